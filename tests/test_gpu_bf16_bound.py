"""GPU: what the bf16 mode is bound by when the DEFAULT loss is used, MR-STFT term (weight 2.0) included.

The MR-STFT gradient cannot be tracked element-wise in any precision (log|X| on near-silent bins,
tests/test_oracle_golden.py::test_mrstft_gradient_is_ill_conditioned_in_fp32), so two things that matter for training are
bounded instead, and the measured numbers are stated in DESIGN.md section 3:

  (a) descent: with the default CodecTrainWeights the bf16 G gradient has a positive inner product with the fp32 oracle's
      gradient, and the fp32-EVALUATED loss_g (CPU oracle) decreases along the bf16 direction for a small step;
  (b) trajectory: 20 iterations (B=2, T=32, d_step_period=1) in bf16 mode and in fp32 mode on the GPU from the same state:
      every logged loss term within a stated relative band at iterations 5 / 10 / 20.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import codec_oracle as O
from tests.gpu_common import build_gpu_models, flat_in_named_order, grab_flat_grads, run_gpu_iteration, stage_cfg_from_oracle
from tests.helpers import make_inputs

DEV = "cuda"


def test_bf16_generator_gradient_is_a_descent_direction_for_the_default_loss():
    from gan_amd.lab3_codec_train import codec_train_iteration
    from gan_amd.optim import FusedAdamW
    g_sd, d_sd, dec_sd = O.fresh_state()
    batch = make_inputs(2, 32, "bf16")
    cfg = O.StepCfg(weights=O.Weights(content=0.0, style=0.0), d_step_period=2)  # step 1: G step only, default weights otherwise
    assert cfg.weights.mrstft == 2.0

    def oracle_eval(gsd, record=False):
        return O.iteration({k: v.clone() for k, v in gsd.items()}, {k: v.clone() for k, v in d_sd.items()}, dec_sd,
                           O.AdamWState(), O.AdamWState(), batch, 1, cfg, record=record)

    want = oracle_eval(g_sd, record=True)
    g_ref = torch.cat([want["g_grads"][k].double().reshape(-1) for k in g_sd])
    G, D, codec = build_gpu_models(g_sd, d_sd, dec_sd, "bf16")
    scfg = stage_cfg_from_oracle(cfg)
    opt_g = FusedAdamW(G.parameters(), lr=scfg.lr_g, betas=scfg.betas)
    opt_d = FusedAdamW(D.parameters(), lr=scfg.lr_d, betas=scfg.betas)
    grabbed = grab_flat_grads({"g": opt_g})
    codec_train_iteration(scfg, G, D, codec, opt_g, opt_d, *[t.to(DEV) for t in batch], 1)
    g_bf = flat_in_named_order(G, opt_g, grabbed["g"]).double().cpu()
    assert g_bf.shape == g_ref.shape
    cos = float(torch.dot(g_bf, g_ref) / (g_bf.norm() * g_ref.norm()))
    print(f"bf16 G gradient vs fp32 oracle, default loss (MR-STFT weight 2.0): cosine {cos:.4f}, "
          f"norms {float(g_bf.norm()):.4f} / {float(g_ref.norm()):.4f}")
    assert cos > 0.1, cos  # measured 0.74: the MR-STFT part is noise-like in every precision, the rest is tracked
    # fp32-evaluated loss along the bf16 direction (unit vector u, step eta in parameter space)
    u = g_bf / g_bf.norm()
    slope = float(torch.dot(u, g_ref))  # first-order decrease per unit step, by the oracle's own gradient
    base = float(want["loss_g"])
    rows = []
    for eta in (3e-4, 1e-3, 3e-3):
        moved, off = {}, 0
        for k, v in g_sd.items():
            moved[k] = (v.double() - eta * u[off:off + v.numel()].reshape(v.shape)).float()
            off += v.numel()
        rows.append((eta, float(oracle_eval(moved)["loss_g"]) - base))
    print("fp32 loss_g change along -g_bf16:", ", ".join(f"eta {e:g}: {d:+.3e} (first order {-e * slope:+.3e})" for e, d in rows))
    assert slope > 0.0
    assert all(d < 0.0 for _, d in rows), rows


def test_bf16_trajectory_tracks_fp32_mode_over_20_iterations():
    """Three 20-iteration runs from the same state on the same batches: fp32 mode, bf16 mode, and - as the yardstick for what a
    GAN trajectory does with ANY perturbation of that size - fp32 mode again with the latent inputs rounded to bf16 (a relative
    2^-9 perturbation of q_src / q_real, nothing else changed).  Stated band: every logged loss term of the
    bf16 run within 25 % (or 0.03 absolute, whichever is larger) of the fp32 run at iterations 5 / 10 / 20 -- loss_adv within 0.10
    absolute; the control's own deviation is printed beside it.  Measured (two boxes): bf16 worst 16 % (loss_adv, a mean of logits near 0.16-0.23), 14 %
    (loss_lat_cont at 0.09), <= 5 % on loss_g / loss_d / loss_fm / loss_mrstft; the control 7 % / 7 % / <= 4 %: the band is the
    trajectory's own sensitivity (AdamW's first steps are +-lr whatever the gradient's size), not an arithmetic error of bf16."""
    g_sd, d_sd, dec_sd = O.fresh_state()
    cfg = O.StepCfg(weights=O.Weights(content=0.0, style=0.0))  # d_step_period 1, default weights incl. MR-STFT 2.0
    keys = ("loss_g", "loss_d", "loss_adv", "loss_fm", "loss_lat_l1", "loss_lat_cont", "loss_mrstft")
    hist = {}
    for run, prec in (("fp32", "fp32"), ("bf16", "bf16"), ("fp32_rounded_inputs", "fp32")):
        G, D, codec = build_gpu_models({k: v.clone() for k, v in g_sd.items()}, {k: v.clone() for k, v in d_sd.items()}, dec_sd, prec)
        og = od = None
        rows = []
        for it in range(1, 21):
            batch = list(make_inputs(2, 32, f"traj{it}"))
            if run == "fp32_rounded_inputs":
                for i in (0, 4, 5):  # q_src, q_real_d, q_real_g
                    batch[i] = batch[i].bfloat16().float()
            out = run_gpu_iteration(G, D, codec, batch, cfg, it, og, od)
            og, od = out["opt_g"], out["opt_d"]
            rows.append({k: float(out[k]) for k in keys})
        hist[run] = rows
    worst = {"bf16": {}, "fp32_rounded_inputs": {}}
    for it in (5, 10, 20):
        a = hist["fp32"][it - 1]
        for run in ("bf16", "fp32_rounded_inputs"):
            b = hist[run][it - 1]
            line = []
            for k in keys:
                rel = abs(b[k] - a[k]) / max(abs(a[k]), 1e-3)
                worst[run][k] = max(worst[run].get(k, 0.0), rel)
                line.append(f"{k} {a[k]:.4f}/{b[k]:.4f} ({rel:.1e})")
            print(f"it {it:2d} fp32/{run}: " + ", ".join(line))
    print("worst relative deviation from the fp32 run over iterations 5/10/20:")
    for k in keys:
        print(f"  {k:14s} bf16 {worst['bf16'][k]:.3f}   fp32 with bf16-rounded inputs {worst['fp32_rounded_inputs'][k]:.3f}")
    for it in (5, 10, 20):
        a, b = hist["fp32"][it - 1], hist["bf16"][it - 1]
        for k in keys:
            # loss_adv is a mean of logits near 0.2 and the most chaotic term: over nine runs the bf16 deviation ranged 5-22 % and the
            # fp32 control's own 10-30 % (both modes sum with atomics, so no two runs agree): 0.10 absolute for it
            band = 0.10 if k == "loss_adv" else max(0.25 * abs(a[k]), 0.03)
            assert abs(b[k] - a[k]) <= band, (it, k, a[k], b[k])
