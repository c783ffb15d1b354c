"""GPU parity on the configurations round 1 left to the CPU oracle only:

  * the full codec shape (B=1, T=375, S=120000) against the reference's own `full/*` fixtures, fp32 mode, rtol 1e-4;
  * BASELINE config 4: 160-D style vector, 192 channels x 8 blocks, 2-scale D, style dropout + jitter with the mask and the
    jitter the reference drew (tests/golden/variant_golden.npz, made by make_golden_variant.py from the reference);
  * bf16 mode: a stated bound on the GRADIENTS of one full iteration against the fp32 oracle.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from pathlib import Path

from oracle import codec_oracle as O
from oracle.procfill import fill_state_dict
from oracle.shapes import decoder_shapes, discriminator_shapes, generator_shapes
from tests.gpu_common import build_gpu_models
from tests.helpers import assert_summary_close, make_inputs, tsummary

DEV = "cuda"
GOLD = Path(__file__).resolve().parent / "golden"


def test_full_codec_shape_matches_reference_fixtures(golden):
    """T=375 / S=120000 through G, the SEANet decoder, the 3-scale D and the losses, against what the reference produced."""
    from gan_amd import lab3_codec_models as M
    from gan_amd.lab3_codec_train import CodecStageTrainConfig, _waveform_mrstft_loss
    G, D, codec = build_gpu_models(*O.fresh_state(), "fp32")
    q_src, zc, zs, noise, q_real_d, _ = [t.to(DEV) for t in make_inputs(1, 375, "full")]
    with torch.no_grad():
        q_hat = G(q_src, zc, zs, noise)
        assert_summary_close(q_hat.contiguous(), golden["full/q_hat"], 1e-4, "q_hat")
        x_hat = codec.decode_embeddings(q_hat)
        assert x_hat.shape == (1, 1, 120000)
        assert_summary_close(x_hat.contiguous(), golden["full/x_hat"], 1e-4, "x_hat")
        np.testing.assert_allclose(x_hat[:, :, ::997].cpu().numpy(), golden["full/x_hat_slice"], rtol=1e-4, atol=3e-5)
        x_real = codec.decode_embeddings(q_real_d)
        outs, outs_r = D(x_hat), D(x_real)
        for s, (lg, feats) in enumerate(outs):
            assert lg.shape[-1] == (469, 235, 118)[s]
            np.testing.assert_allclose(lg.cpu().numpy(), golden[f"full/d_logits{s}"], rtol=1e-4, atol=2e-5)
            for j, f in enumerate(feats):
                assert_summary_close(f.contiguous(), golden[f"full/d_feat{s}_{j}"], 1e-4, f"feat {s}.{j}")
        assert float(M.multiscale_hinge_d_loss(outs_r, outs)) == pytest.approx(float(golden["full/hinge_d"]), rel=1e-4)
        assert float(M.multiscale_feature_matching_loss(outs_r, outs)) == pytest.approx(float(golden["full/fm"]), rel=1e-4)
        res = CodecStageTrainConfig("s", 1).wave_mrstft_resolutions
        assert float(_waveform_mrstft_loss(x_hat, x_real, res)) == pytest.approx(float(golden["full/mrstft"]), rel=1e-4)


@pytest.fixture(scope="module")
def v4():
    with np.load(GOLD / "variant_golden.npz") as z:
        return {k: z[k] for k in z.files}


def _v4_models(precision):
    mk = lambda shapes, seed: fill_state_dict({k: torch.empty(s) for k, s in shapes.items()}, seed)
    g_sd = mk(generator_shapes(z_style_dim=160, hidden=192, n_blocks=8), 21)
    d_sd = mk(discriminator_shapes(2), 22)
    dec_sd = mk(decoder_shapes(), 3)
    return build_gpu_models(g_sd, d_sd, dec_sd, precision, z_style_dim=160, hidden_channels=192, n_blocks=8)


def _draws(B, Ds, p, std, seed):
    """What CodecLatentTranslator.forward draws on the device after torch.manual_seed(seed) (rand [B,1], then randn [B,Ds])."""
    torch.manual_seed(seed)
    keep = (torch.rand((B, 1), device=DEV) >= p).float()
    jit = torch.randn((B, Ds), device=DEV) * std
    return keep, jit


def test_config4_generator_d2_and_gradients_match_reference(v4):
    """The reference's q_hat / x_hat / 2-scale D outputs / losses / parameter-gradient fingerprints for the config-4 model.
    The GPU forward draws its own mask and jitter on the device, so the fixture's draws are injected by composing them into
    the style vector (z*keep + jitter is what forward computes, lab3_codec_models.py:104-108) and calling with p = std = 0;
    the device-side draw path itself is checked against the same formula in the next test."""
    from gan_amd import lab3_codec_models as M
    G, D, codec = _v4_models("fp32")
    q_src, zc, zs160, noise, q_real_d, q_real_g = [t.to(DEV) for t in make_inputs(2, 16, "v4", z_style_dim=160)]
    keep, jit = torch.from_numpy(v4["v4/style_keep"]).to(DEV), torch.from_numpy(v4["v4/style_jitter"]).to(DEV)
    assert keep.flatten().tolist() == [0.0, 1.0]  # one sample went through the dropout branch
    q_hat = G(q_src, zc, zs160 * keep + jit, noise)
    np.testing.assert_allclose(q_hat.detach().cpu().numpy(), v4["v4/q_hat"], rtol=1e-4, atol=1e-5)
    x_hat = codec.decode_embeddings(q_hat)
    np.testing.assert_allclose(x_hat.detach().cpu().numpy(), v4["v4/x_hat"], rtol=1e-4, atol=3e-5)
    with torch.no_grad():
        x_real, x_real_g = codec.decode_embeddings(q_real_d), codec.decode_embeddings(q_real_g)
    # D step
    real_outs, fake_outs_d = D(x_real), D(x_hat.detach())
    assert len(fake_outs_d) == 2
    for s, (lg, feats) in enumerate(fake_outs_d):
        np.testing.assert_allclose(lg.detach().cpu().numpy(), v4[f"v4/d_logits{s}"], rtol=1e-4, atol=1e-5)
        for j, f in enumerate(feats):
            assert_summary_close(f.contiguous(), v4[f"v4/d_feat{s}_{j}"], 1e-4, f"feat {s}.{j}")
    loss_d = M.multiscale_hinge_d_loss(real_outs, fake_outs_d)
    assert float(loss_d) == pytest.approx(float(v4["v4/loss_d"]), rel=1e-4)
    dnames = [k for k, _ in D.named_parameters()]
    dgrads = torch.autograd.grad(loss_d, list(D.parameters()))
    for k, g in zip(dnames, dgrads):
        assert_summary_close(g.contiguous(), v4[f"v4/d_grads/{k}"], 2e-3, f"dD/{k}", sum_atol=1e-7)
    # G step (D frozen, as the reference's discarded D gradients allow)
    for p in D.parameters():
        p.requires_grad_(False)
    real_outs_g, fake_outs = D(x_real_g), D(x_hat)
    loss_adv = M.multiscale_hinge_g_loss(fake_outs)
    loss_fm = M.multiscale_feature_matching_loss(real_outs_g, fake_outs)
    loss_l1 = (q_hat - q_src).abs().mean()
    loss_cont = ((q_hat[:, :, 1:] - q_hat[:, :, :-1]) - (q_src[:, :, 1:] - q_src[:, :, :-1])).abs().mean()
    w = M.CodecTrainWeights()
    loss_g = w.adv * loss_adv + w.feature_match * loss_fm + w.latent_l1 * loss_l1 + w.latent_continuity * loss_cont
    for k, v in (("loss_adv", loss_adv), ("loss_fm", loss_fm), ("loss_lat_l1", loss_l1), ("loss_lat_cont", loss_cont), ("loss_g", loss_g)):
        assert float(v) == pytest.approx(float(v4[f"v4/{k}"]), rel=1e-4), k
    gnames = [k for k, _ in G.named_parameters()]
    ggrads = torch.autograd.grad(loss_g, list(G.parameters()))
    flat = torch.sqrt(sum(g.double().pow(2).sum() for g in ggrads))
    assert float(flat) == pytest.approx(float(v4["v4/g_grad_flat_norm"]), rel=1e-3)
    for k, g in zip(gnames, ggrads):
        assert_summary_close(g.contiguous(), v4[f"v4/g_grads/{k}"], 3e-3, f"dG/{k}", sum_atol=1e-7)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_config4_device_dropout_and_jitter_draws(precision):
    """style_dropout_p / style_jitter_std inside forward: with the device generator seeded, forward(z, p, std) equals
    forward(z*keep + jitter) for the keep / jitter the same seed yields - the reference's order of draws (:104-108)."""
    G, D, codec = _v4_models(precision)
    q_src, zc, zs160, noise, *_ = [t.to(DEV) for t in make_inputs(2, 16, "v4", z_style_dim=160)]
    for seed in (3, 4, 5, 6):  # several masks
        keep, jit = _draws(2, 160, 0.25, 0.08, seed)
        torch.manual_seed(seed)
        with torch.no_grad():
            got = G(q_src, zc, zs160, noise, style_dropout_p=0.25, style_jitter_std=0.08)
            want = G(q_src, zc, zs160 * keep + jit, noise)
        assert torch.equal(got, want), seed


def test_bf16_iteration_gradients_track_fp32_oracle():
    """The stated bf16 bound on gradients: one full iteration (B=2, T=32) in bf16 mode against the fp32
    CPU oracle - flat D and G gradient vectors: cosine >= 0.99, relative L2 error <= 0.15, both clip norms within 5 %.
    The MR-STFT weight is 0 here: that term's gradient is ill-conditioned in ANY precision (the fp32 oracle's own G gradient
    keeps a cosine of only 0.88 / 0.45 with itself when q_src is perturbed by 1e-4 / 2e-3 relative,
    tests/test_oracle_golden.py::test_mrstft_gradient_is_ill_conditioned_in_fp32), so no bf16 run can track it; its bf16
    arithmetic is bounded at the operator level instead (tests/test_gpu_ops.py, MR-STFT hi/lo split against the fp32 path)."""
    from gan_amd.lab3_codec_train import codec_train_iteration
    from gan_amd.optim import FusedAdamW
    from tests.gpu_common import stage_cfg_from_oracle
    g_sd, d_sd, dec_sd = O.fresh_state()
    batch = make_inputs(2, 32, "bf16")
    cfg = O.StepCfg(weights=O.Weights(content=0.0, style=0.0, mrstft=0.0))
    want = O.iteration({k: v.clone() for k, v in g_sd.items()}, {k: v.clone() for k, v in d_sd.items()}, dec_sd,
                       O.AdamWState(), O.AdamWState(), batch, 1, cfg, record=True)
    G, D, codec = build_gpu_models(g_sd, d_sd, dec_sd, "bf16")
    scfg = stage_cfg_from_oracle(cfg)
    opt_g = FusedAdamW(G.parameters(), lr=scfg.lr_g, betas=scfg.betas)
    opt_d = FusedAdamW(D.parameters(), lr=scfg.lr_d, betas=scfg.betas)
    grabbed = {}
    for name, opt in (("g", opt_g), ("d", opt_d)):  # the flat gradient buffer as the optimizer sees it, before its update
        orig = opt.step
        def step(grads_gathered=False, _o=orig, _opt=opt, _n=name):
            grabbed[_n] = _opt.flat_g.detach().clone()
            return _o(grads_gathered=grads_gathered)
        opt.step = step
    q_src, zc, zs, noise, q_real_d, q_real_g = [t.to(DEV) for t in batch]
    out = codec_train_iteration(scfg, G, D, codec, opt_g, opt_d, q_src, zc, zs, noise, q_real_d, q_real_g, 1)
    for name, mod, opt, key in (("d", D, opt_d, "d_grads"), ("g", G, opt_g, "g_grads")):
        flat = grabbed[name].double().cpu()
        ref = torch.zeros_like(flat)
        names = [k for k, p in mod.named_parameters() if p.requires_grad]
        assert len(names) == len(opt.params)
        for k, p, off in zip(names, opt.params, opt.offsets):
            ref[off:off + p.numel()] = want[key][k].double().reshape(-1)
        cos = float(torch.dot(flat, ref) / (flat.norm() * ref.norm()))
        rel = float((flat - ref).norm() / ref.norm())
        print(f"bf16 gradient vs fp32 oracle [{name}]: cosine {cos:.5f}, relative L2 {rel:.4f}")  # measured: d 0.9946 / 0.108, g 0.9936 / 0.113
        assert cos >= 0.99, (name, cos)
        assert rel <= 0.15, (name, rel)
    assert float(out["d_grad_norm"]) == pytest.approx(want["d_grad_norm"], rel=0.05)
    assert float(out["g_grad_norm"]) == pytest.approx(want["g_grad_norm"], rel=0.05)
