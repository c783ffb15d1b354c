"""CPU: the oracle (oracle/codec_oracle.py) reproduces the vectors the reference itself produced."""
import numpy as np
import pytest
import torch

from oracle import codec_oracle as O
from oracle.procfill import fill_state_dict, proc_input
from oracle.shapes import generator_shapes
from tests.helpers import assert_summary_close, make_inputs

RTOL = 1e-4


@pytest.fixture(scope="module")
def state():
    return O.fresh_state()


def test_generator_decoder_discriminator_forward(golden, state):
    g_sd, d_sd, dec_sd = state
    q_src, zc, zs, noise, q_real_d, _ = make_inputs(2, 16, "small")
    q_hat = O.generator_forward(g_sd, q_src, zc, zs, noise)
    np.testing.assert_allclose(q_hat.numpy(), golden["fwd/q_hat"], rtol=RTOL, atol=1e-5)
    for fast in (True, False):
        acts = []
        x_hat = O.decoder_forward(dec_sd, torch.from_numpy(golden["fwd/q_hat"]), fast_lstm=fast, collect=acts)
        np.testing.assert_allclose(x_hat.numpy(), golden["fwd/x_hat"], rtol=RTOL, atol=2e-5)
        for i, a in enumerate(acts):
            assert_summary_close(a, golden[f"fwd/dec_layer{i}"], RTOL, f"dec layer {i}")
    outs = O.discriminator_forward(d_sd, torch.from_numpy(golden["fwd/x_hat"]))
    for s, (lg, feats) in enumerate(outs):
        np.testing.assert_allclose(lg.numpy(), golden[f"fwd/d_logits{s}"], rtol=RTOL, atol=1e-5)
        for j, f in enumerate(feats):
            assert_summary_close(f, golden[f"fwd/d_feat{s}_{j}"], RTOL, f"feat {s}.{j}")
            np.testing.assert_allclose(f[:, ::7, ::13].numpy(), golden[f"fwd/d_feat{s}_{j}_slice"], rtol=RTOL, atol=1e-5)


def test_generator_variant(golden):
    sd = fill_state_dict({k: torch.empty(s) for k, s in generator_shapes(z_style_dim=160, hidden=192, n_blocks=8).items()}, 11)
    q_src, zc, *_ = make_inputs(2, 16, "small")
    zs160 = torch.nn.functional.normalize(proc_input((2, 160), "small.zs160", 0), dim=-1)
    out = O.generator_forward(sd, q_src, zc, zs160, None, direct_output=True)
    np.testing.assert_allclose(out.numpy(), golden["fwd/q_hat_variant"], rtol=RTOL, atol=1e-5)


def test_losses_and_gradients(golden, state):
    g_sd, d_sd, dec_sd = state
    q_src, zc, zs, noise, q_real_d, _ = make_inputs(2, 16, "small")
    q_hat = torch.from_numpy(golden["fwd/q_hat"]).requires_grad_(True)
    x_hat_from_q = O.decoder_forward(dec_sd, q_hat)
    # losses/gradients are evaluated AT the reference's own x_hat (the log-magnitude term is
    # ill-conditioned w.r.t. 1e-6 input differences), so this pins the loss arithmetic itself.
    x_hat = torch.from_numpy(golden["fwd/x_hat"]).requires_grad_(True)
    with torch.no_grad():
        x_real = O.decoder_forward(dec_sd, q_real_d)
        x_src = O.decoder_forward(dec_sd, q_src)
    outs = O.discriminator_forward(d_sd, x_hat)
    outs_r = O.discriminator_forward(d_sd, x_real)
    vals = {
        "hinge_d": O.multiscale_hinge_d_loss(outs_r, outs), "hinge_g": O.multiscale_hinge_g_loss(outs),
        "fm": O.multiscale_feature_matching_loss(outs_r, outs),
        "hinge_d_single": O.hinge_d_loss(outs_r[0][0], outs[0][0]), "hinge_g_single": O.hinge_g_loss(outs[0][0]),
        "fm_single": O.feature_matching_loss(outs_r[0][1], outs[0][1]),
        "mrstft": O.waveform_mrstft_loss(x_hat, x_src),
    }
    for k, v in vals.items():
        assert float(v) == pytest.approx(float(golden[f"loss/{k}"]), rel=RTOL), k
    for name in ("hinge_g", "fm", "mrstft"):
        gx, = torch.autograd.grad(vals[name], x_hat, retain_graph=True)
        want = golden[f"grad/{name}_wrt_x_hat"]
        np.testing.assert_allclose(gx.numpy(), want, rtol=1e-3, atol=1e-4 * np.abs(want).max())
    gq, = torch.autograd.grad(x_hat_from_q.square().mean(), q_hat)
    want = golden["grad/dec_sqmean_wrt_q_hat"]
    np.testing.assert_allclose(gq.numpy(), want, rtol=1e-3, atol=1e-4 * np.abs(want).max())


def test_mrstft_cases(golden):
    x = proc_input((2, 1, 120000), "stft.x", 0, 0.5)
    y = proc_input((2, 1, 120000), "stft.y", 0, 0.5)
    assert float(O.waveform_mrstft_loss(x, y)) == pytest.approx(float(golden["mrstft/full"]), rel=RTOL)
    xs = x[:, :, :1500].clone().requires_grad_(True)
    l = O.waveform_mrstft_loss(xs, y[:, :, :1500])
    assert float(l) == pytest.approx(float(golden["mrstft/short1500"]), rel=RTOL)
    l.backward()
    want = golden["mrstft/short1500_grad"]
    np.testing.assert_allclose(xs.grad.numpy(), want, rtol=1e-3, atol=1e-4 * np.abs(want).max())
    assert float(O.waveform_mrstft_loss(x[:, :, :300], y[:, :, :300])) == pytest.approx(float(golden["mrstft/fallback_l1"]), rel=RTOL)
    z = torch.zeros(1, 1, 4096); z[0, 0, 100] = 1e-3
    assert float(O.waveform_mrstft_loss(z, x[:1, :, :4096])) == pytest.approx(float(golden["mrstft/silence"]), rel=1e-3)
    res5 = ((256, 64, 256), (512, 128, 512), (1024, 256, 1024), (2048, 512, 2048), (4096, 1024, 4096), (8192, 2048, 8192))
    assert float(O.waveform_mrstft_loss(x[:, :, :30000], y[:, :, :30000], res5)) == pytest.approx(float(golden["mrstft/stress"]), rel=RTOL)
    for n_fft, hop, win in O.DEFAULT_RESOLUTIONS:
        sc, lm = O.spectral_terms(x[:, 0, :8192], y[:, 0, :8192], n_fft, hop, win)
        assert float(sc) == pytest.approx(float(golden[f"mrstft/sc_{n_fft}"]), rel=RTOL)
        assert float(lm) == pytest.approx(float(golden[f"mrstft/lm_{n_fft}"]), rel=RTOL)
        assert_summary_close(O.stft_mag(x[:, 0, :8192], n_fft, hop, win).clamp_min(1e-8), golden[f"mrstft/mag_{n_fft}"], RTOL, "mag")
    with pytest.raises(ValueError):
        O.waveform_mrstft_loss(x, y[:, :, :100])


def test_d_only_and_odd_lengths(golden, state):
    _, d_sd, _ = state
    dp = {k: v.clone().requires_grad_(True) for k, v in d_sd.items()}
    wav = proc_input((2, 1, 16384), "donly.wav", 0, 0.8).requires_grad_(True)
    outs = O.discriminator_forward(dp, wav)
    tot = torch.tensor(0.0)
    for s, (lg, feats) in enumerate(outs):
        np.testing.assert_allclose(lg.detach().numpy(), golden[f"donly/logits{s}"], rtol=RTOL, atol=1e-5)
        for j, f in enumerate(feats):
            assert_summary_close(f, golden[f"donly/feat{s}_{j}"], RTOL, f"donly feat {s}.{j}")
        tot = tot + lg.mean() + sum(f.square().mean() for f in feats)
    tot.backward()
    want = golden["donly/grad_wav"]
    np.testing.assert_allclose(wav.grad[:, :, ::17].numpy(), want, rtol=1e-3, atol=1e-4 * np.abs(want).max())
    for k, p in dp.items():
        assert_summary_close(p.grad, golden[f"donly/grads/{k}"], 2e-4, k)
    outs2 = O.discriminator_forward(d_sd, proc_input((1, 1, 3001), "donly.wav2", 0, 0.8))
    for s, (lg, feats) in enumerate(outs2):
        np.testing.assert_allclose(lg.numpy(), golden[f"donly_odd/logits{s}"], rtol=RTOL, atol=1e-5)
        assert [list(f.shape) for f in feats] == golden[f"donly_odd/feat_shapes{s}"].tolist()


def _check_iteration(golden, prefix, out, g_sd, d_sd, rel=2e-4, check_grads=True):
    for k in ("loss_g", "loss_adv", "loss_fm", "loss_lat_l1", "loss_lat_cont", "loss_mrstft", "loss_delta_budget",
              "take_d_step", "do_r1", "r1", "loss_d", "loss_d_hinge", "d_grad_norm", "g_grad_norm"):
        gk = f"{prefix}/{k}"
        if gk in golden:
            assert k in out, f"{gk} missing from oracle output"
            # the G-step gradient runs through the MR-STFT log-magnitude term, whose gradient moves by
            # ~0.5 % under 1e-6 input noise (measured: thread count alone changes the reference-op result
            # by 0.15 %), so the G gradient norm gets a 1e-2 bound; everything else 2e-4.
            assert out[k] == pytest.approx(float(golden[gk]), rel=(max(1e-2, rel) if k == "g_grad_norm" else rel), abs=1e-7), gk
        else:
            assert k not in out or k in ("take_d_step",), f"oracle produced {k} but reference did not"
    for kind, sd in (("g", g_sd), ("d", d_sd)):
        if check_grads and f"{kind}_grads" in out:
            for k, g in out[f"{kind}_grads"].items():
                assert_summary_close(g, golden[f"{prefix}/{kind}_grads/{k}"], 3e-2 if kind == "g" else 5e-3,  # hinge active-set flips
                                     f"{prefix} {kind}_grad {k}")
        for k, p in sd.items():
            # first AdamW steps move every element by ~lr*sign(g): a sign flip of a near-zero gradient
            # shifts one element by 2*lr, hence the absolute element bound of 3*lr*n_steps.
            assert_summary_close(p, golden[f"{prefix}/{kind}_params/{k}"], RTOL if rel <= 2e-4 else 1e-3, f"{prefix} {kind}_param {k}",
                                 elem_atol=2.5e-3, sum_atol=2e-3)


def test_iteration_period1(golden):
    g_sd, d_sd, dec_sd = O.fresh_state()
    cfg = O.StepCfg(weights=O.Weights(content=0.0, style=0.0))
    og, od = O.AdamWState(), O.AdamWState()
    out = O.iteration(g_sd, d_sd, dec_sd, og, od, make_inputs(2, 16, "p1"), 1, cfg, record=True)
    np.testing.assert_allclose(out["q_hat"].numpy(), golden["p1/q_hat"], rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(out["x_hat"].numpy(), golden["p1/x_hat"], rtol=RTOL, atol=2e-5)
    _check_iteration(golden, "p1", out, g_sd, d_sd)


def test_iteration_sequence_period2_r1(golden):
    """4 iterations, d_step_period=2, r1_interval=2 => D steps (with R1) at global_step 2 and 4."""
    g_sd, d_sd, dec_sd = O.fresh_state()
    cfg = O.StepCfg(d_step_period=2, r1_gamma=1.0, r1_interval=2,
                    weights=O.Weights(adv=0.55, latent_l1=0.2, latent_continuity=1.0, content=0.0, style=0.0,
                                      mrstft=0.0, feature_match=1.0, delta_budget=1.0))
    og, od = O.AdamWState(), O.AdamWState()
    for it in range(1, 5):
        out = O.iteration(g_sd, d_sd, dec_sd, og, od, make_inputs(2, 16, f"seq{it}"), it, cfg, record=True)
        assert out["take_d_step"] == float(it % 2 == 0)
        # Adam's first steps move every element by ~lr*sign(g); near-zero gradient elements flip sign under
        # fp32 noise, so trajectories drift apart geometrically: 2e-4 holds through it3, it4 gets 2e-3.
        _check_iteration(golden, f"seq/it{it}", out, g_sd, d_sd, rel=2e-4 if it < 4 else 2e-3, check_grads=it < 4)


def test_iteration_full_shape_gradients(fullgrad_golden):
    """B=2, T=375 (S=120000): the oracle's D-step and G-step gradients, clip norms, losses and post-AdamW parameters against
    what the reference produced at the full codec shape (tests/golden/make_golden_fullgrad.py; MR-STFT weight 0)."""
    gold = fullgrad_golden
    g_sd, d_sd, dec_sd = O.fresh_state()
    cfg = O.StepCfg(weights=O.Weights(content=0.0, style=0.0, mrstft=0.0))
    out = O.iteration(g_sd, d_sd, dec_sd, O.AdamWState(), O.AdamWState(), make_inputs(2, 375, "fullgrad"), 1, cfg, record=True)
    assert_summary_close(out["q_hat"], gold["fullgrad/q_hat"], RTOL, "q_hat")
    assert_summary_close(out["x_hat"], gold["fullgrad/x_hat"], RTOL, "x_hat")
    _check_iteration(gold, "fullgrad", out, g_sd, d_sd)
    for kind, sd in (("d", d_sd), ("g", g_sd)):
        flat = torch.cat([out[f"{kind}_grads"][k].reshape(-1) for k in sd])  # state-dict order = named_parameters order
        sub = flat[::int(gold[f"fullgrad/stride_{kind}"])].double().numpy()
        want = gold[f"fullgrad/{kind}_grad_sub"].astype(np.float64)
        assert sub.shape == want.shape
        rel = np.linalg.norm(sub - want) / np.linalg.norm(want)
        assert rel < 2e-3, (kind, rel)


def test_bank_helpers(golden):
    zs = proc_input((10, 128), "bank.zs", 0).numpy()
    gi = np.array([0, 1, 2, 3, 0, 1, 2, 0, 1, 0], dtype=np.int64)
    np.testing.assert_allclose(O.build_style_centroid_bank(zs, gi, 5).numpy(), golden["bank/centroid"], rtol=1e-6)
    src = torch.tensor([0, 1, 2, 3, 3, 2, 1, 0])
    assert O.shift_targets_balanced(src, 4, offset=3).tolist() == golden["bank/shift_balanced"].tolist()


def test_mrstft_gradient_is_ill_conditioned_in_fp32(state):
    """Why no bf16 bound is stated for the G gradient WITH the MR-STFT term: the fp32 oracle's own G
    gradient of a full iteration loses its direction under a 1e-4 relative perturbation of q_src (cosine < 0.95 with the
    unperturbed gradient; measured 0.88, and 0.45 at the bf16 rounding level 2e-3), while the same perturbation leaves the
    gradient of the remaining terms intact (cosine > 0.999).  log|X| on near-silent bins has gradient 1/|X| with a phase-
    dependent sign; this bounds every implementation, not one precision."""
    g_sd, d_sd, dec_sd = state
    batch = list(make_inputs(2, 32, "bf16"))
    torch.manual_seed(0)
    pert = [t.clone() for t in batch]
    pert[0] = pert[0] * (1 + 1e-4 * torch.randn_like(pert[0]))

    def g_grad(b, mrstft):
        cfg = O.StepCfg(weights=O.Weights(content=0.0, style=0.0, mrstft=mrstft), d_step_period=2)  # step 1: G step only
        out = O.iteration({k: v.clone() for k, v in g_sd.items()}, {k: v.clone() for k, v in d_sd.items()}, dec_sd,
                          O.AdamWState(), O.AdamWState(), b, 1, cfg, record=True)
        return torch.cat([out["g_grads"][k].double().reshape(-1) for k in sorted(out["g_grads"])])

    cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()))
    assert cos(g_grad(batch, 2.0), g_grad(pert, 2.0)) < 0.95
    assert cos(g_grad(batch, 0.0), g_grad(pert, 0.0)) > 0.999


def test_mel_mrstft_sibling_matches_reference_golden():
    """oracle.mel_mrstft_loss vs the reference's multi_resolution_stft_loss (lab3_losses.py:16-71; tests/golden/mel_golden.npz)."""
    from pathlib import Path
    with np.load(Path(__file__).resolve().parent / "golden" / "mel_golden.npz") as z:
        gold = {k: z[k] for k in z.files}
    x = proc_input((2, 96, 256), "mel.pred", 0, 0.8).requires_grad_(True)
    y = proc_input((2, 96, 256), "mel.true", 0, 0.8)
    l = O.mel_mrstft_loss(x, y)
    assert float(l.detach()) == pytest.approx(float(gold["full"]), rel=1e-5)
    g, = torch.autograd.grad(l, x)
    np.testing.assert_allclose(g.numpy()[:, ::7, ::5], gold["full_grad"], rtol=1e-3, atol=1e-9)
    assert float(O.mel_mrstft_loss(x[:, :, :100], y[:, :, :100]).detach()) == pytest.approx(float(gold["short100"]), rel=1e-5)
    assert float(O.mel_mrstft_loss(x[:, :, :40], y[:, :, :40]).detach()) == pytest.approx(float(gold["fallback40"]), rel=1e-5)
    z = torch.zeros(1, 4, 256)
    z[0, 1, 30] = 1e-4
    assert float(O.mel_mrstft_loss(z, y[:1, :4])) == pytest.approx(float(gold["silence"]), rel=1e-5)
    with pytest.raises(ValueError):
        O.mel_mrstft_loss(x, y[:, :, :100])
