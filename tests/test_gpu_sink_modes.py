"""GPU: a full iteration with mode seeking (the generator runs twice, so every G weight receives two gradient contributions)
and R1 (gradients already in p.grad before the hinge backward) gives the same updated parameters with the deferred
weight-gradient path (ops.WeightGradSink) as with plain per-layer autograd."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from gan_amd import lab3_codec_train as T
from gan_amd import ops
from gan_amd.lab3_codec_bridge import FrozenEncodec
from gan_amd.lab3_codec_models import CodecLatentTranslator, CodecTrainWeights, MultiScaleWaveDiscriminator
from gan_amd.optim import FusedAdamW

DEV = torch.device("cuda")


def test_sink_equals_plain_autograd_with_mode_seeking_and_r1():
    torch.manual_seed(44)
    G0 = CodecLatentTranslator(hidden_channels=64, n_blocks=2, precision="fp32").to(DEV).train()
    D0 = MultiScaleWaveDiscriminator(2, precision="fp32").to(DEV).train()
    codec = FrozenEncodec(random_init=True, device="cuda", precision="fp32")
    B, Tn = 2, 12
    q = 0.5 * torch.randn(B, 128, Tn, device=DEV)
    zc, zs, nz = torch.randn(B, 128, device=DEV), torch.randn(B, 128, device=DEV), torch.randn(B, 32, device=DEV)
    nz2 = torch.randn(B, 32, device=DEV)
    qr, qg = 0.5 * torch.randn(B, 128, Tn, device=DEV), 0.5 * torch.randn(B, 128, Tn, device=DEV)
    cfg = T.CodecStageTrainConfig(stage_name="stage3", epochs=1, d_step_period=1, r1_gamma=2.0, r1_interval=1,
                                  weights=CodecTrainWeights(content=0.0, style=0.0, mode_seeking=0.7))
    results = []
    for sink in (True, False):
        G = CodecLatentTranslator(hidden_channels=64, n_blocks=2, precision="fp32").to(DEV).train()
        D = MultiScaleWaveDiscriminator(2, precision="fp32").to(DEV).train()
        G.load_state_dict(G0.state_dict())
        D.load_state_dict(D0.state_dict())
        og, od = FusedAdamW(G.parameters(), lr=1e-3, betas=cfg.betas), FusedAdamW(D.parameters(), lr=1e-3, betas=cfg.betas)
        ops.WGRAD_SINK = sink
        try:
            out = T.codec_train_iteration(cfg, G, D, codec, og, od, q, zc, zs, nz, qr, qg, global_step=1, noise2=nz2)
        finally:
            ops.WGRAD_SINK = True
        results.append((og.flat_p.clone(), od.flat_p.clone(), float(out["loss_g"]), float(out["loss_d"]), float(out["loss_ms"]),
                        og.flat_g.clone(), od.flat_g.clone()))
    a, b = results
    # (loss sums and weight gradients are accumulated with fp32 atomics, whose order changes from run to run: the bounds below are
    # that noise with a margin -- at 1e-6 / 2e-6 this test failed once inside a full-suite run and passed alone and on the re-run)
    assert a[2] == pytest.approx(b[2], rel=1e-5) and a[3] == pytest.approx(b[3], rel=1e-5) and a[4] == pytest.approx(b[4], rel=1e-5)
    # the gradient buffers (what the two paths differ in) agree to fp32 summation order ...
    for ga, gb, name in ((a[5], b[5], "G grad"), (a[6], b[6], "D grad")):
        assert float(gb.abs().max()) > 0
        err = float((ga - gb).abs().max() / gb.abs().max())
        assert err <= 1e-5, (name, err)
    # ... and so do the parameters after the step, up to what AdamW's first step makes of that noise.  The first step is
    # lr * g / (|g| + 1e-8), i.e. +-lr for every entry whose gradient is far above 1e-8, whatever its size: two paths whose
    # gradients differ by summation noise land 2 lr apart exactly where the noise decides the SIGN (an entry whose true gradient
    # is ~0, a few 1e-9 here) or where |g| is of the order of eps.  That is the only way the two paths can differ, and the test
    # states it as such (the earlier form bounded EVERY entry at 1e-4 of max|p| and failed twice in some
    # twenty-five full-suite runs): every entry that differs by more than 1e-4 of max|p| must be one whose gradient is
    # noise-sized in both paths, and the message names the offenders with their gradients if one is not.
    for pa, pb, ga, gb, name in ((a[0], b[0], a[5], b[5], "G"), (a[1], b[1], a[6], b[6], "D")):
        diff = (pa - pb).abs()
        assert float(diff.max()) <= 2.1e-3, (name, float(diff.max()))  # 2 lr: the most a first step can disagree by
        gmax = float(gb.abs().max())
        noise = max(1e-5 * gmax, 1e-6)  # the gradient-agreement bound above / the eps regime of the update
        big = diff > 1e-4 * float(pb.abs().max())
        nbig = int(big.sum())
        if nbig:
            idx = big.nonzero().flatten()
            gmag = torch.maximum(ga[idx].abs(), gb[idx].abs())
            worst = idx[gmag.argmax()]
            report = (f"{name}: {nbig} entries differ by more than 1e-4 max|p|; largest gradient among them at flat index "
                      f"{int(worst)}: g_sink {float(ga[worst]):.3e}, g_plain {float(gb[worst]):.3e}, p_sink {float(pa[worst]):.6e}, "
                      f"p_plain {float(pb[worst]):.6e}, max|g| {gmax:.3e}")
            print(report)
            assert float(gmag.max()) <= noise, report
        clear = gb.abs() >= 1e-3 * gmax
        assert int(clear.sum()) > 0.2 * clear.numel()
        err = float(diff[clear].max() / pb.abs().max())
        assert err <= 1e-4, (name, err)
