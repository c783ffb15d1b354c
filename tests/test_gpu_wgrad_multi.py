"""GPU: the grouped weight-gradient entry point (ga_wgrad_multi, one launch for many small layers) against one ga_wgrad per layer,
and the deferred path that uses it (ops.WeightGradSink.defer: the generator's layers hand their weight gradients to one grouped
launch at the end of the backward pass) against the per-layer path, through a whole training iteration."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from gan_amd import _lib as L
from gan_amd import lab3_codec_train as T
from gan_amd import ops
from gan_amd.lab3_codec_bridge import FrozenEncodec
from gan_amd.lab3_codec_models import CodecLatentTranslator, CodecTrainWeights, MultiScaleWaveDiscriminator
from gan_amd.ops import ConvGeom
from gan_amd.optim import FusedAdamW

DEV = torch.device("cuda")


def _items(cases, bias_flags):
    torch.manual_seed(31)
    n = len(cases)
    items = (L.WgradParams * n)()
    keep, outs = [], []
    for it, (g, B, Ln), wb in zip(items, cases, bias_flags):
        x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
        lo = g.out_len(Ln)
        ga = torch.randn(B, lo, g.cout, device=DEV).bfloat16()
        dw = torch.zeros(g.cout, g.k, g.cin, device=DEV)
        db = torch.zeros(g.cout, device=DEV) if wb else None
        ops._fill(it, g=ga.data_ptr(), x=x.data_ptr(), dw=dw.data_ptr(), dbias=L.ptr(db), B=B, Lin=Ln, Cin=g.cin, M=g.cout, Lout=lo,
                  J=g.k, in_stride=g.stride, tap_step=g.dilation, off0=-g.padding, LgTot=lo, g_stride=1, g_phase=0)
        keep.append((x, ga))
        outs.append((dw, db))
    return items, keep, outs


def test_grouped_weight_gradient_equals_one_launch_per_layer():
    """Heterogeneous problems in one group (more than one chunk of 20): generator-sized dilated layers, a strided k41 layer, one
    without a bias gradient, 128 / 192 / 256 output channels.  Same kernel body, other reduction slices: equal to fp32 atomics order."""
    cases = [(ConvGeom("conv", 256, 256, 3, 1, 2 ** (i % 4), 2 ** (i % 4)), 4, 375) for i in range(19)]
    cases += [(ConvGeom("conv", 128, 256, 3, 1, 1, 1), 4, 375), (ConvGeom("conv", 256, 128, 3, 1, 1, 1), 3, 375),
              (ConvGeom("conv", 64, 128, 41, 4, 20, 1), 2, 3000), (ConvGeom("conv", 192, 192, 5, 1, 4, 2), 5, 211)]
    flags = [i != 7 for i in range(len(cases))]
    items, keep, outs = _items(cases, flags)
    L.check(L.lib().ga_wgrad_multi(C.cast(items, C.c_void_p), len(cases), L.DT_BF16, L.stream()), "ga_wgrad_multi")
    for (g, B, Ln), (x, ga), (dw, db), wb in zip(cases, keep, outs, flags):
        want = ops.conv_wgrad(g, ga, x, want_bias=wb)
        wdw, wdb = want if wb else (want, None)
        assert float(wdw.abs().max()) > 0
        assert float((dw - wdw).abs().max()) <= 2e-6 * float(wdw.abs().max()), g
        if wb:
            assert float((db - wdb).abs().max()) <= 2e-6 * float(wdb.abs().max()), g


def test_grouped_weight_gradient_refuses_what_it_does_not_take():
    """A 64-output-channel layer (the 64 x 256-tile kernel's shape) in the group: GA_ERR_UNSUPPORTED and nothing is written."""
    cases = [(ConvGeom("conv", 256, 256, 3, 1, 1, 1), 2, 375), (ConvGeom("conv", 32, 64, 41, 4, 20, 1), 2, 3000)]
    items, keep, outs = _items(cases, [True, True])
    assert L.lib().ga_wgrad_multi(C.cast(items, C.c_void_p), 2, L.DT_BF16, L.stream()) == 4
    assert L.lib().ga_wgrad_multi(C.cast(items, C.c_void_p), 2, L.DT_F32, L.stream()) == 4
    torch.cuda.synchronize()
    for dw, db in outs:
        assert float(dw.abs().max()) == 0.0 and float(db.abs().max()) == 0.0
    assert L.lib().ga_wgrad_multi(C.cast(items, C.c_void_p), 0, L.DT_BF16, L.stream()) == 0


def test_deferred_generator_weight_gradients_equal_the_per_layer_path():
    """One bf16 G step with the generator's weight gradients deferred to the grouped launch and with one launch per layer: the same
    flat gradient buffer (to fp32 atomics order) and the same loss.  No D step in front (its AdamW update would turn summation
    noise into +-lr differences of D's weights) and the MR-STFT weight at 0: with that term two runs of the SAME path differ by
    7e-3 of the largest gradient entry (its gradient amplifies the atomics-order noise of the spectral sums: measured 4.5e-4 on
    0.068 plain vs plain, 1.9e-9 on 0.020 without the term, the same for deferred vs plain)."""
    torch.manual_seed(45)
    G0 = CodecLatentTranslator(hidden_channels=128, n_blocks=3, precision="bf16").to(DEV).train()
    D0 = MultiScaleWaveDiscriminator(2, precision="bf16").to(DEV).train()
    codec = FrozenEncodec(random_init=True, device="cuda", precision="bf16")
    B, Tn = 3, 96  # (B * Tn >= 256 rows: below that the grouped entry point refuses and the sink launches layer by layer)
    q = 0.5 * torch.randn(B, 128, Tn, device=DEV)
    zc, zs, nz = torch.randn(B, 128, device=DEV), torch.randn(B, 128, device=DEV), torch.randn(B, 32, device=DEV)
    qr, qg = 0.5 * torch.randn(B, 128, Tn, device=DEV), 0.5 * torch.randn(B, 128, Tn, device=DEV)
    cfg = T.CodecStageTrainConfig(stage_name="stage3", epochs=1, d_step_period=2,
                                  weights=CodecTrainWeights(content=0.0, style=0.0, mrstft=0.0))
    res, deferred_layers = [], []
    real_run = ops.WeightGradSink._run_deferred

    def counting_run(self):
        deferred_layers.append(len(self.deferred))
        return real_run(self)

    for defer in (True, False):
        G = CodecLatentTranslator(hidden_channels=128, n_blocks=3, precision="bf16").to(DEV).train()
        D = MultiScaleWaveDiscriminator(2, precision="bf16").to(DEV).train()
        G.load_state_dict(G0.state_dict())
        D.load_state_dict(D0.state_dict())
        og, od = FusedAdamW(G.parameters(), lr=1e-4, betas=cfg.betas), FusedAdamW(D.parameters(), lr=1e-4, betas=cfg.betas)
        ops.WGRAD_DEFER = defer
        ops.WeightGradSink._run_deferred = counting_run
        try:
            out = T.codec_train_iteration(cfg, G, D, codec, og, od, q, zc, zs, nz, qr, qg, global_step=1)
        finally:
            ops.WGRAD_DEFER = True
            ops.WeightGradSink._run_deferred = real_run
        res.append((og.flat_g.clone(), float(out["loss_g"])))
    a, b = res
    assert ops.LAST_WGRAD_GROUP == (2 + 2 * 3, 0), ops.LAST_WGRAD_GROUP
    assert max(deferred_layers) == 2 + 2 * 3, deferred_layers  # conv_in, conv_out and two convs per block went through the group
    assert a[1] == pytest.approx(b[1], rel=1e-5)
    assert float(b[0].abs().max()) > 0
    assert float((a[0] - b[0]).abs().max()) <= 2e-5 * float(b[0].abs().max())
