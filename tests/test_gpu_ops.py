"""GPU: each HIP kernel family, called through the C ABI (gan_amd.ops -> ctypes), against stock torch CPU ops.

fp32 mode (direct kernels): rtol 1e-4 (relative to the tensor's max), the north-star fp32 bound.
bf16 mode (MFMA kernels): inputs/weights rounded to bf16, fp32 accumulate: 2e-2 of the tensor's max.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gan_amd import _lib as L
from gan_amd import ops
from gan_amd.ops import ConvGeom

DEV = "cuda"
TOL = {torch.float32: 1e-4, torch.bfloat16: 2.5e-2}


def close(got, want, tol, name=""):
    got = got.detach().float().cpu().double()
    want = want.detach().double()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = max(want.abs().max().item(), 1e-12)
    err = (got - want).abs().max().item() / scale
    assert err <= tol, f"{name}: max err / max|ref| = {err:.3e} > {tol}"


def _ref_conv(g, x, w, b, pre_act, post_act):
    if pre_act == L.ACT_ELU:
        x = F.elu(x)
    if g.kind == "conv":
        y = F.conv1d(x, w, b, stride=g.stride, padding=g.padding, dilation=g.dilation)
    elif g.kind == "cconv":
        pad = (g.k - 1) * g.dilation
        y = F.conv1d(F.pad(x, (pad, 0), mode="reflect") if pad else x, w, b, dilation=g.dilation)
    else:
        y = F.conv_transpose1d(x, w, b, stride=g.stride)
        y = y[..., : y.shape[-1] - (g.k - g.stride)]
    if post_act == L.ACT_LRELU:
        y = F.leaky_relu(y, 0.2)
    return y


TRAIN_CASES = [
    # (geom, L, weight_norm, post_act)
    (ConvGeom("conv", 128, 256, 3, 1, 1, 1), 75, False, L.ACT_NONE),        # G in_proj
    (ConvGeom("conv", 256, 256, 3, 1, 8, 8), 75, False, L.ACT_NONE),        # G dilated block conv
    (ConvGeom("conv", 1, 32, 15, 1, 7, 1), 1000, True, L.ACT_LRELU),        # D block 0 (Cin = 1)
    (ConvGeom("conv", 32, 64, 41, 4, 20, 1), 1001, True, L.ACT_LRELU),      # D block 1, odd length
    (ConvGeom("conv", 64, 128, 41, 4, 20, 1), 702, True, L.ACT_LRELU),      # D block 2
    (ConvGeom("conv", 128, 256, 41, 4, 20, 1), 375, True, L.ACT_LRELU),     # D block 3
    (ConvGeom("conv", 256, 512, 41, 4, 20, 1), 259, True, L.ACT_LRELU),     # D block 4
    (ConvGeom("conv", 512, 1, 3, 1, 1, 1), 65, True, L.ACT_NONE),           # D logits conv (Cout = 1)
    # long enough for the HBM-bound thin kernels (LDS-staged window) to be the ones that run
    (ConvGeom("conv", 1, 32, 15, 1, 7, 1), 6001, True, L.ACT_LRELU),
    (ConvGeom("conv", 64, 1, 3, 1, 1, 1), 5003, True, L.ACT_NONE),
    (ConvGeom("conv", 512, 1, 3, 1, 1, 1), 4100, True, L.ACT_NONE),
    (ConvGeom("conv", 96, 1, 3, 1, 1, 1), 30011, False, L.ACT_NONE),        # the mel decoder's last conv after the row gather: 12 pieces per row, many rows
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", TRAIN_CASES, ids=lambda c: f"{c[0].cin}-{c[0].cout}-k{c[0].k}s{c[0].stride}d{c[0].dilation}")
def test_trainable_conv_forward_and_all_gradients(case, dtype):
    g, Ln, wn, act = case
    B = 3
    torch.manual_seed(1)
    x = torch.randn(B, g.cin, Ln)
    v = torch.randn(g.cout, g.cin, g.k) / (g.cin * g.k) ** 0.5
    gp = (torch.rand(g.cout, 1, 1) + 0.5) if wn else None
    b = torch.randn(g.cout) * 0.1
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xr, vr, br = x.clone().requires_grad_(True), v.clone().requires_grad_(True), b.clone().requires_grad_(True)
    gr = gp.clone().requires_grad_(True) if wn else None
    w_eff = vr * (gr / vr.reshape(g.cout, -1).norm(dim=1).reshape(-1, 1, 1)) if wn else vr
    if dtype == torch.bfloat16:  # the kernel sees bf16-rounded effective weights
        w_eff = w_eff + (w_eff.detach().bfloat16().float() - w_eff.detach())
    y_ref = _ref_conv(g, xr, w_eff, br, 0, act)
    gy = torch.randn_like(y_ref)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    refs = torch.autograd.grad(y_ref, [xr, vr, br] + ([gr] if wn else []), gy)

    xg = x.permute(0, 2, 1).contiguous().to(DEV, dtype).requires_grad_(True)
    vg, bg = v.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    gg = gp.to(DEV).requires_grad_(True) if wn else None
    wp_f, wp_d, carrier = ops.PackTrainable.apply(vg, gg, g, dtype)
    y = ops.ConvTrain.apply(xg, wp_f, wp_d, carrier, bg, g, act)
    tol = TOL[dtype]
    close(y.permute(0, 2, 1), y_ref, tol, "forward")
    grads = torch.autograd.grad(y, [xg, vg, bg] + ([gg] if wn else []), gy.permute(0, 2, 1).contiguous().to(DEV, dtype))
    close(grads[0].permute(0, 2, 1), refs[0], tol, "dx")
    close(grads[1], refs[1], tol * (4 if dtype == torch.bfloat16 else 1), "dv")
    close(grads[2], refs[2], tol, "dbias")
    if wn:
        close(grads[3], refs[3], tol * (4 if dtype == torch.bfloat16 else 1), "dg")


def test_mfma_path_is_taken_and_matches_direct():
    """bf16: the MFMA kernel must accept the fat layers (no silent direct fallback) and agree with the direct kernel."""
    torch.manual_seed(2)
    for g, Ln in ((ConvGeom("conv", 64, 128, 41, 4, 20, 1), 2050), (ConvGeom("conv", 256, 256, 3, 1, 2, 2), 375),
                  (ConvGeom("convt", 128, 64, 8, 4), 300), (ConvGeom("cconv", 128, 512, 7), 375)):
        B = 2
        x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
        d0, d1, k = g.wdims
        w = (torch.randn(d0, d1, k, device=DEV) / (g.cin * g.k) ** 0.5)
        bias = torch.randn(g.cout, device=DEV)
        for mode, fn in (("fwd", None), ("dgrad", None)):
            pack = g.fwd_pack if mode == "fwd" else g.dgrad_pack
            wp = ops.pack_weight(g, w, pack, torch.bfloat16)
            outs = []
            for impl in (L.IMPL_MFMA, L.IMPL_DIRECT):
                ops.set_conv_impl(impl)
                try:
                    if mode == "fwd":
                        outs.append(ops.conv_forward(g, x, wp, bias, pre_act=L.ACT_ELU if g.kind != "conv" else 0))
                    else:
                        gy = torch.randn(B, g.out_len(Ln), g.cout, device=DEV).bfloat16()
                        torch.manual_seed(3)
                        outs.append(ops.conv_dgrad(g, gy, wp, Ln))
                finally:
                    ops.set_conv_impl(L.IMPL_AUTO)
            if mode == "fwd":
                close(outs[0], outs[1].float().cpu(), 1.5e-2, f"{g} {mode}")


GEMM_CASES = [
    # geom, L, B -- every geometry family the gathered-GEMM kernel serves (pre_act none), ragged N and M tiles
    (ConvGeom("conv", 32, 64, 41, 4, 20, 1), 3001, 3),     # D block 1: 64x256 tile, two taps per K-step
    (ConvGeom("conv", 64, 128, 41, 4, 20, 1), 2050, 2),    # D block 2
    (ConvGeom("conv", 256, 512, 41, 4, 20, 1), 469, 3),    # D block 4: N flattened over the batch, ragged tiles
    (ConvGeom("conv", 256, 256, 3, 1, 8, 8), 375, 3),      # G dilated conv
    (ConvGeom("conv", 128, 256, 3, 1, 1, 1), 37, 5),       # short rows: tiles span several batch items
    (ConvGeom("conv", 16, 48, 5, 2, 2, 1), 301, 2),        # Cin 16 (four taps per K-step), M = 48, K not a multiple of 64
    (ConvGeom("convt", 128, 64, 8, 4), 300, 2),            # transposed conv: phases
    (ConvGeom("convt", 512, 256, 16, 8), 40, 2),
    (ConvGeom("cconv", 128, 512, 7), 375, 2),              # causal reflect conv (dgrad needs the mirrored fix-up launch)
    (ConvGeom("cconv", 64, 128, 1), 500, 2),
    (ConvGeom("convt", 64, 32, 4, 2), 400, 3),             # decoder tail: 32 and 16 output channels (32 x 256 tile)
    (ConvGeom("cconv", 32, 16, 3), 901, 2),
    (ConvGeom("cconv", 16, 32, 1), 901, 2),
]


@pytest.mark.parametrize("case", GEMM_CASES, ids=lambda c: f"{c[0].kind}-{c[0].cin}-{c[0].cout}-k{c[0].k}s{c[0].stride}d{c[0].dilation}-L{c[1]}")
def test_gathered_gemm_kernel_matches_window_and_direct(case):
    """conv_gemm_kernel (direct-to-LDS gathered GEMM) must be the kernel AUTO picks for these shapes and must agree
    with the direct fp32-FMA kernel on the same bf16 operands, forward and data gradient."""
    g, Ln, B = case
    torch.manual_seed(12)
    x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
    d0, d1, k = g.wdims
    w = torch.randn(d0, d1, k, device=DEV) / (g.cin * g.k) ** 0.5
    bias = torch.randn(g.cout, device=DEV)
    gy = torch.randn(B, g.out_len(Ln), g.cout, device=DEV).bfloat16()
    p = ops.fwd_params(g, B, Ln)
    kid = L.lib().ga_conv_kernel_id(__import__("ctypes").byref(p), 1, L.IMPL_AUTO)
    assert kid in L.GEMM_KERNELS, kid
    outs = {}
    for impl in (L.IMPL_MFMA, L.IMPL_DIRECT):
        ops.set_conv_impl(impl)
        try:
            yf = ops.conv_forward(g, x, ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), bias, post_act=L.ACT_LRELU)
            dx = ops.conv_dgrad(g, gy, ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16), Ln)
            outs[impl] = (yf.float().cpu(), dx.float().cpu())
        finally:
            ops.set_conv_impl(L.IMPL_AUTO)
    close(outs[L.IMPL_MFMA][0], outs[L.IMPL_DIRECT][0], 1e-2, f"{g} fwd")
    close(outs[L.IMPL_MFMA][1], outs[L.IMPL_DIRECT][1], 1e-2, f"{g} dgrad")


@pytest.mark.parametrize("case", [
    (ConvGeom("conv", 64, 128, 41, 4, 20, 1), 2050, 2),    # D block 2 (phase-fast taps), windows that start in the left padding of item 0
    (ConvGeom("conv", 256, 512, 41, 4, 20, 1), 469, 3),    # D block 4: tiles span batch items, ragged last tile
    (ConvGeom("conv", 256, 256, 3, 1, 8, 8), 375, 3),      # G dilated conv: padding 8 rows at both ends of every item
    (ConvGeom("conv", 128, 256, 3, 1, 1, 1), 37, 5),       # short rows: several items per tile
    (ConvGeom("conv", 32, 64, 41, 4, 20, 1), 3001, 3),     # Cin 32: two taps per K-step (flat form both times: the K position is per-lane)
    (ConvGeom("conv", 80, 64, 4, 1, 3, 1), 300, 2),        # K = 320 is a multiple of 64 but Cin is not: K-steps straddle taps (flat form)
    (ConvGeom("conv", 192, 96, 5, 2, 2, 1), 401, 3),       # Cin = 3 x 64, tap-fast order, M = 96, stride 2
], ids=lambda c: f"{c[0].cin}-{c[0].cout}-k{c[0].k}s{c[0].stride}-L{c[1]}")
def test_buffer_load_staging_is_bit_identical_to_flat_addresses(case, monkeypatch):
    """Round 3: operands staged by buffer_load ... lds (constant per-lane offsets, zero padding by the hardware range check)
    against the flat-address form with its zero-page select: same K order, same accumulation order, so forward, data gradient
    and (to fp32 atomics order) weight gradient must agree bit for bit / to 1e-6 -- the range check must return zeros exactly
    where the select read the zero page (rows before / after an item, rows past the last output row, channels past M)."""
    g, Ln, B = case
    torch.manual_seed(21)
    x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
    w = torch.randn(*g.wdims, device=DEV) / (g.cin * g.k) ** 0.5
    bias = torch.randn(g.cout, device=DEV)
    gy = torch.randn(B, g.out_len(Ln), g.cout, device=DEV).bfloat16()
    wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
    outs = {}
    for v in ("0", "1"):
        monkeypatch.setenv("GA_GEMM_BUF", v)
        monkeypatch.setenv("GA_WGRAD_BUF", v)
        monkeypatch.setenv("GA_CONV_WIN_BUF", v)
        ops.set_conv_impl(L.IMPL_MFMA)
        try:
            yf = ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU)
            dx = ops.conv_dgrad(g, gy, wd, Ln)
        finally:
            ops.set_conv_impl(L.IMPL_AUTO)
        dw = ops.conv_wgrad(g, gy, x, want_bias=True)  # (the bias gradient rides in the weight-gradient GEMM as a constant-one column)
        outs[v] = (yf, dx, [t.float() for t in (dw if isinstance(dw, (tuple, list)) else [dw]) if t is not None])
    assert torch.equal(outs["0"][0], outs["1"][0]), "forward"
    assert torch.equal(outs["0"][1], outs["1"][1]), "data gradient"
    for a, b in zip(outs["0"][2], outs["1"][2]):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), "weight gradient"


def test_gemm_lds_budget_changes_the_tile_not_the_result():
    """ga_set_gemm_lds_budget: under a 25 KiB budget the generator-sized gathered GEMMs (256 -> 256 channels, k3, 64 x 375 rows) and a
    128 x 128-tile layer take the 64 x 128 single-buffer configuration (0x1221) so that they fit beside a persistent LSTM workgroup;
    same K order, same accumulation order: bit-identical outputs, and the setter returns the previous value."""
    import ctypes as C
    torch.manual_seed(23)
    for g, B, Ln in ((ConvGeom("conv", 256, 256, 3, 1, 2, 2), 8, 375), (ConvGeom("conv", 128, 256, 41, 4, 20, 1), 16, 7500)):
        x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
        w = torch.randn(*g.wdims, device=DEV) / (g.cin * g.k) ** 0.5
        bias = torch.randn(g.cout, device=DEV)
        wf = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16)
        p = ops.fwd_params(g, B, Ln)
        ops.set_conv_impl(L.IMPL_MFMA)
        try:
            v0 = L.lib().ga_conv_gemm_variant(C.byref(p)) & 0xffff
            y0 = ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU)
            assert L.lib().ga_set_gemm_lds_budget(25 * 1024) == 0
            try:
                v1 = L.lib().ga_conv_gemm_variant(C.byref(p)) & 0xffff
                y1 = ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU)
            finally:
                assert L.lib().ga_set_gemm_lds_budget(0) == 25 * 1024
        finally:
            ops.set_conv_impl(L.IMPL_AUTO)
        assert v0 != 0x1221 and v1 == 0x1221, (hex(v0), hex(v1))
        assert torch.equal(y0, y1)


def test_ping_pong_kernel_is_bit_identical_to_the_gathered_kernel():
    """conv_pp_kernel (256 x 256 tiles, eight waves, 16x16x32 MFMAs, buffer-load staging, phase-fast taps) on the shapes AUTO
    gives it (D blocks.4 forward, K = 10496) and on a folded data gradient: same K order and per-output accumulation order
    as conv_gemm_kernel, so the results agree to the last bit; both MFMA shapes of the kernel."""
    torch.manual_seed(22)
    g = ConvGeom("conv", 256, 512, 41, 4, 20, 1)
    B, Ln = 3, 1875
    x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
    w = torch.randn(*g.wdims, device=DEV) / (g.cin * g.k) ** 0.5
    bias = torch.randn(g.cout, device=DEV)
    gy = torch.randn(B, g.out_len(Ln), g.cout, device=DEV).bfloat16()
    wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
    outs = {}
    import os
    for name, impl, mf in (("gemm", L.IMPL_MFMA, None), ("pp16", L.IMPL_MFMA_PP, "16"), ("pp32", L.IMPL_MFMA_PP, "32")):
        if mf is not None:
            os.environ["GA_GEMM_PP_MFMA"] = mf
        ops.set_conv_impl(impl)
        ops.FORCE_KSPLIT = 1  # (the few tiles of this small batch would otherwise be split along K: another summation order)
        try:
            outs[name] = (ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU), ops.conv_dgrad(g, gy, wd, Ln))
        finally:
            ops.set_conv_impl(L.IMPL_AUTO)
            ops.FORCE_KSPLIT = None
            os.environ.pop("GA_GEMM_PP_MFMA", None)
    for name in ("pp16", "pp32"):
        assert torch.equal(outs[name][0], outs["gemm"][0]), name + " forward"
        assert torch.equal(outs[name][1], outs["gemm"][1]), name + " data gradient"
    # and AUTO hands the full-size layer to it (K = 10496, a grid of about two workgroups per CU)
    p = ops.fwd_params(g, 128, 30000)
    assert L.lib().ga_conv_kernel_id(__import__("ctypes").byref(p), 1, L.IMPL_AUTO) == 11


STREAM_CASES = [
    # thin layers (M <= 64, K = J*Cin <= 192): operands straight from global memory into MFMA fragments
    (ConvGeom("cconv", 64, 32, 3), 1201, 3),       # K = 192: 12 k-steps, no register double buffer
    (ConvGeom("cconv", 32, 16, 3), 901, 2),        # K = 96, 16 output channels (half an MFMA tile)
    (ConvGeom("cconv", 16, 32, 1), 901, 2),        # K = 16: one k-step, four row groups per trip
    (ConvGeom("cconv", 32, 64, 1), 777, 2),        # two m tiles
    (ConvGeom("cconv", 64, 64, 1), 515, 3),
    (ConvGeom("convt", 64, 32, 4, 2), 400, 3),     # folded phases: rows of 64 channels, per-item element bound
    (ConvGeom("conv", 16, 48, 5, 2, 2, 1), 301, 2),  # K = 80 (bucket of 6 k-steps), M = 48, strided
    (ConvGeom("conv", 24, 40, 3, 1, 1, 1), 130, 5),  # Cin = 24: taps change inside a k-step pair; tiles span batch items
]


@pytest.mark.parametrize("case", STREAM_CASES, ids=lambda c: f"{c[0].kind}-{c[0].cin}-{c[0].cout}-k{c[0].k}s{c[0].stride}-L{c[1]}")
def test_stream_kernel_matches_direct(case):
    """conv_stream_kernel must be what AUTO picks for the thin bf16 layers, with and without the fused input ELU, and
    must agree with the direct fp32-FMA kernel on the same operands: forward (ELU in, bias, residual) and data
    gradient (ELU derivative from the saved input)."""
    import ctypes
    g, Ln, B = case
    torch.manual_seed(14)
    x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
    d0, d1, k = g.wdims
    w = torch.randn(d0, d1, k, device=DEV) / (g.cin * g.k) ** 0.5
    bias = torch.randn(g.cout, device=DEV)
    Lo = g.out_len(Ln)
    gy = torch.randn(B, Lo, g.cout, device=DEV).bfloat16()
    res = torch.randn(B, Lo, g.cout, device=DEV).bfloat16()
    p = ops.fwd_params(g, B, Ln)
    if g.kind != "convt":
        assert L.lib().ga_conv_kernel_id(ctypes.byref(p), 1, L.IMPL_AUTO) == L.KERNEL_STREAM
        p.pre_act = L.ACT_ELU
        assert L.lib().ga_conv_kernel_id(ctypes.byref(p), 1, L.IMPL_AUTO) == L.KERNEL_STREAM
    wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
    outs = {}
    for impl in (L.IMPL_AUTO, L.IMPL_DIRECT):
        ops.set_conv_impl(impl)
        try:
            ops.PROFILE = []
            y0 = ops.conv_forward(g, x, wf, bias)
            y1 = ops.conv_forward(g, x, wf, bias, pre_act=L.ACT_ELU, res=res)
            dx0 = ops.conv_dgrad(g, gy, wd, Ln)
            dx1 = ops.conv_dgrad(g, gy, wd, Ln, dact_mode=L.DACT_ELU_FROM_IN, dact_src=x)
            kids = [e[0] for e in ops.PROFILE]
            outs[impl] = [t.float().cpu() for t in (y0, y1, dx0, dx1)]
        finally:
            ops.PROFILE = None
            ops.set_conv_impl(L.IMPL_AUTO)
        if impl == L.IMPL_AUTO:
            # (the causal convs' reflected-pad fix-up launches accumulate into dx: those stay with the gathered GEMM)
            # (gathered-GEMM launches are recorded by instantiation, 0xMWNB | BUF << 16: ga_conv_gemm_variant)
            assert sum(kk == L.KERNEL_STREAM for kk in kids) == 4 and all(kk in L.GEMM_KERNELS or kk >= 0x1000 for kk in kids), kids
    for i, name in enumerate(("fwd", "fwd elu+res", "dgrad", "dgrad*elu'")):
        close(outs[L.IMPL_AUTO][i], outs[L.IMPL_DIRECT][i], 1e-2, f"{g} {name}")


def _pack_index_ref(mode, D0, D1, K, S, P):
    """Packed position of every source weight element (s0, s1, k): the documented layouts of ga_pack_weights."""
    s0, s1, k = torch.meshgrid(torch.arange(D0), torch.arange(D1), torch.arange(K), indexing="ij")
    if mode == L.PACK_CONV_FWD:
        return (s0 * K + k) * D1 + s1
    if mode == L.PACK_CONV_DGRAD_S1:
        return (s1 * K + (K - 1 - k)) * D0 + s0
    if mode == L.PACK_CONV_DGRAD_STRIDED:
        jmax = (K + S - 1) // S
        r = (k % S - P) % S
        return ((r * D1 + s1) * jmax + (jmax - 1 - k // S)) * D0 + s0
    if mode == L.PACK_CONVT_FWD:
        return (((k % S) * D1 + s1) * 2 + (1 - k // S)) * D0 + s0
    return (s0 * K + k) * D1 + s1


@pytest.mark.parametrize("shape", [(256, 128, 41, 4, 20), (130, 70, 11, 1, 5), (64, 32, 15, 1, 7), (512, 256, 16, 8, 0), (96, 200, 3, 1, 1)],
                         ids=lambda c: "x".join(map(str, c)))
def test_tiled_pack_and_unpack_match_layout_definition(shape):
    """ga_pack_weights / ga_unpack_grad (LDS-tiled transposes for the larger weights) against the layout definition,
    every mode, ragged tiles, with and without the weight-norm row scale; unpack(pack(w)) == w."""
    D0, D1, K, S, P = shape
    torch.manual_seed(5)
    w = torch.randn(D0, D1, K, device=DEV)
    scale = torch.rand(D0, device=DEV) + 0.5
    lib = L.lib()
    for mode in (L.PACK_CONV_FWD, L.PACK_CONV_DGRAD_S1, L.PACK_CONV_DGRAD_STRIDED, L.PACK_CONVT_FWD, L.PACK_CONVT_DGRAD):
        if mode == L.PACK_CONVT_FWD and K != 2 * S:
            continue
        if mode == L.PACK_CONV_DGRAD_STRIDED and S == 1:
            continue
        n = lib.ga_packed_elems(mode, D0, D1, K, S)
        idx = _pack_index_ref(mode, D0, D1, K, S, P).reshape(-1)
        for sc in (None, scale):
            for dtype in (torch.float32, torch.bfloat16):
                out = torch.full((n,), 7.0, device=DEV, dtype=dtype)
                L.check(lib.ga_pack_weights(w.data_ptr(), L.ptr(sc), out.data_ptr(), L.dt(out), mode, D0, D1, K, S, P, L.stream()), "pack")
                src = w if sc is None else w * sc[:, None, None]
                want = torch.zeros(n, dtype=dtype) if mode == L.PACK_CONV_DGRAD_STRIDED else torch.full((n,), 7.0, dtype=dtype)
                want[idx] = src.cpu().to(dtype).reshape(-1)
                assert torch.equal(out.cpu(), want), (mode, sc is not None, dtype)
        packed = torch.randn(n, device=DEV)
        for acc in (0, 1):
            dw = torch.ones(D0, D1, K, device=DEV)
            L.check(lib.ga_unpack_grad(packed.data_ptr(), dw.data_ptr(), mode, D0, D1, K, S, P, acc, L.stream()), "unpack")
            want = packed.cpu()[idx].reshape(D0, D1, K) + (1.0 if acc else 0.0)
            assert torch.equal(dw.cpu(), want), (mode, acc)


@pytest.mark.parametrize("shape", [(3, 4001, 32, 15), (2, 1000, 64, 7), (5, 130, 16, 3)], ids=lambda c: "x".join(map(str, c)))
def test_single_channel_mfma_kernels_match_direct(shape):
    """D's first conv (one input channel): forward on the streaming MFMA kernel and weight + bias gradient on the Toeplitz
    MFMA kernel, both on a zero-padded copy of the waveform, against the direct kernels on the unpadded input."""
    B, Ln, M, K = shape
    g = ConvGeom("conv", 1, M, K, 1, K // 2, 1)
    torch.manual_seed(40)
    x = torch.randn(B, Ln, 1, device=DEV).bfloat16()
    w = torch.randn(M, 1, K, device=DEV) / K ** 0.5
    bias = torch.randn(M, device=DEV)
    ga = torch.randn(B, Ln, M, device=DEV).bfloat16()
    wf = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16)
    outs = {}
    for impl in (L.IMPL_AUTO, L.IMPL_DIRECT):
        ops.set_conv_impl(impl)
        try:
            ops.PROFILE = []
            y = ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU)
            kid = ops.PROFILE[0][0]
            ops.PROFILE = None
            dw, db = ops.conv_wgrad(g, ga, x, want_bias=True)
            outs[impl] = (y.float().cpu(), dw.clone().cpu(), db.clone().cpu())
        finally:
            ops.PROFILE = None
            ops.set_conv_impl(L.IMPL_AUTO)
        if impl == L.IMPL_AUTO:
            assert kid == L.KERNEL_STREAM, kid
    close(outs[L.IMPL_AUTO][0], outs[L.IMPL_DIRECT][0], 1e-2, "cin1 fwd")
    close(outs[L.IMPL_AUTO][1], outs[L.IMPL_DIRECT][1], 2e-3, "cin1 dw")
    close(outs[L.IMPL_AUTO][2], outs[L.IMPL_DIRECT][2], 2e-3, "cin1 db")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_disc_chain_node_matches_layer_by_layer_autograd(precision):
    """ops.DiscChain (one autograd node per discriminator scale: masks and incoming feature gradients fused into the
    data-gradient epilogues) against the layer-by-layer ConvTrain nodes: outputs, input gradient, every parameter gradient,
    for a logits-only loss (D step), a logits + feature loss (G step: gradients arrive at every feature map), a
    features-only loss, and with D's parameters frozen."""
    from gan_amd import lab3_codec_models as M
    torch.manual_seed(77)
    disc = M.SingleScaleWaveDiscriminator().to(DEV)
    dtype = torch.float32 if precision == "fp32" else torch.bfloat16
    x0 = torch.randn(3, 4099, 1, device=DEV).to(dtype)
    tol = 2e-4 if precision == "fp32" else 2e-2

    def run(fused, mode, freeze):
        M.FUSED_DISC_CHAIN = fused
        for prm in disc.parameters():
            prm.grad = None
            prm.requires_grad_(not freeze)
        x = x0.clone().requires_grad_(True)
        logits, feats = disc.forward_nlc(x)
        torch.manual_seed(5)
        loss = 0.0
        if mode in ("logits", "both"):
            loss = loss + (logits.float() * torch.randn_like(logits.float())).sum()
        if mode in ("feats", "both"):
            for f in feats:
                loss = loss + (f.float() * torch.randn_like(f.float())).mean() * 50.0
        loss.backward()
        outs = [logits.detach().float().cpu()] + [f.detach().float().cpu() for f in feats]
        grads = [x.grad.float().cpu()] + [prm.grad.float().cpu() for prm in disc.parameters() if prm.grad is not None]
        return outs, grads

    try:
        for mode, freeze in (("logits", False), ("both", False), ("feats", False), ("both", True)):
            o1, g1 = run(True, mode, freeze)
            o0, g0 = run(False, mode, freeze)
            want = 1 if freeze else (1 + 18 if mode != "feats" else 1 + 15)  # (features-only: the logits conv gets no gradient)
            assert len(o1) == len(o0) and len(g1) == len(g0) == want, (mode, freeze, len(g1), len(g0))
            for a, b in zip(o1, o0):
                assert torch.equal(a, b), (mode, "forward differs")
            for i, (a, b) in enumerate(zip(g1, g0)):
                close(a, b, tol, f"{mode} frozen={freeze} grad {i}")
    finally:
        M.FUSED_DISC_CHAIN = True
        for prm in disc.parameters():
            prm.requires_grad_(True)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_joined_real_fake_pass_matches_two_passes(precision):
    """MultiScaleWaveDiscriminator.forward_real_fake (the G step's D(real) under no_grad + D(fake) as ONE pass over the joined
    batch, ops.DiscChainSplit) against the two separate passes: every logit / feature map of both halves, and the gradient
    that reaches the fake waveform from a hinge-like + feature-matching-like loss.  Unequal halves and an odd length."""
    from gan_amd import lab3_codec_models as M
    torch.manual_seed(78)
    D = M.MultiScaleWaveDiscriminator(3, precision=precision).to(DEV)
    for prm in D.parameters():
        prm.requires_grad_(False)
    xr = torch.randn(2, 1, 6007, device=DEV)
    xf0 = torch.randn(3, 1, 6007, device=DEV)
    tol = 2e-4 if precision == "fp32" else 2e-2

    def loss_of(real_outs, fake_outs):
        torch.manual_seed(6)
        loss = 0.0
        for (lr, fr), (lf, ff) in zip(real_outs, fake_outs):
            loss = loss - lf.float().mean()
            for a, b in zip(fr, ff):
                loss = loss + (b.float() - a.float().mean(dim=0, keepdim=True)).abs().mean()
        return loss

    xf = xf0.clone().requires_grad_(True)
    with torch.no_grad():
        r_ref = D(xr)
    f_ref = D(xf)
    loss_of(r_ref, f_ref).backward()
    g_ref = xf.grad.clone()
    xf2 = xf0.clone().requires_grad_(True)
    r_new, f_new = D.forward_real_fake(xr, xf2)
    loss_of(r_new, f_new).backward()
    for (lr, fr), (lr2, fr2) in zip(r_ref + f_ref, r_new + f_new):
        assert lr.shape == lr2.shape
        close(lr2.float().cpu(), lr.float().cpu(), tol, "logits")
        for a, b in zip(fr, fr2):
            assert a.shape == b.shape
            close(b.float().cpu(), a.float().cpu(), tol, "feature")
    assert all(not t.requires_grad for lg, fs in r_new for t in [lg] + fs)
    close(xf2.grad.cpu(), g_ref.cpu(), tol, "fake waveform gradient")
    # feature matching inside the node (reductions in forward, gradient in the data-gradient epilogues) against
    # multiscale_feature_matching_loss on the two-pass outputs; equal halves, with and without a logits term
    xr3 = torch.randn(3, 1, 6007, device=DEV)
    for w_adv in (0.5, 0.0):
        xa = xf0.clone().requires_grad_(True)
        with torch.no_grad():
            r_ref = D(xr3)
        f_ref = D(xa)
        fm_ref = M.multiscale_feature_matching_loss(r_ref, f_ref)
        (1.7 * fm_ref + w_adv * M.multiscale_hinge_g_loss(f_ref)).backward()
        xb = xf0.clone().requires_grad_(True)
        r_new, f_new, fm_new = D.forward_real_fake(xr3, xb, want_fm=True)
        (1.7 * fm_new + w_adv * M.multiscale_hinge_g_loss(f_new)).backward()
        assert float(fm_new) == pytest.approx(float(fm_ref), rel=1e-5 if precision == "fp32" else 1e-3)
        close(xb.grad.cpu(), xa.grad.cpu(), tol, f"fake waveform gradient with feature matching (adv weight {w_adv})")
    for prm in D.parameters():
        prm.requires_grad_(True)
    with pytest.raises(RuntimeError):
        D.forward_real_fake(xr, xf0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pack_multi_matches_per_layer_packs(dtype):
    """ops.prepack_layers (ga_pack_multi: the weight-norm scales and both packs of every layer of a module in two launches)
    against the per-layer ga_wn_scale + ga_pack_weights calls, bit for bit, on the discriminator's and the generator's
    layer sets (weight-normed and plain, strided data-gradient packs with zero padding taps, tiny and large weights)."""
    from gan_amd import lab3_codec_models as M
    from gan_amd.optim import FusedAdamW
    torch.manual_seed(12)
    D = M.MultiScaleWaveDiscriminator(2).to(DEV)
    G = M.CodecLatentTranslator(hidden_channels=64, n_blocks=13).to(DEV)  # 28 layers: more than one chunk
    with torch.no_grad():
        for prm in list(D.parameters()) + list(G.parameters()):
            prm.mul_(1.0 + 0.1 * torch.rand_like(prm))
    opts = [FusedAdamW(D.parameters(), lr=1e-4), FusedAdamW(G.parameters(), lr=1e-4)]  # registers the parameter ranges
    layers = D._pack_layers() + [(G.in_proj.weight, None, G._g_in)] + \
        [(c.weight, None, g) for blk in G.blocks for c, g in ((blk.conv1, blk.g1), (blk.conv2, blk.g2))] + [(G.out_proj.weight, None, G._g_out)]
    assert len(layers) == 12 + 28
    ops._PACK_CACHE.clear()
    ops.prepack_layers(layers, dtype)
    for v, gp, geom in layers:
        hit = ops._PACK_CACHE[(v.data_ptr(), geom, dtype)]
        scale = None
        if gp is not None:
            scale, norm = ops.wn_scale(v, gp)
            assert torch.equal(hit[3], norm)
        assert torch.equal(hit[1], ops.pack_weight(geom, v, geom.fwd_pack, dtype, scale)), geom
        assert torch.equal(hit[2], ops.pack_weight(geom, v, geom.dgrad_pack, dtype, scale)), geom
    # the forward pass that follows takes the cached packs: no pack kernel is launched again
    before = dict(ops._PACK_CACHE)
    ops.prepack_layers(layers, dtype)
    assert all(ops._PACK_CACHE[k][1] is before[k][1] for k in before)
    del opts


@pytest.mark.parametrize("preset_grad", [False, True])
def test_deferred_weight_gradients_match_per_layer_path(preset_grad):
    """ops.WeightGradSink + ga_wgrad_finish_multi (unpack + weight-norm backward + copy into the optimizer's flat gradient
    buffer, all conv layers of a module in one launch) against the per-layer autograd path: the flat gradient buffer after
    _backward_and_reduce, for the weight-normed discriminator and the plain-conv generator; with gradients already present
    in p.grad (the R1 term puts them there before the hinge backward) the deferred ones must add to them."""
    from gan_amd import lab3_codec_models as M
    from gan_amd.lab3_codec_train import _backward_and_reduce
    from gan_amd.optim import FusedAdamW
    torch.manual_seed(21)
    D = M.MultiScaleWaveDiscriminator(2, precision="bf16").to(DEV)
    G = M.CodecLatentTranslator(hidden_channels=64, n_blocks=2, precision="bf16").to(DEV)
    opt_d, opt_g = FusedAdamW(D.parameters(), lr=1e-4), FusedAdamW(G.parameters(), lr=1e-4)
    wav = torch.randn(3, 1, 5003, device=DEV)
    q, zc, zs, nz = torch.randn(2, 128, 29, device=DEV), torch.randn(2, 128, device=DEV), torch.randn(2, 128, device=DEV), torch.randn(2, 32, device=DEV)

    def flat(sink_on, which):
        ops.WGRAD_SINK = sink_on
        opt = opt_d if which == "d" else opt_g
        opt.zero_grad(set_to_none=True)
        if preset_grad:
            for i, prm in enumerate(opt.params):
                prm.grad = torch.full_like(prm, 0.01 * (i + 1))
        torch.manual_seed(3)
        if which == "d":
            loss = sum((lg.float() * torch.randn_like(lg.float())).sum() + sum((f.float() * torch.randn_like(f.float())).mean() for f in fs)
                       for lg, fs in D(wav))
        else:
            out = G(q_src=q, z_content=zc, z_style_tgt=zs, noise=nz)
            loss = (out.float() * torch.randn_like(out.float())).sum()
        _backward_and_reduce(loss, opt, None)
        return opt.flat_g.clone()

    try:
        for which in ("d", "g"):
            a, b = flat(True, which), flat(False, which)
            assert float(b.abs().max()) > 0
            close(a.cpu(), b.cpu(), 2e-5, f"flat gradient buffer ({which})")
    finally:
        ops.WGRAD_SINK = True


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_joined_hinge_and_scalar_loss_tails(dtype):
    """ops.HingeDJoined against hinge_d_loss on the two slices (value and gradient, unequal halves, NCL view of an NLC
    buffer); ops.StftCombine against the torch expression it replaces, including a zero sum (sqrt(0) := 0 with zero
    derivative); ops.weighted_sum against the plain weighted sum."""
    from gan_amd import lab3_codec_models as M
    torch.manual_seed(91)
    base = (1.5 * torch.randn(5, 301, 1, device=DEV)).to(dtype)
    for b0 in (2, 3):
        a = base.clone().requires_grad_(True)
        lg = a.permute(0, 2, 1)  # [B, 1, L] view, as the discriminator returns its logits
        ref = M.hinge_d_loss(lg[:b0], lg[b0:])
        ref.backward()
        b = base.clone().requires_grad_(True)
        got = ops.HingeDJoined.apply(b.permute(0, 2, 1), b0)
        got.backward()
        assert float(got) == pytest.approx(float(ref), rel=1e-6)
        close(b.grad.float().cpu(), a.grad.float().cpu(), 1e-6, f"joined hinge gradient (b0={b0})")
    sums = torch.tensor([[4.0, 9.0, 50.0], [0.0, 2.5, 7.0], [3.0, 0.0, 1.0]], device=DEV, requires_grad=True)
    inv = [1.0 / 100.0, 1.0 / 7.0, 0.5]

    def torch_tail(sm):
        tot = 0.0
        for r in range(sm.shape[0]):
            s0, s1, s2 = sm[r, 0], sm[r, 1], sm[r, 2]
            r0 = torch.where(s0 > 0, s0, torch.ones_like(s0)).sqrt() * (s0 > 0)
            r1 = torch.where(s1 > 0, s1, torch.ones_like(s1)).sqrt() * (s1 > 0)
            tot = tot + r0 / (r1 + 1e-8) + s2 * inv[r]
        return tot / float(sm.shape[0])

    ref = torch_tail(sums[:2])  # rows with a positive denominator: same value and gradient
    sm2 = sums.detach()[:2].clone().requires_grad_(True)
    got = ops.StftCombine.apply(sm2, ops.const_vec(inv[:2], DEV))
    g_ref, = torch.autograd.grad(ref, sums)
    g_got, = torch.autograd.grad(got, sm2)
    assert float(got) == pytest.approx(float(ref), rel=1e-6)
    close(g_got.cpu(), g_ref[:2].cpu(), 1e-5, "stft tail gradient")
    sm3 = sums.detach().clone().requires_grad_(True)  # third row: zero denominator sum -> r0 / 1e-8, finite, zero d/ds1
    got3 = ops.StftCombine.apply(sm3, ops.const_vec(inv, DEV))
    g3, = torch.autograd.grad(got3, sm3)
    assert torch.isfinite(got3) and torch.isfinite(g3).all() and float(g3[2, 1]) == 0.0 and float(g3[1, 0]) == 0.0
    terms = [torch.tensor(v, device=DEV, requires_grad=True) for v in (0.3, -1.25, 4.0)]
    ws = ops.weighted_sum(terms, [2.0, 0.5, 0.0])
    ws.backward()
    assert float(ws) == pytest.approx(0.6 - 0.625, rel=1e-6)
    assert [float(t.grad) for t in terms] == [2.0, 0.5, 0.0]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_groupnorm_forward_matches_two_launch_path(dtype):
    """ga_gn_film_act_fused (statistics + normalise + FiLM + SiLU + residual in one launch) against ga_gn_stats followed by
    ga_gn_film_act_fwd: outputs and the saved statistics, bit for bit (same expression order), with and without FiLM /
    residual, a slab longer than the register cache (T = 2100) and one shorter than a pass (T = 5)."""
    torch.manual_seed(17)
    for (B, T, Cn, G) in ((3, 375, 256, 8), (2, 2100, 64, 8), (2, 5, 128, 8)):
        x = (torch.randn(B, T, Cn, device=DEV) * 1.7 + 0.3).to(dtype)
        gamma, beta = torch.randn(Cn, device=DEV), torch.randn(Cn, device=DEV)
        for with_ab, with_res in ((True, True), (False, False), (True, False)):
            ab = 0.3 * torch.randn(B, 2 * Cn, device=DEV) if with_ab else None
            res = torch.randn(B, T, Cn, device=DEV).to(dtype) if with_res else None
            outs = []
            for fused in (True, False):
                ops.GN_FUSED = fused
                try:
                    outs.append(ops.GnFilmAct.apply(x, gamma, beta, ab, res, G, L.ACT_SILU, 1e-5))
                finally:
                    ops.GN_FUSED = True
            assert torch.equal(outs[0], outs[1]), (B, T, Cn, with_ab, with_res)


def test_folded_phases_match_phase_by_phase():
    """Strided-conv data gradients and transposed-conv forwards with their output phases folded into the channel dimension
    (one GEMM over rows of S*M channels) against the S phase-by-phase launches, including item lengths that are not a
    multiple of S (the per-item element bound)."""
    torch.manual_seed(31)
    for g, Ln, B in ((ConvGeom("conv", 32, 64, 41, 4, 20, 1), 3001, 2), (ConvGeom("conv", 128, 256, 41, 4, 20, 1), 1875, 2),
                     (ConvGeom("conv", 256, 512, 41, 4, 20, 1), 470, 3), (ConvGeom("convt", 128, 64, 8, 4), 301, 2),
                     (ConvGeom("convt", 64, 32, 4, 2), 777, 3), (ConvGeom("convt", 512, 256, 16, 8), 40, 2)):
        x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
        d0, d1, k = g.wdims
        w = torch.randn(d0, d1, k, device=DEV) / (g.cin * g.k) ** 0.5
        bias = torch.randn(g.cout, device=DEV)
        gy = torch.randn(B, g.out_len(Ln), g.cout, device=DEV).bfloat16()
        wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
        outs = []
        for merge in (True, False):
            ops.MERGE_PHASES = merge
            try:
                if g.kind == "convt":
                    outs.append(ops.conv_forward(g, x, wf, bias).float().cpu())
                else:
                    outs.append(ops.conv_dgrad(g, gy, wd, Ln).float().cpu())
            finally:
                ops.MERGE_PHASES = True
        close(outs[0], outs[1], 1e-2, f"{g} L={Ln}")


def test_split_k_matches_single_pass():
    """Few-tile, deep-K layers (D blocks 3-4 at the small scales) are split over K with fp32 partial sums; the result must
    agree with the single-pass kernel to bf16 output rounding, forward (bias + LeakyReLU) and data gradient."""
    import ctypes
    torch.manual_seed(21)
    for g, Ln, B in ((ConvGeom("conv", 256, 512, 41, 4, 20, 1), 469, 3), (ConvGeom("conv", 128, 256, 41, 4, 20, 1), 938, 2),
                     (ConvGeom("conv", 256, 512, 41, 4, 20, 1), 117, 5)):
        x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
        d0, d1, k = g.wdims
        w = torch.randn(d0, d1, k, device=DEV) / (g.cin * g.k) ** 0.5
        bias = torch.randn(g.cout, device=DEV)
        gy = torch.randn(B, g.out_len(Ln), g.cout, device=DEV).bfloat16()
        p = ops.fwd_params(g, B, Ln)
        assert L.lib().ga_conv_suggest_ksplit(ctypes.byref(p), 1, L.IMPL_AUTO) > 1, g
        wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
        outs = []
        for split in (True, False):
            ops.SPLIT_K = split
            try:
                outs.append((ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU).float().cpu(),
                             ops.conv_dgrad(g, gy, wd, Ln).float().cpu()))
            finally:
                ops.SPLIT_K = True
        close(outs[0][0], outs[1][0], 1e-2, f"{g} fwd")
        close(outs[0][1], outs[1][1], 1e-2, f"{g} dgrad")


def test_mfma_weight_gradient_matches_direct():
    """bf16 weight gradient: the gathered-GEMM MFMA kernel (direct-to-LDS loads, transposing LDS reads) and the older
    window MFMA kernel against the direct kernel on the same bf16 operands (ragged M / K / reduction tiles included)."""
    torch.manual_seed(9)
    for g, Ln, B in ((ConvGeom("conv", 32, 64, 41, 4, 20, 1), 3001, 2), (ConvGeom("conv", 256, 512, 41, 4, 20, 1), 1875, 2),
                     (ConvGeom("conv", 256, 256, 3, 1, 8, 8), 375, 3), (ConvGeom("conv", 128, 256, 3, 1, 1, 1), 375, 3),
                     (ConvGeom("conv", 64, 128, 41, 4, 20, 1), 938, 2), (ConvGeom("conv", 16, 48, 5, 2, 2, 1), 301, 3),
                     (ConvGeom("conv", 128, 256, 3, 1, 1, 1), 37, 9), (ConvGeom("conv", 24, 40, 7, 1, 3, 2), 1000, 1)):
        x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
        ga = torch.randn(B, g.out_len(Ln), g.cout, device=DEV).bfloat16()
        outs = []
        for impl in (L.IMPL_MFMA, L.IMPL_DIRECT):
            ops.set_conv_impl(impl)
            try:
                outs.append(ops.conv_wgrad(g, ga, x))
            finally:
                ops.set_conv_impl(L.IMPL_AUTO)
        close(outs[0], outs[1].cpu(), 2e-3, f"wgrad {g}")
        ops.set_conv_impl(L.IMPL_MFMA)
        ops.WGRAD_SLABS = True
        try:
            close(ops.conv_wgrad(g, ga, x), outs[1].cpu(), 2e-3, f"slab wgrad {g}")
        finally:
            ops.WGRAD_SLABS = False
            ops.set_conv_impl(L.IMPL_AUTO)
        if g.cin >= 32 and g.cin % 8 == 0 and g.cout % 8 == 0 and B * g.out_len(Ln) >= 256:
            ops.set_conv_impl(L.IMPL_MFMA_WINDOW)
            try:
                close(ops.conv_wgrad(g, ga, x), outs[1].cpu(), 2e-3, f"window wgrad {g}")
            finally:
                ops.set_conv_impl(L.IMPL_AUTO)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("kind", ["cconv7", "cconv3", "cconv1", "convt8", "convt5", "convt2", "tail", "tail_long"])
def test_frozen_decoder_layers(kind, dtype):
    torch.manual_seed(4)
    B = 2
    geom, Ln = {"cconv7": (ConvGeom("cconv", 128, 512, 7), 40), "cconv3": (ConvGeom("cconv", 256, 128, 3), 333),
                "cconv1": (ConvGeom("cconv", 64, 128, 1), 500), "convt8": (ConvGeom("convt", 512, 256, 16, 8), 40),
                "convt5": (ConvGeom("convt", 256, 128, 10, 5), 70), "convt2": (ConvGeom("convt", 64, 32, 4, 2), 400),
                "tail": (ConvGeom("cconv", 32, 1, 7), 900), "tail_long": (ConvGeom("cconv", 32, 1, 7), 7001)}[kind]
    d0, d1, k = geom.wdims
    v = torch.randn(d0, d1, k)
    gp = torch.rand(d0, 1, 1) + 0.5
    bias = torch.randn(geom.cout) * 0.1
    x = torch.randn(B, geom.cin, Ln)
    res = torch.randn(B, geom.cout, geom.out_len(Ln))
    if dtype == torch.bfloat16:
        x, res = x.bfloat16().float(), res.bfloat16().float()
    w_eff = v * (gp / v.reshape(d0, -1).norm(dim=1).reshape(-1, 1, 1))
    if dtype == torch.bfloat16:
        w_eff = w_eff.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    y_ref = _ref_conv(geom, xr, w_eff, bias, L.ACT_ELU, 0) + res
    gy = torch.randn_like(y_ref)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    gx_ref, = torch.autograd.grad(y_ref, xr, gy)
    wp_f, wp_d = ops.pack_frozen(geom, v.to(DEV), gp.to(DEV), dtype)
    xg = x.permute(0, 2, 1).contiguous().to(DEV, dtype).requires_grad_(True)
    y = ops.ConvFrozen.apply(xg, res.permute(0, 2, 1).contiguous().to(DEV, dtype), wp_f, wp_d, bias.to(DEV), geom, L.ACT_ELU)
    close(y.permute(0, 2, 1), y_ref, TOL[dtype], "forward")
    gx, = torch.autograd.grad(y, xg, gy.permute(0, 2, 1).contiguous().to(DEV, dtype))
    close(gx.permute(0, 2, 1), gx_ref, TOL[dtype], "dx")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("C,film,res", [(256, True, True), (256, False, False), (192, True, False)])
def test_groupnorm_film_silu(C, film, res, dtype):
    torch.manual_seed(5)
    B, T = 3, 75
    x = torch.randn(B, C, T) * 1.5 + 0.3
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.1
    ab = torch.randn(B, 2 * C) * 0.3 if film else None
    r = torch.randn(B, C, T) if res else None
    if res and dtype == torch.bfloat16:
        r = r.bfloat16().float()
    leaves = [t.clone().requires_grad_(True) for t in (x, gamma, beta)] + ([ab.clone().requires_grad_(True)] if film else [])
    h = F.group_norm(leaves[0], 8, leaves[1], leaves[2])
    if film:
        h = h * (1 + leaves[3][:, :C].unsqueeze(-1)) + leaves[3][:, C:].unsqueeze(-1)
    y_ref = F.silu(h) + (r if res else 0)
    gy = torch.randn_like(y_ref)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    refs = torch.autograd.grad(y_ref, leaves, gy)
    xg = x.permute(0, 2, 1).contiguous().to(DEV, dtype).requires_grad_(True)
    gl = [gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)]
    abg = ab.to(DEV).requires_grad_(True) if film else None
    rg = r.permute(0, 2, 1).contiguous().to(DEV, dtype) if res else None
    y = ops.GnFilmAct.apply(xg, gl[0], gl[1], abg, rg, 8, L.ACT_SILU, 1e-5)
    tol = TOL[dtype]
    close(y.permute(0, 2, 1), y_ref, tol, "fwd")
    grads = torch.autograd.grad(y, [xg] + gl + ([abg] if film else []), gy.permute(0, 2, 1).contiguous().to(DEV, dtype))
    close(grads[0].permute(0, 2, 1), refs[0], tol, "dx")
    close(grads[1], refs[1], tol, "dgamma")
    close(grads[2], refs[2], tol, "dbeta")
    if film:
        close(grads[3], refs[3], tol, "dab")


def test_small_elementwise_and_reductions():
    torch.manual_seed(6)
    B, Cn, T = 3, 128, 40
    raw, q = torch.randn(B, T, Cn), torch.randn(B, T, Cn)
    rr = raw.clone().requires_grad_(True)
    y_ref = q + 0.5 * torch.tanh(rr)
    gy = torch.randn_like(y_ref)
    g_ref, = torch.autograd.grad(y_ref, rr, gy)
    rg = raw.to(DEV).requires_grad_(True)
    y = ops.TanhResidual.apply(rg, q.to(DEV), 0.5)
    close(y, y_ref, 1e-5, "tanh fwd")
    close(torch.autograd.grad(y, rg, gy.to(DEV))[0], g_ref, 1e-5, "tanh bwd")
    for Ln in (1000, 1001, 37):
        x = torch.randn(B, Ln)
        xr = x.clone().requires_grad_(True)
        p_ref = F.avg_pool1d(xr.unsqueeze(1), 4, 2, 1).squeeze(1)
        gp = torch.randn_like(p_ref)
        gx_ref, = torch.autograd.grad(p_ref, xr, gp)
        xg = x.to(DEV).requires_grad_(True)
        p = ops.AvgPool4s2.apply(xg)
        close(p, p_ref, 1e-6, "pool fwd")
        close(torch.autograd.grad(p, xg, gp.to(DEV))[0], gx_ref, 1e-6, "pool bwd")
    a, b = torch.randn(B, 33, 77), torch.randn(B, 33, 77)
    for op, f in ((L.RED_ABSDIFF, lambda u, w: (u - w).abs().sum()), (L.RED_RELU1M, lambda u, w: F.relu(1 - u).sum()),
                  (L.RED_RELU1P, lambda u, w: F.relu(1 + u).sum()), (L.RED_SUM, lambda u, w: u.sum())):
        ar = a.clone().requires_grad_(True)
        s_ref = f(ar, b)
        ga_ref, = torch.autograd.grad(s_ref * 0.37, ar)
        ag = a.to(DEV).requires_grad_(True)
        s = ops.reduce_sum(ag, b.to(DEV) if op == L.RED_ABSDIFF else None, op)
        assert s.dim() == 0 and float(s) == pytest.approx(float(s_ref), rel=1e-5)
        close(torch.autograd.grad(s * 0.37, ag)[0], ga_ref, 1e-6, f"reduce grad {op}")
        s2 = ops.reduce_sum(ag, b.to(DEV) if op == L.RED_ABSDIFF else None, op, 0.37)  # the scale applied inside the kernels
        assert float(s2) == pytest.approx(0.37 * float(s_ref), rel=1e-5)
        close(torch.autograd.grad(s2, ag)[0], ga_ref, 1e-6, f"scaled reduce grad {op}")
    # permuted ([B,C,L] view of an NLC buffer) operands are reduced without a copy and get a permuted gradient
    an = a.permute(0, 2, 1).contiguous().to(DEV).requires_grad_(True)
    s = ops.reduce_sum(an.permute(0, 2, 1), b.permute(0, 2, 1).contiguous().to(DEV).permute(0, 2, 1), L.RED_ABSDIFF)
    assert float(s) == pytest.approx(float((a - b).abs().sum()), rel=1e-5)
    ar = a.clone().requires_grad_(True)
    close(torch.autograd.grad(s, an)[0].permute(0, 2, 1), torch.autograd.grad((ar - b).abs().sum(), ar)[0], 1e-6, "perm grad")


def test_latent_losses_match_reference_formulas():
    from gan_amd.lab3_codec_train import latent_losses
    torch.manual_seed(7)
    B, Cn, T = 3, 128, 30
    qh, qs = torch.randn(B, Cn, T), torch.randn(B, Cn, T)
    qr = qh.clone().requires_grad_(True)
    l1 = F.l1_loss(qr, qs)
    cont = F.l1_loss(qr[:, :, 1:] - qr[:, :, :-1], qs[:, :, 1:] - qs[:, :, :-1])
    db = F.relu((qr - qs).abs().mean(dim=(1, 2)) - 0.75).mean()
    g_ref, = torch.autograd.grad(4 * l1 + cont + 2 * db, qr)
    qg = qh.to(DEV).requires_grad_(True)
    a, b, c = latent_losses(qg, qs.to(DEV), 0.75)
    assert float(a) == pytest.approx(float(l1), rel=1e-5) and float(b) == pytest.approx(float(cont), rel=1e-5)
    assert float(c) == pytest.approx(float(db), rel=1e-5) and float(db) > 0
    close(torch.autograd.grad(4 * a + b + 2 * c, qg)[0], g_ref, 1e-5, "latent grad")


def test_fused_adamw_matches_torch_adamw_with_clipping():
    from gan_amd.optim import FusedAdamW
    torch.manual_seed(8)
    shapes = [(33, 7, 3), (129,), (64, 64), (5,)]
    ps = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    gp = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ps]
    opt_ref = torch.optim.AdamW(ref, lr=2e-4, betas=(0.5, 0.999))
    opt = FusedAdamW(gp, lr=2e-4, betas=(0.5, 0.999), max_grad_norm=5.0)
    for it in range(3):
        grads = [torch.randn(s) * (10.0 if it == 1 else 0.1) for s in shapes]
        for p, g in zip(ref, grads):
            p.grad = g.clone()
        n_ref = torch.nn.utils.clip_grad_norm_(ref, 5.0)
        opt_ref.step()
        for p, g in zip(gp, grads):
            p.grad = g.to(DEV)
        opt.step()
        assert float(opt.last_grad_norm) == pytest.approx(float(n_ref), rel=1e-5)
        for a, b in zip(gp, ref):
            close(a, b.detach(), 2e-6, f"adamw it{it}")
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    close(sd["state"][2]["exp_avg"], opt_ref.state_dict()["state"][2]["exp_avg"], 1e-5, "exp_avg")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("persist", [True, False], ids=["persistent", "steps"])
@pytest.mark.parametrize("B,T", [(3, 20), (18, 7), (83, 5), (150, 3), (36, 40), (270, 4)])  # > 64 (fwd) / > 32 (bwd): batch chunks in grid z
def test_lstm_skip_matches_torch_lstm(B, T, dtype, persist, monkeypatch):
    """ga_lstm_fwd / ga_lstm_bwd against torch.nn.LSTM on the CPU: the chain of fused two-layer step launches (lstm.hip) and,
    in bf16, the persistent kernels with the weights resident in LDS (lstm_persist.hip; several groups, ragged last tile,
    more than one launch at B = 270)."""
    if persist and dtype != torch.bfloat16:
        pytest.skip("the persistent LSTM kernels are bf16 only")
    monkeypatch.setattr(ops, "LSTM_PERSIST", persist)
    torch.manual_seed(10)
    H = 512
    lstm = torch.nn.LSTM(H, H, 2)
    x = torch.randn(B, T, H) * 0.7
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    seq = xr.permute(1, 0, 2)
    y_ref = (lstm(seq)[0] + seq).permute(1, 0, 2)
    gy = torch.randn_like(y_ref)
    gx_ref, = torch.autograd.grad(y_ref, xr, gy)
    w = ops.prepare_lstm_weights(lstm.to(DEV), dtype)
    xg = x.to(DEV, dtype).requires_grad_(True)
    y = ops.LstmSkip.apply(xg, w)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    close(y, y_ref, tol, "lstm fwd")
    gx, = torch.autograd.grad(y, xg, gy.to(DEV, dtype))
    close(gx, gx_ref, tol, "lstm bwd")
    with torch.no_grad():  # inference path (no saved gates)
        close(ops.LstmSkip.apply(x.to(DEV, dtype), w), y_ref, tol, "lstm fwd nograd")
    assert not ops.lstm_persist_error()


def test_lstm_persist_error_flag_is_sticky_across_launches():
    """the persistent LSTM's error word must survive later launches (a launch clears only its own abort word and
    counters), so a timeout early in an epoch is still visible when the flag is read at its end.  The flag is raised by hand
    here (a real timeout needs a lost residency); a forward and a backward launch later it still stands, those launches ran
    normally (the per-launch abort word is separate), and reading it with clear=True re-arms it."""
    torch.manual_seed(11)
    H, B, T = 512, 5, 6
    lstm = torch.nn.LSTM(H, H, 2)
    w = ops.prepare_lstm_weights(lstm.to(DEV), torch.bfloat16)
    x = (0.5 * torch.randn(B, T, H, device=DEV)).bfloat16().requires_grad_(True)
    y0 = ops.LstmSkip.apply(x, w)
    g0, = torch.autograd.grad(y0, x, torch.ones_like(y0))
    assert not ops.lstm_persist_error()
    ws = ops._lstm_workspace(torch.device(DEV))
    ws[:4].view(torch.int32).fill_(1)  # what lp_wait leaves behind after a timeout
    y1 = ops.LstmSkip.apply(x, w)
    g1, = torch.autograd.grad(y1, x, torch.ones_like(y1))
    assert torch.equal(y0, y1) and torch.equal(g0, g1)   # the raised sticky word does not disturb later launches
    assert ops.lstm_persist_error(clear=False)           # ... and they did not erase it
    assert ops.lstm_persist_error()                      # report and clear
    assert not ops.lstm_persist_error()


def test_lds_window_kernel_matches_gathered_and_direct_kernels():
    """conv_win_kernel (input rows of a tile staged once in LDS, taps read from the window) on the two problem forms it takes:
    D blocks.1 forward (Cin 32, stride 4, k 41, zero padding at both ends, ragged last tile, fewer than 64 output channels too)
    and its phase-folded data gradient (Cin 64, stride 1, 11 taps, LeakyReLU mask in the epilogue).  Same bf16 operands and fp32
    accumulation order per output as the gathered kernel, so the two agree to the last bit; the direct kernel bounds both."""
    torch.manual_seed(33)
    for g, Ln, B in ((ConvGeom("conv", 32, 64, 41, 4, 20, 1), 3001, 3), (ConvGeom("conv", 32, 64, 41, 4, 20, 1), 1024, 2),
                     (ConvGeom("conv", 32, 48, 41, 4, 20, 1), 2050, 2)):
        x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
        w = torch.randn(*g.wdims, device=DEV) / (g.cin * g.k) ** 0.5
        bias = torch.randn(g.cout, device=DEV)
        gy = torch.randn(B, g.out_len(Ln), g.cout, device=DEV).bfloat16()
        src = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
        wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
        outs = {}
        for name, impl in (("window", L.IMPL_MFMA_LDSWIN), ("gather", L.IMPL_MFMA), ("direct", L.IMPL_DIRECT)):
            ops.set_conv_impl(impl)
            try:
                fwd = ops.conv_forward(g, x, wf, bias, post_act=L.ACT_LRELU).float().cpu()
                dgr = None
                if g.cout == 64:  # the folded data gradient has 4 * 32 = 128 output channels and Cin = 64
                    dgr = ops.conv_dgrad(g, gy, wd, Ln, dact_mode=L.DACT_LRELU_FROM_OUT, dact_src=src).float().cpu()
                outs[name] = (fwd, dgr)
            finally:
                ops.set_conv_impl(L.IMPL_AUTO)
        assert torch.equal(outs["window"][0], outs["gather"][0]), g
        close(outs["window"][0], outs["direct"][0], 1e-2, f"window fwd {g} L={Ln}")
        if outs["window"][1] is not None:
            assert torch.equal(outs["window"][1], outs["gather"][1]), g
            close(outs["window"][1], outs["direct"][1], 1e-2, f"window dgrad {g} L={Ln}")


def test_streaming_window_kernel_matches_register_streaming_and_direct_kernels():
    """conv_swin_kernel (per-wave ring of input windows in LDS, ELU once per element in place) on the three forms it takes:
    the decoder's k=3 residual convs at 64 and 32 input channels (causal reflect padding at the left edge) and the k=2s
    transposed conv with its two output phases folded into 64 channels; ragged last tiles, items shorter than one tile, fewer
    output channels than a 32-row MFMA tile, an odd output length behind the fold, with and without ELU, enough tiles that
    every wave walks its ring more than once.  Same bf16 operands, ELU arithmetic and fp32 accumulation order as
    conv_stream_kernel: equal to the last bit; the direct kernel bounds both."""
    torch.manual_seed(34)
    cases = ((ConvGeom("cconv", 64, 32, 3), 4097, 3), (ConvGeom("cconv", 64, 32, 3), 70, 2), (ConvGeom("cconv", 64, 24, 3), 1500, 2),
             (ConvGeom("cconv", 32, 16, 3), 9001, 3), (ConvGeom("cconv", 32, 16, 3), 127, 1), (ConvGeom("cconv", 32, 32, 3), 300, 2),
             (ConvGeom("convt", 64, 32, 4, 2), 2049, 3), (ConvGeom("convt", 64, 32, 4, 2), 31, 2), (ConvGeom("convt", 64, 16, 4, 2), 777, 2),
             (ConvGeom("cconv", 64, 32, 3), 60000, 40))
    for g, Ln, B in cases:
        x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
        w = torch.randn(*g.wdims, device=DEV) / (g.cin * g.k) ** 0.5
        bias = torch.randn(g.cout, device=DEV)
        wf = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16)
        for pre in (L.ACT_ELU, L.ACT_NONE):
            outs = {}
            for name, impl in (("swin", L.IMPL_MFMA_STREAMWIN), ("stream", L.IMPL_MFMA_STREAM), ("direct", L.IMPL_DIRECT)):
                if name == "direct" and Ln > 10000:
                    continue
                ops.set_conv_impl(impl)
                try:
                    outs[name] = ops.conv_forward(g, x, wf, bias, pre_act=pre).float().cpu()
                finally:
                    ops.set_conv_impl(L.IMPL_AUTO)
            assert torch.equal(outs["swin"], outs["stream"]), (g, Ln, pre)
            if "direct" in outs:
                close(outs["swin"], outs["direct"], 1.5e-2, f"swin {g} L={Ln} pre={pre}")
    # AUTO picks it for the long ELU launches only
    p = ops.fwd_params(ConvGeom("cconv", 64, 32, 3), 64, 60000)
    p.x = p.w = p.y = 256
    p.pre_act = L.ACT_ELU
    assert L.lib().ga_conv_kernel_id(C.byref(p), L.DT_BF16, L.IMPL_AUTO) == L.KERNEL_STREAM_WIN
    p.pre_act = L.ACT_NONE
    assert L.lib().ga_conv_kernel_id(C.byref(p), L.DT_BF16, L.IMPL_AUTO) == L.KERNEL_STREAM


def test_single_output_window_kernel_and_last_conv_gradient_match_direct_kernels():
    """conv_swin1_kernel (the window ring with four chains of 16x16x32 MFMAs per 64-step tile) on the two single-output layers it
    takes -- the decoder's last conv (32 -> 1, k7, reflect-causal, ELU in front, fp32 out) and D blocks.0's data gradient
    (32 -> 1, k15, zero padding both sides, bf16 out) -- against the fp32-accumulating direct kernel on the same bf16 operands;
    ragged last tiles, items shorter than a tile, several tiles per wave.  And the last conv's data gradient (1 -> 32 channels,
    ELU' from the layer input in the epilogue), which runs on the single-channel streaming kernel over a zero-padded copy."""
    torch.manual_seed(36)
    g_last = ConvGeom("cconv", 32, 1, 7)
    g_d0 = ConvGeom("conv", 1, 32, 15, 1, 7, 1)
    for B, Ln in ((3, 4099), (2, 50), (5, 66000)):
        x = torch.randn(B, Ln, 32, device=DEV).bfloat16()
        w = torch.randn(*g_last.wdims, device=DEV) / (32 * 7) ** 0.5
        bias = torch.randn(1, device=DEV)
        wf, wd = ops.pack_weight(g_last, w, g_last.fwd_pack, torch.bfloat16), ops.pack_weight(g_last, w, g_last.dgrad_pack, torch.bfloat16)
        outs = {}
        for name, impl in (("swin", L.IMPL_MFMA_STREAMWIN), ("direct", L.IMPL_DIRECT)):
            ops.set_conv_impl(impl)
            try:
                outs[name] = ops.conv_forward(g_last, x, wf, bias, pre_act=L.ACT_ELU, out_f32=True).cpu()
            finally:
                ops.set_conv_impl(L.IMPL_AUTO)
        close(outs["swin"], outs["direct"], 5e-3, f"last conv fwd B={B} L={Ln}")
        gy = torch.randn(B, Ln, 1, device=DEV).bfloat16()
        dx_auto = ops.conv_dgrad(g_last, gy, wd, Ln, dact_mode=L.DACT_ELU_FROM_IN, dact_src=x).float().cpu()
        ops.set_conv_impl(L.IMPL_DIRECT)
        try:
            dx_dir = ops.conv_dgrad(g_last, gy, wd, Ln, dact_mode=L.DACT_ELU_FROM_IN, dact_src=x).float().cpu()
        finally:
            ops.set_conv_impl(L.IMPL_AUTO)
        close(dx_auto, dx_dir, 1.5e-2, f"last conv dgrad B={B} L={Ln}")
        w0 = torch.randn(*g_d0.wdims, device=DEV) / 15 ** 0.5
        wd0 = ops.pack_weight(g_d0, w0, g_d0.dgrad_pack, torch.bfloat16)
        g32 = torch.randn(B, Ln, 32, device=DEV).bfloat16()
        outs = {}
        for name, impl in (("swin", L.IMPL_MFMA_STREAMWIN), ("direct", L.IMPL_DIRECT)):
            ops.set_conv_impl(impl)
            try:
                outs[name] = ops.conv_dgrad(g_d0, g32, wd0, Ln).float().cpu()
            finally:
                ops.set_conv_impl(L.IMPL_AUTO)
        close(outs["swin"], outs["direct"], 1.5e-2, f"D blocks.0 dgrad B={B} L={Ln}")
    # AUTO takes the window kernel for long launches of both
    p = ops.fwd_params(g_last, 4, 120000)
    p.x = p.w = p.y = 256
    p.pre_act, p.out_f32 = L.ACT_ELU, 1
    assert L.lib().ga_conv_kernel_id(C.byref(p), L.DT_BF16, L.IMPL_AUTO) == L.KERNEL_STREAM_WIN
    q = ops.dgrad_params(g_d0, 4, 120000)
    q.x = q.w = q.y = 256
    assert L.lib().ga_conv_kernel_id(C.byref(q), L.DT_BF16, L.IMPL_AUTO) == L.KERNEL_STREAM_WIN


def test_activation_after_residual_and_elu_derivative_from_output():
    """GA_RES_ADD_THEN_ACT (y = ELU(conv + bias + res), gathered-GEMM epilogue) and GA_DACT_ELU_FROM_OUT (the data gradient takes
    ELU'(pre-activation) from u = ELU(pre-activation): 1 for u > 0, else u + 1) against torch on the CPU, bf16 bound."""
    torch.manual_seed(37)
    g = ConvGeom("cconv", 64, 128, 1)
    B, Ln = 3, 700
    x = torch.randn(B, Ln, 64, device=DEV).bfloat16()
    res = torch.randn(B, Ln, 128, device=DEV).bfloat16()
    w = torch.randn(*g.wdims, device=DEV) / 8.0
    bias = torch.randn(128, device=DEV)
    wf, wd = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16), ops.pack_weight(g, w, g.dgrad_pack, torch.bfloat16)
    y = ops.conv_forward(g, x, wf, bias, post_act=L.ACT_ELU, res=res, act_after_res=True).float().cpu()
    wq = w.bfloat16().float().cpu()
    pre = F.conv1d(x.float().cpu().permute(0, 2, 1), wq, bias.cpu()) + res.float().cpu().permute(0, 2, 1)
    close(y, F.elu(pre).permute(0, 2, 1), TOL[torch.bfloat16], "ELU(conv + bias + res)")
    # data gradient of conv(u) w.r.t. the pre-activation behind u = ELU(pre): dgrad(gy) * ELU'(pre), ELU' read off u
    u = F.elu(torch.randn(B, Ln, 64, device=DEV)).bfloat16()
    gy = torch.randn(B, Ln, 128, device=DEV).bfloat16()
    dx = ops.conv_dgrad(g, gy, wd, Ln, dact_mode=L.DACT_ELU_FROM_OUT, dact_src=u).float().cpu()
    uf = u.float().cpu()
    want = F.conv_transpose1d(gy.float().cpu().permute(0, 2, 1), wq).permute(0, 2, 1) * torch.where(uf > 0, torch.ones_like(uf), uf + 1.0)
    close(dx, want, TOL[torch.bfloat16], "dgrad * ELU'(from output)")


def test_streaming_kernel_with_128_output_channels_matches_direct_kernel():
    """conv_stream_kernel's 128-channel form (M <= 128 with K <= 64: four m-tiles, the epilogue in two rounds of 64 channels through
    the same per-wave stage) -- the wide residual unit's second conv 64 -> 128 k1 with the shortcut as residual and ELU after the
    sum, plain, with a residual, with a LeakyReLU, and as a data gradient with a derivative source -- forced (GA_IMPL_MFMA_STREAM)
    against the direct kernel on the same bf16 operands; ragged row counts, M not a multiple of 32."""
    torch.manual_seed(41)
    for g, Ln, B in ((ConvGeom("cconv", 64, 128, 1), 1501, 3), (ConvGeom("cconv", 32, 96, 1), 333, 2), (ConvGeom("cconv", 16, 128, 3), 70, 5),
                     (ConvGeom("cconv", 64, 72, 1), 4099, 2), (ConvGeom("cconv", 64, 128, 1), 15000, 8)):
        x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
        res = torch.randn(B, Ln, g.cout, device=DEV).bfloat16()
        w = torch.randn(*g.wdims, device=DEV) / (g.cin * g.k) ** 0.5
        bias = torch.randn(g.cout, device=DEV)
        wf = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16)
        p = ops.fwd_params(g, B, Ln)
        p.x = p.w = p.y = 256
        assert L.lib().ga_conv_kernel_id(C.byref(p), L.DT_BF16, L.IMPL_AUTO) == L.KERNEL_STREAM
        variants = (dict(), dict(res=res), dict(res=res, post_act=L.ACT_ELU, act_after_res=True), dict(post_act=L.ACT_LRELU))
        for kw in variants:
            outs = {}
            for name, impl in (("stream", L.IMPL_MFMA_STREAM), ("direct", L.IMPL_DIRECT)):
                if name == "direct" and kw.get("act_after_res"):
                    continue
                ops.set_conv_impl(impl)
                try:
                    outs[name] = ops.conv_forward(g, x, wf, bias, **kw).float().cpu()
                finally:
                    ops.set_conv_impl(L.IMPL_AUTO)
            if "direct" in outs:
                close(outs["stream"], outs["direct"], 1e-2, f"stream M={g.cout} {sorted(kw)} L={Ln}")
            else:  # ELU(conv + bias + res): from the plain launch's pieces
                ops.set_conv_impl(L.IMPL_DIRECT)
                try:
                    plain = ops.conv_forward(g, x, wf, bias).float().cpu()
                finally:
                    ops.set_conv_impl(L.IMPL_AUTO)
                close(outs["stream"], F.elu(plain + res.float().cpu()), 1.5e-2, f"stream M={g.cout} ELU after the sum L={Ln}")
        if g.k == 1:  # as a data gradient (of a cout -> cin k1 conv) with ELU' from a derivative source
            gd = ConvGeom("cconv", g.cout, g.cin, 1)
            wd = ops.pack_weight(gd, torch.randn(*gd.wdims, device=DEV) / g.cin ** 0.5, gd.dgrad_pack, torch.bfloat16)
            src = torch.randn(B, Ln, g.cout, device=DEV).bfloat16()
            outs = {}
            for name, impl in (("stream", L.IMPL_MFMA_STREAM), ("direct", L.IMPL_DIRECT)):
                ops.set_conv_impl(impl)
                try:
                    outs[name] = ops.conv_dgrad(gd, x, wd, Ln, dact_mode=L.DACT_ELU_FROM_IN, dact_src=src).float().cpu()
                finally:
                    ops.set_conv_impl(L.IMPL_AUTO)
            close(outs["stream"], outs["direct"], 1e-2, f"stream dgrad M={g.cout} L={Ln}")


def test_cooperative_weights_in_registers_kernel_matches_direct_kernel():
    """conv_coop_kernel (the decoder's transposed conv 128 -> 64 k8 s4 folded to 256 channels x 2 taps: weights in the registers of
    an eight-wave workgroup, shared input window, zero padding at both ends) against the direct kernel on the same bf16 operands,
    and against torch; AUTO takes it for long launches only; lengths: ragged last tile, one tile, several tiles per workgroup."""
    torch.manual_seed(43)
    g = ConvGeom("convt", 128, 64, 8, 4)
    w = torch.randn(*g.wdims, device=DEV) / (128 * 2) ** 0.5
    bias = torch.randn(64, device=DEV)
    wf = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16)
    for B, Ln in ((3, 50000), (1, 140000), (40, 15000), (2, 40)):
        x = torch.randn(B, Ln, 128, device=DEV).bfloat16()
        p = ops.fwd_params(g, B, Ln)
        p.x = p.w = p.y = 256
        q, _ = ops._fold_phases(p, x, None)
        kid = L.lib().ga_conv_kernel_id(C.byref(q), L.DT_BF16, L.IMPL_AUTO)
        assert kid == (L.KERNEL_COOP if B * Ln >= 2 ** 15 else 8), (B, Ln, kid)
        y = ops.conv_forward(g, x, wf, bias).float().cpu()
        assert y.shape == (B, 4 * Ln, 64)
        if Ln <= 50000:
            ops.set_conv_impl(L.IMPL_DIRECT)
            try:
                yd = ops.conv_forward(g, x, wf, bias).float().cpu()
            finally:
                ops.set_conv_impl(L.IMPL_AUTO)
            close(y, yd, 1e-2, f"coop vs direct B={B} L={Ln}")
        if B * Ln <= 200000:
            ref = _ref_conv(g, x.float().cpu().permute(0, 2, 1), w.bfloat16().float().cpu(), bias.cpu(), 0, 0)
            close(y, ref.permute(0, 2, 1), TOL[torch.bfloat16], f"coop vs torch B={B} L={Ln}")


def test_dual_source_k1_kernel_matches_two_launches():
    """ga_conv_dual_k1 (the 256-channel residual unit's closing sum, shortcut 256 -> 256 k1 + last conv 128 -> 256 k1 as one GEMM over
    the concatenated K, weights in the registers of an eight-wave workgroup) against the two ga_conv launches it replaces
    (the second with the first as residual, with and without ELU after the sum) and against torch fp32: bf16 accuracy (the
    one-launch form does not round the shortcut); row counts: less than a tile, ragged last tile, several tiles per workgroup."""
    torch.manual_seed(47)
    g_sc, g3 = ConvGeom("cconv", 256, 256, 1), ConvGeom("cconv", 128, 256, 1)
    wsc = (torch.randn(256, 256, 1) / 16.0).bfloat16().float()
    w3 = (torch.randn(256, 128, 1) / 11.3).bfloat16().float()
    bsc, b3 = torch.randn(256) * 0.2, torch.randn(256) * 0.2
    psc = ops.pack_frozen(g_sc, wsc.to(DEV), None, torch.bfloat16)[0]
    p3 = ops.pack_frozen(g3, w3.to(DEV), None, torch.bfloat16)[0]
    for B, Ln in ((1, 7), (3, 1501), (40, 3000)):
        x = torch.randn(B, Ln, 256).bfloat16()
        t = F.elu(torch.randn(B, Ln, 128)).bfloat16()
        assert ops.dual_k1_ok(x.to(DEV), g_sc, g3)
        for post in (False, True):
            y1 = ops.dual_k1_forward(x.to(DEV), t.to(DEV), psc, bsc.to(DEV), p3, b3.to(DEV), post).float().cpu()
            sc = ops.conv_forward(g_sc, x.to(DEV), psc, bsc.to(DEV))
            y2 = ops.conv_forward(g3, t.to(DEV), p3, b3.to(DEV), res=sc, post_act=L.ACT_ELU if post else 0, act_after_res=post).float().cpu()
            close(y1, y2, 1e-2, f"dual k1 vs two launches rows={B * Ln} elu={post}")
            if B * Ln <= 5000:
                ref = F.conv1d(x.float().permute(0, 2, 1), wsc, bsc) + F.conv1d(t.float().permute(0, 2, 1), w3, b3)
                ref = (F.elu(ref) if post else ref).permute(0, 2, 1)
                close(y1, ref, TOL[torch.bfloat16], f"dual k1 vs torch rows={B * Ln} elu={post}")


def test_fused_residual_unit_matches_three_node_form():
    """SeanetDecoder's residual units as one autograd node (ops.ResBlockFrozen: the shortcut's data-gradient launch adds the
    block's gradient in its epilogue) against the three-ConvFrozen-node form whose two gradients autograd sums: same forward to
    the last bit, gradient equal up to the one bf16 rounding the separate sum adds; and with ELU emitted by the producing
    epilogues (GA_RES_ADD_THEN_ACT, GA_DACT_ELU_FROM_OUT) to bf16 accuracy."""
    from gan_amd.lab3_codec_bridge import SeanetDecoder
    torch.manual_seed(35)
    dec = SeanetDecoder(precision="bf16").to(DEV)
    for p_ in dec.parameters():
        p_.data.normal_(0.0, 0.05)
    dec.invalidate_packs()
    q = torch.randn(2, 128, 9, device=DEV)
    outs = {}
    old = ops.FUSE_RESUNIT
    ops.FUSE_RESUNIT = False  # (the thin units' one-launch forward has a test of its own below; it rounds the shortcut once less)
    try:
        for mode in ("nodes", "fused", "fused+elu"):
            dec.fuse_resblock, dec.fuse_elu = mode != "nodes", mode == "fused+elu"
            qq = q.clone().requires_grad_(True)
            wav = dec(qq)
            g, = torch.autograd.grad(wav.float().square().sum(), qq)
            outs[mode] = (wav.detach().float().cpu(), g.float().cpu())
    finally:
        ops.FUSE_RESUNIT = old
    assert torch.equal(outs["fused"][0], outs["nodes"][0])
    close(outs["fused"][1], outs["nodes"][1], 2e-2, "fused residual unit dq")
    # ELU written by the producing epilogues (inside the wide units and in front of the transposed convs behind them): the
    # activation is taken from the fp32 accumulator instead of from its bf16 rounding, so equal only to bf16 accuracy
    close(outs["fused+elu"][0], outs["nodes"][0], 2e-2, "producer-side ELU wav")
    close(outs["fused+elu"][1], outs["nodes"][1], 3e-2, "producer-side ELU dq")


@pytest.mark.parametrize("c,lin,B", [(32, 3, 2), (32, 131, 3), (32, 4173, 2), (64, 5, 1), (64, 64, 2), (64, 2013, 3),
                                     (32, 120000, 2), (64, 60000, 3), (128, 3, 2), (128, 64, 1), (128, 65, 3), (128, 1501, 2),
                                     (128, 15000, 9)])
def test_one_launch_residual_unit_matches_three_launch_form_and_fp32_reference(c, lin, B):
    """ga_resunit_fwd (the thin residual units of the decoder, HF modeling_encodec.py:320-343, as one pass over x) against the
    three ga_conv launches it replaces and against torch fp32 on the same bf16-rounded weights: forward to bf16 accuracy
    (the one-launch form keeps the shortcut in fp32 where the three-launch form rounds it to bf16 before the sum), the hidden
    tensor it saves for the backward equal to ELU of the three-launch hidden tensor to one bf16 step, and the input gradient of
    the whole unit (which runs the same three data-gradient launches from that saved tensor) to bf16 accuracy.  Lengths: the
    reflect padding alone (3, 5), one partial tile, several tiles with a ragged end, and the bench's item lengths."""
    torch.manual_seed(c + lin)
    g_sc, g1, g3 = ConvGeom("cconv", c, c, 1), ConvGeom("cconv", c, c // 2, 3), ConvGeom("cconv", c // 2, c, 1)
    ws = {n: (torch.randn(g.cout, g.cin, g.k) * (1.5 / (g.cin * g.k) ** 0.5)).bfloat16().float() for n, g in (("sc", g_sc), ("b1", g1), ("b3", g3))}
    bs = {n: torch.randn(g.cout) * 0.1 for n, g in (("sc", g_sc), ("b1", g1), ("b3", g3))}
    packs = {n: ops.pack_frozen(g, ws[n].to(DEV), None, torch.bfloat16) for n, g in (("sc", g_sc), ("b1", g1), ("b3", g3))}
    bd = {n: b.to(DEV) for n, b in bs.items()}
    x = torch.randn(B, lin, c).bfloat16()
    gy = torch.randn(B, lin, c).bfloat16()

    def run(fuse, fuse_bwd=False):
        old = ops.FUSE_RESUNIT, ops.FUSE_RESUNIT_BWD
        ops.FUSE_RESUNIT, ops.FUSE_RESUNIT_BWD = fuse, fuse_bwd
        try:
            xx = x.to(DEV).requires_grad_(True)
            y = ops.ResBlockFrozen.apply(xx, *packs["sc"], bd["sc"], g_sc, *packs["b1"], bd["b1"], g1, *packs["b3"], bd["b3"], g3, False, False)
            hidden = y.grad_fn.saved_tensors[1]
            gx, = torch.autograd.grad(y, xx, gy.to(DEV))
            with torch.no_grad():  # the frozen items' call: no hidden tensor is written
                y0 = ops.ResBlockFrozen.apply(x.to(DEV), *packs["sc"], bd["sc"], g_sc, *packs["b1"], bd["b1"], g1, *packs["b3"], bd["b3"], g3, False, False)
            assert torch.equal(y0, y)
            return y.detach().float().cpu(), gx.float().cpu(), hidden.detach().float().cpu()
        finally:
            ops.FUSE_RESUNIT, ops.FUSE_RESUNIT_BWD = old

    y1, gx1, h1 = run(True)
    y3, gx3, h3 = run(False)
    # ga_resunit_bwd: the unit's input gradient in one launch (hidden gradient, three taps, the reflected pad's rows folded back,
    # ELU'(x), shortcut) against the three data-gradient launches + fix-up launch on the same saved tensors
    y1b, gx1b, _ = run(True, True)
    assert torch.equal(y1b, y1)
    if c == 128:  # the unit that hands ELU(output) to the transposed conv behind it: post_elu against GA_RES_ADD_THEN_ACT
        outs = []
        for fuse in (True, False):
            old = ops.FUSE_RESUNIT
            ops.FUSE_RESUNIT = fuse
            try:
                with torch.no_grad():
                    outs.append(ops.ResBlockFrozen.apply(x.to(DEV), *packs["sc"], bd["sc"], g_sc, *packs["b1"], bd["b1"], g1, *packs["b3"], bd["b3"], g3,
                                                         True, True).float().cpu())
            finally:
                ops.FUSE_RESUNIT = old
        close(outs[0], outs[1], 1e-2, "one-launch 128-channel unit with ELU on the output")
        close(outs[0], F.elu(y1), 1e-2, "ELU(output) vs ELU of the plain output")
    close(gx1b, gx1, 1e-2, "one-launch input gradient vs three launches")
    assert h1.shape == (B, lin, c // 2)
    href = F.elu(h3).bfloat16().float()  # (exp2-based ELU in the kernel, expm1 here: one bf16 step apart at most)
    # (C = 128: the three-launch form's first conv runs on the gathered-GEMM kernel, whose fp32 sums go in another K order, so a
    # few hidden values in ten million round to the neighbouring bf16 before the ELU: two steps there)
    assert ((h1 - href).abs() <= 2.0 ** (-6 if c == 128 else -7) * href.abs().clamp_min(1e-3)).all()
    close(y1, y3, 1e-2, "one-launch unit vs three launches, y")
    close(gx1, gx3, 1e-2, "one-launch unit vs three launches, dx")
    if lin <= 5000:
        xr = x.float().permute(0, 2, 1).requires_grad_(True)
        t = _ref_conv(g1, xr, ws["b1"], bs["b1"], L.ACT_ELU, 0)
        yr = _ref_conv(g_sc, xr, ws["sc"], bs["sc"], 0, 0) + _ref_conv(g3, t, ws["b3"], bs["b3"], L.ACT_ELU, 0)
        gr, = torch.autograd.grad(yr, xr, gy.float().permute(0, 2, 1))
        close(y1, yr.permute(0, 2, 1), TOL[torch.bfloat16], "one-launch unit vs torch fp32, y")
        close(gx1, gr.permute(0, 2, 1), TOL[torch.bfloat16], "one-launch unit vs torch fp32, dx")
        close(gx1b, gr.permute(0, 2, 1), TOL[torch.bfloat16], "one-launch input gradient vs torch fp32")
        # closer to the fp32 result than the three-launch form (one rounding less): not asserted, both are inside the bf16 bound


@pytest.mark.parametrize("lin,B", [(13, 2), (130, 3), (500, 1), (4173, 2), (120000, 2)])
def test_no_gradient_unit_and_closing_conv_in_one_launch_match_separate_launches(lin, B):
    """ga_resunit_fwd's tail form (the last residual unit + the decoder's closing ELU + conv 32 -> 1 k7 in one launch, for
    callers that want no gradient: the unit's output never leaves the chip) against ga_resunit_fwd followed by the closing conv's
    own ga_conv launch: the same bits where that launch runs conv_swin1_kernel (same rounding points, same MFMA chains: the
    120000-sample case, with a ragged last tile), 1e-5 where short launches take thin.hip's vector-ALU kernel (another fp32
    summation order); and against torch fp32.  13 is the shortest length (reflect rows 1..6 inside the first tile)."""
    torch.manual_seed(lin)
    c = 32
    geoms = {"sc": ConvGeom("cconv", c, c, 1), "b1": ConvGeom("cconv", c, c // 2, 3), "b3": ConvGeom("cconv", c // 2, c, 1),
             "last": ConvGeom("cconv", c, 1, 7)}
    ws = {n: (torch.randn(g.cout, g.cin, g.k) * (1.5 / (g.cin * g.k) ** 0.5)).bfloat16().float() for n, g in geoms.items()}
    bs = {n: torch.randn(g.cout) * 0.1 for n, g in geoms.items()}
    pk = {n: ops.pack_frozen(g, ws[n].to(DEV), None, torch.bfloat16)[0] for n, g in geoms.items()}
    bd = {n: b.to(DEV) for n, b in bs.items()}
    x = torch.randn(B, lin, c).bfloat16()
    xd = x.to(DEV)
    args = (xd, pk["sc"], bd["sc"], pk["b1"], bd["b1"], pk["b3"], bd["b3"])
    wav1, none = ops.resunit_forward(*args, want_t=False, last=(pk["last"], bd["last"]))
    y, _ = ops.resunit_forward(*args, want_t=False)
    wav2 = ops.conv_forward(geoms["last"], y, pk["last"], bd["last"], pre_act=L.ACT_ELU, out_f32=True)
    assert none is None and wav1.shape == (B, lin, 1) and wav1.dtype == torch.float32
    if B * lin >= 2 ** 17:
        assert torch.equal(wav1, wav2)
    close(wav1, wav2.cpu(), 1e-5, "tail form vs unit + closing conv")
    if lin <= 5000:
        xr = x.float().permute(0, 2, 1)
        t = _ref_conv(geoms["b1"], xr, ws["b1"], bs["b1"], L.ACT_ELU, 0)
        yr = _ref_conv(geoms["sc"], xr, ws["sc"], bs["sc"], 0, 0) + _ref_conv(geoms["b3"], t, ws["b3"], bs["b3"], L.ACT_ELU, 0)
        wr = _ref_conv(geoms["last"], yr, ws["last"], bs["last"], L.ACT_ELU, 0)
        close(wav1, wr.permute(0, 2, 1), TOL[torch.bfloat16], "tail form vs torch fp32")


@pytest.mark.parametrize("T,B", [(1, 2), (9, 3), (375, 2)])
def test_decoder_without_gradient_uses_the_tail_form_and_matches(T, B):
    """SeanetDecoder under no_grad ends with the one-launch tail; same waveform as with it switched off and as the gradient path."""
    from gan_amd.lab3_codec_bridge import SeanetDecoder
    torch.manual_seed(T)
    dec = SeanetDecoder(precision="bf16").to(DEV)
    for p_ in dec.parameters():
        p_.data.normal_(0.0, 0.12)
    dec.invalidate_packs()
    q = torch.randn(B, 128, T, device=DEV)
    outs = []
    for tail in (True, False):
        dec.fuse_tail = tail
        with torch.no_grad():
            outs.append(dec(q).float().cpu())
    assert outs[0].shape == (B, 1, T * 320)
    close(outs[0], outs[1], 1e-5, "decoder with / without the tail form")
    wav = dec(q.clone().requires_grad_(True))
    assert torch.equal(wav.detach().float().cpu(), outs[1])


def test_second_elu_output_of_the_gathered_gemm_epilogue():
    """ga_conv_params.y2: the decoder's transposed convs 512->256 (k16 s8) and 256->128 (k10 s5) write ELU(y) next to y for the
    residual unit behind them (shortcut reads y, first conv reads ELU(y)): y unchanged to the last bit, y2 = ELU of the stored
    bf16 value to one bf16 step (exp2-based ELU); the decoder's waveform and input gradient with the second output on / off."""
    torch.manual_seed(77)
    for g, lin, B in ((ConvGeom("convt", 256, 128, 10, 5), 601, 3), (ConvGeom("convt", 512, 256, 16, 8), 75, 2), (ConvGeom("cconv", 128, 128, 1), 999, 2)):
        w = torch.randn(*g.wdims) * (1.5 / (g.cin * (g.k if g.kind != "convt" else 2)) ** 0.5)
        b = (torch.randn(g.cout) * 0.3).to(DEV)
        wp = ops.pack_frozen(g, w.to(DEV), None, torch.bfloat16)[0]
        x = torch.randn(B, lin, g.cin, device=DEV).bfloat16()
        assert ops.conv_second_output_ok(g, x)
        y0 = ops.conv_forward(g, x, wp, b)
        y, y2 = ops.conv_forward(g, x, wp, b, elu_out2=True)
        assert torch.equal(y, y0) and y2.shape == y.shape and y2.dtype == torch.bfloat16
        ref = F.elu(y.float()).bfloat16().float().cpu()
        assert ((y2.float().cpu() - ref).abs() <= 2.0 ** -7 * ref.abs().clamp_min(1e-3)).all()
    assert not ops.conv_second_output_ok(ConvGeom("cconv", 128, 128, 1), torch.empty(2, 99, 128, device=DEV))  # fp32: no such epilogue
    from gan_amd.lab3_codec_bridge import SeanetDecoder
    dec = SeanetDecoder(precision="bf16").to(DEV)
    for p_ in dec.parameters():
        p_.data.normal_(0.0, 0.05)
    dec.invalidate_packs()
    q = torch.randn(2, 128, 9, device=DEV)
    outs = {}
    for two in (False, True):
        dec.fuse_elu2 = two
        qq = q.clone().requires_grad_(True)
        wav = dec(qq)
        gq, = torch.autograd.grad(wav.float().square().sum(), qq)
        outs[two] = (wav.detach().float().cpu(), gq.float().cpu())
    close(outs[True][0], outs[False][0], 1e-2, "second ELU output, wav")
    close(outs[True][1], outs[False][1], 2e-2, "second ELU output, dq")


def test_pack_cache_is_released_with_its_optimizer():
    """Packed weights are cached per optimizer epoch; when the FusedAdamW that owns the parameter buffer is dropped (one per
    train_codec_stage call), its packs and its epoch range go with it instead of staying resident for the life of the process."""
    import gc
    from gan_amd.lab3_codec_models import MultiScaleWaveDiscriminator
    from gan_amd.optim import FusedAdamW
    D = MultiScaleWaveDiscriminator(1, precision="bf16").to(DEV)
    opt = FusedAdamW(D.parameters(), lr=1e-3)
    lo, hi = opt.flat_p.data_ptr(), opt.flat_p.data_ptr() + 4 * opt.numel
    D(torch.randn(2, 1, 2048, device=DEV))
    assert any(lo <= k[0] < hi for k in ops._PACK_CACHE) and any(r[0] == lo for r in ops._EPOCH_RANGES)
    del opt
    gc.collect()
    assert not any(lo <= k[0] < hi for k in ops._PACK_CACHE) and not any(r[0] == lo for r in ops._EPOCH_RANGES)
    y1 = D(torch.ones(2, 1, 2048, device=DEV))[0][0]  # still works without an optimizer (packs rebuilt per call, not cached)
    with torch.no_grad():
        for p in D.parameters():
            p.mul_(0.5)
    y2 = D(torch.ones(2, 1, 2048, device=DEV))[0][0]
    assert not torch.equal(y1, y2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_grouped_reduction_matches_one_launch_per_pair(dtype):
    """ga_reduce_multi (the feature-matching sums of one discriminator scale in one launch) against ga_reduce per pair and against
    torch: sizes from 3 elements to 20 M, an odd (unaligned-tail) size, an empty entry."""
    torch.manual_seed(0)
    dev = torch.device("cuda")
    sizes = [20_000_003, 3_000_000, 65_536, 1001, 3, 0, 777_777]
    aa = [torch.randn(n, device=dev).to(dtype) for n in sizes]
    bb = [torch.randn(n, device=dev).to(dtype) for n in sizes]
    scales = [1.0 / max(1, n) / (i + 1) for i, n in enumerate(sizes)]
    out = torch.zeros(1, device=dev)
    ops.reduce_absdiff_multi(aa, bb, scales, out)
    each = torch.zeros(1, device=dev)
    for a, b, sc in zip(aa, bb, scales):
        if a.numel():  # (an empty tensor has no pointer to hand to the single-problem entry point)
            L.check(L.lib().ga_reduce(a.data_ptr(), b.data_ptr(), L.dt(a), a.numel(), L.RED_ABSDIFF, float(sc), each.data_ptr(), L.stream()), "ga_reduce")
    want = sum(sc * (a.double() - b.double()).abs().sum().item() for a, b, sc in zip(aa, bb, scales))
    assert abs(float(out) - want) <= 2e-5 * abs(want) and abs(float(out) - float(each)) <= 2e-6 * abs(want)
    k = 9
    pa = (C.c_void_p * k)(*[aa[0].data_ptr()] * k)
    pn = (C.c_size_t * k)(*[8] * k)
    ps = (C.c_float * k)(*[1.0] * k)
    assert L.lib().ga_reduce_multi(C.cast(pa, C.c_void_p), C.cast(pa, C.c_void_p), C.cast(pn, C.c_void_p), C.cast(ps, C.c_void_p), k, L.dt(aa[0]),
                                   L.RED_ABSDIFF, out.data_ptr(), L.stream()) == 1  # more than 8 entries: GA_ERR_ARG
