"""GPU: ga_conv_splitk_static (the entry point that allocates nothing) against ga_conv + ga_conv_finish with a caller
workspace, through the C ABI.  Same kernels, same slice order, so the outputs must be bit-identical; two calls in a row,
the second on a smaller problem (the scratch array keeps the first call's contents)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

from gan_amd import _lib as L
from gan_amd import ops
from gan_amd.ops import ConvGeom

DEV = "cuda"


def _problem(g, B, Ln, seed):
    torch.manual_seed(seed)
    x = torch.randn(B, Ln, g.cin, device=DEV).bfloat16()
    w = torch.randn(*g.wdims, device=DEV) / (g.cin * g.k) ** 0.5
    bias = torch.randn(g.cout, device=DEV)
    wf = ops.pack_weight(g, w, g.fwd_pack, torch.bfloat16)
    p = ops.fwd_params(g, B, Ln)
    p.post_act = L.ACT_LRELU
    p.x, p.w, p.bias = x.data_ptr(), wf.data_ptr(), bias.data_ptr()
    return p, (x, wf, bias)


def _with_workspace(p, ks):
    y = torch.full((p.B, p.LoutTot, p.M), float("nan"), device=DEV, dtype=torch.bfloat16)
    ws = torch.full((ks * p.B * p.LoutTot * p.M,), float("nan"), device=DEV, dtype=torch.float32)
    q = L.ConvParams()
    ctypes.memmove(ctypes.byref(q), ctypes.byref(p), ctypes.sizeof(L.ConvParams))
    q.y, q.ksplit, q.ws = y.data_ptr(), ks, ws.data_ptr()
    L.check(L.lib().ga_conv(ctypes.byref(q), L.DT_BF16, L.IMPL_AUTO, L.stream()), "ga_conv")
    L.check(L.lib().ga_conv_finish(ctypes.byref(q), L.DT_BF16, L.stream()), "ga_conv_finish")
    return y


def _static(p, ks):
    y = torch.full((p.B, p.LoutTot, p.M), float("nan"), device=DEV, dtype=torch.bfloat16)
    q = L.ConvParams()
    ctypes.memmove(ctypes.byref(q), ctypes.byref(p), ctypes.sizeof(L.ConvParams))
    q.y, q.ksplit, q.ws = y.data_ptr(), ks, None
    # "allocates nothing": the device's free memory as the driver reports it (hipMemGetInfo - it sees a hipMalloc made inside
    # libgan_amd.so, which torch's own allocator statistics would not), read around the call with the stream drained.  The
    # first call on a device looks up the scratch array's address (hipGetSymbolAddress), so callers warm that up first.
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    before = torch.cuda.memory_allocated()
    rc = L.lib().ga_conv_splitk_static(ctypes.byref(q), L.DT_BF16, L.stream())
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    if _static.warm:
        assert torch.cuda.mem_get_info()[0] == free_before, (free_before, torch.cuda.mem_get_info()[0])
    _static.warm = True
    return rc, y


_static.warm = False


def test_static_scratch_split_k_matches_workspace_split_k_twice():
    big = (ConvGeom("conv", 256, 512, 41, 4, 20, 1), 3, 469, 5)    # D blocks.4: 3 x 118 x 512 x ks 5 = 0.9 M floats
    small = (ConvGeom("conv", 128, 256, 41, 4, 20, 1), 2, 301, 3)  # smaller, different shape: stale scratch must not show
    p, keep = _problem(small[0], 1, 301, seed=1)
    assert _static(p, 2)[0] == 0  # first call on the device: symbol lookup; the free-memory check holds from the next call on
    for g, B, Ln, ks in (big, small):
        p, keep = _problem(g, B, Ln, seed=41 + B)
        want = _with_workspace(p, ks)
        rc, got = _static(p, ks)
        assert rc == 0, rc
        torch.cuda.synchronize()
        assert not torch.isnan(got.float()).any()
        assert torch.equal(got, want), (g, (got.float() - want.float()).abs().max().item())
    # library-chosen split (ksplit = 0) agrees with the single-pass kernel to bf16 rounding
    p, keep = _problem(big[0], 3, 469, seed=7)
    rc, got = _static(p, 0)
    assert rc == 0
    y1 = torch.empty_like(got)
    q = L.ConvParams()
    ctypes.memmove(ctypes.byref(q), ctypes.byref(p), ctypes.sizeof(L.ConvParams))
    q.y = y1.data_ptr()
    L.check(L.lib().ga_conv(ctypes.byref(q), L.DT_BF16, L.IMPL_AUTO, L.stream()), "ga_conv")
    torch.cuda.synchronize()
    err = (got.float() - y1.float()).abs().max().item() / y1.float().abs().max().item()
    assert err <= 1e-2, err


def test_static_scratch_rejects_oversize_and_caller_workspace():
    g = ConvGeom("conv", 256, 512, 41, 4, 20, 1)
    p, keep = _problem(g, 24, 1875, seed=3)       # 24 x 469 x 512 x 2 slices = 11.5 M floats > 8 Mi
    assert 2 * p.B * p.LoutTot * p.M > L.SPLITK_STATIC_FLOATS
    rc, y = _static(p, 2)
    assert rc == 5, rc                             # GA_ERR_SCRATCH, nothing launched
    torch.cuda.synchronize()
    assert torch.isnan(y.float()).all()
    with pytest.raises(RuntimeError, match="GA_ERR_SCRATCH"):
        L.check(rc, "ga_conv_splitk_static")
    p, keep = _problem(g, 2, 469, seed=3)
    ws = torch.empty(16, device=DEV)
    p.y, p.ksplit, p.ws = torch.empty(2, p.LoutTot, p.M, device=DEV, dtype=torch.bfloat16).data_ptr(), 2, ws.data_ptr()
    assert L.lib().ga_conv_splitk_static(ctypes.byref(p), L.DT_BF16, L.stream()) == 1


def test_static_scratch_call_is_capturable_in_a_hip_graph():
    """Second check of "allocates nothing": the call is recorded under stream capture (an allocation, a free or a synchronising
    call inside the library would fail the capture) and the replayed graph gives the same bits as the eager call."""
    g = ConvGeom("conv", 128, 256, 41, 4, 20, 1)
    p, keep = _problem(g, 2, 301, seed=9)
    rc, want = _static(p, 3)  # eager (and warm-up: symbol lookup, function attributes)
    assert rc == 0
    y = torch.full((p.B, p.LoutTot, p.M), float("nan"), device=DEV, dtype=torch.bfloat16)
    q = L.ConvParams()
    ctypes.memmove(ctypes.byref(q), ctypes.byref(p), ctypes.sizeof(L.ConvParams))
    q.y, q.ksplit, q.ws = y.data_ptr(), 3, None
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        rc = L.lib().ga_conv_splitk_static(ctypes.byref(q), L.DT_BF16, L.stream())
    assert rc == 0, rc
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, want)
