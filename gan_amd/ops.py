"""Autograd operators of the lab3 codec GAN step, each a thin wrapper over the C ABI (include/gan_amd.h).

All activations are channels-last [B, L, C] ("NLC") CUDA tensors in float32 or bfloat16; parameters,
their gradients and loss sums are float32.  PyTorch supplies device memory, streams and the autograd
tape - every O(N) computation happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib as L

Tensor = torch.Tensor


# ================================================================================================
# geometry of one Conv1d / ConvTranspose1d layer and its three GEMM forms
@dataclass(frozen=True)
class ConvGeom:
    """kind 'conv': Conv1d(cin, cout, k, stride, padding, dilation), zero padding
       kind 'cconv': causal reflect-padded stride-1 conv (HF EncodecConv1d, modeling_encodec.py:157-176)
       kind 'convt': causal ConvTranspose1d(cin, cout, k=2*stride, stride) with the right trim (:206-233)"""
    kind: str
    cin: int
    cout: int
    k: int
    stride: int = 1
    padding: int = 0
    dilation: int = 1

    def out_len(self, lin: int) -> int:
        if self.kind == "conv":
            return (lin + 2 * self.padding - self.dilation * (self.k - 1) - 1) // self.stride + 1
        if self.kind == "cconv":
            return lin
        return lin * self.stride

    # weight source dims (D0, D1, K) as stored in the parameter
    @property
    def wdims(self) -> Tuple[int, int, int]:
        return (self.cin, self.cout, self.k) if self.kind == "convt" else (self.cout, self.cin, self.k)

    @property
    def fwd_pack(self) -> int:
        return L.PACK_CONVT_FWD if self.kind == "convt" else L.PACK_CONV_FWD

    @property
    def dgrad_pack(self) -> int:
        if self.kind == "convt":
            return L.PACK_CONVT_DGRAD
        return L.PACK_CONV_DGRAD_S1 if self.stride == 1 else L.PACK_CONV_DGRAD_STRIDED


def _fill(p: L.ConvParams, **kw) -> L.ConvParams:
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _new_params() -> L.ConvParams:
    p = L.ConvParams()
    p.nphase = 1
    p.out_stride = 1
    p.in_stride = 1
    p.tap_step = 1
    return p


def fwd_params(g: ConvGeom, B: int, lin: int) -> L.ConvParams:
    p = _new_params()
    lout = g.out_len(lin)
    if g.kind == "conv":
        _fill(p, B=B, Lin=lin, Cin=g.cin, M=g.cout, Lout=lout, LoutTot=lout, J=g.k, in_stride=g.stride,
              tap_step=g.dilation, off0=-g.padding)
    elif g.kind == "cconv":
        _fill(p, B=B, Lin=lin, Cin=g.cin, M=g.cout, Lout=lout, LoutTot=lout, J=g.k, in_stride=1,
              tap_step=g.dilation, off0=-(g.k - 1) * g.dilation, pad_mode=1)
    else:
        s = g.stride
        if s > L.GA_MAX_PHASE:
            raise ValueError("transposed-conv stride > 8 unsupported")
        _fill(p, B=B, Lin=lin, Cin=g.cin, M=g.cout, Lout=lin, LoutTot=lout, J=2, in_stride=1, tap_step=1, off0=-1,
              out_stride=s, nphase=s)
    return p


def dgrad_params(g: ConvGeom, B: int, lin: int) -> L.ConvParams:
    """Forward-form problem computing dx [B, lin, cin] from the output gradient [B, lout, cout]."""
    p = _new_params()
    lout = g.out_len(lin)
    if g.kind == "conv" and g.stride == 1:
        _fill(p, B=B, Lin=lout, Cin=g.cout, M=g.cin, Lout=lin, LoutTot=lin, J=g.k, tap_step=g.dilation,
              off0=g.padding - (g.k - 1) * g.dilation)
    elif g.kind == "conv":
        if g.dilation != 1:
            raise ValueError("strided conv with dilation unsupported")
        s = g.stride
        if s > L.GA_MAX_PHASE:
            raise ValueError("conv stride > 8 unsupported")
        jmax = (g.k + s - 1) // s
        _fill(p, B=B, Lin=lout, Cin=g.cout, M=g.cin, Lout=(lin + s - 1) // s, LoutTot=lin, J=jmax, off0=-(jmax - 1),
              out_stride=s, nphase=s)
        for r in range(s):
            p.phase_off0[r] = (r + g.padding) // s
    elif g.kind == "cconv":
        _fill(p, B=B, Lin=lout, Cin=g.cout, M=g.cin, Lout=lin, LoutTot=lin, J=g.k, tap_step=g.dilation, off0=0)
    else:
        _fill(p, B=B, Lin=lout, Cin=g.cout, M=g.cin, Lout=lin, LoutTot=lin, J=g.k, in_stride=g.stride, off0=0)
    return p


def _launch_conv(p: L.ConvParams, x: Tensor, w: Tensor, y: Tensor, bias=None, res=None, dact_src=None, impl=L.IMPL_AUTO,
                 alg_flops: Optional[float] = None):
    """alg_flops: the ALGORITHMIC multiply-add flops of this launch when they differ from the launched ones (a strided data
    gradient runs ceil(k/s)*s taps for k, the bf16 MR-STFT GEMM runs a 3x-deep K for its hi/lo split); benchmark accounting."""
    L.require_cuda(x, w, y)
    assert x.is_contiguous() and w.is_contiguous() and y.is_contiguous()
    assert x.dtype == w.dtype, (x.dtype, w.dtype)
    if p.out_f32 and x.dtype == torch.bfloat16:
        assert y.dtype == torch.float32 and res is None and dact_src is None
    else:
        assert y.dtype == x.dtype, (x.dtype, y.dtype)
    assert x.numel() == p.B * p.Lin * p.Cin, (x.shape, p.B, p.Lin, p.Cin)
    assert y.numel() == p.B * p.LoutTot * p.M, (y.shape, p.B, p.LoutTot, p.M)
    assert w.numel() >= p.nphase * p.M * p.J * p.Cin
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == p.M
    for t in (res, dact_src):
        if t is not None:
            assert t.is_contiguous() and t.dtype == y.dtype and t.numel() == y.numel()
    p.x, p.w, p.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    p.bias, p.res, p.dact_src = L.ptr(bias), L.ptr(res), L.ptr(dact_src)
    if MERGE_PHASES and impl in (L.IMPL_AUTO, L.IMPL_MFMA, L.IMPL_MFMA_PP, L.IMPL_MFMA_LDSWIN, L.IMPL_MFMA_STREAM, L.IMPL_MFMA_STREAMWIN):
        p, bias = _fold_phases(p, x, bias)
    # few output tiles x deep K (D blocks 3-4): split K over workgroups, one fp32 partial-sum slab per slice
    ks = L.lib().ga_conv_suggest_ksplit(C.byref(p), L.dt(x), impl) if SPLIT_K else 1
    if FORCE_KSPLIT is not None and L.lib().ga_conv_kernel_id(C.byref(p), L.dt(x), impl) in (7, 8, 9, 11, 12):
        ks = max(1, min(int(FORCE_KSPLIT), (p.J * p.Cin + 63) // 64))
    ws = None
    if ks > 1:
        ws = torch.empty((ks * p.B * p.LoutTot * p.M,), device=x.device, dtype=torch.float32)
        p.ksplit, p.ws = ks, ws.data_ptr()
    if PROFILE is None:
        L.check(L.lib().ga_conv(C.byref(p), L.dt(x), impl, L.stream()), "ga_conv")
        if ks > 1:
            L.check(L.lib().ga_conv_finish(C.byref(p), L.dt(x), L.stream()), "ga_conv_finish")
        return
    # benchmark accounting: HIP events on the launch stream around this one launch + its multiply-adds as launched
    kid = L.lib().ga_conv_kernel_id(C.byref(p), L.dt(x), impl)
    if kid in (7, 8, 9) and impl in (L.IMPL_AUTO, L.IMPL_MFMA):
        # the gathered-GEMM instantiations are different kernels to a profiler: account for them one by one (kid >= 0x1000)
        var = L.lib().ga_conv_gemm_variant(C.byref(p))
        if var:
            kid = var
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(L.lib().ga_conv(C.byref(p), L.dt(x), impl, L.stream()), "ga_conv")
    if ks > 1:
        L.check(L.lib().ga_conv_finish(C.byref(p), L.dt(x), L.stream()), "ga_conv_finish")
    e1.record()
    launched = 2.0 * p.B * p.nphase * p.Lout * p.M * p.J * p.Cin
    PROFILE.append((kid, launched, e0, e1, (p.B, p.Cin, p.M, p.J, p.Lout, p.nphase, p.in_stride, p.pre_act),
                    launched if alg_flops is None else float(alg_flops)))


_BIAS_REP: dict = {}


def _fold_phases(p: L.ConvParams, x: Tensor, bias):
    """Output phases folded into the channel dimension.  A strided conv's data gradient and a transposed conv's forward
    are `nphase` problems that read the SAME input rows (all phase offsets equal) with different weight slices and write
    interleaved output rows; since y[t*S + r][m] is element (r*M + m) of a row of S*M channels, they are also ONE problem
    with M' = S*M, out_stride 1 and the packed weights unchanged ([S][M][J][Cin] == [S*M][J][Cin]).  One pass over the
    input instead of S, and 128-wide tiles for the 32/64-channel layers.  Only the MFMA GEMM kernel has the per-item
    element bound this needs when the item length is not a multiple of S."""
    S = int(p.nphase)
    if S <= 1 or p.out_stride != S or p.t0 != 0 or x.dtype != torch.bfloat16 or p.ksplit > 1:
        return p, bias
    if any(p.phase_off0[r] != p.phase_off0[0] for r in range(S)):
        return p, bias
    q = L.ConvParams()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(L.ConvParams))
    q.M = p.M * S
    q.nphase, q.out_stride = 1, 1
    q.off0 = p.off0 + p.phase_off0[0]
    for r in range(L.GA_MAX_PHASE):
        q.phase_off0[r] = 0
    q.LoutTot = (p.LoutTot + S - 1) // S
    q.y_batch_elems = p.LoutTot * p.M
    if q.y_batch_elems % 8 or q.y_batch_elems >= 2 ** 31:
        return p, bias
    if bias is not None:
        # cache entries hold the bias tensor itself, so neither its id nor its address can be reused while cached
        key = (id(bias), S)
        ent = _BIAS_REP.get(key)
        if ent is None or ent[0] is not bias or ent[1] != bias._version:
            if len(_BIAS_REP) > 256:
                _BIAS_REP.clear()
            ent = _BIAS_REP[key] = (bias, bias._version, bias.detach().repeat(S).contiguous())
        q.bias = ent[2].data_ptr()
    if L.lib().ga_conv_kernel_id(C.byref(q), L.dt(x), L.IMPL_MFMA) not in L.GEMM_KERNELS:
        return p, bias
    return q, bias


MERGE_PHASES = True  # tests switch this off to compare against the phase-by-phase launches
_IMPL = L.IMPL_AUTO
WGRAD_SLABS = False  # True: per-slice slabs + a reduce kernel instead of fp32 atomics on dw (measured no faster; tests run both)
FORCE_KSPLIT = None  # tuning aid (tools/bench_conv.py): this split for every gathered-GEMM launch
SPLIT_K = True  # tests switch this off to compare split-K against the single-pass kernel
PROFILE = None  # set to a list by bench.py to collect (kernel id, launched flops, start event, end event)


def set_conv_impl(impl: int) -> None:
    """Force the convolution implementation (tests use this to cross-check MFMA against direct)."""
    global _IMPL
    _IMPL = impl


# ---- single-input-channel layers (the discriminators' first conv on the waveform) ---------------------------------
# Their MFMA kernels read 8 consecutive samples per 16-byte piece, so zero padding cannot be applied piece-wise: the input
# is copied once into a zero-padded buffer (a few MB) in which every window is in range, and the problem is restated on it.
CIN1_MFMA = True
_CIN1_BACK = 64      # zero samples behind each item (>= 32-step round-up + 16 taps + one 8-sample piece)
_CIN1_CACHE: list = []  # [(x, version, padded)] most recent first; the forward's copy serves the weight gradient too


def _cin1_padded(g: ConvGeom, x: Tensor):
    """(padded x, front) when the single-channel MFMA kernels apply to conv layer g on x, else (None, 0)."""
    if not (CIN1_MFMA and g.kind == "conv" and g.cin == 1 and g.dilation == 1 and g.stride == 1 and g.k <= 15
            and g.cout % 16 == 0 and g.cout <= 64 and x.dtype == torch.bfloat16 and _IMPL in (L.IMPL_AUTO, L.IMPL_MFMA)):
        return None, 0
    key = (x.data_ptr(), x._version, tuple(x.shape), g.padding)
    for ent in _CIN1_CACHE:  # (an entry keeps its source tensor alive, so its address cannot be reused while cached)
        if ent[1] == key:
            return ent[2], g.padding
    xp = torch.nn.functional.pad(x, (0, 0, g.padding, _CIN1_BACK))
    _CIN1_CACHE.insert(0, (x, key, xp))
    del _CIN1_CACHE[4:]
    return xp, g.padding


def conv_forward(g: ConvGeom, x: Tensor, wp: Tensor, bias: Optional[Tensor], pre_act=0, post_act=0, res=None,
                 out_f32: bool = False, act_after_res: bool = False, elu_out2: bool = False):
    """act_after_res: y = post_act(conv + bias + res) instead of post_act(conv + bias) + res (GA_RES_ADD_THEN_ACT).
    elu_out2: returns (y, ELU(y)), the second written by the same epilogue (ga_conv_params.y2; check conv_second_output_ok)."""
    B, lin, cin = x.shape
    assert cin == g.cin, (cin, g.cin)
    p = fwd_params(g, B, lin)
    p.pre_act, p.post_act = pre_act, post_act
    if act_after_res:
        assert res is not None and x.dtype == torch.bfloat16
        p.res_mode = L.RES_ADD_THEN_ACT
    p.out_f32 = 1 if (out_f32 and x.dtype == torch.bfloat16) else 0
    y = torch.empty((B, p.LoutTot, g.cout), device=x.device, dtype=torch.float32 if p.out_f32 else x.dtype)
    y2 = None
    if elu_out2:
        assert x.dtype == torch.bfloat16 and not p.out_f32
        y2 = torch.empty_like(y)
        p.y2 = y2.data_ptr()
    if not pre_act and res is None and not p.out_f32 and y2 is None:
        xp, front = _cin1_padded(g, x)
        if xp is not None:  # the same problem on the zero-padded copy: no row is ever out of range
            p.Lin, p.off0 = xp.shape[1], p.off0 + front
            x = xp
    _launch_conv(p, x, wp, y, bias=bias, res=res, impl=_IMPL)
    return y if y2 is None else (y, y2)


def conv_second_output_ok(g: ConvGeom, x: Tensor) -> bool:
    """True when ga_conv can write ELU(y) next to y for this forward (the gathered-GEMM kernel's epilogue, ga_conv_params.y2)."""
    if x.dtype != torch.bfloat16 or _IMPL not in (L.IMPL_AUTO, L.IMPL_MFMA):
        return False
    p = fwd_params(g, x.shape[0], x.shape[1])
    p.x = p.w = p.y = p.y2 = 256  # alignment checks only: nothing is dereferenced
    if MERGE_PHASES:
        p, _ = _fold_phases(p, x, None)
    return L.lib().ga_conv_kernel_id(C.byref(p), L.dt(x), _IMPL) in (7, 8, 9)


def conv_dgrad(g: ConvGeom, gy: Tensor, wp_d: Tensor, lin: int, dact_mode=0, dact_src=None, res=None, fm_coef=None) -> Tensor:
    """fm_coef = (scale, device scalar): `res` is the real feature map paired with the fake one in dact_src and the
    epilogue adds scale * scalar * sign(dact_src - res) before the mask (GA_RES_FM_SIGN; check fm_sign_fusable first)."""
    B = gy.shape[0]
    p = dgrad_params(g, B, lin)
    p.dact_mode = dact_mode
    if fm_coef is not None:
        assert g.kind == "conv" and res is not None and dact_src is not None and dact_mode
        p.res_mode, p.res_coef, p.res_coef_dev = L.RES_FM_SIGN, float(fm_coef[0]), fm_coef[1].data_ptr()
    dx = torch.empty((B, lin, g.cin), device=gy.device, dtype=gy.dtype)
    alg = 2.0 * B * g.out_len(lin) * g.cout * g.cin * g.k if (g.kind == "conv" and g.stride > 1) else None
    gy_in = gy
    if (CIN1_MFMA and g.kind == "cconv" and g.cout == 1 and g.dilation == 1 and g.k <= 15 and g.cin % 16 == 0 and g.cin <= 64
            and gy.dtype == torch.bfloat16 and res is None and _IMPL in (L.IMPL_AUTO, L.IMPL_MFMA)):
        # single-channel gradient (the decoder's last conv): the streaming MFMA kernel reads 8 consecutive samples per piece, so
        # the windows that run past the end of an item need real zeros behind it (a copy of B*lin samples)
        gy_in = torch.nn.functional.pad(gy, (0, 0, 0, _CIN1_BACK))
        p.Lin = gy_in.shape[1]
    _launch_conv(p, gy_in, wp_d, dx, res=res, dact_src=dact_src, impl=_IMPL, alg_flops=alg)
    if g.kind == "cconv" and g.k > 1:
        # fold the gradient of the reflected left pad back: dx[l] += dxp[-l], l = 1..P   (in_stride = -1)
        P = min((g.k - 1) * g.dilation, lin - 1)
        if P > 0:
            q = _new_params()
            _fill(q, B=B, Lin=gy.shape[1], Cin=g.cout, M=g.cin, Lout=P, LoutTot=lin, J=g.k, in_stride=-1,
                  tap_step=g.dilation, off0=0, t0=1, accumulate=1, dact_mode=dact_mode)
            # the gathered-GEMM kernel takes any row stride (bf16); otherwise the direct kernel
            kid = L.lib().ga_conv_kernel_id(C.byref(q), L.dt(gy), L.IMPL_MFMA) if _IMPL in (L.IMPL_AUTO, L.IMPL_MFMA) else -1
            fix_impl = L.IMPL_MFMA if kid in L.GEMM_KERNELS else L.IMPL_DIRECT
            _launch_conv(q, gy, wp_d, dx, dact_src=dact_src, impl=fix_impl)
    return dx


def fm_sign_fusable(g: ConvGeom, gy: Tensor, lin: int) -> bool:
    """True when ga_conv runs this layer's data gradient on the gathered-GEMM kernel, whose epilogue can form the
    feature-matching gradient (GA_RES_FM_SIGN)."""
    if gy.dtype != torch.bfloat16 or _IMPL not in (L.IMPL_AUTO, L.IMPL_MFMA) or g.kind != "conv":
        return False
    p = dgrad_params(g, gy.shape[0], lin)
    p.dact_mode, p.res_mode = L.DACT_LRELU_FROM_OUT, L.RES_FM_SIGN
    p.x = p.w = p.y = p.res = p.dact_src = p.res_coef_dev = 256  # alignment checks only: nothing is dereferenced
    if MERGE_PHASES:
        p, _ = _fold_phases(p, gy, None)
    return L.lib().ga_conv_kernel_id(C.byref(p), L.dt(gy), _IMPL) in (7, 8, 9, 11, 12)


def conv_wgrad(g: ConvGeom, ga: Tensor, x: Tensor, want_bias: bool = False):
    """Packed-forward-layout weight gradient [cout, k, cin] (fp32) of a 'conv' layer (and the bias gradient [cout],
    produced by the same C-ABI call, when want_bias)."""
    assert g.kind == "conv"
    L.require_cuda(ga, x)
    assert ga.is_contiguous() and x.is_contiguous() and ga.dtype == x.dtype
    B, lout, cout = ga.shape
    p = L.WgradParams()
    ndw = g.cout * g.k * g.cin
    buf = zeros_f32(ndw + (g.cout if want_bias else 0), x.device)  # one buffer for both
    dw = buf[:ndw].view(g.cout, g.k, g.cin)
    db = buf[ndw:] if want_bias else None
    xp, front = _cin1_padded(g, x)
    if xp is not None:
        x = xp
    _fill(p, g=ga.data_ptr(), x=x.data_ptr(), dw=dw.data_ptr(), dbias=L.ptr(db), B=B, Lin=x.shape[1], Cin=g.cin, M=g.cout,
          Lout=lout, J=g.k, in_stride=g.stride, tap_step=g.dilation, off0=front - g.padding, LgTot=lout, g_stride=1, g_phase=0)
    nws = L.lib().ga_wgrad_workspace_elems(C.byref(p), L.dt(x), _IMPL) if WGRAD_SLABS else 0
    if nws:
        ws = torch.empty((nws,), device=x.device, dtype=torch.float32)  # per-slice partial tiles (plain stores, no atomics)
        p.ws = ws.data_ptr()
    L.check(L.lib().ga_wgrad(C.byref(p), L.dt(x), _IMPL, L.stream()), "ga_wgrad")
    return (dw, db) if want_bias else dw


# ================================================================================================
# weight packing (+ fused weight norm)
def pack_weight(g: ConvGeom, w: Tensor, mode: int, dtype: torch.dtype, scale: Optional[Tensor] = None) -> Tensor:
    L.require_cuda(w)
    d0, d1, k = g.wdims
    assert tuple(w.shape) == (d0, d1, k) and w.dtype == torch.float32 and w.is_contiguous()
    n = L.lib().ga_packed_elems(mode, d0, d1, k, g.stride)
    out = torch.empty((n,), device=w.device, dtype=dtype)
    L.check(L.lib().ga_pack_weights(w.data_ptr(), L.ptr(scale), out.data_ptr(), L.dt(out), mode, d0, d1, k, g.stride,
                                    g.padding, L.stream()), "ga_pack_weights")
    return out


def wn_scale(v: Tensor, gparam: Tensor) -> Tuple[Tensor, Tensor]:
    rows = v.shape[0]
    n = v.numel() // rows
    scale = torch.empty((rows,), device=v.device, dtype=torch.float32)
    norm = torch.empty_like(scale)
    L.check(L.lib().ga_wn_scale(v.data_ptr(), gparam.data_ptr(), rows, n, scale.data_ptr(), norm.data_ptr(), L.stream()),
            "ga_wn_scale")
    return scale, norm


# ---- packed-weight cache -------------------------------------------------------------------------
# A trainable layer's packed weights only change when its optimizer steps (a raw-pointer kernel torch's version counters
# do not see) or when torch writes the parameter in place (which they do see).  FusedAdamW registers the address range of
# its flat parameter buffer with an epoch it bumps on every step, so a pack made in the G step (after D's update) is
# reused by the next iteration's D step instead of being rebuilt from the fp32 masters.
_EPOCH_RANGES: list = []   # (lo, hi, [epoch])
_PACK_CACHE: dict = {}
PACK_CACHE = True


def register_param_range(lo: int, hi: int) -> list:
    global _EPOCH_RANGES
    _EPOCH_RANGES = [r for r in _EPOCH_RANGES if r[1] <= lo or r[0] >= hi]  # address reuse by a new optimizer
    stale = [k for k in _PACK_CACHE if lo <= k[0] < hi]
    for k in stale:
        del _PACK_CACHE[k]
    holder = [0]
    _EPOCH_RANGES.append((lo, hi, holder))
    return holder


def unregister_param_range(lo: int, hi: int) -> None:
    """The optimizer that owned [lo, hi) is gone (FusedAdamW registers this as its finalizer): drop its epoch and every packed
    weight cached for parameters in that range, so neither outlives the buffer (a later tensor at a recycled address can then
    never meet a stale pack, and the bf16 packs of a finished stage do not stay resident)."""
    global _EPOCH_RANGES
    _EPOCH_RANGES = [r for r in _EPOCH_RANGES if not (r[0] == lo and r[1] == hi)]
    for k in [k for k in _PACK_CACHE if lo <= k[0] < hi]:
        del _PACK_CACHE[k]


def _epoch_holder(ptr: int):
    for lo, hi, h in _EPOCH_RANGES:
        if lo <= ptr < hi:
            return h
    return None


_ZERO1: dict = {}


def _zero_scalar(device) -> Tensor:
    """One shared fp32 zero per device (never written): the storage behind every expanded gradient carrier."""
    z = _ZERO1.get(device)
    if z is None:
        z = _ZERO1[device] = torch.zeros((1,), device=device, dtype=torch.float32)
    return z


# ---- pre-zeroed fp32 arena -----------------------------------------------------------------------
# Every atomically accumulated result (weight / bias / affine gradients, loss sums) needs a zeroed fp32 buffer; a training
# step asks for ~180 of them, most a few KB.  They are carved from chunks zeroed by ONE fill each instead of one tiny fill
# launch per buffer.  A carved slice is handed out once and never reused; a chunk lives as long as any slice of it does.
ZERO_ARENA = True
_ARENA_CHUNK = 4 << 20        # floats per chunk (16 MiB: one ~6 us fill)
_ARENA_MAX_CARVE = 1 << 20    # larger requests get their own torch.zeros (the fill is bandwidth, not launch, there)
_ARENA: dict = {}             # device -> [chunk, offset]


def zeros_f32(n: int, device) -> Tensor:
    if not ZERO_ARENA or n > _ARENA_MAX_CARVE:
        return torch.zeros((n,), device=device, dtype=torch.float32)
    device = torch.device(device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ent = _ARENA.get(key)
    n_al = (n + 63) // 64 * 64  # 256-byte granules
    if ent is None or ent[1] + n_al > ent[0].numel():
        ent = _ARENA[key] = [torch.zeros((_ARENA_CHUNK,), device=device, dtype=torch.float32), 0]
    out = ent[0][ent[1]:ent[1] + n]
    ent[1] += n_al
    return out


# ---- deferred weight gradients ---------------------------------------------------------------------
# While a WeightGradSink is active (lab3_codec_train._backward_and_reduce installs one around loss.backward()), the
# PackTrainable nodes of layers whose parameters belong to the sink's optimizer do not unpack / weight-norm-backward /
# return their gradient one layer at a time: they hand the packed gradient to the sink, and flush() takes all of them to
# their final place in the optimizer's flat gradient buffer in one launch (ga_wgrad_finish_multi).  p.grad of those
# parameters stays as it was (None after zero_grad): the flat buffer is what the optimizer and the all-reduce read.
WGRAD_SINK = True
WGRAD_DEFER = os.environ.get("GA_WGRAD_DEFER", "1") == "1"  # plain layers' weight gradients in one grouped launch at the end of the pass
_SINK = None
LAST_WGRAD_GROUP = None


class WeightGradSink:
    def __init__(self, opt):
        self.opt = opt
        m = getattr(opt, "_sink_map", None)
        if m is None:
            m = opt._sink_map = {p.data_ptr(): i for i, p in enumerate(opt.params)}
        self.index = m
        self.pending = {}
        self.deferred = {}    # iv -> (ib | None, v, geom, ga, x): weight gradients not computed yet (ConvTrain.backward, WGRAD_DEFER)
        self.on_offer = None  # dp._OverlapState.learn_pair: told (index of v, index of g | None) of every layer that arrives

    def offer(self, v, gparam, norm, geom, gpacked) -> bool:
        d0, d1, k = geom.wdims
        if geom.fwd_pack != L.PACK_CONV_FWD or d1 * k > L.WFINISH_MAX_ROW or gpacked.dtype != torch.float32:
            return False
        iv = self.index.get(v.data_ptr())
        ig = None if gparam is None else self.index.get(gparam.data_ptr())
        if iv is None or (gparam is not None and ig is None) or iv in self.pending or iv in self.deferred:
            return False
        self.pending[iv] = (ig, v, gparam, norm, (d0, d1, k), gpacked.contiguous())
        if self.on_offer is not None:
            self.on_offer(iv, ig)
        return True

    def defer(self, v, bias, geom, ga, x) -> bool:
        """ConvTrain.backward hands over (masked output gradient, layer input) of a plain (no weight norm) layer instead of
        launching its weight gradient: flush() computes all of them in one grouped launch (ga_wgrad_multi) -- the generator's
        forty layers are 40-50 us launches one by one, 7 us of MFMA time each.  Not under the overlapped all-reduce (its buckets
        leave while the pass is still running)."""
        if self.on_offer is not None or ga.dtype != torch.bfloat16 or geom.kind != "conv" or geom.cin % 8 or geom.cout <= 64:
            return False
        d0, d1, k = geom.wdims
        if geom.fwd_pack != L.PACK_CONV_FWD or d1 * k > L.WFINISH_MAX_ROW:
            return False
        iv = self.index.get(v.data_ptr())
        ib = None if bias is None else self.index.get(bias.data_ptr())
        if iv is None or (bias is not None and ib is None) or iv in self.pending or iv in self.deferred:
            return False
        if self.opt.params[iv].grad is not None or (ib is not None and self.opt.params[ib].grad is not None):
            return False  # (a gradient from elsewhere is already there: keep to the plain path)
        self.deferred[iv] = (ib, v, geom, ga, x)
        return True

    def _run_deferred(self) -> None:
        todo = list(self.deferred.items())
        self.deferred = {}
        if not todo:
            return
        opt, lib = self.opt, L.lib()
        dev = todo[0][1][3].device
        sizes = [g.cout * g.k * g.cin for _, (_, _, g, _, _) in todo]
        slab = torch.zeros((sum(sizes),), device=dev, dtype=torch.float32)
        items = (L.WgradParams * len(todo))()
        views, off = [], 0
        for it, n, (iv, (ib, v, g, ga, x)) in zip(items, sizes, todo):
            dw = slab[off:off + n]
            off += n
            views.append(dw)
            B, lout, _ = ga.shape
            _fill(it, g=ga.data_ptr(), x=x.data_ptr(), dw=dw.data_ptr(), dbias=None if ib is None else opt.grad_views[ib].data_ptr(),
                  B=B, Lin=x.shape[1], Cin=g.cin, M=g.cout, Lout=lout, J=g.k, in_stride=g.stride, tap_step=g.dilation,
                  off0=-g.padding, LgTot=lout, g_stride=1, g_phase=0)
        rc = lib.ga_wgrad_multi(C.cast(items, C.c_void_p), len(todo), L.DT_BF16, L.stream())
        global LAST_WGRAD_GROUP
        LAST_WGRAD_GROUP = (len(todo), rc)  # (layers, status of the grouped call: 0 = one launch per 20, 4 = fell back to one per layer)
        if rc == 4:  # GA_ERR_UNSUPPORTED: one by one
            for it in items:
                L.check(lib.ga_wgrad(C.byref(it), L.DT_BF16, _IMPL, L.stream()), "ga_wgrad")
        else:
            L.check(rc, "ga_wgrad_multi")
        for dw, (iv, (ib, v, g, ga, x)) in zip(views, todo):
            self.pending[iv] = (None, v, None, None, g.wdims, dw)

    def flush(self, only=None) -> None:
        """after opt.gather_grads(): add (p.grad present: it has been copied in) or write (absent) the deferred gradients.
        `only`: finish just these layers (indices of their weights) - the overlapped all-reduce sends bucket by bucket."""
        self._run_deferred()
        if only is None:
            todo = list(self.pending.items())
            self.pending = {}
        else:
            todo = [(iv, self.pending.pop(iv)) for iv in only if iv in self.pending]
        opt, lib = self.opt, L.lib()
        for c0 in range(0, len(todo), L.PACK_MULTI_MAX):
            chunk = todo[c0:c0 + L.PACK_MULTI_MAX]
            items = (L.WFinishItem * len(chunk))()
            for it, (iv, (ig, v, gparam, norm, (d0, d1, k), gpacked)) in zip(items, chunk):
                it.gpacked, it.dv = gpacked.data_ptr(), opt.grad_views[iv].data_ptr()
                it.D0, it.D1, it.K = d0, d1, k
                it.accumulate = 1 if opt.params[iv].grad is not None else 0
                if gparam is not None:
                    it.v, it.g, it.norm, it.dg = v.data_ptr(), gparam.data_ptr(), norm.data_ptr(), opt.grad_views[ig].data_ptr()
                    it.accumulate |= 2 if opt.params[ig].grad is not None else 0
            L.check(lib.ga_wgrad_finish_multi(C.cast(items, C.c_void_p), len(chunk), L.stream()), "ga_wgrad_finish_multi")


class PackTrainable(torch.autograd.Function):
    """(v, g | None) -> (forward pack, data-gradient pack) in the activation dtype.  Backward takes the
    packed-forward-layout gradient from the weight-gradient kernel back to the parameter layout and through
    the weight-norm (w = g*v/||v||, lab3_codec_models.py:125,143)."""

    @staticmethod
    def forward(ctx, v, gparam, geom: ConvGeom, dtype):
        v = v.contiguous()
        key = (v.data_ptr(), geom, dtype)
        holder = _epoch_holder(v.data_ptr()) if PACK_CACHE else None
        stamp = None
        if holder is not None:
            stamp = (holder[0], v._version, None if gparam is None else (gparam.data_ptr(), gparam._version))
            hit = _PACK_CACHE.get(key)
            if hit is not None and hit[0] == stamp:
                _, wp_f, wp_d, norm = hit
            else:
                hit = None
        else:
            hit = None
        if hit is None:
            scale = norm = None
            if gparam is not None:
                scale, norm = wn_scale(v, gparam.contiguous())
            wp_f = pack_weight(geom, v, geom.fwd_pack, dtype, scale)
            wp_d = pack_weight(geom, v, geom.dgrad_pack, dtype, scale)
            if holder is not None:
                _PACK_CACHE[key] = (stamp, wp_f, wp_d, norm)
        ctx.geom = geom
        ctx.has_g = gparam is not None
        ctx.save_for_backward(v, gparam, norm)
        ctx.mark_non_differentiable(wp_f, wp_d)
        ctx.set_materialize_grads(False)  # else autograd zero-fills a weight-sized tensor for each packed output, every backward
        # fp32 gradient carrier: autograd casts a gradient to the dtype of the tensor it belongs to, so the
        # fp32 weight gradient cannot travel on the (possibly bf16) packed weights themselves.
        carrier = _zero_scalar(v.device).expand(wp_f.numel())
        return wp_f, wp_d, carrier

    @staticmethod
    def backward(ctx, _gwp_f, _gwp_d, gwp_f):
        v, gparam, norm = ctx.saved_tensors
        geom = ctx.geom
        d0, d1, k = geom.wdims
        if gwp_f is None:
            return None, None, None, None
        if _SINK is not None and _SINK.offer(v, gparam if ctx.has_g else None, norm, geom, gwp_f):
            return None, None, None, None
        dw = torch.empty_like(v)
        L.check(L.lib().ga_unpack_grad(gwp_f.contiguous().data_ptr(), dw.data_ptr(), geom.fwd_pack, d0, d1, k, geom.stride,
                                       geom.padding, 0, L.stream()), "ga_unpack_grad")
        if not ctx.has_g:
            return dw, None, None, None
        dv = torch.empty_like(v)
        dg = torch.empty_like(gparam)
        L.check(L.lib().ga_wn_bwd(dw.data_ptr(), v.data_ptr(), gparam.contiguous().data_ptr(), norm.data_ptr(), d0, d1 * k,
                                  dv.data_ptr(), dg.data_ptr(), 0, L.stream()), "ga_wn_bwd")
        return dv, dg, None, None


PACK_MULTI = True


def prepack_layers(layers, dtype) -> None:
    """Pack every stale layer of `layers` = [(v, g | None, geom), ...] in two launches (ga_pack_multi) and leave the results
    in the pack cache, where the per-layer PackTrainable nodes of the forward pass that follows find them.  Layers whose
    parameters no optimizer has registered (no epoch to stamp a cache entry with) are left to PackTrainable."""
    if not (PACK_CACHE and PACK_MULTI):
        return
    todo = []
    for v, gparam, geom in layers:
        if not v.is_contiguous() or v.dtype != torch.float32 or (gparam is not None and not gparam.is_contiguous()):
            continue
        holder = _epoch_holder(v.data_ptr())
        if holder is None:
            continue
        key = (v.data_ptr(), geom, dtype)
        stamp = (holder[0], v._version, None if gparam is None else (gparam.data_ptr(), gparam._version))
        hit = _PACK_CACHE.get(key)
        if hit is None or hit[0] != stamp:
            todo.append((key, stamp, v, gparam, geom))
    lib = L.lib()
    for c0 in range(0, len(todo), L.PACK_MULTI_MAX):
        chunk = todo[c0:c0 + L.PACK_MULTI_MAX]
        dev = chunk[0][2].device
        al = 128  # elements: 256-byte (bf16) / 512-byte (fp32) granules
        sizes_f, sizes_d, rows = [], [], 0
        for _, _, v, gparam, geom in chunk:
            d0, d1, k = geom.wdims
            sizes_f.append((lib.ga_packed_elems(geom.fwd_pack, d0, d1, k, geom.stride) + al - 1) // al * al)
            sizes_d.append((lib.ga_packed_elems(geom.dgrad_pack, d0, d1, k, geom.stride) + al - 1) // al * al)
            rows += (d0 + 63) // 64 * 64 if gparam is not None else 0
        slab_f = torch.empty((sum(sizes_f),), device=dev, dtype=dtype)
        strided = any(geom.dgrad_pack == L.PACK_CONV_DGRAD_STRIDED for *_, geom in chunk)
        slab_d = (torch.zeros if strided else torch.empty)((sum(sizes_d),), device=dev, dtype=dtype)
        sn = torch.empty((2, max(rows, 1)), device=dev, dtype=torch.float32)
        items = (L.PackItem * len(chunk))()
        of = od = orow = 0
        results = []
        for i, (key, stamp, v, gparam, geom) in enumerate(chunk):
            d0, d1, k = geom.wdims
            nf = lib.ga_packed_elems(geom.fwd_pack, d0, d1, k, geom.stride)
            nd = lib.ga_packed_elems(geom.dgrad_pack, d0, d1, k, geom.stride)
            wp_f, wp_d = slab_f[of:of + nf], slab_d[od:od + nd]
            norm = None
            it = items[i]
            it.v, it.wp_f, it.wp_d = v.data_ptr(), wp_f.data_ptr(), wp_d.data_ptr()
            if gparam is not None:
                scale, norm = sn[0, orow:orow + d0], sn[1, orow:orow + d0]
                it.g, it.scale, it.norm = gparam.data_ptr(), scale.data_ptr(), norm.data_ptr()
                orow += (d0 + 63) // 64 * 64
            it.D0, it.D1, it.K, it.S, it.P = d0, d1, k, geom.stride, geom.padding
            it.mode_f, it.mode_d = geom.fwd_pack, geom.dgrad_pack
            of += sizes_f[i]
            od += sizes_d[i]
            results.append((key, stamp, wp_f, wp_d, norm))
        L.check(lib.ga_pack_multi(C.cast(items, C.c_void_p), len(chunk), L.DT_BF16 if dtype == torch.bfloat16 else L.DT_F32,
                                  L.stream()), "ga_pack_multi")
        for key, stamp, wp_f, wp_d, norm in results:
            _PACK_CACHE[key] = (stamp, wp_f, wp_d, norm)


def pack_frozen(geom: ConvGeom, v: Tensor, gparam: Optional[Tensor], dtype) -> Tuple[Tensor, Tensor]:
    with torch.no_grad():
        scale = None
        if gparam is not None:
            scale, _ = wn_scale(v.contiguous(), gparam.contiguous())
        return (pack_weight(geom, v.contiguous(), geom.fwd_pack, dtype, scale),
                pack_weight(geom, v.contiguous(), geom.dgrad_pack, dtype, scale))


# ================================================================================================
def act_bwd(gy: Tensor, src: Tensor, mode: int) -> Tensor:
    ga = torch.empty_like(gy)
    L.check(L.lib().ga_act_bwd(gy.data_ptr(), src.data_ptr(), ga.data_ptr(), L.dt(gy), gy.numel(), mode, L.stream()), "ga_act_bwd")
    return ga


def colsum(g: Tensor) -> Tensor:
    M = g.shape[-1]
    out = zeros_f32(M, g.device)
    L.check(L.lib().ga_colsum(g.data_ptr(), L.dt(g), g.numel() // M, M, out.data_ptr(), L.stream()), "ga_colsum")
    return out


class ConvTrain(torch.autograd.Function):
    """y = act(conv(x, w) + bias) for a trainable zero-padded Conv1d (G and D layers).
    wp_f / wp_d come from PackTrainable; the weight gradient is returned in wp_f's packed layout."""

    @staticmethod
    def forward(ctx, x, wp_f, wp_d, carrier, bias, geom: ConvGeom, post_act: int, vparam=None):
        """vparam: the layer's weight parameter when it has no weight norm (the generator's convs) -- lets the backward pass
        leave the weight gradient to the sink's grouped launch."""
        x = x.contiguous()
        y = conv_forward(geom, x, wp_f, bias, post_act=post_act)
        ctx.geom, ctx.post_act = geom, post_act
        ctx.vparam, ctx.bias_param = vparam, bias
        ctx.save_for_backward(x, wp_d, y if post_act else None)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wp_d, y = ctx.saved_tensors
        geom = ctx.geom
        gy = gy.contiguous()
        if ctx.post_act not in (L.ACT_NONE, L.ACT_LRELU):
            raise NotImplementedError("ConvTrain supports LeakyReLU or no activation")
        want_b = ctx.has_bias and ctx.needs_input_grad[4]
        gx = gw = gb = None
        M = gy.shape[-1]
        ev = 8 if gy.dtype == torch.bfloat16 else 4
        if ctx.post_act == L.ACT_LRELU and want_b and M % ev == 0 and 256 % (M // ev) == 0:
            # mask and bias gradient in one pass over gy
            ga = torch.empty_like(gy)
            gb = zeros_f32(M, gy.device)
            L.check(L.lib().ga_act_bwd_bias(gy.data_ptr(), y.data_ptr(), ga.data_ptr(), L.dt(gy), gy.numel() // M, M,
                                            L.DACT_LRELU_FROM_OUT, gb.data_ptr(), L.stream()), "ga_act_bwd_bias")
            want_b = False
        else:
            ga = act_bwd(gy, y, L.DACT_LRELU_FROM_OUT) if ctx.post_act == L.ACT_LRELU else gy
        if ctx.needs_input_grad[0]:
            gx = conv_dgrad(geom, ga, wp_d, x.shape[1])
        if ctx.needs_input_grad[3]:
            if (WGRAD_DEFER and _SINK is not None and ctx.vparam is not None
                    and _SINK.defer(ctx.vparam, ctx.bias_param if want_b else None, geom, ga, x)):
                pass  # weight (and bias) gradient come out of the sink's grouped launch, straight into the optimizer's flat buffer
            elif want_b:
                gw, gb = conv_wgrad(geom, ga, x, want_bias=True)
                gw = gw.reshape(-1)
            else:
                gw = conv_wgrad(geom, ga, x).reshape(-1)
        elif want_b:
            gb = colsum(ga)
        return gx, None, None, gw, gb, None, None, None


def _unpack_wgrad(geom: ConvGeom, dw_packed: Tensor) -> Tensor:
    d0, d1, k = geom.wdims
    dw = torch.empty((d0, d1, k), device=dw_packed.device, dtype=torch.float32)
    L.check(L.lib().ga_unpack_grad(dw_packed.contiguous().data_ptr(), dw.data_ptr(), geom.fwd_pack, d0, d1, k, geom.stride, geom.padding, 0,
                                   L.stream()), "ga_unpack_grad")
    return dw


class ConvTwice(torch.autograd.Function):
    """y = conv(x, w) + bias for a plain 'conv' layer, differentiable TWICE: its backward is built from the two autograd nodes below, so
    a gradient penalty (R1: the squared norm of d logits / d input, reference lab 3/src/lab3_train.py:329-339) can be differentiated
    with respect to the weights.  A convolution is bilinear in (x, w), so every derivative of every order is one of the same three
    kernels with the operands in other roles: forward (ga_conv), data gradient (ga_conv, dgrad form), weight gradient (ga_wgrad).
    w is the [cout, cin, k] fp32 weight itself (packed per call, no cache); no fused activation (the caller applies it as a tensor op)."""

    @staticmethod
    def forward(ctx, x, w, bias, geom: ConvGeom):
        x, w = x.contiguous(), w.contiguous()
        ctx.geom = geom
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return conv_forward(geom, x, pack_weight(geom, w, geom.fwd_pack, x.dtype), bias)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = _ConvDataGrad.apply(gy, w, ctx.geom, x.shape[1]) if ctx.needs_input_grad[0] else None
        gw = _ConvWeightGrad.apply(gy, x, ctx.geom) if ctx.needs_input_grad[1] else None
        gb = gy.float().sum(dim=(0, 1)) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb, None


class _ConvDataGrad(torch.autograd.Function):
    """gx = dgrad(gy, w); linear in each: d<ggx, gx>/d gy = conv(ggx, w), d<ggx, gx>/d w = wgrad(gy, ggx)."""

    @staticmethod
    def forward(ctx, gy, w, geom: ConvGeom, lin: int):
        gy = gy.contiguous()
        ctx.geom = geom
        ctx.save_for_backward(gy, w)
        return conv_dgrad(geom, gy, pack_weight(geom, w, geom.dgrad_pack, gy.dtype), lin)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        ggx = ggx.contiguous()
        g_gy = ConvTwice.apply(ggx, w, None, ctx.geom) if ctx.needs_input_grad[0] else None
        g_w = _ConvWeightGrad.apply(gy, ggx, ctx.geom) if ctx.needs_input_grad[1] else None
        return g_gy, g_w, None, None


class _ConvWeightGrad(torch.autograd.Function):
    """gw = wgrad(gy, x) in the [cout, cin, k] layout; d<ggw, gw>/d gy = conv(x, ggw), d<ggw, gw>/d x = dgrad(gy, ggw)."""

    @staticmethod
    def forward(ctx, gy, x, geom: ConvGeom):
        gy, x = gy.contiguous(), x.contiguous()
        ctx.geom = geom
        ctx.save_for_backward(gy, x)
        return _unpack_wgrad(geom, conv_wgrad(geom, gy, x))

    @staticmethod
    def backward(ctx, ggw):
        gy, x = ctx.saved_tensors
        ggw = ggw.contiguous()
        g_gy = ConvTwice.apply(x, ggw, None, ctx.geom) if ctx.needs_input_grad[0] else None
        g_x = _ConvDataGrad.apply(gy, ggw, ctx.geom, x.shape[1]) if ctx.needs_input_grad[1] else None
        return g_gy, g_x, None


class DiscChain(torch.autograd.Function):
    """A whole single-scale waveform discriminator (reference lab3_codec_models.py:131-152): n-1 x [conv + bias +
    LeakyReLU(0.2)] and the logits conv, as ONE autograd node, so that the backward pass can fuse across layers what
    separate nodes cannot: the data-gradient launch of layer l+1 applies layer l's LeakyReLU mask (and adds the gradient
    that arrives at feature map l from outside, e.g. feature matching) in its epilogue and so emits the masked gradient
    layer l needs directly -- no gradient accumulation pass and no mask pass over the feature maps.
    inputs : x [B, L, C0], geoms, then per layer (forward pack, data-gradient pack, fp32 gradient carrier, bias)
    outputs: the n-1 feature maps (post-activation, NLC) and the logits."""

    @staticmethod
    def forward(ctx, x, geoms, *flat):
        n = len(geoms)
        assert len(flat) == 4 * n
        h = x.contiguous()
        saved_in, outs = [], []
        for l, g in enumerate(geoms):
            wp_f, bias = flat[4 * l], flat[4 * l + 3]
            saved_in.append(h)
            h = conv_forward(g, h, wp_f, bias, post_act=L.ACT_LRELU if l < n - 1 else L.ACT_NONE)
            outs.append(h)
        ctx.geoms = geoms
        ctx.has_bias = [flat[4 * l + 3] is not None for l in range(n)]
        ctx.save_for_backward(saved_in[0], *outs[:-1], *[flat[4 * l + 1] for l in range(n)])
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        geoms = ctx.geoms
        n = len(geoms)
        saved = ctx.saved_tensors
        ins = list(saved[:n])          # input of layer l: x, y_0 .. y_{n-2}
        wp_d = list(saved[n:2 * n])
        grads = [None] * (2 + 4 * n)
        dtype = ins[0].dtype
        ga = gouts[n - 1]              # the logits conv has no activation
        if ga is not None:
            ga = ga.contiguous().to(dtype)
        for l in range(n - 1, -1, -1):
            g_feat = gouts[l - 1] if l > 0 else None
            if ga is None and g_feat is None:
                continue
            if ga is not None:
                need_w, need_b = ctx.needs_input_grad[2 + 4 * l + 2], ctx.has_bias[l] and ctx.needs_input_grad[2 + 4 * l + 3]
                if need_w:
                    if need_b:
                        gw, gb = conv_wgrad(geoms[l], ga, ins[l], want_bias=True)
                        grads[2 + 4 * l + 3] = gb
                    else:
                        gw = conv_wgrad(geoms[l], ga, ins[l])
                    grads[2 + 4 * l + 2] = gw.reshape(-1)
                elif need_b:
                    grads[2 + 4 * l + 3] = colsum(ga)
            if l == 0:
                if ga is not None and ctx.needs_input_grad[0]:
                    grads[0] = conv_dgrad(geoms[0], ga, wp_d[0], ins[0].shape[1])
                break
            # masked gradient of layer l-1:  (dgrad_l(ga) + g_feat) * lrelu'(y_{l-1})
            y_prev = ins[l]
            extra = act_bwd(g_feat.contiguous().to(dtype), y_prev, L.DACT_LRELU_FROM_OUT) if g_feat is not None else None
            if ga is not None:
                ga = conv_dgrad(geoms[l], ga, wp_d[l], y_prev.shape[1], dact_mode=L.DACT_LRELU_FROM_OUT, dact_src=y_prev, res=extra)
            else:
                ga = extra
        return tuple(grads)


class DiscChainSplit(torch.autograd.Function):
    """The generator step's form of DiscChain (reference lab3_codec_train.py:455-462: D(real) under no_grad, then D(fake)
    whose gradient reaches only the generator).  D's weights are frozen there and D acts on every batch item independently,
    so both passes run as ONE pass over [real ; fake]: twice the GEMM N per launch, half the launches.  The backward pass
    runs on the fake items only (contiguous tail slices of the saved feature maps, batch is the outermost axis).
    inputs : x_real [B0, L, C0] (no gradient), x_fake [B1, L, C0], geoms, then per layer (forward pack, data-gradient
             pack, bias), all without gradient
    outputs: n real-half tensors (features then logits; non-differentiable), then the n fake-half tensors."""

    @staticmethod
    def forward(ctx, x_real, x_fake, geoms, want_fm, *flat):
        n = len(geoms)
        assert len(flat) == 3 * n
        b0 = int(x_real.shape[0])
        h = torch.cat([x_real, x_fake], dim=0)
        outs = []
        for l, g in enumerate(geoms):
            h = conv_forward(g, h, flat[3 * l], flat[3 * l + 2], post_act=L.ACT_LRELU if l < n - 1 else L.ACT_NONE)
            outs.append(h)
        ctx.geoms, ctx.b0, ctx.len0 = geoms, b0, int(x_fake.shape[1])
        ctx.save_for_backward(*outs[:-1], *[flat[3 * l + 1] for l in range(n)])
        ctx.set_materialize_grads(False)
        real = tuple(o[:b0] for o in outs)
        ctx.mark_non_differentiable(*real)
        ctx.want_fm = bool(want_fm)
        if not want_fm:
            return real + tuple(o[b0:] for o in outs)
        # feature matching of this scale, (1/n_feats) * sum_l mean|fake_l - real_l| (lab3_codec_models.py:191-199), summed
        # into one scalar here; its gradient is formed inside the data-gradient epilogues of backward()
        if x_fake.shape[0] != b0:
            raise ValueError("feature matching needs as many real as fake items")
        fm = zeros_f32(1, h.device)
        ctx.fm_scales = [1.0 / (o[b0:].numel() * max(1, n - 1)) for o in outs[:-1]]
        reduce_absdiff_multi([o[b0:] for o in outs[:-1]], [o[:b0] for o in outs[:-1]], ctx.fm_scales, fm)
        return real + tuple(o[b0:] for o in outs) + (fm.view(()),)

    @staticmethod
    def backward(ctx, *gouts):
        geoms, b0 = ctx.geoms, ctx.b0
        n = len(geoms)
        saved = ctx.saved_tensors
        feats = saved[:n - 1]
        wp_d = saved[n - 1:]
        g_fm = gouts[2 * n] if ctx.want_fm else None
        gouts = gouts[n:2 * n]
        grads = [None] * (4 + 3 * n)
        if any(ctx.needs_input_grad[4:]):
            raise RuntimeError("DiscChainSplit is for frozen discriminator weights")
        dtype = wp_d[0].dtype
        if g_fm is not None:
            g_fm = g_fm.reshape(1).float().contiguous()
        ga = gouts[n - 1]
        if ga is not None:
            ga = ga.contiguous().to(dtype)
        for l in range(n - 1, -1, -1):
            g_feat = gouts[l - 1] if l > 0 else None
            if ga is None and g_feat is None and (g_fm is None or l == 0):
                continue
            if l == 0:
                if ga is not None and ctx.needs_input_grad[1]:
                    grads[1] = conv_dgrad(geoms[0], ga, wp_d[0], ctx.len0)
                break
            y_prev, r_prev = feats[l - 1][b0:], feats[l - 1][:b0]
            extra = act_bwd(g_feat.contiguous().to(dtype), y_prev, L.DACT_LRELU_FROM_OUT) if g_feat is not None else None
            if g_fm is not None:
                if ga is not None and extra is None and fm_sign_fusable(geoms[l], ga, y_prev.shape[1]):
                    # (dgrad + c * sign(fake - real)) * lrelu'(fake), all in the data-gradient epilogue
                    ga = conv_dgrad(geoms[l], ga, wp_d[l], y_prev.shape[1], dact_mode=L.DACT_LRELU_FROM_OUT, dact_src=y_prev,
                                    res=r_prev, fm_coef=(ctx.fm_scales[l - 1], g_fm))
                    continue
                gf = torch.empty_like(y_prev)
                L.check(L.lib().ga_reduce_grad(y_prev.data_ptr(), r_prev.data_ptr(), L.dt(y_prev), y_prev.numel(), L.RED_ABSDIFF,
                                               g_fm.data_ptr(), ctx.fm_scales[l - 1], gf.data_ptr(), 0, L.stream()), "ga_reduce_grad")
                gf = act_bwd(gf, y_prev, L.DACT_LRELU_FROM_OUT)
                extra = gf if extra is None else extra + gf
            if ga is not None:
                ga = conv_dgrad(geoms[l], ga, wp_d[l], y_prev.shape[1], dact_mode=L.DACT_LRELU_FROM_OUT, dact_src=y_prev, res=extra)
            else:
                ga = extra
        return tuple(grads)


REDUCE_MULTI = os.environ.get("GA_REDUCE_MULTI", "1") == "1"


def reduce_absdiff_multi(a_list, b_list, scales, out: Tensor) -> None:
    """out[0] += sum_i scales[i] * sum |a_i - b_i| over contiguous tensor pairs: one launch per 8 pairs (ga_reduce_multi)."""
    lib = L.lib()
    if not REDUCE_MULTI:
        for a, b, sc in zip(a_list, b_list, scales):
            L.check(lib.ga_reduce(a.data_ptr(), b.data_ptr(), L.dt(a), a.numel(), L.RED_ABSDIFF, float(sc), out.data_ptr(), L.stream()), "ga_reduce")
        return
    for c0 in range(0, len(a_list), 8):
        aa, bb, ss = a_list[c0:c0 + 8], b_list[c0:c0 + 8], scales[c0:c0 + 8]
        k = len(aa)
        assert all(t.is_contiguous() and t.dtype == aa[0].dtype for t in aa + bb)
        pa = (C.c_void_p * k)(*[t.data_ptr() for t in aa])
        pb = (C.c_void_p * k)(*[t.data_ptr() for t in bb])
        pn = (C.c_size_t * k)(*[t.numel() for t in aa])
        ps = (C.c_float * k)(*[float(v) for v in ss])
        L.check(lib.ga_reduce_multi(C.cast(pa, C.c_void_p), C.cast(pb, C.c_void_p), C.cast(pn, C.c_void_p), C.cast(ps, C.c_void_p), k, L.dt(aa[0]),
                                    L.RED_ABSDIFF, out.data_ptr(), L.stream()), "ga_reduce_multi")


def act_fwd(x: Tensor, act: int) -> Tensor:
    y = torch.empty_like(x)
    L.check(L.lib().ga_act_fwd(x.data_ptr(), y.data_ptr(), L.dt(x), x.numel(), act, L.stream()), "ga_act_fwd")
    return y


def _streams_with_pre_act(g: ConvGeom, x: Tensor, pre_act: int) -> bool:
    """True when ga_conv would run this forward, input activation included, on one of the streaming MFMA kernels."""
    if _IMPL not in (L.IMPL_AUTO, L.IMPL_MFMA):
        return False
    p = fwd_params(g, x.shape[0], x.shape[1])
    p.pre_act = pre_act
    if MERGE_PHASES:
        p, _ = _fold_phases(p, x, None)
    return L.lib().ga_conv_kernel_id(C.byref(p), L.dt(x), _IMPL) in (L.KERNEL_STREAM, L.KERNEL_STREAM_WIN)


def elu_needs_a_pass(g: ConvGeom, x: Tensor) -> bool:
    """True when conv(ELU(x)) of frozen layer g would materialise ELU(x) with a pass of its own (bf16 layers that run on the
    gathered-GEMM kernel): the producer of x can then emit ELU(x) itself (ACT_ELU_DONE below)."""
    return x.dtype == torch.bfloat16 and g.cout >= 16 and _IMPL in (L.IMPL_AUTO, L.IMPL_MFMA) and not _streams_with_pre_act(g, x, L.ACT_ELU)


# pre_act code of ConvFrozen / _frozen_forward: the input tensor already holds ELU(pre-activation) -- written by its producer's
# epilogue -- so the forward applies nothing and the data gradient takes ELU' from that output (GA_DACT_ELU_FROM_OUT)
ACT_ELU_DONE = -2


def _frozen_forward(geom: ConvGeom, x: Tensor, wp_f: Tensor, bias, pre_act: int, res=None, out_f32: bool = False, post_act: int = 0,
                    act_after_res: bool = False, elu_out2: bool = False):
    res = None if res is None else res.contiguous()
    if pre_act == ACT_ELU_DONE:
        pre_act = L.ACT_NONE
    if pre_act and x.dtype == torch.bfloat16 and geom.cout >= 16 and (post_act or not _streams_with_pre_act(geom, x, pre_act)):
        # bf16: materialise act(x) once (one streaming pass) so the conv takes the direct-to-LDS GEMM kernel,
        # which cannot transform its operand in flight (the thin layers' streaming kernels can: no copy there)
        return conv_forward(geom, act_fwd(x, pre_act), wp_f, bias, post_act=post_act, res=res, out_f32=out_f32, act_after_res=act_after_res,
                            elu_out2=elu_out2)
    return conv_forward(geom, x, wp_f, bias, pre_act=pre_act, post_act=post_act, res=res, out_f32=out_f32, act_after_res=act_after_res,
                        elu_out2=elu_out2)


FUSE_RESUNIT = os.environ.get("GA_FUSE_RESUNIT", "1") != "0"  # False: a thin residual unit's forward as three ga_conv launches (A/B checks)
FUSE_RESUNIT_WIDE = os.environ.get("GA_FUSE_RESUNIT_WIDE", "1") != "0"  # False: the 128-channel unit stays three launches


def resunit_fusable(x: Tensor, g_sc: ConvGeom, g1: ConvGeom, g3: ConvGeom) -> bool:
    """True when ga_resunit_fwd takes this residual unit: bf16, C in {32, 64, 128} -> C/2 -> C, causal k3 / k1 / k1 shortcut."""
    c = g_sc.cin
    return (FUSE_RESUNIT and x.dtype == torch.bfloat16 and _IMPL in (L.IMPL_AUTO, L.IMPL_MFMA) and x.shape[1] >= 3
            and (c in (32, 64) or (c == 128 and FUSE_RESUNIT_WIDE))
            and all(g.kind == "cconv" and g.dilation == 1 for g in (g_sc, g1, g3))
            and (g_sc.cin, g_sc.cout, g_sc.k) == (c, c, 1) and (g1.cin, g1.cout, g1.k) == (c, c // 2, 3)
            and (g3.cin, g3.cout, g3.k) == (c // 2, c, 1))


def resunit_forward(x: Tensor, sc_f: Tensor, sc_b: Tensor, b1_f: Tensor, b1_b: Tensor, b3_f: Tensor, b3_b: Tensor, want_t: bool, last=None,
                    post_elu: bool = False):
    """(y, ELU(hidden) or None) of one residual unit in one launch (ga_resunit_fwd).
    last = (forward pack [1][7][32], bias) of the decoder's closing conv: the tail form -- returns (wav [B, L, 1] fp32, None)
    and neither y nor the hidden tensor exists (callers that need no gradient)."""
    L.require_cuda(x, sc_f, b1_f, b3_f)
    B, lin, c = x.shape
    assert x.is_contiguous() and x.dtype == torch.bfloat16
    for w, n in ((sc_f, c * c), (b1_f, c // 2 * 3 * c), (b3_f, c * c // 2)):
        assert w.is_contiguous() and w.dtype == torch.bfloat16 and w.numel() >= n
    for b_, n in ((sc_b, c), (b1_b, c // 2), (b3_b, c)):
        assert b_.dtype == torch.float32 and b_.numel() == n
    p = L.ResunitParams()
    if last is not None:
        assert not want_t and c == 32 and lin >= 13
        w_last, b_last = last
        assert w_last.is_contiguous() and w_last.dtype == torch.bfloat16 and w_last.numel() >= 7 * 32 and b_last.dtype == torch.float32
        y = torch.empty((B, lin, 1), device=x.device, dtype=torch.float32)
        t = None
        _fill(p, w_last=w_last.data_ptr(), b_last=b_last.data_ptr(), wav=y.data_ptr())
    else:
        y = torch.empty_like(x)
        t = torch.empty((B, lin, c // 2), device=x.device, dtype=x.dtype) if want_t else None
        _fill(p, y=y.data_ptr(), t_act=L.ptr(t), post_elu=1 if post_elu else 0)
    _fill(p, x=x.data_ptr(), w_sc=sc_f.data_ptr(), b_sc=sc_b.data_ptr(), w1=b1_f.data_ptr(),
          b1=b1_b.data_ptr(), w3=b3_f.data_ptr(), b3=b3_b.data_ptr(), B=B, L=lin, C=c)
    if PROFILE is None:
        L.check(L.lib().ga_resunit_fwd(C.byref(p), L.dt(x), L.stream()), "ga_resunit_fwd")
        return y, t
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(L.lib().ga_resunit_fwd(C.byref(p), L.dt(x), L.stream()), "ga_resunit_fwd")
    e1.record()
    flops = 2.0 * B * lin * (c * c + 3 * c * (c // 2) + (c // 2) * c + (7 * c if last is not None else 0))
    PROFILE.append((L.KERNEL_RESUNIT, flops, e0, e1, (B, c, c if last is None else 1, 3, lin, 1, 1, L.ACT_ELU), flops))
    return y, t


FUSE_RESUNIT_BWD = os.environ.get("GA_FUSE_RESUNIT_BWD", "1") != "0"  # False: the unit's data gradient as three ga_conv launches + the pad fix-up


def resunit_backward(gy: Tensor, t_act: Tensor, x: Tensor, sc_d: Tensor, b1_d: Tensor, b3_d: Tensor) -> Tensor:
    """dL/dx of one residual unit in one launch (ga_resunit_bwd) from dL/dy, the saved ELU(hidden) and x."""
    L.require_cuda(gy, t_act, x, sc_d, b1_d, b3_d)
    B, lin, c = x.shape
    assert gy.shape == x.shape and tuple(t_act.shape) == (B, lin, c // 2)
    for a in (gy, t_act, x, sc_d, b1_d, b3_d):
        assert a.is_contiguous() and a.dtype == torch.bfloat16
    assert sc_d.numel() >= c * c and b1_d.numel() >= c * 3 * (c // 2) and b3_d.numel() >= (c // 2) * c
    gx = torch.empty_like(x)
    p = L.ResunitBwdParams()
    _fill(p, gy=gy.data_ptr(), t_act=t_act.data_ptr(), x=x.data_ptr(), gx=gx.data_ptr(), w_sc_d=sc_d.data_ptr(), w1_d=b1_d.data_ptr(),
          w3_d=b3_d.data_ptr(), B=B, L=lin, C=c)
    if PROFILE is None:
        L.check(L.lib().ga_resunit_bwd(C.byref(p), L.dt(x), L.stream()), "ga_resunit_bwd")
        return gx
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(L.lib().ga_resunit_bwd(C.byref(p), L.dt(x), L.stream()), "ga_resunit_bwd")
    e1.record()
    flops = 2.0 * B * lin * (c * c + 3 * c * (c // 2) + (c // 2) * c)
    PROFILE.append((L.KERNEL_RESUNIT_BWD, flops, e0, e1, (B, c, c, 3, lin, 1, 1, 0), flops))
    return gx


FUSE_DUAL = os.environ.get("GA_FUSE_DUAL", "1") != "0"  # False: the 256-channel unit's shortcut and last conv stay two ga_conv launches


def dual_k1_ok(x: Tensor, g_sc: ConvGeom, g3: ConvGeom) -> bool:
    """True when ga_conv_dual_k1 takes the unit's closing sum: bf16, shortcut 256 -> 256 k1 and last conv 128 -> 256 k1."""
    return (FUSE_DUAL and x.dtype == torch.bfloat16 and _IMPL in (L.IMPL_AUTO, L.IMPL_MFMA)
            and all(g.kind == "cconv" and g.k == 1 and g.dilation == 1 for g in (g_sc, g3))
            and (g_sc.cin, g_sc.cout, g3.cin, g3.cout) == (256, 256, 128, 256))


def dual_k1_forward(x: Tensor, t_act: Tensor, sc_f: Tensor, sc_b: Tensor, b3_f: Tensor, b3_b: Tensor, post_elu: bool) -> Tensor:
    """[ELU](W_sc x + W3 t_act + b_sc + b3) in one launch (ga_conv_dual_k1); t_act holds ELU(hidden)."""
    L.require_cuda(x, t_act, sc_f, b3_f)
    B, lin, ca = x.shape
    assert x.is_contiguous() and t_act.is_contiguous() and tuple(t_act.shape) == (B, lin, 128) and ca == 256
    assert x.dtype == t_act.dtype == sc_f.dtype == b3_f.dtype == torch.bfloat16 and sc_b.dtype == b3_b.dtype == torch.float32
    y = torch.empty_like(x)
    p = L.DualK1Params()
    _fill(p, xa=x.data_ptr(), wa=sc_f.data_ptr(), ba=sc_b.data_ptr(), xb=t_act.data_ptr(), wb=b3_f.data_ptr(), bb=b3_b.data_ptr(),
          y=y.data_ptr(), rows=B * lin, Ca=256, Cb=128, M=256, post_elu=1 if post_elu else 0)
    if PROFILE is None:
        L.check(L.lib().ga_conv_dual_k1(C.byref(p), L.dt(x), L.stream()), "ga_conv_dual_k1")
        return y
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(L.lib().ga_conv_dual_k1(C.byref(p), L.dt(x), L.stream()), "ga_conv_dual_k1")
    e1.record()
    flops = 2.0 * B * lin * 256 * 384
    PROFILE.append((L.KERNEL_DUAL_K1, flops, e0, e1, (B, 384, 256, 1, lin, 1, 1, 0), flops))
    return y


class ResBlockFrozen(torch.autograd.Function):
    """SEANet residual unit with frozen weights (HF modeling_encodec.py:320-343: shortcut(x) + conv_k1(ELU(conv_k(ELU(x))))) as ONE
    autograd node.  As three ConvFrozen nodes, autograd sums the two gradients that reach x (shortcut and block) with an
    element-wise pass over the unit's input -- the largest activations of the step; here the shortcut's data-gradient launch
    takes the block's gradient as its residual operand and adds it in its epilogue.

    elu_mid: the first conv's epilogue writes ELU(t) instead of t (t is used through ELU only), so the second conv needs no
    ELU pass over its input; its data gradient takes ELU'(t) from that output (GA_DACT_ELU_FROM_OUT).
    elu_out: the unit's output y is consumed through ELU only (by the transposed conv behind it): the second conv's epilogue
    writes ELU(conv + bias + shortcut) (GA_RES_ADD_THEN_ACT).  The node then RETURNS ELU(y) and EXPECTS dL/dy as its incoming
    gradient: its consumer must be a ConvFrozen with pre_act = ACT_ELU_DONE, whose data gradient applies ELU'(y) itself."""

    @staticmethod
    def forward(ctx, x, sc_f, sc_d, sc_b, g_sc: ConvGeom, b1_f, b1_d, b1_b, g1: ConvGeom, b3_f, b3_d, b3_b, g3: ConvGeom,
                elu_mid: bool = False, elu_out: bool = False, x_elu=None):
        """x_elu: ELU(x), already written by x's producer (ConvFrozen also_elu): the first conv reads it instead of a pass over x."""
        x = x.contiguous()
        if (not elu_out or g_sc.cin == 128) and resunit_fusable(x, g_sc, g1, g3):
            # the 32 / 64 / 128-channel units: one pass over x instead of three launches; the hidden tensor is kept, as ELU(t),
            # only when a gradient will come back (elu_out: the 128-channel kernel writes ELU(y), as GA_RES_ADD_THEN_ACT does)
            y, t = resunit_forward(x, sc_f, sc_b, b1_f, b1_b, b3_f, b3_b, want_t=ctx.needs_input_grad[0], post_elu=elu_out)
            ctx.geoms, ctx.elu_mid, ctx.one_launch = (g_sc, g1, g3), True, True
            ctx.save_for_backward(x, t, sc_d, b1_d, b3_d)
            return y
        ctx.one_launch = False
        if elu_mid and dual_k1_ok(x, g_sc, g3):
            # the 256-channel unit: first conv as before (its epilogue writes ELU(t)); shortcut and last conv as one launch over
            # the concatenated K, the shortcut never written
            src = x_elu.contiguous() if x_elu is not None else x
            t = _frozen_forward(g1, src, b1_f, b1_b, ACT_ELU_DONE if x_elu is not None else L.ACT_ELU, post_act=L.ACT_ELU)
            y = dual_k1_forward(x, t, sc_f, sc_b, b3_f, b3_b, post_elu=elu_out)
            ctx.geoms, ctx.elu_mid = (g_sc, g1, g3), True
            ctx.save_for_backward(x, t, sc_d, b1_d, b3_d)
            return y
        sc = _frozen_forward(g_sc, x, sc_f, sc_b, L.ACT_NONE)
        if x_elu is not None:
            t = _frozen_forward(g1, x_elu.contiguous(), b1_f, b1_b, ACT_ELU_DONE, post_act=L.ACT_ELU if elu_mid else L.ACT_NONE)
        else:
            t = _frozen_forward(g1, x, b1_f, b1_b, L.ACT_ELU, post_act=L.ACT_ELU if elu_mid else L.ACT_NONE)
        y = _frozen_forward(g3, t, b3_f, b3_b, ACT_ELU_DONE if elu_mid else L.ACT_ELU, res=sc,
                            post_act=L.ACT_ELU if elu_out else L.ACT_NONE, act_after_res=elu_out)
        ctx.geoms, ctx.elu_mid = (g_sc, g1, g3), elu_mid
        ctx.save_for_backward(x, t, sc_d, b1_d, b3_d)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, t, sc_d, b1_d, b3_d = ctx.saved_tensors
        g_sc, g1, g3 = ctx.geoms
        gx = None
        if ctx.needs_input_grad[0]:
            gy = gy.contiguous().to(x.dtype)
            if ctx.one_launch and FUSE_RESUNIT_BWD and x.shape[2] in (32, 64):
                return (resunit_backward(gy, t, x, sc_d, b1_d, b3_d),) + (None,) * 15
            gt = conv_dgrad(g3, gy, b3_d, t.shape[1], dact_mode=L.DACT_ELU_FROM_OUT if ctx.elu_mid else L.DACT_ELU_FROM_IN, dact_src=t)
            gb = conv_dgrad(g1, gt, b1_d, x.shape[1], dact_mode=L.DACT_ELU_FROM_IN, dact_src=x)
            gx = conv_dgrad(g_sc, gy, sc_d, x.shape[1], res=gb)
        return (gx,) + (None,) * 15


class ConvFrozen(torch.autograd.Function):
    """y = conv(pre_act(x)) + bias (+ res) with frozen weights (SEANet decoder): only dx is produced.
    dx = dgrad(gy) * pre_act'(x) (+ the residual branch's gradient when given through `gres_into`)."""

    @staticmethod
    def forward(ctx, x, res, wp_f, wp_d, bias, geom: ConvGeom, pre_act: int, out_f32: bool = False, also_elu: bool = False):
        """also_elu: returns (y, ELU(y)); the second output carries no gradient (its consumer differentiates through y)."""
        x = x.contiguous()
        y = _frozen_forward(geom, x, wp_f, bias, pre_act, res, out_f32, elu_out2=also_elu)
        ctx.geom, ctx.pre_act = geom, pre_act
        ctx.x_dtype = x.dtype
        ctx.save_for_backward(x if pre_act else None, wp_d)
        ctx.lin = x.shape[1]
        if also_elu:
            ctx.mark_non_differentiable(y[1])
        return y

    @staticmethod
    def backward(ctx, gy, gy2=None):
        x, wp_d = ctx.saved_tensors
        gy = gy.contiguous().to(ctx.x_dtype)
        gx = None
        if ctx.needs_input_grad[0]:
            if ctx.pre_act == ACT_ELU_DONE:  # x holds ELU(pre-activation): the returned gradient is w.r.t. the pre-activation
                gx = conv_dgrad(ctx.geom, gy, wp_d, ctx.lin, dact_mode=L.DACT_ELU_FROM_OUT, dact_src=x)
            elif ctx.pre_act == L.ACT_ELU:
                gx = conv_dgrad(ctx.geom, gy, wp_d, ctx.lin, dact_mode=L.DACT_ELU_FROM_IN, dact_src=x)
            elif ctx.pre_act == L.ACT_NONE:
                gx = conv_dgrad(ctx.geom, gy, wp_d, ctx.lin)
            else:
                raise NotImplementedError
        gres = gy if ctx.needs_input_grad[1] else None
        return gx, gres, None, None, None, None, None, None, None


# ================================================================================================
GN_FUSED = True  # tests switch this off to compare against the two-launch path


class GnFilmAct(torch.autograd.Function):
    """GroupNorm(G) -> FiLM (x*(1+a)+b) -> act (+ residual): lab3_codec_models.py:20-25,38-47,114-115."""

    @staticmethod
    def forward(ctx, x, gamma, beta, ab, res, groups: int, act: int, eps: float):
        x = x.contiguous()
        B, T, Cn = x.shape
        mean = torch.empty((B * groups,), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        lib = L.lib()
        y = torch.empty_like(x)
        if ab is not None:
            ab = ab.contiguous()
            assert ab.dtype == torch.float32 and tuple(ab.shape) == (B, 2 * Cn)
        if res is not None:
            res = res.contiguous()
        rc = lib.ga_gn_film_act_fused(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), L.ptr(ab), L.ptr(res), y.data_ptr(),
                                      mean.data_ptr(), rstd.data_ptr(), L.dt(x), B, T, Cn, groups, eps, act, L.stream()) if GN_FUSED else 4
        if rc == 4:  # GA_ERR_UNSUPPORTED: statistics and the normalise/FiLM/activation pass as two launches
            L.check(lib.ga_gn_stats(x.data_ptr(), L.dt(x), B, T, Cn, groups, eps, mean.data_ptr(), rstd.data_ptr(), L.stream()), "ga_gn_stats")
            L.check(lib.ga_gn_film_act_fwd(x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                           L.ptr(ab), L.ptr(res), y.data_ptr(), L.dt(x), B, T, Cn, groups, act, L.stream()),
                    "ga_gn_film_act_fwd")
        else:
            L.check(rc, "ga_gn_film_act_fused")
        ctx.save_for_backward(x, mean, rstd, gamma, beta, ab)
        ctx.groups, ctx.act, ctx.has_res = groups, act, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, beta, ab = ctx.saved_tensors
        dy = dy.contiguous()
        B, T, Cn = x.shape
        dx = torch.empty_like(x)
        nab = ab.numel() if ab is not None else 0
        zbuf = zeros_f32(2 * gamma.numel() + nab, gamma.device)  # one buffer for all three
        dgamma = zbuf[:gamma.numel()].view_as(gamma)
        dbeta = zbuf[gamma.numel():2 * gamma.numel()].view_as(beta)
        dab = zbuf[2 * gamma.numel():].view_as(ab) if ab is not None else None
        L.check(L.lib().ga_gn_film_act_bwd(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                           beta.data_ptr(), L.ptr(ab), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                           L.ptr(dab), L.dt(x), B, T, Cn, ctx.groups, ctx.act, L.stream()), "ga_gn_film_act_bwd")
        return dx, dgamma, dbeta, dab, (dy if ctx.has_res else None), None, None, None


class TanhResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, q, rscale: float):
        raw = raw.contiguous()
        q = q.contiguous()
        y = torch.empty_like(raw)
        L.check(L.lib().ga_tanh_residual_fwd(raw.data_ptr(), q.data_ptr(), y.data_ptr(), L.dt(raw), raw.numel(), rscale, L.stream()),
                "ga_tanh_residual_fwd")
        ctx.save_for_backward(raw)
        ctx.rscale = rscale
        return y

    @staticmethod
    def backward(ctx, dy):
        raw, = ctx.saved_tensors
        dy = dy.contiguous()
        draw = torch.empty_like(raw)
        L.check(L.lib().ga_tanh_residual_bwd(dy.data_ptr(), raw.data_ptr(), draw.data_ptr(), L.dt(raw), raw.numel(), ctx.rscale,
                                             L.stream()), "ga_tanh_residual_bwd")
        return draw, (dy if ctx.needs_input_grad[1] else None), None


class AvgPool4s2(torch.autograd.Function):
    """nn.AvgPool1d(4, 2, 1) on [B, L] rows (lab3_codec_models.py:159)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, Ln = x.shape
        Lo = (Ln + 2 - 4) // 2 + 1
        y = torch.empty((B, Lo), device=x.device, dtype=x.dtype)
        L.check(L.lib().ga_avgpool1d_fwd(x.data_ptr(), y.data_ptr(), L.dt(x), B, Ln, Lo, L.stream()), "ga_avgpool1d_fwd")
        ctx.shape = (B, Ln, Lo)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Ln, Lo = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((B, Ln), device=dy.device, dtype=dy.dtype)
        L.check(L.lib().ga_avgpool1d_bwd(dy.data_ptr(), dx.data_ptr(), L.dt(dy), B, Ln, Lo, L.stream()), "ga_avgpool1d_bwd")
        return dx


# ================================================================================================
def _flat_view(t: Tensor) -> Tuple[Tensor, bool]:
    """(contiguous tensor holding t's elements, whether it is t.permute(0,2,1)): NLC buffers shown as [B,C,L]
    views are reduced in memory order without a copy."""
    if t.is_contiguous():
        return t, False
    if t.dim() == 3 and t.permute(0, 2, 1).is_contiguous():
        return t.permute(0, 2, 1), True
    return t.contiguous(), False


class ReduceSum(torch.autograd.Function):
    """sum_i f(a_i, b_i) as a 1-element fp32 tensor; gradient flows to `a` (and, for |a-b|, to `b` if it needs one)."""

    @staticmethod
    def forward(ctx, a, b, op: int, scale: float = 1.0):
        L.require_cuda(a)
        af, perm = _flat_view(a)
        bf = None
        if b is not None:
            bf, perm_b = _flat_view(b)
            if perm_b != perm:
                bf = (b.permute(0, 2, 1) if perm else b).contiguous()
            assert bf.shape == af.shape and bf.dtype == af.dtype, "operands must share shape and dtype"
        out = zeros_f32(1, a.device)
        L.check(L.lib().ga_reduce(af.data_ptr(), L.ptr(bf), L.dt(af), af.numel(), op, float(scale), out.data_ptr(), L.stream()), "ga_reduce")
        ctx.save_for_backward(af, bf)
        ctx.op, ctx.scale = op, float(scale)
        ctx.permuted = perm
        return out.view(())  # 0-dim: indexing a 1-element result would add a select + its zero-fill backward per loss term

    @staticmethod
    def backward(ctx, gout):
        af, bf = ctx.saved_tensors
        ga = torch.empty_like(af)
        gout = gout.contiguous().float()
        L.check(L.lib().ga_reduce_grad(af.data_ptr(), L.ptr(bf), L.dt(af), af.numel(), ctx.op, gout.data_ptr(), ctx.scale,
                                       ga.data_ptr(), 0, L.stream()), "ga_reduce_grad")
        if ctx.permuted:
            ga = ga.permute(0, 2, 1)
        gb = None
        if bf is not None and ctx.needs_input_grad[1]:
            if ctx.op != L.RED_ABSDIFF:
                raise NotImplementedError("gradient w.r.t. the second operand exists for |a-b| only")
            gb = ga.neg()
        return ga, gb, None, None


class StftCombine(torch.autograd.Function):
    """(1/R) sum_r [ sqrt(s_r0) / (sqrt(s_r1) + 1e-8) + s_r2 / count_r ] of the stacked per-resolution sums [R, 3]
    (spectral convergence + log-magnitude L1, reference lab3_codec_train.py:161-176) in one launch each way."""

    @staticmethod
    def forward(ctx, sums, inv_count):
        sums = sums.contiguous().float()
        out = torch.empty((1,), device=sums.device, dtype=torch.float32)
        L.check(L.lib().ga_stft_combine(sums.data_ptr(), inv_count.data_ptr(), sums.shape[0], None, out.data_ptr(), None, L.stream()),
                "ga_stft_combine")
        ctx.save_for_backward(sums, inv_count)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        sums, inv_count = ctx.saved_tensors
        ds = torch.empty_like(sums)
        gout = gout.reshape(1).float().contiguous()
        L.check(L.lib().ga_stft_combine(sums.data_ptr(), inv_count.data_ptr(), sums.shape[0], gout.data_ptr(), None, ds.data_ptr(),
                                        L.stream()), "ga_stft_combine")
        return ds, None


_CONST_VECS: dict = {}


def const_vec(values, device) -> Tensor:
    """A cached fp32 device vector of Python floats (loss weights, reciprocal counts): built once per distinct tuple."""
    key = (torch.device(device), tuple(float(v) for v in values))
    t = _CONST_VECS.get(key)
    if t is None:
        if len(_CONST_VECS) > 256:
            _CONST_VECS.clear()
        t = _CONST_VECS[key] = torch.tensor(key[1], dtype=torch.float32, device=device)
    return t


def weighted_sum(terms, weights) -> Tensor:
    """sum_i weights[i] * terms[i] for 0-dim device tensors: one stack, one multiply, one sum (and two kernels back)
    instead of a multiply and an add per term each way."""
    st = torch.stack([t.reshape(()) for t in terms])
    return (st * const_vec(weights, st.device)).sum()


class HingeDJoined(torch.autograd.Function):
    """mean(relu(1 - logits[:b0])) + mean(relu(1 + logits[b0:])) for a logits tensor whose batch is [real ; fake]
    (reference hinge_d_loss, lab3_codec_models.py:171-174, on the joined D pass).  One node and one gradient buffer for both
    halves: slicing the logits first would cost a zero-fill, a copy and an add per half in autograd's slice backward."""

    @staticmethod
    def forward(ctx, logits, b0: int):
        L.require_cuda(logits)
        lf, perm = _flat_view(logits)
        per = lf[0].numel()
        n_r, n_f = b0 * per, (lf.shape[0] - b0) * per
        out = zeros_f32(1, lf.device)
        esz = lf.element_size()
        lib = L.lib()
        L.check(lib.ga_reduce(lf.data_ptr(), None, L.dt(lf), n_r, L.RED_RELU1M, 1.0 / n_r, out.data_ptr(), L.stream()), "ga_reduce")
        L.check(lib.ga_reduce(lf.data_ptr() + n_r * esz, None, L.dt(lf), n_f, L.RED_RELU1P, 1.0 / n_f, out.data_ptr(), L.stream()),
                "ga_reduce")
        ctx.save_for_backward(lf)
        ctx.n_r, ctx.n_f, ctx.perm = n_r, n_f, perm
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        lf, = ctx.saved_tensors
        ga = torch.empty_like(lf)
        gout = gout.contiguous().float()
        esz = lf.element_size()
        lib = L.lib()
        L.check(lib.ga_reduce_grad(lf.data_ptr(), None, L.dt(lf), ctx.n_r, L.RED_RELU1M, gout.data_ptr(), 1.0 / ctx.n_r,
                                   ga.data_ptr(), 0, L.stream()), "ga_reduce_grad")
        L.check(lib.ga_reduce_grad(lf.data_ptr() + ctx.n_r * esz, None, L.dt(lf), ctx.n_f, L.RED_RELU1P, gout.data_ptr(),
                                   1.0 / ctx.n_f, ga.data_ptr() + ctx.n_r * esz, 0, L.stream()), "ga_reduce_grad")
        return (ga.permute(0, 2, 1) if ctx.perm else ga), None


def reduce_sum(a: Tensor, b: Optional[Tensor], op: int, scale: float = 1.0) -> Tensor:
    """scale * sum_i f(a_i, b_i) as a 0-dim fp32 tensor (the scale -- typically 1/numel -- is applied inside the kernel)."""
    return ReduceSum.apply(a, b, op, scale)


class LatentSums(torch.autograd.Function):
    """[sum|d|, sum_t|d[t+1]-d[t]|, per-sample sum|d| ...] for d = q_hat - q_src on NLC tensors."""

    @staticmethod
    def forward(ctx, qh, qs):
        qh = qh.contiguous()
        qs = qs.contiguous()
        assert qh.shape == qs.shape and qh.dtype == qs.dtype
        B, T, Cn = qh.shape
        out = zeros_f32(2 + B, qh.device)
        L.check(L.lib().ga_latent_loss_fwd(qh.data_ptr(), qs.data_ptr(), L.dt(qh), B, T, Cn, out.data_ptr(), L.stream()),
                "ga_latent_loss_fwd")
        ctx.save_for_backward(qh, qs)
        return out

    @staticmethod
    def backward(ctx, gout):
        qh, qs = ctx.saved_tensors
        B, T, Cn = qh.shape
        gout = gout.contiguous().float()
        dq = torch.empty_like(qh)
        L.check(L.lib().ga_latent_loss_bwd(qh.data_ptr(), qs.data_ptr(), L.dt(qh), B, T, Cn, gout.data_ptr(), dq.data_ptr(),
                                           L.stream()), "ga_latent_loss_bwd")
        return dq, None


# ================================================================================================
# MR-STFT: windowed DFT basis as conv weights (cached per resolution / dtype / device)
_BASIS_CACHE = {}


def _dft_basis(n_fft: int, hop: int, win: int, dtype: torch.dtype, device, rep: int):
    key = (n_fft, hop, win, dtype, str(device), rep)
    if key in _BASIS_CACHE:
        return _BASIS_CACHE[key]
    if n_fft % hop:
        raise ValueError(f"MR-STFT kernel needs hop | n_fft, got n_fft={n_fft} hop={hop}")
    nb = n_fft // 2 + 1
    M = (2 * nb + 31) // 32 * 32
    J = n_fft // hop
    n = torch.arange(win, dtype=torch.float64)
    window = 0.5 - 0.5 * torch.cos(2.0 * math.pi * n / win)
    if win < n_fft:
        left = (n_fft - win) // 2
        window = torch.nn.functional.pad(window, (left, n_fft - win - left))
    k = torch.arange(n_fft, dtype=torch.float64)
    f = torch.arange(nb, dtype=torch.float64)
    ang = 2.0 * math.pi * torch.outer(f, k) / n_fft
    basis = torch.zeros((M, n_fft), dtype=torch.float64)
    basis[:nb] = torch.cos(ang) * window
    basis[nb:2 * nb] = -torch.sin(ang) * window
    bf = basis.reshape(M, J, hop)  # [m][j][c], k = j*hop + c
    if rep == 1:
        w_f = bf.to(torch.float32).to(dtype)
    else:  # bf16 hi/lo split: x*b ~= xhi*bhi + xlo*bhi + xhi*blo  (channels = [hi | lo | hi] x [bhi | bhi | blo])
        hi = bf.to(torch.float32).to(torch.bfloat16)
        lo = (bf - hi.to(torch.float64)).to(torch.float32).to(torch.bfloat16)
        w_f = torch.cat([hi, hi, lo], dim=2)
    # data-gradient form (stride-1 conv): Wd[c][j'][m] = basis[m][J-1-j'][c]
    w_d = bf.flip(1).permute(2, 1, 0).contiguous().to(torch.float32).to(dtype)
    out = (w_f.contiguous().to(device), w_d.to(device), nb, M, J)
    _BASIS_CACHE[key] = out
    return out


class StftSums(torch.autograd.Function):
    """[sum (my-mx)^2, sum my^2, sum |log mx - log my|] for one resolution; x, y: [B, S].
    split=False: exact fp32 DFT GEMM (direct kernel).  split=True: bf16 MFMA GEMM on the hi/lo split of the fp32 signal
    and basis (x*b ~= xhi*bhi + xlo*bhi + xhi*blo, i.e. 3x the K extent, ~2^-16 relative error) with fp32 output; the
    magnitude / log / reduction kernels are fp32 either way."""

    @staticmethod
    def forward(ctx, x, y, n_fft: int, hop: int, win: int, split: bool, mag_add: bool = False):
        """mag_add: magnitudes floored as |X| + 1e-8 (lab3_losses.multi_resolution_stft_loss) instead of clamp_min(1e-8)."""
        L.require_cuda(x, y)
        ctx.in_dtype = x.dtype
        x = x.contiguous().float()
        y = y.contiguous().float()
        B, S = x.shape
        lib = L.lib()
        gdt = torch.bfloat16 if split else torch.float32
        rep = 3 if split else 1
        w_f, w_d, nb, M, J = _dft_basis(n_fft, hop, win, gdt, x.device, rep)
        P = n_fft // 2
        rows = (S + 2 * P + hop - 1) // hop
        frames = 1 + S // hop
        specs = []
        for sig in (x, y):
            xp = torch.empty((B, rows, hop * rep), device=x.device, dtype=gdt)
            L.check(lib.ga_stft_pad(sig.data_ptr(), xp.data_ptr(), L.dt(sig), L.dt(xp), B, S, n_fft, hop, rows, rep, L.stream()), "ga_stft_pad")
            p = _new_params()
            _fill(p, B=B, Lin=rows, Cin=hop * rep, M=M, Lout=frames, LoutTot=frames, J=J, off0=0, out_f32=1 if split else 0)
            spec = torch.empty((B, frames, M), device=x.device, dtype=torch.float32)
            # algorithmic work of the DFT-basis GEMM: frames x n_fft x 2*(n_fft/2+1) multiply-adds, once (not 3x for the split)
            _launch_conv(p, xp, w_f, spec, impl=_IMPL, alg_flops=2.0 * B * frames * n_fft * 2 * (n_fft // 2 + 1))
            specs.append(spec)
        out = zeros_f32(3, x.device)
        L.check(lib.ga_stft_loss_fwd_mode(specs[0].data_ptr(), specs[1].data_ptr(), L.DT_F32, B * frames, nb, M, out.data_ptr(),
                                          1 if mag_add else 0, L.stream()), "ga_stft_loss_fwd_mode")
        ctx.save_for_backward(specs[0], specs[1], w_d)
        ctx.cfg = (B, S, n_fft, hop, rows, frames, nb, M, J, split)
        ctx.mag_add = bool(mag_add)
        return out

    @staticmethod
    def backward(ctx, gout):
        sx, sy, w_d = ctx.saved_tensors
        B, S, n_fft, hop, rows, frames, nb, M, J, split = ctx.cfg
        lib = L.lib()
        gout = gout.contiguous().float()
        dsx = torch.empty_like(sx)
        L.check(lib.ga_stft_loss_bwd_mode(sx.data_ptr(), sy.data_ptr(), L.DT_F32, B * frames, nb, M, gout.data_ptr(), dsx.data_ptr(),
                                          1 if ctx.mag_add else 0, L.stream()), "ga_stft_loss_bwd_mode")
        if split:
            dsx = dsx.to(torch.bfloat16)  # MFMA operand; accumulation and the output stay fp32
        p = _new_params()
        _fill(p, B=B, Lin=frames, Cin=M, M=hop, Lout=rows, LoutTot=rows, J=J, off0=-(J - 1), out_f32=1 if split else 0)
        dxp = torch.empty((B, rows, hop), device=sx.device, dtype=torch.float32)
        _launch_conv(p, dsx, w_d, dxp, impl=_IMPL, alg_flops=2.0 * B * frames * n_fft * 2 * (n_fft // 2 + 1))
        dx = torch.empty((B, S), device=sx.device, dtype=torch.float32)
        L.check(lib.ga_stft_unpad_bwd(dxp.data_ptr(), dx.data_ptr(), L.DT_F32, L.DT_F32, B, S, n_fft, hop, rows, 0, L.stream()),
                "ga_stft_unpad_bwd")
        return dx.to(ctx.in_dtype), None, None, None, None, None, None


class StftSpec(torch.autograd.Function):
    """(re | im) spectrum [B, frames, M] (fp32) of x [B, S]: torch.stft(center=True, reflect pad, onesided, unnormalised) as the
    reflect-pad kernel + the DFT-basis GEMM of StftSums, with the data gradient back to x.  Returns (spec, nb): columns
    [0, nb) real parts, [nb, 2 nb) imaginary parts, the rest padding.  split: bf16 hi/lo MFMA GEMM (fp32-grade) / exact fp32."""

    @staticmethod
    def forward(ctx, x, n_fft: int, hop: int, win: int, split: bool):
        L.require_cuda(x)
        ctx.in_dtype = x.dtype
        x = x.contiguous().float()
        B, S = x.shape
        lib = L.lib()
        gdt = torch.bfloat16 if split else torch.float32
        rep = 3 if split else 1
        w_f, w_d, nb, M, J = _dft_basis(n_fft, hop, win, gdt, x.device, rep)
        rows = (S + 2 * (n_fft // 2) + hop - 1) // hop
        frames = 1 + S // hop
        xp = torch.empty((B, rows, hop * rep), device=x.device, dtype=gdt)
        L.check(lib.ga_stft_pad(x.data_ptr(), xp.data_ptr(), L.dt(x), L.dt(xp), B, S, n_fft, hop, rows, rep, L.stream()), "ga_stft_pad")
        p = _new_params()
        _fill(p, B=B, Lin=rows, Cin=hop * rep, M=M, Lout=frames, LoutTot=frames, J=J, off0=0, out_f32=1 if split else 0)
        spec = torch.empty((B, frames, M), device=x.device, dtype=torch.float32)
        _launch_conv(p, xp, w_f, spec, impl=_IMPL, alg_flops=2.0 * B * frames * n_fft * 2 * nb)
        ctx.save_for_backward(w_d)
        ctx.cfg = (B, S, n_fft, hop, rows, frames, nb, M, J, split)
        return spec

    @staticmethod
    def backward(ctx, dspec):
        w_d, = ctx.saved_tensors
        B, S, n_fft, hop, rows, frames, nb, M, J, split = ctx.cfg
        lib = L.lib()
        dsx = dspec.contiguous().float()
        if split:
            dsx = dsx.to(torch.bfloat16)
        p = _new_params()
        _fill(p, B=B, Lin=frames, Cin=M, M=hop, Lout=rows, LoutTot=rows, J=J, off0=-(J - 1), out_f32=1 if split else 0)
        dxp = torch.empty((B, rows, hop), device=dsx.device, dtype=torch.float32)
        _launch_conv(p, dsx, w_d, dxp, impl=_IMPL, alg_flops=2.0 * B * frames * n_fft * 2 * nb)
        dx = torch.empty((B, S), device=dsx.device, dtype=torch.float32)
        L.check(lib.ga_stft_unpad_bwd(dxp.data_ptr(), dx.data_ptr(), L.DT_F32, L.DT_F32, B, S, n_fft, hop, rows, 0, L.stream()),
                "ga_stft_unpad_bwd")
        return dx.to(ctx.in_dtype), None, None, None, None


def stft_bins(n_fft: int) -> Tuple[int, int]:
    """(nb, M): bins and padded row width of StftSpec's output."""
    nb = n_fft // 2 + 1
    return nb, (2 * nb + 31) // 32 * 32


class PowerSpec(torch.autograd.Function):
    """power [rows, ldp] = re^2 + im^2 of a (re | im) spectrum [rows, ld] (columns >= nb zero): ga_power_spec_fwd / _bwd."""

    @staticmethod
    def forward(ctx, spec, nb: int, ldp: int):
        L.require_cuda(spec)
        spec = spec.contiguous()
        assert spec.dtype == torch.float32
        ld = int(spec.shape[-1])
        frames = spec.numel() // ld
        pw = torch.empty(spec.shape[:-1] + (ldp,), device=spec.device, dtype=torch.float32)
        L.check(L.lib().ga_power_spec_fwd(spec.data_ptr(), frames, nb, ld, pw.data_ptr(), ldp, L.stream()), "ga_power_spec_fwd")
        ctx.save_for_backward(spec)
        ctx.cfg = (nb, ld, ldp, frames)
        return pw

    @staticmethod
    def backward(ctx, dpw):
        spec, = ctx.saved_tensors
        nb, ld, ldp, frames = ctx.cfg
        dpw = dpw.contiguous().float()
        dspec = torch.empty_like(spec)
        L.check(L.lib().ga_power_spec_bwd(spec.data_ptr(), dpw.data_ptr(), frames, nb, ld, ldp, dspec.data_ptr(), L.stream()), "ga_power_spec_bwd")
        return dspec, None, None


# ================================================================================================
LSTM_PERSIST = True  # bf16, H = 512: one persistent launch per pass (lstm_persist.hip); tests switch it off to compare
_LSTM_WS: dict = {}


def _lstm_workspace(device) -> Tensor:
    """The persistent LSTM kernels' exchange buffer + counters, one per (device, stream): calls on one stream are ordered."""
    device = torch.device(device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _LSTM_WS.get(key)
    if ws is None:
        n = int(L.lib().ga_lstm_persist_workspace_bytes(0))
        ws = _LSTM_WS[key] = torch.zeros((n,), device=device, dtype=torch.uint8)
    return ws


def lstm_persist_error(device=None, clear: bool = True) -> bool:
    """True when ANY persistent LSTM launch on this device since the last cleared report raised the error word (a bounded wait
    ran out: that launch's y / dg1 are invalid).  The word is sticky on the device side -- launches clear only their own abort
    word and counters -- so a timeout in an early launch of an epoch is still seen here.  Synchronises; `clear` re-arms it."""
    hit = False
    for (d, _), ws in _LSTM_WS.items():
        if device is not None and d != torch.device(device):
            continue
        if bool(ws[:4].view(torch.int32).item()):
            hit = True
            if clear:
                ws[:4].zero_()
    return hit


_BUDGET_TO_RESTORE: Optional[int] = None  # the LDS budget that was in force before hold_gemm_lds_budget()


def hold_gemm_lds_budget(nbytes: int) -> None:
    """Gathered GEMMs launched from here on take tiles of at most `nbytes` of LDS (ga_set_gemm_lds_budget) until
    lift_gemm_lds_budget() -- which the next LSTM forward launch calls by itself, so that the budget covers exactly the convolutions
    issued while ANOTHER stream's persistent LSTM launch owns most of every CU's LDS."""
    global _BUDGET_TO_RESTORE
    if _BUDGET_TO_RESTORE is None:
        _BUDGET_TO_RESTORE = int(L.lib().ga_set_gemm_lds_budget(int(nbytes)))


def lift_gemm_lds_budget() -> None:
    global _BUDGET_TO_RESTORE
    if _BUDGET_TO_RESTORE is not None:
        L.lib().ga_set_gemm_lds_budget(_BUDGET_TO_RESTORE)
        _BUDGET_TO_RESTORE = None


class LstmSkip(torch.autograd.Function):
    """y = LSTM2(x) + x for the frozen decoder (HF EncodecLSTM); x, y: [B, T, H] channels-last.
    `w` is the dict prepared by prepare_lstm_weights (frozen, cached by the decoder)."""

    @staticmethod
    def forward(ctx, x, w, n_grad: Optional[int] = None):
        """n_grad: only the first n_grad batch items take part in the backward pass (the rest of the batch rides along
        for the frozen, no-gradient decodes: one chain of step launches instead of two)."""
        L.require_cuda(x)
        x = x.contiguous()
        B, T, H = x.shape
        Bg = B if n_grad is None else int(n_grad)
        assert 0 < Bg <= B
        dtype = x.dtype
        wd = w["dtype"]
        assert wd == dtype, "LSTM weights must be prepared in the activation dtype"
        geom = w["geom_ih0"]
        xg1 = conv_forward(geom, x, w["ih0_f"], w["bias1"])
        need_grad = x.requires_grad
        dev = x.device
        y = torch.empty_like(x)
        h1 = torch.empty((B, T, H), device=dev, dtype=dtype)
        h2 = torch.empty((B, T, H), device=dev, dtype=dtype)
        c = zeros_f32(2 * B * H, dev).view(2, B, H)
        gates = torch.empty((2, Bg, T, 4 * H), device=dev, dtype=torch.float32) if need_grad else None
        cs = torch.empty((2, Bg, T, H), device=dev, dtype=torch.float32) if need_grad else None
        p = L.LstmParams()
        if LSTM_PERSIST and dtype == torch.bfloat16 and H == 512:
            ws = _lstm_workspace(dev)
            p.ws, p.ws_bytes = ws.data_ptr(), ws.numel()
        _fill(p, w1=w["hh0"].data_ptr(), w2=w["cat1"].data_ptr(), bias2=w["bias2"].data_ptr(), xg1=xg1.data_ptr(),
              x_in=x.data_ptr(), y=y.data_ptr(), h1=h1.data_ptr(), h2=h2.data_ptr(), c1=c[0].data_ptr(), c2=c[1].data_ptr(),
              gates1=L.ptr(gates[0]) if need_grad else 0, gates2=L.ptr(gates[1]) if need_grad else 0,
              cs1=L.ptr(cs[0]) if need_grad else 0, cs2=L.ptr(cs[1]) if need_grad else 0, B=B, T=T, H=H,
              act_dtype=L.dt(x), w_dtype=L.dt(h1), B_save=Bg)
        lift_gemm_lds_budget()  # (a caller's budget covers the convs in FRONT of the LSTM launch only)
        L.check(L.lib().ga_lstm_fwd(C.byref(p), L.stream()), "ga_lstm_fwd")
        if need_grad:
            ctx.save_for_backward(gates, cs)
        ctx.w = w
        ctx.shape = (B, T, H)
        ctx.Bg = Bg
        return y

    @staticmethod
    def backward(ctx, dy):
        gates, cs = ctx.saved_tensors
        w = ctx.w
        Bfull, T, H = ctx.shape
        B = ctx.Bg
        dy = dy[:B].contiguous()
        dev, dtype = dy.device, dy.dtype
        dg = torch.empty((2, B, T, 4 * H), device=dev, dtype=dtype)
        dc = zeros_f32(2 * B * H, dev).view(2, B, H)
        p = L.LstmBwdParams()
        if LSTM_PERSIST and dtype == torch.bfloat16 and H == 512:
            ws = _lstm_workspace(dev)
            p.ws, p.ws_bytes = ws.data_ptr(), ws.numel()
        _fill(p, w1t=w["hh0_t"].data_ptr(), w2t=w["cat1_t"].data_ptr(), dy=dy.data_ptr(), gates1=gates[0].data_ptr(),
              gates2=gates[1].data_ptr(), cs1=cs[0].data_ptr(), cs2=cs[1].data_ptr(), dg1=dg[0].data_ptr(), dg2=dg[1].data_ptr(),
              dc1=dc[0].data_ptr(), dc2=dc[1].data_ptr(), B=B, T=T, H=H, act_dtype=L.dt(dy), w_dtype=L.dt(dg))
        L.check(L.lib().ga_lstm_bwd(C.byref(p), L.stream()), "ga_lstm_bwd")
        dx = conv_dgrad(w["geom_ih0"], dg[0], w["ih0_d"], T, res=dy)  # dg1 @ W_ih_l0 + dy (skip path)
        if B < Bfull:
            dx = torch.cat([dx, torch.zeros((Bfull - B, T, H), device=dev, dtype=dtype)], dim=0)
        return dx, None, None


def prepare_lstm_weights(lstm: torch.nn.LSTM, dtype: torch.dtype) -> dict:
    """Frozen nn.LSTM(H, H, 2) parameters -> the layouts ga_lstm_fwd / ga_lstm_bwd stream (done once per dtype)."""
    if lstm.num_layers != 2 or lstm.bidirectional or lstm.input_size != lstm.hidden_size:
        raise ValueError("expected nn.LSTM(H, H, num_layers=2)")
    H = lstm.hidden_size
    with torch.no_grad():
        ih0, hh0 = lstm.weight_ih_l0.detach().float(), lstm.weight_hh_l0.detach().float()
        ih1, hh1 = lstm.weight_ih_l1.detach().float(), lstm.weight_hh_l1.detach().float()
        geom = ConvGeom("conv", H, 4 * H, 1)
        w3 = ih0.reshape(4 * H, H, 1).contiguous()
        return {
            "dtype": dtype, "geom_ih0": geom,
            "ih0_f": pack_weight(geom, w3, geom.fwd_pack, dtype), "ih0_d": pack_weight(geom, w3, geom.dgrad_pack, dtype),
            "bias1": (lstm.bias_ih_l0 + lstm.bias_hh_l0).detach().float().contiguous(),
            "hh0": hh0.to(dtype).contiguous(), "cat1": torch.cat([ih1, hh1], dim=1).to(dtype).contiguous(),
            "bias2": (lstm.bias_ih_l1 + lstm.bias_hh_l1).detach().float().contiguous(),
            "hh0_t": hh0.t().to(dtype).contiguous(), "cat1_t": torch.cat([ih1.t(), hh1.t()], dim=0).to(dtype).contiguous(),
        }
