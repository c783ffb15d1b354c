"""2-D convolutions of the mel track on the library's 1-D implicit-GEMM kernels (channels-last activations [B, H, W, C]).

    y[b, oh, ow, m] = sum_{i, j, c} x[b, s*oh + i - p, s*ow + j - p, c] * w[m, c, i, j]

With channels last, the (j, c) window of one kernel row i is CONTIGUOUS along W -- exactly the im2col row the gathered-GEMM kernel
reads through its per-lane addresses -- so only the kernel ROWS need gathering: xr[b, oh, w, i*C + c] = x[b, s*oh + i - p, w, c] (one
concatenation of kh strided row slices, kh / s times the input; zero rows for the padding).  What is left is a 1-D convolution along W
over B*OH items with kh*C input channels and kw taps: `ops.ConvTrain` (ga_conv forward and data gradient, ga_wgrad, bias /
LeakyReLU in the epilogue).  Autograd carries the gradient through the row gather and the weight re-arrangement (views and copies).

Reference layers this serves: nn.Conv2d / nn.ConvTranspose2d in /root/reference/lab 3/src/lab3_models.py:22-27,88,95,101,114,156-164.
"""
from __future__ import annotations

import os

from typing import Optional

import torch
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .ops import ConvGeom


def to_nhwc(h_nchw: torch.Tensor) -> torch.Tensor:
    """[B, C, H, W] (any strides) -> contiguous [B, H, W, C]; free when the tensor already is channels-last."""
    return h_nchw.permute(0, 2, 3, 1).contiguous()


def to_nchw_view(y_nhwc: torch.Tensor) -> torch.Tensor:
    """contiguous [B, H, W, C] -> the [B, C, H, W] view of the same memory (channels-last strides)."""
    return y_nhwc.permute(0, 3, 1, 2)


class RowGather(torch.autograd.Function):
    """xr[b, oh, w, i*C + c] = x[b, s*oh + i - p, w, c] (zero rows for the padding, zero channels up to kc8): one launch forward
    (ga_rowgather_fwd), one backward (its adjoint) -- as tensor ops this was a pad, kh strided slices and a concatenation, and in
    the backward pass a zero fill, a copy and an add per kernel row.  Linear, so its own backward is differentiable again (R1)."""

    @staticmethod
    def forward(ctx, x, kh: int, s: int, p: int, OH: int, kc8: int):
        B, H, W, C = x.shape
        y = torch.empty((B, OH, W, kc8), device=x.device, dtype=x.dtype)
        L.check(L.lib().ga_rowgather_fwd(x.data_ptr(), y.data_ptr(), L.dt(x), B, H, W, C, kh, s, p, OH, kc8, L.stream()), "ga_rowgather_fwd")
        ctx.geom = (B, H, W, C, kh, s, p, OH, kc8)
        return y

    @staticmethod
    def backward(ctx, gy):
        return _RowScatter.apply(gy.contiguous(), ctx.geom), None, None, None, None, None


class _RowScatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, geom):
        B, H, W, C, kh, s, p, OH, kc8 = geom
        gx = torch.empty((B, H, W, C), device=gy.device, dtype=gy.dtype)
        L.check(L.lib().ga_rowgather_bwd(gy.data_ptr(), gx.data_ptr(), L.dt(gy), B, H, W, C, kh, s, p, OH, kc8, L.stream()), "ga_rowgather_bwd")
        ctx.geom = geom
        return gx

    @staticmethod
    def backward(ctx, ggx):
        B, H, W, C, kh, s, p, OH, kc8 = ctx.geom
        return RowGather.apply(ggx.contiguous(), kh, s, p, OH, kc8), None


TWICE_DIFFERENTIABLE = False  # set by `twice_differentiable()`: convolutions become ops.ConvTwice nodes (gradient penalties)


class twice_differentiable:
    """Context: 2-D convolutions issued inside build a graph whose backward can itself be differentiated (`create_graph=True`)."""

    def __enter__(self):
        global TWICE_DIFFERENTIABLE
        self._was, TWICE_DIFFERENTIABLE = TWICE_DIFFERENTIABLE, True

    def __exit__(self, *exc):
        global TWICE_DIFFERENTIABLE
        TWICE_DIFFERENTIABLE = self._was


def conv2d_nhwc(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int, padding: int,
                post_act: int = L.ACT_NONE) -> torch.Tensor:
    """x [B, H, W, C] (fp32 or bf16, contiguous), weight [M, C, kh, kw] and bias [M] (fp32 parameters) -> [B, OH, OW, M] in x's dtype."""
    L.require_cuda(x, weight)
    B, H, W, C = x.shape
    M, Cw, kh, kw = weight.shape
    if Cw != C:
        raise ValueError(f"weight expects {Cw} input channels, got {C}")
    s, p = int(stride), int(padding)
    OH = (H + 2 * p - kh) // s + 1
    if OH < 1 or (W + 2 * p - kw) // s + 1 < 1:
        raise ValueError("convolution window larger than the padded input")
    kc = kh * C
    kc8 = (kc + 7) // 8 * 8  # the GEMM kernels stage 16-byte (8-channel) pieces
    xr = RowGather.apply(x.contiguous(), kh, s, p, OH, kc8)
    w1 = weight.permute(0, 2, 1, 3).reshape(M, kc, kw)  # [m, (i, c), j]
    if kc8 != kc:
        w1 = F.pad(w1, (0, 0, 0, kc8 - kc))
    geom = ConvGeom("conv", kc8, M, kw, s, p, 1)
    if TWICE_DIFFERENTIABLE:
        y = ops.ConvTwice.apply(xr.view(B * OH, W, kc8), w1.contiguous(), bias, geom).view(B, OH, geom.out_len(W), M)
        return F.leaky_relu(y, 0.2) if post_act == L.ACT_LRELU else y
    wp_f, wp_d, carrier = ops.PackTrainable.apply(w1.contiguous(), None, geom, x.dtype)
    y = ops.ConvTrain.apply(xr.view(B * OH, W, kc8), wp_f, wp_d, carrier, bias, geom, post_act, None)
    return y.view(B, OH, geom.out_len(W), M)


# ConvTranspose2d(kernel 4, stride 2, padding 1): y[2t + phi] = sum over the two kernel taps of phase phi.  Along one axis
#   phi = 0: x[t-1] * w[3] + x[t] * w[1]        phi = 1: x[t] * w[2] + x[t+1] * w[0]
# i.e. a 3-tap convolution (offsets -1, 0, +1) whose taps are a subset of the 4-tap kernel: {tap: kernel index} per phase.
FUSED_NORM = os.environ.get("GA_MEL_FUSED_NORM", "1") != "0"  # 0: the norm layers run as the torch modules (A/B and bisecting)


class ChannelNorm(torch.autograd.Function):
    """InstanceNorm2d (groups = B) / BatchNorm2d (groups = 1) on a contiguous channels-last [B, H, W, C] tensor with an optional
    LeakyReLU(0.2) behind it: ga_cnorm_stats -> ga_cnorm_finalize -> ga_cnorm_apply forward, ga_cnorm_bwd_stats -> ga_cnorm_bwd_apply
    backward.  `running` = (running_mean, running_var, momentum) updates BatchNorm's buffers in training; `frozen` = (mean, rstd) uses
    given statistics (BatchNorm in eval mode).  First-order only: gradient penalties keep the torch modules (`twice_differentiable`)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups: int, eps: float, act: int, running, frozen):
        B, H, W, C = x.shape
        HW = H * W
        lib, st = L.lib(), L.stream()
        if frozen is None:
            sums = torch.zeros((B, C, 2), device=x.device, dtype=torch.float64)
            mean = torch.empty((groups, C), device=x.device, dtype=torch.float64)
            rstd = torch.empty_like(mean)
            L.check(lib.ga_cnorm_stats(x.data_ptr(), L.dt(x), B, HW, C, groups, sums.data_ptr(), st), "ga_cnorm_stats")
            rm, rv, mom = running if running is not None else (None, None, 0.0)
            L.check(lib.ga_cnorm_finalize(x.data_ptr(), L.dt(x), B, HW, C, groups, sums.data_ptr(), float(eps), mean.data_ptr(), rstd.data_ptr(),
                                          L.ptr(rm), L.ptr(rv), float(mom), st), "ga_cnorm_finalize")
        else:
            mean, rstd = frozen
        y = torch.empty_like(x)
        L.check(lib.ga_cnorm_apply(x.data_ptr(), y.data_ptr(), L.dt(x), B, HW, C, groups, mean.data_ptr(), rstd.data_ptr(), L.ptr(gamma),
                                   L.ptr(beta), act, st), "ga_cnorm_apply")
        ctx.save_for_backward(x, y if act != L.ACT_NONE else None, mean, rstd, gamma)
        ctx.cfg = (groups, act, frozen is not None, beta is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, mean, rstd, gamma = ctx.saved_tensors
        groups, act, frozen, has_beta = ctx.cfg
        B, H, W, C = x.shape
        HW = H * W
        gy = gy.contiguous()
        lib, st = L.lib(), L.stream()
        gsums = torch.zeros((B, C, 2), device=x.device, dtype=torch.float64)  # per item: few atomics per address; double: their order does not show
        L.check(lib.ga_cnorm_bwd_stats(gy.data_ptr(), x.data_ptr(), L.ptr(y), L.dt(x), B, HW, C, groups, mean.data_ptr(), rstd.data_ptr(), act,
                                       gsums.data_ptr(), st), "ga_cnorm_bwd_stats")
        gsum = gsums.sum(0)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            L.check(lib.ga_cnorm_bwd_apply(gy.data_ptr(), x.data_ptr(), L.ptr(y), gx.data_ptr(), L.dt(x), B, HW, C, groups, mean.data_ptr(),
                                           rstd.data_ptr(), L.ptr(gamma), (gsum if groups == 1 else gsums).data_ptr(), act, int(frozen), st),
                    "ga_cnorm_bwd_apply")
        ggamma = gsum[:, 1].to(gamma.dtype) if gamma is not None and ctx.needs_input_grad[1] else None
        gbeta = gsum[:, 0].to(gamma.dtype) if has_beta and ctx.needs_input_grad[2] else None
        return gx, ggamma, gbeta, None, None, None, None, None


def norm2d_supported(m: torch.nn.Module, h_nchw: torch.Tensor) -> bool:
    """Whether `norm2d_nhwc` takes this layer on this input (otherwise the torch module runs)."""
    if not FUSED_NORM or TWICE_DIFFERENTIABLE or h_nchw.dim() != 4 or h_nchw.dtype not in (torch.float32, torch.bfloat16):
        return False
    B, C, H, W = h_nchw.shape
    if isinstance(m, torch.nn.InstanceNorm2d):
        if m.track_running_stats or (m.affine and m.weight.dtype != torch.float32):
            return False
        groups = B
    elif isinstance(m, torch.nn.BatchNorm2d):
        if not m.track_running_stats or m.momentum is None or (m.affine and m.weight.dtype != torch.float32):
            return False
        groups = 1
    else:
        return False
    return bool(L.lib().ga_cnorm_supported(L.DT_BF16 if h_nchw.dtype == torch.bfloat16 else L.DT_F32, B, H * W, C, groups))


def norm2d_nhwc(m: torch.nn.Module, x: torch.Tensor, post_act: int = L.ACT_NONE) -> torch.Tensor:
    """nn.InstanceNorm2d / nn.BatchNorm2d `m` (its parameters, buffers, eps, momentum and training flag) on a contiguous channels-last
    [B, H, W, C] tensor, with LeakyReLU(0.2) behind it when post_act = ACT_LRELU."""
    gamma, beta = (m.weight, m.bias) if m.affine else (None, None)
    if isinstance(m, torch.nn.InstanceNorm2d):
        return ChannelNorm.apply(x, gamma, beta, x.shape[0], m.eps, post_act, None, None)
    if m.training:
        m.num_batches_tracked += 1
        return ChannelNorm.apply(x, gamma, beta, 1, m.eps, post_act, (m.running_mean, m.running_var, float(m.momentum)), None)
    frozen = (m.running_mean.double().view(1, -1).contiguous(), torch.rsqrt(m.running_var.double() + m.eps).view(1, -1).contiguous())
    return ChannelNorm.apply(x, gamma, beta, 1, m.eps, post_act, None, frozen)


_TAPS = ({0: 3, 1: 1}, {1: 2, 2: 0})
_TAP_INDEX = {}


def _tap_index(device) -> torch.Tensor:
    t = _TAP_INDEX.get(device)
    if t is None:
        t = _TAP_INDEX[device] = torch.tensor([[_TAPS[ph].get(a, 4) for a in range(3)] for ph in (0, 1)], device=device)
    return t


def conv_transpose2d_k4s2p1_nhwc(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], post_act: int = L.ACT_NONE) -> torch.Tensor:
    """x [B, H, W, C], weight [C, M, 4, 4] (ConvTranspose2d layout), bias [M] -> [B, 2H, 2W, M].  The four output phases are four
    2 x 2 convolutions over x; they run as ONE 3 x 3 implicit GEMM onto 4M channels (m, phi_h, phi_w) -- 4 of the 9 taps of each
    phase are structural zeros: 2.25x the multiply-adds of the four separate GEMMs, on layers of 0.1-0.2 GMAC per sample -- and
    a pixel shuffle puts the phases in place."""
    C, M = weight.shape[0], weight.shape[1]
    if tuple(weight.shape[2:]) != (4, 4):
        raise ValueError("kernel 4 x 4 expected")
    # one gather instead of 16 slice assignments (and as many again in the backward pass): kernel index 4 is a zero row / column
    wp = F.pad(weight, (0, 1, 0, 1))                                         # [C, M, 5, 5]
    idx = _tap_index(weight.device)                                          # [phase][tap] -> kernel index
    w3 = wp[:, :, idx[:, :, None, None], idx[None, None, :, :]]              # [C, M, ph, a, pw, b]
    w3 = w3.permute(1, 2, 4, 0, 3, 5)                                        # [M, ph, pw, C, a, b]
    b4 = None if bias is None else bias.repeat_interleave(4)
    y = conv2d_nhwc(x, w3.reshape(4 * M, C, 3, 3), b4, 1, 1, post_act)  # [B, H, W, (m, ph, pw)]
    B, H, W, _ = y.shape
    return y.view(B, H, W, M, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(B, 2 * H, 2 * W, M)
