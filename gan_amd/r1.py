"""Explicit R1 gradient penalty for the multi-scale waveform discriminator.

Reference: /root/reference/lab 3/src/lab3_codec_train.py:429-442
    real_sum = sum_s mean(logits_s);  grad = d real_sum / d x_real (create_graph);  r1 = mean_b sum grad^2
and loss_d += 0.5*gamma*r1.  Autograd's double backward is replaced by the closed form of that second-order
pass, built from the same convolution kernels.  With the LeakyReLU masks M_l (piecewise constant, so their own
derivative is zero almost everywhere - torch's double backward agrees) and W_l the effective weights,

    grad_x^(s) = W_1^T M_1 W_2^T M_2 ... M_5 W_out^T g_out ,   g_out = 1 / numel(logits_s)

  1. backward chain: ga_l = M_l * dgrad_{l+1}(ga_{l+1})   (masked gradient at every layer), grad_x summed over scales
     through the transposed average pools;
  2. u_x = coef * (2/B) * grad_x;  forward chain through the linearised net: u_l = M_l * conv_l(u_{l-1}) (no bias);
  3. d r1 / d W_l = wgrad(g = ga_l, x = u_{l-1});  weight-norm backward gives d/dv, d/dg; biases get nothing.
The parameter gradients are ACCUMULATED into p.grad (call after zero_grad and before/after loss_d.backward()).
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import _lib as L
from . import ops


def _accumulate(p: torch.nn.Parameter, g: torch.Tensor) -> None:
    if p.grad is None:
        p.grad = g
    else:
        p.grad.add_(g)


def _wn_grads(conv, geom, dwp: torch.Tensor, coef_applied: bool = True) -> None:
    """packed-forward-layout effective-weight gradient -> (weight_v, weight_g).grad"""
    v, g = conv.weight_v, conv.weight_g
    d0, d1, k = geom.wdims
    lib = L.lib()
    dw = torch.empty_like(v)
    L.check(lib.ga_unpack_grad(dwp.data_ptr(), dw.data_ptr(), geom.fwd_pack, d0, d1, k, geom.stride, geom.padding, 0, L.stream()),
            "ga_unpack_grad")
    _, norm = ops.wn_scale(v.data.contiguous(), g.data.contiguous())
    dv = torch.empty_like(v)
    dg = torch.empty_like(g)
    L.check(lib.ga_wn_bwd(dw.data_ptr(), v.data.contiguous().data_ptr(), g.data.contiguous().data_ptr(), norm.data_ptr(), d0,
                          d1 * k, dv.data_ptr(), dg.data_ptr(), 0, L.stream()), "ga_wn_bwd")
    _accumulate(v, dv)
    _accumulate(g, dg)


@torch.no_grad()
def r1_penalty_and_grads(discriminator, x_real: torch.Tensor, real_outs, coef: float) -> torch.Tensor:
    """Returns r1 (device scalar, un-weighted) and accumulates coef * d r1 / d theta into D's parameter gradients.
    `real_outs` is discriminator(x_real)'s output: its feature maps supply the LeakyReLU masks."""
    L.require_cuda(x_real)
    B = x_real.shape[0]
    dtype = real_outs[0][0].dtype
    n_scales = len(discriminator.discs)
    # per-scale layer lists: (conv module, geom, output feature map [B, L, C] or None for the logit conv)
    gx_scales: List[torch.Tensor] = []
    ga_all = []
    packs_all = []
    for s, disc in enumerate(discriminator.discs):
        logits, feats = real_outs[s]
        feats_nlc = [f.detach().permute(0, 2, 1).contiguous() for f in feats]
        layers = [(blk.conv, blk.geom) for blk in disc.blocks] + [(disc.out, disc._g_out)]
        packs = []
        for conv, geom in layers:
            scale, _ = ops.wn_scale(conv.weight_v.data.contiguous(), conv.weight_g.data.contiguous())
            packs.append((ops.pack_weight(geom, conv.weight_v.data.contiguous(), geom.fwd_pack, dtype, scale),
                          ops.pack_weight(geom, conv.weight_v.data.contiguous(), geom.dgrad_pack, dtype, scale)))
        lens = [x_real.shape[-1]]
        for _ in range(s):
            lens[0] = (lens[0] + 2 - 4) // 2 + 1
        for _, geom in layers:
            lens.append(geom.out_len(lens[-1]))
        # 1. backward chain from g_out = 1/numel(logits)
        Lo = logits.shape[-1]
        ga = torch.full((B, Lo, 1), 1.0 / float(B * Lo), device=x_real.device, dtype=dtype)
        gas = [None] * len(layers)
        gas[-1] = ga
        for li in range(len(layers) - 1, 0, -1):
            conv, geom = layers[li]
            h_prev = feats_nlc[li - 1]
            ga = ops.conv_dgrad(geom, ga, packs[li][1], lens[li], dact_mode=L.DACT_LRELU_FROM_OUT, dact_src=h_prev)
            gas[li - 1] = ga
        gx = ops.conv_dgrad(layers[0][1], ga, packs[0][1], lens[0])  # [B, L_s, 1]
        gx_scales.append(gx.reshape(B, -1))
        ga_all.append((layers, gas, feats_nlc, lens))
        packs_all.append(packs)
    # grad_x = sum over scales through the transposed pools
    grad = gx_scales[-1]
    lib = L.lib()
    for s in range(n_scales - 1, 0, -1):
        Lin = gx_scales[s - 1].shape[1]
        up = torch.empty((B, Lin), device=grad.device, dtype=grad.dtype)
        L.check(lib.ga_avgpool1d_bwd(grad.contiguous().data_ptr(), up.data_ptr(), L.dt(grad), B, Lin, grad.shape[1], L.stream()),
                "ga_avgpool1d_bwd")
        grad = up + gx_scales[s - 1]
    gradf = grad.float()
    r1 = gradf.pow(2).sum() / float(B)
    # 2./3. forward chain of u and the weight gradients
    u = (gradf * (2.0 * float(coef) / float(B))).to(dtype)
    for s in range(n_scales):
        if s > 0:
            Lo = (u.shape[1] + 2 - 4) // 2 + 1
            nu = torch.empty((B, Lo), device=u.device, dtype=u.dtype)
            L.check(lib.ga_avgpool1d_fwd(u.contiguous().data_ptr(), nu.data_ptr(), L.dt(u), B, u.shape[1], Lo, L.stream()),
                    "ga_avgpool1d_fwd")
            u = nu
        layers, gas, feats_nlc, lens = ga_all[s]
        ul = u.unsqueeze(-1).contiguous()
        for li, (conv, geom) in enumerate(layers):
            dwp = ops.conv_wgrad(geom, gas[li].contiguous(), ul)
            _wn_grads(conv, geom, dwp)
            if li + 1 < len(layers):
                p = ops.fwd_params(geom, B, ul.shape[1])
                p.dact_mode = L.DACT_LRELU_FROM_OUT
                nxt = torch.empty((B, p.LoutTot, geom.cout), device=ul.device, dtype=ul.dtype)
                ops._launch_conv(p, ul, packs_all[s][li][0], nxt, dact_src=feats_nlc[li], impl=ops._IMPL)
                ul = nxt
    return r1
