"""numpy model of the C-ABI semantics of ga_pack_weights / ga_conv / ga_wgrad (include/gan_amd.h).

Test infrastructure: lets the CPU suite check the geometry builders in gan_amd/ops.py (which conv /
transposed-conv / data-gradient maps onto which generalised forward-form problem, and how weights are packed)
against stock torch ops, without a GPU.  The HIP kernels are checked against the same torch ops on the GPU.
"""
from __future__ import annotations

import numpy as np

from gan_amd import _lib as L


def pack_index(mode, s0, s1, k, D0, D1, K, S, P):
    if mode == L.PACK_CONV_FWD:
        return (s0 * K + k) * D1 + s1
    if mode == L.PACK_CONV_DGRAD_S1:
        return (s1 * K + (K - 1 - k)) * D0 + s0
    if mode == L.PACK_CONV_DGRAD_STRIDED:
        jmax = (K + S - 1) // S
        rho = k % S
        r = (rho - P) % S
        jp = jmax - 1 - k // S
        return ((r * D1 + s1) * jmax + jp) * D0 + s0
    if mode == L.PACK_CONVT_FWD:
        r, j = k % S, 1 - k // S
        return ((r * D1 + s1) * 2 + j) * D0 + s0
    return (s0 * K + k) * D1 + s1


def packed_elems(mode, D0, D1, K, S):
    if mode == L.PACK_CONV_DGRAD_STRIDED:
        return S * D1 * ((K + S - 1) // S) * D0
    return D0 * D1 * K


def pack_weights(src: np.ndarray, scale, mode, S, P):
    D0, D1, K = src.shape
    out = np.zeros(packed_elems(mode, D0, D1, K, S), dtype=np.float64)
    for s0 in range(D0):
        for s1 in range(D1):
            for k in range(K):
                v = src[s0, s1, k] * (1.0 if scale is None else scale[s0])
                out[pack_index(mode, s0, s1, k, D0, D1, K, S, P)] = v
    return out


def _act(a, v):
    if a == L.ACT_LRELU:
        return np.where(v > 0, v, 0.2 * v)
    if a == L.ACT_ELU:
        return np.where(v > 0, v, np.expm1(np.minimum(v, 0)))
    return v


def conv(p: L.ConvParams, x, w, y=None, bias=None, res=None, dact_src=None):
    """x [B, Lin, Cin], w flat packed -> y [B, LoutTot, M] (float64)."""
    B, Lin, Cin, M = p.B, p.Lin, p.Cin, p.M
    if y is None:
        y = np.zeros((B, p.LoutTot, M))
    w = np.asarray(w, dtype=np.float64).reshape(-1)
    xs = _act(p.pre_act, np.asarray(x, dtype=np.float64))
    for ph in range(p.nphase):
        wp = w[ph * M * p.J * Cin:(ph + 1) * M * p.J * Cin].reshape(M, p.J, Cin)
        off0 = p.off0 + p.phase_off0[ph]
        for t in range(p.t0, p.t0 + p.Lout):
            orow = t * p.out_stride + ph
            if orow < 0 or orow >= p.LoutTot:
                continue
            acc = np.zeros((B, M))
            for j in range(p.J):
                row = t * p.in_stride + j * p.tap_step + off0
                if row < 0 and p.pad_mode == 1:
                    row = -row
                if 0 <= row < Lin:
                    acc += xs[:, row, :] @ wp[:, j, :].T
            v = acc + (0 if bias is None else bias[None, :])
            v = _act(p.post_act, v)
            if p.dact_mode == L.DACT_LRELU_FROM_OUT:
                v = v * np.where(dact_src[:, orow, :] > 0, 1.0, 0.2)
            elif p.dact_mode == L.DACT_ELU_FROM_IN:
                v = v * np.where(dact_src[:, orow, :] > 0, 1.0, np.exp(np.minimum(dact_src[:, orow, :], 0)))
            if res is not None:
                v = v + res[:, orow, :]
            if p.accumulate:
                v = v + y[:, orow, :]
            y[:, orow, :] = v
    return y
