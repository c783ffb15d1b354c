"""ctypes binding of libgan_amd.so (the C ABI in include/gan_amd.h).

There is no CPU or PyTorch fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libgan_amd.so"

GA_MAX_PHASE = 8
DT_F32, DT_BF16 = 0, 1
ACT_NONE, ACT_LRELU, ACT_ELU, ACT_SILU, ACT_TANH = 0, 1, 2, 3, 4
DACT_NONE, DACT_LRELU_FROM_OUT, DACT_ELU_FROM_IN, DACT_ELU_FROM_OUT = 0, 1, 2, 3
IMPL_AUTO, IMPL_DIRECT, IMPL_MFMA, IMPL_MFMA_WINDOW, IMPL_MFMA_PP, IMPL_MFMA_LDSWIN = 0, 1, 2, 3, 4, 5
IMPL_MFMA_STREAM, IMPL_MFMA_STREAMWIN = 6, 7
KERNEL_GEMM_64x256, KERNEL_GEMM_128x128, KERNEL_GEMM_32x256 = 7, 8, 9
KERNEL_STREAM = 10
KERNEL_GEMM_PP = 11
KERNEL_CONV_WIN = 12
KERNEL_STREAM_WIN = 13
KERNEL_COOP, KERNEL_DUAL_K1 = 16, 17
KERNEL_RESUNIT, KERNEL_RESUNIT_BWD = 14, 15  # ga_resunit_fwd / _bwd (not ga_conv kernels: the benchmark's per-kernel accounting only)
GEMM_KERNELS = (7, 8, 9, 10, 11, 12, 13, 16)  # MFMA kernels whose operands never pass through a conv window in LDS
PACK_CONV_FWD, PACK_CONV_DGRAD_S1, PACK_CONV_DGRAD_STRIDED, PACK_CONVT_FWD, PACK_CONVT_DGRAD = 0, 1, 2, 3, 4
RED_ABSDIFF, RED_RELU1M, RED_RELU1P, RED_SUM, RED_SQ = 0, 1, 2, 3, 4


class ConvParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
        ("dact_src", C.c_void_p),
        ("B", C.c_int), ("Lin", C.c_int), ("Cin", C.c_int), ("M", C.c_int),
        ("Lout", C.c_int), ("LoutTot", C.c_int),
        ("J", C.c_int), ("in_stride", C.c_int), ("tap_step", C.c_int), ("off0", C.c_int),
        ("out_stride", C.c_int), ("t0", C.c_int),
        ("pad_mode", C.c_int), ("pre_act", C.c_int), ("post_act", C.c_int), ("dact_mode", C.c_int),
        ("accumulate", C.c_int), ("nphase", C.c_int),
        ("phase_off0", C.c_int * GA_MAX_PHASE), ("out_f32", C.c_int), ("ksplit", C.c_int), ("ws", C.c_void_p),
        ("y_batch_elems", C.c_int), ("res_mode", C.c_int), ("res_coef", C.c_float), ("res_coef_dev", C.c_void_p), ("y2", C.c_void_p),
    ]


class PackItem(C.Structure):
    """ga_pack_item (include/gan_amd.h)"""
    _fields_ = [("v", C.c_void_p), ("g", C.c_void_p), ("scale", C.c_void_p), ("norm", C.c_void_p), ("wp_f", C.c_void_p),
                ("wp_d", C.c_void_p), ("D0", C.c_int), ("D1", C.c_int), ("K", C.c_int), ("S", C.c_int), ("P", C.c_int),
                ("mode_f", C.c_int), ("mode_d", C.c_int), ("reserved", C.c_int)]


class WFinishItem(C.Structure):
    """ga_wfinish_item (include/gan_amd.h)"""
    _fields_ = [("gpacked", C.c_void_p), ("v", C.c_void_p), ("g", C.c_void_p), ("norm", C.c_void_p), ("dv", C.c_void_p),
                ("dg", C.c_void_p), ("D0", C.c_int), ("D1", C.c_int), ("K", C.c_int), ("accumulate", C.c_int)]


WFINISH_MAX_ROW = 15360
PACK_CONV_FWD = 0
RES_ADD, RES_FM_SIGN, RES_ADD_THEN_ACT = 0, 1, 2
PACK_MULTI_MAX = 24
PACK_CONV_DGRAD_STRIDED = 2


class WgradParams(C.Structure):
    _fields_ = [
        ("g", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p),
        ("B", C.c_int), ("Lin", C.c_int), ("Cin", C.c_int), ("M", C.c_int), ("Lout", C.c_int), ("J", C.c_int),
        ("in_stride", C.c_int), ("tap_step", C.c_int), ("off0", C.c_int),
        ("LgTot", C.c_int), ("g_stride", C.c_int), ("g_phase", C.c_int), ("rows_per_split", C.c_int), ("ws", C.c_void_p),
    ]


class ResunitParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "y", "t_act", "w_sc", "b_sc", "w1", "b1", "w3", "b3")] + \
               [(n, C.c_int) for n in ("B", "L", "C")] + [(n, C.c_void_p) for n in ("w_last", "b_last", "wav")] + [("post_elu", C.c_int)]


class ResunitBwdParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("gy", "t_act", "x", "gx", "w_sc_d", "w1_d", "w3_d")] + [(n, C.c_int) for n in ("B", "L", "C")]


class DualK1Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("xa", "wa", "ba", "xb", "wb", "bb", "y")] + [("rows", C.c_long)] + \
               [(n, C.c_int) for n in ("Ca", "Cb", "M", "post_elu")]


class LstmParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1", "w2", "bias2", "xg1", "x_in", "y", "h1", "h2", "c1", "c2", "gates1", "gates2",
                                          "cs1", "cs2")] + [(n, C.c_int) for n in ("B", "T", "H", "act_dtype", "w_dtype", "B_save")] + \
               [("ws", C.c_void_p), ("ws_bytes", C.c_size_t)]


class LstmBwdParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1t", "w2t", "dy", "gates1", "gates2", "cs1", "cs2", "dg1", "dg2", "dc1", "dc2")] + \
               [(n, C.c_int) for n in ("B", "T", "H", "act_dtype", "w_dtype")] + [("ws", C.c_void_p), ("ws_bytes", C.c_size_t)]


_P, _I, _F, _Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_SIGNATURES = {
    "ga_conv": [C.POINTER(ConvParams), _I, _I, _P],
    "ga_conv_kernel_id": [C.POINTER(ConvParams), _I, _I],
    "ga_conv_gemm_variant": [C.POINTER(ConvParams)],
    "ga_set_gemm_lds_budget": [C.c_int],
    "ga_conv_suggest_ksplit": [C.POINTER(ConvParams), _I, _I],
    "ga_conv_finish": [C.POINTER(ConvParams), _I, _P],
    "ga_conv_splitk_static": [C.POINTER(ConvParams), _I, _P],
    "ga_resunit_fwd": [C.POINTER(ResunitParams), _I, _P],
    "ga_resunit_bwd": [C.POINTER(ResunitBwdParams), _I, _P],
    "ga_conv_dual_k1": [C.POINTER(DualK1Params), _I, _P],
    "ga_wgrad": [C.POINTER(WgradParams), _I, _I, _P],
    "ga_wgrad_multi": [_P, _I, _I, _P],
    "ga_wgrad_workspace_elems": [C.POINTER(WgradParams), _I, _I],
    "ga_wn_scale": [_P, _P, _I, _I, _P, _P, _P],
    "ga_pack_weights": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ga_pack_multi": [_P, _I, _I, _P],
    "ga_wgrad_finish_multi": [_P, _I, _P],
    "ga_unpack_grad": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ga_wn_bwd": [_P, _P, _P, _P, _I, _I, _P, _P, _I, _P],
    "ga_gn_stats": [_P, _I, _I, _I, _I, _I, _F, _P, _P, _P],
    "ga_gn_film_act_fused": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "ga_gn_film_act_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ga_gn_film_act_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ga_tanh_residual_fwd": [_P, _P, _P, _I, _Z, _F, _P],
    "ga_tanh_residual_bwd": [_P, _P, _P, _I, _Z, _F, _P],
    "ga_avgpool1d_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "ga_avgpool1d_bwd": [_P, _P, _I, _I, _I, _I, _P],
    "ga_act_bwd": [_P, _P, _P, _I, _Z, _I, _P],
    "ga_act_fwd": [_P, _P, _I, _Z, _I, _P],
    "ga_rowgather_fwd": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ga_rowgather_bwd": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ga_cnorm_supported": [_I, _I, C.c_long, _I, _I],
    "ga_cnorm_stats": [_P, _I, _I, C.c_long, _I, _I, _P, _P],
    "ga_cnorm_finalize": [_P, _I, _I, C.c_long, _I, _I, _P, _F, _P, _P, _P, _P, _F, _P],
    "ga_cnorm_apply": [_P, _P, _I, _I, C.c_long, _I, _I, _P, _P, _P, _P, _I, _P],
    "ga_cnorm_bwd_stats": [_P, _P, _P, _I, _I, C.c_long, _I, _I, _P, _P, _I, _P, _P],
    "ga_cnorm_bwd_apply": [_P, _P, _P, _P, _I, _I, C.c_long, _I, _I, _P, _P, _P, _P, _I, _I, _P],
    "ga_act_bwd_bias": [_P, _P, _P, _I, _Z, _I, _I, _P, _P],
    "ga_colsum": [_P, _I, _Z, _I, _P, _P],
    "ga_reduce": [_P, _P, _I, _Z, _I, _F, _P, _P],
    "ga_reduce_multi": [_P, _P, _P, _P, _I, _I, _I, _P, _P],
    "ga_reduce_grad": [_P, _P, _I, _Z, _I, _P, _F, _P, _I, _P],
    "ga_latent_loss_fwd": [_P, _P, _I, _I, _I, _I, _P, _P],
    "ga_latent_loss_bwd": [_P, _P, _I, _I, _I, _I, _P, _P, _P],
    "ga_stft_pad": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ga_power_spec_fwd": [_P, _Z, _I, _I, _P, _I, _P],
    "ga_power_spec_bwd": [_P, _P, _Z, _I, _I, _I, _P, _P],
    "ga_stft_combine": [_P, _P, _I, _P, _P, _P, _P],
    "ga_stft_loss_fwd": [_P, _P, _I, _Z, _I, _I, _P, _P],
    "ga_stft_loss_bwd": [_P, _P, _I, _Z, _I, _I, _P, _P, _P],
    "ga_stft_loss_fwd_mode": [_P, _P, _I, _Z, _I, _I, _P, _I, _P],
    "ga_stft_loss_bwd_mode": [_P, _P, _I, _Z, _I, _I, _P, _P, _I, _P],
    "ga_stft_unpad_bwd": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "ga_lstm_fwd": [C.POINTER(LstmParams), _P],
    "ga_lstm_bwd": [C.POINTER(LstmBwdParams), _P],
    "ga_grad_sqnorm": [_P, _Z, _P, _P],
    "ga_adamw_fused": [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _I, _F, _F, _P, _I, _P],
    "ga_adamw_fused_devstep": [_P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _P, _F, _F, _P, _I, _P],
}
EXPORTED = sorted(list(_SIGNATURES) + ["ga_packed_elems", "ga_build_target", "ga_lstm_persist_workspace_bytes"])

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it was not built)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m gan_amd.build` (hipcc, gfx950). "
                "gan_amd has no CPU/PyTorch fallback for its kernels."
            )
        l = C.CDLL(str(LIB_PATH))
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        l.ga_packed_elems.argtypes = [_I, _I, _I, _I, _I]
        l.ga_packed_elems.restype = C.c_size_t
        l.ga_wgrad_workspace_elems.restype = C.c_size_t
        l.ga_lstm_persist_workspace_bytes.argtypes = [_I]
        l.ga_lstm_persist_workspace_bytes.restype = C.c_size_t
        l.ga_build_target.argtypes = []
        l.ga_build_target.restype = C.c_char_p
        _lib = l
    return _lib


_ERR = {1: "GA_ERR_ARG (bad shape / unsupported configuration)", 2: "GA_ERR_LAUNCH (HIP launch failed)",
        3: "GA_ERR_ALIGN (pointer alignment)", 4: "GA_ERR_UNSUPPORTED (this variant does not take the shape)",
        5: "GA_ERR_SCRATCH (problem larger than the library's fixed split-K scratch: use ga_conv with a workspace)"}
SPLITK_STATIC_FLOATS = 8 << 20


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"gan_amd: {what} failed with status {rc}: {_ERR.get(rc, '?')}")


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.bfloat16:
        return DT_BF16
    raise TypeError(f"gan_amd kernels take float32 or bfloat16 tensors, got {t.dtype}")


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*ts) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("gan_amd ops run on the GPU only (HIP kernels); got a CPU tensor. "
                               "There is no CPU fallback - use oracle/ for CPU checks.")
