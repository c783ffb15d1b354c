// C ABI dispatch for the convolution family.  See include/gan_amd.h.
#include "common.h"

int ga_conv_direct_impl(const ConvParams& p, int dtype, hipStream_t s);
int ga_wgrad_direct_impl(const WgradParams& p, int dtype, hipStream_t s);
int ga_conv_mfma_supported(const ConvParams& p);
int ga_conv_mfma_impl(const ConvParams& p, hipStream_t s);
int ga_conv_gemm_supported(const ConvParams& p);
int ga_conv_gemm_impl(const ConvParams& p, hipStream_t s);
int ga_conv_gemm_finish(const ConvParams& p, hipStream_t s);
int ga_conv_gemm_suggest_ksplit(const ConvParams& p);
int ga_conv_splitk_static_impl(const ConvParams& p, hipStream_t s);
int ga_conv_win_supported(const ConvParams& p);
int ga_conv_win_wanted(const ConvParams& p);
int ga_conv_coop_wanted(const ConvParams& p);
int ga_conv_win_impl(const ConvParams& p, hipStream_t s);
int ga_conv_gemm_variant_impl(const ConvParams& p);
int ga_set_gemm_lds_budget_impl(int bytes);
int ga_wgrad_gemm_multi_impl(const WgradParams* items, int n, hipStream_t s);
int ga_conv_pp_supported(const ConvParams& p);
int ga_conv_pp_wanted(const ConvParams& p);
int ga_conv_pp_impl(const ConvParams& p, hipStream_t s);
int ga_conv_stream_supported(const ConvParams& p);
int ga_conv_stream_impl(const ConvParams& p, hipStream_t s);
int ga_conv_swin_supported(const ConvParams& p);
int ga_conv_swin_wanted(const ConvParams& p);
int ga_conv_swin_impl(const ConvParams& p, hipStream_t s);
int ga_wgrad_mfma_supported(const WgradParams& p);
int ga_wgrad_mfma_impl(const WgradParams& p, hipStream_t s);
int ga_wgrad_gemm_supported(const WgradParams& p);
int ga_wgrad_gemm_impl(const WgradParams& p, hipStream_t s);
size_t ga_wgrad_gemm_workspace_elems(const WgradParams& p);
int ga_thin_conv_supported(const ConvParams& p, int dtype);
int ga_thin_conv_impl(const ConvParams& p, int dtype, hipStream_t s);
int ga_wgrad_cin1_mfma_supported(const WgradParams& p);
int ga_wgrad_cin1_mfma_impl(const WgradParams& p, hipStream_t s);
int ga_thin_wgrad_supported(const WgradParams& p, int dtype);
int ga_thin_wgrad_impl(const WgradParams& p, float* dbias, int dtype, int kind, hipStream_t s);
extern "C" int ga_colsum(const void* g, int dtype, size_t rows, int M, float* out, hipStream_t s);

extern "C" {

int ga_conv(const ga_conv_params* hp, int dtype, int impl, hipStream_t s) {
    if (!hp) return GA_ERR_ARG;
    if (dtype != GA_DTYPE_F32 && dtype != GA_DTYPE_BF16) return GA_ERR_ARG;
    const ConvParams& p = *hp;
    const bool can_mfma = dtype == GA_DTYPE_BF16 && ga_conv_mfma_supported(p);
    const bool can_gemm = dtype == GA_DTYPE_BF16 && ga_conv_gemm_supported(p);
    if (impl == GA_IMPL_MFMA_PP) return (dtype == GA_DTYPE_BF16 && ga_conv_pp_supported(p)) ? ga_conv_pp_impl(p, s) : GA_ERR_ARG;
    if (impl == GA_IMPL_MFMA_LDSWIN) return (dtype == GA_DTYPE_BF16 && ga_conv_win_supported(p)) ? ga_conv_win_impl(p, s) : GA_ERR_ARG;
    if (impl == GA_IMPL_MFMA_STREAM) return (dtype == GA_DTYPE_BF16 && ga_conv_stream_supported(p)) ? ga_conv_stream_impl(p, s) : GA_ERR_ARG;
    if (impl == GA_IMPL_MFMA_STREAMWIN) return (dtype == GA_DTYPE_BF16 && ga_conv_swin_supported(p)) ? ga_conv_swin_impl(p, s) : GA_ERR_ARG;
    // thin bf16 layers (M <= 64, K <= 192): the streaming MFMA kernels, which are also the ones that fuse ELU(x) -- multi-tap
    // layers through a per-wave LDS window ring, the rest straight from global memory into the operand registers
    if (dtype == GA_DTYPE_BF16 && (impl == GA_IMPL_AUTO || impl == GA_IMPL_MFMA) && ga_conv_stream_supported(p))
        return ga_conv_swin_wanted(p) ? ga_conv_swin_impl(p, s) : ga_conv_stream_impl(p, s);
    // single-output-channel thin layers, long launches: the streaming-window kernel's dot-product form
    if (dtype == GA_DTYPE_BF16 && impl == GA_IMPL_AUTO && p.M == 1 && p.res_mode == GA_RES_ADD && !p.y2 && ga_conv_swin_wanted(p))
        return ga_conv_swin_impl(p, s);
    if (p.y2)  // second (ELU) output: the gathered-GEMM kernel's epilogue only
        return (can_gemm && (impl == GA_IMPL_AUTO || impl == GA_IMPL_MFMA) && p.ksplit <= 1 && !p.out_f32) ? ga_conv_gemm_impl(p, s) : GA_ERR_ARG;
    if (p.res_mode == GA_RES_ADD_THEN_ACT)  // activation after the residual sum: the gathered-GEMM kernel's epilogue only
        return (can_gemm && (impl == GA_IMPL_AUTO || impl == GA_IMPL_MFMA) && p.res && p.ksplit <= 1) ? ga_conv_gemm_impl(p, s) : GA_ERR_ARG;
    if (p.res_mode != GA_RES_ADD) {  // only the gathered-GEMM kernel's epilogue forms the feature-matching term
        if (p.res_mode != GA_RES_FM_SIGN || !p.res || !p.dact_src || !p.dact_mode || !p.res_coef_dev) return GA_ERR_ARG;
        return (can_gemm && (impl == GA_IMPL_AUTO || impl == GA_IMPL_MFMA)) ? ga_conv_gemm_impl(p, s) : GA_ERR_ARG;
    }
    if (p.y_batch_elems != 0 && !(can_gemm && (impl == GA_IMPL_AUTO || impl == GA_IMPL_MFMA))) return GA_ERR_ARG;
    if (p.y_batch_elems != 0) return ga_conv_gemm_impl(p, s);
    if (p.ksplit > 1) return (can_gemm && (impl == GA_IMPL_AUTO || impl == GA_IMPL_MFMA)) ? ga_conv_gemm_impl(p, s) : GA_ERR_ARG;
    if (impl == GA_IMPL_MFMA_WINDOW) return can_mfma ? ga_conv_mfma_impl(p, s) : GA_ERR_ARG;
    if (impl == GA_IMPL_MFMA) return can_gemm ? ga_conv_gemm_impl(p, s) : can_mfma ? ga_conv_mfma_impl(p, s) : GA_ERR_ARG;
    if (impl == GA_IMPL_AUTO && ((long)p.Lout * p.B >= 64 || p.out_f32)) {
        if (can_gemm) return ga_conv_gemm_impl(p, s);
        if (can_mfma) return ga_conv_mfma_impl(p, s);
    }
    if (impl != GA_IMPL_AUTO && impl != GA_IMPL_DIRECT) return GA_ERR_ARG;
    const bool want_f32_out = p.out_f32 && dtype == GA_DTYPE_BF16;
    if (want_f32_out && !(p.in_stride >= 1 && ga_thin_conv_supported(p, dtype))) return GA_ERR_ARG;  // only MFMA / thin rows kernels
    if (p.in_stride >= 1 && ga_thin_conv_supported(p, dtype)) {
        if (!p.x || !p.w || !p.y || p.nphase < 1 || p.nphase > GA_MAX_PHASE || (long)p.B * p.nphase > 65535) return GA_ERR_ARG;
        return p.Lout > 0 ? ga_thin_conv_impl(p, dtype, s) : GA_OK;
    }
    return ga_conv_direct_impl(p, dtype, s);
}

int ga_conv_suggest_ksplit(const ga_conv_params* hp, int dtype, int impl) {
    if (!hp || dtype != GA_DTYPE_BF16 || (impl != GA_IMPL_AUTO && impl != GA_IMPL_MFMA) || hp->y2) return 1;
    return ga_conv_gemm_suggest_ksplit(*hp);
}

int ga_conv_finish(const ga_conv_params* hp, int dtype, hipStream_t s) {
    if (!hp || dtype != GA_DTYPE_BF16) return GA_ERR_ARG;
    return ga_conv_gemm_finish(*hp, s);
}

int ga_conv_splitk_static(const ga_conv_params* hp, int dtype, hipStream_t s) {
    if (!hp || dtype != GA_DTYPE_BF16) return GA_ERR_ARG;
    return ga_conv_splitk_static_impl(*hp, s);
}

int ga_conv_gemm_variant(const ga_conv_params* hp) { return hp ? ga_conv_gemm_variant_impl(*hp) : 0; }
int ga_set_gemm_lds_budget(int bytes) { return ga_set_gemm_lds_budget_impl(bytes); }

int ga_conv_kernel_id(const ga_conv_params* hp, int dtype, int impl) {
    if (!hp) return -1;
    const ConvParams& p = *hp;
    const bool can_mfma = dtype == GA_DTYPE_BF16 && ga_conv_mfma_supported(p);
    const bool can_gemm = dtype == GA_DTYPE_BF16 && ga_conv_gemm_supported(p);
    const bool big = (long)p.Lout * p.B >= 64 || p.out_f32;
    if (impl == GA_IMPL_MFMA_PP) return (dtype == GA_DTYPE_BF16 && ga_conv_pp_supported(p)) ? GA_KERNEL_GEMM_PP : -1;
    if (impl == GA_IMPL_MFMA_LDSWIN) return (dtype == GA_DTYPE_BF16 && ga_conv_win_supported(p)) ? GA_KERNEL_CONV_WIN : -1;
    if (impl == GA_IMPL_MFMA_STREAM) return (dtype == GA_DTYPE_BF16 && ga_conv_stream_supported(p)) ? GA_KERNEL_STREAM : -1;
    if (impl == GA_IMPL_MFMA_STREAMWIN) return (dtype == GA_DTYPE_BF16 && ga_conv_swin_supported(p)) ? GA_KERNEL_STREAM_WIN : -1;
    if (dtype == GA_DTYPE_BF16 && (impl == GA_IMPL_AUTO || impl == GA_IMPL_MFMA) && p.res_mode == GA_RES_ADD_THEN_ACT && ga_conv_stream_supported(p))
        return GA_KERNEL_STREAM;  // (as ga_conv: the streaming kernels are asked first)
    if (p.res_mode != GA_RES_ADD || p.y2) {  // ga_conv: gathered-GEMM kernel or nothing
        if (!(can_gemm && (impl == GA_IMPL_AUTO || impl == GA_IMPL_MFMA))) return -1;
        if (ga_conv_coop_wanted(p)) return GA_KERNEL_COOP;
        if (ga_conv_win_wanted(p)) return GA_KERNEL_CONV_WIN;
        if (ga_conv_pp_wanted(p)) return GA_KERNEL_GEMM_PP;
        return p.M <= 32 ? GA_KERNEL_GEMM_32x256 : p.M <= 64 ? GA_KERNEL_GEMM_64x256 : GA_KERNEL_GEMM_128x128;
    }
    if (dtype == GA_DTYPE_BF16 && impl == GA_IMPL_AUTO && p.M == 1 && ga_conv_swin_wanted(p)) return GA_KERNEL_STREAM_WIN;
    if (dtype == GA_DTYPE_BF16 && (impl == GA_IMPL_AUTO || impl == GA_IMPL_MFMA) && ga_conv_stream_supported(p))
        return ga_conv_swin_wanted(p) ? GA_KERNEL_STREAM_WIN : GA_KERNEL_STREAM;
    if (can_gemm && (impl == GA_IMPL_MFMA || (impl == GA_IMPL_AUTO && big)) && ga_conv_coop_wanted(p)) return GA_KERNEL_COOP;
    if (can_gemm && (impl == GA_IMPL_MFMA || (impl == GA_IMPL_AUTO && big)) && ga_conv_win_wanted(p)) return GA_KERNEL_CONV_WIN;
    if (can_gemm && (impl == GA_IMPL_MFMA || (impl == GA_IMPL_AUTO && big)) && ga_conv_pp_wanted(p)) return GA_KERNEL_GEMM_PP;
    if (can_gemm && (impl == GA_IMPL_MFMA || (impl == GA_IMPL_AUTO && big))) return p.M <= 32 ? GA_KERNEL_GEMM_32x256 : p.M <= 64 ? GA_KERNEL_GEMM_64x256 : GA_KERNEL_GEMM_128x128;
    const bool mfma = ((impl == GA_IMPL_MFMA || impl == GA_IMPL_MFMA_WINDOW) && can_mfma) || (impl == GA_IMPL_AUTO && can_mfma && big);
    if (mfma) return p.M <= 64 ? GA_KERNEL_MFMA_64x128 : GA_KERNEL_MFMA_128x128;
    if (p.in_stride >= 1 && ga_thin_conv_supported(p, dtype)) return p.M <= 4 ? GA_KERNEL_THIN_FEWOUT_ROWS : GA_KERNEL_THIN_CIN1_VEC;
    if (p.M <= 4) return GA_KERNEL_DIRECT_FEWOUT;
    if (p.Cin == 1 && p.nphase == 1) return GA_KERNEL_DIRECT_CIN1;
    return GA_KERNEL_DIRECT_TILE;
}

size_t ga_wgrad_workspace_elems(const ga_wgrad_params* hp, int dtype, int impl) {
    if (!hp || dtype != GA_DTYPE_BF16 || (impl != GA_IMPL_AUTO && impl != GA_IMPL_MFMA)) return 0;
    if (ga_thin_wgrad_supported(*hp, dtype) && impl != GA_IMPL_MFMA) return 0;
    return ga_wgrad_gemm_workspace_elems(*hp);
}

int ga_wgrad(const ga_wgrad_params* hp, int dtype, int impl, hipStream_t s) {
    if (!hp) return GA_ERR_ARG;
    if (dtype != GA_DTYPE_F32 && dtype != GA_DTYPE_BF16) return GA_ERR_ARG;
    const WgradParams& p = *hp;
    if (!p.g || !p.x || !p.dw || p.B <= 0 || p.Lin <= 0 || p.Cin <= 0 || p.M <= 0 || p.Lout < 0 || p.J <= 0) return GA_ERR_ARG;
    if (p.g_stride < 1 || p.LgTot < 1 || p.B > 65535) return GA_ERR_ARG;
    if (p.Lout == 0) return GA_OK;
    // single input channel, caller-padded x: the Toeplitz MFMA kernel (fuses the bias gradient)
    if (dtype == GA_DTYPE_BF16 && (impl == GA_IMPL_AUTO || impl == GA_IMPL_MFMA) && ga_wgrad_cin1_mfma_supported(p))
        return ga_wgrad_cin1_mfma_impl(p, s);
    const int thin = ga_thin_wgrad_supported(p, dtype);
    if (thin && impl != GA_IMPL_MFMA && impl != GA_IMPL_MFMA_WINDOW) return ga_thin_wgrad_impl(p, p.dbias, dtype, thin, s);  // fuses the bias gradient
    const bool can_mfma = dtype == GA_DTYPE_BF16 && ga_wgrad_mfma_supported(p);
    const bool can_gemm = dtype == GA_DTYPE_BF16 && ga_wgrad_gemm_supported(p);
    int rc;
    if (impl == GA_IMPL_MFMA_WINDOW) rc = can_mfma ? ga_wgrad_mfma_impl(p, s) : GA_ERR_ARG;
    else if ((impl == GA_IMPL_MFMA || impl == GA_IMPL_AUTO) && can_gemm) {
        rc = ga_wgrad_gemm_impl(p, s);
        // without slabs the kernel has produced the bias gradient itself (a constant-one column of the gathered operand)
        if (rc == GA_OK && p.dbias && !(p.ws && p.M > 64)) return (p.g_stride != 1 || p.g_phase != 0 || p.LgTot != p.Lout) ? GA_ERR_ARG : GA_OK;
    }
    else if (impl == GA_IMPL_MFMA) rc = can_mfma ? ga_wgrad_mfma_impl(p, s) : GA_ERR_ARG;
    else if (impl == GA_IMPL_AUTO && can_mfma) rc = ga_wgrad_mfma_impl(p, s);
    else if (impl != GA_IMPL_AUTO && impl != GA_IMPL_DIRECT) rc = GA_ERR_ARG;
    else rc = ga_wgrad_direct_impl(p, dtype, s);
    if (rc == GA_OK && p.dbias) {
        if (p.g_stride != 1 || p.g_phase != 0 || p.LgTot != p.Lout) return GA_ERR_ARG;
        rc = ga_colsum(p.g, dtype, (size_t)p.B * p.Lout, p.M, p.dbias, s);
    }
    return rc;
}

int ga_wgrad_multi(const ga_wgrad_params* items, int n, int dtype, hipStream_t s) {
    if (!items || n < 0) return GA_ERR_ARG;
    if (n == 0) return GA_OK;
    if (dtype != GA_DTYPE_BF16) return GA_ERR_UNSUPPORTED;
    for (int i = 0; i < n; ++i) {
        const WgradParams& p = items[i];
        if (!p.g || !p.x || !p.dw || p.B <= 0 || p.Lin <= 0 || p.Cin <= 0 || p.M <= 0 || p.Lout < 0 || p.J <= 0) return GA_ERR_ARG;
        if (p.g_stride < 1 || p.LgTot < 1 || p.B > 65535) return GA_ERR_ARG;
    }
    return ga_wgrad_gemm_multi_impl(items, n, s);
}

const char* ga_build_target(void) { return "gfx950"; }

}  // extern "C"
