// C ABI dispatch for the convolution family.  See include/gan_amd.h.
#include "common.h"

int ga_conv_direct_impl(const ConvParams& p, int dtype, hipStream_t s);
int ga_wgrad_direct_impl(const WgradParams& p, int dtype, hipStream_t s);
int ga_conv_mfma_supported(const ConvParams& p);
int ga_conv_mfma_impl(const ConvParams& p, hipStream_t s);
int ga_wgrad_mfma_supported(const WgradParams& p);
int ga_wgrad_mfma_impl(const WgradParams& p, hipStream_t s);

extern "C" {

int ga_conv(const ga_conv_params* hp, int dtype, int impl, hipStream_t s) {
    if (!hp) return GA_ERR_ARG;
    if (dtype != GA_DTYPE_F32 && dtype != GA_DTYPE_BF16) return GA_ERR_ARG;
    const ConvParams& p = *hp;
    const bool can_mfma = dtype == GA_DTYPE_BF16 && ga_conv_mfma_supported(p);
    if (impl == GA_IMPL_MFMA) return can_mfma ? ga_conv_mfma_impl(p, s) : GA_ERR_ARG;
    if (impl == GA_IMPL_AUTO && can_mfma && (long)p.Lout * p.B >= 64) return ga_conv_mfma_impl(p, s);
    if (impl != GA_IMPL_AUTO && impl != GA_IMPL_DIRECT) return GA_ERR_ARG;
    return ga_conv_direct_impl(p, dtype, s);
}

int ga_conv_kernel_id(const ga_conv_params* hp, int dtype, int impl) {
    if (!hp) return -1;
    const ConvParams& p = *hp;
    const bool can_mfma = dtype == GA_DTYPE_BF16 && ga_conv_mfma_supported(p);
    const bool mfma = (impl == GA_IMPL_MFMA && can_mfma) || (impl == GA_IMPL_AUTO && can_mfma && (long)p.Lout * p.B >= 64);
    if (mfma) return p.M <= 64 ? GA_KERNEL_MFMA_64x128 : GA_KERNEL_MFMA_128x128;
    if (p.M <= 4) return GA_KERNEL_DIRECT_FEWOUT;
    if (p.Cin == 1 && p.nphase == 1) return GA_KERNEL_DIRECT_CIN1;
    return GA_KERNEL_DIRECT_TILE;
}

int ga_wgrad(const ga_wgrad_params* hp, int dtype, int impl, hipStream_t s) {
    if (!hp) return GA_ERR_ARG;
    if (dtype != GA_DTYPE_F32 && dtype != GA_DTYPE_BF16) return GA_ERR_ARG;
    const WgradParams& p = *hp;
    const bool can_mfma = dtype == GA_DTYPE_BF16 && ga_wgrad_mfma_supported(p);
    if (impl == GA_IMPL_MFMA) return can_mfma ? ga_wgrad_mfma_impl(p, s) : GA_ERR_ARG;
    if (impl == GA_IMPL_AUTO && can_mfma) return ga_wgrad_mfma_impl(p, s);
    if (impl != GA_IMPL_AUTO && impl != GA_IMPL_DIRECT) return GA_ERR_ARG;
    return ga_wgrad_direct_impl(p, dtype, s);
}

const char* ga_build_target(void) { return "gfx950"; }

}  // extern "C"
