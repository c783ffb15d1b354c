// Shared device/host helpers for the gan_amd HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gan_amd.h"

typedef __bf16 bf16_t;

#define GA_CHECK_LAUNCH()                          \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return GA_ERR_LAUNCH; \
    } while (0)

template <typename T> __device__ __forceinline__ float ld(const T* p) { return (float)(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v) { *p = (T)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); every thread gets the result.
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 16 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += smem[i];
    return r;
}

// activations -------------------------------------------------------------------------------

__device__ __forceinline__ float act_apply(int a, float v) {
    switch (a) {
        case GA_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
        case GA_ACT_ELU: return v > 0.f ? v : expm1f(v);
        case GA_ACT_SILU: return v / (1.f + __expf(-v));
        case GA_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// ELU for a value that is about to be rounded to bf16: v itself down to -2^-12 (expm1(v) = v there to inside half a bf16 step),
// exp(v) - 1 below (the cancellation it has near zero is gone by then); ~6 instructions where expm1f takes ~50
__device__ __forceinline__ float elu_bf16_grade(float v) {
    return v > -2.44140625e-4f ? v : __builtin_amdgcn_exp2f(v * 1.4426950408889634f) - 1.f;
}

// derivative factor modes for conv epilogues (dact_mode)

__device__ __forceinline__ float dact_apply(int m, float s) {
    switch (m) {
        case GA_DACT_LRELU_FROM_OUT: return s > 0.f ? 1.f : 0.2f;
        case GA_DACT_ELU_FROM_IN: return s > 0.f ? 1.f : __expf(s);
        case GA_DACT_ELU_FROM_OUT: return s > 0.f ? 1.f : s + 1.f;
        default: return 1.f;
    }
}

// 16 zero bytes: the source of every padded / out-of-range piece of a direct-to-LDS load (one copy per translation unit)
static __device__ __attribute__((aligned(16))) unsigned int ga_zero_page[4] = {0u, 0u, 0u, 0u};

// The generalised NLC convolution / weight-gradient problem descriptors are part of the C ABI.
typedef ga_conv_params ConvParams;
typedef ga_wgrad_params WgradParams;
