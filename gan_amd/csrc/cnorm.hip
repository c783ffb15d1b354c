// Channels-last InstanceNorm2d / BatchNorm2d (+ LeakyReLU 0.2) of the mel track (lab 3/src/lab3_models.py:80-83 norm2d, :159-165).
//
// x, y: [B][HW][C] (the NHWC image with H and W merged), 16-byte pieces along C.  `groups` = B: statistics per item and channel
// (InstanceNorm2d); `groups` = 1: per channel over all B*HW rows (BatchNorm2d).  HBM-bound: the forward pass reads x twice and writes
// y once, the backward pass reads (gy, x, y) twice and writes gx once; as torch modules on channels-last memory the same layers cost a
// layout copy each way around MIOpen's NCHW kernels plus a separate LeakyReLU pass.
//
// Every kernel uses ONE thread mapping: grid (row blocks, B), 256 threads = (C / EV channel pieces) x (rows per pass), so a thread's
// channels -- and with them gamma, beta, mean, rstd -- are fixed for the whole launch and live in registers.
//
// Sums are taken about a PIVOT (the group's first row), sum(x - p) and sum((x - p)^2), so that var = E[(x-p)^2] - E[x-p]^2 does not
// cancel when |mean| >> std; a thread's fp32 partial sums (a fixed set of rows: deterministic) meet in LDS and leave as one global atomicAdd
// per channel and block, both in DOUBLE, so that the order of the atomics shows at 1e-16 and not at 1e-7 (with float atomics a three-layer
// decoder's output moved by up to 1e-6 from run to run, enough to cross a 2e-6 parity bound now and then); they go into PER-ITEM
// sums [B][C][2] (a batch norm's 2048 blocks on one address per channel cost 15-20 us of serialised atomics); the finalize kernel / the
// caller adds the items of a group.
#include "common.h"

template <typename T> struct alignas(16) CnPack { T v[16 / sizeof(T)]; };

#define CN_MAP()                                                                     \
    constexpr int EV = 16 / (int)sizeof(T);                                          \
    const int cv = C / EV, rpp = 256 / cv;                                           \
    const int piece = threadIdx.x % cv, rl = threadIdx.x / cv;                       \
    const bool active = rl < rpp;                                                    \
    const int b = blockIdx.y, g = groups == 1 ? 0 : b, c0 = piece * EV;              \
    const size_t item = (size_t)b * HW * C;                                          \
    (void)g; (void)item; (void)c0; (void)active

template <typename T>
__global__ __launch_bounds__(256) void cnorm_stats_kernel(const T* __restrict__ x, int HW, int C, int groups, double* __restrict__ sums) {
    CN_MAP();
    extern __shared__ double acc[];  // [C][2]
    for (int i = threadIdx.x; i < 2 * C; i += 256) acc[i] = 0.0;
    __syncthreads();
    if (active) {
        const CnPack<T> pv = *(const CnPack<T>*)(x + (groups == 1 ? (size_t)0 : item) + c0);
        float p[EV], s1[EV], s2[EV];
#pragma unroll
        for (int e = 0; e < EV; ++e) { p[e] = (float)pv.v[e]; s1[e] = 0.f; s2[e] = 0.f; }
        const int step = gridDim.x * rpp;
        int r = blockIdx.x * rpp + rl;
        auto add = [&](const CnPack<T>& v) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < EV; ++e) { const float d = (float)v.v[e] - p[e]; s1[e] += d; s2[e] = fmaf(d, d, s2[e]); }
        };
        for (; (long)r + 3L * step < HW; r += 4 * step) {  // four loads in flight per thread
            const T* q = x + item + (size_t)r * C + c0;
            const CnPack<T> v0 = *(const CnPack<T>*)q, v1 = *(const CnPack<T>*)(q + (size_t)step * C),
                            v2 = *(const CnPack<T>*)(q + (size_t)2 * step * C), v3 = *(const CnPack<T>*)(q + (size_t)3 * step * C);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; r < HW; r += step) add(*(const CnPack<T>*)(x + item + (size_t)r * C + c0));
#pragma unroll
        for (int e = 0; e < EV; ++e) { atomicAdd(&acc[2 * (c0 + e)], (double)s1[e]); atomicAdd(&acc[2 * (c0 + e) + 1], (double)s2[e]); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(&sums[(size_t)b * 2 * C + i], acc[i]);
}

// mean / rstd per (group, channel); BatchNorm's running statistics (momentum update, unbiased variance) when given
template <typename T>
__global__ void cnorm_finalize_kernel(const T* __restrict__ x, int B, int HW, int C, int groups, const double* __restrict__ sums, float eps,
                                      double* __restrict__ mean, double* __restrict__ rstd, float* __restrict__ run_mean,
                                      float* __restrict__ run_var, float momentum) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * C) return;
    const int g = i / C, c = i - g * C;
    const double n = groups == 1 ? (double)B * HW : (double)HW;
    const float p = (float)x[(groups == 1 ? (size_t)0 : (size_t)g * HW * C) + c];
    double t1 = 0.0, t2 = 0.0;  // the per-item partial sums of the group
    for (int bb = (groups == 1 ? 0 : g); bb < (groups == 1 ? B : g + 1); ++bb) {
        t1 += sums[2 * ((size_t)bb * C + c)];
        t2 += sums[2 * ((size_t)bb * C + c) + 1];
    }
    const double m1 = t1 / n;  // (double from here on: a handful of operations per channel)
    double var = t2 / n - m1 * m1;
    var = var > 0.0 ? var : 0.0;
    const double mud = (double)p + m1;
    const float mu = (float)mud;
    mean[i] = mud;
    rstd[i] = 1.0 / sqrt(var + (double)eps);
    if (run_mean && groups == 1) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(var * (n / (n > 1.0 ? n - 1.0 : 1.0)));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cnorm_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int HW, int C, int groups,
                                                          const double* __restrict__ mean, const double* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, int act) {
    CN_MAP();
    if (!active) return;
    // y = ((x - mean) * rstd) * gamma + beta with the subtraction FIRST (x * sc + (beta - mean * sc) cancels two numbers of size
    // |mean| / std and carries their rounding into y); the mean as hi + lo floats
    float mu[EV], mlo[EV], sc[EV], sh[EV];
#pragma unroll
    for (int e = 0; e < EV; ++e) {
        const double rs = rstd[(size_t)g * C + c0 + e], md = mean[(size_t)g * C + c0 + e];
        mu[e] = (float)md;
        mlo[e] = (float)(md - (double)mu[e]);
        sc[e] = (float)(rs * (gamma ? (double)gamma[c0 + e] : 1.0));
        sh[e] = beta ? beta[c0 + e] : 0.f;
    }
    for (int r = blockIdx.x * rpp + rl; r < HW; r += gridDim.x * rpp) {
        const size_t o = item + (size_t)r * C + c0;
        const CnPack<T> v = *(const CnPack<T>*)(x + o);
        CnPack<T> w;
#pragma unroll
        for (int e = 0; e < EV; ++e) {
            float t = fmaf(((float)v.v[e] - mu[e]) - mlo[e], sc[e], sh[e]);
            if (act == GA_ACT_LRELU) t = t > 0.f ? t : 0.2f * t;
            w.v[e] = (T)t;
        }
        *(CnPack<T>*)(y + o) = w;
    }
}

// gsums[b][c] = (sum g, sum g * xhat) over item b, g = gy * act'(y)
template <typename T>
__global__ __launch_bounds__(256) void cnorm_bwd_stats_kernel(const T* __restrict__ gy, const T* __restrict__ x, const T* __restrict__ y, int HW,
                                                              int C, int groups, const double* __restrict__ mean,
                                                              const double* __restrict__ rstd, int act, double* __restrict__ gsums) {
    CN_MAP();
    extern __shared__ double acc[];
    for (int i = threadIdx.x; i < 2 * C; i += 256) acc[i] = 0.0;
    __syncthreads();
    if (active) {
        float mu[EV], mlo[EV], rs[EV], s1[EV], s2[EV];  // mean as hi + lo floats: xhat = ((x - hi) - lo) * rstd keeps the mean's low bits
#pragma unroll
        for (int e = 0; e < EV; ++e) {
            const double md = mean[(size_t)g * C + c0 + e];
            mu[e] = (float)md; mlo[e] = (float)(md - (double)mu[e]);
            rs[e] = (float)rstd[(size_t)g * C + c0 + e]; s1[e] = 0.f; s2[e] = 0.f;
        }
        const int step = gridDim.x * rpp;
        int r = blockIdx.x * rpp + rl;
        auto add = [&](const CnPack<T>& vg, const CnPack<T>& vx, const CnPack<T>& vy) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < EV; ++e) {
                float gg = (float)vg.v[e];
                if (act == GA_ACT_LRELU) gg *= (float)vy.v[e] > 0.f ? 1.f : 0.2f;
                s1[e] += gg;
                s2[e] = fmaf(gg, (((float)vx.v[e] - mu[e]) - mlo[e]) * rs[e], s2[e]);
            }
        };
        const T* yy = act == GA_ACT_LRELU ? y : x;  // without an activation the third stream is not looked at
        for (; (long)r + step < HW; r += 2 * step) {  // six loads in flight per thread
            const size_t o0 = item + (size_t)r * C + c0, o1 = o0 + (size_t)step * C;
            const CnPack<T> g0 = *(const CnPack<T>*)(gy + o0), x0 = *(const CnPack<T>*)(x + o0), y0 = *(const CnPack<T>*)(yy + o0);
            const CnPack<T> g1 = *(const CnPack<T>*)(gy + o1), x1 = *(const CnPack<T>*)(x + o1), y1 = *(const CnPack<T>*)(yy + o1);
            add(g0, x0, y0); add(g1, x1, y1);
        }
        for (; r < HW; r += step) {
            const size_t o = item + (size_t)r * C + c0;
            add(*(const CnPack<T>*)(gy + o), *(const CnPack<T>*)(x + o), *(const CnPack<T>*)(yy + o));
        }
#pragma unroll
        for (int e = 0; e < EV; ++e) { atomicAdd(&acc[2 * (c0 + e)], (double)s1[e]); atomicAdd(&acc[2 * (c0 + e) + 1], (double)s2[e]); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(&gsums[(size_t)b * 2 * C + i], acc[i]);
}

// gx = gamma * rstd * (g - (sum g + xhat * sum g xhat) / N); frozen statistics (eval-mode BatchNorm): gx = gamma * rstd * g
template <typename T>
__global__ __launch_bounds__(256) void cnorm_bwd_apply_kernel(const T* __restrict__ gy, const T* __restrict__ x, const T* __restrict__ y,
                                                              T* __restrict__ gx, int B, int HW, int C, int groups,
                                                              const double* __restrict__ mean, const double* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const double* __restrict__ gsums, int act,
                                                              int frozen) {
    CN_MAP();
    if (!active) return;
    const double inv_n = frozen ? 0.0 : 1.0 / (groups == 1 ? (double)B * (double)HW : (double)HW);
    // The input gradient has to sum to zero over the normalised set (that is what makes the gradient of a bias in FRONT of the layer
    // zero); rounding the mean or sum(g)/N to fp32 leaves a residual of the same sign on every element, N times over -- 1e-7 relative,
    // but AdamW divides by |g| and steps such a bias by a full lr.  So: the mean and both sums stay double, per element in double.
    double mu[EV], rs[EV], k[EV], a1[EV], a2[EV];
#pragma unroll
    for (int e = 0; e < EV; ++e) {
        mu[e] = mean[(size_t)g * C + c0 + e];
        rs[e] = rstd[(size_t)g * C + c0 + e];
        k[e] = rs[e] * (gamma ? (double)gamma[c0 + e] : 1.0);
        a1[e] = gsums[2 * ((size_t)g * C + c0 + e)] * inv_n;
        a2[e] = gsums[2 * ((size_t)g * C + c0 + e) + 1] * inv_n;
    }
    for (int r = blockIdx.x * rpp + rl; r < HW; r += gridDim.x * rpp) {
        const size_t o = item + (size_t)r * C + c0;
        const CnPack<T> vg = *(const CnPack<T>*)(gy + o), vx = *(const CnPack<T>*)(x + o);
        CnPack<T> vy, w;
        if (act == GA_ACT_LRELU) vy = *(const CnPack<T>*)(y + o);
#pragma unroll
        for (int e = 0; e < EV; ++e) {
            float gg = (float)vg.v[e];
            if (act == GA_ACT_LRELU) gg *= (float)vy.v[e] > 0.f ? 1.f : 0.2f;
            const double xh = ((double)(float)vx.v[e] - mu[e]) * rs[e];
            w.v[e] = (T)(float)(k[e] * ((double)gg - a1[e] - xh * a2[e]));
        }
        *(CnPack<T>*)(gx + o) = w;
    }
}

// ================================================================================================
#define CN_DISPATCH(dtype, CALL_F32, CALL_BF16)       \
    do {                                              \
        if ((dtype) == GA_DTYPE_BF16) { CALL_BF16; }  \
        else { CALL_F32; }                            \
    } while (0)

static int cn_check(int dtype, int B, long HW, int C, int groups) {
    if (dtype != GA_DTYPE_F32 && dtype != GA_DTYPE_BF16) return GA_ERR_ARG;
    if (B <= 0 || HW <= 0 || C <= 0 || HW > 0x7fffffffL || (groups != 1 && groups != B)) return GA_ERR_ARG;
    const int ev = dtype == GA_DTYPE_BF16 ? 8 : 4;
    if (C % ev || C / ev > 256 || B > 65535) return GA_ERR_UNSUPPORTED;
    return GA_OK;
}
// the elementwise passes: >= 4 rows per thread, up to 2048 workgroups; the two reductions: >= 16 rows per thread and up to 1024
// workgroups, because every workgroup ends in 2C global atomics (a [64][96][512] tensor cut into 384 workgroups spent one atomic per
// 16-byte piece it had read)
static dim3 cn_grid(int dtype, int B, long HW, int C, bool reduction = false) {
    const int ev = dtype == GA_DTYPE_BF16 ? 8 : 4, rpp = 256 / (C / ev);
    const long per = (long)rpp * (reduction ? 16 : 4), total = reduction ? 1024 : 2048;
    long nb = (HW + per - 1) / per;
    const long cap = total / B > 1 ? total / B : 1;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return dim3((unsigned)nb, (unsigned)B, 1);
}

extern "C" {

int ga_cnorm_supported(int dtype, int B, long HW, int C, int groups) { return cn_check(dtype, B, HW, C, groups) == GA_OK; }

int ga_cnorm_stats(const void* x, int dtype, int B, long HW, int C, int groups, double* sums, hipStream_t s) {
    const int rc = cn_check(dtype, B, HW, C, groups);
    if (rc) return rc;
    if (!x || !sums) return GA_ERR_ARG;
    if (((uintptr_t)x) & 15) return GA_ERR_ALIGN;
    const dim3 grid = cn_grid(dtype, B, HW, C, true);
    const size_t lds = (size_t)2 * C * sizeof(double);
    CN_DISPATCH(dtype, (cnorm_stats_kernel<float><<<grid, 256, lds, s>>>((const float*)x, (int)HW, C, groups, sums)),
                (cnorm_stats_kernel<bf16_t><<<grid, 256, lds, s>>>((const bf16_t*)x, (int)HW, C, groups, sums)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_cnorm_finalize(const void* x, int dtype, int B, long HW, int C, int groups, const double* sums, float eps, double* mean, double* rstd,
                      float* running_mean, float* running_var, float momentum, hipStream_t s) {
    const int rc = cn_check(dtype, B, HW, C, groups);
    if (rc) return rc;
    if (!x || !sums || !mean || !rstd || (running_mean && (!running_var || groups != 1))) return GA_ERR_ARG;
    const int n = groups * C;
    CN_DISPATCH(dtype, (cnorm_finalize_kernel<float><<<(n + 255) / 256, 256, 0, s>>>((const float*)x, B, (int)HW, C, groups, sums, eps, mean, rstd,
                                                                                     running_mean, running_var, momentum)),
                (cnorm_finalize_kernel<bf16_t><<<(n + 255) / 256, 256, 0, s>>>((const bf16_t*)x, B, (int)HW, C, groups, sums, eps, mean, rstd,
                                                                                running_mean, running_var, momentum)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_cnorm_apply(const void* x, void* y, int dtype, int B, long HW, int C, int groups, const double* mean, const double* rstd,
                   const float* gamma, const float* beta, int act, hipStream_t s) {
    const int rc = cn_check(dtype, B, HW, C, groups);
    if (rc) return rc;
    if (!x || !y || !mean || !rstd || (act != GA_ACT_NONE && act != GA_ACT_LRELU)) return GA_ERR_ARG;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return GA_ERR_ALIGN;
    const dim3 grid = cn_grid(dtype, B, HW, C);
    CN_DISPATCH(dtype, (cnorm_apply_kernel<float><<<grid, 256, 0, s>>>((const float*)x, (float*)y, (int)HW, C, groups, mean, rstd, gamma, beta, act)),
                (cnorm_apply_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, (int)HW, C, groups, mean, rstd, gamma, beta, act)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_cnorm_bwd_stats(const void* gy, const void* x, const void* y, int dtype, int B, long HW, int C, int groups, const double* mean,
                       const double* rstd, int act, double* gsums, hipStream_t s) {
    const int rc = cn_check(dtype, B, HW, C, groups);
    if (rc) return rc;
    if (!gy || !x || !mean || !rstd || !gsums || (act != GA_ACT_NONE && act != GA_ACT_LRELU) || (act == GA_ACT_LRELU && !y)) return GA_ERR_ARG;
    if (((uintptr_t)gy | (uintptr_t)x | (uintptr_t)y) & 15) return GA_ERR_ALIGN;
    const dim3 grid = cn_grid(dtype, B, HW, C, true);
    const size_t lds = (size_t)2 * C * sizeof(double);
    CN_DISPATCH(dtype,
                (cnorm_bwd_stats_kernel<float><<<grid, 256, lds, s>>>((const float*)gy, (const float*)x, (const float*)y, (int)HW, C, groups, mean,
                                                                      rstd, act, gsums)),
                (cnorm_bwd_stats_kernel<bf16_t><<<grid, 256, lds, s>>>((const bf16_t*)gy, (const bf16_t*)x, (const bf16_t*)y, (int)HW, C, groups,
                                                                       mean, rstd, act, gsums)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_cnorm_bwd_apply(const void* gy, const void* x, const void* y, void* gx, int dtype, int B, long HW, int C, int groups, const double* mean,
                       const double* rstd, const float* gamma, const double* gsums, int act, int frozen, hipStream_t s) {
    const int rc = cn_check(dtype, B, HW, C, groups);
    if (rc) return rc;
    if (!gy || !x || !gx || !mean || !rstd || !gsums || (act != GA_ACT_NONE && act != GA_ACT_LRELU) || (act == GA_ACT_LRELU && !y))
        return GA_ERR_ARG;
    if (((uintptr_t)gy | (uintptr_t)x | (uintptr_t)y | (uintptr_t)gx) & 15) return GA_ERR_ALIGN;
    const dim3 grid = cn_grid(dtype, B, HW, C);
    CN_DISPATCH(dtype,
                (cnorm_bwd_apply_kernel<float><<<grid, 256, 0, s>>>((const float*)gy, (const float*)x, (const float*)y, (float*)gx, B, (int)HW, C,
                                                                    groups, mean, rstd, gamma, gsums, act, frozen)),
                (cnorm_bwd_apply_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)gy, (const bf16_t*)x, (const bf16_t*)y, (bf16_t*)gx, B, (int)HW,
                                                                     C, groups, mean, rstd, gamma, gsums, act, frozen)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

}  // extern "C"
