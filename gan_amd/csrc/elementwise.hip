// HBM-bound kernels of the lab3 codec GAN step: weight-norm + weight packing, GroupNorm+FiLM+SiLU,
// tanh residual, avg-pool, activation backward, column sums, loss reductions, latent losses,
// gradient square-norm and the fused clipped AdamW.  All activations are channels-last [B][L][C].
#include "common.h"

// ================================================================================================
// weight-norm scale: scale[r] = g[r] / ||v[r,:]||, norm[r] = ||v[r,:]||   (torch _weight_norm, dim=0)
__global__ __launch_bounds__(256) void wn_scale_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                        int n, float* __restrict__ scale, float* __restrict__ norm) {
    __shared__ float red[16];
    const int r = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float a = v[(size_t)r * n + i];
        s = fmaf(a, a, s);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float nr = sqrtf(s);
        norm[r] = nr;
        scale[r] = g[r] / nr;
    }
}

// packed position of source weight element (s0, s1, k); see ga_pack_weights in include/gan_amd.h

__device__ __forceinline__ size_t pack_index(int mode, int s0, int s1, int k, int D0, int D1, int K, int S, int P) {
    switch (mode) {
        case GA_PACK_CONV_FWD:  // src [Co][Ci][K] -> [Co][K][Ci]
            return ((size_t)s0 * K + k) * D1 + s1;
        case GA_PACK_CONV_DGRAD_S1:  // -> [Ci][K][Co], taps flipped
            return ((size_t)s1 * K + (K - 1 - k)) * D0 + s0;
        case GA_PACK_CONV_DGRAD_STRIDED: {  // -> [S][Ci][Jmax][Co]
            const int Jmax = (K + S - 1) / S;
            const int rho = k % S;
            int r = (rho - P) % S;
            if (r < 0) r += S;
            const int jp = Jmax - 1 - k / S;
            return (((size_t)r * D1 + s1) * Jmax + jp) * D0 + s0;
        }
        case GA_PACK_CONVT_FWD: {  // src [Ci][Co][K=2S] -> [S][Co][2][Ci]
            const int r = k % S, j = 1 - k / S;
            return (((size_t)r * D1 + s1) * 2 + j) * D0 + s0;
        }
        default:  // GA_PACK_CONVT_DGRAD: src [Ci][Co][K] -> [Ci][K][Co]
            return ((size_t)s0 * K + k) * D1 + s1;
    }
}

template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ src, const float* __restrict__ scale, T* __restrict__ dst,
                                    int mode, int D0, int D1, int K, int S, int P) {
    const size_t n = (size_t)D0 * D1 * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int s1 = (int)((i / K) % D1);
        const int s0 = (int)(i / ((size_t)K * D1));
        float v = src[i];
        if (scale) v *= scale[s0];
        dst[pack_index(mode, s0, s1, k, D0, D1, K, S, P)] = (T)v;
    }
}

// inverse gather: dw[s0][s1][k] (+)= packed[pack_index]
__global__ void unpack_grad_kernel(const float* __restrict__ packed, float* __restrict__ dw, int mode, int D0, int D1,
                                   int K, int S, int P, int accumulate) {
    const size_t n = (size_t)D0 * D1 * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int s1 = (int)((i / K) % D1);
        const int s0 = (int)(i / ((size_t)K * D1));
        const float v = packed[pack_index(mode, s0, s1, k, D0, D1, K, S, P)];
        dw[i] = accumulate ? dw[i] + v : v;
    }
}

// LDS-tiled forms of the two kernels above: the packed layouts transpose the tap axis against a channel axis, so a
// one-element-per-thread copy is coalesced on one side only (2- or 4-byte accesses at a stride on the other).  A tile is
// read in the source's order, parked in LDS, and written in the destination's order.
//   FAST0 (destination runs along s0: the data-gradient packs, transposed-conv forward): 64 s0 x 64 consecutive (s1,k)
//   else  (destination runs along s1: forward pack, transposed-conv data gradient):       one s0 x 64 s1 x all K (K <= 65)
// UNPACK = true moves packed fp32 gradients back to the parameter layout (the same tiles, roles of the sides swapped).
template <typename T, bool FAST0, bool UNPACK>
__global__ __launch_bounds__(256) void pack_tiled_kernel(const float* __restrict__ src, const float* __restrict__ scale,
                                                          T* __restrict__ dst, float* __restrict__ dw,
                                                          const float* __restrict__ packed, int mode, int D0, int D1, int K,
                                                          int S, int P, int accumulate) {
    __shared__ float tile[64 * 65];
    const int tid = threadIdx.x;
    const size_t DK = (size_t)D1 * K;
    if (FAST0) {
        const int s0b = blockIdx.x * 64;
        const size_t eb = (size_t)blockIdx.y * 64;
        if (!UNPACK) {
            for (int idx = tid; idx < 4096; idx += 256) {
                const int sl = idx >> 6, el = idx & 63;
                const int s0 = s0b + sl;
                const size_t e = eb + el;
                if (s0 < D0 && e < DK) tile[sl * 65 + el] = src[(size_t)s0 * DK + e] * (scale ? scale[s0] : 1.f);
            }
            __syncthreads();
            for (int idx = tid; idx < 4096; idx += 256) {
                const int el = idx >> 6, sl = idx & 63;
                const int s0 = s0b + sl;
                const size_t e = eb + el;
                if (s0 < D0 && e < DK) dst[pack_index(mode, s0, (int)(e / K), (int)(e % K), D0, D1, K, S, P)] = (T)tile[sl * 65 + el];
            }
        } else {
            for (int idx = tid; idx < 4096; idx += 256) {
                const int el = idx >> 6, sl = idx & 63;
                const int s0 = s0b + sl;
                const size_t e = eb + el;
                if (s0 < D0 && e < DK) tile[sl * 65 + el] = packed[pack_index(mode, s0, (int)(e / K), (int)(e % K), D0, D1, K, S, P)];
            }
            __syncthreads();
            for (int idx = tid; idx < 4096; idx += 256) {
                const int sl = idx >> 6, el = idx & 63;
                const int s0 = s0b + sl;
                const size_t e = eb + el;
                if (s0 < D0 && e < DK) {
                    const size_t i = (size_t)s0 * DK + e;
                    dw[i] = accumulate ? dw[i] + tile[sl * 65 + el] : tile[sl * 65 + el];
                }
            }
        }
    } else {
        const int s0 = blockIdx.x;
        const int s1b = blockIdx.y * 64;
        const int ns1 = min(64, D1 - s1b);
        const int ne = ns1 * K;
        const size_t base = ((size_t)s0 * D1 + s1b) * K;
        const float sc = (!UNPACK && scale) ? scale[s0] : 1.f;
        if (!UNPACK) {
            for (int idx = tid; idx < ne; idx += 256) tile[idx] = src[base + idx] * sc;
            __syncthreads();
            for (int idx = tid; idx < 64 * K; idx += 256) {
                const int k = idx >> 6, s1l = idx & 63;
                if (s1l < ns1) dst[pack_index(mode, s0, s1b + s1l, k, D0, D1, K, S, P)] = (T)tile[s1l * K + k];
            }
        } else {
            for (int idx = tid; idx < 64 * K; idx += 256) {
                const int k = idx >> 6, s1l = idx & 63;
                if (s1l < ns1) tile[s1l * K + k] = packed[pack_index(mode, s0, s1b + s1l, k, D0, D1, K, S, P)];
            }
            __syncthreads();
            for (int idx = tid; idx < ne; idx += 256) dw[base + idx] = accumulate ? dw[base + idx] + tile[idx] : tile[idx];
        }
    }
}

static inline bool pack_fast0(int mode) {
    return mode == GA_PACK_CONV_DGRAD_S1 || mode == GA_PACK_CONV_DGRAD_STRIDED || mode == GA_PACK_CONVT_FWD;
}

// launches the tiled pack / unpack; false when the shape does not fit the tiles (callers fall back to the plain kernels)
template <typename T, bool UNPACK>
static bool launch_pack_tiled(const float* src, const float* scale, T* dst, float* dw, const float* packed, int mode, int D0,
                              int D1, int K, int S, int P, int accumulate, hipStream_t s) {
    const size_t DK = (size_t)D1 * K;
    if (D0 * DK < 16384) return false;  // small weights: one plain launch is as good
    if (pack_fast0(mode)) {
        const size_t gy = (DK + 63) / 64;
        if (gy > 65535) return false;
        pack_tiled_kernel<T, true, UNPACK><<<dim3((D0 + 63) / 64, (unsigned)gy), 256, 0, s>>>(src, scale, dst, dw, packed, mode, D0, D1, K, S, P, accumulate);
    } else {
        if (K > 65 || (D1 + 63) / 64 > 65535) return false;
        pack_tiled_kernel<T, false, UNPACK><<<dim3(D0, (D1 + 63) / 64), 256, 0, s>>>(src, scale, dst, dw, packed, mode, D0, D1, K, S, P, accumulate);
    }
    return true;
}

// ---- every layer of a module in two launches (ga_pack_multi) -------------------------------------------------------
// A training step repacks ~40 weight tensors (two layouts each, plus the weight-norm scale): ~100 launches, most of them
// at the launch floor.  Here one launch computes every weight-norm row scale and one launch writes both packs of every
// layer; a workgroup finds its (layer, layout, tile) from a prefix table in the kernel arguments.
struct PackMultiArgs {
    ga_pack_item it[GA_PACK_MULTI_MAX];
    int first[2 * GA_PACK_MULTI_MAX + 1];  // first workgroup of (layer, layout) in the pack grid
    int row0[GA_PACK_MULTI_MAX + 1];       // first workgroup (= weight row) of a layer in the scale grid
    int n;
};

// 0: plain (4096 elements per workgroup), 1: FAST0 tiles, 2: one-s0 row tiles -- the choice launch_pack_tiled makes
__host__ __device__ static inline int pack_kind(int mode, int D0, int D1, int K) {
    const size_t DK = (size_t)D1 * K;
    if (D0 * DK < 16384) return 0;
    if (mode == GA_PACK_CONV_DGRAD_S1 || mode == GA_PACK_CONV_DGRAD_STRIDED || mode == GA_PACK_CONVT_FWD) return 1;
    return K > 65 ? 0 : 2;
}

static inline long pack_blocks(int mode, int D0, int D1, int K) {
    const size_t DK = (size_t)D1 * K;
    switch (pack_kind(mode, D0, D1, K)) {
        case 1: return (long)((D0 + 63) / 64) * (long)((DK + 63) / 64);
        case 2: return (long)D0 * ((D1 + 63) / 64);
        default: return (long)((D0 * DK + 4095) / 4096);
    }
}

template <typename T>
__device__ __forceinline__ void pack_one_block(float* tile, int lb, const float* __restrict__ src, const float* __restrict__ scale,
                                               T* __restrict__ dst, int mode, int D0, int D1, int K, int S, int P) {
    const int tid = threadIdx.x;
    const size_t DK = (size_t)D1 * K;
    const int kind = pack_kind(mode, D0, D1, K);
    if (kind == 1) {
        const int nbx = (D0 + 63) / 64;
        const int s0b = (lb % nbx) * 64;
        const size_t eb = (size_t)(lb / nbx) * 64;
        for (int idx = tid; idx < 4096; idx += 256) {
            const int sl = idx >> 6, el = idx & 63;
            const int s0 = s0b + sl;
            const size_t e = eb + el;
            if (s0 < D0 && e < DK) tile[sl * 65 + el] = src[(size_t)s0 * DK + e] * (scale ? scale[s0] : 1.f);
        }
        __syncthreads();
        for (int idx = tid; idx < 4096; idx += 256) {
            const int el = idx >> 6, sl = idx & 63;
            const int s0 = s0b + sl;
            const size_t e = eb + el;
            if (s0 < D0 && e < DK) dst[pack_index(mode, s0, (int)(e / K), (int)(e % K), D0, D1, K, S, P)] = (T)tile[sl * 65 + el];
        }
    } else if (kind == 2) {
        const int s0 = lb % D0;
        const int s1b = (lb / D0) * 64;
        const int ns1 = min(64, D1 - s1b);
        const int ne = ns1 * K;
        const size_t base = ((size_t)s0 * D1 + s1b) * K;
        const float sc = scale ? scale[s0] : 1.f;
        for (int idx = tid; idx < ne; idx += 256) tile[idx] = src[base + idx] * sc;
        __syncthreads();
        for (int idx = tid; idx < 64 * K; idx += 256) {
            const int k = idx >> 6, s1l = idx & 63;
            if (s1l < ns1) dst[pack_index(mode, s0, s1b + s1l, k, D0, D1, K, S, P)] = (T)tile[s1l * K + k];
        }
    } else {
        const size_t n = (size_t)D0 * DK;
        for (int idx = tid; idx < 4096; idx += 256) {
            const size_t i = (size_t)lb * 4096 + idx;
            if (i < n) {
                const int k = (int)(i % K);
                const int s1 = (int)((i / K) % D1);
                const int s0 = (int)(i / DK);
                dst[pack_index(mode, s0, s1, k, D0, D1, K, S, P)] = (T)(src[i] * (scale ? scale[s0] : 1.f));
            }
        }
    }
}

__global__ __launch_bounds__(256) void wn_scale_multi_kernel(const PackMultiArgs a) {
    __shared__ float red[16];
    int l = 0;
    while (l + 1 < a.n && (int)blockIdx.x >= a.row0[l + 1]) ++l;
    const ga_pack_item& it = a.it[l];
    const int r = blockIdx.x - a.row0[l];
    const int n = it.D1 * it.K;
    const float* __restrict__ v = it.v + (size_t)r * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s = fmaf(v[i], v[i], s);
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        const float nr = sqrtf(s);
        it.norm[r] = nr;
        it.scale[r] = it.g[r] / nr;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackMultiArgs a) {
    __shared__ float tile[64 * 65];
    int e = 0;
    while (e + 1 < 2 * a.n && (int)blockIdx.x >= a.first[e + 1]) ++e;
    const ga_pack_item& it = a.it[e >> 1];
    const int lb = blockIdx.x - a.first[e];
    const float* scale = it.g ? it.scale : nullptr;
    if (e & 1) pack_one_block<T>(tile, lb, it.v, scale, (T*)it.wp_d, it.mode_d, it.D0, it.D1, it.K, it.S, it.P);
    else pack_one_block<T>(tile, lb, it.v, scale, (T*)it.wp_f, it.mode_f, it.D0, it.D1, it.K, it.S, it.P);
}

int ga_pack_multi(const ga_pack_item* items, int n, int dtype, hipStream_t s) {
    if (!items || n < 1 || n > GA_PACK_MULTI_MAX || (dtype != GA_DTYPE_BF16 && dtype != GA_DTYPE_F32)) return GA_ERR_ARG;
    PackMultiArgs a;
    a.n = n;
    long blocks = 0, rows = 0;
    for (int l = 0; l < n; ++l) {
        const ga_pack_item& it = items[l];
        if (!it.v || !it.wp_f || !it.wp_d || it.D0 <= 0 || it.D1 <= 0 || it.K <= 0) return GA_ERR_ARG;
        if (it.g && (!it.scale || !it.norm)) return GA_ERR_ARG;
        for (int which = 0; which < 2; ++which) {
            const int mode = which ? it.mode_d : it.mode_f;
            if (mode < 0 || mode > 4) return GA_ERR_ARG;
            if ((mode == GA_PACK_CONV_DGRAD_STRIDED || mode == GA_PACK_CONVT_FWD) && it.S < 1) return GA_ERR_ARG;
            if (mode == GA_PACK_CONVT_FWD && it.K != 2 * it.S) return GA_ERR_ARG;
            a.first[2 * l + which] = (int)blocks;
            blocks += pack_blocks(mode, it.D0, it.D1, it.K);
        }
        a.row0[l] = (int)rows;
        rows += it.g ? it.D0 : 0;
        a.it[l] = it;
        if (blocks > (1L << 30) || rows > (1L << 30)) return GA_ERR_ARG;
    }
    a.first[2 * n] = (int)blocks;
    a.row0[n] = (int)rows;
    if (rows > 0) {
        wn_scale_multi_kernel<<<(unsigned)rows, 256, 0, s>>>(a);
        GA_CHECK_LAUNCH();
    }
    if (dtype == GA_DTYPE_BF16) pack_multi_kernel<bf16_t><<<(unsigned)blocks, 256, 0, s>>>(a);
    else pack_multi_kernel<float><<<(unsigned)blocks, 256, 0, s>>>(a);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

// weight-norm backward on the effective-weight gradient dw[R][n]:
//   dg[r] = <dw_r, v_r> / ||v_r||;  dv_r = (g/||v||) * dw_r - (g * <dw_r, v_r> / ||v||^3) * v_r
__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                      const float* __restrict__ g, const float* __restrict__ norm, int n,
                                                      float* __restrict__ dv, float* __restrict__ dg, int accumulate) {
    __shared__ float red[16];
    const int r = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s = fmaf(dw[(size_t)r * n + i], v[(size_t)r * n + i], s);
    s = block_sum(s, red);
    const float nr = norm[r], gr = g[r];
    const float a = gr / nr, bcoef = gr * s / (nr * nr * nr);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const size_t idx = (size_t)r * n + i;
        const float val = a * dw[idx] - bcoef * v[idx];
        dv[idx] = accumulate ? dv[idx] + val : val;
    }
    if (threadIdx.x == 0) dg[r] = accumulate ? dg[r] + s / nr : s / nr;
}

// ---- weight gradients of a whole module from the packed layout to their final place in one launch ------------------
// Per layer the backward pass used to run: unpack (packed [D0][K][D1] -> parameter layout), weight-norm backward, and
// the optimizer's copy of the result into its flat gradient buffer.  One workgroup per weight row does all three: the
// row's K*D1 packed gradients are contiguous; they are transposed through LDS, dotted with v, and written (scaled and
// corrected for the weight norm) straight to dv / dg, which the caller points into the flat buffer.
struct WgFinishArgs {
    ga_wfinish_item it[GA_PACK_MULTI_MAX];
    int row0[GA_PACK_MULTI_MAX + 1];
    int n;
};

__global__ __launch_bounds__(256) void wgrad_finish_multi_kernel(const WgFinishArgs a) {
    extern __shared__ float row[];
    __shared__ float red[16];
    int l = 0;
    while (l + 1 < a.n && (int)blockIdx.x >= a.row0[l + 1]) ++l;
    const ga_wfinish_item& it = a.it[l];
    const int r = blockIdx.x - a.row0[l];
    const int D1 = it.D1, K = it.K, n = D1 * K;
    const float* __restrict__ gp = it.gpacked + (size_t)r * n;
    for (int idx = threadIdx.x; idx < n; idx += 256) {
        const int k = idx / D1, s1 = idx - k * D1;
        row[s1 * K + k] = gp[idx];
    }
    __syncthreads();
    float* __restrict__ dv = it.dv + (size_t)r * n;
    const bool acc_v = it.accumulate & 1;
    if (!it.g) {
        for (int i = threadIdx.x; i < n; i += 256) dv[i] = acc_v ? dv[i] + row[i] : row[i];
        return;
    }
    const float* __restrict__ v = it.v + (size_t)r * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s = fmaf(row[i], v[i], s);
    s = block_sum(s, red);
    const float nr = it.norm[r], gr = it.g[r];
    const float ca = gr / nr, cb = gr * s / (nr * nr * nr);
    for (int i = threadIdx.x; i < n; i += 256) {
        const float val = ca * row[i] - cb * v[i];
        dv[i] = acc_v ? dv[i] + val : val;
    }
    if (threadIdx.x == 0) it.dg[r] = (it.accumulate & 2) ? it.dg[r] + s / nr : s / nr;
}

int ga_wgrad_finish_multi(const ga_wfinish_item* items, int n, hipStream_t s) {
    if (!items || n < 1 || n > GA_PACK_MULTI_MAX) return GA_ERR_ARG;
    WgFinishArgs a;
    a.n = n;
    long rows = 0;
    int nmax = 0;
    for (int l = 0; l < n; ++l) {
        const ga_wfinish_item& it = items[l];
        if (!it.gpacked || !it.dv || it.D0 <= 0 || it.D1 <= 0 || it.K <= 0) return GA_ERR_ARG;
        if ((long)it.D1 * it.K > GA_WFINISH_MAX_ROW) return GA_ERR_ARG;
        if (it.g && (!it.v || !it.norm || !it.dg)) return GA_ERR_ARG;
        a.it[l] = it;
        a.row0[l] = (int)rows;
        rows += it.D0;
        if (it.D1 * it.K > nmax) nmax = it.D1 * it.K;
        if (rows > (1L << 30)) return GA_ERR_ARG;
    }
    a.row0[n] = (int)rows;
    wgrad_finish_multi_kernel<<<(unsigned)rows, 256, (size_t)nmax * sizeof(float), s>>>(a);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

// 16 bytes of T (4 floats / 8 bf16)
template <typename T> struct alignas(16) Pack16 { T v[16 / sizeof(T)]; };

// ================================================================================================
// GroupNorm(G) -> FiLM -> SiLU (+ residual), channels-last
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int Tn, int C, int G, float eps,
                                                        float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ float red[16];
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int cg = C / G;
    const int n = Tn * cg;
    const T* xb = x + (size_t)b * Tn * C + g * cg;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += (float)xb[(size_t)(i / cg) * C + (i % cg)];
    const float mu = block_sum(s, red) / n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float d = (float)xb[(size_t)(i / cg) * C + (i % cg)] - mu;
        q = fmaf(d, d, q);
    }
    q = block_sum(q, red) / n;
    if (threadIdx.x == 0) {
        mean[blockIdx.x] = mu;
        rstd[blockIdx.x] = rsqrtf(q + eps);
    }
}

// y = act( ((x-mean)*rstd*gamma+beta) * (1+a[b,c]) + bb[b,c] ) + res ;  ab = [B][2C] (a | b) or null
template <typename T>
__global__ void gn_film_act_fwd_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ ab, const T* __restrict__ res, T* __restrict__ y,
                                       int B, int Tn, int C, int G, int act) {
    const size_t n = (size_t)B * Tn * C;
    const int cg = C / G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int b = (int)(i / ((size_t)Tn * C));
        const int sg = b * G + c / cg;
        float v = ((float)x[i] - mean[sg]) * rstd[sg] * gamma[c] + beta[c];
        if (ab) v = v * (1.f + ab[(size_t)b * 2 * C + c]) + ab[(size_t)b * 2 * C + C + c];
        v = act_apply(act, v);
        if (res) v += (float)res[i];
        y[i] = (T)v;
    }
}

__device__ __forceinline__ float dsilu(float h) {
    const float s = 1.f / (1.f + __expf(-h));
    return s * (1.f + h * (1.f - s));
}

// one block per (b, group).  dgamma/dbeta [C] and dab [B][2C] are accumulated with atomics (caller zeroes).
template <typename T>
__global__ __launch_bounds__(256) void gn_film_act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ ab, T* __restrict__ dx,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ dab, int Tn, int C, int G, int act) {
    __shared__ float red[16];
    __shared__ float chan[4][256];
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int cg = C / G;
    const int rpi = 256 / cg;  // rows per iteration
    const int tid = threadIdx.x;
    const bool active = tid < rpi * cg;
    const int cl = tid % cg, r0 = tid / cg;
    const int c = g * cg + cl;
    const float mu = mean[blockIdx.x], rs = rstd[blockIdx.x];
    const float gm = gamma[c], bt = beta[c];
    const float fa = ab ? ab[(size_t)b * 2 * C + c] : 0.f;
    const float fb = ab ? ab[(size_t)b * 2 * C + C + c] : 0.f;
    const size_t base = (size_t)b * Tn * C + c;
    float s_da = 0.f, s_db = 0.f, s_dg = 0.f, s_dbt = 0.f, s1 = 0.f, s2 = 0.f;
    if (active) {
        for (int t = r0; t < Tn; t += rpi) {
            const size_t idx = base + (size_t)t * C;
            const float xh = ((float)x[idx] - mu) * rs;
            const float nv = xh * gm + bt;
            const float h = nv * (1.f + fa) + fb;
            float dh = (float)dy[idx];
            if (act == GA_ACT_SILU) dh *= dsilu(h);
            s_da = fmaf(dh, nv, s_da);
            s_db += dh;
            const float dn = dh * (1.f + fa);
            s_dg = fmaf(dn, xh, s_dg);
            s_dbt += dn;
            const float dxh = dn * gm;
            s1 += dxh;
            s2 = fmaf(dxh, xh, s2);
        }
    }
    const float S1 = block_sum(s1, red);
    const float S2 = block_sum(s2, red);
    chan[0][tid] = s_da; chan[1][tid] = s_db; chan[2][tid] = s_dg; chan[3][tid] = s_dbt;
    __syncthreads();
    if (tid < cg) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int r = 0; r < rpi; ++r) {
            a0 += chan[0][r * cg + tid]; a1 += chan[1][r * cg + tid];
            a2 += chan[2][r * cg + tid]; a3 += chan[3][r * cg + tid];
        }
        const int cc = g * cg + tid;
        if (dab) {
            atomicAdd(&dab[(size_t)b * 2 * C + cc], a0);
            atomicAdd(&dab[(size_t)b * 2 * C + C + cc], a1);
        }
        atomicAdd(&dgamma[cc], a2);
        atomicAdd(&dbeta[cc], a3);
    }
    const float inv_n = 1.f / (float)(Tn * cg);
    if (active) {
        for (int t = r0; t < Tn; t += rpi) {
            const size_t idx = base + (size_t)t * C;
            const float xh = ((float)x[idx] - mu) * rs;
            const float nv = xh * gm + bt;
            const float h = nv * (1.f + fa) + fb;
            float dh = (float)dy[idx];
            if (act == GA_ACT_SILU) dh *= dsilu(h);
            const float dxh = dh * (1.f + fa) * gm;
            dx[idx] = (T)(rs * (dxh - S1 * inv_n - xh * S2 * inv_n));
        }
    }
}


// Vectorised GroupNorm kernels (channels per group a multiple of 16 bytes).  One block per (b, group) as above, but
// every thread moves 16-byte pieces, the first MAXR row passes stay in registers (one trip to memory for T <= MAXR *
// rows-per-pass), and all loads of a pass are in flight together: the scalar versions were latency-bound.
#define GN_MAXR 8
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_vec_kernel(const T* __restrict__ x, int Tn, int C, int G, float eps,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            const float* __restrict__ gamma = nullptr,
                                                            const float* __restrict__ beta = nullptr,
                                                            const float* __restrict__ ab = nullptr,
                                                            const T* __restrict__ res = nullptr, T* __restrict__ y = nullptr,
                                                            int act = 0) {
    constexpr int EV = 16 / sizeof(T);
    __shared__ float red[16];
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int cg = C / G, parts = cg / EV, rpp = 256 / parts;
    const int tid = threadIdx.x;
    const bool active = tid < rpp * parts;
    const int part = tid % parts, r0 = tid / parts;
    const T* xb = x + (size_t)b * Tn * C + g * cg + part * EV;
    Pack16<T> cache[GN_MAXR];
#pragma unroll
    for (int k = 0; k < GN_MAXR; ++k) {
        const int t = r0 + k * rpp;
        const int tc = t < Tn ? t : Tn - 1;
        cache[k] = *(const Pack16<T>*)(xb + (size_t)tc * C);
    }
    float sacc = 0.f;
#pragma unroll
    for (int k = 0; k < GN_MAXR; ++k) {
        if (active && r0 + k * rpp < Tn) {
#pragma unroll
            for (int e = 0; e < EV; ++e) sacc += (float)cache[k].v[e];
        }
    }
    if (active)
        for (int t = r0 + GN_MAXR * rpp; t < Tn; t += rpp) {
            const Pack16<T> v = *(const Pack16<T>*)(xb + (size_t)t * C);
#pragma unroll
            for (int e = 0; e < EV; ++e) sacc += (float)v.v[e];
        }
    const float n = (float)Tn * cg;
    const float mu = block_sum(sacc, red) / n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < GN_MAXR; ++k) {
        if (active && r0 + k * rpp < Tn) {
#pragma unroll
            for (int e = 0; e < EV; ++e) { const float d = (float)cache[k].v[e] - mu; q = fmaf(d, d, q); }
        }
    }
    if (active)
        for (int t = r0 + GN_MAXR * rpp; t < Tn; t += rpp) {
            const Pack16<T> v = *(const Pack16<T>*)(xb + (size_t)t * C);
#pragma unroll
            for (int e = 0; e < EV; ++e) { const float d = (float)v.v[e] - mu; q = fmaf(d, d, q); }
        }
    q = block_sum(q, red) / n;
    const float rs = rsqrtf(q + eps);
    if (tid == 0) {
        mean[blockIdx.x] = mu;
        rstd[blockIdx.x] = rs;
    }
    if (!y || !active) return;
    // fused forward (ga_gn_film_act_fused): normalise, FiLM, activation (+ residual) on the rows still in registers
    const int c0 = g * cg + part * EV;
    float gm[EV], bt[EV], fa[EV], fb[EV];
#pragma unroll
    for (int e = 0; e < EV; ++e) {
        gm[e] = gamma[c0 + e];
        bt[e] = beta[c0 + e];
        fa[e] = ab ? 1.f + ab[(size_t)b * 2 * C + c0 + e] : 1.f;
        fb[e] = ab ? ab[(size_t)b * 2 * C + C + c0 + e] : 0.f;
    }
    const size_t base = (size_t)b * Tn * C + c0;
    auto emit = [&](const Pack16<T>& vx, int t) __attribute__((always_inline)) {
        Pack16<T> o, r;
        if (res) r = *(const Pack16<T>*)(res + base + (size_t)t * C);
#pragma unroll
        for (int e = 0; e < EV; ++e) {
            float v = ((float)vx.v[e] - mu) * rs * gm[e] + bt[e];  // the same expression, in the same order, as the two-launch path
            if (ab) v = v * fa[e] + fb[e];
            v = act_apply(act, v);
            if (res) v += (float)r.v[e];
            o.v[e] = (T)v;
        }
        *(Pack16<T>*)(y + base + (size_t)t * C) = o;
    };
#pragma unroll
    for (int k = 0; k < GN_MAXR; ++k)
        if (r0 + k * rpp < Tn) emit(cache[k], r0 + k * rpp);
    for (int t = r0 + GN_MAXR * rpp; t < Tn; t += rpp) emit(*(const Pack16<T>*)(xb + (size_t)t * C), t);
}

template <typename T>
__global__ __launch_bounds__(256) void gn_film_act_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   const float* __restrict__ ab, T* __restrict__ dx,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   float* __restrict__ dab, int Tn, int C, int G, int act) {
    constexpr int EV = 16 / sizeof(T);
    __shared__ float red[16];
    __shared__ float chan[4][256][EV + 1];
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int cg = C / G, parts = cg / EV, rpp = 256 / parts;
    const int tid = threadIdx.x;
    const bool active = tid < rpp * parts;
    const int part = tid % parts, r0 = tid / parts;
    const int c0 = g * cg + part * EV;
    const float mu = mean[blockIdx.x], rs = rstd[blockIdx.x];
    float gm[EV], bt[EV], fa[EV], fb[EV];
#pragma unroll
    for (int e = 0; e < EV; ++e) {
        gm[e] = gamma[c0 + e]; bt[e] = beta[c0 + e];
        fa[e] = ab ? ab[(size_t)b * 2 * C + c0 + e] : 0.f;
        fb[e] = ab ? ab[(size_t)b * 2 * C + C + c0 + e] : 0.f;
    }
    const size_t base = (size_t)b * Tn * C + c0;
    Pack16<T> cx[GN_MAXR], cd[GN_MAXR];
#pragma unroll
    for (int k = 0; k < GN_MAXR; ++k) {
        const int t = r0 + k * rpp;
        const int tc = t < Tn ? t : Tn - 1;
        cx[k] = *(const Pack16<T>*)(x + base + (size_t)tc * C);
        cd[k] = *(const Pack16<T>*)(dy + base + (size_t)tc * C);
    }
    float s_da[EV], s_db[EV], s_dg[EV], s_dbt[EV], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < EV; ++e) s_da[e] = s_db[e] = s_dg[e] = s_dbt[e] = 0.f;
    auto accum = [&](const Pack16<T>& vx, const Pack16<T>& vd) {
#pragma unroll
        for (int e = 0; e < EV; ++e) {
            const float xh = ((float)vx.v[e] - mu) * rs;
            const float nv = xh * gm[e] + bt[e];
            const float h = nv * (1.f + fa[e]) + fb[e];
            float dh = (float)vd.v[e];
            if (act == GA_ACT_SILU) dh *= dsilu(h);
            s_da[e] = fmaf(dh, nv, s_da[e]);
            s_db[e] += dh;
            const float dn = dh * (1.f + fa[e]);
            s_dg[e] = fmaf(dn, xh, s_dg[e]);
            s_dbt[e] += dn;
            const float dxh = dn * gm[e];
            s1 += dxh;
            s2 = fmaf(dxh, xh, s2);
        }
    };
#pragma unroll
    for (int k = 0; k < GN_MAXR; ++k)
        if (active && r0 + k * rpp < Tn) accum(cx[k], cd[k]);
    if (active)
        for (int t = r0 + GN_MAXR * rpp; t < Tn; t += rpp)
            accum(*(const Pack16<T>*)(x + base + (size_t)t * C), *(const Pack16<T>*)(dy + base + (size_t)t * C));
    const float S1 = block_sum(s1, red);
    const float S2 = block_sum(s2, red);
#pragma unroll
    for (int e = 0; e < EV; ++e) {
        chan[0][tid][e] = s_da[e]; chan[1][tid][e] = s_db[e]; chan[2][tid][e] = s_dg[e]; chan[3][tid][e] = s_dbt[e];
    }
    __syncthreads();
    for (int o = tid; o < 4 * cg; o += 256) {
        const int which = o / cg, cl = o % cg;
        const int pt = cl / EV, e = cl % EV;
        float a = 0.f;
        for (int r = 0; r < rpp; ++r) a += chan[which][r * parts + pt][e];
        const int cc = g * cg + cl;
        if (which == 0) { if (dab) atomicAdd(&dab[(size_t)b * 2 * C + cc], a); }
        else if (which == 1) { if (dab) atomicAdd(&dab[(size_t)b * 2 * C + C + cc], a); }
        else if (which == 2) atomicAdd(&dgamma[cc], a);
        else atomicAdd(&dbeta[cc], a);
    }
    const float inv_n = 1.f / ((float)Tn * cg);
    auto emit = [&](const Pack16<T>& vx, const Pack16<T>& vd, int t) {
        Pack16<T> o;
#pragma unroll
        for (int e = 0; e < EV; ++e) {
            const float xh = ((float)vx.v[e] - mu) * rs;
            const float nv = xh * gm[e] + bt[e];
            const float h = nv * (1.f + fa[e]) + fb[e];
            float dh = (float)vd.v[e];
            if (act == GA_ACT_SILU) dh *= dsilu(h);
            const float dxh = dh * (1.f + fa[e]) * gm[e];
            o.v[e] = (T)(rs * (dxh - S1 * inv_n - xh * S2 * inv_n));
        }
        *(Pack16<T>*)(dx + base + (size_t)t * C) = o;
    };
#pragma unroll
    for (int k = 0; k < GN_MAXR; ++k)
        if (active && r0 + k * rpp < Tn) emit(cx[k], cd[k], r0 + k * rpp);
    if (active)
        for (int t = r0 + GN_MAXR * rpp; t < Tn; t += rpp)
            emit(*(const Pack16<T>*)(x + base + (size_t)t * C), *(const Pack16<T>*)(dy + base + (size_t)t * C), t);
}

// ================================================================================================
// q_hat = q_src + rscale * tanh(raw)
template <typename T>
__global__ void tanh_residual_fwd_kernel(const T* __restrict__ raw, const T* __restrict__ q, T* __restrict__ y, size_t n, float rscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = (T)((float)q[i] + rscale * tanhf((float)raw[i]));
}
template <typename T>
__global__ void tanh_residual_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ raw, T* __restrict__ draw, size_t n, float rscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float th = tanhf((float)raw[i]);
        draw[i] = (T)((float)dy[i] * rscale * (1.f - th * th));
    }
}

// AvgPool1d(kernel 4, stride 2, padding 1, count_include_pad) on single-channel rows [B][L] -> [B][Lo]
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int L, int Lo) {
    const size_t n = (size_t)B * Lo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / Lo), t = (int)(i % Lo);
        const T* xb = x + (size_t)b * L;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int l = 2 * t - 1 + k;
            if (l >= 0 && l < L) s += (float)xb[l];
        }
        y[i] = (T)(0.25f * s);
    }
}
template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int L, int Lo) {
    const size_t n = (size_t)B * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / L), l = (int)(i % L);
        // outputs t with 2t-1 <= l <= 2t+2  <=>  (l-2)/2 <= t <= (l+1)/2
        float s = 0.f;
        const int tlo = (l - 1) >> 1, thi = (l + 1) >> 1;  // ceil((l-2)/2) == (l-1)>>1 for l>=0 (arith shift handles l=0)
        for (int t = tlo; t <= thi; ++t)
            if (t >= 0 && t < Lo) s += (float)dy[(size_t)b * Lo + t];
        dx[i] = (T)(0.25f * s);
    }
}

// ga = gy * act'(.) : mode LRELU_FROM_OUT uses the post-activation output
// 16 bytes per lane per access (EV = 4 floats / 8 bf16); the last n % EV elements are handled by thread 0 of block 0.

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ src, T* __restrict__ ga, size_t n, int mode) {
    constexpr int EV = 16 / sizeof(T);
    const size_t nv = n / EV;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        const Pack16<T> a = ((const Pack16<T>*)gy)[i], b = ((const Pack16<T>*)src)[i];
        Pack16<T> o;
#pragma unroll
        for (int e = 0; e < EV; ++e) o.v[e] = (T)((float)a.v[e] * dact_apply(mode, (float)b.v[e]));
        ((Pack16<T>*)ga)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = nv * EV; i < n; ++i) ga[i] = (T)((float)gy[i] * dact_apply(mode, (float)src[i]));
}

// ga = gy * act'(src) AND dbias[m] += sum_rows ga[row][m] in the same pass (rows of M channels, M/EV a divisor of 256, so a
// thread keeps the same EV channels for every row it visits: partial sums live in registers, one LDS reduction per block)
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_bias_kernel(const T* __restrict__ gy, const T* __restrict__ src, T* __restrict__ ga,
                                                            size_t nv, int mode, int M, float* __restrict__ dbias) {
    constexpr int EV = 16 / sizeof(T);
    __shared__ float red[256][EV + 1];
    const int MV = M / EV;
    float acc[EV];
#pragma unroll
    for (int e = 0; e < EV; ++e) acc[e] = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        const Pack16<T> a = ((const Pack16<T>*)gy)[i], b = ((const Pack16<T>*)src)[i];
        Pack16<T> o;
#pragma unroll
        for (int e = 0; e < EV; ++e) {
            const float v = (float)a.v[e] * dact_apply(mode, (float)b.v[e]);
            o.v[e] = (T)v;
            acc[e] += (float)o.v[e];  // the rounded value: what the weight-gradient kernel will see
        }
        ((Pack16<T>*)ga)[i] = o;
    }
#pragma unroll
    for (int e = 0; e < EV; ++e) red[threadIdx.x][e] = acc[e];
    __syncthreads();
    // thread tid owns vector column (tid % MV) for all of its rows (gridDim.x * 256 is a multiple of MV)
    for (int o = threadIdx.x; o < M; o += 256) {
        const int col = o / EV, e = o % EV;
        float sacc = 0.f;
        for (int t = col; t < 256; t += MV) sacc += red[t][e];
        atomicAdd(&dbias[o], sacc);
    }
}

// y = act(x)
template <typename T>
__global__ void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n, int act) {
    constexpr int EV = 16 / sizeof(T);
    const size_t nv = n / EV;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        const Pack16<T> a = ((const Pack16<T>*)x)[i];
        Pack16<T> o;
#pragma unroll
        for (int e = 0; e < EV; ++e) o.v[e] = (T)act_apply(act, (float)a.v[e]);
        ((Pack16<T>*)y)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t i = nv * EV; i < n; ++i) y[i] = (T)act_apply(act, (float)x[i]);
}

// out[m] += sum_rows g[row][m]   (bias gradient)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, size_t rows, int M, float* __restrict__ out, size_t rows_per_block) {
    const size_t r0 = (size_t)blockIdx.x * rows_per_block;
    size_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    if (M <= 128) {
        // threads: channel = tid % Mp, row lane = tid / Mp
        __shared__ float part[256];
        int Mp = 1;
        while (Mp < M) Mp <<= 1;
        const int rl = threadIdx.x / Mp, m = threadIdx.x % Mp, nrl = 256 / Mp;
        float s = 0.f;
        if (m < M)
            for (size_t r = r0 + rl; r < r1; r += nrl) s += (float)g[r * M + m];
        part[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < Mp && m < M) {
            float t = 0.f;
            for (int k = 0; k < nrl; ++k) t += part[k * Mp + m];
            atomicAdd(&out[m], t);
        }
    } else {
        for (int m = threadIdx.x; m < M; m += blockDim.x) {
            float s = 0.f;
            for (size_t r = r0; r < r1; ++r) s += (float)g[r * M + m];
            atomicAdd(&out[m], s);
        }
    }
}

// ================================================================================================
// scalar reductions out[0] += scale * sum_i f(a_i, b_i) and their elementwise gradients

__device__ __forceinline__ float red_f(int op, float a, float b) {
    switch (op) {
        case GA_RED_ABSDIFF: return fabsf(a - b);
        case GA_RED_RELU1M: return fmaxf(1.f - a, 0.f);
        case GA_RED_RELU1P: return fmaxf(1.f + a, 0.f);
        case GA_RED_SUM: return a;
        default: return a * a;
    }
}
__device__ __forceinline__ float red_df(int op, float a, float b) {
    switch (op) {
        case GA_RED_ABSDIFF: { const float d = a - b; return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
        case GA_RED_RELU1M: return (1.f - a) > 0.f ? -1.f : 0.f;
        case GA_RED_RELU1P: return (1.f + a) > 0.f ? 1.f : 0.f;
        case GA_RED_SUM: return 1.f;
        default: return 2.f * a;
    }
}

// 16 bytes per lane per trip when the operands are 16-byte aligned (vec != 0); scalar elements otherwise and for the tail.
// Block `lb` of `nb` blocks working on one (a, b, n) problem; returns the block's sum in thread 0.
template <typename T>
__device__ __forceinline__ float reduce_block(const T* __restrict__ a, const T* __restrict__ b, size_t n, int op, int vec, int lb, int nb,
                                              float* red) {
    constexpr int EV = 16 / sizeof(T);
    float s = 0.f;
    const size_t nv = vec ? n / EV : 0;
    // four pieces per operand and trip, all requested before the first is used (bytes in flight, not arithmetic, set the pace:
    // with one piece per trip the feature-matching sums ran at 4 TB/s)
    const size_t stride = (size_t)nb * blockDim.x;
    size_t i = (size_t)lb * blockDim.x + threadIdx.x;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (; i + 3 * stride < nv; i += 4 * stride) {
        Pack16<T> av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            av[u] = ((const Pack16<T>*)a)[i + u * stride];
            if (b) bv[u] = ((const Pack16<T>*)b)[i + u * stride];
        }
#pragma unroll
        for (int e = 0; e < EV; ++e) {
            s += red_f(op, (float)av[0].v[e], b ? (float)bv[0].v[e] : 0.f);
            s1 += red_f(op, (float)av[1].v[e], b ? (float)bv[1].v[e] : 0.f);
            s2 += red_f(op, (float)av[2].v[e], b ? (float)bv[2].v[e] : 0.f);
            s3 += red_f(op, (float)av[3].v[e], b ? (float)bv[3].v[e] : 0.f);
        }
    }
    for (; i < nv; i += stride) {
        const Pack16<T> av = ((const Pack16<T>*)a)[i];
        Pack16<T> bv;
        if (b) bv = ((const Pack16<T>*)b)[i];
#pragma unroll
        for (int e = 0; e < EV; ++e) s += red_f(op, (float)av.v[e], b ? (float)bv.v[e] : 0.f);
    }
    for (size_t k = nv * EV + (size_t)lb * blockDim.x + threadIdx.x; k < n; k += stride)
        s += red_f(op, (float)a[k], b ? (float)b[k] : 0.f);
    return block_sum((s + s1) + (s2 + s3), red);
}

template <typename T>
__global__ __launch_bounds__(256) void reduce_kernel(const T* __restrict__ a, const T* __restrict__ b, size_t n, int op,
                                                      float scale, float* __restrict__ out, int vec) {
    __shared__ float red[16];
    const float s = reduce_block<T>(a, b, n, op, vec, blockIdx.x, gridDim.x, red);
    if (threadIdx.x == 0) atomicAdd(out, s * scale);
}

// several (a, b, n, scale) problems of one op summed into ONE scalar by one launch: the feature-matching term of a discriminator
// scale is five such sums over maps of 1 GB down to 15 MB, and every launch of its own paid ~15 us of ramp and of same-address atomics
#define GA_REDUCE_MULTI_MAX 8
struct ReduceMultiArgs {
    const void* a[GA_REDUCE_MULTI_MAX];
    const void* b[GA_REDUCE_MULTI_MAX];
    unsigned long long n[GA_REDUCE_MULTI_MAX];
    float scale[GA_REDUCE_MULTI_MAX];
    int start[GA_REDUCE_MULTI_MAX + 1];  // first block of each problem
    int vec[GA_REDUCE_MULTI_MAX];
    int cnt, op;
};
template <typename T>
__global__ __launch_bounds__(256) void reduce_multi_kernel(ReduceMultiArgs A, float* __restrict__ out) {
    __shared__ float red[16];
    int i = 0;
    while (i + 1 < A.cnt && (int)blockIdx.x >= A.start[i + 1]) ++i;
    const float s = reduce_block<T>((const T*)A.a[i], (const T*)A.b[i], (size_t)A.n[i], A.op, A.vec[i], blockIdx.x - A.start[i],
                                    A.start[i + 1] - A.start[i], red);
    if (threadIdx.x == 0) atomicAdd(out, s * A.scale[i]);
}

// ga[i] (+)= coef[0] * scale * d f / d a
template <typename T>
__global__ void reduce_grad_kernel(const T* __restrict__ a, const T* __restrict__ b, size_t n, int op, const float* __restrict__ coef,
                                   float scale, T* __restrict__ ga, int accumulate, int vec) {
    constexpr int EV = 16 / sizeof(T);
    const float c = coef[0] * scale;
    const size_t nv = vec ? n / EV : 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        const Pack16<T> av = ((const Pack16<T>*)a)[i];
        Pack16<T> bv, gv, o;
        if (b) bv = ((const Pack16<T>*)b)[i];
        if (accumulate) gv = ((const Pack16<T>*)ga)[i];
#pragma unroll
        for (int e = 0; e < EV; ++e) {
            float v = c * red_df(op, (float)av.v[e], b ? (float)bv.v[e] : 0.f);
            if (accumulate) v += (float)gv.v[e];
            o.v[e] = (T)v;
        }
        ((Pack16<T>*)ga)[i] = o;
    }
    for (size_t i = nv * EV + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = c * red_df(op, (float)a[i], b ? (float)b[i] : 0.f);
        if (accumulate) v += (float)ga[i];
        ga[i] = (T)v;
    }
}

// latent losses on [B][T][C]: out[0] = sum |d|, out[1] = sum_t |d[t+1]-d[t]|, out[2+b] = sum_b |d|, d = qh - qs
template <typename T>
__global__ __launch_bounds__(256) void latent_loss_fwd_kernel(const T* __restrict__ qh, const T* __restrict__ qs, int Tn, int C,
                                                               float* __restrict__ out) {
    __shared__ float red[16];
    const int b = blockIdx.y;
    const size_t base = (size_t)b * Tn * C;
    const size_t n = (size_t)Tn * C;
    float s0 = 0.f, s1 = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = (float)qh[base + i] - (float)qs[base + i];
        s0 += fabsf(d);
        if (i + C < n) {
            const float d2 = (float)qh[base + i + C] - (float)qs[base + i + C];
            s1 += fabsf(d2 - d);
        }
    }
    s0 = block_sum(s0, red);
    s1 = block_sum(s1, red);
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], s0);
        atomicAdd(&out[1], s1);
        atomicAdd(&out[2 + b], s0);
    }
}
__device__ __forceinline__ float sgn(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
template <typename T>
__global__ void latent_loss_bwd_kernel(const T* __restrict__ qh, const T* __restrict__ qs, int B, int Tn, int C,
                                       const float* __restrict__ gout /* [2+B] */, T* __restrict__ dqh) {
    const size_t per = (size_t)Tn * C, n = (size_t)B * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const size_t r = i % per;
        const float d = (float)qh[i] - (float)qs[i];
        float g = (gout[0] + gout[2 + b]) * sgn(d);
        if (r >= (size_t)C) {
            const float dm = (float)qh[i - C] - (float)qs[i - C];
            g += gout[1] * sgn(d - dm);
        }
        if (r + C < per) {
            const float dp = (float)qh[i + C] - (float)qs[i + C];
            g -= gout[1] * sgn(dp - d);
        }
        dqh[i] = (T)g;
    }
}

// ================================================================================================
// optimizer: out[0] += sum g^2 ; fused clip + AdamW over flat fp32 buffers
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, size_t n, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    const size_t n4 = n / 4;
    const float4* g4 = (const float4*)g;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (; i + 3 * stride < n4; i += 4 * stride) {  // four loads in flight per lane
        const float4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
        s += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
        s1 += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
        s2 += v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w;
        s3 += v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w;
    }
    for (; i < n4; i += stride) {
        const float4 v = g4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    for (size_t k = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) s += g[k] * g[k];
    s = block_sum((s + s1) + (s2 + s3), red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

struct AdamWArgs {
    float lr, beta1, beta2, eps, weight_decay, bc1, bc2, max_norm, grad_scale;
};

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamWArgs& a, float coef) {
    g *= coef;
    p *= (1.f - a.lr * a.weight_decay);
    m = a.beta1 * m + (1.f - a.beta1) * g;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / sqrtf(a.bc2) + a.eps;
    p -= (a.lr / a.bc1) * (m / denom);
}

// torch.nn.utils.clip_grad_norm_ (coef = min(1, max_norm/(norm+1e-6))) folded into torch.optim.AdamW's update.
// g_out (optional) receives the clipped gradient so p.grad matches the reference after the step.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, size_t n, AdamWArgs a, const float* __restrict__ sqnorm,
                                                     int write_clipped_grad, const int* __restrict__ step_dev) {
    if (step_dev) {  // the step number lives in device memory (a recorded HIP graph replays this launch with the same arguments)
        const int step = step_dev[0];
        a.bc1 = (float)(1.0 - pow((double)a.beta1, (double)step));
        a.bc2 = (float)(1.0 - pow((double)a.beta2, (double)step));
    }
    float coef = a.grad_scale;
    if (a.max_norm > 0.f && sqnorm) {
        const float total = sqrtf(sqnorm[0]) * a.grad_scale;
        const float c = a.max_norm / (total + 1e-6f);
        if (c < 1.f) coef *= c;
    }
    const size_t n4 = n / 4;
    float4* p4 = (float4*)p; float4* g4 = (float4*)g; float4* m4 = (float4*)m; float4* v4 = (float4*)v;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        adamw_one(pp.x, gg.x, mm.x, vv.x, a, coef);
        adamw_one(pp.y, gg.y, mm.y, vv.y, a, coef);
        adamw_one(pp.z, gg.z, mm.z, vv.z, a, coef);
        adamw_one(pp.w, gg.w, mm.w, vv.w, a, coef);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (write_clipped_grad) g4[i] = make_float4(gg.x * coef, gg.y * coef, gg.z * coef, gg.w * coef);
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float pp = p[i], mm = m[i], vv = v[i];
        adamw_one(pp, g[i], mm, vv, a, coef);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (write_clipped_grad) g[i] *= coef;
    }
}

// ================================================================================================
// C ABI
int ga_colsum_vec_supported(const void* g, int dtype, int M);
int ga_colsum_vec_impl(const void* g, int dtype, size_t rows, int M, float* out, hipStream_t s);
// ================================================================================================
// 2-D convolutions as 1-D implicit GEMMs (gan_amd/conv2d.py): the kernel ROWS of a channels-last image gathered into the channel axis.
//   y[b][oh][w][i*C + c] = x[b][S*oh + i - P][w][c]   (zero outside the image; channels KH*C .. KC-1 of y are zero padding)
// and the adjoint  gx[b][h][w][c] = sum_{i : (h + P - i) % S == 0, oh = (h + P - i) / S < OH} gy[b][oh][w][i*C + c].
// One thread per 16-byte piece when C is a multiple of the piece, per element otherwise (the discriminator's one-channel input).
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void rowgather_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int KH, int S,
                                                            int P, int OH, int KC) {
    constexpr int EV = VEC ? 16 / (int)sizeof(T) : 1;
    const int kcv = KC / EV, cv = C / EV;
    const size_t total = (size_t)B * OH * W * kcv;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % kcv);
        size_t r = idx / kcv;
        const int w = (int)(r % W);
        r /= W;
        const int oh = (int)(r % OH), b = (int)(r / OH);
        const int i = k / cv, c = (k - i * cv) * EV;
        const int h = S * oh + i - P;
        const bool ok = i < KH && h >= 0 && h < H;
        if (VEC) {
            Pack16<T> v;
            if (ok) v = *(const Pack16<T>*)(x + (((size_t)b * H + h) * W + w) * C + c);
            else
#pragma unroll
                for (int e = 0; e < EV; ++e) v.v[e] = (T)0.f;
            *(Pack16<T>*)(y + idx * EV) = v;
        } else {
            y[idx] = ok ? x[(((size_t)b * H + h) * W + w) * C + c] : (T)0.f;
        }
    }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void rowgather_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int B, int H, int W, int C, int KH, int S,
                                                            int P, int OH, int KC) {
    constexpr int EV = VEC ? 16 / (int)sizeof(T) : 1;
    const int cv = C / EV;
    const size_t total = (size_t)B * H * W * cv;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv) * EV;
        size_t r = idx / cv;
        const int w = (int)(r % W);
        r /= W;
        const int h = (int)(r % H), b = (int)(r / H);
        float acc[EV];
#pragma unroll
        for (int e = 0; e < EV; ++e) acc[e] = 0.f;
        for (int i = 0; i < KH; ++i) {
            const int t = h + P - i;
            if (t < 0 || t % S) continue;
            const int oh = t / S;
            if (oh >= OH) continue;
            const T* src = gy + (((size_t)b * OH + oh) * W + w) * KC + (size_t)i * C + c;
            if (VEC) {
                const Pack16<T> v = *(const Pack16<T>*)src;
#pragma unroll
                for (int e = 0; e < EV; ++e) acc[e] += (float)v.v[e];
            } else {
                acc[0] += (float)src[0];
            }
        }
        if (VEC) {
            Pack16<T> o;
#pragma unroll
            for (int e = 0; e < EV; ++e) o.v[e] = (T)acc[e];
            *(Pack16<T>*)(gx + idx * EV) = o;
        } else {
            gx[idx] = (T)acc[0];
        }
    }
}

static inline int grid_for(size_t n, int block = 256, int cap = 2048) {
    size_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (size_t)cap) g = cap;
    return (int)g;
}
#define DISPATCH_T(dtype, CALL_F32, CALL_BF16)        \
    do {                                              \
        if ((dtype) == GA_DTYPE_BF16) { CALL_BF16; }  \
        else if ((dtype) == GA_DTYPE_F32) { CALL_F32; } \
        else return GA_ERR_ARG;                       \
    } while (0)

extern "C" {

int ga_wn_scale(const float* v, const float* g, int rows, int n, float* scale, float* norm, hipStream_t s) {
    if (!v || !g || !scale || !norm || rows <= 0 || n <= 0) return GA_ERR_ARG;
    wn_scale_kernel<<<rows, 256, 0, s>>>(v, g, n, scale, norm);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

size_t ga_packed_elems(int mode, int D0, int D1, int K, int S) {
    switch (mode) {
        case GA_PACK_CONV_DGRAD_STRIDED: return (size_t)S * D1 * ((K + S - 1) / S) * D0;
        default: return (size_t)D0 * D1 * K;
    }
}

int ga_pack_weights(const float* src, const float* scale, void* dst, int dtype, int mode, int D0, int D1, int K, int S, int P,
                    hipStream_t s) {
    if (!src || !dst || D0 <= 0 || D1 <= 0 || K <= 0 || mode < 0 || mode > 4) return GA_ERR_ARG;
    if ((mode == GA_PACK_CONV_DGRAD_STRIDED || mode == GA_PACK_CONVT_FWD) && S < 1) return GA_ERR_ARG;
    if (mode == GA_PACK_CONVT_FWD && K != 2 * S) return GA_ERR_ARG;
    const size_t n = (size_t)D0 * D1 * K;
    const size_t esz = dtype == GA_DTYPE_BF16 ? 2 : 4;
    if (mode == GA_PACK_CONV_DGRAD_STRIDED) {
        if (hipMemsetAsync(dst, 0, ga_packed_elems(mode, D0, D1, K, S) * esz, s) != hipSuccess) return GA_ERR_LAUNCH;
    }
    const bool tiled = dtype == GA_DTYPE_BF16
        ? launch_pack_tiled<bf16_t, false>(src, scale, (bf16_t*)dst, nullptr, nullptr, mode, D0, D1, K, S, P, 0, s)
        : launch_pack_tiled<float, false>(src, scale, (float*)dst, nullptr, nullptr, mode, D0, D1, K, S, P, 0, s);
    if (!tiled) {
        DISPATCH_T(dtype, (pack_weights_kernel<float><<<grid_for(n), 256, 0, s>>>(src, scale, (float*)dst, mode, D0, D1, K, S, P)),
                   (pack_weights_kernel<bf16_t><<<grid_for(n), 256, 0, s>>>(src, scale, (bf16_t*)dst, mode, D0, D1, K, S, P)));
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_unpack_grad(const float* packed, float* dw, int mode, int D0, int D1, int K, int S, int P, int accumulate, hipStream_t s) {
    if (!packed || !dw || D0 <= 0 || D1 <= 0 || K <= 0 || mode < 0 || mode > 4) return GA_ERR_ARG;
    if (!launch_pack_tiled<float, true>(nullptr, nullptr, (float*)nullptr, dw, packed, mode, D0, D1, K, S, P, accumulate, s))
        unpack_grad_kernel<<<grid_for((size_t)D0 * D1 * K), 256, 0, s>>>(packed, dw, mode, D0, D1, K, S, P, accumulate);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_wn_bwd(const float* dw, const float* v, const float* g, const float* norm, int rows, int n, float* dv, float* dg,
              int accumulate, hipStream_t s) {
    if (!dw || !v || !g || !norm || !dv || !dg || rows <= 0 || n <= 0) return GA_ERR_ARG;
    wn_bwd_kernel<<<rows, 256, 0, s>>>(dw, v, g, norm, n, dv, dg, accumulate);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_gn_stats(const void* x, int dtype, int B, int T, int C, int G, float eps, float* mean, float* rstd, hipStream_t s) {
    if (!x || !mean || !rstd || B <= 0 || T <= 0 || C <= 0 || G <= 0 || C % G) return GA_ERR_ARG;
    const int ev = dtype == GA_DTYPE_BF16 ? 8 : 4;
    const int cg = C / G;
    const bool vec = cg % ev == 0 && cg / ev <= 256 && C % ev == 0 && ((uintptr_t)x & 15) == 0;
    if (vec) {
        DISPATCH_T(dtype, (gn_stats_vec_kernel<float><<<B * G, 256, 0, s>>>((const float*)x, T, C, G, eps, mean, rstd)),
                   (gn_stats_vec_kernel<bf16_t><<<B * G, 256, 0, s>>>((const bf16_t*)x, T, C, G, eps, mean, rstd)));
    } else {
        DISPATCH_T(dtype, (gn_stats_kernel<float><<<B * G, 256, 0, s>>>((const float*)x, T, C, G, eps, mean, rstd)),
                   (gn_stats_kernel<bf16_t><<<B * G, 256, 0, s>>>((const bf16_t*)x, T, C, G, eps, mean, rstd)));
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_gn_film_act_fused(const void* x, const float* gamma, const float* beta, const float* ab, const void* res, void* y,
                         float* mean, float* rstd, int dtype, int B, int T, int C, int G, float eps, int act, hipStream_t s) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || B <= 0 || T <= 0 || C <= 0 || G <= 0 || C % G) return GA_ERR_ARG;
    const int ev = dtype == GA_DTYPE_BF16 ? 8 : 4;
    const int cg = C / G;
    const bool vec = cg % ev == 0 && cg / ev <= 256 && C % ev == 0 &&
                     (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15) == 0;
    if (!vec) return GA_ERR_UNSUPPORTED;
    DISPATCH_T(dtype,
               (gn_stats_vec_kernel<float><<<B * G, 256, 0, s>>>((const float*)x, T, C, G, eps, mean, rstd, gamma, beta, ab, (const float*)res, (float*)y, act)),
               (gn_stats_vec_kernel<bf16_t><<<B * G, 256, 0, s>>>((const bf16_t*)x, T, C, G, eps, mean, rstd, gamma, beta, ab, (const bf16_t*)res, (bf16_t*)y, act)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_gn_film_act_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                       const float* ab, const void* res, void* y, int dtype, int B, int T, int C, int G, int act, hipStream_t s) {
    if (!x || !mean || !rstd || !gamma || !beta || !y || C % G) return GA_ERR_ARG;
    const size_t n = (size_t)B * T * C;
    DISPATCH_T(dtype,
               (gn_film_act_fwd_kernel<float><<<grid_for(n), 256, 0, s>>>((const float*)x, mean, rstd, gamma, beta, ab, (const float*)res, (float*)y, B, T, C, G, act)),
               (gn_film_act_fwd_kernel<bf16_t><<<grid_for(n), 256, 0, s>>>((const bf16_t*)x, mean, rstd, gamma, beta, ab, (const bf16_t*)res, (bf16_t*)y, B, T, C, G, act)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_gn_film_act_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                       const float* ab, void* dx, float* dgamma, float* dbeta, float* dab, int dtype, int B, int T, int C, int G,
                       int act, hipStream_t s) {
    if (!dy || !x || !mean || !rstd || !gamma || !beta || !dx || !dgamma || !dbeta || C % G) return GA_ERR_ARG;
    if (C / G > 256 || (ab && !dab)) return GA_ERR_ARG;
    const int ev = dtype == GA_DTYPE_BF16 ? 8 : 4;
    const int cg = C / G;
    const bool vec = cg % ev == 0 && C % ev == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
    if (vec) {
        DISPATCH_T(dtype,
                   (gn_film_act_bwd_vec_kernel<float><<<B * G, 256, 0, s>>>((const float*)dy, (const float*)x, mean, rstd, gamma, beta, ab, (float*)dx, dgamma, dbeta, dab, T, C, G, act)),
                   (gn_film_act_bwd_vec_kernel<bf16_t><<<B * G, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)x, mean, rstd, gamma, beta, ab, (bf16_t*)dx, dgamma, dbeta, dab, T, C, G, act)));
    } else {
        DISPATCH_T(dtype,
                   (gn_film_act_bwd_kernel<float><<<B * G, 256, 0, s>>>((const float*)dy, (const float*)x, mean, rstd, gamma, beta, ab, (float*)dx, dgamma, dbeta, dab, T, C, G, act)),
                   (gn_film_act_bwd_kernel<bf16_t><<<B * G, 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)x, mean, rstd, gamma, beta, ab, (bf16_t*)dx, dgamma, dbeta, dab, T, C, G, act)));
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_tanh_residual_fwd(const void* raw, const void* q, void* y, int dtype, size_t n, float rscale, hipStream_t s) {
    if (!raw || !q || !y) return GA_ERR_ARG;
    DISPATCH_T(dtype, (tanh_residual_fwd_kernel<float><<<grid_for(n), 256, 0, s>>>((const float*)raw, (const float*)q, (float*)y, n, rscale)),
               (tanh_residual_fwd_kernel<bf16_t><<<grid_for(n), 256, 0, s>>>((const bf16_t*)raw, (const bf16_t*)q, (bf16_t*)y, n, rscale)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}
int ga_tanh_residual_bwd(const void* dy, const void* raw, void* draw, int dtype, size_t n, float rscale, hipStream_t s) {
    if (!dy || !raw || !draw) return GA_ERR_ARG;
    DISPATCH_T(dtype, (tanh_residual_bwd_kernel<float><<<grid_for(n), 256, 0, s>>>((const float*)dy, (const float*)raw, (float*)draw, n, rscale)),
               (tanh_residual_bwd_kernel<bf16_t><<<grid_for(n), 256, 0, s>>>((const bf16_t*)dy, (const bf16_t*)raw, (bf16_t*)draw, n, rscale)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_avgpool1d_fwd(const void* x, void* y, int dtype, int B, int L, int Lo, hipStream_t s) {
    if (!x || !y || B <= 0 || L <= 0 || Lo != (L + 2 - 4) / 2 + 1) return GA_ERR_ARG;
    const size_t n = (size_t)B * Lo;
    DISPATCH_T(dtype, (avgpool_fwd_kernel<float><<<grid_for(n), 256, 0, s>>>((const float*)x, (float*)y, B, L, Lo)),
               (avgpool_fwd_kernel<bf16_t><<<grid_for(n), 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, B, L, Lo)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}
int ga_avgpool1d_bwd(const void* dy, void* dx, int dtype, int B, int L, int Lo, hipStream_t s) {
    if (!dy || !dx || B <= 0 || L <= 0 || Lo != (L + 2 - 4) / 2 + 1) return GA_ERR_ARG;
    const size_t n = (size_t)B * L;
    DISPATCH_T(dtype, (avgpool_bwd_kernel<float><<<grid_for(n), 256, 0, s>>>((const float*)dy, (float*)dx, B, L, Lo)),
               (avgpool_bwd_kernel<bf16_t><<<grid_for(n), 256, 0, s>>>((const bf16_t*)dy, (bf16_t*)dx, B, L, Lo)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_act_bwd(const void* gy, const void* src, void* ga, int dtype, size_t n, int mode, hipStream_t s) {
    if (!gy || !src || !ga) return GA_ERR_ARG;
    if (((uintptr_t)gy | (uintptr_t)src | (uintptr_t)ga) & 15) return GA_ERR_ALIGN;
    const int grid = grid_for(n / 4 + 1, 256, 4096);
    DISPATCH_T(dtype, (act_bwd_kernel<float><<<grid, 256, 0, s>>>((const float*)gy, (const float*)src, (float*)ga, n, mode)),
               (act_bwd_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)gy, (const bf16_t*)src, (bf16_t*)ga, n, mode)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_act_bwd_bias(const void* gy, const void* src, void* ga, int dtype, size_t rows, int M, int mode, float* dbias, hipStream_t s) {
    if (!gy || !src || !ga || !dbias || M <= 0) return GA_ERR_ARG;
    if (((uintptr_t)gy | (uintptr_t)src | (uintptr_t)ga) & 15) return GA_ERR_ALIGN;
    const int ev = dtype == GA_DTYPE_BF16 ? 8 : 4;
    if (M % ev || 256 % (M / ev)) return GA_ERR_ARG;
    const size_t nv = rows * (size_t)(M / ev);
    const int grid = grid_for(nv, 256, 2048);
    DISPATCH_T(dtype, (act_bwd_bias_kernel<float><<<grid, 256, 0, s>>>((const float*)gy, (const float*)src, (float*)ga, nv, mode, M, dbias)),
               (act_bwd_bias_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)gy, (const bf16_t*)src, (bf16_t*)ga, nv, mode, M, dbias)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

static int rowgather_check(const void* a, const void* b, int dtype, int B, int H, int W, int C, int KH, int S, int P, int OH, int KC) {
    if (!a || !b || (dtype != GA_DTYPE_F32 && dtype != GA_DTYPE_BF16)) return GA_ERR_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || S <= 0 || P < 0 || OH <= 0 || KC < KH * C) return GA_ERR_ARG;
    if ((long)S * (OH - 1) - P + (KH - 1) >= (long)H + P) return GA_ERR_ARG;  // the last window reaches past the padded image
    return GA_OK;
}
int ga_rowgather_fwd(const void* x, void* y, int dtype, int B, int H, int W, int C, int KH, int S, int P, int OH, int KC, hipStream_t s) {
    const int rc = rowgather_check(x, y, dtype, B, H, W, C, KH, S, P, OH, KC);
    if (rc) return rc;
    const int ev = dtype == GA_DTYPE_BF16 ? 8 : 4;
    const bool vec = C % ev == 0 && KC % ev == 0 && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
    const size_t work = (size_t)B * OH * W * (vec ? KC / ev : KC);
    const int grid = grid_for(work, 256, 8192);
    if (vec) {
        DISPATCH_T(dtype, (rowgather_fwd_kernel<float, true><<<grid, 256, 0, s>>>((const float*)x, (float*)y, B, H, W, C, KH, S, P, OH, KC)),
                   (rowgather_fwd_kernel<bf16_t, true><<<grid, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, B, H, W, C, KH, S, P, OH, KC)));
    } else {
        DISPATCH_T(dtype, (rowgather_fwd_kernel<float, false><<<grid, 256, 0, s>>>((const float*)x, (float*)y, B, H, W, C, KH, S, P, OH, KC)),
                   (rowgather_fwd_kernel<bf16_t, false><<<grid, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, B, H, W, C, KH, S, P, OH, KC)));
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}
int ga_rowgather_bwd(const void* gy, void* gx, int dtype, int B, int H, int W, int C, int KH, int S, int P, int OH, int KC, hipStream_t s) {
    const int rc = rowgather_check(gy, gx, dtype, B, H, W, C, KH, S, P, OH, KC);
    if (rc) return rc;
    const int ev = dtype == GA_DTYPE_BF16 ? 8 : 4;
    const bool vec = C % ev == 0 && KC % ev == 0 && ((((uintptr_t)gy | (uintptr_t)gx) & 15) == 0);
    const size_t work = (size_t)B * H * W * (vec ? C / ev : C);
    const int grid = grid_for(work, 256, 8192);
    if (vec) {
        DISPATCH_T(dtype, (rowgather_bwd_kernel<float, true><<<grid, 256, 0, s>>>((const float*)gy, (float*)gx, B, H, W, C, KH, S, P, OH, KC)),
                   (rowgather_bwd_kernel<bf16_t, true><<<grid, 256, 0, s>>>((const bf16_t*)gy, (bf16_t*)gx, B, H, W, C, KH, S, P, OH, KC)));
    } else {
        DISPATCH_T(dtype, (rowgather_bwd_kernel<float, false><<<grid, 256, 0, s>>>((const float*)gy, (float*)gx, B, H, W, C, KH, S, P, OH, KC)),
                   (rowgather_bwd_kernel<bf16_t, false><<<grid, 256, 0, s>>>((const bf16_t*)gy, (bf16_t*)gx, B, H, W, C, KH, S, P, OH, KC)));
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_act_fwd(const void* x, void* y, int dtype, size_t n, int act, hipStream_t s) {
    if (!x || !y) return GA_ERR_ARG;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return GA_ERR_ALIGN;
    const int grid = grid_for(n / 4 + 1, 256, 4096);
    DISPATCH_T(dtype, (act_fwd_kernel<float><<<grid, 256, 0, s>>>((const float*)x, (float*)y, n, act)),
               (act_fwd_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, n, act)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_colsum(const void* g, int dtype, size_t rows, int M, float* out, hipStream_t s) {
    if (!g || !out || M <= 0) return GA_ERR_ARG;
    if (rows == 0) return GA_OK;
    if (dtype != GA_DTYPE_F32 && dtype != GA_DTYPE_BF16) return GA_ERR_ARG;
    if (ga_colsum_vec_supported(g, dtype, M)) return ga_colsum_vec_impl(g, dtype, rows, M, out, s);
    size_t blocks = (rows + 511) / 512;
    if (blocks > 1024) blocks = 1024;
    const size_t rpb = (rows + blocks - 1) / blocks;
    blocks = (rows + rpb - 1) / rpb;
    DISPATCH_T(dtype, (colsum_kernel<float><<<(int)blocks, 256, 0, s>>>((const float*)g, rows, M, out, rpb)),
               (colsum_kernel<bf16_t><<<(int)blocks, 256, 0, s>>>((const bf16_t*)g, rows, M, out, rpb)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_reduce(const void* a, const void* b, int dtype, size_t n, int op, float scale, float* out, hipStream_t s) {
    if (!a || !out || op < 0 || op > 4 || (op == GA_RED_ABSDIFF && !b)) return GA_ERR_ARG;
    if (n == 0) return GA_OK;
    const int vec = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
    const size_t work = vec ? n / (dtype == GA_DTYPE_BF16 ? 8 : 4) + 1 : n;
    DISPATCH_T(dtype, (reduce_kernel<float><<<grid_for(work, 256, 1024), 256, 0, s>>>((const float*)a, (const float*)b, n, op, scale, out, vec)),
               (reduce_kernel<bf16_t><<<grid_for(work, 256, 1024), 256, 0, s>>>((const bf16_t*)a, (const bf16_t*)b, n, op, scale, out, vec)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}
int ga_reduce_multi(const void* const* a, const void* const* b, const size_t* n, const float* scale, int count, int dtype, int op, float* out,
                    hipStream_t s) {
    if (!a || !n || !scale || !out || count < 1 || count > GA_REDUCE_MULTI_MAX || op < 0 || op > 4) return GA_ERR_ARG;
    if (dtype != GA_DTYPE_F32 && dtype != GA_DTYPE_BF16) return GA_ERR_ARG;
    ReduceMultiArgs A;
    const int ev = dtype == GA_DTYPE_BF16 ? 8 : 4;
    double total = 0.0;
    for (int i = 0; i < count; ++i) {
        if (n[i] && (!a[i] || (op == GA_RED_ABSDIFF && (!b || !b[i])))) return GA_ERR_ARG;  // (an empty entry may have null pointers)
        total += (double)n[i];
    }
    if (total == 0.0) return GA_OK;
    const int cap = 1536;  // blocks in all: the large maps need ~1k blocks for bandwidth, every block costs one same-address atomic
    int at = 0;
    for (int i = 0; i < count; ++i) {
        A.a[i] = a[i];
        A.b[i] = b ? b[i] : nullptr;
        A.n[i] = n[i];
        A.scale[i] = scale[i];
        A.vec[i] = (((uintptr_t)A.a[i] | (uintptr_t)A.b[i]) & 15) == 0;
        const size_t work = A.vec[i] ? n[i] / ev + 1 : n[i];
        long nb = (long)(cap * ((double)n[i] / total) + 0.5);
        const long most = (long)((work + 255) / 256);
        if (nb > most) nb = most;
        if (nb > 1024) nb = 1024;
        if (nb < 1) nb = 1;
        A.start[i] = at;
        at += (int)nb;
    }
    A.start[count] = at;
    A.cnt = count;
    A.op = op;
    DISPATCH_T(dtype, (reduce_multi_kernel<float><<<at, 256, 0, s>>>(A, out)), (reduce_multi_kernel<bf16_t><<<at, 256, 0, s>>>(A, out)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}
int ga_reduce_grad(const void* a, const void* b, int dtype, size_t n, int op, const float* coef, float scale, void* ga,
                   int accumulate, hipStream_t s) {
    if (!a || !coef || !ga || op < 0 || op > 4 || (op == GA_RED_ABSDIFF && !b)) return GA_ERR_ARG;
    if (n == 0) return GA_OK;
    const int vec = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)ga) & 15) == 0;
    const size_t work = vec ? n / (dtype == GA_DTYPE_BF16 ? 8 : 4) + 1 : n;
    DISPATCH_T(dtype, (reduce_grad_kernel<float><<<grid_for(work), 256, 0, s>>>((const float*)a, (const float*)b, n, op, coef, scale, (float*)ga, accumulate, vec)),
               (reduce_grad_kernel<bf16_t><<<grid_for(work), 256, 0, s>>>((const bf16_t*)a, (const bf16_t*)b, n, op, coef, scale, (bf16_t*)ga, accumulate, vec)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_latent_loss_fwd(const void* qh, const void* qs, int dtype, int B, int T, int C, float* out, hipStream_t s) {
    if (!qh || !qs || !out || B <= 0 || B > 65535 || T <= 0 || C <= 0) return GA_ERR_ARG;
    const int gx = grid_for((size_t)T * C, 256, 64);
    DISPATCH_T(dtype, (latent_loss_fwd_kernel<float><<<dim3(gx, B), 256, 0, s>>>((const float*)qh, (const float*)qs, T, C, out)),
               (latent_loss_fwd_kernel<bf16_t><<<dim3(gx, B), 256, 0, s>>>((const bf16_t*)qh, (const bf16_t*)qs, T, C, out)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}
int ga_latent_loss_bwd(const void* qh, const void* qs, int dtype, int B, int T, int C, const float* gout, void* dqh, hipStream_t s) {
    if (!qh || !qs || !gout || !dqh || B <= 0 || T <= 0 || C <= 0) return GA_ERR_ARG;
    const size_t n = (size_t)B * T * C;
    DISPATCH_T(dtype, (latent_loss_bwd_kernel<float><<<grid_for(n), 256, 0, s>>>((const float*)qh, (const float*)qs, B, T, C, gout, (float*)dqh)),
               (latent_loss_bwd_kernel<bf16_t><<<grid_for(n), 256, 0, s>>>((const bf16_t*)qh, (const bf16_t*)qs, B, T, C, gout, (bf16_t*)dqh)));
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_grad_sqnorm(const float* g, size_t n, float* out, hipStream_t s) {
    if (!g || !out || ((uintptr_t)g & 15)) return g && out ? GA_ERR_ALIGN : GA_ERR_ARG;
    if (n == 0) return GA_OK;
    sqnorm_kernel<<<grid_for(n / 4 + 1, 256, 1024), 256, 0, s>>>(g, n, out);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_adamw_fused(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float max_norm, float grad_scale, const float* sqnorm,
                   int write_clipped_grad, hipStream_t s) {
    if (!p || !g || !m || !v || step < 1) return GA_ERR_ARG;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return GA_ERR_ALIGN;
    if (n == 0) return GA_OK;
    AdamWArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, step));
    a.bc2 = (float)(1.0 - pow((double)beta2, step));
    a.max_norm = max_norm; a.grad_scale = grad_scale;
    adamw_kernel<<<grid_for(n / 4 + 1, 256, 2048), 256, 0, s>>>(p, g, m, v, n, a, sqnorm, write_clipped_grad, nullptr);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_adamw_fused_devstep(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, const int* step_dev, float max_norm, float grad_scale, const float* sqnorm,
                           int write_clipped_grad, hipStream_t s) {
    if (!p || !g || !m || !v || !step_dev) return GA_ERR_ARG;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return GA_ERR_ALIGN;
    if (n == 0) return GA_OK;
    AdamWArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.bc1 = a.bc2 = 1.f;  // from step_dev[0], in the kernel
    a.max_norm = max_norm; a.grad_scale = grad_scale;
    adamw_kernel<<<grid_for(n / 4 + 1, 256, 2048), 256, 0, s>>>(p, g, m, v, n, a, sqnorm, write_clipped_grad, step_dev);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

}  // extern "C"
