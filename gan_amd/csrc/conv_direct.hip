// Direct (vector-ALU, fp32 FMA chain) kernels for the generalised NLC convolution.
// They are the exact-fp32 path ("fp32 mode", parity at rtol 1e-4 against the CPU reference) and the
// path for thin layers (Cin == 1 or M == 1) whose GEMM view would leave MFMA tiles empty; those are
// HBM-bound and get dedicated row kernels below.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// generic tile kernel: 256 threads, 64 output steps x 64 output channels, 16-deep K chunks
template <typename T>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvParams p) {
    __shared__ float xs[16][65];
    __shared__ float ws[16][65];
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ w = (const T*)p.w;
    const int phase = blockIdx.z % p.nphase;
    const int b = blockIdx.z / p.nphase;
    const int tbase = p.t0 + blockIdx.x * 64;
    const int mbase = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    const int off0 = p.off0 + p.phase_off0[phase];
    w += (size_t)phase * p.M * p.J * p.Cin;
    x += (size_t)b * p.Lin * p.Cin;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;

    const int lr = tid >> 2;        // 0..63: tile row (t for x, m for w)
    const int lc = (tid & 3) * 4;   // first of 4 k-elements this thread stages
    const int t_l = tbase + lr;
    const bool t_ok = t_l < p.t0 + p.Lout;

    for (int j = 0; j < p.J; ++j) {
        int row = t_l * p.in_stride + j * p.tap_step + off0;
        if (row < 0 && p.pad_mode == 1) row = -row;
        const bool row_ok = t_ok && row >= 0 && row < p.Lin;
        for (int c0 = 0; c0 < p.Cin; c0 += 16) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + lc + i;
                float xv = 0.f, wv = 0.f;
                if (row_ok && c < p.Cin) {
                    xv = (float)x[(size_t)row * p.Cin + c];
                    if (p.pre_act) xv = act_apply(p.pre_act, xv);
                }
                const int m = mbase + lr;
                if (m < p.M && c < p.Cin) wv = (float)w[((size_t)m * p.J + j) * p.Cin + c];
                xs[lc + i][lr] = xv;
                ws[lc + i][lr] = wv;
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float a[4], bb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = xs[c][ty * 4 + i];
#pragma unroll
                for (int k = 0; k < 4; ++k) bb[k] = ws[c][tx * 4 + k];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[i][k] = fmaf(a[i], bb[k], acc[i][k]);
            }
        }
    }

    T* __restrict__ y = (T*)p.y;
    const T* __restrict__ res = (const T*)p.res;
    const T* __restrict__ ds = (const T*)p.dact_src;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = tbase + ty * 4 + i;
        if (t >= p.t0 + p.Lout) continue;
        const int orow = t * p.out_stride + phase;
        if (orow < 0 || orow >= p.LoutTot) continue;
        const size_t base = ((size_t)b * p.LoutTot + orow) * p.M;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = mbase + tx * 4 + k;
            if (m >= p.M) continue;
            float v = acc[i][k];
            if (p.bias) v += p.bias[m];
            v = act_apply(p.post_act, v);
            if (p.dact_mode) v *= dact_apply(p.dact_mode, (float)ds[base + m]);
            if (res) v += (float)res[base + m];
            if (p.accumulate) v += (float)y[base + m];
            y[base + m] = (T)v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// thin kernel A: few output channels (M <= 4, e.g. D.out 512->1 k3, decoder tail 32->1 k7, and the
// data-gradient of a Cin==1 layer).  One wave per output step: lanes split the (j, c) reduction.
template <typename T>
__global__ __launch_bounds__(256) void conv_fewout_kernel(ConvParams p) {
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ w = (const T*)p.w;
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int phase = blockIdx.z % p.nphase;
    const int b = blockIdx.z / p.nphase;
    const int off0 = p.off0 + p.phase_off0[phase];
    w += (size_t)phase * p.M * p.J * p.Cin;
    x += (size_t)b * p.Lin * p.Cin;
    const int KC = p.J * p.Cin;
    for (int t = p.t0 + blockIdx.x * 4 + wid; t < p.t0 + p.Lout; t += gridDim.x * 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int kc = lane; kc < KC; kc += 64) {
            const int j = kc / p.Cin, c = kc - j * p.Cin;
            int row = t * p.in_stride + j * p.tap_step + off0;
            if (row < 0 && p.pad_mode == 1) row = -row;
            if (row < 0 || row >= p.Lin) continue;
            float xv = (float)x[(size_t)row * p.Cin + c];
            if (p.pre_act) xv = act_apply(p.pre_act, xv);
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (m < p.M) acc[m] = fmaf(xv, (float)w[(size_t)m * KC + kc], acc[m]);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = wave_sum(acc[m]);
        const int orow = t * p.out_stride + phase;
        if (lane < p.M && orow >= 0 && orow < p.LoutTot) {
            const int m = lane;
            const size_t idx = ((size_t)b * p.LoutTot + orow) * p.M + m;
            float v = acc[0];
            if (m == 1) v = acc[1];
            if (m == 2) v = acc[2];
            if (m == 3) v = acc[3];
            if (p.bias) v += p.bias[m];
            v = act_apply(p.post_act, v);
            if (p.dact_mode) v *= dact_apply(p.dact_mode, (float)((const T*)p.dact_src)[idx]);
            if (p.res) v += (float)((const T*)p.res)[idx];
            if (p.accumulate) v += (float)((T*)p.y)[idx];
            ((T*)p.y)[idx] = (T)v;
        }
    }
}

// thin kernel A', M == 1 and Cin % 8 == 0 with at most 256 16-byte pieces per output (D.out: 512 -> 1, k3 = 192 pieces): one wave
// per output step like kernel A, but a lane owns fixed pieces (tap j, 8 consecutive channels), keeps their weights in
// registers for every step it visits and fetches each with ONE 16-byte load -- no per-element division, no scalar loads.
template <typename T> __device__ __forceinline__ void load8(const T* q, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* q, float (&v)[8]) {
    const float4 a = *(const float4*)q, b = *(const float4*)(q + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* q, float (&v)[8]) {
    typedef __attribute__((ext_vector_type(8))) __bf16 raw_t;
    const raw_t r = *(const raw_t*)q;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
}

template <typename T>
__global__ __launch_bounds__(256) void conv_m1_wave_kernel(ConvParams p) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int phase = blockIdx.z % p.nphase, b = blockIdx.z / p.nphase;
    const int off0 = p.off0 + p.phase_off0[phase];
    const T* __restrict__ w = (const T*)p.w + (size_t)phase * p.J * p.Cin;
    const T* __restrict__ x = (const T*)p.x + (size_t)b * p.Lin * p.Cin;
    const int parts = p.Cin / 8, KP = p.J * parts;
    float wr[4][8];
    int dr[4], dc[4];
    bool on[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = lane + 64 * q;
        on[q] = piece < KP;
        const int j = on[q] ? piece / parts : 0, part = on[q] ? piece - j * parts : 0;
        dr[q] = j * p.tap_step;
        dc[q] = part * 8;
        load8<T>(w + (size_t)j * p.Cin + part * 8, wr[q]);
    }
    for (int t = p.t0 + blockIdx.x * 4 + wid; t < p.t0 + p.Lout; t += gridDim.x * 4) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int row = t * p.in_stride + dr[q] + off0;
            if (row < 0 && p.pad_mode == 1) row = -row;
            if (!on[q] || row < 0 || row >= p.Lin) continue;
            float xv[8];
            load8<T>(x + (size_t)row * p.Cin + dc[q], xv);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(p.pre_act ? act_apply(p.pre_act, xv[e]) : xv[e], wr[q][e], acc);
        }
        acc = wave_sum(acc);
        const int orow = t * p.out_stride + phase;
        if (lane == 0 && orow >= 0 && orow < p.LoutTot) {
            const size_t idx = (size_t)b * p.LoutTot + orow;
            float v = acc;
            if (p.bias) v += p.bias[0];
            v = act_apply(p.post_act, v);
            if (p.dact_mode) v *= dact_apply(p.dact_mode, (float)((const T*)p.dact_src)[idx]);
            if (p.res) v += (float)((const T*)p.res)[idx];
            if (p.accumulate) v += (float)((T*)p.y)[idx];
            ((T*)p.y)[idx] = (T)v;
        }
    }
}

// thin kernel B: one input channel (D block 0: 1 -> 32, k15).  One thread per (t, m); the J inputs
// of a step are contiguous and shared through L1, weights sit in LDS.  HBM-bound on the output.
template <typename T>
__global__ __launch_bounds__(256) void conv_cin1_kernel(ConvParams p) {
    extern __shared__ float wsm[];  // [M][J] + [M] bias
    const T* __restrict__ x = (const T*)p.x;
    const T* __restrict__ w = (const T*)p.w;
    const int b = blockIdx.z;
    for (int i = threadIdx.x; i < p.M * p.J; i += blockDim.x) wsm[i] = (float)w[i];
    for (int i = threadIdx.x; i < p.M; i += blockDim.x) wsm[p.M * p.J + i] = p.bias ? p.bias[i] : 0.f;
    __syncthreads();
    x += (size_t)b * p.Lin;
    const size_t total = (size_t)p.Lout * p.M;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int t = p.t0 + (int)(e / p.M), m = (int)(e % p.M);
        float acc = 0.f;
        for (int j = 0; j < p.J; ++j) {
            int row = t * p.in_stride + j * p.tap_step + p.off0;
            if (row < 0 && p.pad_mode == 1) row = -row;
            if (row >= 0 && row < p.Lin) {
                float xv = (float)x[row];
                if (p.pre_act) xv = act_apply(p.pre_act, xv);
                acc = fmaf(xv, wsm[m * p.J + j], acc);
            }
        }
        const int orow = t * p.out_stride;
        if (orow >= p.LoutTot) continue;
        const size_t idx = ((size_t)b * p.LoutTot + orow) * p.M + m;
        float v = act_apply(p.post_act, acc + wsm[p.M * p.J + m]);
        if (p.dact_mode) v *= dact_apply(p.dact_mode, (float)((const T*)p.dact_src)[idx]);
        if (p.res) v += (float)((const T*)p.res)[idx];
        if (p.accumulate) v += (float)((T*)p.y)[idx];
        ((T*)p.y)[idx] = (T)v;
    }
}

// ------------------------------------------------------------------------------------------------
// generic weight gradient: 64 (m) x 64 (c) tile for one tap j, reduction over flattened (b,t) rows in
// chunks of 16, split over gridDim.z; fp32 atomics into dW (placement-independent, caller zeroes).
template <typename T>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(WgradParams p) {
    __shared__ float gs[16][65];
    __shared__ float xs[16][65];
    const T* __restrict__ g = (const T*)p.g;
    const T* __restrict__ x = (const T*)p.x;
    const int ctiles = (p.Cin + 63) / 64;
    const int mbase = (blockIdx.x / ctiles) * 64;
    const int cbase = (blockIdx.x % ctiles) * 64;
    const int j = blockIdx.y;
    const long rows_total = (long)p.B * p.Lout;
    const long r_begin = (long)blockIdx.z * p.rows_per_split;
    long r_end = r_begin + p.rows_per_split;
    if (r_end > rows_total) r_end = rows_total;
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    const int lr = tid >> 4;        // 0..15 row within chunk
    const int lc = (tid & 15) * 4;  // 4 consecutive channels

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = 0.f;

    for (long r0 = r_begin; r0 < r_end; r0 += 16) {
        const long r = r0 + lr;
        __syncthreads();
        if (r < r_end) {
            const int b = (int)(r / p.Lout), t = (int)(r % p.Lout);
            const int row = t * p.in_stride + j * p.tap_step + p.off0;
            const bool row_ok = row >= 0 && row < p.Lin;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mbase + lc + i, c = cbase + lc + i;
                gs[lr][lc + i] = (m < p.M) ? (float)g[((size_t)b * p.LgTot + (size_t)t * p.g_stride + p.g_phase) * p.M + m] : 0.f;
                xs[lr][lc + i] = (row_ok && c < p.Cin) ? (float)x[((size_t)b * p.Lin + row) * p.Cin + c] : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { gs[lr][lc + i] = 0.f; xs[lr][lc + i] = 0.f; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], bb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = gs[k][ty * 4 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) bb[i] = xs[k][tx * 4 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][q] = fmaf(a[i], bb[q], acc[i][q]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mbase + ty * 4 + i;
        if (m >= p.M) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = cbase + tx * 4 + q;
            if (c >= p.Cin) continue;
            atomicAdd(&p.dw[((size_t)m * p.J + j) * p.Cin + c], acc[i][q]);
        }
    }
}

// thin weight gradient (M <= 4 or Cin == 1): one block per (m, j, c-chunk of 64) slice reduces over rows.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(WgradParams p) {
    __shared__ float red[16];
    const T* __restrict__ g = (const T*)p.g;
    const T* __restrict__ x = (const T*)p.x;
    const int m = blockIdx.x, j = blockIdx.y;
    const long rows_total = (long)p.B * p.Lout;
    const long r_begin = (long)blockIdx.z * p.rows_per_split;
    long r_end = r_begin + p.rows_per_split;
    if (r_end > rows_total) r_end = rows_total;
    if (p.Cin <= 4) {
        // lanes over rows, loop channels
        for (int c = 0; c < p.Cin; ++c) {
            float acc = 0.f;
            for (long r = r_begin + threadIdx.x; r < r_end; r += blockDim.x) {
                const int b = (int)(r / p.Lout), t = (int)(r % p.Lout);
                const int row = t * p.in_stride + j * p.tap_step + p.off0;
                if (row >= 0 && row < p.Lin)
                    acc = fmaf((float)g[((size_t)b * p.LgTot + (size_t)t * p.g_stride + p.g_phase) * p.M + m], (float)x[((size_t)b * p.Lin + row) * p.Cin + c], acc);
            }
            acc = block_sum(acc, red);
            if (threadIdx.x == 0) atomicAdd(&p.dw[((size_t)m * p.J + j) * p.Cin + c], acc);
        }
    } else {
        // lanes over channels (coalesced rows of x), loop rows
        for (int c = threadIdx.x; c < p.Cin; c += blockDim.x) {
            float acc = 0.f;
            for (long r = r_begin; r < r_end; ++r) {
                const int b = (int)(r / p.Lout), t = (int)(r % p.Lout);
                const int row = t * p.in_stride + j * p.tap_step + p.off0;
                if (row >= 0 && row < p.Lin)
                    acc = fmaf((float)g[((size_t)b * p.LgTot + (size_t)t * p.g_stride + p.g_phase) * p.M + m], (float)x[((size_t)b * p.Lin + row) * p.Cin + c], acc);
            }
            atomicAdd(&p.dw[((size_t)m * p.J + j) * p.Cin + c], acc);
        }
    }
}

// shapes of conv_m1_wave_kernel (also consulted by the thin-kernel dispatch, which hands these over)
int ga_conv_m1_wave_ok(const ConvParams& p, int esz) {
    return p.M == 1 && p.Cin % 8 == 0 && p.Cin >= 128 && (long)p.J * (p.Cin / 8) <= 256 && !p.out_f32 &&
           ((uintptr_t)p.x & 15) == 0 && ((uintptr_t)p.w & 15) == 0 && ((size_t)p.Cin * esz) % 16 == 0;
}

// ------------------------------------------------------------------------------------------------
static int check_conv(const ConvParams& p) {
    if (!p.x || !p.w || !p.y) return GA_ERR_ARG;
    if (p.B <= 0 || p.Lin <= 0 || p.Cin <= 0 || p.M <= 0 || p.Lout < 0 || p.LoutTot <= 0 || p.J <= 0) return GA_ERR_ARG;
    if (p.nphase < 1 || p.nphase > GA_MAX_PHASE || p.out_stride < 1) return GA_ERR_ARG;
    if (p.dact_mode && !p.dact_src) return GA_ERR_ARG;
    if ((long)p.B * p.nphase > 65535) return GA_ERR_ARG;
    return GA_OK;
}

template <typename T>
static int launch_conv_direct(const ConvParams& p, hipStream_t s) {
    if (p.Lout == 0) return GA_OK;
    if (p.M <= 4) {
        int gx = (p.Lout + 3) / 4;
        if (gx > 8192) gx = 8192;
        if (ga_conv_m1_wave_ok(p, (int)sizeof(T)))
            conv_m1_wave_kernel<T><<<dim3(gx, 1, p.B * p.nphase), 256, 0, s>>>(p);
        else
            conv_fewout_kernel<T><<<dim3(gx, 1, p.B * p.nphase), 256, 0, s>>>(p);
    } else if (p.Cin == 1 && p.nphase == 1 && (size_t)(p.M * p.J + p.M) * 4 <= 48 * 1024) {
        size_t total = (size_t)p.Lout * p.M;
        int gx = (int)((total + 255) / 256);
        if (gx > 4096) gx = 4096;
        conv_cin1_kernel<T><<<dim3(gx, 1, p.B), 256, (p.M * p.J + p.M) * sizeof(float), s>>>(p);
    } else {
        conv_direct_kernel<T><<<dim3((p.Lout + 63) / 64, (p.M + 63) / 64, p.B * p.nphase), 256, 0, s>>>(p);
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_conv_direct_impl(const ConvParams& p, int dtype, hipStream_t s) {
    int rc = check_conv(p);
    if (rc) return rc;
    return dtype == GA_DTYPE_BF16 ? launch_conv_direct<bf16_t>(p, s) : launch_conv_direct<float>(p, s);
}

template <typename T>
static int launch_wgrad_direct(WgradParams p, hipStream_t s) {
    const long rows = (long)p.B * p.Lout;
    if (rows == 0) return GA_OK;
    if (p.M <= 4 || p.Cin <= 4) {
        int splits = (int)((rows + 8191) / 8192);
        if (splits > 1024) splits = 1024;
        p.rows_per_split = (int)((rows + splits - 1) / splits);
        wgrad_thin_kernel<T><<<dim3(p.M, p.J, splits), 256, 0, s>>>(p);
    } else {
        const int tiles = ((p.M + 63) / 64) * ((p.Cin + 63) / 64);
        long want = 2048 / ((long)tiles * p.J) + 1;  // aim for ~2k blocks
        long maxs = (rows + 255) / 256;
        if (want > maxs) want = maxs;
        if (want < 1) want = 1;
        if (want > 4096) want = 4096;
        int rps = (int)((rows + want - 1) / want);
        rps = (rps + 15) / 16 * 16;
        p.rows_per_split = rps;
        const int splits = (int)((rows + rps - 1) / rps);
        wgrad_direct_kernel<T><<<dim3(tiles, p.J, splits), 256, 0, s>>>(p);
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_wgrad_direct_impl(const WgradParams& p, int dtype, hipStream_t s) {
    if (p.g_stride < 1 || p.LgTot < 1) return GA_ERR_ARG;
    if (!p.g || !p.x || !p.dw || p.B <= 0 || p.Lin <= 0 || p.Cin <= 0 || p.M <= 0 || p.Lout < 0 || p.J <= 0) return GA_ERR_ARG;
    if (p.J > 65535) return GA_ERR_ARG;
    return dtype == GA_DTYPE_BF16 ? launch_wgrad_direct<bf16_t>(p, s) : launch_wgrad_direct<float>(p, s);
}
