// bf16 MFMA implicit-GEMM kernel for the generalised NLC convolution (forward form), gfx950.
//
//   GEMM view   M = output channels, N = output steps of one batch item, K = (tap j, input channel c)
//   A = packed weights  Wp[m][j][c]      (K-contiguous, streamed per K-step, double-buffered in LDS)
//   B = input window    x[row(t,j)][c]   (staged ONCE per 32-channel chunk, reused by all J taps)
//   v_mfma_f32_32x32x16_bf16, fp32 accumulate; wave tile 64 (m) x WTN (t); epilogue fuses bias,
//   LeakyReLU / ELU', residual add and the transposed-conv output phase.
//
// LDS image of the window: rows are split by (row - row0) mod in_stride so that, for one tap, the
// lanes of a wave read CONSECUTIVE rows of one sub-image; the 64-byte rows are padded to 80 bytes, an
// odd number of 16-byte slots, so 16 consecutive rows hit 16 different slots of the 256-byte bank row
// (conflict-free ds_read_b128; cdna_hip_programming.md section 2 / Guideline 4).
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CC 32        // input channels per K-step
#define PITCH 80     // bytes per LDS row: 64 data + 16 pad

__device__ __forceinline__ bf16x8 apply_pre(bf16x8 v, int pre_act) {
    if (pre_act == GA_ACT_NONE) return v;
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (__bf16)act_apply(pre_act, (float)v[i]);
    return o;
}

template <int WM, int WN, int WTN>
__global__ __launch_bounds__(WM * WN * 64) void conv_mfma_kernel(ConvParams p, int nx, int ny, int total, int rpp) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * 64;
    constexpr int BN = WN * WTN;
    constexpr int NI = WTN / 32;
    constexpr int PW = (BM * 4 + NT - 1) / NT;  // 16-byte weight pieces per thread per K-step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Wl = smem;                       // [2][BM][PITCH]
    char* Xl = smem + 2 * BM * PITCH;      // [in_stride][rpp][PITCH]

    // XCD-aware work mapping: blocks b and b+8 share an XCD (and its L2); give every XCD a contiguous
    // run of the (m-tile major) work list so that its blocks stream the same weight slice.
    const int id = blockIdx.x;
    const int chunk = (total + 7) / 8;
    const int wk = (id & 7) * chunk + (id >> 3);
    if (wk >= total) return;
    const int my = wk / (nx * (p.B * p.nphase));
    const int rem = wk % (nx * (p.B * p.nphase));
    const int bz = rem / nx;
    const int bx = rem % nx;
    const int phase = bz % p.nphase;
    const int b = bz / p.nphase;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;
    const int tbase = p.t0 + bx * BN;
    const int mbase = my * BM;
    const int S = p.in_stride;
    const int off0 = p.off0 + p.phase_off0[phase];
    const int row0 = tbase * S + off0;
    const int span = (BN - 1) * S + (p.J - 1) * p.tap_step + 1;
    const bf16_t* __restrict__ x = (const bf16_t*)p.x + (size_t)b * p.Lin * p.Cin;
    const bf16_t* __restrict__ w = (const bf16_t*)p.w + (size_t)phase * p.M * p.J * p.Cin;
    const bool wave_active = (mbase + wm * 64) < p.M;

    f32x16 acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    const int nchunks = (p.Cin + CC - 1) / CC;
    const int nsteps = nchunks * p.J;
    bf16x8 wreg[PW];

    auto load_w = [&](int step) {
        const int ch = step / p.J, j = step - ch * p.J;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int q = tid + i * NT;
            const int m = q >> 2, part = q & 3;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
            if (q < BM * 4 && mbase + m < p.M && ch * CC + part * 8 < p.Cin)
                v = *(const bf16x8*)(w + ((size_t)(mbase + m) * p.J + j) * p.Cin + ch * CC + part * 8);
            wreg[i] = v;
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int q = tid + i * NT;
            if (q < BM * 4) *(bf16x8*)(Wl + buf * BM * PITCH + (q >> 2) * PITCH + (q & 3) * 16) = wreg[i];
        }
    };
    auto stage_x = [&](int ch) {
        for (int q = tid; q < span * 4; q += NT) {
            const int u = q >> 2, part = q & 3;
            int R = row0 + u;
            if (R < 0 && p.pad_mode == 1) R = -R;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
            if (R >= 0 && R < p.Lin && ch * CC + part * 8 < p.Cin) v = apply_pre(*(const bf16x8*)(x + (size_t)R * p.Cin + ch * CC + part * 8), p.pre_act);
            const int ph = u % S, idx = u / S;
            *(bf16x8*)(Xl + (ph * rpp + idx) * PITCH + part * 16) = v;
        }
    };

    load_w(0);
    for (int step = 0; step < nsteps; ++step) {
        const int ch = step / p.J, j = step - ch * p.J;
        if (j == 0) {
            if (step > 0) __syncthreads();  // every wave is done reading the previous window
            stage_x(ch);
        }
        store_w(step & 1);
        __syncthreads();
        if (step + 1 < nsteps) load_w(step + 1);
        if (wave_active) {
            const int uj = j * p.tap_step;
            const int ph = uj % S, ibase = uj / S;
            const char* Wb = Wl + (step & 1) * BM * PITCH + (wm * 64 + r) * PITCH + h * 16;
            const char* Xb = Xl + (ph * rpp + ibase + wn * WTN + r) * PITCH + h * 16;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 a[2], bb[NI];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = *(const bf16x8*)(Wb + mi * 32 * PITCH + kk * 32);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bb[ni] = *(const bf16x8*)(Xb + ni * 32 * PITCH + kk * 32);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], bb[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    }

    if (!wave_active) return;
    bf16_t* __restrict__ y = (bf16_t*)p.y;
    const bf16_t* __restrict__ res = (const bf16_t*)p.res;
    const bf16_t* __restrict__ ds = (const bf16_t*)p.dact_src;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int t = tbase + wn * WTN + ni * 32 + r;
        if (t >= p.t0 + p.Lout) continue;
        const int orow = t * p.out_stride + phase;
        if (orow < 0 || orow >= p.LoutTot) continue;
        const size_t rowbase = ((size_t)b * p.LoutTot + orow) * p.M;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m0 = mbase + wm * 64 + mi * 32 + 8 * q + 4 * h;
                if (m0 >= p.M) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[mi][ni][4 * q + i];
                if (p.bias) {
                    const float4 bv = *(const float4*)(p.bias + m0);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = act_apply(p.post_act, v[i]);
                if (p.dact_mode) {
                    const bf16x4 d = *(const bf16x4*)(ds + rowbase + m0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] *= dact_apply(p.dact_mode, (float)d[i]);
                }
                if (res) {
                    const bf16x4 d = *(const bf16x4*)(res + rowbase + m0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)d[i];
                }
                if (p.accumulate) {
                    const bf16x4 d = *(const bf16x4*)(y + rowbase + m0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)d[i];
                }
                if (p.out_f32) {
                    *(float4*)((float*)p.y + rowbase + m0) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
                    *(bf16x4*)(y + rowbase + m0) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
int ga_conv_mfma_supported(const ConvParams& p) {
    if (p.Cin % 8 || p.Cin < 16 || p.M % 4 || p.M < 16 || p.in_stride < 1 || p.tap_step < 0) return 0;
    if (p.out_f32 && (p.res || p.dact_src || p.accumulate)) return 0;
    if (((uintptr_t)p.x | (uintptr_t)p.w | (uintptr_t)p.y | (uintptr_t)p.res | (uintptr_t)p.dact_src | (uintptr_t)p.bias) & 15) return 0;
    const int span = 127 * p.in_stride + (p.J - 1) * p.tap_step + 1;
    const int rpp = (span + p.in_stride - 1) / p.in_stride + 1;
    const size_t lds = 2 * 128 * PITCH + (size_t)p.in_stride * rpp * PITCH;
    if (lds > 150 * 1024) return 0;
    return 1;
}

template <int WM, int WN, int WTN>
static int launch_mfma(const ConvParams& p, hipStream_t s) {
    constexpr int BM = WM * 64, BN = WN * WTN;
    const int nx = (p.Lout + BN - 1) / BN, ny = (p.M + BM - 1) / BM;
    const long total = (long)nx * ny * p.B * p.nphase;
    if (total <= 0) return GA_OK;
    if (total > (1L << 30)) return GA_ERR_ARG;
    const int span = (BN - 1) * p.in_stride + (p.J - 1) * p.tap_step + 1;
    const int rpp = (span + p.in_stride - 1) / p.in_stride + 1;
    const size_t lds = 2 * BM * PITCH + (size_t)p.in_stride * rpp * PITCH;
    auto kern = conv_mfma_kernel<WM, WN, WTN>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GA_ERR_LAUNCH;
    }
    const int grid = (int)((total + 7) / 8 * 8);
    kern<<<grid, WM * WN * 64, lds, s>>>(p, nx, ny, (int)total, rpp);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_conv_mfma_impl(const ConvParams& p, hipStream_t s) {
    if (!ga_conv_mfma_supported(p)) return GA_ERR_ARG;
    if (p.Lout == 0) return GA_OK;
    if (p.M <= 64) return launch_mfma<1, 4, 32>(p, s);
    return launch_mfma<2, 2, 64>(p, s);
}
