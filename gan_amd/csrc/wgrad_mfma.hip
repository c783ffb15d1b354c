// bf16 MFMA weight-gradient kernel for the generalised NLC convolution, gfx950.
//
//   dW[m][j][c] += sum_{b,t} G[b][t][m] * X[b][t*S + j*step + off0][c]
//
// GEMM view per tap j: A = G^T (m x t), B = X_j (t x c), reduction over time.  In channels-last memory BOTH operands
// have the reduction index on the slow axis, so the LDS tiles stay row-major [t][channel] (coalesced 16-byte staging)
// and the MFMA fragments are fetched with the CDNA4 transposing LDS read ds_read_b64_tr_b16
// (cdna_hip_programming.md T10), which hands each lane 4 consecutive K (time) values of its channel column.
// A workgroup owns a 128 (m) x 64 (c) tile for a GROUP of TG=4 consecutive taps: the G tile and one input window
// are staged once per 32-step time chunk and reused by the four taps (for the stride-4 discriminator layers the four
// taps are the four row phases of the window, so every staged row is used).  Time is split over gridDim.y; partial
// sums are added with fp32 atomics, whole 128-byte row segments per wave instruction.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WG_BM 128
#define WG_BC 64
#define WG_KT 32
#define WG_TG 4
#define WG_PG (WG_BM * 2 + 64)  // 320 B: 4 consecutive rows start 64 B apart modulo the 256-B bank row
#define WG_PX (WG_BC * 2 + 64)  // 192 B
#define WG_MAXSPAN 256
#define WG_NXP (WG_MAXSPAN * 8 / 256)

__device__ __forceinline__ bf16x8 tr_frag(const char* base /* address of [k = 8h+q][col 4p] for s = 0 */, int pitch) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + 4 * pitch));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradParams p, int nchunks_per_b, int total_chunks, int chunks_per_split,
                                                          int ctiles, int tgroups, int span, int rpp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Gl = smem;                     // [WG_KT][WG_PG]
    char* Xl = smem + WG_KT * WG_PG;     // [S][rpp][WG_PX]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wc = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    const int g4 = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;

    int bx = blockIdx.x;
    const int tg = bx % tgroups; bx /= tgroups;
    const int ct = bx % ctiles;
    const int mt = bx / ctiles;
    const int mbase = mt * WG_BM, cbase = ct * WG_BC, j0 = tg * WG_TG;
    const int S = p.in_stride;
    const bf16_t* __restrict__ g = (const bf16_t*)p.g;
    const bf16_t* __restrict__ x = (const bf16_t*)p.x;

    f32x16 acc[2][WG_TG];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int jl = 0; jl < WG_TG; ++jl)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][jl][i] = 0.f;

    const int c_begin = blockIdx.y * chunks_per_split;
    int c_end = c_begin + chunks_per_split;
    if (c_end > total_chunks) c_end = total_chunks;

    bf16x8 greg[2], xreg[WG_NXP];
    auto zero8 = []() { bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
        return v; };
    auto load_chunk = [&](int ci) {
        const int b = ci / nchunks_per_b;
        const int t0 = (ci - b * nchunks_per_b) * WG_KT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * 256;
            const int row = id >> 4, part = id & 15;
            const int t = t0 + row, m = mbase + part * 8;
            bf16x8 v = zero8();
            if (t < p.Lout && m < p.M) v = *(const bf16x8*)(g + ((size_t)b * p.LgTot + t) * p.M + m);
            greg[i] = v;
        }
        const int R0 = t0 * S + j0 * p.tap_step + p.off0;
#pragma unroll
        for (int i = 0; i < WG_NXP; ++i) {
            const int id = tid + i * 256;
            bf16x8 v = zero8();
            if (id < span * 8) {
                const int u = id >> 3, part = id & 7;
                const int R = R0 + u, c = cbase + part * 8;
                if (R >= 0 && R < p.Lin && c < p.Cin) v = *(const bf16x8*)(x + ((size_t)b * p.Lin + R) * p.Cin + c);
            }
            xreg[i] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * 256;
            *(bf16x8*)(Gl + (id >> 4) * WG_PG + (id & 15) * 16) = greg[i];
        }
#pragma unroll
        for (int i = 0; i < WG_NXP; ++i) {
            const int id = tid + i * 256;
            if (id < span * 8) {
                const int u = id >> 3, part = id & 7;
                *(bf16x8*)(Xl + ((u % S) * rpp + u / S) * WG_PX + part * 16) = xreg[i];
            }
        }
    };

    // per-lane base addresses of the transposed reads: row = 8h + q (+16*kk2 +4*s), column = 16*(g4&1) + 4*pp
    const int krow = 8 * h + q;
    const char* a_base = Gl + krow * WG_PG + (wm * 64 + 16 * (g4 & 1) + 4 * pp) * 2;
    const int xcol = (wc * 32 + 16 * (g4 & 1) + 4 * pp) * 2;

    if (c_begin < c_end) load_chunk(c_begin);
    for (int ci = c_begin; ci < c_end; ++ci) {
        __syncthreads();  // previous chunk's reads are done
        store_chunk();
        __syncthreads();
        if (ci + 1 < c_end) load_chunk(ci + 1);
#pragma unroll
        for (int kk2 = 0; kk2 < 2; ++kk2) {
            bf16x8 a[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = tr_frag(a_base + kk2 * 16 * WG_PG + mi * 32 * 2, WG_PG);
#pragma unroll
            for (int jl = 0; jl < WG_TG; ++jl) {
                const int uj = jl * p.tap_step;
                const char* xb = Xl + ((uj % S) * rpp + uj / S + kk2 * 16 + krow) * WG_PX + xcol;
                const bf16x8 bfrag = tr_frag(xb, WG_PX);
                if (j0 + jl < p.J) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
                        acc[mi][jl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], bfrag, acc[mi][jl], 0, 0, 0);
                }
            }
        }
    }

    const int c = cbase + wc * 32 + r;
    if (c >= p.Cin) return;
#pragma unroll
    for (int jl = 0; jl < WG_TG; ++jl) {
        const int j = j0 + jl;
        if (j >= p.J) continue;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = mbase + wm * 64 + mi * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (m < p.M) atomicAdd(&p.dw[((size_t)m * p.J + j) * p.Cin + c], acc[mi][jl][i]);
            }
        }
    }
}

static int wg_span(const WgradParams& p) { return (WG_KT - 1) * p.in_stride + (WG_TG - 1) * p.tap_step + 1; }

int ga_wgrad_mfma_supported(const WgradParams& p) {
    if (p.M % 8 || p.Cin % 8 || p.M < 32 || p.Cin < 16) return 0;
    if (p.g_stride != 1 || p.g_phase != 0 || p.in_stride < 1 || p.tap_step < 1) return 0;
    if (((uintptr_t)p.g | (uintptr_t)p.x | (uintptr_t)p.dw) & 15) return 0;
    if (wg_span(p) > WG_MAXSPAN) return 0;
    if ((long)p.B * p.Lout < 256) return 0;
    return 1;
}

int ga_wgrad_mfma_impl(const WgradParams& p, hipStream_t s) {
    if (!ga_wgrad_mfma_supported(p)) return GA_ERR_ARG;
    const int span = wg_span(p);
    const int rpp = (span + p.in_stride - 1) / p.in_stride + 1;
    const int mtiles = (p.M + WG_BM - 1) / WG_BM, ctiles = (p.Cin + WG_BC - 1) / WG_BC, tgroups = (p.J + WG_TG - 1) / WG_TG;
    const int nchunks_per_b = (p.Lout + WG_KT - 1) / WG_KT;
    const long total_chunks = (long)nchunks_per_b * p.B;
    const long base = (long)mtiles * ctiles * tgroups;
    if (base > 65535L * 16 || total_chunks > (1L << 30)) return GA_ERR_ARG;
    long splits = (1024 + base - 1) / base;           // aim for ~1k workgroups
    const long max_splits = (total_chunks + 7) / 8;    // at least 8 chunks (256 time steps) per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    const int cps = (int)((total_chunks + splits - 1) / splits);
    splits = (total_chunks + cps - 1) / cps;
    const size_t lds = WG_KT * WG_PG + (size_t)p.in_stride * rpp * WG_PX;
    if (lds > 64 * 1024) return GA_ERR_ARG;
    wgrad_mfma_kernel<<<dim3((unsigned)base, (unsigned)splits), 256, lds, s>>>(p, nchunks_per_b, (int)total_chunks, cps, ctiles,
                                                                              tgroups, span, rpp);
    GA_CHECK_LAUNCH();
    return GA_OK;
}
