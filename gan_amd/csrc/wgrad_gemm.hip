// bf16 MFMA weight-gradient kernel, "gathered GEMM" form (gfx950):
//
//   dW[m][k] += sum_n G[n][m] * X[n][k]      n = (batch item, output step) flattened, k = (tap j, channel c)
//
// Both operands have the REDUCTION index n on the slow memory axis (channels-last), so both LDS tiles stay
// row-major [n][channel]: rows of G are contiguous in m, and the im2col row of X for step n is gathered piece by
// piece (8 channels = 16 bytes) exactly as in conv_gemm.hip.  Every piece goes global -> LDS with
// global_load_lds_dwordx4 (no register staging); MFMA fragments come out of LDS with the CDNA4 transposing read
// ds_read_b64_tr_b16, which hands each lane 4 consecutive n of its column.
//
//   block tile 128 (m) x 128 (k) of dW, 4 waves as 2 x 2, wave tile 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_bf16
//   reduction step 64 rows; two LDS buffers per operand ([64][128] bf16 = 16 KiB each), next step in flight
//   LDS rows are 256 B; the 64-byte quarter q of row r is stored at quarter q ^ (r & 3), so the four rows one
//   transposing read touches land on different bank quarters (conflict-free); the XOR sits on the SOURCE piece.
//   The reduction range is split over gridDim.y; partial tiles are added with fp32 atomics (128-byte row segments).
//   M <= 64 (D blocks.1): the block tile is 64 (m) x 256 (k) instead -- one G tile, TWO gathered-X tiles of 128 columns, the four
//   waves side by side along k -- so that a reduction step carries the same 16 MFMAs per wave as the 128 x 128 tile does (with
//   the 128 x 128 tile half of the G tile was padding and every wave had 8 MFMAs per barrier pair: half the rate).
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;


typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void wg_glds16(uintptr_t src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}
// 16 bytes per lane: buffer resource + 32-bit byte offset -> LDS (a plain function: see bglds16 in conv_gemm.hip)
__device__ __forceinline__ void wg_bglds16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds_wave_base, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ uintptr_t wg_sel(bool ok, uintptr_t a, uintptr_t z) {
    return z ^ ((z ^ a) & (uintptr_t)(-(long long)ok));
}
// fragment of 32 columns x 16 reduction rows: two transposing reads, rows q and q + 4 of this lane's 8-row half
__device__ __forceinline__ bf16x8 wg_tr_frag(const char* p0, const char* p1) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)p1);
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}

// Bias gradient from the same MFMAs: db[m] = sum_n G[n][m] * 1 is column k = Ktot of dW when the gathered operand gets a
// constant-one column there.  The 16-byte piece at k = Ktot reads this page (bf16 1.0 followed by seven zeros).
static __device__ __attribute__((aligned(16))) unsigned int wgg_one_page[4] = {0x00003F80u, 0u, 0u, 0u};

#define WGG_RK 64    // reduction rows per step
#define WGG_TILE (WGG_RK * 256)  // bytes of one operand tile

// NBUF = 2: next step's loads in flight behind the current multiply (64 KiB of LDS, two workgroups per CU).
// NBUF = 1: one buffer, 32 KiB, 128 registers: four workgroups per CU hide each other's load round trips (the form that won for
//           the forward kernel on grids several workgroups per CU deep).
// NB = number of 128-column gathered-X tiles per workgroup: 1 (128 x 128 block tile) or 2 (64 x 256, M <= 64 only)
// BUF: both operands staged by direct-to-LDS BUFFER loads (tensors under 2 GiB): every row this lane stages carries a running
// 32-bit byte offset into g and into x that advances by a constant per reduction step (and by another constant when the row
// wraps into the next batch item), so a load needs no 64-bit address arithmetic; rows past the end, columns past M / K and
// taps outside the item get an offset beyond the buffer and the hardware's range check returns zeros for them.  The bias
// column's constant-one piece is written to its LDS slot once, up front: the lane that owns the slot never loads into it.
template <int NBUF, int NB, bool BUF>
__device__ __forceinline__ void wgrad_tile(const WgradParams& p, char* const smem, const int wk, const int ktiles, const int Ktot,
                                           const int steps_per_split, const int nsteps_total, const int base) {
    const int bx = wk % base, by = wk / base;
    // NB = 2: the G tile has 64 columns; reduction rows n and n + 32 share one 256-byte LDS row (n + 32 in its second half), so
    // the tile is 8 KiB, the row structure the transposing reads and their swizzle rely on stays, and four workgroups fit a CU
    constexpr int ATILE = NB == 2 ? WGG_TILE / 2 : WGG_TILE;
    char* const Al = smem;                    // [NBUF][64][256 B]  G rows   (NB = 2: [32][256 B])
    char* const Bl = smem + NBUF * ATILE;     // [NBUF][NB][64][256 B]  gathered X rows

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int r = lane & 31, h = lane >> 5;
    const int g4 = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int kt = bx % ktiles, mt = bx / ktiles;
    const int mbase = mt * 128, kbase = kt * 128 * NB;
    const int S = p.in_stride, Cin = p.Cin;
    const uintptr_t zero = (uintptr_t)ga_zero_page;

    const int s_begin = by * steps_per_split;
    int s_end = s_begin + steps_per_split;
    if (s_end > nsteps_total) s_end = nsteps_total;
    if (s_begin >= s_end && !p.ws) return;  // (slab mode: an empty slice still writes its zero tile)

    // staging role: instruction i covers rows i*16 + wid*4 + (lane>>4); this lane's physical 16-byte slot lane&15
    const int prow = lane >> 4;
    const int lpiece = (lane & 15) ^ (prow << 2);  // logical piece held by this lane's slot (row & 3 == prow)
    const int am = mbase + lpiece * 8;
    const bool a_ok = am < p.M;
    const uintptr_t one = (uintptr_t)wgg_one_page;
    const int Kcols = Ktot + (p.dbias ? 1 : 0);
    bool b_ok[NB], b_one[NB];
    int bc[NB], bdR[NB];
#pragma unroll
    for (int sb = 0; sb < NB; ++sb) {
        const int bk = kbase + sb * 128 + lpiece * 8;
        b_ok[sb] = bk < Ktot;
        b_one[sb] = p.dbias && bk == Ktot;  // the bias column (fused only without slabs: see ga_wgrad_gemm_impl)
        const int bj = b_ok[sb] ? bk / Cin : 0;
        bc[sb] = b_ok[sb] ? bk - bj * Cin : 0;
        bdR[sb] = bj * p.tap_step + p.off0;
    }
    // per-row running (b, t) for the 4 rows this lane stages
    int rb[4], rt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long n = (long)s_begin * WGG_RK + i * 16 + wid * 4 + prow;
        rb[i] = (int)(n / p.Lout);
        rt[i] = (int)(n - (long)rb[i] * p.Lout);
    }
    const bf16_t* __restrict__ g = (const bf16_t*)p.g;
    const bf16_t* __restrict__ x = (const bf16_t*)p.x;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, BUF ? (int)((long)p.B * p.LgTot * p.M * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, BUF ? (int)((long)p.B * p.Lin * Cin * 2) : 0, 0x00020000);
    unsigned grow[4], xrow[4];  // byte offsets of (row's batch item, row's step t, column 0) in g / (item, input row t*S, channel 0) in x
    unsigned xcol[NB];          // + this lane's (tap, channel) inside the row window
    if constexpr (BUF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            grow[i] = (unsigned)((rb[i] * p.LgTot + rt[i]) * p.M) * 2u;
            xrow[i] = (unsigned)((rb[i] * p.Lin + rt[i] * S) * Cin) * 2u;
        }
#pragma unroll
        for (int sb = 0; sb < NB; ++sb) xcol[sb] = (unsigned)(bdR[sb] * Cin + bc[sb]) * 2u;  // (wrapping: bdR may be negative)
        // the bias column: its piece (bf16 1.0, seven zeros) goes into this lane's slot of every row instruction and buffer, once
#pragma unroll
        for (int sb = 0; sb < NB; ++sb)
            if (b_one[sb]) {
                for (int buf = 0; buf < NBUF; ++buf)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *(uint4*)(Bl + buf * NB * WGG_TILE + wid * 1024 + sb * WGG_TILE + i * 4096 + lane * 16) = make_uint4(0x00003F80u, 0u, 0u, 0u);
            }
    }
    const unsigned am2x = (unsigned)am * 2u;
    // workgroup-uniform: does this k-tile hold the bias column (only the last one can), does a step cross at most one item boundary
    const bool tile_has_one = p.dbias && kbase <= Ktot && Ktot < kbase + 128 * NB;
    const bool single_wrap = p.Lout >= WGG_RK;
    const unsigned g_step = (unsigned)(WGG_RK * p.M) * 2u, g_wrap = (unsigned)((p.LgTot - p.Lout) * p.M) * 2u;
    const unsigned x_step = (unsigned)(WGG_RK * S * Cin) * 2u, x_wrap = (unsigned)((p.Lin - p.Lout * S) * Cin) * 2u;

    auto issue = [&](int buf) {
        char* const ab = Al + buf * ATILE + wid * 1024;
        char* const bb = Bl + buf * NB * WGG_TILE + wid * 1024;
        if constexpr (BUF) {
            if (NB == 2) {
                const bool up = lpiece >= 8;
                const unsigned am2b = (unsigned)((lpiece & 7) * 8) * 2u;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int b_ = up ? rb[i + 2] : rb[i];
                    const unsigned go = up ? grow[i + 2] : grow[i];
                    wg_bglds16(rsG, ab + i * 4096, (b_ < p.B && (lpiece & 7) * 8 < p.M) ? go + am2b : OOB);
                }
            }
            // (the two conditions below are workgroup-uniform: the straight-line forms carry no exec-mask regions and no loops;
            // measured: no change in kernel time -- the staging instructions are not what bounds this kernel -- kept for the ISA)
            if (!tile_has_one) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool n_ok = rb[i] < p.B;
                    if (NB == 1) wg_bglds16(rsG, ab + i * 4096, (n_ok && a_ok) ? grow[i] + am2x : OOB);
#pragma unroll
                    for (int sb = 0; sb < NB; ++sb) {
                        const bool ok = n_ok && b_ok[sb] && (unsigned)(rt[i] * S + bdR[sb]) < (unsigned)p.Lin;
                        wg_bglds16(rsX, bb + sb * WGG_TILE + i * 4096, ok ? xrow[i] + xcol[sb] : OOB);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool n_ok = rb[i] < p.B;
                    if (NB == 1) wg_bglds16(rsG, ab + i * 4096, (n_ok && a_ok) ? grow[i] + am2x : OOB);
#pragma unroll
                    for (int sb = 0; sb < NB; ++sb) {
                        const bool ok = n_ok && b_ok[sb] && (unsigned)(rt[i] * S + bdR[sb]) < (unsigned)p.Lin;
                        if (!b_one[sb]) wg_bglds16(rsX, bb + sb * WGG_TILE + i * 4096, ok ? xrow[i] + xcol[sb] : OOB);
                    }
                }
            }
            if (single_wrap) {  // Lout >= 64: a row crosses at most one item boundary per step
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rt[i] += WGG_RK;
                    const bool wr = rt[i] >= p.Lout;
                    rt[i] -= wr ? p.Lout : 0;
                    rb[i] += wr ? 1 : 0;
                    grow[i] += g_step + (wr ? g_wrap : 0u);
                    xrow[i] += x_step + (wr ? x_wrap : 0u);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rt[i] += WGG_RK;
                    grow[i] += g_step;
                    xrow[i] += x_step;
                    while (rt[i] >= p.Lout) {
                        rt[i] -= p.Lout;
                        ++rb[i];
                        grow[i] += g_wrap;
                        xrow[i] += x_wrap;
                    }
                }
            }
            return;
        }
        if (NB == 2) {  // LDS row rho = i*16 + wid*4 + prow: pieces 0-7 = columns of reduction row rho, pieces 8-15 = of row rho + 32
            const bool up = lpiece >= 8;
            const int am2 = (lpiece & 7) * 8;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int b_ = up ? rb[i + 2] : rb[i], t_ = up ? rt[i + 2] : rt[i];
                const uintptr_t asrc = (uintptr_t)(g + ((size_t)b_ * p.LgTot + t_) * p.M + am2);
                wg_glds16(wg_sel(b_ < p.B && am2 < p.M, asrc, zero), ab + i * 4096);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool n_ok = rb[i] < p.B;
            if (NB == 1) {
                const uintptr_t asrc = (uintptr_t)(g + ((size_t)rb[i] * p.LgTot + rt[i]) * p.M + am);
                wg_glds16(wg_sel(n_ok && a_ok, asrc, zero), ab + i * 4096);
            }
#pragma unroll
            for (int sb = 0; sb < NB; ++sb) {
                const int R = rt[i] * S + bdR[sb];
                const uintptr_t bsrc = (uintptr_t)(x + ((size_t)rb[i] * p.Lin + R) * Cin + bc[sb]);
                wg_glds16(wg_sel(b_one[sb], one, wg_sel(n_ok && b_ok[sb] && R >= 0 && R < p.Lin, bsrc, zero)), bb + sb * WGG_TILE + i * 4096);
            }
            // advance this row by one reduction step
            rt[i] += WGG_RK;
            while (rt[i] >= p.Lout) { rt[i] -= p.Lout; ++rb[i]; }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    // transposed-read addressing: row = kk*16 + 8h + q (+4), columns 16*(g4&1) + 4*pp .. +3 of the wave's 32-column
    // fragment; element column e -> logical piece e>>3, byte (e&7)*2; physical piece = logical ^ ((row&3)<<2), row&3 == q
    // M <= 64: only the first 64 rows of the 128-row tile exist.  Instead of idling, waves wm = 1 take the second half
    // of every 64-row reduction step for the SAME 64 x 64 output sub-tile (the partial sums meet in the atomics).
    // (NB = 2: all four waves work on the 64 rows, side by side along k -- waves wm = 1 on the second X tile -- over the whole step)
    const bool halfm = NB == 1 && p.M <= 64;
    const int wme = (halfm || NB == 2) ? 0 : wm;
    const int bsub = NB == 2 ? wm : 0;
    const int kk0 = halfm ? wm * 2 : 0, kk1 = halfm ? wm * 2 + 2 : 4;
    const int ecolA[2] = {wme * 64 + 16 * (g4 & 1) + 4 * pp, wme * 64 + 32 + 16 * (g4 & 1) + 4 * pp};
    const int ecolB[2] = {wn * 64 + 16 * (g4 & 1) + 4 * pp, wn * 64 + 32 + 16 * (g4 & 1) + 4 * pp};
    int offA[2], offB[2], offA2[2];  // offA2: the same fragment of reduction rows + 32 (NB = 2: second half of the LDS row)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offA[i] = ((((ecolA[i] >> 3) ^ (q << 2)) << 4) | ((ecolA[i] & 7) << 1)) + (8 * h + q) * 256;
        offB[i] = ((((ecolB[i] >> 3) ^ (q << 2)) << 4) | ((ecolB[i] & 7) << 1)) + (8 * h + q) * 256;
        const int e2 = ecolA[i] + 64;
        offA2[i] = ((((e2 >> 3) ^ (q << 2)) << 4) | ((e2 & 7) << 1)) + (8 * h + q) * 256;
    }
    const bool wave_active = (mbase + wme * 64) < p.M && (kbase + bsub * 128 + wn * 64) < Kcols;

    if (NBUF == 2 && s_begin < s_end) issue(0);
    for (int st = s_begin; st < s_end; ++st) {
        const int buf = NBUF == 2 ? (st - s_begin) & 1 : 0;
        if (NBUF == 1) {
            if (st > s_begin) __syncthreads();  // everyone is done reading the single buffer
            issue(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (NBUF == 2 && st + 1 < s_end) issue(buf ^ 1);
        if (wave_active) {
            const char* Ab = Al + buf * ATILE;
            const char* Bb = Bl + (buf * NB + bsub) * WGG_TILE;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk < kk0 || kk >= kk1) continue;
                bf16x8 a[2], b[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const char* ap = NB == 2 ? Ab + (kk & 1) * 4096 + (kk < 2 ? offA[mi] : offA2[mi]) : Ab + kk * 4096 + offA[mi];
                    a[mi] = wg_tr_frag(ap, ap + 1024);
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) b[ni] = wg_tr_frag(Bb + kk * 4096 + offB[ni], Bb + kk * 4096 + offB[ni] + 1024);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
        }
    }

    if (!wave_active) return;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int k = kbase + bsub * 128 + wn * 64 + ni * 32 + r;
        if (k >= Kcols) continue;
        if (k == Ktot) {  // bias column
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = mbase + wme * 64 + mi * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (m < p.M) atomicAdd(&p.dbias[m], acc[mi][ni][i]);
                }
            continue;
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = mbase + wme * 64 + mi * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (m < p.M) {
                    if (p.ws) p.ws[(size_t)by * p.M * Ktot + (size_t)m * Ktot + k] = acc[mi][ni][i];
                    else atomicAdd(&p.dw[(size_t)m * Ktot + k], acc[mi][ni][i]);
                }
            }
        }
    }
}

template <int NBUF, int NB, bool BUF = false>
__global__ __launch_bounds__(256, NBUF == 1 ? 4 : 1) void wgrad_gemm_kernel(WgradParams p, int ktiles, int Ktot, int Ntot, int steps_per_split, int nsteps_total, int base, int total, int xcd_order) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    // work item = (output tile bx, reduction slice by), tiles fastest.  Workgroups are dealt to the 8 XCDs round-robin, so in
    // launch order the k-tiles of one reduction slice -- which read the SAME rows of G and overlapping rows of x -- would land on
    // eight different L2s and each fetch them from the fabric again (PMC: 4.9 GB per launch for 1.5 GB of operands on D blocks.1);
    // with xcd_order the ids an XCD receives are consecutive work items instead.  Measured at B = 128: +8-16 % on the 64 x 256
    // tiles of D blocks.1, +2 % / -4 % / 0 on blocks 2 / 3 / 4 with 128 x 128 tiles: on for the former only (GA_WGRAD_XCD=2: all).
    int wk = blockIdx.x;
    if (xcd_order) wk = (wk & 7) * (gridDim.x >> 3) + (wk >> 3);
    if (wk >= total) return;
    wgrad_tile<NBUF, NB, BUF>(p, smem, wk, ktiles, Ktot, steps_per_split, nsteps_total, base);
}

// Several weight-gradient problems in ONE launch (the generator's forty 3.7-GFLOP layers: each alone is a 40-50 us launch of
// ~500 workgroups whose MFMA time is 7 us).  Workgroup ids are dealt problem after problem; the plan of each problem (tiles,
// slices) is the single-launch plan with a smaller workgroup target.
#define WGM_MAX 20
struct WgradMultiArgs {
    WgradParams p[WGM_MAX];
    int ktiles[WGM_MAX], sps[WGM_MAX], nsteps[WGM_MAX], base[WGM_MAX];
    int start[WGM_MAX + 1];  // first workgroup of a problem
    int n;
    int xcd_order;           // consecutive work items on one XCD (its L2 then serves the tiles of a reduction slice: they share rows)
};
__global__ __launch_bounds__(256, 4) void wgrad_gemm_multi_kernel(const WgradMultiArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    int id = blockIdx.x;
    if (a.xcd_order) id = (id & 7) * (gridDim.x >> 3) + (id >> 3);
    if (id >= a.start[a.n]) return;
    int i = 0;
    while (i + 1 < a.n && id >= a.start[i + 1]) ++i;
    const WgradParams& p = a.p[i];
    wgrad_tile<1, 1, true>(p, smem, id - a.start[i], a.ktiles[i], p.J * p.Cin, a.sps[i], a.nsteps[i], a.base[i]);
}

int ga_wgrad_gemm_supported(const WgradParams& p) {
    if (p.M % 8 || p.Cin % 8 || p.M < 32 || p.Cin < 8 || p.J < 1) return 0;
    if (p.g_stride != 1 || p.g_phase != 0 || p.tap_step < 0) return 0;
    if (((uintptr_t)p.g | (uintptr_t)p.x | (uintptr_t)p.dw) & 15) return 0;
    if ((long)p.B * p.Lout < 256 || (long)p.B * p.Lout > (1L << 30) || (long)p.J * p.Cin > (1L << 24)) return 0;
    return 1;
}

// dw[i] += sum over slabs
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, long slab, int nslab) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 a = ((const float4*)dw)[i];
        for (int sidx = 0; sidx < nslab; ++sidx) {
            const float4 v = ((const float4*)(ws + sidx * slab))[i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        ((float4*)dw)[i] = a;
    }
}

// GA_WGRAD_TARGET overrides the number of workgroups a launch aims for (tuning aid)
static long wgg_target_blocks() {
    static long v = -1;
    if (v < 0) {
        const char* e = getenv("GA_WGRAD_TARGET");
        v = e ? atol(e) : 2048;  // four single-buffer workgroups are resident per CU; swept 768 / 1024 / 1536 / 2560 at B = 24 and 128
        if (v < 1) v = 1;
    }
    return v;
}

// 64 x 256 block tiles for M <= 64 (GA_WGRAD_WIDE64=0: the 128 x 128 tile for every layer)
static bool wgg_wide64(const WgradParams& p) {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("GA_WGRAD_WIDE64");
        on = (e && e[0] == '0') ? 0 : 1;
    }
    return on && p.M <= 64;
}

static void wgg_plan(const WgradParams& p, long& base, long& splits, int& sps, long& nsteps, int& ktiles, long target = 0) {
    const int Ktot = p.J * p.Cin + (p.dbias ? 8 : 0);  // (+ one 16-byte piece for the bias column)
    const long Ntot = (long)p.B * p.Lout;
    const int mtiles = (p.M + 127) / 128;
    const int kw = wgg_wide64(p) ? 256 : 128;
    ktiles = (Ktot + kw - 1) / kw;
    base = (long)mtiles * ktiles;
    nsteps = (Ntot + WGG_RK - 1) / WGG_RK;
    if (target <= 0) target = wgg_target_blocks();
    splits = (target + base - 1) / base;  // workgroups aimed for
    // at least 48 steps (3072 rows) per split: shorter slices are dominated by their prologue and the fp32 atomics of their
    // 64 KiB tile (swept at B = 24) - unless the grid would then leave CUs idle (the generator's few-tile layers), where
    // slices go down to 8 steps to reach about two workgroups per CU
    const long cap48 = (nsteps + 47) / 48, cap8 = (nsteps + 7) / 8;
    if (splits > cap48) splits = cap48;
    if (splits < 1) splits = 1;
    const long floor_blocks = target < 512 ? target : 512;
    if (base * splits < floor_blocks) {
        long want = (floor_blocks + base - 1) / base;
        if (want > cap8) want = cap8;
        if (want > splits) splits = want;
    }
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    sps = (int)((nsteps + splits - 1) / splits);
    splits = (nsteps + sps - 1) / sps;
}

size_t ga_wgrad_gemm_workspace_elems(const WgradParams& p) {
    if (!ga_wgrad_gemm_supported(p) || p.M <= 64) return 0;  // (M <= 64: two waves share an output sub-tile: atomics only)
    long base, splits, nsteps;
    int sps, ktiles;
    WgradParams q = p;
    q.dbias = nullptr;  // (slab mode never fuses the bias column: plan exactly as ga_wgrad_gemm_impl will)
    wgg_plan(q, base, splits, sps, nsteps, ktiles);
    return (size_t)splits * p.M * p.J * p.Cin;
}

int ga_wgrad_gemm_impl(const WgradParams& p_in, hipStream_t s) {
    if (!ga_wgrad_gemm_supported(p_in)) return GA_ERR_ARG;
    WgradParams p = p_in;
    if (p.M <= 64) p.ws = nullptr;
    if (p.ws) p.dbias = nullptr;  // slab mode reduces dW only: the caller adds the bias gradient (ga_colsum)
    const int Ktot = p.J * p.Cin;
    const long Ntot = (long)p.B * p.Lout;
    long base, splits, nsteps;
    int sps, ktiles;
    wgg_plan(p, base, splits, sps, nsteps, ktiles);
    if (p.ws && (((uintptr_t)p.ws & 15) || ((size_t)p.M * Ktot) % 4)) return GA_ERR_ALIGN;
    static int nbuf = -1;
    if (nbuf < 0) {
        const char* e = getenv("GA_WGRAD_NBUF");
        nbuf = (e && e[0] == '2') ? 2 : 1;  // single buffer by default: 20-40 % faster on every D layer (B = 24 and 128)
    }
    int xcd = 1;  // (read per launch: same-process A/B)
    {
        const char* e = getenv("GA_WGRAD_XCD");
        if (e) xcd = atoi(e);
    }
    const int xcd_wide = xcd >= 1, xcd_all = xcd >= 2;
    const long total = base * splits;
    if (total > (1L << 30)) return GA_ERR_ARG;
    const unsigned grid = (unsigned)((total + 7) / 8 * 8);
    // buffer-load staging where the 32-bit byte offsets reach (GA_WGRAD_BUF=0: the flat-address form)
    const char* eb = getenv("GA_WGRAD_BUF");
    const bool buf = !(eb && eb[0] == '0') && (long)p.B * p.LgTot * p.M * 2 < (1L << 31) - 64 && (long)p.B * p.Lin * p.Cin * 2 < (1L << 31) - 64;
#define WGG_LAUNCH(NBUF_, NB_, LDS_, XCD_)                                                                                          \
    do {                                                                                                                            \
        if (buf) wgrad_gemm_kernel<NBUF_, NB_, true><<<grid, 256, LDS_, s>>>(p, ktiles, Ktot, (int)Ntot, sps, (int)nsteps, (int)base, (int)total, XCD_);  \
        else wgrad_gemm_kernel<NBUF_, NB_, false><<<grid, 256, LDS_, s>>>(p, ktiles, Ktot, (int)Ntot, sps, (int)nsteps, (int)base, (int)total, XCD_);    \
    } while (0)
    if (wgg_wide64(p)) {
        WGG_LAUNCH(1, 2, WGG_TILE / 2 + 2 * WGG_TILE, xcd_wide);
    } else if (nbuf == 1) {
        WGG_LAUNCH(1, 1, 2 * WGG_TILE, xcd_all);
    } else {
        const size_t lds = 4 * WGG_TILE;
        if (hipFuncSetAttribute(buf ? (const void*)wgrad_gemm_kernel<2, 1, true> : (const void*)wgrad_gemm_kernel<2, 1, false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GA_ERR_LAUNCH;
        WGG_LAUNCH(2, 1, lds, xcd_all);
    }
#undef WGG_LAUNCH
    GA_CHECK_LAUNCH();
    if (p.ws) {
        const long n4 = (long)p.M * Ktot / 4;
        long grid = (n4 + 255) / 256;
        if (grid > 4096) grid = 4096;
        wgrad_slab_reduce_kernel<<<(unsigned)grid, 256, 0, s>>>(p.ws, p.dw, n4, (long)p.M * Ktot, (int)splits);
        GA_CHECK_LAUNCH();
    }
    return GA_OK;
}

// Can ga_wgrad_gemm_multi_impl take this problem?  (The 128 x 128 single-buffer, buffer-staged, atomics form only.)
int ga_wgrad_gemm_multi_supported(const WgradParams& p) {
    if (!ga_wgrad_gemm_supported(p) || p.M <= 64 || p.ws || p.Lout <= 0) return 0;
    if (p.dbias && p.LgTot != p.Lout) return 0;
    return (long)p.B * p.LgTot * p.M * 2 < (1L << 31) - 64 && (long)p.B * p.Lin * p.Cin * 2 < (1L << 31) - 64;
}

int ga_wgrad_gemm_multi_impl(const WgradParams* items, int n, hipStream_t s) {
    for (int i = 0; i < n; ++i)
        if (!ga_wgrad_gemm_multi_supported(items[i])) return GA_ERR_UNSUPPORTED;
    for (int c0 = 0; c0 < n; c0 += WGM_MAX) {
        const int cn = n - c0 < WGM_MAX ? n - c0 : WGM_MAX;
        WgradMultiArgs a;
        a.n = cn;
        // about two launches' worth of workgroups over the whole group, at least one problem's tiles each
        long tpct = 100;  // workgroup target of a group, in percent of a single launch's (swept 25 .. 400 on the generator's 22 layers)
        {
            const char* e = getenv("GA_WGRAD_MULTI_TARGET");
            if (e && atoi(e) > 0) tpct = atoi(e);
        }
        const long target = (tpct * wgg_target_blocks() / 100 + cn - 1) / cn;
        long at = 0;
        for (int i = 0; i < cn; ++i) {
            a.p[i] = items[c0 + i];
            long base, splits, nsteps;
            int sps, ktiles;
            wgg_plan(a.p[i], base, splits, sps, nsteps, ktiles, target);
            a.ktiles[i] = ktiles;
            a.sps[i] = sps;
            a.nsteps[i] = (int)nsteps;
            a.base[i] = (int)base;
            a.start[i] = (int)at;
            at += base * splits;
            if (at > (1L << 30)) return GA_ERR_ARG;
        }
        a.start[cn] = (int)at;
        for (int i = cn + 1; i <= WGM_MAX; ++i) a.start[i] = (int)at;
        {
            const char* e = getenv("GA_WGRAD_MULTI_XCD");
            a.xcd_order = e ? atoi(e) : 1;
        }
        wgrad_gemm_multi_kernel<<<(unsigned)((at + 7) / 8 * 8), 256, 2 * WGG_TILE, s>>>(a);
        GA_CHECK_LAUNCH();
    }
    return GA_OK;
}
