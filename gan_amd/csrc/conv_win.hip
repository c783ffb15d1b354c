// bf16 MFMA implicit-GEMM kernel, "LDS window" form, for the narrow-channel layers of the discriminator (gfx950).
//
//   Y[n][m] = sum_{j < J} sum_{c < Cin} Wp[m][j][c] * X[n*S + j + off0][c]         one batch item per workgroup tile
//
// The gathered-GEMM kernels (conv_gemm.hip, conv_pp.hip) fetch every im2col piece of x from L2 once per tap: with 41 taps at
// stride 4 (or 11 taps at stride 1) each input element goes L2 -> LDS about ten times, and for Cin = 32 / 64 that gather, not
// the matrix pipe, sets the pace (500-550 TFLOP/s on D blocks.1 against 850-900 on the wide layers).  Here the input rows a
// tile of BN output steps touches - a CONTIGUOUS run of (BN-1)*S + J rows of channels-last x - are staged in LDS ONCE, and all
// J taps read their B fragments from that window at shifted rows; only the weights stream through a three-stage ring.
//
//   Cin = 32 (S = 4, J <= 44: D blocks.1 forward)       tile 64 (m) x 128 (n), 4 waves as 1 x 4, wave tile 64 x 32
//   Cin = 64 (S = 1, J <= 12: D blocks.1 data gradient with its four output phases folded into 128 channels)
//                                                        tile 128 (m) x 128 (n), 4 waves as 2 x 2, wave tile 64 x 64
//   K-step 64 = 2 taps x 32 channels or 1 tap x 64 channels; weights [BM][64] per step, 128-byte rows, XOR-swizzled as in
//   conv_gemm.hip; window 35 KiB / 18 KiB + a weight ring of two stages (16 / 32 KiB): three workgroups per CU, so one
//   workgroup's window fill hides behind the others' multiply (a three-stage ring, two workgroups per CU, is 15-20 % slower).
//   Window image: row r of the window sits at its natural byte offset r*2*Cin, except that inside every aligned 256-byte line
//   (4 rows of 64 B or 2 rows of 128 B = 16 sixteen-byte slots) slot s is stored at s ^ (line & 15).  A B fragment reads 32
//   rows S*2*Cin bytes apart at one channel slot: without the XOR those are 16 reads of the same bank quarter per 16-lane group,
//   with it the 16 lanes of a group hit 16 different slots (S*2*Cin = 256 B: consecutive lines; 128 B: row pairs share a line
//   and differ in the slot's top bit).  The XOR is applied to the SOURCE address of the direct-to-LDS load (the window is one
//   contiguous range of global memory, so a wave's load still covers four whole 256-byte lines).
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void cw_glds16(uintptr_t src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}
// 16 bytes per lane: buffer resource + 32-bit vector byte offset + scalar byte offset -> LDS (plain function: see bglds16, conv_gemm.hip)
__device__ __forceinline__ void cw_bglds16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ uintptr_t cw_sel(bool ok, uintptr_t a, uintptr_t z) {
    return z ^ ((z ^ a) & (uintptr_t)(-(long long)ok));
}

#define CW_BN 128

template <int CIN, int S, int WM, int WN, int RING>
__global__ __launch_bounds__(256, RING == 2 ? 3 : 2) void conv_win_kernel(ConvParams p, int tiles_per_item, int total, int Ktot, int nsteps, int mode_flags) {
    constexpr int MI = 2;                        // 32-row MFMA tiles per wave along m
    constexpr int NI = CW_BN / (WN * 32);        // along n
    constexpr int BM = WM * MI * 32;
    constexpr int ROWB = CIN * 2;                // bytes per window row
    constexpr int TPS = 64 / CIN;                // taps per K-step
    constexpr int CPR = ROWB / 16;               // 16-byte chunks per row (4 or 8)
    constexpr int JMAX = CIN == 32 ? 44 : 12;
    constexpr int WROWS = (CW_BN - 1) * S + JMAX;  // window rows
    constexpr int WBYTES = ((WROWS * ROWB + 1023) / 1024) * 1024;
    constexpr int ABYTES = BM * 128;             // one stage of the weight ring
    constexpr int NA = ABYTES / (256 * 16);      // direct-to-LDS loads per thread per stage (2 or 4)
    constexpr int EPF = MI * 32 + 4;
    static_assert(WM * WN == 4, "four waves");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const Wl = smem;                       // window
    char* const Al = smem + WBYTES;              // [RING][BM][128 B]

    const int id = blockIdx.x;
    const int chunk = (total + 7) / 8;
    const int wk = (id & 7) * chunk + (id >> 3);  // consecutive tiles of an item side by side on one XCD: shared halo rows in its L2
    if (wk >= total) return;
    // wk -> (m tile, batch item, tile inside the item); m tiles fastest
    const int nM = (p.M + BM - 1) / BM;
    const int my = wk % nM;
    const int rest = wk / nM;
    const int b = rest / tiles_per_item, ti = rest - b * tiles_per_item;
    const int tl0 = ti * CW_BN;                  // first output step of the tile inside the item
    const int mbase = my * BM;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;
    const uintptr_t zero = (uintptr_t)ga_zero_page;

    // ---- window fill: rows R0 .. R0 + WROWS of item b (rows outside [0, Lin) are zero padding)
    const int R0 = (p.t0 + tl0) * S + p.off0 + p.phase_off0[0];
    {
        const uintptr_t xb = (uintptr_t)p.x + ((size_t)b * p.Lin) * ROWB;
        // LDS slot P (16-byte units) = line*16 + s' holds logical slot s = s' ^ (line & 15) of that line
        for (int P0 = wid * 64; P0 < WBYTES / 16; P0 += 256) {
            const int P = P0 + lane;
            const int line = P >> 4, sl = (P & 15) ^ (line & 15);
            const int row = line * (16 / CPR) + sl / CPR, ch = sl % CPR;
            const int R = R0 + row;
            const bool ok = row < WROWS && R >= 0 && R < p.Lin;
            cw_glds16(cw_sel(ok, xb + (size_t)(ok ? R : 0) * ROWB + ch * 16, zero), Wl + P0 * 16);
        }
    }
    // ---- weight ring: stage = K-step; piece q = i*256 + tid -> row q >> 3, slot q & 7 holding logical piece (q&7) ^ ((row>>1)&7)
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;
    uintptr_t abase[NA];
    const int apiece = (tid & 7) ^ ((tid >> 4) & 7);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = mbase + i * 32 + (tid >> 3);
        abase[i] = m < p.M ? (uintptr_t)(w + (size_t)m * Ktot + apiece * 8) : 0;
    }
    // weights by buffer loads: constant per-lane byte offset + the K-step as the scalar offset; rows past M and, in the last
    // K-step of a K that is no multiple of 64, pieces past Ktot get an offset beyond the buffer (the range check returns zeros)
    const bool wbuf = (mode_flags & 1) && (long)p.M * Ktot * 2 < (1L << 31) - 64;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, wbuf ? (int)((long)p.M * Ktot * 2) : 0, 0x00020000);
    unsigned avo[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) avo[i] = abase[i] ? (unsigned)(abase[i] - (uintptr_t)w) : 0x80000000u;
    auto issue_A = [&](int step, int buf) {
        const int k = step * 64 + apiece * 8;
        const bool kin = k < Ktot;
        char* const dst = Al + buf * ABYTES + wid * 1024;
        if (wbuf) {
#pragma unroll
            for (int i = 0; i < NA; ++i) cw_bglds16(rsA, dst + i * 4096, kin ? avo[i] : 0x80000000u, step * 128);
            return;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
            cw_glds16(cw_sel(kin && abase[i], abase[i] + (uintptr_t)step * 128, zero), dst + i * 4096);
    };
    issue_A(0, 0);
    if (RING == 3) issue_A(1, 1);

    const int m0_ = mbase + wm * (MI * 32) + (lane % (MI * 4)) * 8;
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias && m0_ < p.M) {
        const float4 b0 = *(const float4*)(p.bias + m0_), b1 = *(const float4*)(p.bias + m0_ + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
        bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    const int aswz = (r >> 1) & 7;
    const int aoff = (wm * MI * 32 + r) * 128;
    // B fragment of (ni, step, kk): window row (wn*NI*32 + ni*32 + r)*S + step*TPS + tap(kk), 16-byte chunk ch(kk) + ...
    int brow[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) brow[ni] = ((wn * NI + ni) * 32 + r) * S;

    for (int ks = 0; ks < nsteps; ++ks) {
        // stage ks has landed (and, at ks == 0, the window): all but the newest stage's loads are retired
        if (RING == 3 && ks + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // refill the buffer stage ks-1 was read from (everyone is past that read)
        if (ks + RING - 1 < nsteps) issue_A(ks + RING - 1, (ks + RING - 1) % RING);
        const char* Ab = Al + (ks % RING) * ABYTES + aoff;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int tap = CIN == 32 ? (kk >> 1) : 0;
            const int ch = CIN == 32 ? ((kk & 1) * 2 + h) : (kk * 2 + h);
            bf16x8 a[MI], bf[NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = *(const bf16x8*)(Ab + mi * 32 * 128 + (((kk * 2 + h) ^ aswz) << 4));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row = brow[ni] + ks * TPS + tap;
                const int line = row / (16 / CPR), sl = (row % (16 / CPR)) * CPR + ch;
                bf[ni] = *(const bf16x8*)(Wl + line * 256 + ((sl ^ (line & 15)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
    }

    // ---- epilogue: accumulators -> wave-private LDS rows [n][m] -> 16-byte, row-contiguous global accesses ----
    __syncthreads();
    if (mbase + wm * MI * 32 >= p.M) return;
    float* const stage = (float*)smem + wid * (32 * EPF);
    bf16_t* __restrict__ y = (bf16_t*)p.y;
    const bf16_t* __restrict__ res = (const bf16_t*)p.res;
    const bf16_t* __restrict__ ds = (const bf16_t*)p.dact_src;
    constexpr int LPR = MI * 4;         // lanes per staged row (8 output channels each)
    constexpr int RPP = 64 / LPR;       // rows per pass
    const int erow = lane / LPR, ec8 = (lane % LPR) * 8;
    const int m0 = mbase + wm * MI * 32 + ec8;
    const size_t item_elems = p.y_batch_elems > 0 ? (size_t)p.y_batch_elems : (size_t)p.LoutTot * p.M;
    const bool simple_act = p.post_act == GA_ACT_NONE || p.post_act == GA_ACT_LRELU;
    const float slope = p.post_act == GA_ACT_LRELU ? 0.2f : 1.f;
    const float fm_c = p.res_mode == GA_RES_FM_SIGN ? p.res_coef * p.res_coef_dev[0] : 0.f;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4*)(stage + r * EPF + mi * 32 + 8 * q + 4 * h) =
                    make_float4(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int row = ps * RPP + erow;
            const int tl = tl0 + (wn * NI + ni) * 32 + row;
            if (tl >= p.Lout || m0 >= p.M) continue;
            const int orow = (p.t0 + tl) * p.out_stride;
            if (orow < 0 || orow >= p.LoutTot) continue;
            const size_t in_item = (size_t)orow * p.M + m0;
            if (in_item + 8 > item_elems) continue;  // (folded phases: the last row may run past the item)
            const size_t o = (size_t)b * item_elems + in_item;
            float v[8];
            *(float4*)&v[0] = *(const float4*)(stage + row * EPF + ec8);
            *(float4*)&v[4] = *(const float4*)(stage + row * EPF + ec8 + 4);
            if (simple_act) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] += bv[i];
                    v[i] = v[i] > 0.f ? v[i] : slope * v[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = act_apply(p.post_act, v[i] + bv[i]);
            }
            if (p.res_mode == GA_RES_FM_SIGN) {  // (v + c * sign(fake - real)) * dact(fake)
                const bf16x8 d = *(const bf16x8*)(ds + o);
                const bf16x8 rr = *(const bf16x8*)(res + o);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float f = (float)d[i], q = (float)rr[i];
                    v[i] = (v[i] + (f > q ? fm_c : f < q ? -fm_c : 0.f)) * dact_apply(p.dact_mode, f);
                }
            } else {
                if (p.dact_mode) {
                    const bf16x8 d = *(const bf16x8*)(ds + o);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= dact_apply(p.dact_mode, (float)d[i]);
                }
                if (res) {
                    const bf16x8 d = *(const bf16x8*)(res + o);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += (float)d[i];
                }
            }
            if (p.accumulate) {
                const bf16x8 d = *(const bf16x8*)(y + o);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += (float)d[i];
            }
            bf16x8 ov;
#pragma unroll
            for (int i = 0; i < 8; ++i) ov[i] = (__bf16)v[i];
            *(bf16x8*)(y + o) = ov;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int ga_conv_gemm_supported(const ConvParams& p);

// 1: Cin = 32 / stride 4 form, 2: Cin = 64 / stride 1 form, 0: not a window-kernel problem
int ga_conv_win_supported(const ConvParams& p) {
    if (!ga_conv_gemm_supported(p) || p.nphase != 1 || p.pad_mode != 0 || p.tap_step != 1 || p.out_f32 || p.ksplit > 1) return 0;
    if (p.res_mode == GA_RES_ADD_THEN_ACT || p.y2) return 0;
    if ((long)p.B * p.Lin * p.Cin >= (1L << 31)) return 0;
    if (p.Cin == 32 && p.in_stride == 4 && p.J <= 44 && p.M <= 64 && p.M % 8 == 0) return 1;
    if (p.Cin == 64 && p.in_stride == 1 && p.J <= 12 && p.M >= 96 && p.M % 8 == 0) return 2;
    return 0;
}

int ga_conv_win_wanted(const ConvParams& p) {
    const char* e = getenv("GA_CONV_WIN");  // (read per call: in-process A/B)
    const int mode = e ? atoi(e) : 2;
    if (!mode) return 0;
    const int kind = ga_conv_win_supported(p);
    if (!kind) return 0;
    if (p.J * p.Cin < 512 || p.Lout < 2 * CW_BN) return 0;            // short K / short items: the gathered kernel's ground
    const long tiles = (long)p.B * ((p.Lout + CW_BN - 1) / CW_BN) * ((p.M + (kind == 1 ? 63 : 127)) / (kind == 1 ? 64 : 128));
    if (tiles < 1024) return 0;                                        // two workgroups per CU, a few rounds
    // measured at B = 128 (tools/win_check.py), two-stage weight ring: the stride-4 form 788-840 TFLOP/s against 514-550 for the
    // gathered kernel on D blocks.1 forward; the stride-1 form 625-630 against 520-540 on its folded data gradient (with the
    // three-stage ring, two workgroups per CU, it only tied).  GA_CONV_WIN=1 keeps the stride-1 form off, 0 both.
    return (kind == 1 || mode >= 2) ? kind : 0;
}

template <int CIN, int S, int WM, int WN, int RING>
static int launch_win(const ConvParams& p, hipStream_t s) {
    constexpr int BM = WM * 64;
    constexpr int JMAX = CIN == 32 ? 44 : 12;
    constexpr int WROWS = (CW_BN - 1) * S + JMAX;
    constexpr int WBYTES = ((WROWS * CIN * 2 + 1023) / 1024) * 1024;
    const int Ktot = p.J * p.Cin;
    const int nsteps = (Ktot + 63) / 64;
    const int tpi = (p.Lout + CW_BN - 1) / CW_BN;
    const long total = (long)p.B * tpi * ((p.M + BM - 1) / BM);
    if (total <= 0) return GA_OK;
    if (total > (1L << 30)) return GA_ERR_ARG;
    size_t lds = WBYTES + RING * BM * 128;
    const size_t epi = 4 * 32 * (64 + 4) * sizeof(float);
    if (lds < epi) lds = epi;
    auto kern = conv_win_kernel<CIN, S, WM, WN, RING>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GA_ERR_LAUNCH;
    const int grid = (int)((total + 7) / 8 * 8);
    const char* eb = getenv("GA_CONV_WIN_BUF");  // "0": flat-address weight loads (A/B aid)
    kern<<<grid, 256, lds, s>>>(p, tpi, (int)total, Ktot, nsteps, (eb && eb[0] == '0') ? 0 : 1);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

static int win_ring() {
    static int ring = -1;
    if (ring < 0) {
        const char* e = getenv("GA_CONV_WIN_RING");
        ring = e ? atoi(e) : 2;
    }
    return ring;
}

int ga_conv_win_impl(const ConvParams& p, hipStream_t s) {
    if (!p.x || !p.w || !p.y) return GA_ERR_ARG;
    switch (ga_conv_win_supported(p)) {
        // weight ring of two stages (three workgroups per CU) or three (two per CU): GA_CONV_WIN_RING, default 2 -- measured at
        // B = 128 on the stride-4 form: 667-700 -> 788-840 TFLOP/s; occupancy, not prefetch depth, was the limit
        case 1: return win_ring() == 3 ? launch_win<32, 4, 1, 4, 3>(p, s) : launch_win<32, 4, 1, 4, 2>(p, s);
        case 2: return win_ring() == 3 ? launch_win<64, 1, 2, 2, 3>(p, s) : launch_win<64, 1, 2, 2, 2>(p, s);
        default: return GA_ERR_ARG;
    }
}
