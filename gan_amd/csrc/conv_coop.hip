// bf16 MFMA kernel for the decoder's transposed conv 128 -> 64 (k8, stride 4) in its folded form -- 256 output channels (4 phases
// x 64), K = 2 taps x 128 channels, 15000 steps per item, 2.9M rows per launch -- and shapes like it: many rows, K of a few
// hundred, no epilogue operands.  On the gathered-GEMM kernel such a layer is four K-steps per 128x128 tile: every tile pays
// the index set-up, a first load round trip and the LDS-staged epilogue for 64 MFMAs per wave, and re-reads the 128 KB of
// weights (416-450 TFLOP/s, 2.4 TB/s).  Here the weights never move: a workgroup of WAVES waves holds them in registers, wave w
// the rows of output channels 32w .. 32w+31 as K/16 MFMA A fragments (64 registers), for the whole launch.  The input rows of a
// 64-step tile (one contiguous run, J - 1 rows of halo) go global -> LDS once per workgroup with global_load_lds_dwordx4, one
// tile ahead, and every wave reads its B fragments from that shared window (one 256-byte row per line, slots XOR-swizzled by
// the row); one workgroup barrier per tile; bias, rounding, wave-private stage, 16-byte stores of each wave's 64-byte piece of
// the 512-byte output rows.  Same K order (tap-major, then channel) and one accumulation chain per output: results equal the
// gathered kernel's up to its other K order (phase-/tap-fast) -- compared against the direct kernel in tests/test_gpu_ops.py.
#include "swin_common.h"
#include <stdlib.h>

typedef float cc_f32x4 __attribute__((ext_vector_type(4)));

#define CC_WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int CIN, int J, int WAVES>
struct CoopCfg {
    static constexpr int NT = 2, NR = 64;
    static constexpr int ROWB = CIN * 2;
    static_assert(ROWB == 256, "one window row per 256-byte line");
    static constexpr int WROWS = NR + J - 1;
    static constexpr int WCH = WROWS * 16;
    static constexpr int NLW = (WCH + WAVES * 64 - 1) / (WAVES * 64);  // loads per wave and tile
    static constexpr int SLOT = NLW * WAVES * 1024;
    static constexpr int KS = J * CIN / 16;
    static constexpr int YROW = 72, YSTAGE = 2304;
    static constexpr int RD = 2;
    static constexpr int LDS = RD * SLOT + WAVES * YSTAGE;
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
};

template <int CIN, int J, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void conv_coop_kernel(ConvParams p, int tiles_per_item, int total) {
    typedef CoopCfg<CIN, J, WAVES> G;
    constexpr int NT = G::NT, NR = G::NR, NLW = G::NLW, SLOT = G::SLOT, KS = G::KS, YROW = G::YROW, RD = G::RD, M = WAVES * 32;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uintptr_t zero = (uintptr_t)ga_zero_page;
    char* const ystage = smem + RD * SLOT + wid * G::YSTAGE;
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;
    bf16x8 a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = *(gfrag_ptr_t)(w + (size_t)(wid * 32 + r) * (J * CIN) + ks * 16 + h * 8);
    float bq[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) bq[q][i] = p.bias ? p.bias[wid * 32 + 8 * q + 4 * h + i] : 0.f;

    // cooperative fill: load i of wave w puts chunk P = (i*WAVES + w)*64 + lane at LDS offset 16 P: window row P >> 4, logical
    // chunk (P & 15) ^ (row & 15)
    int wsrc[NLW];
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int P = (i * WAVES + wid) * 64 + lane, row = P >> 4, ch = (P & 15) ^ (row & 15);
        wsrc[i] = P < G::WCH ? row * 256 + ch * 16 : -1;
    }
    const int off0 = p.off0 + p.phase_off0[0];
    const int Lin = p.Lin, Lout = p.Lout;
    bf16_t* __restrict__ y = (bf16_t*)p.y;
    const size_t item_elems = (size_t)p.LoutTot * M;

    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
        const bool valid = t < total;
        const int tt = valid ? t : 0;
        const int b = tt / tiles_per_item, R0 = (tt - b * tiles_per_item) * NR + off0;
        const uintptr_t xb = (uintptr_t)p.x + (size_t)b * Lin * 256;
        char* const dst = smem + slot * SLOT + wid * 1024;
#pragma unroll
        for (int i = 0; i < NLW; ++i) {
            const int R = R0 + (wsrc[i] >> 8);
            const bool ok = valid && wsrc[i] >= 0 && R >= 0 && R < Lin;  // zero padding on both sides
            sw_glds16(sw_sel(ok, xb + (size_t)(ok ? R : 0) * 256 + (wsrc[i] & 255), zero), dst + i * WAVES * 1024);
        }
    };

    int t = blockIdx.x;
    issue(t, 0);
    for (int it = 0; t < total; t += gridDim.x, ++it) {
        // this wave's part of tile t has landed (and its stores of the tile before are out); after the barrier everybody's
        // has, and nobody reads the other slot any more: the next tile goes there while this one is multiplied
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CC_WG_BARRIER();
        issue(t + gridDim.x, (it + 1) % RD);
        const char* const Wl = smem + (it % RD) * SLOT;
        const int b = t / tiles_per_item, tl0 = (t - b * tiles_per_item) * NR;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int tap = (ks * 16) / CIN, kk = (ks * 16 % CIN) / 16;
                const int row = nt * 32 + r + tap;
                const bf16x8 xb = *(const bf16x8*)(Wl + row * 256 + (((2 * kk + h) ^ (row & 15)) << 4));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], xb, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x2 o;
                o.x = sw_pack_bf16(acc[4 * q] + bq[q][0], acc[4 * q + 1] + bq[q][1]);
                o.y = sw_pack_bf16(acc[4 * q + 2] + bq[q][2], acc[4 * q + 3] + bq[q][3]);
                sw_lds_write8(ystage + r * YROW + (8 * q + 4 * h) * 2, o);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int row = ps * 16 + (lane >> 2), m0 = (lane & 3) * 8;
                const int tl = tl0 + nt * 32 + row;
                const u32x2 lo = *(const u32x2*)(ystage + row * YROW + m0 * 2), hi = *(const u32x2*)(ystage + row * YROW + m0 * 2 + 8);
                u32x4 ov;
                ov.x = lo.x; ov.y = lo.y; ov.z = hi.x; ov.w = hi.y;
                if (tl < Lout) *(u32x4*)(y + (size_t)b * item_elems + (size_t)tl * M + wid * 32 + m0) = ov;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
int ga_conv_gemm_supported(const ConvParams& p);

int ga_conv_coop_supported(const ConvParams& p) {
    if (!ga_conv_gemm_supported(p)) return 0;
    if (p.Cin != 128 || p.J != 2 || p.M != 256 || p.in_stride != 1 || p.tap_step != 1 || p.nphase != 1 || p.out_stride != 1 || p.t0 != 0) return 0;
    if (p.pad_mode != 0 || p.pre_act || p.post_act || p.res || p.dact_mode || p.dact_src || p.accumulate || p.out_f32 || p.y2) return 0;
    if (p.res_mode != GA_RES_ADD || p.ksplit > 1 || p.Lout != p.LoutTot) return 0;
    if (p.y_batch_elems != 0 && (long)p.y_batch_elems != (long)p.LoutTot * p.M) return 0;
    if ((long)p.B * ((p.Lout + 63) / 64) > (1L << 30)) return 0;
    return 1;
}

int ga_conv_coop_wanted(const ConvParams& p) {
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("GA_CONV_COOP");
        mode = e ? atoi(e) : 1;
    }
    // (short launches stay with the gathered kernel: this one wants about a tile per workgroup or more to amortise its weight loads)
    return mode && ga_conv_coop_supported(p) && (long)p.B * p.Lout >= (1L << 15);
}

int ga_conv_coop_impl(const ConvParams& p, hipStream_t s) {
    if (!ga_conv_coop_supported(p) || !p.x || !p.w || !p.y) return GA_ERR_ARG;
    if (p.Lout == 0) return GA_OK;
    typedef CoopCfg<128, 2, 8> G;
    const int tpi = (p.Lout + G::NR - 1) / G::NR;
    const long total = (long)p.B * tpi;
    auto kern = conv_coop_kernel<128, 2, 8>;
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return GA_ERR_LAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) return GA_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const int grid = (int)(total < 512 ? total : 512);  // two workgroups per CU
    kern<<<grid, 8 * 64, G::LDS, s>>>(p, tpi, (int)total);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

// ------------------------------------------------------------------------------------------------
// The closing sum of the 256-channel residual unit in one pass: y = [ELU]( W_sc x + W3 t + b_sc + b3 ), x [rows][256], t [rows][128],
// 256 output channels (HF modeling_encodec.py:333-343: shortcut(x) + block's last conv k1).  Both convs are k1, so the problem is
// row-wise -- one GEMM over the concatenated K = 256 + 128 -- and has no item boundaries.  As two gathered-GEMM launches the
// shortcut was written and read back and each launch was two / four K-steps per tile.  Same scheme as conv_coop_kernel: the 192 KB
// of [W_sc | W3] live in the registers of an eight-wave workgroup (96 per wave), the 64 rows of x and t of a tile go global -> LDS
// once per workgroup one tile ahead (exactly six loads per wave), one barrier per tile.
struct DualCfg {
    static constexpr int CA = 256, CB = 128, M = 256, WAVES = 8, NR = 64, NT = 2;
    static constexpr int XA_BYTES = NR * CA * 2, XB_BYTES = NR * CB * 2;   // 32 KB + 16 KB
    static constexpr int SLOT = XA_BYTES + XB_BYTES;
    static constexpr int NLA = XA_BYTES / (WAVES * 1024), NLB = XB_BYTES / (WAVES * 1024);  // 4 + 2 loads per wave
    static constexpr int KSA = CA / 16, KSB = CB / 16;
    static constexpr int YROW = 72, YSTAGE = 2304;
    static constexpr int RD = 2;
    static constexpr int LDS = RD * SLOT + WAVES * YSTAGE;
    static_assert(LDS <= 160 * 1024, "LDS");
};

template <bool POST_ELU>
__global__ __launch_bounds__(512) void conv_dual_k1_kernel(ga_dual_k1_params p, int total) {
    typedef DualCfg G;
    constexpr int NR = G::NR, NT = G::NT, SLOT = G::SLOT, YROW = G::YROW, RD = G::RD, M = G::M, WAVES = G::WAVES;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uintptr_t zero = (uintptr_t)ga_zero_page;
    char* const ystage = smem + RD * SLOT + wid * G::YSTAGE;
    const bf16_t* __restrict__ wa = (const bf16_t*)p.wa;
    const bf16_t* __restrict__ wb = (const bf16_t*)p.wb;
    bf16x8 aa[G::KSA], ab[G::KSB];
#pragma unroll
    for (int ks = 0; ks < G::KSA; ++ks) aa[ks] = *(gfrag_ptr_t)(wa + (size_t)(wid * 32 + r) * G::CA + ks * 16 + h * 8);
#pragma unroll
    for (int ks = 0; ks < G::KSB; ++ks) ab[ks] = *(gfrag_ptr_t)(wb + (size_t)(wid * 32 + r) * G::CB + ks * 16 + h * 8);
    float bq[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = wid * 32 + 8 * q + 4 * h + i;
            bq[q][i] = (p.ba ? p.ba[m] : 0.f) + (p.bb ? p.bb[m] : 0.f);
        }
    // fill: load i of wave w covers LDS bytes [(i*WAVES + w)*1024, +1024) of a region = lines 4(i*WAVES + w) .. +3; lane's slot holds
    // logical slot (P & 15) ^ (line & 15) of line P >> 4.  x rows are two lines (512 bytes), t rows one.
    int srcA[G::NLA], srcB[G::NLB];  // byte offset inside the tile's rows of the source
#pragma unroll
    for (int i = 0; i < G::NLA; ++i) {
        const int P = (i * WAVES + wid) * 64 + lane, line = P >> 4, sl = (P & 15) ^ (line & 15);
        srcA[i] = line * 256 + sl * 16;
    }
#pragma unroll
    for (int i = 0; i < G::NLB; ++i) {
        const int P = (i * WAVES + wid) * 64 + lane, line = P >> 4, sl = (P & 15) ^ (line & 15);
        srcB[i] = line * 256 + sl * 16;
    }
    const long rows = p.rows;
    bf16_t* __restrict__ y = (bf16_t*)p.y;

    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
        const long row0 = (long)t * NR;
        const long left = t < total ? rows - row0 : 0;  // rows of this tile that exist
        const uintptr_t xa = (uintptr_t)p.xa + (size_t)(t < total ? row0 : 0) * (G::CA * 2);
        const uintptr_t xb = (uintptr_t)p.xb + (size_t)(t < total ? row0 : 0) * (G::CB * 2);
        char* const dst = smem + slot * SLOT + wid * 1024;
#pragma unroll
        for (int i = 0; i < G::NLA; ++i)
            sw_glds16(sw_sel((srcA[i] >> 9) < left, xa + srcA[i], zero), dst + i * WAVES * 1024);
#pragma unroll
        for (int i = 0; i < G::NLB; ++i)
            sw_glds16(sw_sel((srcB[i] >> 8) < left, xb + srcB[i], zero), dst + G::XA_BYTES + i * WAVES * 1024);
    };

    int t = blockIdx.x;
    issue(t, 0);
    for (int it = 0; t < total; t += gridDim.x, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CC_WG_BARRIER();
        issue(t + gridDim.x, (it + 1) % RD);
        const char* const Wa = smem + (it % RD) * SLOT;
        const char* const Wb = Wa + G::XA_BYTES;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const int row = nt * 32 + r;
#pragma unroll
            for (int ks = 0; ks < G::KSA; ++ks) {  // chunk 2 ks + h of x row `row`: line 2 row + (chunk >> 4)
                const int c = 2 * ks, line = 2 * row + (c >> 4);
                const bf16x8 xb = *(const bf16x8*)(Wa + line * 256 + ((((c & 15) + h) ^ (line & 15)) << 4));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[ks], xb, acc, 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < G::KSB; ++ks) {
                const bf16x8 xb = *(const bf16x8*)(Wb + row * 256 + (((2 * ks + h) ^ (row & 15)) << 4));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[ks], xb, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[4 * q + i] + bq[q][i];
                    if (POST_ELU) v[i] = elu_bf16_grade(v[i]);
                }
                u32x2 o;
                o.x = sw_pack_bf16(v[0], v[1]);
                o.y = sw_pack_bf16(v[2], v[3]);
                sw_lds_write8(ystage + r * YROW + (8 * q + 4 * h) * 2, o);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int rr = ps * 16 + (lane >> 2), m0 = (lane & 3) * 8;
                const long grow = (long)t * NR + nt * 32 + rr;
                const u32x2 lo = *(const u32x2*)(ystage + rr * YROW + m0 * 2), hi = *(const u32x2*)(ystage + rr * YROW + m0 * 2 + 8);
                u32x4 ov;
                ov.x = lo.x; ov.y = lo.y; ov.z = hi.x; ov.w = hi.y;
                if (grow < rows) *(u32x4*)(y + (size_t)grow * M + wid * 32 + m0) = ov;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

extern "C" int ga_conv_dual_k1(const ga_dual_k1_params* hp, int dtype, hipStream_t s) {
    if (!hp) return GA_ERR_ARG;
    const ga_dual_k1_params& p = *hp;
    if (!p.xa || !p.wa || !p.xb || !p.wb || !p.y || p.rows < 0) return GA_ERR_ARG;
    if (dtype != GA_DTYPE_BF16 || p.Ca != 256 || p.Cb != 128 || p.M != 256) return GA_ERR_UNSUPPORTED;
    if (((uintptr_t)p.xa | (uintptr_t)p.wa | (uintptr_t)p.xb | (uintptr_t)p.wb | (uintptr_t)p.y) & 15) return GA_ERR_ALIGN;
    if (p.rows == 0) return GA_OK;
    const long total = (p.rows + DualCfg::NR - 1) / DualCfg::NR;
    if (total > (1L << 30)) return GA_ERR_UNSUPPORTED;
    static bool attr_set[16][2] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return GA_ERR_LAUNCH;
    const int v = p.post_elu ? 1 : 0;
    if (!attr_set[dev][v]) {
        const void* k = v ? (const void*)conv_dual_k1_kernel<true> : (const void*)conv_dual_k1_kernel<false>;
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, DualCfg::LDS) != hipSuccess) return GA_ERR_LAUNCH;
        attr_set[dev][v] = true;
    }
    const int grid = (int)(total < 256 ? total : 256);  // one workgroup per CU (its LDS: 114 KB)
    if (v) conv_dual_k1_kernel<true><<<grid, 512, DualCfg::LDS, s>>>(p, (int)total);
    else conv_dual_k1_kernel<false><<<grid, 512, DualCfg::LDS, s>>>(p, (int)total);
    GA_CHECK_LAUNCH();
    return GA_OK;
}
