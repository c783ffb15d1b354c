// bf16 MFMA "streaming window" kernel for the thin MULTI-TAP forward layers of the SEANet decoder (gfx950): M <= 64 output
// channels, Cin in {32, 64}, 2 or 3 taps one row apart, ELU in front -- the k=3 residual convs and the k=2s transposed convs at
// 60000..120000 samples per item.  Like conv_stream.hip these layers are HBM-bound by their bytes (0.7-3 GB per launch for a few
// GFLOP), but there every lane fetches its own 16-byte fragment of every tap and applies ELU to it: each input element goes
// through the (quarter-rate) exponential J times, and the vector ALU, not memory, sets the pace (2.3-3.4 TB/s against the
// 5.3 TB/s of the single-tap layers; the data gradients of the same layers carry no ELU and run at 5 TB/s there).  Here:
//
//   * the packed weights live in registers as MFMA A fragments for the whole launch (as in conv_stream.hip)
//   * every wave owns a ring of RD window slots in LDS.  The NR-1+J input rows that a tile of NR = 32*NT output steps touches
//     are ONE contiguous run of channels-last x (a tile never crosses an item); they go global -> LDS once, with
//     global_load_lds_dwordx4 (no registers, every row fetched once), RD-1 tiles ahead of the multiply.  The ring is private
//     to its wave: no workgroup barrier anywhere, one counted s_waitcnt vmcnt per tile; eight waves per CU overlap one another's
//     ELU, LDS and MFMA phases.
//   * ELU (HF modeling_encodec.py:268,330,340) is applied ONCE per input element, in place in the window
//   * all J taps read their B fragments (v_mfma_f32_32x32x16_bf16: 8 consecutive k of one im2col row = 16 bytes of one row)
//     from the window at shifted rows.  Window image as in conv_win.hip: row r at its natural byte offset, 16-byte slots
//     XOR-swizzled inside every aligned 256-byte line (applied to the load's SOURCE address) so that the 16 lanes of a
//     ds_read_b128 group hit 16 different slots.
//   * epilogue: bias and activation on the accumulators, bf16 rows through a wave-private LDS stage, 16-byte row-contiguous stores
// The K order (tap-major, then channel), the accumulation chain and the rounding points are those of conv_stream_kernel:
// results are bit-identical (tools/swin_check.py, tests/test_gpu_ops.py).
#include "swin_common.h"
#include <stdlib.h>

template <int CIN, int J, int MT, int NT, int RD, int WPB>
struct SwinCfg {
    static constexpr int NR = NT * 32;                     // output steps per tile
    static constexpr int ROWB = CIN * 2;                   // bytes per window row
    static constexpr int CPR = ROWB / 16;                  // 16-byte chunks per row
    static constexpr int RPL = 16 / CPR;                   // rows per 256-byte line
    static constexpr int WROWS = NR - 1 + J;               // window rows
    static constexpr int WCH = WROWS * CPR;                // window chunks
    static constexpr int NLW = (WCH + 63) / 64;            // direct-to-LDS loads per tile (the last one may be partial)
    static constexpr int SLOT = ((WROWS * ROWB + 255) / 256) * 256;
    static constexpr int WMR = MT * 32;
    static constexpr int KS = J * CIN / 16;
    static constexpr int SROW = WMR * 2 + 8;               // bytes per staged output row (bf16)
    static constexpr int STAGE = ((32 * SROW + 255) / 256) * 256;
    static constexpr int WAVE_LDS = RD * SLOT + STAGE;
    static constexpr int LDS = WPB * WAVE_LDS;
    static_assert(CIN == 32 || CIN == 64, "Cin");
    static_assert((RD - 1) * NLW <= 63, "vmcnt range");
    static_assert(LDS <= 160 * 1024, "LDS");
};

template <int CIN, int J, int MT, int NT, int RD, int WPB>
__global__ __launch_bounds__(WPB * 64) void conv_swin_kernel(ConvParams p, int tiles_per_item, int total) {
    typedef SwinCfg<CIN, J, MT, NT, RD, WPB> C;
    constexpr int NR = C::NR, ROWB = C::ROWB, CPR = C::CPR, RPL = C::RPL, WROWS = C::WROWS, WCH = C::WCH, NLW = C::NLW;
    constexpr int WMR = C::WMR, SLOT = C::SLOT, KS = C::KS, SROW = C::SROW;
    constexpr int LPR = WMR / 8;    // lanes per staged row (8 output channels each)
    constexpr int RPP = 64 / LPR;   // rows per pass
    constexpr int NPS = 32 / RPP;   // passes per 32-row tile
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int M = p.M, Ktot = J * CIN;
    const int off0 = p.off0 + p.phase_off0[0];
    const uintptr_t zero = (uintptr_t)ga_zero_page;
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;
    char* const ring = smem + wid * C::WAVE_LDS;
    char* const stage = ring + RD * SLOT;

    // weights: A fragment (mt, ks) of lane (r, h) = w[mt*32 + r][ks*16 + h*8 .. +8]
    bf16x8 a[MT][KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mt * 32 + r;
            a[mt][ks] = *(gfrag_ptr_t)sw_sel(m < M, (uintptr_t)(w + (size_t)m * Ktot + ks * 16 + h * 8), zero);
        }
    // bias in the accumulator layout: element (q, i) of lane (r, h) is output channel mt*32 + 8q + 4h + i
    float bq[MT][4][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mt * 32 + 8 * q + 4 * h + i;
                bq[mt][q][i] = (p.bias && m < M) ? p.bias[m] : 0.f;
            }

    // window fill pattern of this lane: LDS slot P = i*64 + lane (16-byte units) = line*16 + s' holds logical slot
    // s = s' ^ (line & 15) of that line, i.e. chunk s % CPR of window row line*RPL + s / CPR
    int wrow[NLW], woff[NLW];
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int P = i * 64 + lane;
        const int line = P >> 4, sl = (P & 15) ^ (line & 15);
        wrow[i] = line * RPL + sl / CPR;
        woff[i] = wrow[i] * ROWB + (sl % CPR) * 16;
    }
    // B fragment (nt, ks) of lane (r, h): window row nt*32 + r + tap, chunk ch
    int foff[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int tap = (ks * 16) / CIN, ch = ((ks * 16) % CIN) / 8 + h;
            const int row = nt * 32 + r + tap;
            const int line = row / RPL, sl = (row % RPL) * CPR + ch;
            foff[nt][ks] = line * 256 + ((sl ^ (line & 15)) << 4);
        }

    bf16_t* __restrict__ y = (bf16_t*)p.y;
    const int erow = lane / LPR, m0 = (lane % LPR) * 8;
    const size_t item_elems = p.y_batch_elems > 0 ? (size_t)p.y_batch_elems : (size_t)p.LoutTot * M;
    const bool lrelu = p.post_act == GA_ACT_LRELU;
    const bool elu = p.pre_act == GA_ACT_ELU;

    // every load of tile t into ring slot `slot` (NLW instructions, always all of them: the wait below counts them)
    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
        const bool valid = t < total;
        const int tt = valid ? t : 0;
        const int b = tt / tiles_per_item, tl0 = (tt - b * tiles_per_item) * NR;
        const int R0 = p.t0 + tl0 + off0;
        const uintptr_t xb = (uintptr_t)p.x + (size_t)b * p.Lin * ROWB;
        char* const dst = ring + slot * SLOT;
        if (valid && R0 >= 0 && R0 + WROWS <= p.Lin) {  // every row of the window is a row of the item (wave-uniform)
            const uintptr_t base = xb + (size_t)R0 * ROWB;
#pragma unroll
            for (int i = 0; i < NLW; ++i)
                if ((i + 1) * 64 <= WCH || lane < WCH - i * 64) sw_glds16(base + woff[i], dst + i * 1024);
        } else {
#pragma unroll
            for (int i = 0; i < NLW; ++i) {
                int R = R0 + wrow[i];
                if (R < 0 && p.pad_mode == 1) R = -R;
                const bool ok = valid && R >= 0 && R < p.Lin;
                if ((i + 1) * 64 <= WCH || lane < WCH - i * 64)
                    sw_glds16(sw_sel(ok, xb + (size_t)(ok ? R : 0) * ROWB + (woff[i] - wrow[i] * ROWB), zero), dst + i * 1024);
            }
        }
    };

    const int nw = gridDim.x * WPB;
    int t = blockIdx.x * WPB + wid;
#pragma unroll
    for (int d = 0; d < RD - 1; ++d) issue(t + d * nw, d);

    for (int it = 0; t < total; t += nw, ++it) {
        asm volatile("" ::: "memory");
        issue(t + (RD - 1) * nw, (it + RD - 1) % RD);
        // tile t has landed: at most the (RD-1)*NLW loads issued after it are outstanding (loads retire in order; the stores of
        // earlier tiles in between only make the wait conservative)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 1) * NLW) : "memory");
        char* const Wl = ring + (it % RD) * SLOT;
        const int b = t / tiles_per_item, tl0 = (t - b * tiles_per_item) * NR;

        if (elu) {  // once per element, in place (each lane the chunks it loaded)
            sw_elu_window<NLW, WCH, 3>(Wl, lane);
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 xb = *(const bf16x8*)(Wl + foff[nt][ks]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][ks], xb, acc[mt], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = acc[mt][4 * q + i] + bq[mt][q][i];
                        if (lrelu) v[i] = v[i] > 0.f ? v[i] : 0.2f * v[i];
                    }
                    u32x2 o;
                    o.x = sw_pack_bf16(v[0], v[1]);
                    o.y = sw_pack_bf16(v[2], v[3]);
                    sw_lds_write8(stage + r * SROW + (mt * 32 + 8 * q + 4 * h) * 2, o);
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int row = ps * RPP + erow;
                const int tl = tl0 + nt * 32 + row;
                const int orow = (p.t0 + tl) * p.out_stride;
                const size_t in_item = (size_t)orow * M + m0;
                const bool ok = tl < p.Lout && m0 < M && orow >= 0 && orow < p.LoutTot && in_item + 8 <= item_elems;
                const u32x2 lo = *(const u32x2*)(stage + row * SROW + m0 * 2), hi = *(const u32x2*)(stage + row * SROW + m0 * 2 + 8);
                u32x4 ov;
                ov.x = lo.x; ov.y = lo.y; ov.z = hi.x; ov.w = hi.y;
                if (ok) *(u32x4*)(y + (size_t)b * item_elems + in_item) = ov;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (prefetches past the end read the zero page; let them land before the LDS goes)
}

// ------------------------------------------------------------------------------------------------
// Single output channel (the decoder's last conv 32 -> 1, k7, ELU in front, fp32 out; D blocks.0 data gradient 32 -> 1, k15):
// the same per-wave window ring and in-place ELU; the inner products of a tile's 64 output steps run on the matrix pipe as four
// chains of v_mfma_f32_16x16x32_bf16 (one K-step = one tap x 32 channels; B fragment = one 16-byte chunk of a window row).  Only
// one row of the 16-row A operand is a real filter: rows 0, 4, 8 and 12 all carry it, so that every lane group finds the result
// of its own steps in accumulator register 0, and one 64-lane store writes the tile.  15/16 of the MFMA work is padding and
// it is still far cheaper than the vector ALU: 224-480 multiply-adds per output cost thin.hip's one-thread-per-step kernel
// ~1800 cycles per 64 steps as FMAs and the same as packed v_dot2c_f32_bf16 (2.0-2.7 TB/s either way), 450-960 MFMA cycles here.
typedef float sw_f32x4 __attribute__((ext_vector_type(4)));

template <int CIN, int J, int RD, int WPB>
struct Swin1Cfg {
    static constexpr int NR = 64;                          // output steps per tile: one per lane
    static constexpr int ROWB = CIN * 2;
    static constexpr int CPR = ROWB / 16;
    static constexpr int RPL = 16 / CPR;
    static constexpr int WROWS = NR - 1 + J;
    static constexpr int WCH = WROWS * CPR;
    static constexpr int NLW = (WCH + 63) / 64;
    static constexpr int SLOT = ((WROWS * ROWB + 255) / 256) * 256;
    static constexpr int WAVE_LDS = RD * SLOT;
    static constexpr int LDS = WPB * WAVE_LDS;
    static_assert(CIN == 32, "one tap per 32-deep K-step");
    static_assert((RD - 1) * NLW <= 63, "vmcnt range");
    static_assert(LDS <= 160 * 1024, "LDS");
};

template <int CIN, int J, int RD, int WPB>
__global__ __launch_bounds__(WPB * 64) void conv_swin1_kernel(ConvParams p, int tiles_per_item, int total) {
    typedef Swin1Cfg<CIN, J, RD, WPB> C;
    constexpr int NR = C::NR, ROWB = C::ROWB, CPR = C::CPR, RPL = C::RPL, WROWS = C::WROWS, WCH = C::WCH, NLW = C::NLW, SLOT = C::SLOT;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int off0 = p.off0 + p.phase_off0[0];
    const uintptr_t zero = (uintptr_t)ga_zero_page;
    char* const ring = smem + wid * C::WAVE_LDS;
    const int n16 = lane & 15, g = lane >> 4;
    // A fragment of tap j: lane (m = n16, g) holds w[j*32 + 8g .. +8] when m is 0, 4, 8 or 12, zeros otherwise
    bf16x8 a[J];
#pragma unroll
    for (int j = 0; j < J; ++j)
        a[j] = *(gfrag_ptr_t)sw_sel((n16 & 3) == 0, (uintptr_t)p.w + (size_t)(j * CIN + g * 8) * 2, zero);

    int wrow[NLW], woff[NLW];
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int P = i * 64 + lane;
        const int line = P >> 4, sl = (P & 15) ^ (line & 15);
        wrow[i] = line * RPL + sl / CPR;
        woff[i] = wrow[i] * ROWB + (sl % CPR) * 16;
    }
    // B fragment of (chain c, tap j): window row 4*n16 + c + j, chunk g -- chain c holds the tile's steps 4n + c, so that the
    // offset depends on c + j only (J + 3 registers instead of 4J) and the 16 lanes of a group read 16 different 256-byte lines
    int xoff[J + 3];
#pragma unroll
    for (int sft = 0; sft < J + 3; ++sft) {
        const int row = 4 * n16 + sft;
        const int line = row / RPL, sl = (row % RPL) * CPR + g;
        xoff[sft] = line * 256 + ((sl ^ (line & 15)) << 4);
    }
    const float bias = p.bias ? p.bias[0] : 0.f;
    const bool elu = p.pre_act == GA_ACT_ELU;

    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
        const bool valid = t < total;
        const int tt = valid ? t : 0;
        const int b = tt / tiles_per_item, tl0 = (tt - b * tiles_per_item) * NR;
        const int R0 = p.t0 + tl0 + off0;
        const uintptr_t xb = (uintptr_t)p.x + (size_t)b * p.Lin * ROWB;
        char* const dst = ring + slot * SLOT;
        if (valid && R0 >= 0 && R0 + WROWS <= p.Lin) {
            const uintptr_t base = xb + (size_t)R0 * ROWB;
#pragma unroll
            for (int i = 0; i < NLW; ++i)
                if ((i + 1) * 64 <= WCH || lane < WCH - i * 64) sw_glds16(base + woff[i], dst + i * 1024);
        } else {
#pragma unroll
            for (int i = 0; i < NLW; ++i) {
                int R = R0 + wrow[i];
                if (R < 0 && p.pad_mode == 1) R = -R;
                const bool ok = valid && R >= 0 && R < p.Lin;
                if ((i + 1) * 64 <= WCH || lane < WCH - i * 64)
                    sw_glds16(sw_sel(ok, xb + (size_t)(ok ? R : 0) * ROWB + (woff[i] - wrow[i] * ROWB), zero), dst + i * 1024);
            }
        }
    };

    const int nw = gridDim.x * WPB;
    int t = blockIdx.x * WPB + wid;
#pragma unroll
    for (int d = 0; d < RD - 1; ++d) issue(t + d * nw, d);

    for (int it = 0; t < total; t += nw, ++it) {
        asm volatile("" ::: "memory");
        issue(t + (RD - 1) * nw, (it + RD - 1) % RD);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 1) * NLW) : "memory");
        char* const Wl = ring + (it % RD) * SLOT;
        const int b = t / tiles_per_item, tl0 = (t - b * tiles_per_item) * NR;
        if (elu) {
            sw_elu_window<NLW, WCH, 1>(Wl, lane);  // (sixteen waves per CU: 128 registers, none to spare for reading ahead)
            __builtin_amdgcn_wave_barrier();
        }
        sw_f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = sw_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], *(const bf16x8*)(Wl + xoff[c + j]), acc[c], 0, 0, 0);
        const float dot = g == 0 ? acc[0][0] : g == 1 ? acc[1][0] : g == 2 ? acc[2][0] : acc[3][0];  // lane (n16, g): step 4*n16 + g
        const int tl = tl0 + 4 * n16 + g;
        const int orow = (p.t0 + tl) * p.out_stride;
        if (tl < p.Lout && orow >= 0 && orow < p.LoutTot) {
            const size_t idx = (size_t)b * p.LoutTot + orow;
            const float v = act_apply(p.post_act, dot + bias);
            if (p.out_f32) ((float*)p.y)[idx] = v;
            else ((bf16_t*)p.y)[idx] = (bf16_t)v;
        }
        __builtin_amdgcn_wave_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------------------
int ga_conv_stream_supported(const ConvParams& p);

// single output channel: 0 or the instantiation (10, 11)
static int swin1_kind(const ConvParams& p) {
    if (p.M != 1 || p.Cin != 32 || p.in_stride != 1 || p.tap_step != 1 || p.nphase != 1 || p.ksplit > 1) return 0;
    if (p.res || p.dact_mode || p.dact_src || p.accumulate || p.B < 1 || p.Lout < 0) return 0;
    if (p.pre_act != GA_ACT_NONE && p.pre_act != GA_ACT_ELU) return 0;
    if (p.y_batch_elems != 0 || ((uintptr_t)p.x | (uintptr_t)p.w) & 15) return 0;
    if ((long)p.B * p.Lout > (1L << 30)) return 0;
    return p.J == 7 ? 10 : p.J == 15 ? 11 : 0;
}

// 0: not this kernel's problem, else the instantiation
int ga_conv_swin_supported(const ConvParams& p) {
    if (p.M == 1) return swin1_kind(p);
    if (!ga_conv_stream_supported(p)) return 0;
    if (p.res || p.dact_mode || p.dact_src || p.tap_step != 1 || p.in_stride != 1 || p.M % 8 || p.M < 8) return 0;
    if (p.Cin == 64 && p.J == 3 && p.M <= 32) return 1;
    if (p.Cin == 32 && p.J == 3 && p.M <= 32) return 2;
    if (p.Cin == 64 && p.J == 2 && p.M <= 64) return 3;
    return 0;
}

static int swin_mode() {
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("GA_CONV_SWIN");
        mode = e ? atoi(e) : 1;
    }
    return mode;
}

int ga_conv_swin_wanted(const ConvParams& p) {
    if (!swin_mode()) return 0;
    const int kind = ga_conv_swin_supported(p);
    if (!kind) return 0;
    if (kind >= 10) return (long)p.B * p.Lout >= (1L << 17) ? kind : 0;  // single output channel: long launches
    // what it buys is the exponential once per element instead of once per tap; without ELU the register-streaming kernel
    // already runs these shapes at 5 TB/s, and short launches are that kernel's ground too (many small groups)
    if (p.pre_act != GA_ACT_ELU || (long)p.B * p.Lout < (1L << 17)) return 0;
    return kind;
}

template <int CIN, int J, int RD, int WPB>
static int launch_swin1(const ConvParams& p, hipStream_t s) {
    typedef Swin1Cfg<CIN, J, RD, WPB> C;
    const int tpi = (p.Lout + C::NR - 1) / C::NR;
    const long total = (long)p.B * tpi;
    if (total <= 0) return GA_OK;
    auto kern = conv_swin1_kernel<CIN, J, RD, WPB>;
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return GA_ERR_LAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) return GA_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    long grid = (total + WPB - 1) / WPB;
    if (grid > 256) grid = 256;
    kern<<<(int)grid, WPB * 64, C::LDS, s>>>(p, tpi, (int)total);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

template <int CIN, int J, int MT, int NT, int RD, int WPB>
static int launch_swin(const ConvParams& p, hipStream_t s) {
    typedef SwinCfg<CIN, J, MT, NT, RD, WPB> C;
    const int tpi = (p.Lout + C::NR - 1) / C::NR;
    const long total = (long)p.B * tpi;
    if (total <= 0) return GA_OK;
    if (total > (1L << 30)) return GA_ERR_ARG;
    auto kern = conv_swin_kernel<CIN, J, MT, NT, RD, WPB>;
    static bool attr_set[16] = {};  // per device: the attribute is a property of the loaded function
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return GA_ERR_LAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) return GA_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const int per_cu = (160 * 1024) / C::LDS;
    long grid = (total + WPB - 1) / WPB;
    if (grid > 256L * per_cu) grid = 256L * per_cu;
    kern<<<(int)grid, WPB * 64, C::LDS, s>>>(p, tpi, (int)total);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_conv_swin_impl(const ConvParams& p, hipStream_t s) {
    if (!p.x || !p.w || !p.y) return GA_ERR_ARG;
    const int kind = ga_conv_swin_supported(p);
    if (!kind) return GA_ERR_ARG;
    if (p.Lout == 0) return GA_OK;
    // (swept at B = 192: one 32-row tile x three slots and four waves per CU are 8-25 % slower)
    switch (kind) {
        //                        CIN J MT NT RD WPB
        case 1: return launch_swin<64, 3, 1, 2, 2, 8>(p, s);
        case 2: return launch_swin<32, 3, 1, 4, 2, 8>(p, s);
        case 3: return launch_swin<64, 2, 2, 1, 3, 8>(p, s);
        case 10: return launch_swin1<32, 7, 2, 16>(p, s);
        case 11: return launch_swin1<32, 15, 2, 16>(p, s);
    }
    return GA_ERR_ARG;
}
