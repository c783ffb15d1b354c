// One SEANet residual unit of the frozen decoder as ONE pass over its input (gfx950, bf16):
//     y = W_sc x + b_sc + W3 ELU(W1 * ELU(x) + b1) + b3         (HF modeling_encodec.py:320-343; C -> C/2 -> C, causal k3)
// for the thin end of the decoder (C = 32 at 120000 steps per item, C = 64 at 60000).  As three ga_conv launches the unit reads
// x twice, writes and reads the shortcut and the hidden tensor, and writes y: 4.5 C-wide passes for 2; every launch is HBM-bound
// (conv_stream.hip / conv_swin.hip), so the unit's time is its bytes.  Here, per wave (the ring of conv_swin.hip: windows of
// NR + 2 input rows go global -> LDS with global_load_lds_dwordx4 one tile ahead, private to the wave, no workgroup barrier in
// the loop, one counted vmcnt per tile):
//   1. shortcut: W_sc x on the RAW window (the B fragments of the k3 conv's last tap, before the ELU) into the accumulators
//      that the unit's output is summed in;
//   2. ELU once per element, in place in the window;
//   3. per 32 steps: hidden = W1 * window (3 taps at shifted rows) + b1, rounded to bf16 (the value the three-launch form
//      stores), ELU, into a wave-private LDS stage laid out as MFMA B fragments; W3 * stage accumulates onto the shortcut;
//      the stage's rows go to t_act (when the caller wants the backward's ELU'(t) source); bias, one rounding, rows through
//      the stage, 16-byte row-contiguous stores.
// All three weight matrices live in registers as MFMA A fragments (36 / 96 registers), biases in LDS.
#include "swin_common.h"
#include <stdlib.h>

typedef float ru_f32x4 __attribute__((ext_vector_type(4)));

template <int C, int NT, int RD, int WPB>
struct RuCfg {
    static constexpr int H = C / 2;                        // hidden channels
    static constexpr int NR = NT * 32;                     // output steps per tile
    static constexpr int ROWB = C * 2;
    static constexpr int CPR = ROWB / 16;
    static constexpr int RPL = 16 / CPR;
    static constexpr int WROWS = NR + 2;
    static constexpr int WCH = WROWS * CPR;
    static constexpr int NLW = (WCH + 63) / 64;
    static constexpr int SLOT = ((WROWS * ROWB + 255) / 256) * 256;
    static constexpr int MT = C / 32;
    static constexpr int KSC = C / 16;                     // K-steps of one tap (and of the shortcut)
    static constexpr int KS1 = 3 * KSC;
    static constexpr int KS3 = H / 16;
    static constexpr int TROW = H * 2 + 16;                // staged hidden row: 48 / 80 bytes, 16 lanes of a b128 read hit 16 different slots
    static constexpr int YROW = 32 * 2 + 8;                // staged output half-row (32 channels)
    static constexpr int STAGE = ((32 * (TROW > YROW ? TROW : YROW) + 255) / 256) * 256;
    static constexpr int NTB = C == 64 ? 1 : 2;            // 32 rows further = 16 / 8 lines further: the swizzle repeats every NTB sub-tiles
    static constexpr int WAVE_LDS = RD * SLOT + STAGE;
    static constexpr int BIAS_OFF = WPB * WAVE_LDS;
    static constexpr int LDS = BIAS_OFF + 512;             // [C] b_sc + b3, [H] b1 (fp32)
    static_assert(C == 32 || C == 64, "C");
    static_assert(NT % NTB == 0, "NT");
    static_assert((RD - 1) * NLW <= 63, "vmcnt range");
    static_assert(WAVE_LDS % 256 == 0, "lines");
    static_assert(LDS <= 160 * 1024, "LDS");
};

typedef float ru_f32x4b __attribute__((ext_vector_type(4)));

// TAIL (C = 32 only): the decoder's last layer behind the unit -- wav = conv_k7(ELU(y)) + b (HF modeling_encodec.py: the decoder's
// closing nn.ELU + EncodecConv1d(32, 1, 7)) -- in the same pass, for callers that need no gradient: y is never written.  A tile's
// 128 rows of y (bias, one rounding to bf16: the bits ga_resunit_fwd would store; ELU) go back into the tile's own, by then
// dead, input window, in the window layout of conv_swin1_kernel, whose four MFMA chains then produce 122 samples: the tile
// stride is NR - 6 and every tile recomputes the six rows of y in front of its samples (the first tile of an item reflects them).
template <int C, int NT, int RD, int WPB, bool TAIL>
__global__ __launch_bounds__(WPB * 64) void resunit_fwd_kernel(ga_resunit_params p, int tiles_per_item, int total) {
    typedef RuCfg<C, NT, RD, WPB> G;
    static_assert(!TAIL || (C == 32 && NT == 4), "tail form");
    constexpr int HALO = TAIL ? 6 : 0, NRO = G::NR - HALO;  // rows of y recomputed in front of a tile; tile stride
    constexpr int H = G::H, ROWB = G::ROWB, CPR = G::CPR, RPL = G::RPL, WROWS = G::WROWS, WCH = G::WCH, NLW = G::NLW;
    constexpr int SLOT = G::SLOT, MT = G::MT, KSC = G::KSC, KS1 = G::KS1, KS3 = G::KS3, TROW = G::TROW, YROW = G::YROW, NTB = G::NTB;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uintptr_t zero = (uintptr_t)ga_zero_page;
    char* const ring = smem + wid * G::WAVE_LDS;
    char* const stage = ring + RD * SLOT;
    float* const bsum = (float*)(smem + G::BIAS_OFF);
    float* const b1s = bsum + C;
    for (int i = tid; i < C + H; i += WPB * 64) bsum[i] = i < C ? p.b_sc[i] + p.b3[i] : p.b1[i - C];
    __syncthreads();

    // A fragment (mt, ks) of lane (r, h) of a row-major [M][K] matrix: w[mt*32 + r][ks*16 + h*8 .. +8]
    const bf16_t* __restrict__ wsc = (const bf16_t*)p.w_sc;
    const bf16_t* __restrict__ w1 = (const bf16_t*)p.w1;
    const bf16_t* __restrict__ w3 = (const bf16_t*)p.w3;
    bf16x8 asc[MT][KSC], a1[KS1], a3[MT][KS3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) asc[mt][ks] = *(gfrag_ptr_t)(wsc + (size_t)(mt * 32 + r) * C + ks * 16 + h * 8);
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) a3[mt][ks] = *(gfrag_ptr_t)(w3 + (size_t)(mt * 32 + r) * H + ks * 16 + h * 8);
    }
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) a1[ks] = *(gfrag_ptr_t)sw_sel(r < H, (uintptr_t)(w1 + (size_t)r * (3 * C) + ks * 16 + h * 8), zero);

    // window fill pattern (conv_swin.hip): LDS slot P = i*64 + lane holds logical slot (P & 15) ^ (line & 15) of line P >> 4
    // (load i + 4 is 16 lines further: the same swizzle, 4096 bytes and 16*RPL rows on)
    constexpr int NLD = NLW < 4 ? NLW : 4;
    int wrow[NLD], woff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int P = i * 64 + lane;
        const int line = P >> 4, sl = (P & 15) ^ (line & 15);
        wrow[i] = line * RPL + sl / CPR;
        woff[i] = wrow[i] * ROWB + (sl % CPR) * 16;
    }
    // B fragment (nt, ks) of lane (r, h): window row nt*32 + r + tap, chunk ch; sub-tile nt + NTB sits 4096 bytes further
    int foff[NTB][KS1];
#pragma unroll
    for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const int tap = (ks * 16) / C, ch = ((ks * 16) % C) / 8 + h;
            const int row = nb * 32 + r + tap;
            const int line = row / RPL, sl = (row % RPL) * CPR + ch;
            foff[nb][ks] = line * 256 + ((sl ^ (line & 15)) << 4);
        }

    bf16_t* __restrict__ y = (bf16_t*)p.y;
    bf16_t* __restrict__ tact = (bf16_t*)p.t_act;
    const int L = p.L;

    // TAIL: A fragments of the k7 filter (conv_swin1_kernel: rows 0/4/8/12 of the 16-row operand carry it), fragment offsets of
    // chain c, tap j (window row 4*n16 + c + j, chunk g), where this lane's four y channels of a row go (chunk q, + 8h bytes)
    constexpr int JL = 7;
    const int n16 = lane & 15, g4 = lane >> 4;
    bf16x8 a7[TAIL ? JL : 1];
    int xo7[TAIL ? JL + 3 : 1], yoff[TAIL ? 2 : 1][TAIL ? 4 : 1];
    float blast = 0.f;
    if (TAIL) {
#pragma unroll
        for (int j = 0; j < JL; ++j) a7[j] = *(gfrag_ptr_t)sw_sel((n16 & 3) == 0, (uintptr_t)p.w_last + (size_t)(j * C + g4 * 8) * 2, zero);
#pragma unroll
        for (int sft = 0; sft < JL + 3; ++sft) {
            const int row = 4 * n16 + sft;
            const int line = row / RPL, sl = (row % RPL) * CPR + g4;
            xo7[sft] = line * 256 + ((sl ^ (line & 15)) << 4);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = nb * 32 + r;
                const int line = row / RPL, sl = (row % RPL) * CPR + q;
                yoff[nb][q] = line * 256 + ((sl ^ (line & 15)) << 4) + 8 * h;
            }
        blast = p.b_last ? p.b_last[0] : 0.f;
    }

    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
        const bool valid = t < total;
        const int tt = valid ? t : 0;
        const int b = tt / tiles_per_item, tl0 = (tt - b * tiles_per_item) * NRO - HALO;
        const int R0 = tl0 - 2;
        const uintptr_t xb = (uintptr_t)p.x + (size_t)b * L * ROWB;
        char* const dst = ring + slot * SLOT;
        if (valid && R0 >= 0 && R0 + WROWS <= L) {  // every row of the window is a row of the item (wave-uniform)
            const uintptr_t base = xb + (size_t)R0 * ROWB;
#pragma unroll
            for (int i = 0; i < NLW; ++i)
                if ((i + 1) * 64 <= WCH || lane < WCH - i * 64) sw_glds16(base + woff[i & 3] + (i >> 2) * 4096, dst + i * 1024);
        } else {
#pragma unroll
            for (int i = 0; i < NLW; ++i) {
                int R = R0 + wrow[i & 3] + (i >> 2) * 16 * RPL;
                if (R < 0) R = -R;  // reflect-left padding of the causal conv (HF modeling_encodec.py:157-176)
                const bool ok = valid && R < L;
                if ((i + 1) * 64 <= WCH || lane < WCH - i * 64)
                    sw_glds16(sw_sel(ok, xb + (size_t)(ok ? R : 0) * ROWB + (woff[i & 3] - wrow[i & 3] * ROWB), zero), dst + i * 1024);
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
        const int b = t / tiles_per_item, tl0 = (t - b * tiles_per_item) * NRO - HALO;  // item row of the tile's first row of y

        // 1. shortcut on the raw rows (tap 2 of the window = the step itself)
        f32x16 acc[NT][MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks) {
                const bf16x8 xb = *(const bf16x8*)(Wl + foff[nt % NTB][2 * KSC + ks] + (nt / NTB) * 4096);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asc[mt][ks], xb, acc[nt][mt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // 2. ELU once per element, in place (each lane the chunks it loaded)
        sw_elu_window<NLW, WCH, MT == 1 ? 3 : 1>(Wl, lane);
        __builtin_amdgcn_wave_barrier();

#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // 3a. hidden rows of these 32 steps
            f32x16 ta;
#pragma unroll
            for (int i = 0; i < 16; ++i) ta[i] = 0.f;
            // (the bias vectors are read from LDS ahead of the MFMAs that hide the round trip: read where they are used, every
            // group of exponentials waited for its own LDS read -- the asm stores keep the compiler from hoisting them itself)
            // (C = 64 has no registers to spare for this: 250 of 256)
            ru_f32x4 bb1[H / 8];
            if (MT == 1) {
#pragma unroll
                for (int q = 0; q < H / 8; ++q) bb1[q] = *(const ru_f32x4*)(b1s + 8 * q + 4 * h);
            }
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const bf16x8 xb = *(const bf16x8*)(Wl + foff[nt % NTB][ks] + (nt / NTB) * 4096);
                ta = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ks], xb, ta, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < H / 8; ++q) {  // accumulator (q, i) of lane (r, h): hidden channel 8q + 4h + i of step r
                const ru_f32x4 bb = MT == 1 ? bb1[q] : *(const ru_f32x4*)(b1s + 8 * q + 4 * h);
                u32x2 o;
                o.x = sw_elu2(sw_pack_bf16(ta[4 * q] + bb[0], ta[4 * q + 1] + bb[1]));
                o.y = sw_elu2(sw_pack_bf16(ta[4 * q + 2] + bb[2], ta[4 * q + 3] + bb[3]));
                sw_lds_write8(stage + r * TROW + (8 * q + 4 * h) * 2, o);
            }
            __builtin_amdgcn_wave_barrier();
            // 3b. second conv onto the shortcut
            constexpr bool EARLY = !TAIL && MT == 1;  // (C = 64 has no registers to spare: its vectors are read per 32 channels below)
            ru_f32x4 bbs[4];
            if (EARLY) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bbs[q] = *(const ru_f32x4*)(bsum + 8 * q + 4 * h);
            }
#pragma unroll
            for (int ks = 0; ks < KS3; ++ks) {
                const bf16x8 tb = *(const bf16x8*)(stage + r * TROW + (ks * 16 + h * 8) * 2);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[mt][ks], tb, acc[nt][mt], 0, 0, 0);
            }
            if (tact) {  // ELU(hidden) for the backward pass: rows of H channels, 16 bytes per lane
                constexpr int LPRT = H / 8, RPPT = 64 / LPRT;
#pragma unroll
                for (int ps = 0; ps < 32 / RPPT; ++ps) {
                    const int row = ps * RPPT + lane / LPRT, c0 = (lane % LPRT) * 8;
                    const int tl = tl0 + nt * 32 + row;
                    const u32x4 v = *(const u32x4*)(stage + row * TROW + c0 * 2);
                    if (tl < L) *(u32x4*)(tact + ((size_t)b * L + tl) * H + c0) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (TAIL) continue;
            // 3c. bias, one rounding, rows through the stage (32 channels at a time)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const ru_f32x4 bb = EARLY ? bbs[q] : *(const ru_f32x4*)(bsum + mt * 32 + 8 * q + 4 * h);
                    u32x2 o;
                    o.x = sw_pack_bf16(acc[nt][mt][4 * q] + bb[0], acc[nt][mt][4 * q + 1] + bb[1]);
                    o.y = sw_pack_bf16(acc[nt][mt][4 * q + 2] + bb[2], acc[nt][mt][4 * q + 3] + bb[3]);
                    sw_lds_write8(stage + r * YROW + (8 * q + 4 * h) * 2, o);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int row = ps * 16 + (lane >> 2), m0 = (lane & 3) * 8;
                    const int tl = tl0 + nt * 32 + row;
                    const u32x2 lo = *(const u32x2*)(stage + row * YROW + m0 * 2), hi = *(const u32x2*)(stage + row * YROW + m0 * 2 + 8);
                    u32x4 ov;
                    ov.x = lo.x; ov.y = lo.y; ov.z = hi.x; ov.w = hi.y;
                    if (tl < L) *(u32x4*)(y + ((size_t)b * L + tl) * C + mt * 32 + m0) = ov;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (TAIL) {
            // every read of the input window is done: ELU(y) takes its place (row w of the window = row tl0 + w of the item)
            ru_f32x4 bby[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bby[q] = *(const ru_f32x4*)(bsum + 8 * q + 4 * h);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const ru_f32x4 bb = bby[q];
                    u32x2 o;
                    o.x = sw_elu2(sw_pack_bf16(acc[nt][0][4 * q] + bb[0], acc[nt][0][4 * q + 1] + bb[1]));
                    o.y = sw_elu2(sw_pack_bf16(acc[nt][0][4 * q + 2] + bb[2], acc[nt][0][4 * q + 3] + bb[3]));
                    sw_lds_write8(Wl + yoff[nt & 1][q] + (nt >> 1) * 4096, o);
                }
            __builtin_amdgcn_wave_barrier();
            if (tl0 < 0) {  // first tile of an item: rows -6 .. -1 of the causal conv's reflected pad are rows 6 .. 1
                const int w = lane >> 2, c = lane & 3;  // lanes 0..23: chunk c of window row w <- window row 12 - w
                if (w < HALO) {
                    const int rs = 2 * HALO - w;
                    const int ls = rs / RPL, ss = (rs % RPL) * CPR + c, ld = w / RPL, sd = (w % RPL) * CPR + c;
                    const u32x4 v = *(const u32x4*)(Wl + ls * 256 + ((ss ^ (ls & 15)) << 4));
                    sw_lds_write16(Wl + ld * 256 + ((sd ^ (ld & 15)) << 4), v);
                }
                __builtin_amdgcn_wave_barrier();
            }
            float* __restrict__ wav = p.wav;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {  // 64 samples per pass; sample s of the tile reads window rows s .. s + 6
                ru_f32x4b ch[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) ch[c] = ru_f32x4b{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < JL; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        ch[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a7[j], *(const bf16x8*)(Wl + xo7[c + j] + sub * 4096), ch[c], 0, 0, 0);
                const float dot = g4 == 0 ? ch[0][0] : g4 == 1 ? ch[1][0] : g4 == 2 ? ch[2][0] : ch[3][0];
                const int sidx = sub * 64 + 4 * n16 + g4;
                const int n = tl0 + HALO + sidx;
                if (sidx < NRO && n < L) wav[(size_t)b * L + n] = dot + blast;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (prefetches past the end read the zero page; let them land before the LDS goes)
}

template <int C, int NT, int RD, int WPB, bool TAIL>
static int launch_resunit(const ga_resunit_params& p, hipStream_t s) {
    typedef RuCfg<C, NT, RD, WPB> G;
    constexpr int NRO = G::NR - (TAIL ? 6 : 0);
    const int tpi = (p.L + NRO - 1) / NRO;
    const long total = (long)p.B * tpi;
    if (total <= 0) return GA_OK;
    if (total > (1L << 30)) return GA_ERR_UNSUPPORTED;
    auto kern = resunit_fwd_kernel<C, NT, RD, WPB, TAIL>;
    static bool attr_set[16] = {};  // per device: the attribute is a property of the loaded function
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return GA_ERR_LAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) return GA_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    long grid = (total + WPB - 1) / WPB;
    if (grid > 256) grid = 256;  // one workgroup per CU (its LDS)
    kern<<<(int)grid, WPB * 64, G::LDS, s>>>(p, tpi, (int)total);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

static int launch_resunit_wide(const ga_resunit_params& p, hipStream_t s);

extern "C" int ga_resunit_fwd(const ga_resunit_params* hp, int dtype, hipStream_t s) {
    if (!hp) return GA_ERR_ARG;
    const ga_resunit_params& p = *hp;
    if (!p.x || !p.w_sc || !p.b_sc || !p.w1 || !p.b1 || !p.w3 || !p.b3 || p.B < 0 || p.L < 0) return GA_ERR_ARG;
    if (p.wav ? (p.y || p.t_act || !p.w_last) : !p.y) return GA_ERR_ARG;  // the tail form writes the waveform only
    if (dtype != GA_DTYPE_BF16 || (p.C != 32 && p.C != 64 && p.C != 128) || p.L < 3) return GA_ERR_UNSUPPORTED;
    if (p.wav && (p.C != 32 || p.L < 13)) return GA_ERR_UNSUPPORTED;
    if (p.post_elu && p.C != 128) return GA_ERR_UNSUPPORTED;
    if (((uintptr_t)p.x | (uintptr_t)p.y | (uintptr_t)p.t_act | (uintptr_t)p.w_sc | (uintptr_t)p.w1 | (uintptr_t)p.w3 | (uintptr_t)p.w_last) & 15)
        return GA_ERR_ALIGN;
    if (p.B == 0) return GA_OK;
    if (p.C == 128) return launch_resunit_wide(p, s);
    if (p.wav) return launch_resunit<32, 4, 2, 8, true>(p, s);
    return p.C == 32 ? launch_resunit<32, 4, 2, 8, false>(p, s) : launch_resunit<64, 2, 2, 8, false>(p, s);
}

// ------------------------------------------------------------------------------------------------
// The same unit's input gradient as one pass:
//     gt = (W3^T gy) * ELU'(hidden)              (from t_act = ELU(hidden): 1 where positive, else t_act + 1)
//     gx = W_sc^T gy + ( sum_j W1[j]^T gt[n + 2 - j]  (+ the rows folded back from the reflected left pad) ) * ELU'(x)
// As three ga_conv launches (+ the pad fix-up) it moved 7.5 C-wide passes where 3.5 are needed (gy, x, half-width t_act in; gx
// out).  Per wave and tile: NW rows of gy, t_act and x (the same rows n0 .. n0+NW-1 of the item: the taps look FORWARD here, so
// nothing is reflected in the loads) go global -> LDS one tile ahead, exactly 10 direct-to-LDS loads; the tile yields
// NW - 2 rows of gx.  gy's B fragments feed two products (hidden gradient for all NW rows, shortcut for the output rows); the
// masked hidden gradient goes through a wave-private stage in B-fragment layout (rounded to bf16 as the three-launch form
// stores it); the three taps read it at shifted rows; gb stays in fp32 until the one rounding of gx.
template <int C, int RD, int WPB>
struct RuBwdCfg {
    static constexpr int H = C / 2;
    static constexpr int ROWB = C * 2;
    static constexpr int NW = 4096 / ROWB;                 // window rows: 32 (C = 64) / 64 (C = 32)
    static constexpr int NTW = NW / 32;
    static constexpr int NO = NW - 2;                      // output rows per tile
    static constexpr int CPR = ROWB / 16, RPL = 16 / CPR;  // gy / x rows
    static constexpr int TROWB = H * 2, CPRT = TROWB / 16, RPLT = 16 / CPRT;
    static constexpr int GY_OFF = 0, T_OFF = 4096, X_OFF = 6144, SLOT = 10240;
    static constexpr int NLOAD = 10;                       // 4 + 2 + 4 loads of 1024 bytes
    static constexpr int MT = C / 32;
    static constexpr int KSC = C / 16;
    static constexpr int KS3 = H / 16;
    static constexpr int TROW = H * 2 + 16;                // staged hidden-gradient row
    static constexpr int GSTAGE = (((NW + 2) * TROW + 255) / 256) * 256;
    static constexpr int YROW = 32 * 2 + 8;
    static constexpr int YSTAGE = 2304;
    static constexpr int WAVE_LDS = RD * SLOT + GSTAGE + YSTAGE;
    static constexpr int LDS = WPB * WAVE_LDS;
    static_assert(C == 32 || C == 64, "C");
    static_assert((RD - 1) * NLOAD <= 63, "vmcnt range");
    static_assert(WAVE_LDS % 256 == 0, "lines");
    static_assert(LDS <= 160 * 1024, "LDS");
};

template <int C, int RD, int WPB>
__global__ __launch_bounds__(WPB * 64) void resunit_bwd_kernel(ga_resunit_bwd_params p, int tiles_per_item, int total) {
    typedef RuBwdCfg<C, RD, WPB> G;
    constexpr int H = G::H, ROWB = G::ROWB, NW = G::NW, NTW = G::NTW, NO = G::NO, CPR = G::CPR, RPL = G::RPL;
    constexpr int TROWB = G::TROWB, CPRT = G::CPRT, RPLT = G::RPLT, SLOT = G::SLOT, MT = G::MT, KSC = G::KSC, KS3 = G::KS3;
    constexpr int TROW = G::TROW, YROW = G::YROW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uintptr_t zero = (uintptr_t)ga_zero_page;
    char* const ring = smem + wid * G::WAVE_LDS;
    char* const gstage = ring + RD * SLOT;
    char* const ystage = gstage + G::GSTAGE;

    // A fragments of the data-gradient packs (GA_PACK_CONV_DGRAD_S1: [rows = input channels][tap, flipped][output channels])
    const bf16_t* __restrict__ wsc = (const bf16_t*)p.w_sc_d;  // [C][C]
    const bf16_t* __restrict__ w1 = (const bf16_t*)p.w1_d;     // [C][3][H]
    const bf16_t* __restrict__ w3 = (const bf16_t*)p.w3_d;     // [H][C]
    bf16x8 asc[MT][KSC], a1[MT][3 * KS3], a3[KSC];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) asc[mt][ks] = *(gfrag_ptr_t)(wsc + (size_t)(mt * 32 + r) * C + ks * 16 + h * 8);
#pragma unroll
        for (int ks = 0; ks < 3 * KS3; ++ks) a1[mt][ks] = *(gfrag_ptr_t)(w1 + (size_t)(mt * 32 + r) * (3 * H) + ks * 16 + h * 8);
    }
#pragma unroll
    for (int ks = 0; ks < KSC; ++ks) a3[ks] = *(gfrag_ptr_t)sw_sel(r < H, (uintptr_t)(w3 + (size_t)r * C + ks * 16 + h * 8), zero);

    // fill patterns: load i of a region puts 16 bytes per lane at LDS slot P = i*64 + lane of the region, which holds logical
    // slot (P & 15) ^ (line & 15) of line P >> 4 (the swizzle of conv_swin.hip, applied to the source address)
    int woffx[4], wofft[2];  // byte offset inside the region's rows of the item
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int P = i * 64 + lane, line = P >> 4, sl = (P & 15) ^ (line & 15);
        woffx[i] = (line * RPL + sl / CPR) * ROWB + (sl % CPR) * 16;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int P = i * 64 + lane, line = P >> 4, sl = (P & 15) ^ (line & 15);
        wofft[i] = (line * RPLT + sl / CPRT) * TROWB + (sl % CPRT) * 16;
    }
    // reads.  gy B fragment (nt, ks) of lane (r, h): row nt*32 + r, chunk 2 ks + h
    int foff[NTW][KSC];
    // t_act / x in the ACCUMULATOR layout: element (q, i) of lane (r, h) is channel 8q + 4h + i of row r: 8 bytes at chunk q (+ 8h)
    int toff[NTW][H / 8], xoff[NTW][C / 8];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int row = nt * 32 + r;
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) {
            const int line = row / RPL, sl = (row % RPL) * CPR + 2 * ks + h;
            foff[nt][ks] = G::GY_OFF + line * 256 + ((sl ^ (line & 15)) << 4);
        }
#pragma unroll
        for (int q = 0; q < H / 8; ++q) {
            const int line = row / RPLT, sl = (row % RPLT) * CPRT + q;
            toff[nt][q] = G::T_OFF + line * 256 + ((sl ^ (line & 15)) << 4) + 8 * h;
        }
#pragma unroll
        for (int q = 0; q < C / 8; ++q) {
            const int line = row / RPL, sl = (row % RPL) * CPR + q;
            xoff[nt][q] = G::X_OFF + line * 256 + ((sl ^ (line & 15)) << 4) + 8 * h;
        }
    }

    bf16_t* __restrict__ gx = (bf16_t*)p.gx;
    const int L = p.L;

    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
        const bool valid = t < total;
        const int tt = valid ? t : 0;
        const int b = tt / tiles_per_item, R0 = (tt - b * tiles_per_item) * NO;
        const uintptr_t gyb = (uintptr_t)p.gy + ((size_t)b * L + R0) * ROWB;
        const uintptr_t xb = (uintptr_t)p.x + ((size_t)b * L + R0) * ROWB;
        const uintptr_t tb = (uintptr_t)p.t_act + ((size_t)b * L + R0) * TROWB;
        char* const dst = ring + slot * SLOT;
        if (valid && R0 + NW <= L) {  // (wave-uniform)
#pragma unroll
            for (int i = 0; i < 4; ++i) sw_glds16(gyb + woffx[i], dst + G::GY_OFF + i * 1024);
#pragma unroll
            for (int i = 0; i < 2; ++i) sw_glds16(tb + wofft[i], dst + G::T_OFF + i * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) sw_glds16(xb + woffx[i], dst + G::X_OFF + i * 1024);
        } else {  // rows past the end of the item: zeros (gy = 0 there makes the hidden gradient 0, as the three-launch form has it)
            const int left = valid ? L - R0 : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) sw_glds16(sw_sel(woffx[i] / ROWB < left, gyb + woffx[i], zero), dst + G::GY_OFF + i * 1024);
#pragma unroll
            for (int i = 0; i < 2; ++i) sw_glds16(sw_sel(wofft[i] / TROWB < left, tb + wofft[i], zero), dst + G::T_OFF + i * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) sw_glds16(sw_sel(woffx[i] / ROWB < left, xb + woffx[i], zero), dst + G::X_OFF + i * 1024);
        }
    };

    const int nw = gridDim.x * WPB;
    int t = blockIdx.x * WPB + wid;
#pragma unroll
    for (int d = 0; d < RD - 1; ++d) issue(t + d * nw, d);

    for (int it = 0; t < total; t += nw, ++it) {
        asm volatile("" ::: "memory");
        issue(t + (RD - 1) * nw, (it + RD - 1) % RD);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 1) * G::NLOAD) : "memory");
        char* const Wl = ring + (it % RD) * SLOT;
        const int b = t / tiles_per_item, tl0 = (t - b * tiles_per_item) * NO;

        // 1. hidden gradient of every window row, masked, to the stage
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            f32x16 ga;
#pragma unroll
            for (int i = 0; i < 16; ++i) ga[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks) ga = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[ks], *(const bf16x8*)(Wl + foff[nt][ks]), ga, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < H / 8; ++q) {
                const u32x2 tv = *(const u32x2*)(Wl + toff[nt][q]);
                float m[4];
                m[0] = __uint_as_float(tv.x << 16); m[1] = __uint_as_float(tv.x & 0xffff0000u);
                m[2] = __uint_as_float(tv.y << 16); m[3] = __uint_as_float(tv.y & 0xffff0000u);
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = ga[4 * q + i] * (m[i] > 0.f ? 1.f : m[i] + 1.f);
                u32x2 o;
                o.x = sw_pack_bf16(m[0], m[1]);
                o.y = sw_pack_bf16(m[2], m[3]);
                sw_lds_write8(gstage + (nt * 32 + r) * TROW + (8 * q + 4 * h) * 2, o);
            }
        }
        __builtin_amdgcn_wave_barrier();

        // 2. per 32 output rows: three taps of the hidden gradient, ELU'(x), the shortcut on top, one rounding
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            f32x16 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int ks = 0; ks < KS3; ++ks) {
                    const bf16x8 gb = *(const bf16x8*)(gstage + (nt * 32 + r + j) * TROW + (ks * 16 + h * 8) * 2);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[mt][j * KS3 + ks], gb, acc[mt], 0, 0, 0);
                }
            if (nt == 0 && tl0 == 0) {
                // the reflected left pad of the causal conv (HF modeling_encodec.py:157-176: padded row -l is row l): the gradient of
                // padded rows -1, -2 returns to rows 1, 2 -- row l takes tap j of hidden-gradient row j - l, 1 <= l <= j
#pragma unroll
                for (int j = 1; j < 3; ++j)
#pragma unroll
                    for (int ks = 0; ks < KS3; ++ks) {
                        const bool on = r >= 1 && r <= j;
                        bf16x8 gb = *(const bf16x8*)(gstage + (on ? j - r : 0) * TROW + (ks * 16 + h * 8) * 2);
                        if (!on) gb = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[mt][j * KS3 + ks], gb, acc[mt], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x2 xv = *(const u32x2*)(Wl + xoff[nt][mt * 4 + q]);
                    float m[4];
                    m[0] = __uint_as_float(xv.x << 16); m[1] = __uint_as_float(xv.x & 0xffff0000u);
                    m[2] = __uint_as_float(xv.y << 16); m[3] = __uint_as_float(xv.y & 0xffff0000u);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[mt][4 * q + i] *= m[i] > 0.f ? 1.f : __builtin_amdgcn_exp2f(m[i] * 1.4426950408889634f);
                }
#pragma unroll
            for (int ks = 0; ks < KSC; ++ks) {
                const bf16x8 gf = *(const bf16x8*)(Wl + foff[nt][ks]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asc[mt][ks], gf, acc[mt], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x2 o;
                    o.x = sw_pack_bf16(acc[mt][4 * q], acc[mt][4 * q + 1]);
                    o.y = sw_pack_bf16(acc[mt][4 * q + 2], acc[mt][4 * q + 3]);
                    sw_lds_write8(ystage + r * YROW + (8 * q + 4 * h) * 2, o);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int row = ps * 16 + (lane >> 2), m0 = (lane & 3) * 8;
                    const int wr = nt * 32 + row, tl = tl0 + wr;
                    const u32x2 lo = *(const u32x2*)(ystage + row * YROW + m0 * 2), hi = *(const u32x2*)(ystage + row * YROW + m0 * 2 + 8);
                    u32x4 ov;
                    ov.x = lo.x; ov.y = lo.y; ov.z = hi.x; ov.w = hi.y;
                    if (wr < NO && tl < L) *(u32x4*)(gx + ((size_t)b * L + tl) * C + mt * 32 + m0) = ov;
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int C, int RD, int WPB>
static int launch_resunit_bwd(const ga_resunit_bwd_params& p, hipStream_t s) {
    typedef RuBwdCfg<C, RD, WPB> G;
    const int tpi = (p.L + G::NO - 1) / G::NO;
    const long total = (long)p.B * tpi;
    if (total <= 0) return GA_OK;
    if (total > (1L << 30)) return GA_ERR_UNSUPPORTED;
    auto kern = resunit_bwd_kernel<C, RD, WPB>;
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return GA_ERR_LAUNCH;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) return GA_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    long grid = (total + WPB - 1) / WPB;
    if (grid > 256) grid = 256;
    kern<<<(int)grid, WPB * 64, G::LDS, s>>>(p, tpi, (int)total);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

extern "C" int ga_resunit_bwd(const ga_resunit_bwd_params* hp, int dtype, hipStream_t s) {
    if (!hp) return GA_ERR_ARG;
    const ga_resunit_bwd_params& p = *hp;
    if (!p.gy || !p.t_act || !p.x || !p.gx || !p.w_sc_d || !p.w1_d || !p.w3_d || p.B < 0 || p.L < 0) return GA_ERR_ARG;
    if (dtype != GA_DTYPE_BF16 || (p.C != 32 && p.C != 64) || p.L < 3) return GA_ERR_UNSUPPORTED;
    if (((uintptr_t)p.gy | (uintptr_t)p.t_act | (uintptr_t)p.x | (uintptr_t)p.gx | (uintptr_t)p.w_sc_d | (uintptr_t)p.w1_d | (uintptr_t)p.w3_d) & 15)
        return GA_ERR_ALIGN;
    if (p.B == 0) return GA_OK;
    return p.C == 32 ? launch_resunit_bwd<32, 2, 6>(p, s) : launch_resunit_bwd<64, 2, 6>(p, s);
}

// ------------------------------------------------------------------------------------------------
// The 128-channel unit (15000 steps per item) as one pass.  Its three weight matrices are 96 KB: too many registers for one
// wave, so here a WORKGROUP of four waves owns a tile of 64 steps and splits the output channels: wave w holds rows 32w..32w+31
// of W_sc and W3 and one (32 hidden channels x 32 steps) quarter of the hidden product (all 24 K-steps of W1's rows) -- 144
// weight registers.  The window (66 rows of 256 bytes, one row per 256-byte line, slots XOR-swizzled by the row) is loaded
// cooperatively one tile ahead (five direct-to-LDS loads per wave) and shared; four workgroup barriers per tile order the
// phases: shortcut on the raw window | ELU in place | hidden quarter -> bf16, ELU -> shared stage | second conv, epilogue.
// Two workgroups per CU overlap one another's phases.
struct RuWideCfg {
    static constexpr int C = 128, H = 64, NT = 2, NR = 64, RD = 2;
    static constexpr int WROWS = NR + 2;
    static constexpr int WCH = WROWS * 16;                 // 16-byte chunks of a window
    static constexpr int NLW = 5;                          // loads per wave and tile (4 waves x 64 lanes x 5 >= WCH; the rest hits a pad)
    static constexpr int SLOT = NLW * 4096;
    static constexpr int HROW = H * 2 + 16;                // shared hidden stage: 64 rows
    static constexpr int HSTAGE = NR * HROW;
    static constexpr int YROW = 72, YSTAGE = 2304;         // per-wave output stage (32 rows x 32 channels)
    static constexpr int BIAS_OFF = RD * SLOT + HSTAGE + 4 * YSTAGE;
    static constexpr int LDS = BIAS_OFF + 1024;
    static_assert(WCH <= NLW * 256, "loads");
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
};

// workgroup barrier that the compiler cannot move LDS accesses across (the s_barrier builtin is "no memory" to it), with this
// wave's LDS operations complete first
#define RU_WG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <bool POST_ELU>
__global__ __launch_bounds__(256) void resunit_wide_fwd_kernel(ga_resunit_params p, int tiles_per_item, int total) {
    typedef RuWideCfg G;
    constexpr int C = G::C, H = G::H, NT = G::NT, NR = G::NR, RD = G::RD, SLOT = G::SLOT, HROW = G::HROW, YROW = G::YROW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uintptr_t zero = (uintptr_t)ga_zero_page;
    char* const hstage = smem + RD * SLOT;
    char* const ystage = hstage + G::HSTAGE + wid * G::YSTAGE;
    float* const bsum = (float*)(smem + G::BIAS_OFF);
    float* const b1s = bsum + C;
    for (int i = tid; i < C + H; i += 256) bsum[i] = i < C ? p.b_sc[i] + p.b3[i] : p.b1[i - C];
    __syncthreads();

    const bf16_t* __restrict__ wsc = (const bf16_t*)p.w_sc;
    const bf16_t* __restrict__ w1 = (const bf16_t*)p.w1;
    const bf16_t* __restrict__ w3 = (const bf16_t*)p.w3;
    const int hm = wid & 1, hn = wid >> 1;  // this wave's quarter of the hidden product: hidden channels 32hm.., steps 32hn..
    bf16x8 asc[8], a3[4], a1[24];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asc[ks] = *(gfrag_ptr_t)(wsc + (size_t)(wid * 32 + r) * C + ks * 16 + h * 8);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a3[ks] = *(gfrag_ptr_t)(w3 + (size_t)(wid * 32 + r) * H + ks * 16 + h * 8);
#pragma unroll
    for (int ks = 0; ks < 24; ++ks) a1[ks] = *(gfrag_ptr_t)(w1 + (size_t)(hm * 32 + r) * (3 * C) + ks * 16 + h * 8);

    // cooperative fill: load i of wave w puts chunk P = i*256 + w*64 + lane at LDS offset 16 P: row P >> 4, logical chunk (P & 15) ^ (row & 15)
    int wsrc[G::NLW];  // byte offset in the item's rows, or -1: pad
#pragma unroll
    for (int i = 0; i < G::NLW; ++i) {
        const int P = i * 256 + wid * 64 + lane, row = P >> 4, ch = (P & 15) ^ (row & 15);
        wsrc[i] = P < G::WCH ? row * 256 + ch * 16 : -1;
    }
    bf16_t* __restrict__ y = (bf16_t*)p.y;
    bf16_t* __restrict__ tact = (bf16_t*)p.t_act;
    const int L = p.L;

    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
        const bool valid = t < total;
        const int tt = valid ? t : 0;
        const int b = tt / tiles_per_item, R0 = (tt - b * tiles_per_item) * NR - 2;
        const uintptr_t xb = (uintptr_t)p.x + (size_t)b * L * 256;
        char* const dst = smem + slot * SLOT + wid * 1024;
#pragma unroll
        for (int i = 0; i < G::NLW; ++i) {
            int R = R0 + (wsrc[i] >> 8);
            if (R < 0) R = -R;  // reflect-left padding of the causal conv
            const bool ok = valid && wsrc[i] >= 0 && R < L;
            sw_glds16(sw_sel(ok, xb + (size_t)(ok ? R : 0) * 256 + (wsrc[i] & 255), zero), dst + i * 4096);
        }
    };
    // B fragment of (tap, k-step kk) for row r (+ 32 rows: + 8192 bytes): chunk 2 kk + h of window row r + tap
    auto frag = [&](const char* Wl, int nt, int tap, int kk) __attribute__((always_inline)) -> bf16x8 {
        const int row = r + tap;
        return *(const bf16x8*)(Wl + nt * 8192 + row * 256 + (((2 * kk + h) ^ (row & 15)) << 4));
    };

    int t = blockIdx.x;
#pragma unroll
    for (int d = 0; d < RD - 1; ++d) issue(t + d * gridDim.x, d);
    for (int it = 0; t < total; t += gridDim.x, ++it) {
        asm volatile("" ::: "memory");
        issue(t + (RD - 1) * gridDim.x, (it + RD - 1) % RD);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 1) * G::NLW) : "memory");
        RU_WG_BARRIER();  // every wave's part of the window has landed
        char* const Wl = smem + (it % RD) * SLOT;
        const int b = t / tiles_per_item, tl0 = (t - b * tiles_per_item) * NR;

        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
            // (fragments in groups of eight ahead of their MFMAs: left to itself the compiler issues every LDS read right in front
            // of its MFMA and waits for it -- one LDS round trip per K-step, which is what bounded this kernel)
            bf16x8 f[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) f[kk] = frag(Wl, nt, 2, kk);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(asc[kk], f[kk], acc[nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        RU_WG_BARRIER();  // the raw window has been read by everyone
#pragma unroll
        for (int i = 0; i < G::NLW; ++i)
            if (wsrc[i] >= 0) {
                char* const q = Wl + (i * 256 + wid * 64 + lane) * 16;
                sw_lds_write16(q, sw_elu8(*(const u32x4*)q));
            }
        RU_WG_BARRIER();  // ELU(window) complete
        {
            f32x16 ta;
#pragma unroll
            for (int i = 0; i < 16; ++i) ta[i] = 0.f;
            ru_f32x4 bbh[4];  // (read ahead of the MFMAs that hide the LDS round trip)
#pragma unroll
            for (int q = 0; q < 4; ++q) bbh[q] = *(const ru_f32x4*)(b1s + hm * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                bf16x8 f[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) f[kk] = frag(Wl, hn, tap, kk);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) ta = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[tap * 8 + kk], f[kk], ta, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const ru_f32x4 bb = bbh[q];
                u32x2 o;
                o.x = sw_elu2(sw_pack_bf16(ta[4 * q] + bb[0], ta[4 * q + 1] + bb[1]));
                o.y = sw_elu2(sw_pack_bf16(ta[4 * q + 2] + bb[2], ta[4 * q + 3] + bb[3]));
                sw_lds_write8(hstage + (hn * 32 + r) * HROW + (hm * 32 + 8 * q + 4 * h) * 2, o);
            }
        }
        RU_WG_BARRIER();  // hidden stage complete (and nobody reads the window any more)
        if (tact) {  // 64 rows x 128 bytes of ELU(hidden): 512 pieces over 256 lanes
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int piece = ps * 256 + tid, row = piece >> 3, c0 = (piece & 7) * 8;
                const int tl = tl0 + row;
                const u32x4 v = *(const u32x4*)(hstage + row * HROW + c0 * 2);
                if (tl < L) *(u32x4*)(tact + ((size_t)b * L + tl) * H + c0) = v;
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bf16x8 tb[4];
            ru_f32x4 bbo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bbo[q] = *(const ru_f32x4*)(bsum + wid * 32 + 8 * q + 4 * h);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) tb[ks] = *(const bf16x8*)(hstage + (nt * 32 + r) * HROW + (ks * 16 + h * 8) * 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[ks], tb[ks], acc[nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const ru_f32x4 bb = bbo[q];
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[nt][4 * q + i] + bb[i];
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
                const int row = ps * 16 + (lane >> 2), m0 = (lane & 3) * 8;
                const int tl = tl0 + nt * 32 + row;
                const u32x2 lo = *(const u32x2*)(ystage + row * YROW + m0 * 2), hi = *(const u32x2*)(ystage + row * YROW + m0 * 2 + 8);
                u32x4 ov;
                ov.x = lo.x; ov.y = lo.y; ov.z = hi.x; ov.w = hi.y;
                if (tl < L) *(u32x4*)(y + ((size_t)b * L + tl) * C + wid * 32 + m0) = ov;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static int launch_resunit_wide(const ga_resunit_params& p, hipStream_t s) {
    typedef RuWideCfg G;
    const int tpi = (p.L + G::NR - 1) / G::NR;
    const long total = (long)p.B * tpi;
    if (total <= 0) return GA_OK;
    if (total > (1L << 30)) return GA_ERR_UNSUPPORTED;
    static bool attr_set[16][2] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return GA_ERR_LAUNCH;
    const int v = p.post_elu ? 1 : 0;
    if (!attr_set[dev][v]) {
        const void* k = v ? (const void*)resunit_wide_fwd_kernel<true> : (const void*)resunit_wide_fwd_kernel<false>;
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) return GA_ERR_LAUNCH;
        attr_set[dev][v] = true;
    }
    const int grid = (int)(total < 512 ? total : 512);  // two workgroups per CU
    if (v) resunit_wide_fwd_kernel<true><<<grid, 256, G::LDS, s>>>(p, tpi, (int)total);
    else resunit_wide_fwd_kernel<false><<<grid, 256, G::LDS, s>>>(p, tpi, (int)total);
    GA_CHECK_LAUNCH();
    return GA_OK;
}
