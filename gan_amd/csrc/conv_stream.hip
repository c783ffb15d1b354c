// bf16 MFMA "streaming" kernel for the thin layers of the generalised NLC convolution (gfx950):
// M <= 64 output channels and K = J*Cin <= 192.  These layers (the 16..64-channel end of the SEANet decoder at
// 60000..120000 samples per item, and their data gradients) move 200-550 MB per launch for a few GFLOP: they are
// HBM-bound, and what limits a tiled GEMM on them is bytes in flight per CU, not arithmetic.
//
//   * the packed weights (<= 24 KiB) live in registers as MFMA A fragments for the whole launch
//   * the B operand needs no LDS: in channels-last layout the fragment of v_mfma_f32_32x32x16_bf16 that lane (n, h)
//     supplies -- 8 consecutive k of im2col row n -- is 16 contiguous bytes of x (one tap, 8 channels), loaded
//     straight from global memory into the operand registers (padding reads a 16-byte zero page)
//   * every wave walks groups of NT*32 rows of the flattened (item, step) axis, grid-stride; per trip it issues, in this
//     order and with no branch around any load, the residual / derivative-source pieces of the CURRENT group and the x
//     fragments of the NEXT group, then multiplies and stores the current group: the memory counter is in-order, so the
//     waits the compiler places leave the next group's loads in flight behind the current group's epilogue
//   * the input activation (ELU in front of every decoder conv, HF modeling_encodec.py:268,330,340) is applied to
//     the fragments in registers, so no act(x) copy is written and read back
//   * epilogue: accumulators -> wave-private LDS rows -> 16-byte row-contiguous global accesses
// Epilogue variants are compile-time (SIDE: none / residual / derivative source) so that the loop stays small and
// branch-free; other combinations (accumulate, fp32 output, residual AND derivative) stay with conv_gemm.hip.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define STREAM_KMAX 192
#define SIDE_NONE 0
#define SIDE_RES 1
#define SIDE_DACT 2

// loads through an integer-selected address must say "global": a generic pointer becomes flat_load, whose waits drain
// both counters to zero and serialise the pipeline
typedef const __attribute__((address_space(1))) bf16x8* gfrag_ptr_t;
__device__ __forceinline__ bf16x8 ldg_frag(uintptr_t a) { return *(gfrag_ptr_t)a; }
// the same from a 2-byte-aligned address (single-channel inputs: 8 consecutive samples starting anywhere)
typedef bf16x8 __attribute__((aligned(2))) bf16x8_u;
typedef const __attribute__((address_space(1))) bf16x8_u* gfrag_u_ptr_t;
__device__ __forceinline__ bf16x8 ldg_frag_u(uintptr_t a) { return *(gfrag_u_ptr_t)a; }

__device__ __forceinline__ uintptr_t sel_addr_s(bool ok, uintptr_t a, uintptr_t z) {
    return z ^ ((z ^ a) & (uintptr_t)(-(long long)ok));
}

// ELU on a bf16 fragment, two elements per step so that the multiply / add / convert are packed instructions.
// f > -2^-12 returns f itself (for -2^-12 < f < 0, expm1(f) = f to well inside half a bf16 ulp, and exp(f) - 1 would
// cancel there), so one compare serves both the positive side and the cancellation guard.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 elu_frag(bf16x8 v) {
    union { bf16x8 b; unsigned int u[4]; } in, out;
    in.b = v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x2 f;
        f.x = __uint_as_float(in.u[i] << 16);
        f.y = __uint_as_float(in.u[i] & 0xffff0000u);
        const f32x2 t = f * 1.4426950408889634f;
        f32x2 e;
        e.x = __builtin_amdgcn_exp2f(t.x);
        e.y = __builtin_amdgcn_exp2f(t.y);
        e = e - 1.0f;
        const float rx = f.x > -2.44140625e-4f ? f.x : e.x;
        const float ry = f.y > -2.44140625e-4f ? f.y : e.y;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        union { bf16x2 b; unsigned int u; } pk;
        pk.b[0] = (__bf16)rx;
        pk.b[1] = (__bf16)ry;
        out.u[i] = pk.u;
    }
    return out.b;
}

// n -> (n / L, n % L) for 0 <= n < 2^30 and quotients far below 2^20: float estimate, one correction each way
__device__ __forceinline__ void divmod_small(int n, int L, float rcpL, int& q, int& rem) {
    q = (int)((float)n * rcpL);
    rem = n - q * L;
    if (rem < 0) { rem += L; --q; }
    if (rem >= L) { rem -= L; ++q; }
}

// CIN1: single input channel (the discriminators' first conv on the waveform).  K is the tap axis itself: the fragment of
// lane (n, h) is 8 consecutive SAMPLES x[row(n) + ks*16 + 8h ..], one unaligned 16-byte load.  Padding cannot be applied per
// 16-byte piece here, so the caller hands in a zero-padded copy of x in which every window is in range (checked on the host).
template <int MT, int KS, int NT, bool PRE_ELU, int SIDE, bool CIN1 = false>
__global__ __launch_bounds__(256) void conv_stream_kernel(ConvParams p, int Ntot, int ng, float rcpL) {
    constexpr int EM = MT > 2 ? 2 : MT;  // m-tiles per epilogue round (the stage holds at most 64 channels per row)
    constexpr int ER = MT / EM;          // epilogue rounds: 2 for the 128-channel form
    static_assert(MT % EM == 0, "rounds");
    constexpr int WMR = EM * 32;
    constexpr int EPF = WMR + 4;
    constexpr int LPR = WMR / 8;    // lanes per staged row (8 output channels each)
    constexpr int RPP = 64 / LPR;   // rows per pass
    constexpr int NPS = 32 / RPP;   // passes per 32-row tile
    __shared__ __attribute__((aligned(16))) float stage_all[4 * 32 * EPF];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int Cin = p.Cin, Ktot = p.J * Cin, S = p.in_stride, Lout = p.Lout;
    const int off0 = p.off0 + p.phase_off0[0];
    const uintptr_t zero = (uintptr_t)ga_zero_page;
    const bf16_t* __restrict__ w = (const bf16_t*)p.w;

    // weights: A fragment (mt, ks) of lane (r, h) = w[mt*32 + r][ks*16 + h*8 .. +8]
    bf16x8 a[MT][KS];
    int dR[KS], cc[KS];
    bool kin[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = ks * 16 + h * 8;
        if (CIN1) {
            kin[ks] = true;
            dR[ks] = k;  // (tap_step == 1: the k-th tap is k samples further)
            cc[ks] = 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 32 + r;
#pragma unroll
                for (int i = 0; i < 8; ++i)  // rows of J (not a multiple of 8) taps: element-wise, once per launch
                    a[mt][ks][i] = (m < p.M && k + i < Ktot) ? w[(size_t)m * Ktot + k + i] : (__bf16)0.f;
            }
            continue;
        }
        kin[ks] = k < Ktot;
        const int j = k / Cin;
        dR[ks] = j * p.tap_step;
        cc[ks] = k - j * Cin;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mt * 32 + r;
            const bool ok = kin[ks] && m < p.M;
            a[mt][ks] = ldg_frag(sel_addr_s(ok, (uintptr_t)(w + (size_t)m * Ktot + k), zero));
        }
    }

    float* const stage = stage_all + wid * (32 * EPF);
    bf16_t* __restrict__ y = (bf16_t*)p.y;
    const bf16_t* __restrict__ side_src = SIDE == SIDE_RES ? (const bf16_t*)p.res : (const bf16_t*)p.dact_src;
    const int erow = lane / LPR, m0 = (lane % LPR) * 8;
    const size_t item_elems = p.y_batch_elems > 0 ? (size_t)p.y_batch_elems : (size_t)p.LoutTot * p.M;
    const float slope = p.post_act == GA_ACT_LRELU ? 0.2f : 1.f;
    const int dmode = p.dact_mode;  // (uniform: the compiler hoists the choice out of the element loop)
    const bool elu_after = p.res_mode == GA_RES_ADD_THEN_ACT;  // ELU(acc + bias + res)
    float bv[ER][8];
#pragma unroll
    for (int er = 0; er < ER; ++er) {
#pragma unroll
        for (int i = 0; i < 8; ++i) bv[er][i] = 0.f;
        const int mm = er * WMR + m0;
        if (p.bias && mm < p.M) {
            const float4 b0 = *(const float4*)(p.bias + mm), b1 = *(const float4*)(p.bias + mm + 4);
            bv[er][0] = b0.x; bv[er][1] = b0.y; bv[er][2] = b0.z; bv[er][3] = b0.w;
            bv[er][4] = b1.x; bv[er][5] = b1.y; bv[er][6] = b1.z; bv[er][7] = b1.w;
        }
    }

    // x fragments of group g (every load is issued; rows or k past the end read the zero page)
    auto load = [&](bf16x8 (&buf)[NT][KS], int g) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = (g * NT + nt) * 32 + r;
            int b, tl;
            divmod_small(n, Lout, rcpL, b, tl);
            const bool valid = n < Ntot;
            const bf16_t* base = (const bf16_t*)p.x + (size_t)b * p.Lin * Cin;
            const int R0 = (p.t0 + tl) * S + off0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                int R = R0 + dR[ks];
                if (CIN1) {
                    const bool ok = valid && R >= 0 && R + 8 <= p.Lin;
                    buf[nt][ks] = ldg_frag_u(sel_addr_s(ok, (uintptr_t)(base + R), zero));
                    continue;
                }
                if (R < 0 && p.pad_mode == 1) R = -R;
                const bool ok = valid && kin[ks] && R >= 0 && R < p.Lin;
                buf[nt][ks] = ldg_frag(sel_addr_s(ok, (uintptr_t)(base + (size_t)R * Cin + cc[ks]), zero));
            }
        }
    };

    // output element offset of this lane's 8 channels in pass ps of tile nt of group g (-1: nothing to write)
    auto out_offset = [&](int g, int nt, int ps, int er) __attribute__((always_inline)) -> long {
        const int n = (g * NT + nt) * 32 + ps * RPP + erow;
        int b, tl;
        divmod_small(n, Lout, rcpL, b, tl);
        const int orow = (p.t0 + tl) * p.out_stride;
        const int mm = er * WMR + m0;
        const size_t in_item = (size_t)orow * p.M + mm;
        const bool ok = n < Ntot && mm < p.M && orow >= 0 && orow < p.LoutTot && in_item + 8 <= item_elems;
        return ok ? (long)((size_t)b * item_elems + in_item) : -1L;
    };

    auto compute = [&](bf16x8 (&buf)[NT][KS], bf16x8 (&side)[NT][ER][NPS], int g) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 xb = PRE_ELU ? elu_frag(buf[nt][ks]) : buf[nt][ks];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][ks], xb, acc[mt], 0, 0, 0);
            }
#pragma unroll
          for (int er = 0; er < ER; ++er) {
#pragma unroll
            for (int me = 0; me < EM; ++me)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(float4*)(stage + r * EPF + me * 32 + 8 * q + 4 * h) =
                        make_float4(acc[er * EM + me][4 * q], acc[er * EM + me][4 * q + 1], acc[er * EM + me][4 * q + 2], acc[er * EM + me][4 * q + 3]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int row = ps * RPP + erow;
                const long o = out_offset(g, nt, ps, er);
                float v[8];
                *(float4*)&v[0] = *(const float4*)(stage + row * EPF + m0);
                *(float4*)&v[4] = *(const float4*)(stage + row * EPF + m0 + 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] += bv[er][i];
                    v[i] = v[i] > 0.f ? v[i] : slope * v[i];
                }
                if (SIDE == SIDE_DACT) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float sv = (float)side[nt][er][ps][i];
                        v[i] *= sv > 0.f ? 1.f : (dmode == GA_DACT_ELU_FROM_IN ? __expf(sv) : dmode == GA_DACT_ELU_FROM_OUT ? sv + 1.f : 0.2f);
                    }
                }
                if (SIDE == SIDE_RES) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += (float)side[nt][er][ps][i];
                    if (elu_after) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = elu_bf16_grade(v[i]);
                    }
                }
                bf16x8 ov;
#pragma unroll
                for (int i = 0; i < 8; ++i) ov[i] = (__bf16)v[i];
                if (o >= 0) *(bf16x8*)(y + o) = ov;
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
    };

    auto load_side = [&](bf16x8 (&side)[NT][ER][NPS], int g) __attribute__((always_inline)) {
        if (SIDE == SIDE_NONE) return;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int er = 0; er < ER; ++er)
#pragma unroll
                for (int ps = 0; ps < NPS; ++ps) {
                    const long o = out_offset(g, nt, ps, er);
                    side[nt][er][ps] = ldg_frag(sel_addr_s(o >= 0, (uintptr_t)(side_src + (o >= 0 ? o : 0)), zero));
                }
    };

    const int stride = gridDim.x * 4;
    int g = blockIdx.x * 4 + wid;
    if (g >= ng) return;
    bf16x8 b0[NT][KS], b1[NT][KS], sd[NT][ER][NPS];
    load(b0, g);
    while (true) {  // two groups per trip: the buffers swap roles by name, never by copy
        load_side(sd, g);
        load(b1, min(g + stride, ng - 1));
        compute(b0, sd, g);
        g += stride;
        if (g >= ng) break;
        load_side(sd, g);
        load(b0, min(g + stride, ng - 1));
        compute(b1, sd, g);
        g += stride;
        if (g >= ng) break;
    }
}

// ------------------------------------------------------------------------------------------------
static int stream_side(const ConvParams& p) {
    if (p.res && (p.dact_mode || p.dact_src)) return -1;
    if (p.res) return SIDE_RES;
    if (p.dact_mode) return SIDE_DACT;
    return SIDE_NONE;
}

// single-channel form: 1 <= J <= 32 taps one sample apart, and every 8-sample piece any lane touches inside its item
static int stream_cin1_ok(const ConvParams& p) {
    if (p.Cin != 1 || p.tap_step != 1 || p.J < 1 || p.J > 32 || p.in_stride < 1 || p.t0 != 0 || p.pad_mode != 0) return 0;
    if (p.pre_act != GA_ACT_NONE || p.res) return 0;  // (a derivative source is fine: the decoder's last conv's data gradient)
    const int ks = (p.J + 15) / 16;
    if (p.off0 < 0 || (long)(p.Lout - 1) * p.in_stride + p.off0 + ks * 16 > p.Lin) return 0;
    return 1;
}

int ga_conv_stream_supported(const ConvParams& p) {
    // ELU after the residual sum (a residual unit handing ELU(output) on): only what the unit's second conv looks like
    const bool elu_after = p.res_mode == GA_RES_ADD_THEN_ACT;
    if (elu_after ? !(p.res && p.post_act == GA_ACT_ELU && !p.dact_mode && !p.dact_src) : p.res_mode != GA_RES_ADD) return 0;
    if (p.y2) return 0;
    if (p.pre_act != GA_ACT_NONE && p.pre_act != GA_ACT_ELU) return 0;
    if (!elu_after && p.post_act != GA_ACT_NONE && p.post_act != GA_ACT_LRELU) return 0;
    if (p.dact_mode != GA_DACT_NONE && p.dact_mode != GA_DACT_ELU_FROM_IN && p.dact_mode != GA_DACT_LRELU_FROM_OUT &&
        p.dact_mode != GA_DACT_ELU_FROM_OUT) return 0;
    if (p.dact_mode && !p.dact_src) return 0;
    if (p.accumulate || p.out_f32 || stream_side(p) < 0) return 0;
    if (p.pre_act == GA_ACT_ELU && p.dact_mode) return 0;  // (not instantiated: a forward fuses ELU, a data gradient its derivative)
    if (p.nphase != 1 || p.ksplit > 1) return 0;
    if (p.M % 8 || p.M < 8 || p.M > 128 || p.tap_step < 0 || p.J < 1 || p.B < 1 || p.Lout < 0) return 0;
    if (p.M > 64 && (p.Cin == 1 || (long)p.J * p.Cin > 64 || p.pre_act)) return 0;  // 128 channels: K <= 64 (64 weight registers)
    if (p.M > 64 || elu_after) {
        static int wide = -1;  // GA_STREAM_M128=0: these stay with the gathered-GEMM kernel (A/B checks)
        if (wide < 0) {
            const char* e = getenv("GA_STREAM_M128");
            wide = e ? atoi(e) : 1;
        }
        if (!wide) return 0;
    }
    if (p.Cin == 1) {
        if (!stream_cin1_ok(p)) return 0;
    } else if (p.Cin % 8 || p.Cin < 8) {
        return 0;
    }
    if ((long)p.J * p.Cin > STREAM_KMAX) return 0;
    if (p.y_batch_elems < 0 || (p.y_batch_elems > 0 && (p.y_batch_elems % 8 || (long)p.y_batch_elems > (long)p.LoutTot * p.M))) return 0;
    if (((uintptr_t)p.y | (uintptr_t)p.res | (uintptr_t)p.dact_src | (uintptr_t)p.bias) & 15) return 0;
    if (p.Cin == 1 ? (((uintptr_t)p.x | (uintptr_t)p.w) & 1) : (((uintptr_t)p.x | (uintptr_t)p.w) & 15)) return 0;
    if ((long)p.B * p.Lout > (1L << 30) || p.B > (1 << 18)) return 0;
    return 1;
}

template <int MT, int KS, int NT>
static int launch_stream(const ConvParams& p, hipStream_t s) {
    const int Ntot = p.B * p.Lout;
    const int ng = (Ntot + 32 * NT - 1) / (32 * NT);
    if (ng <= 0) return GA_OK;
    int grid = (ng + 3) / 4;
    if (grid > 2048) grid = 2048;  // 8 workgroups per CU; the waves stride over the rest
    const float rcpL = 1.0f / (float)p.Lout;
    const int side = stream_side(p);
    if (p.pre_act == GA_ACT_ELU) {
        if (side == SIDE_RES) conv_stream_kernel<MT, KS, NT, true, SIDE_RES><<<grid, 256, 0, s>>>(p, Ntot, ng, rcpL);
        else conv_stream_kernel<MT, KS, NT, true, SIDE_NONE><<<grid, 256, 0, s>>>(p, Ntot, ng, rcpL);
    } else {
        if (side == SIDE_RES) conv_stream_kernel<MT, KS, NT, false, SIDE_RES><<<grid, 256, 0, s>>>(p, Ntot, ng, rcpL);
        else if (side == SIDE_DACT) conv_stream_kernel<MT, KS, NT, false, SIDE_DACT><<<grid, 256, 0, s>>>(p, Ntot, ng, rcpL);
        else conv_stream_kernel<MT, KS, NT, false, SIDE_NONE><<<grid, 256, 0, s>>>(p, Ntot, ng, rcpL);
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}

template <int MT, int KS>
static int launch_stream_cin1(const ConvParams& p, hipStream_t s) {
    constexpr int NT = 4;
    const int Ntot = p.B * p.Lout;
    const int ng = (Ntot + 32 * NT - 1) / (32 * NT);
    if (ng <= 0) return GA_OK;
    int grid = (ng + 3) / 4;
    if (grid > 2048) grid = 2048;
    if (stream_side(p) == SIDE_DACT) conv_stream_kernel<MT, KS, NT, false, SIDE_DACT, true><<<grid, 256, 0, s>>>(p, Ntot, ng, 1.0f / (float)p.Lout);
    else conv_stream_kernel<MT, KS, NT, false, SIDE_NONE, true><<<grid, 256, 0, s>>>(p, Ntot, ng, 1.0f / (float)p.Lout);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

template <int MT>
static int launch_stream_mt(const ConvParams& p, hipStream_t s) {
    if (p.Cin == 1) return p.J <= 16 ? launch_stream_cin1<MT, 1>(p, s) : launch_stream_cin1<MT, 2>(p, s);
    const int ks = (p.J * p.Cin + 15) / 16;
    if (ks <= 1) return launch_stream<MT, 1, 4>(p, s);
    if (ks <= 2) return launch_stream<MT, 2, MT == 1 ? 4 : 2>(p, s);
    if (ks <= 4) return launch_stream<MT, 4, 2>(p, s);
    if (ks <= 6) return launch_stream<MT, 6, 1>(p, s);
    if (ks <= 8) return launch_stream<MT, 8, 1>(p, s);
    return launch_stream<MT, 12, 1>(p, s);
}

int ga_conv_stream_impl(const ConvParams& p, hipStream_t s) {
    if (!ga_conv_stream_supported(p) || !p.x || !p.w || !p.y) return GA_ERR_ARG;
    if (p.Lout == 0) return GA_OK;
    if (p.M > 64) {  // (K <= 64, no single-channel form: checked above)
        const int ks = (p.J * p.Cin + 15) / 16;
        return ks <= 1 ? launch_stream<4, 1, 2>(p, s) : ks <= 2 ? launch_stream<4, 2, 2>(p, s) : launch_stream<4, 4, 1>(p, s);
    }
    return p.M <= 32 ? launch_stream_mt<1>(p, s) : launch_stream_mt<2>(p, s);
}
