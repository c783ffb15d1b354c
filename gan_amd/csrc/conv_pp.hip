// bf16 MFMA implicit-GEMM kernel, "ping-pong" form, for the deep-K layers of the generalised NLC convolution (gfx950).
//
//   Y[n][m] = sum_k  Wp[m][k] * X[n][k]        n = (batch item, output step) flattened, k = (tap j, channel c)
//
// Same gathered-GEMM operand staging as conv_gemm.hip (every 16-byte piece of both operands goes global -> LDS by a
// direct-to-LDS load whose per-lane SOURCE address does the im2col gather and the XOR swizzle), but a different schedule:
//
//   block tile 256 (m) x 256 (n), K-step 64, eight waves as 2 (m) x 4 (n), each wave 128 x 64 = 4 x 2 MFMA 32x32x16 tiles
//   (0.75 KiB of LDS reads per MFMA instead of 1 KiB for a 64 x 64 wave tile), one workgroup per CU, 2 x 64 KiB of LDS.
//   A K-step is four phases; a phase is a LOAD segment (fragment reads for one 64 x 32 quadrant of the wave tile + two
//   direct-to-LDS loads of a later K-step) and an MFMA segment (8 MFMAs), each ended by a raw s_barrier.  Waves 4-7 run one
//   barrier behind waves 0-3, so on every SIMD one wave is in its MFMA segment while its partner issues loads: the
//   address arithmetic, LDS reads and load issue of one hide behind the matrix work of the other.
//   The direct-to-LDS loads of K-step t+1 are issued during K-step t (and the first quarter of t+2), and waited for with a
//   counted vmcnt(2) once per K-step, never vmcnt(0): two quarter-tiles stay in flight across every barrier.
//
// Buffer safety (slot = the interval between two consecutive barriers; waves 0-3 are "g0", waves 4-7 "g1"; Lp / Mp = LOAD
// / MFMA segment of phase p; g0 runs L1 M1 L2 M2 L3 M3 L4 M4 of K-step t in slots 8t .. 8t+7, g1 in slots 8t+1 .. 8t+8):
//   reads of K-step t:   B fragments in L1, L2 (last: g1 in slot 8t+3, retired by its lgkmcnt wait at the start of 8t+4),
//                        A fragments in L1, L3 (last: g1 in slot 8t+5, retired at the start of 8t+6)
//   refills of buffer t&1 for K-step t+2: B rows 0-127 in L4(t) (slots 8t+6 / 8t+7), B rows 128-255 in L1(t+1) (8t+8 / 9),
//                        A rows 0-127 in L2(t+1) (8t+10 / 11), A rows 128-255 in L3(t+1) (8t+12 / 13): every refill is
//                        issued at least two barriers after the last read of the bytes it replaces has been retired
//   landing: every wave waits vmcnt(2) (all but the newest quarter-tile) before the barrier that ends slot 8t+15; the first
//                        read of K-step t+2 is g0's L1 in slot 8t+16, behind that barrier.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define PBK 64          // K elements per step
#define PP_BM 256
#define PP_BN 256
#define PP_BUF 65536    // bytes per LDS buffer: A rows 0-255 then B rows 0-255, 128 bytes each
#define PP_EPF 132      // floats per staged epilogue row (128 channels + 4: odd number of 16-byte slots)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void pp_glds16(uintptr_t src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ uintptr_t pp_sel(bool ok, uintptr_t a, uintptr_t z) {
    return z ^ ((z ^ a) & (uintptr_t)(-(long long)ok));
}
__device__ __forceinline__ void pp_divmod(int n, int d, float rcp, int& q, int& rem) {
    q = (int)((float)n * rcp);
    rem = n - q * d;
    if (rem < 0) { rem += d; --q; }
    if (rem >= d) { rem -= d; ++q; }
}

#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PP_BARRIER()                       \
    do {                                   \
        PP_FENCE();                        \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_s_barrier();      \
        asm volatile("" ::: "memory");     \
        PP_FENCE();                        \
    } while (0)

__global__ __launch_bounds__(512, 2) void conv_pp_kernel(ConvParams p, int nN, int total, int Ktot, int nsteps, int Ntot, int mgroup_flags) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];  // [2][PP_BUF]; reused by the epilogue as 8 x [32][PP_EPF] floats

    const int id = blockIdx.x;
    const int chunk = (total + 7) / 8;
    int wk = (id & 7) * chunk + (id >> 3);
    if (wk >= total) return;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    int split = 0;
    if (ksplit > 1) {
        const int q = wk / ksplit;
        split = wk - q * ksplit;
        wk = q;
    }
    const int ks_begin = ksplit > 1 ? (int)((long)nsteps * split / ksplit) : 0;
    const int ks_end = ksplit > 1 ? (int)((long)nsteps * (split + 1) / ksplit) : nsteps;
    int my = 0, bx;
    const int mgroup = mgroup_flags & 0xffff;
    if (mgroup > 1) {
        // groups of `mgroup` m-tiles walk the n-tiles together (m fastest inside a group): the workgroups that run side by
        // side on an XCD share one x tile in its L2
        const int per_g = mgroup * nN;
        int gm, r2;
        pp_divmod(wk, per_g, __builtin_amdgcn_rcpf((float)per_g), gm, r2);
        const int nM_ = (p.M + PP_BM - 1) / PP_BM;
        const int gcur = min(mgroup, nM_ - gm * mgroup);
        int q;
        pp_divmod(r2, gcur, __builtin_amdgcn_rcpf((float)gcur), bx, q);
        my = gm * mgroup + q;
    } else {
        int rem = wk;
        if (wk >= nN) pp_divmod(wk, nN, __builtin_amdgcn_rcpf((float)nN), my, rem);
        bx = rem;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wr = wid >> 2, wc = wid & 3;      // wave tile: rows wr*128 .. +128 of m, wc*64 .. +64 of n
    const int mbase = my * PP_BM, nbase = bx * PP_BN;
    const int S = p.in_stride, Cin = p.Cin;
    const int off0 = p.off0 + p.phase_off0[0];
    const uintptr_t wbase = (uintptr_t)p.w, xbase = (uintptr_t)p.x;
    const uintptr_t zero = (uintptr_t)ga_zero_page;

    // staging geometry: one load instruction of the workgroup covers 64 tile rows (8 rows x 128 B per wave);
    // this thread: tile rows srow, srow+64, srow+128, srow+192 of each operand, 16-byte slot tid&7 = logical piece `piece`
    const int srow = tid >> 3;
    const int piece = (tid & 7) ^ ((tid >> 4) & 7);
    const float rcpL = __builtin_amdgcn_rcpf((float)p.Lout);
    int aoff[4];   // element offset of the row's first weight, -1: row past M
    int rbo[4];    // element offset of (tap 0, channel 0) of the row's window (may point before the item)
    int bR[4];     // input row of tap 0 (a huge value for rows past the end: every bounds test fails)
    int xit[4];    // element offset of the row's batch item (reflect padding only)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mbase + i * 64 + srow;
        aoff[i] = m < p.M ? m * Ktot : -1;
        const int n = nbase + i * 64 + srow;
        int b, tl;
        pp_divmod(n < Ntot ? n : 0, p.Lout, rcpL, b, tl);
        xit[i] = b * p.Lin * Cin;
        bR[i] = n < Ntot ? (p.t0 + tl) * S + off0 : 0x3fffffff;
        rbo[i] = xit[i] + (n < Ntot ? bR[i] * Cin : 0);
    }
    const int tapE = p.tap_step * Cin;  // elements between taps in a row window
    int st_dj, st_dc;
    pp_divmod(PBK, Cin, __builtin_amdgcn_rcpf((float)Cin), st_dj, st_dc);
    const int st_ddR = st_dj * p.tap_step, st_dkk = st_dj * tapE + st_dc;
    // K order: (tap, channel) 64 at a time, or "tap-fast" when a tap holds several 64-channel chunks (see conv_gemm.hip)
    const bool tapfast = !(mgroup_flags & 0x10000) && p.J > 1 && Cin > PBK && Cin % PBK == 0;
    int st_k, st_c, st_dR, st_kk, st_j = 0;
    if (tapfast) {
        int cc0;
        pp_divmod(ks_begin, p.J, __builtin_amdgcn_rcpf((float)p.J), cc0, st_j);
        st_c = cc0 * PBK + piece * 8;
        st_k = st_j * Cin + st_c;
        st_dR = st_j * p.tap_step;
        st_kk = st_j * tapE + st_c;
    } else {
        st_k = ks_begin * PBK + piece * 8;
        int j0;
        pp_divmod(st_k, Cin, __builtin_amdgcn_rcpf((float)Cin), j0, st_c);
        st_dR = j0 * p.tap_step;
        st_kk = j0 * tapE + st_c;
    }
    const bool reflect = p.pad_mode == 1;

    // two direct-to-LDS loads: rows q*128 + {0, 64} + srow of operand A (weights) / B (gathered activations) of the K-step
    // the gather state stands at, into LDS buffer `buf`
    auto issue_A = [&](int q, int buf) {
        const bool kin = st_k < Ktot;
        char* const dst = smem + buf * PP_BUF + q * 16384 + wid * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ao = aoff[q * 2 + i];
            pp_glds16(pp_sel(kin && ao >= 0, wbase + ((uintptr_t)(unsigned)(ao + st_k) << 1), zero), dst + i * 8192);
        }
    };
    auto issue_B = [&](int q, int buf) {
        const bool kin = st_k < Ktot;
        char* const dst = smem + buf * PP_BUF + 32768 + q * 16384 + wid * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int R = bR[q * 2 + i] + st_dR;
            if (!reflect) {
                const bool ok = kin && R >= 0 && R < p.Lin;
                pp_glds16(pp_sel(ok, xbase + ((uintptr_t)(unsigned)(rbo[q * 2 + i] + st_kk) << 1), zero), dst + i * 8192);
            } else {
                if (R < 0) R = -R;
                const bool ok = kin && R < p.Lin;
                pp_glds16(pp_sel(ok, xbase + ((uintptr_t)(unsigned)(xit[q * 2 + i] + R * Cin + st_c) << 1), zero), dst + i * 8192);
            }
        }
    };
    auto advance = [&]() {
        if (tapfast) {
            st_k += Cin;
            st_dR += p.tap_step;
            st_kk += tapE;
            if (++st_j == p.J) {  // next channel chunk, back to tap 0
                st_j = 0;
                st_c += PBK;
                st_k += PBK - p.J * Cin;
                st_dR = 0;
                st_kk += PBK - p.J * tapE;
            }
            return;
        }
        st_k += PBK;
        st_c += st_dc;
        st_dR += st_ddR;
        st_kk += st_dkk;
        if (st_c >= Cin) {
            st_c -= Cin;
            st_dR += p.tap_step;
            st_kk += tapE - Cin;
        }
    };

    // this lane's 8 bias values (the epilogue's channel group depends on the lane only)
    const int m0_ = mbase + wr * 128 + (lane & 15) * 8;
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias && ksplit == 1 && m0_ < p.M) {
        const float4 b0 = *(const float4*)(p.bias + m0_), b1 = *(const float4*)(p.bias + m0_ + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
        bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    const int swz = (r >> 1) & 7;
    const int aoffs = (wr * 128 + r) * 128, boffs = 32768 + (wc * 64 + r) * 128;
    bf16x8 fa[2][4], fb[2][4];  // A fragments of the current 64-row half [mi][kk]; B fragments of both 32-row halves [ni][kk]

#define PP_READ_A(buf, half)                                                                          \
    _Pragma("unroll") for (int mi_ = 0; mi_ < 2; ++mi_)                                               \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                               \
        fa[mi_][kk_] = *(const bf16x8*)(smem + (buf) * PP_BUF + aoffs + ((half) * 2 + mi_) * 4096 + (((kk_ * 2 + h) ^ swz) << 4));
#define PP_READ_B(buf, half)                                                                          \
    _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                               \
        fb[half][kk_] = *(const bf16x8*)(smem + (buf) * PP_BUF + boffs + (half) * 4096 + (((kk_ * 2 + h) ^ swz) << 4));
#define PP_MFMA(ahalf, bhalf)                                                                         \
    do {                                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                \
        _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                           \
        _Pragma("unroll") for (int mi_ = 0; mi_ < 2; ++mi_)                                           \
            acc[(ahalf) * 2 + mi_][bhalf] =                                                           \
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi_][kk_], fb[bhalf][kk_], acc[(ahalf) * 2 + mi_][bhalf], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                \
    } while (0)
    // one K-step out of buffer `cur`; `nxt` is the other buffer.  `live`: this K-step exists (wave-uniform).
#define PP_KSTEP(cur, nxt, live)                                                                      \
    do {                                                                                              \
        if (live) { PP_READ_A(cur, 0) PP_READ_B(cur, 0) }                                             \
        issue_B(1, nxt);                                                                              \
        PP_BARRIER();                                                                                 \
        if (live) PP_MFMA(0, 0);                                                                      \
        PP_BARRIER();                                                                                 \
        if (live) { PP_READ_B(cur, 1) }                                                               \
        issue_A(0, nxt);                                                                              \
        PP_BARRIER();                                                                                 \
        if (live) PP_MFMA(0, 1);                                                                      \
        PP_BARRIER();                                                                                 \
        if (live) { PP_READ_A(cur, 1) }                                                               \
        issue_A(1, nxt);                                                                              \
        advance();                                                                                    \
        PP_BARRIER();                                                                                 \
        if (live) PP_MFMA(1, 1);                                                                      \
        PP_BARRIER();                                                                                 \
        issue_B(0, cur);                                                                              \
        if (wr == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                 \
        PP_BARRIER();                                                                                 \
        if (live) PP_MFMA(1, 0);                                                                      \
        if (wr == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                 \
        PP_BARRIER();                                                                                 \
    } while (0)

    // prologue: K-step ks_begin into buffer 0, the first quarter of the next one into buffer 1
    issue_B(0, 0);
    issue_B(1, 0);
    issue_A(0, 0);
    issue_A(1, 0);
    advance();
    issue_B(0, 1);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    PP_BARRIER();
    if (wr == 1) PP_BARRIER();  // waves 4-7 run one barrier behind waves 0-3
    for (int ks = ks_begin; ks < ks_end; ks += 2) {
        PP_KSTEP(0, 1, true);
        const bool second = ks + 1 < ks_end;
        PP_KSTEP(1, 0, second);
    }
    if (wr == 0) PP_BARRIER();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: accumulators -> wave-private LDS rows [n][m] -> 16-byte, row-contiguous global accesses ----
    __syncthreads();  // every wave is done with the operand tiles (and every load into them has landed)
    if (mbase + wr * 128 >= p.M) return;
    float* const stage = (float*)smem + wid * (32 * PP_EPF);
    bf16_t* __restrict__ y = (bf16_t*)p.y;
    const bf16_t* __restrict__ res = (const bf16_t*)p.res;
    const bf16_t* __restrict__ ds = (const bf16_t*)p.dact_src;
    const int erow = lane >> 4, ec8 = (lane & 15) * 8;  // 16 lanes per staged row (8 output channels each), 4 rows per pass
    const int m0 = mbase + wr * 128 + ec8;
    const size_t item_elems = p.y_batch_elems > 0 ? (size_t)p.y_batch_elems : (size_t)p.LoutTot * p.M;
    const bool simple_act = p.post_act == GA_ACT_NONE || p.post_act == GA_ACT_LRELU;
    const float slope = p.post_act == GA_ACT_LRELU ? 0.2f : 1.f;
    const float fm_c = (p.res_mode == GA_RES_FM_SIGN && ksplit == 1) ? p.res_coef * p.res_coef_dev[0] : 0.f;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4*)(stage + r * PP_EPF + mi * 32 + 8 * q + 4 * h) =
                    make_float4(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 4 + erow;
            const int n = nbase + wc * 64 + ni * 32 + row;
            if (n >= Ntot || m0 >= p.M) continue;
            int b, tl;
            pp_divmod(n, p.Lout, rcpL, b, tl);
            const int orow = (p.t0 + tl) * p.out_stride;
            if (orow < 0 || orow >= p.LoutTot) continue;
            const size_t in_item = (size_t)orow * p.M + m0;
            if (in_item + 8 > item_elems) continue;  // (folded phases: the last row may run past the item)
            const size_t o = (size_t)b * item_elems + in_item;
            float v[8];
            *(float4*)&v[0] = *(const float4*)(stage + row * PP_EPF + ec8);
            *(float4*)&v[4] = *(const float4*)(stage + row * PP_EPF + ec8 + 4);
            if (ksplit > 1) {  // partial sums only, into this slice's own slab; ga_conv_finish adds the slabs and applies the epilogue
                float* wo = p.ws + (size_t)split * ((size_t)p.B * item_elems) + o;
                *(float4*)wo = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(wo + 4) = make_float4(v[4], v[5], v[6], v[7]);
                continue;
            }
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
            if (p.out_f32) {
                float* yo = (float*)p.y + o;
                *(float4*)yo = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(yo + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
                bf16x8 ov;
#pragma unroll
                for (int i = 0; i < 8; ++i) ov[i] = (__bf16)v[i];
                *(bf16x8*)(y + o) = ov;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int ga_conv_gemm_supported(const ConvParams& p);

// The ping-pong kernel takes what the gathered-GEMM kernel takes, single phase, with 32-bit element offsets.
int ga_conv_pp_supported(const ConvParams& p) {
    if (!ga_conv_gemm_supported(p) || p.nphase != 1 || p.res_mode == GA_RES_ADD_THEN_ACT || p.y2) return 0;
    if ((long)p.B * p.Lin * p.Cin >= (1L << 31) - (1L << 24) || (long)p.M * p.J * p.Cin >= (1L << 31) - (1L << 24)) return 0;
    if ((long)(labs((long)p.off0) + (long)p.J * p.tap_step + 2) * p.Cin >= (1L << 24)) return 0;  // window offsets stay far from overflow
    return 1;
}

static int pp_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Should ga_conv hand this problem to the ping-pong kernel?  It wants >= 128 output channels' worth of 256-row m-tiles that
// are not mostly padding, a K loop long enough to amortise its prologue and un-overlapped epilogue (one workgroup per CU),
// and about a CU-count of workgroups or more.
int ga_conv_pp_wanted(const ConvParams& p) {
    static int mode = -1, min_steps = -1, min_blocks = -1;
    if (mode < 0) {
        mode = pp_env("GA_GEMM_PP", 0);
        min_steps = pp_env("GA_GEMM_PP_MIN_STEPS", 16);
        min_blocks = pp_env("GA_GEMM_PP_MIN_BLOCKS", 192);
    }
    if (!mode || !ga_conv_pp_supported(p)) return 0;
    if (p.M < 192 || (p.M % 256 != 0 && p.M % 256 < 192)) return 0;
    const int nsteps = (p.J * p.Cin + PBK - 1) / PBK;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    if (nsteps / ksplit < min_steps) return 0;
    const long blocks = (((long)p.B * p.Lout + PP_BN - 1) / PP_BN) * ((p.M + PP_BM - 1) / PP_BM) * ksplit;
    return blocks >= min_blocks;
}

int ga_conv_pp_impl(const ConvParams& p, hipStream_t s) {
    if (!ga_conv_pp_supported(p) || !p.x || !p.w || !p.y) return GA_ERR_ARG;
    if (p.Lout == 0) return GA_OK;
    const int Ntot = p.B * p.Lout;
    const int nN = (Ntot + PP_BN - 1) / PP_BN, nM = (p.M + PP_BM - 1) / PP_BM;
    const int Ktot = p.J * p.Cin;
    const int nsteps = (Ktot + PBK - 1) / PBK;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    if (ksplit > 1 && (!p.ws || ((uintptr_t)p.ws & 15) || ksplit > nsteps)) return GA_ERR_ARG;
    const long total = (long)nN * nM * ksplit;
    if (total <= 0) return GA_OK;
    if (total > (1L << 30)) return GA_ERR_ARG;
    const size_t lds = 8 * 32 * PP_EPF * sizeof(float);  // 135168 >= 2 * PP_BUF
    static_assert(8 * 32 * PP_EPF * sizeof(float) >= 2 * PP_BUF, "epilogue staging covers the operand buffers");
    if (hipFuncSetAttribute((const void*)conv_pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GA_ERR_LAUNCH;
    static int mg = -1;
    if (mg < 0) {
        mg = pp_env("GA_GEMM_PP_MGROUP", 4);
        if (mg < 1) mg = 1;
        const char* t = getenv("GA_GEMM_TAPFAST");
        if (t && t[0] == '0') mg |= 0x10000;
    }
    const int grid = (int)((total + 7) / 8 * 8);
    conv_pp_kernel<<<grid, 512, lds, s>>>(p, nN, (int)total, Ktot, nsteps, Ntot, ksplit > 1 ? (1 | (mg & 0x10000)) : mg);
    GA_CHECK_LAUNCH();
    return GA_OK;
}
