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
typedef __attribute__((ext_vector_type(4))) float f32x4;

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

// MF: the MFMA shape.  32: v_mfma_f32_32x32x16_bf16 (wave tile = 4 x 2 tiles), 16: v_mfma_f32_16x16x32_bf16 (8 x 4 tiles): same
// operand bytes and the same cycles per flop, but the chip holds a higher clock on the 16x16x32 shape under load
// (MI355X_MICROARCH.md, DVFS give-back item 7: ~1.12-1.15x the FLOP/s at equal cycles with operands re-read from LDS).
template <int MF>
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
    // Operand staging: direct-to-LDS BUFFER loads (buffer_load_dwordx4 ... lds).  The per-lane part of every address is a
    // 32-bit byte offset that never changes (row start + 16-byte piece); the K position is wave-uniform and goes into the
    // instruction's scalar offset.  Padding needs no address select: a lane whose input row is outside its batch item gets
    // an offset beyond the buffer, and the hardware's range check returns zeros for it (rows past M / past the last output
    // row likewise, for good).  Per load that leaves add + compare + select for the gathered operand and nothing at all
    // for the weights -- the flat-address form this replaces spent ~10 vector and ~12 scalar instructions per load on the
    // select against a zero page, and with 8 loads per K-step and wave the LOAD segments, not the MFMA segments, set the
    // pace (ablation: 1600 TFLOP/s without the loads, 1100 with them served from cache; tools/pp_ab.py).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long)p.M * Ktot * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((long)p.B * p.Lin * Cin * 2), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;  // beyond any buffer this kernel accepts (< 2^31 bytes), with or without the scalar offset

    // staging geometry: one load instruction of the workgroup covers 64 tile rows (8 rows x 128 B per wave);
    // this thread: tile rows srow, srow+64, srow+128, srow+192 of each operand, 16-byte slot tid&7 = logical piece `piece`
    const int srow = tid >> 3;
    const int piece = (tid & 7) ^ ((tid >> 4) & 7);
    const float rcpL = __builtin_amdgcn_rcpf((float)p.Lout);
    unsigned avo[4];   // byte offset of (row m, this lane's piece) in the packed weights; OOB: row past M
    unsigned bvo[4];   // byte offset of (tap 0, channel 0, this lane's piece) of the row's window (may "point" before the item)
    int bR[4];         // input row of tap 0 (a huge value for rows past the end: every bounds test fails)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = mbase + i * 64 + srow;
        avo[i] = m < p.M ? (unsigned)(m * Ktot + piece * 8) * 2u : OOB;
        const int n = nbase + i * 64 + srow;
        int b, tl;
        pp_divmod(n < Ntot ? n : 0, p.Lout, rcpL, b, tl);
        bR[i] = n < Ntot ? (p.t0 + tl) * S + off0 : 0x3fffffff;
        bvo[i] = (unsigned)(b * p.Lin * Cin + (n < Ntot ? bR[i] * Cin : 0) + piece * 8) * 2u;
    }
    const int tapE = p.tap_step * Cin;  // elements between taps in a row window
    int st_dj, st_dc;
    pp_divmod(PBK, Cin, __builtin_amdgcn_rcpf((float)Cin), st_dj, st_dc);
    st_dj = __builtin_amdgcn_readfirstlane(st_dj);
    st_dc = __builtin_amdgcn_readfirstlane(st_dc);
    const int st_ddR = st_dj * p.tap_step, st_dkk = st_dj * tapE + st_dc;
    // K order: (tap, channel) 64 at a time, or "tap-fast" when a tap holds several 64-channel chunks (see conv_gemm.hip)
    // strided layers: "phase-fast" tap order within a channel chunk (taps phi, phi+S, phi+2S, ... for phi = 0 .. S-1), so that
    // consecutive stages gather the rows of the stage before from L2 (conv_gemm.hip has the measurements)
    const int TS = (mgroup_flags & 0x20000) && S > 1 && S <= 8 && p.tap_step == 1 && p.J > S ? S : 1;
    const bool tapfast = !(mgroup_flags & 0x10000) && p.J > 1 && (TS > 1 ? Cin >= PBK : Cin > PBK) && Cin % PBK == 0;
    // the K position (all wave-uniform, kept in scalar registers): element offset st_k into a weight row, tap row offset
    // st_dR, element offset st_kk into a row window, channel st_c / tap st_j / phase st_phi of the stage to be issued next
    int st_k, st_c, st_dR, st_kk, st_j = 0, st_phi = 0;
    if (tapfast) {
        int cc0;
        pp_divmod(ks_begin, p.J, __builtin_amdgcn_rcpf((float)p.J), cc0, st_j);
        if (TS > 1) {  // st_j is the position in the chunk's tap sequence: find its phase and tap
            int t = st_j;
            while (t >= (p.J - st_phi + TS - 1) / TS) {
                t -= (p.J - st_phi + TS - 1) / TS;
                ++st_phi;
            }
            st_j = st_phi + t * TS;
        }
        st_c = cc0 * PBK;
        st_k = st_j * Cin + st_c;
        st_dR = st_j * p.tap_step;
        st_kk = st_j * tapE + st_c;
    } else {
        st_k = ks_begin * PBK;
        int j0;
        pp_divmod(st_k, Cin, __builtin_amdgcn_rcpf((float)Cin), j0, st_c);
        st_dR = j0 * p.tap_step;
        st_kk = j0 * tapE + st_c;
    }
    st_k = __builtin_amdgcn_readfirstlane(st_k);
    st_c = __builtin_amdgcn_readfirstlane(st_c);
    st_dR = __builtin_amdgcn_readfirstlane(st_dR);
    st_kk = __builtin_amdgcn_readfirstlane(st_kk);
    st_j = __builtin_amdgcn_readfirstlane(st_j);
    st_phi = __builtin_amdgcn_readfirstlane(st_phi);

    // two direct-to-LDS loads: rows q*128 + {0, 64} + srow of operand A (weights) / B (gathered activations) of the K-step
    // the gather state stands at, into LDS buffer `buf`.  K-steps past the end are issued like any other (the counted waits
    // rely on a fixed number of loads per K-step): they land in a buffer nobody reads, and the range check keeps them in bounds.
    const bool abl_noload = mgroup_flags & 0x80000, abl_noread = mgroup_flags & 0x100000;  // timing experiments (wrong results)
    auto issue_A = [&](int q, int buf) {
        if (abl_noload) return;
        char* const dst = smem + buf * PP_BUF + q * 16384 + wid * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(dst + i * 8192), 16, avo[q * 2 + i], st_k * 2, 0, 0);
    };
    auto issue_B = [&](int q, int buf) {
        if (abl_noload) return;
        char* const dst = smem + buf * PP_BUF + 32768 + q * 16384 + wid * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = (unsigned)(bR[q * 2 + i] + st_dR) < (unsigned)p.Lin;
            // (tap offset added on the vector side, wrapping: a window that starts in the left padding has a "negative" bvo)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(dst + i * 8192), 16, ok ? bvo[q * 2 + i] + (unsigned)(st_kk * 2) : OOB, 0, 0, 0);
        }
    };
    auto advance = [&]() {
        if (mgroup_flags & 0x40000) return;  // timing experiment (GA_GEMM_PP_NOADV=1): every K-step re-reads the first one's bytes (wrong results)
        if (tapfast && TS > 1) {
            st_j += TS;
            if (st_j >= p.J) {  // next phase; after the last one the next channel chunk
                if (++st_phi == TS) {
                    st_phi = 0;
                    st_c += PBK;
                }
                st_j = st_phi;
            }
            st_k = st_j * Cin + st_c;
            st_dR = st_j;  // (tap_step == 1)
            st_kk = st_j * tapE + st_c;
            return;
        }
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

    // accumulators of the 128 x 64 wave tile: MF == 32: acc32[mi][ni] (4 x 2 tiles of 32 x 32), MF == 16: acc16[mt][nt] (8 x 4 of 16 x 16)
    f32x16 acc32[MF == 32 ? 4 : 1][MF == 32 ? 2 : 1];
    f32x4 acc16[MF == 16 ? 8 : 1][MF == 16 ? 4 : 1];
    if constexpr (MF == 32) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc32[mi][ni][i] = 0.f;
    } else {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc16[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // fragment geometry.  MF == 32: lane -> tile row lane & 31, 16-byte k piece (2 kk + (lane >> 5)), kk = 0..3.
    //                     MF == 16: lane -> tile row lane & 15, 16-byte k piece (4 kh + (lane >> 4)), kh = 0..1.
    // Either way the piece sits in LDS slot piece ^ ((row >> 1) & 7): conflict-free for both read shapes.
    const int fr = MF == 32 ? r : (lane & 15), fh = MF == 32 ? h : (lane >> 4);
    const int swz = (fr >> 1) & 7;
    const int aoffs = (wr * 128 + fr) * 128, boffs = 32768 + (wc * 64 + fr) * 128;
    // fragments of one 64-row half of A and of both 32-row halves of B: 8 x 16 bytes each way, whatever the shape
    bf16x8 fa[8], fb[2][4];

#define PP_READ_A(buf, half)                                                                          \
    do {                                                                                              \
        if constexpr (MF == 32) {                                                                     \
            _Pragma("unroll") for (int mi_ = 0; mi_ < 2; ++mi_)                                       \
            _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                       \
                fa[mi_ * 4 + kk_] = *(const bf16x8*)(smem + (buf) * PP_BUF + aoffs + ((half) * 2 + mi_) * 4096 + (((kk_ * 2 + fh) ^ swz) << 4)); \
        } else {                                                                                      \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_)                                       \
            _Pragma("unroll") for (int kh_ = 0; kh_ < 2; ++kh_)                                       \
                fa[mt_ * 2 + kh_] = *(const bf16x8*)(smem + (buf) * PP_BUF + aoffs + ((half) * 4 + mt_) * 2048 + (((kh_ * 4 + fh) ^ swz) << 4)); \
        }                                                                                             \
    } while (0);
#define PP_READ_B(buf, half)                                                                          \
    do {                                                                                              \
        if constexpr (MF == 32) {                                                                     \
            _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                       \
                fb[half][kk_] = *(const bf16x8*)(smem + (buf) * PP_BUF + boffs + (half) * 4096 + (((kk_ * 2 + fh) ^ swz) << 4)); \
        } else {                                                                                      \
            _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)                                       \
            _Pragma("unroll") for (int kh_ = 0; kh_ < 2; ++kh_)                                       \
                fb[half][nt_ * 2 + kh_] = *(const bf16x8*)(smem + (buf) * PP_BUF + boffs + ((half) * 2 + nt_) * 2048 + (((kh_ * 4 + fh) ^ swz) << 4)); \
        }                                                                                             \
    } while (0);
#define PP_MFMA(ahalf, bhalf)                                                                         \
    do {                                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                \
        if constexpr (MF == 32) {                                                                     \
            _Pragma("unroll") for (int kk_ = 0; kk_ < 4; ++kk_)                                       \
            _Pragma("unroll") for (int mi_ = 0; mi_ < 2; ++mi_)                                       \
                acc32[(ahalf) * 2 + mi_][bhalf] =                                                     \
                    __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi_ * 4 + kk_], fb[bhalf][kk_], acc32[(ahalf) * 2 + mi_][bhalf], 0, 0, 0); \
        } else {                                                                                      \
            _Pragma("unroll") for (int kh_ = 0; kh_ < 2; ++kh_)                                       \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_)                                       \
            _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)                                       \
                acc16[(ahalf) * 4 + mt_][(bhalf) * 2 + nt_] =                                         \
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt_ * 2 + kh_], fb[bhalf][nt_ * 2 + kh_], acc16[(ahalf) * 4 + mt_][(bhalf) * 2 + nt_], 0, 0, 0); \
        }                                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                \
    } while (0)
    // one K-step out of buffer `cur`; `nxt` is the other buffer.  `live`: this K-step exists (wave-uniform).
#define PP_KSTEP(cur, nxt, live)                                                                      \
    do {                                                                                              \
        if (live && !abl_noread) { PP_READ_A(cur, 0) PP_READ_B(cur, 0) }                              \
        issue_B(1, nxt);                                                                              \
        PP_BARRIER();                                                                                 \
        if (live) PP_MFMA(0, 0);                                                                      \
        PP_BARRIER();                                                                                 \
        if (live && !abl_noread) { PP_READ_B(cur, 1) }                                                \
        issue_A(0, nxt);                                                                              \
        PP_BARRIER();                                                                                 \
        if (live) PP_MFMA(0, 1);                                                                      \
        PP_BARRIER();                                                                                 \
        if (live && !abl_noread) { PP_READ_A(cur, 1) }                                                \
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

    if (abl_noread) {  // (fragments defined once)
        PP_READ_A(0, 0) PP_READ_B(0, 0) PP_READ_B(0, 1)
    }
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
        if constexpr (MF == 32) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(float4*)(stage + r * PP_EPF + mi * 32 + 8 * q + 4 * h) =
                        make_float4(acc32[mi][ni][4 * q], acc32[mi][ni][4 * q + 1], acc32[mi][ni][4 * q + 2], acc32[mi][ni][4 * q + 3]);
        } else {
            // 16 x 16 accumulator: lane holds channels m = 16 mt + 4 (lane >> 4) + i of output row n = 16 nt + (lane & 15)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
                    *(float4*)(stage + (nt * 16 + fr) * PP_EPF + mt * 16 + 4 * fh) =
                        make_float4(acc16[mt][ni * 2 + nt][0], acc16[mt][ni * 2 + nt][1], acc16[mt][ni * 2 + nt][2], acc16[mt][ni * 2 + nt][3]);
        }
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
    if (p.pad_mode != 0 || p.Cin % PBK != 0) return 0;  // zero padding by the buffer range check; a K-step lies inside one tap (wave-uniform K position)
    if ((long)p.B * p.Lin * p.Cin * 2 >= (1L << 31) - 64 || (long)p.M * p.J * p.Cin * 2 >= (1L << 31) - 64) return 0;  // 32-bit byte offsets
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
        // on by default since round 3 (16x16x32 MFMAs + buffer-load staging): it wins from K ~ 5000 up once the grid is about a
        // workgroup per CU (same-process A/B at B = 128, tools/pp_ab.py: D blocks.4 forward 1300 vs 1077 TFLOP/s, blocks.3 forward
        // 1094-1205 vs 1038-1090, blocks.4 data gradient 1136 vs 1046; shorter K and half-empty grids stay on the 128x128 kernel)
        mode = pp_env("GA_GEMM_PP", 1);
        min_steps = pp_env("GA_GEMM_PP_MIN_STEPS", 128);  // (in the step, with its colder caches and fuller epilogues, only K >= 8192 wins: D blocks.4 forward)
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
    const int mf = pp_env("GA_GEMM_PP_MFMA", 16) == 32 ? 32 : 16;  // (read per launch: in-process A/B by os.environ)
    auto kern = mf == 32 ? conv_pp_kernel<32> : conv_pp_kernel<16>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GA_ERR_LAUNCH;
    int mg = -1;
    {
        mg = pp_env("GA_GEMM_PP_MGROUP", 4);
        if (mg < 1) mg = 1;
        const char* t = getenv("GA_GEMM_TAPFAST");
        if (t && t[0] == '0') mg |= 0x10000;
        const char* ph = getenv("GA_GEMM_PHASEFAST");  // "0": plain tap order on strided layers too
        if (!(ph && ph[0] == '0')) mg |= 0x20000;
        if (pp_env("GA_GEMM_PP_NOADV", 0)) mg |= 0x40000;
        if (pp_env("GA_GEMM_PP_NOLOAD", 0)) mg |= 0x80000;
        if (pp_env("GA_GEMM_PP_NOREAD", 0)) mg |= 0x100000;
    }
    const int grid = (int)((total + 7) / 8 * 8);
    kern<<<grid, 512, lds, s>>>(p, nN, (int)total, Ktot, nsteps, Ntot, ksplit > 1 ? (1 | (mg & 0x1f0000)) : mg);
    GA_CHECK_LAUNCH();
    return GA_OK;
}
