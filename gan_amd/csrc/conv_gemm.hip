// bf16 MFMA implicit-GEMM kernel, "gathered GEMM" form, for the generalised NLC convolution (gfx950).
//
//   Y[n][m] = sum_k  Wp[m][k] * X[n][k]        n = (batch item, output step) flattened, k = (tap j, channel c)
//
// Channels-last activations make the im2col row of output step t a set of J contiguous Cin-runs of x, so the
// B operand needs no window in LDS: every 16-byte piece (8 channels of one input row) is fetched straight into
// its LDS slot by a direct-to-LDS load (global_load_lds_dwordx4) whose per-lane SOURCE address does the im2col
// gather (zero padding and out-of-range pieces read a 16-byte zero page; reflect padding mirrors the row).
// Re-reads of x between taps are served by L2, exactly like the B tile re-reads of a plain GEMM.
//
//   block tile  BM x BN = (WM*64) x (WN*64), 4 waves, each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16
//   K-step 64, two LDS buffers per operand; loads of K-step k+1 are in flight while k is multiplied
//   LDS image: [rows][64 k] bf16 = 128-byte rows, lane-linear per load instruction (8 rows x 128 B = 1 KiB),
//   16-byte slot s of row r holds logical piece s ^ ((r>>1)&7): with that XOR the ds_read_b128 of a 32-row
//   fragment is bank-conflict-free (cdna_hip_programming.md 5.5 T2; the XOR is applied on the SOURCE address).
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GBK 64  // K elements per step


typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void glds16(uintptr_t src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}
// 16 bytes per lane, buffer resource + 32-bit vector offset + scalar offset -> LDS (wave-uniform base + lane * 16).  (A plain
// function on purpose: called from a value-dependent context -- `if constexpr` inside a kernel template -- the builtin fails
// the HOST pass's target-feature check at instantiation time and the whole instantiation silently loses its host stub.)
__device__ __forceinline__ void bglds16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
// branch-free "ok ? a : z" on addresses (a ternary on pointers makes hipcc branch around two separate loads)
__device__ __forceinline__ uintptr_t sel_addr(bool ok, uintptr_t a, uintptr_t z) {
    return z ^ ((z ^ a) & (uintptr_t)(-(long long)ok));
}

// n -> (n / d, n % d) for 0 <= n < 2^30 and quotients far below 2^20: reciprocal estimate, one correction each way.
// (An integer division is ~40 dependent instructions; with one wave per SIMD -- small grids -- they are not hidden.)
__device__ __forceinline__ void divmod_small(int n, int d, float rcp, int& q, int& rem) {
    q = (int)((float)n * rcp);
    rem = n - q * d;
    if (rem < 0) { rem += d; --q; }
    if (rem >= d) { rem -= d; ++q; }
}

// MI: 32-row MFMA tiles per wave along m (2: wave tile 64 x 64, 1: wave tile 32 x 64 for layers with <= 32 output channels)
// waves per SIMD the single-buffer 128x128 instantiation is compiled for (4: 128 registers, four workgroups per CU hide
// each other's load round trips; 3: 144 registers, no spills)
#ifndef GA_GEMM_NBUF1_WAVES
#define GA_GEMM_NBUF1_WAVES 4
#endif
#ifndef GA_GEMM_SETPRIO
#define GA_GEMM_SETPRIO 0  // measured: no change in kernel time at 1-3 workgroups per CU
#endif
static constexpr bool SETPRIO = GA_GEMM_SETPRIO;

// BUF: operand staging by direct-to-LDS BUFFER loads (zero padding, whole K-steps, tensors under 2 GiB): the per-lane part of an
// address is a constant 32-bit byte offset, the K position a scalar offset, and padding is the hardware's range check (a lane
// whose input row lies outside its batch item gets an offset beyond the buffer and receives zeros) -- per load: add + compare +
// select for the gathered operand, nothing for the weights, where the flat-address form selects a 64-bit address against a zero
// page with ~10 vector and ~12 scalar instructions per load.
template <int MI, int WM, int WN, int NBUF, bool BUF = false>
__global__ __launch_bounds__(WM * WN * 64, ((NBUF == 1 && WM == 2 && WN == 2 && MI == 2) ? GA_GEMM_NBUF1_WAVES : 1)) void conv_gemm_kernel(ConvParams p, int nN, int total, int Ktot, int nsteps, int Ntot, int mgroup_flags) {
    constexpr int WMR = MI * 32;               // rows (m) per wave
    constexpr int BM = WM * WMR, BN = WN * 64;
    constexpr int NT = WM * WN * 64;           // threads: WM x WN waves
    constexpr int RPI = NT / 8;                // tile rows one load instruction of the workgroup covers
    constexpr int NA = (BM * 8 + NT - 1) / NT; // 16-byte pieces per thread per K-step, A (weights)
    constexpr int NB = BN * 8 / NT;            // same, B (gathered activations)
    constexpr int EPF = WMR + 4;               // floats per staged epilogue row (odd number of 16-byte slots)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    // [NBUF][BM*128] A, then [NBUF][BN*128] B; reused by the epilogue as 4 x [32][EPF] floats
    char* const Al = smem;
    char* const Bl = smem + NBUF * BM * 128;

    const int id = blockIdx.x;
    const int chunk = (total + 7) / 8;
    int wk = (id & 7) * chunk + (id >> 3);
    if (wk >= total) return;
    // split-K: consecutive work items are the K slices of one output tile
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    int split = 0;
    if (ksplit > 1) {
        const int q = wk / ksplit;
        split = wk - q * ksplit;
        wk = q;
    }
    const int ks_begin = ksplit > 1 ? (int)((long)nsteps * split / ksplit) : 0;
    const int ks_end = ksplit > 1 ? (int)((long)nsteps * (split + 1) / ksplit) : nsteps;
    int my = 0, rem = wk, phase = 0, bx;
    const int mgroup = mgroup_flags & 0xffff;
    if (mgroup > 1 && p.nphase == 1) {
        // groups of `mgroup` m-tiles walk the n-tiles together (m fastest inside a group): the workgroups that run side by
        // side on an XCD then share one x tile in its L2 instead of each streaming a different one
        const int per_g = mgroup * nN;
        int gm, r2;
        divmod_small(wk, per_g, __builtin_amdgcn_rcpf((float)per_g), gm, r2);
        const int nM_ = (p.M + BM - 1) / BM;
        const int gcur = min(mgroup, nM_ - gm * mgroup);
        int q;
        divmod_small(r2, gcur, __builtin_amdgcn_rcpf((float)gcur), bx, q);
        my = gm * mgroup + q;
    } else {
        const int per_m = nN * p.nphase;
        if (wk >= per_m) divmod_small(wk, per_m, __builtin_amdgcn_rcpf((float)per_m), my, rem);
        bx = rem;
        if (p.nphase > 1) divmod_small(rem, nN, __builtin_amdgcn_rcpf((float)nN), phase, bx);
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;
    const int mbase = my * BM, nbase = bx * BN;
    const int S = p.in_stride, Cin = p.Cin;
    const int off0 = p.off0 + p.phase_off0[phase];
    const bf16_t* __restrict__ w = (const bf16_t*)p.w + (size_t)phase * p.M * Ktot;
    const uintptr_t zero = (uintptr_t)ga_zero_page;

    // staging geometry: piece q = i*NT + tid -> row q>>3 = i*RPI + (tid>>3), slot tid&7, logical piece below
    const int srow = tid >> 3;
    const int piece = (tid & 7) ^ ((tid >> 4) & 7);
    // per-thread gather state.  The K position advances by 64 per stage, so tap j, channel c and the byte offset of
    // (j, c) inside a row window are carried from stage to stage instead of being re-derived (no division, no multiply).
    const float rcpL = __builtin_amdgcn_rcpf((float)p.Lout);
    uintptr_t abase[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = mbase + i * RPI + srow;
        abase[i] = (m < p.M && i * RPI + srow < BM) ? (uintptr_t)(w + (size_t)m * Ktot) : 0;
    }
    uintptr_t xitem[NB];  // start of the row's batch item (0: row past the end)
    uintptr_t rb[NB];     // xitem + bR*Cin elements: address of (tap 0, channel 0) of the row's window, possibly out of range
    int bR[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = nbase + i * RPI + srow;
        int b, tl;
        divmod_small(n < Ntot ? n : 0, p.Lout, rcpL, b, tl);
        xitem[i] = n < Ntot ? (uintptr_t)((const bf16_t*)p.x + (size_t)b * p.Lin * Cin) : 0;
        bR[i] = (p.t0 + tl) * S + off0;
        rb[i] = xitem[i] + (uintptr_t)((long)bR[i] * Cin * 2);
    }
    // BUF: constant per-lane byte offsets (row start + this lane's 16-byte piece); the K position below then leaves `piece` out
    constexpr unsigned OOB = 0x80000000u;
    unsigned avo[NA], bvo[NB];
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, BUF ? (int)((long)p.M * Ktot * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, BUF ? (int)((long)p.B * p.Lin * Cin * 2) : 0, 0x00020000);
    if constexpr (BUF) {
#pragma unroll
        for (int i = 0; i < NA; ++i) avo[i] = abase[i] ? (unsigned)(abase[i] - (uintptr_t)w) + piece * 16u : OOB;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            bvo[i] = (unsigned)(rb[i] - (uintptr_t)p.x) + piece * 16u;
            if (!xitem[i]) bR[i] = 0x3fffffff;  // rows past the end fail every bounds test
        }
    }
    const int pk = BUF ? 0 : piece * 8;  // (the flat form carries the lane's piece inside the K position)
    const int tapB = p.tap_step * Cin * 2;  // bytes between taps in a row window
    int st_dj, st_dc;
    divmod_small(GBK, Cin, __builtin_amdgcn_rcpf((float)Cin), st_dj, st_dc);
    const int st_ddR = st_dj * p.tap_step, st_dkk = st_dj * tapB + st_dc * 2;
    // K order.  Default: k = (tap, channel), 64 at a time.  With several 64-channel chunks per tap (Cin = 128 .. 1024) the
    // stages walk the TAPS of one channel chunk first ("tap-fast"): consecutive stages then gather the same x rows shifted
    // by one tap, so the re-read comes a stage later and hits L2, instead of Cin/64 stages later behind everything the other
    // workgroups of the XCD have streamed through it meanwhile (PMC: 11-tap layers fetched their input 11 times over).
    // Strided layers (the k41 stride-4 forward convs): tap j of output step n reads row S*n + j, so taps j and j + S read the SAME
    // rows one step apart.  "Phase-fast": within a channel chunk the taps go phi, phi+S, phi+2S, ... for phi = 0 .. S-1, so that
    // consecutive stages gather (all but one of) the rows of the stage before -- from L2 -- where plain tap order comes back to a
    // row S stages later, after the XCD's other workgroups have pushed ~16 MB through its 4 MB L2 (PMC, per launch at B = 128:
    // D blocks.4 forward fetched 1.45 GB from the fabric for 0.12 GB of input, blocks.2 / .3 the same ~10x).
    const int TS = (mgroup_flags & 0x20000) && S > 1 && S <= 8 && p.tap_step == 1 && p.J > S ? S : 1;
    const bool tapfast = !(mgroup_flags & 0x10000) && p.J > 1 && (TS > 1 ? Cin >= GBK : Cin > GBK) && Cin % GBK == 0;
    int st_k, st_c, st_dR, st_kk, st_j = 0, st_phi = 0;
    if (tapfast) {
        int cc0;
        divmod_small(ks_begin, p.J, __builtin_amdgcn_rcpf((float)p.J), cc0, st_j);
        if (TS > 1) {  // st_j is the position in the chunk's tap sequence: find its phase and tap
            int t = st_j;
            while (t >= (p.J - st_phi + TS - 1) / TS) {
                t -= (p.J - st_phi + TS - 1) / TS;
                ++st_phi;
            }
            st_j = st_phi + t * TS;
        }
        st_c = cc0 * GBK + pk;
        st_k = st_j * Cin + st_c;
        st_dR = st_j * p.tap_step;
        st_kk = st_j * tapB + st_c * 2;
    } else {
        st_k = ks_begin * GBK + pk;
        int j0;
        divmod_small(st_k, Cin, __builtin_amdgcn_rcpf((float)Cin), j0, st_c);
        st_dR = j0 * p.tap_step;
        st_kk = j0 * tapB + st_c * 2;
    }
    const bool reflect = p.pad_mode == 1;
    if constexpr (BUF) {  // the K position is wave-uniform: keep it in scalar registers
        st_k = __builtin_amdgcn_readfirstlane(st_k);
        st_c = __builtin_amdgcn_readfirstlane(st_c);
        st_dR = __builtin_amdgcn_readfirstlane(st_dR);
        st_kk = __builtin_amdgcn_readfirstlane(st_kk);
        st_j = __builtin_amdgcn_readfirstlane(st_j);
        st_phi = __builtin_amdgcn_readfirstlane(st_phi);
    }

    // issues the direct-to-LDS loads of the NEXT stage in sequence (stages are issued in K order, each exactly once)
    auto issue = [&](int buf) {
        const bool kin = st_k < Ktot;
        char* const ab = Al + buf * (BM * 128) + wid * 1024;
        char* const bb = Bl + buf * (BN * 128) + wid * 1024;
        if constexpr (BUF) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                bglds16(rsA, ab + i * (NT * 16), avo[i], st_k * 2);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const bool ok = (unsigned)(bR[i] + st_dR) < (unsigned)p.Lin;
                // (the tap offset is added on the vector side, in wrapping 32-bit arithmetic: a window that starts in the left
                // padding has a "negative" bvo, and only the sum is a valid offset)
                bglds16(rsB, bb + i * (NT * 16), ok ? bvo[i] + (unsigned)st_kk : OOB, 0);
            }
        } else {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            glds16(sel_addr(kin && abase[i], abase[i] + (uintptr_t)(st_k * 2), zero), ab + i * (NT * 16));
        }
        if (!reflect) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int R = bR[i] + st_dR;
                const bool ok = kin && xitem[i] && R >= 0 && R < p.Lin;
                glds16(sel_addr(ok, rb[i] + (uintptr_t)(long)st_kk, zero), bb + i * (NT * 16));
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                int R = bR[i] + st_dR;
                if (R < 0) R = -R;
                const bool ok = kin && xitem[i] && R < p.Lin;
                glds16(sel_addr(ok, xitem[i] + (uintptr_t)(((long)R * Cin + st_c) * 2), zero), bb + i * (NT * 16));
            }
        }
        }
        if (tapfast && TS > 1) {
            st_j += TS;
            if (st_j >= p.J) {  // next phase; after the last one the next channel chunk
                if (++st_phi == TS) {
                    st_phi = 0;
                    st_c += GBK;
                }
                st_j = st_phi;
            }
            st_k = st_j * Cin + st_c;
            st_dR = st_j;  // (tap_step == 1)
            st_kk = st_j * tapB + st_c * 2;
            return;
        }
        if (tapfast) {
            st_k += Cin;
            st_dR += p.tap_step;
            st_kk += tapB;
            if (++st_j == p.J) {  // next channel chunk, back to tap 0
                st_j = 0;
                st_c += GBK;
                st_k += GBK - p.J * Cin;
                st_dR = 0;
                st_kk += GBK * 2 - p.J * tapB;
            }
            return;
        }
        st_k += GBK;
        st_c += st_dc;
        st_dR += st_ddR;
        st_kk += st_dkk;
        if (st_c >= Cin) {
            st_c -= Cin;
            st_dR += p.tap_step;
            st_kk += tapB - Cin * 2;
        }
    };

    // this lane's 8 bias values (the epilogue's channel group depends on the lane only): fetched here, behind the whole K loop,
    // not once per epilogue pass in front of every store
    constexpr int LPR_ = (MI * 32) / 8;
    const int m0_ = mbase + wm * (MI * 32) + (lane % LPR_) * 8;
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias && ksplit == 1 && m0_ < p.M) {
        const float4 b0 = *(const float4*)(p.bias + m0_), b1 = *(const float4*)(p.bias + m0_ + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w;
        bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }

    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    const int swz = (r >> 1) & 7;
    const int aoff = (wm * WMR + r) * 128, boff = (wn * 64 + r) * 128;

    // NBUF >= 2: stages ks+1 .. ks+NBUF-1 are in flight while stage ks is multiplied (NBUF = 4 is for grids of about one
    // workgroup per CU, where nothing else on the CU hides the load latency).  One barrier per K-step: behind it every
    // wave's share of stage ks has landed and everyone is done reading the buffer that stage ks+NBUF-1 is about to refill.
    constexpr int DEPTH = NBUF - 1;           // stages in flight ahead of the one being multiplied
    constexpr int LPS = NA + NB;              // direct-to-LDS loads per thread per stage (the memory counter is in-order)
    if (NBUF >= 2) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (ks_begin + d < ks_end) issue(d);
    }
    int buf = 0;
    for (int ks = ks_begin; ks < ks_end; ++ks) {
        if (NBUF == 1) {
            if (ks > ks_begin) __syncthreads();  // everyone is done reading the single buffer
            issue(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (NBUF == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int ahead = ks_end - 1 - ks;  // stages issued after stage ks (at most DEPTH - 1 of them are in flight here)
            if (DEPTH >= 3 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
            else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (NBUF > 2) {
            // raw barrier: __syncthreads() would drain the memory counter (vmcnt(0)) and with it the stages in flight
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();
        }
        if (NBUF >= 2 && ks + DEPTH < ks_end) {
            int nb = buf + DEPTH;
            if (nb >= NBUF) nb -= NBUF;
            issue(nb);
        }
        const char* Ab = Al + buf * (BM * 128) + aoff;
        const char* Bb = Bl + buf * (BN * 128) + boff;
        if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int po = ((kk * 2 + h) ^ swz) << 4;
            bf16x8 a[MI], b[2];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = *(const bf16x8*)(Ab + mi * 32 * 128 + po);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = *(const bf16x8*)(Bb + ni * 32 * 128 + po);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (SETPRIO) __builtin_amdgcn_s_setprio(0);
        if (NBUF >= 2 && ++buf == NBUF) buf = 0;
    }

    // ---- epilogue: accumulators -> wave-private LDS rows [n][m] -> 16-byte, row-contiguous global accesses ----
    __syncthreads();  // every wave is done with the operand tiles; their LDS is reused below
    if (mbase + wm * WMR >= p.M) return;
    float* const stage = (float*)smem + wid * (32 * EPF);
    bf16_t* __restrict__ y = (bf16_t*)p.y;
    const bf16_t* __restrict__ res = (const bf16_t*)p.res;
    const bf16_t* __restrict__ ds = (const bf16_t*)p.dact_src;
    constexpr int LPR = WMR / 8;        // lanes per staged row (8 output channels each)
    constexpr int RPP = 64 / LPR;       // rows per pass
    const int erow = lane / LPR, ec8 = (lane % LPR) * 8;
    const int m0 = mbase + wm * WMR + ec8;
    const size_t item_elems = p.y_batch_elems > 0 ? (size_t)p.y_batch_elems : (size_t)p.LoutTot * p.M;
    const bool simple_act = p.post_act == GA_ACT_NONE || p.post_act == GA_ACT_LRELU;
    const float slope = p.post_act == GA_ACT_LRELU ? 0.2f : 1.f;
    const float fm_c = (p.res_mode == GA_RES_FM_SIGN && ksplit == 1) ? p.res_coef * p.res_coef_dev[0] : 0.f;
    const bool fast_elu = p.post_act == GA_ACT_ELU && !p.out_f32;  // (bf16 output)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
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
            const int n = nbase + wn * 64 + ni * 32 + row;
            if (n >= Ntot || m0 >= p.M) continue;
            int b, tl;
            divmod_small(n, p.Lout, rcpL, b, tl);
            const int orow = (p.t0 + tl) * p.out_stride + phase;
            if (orow < 0 || orow >= p.LoutTot) continue;
            const size_t in_item = (size_t)orow * p.M + m0;
            if (in_item + 8 > item_elems) continue;  // (folded phases: the last row may run past the item)
            const size_t o = (size_t)b * item_elems + in_item;
            float v[8];
            *(float4*)&v[0] = *(const float4*)(stage + row * EPF + ec8);
            *(float4*)&v[4] = *(const float4*)(stage + row * EPF + ec8 + 4);
            if (ksplit > 1) {  // partial sums only, into this slice's own slab; ga_conv_finish adds the slabs and applies the epilogue
                float* wo = p.ws + (size_t)split * ((size_t)p.B * item_elems) + o;
                *(float4*)wo = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(wo + 4) = make_float4(v[4], v[5], v[6], v[7]);
                continue;
            }
            const bool act_after = p.res_mode == GA_RES_ADD_THEN_ACT;  // post_act(acc + bias + res)
            if (act_after) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += bv[i];
            } else if (simple_act) {  // none / LeakyReLU: one multiply-select per element
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] += bv[i];
                    v[i] = v[i] > 0.f ? v[i] : slope * v[i];
                }
            } else if (fast_elu) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = elu_bf16_grade(v[i] + bv[i]);
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
                if (act_after) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fast_elu ? elu_bf16_grade(v[i]) : act_apply(p.post_act, v[i]);
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
                if (p.y2) {  // ELU of the value just stored (what a separate pass over y would read)
                    bf16x8 ev;
#pragma unroll
                    for (int i = 0; i < 8; ++i) ev[i] = (__bf16)elu_bf16_grade((float)ov[i]);
                    *(bf16x8*)((bf16_t*)p.y2 + o) = ev;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// y = epi(ws) for a split-K launch: one thread per (row, 8 output channels)
__global__ __launch_bounds__(256) void conv_finish_kernel(ConvParams p, long total8) {
    const int M8 = p.M / 8;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total8; e += (long)gridDim.x * 256) {
        const int m0 = (int)(e % M8) * 8;
        long row = e / M8;
        const int phase = (int)(row % p.nphase);
        row /= p.nphase;
        const int tl = (int)(row % p.Lout), b = (int)(row / p.Lout);
        const int orow = (p.t0 + tl) * p.out_stride + phase;
        if (orow < 0 || orow >= p.LoutTot) continue;
        const size_t item_elems = p.y_batch_elems > 0 ? (size_t)p.y_batch_elems : (size_t)p.LoutTot * p.M;
        const size_t in_item = (size_t)orow * p.M + m0;
        if (in_item + 8 > item_elems) continue;
        const size_t o = (size_t)b * item_elems + in_item;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const size_t slab = (size_t)p.B * item_elems;
        for (int sp = 0; sp < p.ksplit; ++sp) {
            const float4 a0 = *(const float4*)(p.ws + sp * slab + o), a1 = *(const float4*)(p.ws + sp * slab + o + 4);
            v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w;
            v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
        }
        if (p.bias) {
            const float4 b0 = *(const float4*)(p.bias + m0), b1 = *(const float4*)(p.bias + m0 + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
            v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = act_apply(p.post_act, v[i]);
        if (p.res_mode == GA_RES_FM_SIGN) {
            const float fm_c = p.res_coef * p.res_coef_dev[0];
            const bf16x8 d = *(const bf16x8*)((const bf16_t*)p.dact_src + o);
            const bf16x8 rr = *(const bf16x8*)((const bf16_t*)p.res + o);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float f = (float)d[i], q = (float)rr[i];
                v[i] = (v[i] + (f > q ? fm_c : f < q ? -fm_c : 0.f)) * dact_apply(p.dact_mode, f);
            }
        } else {
            if (p.dact_mode) {
                const bf16x8 d = *(const bf16x8*)((const bf16_t*)p.dact_src + o);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] *= dact_apply(p.dact_mode, (float)d[i]);
            }
            if (p.res) {
                const bf16x8 d = *(const bf16x8*)((const bf16_t*)p.res + o);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += (float)d[i];
            }
        }
        if (p.out_f32) {
            float* yo = (float*)p.y + o;
            *(float4*)yo = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(yo + 4) = make_float4(v[4], v[5], v[6], v[7]);
            continue;
        }
        if (p.accumulate) {
            const bf16x8 d = *(const bf16x8*)((const bf16_t*)p.y + o);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += (float)d[i];
        }
        bf16x8 ov;
#pragma unroll
        for (int i = 0; i < 8; ++i) ov[i] = (__bf16)v[i];
        *(bf16x8*)((bf16_t*)p.y + o) = ov;
    }
}

int ga_conv_gemm_supported(const ConvParams& p);

int ga_conv_gemm_finish(const ConvParams& p, hipStream_t s) {
    if (!ga_conv_gemm_supported(p) || p.ksplit <= 1 || !p.ws || !p.y || ((uintptr_t)p.ws & 15)) return GA_ERR_ARG;
    const long total8 = (long)p.B * p.Lout * p.nphase * (p.M / 8);
    if (total8 <= 0) return GA_OK;
    long grid = (total8 + 255) / 256;
    if (grid > 8192) grid = 8192;
    conv_finish_kernel<<<(unsigned)grid, 256, 0, s>>>(p, total8);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

// the split that brings a few-tile, deep-K problem to about one and a half workgroups per CU (1 = leave it alone)
int ga_conv_gemm_suggest_ksplit(const ConvParams& p) {
    if (!ga_conv_gemm_supported(p) || p.M <= 64 || p.Lout <= 0) return 1;
    const long tiles = (((long)p.B * p.Lout + 127) / 128) * ((p.M + 127) / 128) * p.nphase;
    const int nsteps = (p.J * p.Cin + GBK - 1) / GBK;
    if (tiles >= 200 || nsteps < 24) return 1;
    long ks = (384 + tiles - 1) / tiles;
    if (ks > nsteps / 8) ks = nsteps / 8;
    if (ks > 16) ks = 16;
    return ks < 2 ? 1 : (int)ks;
}

// ------------------------------------------------------------------------------------------------
int ga_conv_gemm_supported(const ConvParams& p) {
    if (p.pre_act != GA_ACT_NONE) return 0;  // direct-to-LDS loads cannot transform the operand in flight
    if (p.Cin % 8 || p.Cin < 8 || p.M % 8 || p.M < 16 || p.tap_step < 0 || p.J < 1 || p.B < 1 || p.Lout < 0) return 0;
    if (p.nphase < 1 || p.nphase > GA_MAX_PHASE) return 0;
    if (p.y_batch_elems < 0 || (p.y_batch_elems > 0 && (p.y_batch_elems % 8 || (long)p.y_batch_elems > (long)p.LoutTot * p.M))) return 0;
    if (p.out_f32 && (p.res || p.dact_src || p.accumulate)) return 0;
    if (((uintptr_t)p.x | (uintptr_t)p.w | (uintptr_t)p.y | (uintptr_t)p.res | (uintptr_t)p.dact_src | (uintptr_t)p.bias) & 15) return 0;
    if ((long)p.J * p.Cin > (1L << 30) || (long)p.B * p.Lout > (1L << 30)) return 0;
    return 1;
}

// GA_GEMM_NBUF=1|2|4 in the environment pins the LDS buffering of the 128x128 configuration (tuning aid); default: by grid size
static int gemm_nbuf_override() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GA_GEMM_NBUF");
        v = (e && (e[0] == '1' || e[0] == '2' || e[0] == '4')) ? e[0] - '0' : 0;
    }
    return v;
}

// grids up to this many workgroups (about one per CU) take the four-buffer pipeline; GA_GEMM_DEEP_MAX overrides (tuning aid)
static int gemm_deep_max_blocks() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GA_GEMM_DEEP_MAX");
        v = e ? atoi(e) : 288;
        if (v < 0) v = 0;
    }
    return v;
}

// grids of at most this many 128x128 workgroups use 64x128 tiles instead; GA_GEMM_HALF_MAX overrides (tuning aid)
static int gemm_half_max_blocks() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GA_GEMM_HALF_MAX");
        v = e ? atoi(e) : 400;  // (re-swept with buffer-load staging: 128 -> 400 is +0.5 % at B = 64 and at B = 12)
        if (v < 0) v = 0;
    }
    return v;
}

// grids of at least this many 128x128 workgroups take the single-buffer kernel (four workgroups per CU); GA_GEMM_NBUF1_MIN
// overrides (tuning aid)
static int gemm_single_min_blocks() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GA_GEMM_NBUF1_MIN");
        v = e ? atoi(e) : 450;  // swept 300 / 450 / 600 / 800 / 1100 at four workgroups per CU: 450 is the best by 1.7 % of GEMM time
        if (v < 1) v = 1;
    }
    return v;
}

// m-tiles per group in the tile order (see the kernel); GA_GEMM_MGROUP overrides (tuning aid)
static int gemm_mgroup() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GA_GEMM_MGROUP");
        v = e ? atoi(e) : 8;
        if (v < 1) v = 1;
        const char* t = getenv("GA_GEMM_TAPFAST");  // "0": keep the plain (tap, channel) K order
        if (t && t[0] == '0') v |= 0x10000;
        const char* ph = getenv("GA_GEMM_PHASEFAST");  // "0": plain tap order on strided layers too
        if (!(ph && ph[0] == '0')) v |= 0x20000;
    }
    return v;
}

// buffer-load staging where it applies (zero padding, 32-bit byte offsets); GA_GEMM_BUF=0 keeps the flat form.
// (Cin a multiple of the K-step: a K-step then lies inside ONE tap for every lane, which is what makes the K position
// wave-uniform; with other channel counts the lanes of a K-step straddle taps and the flat form's per-lane state is needed)
static bool gemm_use_buf(const ConvParams& p) {
    const char* eb = getenv("GA_GEMM_BUF");
    return !(eb && eb[0] == '0') && p.pad_mode == 0 && p.Cin % GBK == 0 &&
           (long)p.B * p.Lin * p.Cin * 2 < (1L << 31) - 64 && (long)p.M * p.J * p.Cin * 2 < (1L << 31) - 64;
}

template <int MI, int WM, int WN, int NBUF>
static int launch_gemm(const ConvParams& p, hipStream_t s) {
    constexpr int BM = WM * MI * 32, BN = WN * 64;
    const int Ntot = p.B * p.Lout;
    const int nN = (Ntot + BN - 1) / BN, nM = (p.M + BM - 1) / BM;
    const int Ktot = p.J * p.Cin;
    const int nsteps = (Ktot + GBK - 1) / GBK;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    if (ksplit > 1 && (!p.ws || ((uintptr_t)p.ws & 15) || ksplit > nsteps)) return GA_ERR_ARG;
    const long total = (long)nN * nM * p.nphase * ksplit;
    if (total <= 0) return GA_OK;
    if (total > (1L << 30)) return GA_ERR_ARG;
    size_t lds = NBUF * (BM + BN) * 128;
    const size_t epi = WM * WN * 32 * (MI * 32 + 4) * sizeof(float);
    if (lds < epi) lds = epi;
    const bool buf = gemm_use_buf(p);
    const int grid = (int)((total + 7) / 8 * 8);
    const int flags = ksplit > 1 ? (1 | (gemm_mgroup() & 0x10000)) : gemm_mgroup();
    // (two explicit launches: taking the address of the BUF = true instantiation into a function pointer fails to compile
    // for the host with this hipcc, a direct launch or a cast to const void* does not)
    if (buf) {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv_gemm_kernel<MI, WM, WN, NBUF, true>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GA_ERR_LAUNCH;
        conv_gemm_kernel<MI, WM, WN, NBUF, true><<<grid, WM * WN * 64, lds, s>>>(p, nN, (int)total, Ktot, nsteps, Ntot, flags);
    } else {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv_gemm_kernel<MI, WM, WN, NBUF, false>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GA_ERR_LAUNCH;
        conv_gemm_kernel<MI, WM, WN, NBUF, false><<<grid, WM * WN * 64, lds, s>>>(p, nN, (int)total, Ktot, nsteps, Ntot, flags);
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_conv_pp_wanted(const ConvParams& p);
int ga_conv_pp_impl(const ConvParams& p, hipStream_t s);

// LDS budget per workgroup for gathered-GEMM launches (0: none), see ga_set_gemm_lds_budget in gan_amd.h: a caller that runs these
// launches BESIDE a persistent LSTM launch (one workgroup of ~111-137 KiB of LDS on every CU) bounds their LDS so that the two
// co-reside on a CU instead of the GEMM waiting for the LSTM to leave.  A process-wide setting, read at launch time.
static int g_gemm_lds_budget = 0;
int ga_set_gemm_lds_budget_impl(int bytes) {
    const int old = g_gemm_lds_budget;
    g_gemm_lds_budget = bytes > 0 ? bytes : 0;
    return old;
}
static int gemm_variant_lds(int v) {
    const int mi = (v >> 12) & 15, wm = (v >> 8) & 15, wn = (v >> 4) & 15, nb = v & 15;
    const int lds = nb * (wm * mi * 32 + wn * 64) * 128, epi = wm * wn * 32 * (mi * 32 + 4) * 4;
    return lds > epi ? lds : epi;
}

static int gemm_variant_unbounded(const ConvParams& p);
// the instantiation conv_gemm_kernel<MI, WM, WN, NBUF> of a problem, as 0xMWNB
static int gemm_variant(const ConvParams& p) {
    const int v = gemm_variant_unbounded(p);
    const int budget = g_gemm_lds_budget;
    if (budget <= 0 || gemm_variant_lds(v) <= budget) return v;
    // the smallest configuration: 64 x 128 tile, one buffer = 24 KiB (only layers with M > 64 have it; others keep their choice)
    if (p.M > 64 && gemm_variant_lds(0x1221) <= budget) return 0x1221;
    return v;
}
static int gemm_variant_unbounded(const ConvParams& p) {
    // thin layers: short K (a few steps) is HBM-bound and wants occupancy (one buffer, 4 workgroups per CU); long K
    // wants the second buffer
    const int ksteps = (p.J * p.Cin + GBK - 1) / GBK;
    const int nb_small = gemm_nbuf_override() ? gemm_nbuf_override() : (ksteps <= 8 ? 1 : 2);
    if (p.M <= 32) return nb_small == 1 ? 0x1141 : 0x1142;
    if (p.M <= 64) return nb_small == 1 ? 0x2141 : 0x2142;
    // 128x128: one LDS buffer (32 KiB, three workgroups per CU hide each other's loads) once the grid is several
    // waves of workgroups deep; two buffers (own prefetch) when the grid is about one workgroup per CU or less
    const long blocks = (((long)p.B * p.Lout + 127) / 128) * ((p.M + 127) / 128) * p.nphase * (p.ksplit > 1 ? p.ksplit : 1);
    // few 128x128 tiles (short sequences: the generator's and the judge's convs): 64x128 tiles double the workgroups
    if (blocks <= gemm_half_max_blocks() && !gemm_nbuf_override()) return (2 * blocks <= gemm_deep_max_blocks() && ksteps >= 6) ? 0x1224 : 0x1222;
    int nbuf = gemm_nbuf_override();
    if (!nbuf) nbuf = blocks >= gemm_single_min_blocks() ? 1 : (blocks <= gemm_deep_max_blocks() && ksteps >= 6) ? 4 : 2;
    return 0x2220 | nbuf;
}
int ga_conv_win_wanted(const ConvParams& p);
int ga_conv_win_impl(const ConvParams& p, hipStream_t s);
int ga_conv_coop_wanted(const ConvParams& p);
int ga_conv_coop_impl(const ConvParams& p, hipStream_t s);

int ga_conv_gemm_impl(const ConvParams& p, hipStream_t s) {
    if (!ga_conv_gemm_supported(p) || !p.x || !p.w || !p.y) return GA_ERR_ARG;
    if (p.res_mode == GA_RES_ADD_THEN_ACT && (!p.res || p.ksplit > 1 || p.dact_mode)) return GA_ERR_ARG;
    if (p.y2 && (p.ksplit > 1 || p.out_f32 || ((uintptr_t)p.y2 & 15))) return GA_ERR_ARG;
    if (p.Lout == 0) return GA_OK;
    const bool bounded = g_gemm_lds_budget > 0;  // (the special-shape kernels below all need more LDS than any budget worth setting)
    if (!bounded && ga_conv_coop_wanted(p)) return ga_conv_coop_impl(p, s);  // many rows, K of a few hundred, 256 channels: weights in registers (conv_coop.hip)
    if (!bounded && ga_conv_win_wanted(p)) return ga_conv_win_impl(p, s);  // narrow-channel D layers: input window staged once in LDS (conv_win.hip)
    if (!bounded && ga_conv_pp_wanted(p)) return ga_conv_pp_impl(p, s);  // deep-K, >= 192-channel layers: the ping-pong kernel (conv_pp.hip)
    switch (gemm_variant(p)) {
        case 0x1141: return launch_gemm<1, 1, 4, 1>(p, s);
        case 0x1142: return launch_gemm<1, 1, 4, 2>(p, s);
        case 0x2141: return launch_gemm<2, 1, 4, 1>(p, s);
        case 0x2142: return launch_gemm<2, 1, 4, 2>(p, s);
        case 0x1224: return launch_gemm<1, 2, 2, 4>(p, s);
        case 0x1222: return launch_gemm<1, 2, 2, 2>(p, s);
        case 0x1221: return launch_gemm<1, 2, 2, 1>(p, s);
        case 0x2221: return launch_gemm<2, 2, 2, 1>(p, s);
        case 0x2224: return launch_gemm<2, 2, 2, 4>(p, s);
        default: return launch_gemm<2, 2, 2, 2>(p, s);
    }
}

// which conv_gemm_kernel<MI, WM, WN, NBUF, BUF> a problem runs on, as 0xMWNB | BUF << 16 (0: another kernel takes it).  For the
// benchmark's per-kernel accounting: the instantiations are different kernels to a profiler (ga_conv_gemm_variant in gan_amd.h).
int ga_conv_gemm_variant_impl(const ConvParams& p) {
    if (!ga_conv_gemm_supported(p)) return 0;
    if (g_gemm_lds_budget <= 0 && (ga_conv_coop_wanted(p) || ga_conv_win_wanted(p) || ga_conv_pp_wanted(p))) return 0;
    return gemm_variant(p) | (gemm_use_buf(p) ? 0x10000 : 0);
}

// ------------------------------------------------------------------------------------------------
// Split-K through a library-owned scratch: the one entry point that allocates nothing (include/gan_amd.h,
// ga_conv_splitk_static).  The partial-sum slabs of conv_gemm_kernel + conv_finish_kernel live in a file-scope device
// array instead of a caller workspace.  Each GPU the library runs on holds its own copy of the array (32 MiB) at its own
// address until the process ends; the array's NAME is only a host-side handle, so the device address is looked up once per
// device with hipGetSymbolAddress and that pointer, never the name, is what the kernels receive.
#include <atomic>
#include <mutex>

#define GA_SPLITK_STATIC_MAX_DEVICES 64
static __device__ __align__(256) float ga_splitk_scratch[GA_SPLITK_STATIC_FLOATS];
static std::atomic<float*> splitk_scratch_ptr[GA_SPLITK_STATIC_MAX_DEVICES];
static std::mutex splitk_scratch_mutex;

// the calling thread's current device's copy of the scratch array (nullptr: lookup failed)
static float* splitk_scratch_for_current_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GA_SPLITK_STATIC_MAX_DEVICES) return nullptr;
    float* ptr = splitk_scratch_ptr[dev].load(std::memory_order_acquire);
    if (ptr) return ptr;
    std::lock_guard<std::mutex> lock(splitk_scratch_mutex);
    ptr = splitk_scratch_ptr[dev].load(std::memory_order_relaxed);
    if (ptr) return ptr;
    void* sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(ga_splitk_scratch)) != hipSuccess || !sym) return nullptr;
    ptr = (float*)sym;
    splitk_scratch_ptr[dev].store(ptr, std::memory_order_release);
    return ptr;
}

int ga_conv_splitk_static_impl(const ConvParams& p0, hipStream_t s) {
    if (!ga_conv_gemm_supported(p0) || !p0.x || !p0.w || !p0.y) return GA_ERR_ARG;
    if (p0.ws) return GA_ERR_ARG;  // the library supplies the workspace here; a caller-owned one goes through ga_conv
    if (p0.Lout == 0) return GA_OK;
    ConvParams p = p0;
    const int nsteps = (p.J * p.Cin + GBK - 1) / GBK;
    if (p.ksplit <= 1) {  // library's choice, at least two slices (this entry point IS the split-K path)
        p.ksplit = ga_conv_gemm_suggest_ksplit(p);
        if (p.ksplit < 2) p.ksplit = 2;
    }
    if (p.ksplit > nsteps) p.ksplit = nsteps;
    if (p.ksplit < 2) return GA_ERR_UNSUPPORTED;  // a single 64-deep K step cannot be split
    const size_t item_elems = p.y_batch_elems > 0 ? (size_t)p.y_batch_elems : (size_t)p.LoutTot * p.M;
    const size_t need = (size_t)p.ksplit * (size_t)p.B * item_elems;
    if (need > (size_t)GA_SPLITK_STATIC_FLOATS) return GA_ERR_SCRATCH;
    float* const scratch = splitk_scratch_for_current_device();
    if (!scratch) return GA_ERR_LAUNCH;
    p.ws = scratch;
    // Nothing needs clearing: every slab word conv_finish_kernel reads was written by this call's conv_gemm_kernel (both
    // walk the same rows / channels of the problem), so what an earlier call left in the array is never observed.
    const int rc = ga_conv_gemm_impl(p, s);
    if (rc != GA_OK) return rc;
    return ga_conv_gemm_finish(p, s);
}
