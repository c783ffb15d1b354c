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
// branch-free "ok ? a : z" on addresses (a ternary on pointers makes hipcc branch around two separate loads)
__device__ __forceinline__ uintptr_t sel_addr(bool ok, uintptr_t a, uintptr_t z) {
    return z ^ ((z ^ a) & (uintptr_t)(-(long long)ok));
}

template <int WM, int WN, int NBUF>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvParams p, int nN, int total, int Ktot, int nsteps, int Ntot) {
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int NA = BM * 8 / 256;  // 16-byte pieces per thread per K-step, A (weights)
    constexpr int NB = BN * 8 / 256;  // same, B (gathered activations)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    // [2][BM*128] A, then [2][BN*128] B
    char* const Al = smem;
    char* const Bl = smem + NBUF * BM * 128;

    const int id = blockIdx.x;
    const int chunk = (total + 7) / 8;
    const int wk = (id & 7) * chunk + (id >> 3);
    if (wk >= total) return;
    const int my = wk / (nN * p.nphase);
    const int rem = wk - my * (nN * p.nphase);
    const int phase = rem / nN;
    const int bx = rem - phase * nN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;
    const int mbase = my * BM, nbase = bx * BN;
    const int S = p.in_stride, Cin = p.Cin;
    const int off0 = p.off0 + p.phase_off0[phase];
    const bf16_t* __restrict__ w = (const bf16_t*)p.w + (size_t)phase * p.M * Ktot;
    const uintptr_t zero = (uintptr_t)ga_zero_page;

    // staging geometry: piece q = i*256 + tid -> row q>>3 = i*32 + (tid>>3), slot tid&7, logical piece below
    const int srow = tid >> 3;
    const int piece = (tid & 7) ^ ((tid >> 4) & 7);
    const bf16_t* arow[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = mbase + i * 32 + srow;
        arow[i] = m < p.M ? w + (size_t)m * Ktot : nullptr;
    }
    const bf16_t* brow[NB];
    int bR[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = nbase + i * 32 + srow;
        if (n < Ntot) {
            const int b = n / p.Lout, t = p.t0 + (n - b * p.Lout);
            brow[i] = (const bf16_t*)p.x + (size_t)b * p.Lin * Cin;
            bR[i] = t * S + off0;
        } else {
            brow[i] = nullptr;
            bR[i] = 0;
        }
    }

    auto issue = [&](int ks, int buf) {
        const int k = ks * GBK + piece * 8;
        const bool kin = k < Ktot;
        const int j = k / Cin, c = k - j * Cin;
        const int dR = j * p.tap_step;
        char* const ab = Al + buf * (BM * 128) + wid * 1024;
        char* const bb = Bl + buf * (BN * 128) + wid * 1024;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            glds16(sel_addr(kin && arow[i], (uintptr_t)(arow[i] + k), zero), ab + i * 4096);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            int R = bR[i] + dR;
            if (R < 0 && p.pad_mode == 1) R = -R;
            const bool ok = kin && brow[i] && R >= 0 && R < p.Lin;
            glds16(sel_addr(ok, (uintptr_t)(brow[i] + (size_t)R * Cin + c), zero), bb + i * 4096);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

    const int swz = (r >> 1) & 7;
    const int aoff = (wm * 64 + r) * 128, boff = (wn * 64 + r) * 128;

    if (NBUF == 2) issue(0, 0);
    for (int ks = 0; ks < nsteps; ++ks) {
        const int buf = NBUF == 2 ? (ks & 1) : 0;
        if (NBUF == 1) {
            if (ks > 0) __syncthreads();  // everyone is done reading the single buffer
            issue(ks, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // K-step ks has landed for every wave; everyone is done reading the other buffer
        if (NBUF == 2 && ks + 1 < nsteps) issue(ks + 1, buf ^ 1);
        const char* Ab = Al + buf * (BM * 128) + aoff;
        const char* Bb = Bl + buf * (BN * 128) + boff;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int po = ((kk * 2 + h) ^ swz) << 4;
            bf16x8 a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[mi] = *(const bf16x8*)(Ab + mi * 32 * 128 + po);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = *(const bf16x8*)(Bb + ni * 32 * 128 + po);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }

    if (mbase + wm * 64 >= p.M) return;
    bf16_t* __restrict__ y = (bf16_t*)p.y;
    const bf16_t* __restrict__ res = (const bf16_t*)p.res;
    const bf16_t* __restrict__ ds = (const bf16_t*)p.dact_src;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = nbase + wn * 64 + ni * 32 + r;
        if (n >= Ntot) continue;
        const int b = n / p.Lout, t = p.t0 + (n - b * p.Lout);
        const int orow = t * p.out_stride + phase;
        if (orow < 0 || orow >= p.LoutTot) continue;
        const size_t rowbase = ((size_t)b * p.LoutTot + orow) * p.M;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m0 = mbase + wm * 64 + mi * 32 + 8 * q + 4 * h;
                if (m0 >= p.M) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[mi][ni][4 * q + i];
                if (p.bias) {
                    const float4 bv = *(const float4*)(p.bias + m0);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = act_apply(p.post_act, v[i]);
                if (p.dact_mode) {
                    const bf16x4 d = *(const bf16x4*)(ds + rowbase + m0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] *= dact_apply(p.dact_mode, (float)d[i]);
                }
                if (res) {
                    const bf16x4 d = *(const bf16x4*)(res + rowbase + m0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)d[i];
                }
                if (p.accumulate) {
                    const bf16x4 d = *(const bf16x4*)(y + rowbase + m0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += (float)d[i];
                }
                if (p.out_f32) {
                    *(float4*)((float*)p.y + rowbase + m0) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
                    *(bf16x4*)(y + rowbase + m0) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
int ga_conv_gemm_supported(const ConvParams& p) {
    if (p.pre_act != GA_ACT_NONE) return 0;  // direct-to-LDS loads cannot transform the operand in flight
    if (p.Cin % 8 || p.Cin < 8 || p.M % 4 || p.M < 33 || p.tap_step < 0 || p.J < 1 || p.B < 1 || p.Lout < 0) return 0;
    if (p.nphase < 1 || p.nphase > GA_MAX_PHASE) return 0;
    if (p.out_f32 && (p.res || p.dact_src || p.accumulate)) return 0;
    if (((uintptr_t)p.x | (uintptr_t)p.w | (uintptr_t)p.y | (uintptr_t)p.res | (uintptr_t)p.dact_src | (uintptr_t)p.bias) & 15) return 0;
    if ((long)p.J * p.Cin > (1L << 30) || (long)p.B * p.Lout > (1L << 30)) return 0;
    return 1;
}

// GA_GEMM_NBUF=1|2 in the environment pins the LDS buffering of the 128x128 configuration (tuning aid); default: by grid size
static int gemm_nbuf_override() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GA_GEMM_NBUF");
        v = (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 0;
    }
    return v;
}

template <int WM, int WN, int NBUF>
static int launch_gemm(const ConvParams& p, hipStream_t s) {
    constexpr int BM = WM * 64, BN = WN * 64;
    const int Ntot = p.B * p.Lout;
    const int nN = (Ntot + BN - 1) / BN, nM = (p.M + BM - 1) / BM;
    const long total = (long)nN * nM * p.nphase;
    if (total <= 0) return GA_OK;
    if (total > (1L << 30)) return GA_ERR_ARG;
    const int Ktot = p.J * p.Cin;
    const int nsteps = (Ktot + GBK - 1) / GBK;
    const size_t lds = NBUF * (BM + BN) * 128;
    auto kern = conv_gemm_kernel<WM, WN, NBUF>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GA_ERR_LAUNCH;
    }
    const int grid = (int)((total + 7) / 8 * 8);
    kern<<<grid, 256, lds, s>>>(p, nN, (int)total, Ktot, nsteps, Ntot);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_conv_gemm_impl(const ConvParams& p, hipStream_t s) {
    if (!ga_conv_gemm_supported(p) || !p.x || !p.w || !p.y) return GA_ERR_ARG;
    if (p.Lout == 0) return GA_OK;
    if (p.M <= 64) return launch_gemm<1, 4, 2>(p, s);
    // 128x128: one LDS buffer (32 KiB, three workgroups per CU hide each other's loads) once the grid is several
    // waves of workgroups deep; two buffers (own prefetch) when the grid is about one workgroup per CU or less
    const long blocks = (((long)p.B * p.Lout + 127) / 128) * ((p.M + 127) / 128) * p.nphase;
    int nbuf = gemm_nbuf_override();
    if (!nbuf) nbuf = blocks >= 600 ? 1 : 2;
    return nbuf == 1 ? launch_gemm<2, 2, 1>(p, s) : launch_gemm<2, 2, 2>(p, s);
}
