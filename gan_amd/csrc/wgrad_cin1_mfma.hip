// bf16 MFMA weight gradient of a single-input-channel convolution (the discriminators' first conv on the waveform):
//   dw[m][j] += sum_{b,t} g[b][t][m] * x[b][t + j + off0]      db[m] += sum_{b,t} g[b][t][m]
// As a GEMM this is D[m][j] = sum_n G^T[m][n] X[n][j] with X[n][j] = x[n + j + off0] (a Toeplitz matrix that is never
// formed): v_mfma_f32_16x16x32_bf16 with the step axis n as K.  The B fragment of lane (tap j, quarter kq) -- 8
// consecutive n of column j -- is 8 consecutive samples of x, ONE unaligned 16-byte load; a constant-ones column at
// j = 15 yields the bias gradient from the same MFMAs.  The A fragment (channel m, 8 consecutive n) is gathered with
// 2-byte loads (channels are the contiguous axis of g; the lanes of a load instruction cover whole 32-byte runs).
// The kernel is bound by the one pass over g (HBM); the direct kernel it replaces was bound by vector FMAs (~5x slower).
// Padding cannot be applied per 16-byte piece, so the caller hands in a zero-padded copy of x in which every window the
// kernel touches is in range (checked on the host; otherwise the direct kernel runs).
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef bf16x8 __attribute__((aligned(2))) bf16x8_u;
typedef const __attribute__((address_space(1))) bf16x8_u* gfrag_u_ptr_t;
typedef const __attribute__((address_space(1))) unsigned short* gu16_ptr_t;

#define WGC_STEP 32   // steps (rows of g) per MFMA
#define WGC_UNROLL 2  // MFMA steps whose loads are issued together

template <int MT>
__global__ __launch_bounds__(256) void wgrad_cin1_mfma_kernel(WgradParams p, int steps_per_item, int total, int bias_col) {
    __shared__ float red[4][MT][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int rc = lane & 15, kq = lane >> 4;
    const uintptr_t zero = (uintptr_t)ga_zero_page;
    const bf16_t* __restrict__ g = (const bf16_t*)p.g;
    const bf16_t* __restrict__ x = (const bf16_t*)p.x;
    const int jcol = rc < p.J ? rc : p.J - 1;  // columns past J are computed on a valid window and dropped
    const bool ones = rc == bias_col;

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 one8;
#pragma unroll
    for (int i = 0; i < 8; ++i) one8[i] = (__bf16)1.0f;

    const int nwaves = gridDim.x * 4;
    for (int w0 = blockIdx.x * 4 + wid; w0 < total; w0 += nwaves * WGC_UNROLL) {
        unsigned short ga[WGC_UNROLL][MT][8];
        bf16x8 xb[WGC_UNROLL];
#pragma unroll
        for (int u = 0; u < WGC_UNROLL; ++u) {
            const int wk = w0 + u * nwaves;
            const bool on = wk < total;
            const int wkc = on ? wk : total - 1;
            const int b = wkc / steps_per_item;
            const int t0 = (wkc - b * steps_per_item) * WGC_STEP + kq * 8;
            xb[u] = *(gfrag_u_ptr_t)(x + (size_t)b * p.Lin + t0 + jcol + p.off0);  // in range by the host-side check
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t = t0 + i;
                const bool ok = on && t < p.Lout;
                const bf16_t* row = g + ((size_t)b * p.LgTot + (size_t)t * p.g_stride + p.g_phase) * p.M + rc;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const uintptr_t a = (uintptr_t)(row + mt * 16);
                    ga[u][mt][i] = *(gu16_ptr_t)(zero ^ ((zero ^ a) & (uintptr_t)(-(long long)ok)));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < WGC_UNROLL; ++u) {
            const bf16x8 bfrag = ones ? one8 : xb[u];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                union { bf16x8 v; unsigned short s[8]; } af;
#pragma unroll
                for (int i = 0; i < 8; ++i) af.s[i] = ga[u][mt][i];
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bfrag, acc[mt], 0, 0, 0);
            }
        }
    }
    // D: row (channel) = 4*kq + i, column (tap) = rc.  Sum the four waves, then one atomic per output per workgroup.
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wid][mt][kq * 4 + i][rc] = acc[mt][i];
    __syncthreads();
    for (int o = tid; o < MT * 256; o += 256) {
        const int mt = o >> 8, row = (o >> 4) & 15, col = o & 15;
        const int m = mt * 16 + row;
        if (m >= p.M) continue;
        const float v = red[0][mt][row][col] + red[1][mt][row][col] + red[2][mt][row][col] + red[3][mt][row][col];
        if (col < p.J && col != bias_col) atomicAdd(&p.dw[(size_t)m * p.J + col], v);
        else if (col == bias_col && p.dbias) atomicAdd(&p.dbias[m], v);
    }
}

int ga_wgrad_cin1_mfma_supported(const WgradParams& p) {
    if (p.Cin != 1 || p.in_stride != 1 || p.tap_step != 1 || p.M % 16 || p.M < 16 || p.M > 64 || p.J < 1) return 0;
    if (p.J > (p.dbias ? 15 : 16) || p.off0 < 0 || p.Lout < 1 || p.B < 1 || p.g_stride < 1) return 0;
    if (((uintptr_t)p.g | (uintptr_t)p.x) & 1) return 0;
    // furthest sample any lane reads: step rounded up to 32, + 15 taps (columns are clamped to J - 1), + 8-sample piece
    const long last = (long)(p.Lout + WGC_STEP - 1) / WGC_STEP * WGC_STEP - 8 + (p.J - 1) + p.off0 + 8;
    if (last > p.Lin) return 0;
    if ((long)p.B * ((p.Lout + WGC_STEP - 1) / WGC_STEP) > (1L << 30)) return 0;
    return 1;
}

int ga_wgrad_cin1_mfma_impl(const WgradParams& p, hipStream_t s) {
    if (!ga_wgrad_cin1_mfma_supported(p) || !p.g || !p.x || !p.dw) return GA_ERR_ARG;
    const int spi = (p.Lout + WGC_STEP - 1) / WGC_STEP;
    const int total = p.B * spi;
    int grid = (total + 4 * WGC_UNROLL * 4 - 1) / (4 * WGC_UNROLL * 4);  // at least four trips per wave
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    const int bias_col = p.dbias ? 15 : -1;
    switch (p.M / 16) {
        case 1: wgrad_cin1_mfma_kernel<1><<<grid, 256, 0, s>>>(p, spi, total, bias_col); break;
        case 2: wgrad_cin1_mfma_kernel<2><<<grid, 256, 0, s>>>(p, spi, total, bias_col); break;
        case 3: wgrad_cin1_mfma_kernel<3><<<grid, 256, 0, s>>>(p, spi, total, bias_col); break;
        default: wgrad_cin1_mfma_kernel<4><<<grid, 256, 0, s>>>(p, spi, total, bias_col); break;
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}
