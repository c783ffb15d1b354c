// Two-layer LSTM of the frozen EnCodec decoder (HF EncodecLSTM, modeling_encodec.py:236-249: nn.LSTM(512, 512, 2)
// over time, plus the input skip) as ONE chain of small step kernels per decode instead of a library RNN.
//
// The recurrence is latency-bound (375 dependent steps, 12.6 MFLOP each at B=12), so the design goal is the fewest,
// shortest dependent launches: one launch per time step covers BOTH layers (layer 2 runs one step behind layer 1
// and consumes [h1_t | h2_{t-1}] with the concatenated [W_ih_l1 | W_hh_l1]), the layer-1 input projection for all
// steps is hoisted into one GEMM (ga_conv, k=1), kernel boundaries are the only inter-workgroup synchronisation
// (no spin waits), and every workgroup owns 4 hidden units (forward) / 16 hidden units (backward) whose gate rows
// form exactly one 16-row MFMA tile.  Operands stream from L2 straight into MFMA fragments (no LDS staging: nothing is
// reused inside a step); the 4 waves of a workgroup split K and combine through 4 KiB of LDS.
//   bf16 weights/state: v_mfma_f32_16x16x32_bf16;  fp32: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains)
// Backward produces only the input gradient (weights are frozen): d(gates) of layer 1 for all steps, which one more
// k=1 GEMM turns into dx.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }

template <typename WT> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int KSTEP = 32;
    typedef bf16x8 V;
    static __device__ __forceinline__ V load(const bf16_t* row, int k0, int kq) { return *(const bf16x8*)(row + k0 + 8 * kq); }
    static __device__ __forceinline__ f32x4 mma(V a, V b, f32x4 acc) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); }
    // a_row / b_row: pointers to element k=0 of this lane's row (A) / column (B); lane group kq = lane >> 4
    static __device__ __forceinline__ f32x4 run(const bf16_t* a_row, const bf16_t* b_row, int k0, int kq, f32x4 acc) {
        const bf16x8 a = *(const bf16x8*)(a_row + k0 + 8 * kq);
        const bf16x8 b = *(const bf16x8*)(b_row + k0 + 8 * kq);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int KSTEP = 16;
    typedef float4 V;
    static __device__ __forceinline__ V load(const float* row, int k0, int kq) { return *(const float4*)(row + k0 + 4 * kq); }
    static __device__ __forceinline__ f32x4 mma(V a, V b, f32x4 acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 run(const float* a_row, const float* b_row, int k0, int kq, f32x4 acc) {
        const float4 a = *(const float4*)(a_row + k0 + 4 * kq);
        const float4 b = *(const float4*)(b_row + k0 + 4 * kq);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
        return acc;
    }
};

// accumulate acc += A[16 rows][k0..k1) * B[k0..k1)[16 cols], K range split over the 4 waves of the block
template <typename WT>
__device__ __forceinline__ f32x4 k_range(const WT* a_row, const WT* b_row, int K, int wid, int kq, f32x4 acc) {
    constexpr int KS = Mma<WT>::KSTEP;
    const int steps = K / KS, per = (steps + 3) / 4;
    const int s0 = wid * per;
    int s1 = s0 + per;
    if (s1 > steps) s1 = steps;
#pragma unroll 4
    for (int s = s0; s < s1; ++s) acc = Mma<WT>::run(a_row, b_row, s * KS, kq, acc);
    return acc;
}

// ------------------------------------------------------------------------------------------------
template <typename WT, typename AT>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(ga_lstm_params p, int s) {
    __shared__ float red[4][16][17];
    const int layer = blockIdx.y;
    const int t = layer == 0 ? s : s - 1;
    if (t < 0 || t >= p.T) return;
    const int H = p.H;
    const int u0 = blockIdx.x * 4;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int rc = lane & 15, kq = lane >> 4;
    // A row of this lane: gate = rc / 4, unit = u0 + rc % 4
    const int arow = (rc >> 2) * H + u0 + (rc & 3);
    const WT* __restrict__ h1 = (const WT*)p.h1;
    const WT* __restrict__ h2 = (const WT*)p.h2;
    float* cst = layer == 0 ? p.c1 : p.c2;
    float* gsave = layer == 0 ? p.gates1 : p.gates2;
    float* csave = layer == 0 ? p.cs1 : p.cs2;
    WT* hout = layer == 0 ? (WT*)p.h1 : (WT*)p.h2;

    for (int b0 = 0; b0 < p.B; b0 += 16) {
        const int bcol = b0 + rc;                 // this lane's batch column (B operand / D column)
        const int bld = bcol < p.B ? bcol : p.B - 1;  // clamp loads, results of padded columns are discarded
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (layer == 0) {
            if (t > 0) acc = k_range<WT>((const WT*)p.w1 + (size_t)arow * H, h1 + ((size_t)bld * p.T + (t - 1)) * H, H, wid, kq, acc);
        } else {
            const WT* a = (const WT*)p.w2 + (size_t)arow * 2 * H;
            acc = k_range<WT>(a, h1 + ((size_t)bld * p.T + t) * H, H, wid, kq, acc);
            if (t > 0) acc = k_range<WT>(a + H, h2 + ((size_t)bld * p.T + (t - 1)) * H, H, wid, kq, acc);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wid][kq * 4 + i][rc] = acc[i];  // D: row = 4*kq + i, col = rc
        __syncthreads();
        if (tid < 64) {
            const int uu = tid >> 4, bb = tid & 15;
            const int b = b0 + bb;
            if (b < p.B) {
                const int u = u0 + uu;
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int r = g * 4 + uu;
                    float v = red[0][r][bb] + red[1][r][bb] + red[2][r][bb] + red[3][r][bb];
                    if (layer == 0) v += (float)((const AT*)p.xg1)[((size_t)b * p.T + t) * 4 * H + g * H + u];
                    else v += p.bias2[g * H + u];
                    pre[g] = v;
                }
                const float ig = sigm(pre[0]), fg = sigm(pre[1]), gg = tanhf(pre[2]), og = sigm(pre[3]);
                const size_t ci = (size_t)b * H + u;
                const float c = fg * cst[ci] + ig * gg;
                cst[ci] = c;
                const float h = og * tanhf(c);
                const size_t hi = ((size_t)b * p.T + t) * H + u;
                hout[hi] = (WT)h;
                if (gsave && (p.B_save <= 0 || b < p.B_save)) {
                    *(float4*)(gsave + (((size_t)b * p.T + t) * H + u) * 4) = make_float4(ig, fg, gg, og);
                    csave[hi] = c;
                }
                if (layer == 1) ((AT*)p.y)[hi] = (AT)(h + (float)((const AT*)p.x_in)[hi]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward step: layer 2 at t = T-1-s, layer 1 at t = T-s (one launch behind).  Workgroup = 16 hidden units.
template <typename WT, typename AT>
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(ga_lstm_bwd_params p, int s) {
    __shared__ float red[4][16][17];
    const int layer = blockIdx.y;  // 0: layer 2 (leads), 1: layer 1
    const int t = layer == 0 ? p.T - 1 - s : p.T - s;
    if (t < 0 || t >= p.T) return;
    const int H = p.H;
    const int u0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int rc = lane & 15, kq = lane >> 4;
    const WT* __restrict__ dg1 = (const WT*)p.dg1;
    const WT* __restrict__ dg2 = (const WT*)p.dg2;
    const float* gates = layer == 0 ? p.gates2 : p.gates1;
    const float* cs = layer == 0 ? p.cs2 : p.cs1;
    float* dcs = layer == 0 ? p.dc2 : p.dc1;
    WT* dgout = layer == 0 ? (WT*)p.dg2 : (WT*)p.dg1;

    for (int b0 = 0; b0 < p.B; b0 += 16) {
        const int bcol = b0 + rc;
        const int bld = bcol < p.B ? bcol : p.B - 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (layer == 0) {
            // dh2_t[u] = dy_t[u] + sum_k dg2_{t+1}[k] * W_hh_l1[k][u]       (w2t row H + u)
            if (t + 1 < p.T)
                acc = k_range<WT>((const WT*)p.w2t + (size_t)(H + u0 + rc) * 4 * H, dg2 + ((size_t)bld * p.T + (t + 1)) * 4 * H, 4 * H, wid, kq, acc);
        } else {
            // dh1_t[u] = sum_k dg2_t[k] * W_ih_l1[k][u] + sum_k dg1_{t+1}[k] * W_hh_l0[k][u]
            acc = k_range<WT>((const WT*)p.w2t + (size_t)(u0 + rc) * 4 * H, dg2 + ((size_t)bld * p.T + t) * 4 * H, 4 * H, wid, kq, acc);
            if (t + 1 < p.T)
                acc = k_range<WT>((const WT*)p.w1t + (size_t)(u0 + rc) * 4 * H, dg1 + ((size_t)bld * p.T + (t + 1)) * 4 * H, 4 * H, wid, kq, acc);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wid][kq * 4 + i][rc] = acc[i];  // row = unit, col = batch
        __syncthreads();
        {
            const int uu = tid >> 4, bb = tid & 15;  // 256 threads = 16 units x 16 batch columns
            const int b = b0 + bb;
            if (b < p.B) {
                const int u = u0 + uu;
                float dh = red[0][uu][bb] + red[1][uu][bb] + red[2][uu][bb] + red[3][uu][bb];
                const size_t hi = ((size_t)b * p.T + t) * H + u;
                if (layer == 0) dh += (float)((const AT*)p.dy)[hi];
                const float4 gs = *(const float4*)(gates + (((size_t)b * p.T + t) * H + u) * 4);
                const float ig = gs.x, fg = gs.y, gg = gs.z, og = gs.w;
                const float c = cs[hi];
                const float cprev = t > 0 ? cs[hi - H] : 0.f;
                const float tc = tanhf(c);
                const size_t ci = (size_t)b * H + u;
                const float dc = dcs[ci] + dh * og * (1.f - tc * tc);
                dcs[ci] = dc * fg;
                WT* o = dgout + ((size_t)b * p.T + t) * 4 * H + u;
                o[0] = (WT)(dc * gg * ig * (1.f - ig));
                o[H] = (WT)(dc * cprev * fg * (1.f - fg));
                o[2 * H] = (WT)(dc * ig * (1.f - gg * gg));
                o[3 * H] = (WT)(dh * tc * og * (1.f - og));
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Latency-shaped step kernels (compile-time H, up to NBT batch tiles of 16 per pass).  A step is one dependent
// L2 round trip, so everything a lane needs is requested BEFORE anything is consumed: the epilogue's gate inputs
// and cell state first, then every A/B fragment of the lane's K range (CH k-steps per burst, sized to the VGPR
// budget), and only then the MFMAs.  All batch tiles share one pass over the weights.
template <typename WT, int NK, int NBT>
__device__ __forceinline__ void seg_mma(const WT* a_row, const WT* const (&b_row)[NBT], int kq, f32x4 (&acc)[NBT]) {
    typedef typename Mma<WT>::V V;
    constexpr int KS = Mma<WT>::KSTEP;
    constexpr int CAP = (sizeof(WT) == 2 ? 32 : 24) / (1 + NBT);
    constexpr int CH = NK < CAP ? NK : CAP;
    for (int s0 = 0; s0 < NK; s0 += CH) {
        V a[CH], b[NBT][CH];
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            const int sk = (s0 + s < NK ? s0 + s : NK - 1) * KS;
            a[s] = Mma<WT>::load(a_row, sk, kq);
#pragma unroll
            for (int i = 0; i < NBT; ++i) b[i][s] = Mma<WT>::load(b_row[i], sk, kq);
        }
        __builtin_amdgcn_sched_barrier(0);  // every load of the burst is issued before the first MFMA waits on one
#pragma unroll
        for (int s = 0; s < CH; ++s)
            if (s0 + s < NK) {
#pragma unroll
                for (int i = 0; i < NBT; ++i) acc[i] = Mma<WT>::mma(a[s], b[i][s], acc[i]);
            }
    }
}

// two K segments in ONE burst (backward, layer 1): all loads of both, then all MFMAs
template <typename WT, int NK, int NBT>
__device__ __forceinline__ void seg_mma2(const WT* a0, const WT* const (&b0)[NBT], const WT* a1, const WT* const (&b1)[NBT], int kq,
                                         f32x4 (&acc)[NBT]) {
    typedef typename Mma<WT>::V V;
    constexpr int KS = Mma<WT>::KSTEP;
    V a[2][NK], b[2][NBT][NK];
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        a[0][s] = Mma<WT>::load(a0, s * KS, kq);
        a[1][s] = Mma<WT>::load(a1, s * KS, kq);
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
            b[0][i][s] = Mma<WT>::load(b0[i], s * KS, kq);
            b[1][i][s] = Mma<WT>::load(b1[i], s * KS, kq);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NK; ++s)
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
            acc[i] = Mma<WT>::mma(a[0][s], b[0][i][s], acc[i]);
            acc[i] = Mma<WT>::mma(a[1][s], b[1][i][s], acc[i]);
        }
}

template <typename WT, typename AT, int H, int NBT, int NW>
__global__ __launch_bounds__(NW * 64) void lstm_fwd_fast_kernel(ga_lstm_params p, int s, int b_base0) {
    __shared__ float red[NW][NBT][16][17];
    const int b_base = b_base0 + blockIdx.z * (NBT * 16);  // batch chunks of one step run side by side (grid z), not as a chain of launches
    constexpr int KS = Mma<WT>::KSTEP;
    const int layer = blockIdx.y;
    const int t = layer == 0 ? s : s - 1;
    if (t < 0 || t >= p.T) return;
    const int u0 = blockIdx.x * 4;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int rc = lane & 15, kq = lane >> 4;
    const int arow = (rc >> 2) * H + u0 + (rc & 3);
    const WT* __restrict__ h1 = (const WT*)p.h1;
    const WT* __restrict__ h2 = (const WT*)p.h2;
    float* cst = layer == 0 ? p.c1 : p.c2;
    float* gsave = layer == 0 ? p.gates1 : p.gates2;
    float* csave = layer == 0 ? p.cs1 : p.cs2;
    WT* hout = layer == 0 ? (WT*)p.h1 : (WT*)p.h2;

    // epilogue role (tile, unit, batch column) and its inputs, requested first
    const int e_tile = tid >> 6, e_uu = tid & 3, e_bb = (tid >> 2) & 15;  // unit fastest: 4 lanes share a row segment
    const int e_b = b_base + e_tile * 16 + e_bb, e_u = u0 + e_uu;
    const bool e_on = e_tile < NBT && e_b < p.B;
    // (unconditional raw loads from clamped addresses: a branch, a conversion or a per-element select here would make
    // the compiler wait for each load in turn)
    const int e_bl = e_b < p.B ? e_b : p.B - 1;
    const AT* xgp = (const AT*)p.xg1 + ((size_t)e_bl * p.T + t) * 4 * H + e_u;
    const float* bsp = p.bias2 + e_u;
    const AT xg0 = xgp[0], xg1v = xgp[H], xg2 = xgp[2 * H], xg3 = xgp[3 * H];
    const float bs0 = bsp[0], bs1 = bsp[H], bs2 = bsp[2 * H], bs3 = bsp[3 * H];
    const AT xs_raw = ((const AT*)p.x_in)[((size_t)e_bl * p.T + t) * H + e_u];
    const float c_prev = cst[(size_t)e_bl * H + e_u];

    f32x4 acc[NBT];
#pragma unroll
    for (int i = 0; i < NBT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int bld[NBT];
#pragma unroll
    for (int i = 0; i < NBT; ++i) {
        const int bcol = b_base + i * 16 + rc;
        bld[i] = bcol < p.B ? bcol : p.B - 1;  // clamp loads; results of padded columns are discarded
    }
    if (layer == 0) {
        if (t > 0) {
            constexpr int NK = H / KS / NW;
            const WT* a = (const WT*)p.w1 + (size_t)arow * H + wid * (H / NW);
            const WT* b[NBT];
#pragma unroll
            for (int i = 0; i < NBT; ++i) b[i] = h1 + ((size_t)bld[i] * p.T + (t - 1)) * H + wid * (H / NW);
            seg_mma<WT, NK, NBT>(a, b, kq, acc);
        }
    } else {
        // combined K = [h1_t | h2_{t-1}]: the first half of the waves takes the h1 half, the rest the h2 half (absent at t = 0)
        constexpr int NK = 2 * H / KS / NW;
        const int koff = wid * (2 * H / NW);
        if (koff < H || t > 0) {
            const WT* a = (const WT*)p.w2 + (size_t)arow * 2 * H + koff;
            const WT* b[NBT];
#pragma unroll
            for (int i = 0; i < NBT; ++i)
                b[i] = koff < H ? h1 + ((size_t)bld[i] * p.T + t) * H + koff : h2 + ((size_t)bld[i] * p.T + (t - 1)) * H + (koff - H);
            seg_mma<WT, NK, NBT>(a, b, kq, acc);
        }
    }
#pragma unroll
    for (int i = 0; i < NBT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wid][i][kq * 4 + j][rc] = acc[i][j];  // D: row = 4*kq + j, col = rc
    __syncthreads();
    if (e_on) {
        float pre[4];
        const float x_skip = (float)xs_raw;
        pre[0] = layer == 0 ? (float)xg0 : bs0;
        pre[1] = layer == 0 ? (float)xg1v : bs1;
        pre[2] = layer == 0 ? (float)xg2 : bs2;
        pre[3] = layer == 0 ? (float)xg3 : bs3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = g * 4 + e_uu;
#pragma unroll
            for (int w8 = 0; w8 < NW; ++w8) pre[g] += red[w8][e_tile][r][e_bb];
        }
        const float ig = sigm(pre[0]), fg = sigm(pre[1]), gg = tanhf(pre[2]), og = sigm(pre[3]);
        const float c = fg * c_prev + ig * gg;
        cst[(size_t)e_b * H + e_u] = c;
        const float h = og * tanhf(c);
        const size_t hi = ((size_t)e_b * p.T + t) * H + e_u;
        hout[hi] = (WT)h;
        if (gsave && (p.B_save <= 0 || e_b < p.B_save)) {
            *(float4*)(gsave + (((size_t)e_b * p.T + t) * H + e_u) * 4) = make_float4(ig, fg, gg, og);
            csave[hi] = c;
        }
        if (layer == 1) ((AT*)p.y)[hi] = (AT)(h + x_skip);
    }
}

template <typename WT, typename AT, int H, int NBT, int NW>
__global__ __launch_bounds__(NW * 64) void lstm_bwd_fast_kernel(ga_lstm_bwd_params p, int s, int b_base0) {
    __shared__ float red[NW][NBT][16][17];
    const int b_base = b_base0 + blockIdx.z * (NBT * 16);
    constexpr int KS = Mma<WT>::KSTEP;
    constexpr int NK = 4 * H / KS / NW;
    const int layer = blockIdx.y;  // 0: layer 2 (leads), 1: layer 1
    const int t = layer == 0 ? p.T - 1 - s : p.T - s;
    if (t < 0 || t >= p.T) return;
    const int u0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int rc = lane & 15, kq = lane >> 4;
    const WT* __restrict__ dg1 = (const WT*)p.dg1;
    const WT* __restrict__ dg2 = (const WT*)p.dg2;
    const float* gates = layer == 0 ? p.gates2 : p.gates1;
    const float* cs = layer == 0 ? p.cs2 : p.cs1;
    float* dcs = layer == 0 ? p.dc2 : p.dc1;
    WT* dgout = layer == 0 ? (WT*)p.dg2 : (WT*)p.dg1;

    // epilogue role: threads 0..255 = 16 units x 16 batch columns, looped over the NBT tiles; inputs requested first
    // (unit fastest: one load instruction covers 16 consecutive units = 64 contiguous bytes of 4 batch rows)
    const int e_uu = tid & 15, e_bb = (tid >> 4) & 15, e_u = u0 + e_uu;
    const bool e_on = tid < 256;
    float e_ig[NBT], e_fg[NBT], e_gg[NBT], e_og[NBT], e_c[NBT], e_cp[NBT], e_dc[NBT];
    AT e_dyr[NBT];
#pragma unroll
    for (int i = 0; i < NBT; ++i) {  // unconditional loads from clamped addresses (see the forward kernel)
        const int b = b_base + i * 16 + e_bb;
        const int bl = b < p.B ? b : p.B - 1;
        const size_t hi = ((size_t)bl * p.T + t) * H + e_u;
        const float4 gs = *(const float4*)(gates + (((size_t)bl * p.T + t) * H + e_u) * 4);
        e_ig[i] = gs.x; e_fg[i] = gs.y; e_gg[i] = gs.z; e_og[i] = gs.w;
        e_c[i] = cs[hi];
        e_cp[i] = cs[t > 0 ? hi - H : hi];
        e_dc[i] = dcs[(size_t)bl * H + e_u];
        e_dyr[i] = ((const AT*)p.dy)[hi];
    }

    f32x4 acc[NBT];
#pragma unroll
    for (int i = 0; i < NBT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int bld[NBT];
#pragma unroll
    for (int i = 0; i < NBT; ++i) {
        const int bcol = b_base + i * 16 + rc;
        bld[i] = bcol < p.B ? bcol : p.B - 1;
    }
    const int koff = wid * (4 * H / NW);  // this wave's share of K = 4H
    const WT* b[NBT];
    if (layer == 0) {
        // dh2_t[u] = dy_t[u] + sum_k dg2_{t+1}[k] * W_hh_l1[k][u]       (w2t row H + u)
        if (t + 1 < p.T) {
#pragma unroll
            for (int i = 0; i < NBT; ++i) b[i] = dg2 + ((size_t)bld[i] * p.T + (t + 1)) * 4 * H + koff;
            seg_mma<WT, NK, NBT>((const WT*)p.w2t + (size_t)(H + u0 + rc) * 4 * H + koff, b, kq, acc);
        }
    } else {
        // dh1_t[u] = sum_k dg2_t[k] * W_ih_l1[k][u] + sum_k dg1_{t+1}[k] * W_hh_l0[k][u]
#pragma unroll
        for (int i = 0; i < NBT; ++i) b[i] = dg2 + ((size_t)bld[i] * p.T + t) * 4 * H + koff;
        const WT* a2 = (const WT*)p.w2t + (size_t)(u0 + rc) * 4 * H + koff;
        if (t + 1 < p.T) {
            const WT* bn[NBT];
#pragma unroll
            for (int i = 0; i < NBT; ++i) bn[i] = dg1 + ((size_t)bld[i] * p.T + (t + 1)) * 4 * H + koff;
            const WT* a1 = (const WT*)p.w1t + (size_t)(u0 + rc) * 4 * H + koff;
            if constexpr (sizeof(WT) == 2 && NK * (1 + NBT) <= 16) {
                seg_mma2<WT, NK, NBT>(a2, b, a1, bn, kq, acc);
            } else {
                seg_mma<WT, NK, NBT>(a2, b, kq, acc);
                seg_mma<WT, NK, NBT>(a1, bn, kq, acc);
            }
        } else {
            seg_mma<WT, NK, NBT>(a2, b, kq, acc);
        }
    }
#pragma unroll
    for (int i = 0; i < NBT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wid][i][kq * 4 + j][rc] = acc[i][j];  // row = unit, col = batch
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NBT; ++i) {
        const int bq = b_base + i * 16 + e_bb;
        if (bq < p.B && e_on) {
            float dh = layer == 0 ? (float)e_dyr[i] : 0.f;
#pragma unroll
            for (int w8 = 0; w8 < NW; ++w8) dh += red[w8][i][e_uu][e_bb];
            const float cprev = t > 0 ? e_cp[i] : 0.f;
            const float ig = e_ig[i], fg = e_fg[i], gg = e_gg[i], og = e_og[i];
            const float tc = tanhf(e_c[i]);
            const float dc = e_dc[i] + dh * og * (1.f - tc * tc);
            dcs[(size_t)bq * H + e_u] = dc * fg;
            WT* o = dgout + ((size_t)bq * p.T + t) * 4 * H + e_u;
            o[0] = (WT)(dc * gg * ig * (1.f - ig));
            o[H] = (WT)(dc * cprev * fg * (1.f - fg));
            o[2 * H] = (WT)(dc * ig * (1.f - gg * gg));
            o[3 * H] = (WT)(dh * tc * og * (1.f - og));
        }
    }
}

#ifndef GA_LSTM_FNW
#define GA_LSTM_FNW 4
#endif
#ifndef GA_LSTM_BNW
#define GA_LSTM_BNW 8
#endif
template <typename WT, typename AT, int H>
static void launch_fwd_fast(const ga_lstm_params& p, int st, hipStream_t s) {
    constexpr int FNW = GA_LSTM_FNW;
    if (p.B > 64) {  // every 64-item chunk of the step in ONE launch (columns past B are clamped on load and never stored)
        const dim3 gridz(H / 4, 2, (p.B + 63) / 64);
        lstm_fwd_fast_kernel<WT, AT, H, 4, FNW><<<gridz, FNW * 64, 0, s>>>(p, st, 0);
        return;
    }
    const dim3 grid(H / 4, 2);
    for (int b0 = 0; b0 < p.B; b0 += 64) {
        const int nbt = (((p.B - b0) < 64 ? (p.B - b0) : 64) + 15) / 16;
        switch (nbt) {
            case 1: lstm_fwd_fast_kernel<WT, AT, H, 1, FNW><<<grid, FNW * 64, 0, s>>>(p, st, b0); break;
            case 2: lstm_fwd_fast_kernel<WT, AT, H, 2, FNW><<<grid, FNW * 64, 0, s>>>(p, st, b0); break;
            case 3: lstm_fwd_fast_kernel<WT, AT, H, 3, FNW><<<grid, FNW * 64, 0, s>>>(p, st, b0); break;
            default: lstm_fwd_fast_kernel<WT, AT, H, 4, FNW><<<grid, FNW * 64, 0, s>>>(p, st, b0); break;
        }
    }
}

template <typename WT, typename AT, int H>
static void launch_bwd_fast(const ga_lstm_bwd_params& p, int st, hipStream_t s) {
    constexpr int BNW = GA_LSTM_BNW;
    if (p.B > 32) {
        const dim3 gridz(H / 16, 2, (p.B + 31) / 32);
        lstm_bwd_fast_kernel<WT, AT, H, 2, BNW><<<gridz, BNW * 64, 0, s>>>(p, st, 0);
        return;
    }
    const dim3 grid(H / 16, 2);
    for (int b0 = 0; b0 < p.B; b0 += 32) {
        const int nbt = (((p.B - b0) < 32 ? (p.B - b0) : 32) + 15) / 16;
        if (nbt == 1) lstm_bwd_fast_kernel<WT, AT, H, 1, BNW><<<grid, BNW * 64, 0, s>>>(p, st, b0);
        else lstm_bwd_fast_kernel<WT, AT, H, 2, BNW><<<grid, BNW * 64, 0, s>>>(p, st, b0);
    }
}

// ------------------------------------------------------------------------------------------------
int ga_lstm_fwd_persist(const ga_lstm_params& p, hipStream_t s);
int ga_lstm_bwd_persist(const ga_lstm_bwd_params& p, hipStream_t s);

extern "C" {

int ga_lstm_fwd(const ga_lstm_params* hp, hipStream_t s) {
    if (!hp) return GA_ERR_ARG;
    const ga_lstm_params& p = *hp;
    if (!p.w1 || !p.w2 || !p.bias2 || !p.xg1 || !p.x_in || !p.y) return GA_ERR_ARG;
    if (p.B <= 0 || p.T <= 0 || p.H <= 0 || p.H % 64) return GA_ERR_ARG;
    if ((p.gates1 == nullptr) != (p.gates2 == nullptr) || (p.gates1 && (!p.cs1 || !p.cs2))) return GA_ERR_ARG;
    if (((uintptr_t)p.w1 | (uintptr_t)p.w2) & 15) return GA_ERR_ALIGN;
    if (p.ws) {  // one persistent launch (lstm_persist.hip) when it applies
        const int rc = ga_lstm_fwd_persist(p, s);
        if (rc != GA_ERR_UNSUPPORTED) return rc;
    }
    if (!p.h1 || !p.h2 || !p.c1 || !p.c2) return GA_ERR_ARG;
    if (((uintptr_t)p.h1 | (uintptr_t)p.h2) & 15) return GA_ERR_ALIGN;
    const dim3 grid(p.H / 4, 2);
    for (int st = 0; st <= p.T; ++st) {
        if (p.H == 512 && p.w_dtype == GA_DTYPE_BF16 && p.act_dtype == GA_DTYPE_BF16)
            launch_fwd_fast<bf16_t, bf16_t, 512>(p, st, s);
        else if (p.H == 512 && p.w_dtype == GA_DTYPE_F32 && p.act_dtype == GA_DTYPE_F32)
            launch_fwd_fast<float, float, 512>(p, st, s);
        else if (p.w_dtype == GA_DTYPE_BF16 && p.act_dtype == GA_DTYPE_BF16)
            lstm_fwd_step_kernel<bf16_t, bf16_t><<<grid, 256, 0, s>>>(p, st);
        else if (p.w_dtype == GA_DTYPE_F32 && p.act_dtype == GA_DTYPE_F32)
            lstm_fwd_step_kernel<float, float><<<grid, 256, 0, s>>>(p, st);
        else
            return GA_ERR_ARG;
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_lstm_bwd(const ga_lstm_bwd_params* hp, hipStream_t s) {
    if (!hp) return GA_ERR_ARG;
    const ga_lstm_bwd_params& p = *hp;
    if (!p.w1t || !p.w2t || !p.dy || !p.gates1 || !p.gates2 || !p.cs1 || !p.cs2 || !p.dg1) return GA_ERR_ARG;
    if (p.B <= 0 || p.T <= 0 || p.H <= 0 || p.H % 64) return GA_ERR_ARG;
    if (((uintptr_t)p.w1t | (uintptr_t)p.w2t | (uintptr_t)p.dg1) & 15) return GA_ERR_ALIGN;
    if (p.ws) {
        const int rc = ga_lstm_bwd_persist(p, s);
        if (rc != GA_ERR_UNSUPPORTED) return rc;
    }
    if (!p.dg2 || !p.dc1 || !p.dc2) return GA_ERR_ARG;
    if ((uintptr_t)p.dg2 & 15) return GA_ERR_ALIGN;
    const dim3 grid(p.H / 16, 2);
    for (int st = 0; st <= p.T; ++st) {
        if (p.H == 512 && p.w_dtype == GA_DTYPE_BF16 && p.act_dtype == GA_DTYPE_BF16)
            launch_bwd_fast<bf16_t, bf16_t, 512>(p, st, s);
        else if (p.H == 512 && p.w_dtype == GA_DTYPE_F32 && p.act_dtype == GA_DTYPE_F32)
            launch_bwd_fast<float, float, 512>(p, st, s);
        else if (p.w_dtype == GA_DTYPE_BF16 && p.act_dtype == GA_DTYPE_BF16)
            lstm_bwd_step_kernel<bf16_t, bf16_t><<<grid, 256, 0, s>>>(p, st);
        else if (p.w_dtype == GA_DTYPE_F32 && p.act_dtype == GA_DTYPE_F32)
            lstm_bwd_step_kernel<float, float><<<grid, 256, 0, s>>>(p, st);
        else
            return GA_ERR_ARG;
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}

}  // extern "C"
