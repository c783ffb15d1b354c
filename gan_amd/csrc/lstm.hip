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
    // a_row / b_row: pointers to element k=0 of this lane's row (A) / column (B); lane group kq = lane >> 4
    static __device__ __forceinline__ f32x4 run(const bf16_t* a_row, const bf16_t* b_row, int k0, int kq, f32x4 acc) {
        const bf16x8 a = *(const bf16x8*)(a_row + k0 + 8 * kq);
        const bf16x8 b = *(const bf16x8*)(b_row + k0 + 8 * kq);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int KSTEP = 16;
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
                if (gsave) {
                    float* gs = gsave + ((size_t)b * p.T + t) * 4 * H + u;
                    gs[0] = ig; gs[H] = fg; gs[2 * H] = gg; gs[3 * H] = og;
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
                const float* gs = gates + ((size_t)b * p.T + t) * 4 * H + u;
                const float ig = gs[0], fg = gs[H], gg = gs[2 * H], og = gs[3 * H];
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
extern "C" {

int ga_lstm_fwd(const ga_lstm_params* hp, hipStream_t s) {
    if (!hp) return GA_ERR_ARG;
    const ga_lstm_params& p = *hp;
    if (!p.w1 || !p.w2 || !p.bias2 || !p.xg1 || !p.x_in || !p.y || !p.h1 || !p.h2 || !p.c1 || !p.c2) return GA_ERR_ARG;
    if (p.B <= 0 || p.T <= 0 || p.H <= 0 || p.H % 64) return GA_ERR_ARG;
    if ((p.gates1 == nullptr) != (p.gates2 == nullptr) || (p.gates1 && (!p.cs1 || !p.cs2))) return GA_ERR_ARG;
    if (((uintptr_t)p.w1 | (uintptr_t)p.w2 | (uintptr_t)p.h1 | (uintptr_t)p.h2) & 15) return GA_ERR_ALIGN;
    const dim3 grid(p.H / 4, 2);
    for (int st = 0; st <= p.T; ++st) {
        if (p.w_dtype == GA_DTYPE_BF16 && p.act_dtype == GA_DTYPE_BF16)
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
    if (!p.w1t || !p.w2t || !p.dy || !p.gates1 || !p.gates2 || !p.cs1 || !p.cs2 || !p.dg1 || !p.dg2 || !p.dc1 || !p.dc2) return GA_ERR_ARG;
    if (p.B <= 0 || p.T <= 0 || p.H <= 0 || p.H % 64) return GA_ERR_ARG;
    if (((uintptr_t)p.w1t | (uintptr_t)p.w2t | (uintptr_t)p.dg1 | (uintptr_t)p.dg2) & 15) return GA_ERR_ALIGN;
    const dim3 grid(p.H / 16, 2);
    for (int st = 0; st <= p.T; ++st) {
        if (p.w_dtype == GA_DTYPE_BF16 && p.act_dtype == GA_DTYPE_BF16)
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
