// HBM-bound kernels for the thin ends of the networks (one input channel or a single output channel), where a GEMM
// view would leave MFMA tiles empty: D block 0 (1 -> 32, k15), D's logit conv (512 -> 1, k3), the decoder's last conv
// (32 -> 1, k7) and their gradients.  Every thread moves 16-byte vectors of 8 channels; fp32 accumulation.
#include "common.h"

template <typename T> struct Vec8;
template <> struct Vec8<float> {
    float v[8];
    __device__ __forceinline__ void load(const float* p) {
        const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct Vec8<bf16_t> {
    float v[8];
    typedef __attribute__((ext_vector_type(8))) __bf16 raw_t;
    __device__ __forceinline__ void load(const bf16_t* p) {
        const raw_t r = *(const raw_t*)p;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
    }
    __device__ __forceinline__ void store(bf16_t* p) const {
        raw_t r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[i];
        *(raw_t*)p = r;
    }
};

// ------------------------------------------------------------------------------------------------
// forward form, M <= 4, one THREAD per output step: y[t][m] = sum_j sum_c w[m][j][c] * pre(x[row(t,j)][c]).
// Weights (M*J*Cin floats) sit in LDS and are read as broadcasts; x rows are read as 16-byte vectors and overlap
// between neighbouring threads (served by L1/L2).  Cin % 8 == 0.
template <typename T>
__global__ __launch_bounds__(256) void conv_fewout_rows_kernel(ConvParams p) {
    extern __shared__ float wsm[];  // [M][J*Cin]
    const T* __restrict__ x = (const T*)p.x;
    const int phase = blockIdx.z % p.nphase, b = blockIdx.z / p.nphase;
    const int KC = p.J * p.Cin;
    const T* __restrict__ w = (const T*)p.w + (size_t)phase * p.M * KC;
    for (int i = threadIdx.x; i < p.M * KC; i += blockDim.x) wsm[i] = (float)w[i];
    __syncthreads();
    const int off0 = p.off0 + p.phase_off0[phase];
    x += (size_t)b * p.Lin * p.Cin;
    for (int t = p.t0 + blockIdx.x * blockDim.x + threadIdx.x; t < p.t0 + p.Lout; t += gridDim.x * blockDim.x) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < p.J; ++j) {
            int row = t * p.in_stride + j * p.tap_step + off0;
            if (row < 0 && p.pad_mode == 1) row = -row;
            if (row < 0 || row >= p.Lin) continue;
            const T* xr = x + (size_t)row * p.Cin;
            const float* wj = wsm + j * p.Cin;
            for (int c = 0; c < p.Cin; c += 8) {
                Vec8<T> xv;
                xv.load(xr + c);
                if (p.pre_act) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xv.v[e] = act_apply(p.pre_act, xv.v[e]);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (m < p.M) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[m] = fmaf(xv.v[e], wj[m * KC + c + e], acc[m]);
                    }
                }
            }
        }
        const int orow = t * p.out_stride + phase;
        if (orow < 0 || orow >= p.LoutTot) continue;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m >= p.M) continue;
            const size_t idx = ((size_t)b * p.LoutTot + orow) * p.M + m;
            float v = acc[m];
            if (p.bias) v += p.bias[m];
            v = act_apply(p.post_act, v);
            if (p.dact_mode) v *= dact_apply(p.dact_mode, (float)((const T*)p.dact_src)[idx]);
            if (p.res) v += (float)((const T*)p.res)[idx];
            if (p.out_f32) { ((float*)p.y)[idx] = v; continue; }
            if (p.accumulate) v += (float)((T*)p.y)[idx];
            ((T*)p.y)[idx] = (T)v;
        }
    }
}

// forward form, Cin == 1, M % 8 == 0 (D block 0): thread = (output step, group of 8 output channels), 16-byte stores.
template <typename T>
__global__ __launch_bounds__(256) void conv_cin1_vec_kernel(ConvParams p) {
    extern __shared__ float wsm[];  // [J][M] (transposed for contiguous 8-channel reads) + [M] bias
    const T* __restrict__ x = (const T*)p.x + (size_t)blockIdx.z * p.Lin;
    const T* __restrict__ w = (const T*)p.w;
    for (int i = threadIdx.x; i < p.M * p.J; i += blockDim.x) wsm[(i % p.J) * p.M + i / p.J] = (float)w[i];
    for (int i = threadIdx.x; i < p.M; i += blockDim.x) wsm[p.M * p.J + i] = p.bias ? p.bias[i] : 0.f;
    __syncthreads();
    const int parts = p.M / 8;
    const size_t total = (size_t)p.Lout * parts;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int t = p.t0 + (int)(e / parts), m0 = (int)(e % parts) * 8;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = wsm[p.M * p.J + m0 + i];
        for (int j = 0; j < p.J; ++j) {
            int row = t * p.in_stride + j * p.tap_step + p.off0;
            if (row < 0 && p.pad_mode == 1) row = -row;
            if (row < 0 || row >= p.Lin) continue;
            float xv = (float)x[row];
            if (p.pre_act) xv = act_apply(p.pre_act, xv);
            const float* wj = wsm + j * p.M + m0;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(xv, wj[i], acc[i]);
        }
        const int orow = t * p.out_stride;
        if (orow >= p.LoutTot) continue;
        const size_t idx = ((size_t)blockIdx.z * p.LoutTot + orow) * p.M + m0;
        Vec8<T> o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] = act_apply(p.post_act, acc[i]);
        if (p.dact_mode) {
            Vec8<T> d;
            d.load((const T*)p.dact_src + idx);
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] *= dact_apply(p.dact_mode, d.v[i]);
        }
        if (p.res) {
            Vec8<T> d;
            d.load((const T*)p.res + idx);
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] += d.v[i];
        }
        if (p.accumulate) {
            Vec8<T> d;
            d.load((T*)p.y + idx);
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] += d.v[i];
        }
        o.store((T*)p.y + idx);
    }
}


// ------------------------------------------------------------------------------------------------
// LDS-staged variants.  The per-thread kernels above re-read every input row once per tap through L1 (and re-apply
// the ELU prologue each time); here a block stages its input window ONCE with coalesced 16-byte loads (prologue
// activation applied once), and all tap reads come from LDS.
//
// M <= 4: block = 256 consecutive output steps of one (batch item, phase); LDS rows are padded to an odd number of
// 16-byte slots so that the lanes' ds_read_b128 of consecutive rows do not collide.
typedef unsigned int fo_u32x4 __attribute__((ext_vector_type(4)));
typedef float fo_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) __bf16 fo_bf16x2;

// ELU on eight packed bf16: f > -2^-12 returns f (ELU(f) = f there within half a bf16 step), else exp(f) - 1; two elements per
// step so that the multiply / add / convert are packed instructions (the arithmetic of conv_stream.hip / conv_swin.hip)
__device__ __forceinline__ fo_u32x4 fo_elu8(fo_u32x4 in) {
    fo_u32x4 out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fo_f32x2 f;
        f.x = __uint_as_float(in[i] << 16);
        f.y = __uint_as_float(in[i] & 0xffff0000u);
        const fo_f32x2 t = f * 1.4426950408889634f;
        fo_f32x2 e;
        e.x = __builtin_amdgcn_exp2f(t.x);
        e.y = __builtin_amdgcn_exp2f(t.y);
        e = e - 1.0f;
        union { fo_bf16x2 b; unsigned int u; } pk;
        pk.b[0] = (__bf16)(f.x > -2.44140625e-4f ? f.x : e.x);
        pk.b[1] = (__bf16)(f.y > -2.44140625e-4f ? f.y : e.y);
        out[i] = pk.u;
    }
    return out;
}
__device__ __forceinline__ float fo_dot2(unsigned int a, unsigned int b, float c) {
    union { unsigned int u; fo_bf16x2 v; } x, y;
    x.u = a;
    y.u = b;
    return __builtin_amdgcn_fdot2_f32_bf16(x.v, y.v, c, false);  // v_dot2c_f32_bf16: c + a.lo*b.lo + a.hi*b.hi in fp32
}

template <typename T>
__global__ __launch_bounds__(256) void conv_fewout_lds_kernel(ConvParams p, int span, int pitch) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wsm = (float*)smem_raw;                                  // [M][J*Cin]
    const int KC = p.J * p.Cin;
    char* xs = smem_raw + (((size_t)p.M * KC * 4 + 15) & ~(size_t)15);  // [span][pitch]
    const int phase = blockIdx.z % p.nphase, b = blockIdx.z / p.nphase;
    const T* __restrict__ w = (const T*)p.w + (size_t)phase * p.M * KC;
    const T* __restrict__ x = (const T*)p.x + (size_t)b * p.Lin * p.Cin;
    const int off0 = p.off0 + p.phase_off0[phase];
    const int tblk = p.t0 + blockIdx.x * 256;
    const int R0 = tblk * p.in_stride + off0;
    const int parts = p.Cin / 8;
    if constexpr (sizeof(T) == 2) {
        // bf16, one output channel (decoder tail, D block 0 data gradient), no or ELU input activation: everything stays packed --
        // 16-byte pieces straight into LDS (ELU on the packed pairs), the weights as bf16 pairs too, and the inner product by the
        // packed dot instruction: 4 vector instructions per 8 channels where unpacking + eight FMAs took 16, which, not memory,
        // set this kernel's pace (2.0-2.7 TB/s)
        if (p.M == 1 && (p.pre_act == GA_ACT_NONE || p.pre_act == GA_ACT_ELU) && KC % 8 == 0 && ((uintptr_t)w & 15) == 0) {
            fo_u32x4* wpk = (fo_u32x4*)smem_raw;                    // [J*Cin / 8] pieces of 8 bf16
            for (int i = threadIdx.x; i < KC / 8; i += 256) wpk[i] = ((const fo_u32x4*)w)[i];
            const bool elu = p.pre_act == GA_ACT_ELU;
            // the window in rounds of eight pieces per thread, every load of a round issued before the first is used: one memory
            // round trip per round, not per piece (a block has about five pieces per thread)
            const fo_u32x4* const zp = (const fo_u32x4*)ga_zero_page;
            for (int base = 0; base < span * parts; base += 8 * 256) {
                fo_u32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256 + threadIdx.x;
                    const int row = idx / parts, part = idx - row * parts;
                    int R = R0 + row;
                    if (R < 0 && p.pad_mode == 1) R = -R;
                    const bool ok = idx < span * parts && R >= 0 && R < p.Lin;
                    v[u] = *(ok ? (const fo_u32x4*)(x + (size_t)R * p.Cin + part * 8) : zp);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256 + threadIdx.x;
                    const int row = idx / parts, part = idx - row * parts;
                    if (idx < span * parts) *(fo_u32x4*)(xs + (size_t)row * pitch + part * 16) = elu ? fo_elu8(v[u]) : v[u];
                }
            }
            __syncthreads();
            const int t = tblk + threadIdx.x;
            if (t >= p.t0 + p.Lout) return;
            float a0 = 0.f, a1 = 0.f;
            for (int j = 0; j < p.J; ++j) {
                const char* xr = xs + (size_t)(threadIdx.x * p.in_stride + j * p.tap_step) * pitch;
                const fo_u32x4* wj = wpk + j * parts;
                for (int c = 0; c < parts; ++c) {
                    const fo_u32x4 xv = *(const fo_u32x4*)(xr + c * 16), wv = wj[c];
                    a0 = fo_dot2(xv.x, wv.x, a0);
                    a1 = fo_dot2(xv.y, wv.y, a1);
                    a0 = fo_dot2(xv.z, wv.z, a0);
                    a1 = fo_dot2(xv.w, wv.w, a1);
                }
            }
            const int orow = t * p.out_stride + phase;
            if (orow < 0 || orow >= p.LoutTot) return;
            const size_t idx = ((size_t)b * p.LoutTot + orow);
            float v = a0 + a1;
            if (p.bias) v += p.bias[0];
            v = act_apply(p.post_act, v);
            if (p.dact_mode) v *= dact_apply(p.dact_mode, (float)((const T*)p.dact_src)[idx]);
            if (p.res) v += (float)((const T*)p.res)[idx];
            if (p.out_f32) { ((float*)p.y)[idx] = v; return; }
            if (p.accumulate) v += (float)((T*)p.y)[idx];
            ((T*)p.y)[idx] = (T)v;
            return;
        }
    }
    for (int i = threadIdx.x; i < p.M * KC; i += 256) wsm[i] = (float)w[i];
    for (int idx = threadIdx.x; idx < span * parts; idx += 256) {
        const int row = idx / parts, part = idx - row * parts;
        int R = R0 + row;
        if (R < 0 && p.pad_mode == 1) R = -R;
        Vec8<T> v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v.v[e] = 0.f;
        if (R >= 0 && R < p.Lin) {
            v.load(x + (size_t)R * p.Cin + part * 8);
            if (p.pre_act == GA_ACT_ELU && sizeof(T) == 2) {
                // bf16 staging: ELU(v) = v down to -2^-12 within half a bf16 step, and exp(v) - 1 below that loses nothing
                // a bf16 keeps; expm1f's exactness near zero (the fp32 path below) costs ~50 instructions per element
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = v.v[e];
                    v.v[e] = f > -0.000244140625f ? f : __expf(f) - 1.f;
                }
            } else if (p.pre_act) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v.v[e] = act_apply(p.pre_act, v.v[e]);
            }
        }
        v.store((T*)(xs + (size_t)row * pitch) + part * 8);
    }
    __syncthreads();
    const int t = tblk + threadIdx.x;
    if (t >= p.t0 + p.Lout) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.M == 1) {
        // single output channel (decoder tail, D block 0 data gradient): the 8 weights of a piece as two 16-byte broadcast
        // reads -- with one 4-byte read per weight the LDS instruction stream, not memory, set the pace
        for (int j = 0; j < p.J; ++j) {
            const T* xr = (const T*)(xs + (size_t)(threadIdx.x * p.in_stride + j * p.tap_step) * pitch);
            const float* wj = wsm + j * p.Cin;
            for (int c = 0; c < p.Cin; c += 8) {
                Vec8<T> xv;
                xv.load(xr + c);
                const float4 w0 = *(const float4*)(wj + c), w1 = *(const float4*)(wj + c + 4);
                acc[0] = fmaf(xv.v[0], w0.x, acc[0]); acc[0] = fmaf(xv.v[1], w0.y, acc[0]);
                acc[0] = fmaf(xv.v[2], w0.z, acc[0]); acc[0] = fmaf(xv.v[3], w0.w, acc[0]);
                acc[0] = fmaf(xv.v[4], w1.x, acc[0]); acc[0] = fmaf(xv.v[5], w1.y, acc[0]);
                acc[0] = fmaf(xv.v[6], w1.z, acc[0]); acc[0] = fmaf(xv.v[7], w1.w, acc[0]);
            }
        }
    } else {
        for (int j = 0; j < p.J; ++j) {
            const T* xr = (const T*)(xs + (size_t)(threadIdx.x * p.in_stride + j * p.tap_step) * pitch);
            const float* wj = wsm + j * p.Cin;
            for (int c = 0; c < p.Cin; c += 8) {
                Vec8<T> xv;
                xv.load(xr + c);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (m < p.M) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[m] = fmaf(xv.v[e], wj[m * KC + c + e], acc[m]);
                    }
                }
            }
        }
    }
    const int orow = t * p.out_stride + phase;
    if (orow < 0 || orow >= p.LoutTot) return;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m >= p.M) continue;
        const size_t idx = ((size_t)b * p.LoutTot + orow) * p.M + m;
        float v = acc[m];
        if (p.bias) v += p.bias[m];
        v = act_apply(p.post_act, v);
        if (p.dact_mode) v *= dact_apply(p.dact_mode, (float)((const T*)p.dact_src)[idx]);
        if (p.res) v += (float)((const T*)p.res)[idx];
        if (p.out_f32) { ((float*)p.y)[idx] = v; continue; }
        if (p.accumulate) v += (float)((T*)p.y)[idx];
        ((T*)p.y)[idx] = (T)v;
    }
}

// Cin == 1, M % 8 == 0, J <= 16 (D block 0): a block covers CIN1_ROWS output steps; the input window sits in LDS as
// floats, every thread keeps the J x 8 weights of its 8 output channels in registers and writes 16-byte pieces, so one
// wave instruction stores whole contiguous output rows.
#define CIN1_ROWS 2048
template <typename T>
__global__ __launch_bounds__(256) void conv_cin1_lds_kernel(ConvParams p, int span) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* ws = (float*)smem_raw;          // [J][M], transposed so a thread's 8 channels are contiguous
    float* xs = ws + ((p.J * p.M + 3) & ~3);  // [span]
    const int b = blockIdx.z;
    const T* __restrict__ x = (const T*)p.x + (size_t)b * p.Lin;
    const T* __restrict__ w = (const T*)p.w;
    for (int i = threadIdx.x; i < p.M * p.J; i += 256) ws[(i % p.J) * p.M + i / p.J] = (float)w[i];
    const int tblk = p.t0 + blockIdx.x * CIN1_ROWS;
    const int R0 = tblk * p.in_stride + p.off0;
    for (int i = threadIdx.x; i < span; i += 256) {
        int R = R0 + i;
        if (R < 0 && p.pad_mode == 1) R = -R;
        float v = (R >= 0 && R < p.Lin) ? (float)x[R] : 0.f;
        if (p.pre_act) v = act_apply(p.pre_act, v);
        xs[i] = v;
    }
    const int parts = p.M / 8, rows_per_pass = 256 / parts;
    const int part = threadIdx.x % parts, rl = threadIdx.x / parts;
    const int m0 = part * 8;
    float wr[16][8], bs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bs[i] = p.bias ? p.bias[m0 + i] : 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int jc = j < p.J ? j : 0;
        const float4 w0 = *(const float4*)(ws + jc * p.M + m0), w1 = *(const float4*)(ws + jc * p.M + m0 + 4);
        const float keep = j < p.J ? 1.f : 0.f;
        wr[j][0] = w0.x * keep; wr[j][1] = w0.y * keep; wr[j][2] = w0.z * keep; wr[j][3] = w0.w * keep;
        wr[j][4] = w1.x * keep; wr[j][5] = w1.y * keep; wr[j][6] = w1.z * keep; wr[j][7] = w1.w * keep;
    }
    if (rl >= rows_per_pass) return;
    int tend = p.t0 + p.Lout - tblk;
    if (tend > CIN1_ROWS) tend = CIN1_ROWS;
    for (int tl = rl; tl < tend; tl += rows_per_pass) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = bs[i];
        const float* xw = xs + tl * p.in_stride;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < p.J) {
                const float xv = xw[j * p.tap_step];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fmaf(xv, wr[j][i], acc[i]);
            }
        }
        const int orow = (tblk + tl) * p.out_stride;
        if (orow >= p.LoutTot) continue;
        const size_t idx = ((size_t)b * p.LoutTot + orow) * p.M + m0;
        Vec8<T> o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] = act_apply(p.post_act, acc[i]);
        if (p.dact_mode) {
            Vec8<T> d;
            d.load((const T*)p.dact_src + idx);
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] *= dact_apply(p.dact_mode, d.v[i]);
        }
        if (p.res) {
            Vec8<T> d;
            d.load((const T*)p.res + idx);
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] += d.v[i];
        }
        if (p.accumulate) {
            Vec8<T> d;
            d.load((T*)p.y + idx);
#pragma unroll
            for (int i = 0; i < 8; ++i) o.v[i] += d.v[i];
        }
        o.store((T*)p.y + idx);
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient, Cin == 1, M % 8 == 0, J <= 16:  dw[m][j] += sum_{b,t} g[b][t][m] * x[b][t*S + j*step + off0]
// and, fused, the bias gradient db[m] += sum g.  A block owns a run of consecutive steps of one batch item.
#define WG1_ROWS 2048
template <typename T>
__global__ __launch_bounds__(256) void wgrad_cin1_kernel(WgradParams p, float* __restrict__ dbias, int chunks_per_b) {
    __shared__ float xs[WG1_ROWS * 4 + 64];   // window for in_stride <= 4
    __shared__ float red[256 * 17];
    const T* __restrict__ g = (const T*)p.g;
    const T* __restrict__ x = (const T*)p.x;
    const int b = blockIdx.x / chunks_per_b;
    const int t0 = (blockIdx.x % chunks_per_b) * WG1_ROWS;
    int nrows = p.Lout - t0;
    if (nrows > WG1_ROWS) nrows = WG1_ROWS;
    const int span = (nrows - 1) * p.in_stride + (p.J - 1) * p.tap_step + 1;
    const int R0 = t0 * p.in_stride + p.off0;
    for (int i = threadIdx.x; i < span; i += blockDim.x) {
        const int R = R0 + i;
        xs[i] = (R >= 0 && R < p.Lin) ? (float)x[(size_t)b * p.Lin + R] : 0.f;
    }
    __syncthreads();
    const int parts = p.M / 8;            // threads per row
    const int lanes = 256 / parts;        // rows in flight
    const int part = threadIdx.x % parts, rl = threadIdx.x / parts;
    float acc[8][16];
    float accb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        accb[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    }
    if (rl < lanes) {
        // 2 rows per trip, their loads issued together (the kernel is HBM-latency-bound otherwise)
        for (int r0 = rl; r0 < nrows; r0 += 2 * lanes) {
            Vec8<T> gv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = r0 + u * lanes;
                const int rc = r < nrows ? r : nrows - 1;
                gv[u].load(g + ((size_t)b * p.LgTot + (size_t)(t0 + rc) * p.g_stride + p.g_phase) * p.M + part * 8);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = r0 + u * lanes;
                if (r >= nrows) continue;
                const float* xw = xs + r * p.in_stride;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (j < p.J) {
                        const float xv = xw[j * p.tap_step];
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[i][j] = fmaf(gv[u].v[i], xv, acc[i][j]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) accb[i] += gv[u].v[i];
            }
        }
    }
    // reduce over the row lanes: 17 values (16 taps + bias) per (part, channel)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) red[threadIdx.x * 17 + j] = acc[i][j];
        red[threadIdx.x * 17 + 16] = accb[i];
        __syncthreads();
        // thread (part', j) sums over row lanes
        for (int o = threadIdx.x; o < parts * 17; o += blockDim.x) {
            const int pt = o / 17, j = o % 17;
            float s = 0.f;
            for (int l = 0; l < lanes; ++l) s += red[(l * parts + pt) * 17 + j];
            const int m = pt * 8 + i;
            if (j < p.J) atomicAdd(&p.dw[(size_t)m * p.J + j], s);
            else if (j == 16 && dbias) atomicAdd(&dbias[m], s);
        }
    }
}

// weight gradient, M == 1, Cin % 8 == 0, J <= 8: dw[j][c] += sum_{b,t} g[b][t] * x[b][t*S + j*step + off0][c]; db += sum g.
// Each thread owns 8 channels and walks a run of INPUT rows once, scattering each row into the taps that use it.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_m1_kernel(WgradParams p, float* __restrict__ dbias, int rows_per_block) {
    __shared__ float red[16];
    const T* __restrict__ g = (const T*)p.g;
    const T* __restrict__ x = (const T*)p.x;
    const int parts = p.Cin / 8;
    const int lanes = 256 / parts;
    const int part = threadIdx.x % parts, rl = threadIdx.x / parts;
    const int b = blockIdx.y;
    const int r_begin = blockIdx.x * rows_per_block;
    int r_end = r_begin + rows_per_block;
    if (r_end > p.Lin) r_end = p.Lin;
    float acc[8][8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
    if (rl < lanes) {
        for (int R = r_begin + rl; R < r_end; R += lanes) {
            Vec8<T> xv;
            xv.load(x + ((size_t)b * p.Lin + R) * p.Cin + part * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < p.J) {
                    const int num = R - j * p.tap_step - p.off0;   // t * S
                    if (num >= 0 && num % p.in_stride == 0) {
                        const int t = num / p.in_stride;
                        if (t < p.Lout) {
                            const float gv = (float)g[(size_t)b * p.LgTot + (size_t)t * p.g_stride + p.g_phase];
#pragma unroll
                            for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(gv, xv.v[i], acc[j][i]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < p.J)
#pragma unroll
                for (int i = 0; i < 8; ++i) atomicAdd(&p.dw[(size_t)j * p.Cin + part * 8 + i], acc[j][i]);
    }
    if (dbias && blockIdx.x == 0) {
        float s = 0.f;
        for (int t = threadIdx.x; t < p.Lout; t += blockDim.x) s += (float)g[(size_t)b * p.LgTot + (size_t)t * p.g_stride + p.g_phase];
        s = block_sum(s, red);
        if (threadIdx.x == 0) atomicAdd(dbias, s);
    }
}

// column sums with 16-byte loads: out[m] += sum_rows g[row][m], M % 8 == 0, M <= 2048
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ g, size_t rows, int M, float* __restrict__ out,
                                                          size_t rows_per_block) {
    __shared__ float red[256 * 8];
    const int parts = M / 8;
    const int lanes = 256 / parts > 0 ? 256 / parts : 1;
    const size_t r0 = (size_t)blockIdx.x * rows_per_block;
    size_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (int pbase = 0; pbase < parts; pbase += 256) {
        const int part = pbase + (threadIdx.x % (parts < 256 ? parts : 256));
        const int rl = parts < 256 ? threadIdx.x / parts : 0;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (rl < lanes && part < parts) {
            for (size_t r = r0 + rl; r < r1; r += lanes) {
                Vec8<T> v;
                v.load(g + r * M + part * 8);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += v.v[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = acc[i];
        __syncthreads();
        const int np = parts < 256 ? parts : 256;
        for (int o = threadIdx.x; o < np * 8; o += blockDim.x) {
            const int pt = o / 8, i = o % 8;
            float s = 0.f;
            if (parts < 256) {
                for (int l = 0; l < lanes; ++l) s += red[(l * parts + pt) * 8 + i];
            } else {
                s = red[pt * 8 + i];
            }
            if (pbase + pt < parts) atomicAdd(&out[(pbase + pt) * 8 + i], s);
        }
    }
}

// ------------------------------------------------------------------------------------------------
static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <typename T>
static int thin_conv_t(const ConvParams& p, hipStream_t s) {
    if (p.M <= 4) {
        // LDS-staged window when it fits (64 KiB) and the block grid stays within limits
        const long span = 255L * p.in_stride + (long)(p.J - 1) * p.tap_step + 1;
        const int pitch = p.Cin * (int)sizeof(T) + 16;
        const size_t wbytes = ((size_t)p.M * p.J * p.Cin * 4 + 15) & ~(size_t)15;
        const size_t lds = wbytes + (size_t)span * pitch;
        const long gxl = ((long)p.Lout + 255) / 256;
        if (p.in_stride >= 1 && p.tap_step >= 0 && lds <= 64 * 1024 && gxl <= 0x7fffffffL && (long)p.B * p.nphase <= 65535) {
            conv_fewout_lds_kernel<T><<<dim3((unsigned)gxl, 1, p.B * p.nphase), 256, lds, s>>>(p, (int)span, pitch);
        } else {
            int gx = (p.Lout + 255) / 256;
            if (gx > 16384) gx = 16384;
            conv_fewout_rows_kernel<T><<<dim3(gx, 1, p.B * p.nphase), 256, (size_t)p.M * p.J * p.Cin * sizeof(float), s>>>(p);
        }
    } else {
        const long span = (long)(CIN1_ROWS - 1) * p.in_stride + (long)(p.J - 1) * p.tap_step + 1;
        const size_t lds1 = (size_t)(((p.J * p.M + 3) & ~3) + span) * 4;
        if (p.J <= 16 && p.M <= 256 && 256 % (p.M / 8) == 0 && p.in_stride >= 1 && p.tap_step >= 0 && lds1 <= 60 * 1024 && p.B <= 65535) {
            const unsigned gx = (unsigned)((p.Lout + CIN1_ROWS - 1) / CIN1_ROWS);
            conv_cin1_lds_kernel<T><<<dim3(gx, 1, p.B), 256, lds1, s>>>(p, (int)span);
        } else {
            const size_t total = (size_t)p.Lout * (p.M / 8);
            size_t gx = (total + 255) / 256;
            if (gx > 8192) gx = 8192;
            conv_cin1_vec_kernel<T><<<dim3((unsigned)gx, 1, p.B), 256, (size_t)(p.M * p.J + p.M) * sizeof(float), s>>>(p);
        }
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}

// returns 1 when one of the vectorised thin forward kernels applies
int ga_conv_m1_wave_ok(const ConvParams& p, int esz);  // conv_direct.hip

int ga_thin_conv_supported(const ConvParams& p, int dtype) {
    const size_t esz = dtype == GA_DTYPE_BF16 ? 2 : 4;
    if (p.M <= 4) {
        if (ga_conv_m1_wave_ok(p, (int)esz)) return 0;  // wide single-output layers (D.out): the wave-per-step vector kernel
        if (p.Cin % 8 || (size_t)p.M * p.J * p.Cin * 4 > 48 * 1024) return 0;
        if (p.out_f32 && (p.res || p.dact_src || p.accumulate)) return 0;
        if ((long)p.Lout * p.B < 8192 && !p.out_f32) return 0;  // short problems: the wave-per-row kernel has more parallelism
        if (!aligned16(p.x) || (p.Cin * esz) % 16) return 0;
        return 1;
    }
    if (p.out_f32) return 0;
    if (p.Cin == 1 && p.nphase == 1 && p.M % 8 == 0 && (size_t)(p.M * p.J + p.M) * 4 <= 48 * 1024) {
        if (!aligned16(p.y) || !aligned16(p.res) || !aligned16(p.dact_src) || (p.M * esz) % 16) return 0;
        return 1;
    }
    return 0;
}
int ga_thin_conv_impl(const ConvParams& p, int dtype, hipStream_t s) {
    return dtype == GA_DTYPE_BF16 ? thin_conv_t<bf16_t>(p, s) : thin_conv_t<float>(p, s);
}

int ga_thin_wgrad_supported(const WgradParams& p, int dtype) {
    const size_t esz = dtype == GA_DTYPE_BF16 ? 2 : 4;
    if (p.Cin == 1 && p.M % 8 == 0 && p.M <= 256 && p.J <= 16 && p.in_stride >= 1 && p.in_stride <= 4 &&
        (p.J - 1) * p.tap_step < 64 && aligned16(p.g) && (p.M * esz) % 16 == 0)
        return 1;
    // (the row-walking kernel below takes any multiple of 32 channels at stride 1; the one-row-per-lane-group kernel needs the row's
    // 16-byte pieces to divide a workgroup)
    if (p.M == 1 && p.Cin % 8 == 0 && p.Cin <= 2048 && (256 % (p.Cin / 8) == 0 || (p.Cin % 32 == 0 && p.in_stride == 1)) && p.J <= 8 &&
        p.in_stride >= 1 && aligned16(p.x) && (p.Cin * esz) % 16 == 0)
        return 2;
    return 0;
}

// The same for wide inputs (D's last conv: 1024 channels, a few hundred rows per item): one workgroup per 32 channels walks
// EVERY row (64 row lanes x 4 channel parts), so each output is summed inside one workgroup -- no contended atomics -- and
// the loop is branch-free and unrolled so that a thread's loads are in flight together.  in_stride == 1.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_m1_wide_kernel(WgradParams p, float* __restrict__ dbias) {
    __shared__ float red[256 * 9];
    const T* __restrict__ g = (const T*)p.g;
    const T* __restrict__ x = (const T*)p.x;
    const int part = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const int c0 = blockIdx.x * 32 + part * 8;
    const long nrows = (long)p.B * p.Lin;
    float acc[8][8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
    float gsum = 0.f;
    // gridDim.y slices of the row range (the per-output sums already end in atomics): 16 workgroups walking every row
    // left the chip idle for 70 us on D's last conv
    const long per = ((nrows + gridDim.y - 1) / gridDim.y + 255) / 256 * 256;
    const long n_lo = (long)blockIdx.y * per, n_hi = n_lo + per < nrows ? n_lo + per : nrows;
    for (long n0 = n_lo + rl; n0 < n_hi; n0 += 64 * 4) {
        Vec8<T> xv[4];
        float gv[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long n = n0 + u * 64;
            const long nc = n < nrows ? n : nrows - 1;
            xv[u].load(x + nc * p.Cin + c0);
            const int b = (int)(nc / p.Lin), R = (int)(nc - (long)b * p.Lin);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = R - j * p.tap_step - p.off0;
                const bool ok = j < p.J && n < n_hi && t >= 0 && t < p.Lout;
                const int tc = t < 0 ? 0 : (t >= p.Lout ? p.Lout - 1 : t);
                const float v = (float)g[(size_t)b * p.LgTot + (size_t)tc * p.g_stride + p.g_phase];
                gv[u][j] = ok ? v : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[j][i] = fmaf(gv[u][j], xv[u].v[i], acc[j][i]);
    }
    // sum over the 64 row lanes: per tap, [256 threads][8 channels] through LDS
    for (int j = 0; j < p.J && j < 8; ++j) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) red[threadIdx.x * 9 + i] = acc[j][i];
        __syncthreads();
        if (threadIdx.x < 32) {
            const int pt = threadIdx.x >> 3, i = threadIdx.x & 7;
            float sum = 0.f;
            for (int l = 0; l < 64; ++l) sum += red[(l * 4 + pt) * 9 + i];
            atomicAdd(&p.dw[(size_t)j * p.Cin + blockIdx.x * 32 + threadIdx.x], sum);
        }
    }
    if (dbias && blockIdx.x == 0) {
        __syncthreads();
        for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < (long)p.B * p.Lout; i += 256L * gridDim.y) {
            const long b = i / p.Lout, t = i - b * p.Lout;
            gsum += (float)g[(size_t)b * p.LgTot + (size_t)t * p.g_stride + p.g_phase];
        }
        gsum = block_sum(gsum, red);
        if (threadIdx.x == 0) atomicAdd(dbias, gsum);
    }
}

template <typename T>
static int thin_wgrad_t(const WgradParams& p, float* dbias, int kind, hipStream_t s) {
    if (kind == 1) {
        const int chunks_per_b = (p.Lout + WG1_ROWS - 1) / WG1_ROWS;
        wgrad_cin1_kernel<T><<<p.B * chunks_per_b, 256, 0, s>>>(p, dbias, chunks_per_b);
    } else {
        // wide inputs, and narrower ones with very many rows (the mel decoder's last conv: 96 gathered channels x 1.5M rows, where
        // the kernel below it took 4.4 ms for one pass over 300 MB) or with a piece count that does not divide a workgroup
        const bool many_rows = (long)p.B * p.Lin >= (1L << 16) || 256 % (p.Cin / 8) != 0;
        if ((p.Cin >= 256 || many_rows) && p.Cin % 32 == 0 && p.in_stride == 1 && (long)p.B * p.Lin < (1L << 30)) {
            long ny = ((long)p.B * p.Lin + 1023) / 1024;  // >= 4 trips of 256 rows per workgroup
            const long cap = p.Cin >= 256 ? 32 : 1536 / (p.Cin / 32);  // ~1-1.5k workgroups in all
            if (ny > cap) ny = cap;
            if (ny < 1) ny = 1;
            wgrad_m1_wide_kernel<T><<<dim3(p.Cin / 32, (unsigned)ny), 256, 0, s>>>(p, dbias);
            GA_CHECK_LAUNCH();
            return GA_OK;
        }
        int rows_per_block = 256;
        int gx = (p.Lin + rows_per_block - 1) / rows_per_block;
        if (gx > 4096) { gx = 4096; rows_per_block = (p.Lin + gx - 1) / gx; }
        wgrad_m1_kernel<T><<<dim3(gx, p.B), 256, 0, s>>>(p, dbias, rows_per_block);
    }
    GA_CHECK_LAUNCH();
    return GA_OK;
}
int ga_thin_wgrad_impl(const WgradParams& p, float* dbias, int dtype, int kind, hipStream_t s) {
    return dtype == GA_DTYPE_BF16 ? thin_wgrad_t<bf16_t>(p, dbias, kind, s) : thin_wgrad_t<float>(p, dbias, kind, s);
}

int ga_colsum_vec_supported(const void* g, int dtype, int M) {
    const size_t esz = dtype == GA_DTYPE_BF16 ? 2 : 4;
    return M % 8 == 0 && M <= 2048 && aligned16(g) && (M * esz) % 16 == 0 && 256 % (M / 8) == 0;
}
int ga_colsum_vec_impl(const void* g, int dtype, size_t rows, int M, float* out, hipStream_t s) {
    size_t blocks = (rows + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const size_t rpb = (rows + blocks - 1) / blocks;
    blocks = (rows + rpb - 1) / rpb;
    if (dtype == GA_DTYPE_BF16)
        colsum_vec_kernel<bf16_t><<<(unsigned)blocks, 256, 0, s>>>((const bf16_t*)g, rows, M, out, rpb);
    else
        colsum_vec_kernel<float><<<(unsigned)blocks, 256, 0, s>>>((const float*)g, rows, M, out, rpb);
    GA_CHECK_LAUNCH();
    return GA_OK;
}
