// MR-STFT loss kernels either side of the DFT-basis GEMM (which is ga_conv on the framed signal).
// Reference: /root/reference/lab 3/src/lab3_codec_train.py:148-176 (torch.stft center=True reflect pad,
// abs, clamp_min(1e-8), Frobenius-ratio spectral convergence + log-magnitude L1).
#include "common.h"

__device__ __forceinline__ int reflect_idx(int s, int S) {
    if (s < 0) s = -s;
    if (s >= S) s = 2 * (S - 1) - s;
    return s;
}

template <typename TI, typename TO>
__global__ void stft_pad_kernel(const TI* __restrict__ x, TO* __restrict__ xp, int B, int S, int P, int hop, int rows, int rep) {
    const size_t per = (size_t)rows * hop, n = (size_t)B * per;
    const int Lp = S + 2 * P;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const int q = (int)(i % per);
        float v = 0.f;
        if (q < Lp) v = (float)x[(size_t)b * S + reflect_idx(q - P, S)];
        if (rep == 1) {
            xp[i] = (TO)v;
        } else {
            const int r = q / hop, c = q % hop;
            TO* o = xp + ((size_t)b * rows + r) * (size_t)(3 * hop);
            const TO hi = (TO)v;
            const TO lo = (TO)(v - (float)hi);
            o[c] = hi; o[hop + c] = lo; o[2 * hop + c] = hi;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void stft_loss_fwd_kernel(const T* __restrict__ sx, const T* __restrict__ sy, size_t frames, int nb,
                                                             int ld, float* __restrict__ out, int mag_add) {
    __shared__ float red[16];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const size_t n = frames * nb;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / nb;
        const int k = (int)(i % nb);
        const float xr = (float)sx[f * ld + k], xi = (float)sx[f * ld + nb + k];
        const float yr = (float)sy[f * ld + k], yi = (float)sy[f * ld + nb + k];
        // magnitude floor: clamp_min(1e-8) (lab3_codec_train.py:168-169) or + 1e-8 (lab3_losses.py:60-61)
        const float ax = sqrtf(xr * xr + xi * xi), ay = sqrtf(yr * yr + yi * yi);
        const float mx = mag_add ? ax + 1e-8f : fmaxf(ax, 1e-8f);
        const float my = mag_add ? ay + 1e-8f : fmaxf(ay, 1e-8f);
        const float d = my - mx;
        s0 = fmaf(d, d, s0);
        s1 = fmaf(my, my, s1);
        s2 += fabsf(logf(mx) - logf(my));
    }
    s0 = block_sum(s0, red);
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], s0);
        atomicAdd(&out[1], s1);
        atomicAdd(&out[2], s2);
    }
}

template <typename T>
__global__ void stft_loss_bwd_kernel(const T* __restrict__ sx, const T* __restrict__ sy, size_t frames, int nb, int ld,
                                     const float* __restrict__ coef, T* __restrict__ dsx, int mag_add) {
    const float c0 = coef[0], c2 = coef[2];
    const size_t n = frames * (size_t)ld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / ld;
        const int col = (int)(i % ld);
        float g = 0.f;
        if (col < 2 * nb) {
            const int k = col < nb ? col : col - nb;
            const float xr = (float)sx[f * ld + k], xi = (float)sx[f * ld + nb + k];
            const float yr = (float)sy[f * ld + k], yi = (float)sy[f * ld + nb + k];
            const float ax = sqrtf(xr * xr + xi * xi);
            const float ay = sqrtf(yr * yr + yi * yi);
            if (mag_add ? ax > 0.f : ax >= 1e-8f) {  // (clamped bins and |X| = 0 pass no gradient)
                const float mx = mag_add ? ax + 1e-8f : ax;
                const float my = mag_add ? ay + 1e-8f : fmaxf(ay, 1e-8f);
                const float dl = logf(mx) - logf(my);
                const float dmx = c0 * 2.f * (mx - my) + c2 * (dl > 0.f ? 1.f : (dl < 0.f ? -1.f : 0.f)) / mx;
                g = dmx * (col < nb ? xr : xi) / ax;
            }
        }
        dsx[i] = (T)g;
    }
}

template <typename TI, typename TO>
__global__ void stft_unpad_bwd_kernel(const TI* __restrict__ dxp, TO* __restrict__ dx, int B, int S, int P, size_t per, int accumulate) {
    const size_t n = (size_t)B * S;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / S), s = (int)(i % S);
        const TI* d = dxp + (size_t)b * per;
        float v = (float)d[s + P];
        if (s >= 1 && s <= P) v += (float)d[P - s];
        if (s >= S - 1 - P && s <= S - 2) v += (float)d[P + 2 * (S - 1) - s];
        if (accumulate) v += (float)dx[i];
        dx[i] = (TO)v;
    }
}

static inline int grid_for(size_t n, int cap = 2048) {
    size_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > (size_t)cap) g = cap;
    return (int)g;
}

// power spectrum of a (re | im) spectrum row: p[f][k] = re^2 + im^2, k < nb; columns nb .. ldp-1 of p are written as zeros
// (the mel filterbank GEMM behind it reads rows of ldp channels)
__global__ __launch_bounds__(256) void power_spec_fwd_kernel(const float* __restrict__ spec, size_t frames, int nb, int ld, float* __restrict__ pw, int ldp) {
    const size_t n = frames * (size_t)ldp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / ldp;
        const int k = (int)(i - f * ldp);
        float v = 0.f;
        if (k < nb) {
            const float re = spec[f * ld + k], im = spec[f * ld + nb + k];
            v = re * re + im * im;
        }
        pw[i] = v;
    }
}
// dspec = d/dspec of sum dp * p: (2 re dp | 2 im dp); the padding columns 2*nb .. ld-1 of dspec are zeroed
__global__ __launch_bounds__(256) void power_spec_bwd_kernel(const float* __restrict__ spec, const float* __restrict__ dp, size_t frames, int nb, int ld,
                                                             int ldp, float* __restrict__ dspec) {
    const size_t n = frames * (size_t)ld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t f = i / ld;
        const int c = (int)(i - f * ld);
        float v = 0.f;
        if (c < 2 * nb) {
            const int k = c < nb ? c : c - nb;
            v = 2.f * spec[i] * dp[f * ldp + k];
        }
        dspec[i] = v;
    }
}

extern "C" {

int ga_power_spec_fwd(const float* spec, size_t frames, int nb, int ldspec, float* power, int ldp, hipStream_t s) {
    if (!spec || !power || nb < 1 || ldspec < 2 * nb || ldp < nb) return GA_ERR_ARG;
    const size_t n = frames * (size_t)ldp;
    if (!n) return GA_OK;
    size_t grid = (n + 255) / 256;
    if (grid > 8192) grid = 8192;
    power_spec_fwd_kernel<<<(unsigned)grid, 256, 0, s>>>(spec, frames, nb, ldspec, power, ldp);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_power_spec_bwd(const float* spec, const float* dpower, size_t frames, int nb, int ldspec, int ldp, float* dspec, hipStream_t s) {
    if (!spec || !dpower || !dspec || nb < 1 || ldspec < 2 * nb || ldp < nb) return GA_ERR_ARG;
    const size_t n = frames * (size_t)ldspec;
    if (!n) return GA_OK;
    size_t grid = (n + 255) / 256;
    if (grid > 8192) grid = 8192;
    power_spec_bwd_kernel<<<(unsigned)grid, 256, 0, s>>>(spec, dpower, frames, nb, ldspec, ldp, dspec);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_stft_pad(const void* x, void* xp, int dtype_in, int dtype_out, int B, int S, int n_fft, int hop, int rows, int rep,
                hipStream_t s) {
    if (!x || !xp || B <= 0 || S <= 0 || n_fft <= 0 || hop <= 0 || (rep != 1 && rep != 3)) return GA_ERR_ARG;
    const int P = n_fft / 2;
    if (P >= S || (size_t)rows * hop < (size_t)S + 2 * P) return GA_ERR_ARG;
    const size_t n = (size_t)B * rows * hop;
    if (dtype_in == GA_DTYPE_F32 && dtype_out == GA_DTYPE_F32)
        stft_pad_kernel<float, float><<<grid_for(n), 256, 0, s>>>((const float*)x, (float*)xp, B, S, P, hop, rows, rep);
    else if (dtype_in == GA_DTYPE_F32 && dtype_out == GA_DTYPE_BF16)
        stft_pad_kernel<float, bf16_t><<<grid_for(n), 256, 0, s>>>((const float*)x, (bf16_t*)xp, B, S, P, hop, rows, rep);
    else if (dtype_in == GA_DTYPE_BF16 && dtype_out == GA_DTYPE_BF16)
        stft_pad_kernel<bf16_t, bf16_t><<<grid_for(n), 256, 0, s>>>((const bf16_t*)x, (bf16_t*)xp, B, S, P, hop, rows, rep);
    else if (dtype_in == GA_DTYPE_BF16 && dtype_out == GA_DTYPE_F32)
        stft_pad_kernel<bf16_t, float><<<grid_for(n), 256, 0, s>>>((const bf16_t*)x, (float*)xp, B, S, P, hop, rows, rep);
    else
        return GA_ERR_ARG;
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_stft_loss_fwd_mode(const void* sx, const void* sy, int dtype, size_t frames, int nb, int ld, float* out, int mag_mode, hipStream_t s);
int ga_stft_loss_bwd_mode(const void* sx, const void* sy, int dtype, size_t frames, int nb, int ld, const float* coef, void* dsx,
                          int mag_mode, hipStream_t s);

int ga_stft_loss_fwd(const void* sx, const void* sy, int dtype, size_t frames, int nb, int ld, float* out, hipStream_t s) {
    return ga_stft_loss_fwd_mode(sx, sy, dtype, frames, nb, ld, out, GA_STFT_MAG_CLAMP, s);
}

int ga_stft_loss_bwd(const void* sx, const void* sy, int dtype, size_t frames, int nb, int ld, const float* coef, void* dsx,
                     hipStream_t s) {
    return ga_stft_loss_bwd_mode(sx, sy, dtype, frames, nb, ld, coef, dsx, GA_STFT_MAG_CLAMP, s);
}

int ga_stft_loss_fwd_mode(const void* sx, const void* sy, int dtype, size_t frames, int nb, int ld, float* out, int mag_mode, hipStream_t s) {
    if (!sx || !sy || !out || nb <= 0 || ld < 2 * nb || (mag_mode != GA_STFT_MAG_CLAMP && mag_mode != GA_STFT_MAG_ADD)) return GA_ERR_ARG;
    if (frames == 0) return GA_OK;
    const size_t n = frames * nb;
    if (dtype == GA_DTYPE_BF16)
        stft_loss_fwd_kernel<bf16_t><<<grid_for(n, 1024), 256, 0, s>>>((const bf16_t*)sx, (const bf16_t*)sy, frames, nb, ld, out, mag_mode);
    else if (dtype == GA_DTYPE_F32)
        stft_loss_fwd_kernel<float><<<grid_for(n, 1024), 256, 0, s>>>((const float*)sx, (const float*)sy, frames, nb, ld, out, mag_mode);
    else
        return GA_ERR_ARG;
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_stft_loss_bwd_mode(const void* sx, const void* sy, int dtype, size_t frames, int nb, int ld, const float* coef, void* dsx,
                          int mag_mode, hipStream_t s) {
    if (!sx || !sy || !coef || !dsx || nb <= 0 || ld < 2 * nb || (mag_mode != GA_STFT_MAG_CLAMP && mag_mode != GA_STFT_MAG_ADD)) return GA_ERR_ARG;
    if (frames == 0) return GA_OK;
    const size_t n = frames * ld;
    if (dtype == GA_DTYPE_BF16)
        stft_loss_bwd_kernel<bf16_t><<<grid_for(n), 256, 0, s>>>((const bf16_t*)sx, (const bf16_t*)sy, frames, nb, ld, coef, (bf16_t*)dsx, mag_mode);
    else if (dtype == GA_DTYPE_F32)
        stft_loss_bwd_kernel<float><<<grid_for(n), 256, 0, s>>>((const float*)sx, (const float*)sy, frames, nb, ld, coef, (float*)dsx, mag_mode);
    else
        return GA_ERR_ARG;
    GA_CHECK_LAUNCH();
    return GA_OK;
}

// loss = (1/R) sum_r [ sqrt(s_r0) / (sqrt(s_r1) + 1e-8) + s_r2 * inv_count_r ] from the R x 3 sums of the resolutions, and its
// gradient: ~90 one-element torch kernels (sqrt / where / div / add, forward and backward) as two launches.
// sqrt of a non-positive sum is taken as 0 with zero derivative (the torch expression it replaces did the same).
__global__ void stft_combine_kernel(const float* __restrict__ sums, const float* __restrict__ inv_count, int R,
                                    const float* __restrict__ gout, float* __restrict__ out, float* __restrict__ dsums) {
    const int r = threadIdx.x;
    float term = 0.f;
    if (r < R) {
        const float s0 = sums[3 * r], s1 = sums[3 * r + 1], s2 = sums[3 * r + 2];
        const float r0 = s0 > 0.f ? sqrtf(s0) : 0.f, r1 = s1 > 0.f ? sqrtf(s1) : 0.f;
        const float den = r1 + 1e-8f;
        term = r0 / den + s2 * inv_count[r];
        if (dsums) {
            const float g = gout[0] / (float)R;
            dsums[3 * r] = s0 > 0.f ? g * 0.5f / (r0 * den) : 0.f;
            dsums[3 * r + 1] = s1 > 0.f ? -g * r0 / (den * den) * 0.5f / r1 : 0.f;
            dsums[3 * r + 2] = g * inv_count[r];
        }
    }
    if (out) {
        term = wave_sum(term);
        if (threadIdx.x == 0) out[0] = term / (float)R;
    }
}

int ga_stft_combine(const float* sums, const float* inv_count, int R, const float* gout, float* out, float* dsums, hipStream_t s) {
    if (!sums || !inv_count || R < 1 || R > 64 || (!out && !dsums) || (dsums && !gout)) return GA_ERR_ARG;
    stft_combine_kernel<<<1, 64, 0, s>>>(sums, inv_count, R, gout, out, dsums);
    GA_CHECK_LAUNCH();
    return GA_OK;
}

int ga_stft_unpad_bwd(const void* dxp, void* dx, int dtype_in, int dtype_out, int B, int S, int n_fft, int hop, int rows,
                      int accumulate, hipStream_t s) {
    if (!dxp || !dx || B <= 0 || S <= 0) return GA_ERR_ARG;
    const int P = n_fft / 2;
    if (P >= S || (size_t)rows * hop < (size_t)S + 2 * P) return GA_ERR_ARG;
    const size_t per = (size_t)rows * hop, n = (size_t)B * S;
    if (dtype_in == GA_DTYPE_F32 && dtype_out == GA_DTYPE_F32)
        stft_unpad_bwd_kernel<float, float><<<grid_for(n), 256, 0, s>>>((const float*)dxp, (float*)dx, B, S, P, per, accumulate);
    else if (dtype_in == GA_DTYPE_BF16 && dtype_out == GA_DTYPE_BF16)
        stft_unpad_bwd_kernel<bf16_t, bf16_t><<<grid_for(n), 256, 0, s>>>((const bf16_t*)dxp, (bf16_t*)dx, B, S, P, per, accumulate);
    else if (dtype_in == GA_DTYPE_F32 && dtype_out == GA_DTYPE_BF16)
        stft_unpad_bwd_kernel<float, bf16_t><<<grid_for(n), 256, 0, s>>>((const float*)dxp, (bf16_t*)dx, B, S, P, per, accumulate);
    else if (dtype_in == GA_DTYPE_BF16 && dtype_out == GA_DTYPE_F32)
        stft_unpad_bwd_kernel<bf16_t, float><<<grid_for(n), 256, 0, s>>>((const bf16_t*)dxp, (float*)dx, B, S, P, per, accumulate);
    else
        return GA_ERR_ARG;
    GA_CHECK_LAUNCH();
    return GA_OK;
}

}  // extern "C"
