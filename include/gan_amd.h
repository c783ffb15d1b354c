/* gan_amd C ABI - the drop-in boundary of the MI355X-native lab3 codec-GAN training step.
 *
 * The reference (asajid2-cell/gan) is pure Python on torch ATen ops; it has no FFI of its own.  Each entry
 * point below replaces one ATen call site of the hot path; the citation on
 * each declaration names that call site (paths relative to /root/reference).  The Python host in
 * gan_amd/ binds these symbols with ctypes (gan_amd/_lib.py) and mirrors the reference's module API
 * (lab 3/src/lab3_codec_models.py, lab3_codec_train.py, lab3_codec_bridge.py::decode_embeddings).
 *
 * Conventions
 *   - every pointer is a device pointer in HBM unless its name starts with h_; no torch types
 *   - activations are channels-last "NLC": [B][L][C], C contiguous; dtype GA_DTYPE_F32 or GA_DTYPE_BF16
 *   - parameters, gradients of parameters, statistics and loss sums are fp32
 *   - every function enqueues on `stream`, returns an int status (GA_OK == 0), never throws, never
 *     synchronises and never allocates: callers pass outputs and workspaces
 *   - functions that accumulate with atomics say so; the caller zeroes those outputs first
 */
#ifndef GAN_AMD_H
#define GAN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* ga_stream_t; /* == hipStream_t */

#define GA_OK 0
#define GA_ERR_ARG 1
#define GA_ERR_LAUNCH 2
#define GA_ERR_ALIGN 3
#define GA_ERR_UNSUPPORTED 4 /* the fused / fast variant asked for does not take this shape: use the general entry points */
#define GA_ERR_SCRATCH 5     /* ga_conv_splitk_static: the problem needs more than the library's fixed scratch array */

#define GA_DTYPE_F32 0
#define GA_DTYPE_BF16 1

#define GA_ACT_NONE 0
#define GA_ACT_LRELU 1 /* leaky_relu(0.2): lab3_codec_models.py:128 */
#define GA_ACT_ELU 2   /* nn.ELU(): HF modeling_encodec.py:268,330,340 */
#define GA_ACT_SILU 3  /* F.silu: lab3_codec_models.py:42,46,115 */
#define GA_ACT_TANH 4

#define GA_DACT_NONE 0
#define GA_DACT_LRELU_FROM_OUT 1
#define GA_DACT_ELU_FROM_IN 2
#define GA_DACT_ELU_FROM_OUT 3 /* dact_src holds u = ELU(pre-activation): factor 1 for u > 0, else u + 1 (= exp of the pre-activation) */

#define GA_IMPL_AUTO 0   /* MFMA for bf16 when the shape fills tiles, else direct */
#define GA_IMPL_DIRECT 1 /* vector-ALU fp32 FMA chains (exact fp32 mode, thin layers) */
#define GA_IMPL_MFMA 2   /* bf16 MFMA implicit GEMM (gathered-GEMM kernel when it applies, else the window kernel); error if unsupported */
#define GA_IMPL_MFMA_WINDOW 3 /* force the older LDS-window MFMA kernels (conv_mfma_kernel / wgrad_mfma_kernel).  TEST CROSS-CHECK: they are
                                 compiled in so the tests can compare the product kernels against an independent implementation;
                                 GA_IMPL_AUTO reaches conv_mfma_kernel only for shapes the gathered kernel refuses (ELU on load with
                                 fp32 output, M % 8 != 0) and never reaches wgrad_mfma_kernel */
#define GA_IMPL_MFMA_PP 4     /* force the ping-pong gathered-GEMM kernel (conv_pp_kernel: 256x256 tiles, eight waves); error if
                                 unsupported (needs zero padding, Cin % 64 == 0, tensors < 2^31 bytes).  AUTO / MFMA pick it for K = J * Cin >= 8192 with
                                 >= 192 tiles of 256 x 256 (D blocks.4 forward at the bench batch) */

#define GA_IMPL_MFMA_LDSWIN 5 /* force the LDS-window kernel of the narrow-channel D layers (conv_win_kernel: the input rows of a
                                 tile staged once in LDS, every tap read from there); error if unsupported.  AUTO / MFMA pick it
                                 for Cin 32 / stride 4 (D blocks.1 forward) and Cin 64 / stride 1 (its folded data gradient) */
#define GA_IMPL_MFMA_STREAM 6    /* force the register-streaming kernel of the thin layers (conv_stream_kernel); error if unsupported */
#define GA_IMPL_MFMA_STREAMWIN 7 /* force the streaming-window kernel of the thin multi-tap layers (conv_swin_kernel: a per-wave ring
                                    of input windows in LDS); error if unsupported.  AUTO / MFMA pick it for long launches */

#define GA_MAX_PHASE 8

/* Generalised 1-D convolution in NLC layout ("forward form").
 *   y[b][t*out_stride + phase][m] = epi( sum_{j<J} sum_{c<Cin} w[phase][m][j][c] * pre(x[b][row(t,j)][c]) )
 *   row(t,j) = t*in_stride + j*tap_step + off0 + phase_off0[phase],  t in [t0, t0+Lout)
 *   rows < 0: zero (pad_mode 0) or reflected to -row (pad_mode 1); rows >= Lin: zero
 *   epi(v) = ( act(v + bias[m]) * dact(dact_src) ) + res  (+ old y when accumulate)
 * Instances: Conv1d forward, its data gradient, ConvTranspose1d forward (one phase per output
 * residue) and its data gradient, and the DFT-basis GEMM of the STFT. */
typedef struct ga_conv_params {
    const void* x;        /* [B][Lin][Cin] */
    const void* w;        /* [nphase][M][J][Cin], packed by ga_pack_weights, same dtype as x */
    void* y;              /* [B][LoutTot][M] */
    const float* bias;    /* [M] or NULL */
    const void* res;      /* [B][LoutTot][M] or NULL */
    const void* dact_src; /* [B][LoutTot][M] or NULL */
    int B, Lin, Cin, M;
    int Lout;    /* number of output steps t */
    int LoutTot; /* rows per batch item of y / res / dact_src */
    int J, in_stride, tap_step, off0;
    int out_stride, t0;
    int pad_mode, pre_act, post_act, dact_mode, accumulate;
    int nphase;
    int phase_off0[GA_MAX_PHASE];
    int out_f32; /* bf16 problems only: write y as fp32 (no res / dact_src / accumulate); used by the STFT GEMM and the
                    decoder's waveform output so that loss arithmetic downstream sees unrounded values */
    int ksplit;  /* > 1 (MFMA GEMM kernel only): the K = J*Cin range is divided over ksplit workgroups per output tile,
                    each writing its partial sums to its own slab of ws; ga_conv_finish then adds the slabs and applies
                    the epilogue (bias, act, dact, res, y).  Fills the chip when a deep-K layer has few output tiles
                    (D blocks 3-4). */
    float* ws;   /* split-K workspace [ksplit][B][LoutTot][M] fp32 (need not be initialised); NULL when ksplit <= 1 */
    int y_batch_elems; /* 0: batch items of y / res / dact_src are LoutTot*M elements apart.  Otherwise (MFMA GEMM kernel
                    only) that distance AND the number of valid elements per item: lets the output phases of a strided
                    data gradient / transposed conv be folded into M (rows of nphase*M channels, which is the same
                    memory as nphase rows of M) when the item length is not a multiple of nphase */
    int res_mode;   /* GA_RES_ADD (0): epi adds res.  GA_RES_FM_SIGN (MFMA GEMM kernel only; needs dact_mode, dact_src, res
                    and res_coef_dev): res holds the REAL feature map that pairs with the fake one in dact_src, and epi adds
                    res_coef * res_coef_dev[0] * sign(dact_src - res) * dact(dact_src) -- the feature-matching L1 gradient
                    (lab3_codec_models.py:191-199) formed in the data-gradient epilogue instead of in two passes of its own.
                    GA_RES_ADD_THEN_ACT (MFMA GEMM kernel only, needs res, no split-K): y = post_act(acc + bias + res) -- the
                    activation AFTER the residual sum, so that a SEANet residual unit can hand ELU(unit output) straight to
                    the transposed conv behind it (modeling_encodec.py:340-343 then :268) with no pass of its own */
    float res_coef;
    const float* res_coef_dev; /* device scalar (the loss term's upstream gradient) */
    void* y2;       /* NULL, or (bf16, gathered-GEMM kernel only, no split-K, not with out_f32) a second output of y's shape that
                    receives ELU(y), taken from the rounded value y holds: a layer whose output is consumed both raw (a residual
                    unit's shortcut, modeling_encodec.py:333-336) and through ELU (the unit's first conv, :326-328) hands the
                    ELU over with no pass of its own */
} ga_conv_params;
#define GA_RES_ADD 0
#define GA_RES_FM_SIGN 1
#define GA_RES_ADD_THEN_ACT 2

/* replaces: F.conv1d / F.conv_transpose1d forward and their input gradients
 *   G   lab3_codec_models.py:31-32,75,86 (k3, dilation 1/2/4/8)
 *   D   lab3_codec_models.py:125,136-143 (k15 s1, k41 s4 x4, k3) + leaky_relu :128
 *   dec HF modeling_encodec.py:106 (EncodecConv1d, reflect causal pad :157-176), :193 (EncodecConvTranspose1d,
 *       right trim :206-233), called through lab3_codec_bridge.py:143 */
int ga_conv(const ga_conv_params* h_params, int dtype, int impl, ga_stream_t stream);
/* Kernel selection (bf16, impl AUTO / MFMA), all the same arithmetic:
 *   - M <= 64 and J*Cin <= 192: conv_stream_kernel (operands global -> MFMA registers; fuses pre_act ELU).  It also serves
 *     Cin == 1 (D blocks.0 on the waveform) when the caller passes a ZERO-PADDED copy of x in which every 8-sample window
 *     the problem touches is in range: pad_mode 0, tap_step 1, off0 >= 0 and (Lout-1)*in_stride + off0 + 16*ceil(J/16)
 *     <= Lin (gan_amd/ops.py pads once per waveform); otherwise Cin == 1 runs the vector-ALU kernels of thin.hip;
 *   - otherwise Cin % 8 == 0, M >= 16, no pre_act: conv_gemm_kernel (gathered GEMM through LDS); a layer with pre_act there
 *     is given act(x) materialised by ga_act_fwd;
 *   - M <= 4 (logits / waveform convs): conv_fewout kernels.  fp32 problems run conv_direct (exact fp32 FMA chains). */
/* split-K helpers (no counterpart in the reference: a scheduling detail of the same convolutions).
 * ga_conv_suggest_ksplit: the split ga_conv would profit from for this problem (1 = do not split);
 * ga_conv_finish: y = epi(ws) over the rows the problem covers, after a ga_conv launch with ksplit > 1. */
int ga_conv_suggest_ksplit(const ga_conv_params* h_params, int dtype, int impl);
int ga_conv_finish(const ga_conv_params* h_params, int dtype, ga_stream_t stream);

/* Split-K convolution that allocates nothing: ga_conv with ksplit > 1 followed by ga_conv_finish, with the partial-sum
 * slabs in a scratch array that is part of the library's code object instead of a caller workspace.
 * No counterpart in the reference.  It is for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  It calls no hipMalloc / hipFree / allocating library routine.
 *   - h_params as for ga_conv (bf16 problems the gathered-GEMM kernel takes; inputs and outputs are still the caller's);
 *     ws must be NULL; ksplit <= 1 lets the library choose (at least 2), otherwise it is used (capped at the K steps).
 *   - limit: ksplit * B * LoutTot * M (y_batch_elems per item when set) <= GA_SPLITK_STATIC_FLOATS fp32 words (32 MiB);
 *     larger problems return GA_ERR_SCRATCH and launch nothing: use ga_conv + ga_conv_finish with a workspace.
 *   - every GPU the library runs on holds its own copy of the array until the process ends.  All calls on one GPU share
 *     it: calls that overlap on one GPU (two streams, two threads) race; calls on one stream are ordered and safe.  The
 *     array keeps the previous call's contents; nothing has to be cleared because every word the finishing kernel reads
 *     is written first by the same call.
 *   - the first call on a device looks the array's address up (hipGetSymbolAddress); make that call outside graph capture. */
#define GA_SPLITK_STATIC_FLOATS (8u << 20)
int ga_conv_splitk_static(const ga_conv_params* h_params, int dtype, ga_stream_t stream);

/* which kernel ga_conv would launch for these parameters (for tests and the benchmark's per-kernel accounting) */
#define GA_KERNEL_DIRECT_TILE 0
#define GA_KERNEL_DIRECT_FEWOUT 1
#define GA_KERNEL_DIRECT_CIN1 2
#define GA_KERNEL_MFMA_64x128 3  /* conv_mfma_kernel<1,4,32> */
#define GA_KERNEL_MFMA_128x128 4 /* conv_mfma_kernel<2,2,64> */
#define GA_KERNEL_THIN_FEWOUT_ROWS 5
#define GA_KERNEL_THIN_CIN1_VEC 6
#define GA_KERNEL_GEMM_64x256 7  /* conv_gemm_kernel<2,1,4> */
#define GA_KERNEL_GEMM_128x128 8 /* conv_gemm_kernel<2,2,2>; <1,2,2> (64x128 tiles) for grids of few workgroups */
#define GA_KERNEL_GEMM_32x256 9  /* conv_gemm_kernel<1,1,4> */
#define GA_KERNEL_STREAM 10      /* conv_stream_kernel: M <= 64, K <= 192, operands straight from global memory */
#define GA_KERNEL_GEMM_PP 11     /* conv_pp_kernel: 256x256 tiles, wave tile 128x64, ping-pong schedule (deep K, M >= 192) */
#define GA_KERNEL_CONV_WIN 12    /* conv_win_kernel: input window in LDS, weights streamed (narrow-channel D layers) */
#define GA_KERNEL_STREAM_WIN 13  /* conv_swin_kernel: M <= 64, 2-4 taps, weights in registers, per-wave ring of input windows in LDS */
#define GA_KERNEL_RESUNIT 14     /* resunit_fwd_kernel (ga_resunit_fwd below; never returned by ga_conv_kernel_id) */
#define GA_KERNEL_RESUNIT_BWD 15 /* resunit_bwd_kernel (ga_resunit_bwd) */
#define GA_KERNEL_DUAL_K1 17     /* conv_dual_k1_kernel (ga_conv_dual_k1) */
#define GA_KERNEL_COOP 16        /* conv_coop_kernel: 256 output channels, K = 2 x 128, many rows: weights in the registers of an
                                    eight-wave workgroup, shared input window (the decoder's transposed conv 128 -> 4 x 64) */
int ga_conv_kernel_id(const ga_conv_params* h_params, int dtype, int impl);
/* Which instantiation conv_gemm_kernel<MI, WM, WN, NBUF, BUF> ga_conv would run the problem on with GA_IMPL_AUTO / GA_IMPL_MFMA
 * (bf16), as 0xMWNB | BUF << 16: tile rows per wave (MI x 32), waves along m / n, LDS buffers, buffer-load staging.  0: another
 * kernel takes it.  The instantiations are different kernels to a profiler; bench.py's per-kernel accounting asks this. */
int ga_conv_gemm_variant(const ga_conv_params* h_params);
/* LDS budget per workgroup, in bytes, for the gathered-GEMM launches ga_conv makes from now on (0: none; returns the previous
 * value).  For callers that run small convolutions on one stream BESIDE a persistent LSTM launch (ga_lstm_fwd with a workspace: one
 * workgroup holding 111-137 KiB of LDS on every CU) on another: with 24 KiB <= budget < 32 KiB the launches take the 64 x 128,
 * single-buffer configuration (24 KiB) and share the CUs with the LSTM's workgroups instead of queueing behind them.  Results do
 * not depend on it.  Process-wide and not synchronised: set it from the thread that launches (the reference has no counterpart:
 * lab3_codec_train.py:404-470 runs everything on one stream). */
int ga_set_gemm_lds_budget(int bytes);

/* One SEANet residual unit of the frozen decoder, forward, as ONE pass over its input (bf16):
 *     y = shortcut(x) + conv_k1(ELU(conv_k3(ELU(x))))          C -> C/2 -> C channels, causal k3 with reflect-left padding
 * replaces: EncodecResnetBlock.forward, HF modeling_encodec.py:320-343 (shortcut conv k1 :333-336, block = ELU, conv k3,
 *   ELU, conv k1 :326-332), reached through FrozenEncodec.decode_embeddings (lab3_codec_bridge.py:143) -- the three ga_conv
 *   launches of the unit at the thin end of the decoder (32 and 64 channels at 60000-120000 steps per item), which each read
 *   and write the unit's largest tensors; here x is read once and y written once.
 *   x, y [B][L][C] channels-last; weights in ga_conv's packed forward layout (GA_PACK_CONV_FWD): w_sc [C][C], w1 [C/2][3][C],
 *   w3 [C][C/2]; biases fp32.  t_act (optional, [B][L][C/2]) receives ELU(conv_k3(ELU(x)) + b1), which the data gradient of
 *   conv_k1 needs (GA_DACT_ELU_FROM_OUT).  The shortcut stays in fp32 until the single rounding of y (the three-launch form
 *   rounds it to bf16 first).  C in {32, 64, 128}, L >= 3, 16-byte aligned pointers; anything else returns GA_ERR_UNSUPPORTED.
 *   (C = 128: the weights do not fit one wave's registers; a workgroup of four waves shares the window and splits the channels.) */
typedef struct ga_resunit_params {
    const void* x;
    void* y;
    void* t_act;
    const void* w_sc;
    const float* b_sc;
    const void* w1;
    const float* b1;
    const void* w3;
    const float* b3;
    int B, L, C;
    /* tail form (C == 32, L >= 13, no gradient wanted): wav != NULL makes the call ALSO run the decoder's closing layers on
     * the unit's output -- wav[b][n] = b_last + sum_{j<7, c<32} w_last[j][c] * ELU(y)[b][n - 6 + j][c], reflect-left padded
     * (modeling_encodec.py: the decoder's last nn.ELU + EncodecConv1d(32, 1, 7)) -- and write only that: [B][L] fp32; y and
     * t_act must then be NULL.  w_last is the layer's GA_PACK_CONV_FWD pack [1][7][32]; the result equals
     * ga_resunit_fwd followed by ga_conv (pre_act ELU, out_f32) to the last bit. */
    const void* w_last;
    const float* b_last;
    float* wav;
    int post_elu; /* C == 128 only: y receives ELU(unit output) (the transposed conv behind the unit reads it through ELU only:
                     what GA_RES_ADD_THEN_ACT does in the three-launch form) */
} ga_resunit_params;
int ga_resunit_fwd(const ga_resunit_params* h_params, int dtype, ga_stream_t stream);
/* The same unit's input gradient as one pass (autograd of the call above, reached from loss_g.backward(),
 * lab3_codec_train.py:588):
 *     gt = (W3^T gy) * ELU'(hidden)          ELU' taken from t_act = ELU(hidden), as GA_DACT_ELU_FROM_OUT
 *     gx = W_sc^T gy + (conv_k3^T gt, with the rows of the reflected left pad folded back) * ELU'(x)
 * replaces the unit's three data-gradient ga_conv launches and the pad fix-up launch.  Weights in ga_conv's data-gradient
 * layout (GA_PACK_CONV_DGRAD_S1): w_sc_d [C][1][C], w1_d [C][3][C/2], w3_d [C/2][1][C].  C in {32, 64} only (the 128-channel unit's
 * backward returns GA_ERR_UNSUPPORTED: its data gradient runs as three ga_conv launches); other limits as ga_resunit_fwd. */
typedef struct ga_resunit_bwd_params {
    const void* gy;
    const void* t_act;
    const void* x;
    void* gx;
    const void* w_sc_d;
    const void* w1_d;
    const void* w3_d;
    int B, L, C;
} ga_resunit_bwd_params;
int ga_resunit_bwd(const ga_resunit_bwd_params* h_params, int dtype, ga_stream_t stream);

/* The closing sum of the 256-channel residual unit in one pass (bf16): y = [ELU]( W_a xa + W_b xb + ba + bb ), row-wise (both convs
 * are k1): xa [rows][256] = the unit's input with wa = the shortcut's pack [256][256], xb [rows][128] = ELU(hidden) with wb = the
 * block's last conv's pack [256][128]; y [rows][256].
 * replaces: shortcut(x) + block[-1](...) of EncodecResnetBlock.forward, HF modeling_encodec.py:333-343 -- two ga_conv launches, the
 * first one's output written and read back.  Ca = 256, Cb = 128, M = 256 only (GA_ERR_UNSUPPORTED otherwise); no padding, no
 * item boundaries: rows = B * L. */
typedef struct ga_dual_k1_params {
    const void* xa;
    const void* wa;
    const float* ba;
    const void* xb;
    const void* wb;
    const float* bb;
    void* y;
    long rows;
    int Ca, Cb, M;
    int post_elu;
} ga_dual_k1_params;
int ga_conv_dual_k1(const ga_dual_k1_params* h_params, int dtype, ga_stream_t stream);

/* weight gradient: dw[m][j][c] += sum_{b,t} g[b][t*g_stride + g_phase][m] * x[b][t*in_stride + j*tap_step + off0][c]
 * fp32 atomics into dw (caller zeroes); dw is in the packed forward layout [M][J][Cin].
 * replaces: the weight-gradient half of conv1d backward for G and D (autograd of the call sites above,
 * reached from loss_d.backward() / loss_g.backward(), lab3_codec_train.py:445,588) */
typedef struct ga_wgrad_params {
    const void* g; /* [B][LgTot][M] */
    const void* x; /* [B][Lin][Cin] */
    float* dw;     /* [M][J][Cin] */
    float* dbias;  /* [M] or NULL: when given, dbias[m] += sum_{b,t} g[b][t][m] (bias gradient, atomics) */
    int B, Lin, Cin, M, Lout, J, in_stride, tap_step, off0;
    int LgTot, g_stride, g_phase;
    int rows_per_split; /* set by the library */
    float* ws;          /* optional workspace of ga_wgrad_workspace_elems() floats (need not be initialised): the slices of
                           the reduction range then write their partial tiles as plain stores to slabs that one more
                           kernel adds into dw, instead of fp32 atomics on dw; NULL = atomics */
} ga_wgrad_params;
/* bf16 kernels: wgrad_gemm_kernel (Cin % 8 == 0); Cin == 1 with a zero-padded x (in_stride 1, tap_step 1, off0 >= 0, J <= 15
 * with dbias, every window in range as for ga_conv): wgrad_cin1_mfma_kernel, which also fills dbias; M == 1: wgrad_m1 kernels. */
int ga_wgrad(const ga_wgrad_params* h_params, int dtype, int impl, ga_stream_t stream);
/* n weight-gradient problems (h_items: a HOST array) in one launch per 20 of them, each with the semantics of ga_wgrad (dw and dbias
 * are added to: the caller zeroes them).  For the generator's many small layers (CodecLatentTranslator, lab3_codec_models.py:50-119:
 * forty Conv1d(256, 256, 3) over 375 steps), whose loss_g.backward() (lab3_codec_train.py:586) weight gradients are 40-50 us launches
 * one by one.  bf16, M > 64, ws == NULL, tensors under 2 GiB only: GA_ERR_UNSUPPORTED (nothing launched) if any item does not
 * qualify -- issue ga_wgrad per item then. */
int ga_wgrad_multi(const ga_wgrad_params* h_items, int n, int dtype, ga_stream_t stream);
/* floats of workspace ga_wgrad can use for this problem (0: the kernel it would pick has no slab path) */
size_t ga_wgrad_workspace_elems(const ga_wgrad_params* h_params, int dtype, int impl);

/* weight norm (torch.nn.utils.weight_norm, dim=0: lab3_codec_models.py:125,143; parametrizations.weight_norm in the
 * HF decoder, modeling_encodec.py:109-111,196-201): scale[r] = g[r]/||v[r,:]||, norm[r] = ||v[r,:]|| */
int ga_wn_scale(const float* v, const float* g, int rows, int n, float* scale, float* norm, ga_stream_t stream);

#define GA_PACK_CONV_FWD 0           /* src [Co][Ci][K] -> [Co][K][Ci] */
#define GA_PACK_CONV_DGRAD_S1 1      /* src [Co][Ci][K] -> [Ci][K][Co], taps flipped (stride-1 data gradient) */
#define GA_PACK_CONV_DGRAD_STRIDED 2 /* src [Co][Ci][K] -> [S][Ci][ceil(K/S)][Co], one phase per input residue */
#define GA_PACK_CONVT_FWD 3          /* src [Ci][Co][2S] -> [S][Co][2][Ci], one phase per output residue */
#define GA_PACK_CONVT_DGRAD 4        /* src [Ci][Co][K] -> [Ci][K][Co] */
size_t ga_packed_elems(int mode, int D0, int D1, int K, int S);
/* dst[pack(s0,s1,k)] = src[s0][s1][k] * scale[s0]  (scale NULL => 1); dst dtype f32 or bf16.
 * This is where "w = g * v / ||v||" is applied (fused weight-norm), once per step per layer. */
int ga_pack_weights(const float* src, const float* scale, void* dst, int dtype, int mode, int D0, int D1, int K, int S, int P,
                    ga_stream_t stream);
/* Both packs (and the weight-norm scale) of up to GA_PACK_MULTI_MAX layers in two launches: what ga_wn_scale +
 * 2 x ga_pack_weights do per layer, for every conv of a module at once (the per-iteration repack of G's and D's weights,
 * lab3_codec_models.py:125,143 applied once per optimizer step).  `items` is a HOST array.  Layers with g == NULL are packed
 * unscaled.  A wp_d written in mode GA_PACK_CONV_DGRAD_STRIDED must be zero-filled by the caller (its padding taps are
 * not written). */
#define GA_PACK_MULTI_MAX 24
typedef struct ga_pack_item {
    const float* v;  /* [D0][D1][K] fp32 master weights */
    const float* g;  /* weight-norm gain [D0], or NULL */
    float* scale;    /* out [D0]: g/||v||   (g != NULL) */
    float* norm;     /* out [D0]: ||v||     (g != NULL) */
    void* wp_f;      /* out: pack in mode_f */
    void* wp_d;      /* out: pack in mode_d */
    int D0, D1, K, S, P, mode_f, mode_d, reserved;
} ga_pack_item;
int ga_pack_multi(const ga_pack_item* items, int n, int dtype, ga_stream_t stream);
/* The weight gradients of up to GA_PACK_MULTI_MAX Conv1d layers, from the packed GA_PACK_CONV_FWD layout [D0][K][D1]
 * (what ga_wgrad writes) to their final place, in one launch: ga_unpack_grad + ga_wn_bwd per layer, and the copy into the
 * optimizer's gradient buffer, fused (torch weight_norm backward, lab3_codec_models.py:125,143; opt step,
 * lab3_codec_train.py:444-452,585-590).  g == NULL: plain layer, dv = unpacked gradient.  accumulate bit 0: add to dv,
 * bit 1: add to dg.  D1*K <= GA_WFINISH_MAX_ROW (one weight row is staged in LDS).  `items` is a HOST array. */
#define GA_WFINISH_MAX_ROW 15360
typedef struct ga_wfinish_item {
    const float* gpacked; /* [D0][K][D1] fp32 */
    const float* v;       /* [D0][D1][K]   (g != NULL) */
    const float* g;       /* [D0] or NULL */
    const float* norm;    /* [D0] ||v||    (g != NULL) */
    float* dv;            /* out [D0][D1][K] */
    float* dg;            /* out [D0]      (g != NULL) */
    int D0, D1, K, accumulate;
} ga_wfinish_item;
int ga_wgrad_finish_multi(const ga_wfinish_item* items, int n, ga_stream_t stream);
/* dw[s0][s1][k] (+)= packed[pack(s0,s1,k)] : brings a packed-layout gradient back to the parameter layout */
int ga_unpack_grad(const float* packed, float* dw, int mode, int D0, int D1, int K, int S, int P, int accumulate,
                   ga_stream_t stream);
/* backward of w = g*v/||v|| given dw (parameter layout [rows][n]) -> dv, dg */
int ga_wn_bwd(const float* dw, const float* v, const float* g, const float* norm, int rows, int n, float* dv, float* dg,
              int accumulate, ga_stream_t stream);

/* GroupNorm(8) -> FiLM -> SiLU (+ residual): lab3_codec_models.py:20-25,33-47,85,114-115
 *   y = act(((x-mean)*rstd*gamma+beta)*(1+a[b,c]) + bb[b,c]) + res;  ab = [B][2C] = (a | bb) or NULL (no FiLM) */
int ga_gn_stats(const void* x, int dtype, int B, int T, int C, int G, float eps, float* mean, float* rstd, ga_stream_t stream);
/* ga_gn_stats + ga_gn_film_act_fwd in one launch: GroupNorm(8) -> FiLM -> SiLU (+ residual) of lab3_codec_models.py:20-25,38-47,
 * 114-115 and lab3_codec_judge.py:98-108 (a (b, group) slab is normalised from the registers its statistics were
 * taken from); mean / rstd are still written for the backward pass.  GA_ERR_UNSUPPORTED when C/G is not a multiple of the
 * 16-byte vector width or a pointer is not 16-byte aligned: call the two entry points instead. */
int ga_gn_film_act_fused(const void* x, const float* gamma, const float* beta, const float* ab, const void* res, void* y,
                         float* mean, float* rstd, int dtype, int B, int T, int C, int G, float eps, int act, ga_stream_t stream);
int ga_gn_film_act_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                       const float* ab, const void* res, void* y, int dtype, int B, int T, int C, int G, int act,
                       ga_stream_t stream);
/* dgamma, dbeta [C] and dab [B][2C] are accumulated with atomics (caller zeroes) */
int ga_gn_film_act_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, const float* ab, void* dx, float* dgamma, float* dbeta, float* dab, int dtype,
                       int B, int T, int C, int G, int act, ga_stream_t stream);

/* q_hat = q_src + residual_scale * tanh(raw): lab3_codec_models.py:119 */
int ga_tanh_residual_fwd(const void* raw, const void* q, void* y, int dtype, size_t n, float rscale, ga_stream_t stream);
int ga_tanh_residual_bwd(const void* dy, const void* raw, void* draw, int dtype, size_t n, float rscale, ga_stream_t stream);

/* nn.AvgPool1d(4, 2, 1) on [B][L]: lab3_codec_models.py:159,166 */
int ga_avgpool1d_fwd(const void* x, void* y, int dtype, int B, int L, int Lo, ga_stream_t stream);
int ga_avgpool1d_bwd(const void* dy, void* dx, int dtype, int B, int L, int Lo, ga_stream_t stream);

/* ga = gy * act'(src): backward of leaky_relu (lab3_codec_models.py:128) */
int ga_act_bwd(const void* gy, const void* src, void* ga, int dtype, size_t n, int mode, ga_stream_t stream);
/* the same on [rows][M] plus the conv bias gradient dbias[m] += sum_rows ga[row][m] in one pass (caller zeroes dbias);
 * M / (16 bytes of elements) must divide 256 */
int ga_act_bwd_bias(const void* gy, const void* src, void* ga, int dtype, size_t rows, int M, int mode, float* dbias,
                    ga_stream_t stream);
/* y = act(x), elementwise (GA_ACT_*): the ELU in front of every SEANet decoder conv (modeling_encodec.py:265-275,
 * 330-343), materialised once so that the conv itself can stream its operand straight into LDS */
int ga_act_fwd(const void* x, void* y, int dtype, size_t n, int act, ga_stream_t stream);
/* nn.Conv2d / nn.ConvTranspose2d of the mel track (lab 3/src/lab3_models.py:22-27,88,95,101,114,156-164) run as 1-D implicit GEMMs along W
 * after the KH kernel rows of a channels-last image [B][H][W][C] have been gathered into the channel axis:
 *   y[b][oh][w][i*C + c] = x[b][S*oh + i - P][w][c], zero outside the image and in the padding channels KH*C .. KC-1   (y: [B][OH][W][KC])
 * ga_rowgather_bwd is the adjoint (gx[b][h][w][c] = sum of the gy entries that read it).  16-byte pieces when C and KC allow. */
int ga_rowgather_fwd(const void* x, void* y, int dtype, int B, int H, int W, int C, int KH, int S, int P, int OH, int KC, ga_stream_t stream);
int ga_rowgather_bwd(const void* gy, void* gx, int dtype, int B, int H, int W, int C, int KH, int S, int P, int OH, int KC, ga_stream_t stream);
/* nn.InstanceNorm2d(affine) / nn.BatchNorm2d followed by LeakyReLU(0.2) in the mel track's decoder up-blocks and discriminator
 * (lab 3/src/lab3_models.py:80-83 `norm2d`, :85-105 up-blocks, :159-165), on channels-last memory x, y: [B][HW][C], C a multiple of the
 * 16-byte piece and at most 256 pieces (ga_cnorm_supported says).  groups = B: statistics per item and channel (instance norm); groups = 1:
 * per channel over B*HW rows (batch norm).  Forward: ga_cnorm_stats adds (sum(x - p), sum((x - p)^2)) about the pivot p = the group's first
 * row into the per-item sums[B][C][2] (DOUBLE: the atomics' order then does not show in fp32; caller zeroes); ga_cnorm_finalize adds the items of a group and turns them into mean / rstd [groups][C] (DOUBLE as well: the input gradient must sum to zero, and fp32 statistics leave a same-signed residual) (biased variance, eps inside the
 * root) and, when running_mean / running_var are given (groups = 1), applies torch's momentum update with the unbiased variance;
 * ga_cnorm_apply writes y = act((x - mean) * rstd * gamma + beta), act GA_ACT_NONE or GA_ACT_LRELU, gamma / beta may be NULL (1 / 0).
 * Backward: ga_cnorm_bwd_stats adds (sum g, sum g * xhat), g = gy * act'(y), into the per-item gsums[B][C][2] (caller zeroes; summed over B
 * these ARE the beta and gamma gradients); ga_cnorm_bwd_apply, given gsums[groups][C][2] (for groups = 1 the sum over B), writes gx = gamma * rstd * (g - (sum g + xhat * sum g xhat) / N), or
 * gamma * rstd * g when `frozen` (eval-mode batch norm: mean / rstd from the running statistics). */
int ga_cnorm_supported(int dtype, int B, long HW, int C, int groups);
int ga_cnorm_stats(const void* x, int dtype, int B, long HW, int C, int groups, double* sums, ga_stream_t stream);
int ga_cnorm_finalize(const void* x, int dtype, int B, long HW, int C, int groups, const double* sums, float eps, double* mean, double* rstd,
                      float* running_mean, float* running_var, float momentum, ga_stream_t stream);
int ga_cnorm_apply(const void* x, void* y, int dtype, int B, long HW, int C, int groups, const double* mean, const double* rstd,
                   const float* gamma, const float* beta, int act, ga_stream_t stream);
int ga_cnorm_bwd_stats(const void* gy, const void* x, const void* y, int dtype, int B, long HW, int C, int groups, const double* mean,
                       const double* rstd, int act, double* gsums, ga_stream_t stream);
int ga_cnorm_bwd_apply(const void* gy, const void* x, const void* y, void* gx, int dtype, int B, long HW, int C, int groups, const double* mean,
                       const double* rstd, const float* gamma, const double* gsums, int act, int frozen, ga_stream_t stream);
/* out[m] += sum_rows g[row][m]: bias gradient of conv1d (atomics; caller zeroes) */
int ga_colsum(const void* g, int dtype, size_t rows, int M, float* out, ga_stream_t stream);

/* loss reductions: out[0] += scale * sum f(a,b) (atomics; caller zeroes)
 *   hinge_d_loss / hinge_g_loss lab3_codec_models.py:171-176, F.l1_loss in feature_matching_loss :185 */
#define GA_RED_ABSDIFF 0
#define GA_RED_RELU1M 1
#define GA_RED_RELU1P 2
#define GA_RED_SUM 3
#define GA_RED_SQ 4
int ga_reduce(const void* a, const void* b, int dtype, size_t n, int op, float scale, float* out, ga_stream_t stream);
/* `count` (<= 8) such sums, each with its own scale, added into ONE scalar by one launch: multiscale_feature_matching_loss
 * (lab3_codec_models.py:179-186,208-215) is five l1 means per discriminator scale.  n = 0 entries are allowed. */
int ga_reduce_multi(const void* const* a, const void* const* b, const size_t* n, const float* scale, int count, int dtype, int op,
                    float* out, ga_stream_t stream);
int ga_reduce_grad(const void* a, const void* b, int dtype, size_t n, int op, const float* coef, float scale, void* ga,
                   int accumulate, ga_stream_t stream);

/* latent L1, first-difference L1 and per-sample |delta| sums in one pass: lab3_codec_train.py:464-465,552
 *   out[0] = sum|d|, out[1] = sum_t |d[t+1]-d[t]|, out[2+b] = sum over sample b of |d|, d = q_hat - q_src (atomics) */
int ga_latent_loss_fwd(const void* qh, const void* qs, int dtype, int B, int T, int C, float* out, ga_stream_t stream);
int ga_latent_loss_bwd(const void* qh, const void* qs, int dtype, int B, int T, int C, const float* gout, void* dqh,
                       ga_stream_t stream);

/* MR-STFT loss (lab3_codec_train.py:148-176).  The DFT itself is ga_conv on the framed signal with the
 * windowed cos/sin basis as weights; these are the kernels either side of it.
 *   ga_stft_pad:      x [B][S] -> xp [B][rows][width] row-major view of the reflect-padded (n_fft/2 each side) signal,
 *                     zero tail; width = hop*rep, rep=1 plain, rep=3 writes (hi, lo, hi) bf16 split of fp32 x
 *   ga_stft_loss_fwd: spec [B][F][2*nb] (re | im) of prediction and target -> out[0] += sum (my-mx)^2,
 *                     out[1] += sum my^2, out[2] += sum |log mx - log my|, magnitudes clamped at 1e-8
 *   ga_stft_loss_bwd: dspec_x = d/dspec_x of c[0]*out0 + c[2]*out2 (target side has no gradient)
 *   ga_stft_unpad_bwd: folds the gradient of the padded signal back onto x */
int ga_stft_pad(const void* x, void* xp, int dtype_in, int dtype_out, int B, int S, int n_fft, int hop, int rows, int rep,
                ga_stream_t stream);
int ga_stft_loss_fwd(const void* spec_x, const void* spec_y, int dtype, size_t frames_total, int nb, int ldspec, float* out,
                     ga_stream_t stream);
int ga_stft_loss_bwd(const void* spec_x, const void* spec_y, int dtype, size_t frames_total, int nb, int ldspec,
                     const float* coef, void* dspec_x, ga_stream_t stream);
/* The same two kernels with the magnitude floor as a parameter: GA_STFT_MAG_CLAMP = |X|.clamp_min(1e-8) (the waveform loss
 * above), GA_STFT_MAG_ADD = |X| + 1e-8 (the mel-trajectory sibling multi_resolution_stft_loss, lab 3/src/lab3_losses.py:16-71) */
#define GA_STFT_MAG_CLAMP 0
#define GA_STFT_MAG_ADD 1
int ga_stft_loss_fwd_mode(const void* spec_x, const void* spec_y, int dtype, size_t frames_total, int nb, int ldspec, float* out,
                          int mag_mode, ga_stream_t stream);
int ga_stft_loss_bwd_mode(const void* spec_x, const void* spec_y, int dtype, size_t frames_total, int nb, int ldspec,
                          const float* coef, void* dspec_x, int mag_mode, ga_stream_t stream);
/* MR-STFT scalar tail (lab3_codec_train.py:161-176): out[0] = (1/R) sum_r [ sqrt(sums[r][0]) / (sqrt(sums[r][1]) + 1e-8)
 * + sums[r][2] * inv_count[r] ] (out != NULL) and/or dsums = gout[0] * d out / d sums (dsums != NULL).  R <= 64. */
int ga_stft_combine(const float* sums, const float* inv_count, int R, const float* gout, float* out, float* dsums,
                    ga_stream_t stream);
int ga_stft_unpad_bwd(const void* dxp, void* dx, int dtype_in, int dtype_out, int B, int S, int n_fft, int hop, int rows,
                      int accumulate, ga_stream_t stream);

/* Power spectrum either side of the mel filterbank GEMM of the Lab1 waveform -> log-mel adapter (lab3_codec_train.py:185-194,
 * 208: torchaudio MelSpectrogram(power=2) = |STFT|^2 @ filterbank): power[f][k] = re^2 + im^2 from a (re | im) spectrum row of
 * ldspec floats, written as rows of ldp floats (columns >= nb zero: the filterbank GEMM reads whole rows); and its gradient
 * dspec = (2 re dpower | 2 im dpower), padding columns zero. */
int ga_power_spec_fwd(const float* spec, size_t frames_total, int nb, int ldspec, float* power, int ldp, ga_stream_t stream);
int ga_power_spec_bwd(const float* spec, const float* dpower, size_t frames_total, int nb, int ldspec, int ldp, float* dspec,
                      ga_stream_t stream);

/* torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW.step (lab3_codec_train.py:302-311,446-448,589-591) over flat
 * fp32 buffers: out[0] += sum g^2 (atomics); then p,m,v updated in one pass, clip coefficient
 * min(1, max_norm/(sqrt(sqnorm[0])*grad_scale + 1e-6)) folded in; grad_scale = 1/world_size under data parallel. */
int ga_grad_sqnorm(const float* g, size_t n, float* out, ga_stream_t stream);
int ga_adamw_fused(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float max_norm, float grad_scale, const float* sqnorm,
                   int write_clipped_grad, ga_stream_t stream);
/* The same launch with the step number (>= 1) read from device memory by the kernel: for an iteration recorded as a HIP graph, whose
 * replays repeat every launch with the recorded arguments (the caller increments *step_dev on the stream before this call). */
int ga_adamw_fused_devstep(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                           float weight_decay, const int* step_dev, float max_norm, float grad_scale, const float* sqnorm,
                           int write_clipped_grad, ga_stream_t stream);

/* Two-layer LSTM + input skip of the frozen decoder (HF EncodecLSTM, modeling_encodec.py:236-249; reached through
 * lab3_codec_bridge.py:143).  All [B][T][*] buffers are channels-last.  Gate order i,f,g,o (torch nn.LSTM).
 * forward: T+1 dependent step launches covering both layers (layer 2 one step behind layer 1).
 *   w1 = weight_hh_l0 [4H][H]; w2 = [weight_ih_l1 | weight_hh_l1] [4H][2H]; bias2 = bias_ih_l1 + bias_hh_l1
 *   xg1 = x @ weight_ih_l0^T + bias_ih_l0 + bias_hh_l0 for all steps (one ga_conv with k=1)
 *   y = h2 + x_in.  c1, c2 are the running cell states (caller zeroes).  gatesN and csN (all or none) save i,f,g,o and c
 *   for the backward pass.  w_dtype/act_dtype: both bf16 or both f32. */
typedef struct ga_lstm_params {
    const void* w1; const void* w2; const float* bias2;
    const void* xg1;  /* [B][T][4H] act dtype */
    const void* x_in; /* [B][T][H] act dtype */
    void* y;          /* [B][T][H] act dtype */
    void* h1; void* h2; /* [B][T][H] w dtype: hidden-state history of the two layers */
    float* c1; float* c2; /* [B][H] */
    float* gates1; float* gates2; /* [B][T][H][4] (i, f, g, o of a unit side by side: one 16-byte access) or NULL */
    float* cs1; float* cs2;       /* [B][T][H] or NULL */
    int B, T, H, act_dtype, w_dtype;
    int B_save; /* gates / cell states are recorded for batch items [0, B_save) only; <= 0 means all B */
    void* ws;        /* optional (bf16, H = 512): workspace of ga_lstm_persist_workspace_bytes() bytes, 256-byte aligned.  With
                        it the pass runs as ONE persistent launch (lstm_persist.hip: recurrent weights resident in LDS, hidden
                        state exchanged between workgroups through ws with write-through stores and a counter) instead of T+1
                        step launches; h1 / h2 / c1 / c2 are then not written.  The first word of ws is a STICKY error flag
                        a kernel raises when a bounded wait ran out (results of that launch invalid): the caller zeroes ws once
                        when it allocates it, no launch ever clears that word (a launch clears bytes 64 .. 1023 only), so the
                        flag covers every launch since the caller last read and cleared it; calls sharing one ws must not overlap. */
    size_t ws_bytes;
} ga_lstm_params;
int ga_lstm_fwd(const ga_lstm_params* h_params, ga_stream_t stream);
size_t ga_lstm_persist_workspace_bytes(int B);
/* backward w.r.t. the input only (weights are frozen): fills dg1 = d(loss)/d(gate pre-activations of layer 1) for all
 * steps; dx = dg1 @ weight_ih_l0 + dy is one more ga_conv (k=1, residual epilogue).
 *   w1t = weight_hh_l0^T [H][4H]; w2t = [weight_ih_l1^T ; weight_hh_l1^T] [2H][4H]; dc1, dc2 [B][H] zeroed by the caller */
typedef struct ga_lstm_bwd_params {
    const void* w1t; const void* w2t;
    const void* dy; /* [B][T][H] act dtype */
    const float* gates1; const float* gates2; const float* cs1; const float* cs2;
    void* dg1; void* dg2; /* [B][T][4H] w dtype */
    float* dc1; float* dc2;
    int B, T, H, act_dtype, w_dtype;
    void* ws;        /* optional, as in ga_lstm_params: one persistent launch; dg2 / dc1 / dc2 are then not written */
    size_t ws_bytes;
} ga_lstm_bwd_params;
int ga_lstm_bwd(const ga_lstm_bwd_params* h_params, ga_stream_t stream);

/* library identification: returns the gfx target the code objects were built for ("gfx950") */
const char* ga_build_target(void);

#ifdef __cplusplus
}
#endif
#endif /* GAN_AMD_H */
