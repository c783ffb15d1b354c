// Two-layer LSTM of the frozen EnCodec decoder (HF EncodecLSTM, modeling_encodec.py:236-249) as ONE persistent launch per
// pass instead of a chain of T+1 step launches (lstm.hip).  bf16 weights / activations, H = 512.
//
// Why: the recurrence is latency-bound.  A step launch costs a kernel boundary (~2.7 us) plus a fresh fetch of the 6 MB of
// recurrent weights (they do not survive the boundary in L2: 12.7 MB fetched per step launch, profiles/r01*).  Here every
// workgroup keeps ITS slice of the weights in LDS for the whole sequence and only the hidden state crosses workgroups.
//
// Decomposition: the batch is cut into `groups` (<= 4) groups of NB = 16*NBT items; a group is served by 64 workgroups (one
// per CU, 256 threads), workgroup j owning hidden units 8j .. 8j+7 of BOTH layers: 32 gate rows x 512 of W_hh_l0 (32 KiB)
// and 32 gate rows x 1024 of [W_ih_l1 | W_hh_l1] (64 KiB) in LDS.  Step s computes layer 1 at t = s from h1[s-1] and
// layer 2 at t = s-1 from [h1[s-1] | h2[s-2]] (layer 2 runs one step behind, as in lstm.hip), so one exchange per step
// carries everything: each workgroup publishes its 8 units of h1[s] and h2[s-1] for its group's items (16 bytes per item
// and layer) and then needs the other 63 workgroups' pieces.
//
// Hand-off (MI355X_MICROARCH.md, "Valid forms", first row of the sc1 table): pieces are written with 16-byte sc1
// (write-through) stores, every storing wave drains them (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane
// adds 1 to the group's monotonic counter (agent-scope atomic); consumers poll that counter with an sc1 load from one
// lane, meet at a workgroup barrier and then read every piece with 16-byte sc1 loads (never a plain load of exchanged
// bytes).  Two parity buffers: a workgroup overwrites parity s&1 at step s+2 only after the counter has shown that every
// workgroup finished step s+1, i.e. finished reading it.  Every spin is bounded: on a timeout the workgroup raises the
// error word and runs through without waiting (results are then wrong, the error word says so, nothing hangs).
// All 64*groups workgroups must be resident at once: the host launches at most one workgroup per CU.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define LP_H 512
#define LP_WG_PER_GROUP 64
#define LP_W1_STRIDE 1040  // bytes per LDS row of W1 (512 bf16 + 16: the 16 rows of an MFMA tile spread over the banks)
#define LP_W2_STRIDE 2064
#define LP_SPIN_LIMIT (1 << 22)
#define LP_WS_HEADER 1024  // bytes: STICKY error word at 0 (never cleared by a launch: only the host clears it after reading it),
                           // this launch's abort word at 64, counter of group g at 128 + 128*g; a launch clears bytes 64 .. 1023
#define LP_WS_STICKY 64    // bytes in front of the per-launch part

__device__ __forceinline__ float lp_sigm(float v) { return 1.f / (1.f + __expf(-v)); }

struct LstmPersistWs {
    unsigned* err;   // this launch's abort word (cleared before every launch): lets the other workgroups stop waiting early
    unsigned* sticky;  // raised together with it and left standing, so that a timeout in ANY launch since the host last looked is seen
    unsigned* cnt;   // this launch's counters (stride 32 words)
    char* hx;        // [groups][2 parity][2 layer][64 wg][NB][8] bf16
};

// wait until the group's counter reaches `target`; false once the launch is known to be broken
__device__ __forceinline__ bool lp_wait(unsigned* cnt, unsigned* err, unsigned* sticky, unsigned target, bool dead) {
    if (dead) return true;
    int spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > LP_SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
    }
    return false;
}

template <int NBT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_fwd_persist_kernel(ga_lstm_params p, LstmPersistWs ws, int b_first, int affine) {
    constexpr int H = LP_H, NB = NBT * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const W1s = smem;                                   // [32][LP_W1_STRIDE]
    char* const W2s = smem + 32 * LP_W1_STRIDE;               // [32][LP_W2_STRIDE]
    f32x4* const red = (f32x4*)(W2s + 32 * LP_W2_STRIDE);     // [4 source waves][3 other (layer, m) combos][NBT][64 lanes]
    __bf16* const hst = (__bf16*)(red + 4 * 3 * NBT * 64);    // [2 layers][NB][8]
    __shared__ int dead_s;

    // workgroup -> (group g, unit slice j).  affine > 0 (= the number of groups, grid of 256): the 64 workgroups of a group are
    // the ones the dispatcher deals to TWO of the eight XCDs (blocks b and b + 8 share an XCD), so a group's exchange lines
    // are fetched into two L2s instead of eight; workgroups of unused XCD pairs leave at once.  Placement only changes speed.
    int j, g;
    if (affine > 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        g = xcd >> 1;
        j = (xcd & 1) * 32 + slot;
        if (g >= affine) return;
    } else {
        j = blockIdx.x % LP_WG_PER_GROUP;
        g = blockIdx.x / LP_WG_PER_GROUP;
    }
    const int b0 = b_first + g * NB;          // first batch item of this group
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int T = p.T;
    unsigned* const cnt = ws.cnt + 32 * g;
    char* const hx = ws.hx + (size_t)g * (2 * 2 * LP_WG_PER_GROUP * NB * 16);
    const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(hx, 0, 2 * 2 * LP_WG_PER_GROUP * NB * 16, 0x00020000);

    // ---- weights of this workgroup's 8 units into LDS, once.  LDS row 16m + 4q + gate <-> unit 8j + 4m + q: lane kq of an
    // MFMA accumulator then holds the four gates (register index = gate) of ONE unit.
    {
        const bf16_t* w1 = (const bf16_t*)p.w1;  // weight_hh_l0 [4H][H]
        const bf16_t* w2 = (const bf16_t*)p.w2;  // [weight_ih_l1 | weight_hh_l1] [4H][2H]
        for (int e = tid; e < 32 * 64; e += 256) {
            const int row = e >> 6, pc = e & 63;
            const int grow = (row & 3) * H + 8 * j + 4 * (row >> 4) + ((row >> 2) & 3);
            *(uint4*)(W1s + row * LP_W1_STRIDE + pc * 16) = *(const uint4*)(w1 + (size_t)grow * H + pc * 8);
        }
        for (int e = tid; e < 32 * 128; e += 256) {
            const int row = e >> 7, pc = e & 127;
            const int grow = (row & 3) * H + 8 * j + 4 * (row >> 4) + ((row >> 2) & 3);
            *(uint4*)(W2s + row * LP_W2_STRIDE + pc * 16) = *(const uint4*)(w2 + (size_t)grow * 2 * H + pc * 8);
        }
    }
    if (tid == 0) dead_s = 0;
    __syncthreads();

    // ---- roles.  MFMA phase: wave `wid` takes K range [128 wid, 128 wid + 128) of every product (both tiles m, all item
    // tiles).  Cell phase: wave wid owns (layer = wid >> 1, m = wid & 1): lane (n, kq) = unit 8j + 4m + kq, item nt*16 + n.
    const int c_layer = wid >> 1, c_m = wid & 1;
    const int c_u = 8 * j + 4 * c_m + kq;
    float cstate[NBT];
#pragma unroll
    for (int i = 0; i < NBT; ++i) cstate[i] = 0.f;
    const float bs0 = p.bias2[c_u], bs1 = p.bias2[H + c_u], bs2 = p.bias2[2 * H + c_u], bs3 = p.bias2[3 * H + c_u];
    const bf16_t* __restrict__ xg1 = (const bf16_t*)p.xg1;
    const bf16_t* __restrict__ x_in = (const bf16_t*)p.x_in;
    bf16_t* __restrict__ y = (bf16_t*)p.y;
    const int a_row_off1 = n * LP_W1_STRIDE + (128 * wid + 8 * kq) * 2;
    const int a_row_off2 = n * LP_W2_STRIDE + (128 * wid + 8 * kq) * 2;
    bool dead = false;

    for (int s = 0; s <= T; ++s) {
        // ---- inputs of the cell phase that do not depend on the recurrence: requested before the wait
        const int c_t = c_layer == 0 ? s : s - 1;
        const bool c_live = c_t >= 0 && c_t < T;
        bf16_t e_in[NBT][4];
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
            const int b = b0 + i * 16 + n;
            const int bl = b < p.B ? b : p.B - 1;
            const int tl = c_live ? c_t : 0;
            if (c_layer == 0) {
                const bf16_t* q = xg1 + ((size_t)bl * T + tl) * 4 * H + c_u;
                e_in[i][0] = q[0]; e_in[i][1] = q[H]; e_in[i][2] = q[2 * H]; e_in[i][3] = q[3 * H];
            } else {
                e_in[i][0] = x_in[((size_t)bl * T + tl) * H + c_u];
                e_in[i][1] = e_in[i][2] = e_in[i][3] = e_in[i][0];
            }
        }

        // ---- wait for every workgroup's publication of step s-1
        if (s > 0) {
            if (tid == 0 && lp_wait(cnt, ws.err, ws.sticky, (unsigned)(LP_WG_PER_GROUP * s), dead)) dead_s = 1;
            __syncthreads();
            dead = dead_s != 0;
        }

        // ---- gate pre-activations: acc1[m][i] = W1 . h1[s-1],  acc2[m][i] = W2a . h1[s-1] + W2b . h2[s-2]
        f32x4 acc1[2][NBT], acc2[2][NBT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < NBT; ++i) {
                acc1[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc2[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        if (s > 0) {
            const int par = (s - 1) & 1;
            // piece index of this lane's 8 k values at K-step ks: (128 wid + 32 ks + 8 kq) / 8 = 16 wid + 4 ks + kq
            u32x4 hb1[4][NBT], hb2[4][NBT];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < NBT; ++i) {
                    const int pc = 16 * wid + 4 * ks + kq;
                    const int off = (((par * 2 + 0) * LP_WG_PER_GROUP + pc) * NB + i * 16 + n) * 16;
                    hb1[ks][i] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, off, 0, 16);  // sc1
                    hb2[ks][i] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, off + LP_WG_PER_GROUP * NB * 16, 0, 16);
                }
            __builtin_amdgcn_sched_barrier(0);  // every exchange load of the step is in flight before the first MFMA waits on one
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const bf16x8 a1 = *(const bf16x8*)(W1s + m * 16 * LP_W1_STRIDE + a_row_off1 + ks * 64);
                    const bf16x8 a2a = *(const bf16x8*)(W2s + m * 16 * LP_W2_STRIDE + a_row_off2 + ks * 64);
                    const bf16x8 a2b = *(const bf16x8*)(W2s + m * 16 * LP_W2_STRIDE + a_row_off2 + H * 2 + ks * 64);
#pragma unroll
                    for (int i = 0; i < NBT; ++i) {
                        const bf16x8 b1 = __builtin_bit_cast(bf16x8, hb1[ks][i]);
                        const bf16x8 b2 = __builtin_bit_cast(bf16x8, hb2[ks][i]);
                        acc1[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc1[m][i], 0, 0, 0);
                        acc2[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2a, b1, acc2[m][i], 0, 0, 0);
                        acc2[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2b, b2, acc2[m][i], 0, 0, 0);
                    }
                }
        }
        // (h2[s-2] does not exist at s == 1: its parity slot holds zeros written by the step-0 publication below)
        // K was split over the four waves: wave c owns combo c = 2*layer + m in the cell phase, so every wave hands its
        // partial sums of the three combos it does not own to their owners (slot = combo index with the own one left out)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c != wid) {
                const int slot = c < wid ? c : c - 1;
#pragma unroll
                for (int i = 0; i < NBT; ++i) red[((wid * 3 + slot) * NBT + i) * 64 + lane] = (c & 2) ? acc2[c & 1][i] : acc1[c & 1][i];
            }
        __syncthreads();

        // ---- cell update for (c_layer, unit c_u, items i*16 + n)
        float o_i[NBT], o_f[NBT], o_g[NBT], o_o[NBT], o_c[NBT], o_h[NBT];
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
            // own partial (a wave-uniform choice among the four accumulators) + the three other waves', in wave order
            f32x4 pre = wid == 0 ? acc1[0][i] : wid == 1 ? acc1[1][i] : wid == 2 ? acc2[0][i] : acc2[1][i];
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (w != wid) {
                    const int slot = wid < w ? wid : wid - 1;
                    const f32x4 t4 = red[((w * 3 + slot) * NBT + i) * 64 + lane];
                    pre[0] += t4[0]; pre[1] += t4[1]; pre[2] += t4[2]; pre[3] += t4[3];
                }
            if (c_layer == 0) {
                pre[0] += (float)e_in[i][0]; pre[1] += (float)e_in[i][1]; pre[2] += (float)e_in[i][2]; pre[3] += (float)e_in[i][3];
            } else {
                pre[0] += bs0; pre[1] += bs1; pre[2] += bs2; pre[3] += bs3;
            }
            const float ig = lp_sigm(pre[0]), fg = lp_sigm(pre[1]), gg = tanhf(pre[2]), og = lp_sigm(pre[3]);
            const float c = fg * cstate[i] + ig * gg;
            const float h = c_live ? og * tanhf(c) : 0.f;
            if (c_live) cstate[i] = c;
            o_i[i] = ig; o_f[i] = fg; o_g[i] = gg; o_o[i] = og; o_c[i] = c; o_h[i] = h;
            hst[(c_layer * NB + i * 16 + n) * 8 + 4 * c_m + kq] = (__bf16)h;
        }
        __syncthreads();

        // ---- publish h1[s] and h2[s-1] of this workgroup's units (not needed after the last step)
        if (s < T) {
            if (tid < 2 * NB) {
                const int layer = tid / NB, item = tid - layer * NB;
                const u32x4 v = *(const u32x4*)(hst + (layer * NB + item) * 8);
                const int off = ((((s & 1) * 2 + layer) * LP_WG_PER_GROUP + j) * NB + item) * 16;
                __builtin_amdgcn_raw_buffer_store_b128(v, hx_rsrc, off, 0, 16);  // sc1: write-through
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }

        // ---- results to memory (off the critical path: the next step's wait covers them)
        if (c_live) {
#pragma unroll
            for (int i = 0; i < NBT; ++i) {
                const int b = b0 + i * 16 + n;
                if (b >= p.B) continue;
                const size_t hi = ((size_t)b * T + c_t) * H + c_u;
                float* gsave = c_layer == 0 ? p.gates1 : p.gates2;
                if (gsave && (p.B_save <= 0 || b < p.B_save)) {
                    *(float4*)(gsave + (((size_t)b * T + c_t) * H + c_u) * 4) = make_float4(o_i[i], o_f[i], o_g[i], o_o[i]);  // one 16-byte store
                    (c_layer == 0 ? p.cs1 : p.cs2)[hi] = o_c[i];
                }
                if (c_layer == 1) y[hi] = (bf16_t)(o_h[i] + (float)e_in[i][0]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
static size_t lp_lds_bytes(int nbt) {
    return 32 * LP_W1_STRIDE + 32 * LP_W2_STRIDE + (size_t)4 * 3 * nbt * 64 * 16 + (size_t)2 * nbt * 16 * 8 * 2 + 64;
}

// groups and item tiles per group for `B` items on `cus` compute units (at most one workgroup per CU)
static void lp_shape(int B, int cus, int& groups, int& nbt) {
    int gmax = cus / LP_WG_PER_GROUP;
    if (gmax > 4) gmax = 4;
    if (gmax < 1) gmax = 1;
    const int tiles = (B + 15) / 16;
    groups = tiles < gmax ? tiles : gmax;
    nbt = (tiles + groups - 1) / groups;
    if (nbt > 4) nbt = 4;
    if (groups * nbt > tiles) groups = (tiles + nbt - 1) / nbt;  // no group made of padding only
}

extern "C" size_t ga_lstm_persist_workspace_bytes(int B) {
    (void)B;  // sized for the largest launch: 4 groups x 64 items, backward exchange (4 gate pieces per workgroup and layer)
    return LP_WS_HEADER + (size_t)4 * (2 * 2 * 4 * LP_WG_PER_GROUP * 64 * 16);
}

static int lp_affine() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("GA_LSTM_AFFINE");
        v = e ? atoi(e) : 1;
    }
    return v;
}

static int lp_cu_count() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
    }
    return cus;
}

// forward pass of `p` through the persistent kernel; GA_ERR_UNSUPPORTED when it does not apply (the caller then runs the
// step-launch chain of lstm.hip)
int ga_lstm_fwd_persist(const ga_lstm_params& p, hipStream_t s) {
    if (!p.ws || p.H != LP_H || p.w_dtype != GA_DTYPE_BF16 || p.act_dtype != GA_DTYPE_BF16) return GA_ERR_UNSUPPORTED;
    if (p.ws_bytes < ga_lstm_persist_workspace_bytes(p.B) || ((uintptr_t)p.ws & 255)) return GA_ERR_ARG;
    const int cus = lp_cu_count();
    if (cus < LP_WG_PER_GROUP) return GA_ERR_UNSUPPORTED;
    for (int b_first = 0; b_first < p.B;) {
        int groups, nbt;
        lp_shape(p.B - b_first, cus, groups, nbt);
        if (hipMemsetAsync((char*)p.ws + LP_WS_STICKY, 0, LP_WS_HEADER - LP_WS_STICKY, s) != hipSuccess) return GA_ERR_LAUNCH;
        LstmPersistWs ws;
        ws.err = (unsigned*)p.ws + LP_WS_STICKY / 4;
        ws.sticky = (unsigned*)p.ws;
        ws.cnt = (unsigned*)p.ws + 32;
        ws.hx = (char*)p.ws + LP_WS_HEADER;
        const size_t lds = lp_lds_bytes(nbt);
        const int affine = (lp_affine() && cus >= 256) ? groups : 0;
        const dim3 grid(affine ? 256 : groups * LP_WG_PER_GROUP);
#define LP_LAUNCH(N)                                                                                                     \
    do {                                                                                                                 \
        if (hipFuncSetAttribute((const void*)lstm_fwd_persist_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                (int)lds) != hipSuccess)                                                                 \
            return GA_ERR_LAUNCH;                                                                                        \
        lstm_fwd_persist_kernel<N><<<grid, 256, lds, s>>>(p, ws, b_first, affine);                                               \
    } while (0)
        switch (nbt) {
            case 1: LP_LAUNCH(1); break;
            case 2: LP_LAUNCH(2); break;
            case 3: LP_LAUNCH(3); break;
            default: LP_LAUNCH(4); break;
        }
#undef LP_LAUNCH
        GA_CHECK_LAUNCH();
        b_first += groups * nbt * 16;
    }
    return GA_OK;
}

// ================================================================================================
// Backward (input gradient only; the weights are frozen): ONE persistent launch, same decomposition and hand-off.
//   step s: layer 2 at t2 = T-1-s:  dh2 = dy[t2] + W_hh_l1^T dg2[t2+1]
//           layer 1 at t1 = T-s:    dh1 = W_ih_l1^T dg2[t1] + W_hh_l0^T dg1[t1+1]          (t2 + 1 == t1 == T - s)
//   Workgroup j owns units 8j .. 8j+7: rows u of W_hh_l1^T, W_ih_l1^T and W_hh_l0^T (8 x 2048 bf16 each, 96 KiB in LDS).
//   MFMA tile X (16 rows = W_hh_l1^T rows | W_ih_l1^T rows) x dg2[T-s], tile Y (W_hh_l0^T rows, upper half unused) x
//   dg1[T-s+1]; K = 4H = 2048 split over the four waves.  Exchanged per step: the four gate gradients of the workgroup's 8
//   units, both layers (pieces of 8 units x 16 bytes, piece index = gate*64 + workgroup, i.e. k / 8 in the weights' K order).
//   dg1 is also written to memory (the caller turns it into dx with one GEMM); dg2 only lives in the exchange buffer.
#define LP_WT_STRIDE 4112  // bytes per LDS row (2048 bf16 + 16)

template <int NBT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_bwd_persist_kernel(ga_lstm_bwd_params p, LstmPersistWs ws, int b_first, int affine) {
    constexpr int H = LP_H, NB = NBT * 16, NPC = 4 * LP_WG_PER_GROUP;  // pieces per (parity, layer)
    constexpr int CH = NBT == 1 ? 16 : NBT == 2 ? 8 : 4;                // K-steps whose operands are requested together (<= 128 registers)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Ws = smem;                                             // [24][LP_WT_STRIDE]: W_hh_l1^T | W_ih_l1^T | W_hh_l0^T rows
    float* const red = (float*)(Ws + 24 * LP_WT_STRIDE);               // [2 tiles X,Y][4 waves][NBT][64 lanes][4]
    __bf16* const dgst = (__bf16*)(red + 2 * 4 * NBT * 64 * 4);        // [2 layers][4 gates][NB][8]
    __shared__ int dead_s;

    // workgroup -> (group g, unit slice j).  affine > 0 (= the number of groups, grid of 256): the 64 workgroups of a group are
    // the ones the dispatcher deals to TWO of the eight XCDs (blocks b and b + 8 share an XCD), so a group's exchange lines
    // are fetched into two L2s instead of eight; workgroups of unused XCD pairs leave at once.  Placement only changes speed.
    int j, g;
    if (affine > 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        g = xcd >> 1;
        j = (xcd & 1) * 32 + slot;
        if (g >= affine) return;
    } else {
        j = blockIdx.x % LP_WG_PER_GROUP;
        g = blockIdx.x / LP_WG_PER_GROUP;
    }
    const int b0 = b_first + g * NB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int T = p.T;
    unsigned* const cnt = ws.cnt + 32 * g;
    char* const hx = ws.hx + (size_t)g * (2 * 2 * NPC * NB * 16);
    const __amdgpu_buffer_rsrc_t hx_rsrc = __builtin_amdgcn_make_buffer_rsrc(hx, 0, 2 * 2 * NPC * NB * 16, 0x00020000);

    {
        const bf16_t* w1t = (const bf16_t*)p.w1t;  // weight_hh_l0^T [H][4H]
        const bf16_t* w2t = (const bf16_t*)p.w2t;  // [weight_ih_l1^T ; weight_hh_l1^T] [2H][4H]
        for (int e = tid; e < 24 * 256; e += 256) {
            const int row = e >> 8, pc = e & 255;
            const int u = 8 * j + (row & 7);
            const bf16_t* src = row < 8 ? w2t + (size_t)(H + u) * 4 * H : row < 16 ? w2t + (size_t)u * 4 * H : w1t + (size_t)u * 4 * H;
            *(uint4*)(Ws + row * LP_WT_STRIDE + pc * 16) = *(const uint4*)(src + pc * 8);
        }
    }
    if (tid == 0) dead_s = 0;
    __syncthreads();

    // cell-phase role: thread = (layer, unit, item column); layer 0 = LSTM layer 2 (leads), layer 1 = LSTM layer 1
    const int c_n = tid & 15, c_uu = (tid >> 4) & 7, c_layer = tid >> 7;
    const int c_u = 8 * j + c_uu;
    const float* __restrict__ gates = c_layer == 0 ? p.gates2 : p.gates1;
    const float* __restrict__ cs = c_layer == 0 ? p.cs2 : p.cs1;
    const bf16_t* __restrict__ dy = (const bf16_t*)p.dy;
    bf16_t* __restrict__ dg1out = (bf16_t*)p.dg1;
    float dcst[NBT];
#pragma unroll
    for (int i = 0; i < NBT; ++i) dcst[i] = 0.f;
    const int a_off_x = n * LP_WT_STRIDE + (512 * wid + 8 * kq) * 2;
    const int a_off_y = (16 + (n & 7)) * LP_WT_STRIDE + (512 * wid + 8 * kq) * 2;
    bool dead = false;

    for (int s = 0; s <= T; ++s) {
        const int c_t = c_layer == 0 ? T - 1 - s : T - s;
        const bool c_live = c_t >= 0 && c_t < T;
        // ---- cell-phase inputs (independent of the recurrence), requested before the wait
        float e_ig[NBT], e_fg[NBT], e_gg[NBT], e_og[NBT], e_c[NBT], e_cp[NBT];
        bf16_t e_dy[NBT];
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
            const int b = b0 + i * 16 + c_n;
            const int bl = b < p.B ? b : p.B - 1;
            const int tl = c_live ? c_t : 0;
            const size_t hi = ((size_t)bl * T + tl) * H + c_u;
            const float4 gs = *(const float4*)(gates + (((size_t)bl * T + tl) * H + c_u) * 4);
            e_ig[i] = gs.x; e_fg[i] = gs.y; e_gg[i] = gs.z; e_og[i] = gs.w;
            e_c[i] = cs[hi];
            e_cp[i] = cs[tl > 0 ? hi - H : hi];
            e_dy[i] = dy[hi];
        }

        if (s > 0) {
            if (tid == 0 && lp_wait(cnt, ws.err, ws.sticky, (unsigned)(LP_WG_PER_GROUP * s), dead)) dead_s = 1;
            __syncthreads();
            dead = dead_s != 0;
        }

        f32x4 accx[NBT], accy[NBT];
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
            accx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            accy[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (s > 0) {
            const int par = (s - 1) & 1;
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += CH) {
                u32x4 bx[CH][NBT], by[CH][NBT];
#pragma unroll
                for (int ks = 0; ks < CH; ++ks)
#pragma unroll
                    for (int i = 0; i < NBT; ++i) {
                        const int pc = 64 * wid + 4 * (k0 + ks) + kq;  // (512 wid + 32 (k0 + ks) + 8 kq) / 8
                        const int off = (((par * 2 + 0) * NPC + pc) * NB + i * 16 + n) * 16;
                        bx[ks][i] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, off, 0, 16);                   // dg2[T-s], sc1
                        by[ks][i] = __builtin_amdgcn_raw_buffer_load_b128(hx_rsrc, off + NPC * NB * 16, 0, 16);   // dg1[T-s+1]
                    }
#pragma unroll
                for (int ks = 0; ks < CH; ++ks) {
                    const bf16x8 ax = *(const bf16x8*)(Ws + a_off_x + (k0 + ks) * 64);
                    const bf16x8 ay = *(const bf16x8*)(Ws + a_off_y + (k0 + ks) * 64);
#pragma unroll
                    for (int i = 0; i < NBT; ++i) {
                        accx[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, __builtin_bit_cast(bf16x8, bx[ks][i]), accx[i], 0, 0, 0);
                        accy[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ay, __builtin_bit_cast(bf16x8, by[ks][i]), accy[i], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
            *(f32x4*)(red + (((0 * 4 + wid) * NBT + i) * 64 + lane) * 4) = accx[i];
            *(f32x4*)(red + (((1 * 4 + wid) * NBT + i) * 64 + lane) * 4) = accy[i];
        }
        __syncthreads();

        // ---- cell phase: dh -> gate gradients of (c_layer, unit c_uu, items i*16 + c_n)
        float o_dg[NBT][4];
#pragma unroll
        for (int i = 0; i < NBT; ++i) {
            float dh = c_layer == 0 ? (float)e_dy[i] : 0.f;
            // tile X rows 0-7 (dh2) / 8-15 (dh1 through W_ih_l1); tile Y rows 0-7 (dh1 through W_hh_l0): row = 4*kq' + reg
            const int xl = c_n + 16 * ((c_layer == 0 ? 0 : 2) + (c_uu >> 2)), yl = c_n + 16 * (c_uu >> 2), rg = c_uu & 3;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                dh += red[(((0 * 4 + w) * NBT + i) * 64 + xl) * 4 + rg];
                if (c_layer == 1) dh += red[(((1 * 4 + w) * NBT + i) * 64 + yl) * 4 + rg];
            }
            const float cprev = c_t > 0 ? e_cp[i] : 0.f;
            const float ig = e_ig[i], fg = e_fg[i], gg = e_gg[i], og = e_og[i];
            const float tc = tanhf(e_c[i]);
            const float dc = dcst[i] + dh * og * (1.f - tc * tc);
            if (c_live) dcst[i] = dc * fg;
            o_dg[i][0] = c_live ? dc * gg * ig * (1.f - ig) : 0.f;
            o_dg[i][1] = c_live ? dc * cprev * fg * (1.f - fg) : 0.f;
            o_dg[i][2] = c_live ? dc * ig * (1.f - gg * gg) : 0.f;
            o_dg[i][3] = c_live ? dh * tc * og * (1.f - og) : 0.f;
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) dgst[((c_layer * 4 + gt) * NB + i * 16 + c_n) * 8 + c_uu] = (__bf16)o_dg[i][gt];
        }
        __syncthreads();

        // ---- publish (2 layers x 4 gates x NB items pieces of 16 bytes), and dg1[t1] to memory for the dx GEMM
        for (int e = tid; e < 8 * NB; e += 256) {
            const int lg = e / NB, item = e - lg * NB;          // lg = layer*4 + gate
            const u32x4 v = *(const u32x4*)(dgst + (lg * NB + item) * 8);
            if (s < T) {
                const int off = ((((s & 1) * 2 + (lg >> 2)) * NPC + (lg & 3) * LP_WG_PER_GROUP + j) * NB + item) * 16;
                __builtin_amdgcn_raw_buffer_store_b128(v, hx_rsrc, off, 0, 16);  // sc1: write-through
            }
        }
        if (s < T) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (s >= 1) {  // layer 1 was live at t1 = T - s
            const int t1 = T - s;
            for (int e = tid; e < 4 * NB; e += 256) {
                const int gt = e / NB, item = e - gt * NB;
                const int b = b0 + item;
                if (b < p.B)
                    *(u32x4*)(dg1out + ((size_t)b * T + t1) * 4 * H + gt * H + 8 * j) = *(const u32x4*)(dgst + ((4 + gt) * NB + item) * 8);
            }
        }
        // (dgst is rewritten only behind the next step's first two barriers, after these reads)
    }
}

static size_t lp_bwd_lds_bytes(int nbt) {
    return 24 * LP_WT_STRIDE + (size_t)2 * 4 * nbt * 64 * 16 + (size_t)2 * 4 * nbt * 16 * 8 * 2 + 64;
}

int ga_lstm_bwd_persist(const ga_lstm_bwd_params& p, hipStream_t s) {
    if (!p.ws || p.H != LP_H || p.w_dtype != GA_DTYPE_BF16 || p.act_dtype != GA_DTYPE_BF16) return GA_ERR_UNSUPPORTED;
    if (p.ws_bytes < ga_lstm_persist_workspace_bytes(p.B) || ((uintptr_t)p.ws & 255)) return GA_ERR_ARG;
    const int cus = lp_cu_count();
    if (cus < LP_WG_PER_GROUP) return GA_ERR_UNSUPPORTED;
    for (int b_first = 0; b_first < p.B;) {
        int groups, nbt;
        lp_shape(p.B - b_first, cus, groups, nbt);
        if (hipMemsetAsync((char*)p.ws + LP_WS_STICKY, 0, LP_WS_HEADER - LP_WS_STICKY, s) != hipSuccess) return GA_ERR_LAUNCH;
        LstmPersistWs ws;
        ws.err = (unsigned*)p.ws + LP_WS_STICKY / 4;
        ws.sticky = (unsigned*)p.ws;
        ws.cnt = (unsigned*)p.ws + 32;
        ws.hx = (char*)p.ws + LP_WS_HEADER;
        const size_t lds = lp_bwd_lds_bytes(nbt);
        const int affine = (lp_affine() && cus >= 256) ? groups : 0;
        const dim3 grid(affine ? 256 : groups * LP_WG_PER_GROUP);
#define LP_LAUNCH(N)                                                                                                     \
    do {                                                                                                                 \
        if (hipFuncSetAttribute((const void*)lstm_bwd_persist_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                (int)lds) != hipSuccess)                                                                 \
            return GA_ERR_LAUNCH;                                                                                        \
        lstm_bwd_persist_kernel<N><<<grid, 256, lds, s>>>(p, ws, b_first, affine);                                               \
    } while (0)
        switch (nbt) {
            case 1: LP_LAUNCH(1); break;
            case 2: LP_LAUNCH(2); break;
            case 3: LP_LAUNCH(3); break;
            default: LP_LAUNCH(4); break;
        }
#undef LP_LAUNCH
        GA_CHECK_LAUNCH();
        b_first += groups * nbt * 16;
    }
    return GA_OK;
}
