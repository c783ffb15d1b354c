// Pieces shared by the per-wave LDS-window kernels (conv_swin.hip, resunit.hip): vector types, the direct-to-LDS load, LDS stores that
// do not drain pending LDS-DMA, packed bf16 ELU.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef const __attribute__((address_space(1))) bf16x8* gfrag_ptr_t;
typedef __attribute__((address_space(3))) char* lds_char_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void sw_glds16(uintptr_t src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ uintptr_t sw_sel(bool ok, uintptr_t a, uintptr_t z) {
    return z ^ ((z ^ a) & (uintptr_t)(-(long long)ok));
}

// LDS stores while direct-to-LDS loads are in flight: the compiler cannot tell that a ds_write and a pending LDS-DMA write do not
// alias and puts s_waitcnt vmcnt(0) in front of every ds_write it emits itself (and in front of LDS reads through struct types:
// the reads below go through vector types, which it lets pass), which would drain the prefetched tiles once per tile.  The
// stores are therefore issued as asm: LDS executes one wave's instructions in order, so the plain reads that follow see them,
// and the data registers are read at issue.
__device__ __forceinline__ void sw_lds_write16(void* dst, u32x4 v) {
    const unsigned a = (unsigned)(uintptr_t)(lds_char_ptr_t)dst;
    asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory");
}
__device__ __forceinline__ void sw_lds_write8(void* dst, u32x2 v) {
    const unsigned a = (unsigned)(uintptr_t)(lds_char_ptr_t)dst;
    asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory");
}

__device__ __forceinline__ unsigned int sw_pack_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    union { bf16x2 b; unsigned int u; } pk;
    pk.b[0] = (__bf16)lo;
    pk.b[1] = (__bf16)hi;
    return pk.u;
}

// ELU on eight bf16 (the arithmetic of conv_stream.hip's elu_frag: identical results), two elements per step so that the
// multiply / add / convert are packed instructions; f > -2^-12 returns f itself (expm1(f) = f there to inside half a bf16 ulp)
__device__ __forceinline__ unsigned int sw_elu2(unsigned int in) {
    f32x2 f;
    f.x = __uint_as_float(in << 16);
    f.y = __uint_as_float(in & 0xffff0000u);
    const f32x2 t = f * 1.4426950408889634f;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    e = e - 1.0f;
    const float rx = f.x > -2.44140625e-4f ? f.x : e.x;
    const float ry = f.y > -2.44140625e-4f ? f.y : e.y;
    return sw_pack_bf16(rx, ry);
}
__device__ __forceinline__ u32x4 sw_elu8(u32x4 in) {
    u32x4 out;
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = sw_elu2(in[i]);
    return out;
}

// In-place ELU over the chunks a lane loaded into a window (chunk i of the lane at byte (i*64 + lane)*16; the last of the NLW
// chunks exists for lane < WCH - (NLW-1)*64 only), EG chunks read ahead of their exponentials: one chunk at a time, every read
// is waited for before its eight exponentials start (the asm stores pin the order), an LDS round trip per chunk.  Reads past the
// window's last chunk stay inside the wave's LDS (the slot is followed by another slot or the stage) and are not written back.
template <int NLW, int WCH, int EG>
__device__ __forceinline__ void sw_elu_window(char* Wl, int lane) {
#pragma unroll
    for (int i0 = 0; i0 < NLW; i0 += EG) {
        u32x4 v[EG];
#pragma unroll
        for (int j = 0; j < EG; ++j)
            if (i0 + j < NLW) v[j] = *(const u32x4*)(Wl + ((i0 + j) * 64 + lane) * 16);
#pragma unroll
        for (int j = 0; j < EG; ++j) {
            const int i = i0 + j;
            if (i < NLW && ((i + 1) * 64 <= WCH || lane < WCH - i * 64)) sw_lds_write16(Wl + (i * 64 + lane) * 16, sw_elu8(v[j]));
        }
    }
}
