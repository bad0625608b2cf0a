// wino1d_common.h -- what the two Winograd F(4, 5) kernels of the L = 13 / 26 levels share (wino1d_kernels.hip: items of (agent, tile)
// rows; wino1d_edge.hip: items of agent rows with the ragged end of the sequence in the direct form): vector types, cycle stamps of
// the diagnostic build, packed Mish, the LDS slot permutation and the list of layer shapes.
#pragma once
#include "cld_kernels.h"

namespace cld {

typedef float v4f __attribute__((ext_vector_type(4)));

#ifdef CLD_STAMPS
// diagnostic build: in-kernel cycle stamps (never compiled into the shipped library); scripts/wino1d_stamps.py reads them
#define W1STAMP(k)                                                                                 \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (p.stamps && tid == 0) {                                                                \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            p.stamps[(size_t)blockIdx.x * 16 + (k)] = t_;                                          \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define W1STAMP_RT(k)                                                                              \
    do {                                                                                           \
        if (p.stamps && tid == 0) {                                                                \
            unsigned long long t_;                                                                 \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            p.stamps[(size_t)blockIdx.x * 16 + (k)] = t_;                                          \
        }                                                                                          \
    } while (0)
#else
#define W1STAMP(k) do {} while (0)
#define W1STAMP_RT(k) do {} while (0)
#endif

namespace {

__device__ __forceinline__ v4f fma4(const v4f a, const float s, const v4f b) { return __builtin_elementwise_fma(a, v4f{s, s, s, s}, b); }      // a s + b

__device__ __forceinline__ int hsw1(int k) { return ((k & 1) * 3) ^ (k >> 1); }      // wino_kernels.hip hsw

#define W1_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
// sum over the 4 lanes of a quad (the four tiles of an agent) and the 4 lane groups (channel quads), left in every lane involved
__device__ __forceinline__ float agent_sum(float s) {
    s += W1_DPP(s, 0xB1);       // quad_perm:[1,0,3,2]
    s += W1_DPP(s, 0x4E);       // quad_perm:[2,3,0,1]
    const auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s), __builtin_bit_cast(unsigned, s), false, false);
    const unsigned a16 = r16[0], b16 = r16[1];      // (scalars first: __builtin_bit_cast of a vector ELEMENT reads element 0)
    s = __builtin_bit_cast(float, a16) + __builtin_bit_cast(float, b16);
    const auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s), __builtin_bit_cast(unsigned, s), false, false);
    const unsigned a32 = r32[0], b32 = r32[1];
    return __builtin_bit_cast(float, a32) + __builtin_bit_cast(float, b32);
}
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
// Mish = x n / (n + 2), n = e^x (e^x + 2) (conv_block.hip mish_f: one v_exp_f32 and one v_rcp_f32 per value), on register pairs so that
// everything but the exponential, the reciprocal and the clamp is a packed fp32 instruction (two values per issue slot)
__device__ __forceinline__ v2f mish2(const v2f x) {
    const v2f c = v2f{fminf(x[0], 30.0f), fminf(x[1], 30.0f)} * v2f{1.4426950408889634f, 1.4426950408889634f};
    const v2f e = {__builtin_amdgcn_exp2f(c[0]), __builtin_amdgcn_exp2f(c[1])};
    const v2f two = {2.0f, 2.0f};
    const v2f n = e * (e + two);
    const v2f d = n + two;
    const v2f r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return (x * n) * r;
}
__device__ __forceinline__ v4f mish4(const v4f x) {
    const v2f lo = mish2(v2f{x[0], x[1]}), hi = mish2(v2f{x[2], x[3]});
    return v4f{lo[0], lo[1], hi[0], hi[1]};
}

}  // namespace

// (L, C_in, channels per source, C_out): the k5 + GroupNorm + Mish layers of the L = 13 and L = 26 levels
#define CLD_WINO1D_INSTANCES(X) \
    X(13, 256, 256, 256)        \
    X(13, 128, 128, 128)        \
    X(13, 128, 128, 256)        \
    X(13, 512, 256, 128)        \
    X(26, 128, 128, 128)        \
    X(26, 64, 64, 128)          \
    X(26, 256, 128, 64)

}  // namespace cld
