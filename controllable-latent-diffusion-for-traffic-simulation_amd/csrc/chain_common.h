// chain_common.h -- helpers shared by the layer-chain kernels (conv_chain.hip: direct form; chain_wino.hip: Winograd F(4, 5) form):
// tile geometry of the LDS-resident image, the direct-form K loop over it, weight-fragment queues, GroupNorm + Mish on accumulators.
#pragma once
#include "cld_kernels.h"

#ifndef CLD_STORE_AUX
#define CLD_STORE_AUX 16
#endif

namespace cld {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#ifdef CLD_STAMPS
// diagnostic build: in-kernel cycle stamps (never compiled into the shipped library)
#define CSTAMP(k)                                                                                  \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (p.stamps && tid == 0) {                                                                \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            p.stamps[(size_t)blockIdx.x * 16 + (k)] = t_;                                          \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define CSTAMP_RT(k)                                                                               \
    do {                                                                                           \
        if (p.stamps && tid == 0) {                                                                \
            unsigned long long t_;                                                                 \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            p.stamps[(size_t)blockIdx.x * 16 + (k)] = t_;                                          \
        }                                                                                          \
    } while (0)
#else
#define CSTAMP(k) do {} while (0)
#define CSTAMP_RT(k) do {} while (0)
#endif

namespace {

__device__ __forceinline__ float mish_c(float x) {      // conv_block.hip mish_f
    const float e = __expf(fminf(x, 30.0f));
    const float n = e * (e + 2.0f);
    return x * n * __builtin_amdgcn_rcpf(n + 2.0f);
}
__device__ __forceinline__ v4f bload16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// Geometry of a 64-channel tile of AG agents x L rows (AG = 4: the throughput tile, 13 M-tiles at L = 52; AG = 1: the
// small-batch tile, one workgroup per agent).  LDS image: rows of KCP floats; agent a's rows start at a * ASTR, two zero rows
// lead every agent block (the next agent's lead rows are the previous one's trailing halo), AEX extra floats per block keep the
// fragment reads conflict-free (scripts/lds_conflicts.py: AG = 4: L = 52 -> 0, L = 26 -> 16; AG = 1: 16 consecutive rows of
// 72 floats are conflict-free as they are).  M-tile m holds rows RPT m .. RPT m + RPT - 1 of every agent: GEMM row i of a tile
// (lane i of a fragment read; accumulator register r of lane group q is row 4 q + r) is agent i % AG, row RPT m + i / AG.
template <int L_, int AG_>
struct Geo {
    static constexpr int C = 64, L = L_, AG = AG_;
    static constexpr int KCP = C + 8;
    static constexpr int LP = L + 2;
    static constexpr int AEX = (AG == 4 && L == 26) ? 16 : (AG == 2 && L == 52) ? 8 : 0;      // (AG = 2, chain_wino.hip: the L = 52 image is read with stride 2)
    static constexpr int ASTR = LP * KCP + AEX;
    static constexpr int RPT = 16 / AG;                 // rows of one agent per M-tile
    static constexpr int NMT = (AG * L + 15) / 16;
    static constexpr bool RAGGED = NMT * 16 != AG * L;  // the last M-tile carries rows past the agents' ends
    static constexpr int IMG = (AG * LP + 2) * KCP + AG * AEX;
    static_assert(AG == 4 || AG == 2 || AG == 1, "tiles of 4 agents, of 2 (chain_wino.hip only) or of 1");
    static __device__ __forceinline__ int agent(int q, int r) { return AG == 4 ? r : AG == 2 ? (r & 1) : 0; }
    static __device__ __forceinline__ int pos(int m, int q, int r) { return AG == 4 ? 4 * m + q : AG == 2 ? 8 * m + 2 * q + (r >> 1) : 16 * m + 4 * q + r; }
    static __device__ __forceinline__ bool ok(int m, int q, int r) { return !RAGGED || m < NMT - 1 || pos(m, q, r) < L; }
    // float offset of (agent, row) of accumulator register r of M-tile m relative to the lane's base  row0(q) * KCP
    static constexpr int roff(int m, int r) { return AG == 4 ? r * ASTR + 4 * m * KCP : (16 * m + r) * KCP; }
    static __device__ __forceinline__ int row0(int q) { return AG == 4 ? q : 4 * q; }
    // fragment base (floats) of lane i16: agent i % AG, row i / AG of M-tile 0
    static __device__ __forceinline__ int frag0(int i16) { return (i16 % AG) * ASTR + (i16 / AG) * KCP; }
};
constexpr int kSlackFloats = 16 * 72;       // rows behind the image that ragged / stride-2 fragment reads run into (read, never used)
template <int AG> constexpr int img_floats() { return (Geo<52, AG>::IMG > Geo<26, AG>::IMG ? Geo<52, AG>::IMG : Geo<26, AG>::IMG) + kSlackFloats; }

// Weight fragments run WD (tap, group) iterations ahead of the MFMAs that consume them: an iteration is 4 NMT MFMAs, i.e.
// 128 NMT cycles of cover, and a fragment comes from L2 (~1.5k cycles) -- two iterations ahead is enough for the 13-M-tile
// throughput tile and far too little for the 4- and 2-M-tile small-batch tiles, whose loop otherwise waits on every fragment.
constexpr int weight_depth(int nmt, int nit) {
    int d = (3400 + 128 * nmt - 1) / (128 * nmt);
    d = d < 2 ? 2 : d;
    return d > nit ? nit : d;
}
template <int NTAPS>
__device__ __forceinline__ v4f wfrag_load(const __amdgpu_buffer_rsrc_t rsw, int wlane, int ntn, int ntile, int it) {
    const int c = it / (2 * NTAPS), ii = it % (2 * NTAPS), t = ii / 2, g = ii % 2;
    return bload16(rsw, wlane, (((2 * c + g) * NTAPS + t) * ntn + ntile) * 1024);
}
// the first WD fragments of a layer: issued by the caller BEFORE the previous layer's epilogue, so that a layer does not start
// behind an L2 round trip
template <int C_IN, int NTAPS, int NMT>
struct WQueue {
    static constexpr int NIT = (C_IN / 16) * NTAPS, WD = weight_depth(NMT, NIT);
    v4f q[WD];
    __device__ __forceinline__ void prime(const __amdgpu_buffer_rsrc_t rsw, int wlane, int ntn, int ntile) {
#pragma unroll
        for (int i = 0; i < WD; ++i) q[i] = wfrag_load<NTAPS>(rsw, wlane, ntn, ntile, i);
    }
};

// K loop over an LDS-resident image: acc[m] += sum over (chunk c, tap t, group g) in conv_block.hip's order.
// abase: byte address of this lane's fragment for (M-tile 0, tap 0, channel group 0); M-tile m is MSTEP bytes further.
template <int KCP, int MSTEP, int C_IN, int NTAPS, int NMT>
__device__ __forceinline__ void kloop(v4f (&acc)[NMT], const char* ldsb, const int abase, const __amdgpu_buffer_rsrc_t rsw,
                                      const int wlane, const int ntn, const int ntile, WQueue<C_IN, NTAPS, NMT>& wq) {
    constexpr int NIT = (C_IN / 16) * NTAPS, WD = WQueue<C_IN, NTAPS, NMT>::WD;
    auto loff = [](int it) {                 // LDS byte offset of iteration it = (chunk, tap, group)
        const int c = it / (2 * NTAPS), ii = it % (2 * NTAPS), t = ii / 2, g = ii % 2;
        return (t * KCP + 16 * (2 * c + g)) * 4;
    };
    v4f af[2][NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) af[0][m] = *reinterpret_cast<const v4f*>(ldsb + abase + m * MSTEP + loff(0));
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int cur = it & 1;
        const v4f bcur = wq.q[it % WD];
        if (it + WD < NIT) wq.q[it % WD] = wfrag_load<NTAPS>(rsw, wlane, ntn, ntile, it + WD);
#pragma unroll
        for (int g = 0; g < NMT; ++g) {
            if (it + 1 < NIT) af[cur ^ 1][g] = *reinterpret_cast<const v4f*>(ldsb + abase + g * MSTEP + loff(it + 1 < NIT ? it + 1 : 0));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = 4 * g + q, sidx = idx / NMT, m = idx % NMT;
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][m][sidx], bcur[sidx], acc[m], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// sum over the 8 lanes of a GroupNorm group (channels) and the 4 lane groups (rows) that hold one agent's values, left in every
// lane: three DPP adds inside the 16-lane row (after the two quad steps every lane of a quad holds the quad's sum, so the
// half-row mirror is as good as an xor by 4) and the gfx950 row / half swaps -- pure VALU, no LDS crossbar
#define CLD_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
__device__ __forceinline__ float group_sum(float s) {
    s += CLD_DPP(s, 0xB1);       // quad_perm:[1,0,3,2]
    s += CLD_DPP(s, 0x4E);       // quad_perm:[2,3,0,1]
    s += CLD_DPP(s, 0x141);      // row_half_mirror
    const auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s), __builtin_bit_cast(unsigned, s), false, false);
    const unsigned a16 = r16[0], b16 = r16[1];      // (scalars first: __builtin_bit_cast of a vector ELEMENT reads element 0)
    s = __builtin_bit_cast(float, a16) + __builtin_bit_cast(float, b16);
    const auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s), __builtin_bit_cast(unsigned, s), false, false);
    const unsigned a32 = r32[0], b32 = r32[1];
    return __builtin_bit_cast(float, a32) + __builtin_bit_cast(float, b32);
}

// GroupNorm(8 channels x L rows per agent, eps 1e-5, biased variance; diffuser_helpers.py:61) + Mish + per-agent vector, on the
// accumulators of one wave: lane (n, q) register r of M-tile m = (agent r, row RPT m + q, channel 16 wave + n).  Written on
// register PAIRS (agents 0 | 1 and 2 | 3) so that the adds / multiplies / FMAs compile to the packed fp32 instructions
// (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32: two values per issue slot) -- the epilogue is issue-bound next to the other
// workgroup's MFMA loop, and only the exponential, the reciprocal and the clamp stay one value per instruction.
template <class G>
__device__ __forceinline__ void gn_mish(v4f (&acc)[G::NMT], const float bias, const float gam, const float bet, const float (&add)[4], const int q) {
    const v2f bias2 = {bias, bias};
    v2f s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < G::NMT; ++m) {
        v2f lo = v2f{acc[m][0], acc[m][1]} + bias2, hi = v2f{acc[m][2], acc[m][3]} + bias2;
        acc[m] = v4f{lo[0], lo[1], hi[0], hi[1]};
        if (G::RAGGED && m == G::NMT - 1) {
            lo = v2f{G::ok(m, q, 0) ? lo[0] : 0.f, G::ok(m, q, 1) ? lo[1] : 0.f};
            hi = v2f{G::ok(m, q, 2) ? hi[0] : 0.f, G::ok(m, q, 3) ? hi[1] : 0.f};
        }
        s01 += lo; s23 += hi;
    }
    const float inv = 1.0f / (float)(8 * G::L);
    v2f mean01, mean23;
    if (G::AG == 1) {                // the four registers are four rows of the one agent
        const float mu = group_sum((s01[0] + s01[1]) + (s23[0] + s23[1])) * inv;
        mean01 = v2f{mu, mu}; mean23 = mean01;
    } else {
        mean01 = v2f{group_sum(s01[0]) * inv, group_sum(s01[1]) * inv}; mean23 = v2f{group_sum(s23[0]) * inv, group_sum(s23[1]) * inv};
    }
    v2f q01 = {0.f, 0.f}, q23 = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < G::NMT; ++m) {
        v2f lo = v2f{acc[m][0], acc[m][1]} - mean01, hi = v2f{acc[m][2], acc[m][3]} - mean23;
        if (G::RAGGED && m == G::NMT - 1) {
            lo = v2f{G::ok(m, q, 0) ? lo[0] : 0.f, G::ok(m, q, 1) ? lo[1] : 0.f};
            hi = v2f{G::ok(m, q, 2) ? hi[0] : 0.f, G::ok(m, q, 3) ? hi[1] : 0.f};
        }
        q01 += lo * lo; q23 += hi * hi;
    }
    v2f sc01, sc23;
    if (G::AG == 1) {
        const float sc = (1.0f / sqrtf(group_sum((q01[0] + q01[1]) + (q23[0] + q23[1])) * inv + 1e-5f)) * gam;
        sc01 = v2f{sc, sc}; sc23 = sc01;
    } else {
        sc01 = v2f{(1.0f / sqrtf(group_sum(q01[0]) * inv + 1e-5f)) * gam, (1.0f / sqrtf(group_sum(q01[1]) * inv + 1e-5f)) * gam};
        sc23 = v2f{(1.0f / sqrtf(group_sum(q23[0]) * inv + 1e-5f)) * gam, (1.0f / sqrtf(group_sum(q23[1]) * inv + 1e-5f)) * gam};
    }
    const v2f bet2 = {bet, bet}, add01 = {add[0], add[1]}, add23 = {add[2], add[3]}, two = {2.0f, 2.0f};
    auto mish2 = [&](const v2f x, const v2f ad) {          // x n / (n + 2) + ad, n = e^x (e^x + 2)   (conv_block.hip mish_f)
        const v2f c = v2f{fminf(x[0], 30.0f), fminf(x[1], 30.0f)} * v2f{1.4426950408889634f, 1.4426950408889634f};
        const v2f e = {__builtin_amdgcn_exp2f(c[0]), __builtin_amdgcn_exp2f(c[1])};
        const v2f nn = e * (e + two);
        const v2f d = nn + two;
        const v2f r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        return (x * nn) * r + ad;
    };
#pragma unroll
    for (int m = 0; m < G::NMT; ++m) {
        const v2f lo = mish2((v2f{acc[m][0], acc[m][1]} - mean01) * sc01 + bet2, add01);
        const v2f hi = mish2((v2f{acc[m][2], acc[m][3]} - mean23) * sc23 + bet2, add23);
        acc[m] = v4f{lo[0], lo[1], hi[0], hi[1]};
    }
}

// accumulators -> the image rows of the next layer (every lane one float per (M-tile, agent): 64-byte runs per lane group)
template <class G>
__device__ __forceinline__ void to_image(const v4f (&acc)[G::NMT], float* lds, const int wbase, const int q) {
#pragma unroll
    for (int m = 0; m < G::NMT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (G::ok(m, q, r)) lds[wbase + G::roff(m, r)] = acc[m][r];
}

template <class G>
__device__ __forceinline__ void zero_halo(float* lds, const int tid, const int nthr) {
    // the two leading rows and, behind every agent, its two halo rows (+ the AEX floats, + the slack rows behind the last agent)
    constexpr int GQ = (2 * G::KCP + G::AEX) / 4;
    for (int i = tid; i < (G::AG + 1) * GQ; i += nthr) {
        const int qd = i % GQ, g = i / GQ;
        if (g == 0 && qd >= 2 * G::KCP / 4) continue;
        const int at = g == 0 ? 0 : (g - 1) * G::ASTR + (2 + G::L) * G::KCP;
        *reinterpret_cast<v4f*>(lds + at + qd * 4) = v4f{0.f, 0.f, 0.f, 0.f};
    }
}

}  // namespace

}  // namespace cld
