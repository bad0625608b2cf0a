// conv_chain.hip -- runs of 64-channel layers of TemporalMapUnet as ONE launch each, the activations never leaving the CU.
//
// At the 64-channel levels of the U-Net (reference: src/tbsim/models/temporal.py:148-176; L = 52: downs.0.*, final_conv;
// L = 26: ups.1.*) one conv workgroup owns whole agents AND every output channel, so the output tile of one layer IS the
// input tile of the next: nothing but the kernel boundary forced it through HBM.  conv_block_kernel runs these layers at
// 62-71 % of the fp32-MFMA peak at 4,096 rows (a third of each workgroup's life is entry latency + write-through stores at
// the HBM rate); the chains below keep the tile in LDS from layer to layer:
//
//   chain_head_kernel :  latent [B,52,4] -> downs.0.0 (conv 4->64 | conv 64->64 + residual_conv(latent)) -> downs.0.1
//                        (two convs, identity residual) -> downs.0.2 (Conv1d k3 s2) -> [B,26,64]            5 launches -> 1
//                        (the skip h[0] the reference pushes here is never popped, temporal.py:155,164: nothing else leaves)
//   chain_tail_kernel :  ups.1.0's second conv -> ups.1.1 -> ups.1.2 (ConvTranspose1d k4 s2) -> final_conv.0 -> final_conv.1
//                        (1x1, 64 -> 4): [B,26,64] -> eps [B,52,4]                                          6 launches -> 1
//
// Same arithmetic as conv_block.hip: v_mfma_f32_16x16x4_f32 (exact fp32), the same k order per accumulator (chunk of 32
// channels, tap, 16-channel group), two-pass GroupNorm statistics, Mish by one v_exp_f32 + one v_rcp_f32.  What differs:
//   * the A image holds ALL input channels of the tile (rows of 64 + 8 floats), written once by the previous layer's
//     epilogue; the K loop touches HBM / L2 only for its weight fragments (one coalesced 1-KiB load per 52 MFMAs);
//   * GroupNorm + Mish run on the accumulators IN REGISTERS: under the transposed M-tile mapping (M-tile m = rows
//     RPT m .. of every agent, conv_block.hip TMAP) register r of a lane is agent r, and a group's 8 channels x L rows are
//     13 registers x 8 lanes x 4 lane groups -- five cross-lane steps per pass instead of a tile exchange through LDS;
//   * a block's input (the identity residual two layers later) waits in a per-thread spill slot in the workspace.
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
    static constexpr int AEX = (AG == 4 && L == 26) ? 16 : 0;
    static constexpr int ASTR = LP * KCP + AEX;
    static constexpr int RPT = 16 / AG;                 // rows of one agent per M-tile
    static constexpr int NMT = (AG * L + 15) / 16;
    static constexpr bool RAGGED = NMT * 16 != AG * L;  // the last M-tile carries rows past the agents' ends
    static constexpr int IMG = (AG * LP + 2) * KCP + AG * AEX;
    static_assert(AG == 4 || AG == 1, "tiles of 4 agents or of 1");
    static __device__ __forceinline__ int agent(int q, int r) { return AG == 4 ? r : 0; }
    static __device__ __forceinline__ int pos(int m, int q, int r) { return AG == 4 ? 4 * m + q : 16 * m + 4 * q + r; }
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

// ---------------------------------------------------------------------------------------------------------------------
// downs.0 as one launch
// ---------------------------------------------------------------------------------------------------------------------
template <int AG>
__global__ __launch_bounds__(256, 2) void chain_head_kernel(const ChainHeadArgs p) {
    typedef Geo<52, AG> G;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xl = lds + img_floats<AG>();                  // the tile's latent rows [AG][52][4]: operand of residual_conv in stage 1
    const char* ldsb = reinterpret_cast<const char*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, n16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * AG;
    const int n = 16 * wave + n16;                       // this lane's output channel
    CSTAMP(0);
    CSTAMP_RT(14);

    // ---- latent rows -> image channel slots 0..3 (+ a compact copy); halos zeroed once: no layer writes them ----
    if (tid < AG * 52) {
        const int a = tid / 52, l = tid % 52;
        const v4f v = *reinterpret_cast<const v4f*>(p.x + ((size_t)(b0 + a) * 52 + l) * 4);
        *reinterpret_cast<v4f*>(lds + a * G::ASTR + (2 + l) * G::KCP) = v;
        *reinterpret_cast<v4f*>(xl + tid * 4) = v;
    }
    zero_halo<G>(lds, tid, 256);
    for (int i = tid; i < kSlackFloats / 4; i += 256)    // slack rows behind the image (read by ragged / stride-2 fragments, never used)
        *reinterpret_cast<v4f*>(lds + G::IMG + i * 4) = v4f{0.f, 0.f, 0.f, 0.f};

    v4f acc[G::NMT];
    const int arow = G::frag0(n16);                      // floats, relative to the agent block's row 0 (= first halo row)
    const int wbase = (2 + G::row0(q)) * G::KCP + n;     // epilogue: register (m, r) of this lane lives at wbase + roff(m, r)
    constexpr int MSTEP = G::RPT * G::KCP * 4;

    // ---- stage 0: Conv1d(4 -> 64, k5) with K folded over (tap, channel) (conv_block.hip PADC) ----
    {
        const ChainStage& st = p.st[0];
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st.wfrag), 0, 4 * 2048, 0x00020000);
        const v4f bq0 = bload16(rsw, lane * 32, wave * 2048), bq1 = bload16(rsw, lane * 32 + 16, wave * 2048);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < G::NMT; ++m) acc[m] = v4f{0.f, 0.f, 0.f, 0.f};
        const int ab = (arow + q * G::KCP) * 4;                                // row j + kk - 2 + 2 of tap kk = q, channels 0..3
        const int t4 = ((4 - q) * G::KCP + q) * 4;                             // -> (row of tap 4, channel q)
#pragma unroll
        for (int m = 0; m < G::NMT; ++m) {
            const v4f a4 = *reinterpret_cast<const v4f*>(ldsb + ab + m * MSTEP);
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[sidx], bq0[sidx], acc[m], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < G::NMT; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(*reinterpret_cast<const float*>(ldsb + ab + m * MSTEP + t4), bq1[0], acc[m], 0, 0, 0);
        float add[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) add[r] = p.cbias[(size_t)(b0 + G::agent(q, r)) * p.cb_stride + st.cb_off + n] + (p.tbias ? p.tbias[st.cb_off + n] : 0.f);
        gn_mish<G>(acc, st.bias[n], st.gamma[n], st.beta[n], add, q);
        __syncthreads();                                 // every wave has read the latent slots
        to_image<G>(acc, lds, wbase, q);
        __syncthreads();
    }
    CSTAMP(1);

    // ---- stages 1..3: Conv1d(64 -> 64, k5) + GroupNorm + Mish [+ time / cond vector] [+ residual]; one code instance ----
    v4f* keep = reinterpret_cast<v4f*>(p.keep) + (size_t)blockIdx.x * G::NMT * 256 + tid;
    typedef Geo<26, AG> GO;
    WQueue<64, 5, G::NMT> wq;
    WQueue<64, 3, GO::NMT> wqd;
    wq.prime(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.st[1].wfrag), 0, 4 * 5 * 4 * 1024, 0x00020000), lane * 16, 4, wave);
#pragma clang loop unroll(disable)
    for (int s = 1; s <= 3; ++s) {
        const ChainStage& st = p.st[s];
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st.wfrag), 0, 4 * 5 * 4 * 1024, 0x00020000);
#pragma unroll
        for (int m = 0; m < G::NMT; ++m) acc[m] = v4f{0.f, 0.f, 0.f, 0.f};
        kloop<G::KCP, MSTEP, 64, 5, G::NMT>(acc, ldsb, (arow + 4 * q) * 4, rsw, lane * 16, 4, wave, wq);
        // the next layer's first weight fragments travel under this layer's epilogue
        if (s < 3) wq.prime(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.st[s + 1].wfrag), 0, 4 * 5 * 4 * 1024, 0x00020000), lane * 16, 4, wave);
        else wqd.prime(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.st[4].wfrag), 0, 4 * 3 * 4 * 1024, 0x00020000), lane * 16, 4, wave);
        CSTAMP(2 * s);
        float add[4] = {0.f, 0.f, 0.f, 0.f};
        if (st.cb_off >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) add[r] = p.cbias[(size_t)(b0 + G::agent(q, r)) * p.cb_stride + st.cb_off + n] + (p.tbias ? p.tbias[st.cb_off + n] : 0.f);
        }
        gn_mish<G>(acc, st.bias[n], st.gamma[n], st.beta[n], add, q);
        if (st.res_kind == CHAIN_RES_LATENT) {
            // residual_conv = Conv1d(4 -> 64, k = 1) of the block input, the latent (temporal.py:32-34)
            const v4f w4 = *reinterpret_cast<const v4f*>(p.res4_w + (size_t)n * 4);
            const float b4 = p.res4_b[n];
#pragma unroll
            for (int m = 0; m < G::NMT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!G::ok(m, q, r)) continue;
                    const v4f x4 = *reinterpret_cast<const v4f*>(xl + (G::agent(q, r) * 52 + G::pos(m, q, r)) * 4);
                    // explicit FMAs: whatever the vectoriser does with the four agents of a register quad, every element
                    // sees the same operation sequence (a row's result must not depend on its place in the tile)
                    acc[m][r] += fmaf(w4[3], x4[3], fmaf(w4[2], x4[2], fmaf(w4[1], x4[1], fmaf(w4[0], x4[0], b4))));
                }
        } else if (st.res_kind == CHAIN_RES_KEPT) {
#pragma unroll
            for (int m = 0; m < G::NMT; ++m) {
                const v4f k4 = keep[m * 256];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][r] += k4[r];
            }
        }
        if (st.keep) {
#pragma unroll
            for (int m = 0; m < G::NMT; ++m) keep[m * 256] = acc[m];
        }
        __syncthreads();                                 // every wave is done reading the image this layer consumed
        to_image<G>(acc, lds, wbase, q);
        __syncthreads();
        CSTAMP(2 * s + 1);
    }

    // ---- stage 4: Conv1d(64 -> 64, k3, stride 2, pad 1) + bias -> [B,26,64] ----
    {
        const ChainStage& st = p.st[4];
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st.wfrag), 0, 4 * 3 * 4 * 1024, 0x00020000);
        v4f acd[GO::NMT];
#pragma unroll
        for (int m = 0; m < GO::NMT; ++m) acd[m] = v4f{0.f, 0.f, 0.f, 0.f};
        // input row of tap 0 of output row j = 2 j - 1 (+ 2 halo rows): lane i16 is agent i16 % AG, j = RPT m + i16 / AG
        kloop<G::KCP, 2 * MSTEP, 64, 3, GO::NMT>(acd, ldsb, ((n16 % AG) * G::ASTR + (1 + 2 * (n16 / AG)) * G::KCP + 4 * q) * 4, rsw, lane * 16, 4, wave, wqd);
        CSTAMP(8);
        const float bias = st.bias[n];
        const size_t ybase = (size_t)b0 * 26 * 64;
        const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(p.y + ybase, 0, AG * 26 * 64 * 4, 0x00020000);
#pragma unroll
        for (int m = 0; m < GO::NMT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (GO::ok(m, q, r))
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acd[m][r] + bias), rsy,
                                                          ((GO::agent(q, r) * 26 + GO::pos(m, q, r)) * 64 + n) * 4, 0, CLD_STORE_AUX);
    }
    CSTAMP(9);
    CSTAMP_RT(15);
}

template <int AG>
static hipError_t launch_chain_head_inst(const ChainHeadArgs& a, int b_pad, hipStream_t s) {
    constexpr size_t lds_bytes = sizeof(float) * (img_floats<AG>() + AG * 52 * 4);
    static_assert(2 * lds_bytes <= 160 * 1024, "two workgroups per CU");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_head_kernel<AG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (b_pad % AG) return hipErrorInvalidValue;
    hipLaunchKernelGGL(chain_head_kernel<AG>, dim3(b_pad / AG), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}
// MFMAs per wave and workgroup: the latent's conv (5 per M-tile), three k5 layers (20 (tap, group) iterations x 4 per M-tile), the
// stride-2 conv (12 iterations x 4 per M-tile of the L = 26 output)
template <int AG> static constexpr double head_mfma_per_wave() { return 5.0 * Geo<52, AG>::NMT + 3 * 80.0 * Geo<52, AG>::NMT + 48.0 * Geo<26, AG>::NMT; }
double chain_head_exec_flop(int b_pad, int agents_per_tile) {
    return agents_per_tile == 1 ? head_mfma_per_wave<1>() * 4 * 2048.0 * b_pad : head_mfma_per_wave<4>() * 4 * 2048.0 * (b_pad / 4);
}
hipError_t launch_chain_head(const ChainHeadArgs& a, int b_pad, int agents_per_tile, hipStream_t s) {
    return agents_per_tile == 1 ? launch_chain_head_inst<1>(a, b_pad, s) : launch_chain_head_inst<4>(a, b_pad, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// ups.1.0's second conv + ups.1.1 + ups.1.2 + final_conv as one launch
// ---------------------------------------------------------------------------------------------------------------------
template <int AG>
__global__ __launch_bounds__(256, 2) void chain_tail_kernel(const ChainTailArgs p) {
    typedef Geo<26, AG> G;         // stages 0..2 and the transposed conv's input
    typedef Geo<52, AG> H;         // its output, final_conv
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const char* ldsb = reinterpret_cast<const char*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, n16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * AG;
    const int n = 16 * wave + n16;

    // ---- input rows [AG agents][26][64] -> image (coalesced 256-byte rows) ----
    {
        constexpr int NPC = AG * 26 * 16, NP = (NPC + 255) / 256;
        const v4f* src = reinterpret_cast<const v4f*>(p.x + (size_t)b0 * 26 * 64);
        v4f st[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + 256 * i;
            st[i] = src[idx < NPC ? idx : tid % NPC];
        }
        zero_halo<G>(lds, tid, 256);
        for (int i = tid; i < kSlackFloats / 4; i += 256)    // rows behind the L = 26 image: read by the ragged last M-tile, never used (but keep them finite)
            *reinterpret_cast<v4f*>(lds + G::IMG + i * 4) = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int idx = tid + 256 * i;
            if (idx < NPC) {
                const int row = idx >> 4, a = row / 26, l = row - a * 26;
                *reinterpret_cast<v4f*>(lds + a * G::ASTR + (2 + l) * G::KCP + (idx & 15) * 4) = st[i];
            }
        }
        __syncthreads();
    }

    const int arow = G::frag0(n16);
    const int wbase = (2 + G::row0(q)) * G::KCP + n;
    constexpr int MSTEP = G::RPT * G::KCP * 4;
    v4f* keep = reinterpret_cast<v4f*>(p.keep) + (size_t)blockIdx.x * G::NMT * 256 + tid;
    v4f acc[G::NMT];
    WQueue<64, 5, G::NMT> wq;
    WQueue<64, 2, G::NMT> wqe, wqo;
    WQueue<64, 5, H::NMT> wqf;
    const __amdgpu_buffer_rsrc_t rse = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.up_even.wfrag), 0, 4 * 2 * 4 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.up_odd.wfrag), 0, 4 * 2 * 4 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.fin.wfrag), 0, 4 * 5 * 4 * 1024, 0x00020000);
    wq.prime(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.st[0].wfrag), 0, 4 * 5 * 4 * 1024, 0x00020000), lane * 16, 4, wave);

    // ---- stages 0..2: Conv1d(64 -> 64, k5) + GroupNorm + Mish at L = 26 (the last M-tile is ragged: dummy rows masked) ----
#pragma clang loop unroll(disable)
    for (int s = 0; s < 3; ++s) {
        const ChainStage& st = p.st[s];
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st.wfrag), 0, 4 * 5 * 4 * 1024, 0x00020000);
#pragma unroll
        for (int m = 0; m < G::NMT; ++m) acc[m] = v4f{0.f, 0.f, 0.f, 0.f};
        kloop<G::KCP, MSTEP, 64, 5, G::NMT>(acc, ldsb, (arow + 4 * q) * 4, rsw, lane * 16, 4, wave, wq);
        if (s < 2) wq.prime(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.st[s + 1].wfrag), 0, 4 * 5 * 4 * 1024, 0x00020000), lane * 16, 4, wave);
        else wqe.prime(rse, lane * 16, 4, wave);
        float add[4] = {0.f, 0.f, 0.f, 0.f};
        if (st.cb_off >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) add[r] = p.cbias[(size_t)(b0 + G::agent(q, r)) * p.cb_stride + st.cb_off + n] + (p.tbias ? p.tbias[st.cb_off + n] : 0.f);
        }
        gn_mish<G>(acc, st.bias[n], st.gamma[n], st.beta[n], add, q);
        if (st.res_kind == CHAIN_RES_TENSOR) {
            const float* rp = st.res + (size_t)b0 * 26 * 64 + n;
#pragma unroll
            for (int m = 0; m < G::NMT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (G::ok(m, q, r)) acc[m][r] += rp[(G::agent(q, r) * 26 + G::pos(m, q, r)) * 64];
        } else if (st.res_kind == CHAIN_RES_KEPT) {
#pragma unroll
            for (int m = 0; m < G::NMT; ++m) {
                const v4f k4 = keep[m * 256];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][r] += k4[r];
            }
        }
        if (st.keep) {
#pragma unroll
            for (int m = 0; m < G::NMT; ++m) keep[m * 256] = acc[m];
        }
        __syncthreads();
        to_image<G>(acc, lds, wbase, q);
        __syncthreads();
    }

    // ---- ConvTranspose1d(64 -> 64, k4, s2, p1) as two 2-tap parity convolutions: out[2j] = x[j-1] W3 + x[j] W1,
    //      out[2j+1] = x[j] W2 + x[j+1] W0 (conv_block.hip); both from the L = 26 image, then the tile becomes an L = 52 image ----
    {
        v4f ae[G::NMT], ao[G::NMT];
#pragma unroll
        for (int m = 0; m < G::NMT; ++m) { ae[m] = v4f{0.f, 0.f, 0.f, 0.f}; ao[m] = v4f{0.f, 0.f, 0.f, 0.f}; }
        wqo.prime(rso, lane * 16, 4, wave);
        kloop<G::KCP, MSTEP, 64, 2, G::NMT>(ae, ldsb, (arow + 1 * G::KCP + 4 * q) * 4, rse, lane * 16, 4, wave, wqe);    // tap 0 reads x[j - 1] = image row 2 + j - 1
        wqf.prime(rsf, lane * 16, 4, wave);
        kloop<G::KCP, MSTEP, 64, 2, G::NMT>(ao, ldsb, (arow + 2 * G::KCP + 4 * q) * 4, rso, lane * 16, 4, wave, wqo);    // tap 0 reads x[j]
        const float be = p.up_even.bias[n], bo = p.up_odd.bias[n];
        __syncthreads();                                 // the L = 26 image is dead
        zero_halo<H>(lds, tid, 256);
#pragma unroll
        for (int m = 0; m < G::NMT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!G::ok(m, q, r)) continue;
                float* o = lds + G::agent(q, r) * H::ASTR + (2 + 2 * G::pos(m, q, r)) * H::KCP + n;
                o[0] = ae[m][r] + be;
                o[H::KCP] = ao[m][r] + bo;
            }
        __syncthreads();
    }

    // ---- final_conv.0: Conv1d(64 -> 64, k5) + GroupNorm + Mish at L = 52 ----
    const int hrow = H::frag0(n16);
    constexpr int HSTEP = H::RPT * H::KCP * 4;
    {
        v4f af[H::NMT];
#pragma unroll
        for (int m = 0; m < H::NMT; ++m) af[m] = v4f{0.f, 0.f, 0.f, 0.f};
        kloop<H::KCP, HSTEP, 64, 5, H::NMT>(af, ldsb, (hrow + 4 * q) * 4, rsf, lane * 16, 4, wave, wqf);
        const float add[4] = {0.f, 0.f, 0.f, 0.f};
        gn_mish<H>(af, p.fin.bias[n], p.fin.gamma[n], p.fin.beta[n], add, q);
        __syncthreads();
        to_image<H>(af, lds, (2 + H::row0(q)) * H::KCP + n, q);
        __syncthreads();
    }

    // ---- final_conv.1: Conv1d(64 -> 4, k1): one N tile (4 of its 16 columns real); wave w takes M-tiles w, w + 4, ... ----
    {
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.head_wfrag), 0, 4 * 1024, 0x00020000);
        v4f bw[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bw[g] = bload16(rsw, lane * 16, g * 1024);
        const int ab = (hrow + 2 * H::KCP + 4 * q) * 4 + wave * HSTEP;       // the centre tap: image row 2 + j
        constexpr int NMI = (H::NMT + 3) / 4;
        v4f ah[NMI];
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) ah[mi] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mi = 0; mi < NMI; ++mi) {
            if (wave + 4 * mi >= H::NMT) continue;       // (wave-uniform)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4f a4 = *reinterpret_cast<const v4f*>(ldsb + ab + mi * 4 * HSTEP + g * 64);
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) ah[mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[sidx], bw[g][sidx], ah[mi], 0, 0, 0);
            }
        }
        if (n16 < 4) {
            const float hb = p.head_b[n16];
#pragma unroll
            for (int mi = 0; mi < NMI; ++mi) {
                const int m = wave + 4 * mi;
                if (m >= H::NMT) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pos = AG == 4 ? 4 * m + q : 16 * m + 4 * q + r;      // H::pos with a run-time M-tile
                    if (pos >= 52) continue;
                    const int b = b0 + H::agent(q, r);
                    const size_t row = (size_t)b * 52 + pos, e = row * 4 + n16;
                    const float ev = ah[mi][r] + hb;
                    if (p.eps) p.eps[e] = ev;
                    if (p.upd_x) {                       // x_{t-1} = x_t_cof x - noise_cof eps + sigma z, as head_kernel writes it
                        const float mean = p.xc * p.upd_x[e] - p.nc * ev;
                        if (p.upd_mean_out) p.upd_mean_out[e] = mean;
                        if (p.upd_x_out) {
                            float zz = 0.f;
                            if (p.sg != 0.f && b < p.B) zz = p.upd_z ? p.upd_z[e] : normal4(p.seed, p.step_salt, (unsigned)row)[n16];
                            p.upd_x_out[e] = mean + p.sg * zz;
                        }
                    }
                }
            }
        }
    }
}

template <int AG>
static hipError_t launch_chain_tail_inst(const ChainTailArgs& a, int b_pad, hipStream_t s) {
    constexpr size_t lds_bytes = sizeof(float) * img_floats<AG>();
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_tail_kernel<AG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (b_pad % AG) return hipErrorInvalidValue;
    hipLaunchKernelGGL(chain_tail_kernel<AG>, dim3(b_pad / AG), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}
// three k5 layers at L = 26, the two parity convolutions (8 iterations x 4 per M-tile each), final_conv.0 at L = 52, final_conv.1 (16 per
// M-tile, shared between the four waves)
template <int AG> static constexpr double tail_mfma_per_wave() {
    return 3 * 80.0 * Geo<26, AG>::NMT + 2 * 32.0 * Geo<26, AG>::NMT + 80.0 * Geo<52, AG>::NMT + 4.0 * Geo<52, AG>::NMT;
}
double chain_tail_exec_flop(int b_pad, int agents_per_tile) {
    return agents_per_tile == 1 ? tail_mfma_per_wave<1>() * 4 * 2048.0 * b_pad : tail_mfma_per_wave<4>() * 4 * 2048.0 * (b_pad / 4);
}
hipError_t launch_chain_tail(const ChainTailArgs& a, int b_pad, int agents_per_tile, hipStream_t s) {
    return agents_per_tile == 1 ? launch_chain_tail_inst<1>(a, b_pad, s) : launch_chain_tail_inst<4>(a, b_pad, s);
}

}  // namespace cld
