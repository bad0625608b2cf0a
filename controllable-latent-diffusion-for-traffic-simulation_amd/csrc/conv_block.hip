// conv_block.hip -- the hot kernel of the CLD denoiser on gfx950.
//
// One kernel template covers every convolution of TemporalMapUnet
// (reference: src/tbsim/models/temporal.py:16-45,122-180 and
// src/tbsim/models/diffuser_helpers.py:34-67):
//   * Conv1d(k=5,p=2) -> GroupNorm(8) -> Mish [-> + time/cond bias] [-> + residual]
//   * Conv1d(k=1) residual projections, Conv1d(k=3,s=2,p=1) down-sampling,
//     ConvTranspose1d(k=4,s=2,p=1) up-sampling (as two 2-tap parity convolutions)
// as an implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32: parity needs 1e-3
// end to end and bf16 inputs miss it by 10x).
//
// Work decomposition (see cld_kernels.h): a 256-thread workgroup owns 208 GEMM
// rows (whole agents) x NT = 16*NWN output channels; wave (nw, ks) owns 13
// M-tiles x 1 N-tile and, when KS = 2, one half of every K chunk.  A workgroup
// therefore always holds complete GroupNorm groups (all rows of an agent x whole
// channel groups), so GroupNorm + Mish are fused into the epilogue with no
// cross-workgroup reduction.
//
//   A operand  : activations, channels-last in HBM, staged per K chunk (KC input
//                channels) into a double-buffered LDS image whose rows carry a
//                2-row zero halo between agents, so the 5 taps are 5 shifted
//                ds_read_b128 of the same image and need no boundary tests.
//   B operand  : weights pre-packed on the host in MFMA fragment order; each
//                lane fetches its fragment with ONE coalesced global_load_dwordx4
//                per (tap, 16-channel group) -- weights never touch LDS.
//   k order    : lane (i, kk) holds channels 4kk..4kk+3 of a 16-channel group and
//                feeds them to 4 successive MFMAs; A and B use the same
//                permutation, so the sum over k is unchanged.
#include "cld_kernels.h"

namespace cld {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float mish_f(float x) {
    // x * tanh(softplus(x)) == x * n / (n + 2), n = e^x (e^x + 2): one exp, no cancellation.
    // v_exp_f32 / v_rcp_f32 are 1-ulp instructions: relative error ~2e-7, far inside the parity bar.
    const float e = __expf(fminf(x, 30.0f));
    const float n = e * (e + 2.0f);
    return x * n * __frcp_rn(n + 2.0f);
}

template <int W> struct VecT;
template <> struct VecT<4> { typedef v4f type; };
template <> struct VecT<2> { typedef v2f type; };

template <int L_IN, int LM, int STRIDE, int NTAPS, int KC, int NWN, int EPI, int GS, int OSTR>
__global__ __launch_bounds__(256) void conv_block_kernel(const ConvArgs p) {
    constexpr int KS = 4 / NWN;            // K split across waves
    constexpr int AG = MT / LM;            // agents per workgroup
    constexpr int LP = L_IN + 2;           // LDS rows per agent (2 halo rows shared with the neighbour)
    constexpr int AROWS = AG * LP + 2;
    constexpr int KCP = KC + 4;            // padded LDS row, floats (keeps 16-B alignment)
    constexpr int ABUF = AROWS * KCP;      // floats per A image
    constexpr int ABUFP = ABUF + KCP;      // + one dump row for the staging pieces past the tile
    constexpr int NT = 16 * NWN;
    constexpr int OP = NT + 4;             // padded row of the output tile
    constexpr int NKG = KC / 16;           // 16-channel groups per chunk
    constexpr int KGW = NKG / KS;          // groups per wave per chunk
    static_assert(NWN * KS == 4, "4 waves");
    static_assert(KGW >= 1 && KGW * KS == NKG, "K split must divide the chunk");
    static_assert(AG * LM == MT, "whole agents per tile");
    constexpr int IN_ROWS = AG * L_IN;
    constexpr int PPR = KC / 4;            // 16-byte pieces per staged row
    constexpr int NPC = IN_ROWS * PPR;
    constexpr int NPIECE = (NPC + 255) / 256;
    constexpr int NIT = NTAPS * KGW;       // (tap, group) iterations per chunk per wave

    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = wave % NWN;
    const int ks = wave / NWN;
    const int b0 = blockIdx.x * AG;
    const int ntile_g = blockIdx.y * NWN + nw;
    const int ntn = p.c_out >> 4;
    const int nchunk = (p.c1_pad + p.c2) / KC;

    // ---- staging map: piece i of this thread -> (global row, LDS offset) ------------------
    // Every piece loads and stores unconditionally (no divergent branches, so the compiler can count
    // its vmcnt waits): pieces past the tile read this thread's piece 0 again and land in a dump row
    // behind the image.
    const int pc4 = (tid % PPR) * 4;
    int soff[NPIECE];
    int grow[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int idx = tid + 256 * i;
        const bool ok = idx < NPC;
        const int r = ok ? idx / PPR : tid / PPR;
        const int a = r / L_IN;
        const int l = r - a * L_IN;
        soff[i] = ok ? (2 + a * LP + l) * KCP + pc4 : AROWS * KCP + pc4;
        grow[i] = b0 * L_IN + r;
    }
    v4f st[NPIECE];
    auto load_chunk = [&](int c) {
        const int cv = c * KC;                          // virtual input channel of this chunk
        const bool first = cv < p.c1_pad;               // wave-uniform: which source feeds the chunk
        const float* src = first ? p.x1 : p.x2;
        const int stride = first ? p.c1_real : p.c2;
        int cb = (first ? cv : cv - p.c1_pad) + pc4;
        const bool real = !first || cb < p.c1_real;     // only the 4-channel latent has a padded chunk
        cb = real ? cb : 0;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const v4f v = *reinterpret_cast<const v4f*>(src + (size_t)grow[i] * stride + cb);
            st[i] = real ? v : v4f{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_chunk = [&](int buf) {
        float* A = lds + buf * ABUFP;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) *reinterpret_cast<v4f*>(A + soff[i]) = st[i];
    };

    // ---- A fragment offsets: lane (i = lane&15, kk = lane>>4) of M-tile m --------------------
    int aoff[NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
        const int r = 16 * m + (lane & 15);
        const int a = r / LM;
        const int j = r - a * LM;
        aoff[m] = (2 + a * LP + STRIDE * j + p.off0) * KCP + 4 * (lane >> 4) + 16 * ks;
    }
    const float* wlane = p.wfrag + ((size_t)ntile_g * 64 + lane) * 4;
    const size_t wstride = (size_t)ntn * 256;      // floats per (chunk, tap, group) slab
    auto wptr = [&](int c, int it) {
        const int t = it / KGW;
        const int kg = ks + KS * (it % KGW);
        return reinterpret_cast<const v4f*>(wlane + (size_t)((c * NTAPS + t) * NKG + kg) * wstride);
    };

    v4f acc[NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) acc[m] = v4f{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: zero both images (halo rows stay zero for the whole kernel), stage chunk 0 ----
    load_chunk(0);
    // B fragments run two (tap, group) iterations ahead of the MFMAs that consume them
    auto wptr_lin = [&](int c, int it) { return wptr(c + it / NIT, it % NIT); };   // `it` may run past the chunk
    v4f bq0 = *wptr(0, 0);
    v4f bq1 = (1 / NIT < nchunk) ? *wptr_lin(0, 1) : bq0;
    // zero the halo rows of both images: rows 0,1 and the 2 rows behind every agent
    for (int i = tid; i < 2 * (AG + 1) * 2 * (KCP / 4); i += 256) {
        const int q = i % (KCP / 4), hr = (i / (KCP / 4)) % (2 * (AG + 1)), buf = i / ((KCP / 4) * 2 * (AG + 1));
        const int g = hr >> 1;                          // gap index: 0 = leading rows, g>0 = behind agent g-1
        const int row = (g == 0 ? 0 : g * LP) + (hr & 1);
        *reinterpret_cast<v4f*>(lds + buf * ABUFP + row * KCP + q * 4) = v4f{0.f, 0.f, 0.f, 0.f};
    }
    store_chunk(0);
    __syncthreads();

    for (int c = 0; c < nchunk; ++c) {
        const float* A = lds + (c & 1) * ABUFP;
        const bool more = (c + 1 < nchunk);
        v4f af[2][NMT];
#pragma unroll
        for (int m = 0; m < NMT; ++m) af[0][m] = *reinterpret_cast<const v4f*>(A + aoff[m]);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const v4f bcur = bq0;
            bq0 = bq1;
            if (c + (it + 2) / NIT < nchunk) bq1 = *wptr_lin(c, it + 2);
            if (it == 0 && more) load_chunk(c + 1);      // next chunk's activations: in flight under this chunk's MFMAs
            if (it + 1 < NIT) {
                const int t1 = (it + 1) / KGW, g1 = (it + 1) % KGW;
#pragma unroll
                for (int m = 0; m < NMT; ++m)
                    af[(it + 1) & 1][m] = *reinterpret_cast<const v4f*>(A + aoff[m] + t1 * KCP + 16 * KS * g1);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < NMT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[it & 1][m][s], bcur[s], acc[m], 0, 0, 0);
        }
        if (more) store_chunk((c + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: accumulators -> LDS tile [208][OP] (aliases the A images) -----------------
    float* O = lds;
    {
        const int col = nw * 16 + (lane & 15);
        const int rb = 4 * (lane >> 4);
        if (KS == 1 || ks == 0) {
#pragma unroll
            for (int m = 0; m < NMT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) O[(16 * m + rb + r) * OP + col] = acc[m][r];
        }
        if (KS > 1) {
            __syncthreads();
            if (ks == 1) {   // same element, same lane position of the partner wave: plain read-modify-write
#pragma unroll
                for (int m = 0; m < NMT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) O[(16 * m + rb + r) * OP + col] += acc[m][r];
            }
        }
    }
    __syncthreads();

    // ---- per-(agent, group) epilogue: TPP lanes x 13 vectors of VW channels -----------------
    constexpr int VW = NT / 16;
    constexpr int NG = NT / GS;
    constexpr int PAIRS = AG * NG;
    constexpr int TPP = 256 / PAIRS;
    constexpr int VPR = GS / VW;
    constexpr int RPI = TPP / VPR;
    static_assert(PAIRS * TPP == 256 && VPR * RPI == TPP && RPI * 13 == LM, "epilogue mapping");
    typedef typename VecT<VW>::type vec_t;

    const int pair = tid / TPP, q = tid % TPP;
    const int a = pair / NG, g = pair % NG;
    const int ch = g * GS + (q % VPR) * VW;          // channel within the tile
    const int n = blockIdx.y * NT + ch;              // global output channel
    const int jr = q / VPR;

    float v[13][VW];
    const vec_t bias = *reinterpret_cast<const vec_t*>(p.bias + n);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int j = RPI * i + jr;
        const vec_t o = *reinterpret_cast<const vec_t*>(O + (a * LM + j) * OP + ch);
#pragma unroll
        for (int e = 0; e < VW; ++e) v[i][e] = o[e] + bias[e];
    }

    if (EPI == EPI_GN_MISH) {
        // GroupNorm over (GS channels x LM rows) of one agent: two-pass, biased variance, eps 1e-5
        // (torch.nn.GroupNorm as used in diffuser_helpers.py:61).
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 13; ++i)
#pragma unroll
            for (int e = 0; e < VW; ++e) s += v[i][e];
#pragma unroll
        for (int o = 1; o < TPP; o <<= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / (float)(GS * LM));
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 13; ++i)
#pragma unroll
            for (int e = 0; e < VW; ++e) { const float d = v[i][e] - mean; ss += d * d; }
#pragma unroll
        for (int o = 1; o < TPP; o <<= 1) ss += __shfl_xor(ss, o);
        const float rstd = 1.0f / sqrtf(ss * (1.0f / (float)(GS * LM)) + 1e-5f);
        const vec_t gam = *reinterpret_cast<const vec_t*>(p.gamma + n);
        const vec_t bet = *reinterpret_cast<const vec_t*>(p.beta + n);
        float add[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) add[e] = 0.f;
        if (p.cbias) {
            const vec_t cbv = *reinterpret_cast<const vec_t*>(p.cbias + (size_t)(b0 + a) * p.cb_stride + n);
#pragma unroll
            for (int e = 0; e < VW; ++e) add[e] += cbv[e];
        }
        if (p.tbias) {
            const vec_t tbv = *reinterpret_cast<const vec_t*>(p.tbias + n);
#pragma unroll
            for (int e = 0; e < VW; ++e) add[e] += tbv[e];
        }
#pragma unroll
        for (int i = 0; i < 13; ++i)
#pragma unroll
            for (int e = 0; e < VW; ++e)
                v[i][e] = mish_f((v[i][e] - mean) * rstd * gam[e] + bet[e]) + add[e];
    }

#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int j = RPI * i + jr;
        const size_t idx = ((size_t)(b0 + a) * p.ly + (OSTR * j + p.orow0)) * p.c_out + n;
        vec_t o;
        if (p.res) {
            const vec_t rv = *reinterpret_cast<const vec_t*>(p.res + idx);
#pragma unroll
            for (int e = 0; e < VW; ++e) o[e] = v[i][e] + rv[e];
        } else {
#pragma unroll
            for (int e = 0; e < VW; ++e) o[e] = v[i][e];
        }
        *reinterpret_cast<vec_t*>(p.y + idx) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------
template <int L_IN, int LM, int STRIDE, int NTAPS, int KC, int NWN, int EPI, int GS, int OSTR>
static hipError_t launch_inst(const ConvArgs& a, int b_pad, hipStream_t s) {
    constexpr int AG = MT / LM;
    constexpr int ABUF = (AG * (L_IN + 2) + 2 + 1) * (KC + 4);   // image + dump row
    constexpr int OTILE = MT * (16 * NWN + 4);     // the epilogue's output tile aliases the A images
    constexpr size_t lds_bytes = sizeof(float) * (size_t)(2 * ABUF > OTILE ? 2 * ABUF : OTILE);
    auto kern = conv_block_kernel<L_IN, LM, STRIDE, NTAPS, KC, NWN, EPI, GS, OSTR>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    dim3 grid(b_pad / AG, a.c_out / (16 * NWN), 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

#define CLD_CONV_INSTANCES(X)                        \
    X(52, 52, 1, 5, 16, 4, EPI_GN_MISH, 8, 1)        \
    X(52, 52, 1, 5, 32, 4, EPI_GN_MISH, 8, 1)        \
    X(26, 26, 1, 5, 32, 4, EPI_GN_MISH, 16, 1)       \
    X(13, 13, 1, 5, 32, 4, EPI_GN_MISH, 32, 1)       \
    X(13, 13, 1, 5, 32, 2, EPI_GN_MISH, 16, 1)       \
    X(26, 26, 1, 5, 32, 2, EPI_GN_MISH, 8, 1)        \
    X(52, 52, 1, 1, 16, 4, EPI_BIAS, 8, 1)           \
    X(26, 26, 1, 1, 32, 4, EPI_BIAS, 16, 1)          \
    X(13, 13, 1, 1, 32, 4, EPI_BIAS, 32, 1)          \
    X(13, 13, 1, 1, 32, 2, EPI_BIAS, 16, 1)          \
    X(26, 26, 1, 1, 32, 2, EPI_BIAS, 8, 1)           \
    X(52, 26, 2, 3, 32, 2, EPI_BIAS, 8, 1)           \
    X(26, 13, 2, 3, 32, 2, EPI_BIAS, 16, 1)          \
    X(13, 13, 1, 2, 32, 4, EPI_BIAS, 16, 2)          \
    X(26, 26, 1, 2, 32, 4, EPI_BIAS, 8, 2)

static inline bool geom_is(const ConvGeom& g, int l_in, int lm, int stride, int ntaps, int kc, int nwn,
                           int epi, int gs, int ostr) {
    return g.l_in == l_in && g.lm == lm && g.stride == stride && g.ntaps == ntaps && g.kc == kc &&
           g.nwn == nwn && g.epi == epi && g.gs == gs && g.ostr == ostr;
}

bool conv_geom_supported(const ConvGeom& g) {
#define X(a, b, c, d, e, f, h, i, j) if (geom_is(g, a, b, c, d, e, f, h, i, j)) return true;
    CLD_CONV_INSTANCES(X)
#undef X
    return false;
}

hipError_t launch_conv(const ConvGeom& g, const ConvArgs& a, int b_pad, int /*grid_z_index*/, hipStream_t s) {
#define X(a_, b_, c_, d_, e_, f_, h_, i_, j_) \
    if (geom_is(g, a_, b_, c_, d_, e_, f_, h_, i_, j_)) return launch_inst<a_, b_, c_, d_, e_, f_, h_, i_, j_>(a, b_pad, s);
    CLD_CONV_INSTANCES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace cld
