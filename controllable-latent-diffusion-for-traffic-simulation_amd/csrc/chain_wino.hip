// chain_wino.hip -- the layer chains of conv_chain.hip with their 64 -> 64 Conv1d(k = 5) layers in Winograd / Cook-Toom F(4, 5) form.
//
// conv_chain.hip keeps a four-agent tile of the 64-channel levels of TemporalMapUnet (reference: src/tbsim/models/temporal.py:148-176;
// Conv1dBlock, src/tbsim/models/diffuser_helpers.py:34-67) in LDS from layer to layer.  Its stamps show the launches bound by MFMA
// issue (two workgroups per CU keep the fp32 MFMA pipe ~94 % busy), so what is left to take out is the arithmetic -- the step
// wino1d_kernels.hip took for the L = 13 / 26 levels: four outputs of a 5-tap correlation from 8 multiplies at the points
// {0, +-1, +-2, +-1/2, inf} (matrices: wino1d_kernels.hip header; G g formed in double at cld_finalize).  Per k5 layer a wave issues
// 8 x 4 x 16 = 512 MFMAs for four agents at L = 52 (13 tiles per agent, 52 of 64 GEMM rows live) instead of 1,040, and 256 instead of
// 560 at L = 26 (7 tiles per agent, 28 of 32 rows).
//
// What makes the form fit a chain: the output transform leaves a lane with FOUR CONSECUTIVE CHANNELS of the four outputs of one
// (agent, tile) -- and the input transform of the NEXT layer needs, for those channels, exactly these four rows plus two rows of each
// neighbouring tile, which sit in the neighbouring LANES (GEMM row = 13 agent + tile: lane i16 +- 1 of the 16-lane DPP row).  So between
// two Winograd layers the activations never take the spatial form: epilogue (A^T, GroupNorm, Mish, vectors, residual) in registers ->
// two DPP row rotations per value for the halo -> B^T in registers -> V[xi][row][64 channels] in LDS, which is the next layer's MFMA
// operand.  V of all 8 xi is 128 KB at L = 52: a layer runs in two phases of four xi (64 KB resident, two workgroups per CU), the
// second half of V waiting in registers.  The layers that are not k5 / stride 1 take the spatial image as before: the latent's conv
// (evaluated straight into the (agent, tile) layout), the stride-2 conv, the transposed conv (evaluated from the L = 26 image straight
// into the L = 52 (agent, tile) layout: the four outputs of tile t are parities 0 | 1 of input rows 2 t and 2 t + 1) and final_conv.1.
//
// The kernels are templates over the agents per tile (WGeo<L, AG>): 4 (64 KB of V, two workgroups per CU), 2 (32 KB, three per CU: what the
// library takes above 944 rows per launch set) and 1 (all of V resident in 32 KB: up to 944 rows).  Every agent's sums are formed in the same order
// whatever the tile: the three sizes agree bit for bit.
//
// A row's result must not depend on its place in the workgroup (tests: shuffled batches reproduce their rows bit for bit): no implicit
// contraction in this file -- every fused multiply-add below is written as one.
#pragma clang fp contract(off)
#include "chain_common.h"

namespace cld {

namespace {

typedef unsigned int u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4f fma4(const v4f a, const float s, const v4f b) { return __builtin_elementwise_fma(a, v4f{s, s, s, s}, b); }      // a s + b

// Mish = x n / (n + 2), n = e^x (e^x + 2) (conv_block.hip mish_f) on register pairs
__device__ __forceinline__ v2f mish2w(const v2f x) {
    const v2f c = v2f{fminf(x[0], 30.0f), fminf(x[1], 30.0f)} * v2f{1.4426950408889634f, 1.4426950408889634f};
    const v2f e = {__builtin_amdgcn_exp2f(c[0]), __builtin_amdgcn_exp2f(c[1])};
    const v2f two = {2.0f, 2.0f};
    const v2f n = e * (e + two);
    const v2f d = n + two;
    const v2f r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return (x * n) * r;
}
__device__ __forceinline__ v4f mish4w(const v4f x) {
    const v2f lo = mish2w(v2f{x[0], x[1]}), hi = mish2w(v2f{x[2], x[3]});
    return v4f{lo[0], lo[1], hi[0], hi[1]};
}

// value of the lane one to the left / right in the 16-lane DPP row, with wrap-around (row_ror:1 / row_ror:15)
__device__ __forceinline__ float ror1(const float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
}
__device__ __forceinline__ float ror15(const float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x12F, 0xf, 0xf, false));
}
__device__ __forceinline__ v4f ror1v(const v4f v) {
    const float a = v[0], b = v[1], c = v[2], d = v[3];      // (scalars first: __builtin_bit_cast of a vector ELEMENT reads element 0)
    return v4f{ror1(a), ror1(b), ror1(c), ror1(d)};
}
__device__ __forceinline__ v4f ror15v(const v4f v) {
    const float a = v[0], b = v[1], c = v[2], d = v[3];
    return v4f{ror15(a), ror15(b), ror15(c), ror15(d)};
}
__device__ __forceinline__ float swap16_sum(const float s) {      // s + the value of lane ^ 16
    const auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s), __builtin_bit_cast(unsigned, s), false, false);
    const unsigned a16 = r16[0], b16 = r16[1];
    return __builtin_bit_cast(float, a16) + __builtin_bit_cast(float, b16);
}

// Geometry of the (agent, tile) layout of a four-agent tile at L rows: GEMM row = TPA agent + tile; lane (i16, kk) of wave w, M-tile m:
// row 16 m + i16, channels 16 w + 4 kk .. + 3, the four outputs 4 tile + o.
// Tiles of AG agents: 4 (64 KB of V resident, two workgroups per CU: the form of rounds 4's first measurements), 2 or 1 (32 KB of V, <= 170
// registers: three workgroups per CU -- a chain wave spends about a third of its life issuing MFMAs, so a third wave per SIMD fills more of the
// pipe than the doubled / quadrupled U traffic per MFMA costs; AG = 1 is also the small-batch tile).
template <int L_, int AG_ = 4>
struct WGeo {
    static constexpr int L = L_, AG = AG_, TPA = (L + 3) / 4, ROWS = AG * TPA, NM = (ROWS + 15) / 16, R = 16 * NM;
    static constexpr int XIB = R * 256;                   // bytes of one xi image: R rows x 64 channels
    static constexpr int VBYTES = AG == 4 ? 65536 : 32768;
    static constexpr bool TWO_PHASE = 8 * XIB > VBYTES;   // V of four xi resident at a time
    static constexpr int NS = TWO_PHASE ? 4 : 8;          // xi images resident at once
    static constexpr int WD = NM == 4 ? 5 : NM == 2 ? 10 : 12;      // U fragments in flight: an item is 4 NM MFMAs = 128 NM cycles, a fragment comes from L2
    static_assert(NS * XIB <= VBYTES, "V fits its region");
};
template <int AG> constexpr int v_floats() { return AG == 4 ? 16384 : 8192; }      // the V region
constexpr int kGnFloats = 1024;                           // GroupNorm row sums behind it: [pass 2][wave 4][group of the wave 2][row 64]
// transform point of image slot s of phase PH (0 | 1: the halves of a two-phase layer; 2: all eight)
__device__ __forceinline__ constexpr int xi_of(int ph, int s) {
    return ph == 0 ? s + 1 : ph == 1 ? (s == 0 ? 5 : s == 1 ? 6 : s == 2 ? 0 : 7) : (s < 6 ? s + 1 : s == 6 ? 0 : 7);
}

// what a lane knows about its rows
template <class W>
struct Rows {
    int al[W::NM], tl[W::NM];       // agent (0..3) and tile of row 16 m + i16 (idle rows: agent 0, tile 0)
    bool lv[W::NM];
    __device__ __forceinline__ void init(const int i16) {
#pragma unroll
        for (int m = 0; m < W::NM; ++m) {
            const int r = 16 * m + i16;
            lv[m] = r < W::ROWS;
            const int rr = lv[m] ? r : 0;
            al[m] = rr / W::TPA;
            tl[m] = rr - al[m] * W::TPA;
        }
    }
};

// U fragments (G g in pack_conv_weights layout, the 8 points as taps) of items (chunk c = i / NS, slot s = i % NS) of phase PH
template <class W, int PH>
__device__ __forceinline__ v4f uload(const __amdgpu_buffer_rsrc_t rsw, const int wvoff, const int wsoff, const int i) {
    constexpr int NS = W::NS;
    const int c = i / NS, xi = xi_of(PH, i % NS);
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff, wsoff + ((c * 8 + xi) * 4) * 1024, 0));
}
template <class W, int PH>
__device__ __forceinline__ void uprime(v4f (&uq)[W::WD], const float* ufrag, const int wvoff, const int wsoff) {
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ufrag), 0, 4 * 8 * 4 * 1024, 0x00020000);
#pragma unroll
    for (int i = 0; i < W::WD; ++i) uq[i] = uload<W, PH>(rsw, wvoff, wsoff, i);
}

// one phase of a layer: acc[xi][m] += sum over the 64 input channels of U_xi (A operand: channels 16 w .. of the output) x V_xi (B operand:
// the rows), for the NS xi whose images are resident.  No barrier inside: all four 16-channel chunks of V are in LDS.
template <class W, int PH>
__device__ __forceinline__ void wino_phase(v4f (&acc)[8][W::NM], const char* ldsb, const int (&aoff)[4], v4f (&uq)[W::WD], const float* ufrag,
                                           const int wvoff, const int wsoff) {
    constexpr int NM = W::NM, NS = W::NS, NI = 4 * NS, WD = W::WD;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ufrag), 0, 4 * 8 * 4 * 1024, 0x00020000);
    auto frag = [&](const int it) {      // it = item * NM + m
        const int i = it / NM, m = it % NM, c = i / NS, s = i % NS;
        return *reinterpret_cast<const v4f*>(ldsb + aoff[c] + s * W::XIB + m * 4096);
    };
    v4f ar[3];
    ar[0] = frag(0);
    ar[1] = frag(1);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int xi = xi_of(PH, i % NS);
        const v4f bcur = uq[i % WD];
        if (i + WD < NI) uq[i % WD] = uload<W, PH>(rsw, wvoff, wsoff, i + WD);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int it = NM * i + m;
            if (it + 2 < NI * NM) ar[(it + 2) % 3] = frag(it + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[xi][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[e], ar[it % 3][e], acc[xi][m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Y (a lane's four channels of the four outputs of its rows; outputs past the agent's end and idle rows are zero) -> B^T d:
// d = [left tile's outputs 2, 3 | own 0..3 | right tile's outputs 0, 1], zero beyond the agent's ends (the convolution's padding).
// Slots 0..3 of a two-phase layer (xi 1..4) and all eight of a one-phase layer go to LDS at `vb` (+ slot * XIB + m * 4096); a
// two-phase layer's xi 5, 6, 0, 7 come back in vkeep for the second phase.
template <class W>
__device__ __forceinline__ void to_winograd(const v4f (&Y)[W::NM][4], const Rows<W>& rw, const int i16, char* vb, v4f (&vkeep)[4][W::NM]) {
    constexpr int NM = W::NM;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        // row 16 m + i16 - 1 is lane i16 - 1 of this M-tile, or lane 15 of the previous one (likewise to the right): the SOURCE lane picks
        // which M-tile it hands over -- lane 15 is read by lane 0 only -- so a value costs one select and one DPP rotation, no storage
        const bool has_l = rw.lv[m] && rw.tl[m] != 0, has_r = rw.lv[m] && rw.tl[m] != W::TPA - 1;
        v4f d0 = ror1v((m > 0 && i16 == 15) ? Y[m > 0 ? m - 1 : 0][2] : Y[m][2]), d1 = ror1v((m > 0 && i16 == 15) ? Y[m > 0 ? m - 1 : 0][3] : Y[m][3]);
        v4f d6 = ror15v((m + 1 < NM && i16 == 0) ? Y[m + 1 < NM ? m + 1 : m][0] : Y[m][0]), d7 = ror15v((m + 1 < NM && i16 == 0) ? Y[m + 1 < NM ? m + 1 : m][1] : Y[m][1]);
        d0 = has_l ? d0 : zero; d1 = has_l ? d1 : zero;
        d6 = has_r ? d6 : zero; d7 = has_r ? d7 : zero;
        const v4f d2 = Y[m][0], d3 = Y[m][1], d4 = Y[m][2], d5 = Y[m][3];
        auto st = [&](const int slot, const v4f v) { *reinterpret_cast<v4f*>(vb + slot * W::XIB + m * 4096) = v; };
        {
            const v4f e = fma4(d4, -4.25f, d2 + d6), o = fma4(d3, -4.25f, d1 + d5);
            st(0, e + o); st(1, e - o);                                                   // xi 1, 2
        }
        {
            const v4f e = fma4(d2, 0.25f, fma4(d4, -1.25f, d6)), o = fma4(d1, 0.5f, fma4(d3, -2.5f, 2.0f * d5));
            st(2, e + o); st(3, e - o);                                                   // xi 3, 4
        }
        {
            const v4f e = fma4(d2, 4.0f, fma4(d4, -5.0f, d6)), o = fma4(d1, 2.0f, fma4(d3, -2.5f, 0.5f * d5));
            const v4f v0 = fma4(d2 - d4, 5.25f, d6 - d0), v7 = fma4(d3 - d5, 5.25f, d7 - d1);
            if constexpr (W::TWO_PHASE) { vkeep[0][m] = e + o; vkeep[1][m] = e - o; vkeep[2][m] = v0; vkeep[3][m] = v7; }
            else { st(4, e + o); st(5, e - o); st(6, v0); st(7, v7); }
        }
    }
}

// Output transform A^T + conv bias: acc[xi][m] -> Y[m][o]
template <int NM>
__device__ __forceinline__ void out_transform(const v4f (&acc)[8][NM], const v4f bias, v4f (&Y)[NM][4]) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const v4f p12 = acc[1][m] + acc[2][m], m12 = acc[1][m] - acc[2][m];
        const v4f p34 = acc[3][m] + acc[4][m], m34 = acc[3][m] - acc[4][m];
        const v4f p56 = acc[5][m] + acc[6][m], m56 = acc[5][m] - acc[6][m];
        Y[m][0] = ((acc[0][m] + p12) + (p34 + p56)) + bias;
        Y[m][1] = fma4(m56, 0.5f, fma4(m34, 2.0f, m12)) + bias;
        Y[m][2] = fma4(p56, 0.25f, fma4(p34, 4.0f, p12)) + bias;
        Y[m][3] = (fma4(m56, 0.125f, fma4(m34, 8.0f, m12)) + acc[7][m]) + bias;
    }
}

// GroupNorm(8 channels x L rows per agent, eps 1e-5, biased variance, two passes; diffuser_helpers.py:61) + Mish + per-agent vector on
// Y (conv bias included).  A group is two channel quads (lanes kk, kk ^ 1: one permlane16 swap) x the agent's TPA rows, which are
// spread over lanes and M-tiles: the per-row sums go through wave-private LDS (a wave holds whole groups), one lane per (agent, group)
// adds them in tile order and the totals come back by ds_bpermute.  Outputs past the agent's end / idle rows leave as zeros.
template <class W>
__device__ __forceinline__ void gn_mish_rows(v4f (&Y)[W::NM][4], const Rows<W>& rw, const v4f gam, const v4f bet, const v4f (&add)[W::NM],
                                             float* gnw /* this wave's [pass 2][group 2][64] */, const int i16, const int kk) {
    constexpr int NM = W::NM, L = W::L, TPA = W::TPA;
    auto live = [&](const int m, const int o) { return rw.lv[m] && 4 * rw.tl[m] + o < L; };
    const float inv = 1.0f / (float)(8 * L);
    const int gsel = kk >> 1;
    auto totals = [&](float (&v)[NM], float* rowsum) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const float sv = swap16_sum(v[m]);
            if ((kk & 1) == 0) rowsum[gsel * 64 + 16 * m + i16] = sv;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the wave's own LDS writes have landed (a wave's LDS operations complete in order)
        __builtin_amdgcn_wave_barrier();
        float tot = 0.f;                         // lane (i16, kk): agent i16 & 3 of group kk >> 1, its tiles in order
        const float* src = rowsum + gsel * 64 + (i16 & 3) * TPA;
#pragma unroll
        for (int j = 0; j < TPA; ++j) tot += src[j];
#pragma unroll
        for (int m = 0; m < NM; ++m)
            v[m] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((rw.al[m] | (kk << 4)) << 2, __builtin_bit_cast(int, tot)));
        __builtin_amdgcn_wave_barrier();
    };
    float mean[NM], s2[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        float sv = 0.f;
#pragma unroll
        for (int o = 0; o < 4; ++o) {          // (selects, not branches: the guards differ from lane to lane)
            const float c = (Y[m][o][0] + Y[m][o][1]) + (Y[m][o][2] + Y[m][o][3]);
            sv += live(m, o) ? c : 0.f;
        }
        mean[m] = sv;
    }
    totals(mean, gnw);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        mean[m] *= inv;
        float sv = 0.f;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const v4f dv = Y[m][o] - mean[m];
            const float c = __builtin_fmaf(dv[0], dv[0], dv[1] * dv[1]) + __builtin_fmaf(dv[2], dv[2], dv[3] * dv[3]);
            sv += live(m, o) ? c : 0.f;
        }
        s2[m] = sv;
    }
    totals(s2, gnw + 128);
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const v4f sc = (1.0f / sqrtf(s2[m] * inv + 1e-5f)) * gam;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const v4f x = __builtin_elementwise_fma(Y[m][o] - mean[m], sc, bet);
            Y[m][o] = live(m, o) ? mish4w(x) + add[m] : zero;
        }
    }
}

// a k5 layer in the (agent, tile) layout: Y -> V (LDS) -> 8 transform-domain products -> acc.  Barriers: before V is overwritten (every wave
// has finished the products that read it) and after it is written.  `uq` holds the first U fragments of the layer on entry; the caller
// primes the next layer's once the accumulators are dead.
template <class W>
__device__ __forceinline__ void wino_layer(const v4f (&Y)[W::NM][4], v4f (&acc)[8][W::NM], const Rows<W>& rw, char* ldsb, const int (&aoff)[4],
                                           const int wofs, const int i16, v4f (&uq)[W::WD], const float* ufrag, const int wvoff, const int wsoff) {
    constexpr int NM = W::NM;
    v4f vkeep[4][NM];
    __syncthreads();                                     // V (or the spatial image in its place) is dead in every wave
    to_winograd<W>(Y, rw, i16, ldsb + wofs, vkeep);
    __syncthreads();
    if constexpr (W::TWO_PHASE) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[xi_of(0, s)][m] = v4f{0.f, 0.f, 0.f, 0.f};
        wino_phase<W, 0>(acc, ldsb, aoff, uq, ufrag, wvoff, wsoff);
        uprime<W, 1>(uq, ufrag, wvoff, wsoff);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                *reinterpret_cast<v4f*>(ldsb + wofs + s * W::XIB + m * 4096) = vkeep[s][m];
                acc[xi_of(1, s)][m] = v4f{0.f, 0.f, 0.f, 0.f};
            }
        __syncthreads();
        wino_phase<W, 1>(acc, ldsb, aoff, uq, ufrag, wvoff, wsoff);
    } else {
#pragma unroll
        for (int xi = 0; xi < 8; ++xi)
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[xi][m] = v4f{0.f, 0.f, 0.f, 0.f};
        wino_phase<W, 2>(acc, ldsb, aoff, uq, ufrag, wvoff, wsoff);
    }
}

// Y -> rows of the spatial image (conv_chain.hip Geo layout) for the layers that read it (stride-2 conv, transposed conv, final_conv.1)
template <class W, class G>
__device__ __forceinline__ void to_image_rows(const v4f (&Y)[W::NM][4], const Rows<W>& rw, float* lds, const int n4) {
#pragma unroll
    for (int m = 0; m < W::NM; ++m)
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (rw.lv[m] && 4 * rw.tl[m] + o < W::L)
                *reinterpret_cast<v4f*>(lds + rw.al[m] * G::ASTR + (2 + 4 * rw.tl[m] + o) * G::KCP + n4) = Y[m][o];
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// downs.0 as one launch, k5 layers in Winograd form
// ---------------------------------------------------------------------------------------------------------------------
template <int AG>
__global__ __launch_bounds__(256, AG == 4 ? 2 : 3) void chain_head_wino_kernel(const ChainHeadArgs p) {
    typedef WGeo<52, AG> W;
    typedef Geo<52, AG> G;
    typedef Geo<26, AG> GO;
    constexpr int NM = W::NM;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* gn = lds + v_floats<AG>();
    float* xl = lds + v_floats<AG>() + kGnFloats;        // latent rows [AG agents][2 + 52 + 2][4], the halo rows zero
    char* ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * AG;
    const int n4 = 16 * wave + 4 * kk;                   // this lane's four output channels
    Rows<W> rw;
    rw.init(i16);
    float* gnw = gn + wave * 256;
    CSTAMP(0);
    CSTAMP_RT(14);
#ifdef CLD_CHAIN_STAGGER      // experiment builds only: second-slot workgroups of the first generation start n x 8k cycles late (no gain)
    if (((blockIdx.x >> 8) & 1) && blockIdx.x < 512)
        for (int k_ = 0; k_ < CLD_CHAIN_STAGGER; ++k_) __builtin_amdgcn_s_sleep(127);
#endif

    // V addressing: rows of 64 channels = 16 slots of 16 bytes, slot s of row r at s ^ (r & 15): the four 16-lane groups of a ds_read_b128
    // then touch every bank once.  Fragment of chunk c: slot 4 c + kk; this lane's own channels: slot 4 wave + kk.
    int aoff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) aoff[c] = i16 * 256 + (((4 * c + kk) ^ i16) << 4);
    const int wofs = i16 * 256 + (((4 * wave + kk) ^ i16) << 4);
    const int wvoff = lane * 16, wsoff = wave * 1024;

    v4f uq[W::WD];
    uprime<W, W::TWO_PHASE ? 0 : 2>(uq, p.st[1].ufrag, wvoff, wsoff);      // (the first phase of the layer form this tile takes)

    // ---- latent rows -> xl ----
    if (tid < AG * 56) {
        const int a = tid / 56, l = tid % 56 - 2;
        v4f v = {0.f, 0.f, 0.f, 0.f};
        if (l >= 0 && l < 52) v = *reinterpret_cast<const v4f*>(p.x + ((size_t)(b0 + a) * 52 + l) * 4);
        *reinterpret_cast<v4f*>(xl + tid * 4) = v;
    }
    v4f Y[NM][4];
    v4f add[NM];
    auto load_add = [&](const int cb_off) {
        const v4f tb = (p.tbias && cb_off >= 0) ? *reinterpret_cast<const v4f*>(p.tbias + cb_off + n4) : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            add[m] = tb;
            if (cb_off >= 0) add[m] += *reinterpret_cast<const v4f*>(p.cbias + (size_t)(b0 + rw.al[m]) * p.cb_stride + cb_off + n4);
        }
    };

    // ---- stage 0: Conv1d(4 -> 64, k5), K folded over (tap, channel) (conv_block.hip PADC), straight into the (agent, tile) layout:
    //      the weights are the A operand; output o of tile t reads latent rows 4 t + o - 2 + tap ----
    {
        const ChainStage& st = p.st[0];
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st.wfrag), 0, 4 * 2048, 0x00020000);
        const v4f bq0 = bload16(rsw, lane * 32, wave * 2048), bq1 = bload16(rsw, lane * 32 + 16, wave * 2048);
        load_add(st.cb_off);
        const v4f bias = *reinterpret_cast<const v4f*>(st.bias + n4);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const float* xb = xl + (rw.al[m] * 56 + 4 * rw.tl[m]) * 4;       // row 4 t - 2 (+ 2 halo rows)
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const v4f x4 = *reinterpret_cast<const v4f*>(xb + (o + kk) * 4);      // taps 0..3: row 4 t + o - 2 + kk, the four channels
                const float x1 = xb[(o + 4) * 4 + kk];                               // tap 4: row 4 t + o + 2, channel kk
                v4f a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) a = __builtin_amdgcn_mfma_f32_16x16x4f32(bq0[s], x4[s], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(bq1[0], x1, a, 0, 0, 0);
                Y[m][o] = a + bias;
            }
        }
        gn_mish_rows<W>(Y, rw, *reinterpret_cast<const v4f*>(st.gamma + n4), *reinterpret_cast<const v4f*>(st.beta + n4), add, gnw, i16, kk);
    }
    CSTAMP(1);

    // ---- stages 1..3: Conv1d(64 -> 64, k5) + GroupNorm + Mish [+ vectors] [+ residual] in Winograd form; one code instance ----
    v4f* keep = reinterpret_cast<v4f*>(p.keep) + (size_t)blockIdx.x * (NM * 4) * 256 + tid;
#pragma clang loop unroll(disable)
    for (int s = 1; s <= 3; ++s) {
        const ChainStage& st = p.st[s];
        v4f acc[8][NM];
        wino_layer<W>(Y, acc, rw, ldsb, aoff, wofs, i16, uq, st.ufrag, wvoff, wsoff);
        CSTAMP(2 * s);
        out_transform<NM>(acc, *reinterpret_cast<const v4f*>(st.bias + n4), Y);
        // the next layer's first weight fragments travel under this layer's epilogue
        if (s < 3) uprime<W, W::TWO_PHASE ? 0 : 2>(uq, p.st[s + 1].ufrag, wvoff, wsoff);
        load_add(st.cb_off);
        gn_mish_rows<W>(Y, rw, *reinterpret_cast<const v4f*>(st.gamma + n4), *reinterpret_cast<const v4f*>(st.beta + n4), add, gnw, i16, kk);
        if (st.res_kind == CHAIN_RES_LATENT) {
            // residual_conv = Conv1d(4 -> 64, k = 1) of the block input, the latent (temporal.py:32-34)
            v4f w4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w4[j] = *reinterpret_cast<const v4f*>(p.res4_w + (size_t)(n4 + j) * 4);
            const v4f b4 = *reinterpret_cast<const v4f*>(p.res4_b + n4);
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const v4f x4 = *reinterpret_cast<const v4f*>(xl + (rw.al[m] * 56 + 2 + 4 * rw.tl[m] + o) * 4);
                    v4f r4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) r4[j] = fmaf(w4[j][3], x4[3], fmaf(w4[j][2], x4[2], fmaf(w4[j][1], x4[1], fmaf(w4[j][0], x4[0], b4[j]))));
                    Y[m][o] += rw.lv[m] ? r4 : v4f{0.f, 0.f, 0.f, 0.f};      // idle rows stay zero
                }
        } else if (st.res_kind == CHAIN_RES_KEPT) {
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int o = 0; o < 4; ++o) Y[m][o] += keep[(m * 4 + o) * 256];      // (written by this workgroup two layers ago: an L2 hit)
        }
        if (st.keep) {
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int o = 0; o < 4; ++o) keep[(m * 4 + o) * 256] = Y[m][o];
        }
        CSTAMP(2 * s + 1);
    }

    // ---- stage 4: Conv1d(64 -> 64, k3, stride 2, pad 1) + bias -> [B,26,64], from the spatial image (conv_chain.hip) ----
    {
        const int n16 = i16, q = kk, n = 16 * wave + n16;
        WQueue<64, 3, GO::NMT> wqd;
        wqd.prime(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.st[4].wfrag), 0, 4 * 3 * 4 * 1024, 0x00020000), lane * 16, 4, wave);
        __syncthreads();                                 // V is dead
        zero_halo<G>(lds, tid, 256);
        for (int i = tid; i < kSlackFloats / 4; i += 256)
            *reinterpret_cast<v4f*>(lds + G::IMG + i * 4) = v4f{0.f, 0.f, 0.f, 0.f};
        to_image_rows<W, G>(Y, rw, lds, n4);
        __syncthreads();
        const ChainStage& st = p.st[4];
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st.wfrag), 0, 4 * 3 * 4 * 1024, 0x00020000);
        constexpr int MSTEP = G::RPT * G::KCP * 4;
        v4f acd[GO::NMT];
#pragma unroll
        for (int m = 0; m < GO::NMT; ++m) acd[m] = v4f{0.f, 0.f, 0.f, 0.f};
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

template <int AG> constexpr size_t head_wino_lds() { return sizeof(float) * (v_floats<AG>() + kGnFloats + AG * 56 * 4); }
static_assert(img_floats<4>() <= v_floats<4>() + kGnFloats && img_floats<2>() <= v_floats<2>() + kGnFloats && img_floats<1>() <= v_floats<1>() + kGnFloats,
              "the stride-2 conv's image fits in front of the latent rows");
static_assert(2 * head_wino_lds<4>() <= 160 * 1024 && 3 * head_wino_lds<2>() <= 160 * 1024, "two / three workgroups per CU");

template <int AG>
static hipError_t launch_chain_head_wino_inst(const ChainHeadArgs& a, int b_pad, hipStream_t s) {
    static unsigned long long attr_done = 0;      // one bit per device (a second handle on another device sets it there too)
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(chain_head_wino_kernel<AG>), 160 * 1024, &attr_done); e != hipSuccess) return e;
    if (b_pad % AG) return hipErrorInvalidValue;
    hipLaunchKernelGGL(chain_head_wino_kernel<AG>, dim3(b_pad / AG), dim3(256), head_wino_lds<AG>(), s, a);
    return hipGetLastError();
}
hipError_t launch_chain_head_wino(const ChainHeadArgs& a, int b_pad, int agents_per_tile, hipStream_t s) {
    for (int i = 1; i <= 3; ++i)
        if (!a.st[i].ufrag) return hipErrorInvalidValue;
    return agents_per_tile == 4 ? launch_chain_head_wino_inst<4>(a, b_pad, s)
         : agents_per_tile == 2 ? launch_chain_head_wino_inst<2>(a, b_pad, s) : launch_chain_head_wino_inst<1>(a, b_pad, s);
}
// MFMAs per wave and workgroup: the latent's conv (5 per M-tile and output), three Winograd layers (8 xi x 4 chunks x NM M-tiles x 4), the stride-2 conv
template <int AG> static constexpr double head_wino_mfma() { return 20.0 * WGeo<52, AG>::NM + 3 * 128.0 * WGeo<52, AG>::NM + 48.0 * Geo<26, AG>::NMT; }
double chain_head_wino_exec_flop(int b_pad, int agents_per_tile) {
    return (agents_per_tile == 4 ? head_wino_mfma<4>() : agents_per_tile == 2 ? head_wino_mfma<2>() : head_wino_mfma<1>()) * 4 * 2048.0 * (b_pad / agents_per_tile);
}

// ---------------------------------------------------------------------------------------------------------------------
// ups.1.0's second conv + ups.1.1 + ups.1.2 + final_conv as one launch, k5 layers in Winograd form
// ---------------------------------------------------------------------------------------------------------------------
template <int AG>
__global__ __launch_bounds__(256, AG == 4 ? 2 : 3) void chain_tail_wino_kernel(const ChainTailArgs p) {
    typedef WGeo<26, AG> W;        // stages 0..2
    typedef WGeo<52, AG> V;        // the transposed conv's output, final_conv.0
    typedef Geo<26, AG> G;
    typedef Geo<52, AG> H;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* gn = lds + v_floats<AG>();
    char* ldsb = reinterpret_cast<char*>(lds);
    const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * AG;
    const int n4 = 16 * wave + 4 * kk;
    Rows<W> rw;
    rw.init(i16);
    Rows<V> rv;
    rv.init(i16);
    float* gnw = gn + wave * 256;
    int aoff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) aoff[c] = i16 * 256 + (((4 * c + kk) ^ i16) << 4);
    const int wofs = i16 * 256 + (((4 * wave + kk) ^ i16) << 4);
    const int wvoff = lane * 16, wsoff = wave * 1024;

    v4f Yf[V::NM][4];              // the L = 52 half of the kernel; the L = 26 stages use the first two M-tiles' worth
    {
        v4f uq[W::WD];
        uprime<W, W::TWO_PHASE ? 0 : 2>(uq, p.st[0].ufrag, wvoff, wsoff);
        v4f Y[W::NM][4];
        // ---- input rows [4 agents][26][64] -> the (agent, tile) layout: 16 bytes per (row, output) and lane; rows past the agent's end read 0 ----
        {
            const int xbytes = AG * 26 * 64 * 4;
            const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)b0 * 26 * 64), 0, xbytes, 0x00020000);
#pragma unroll
            for (int m = 0; m < W::NM; ++m)
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int pos = 4 * rw.tl[m] + o;
                    Y[m][o] = bload16(rsx, (rw.lv[m] && pos < 26) ? ((rw.al[m] * 26 + pos) * 64 + n4) * 4 : xbytes, 0);
                }
        }
        v4f* keep = reinterpret_cast<v4f*>(p.keep) + (size_t)blockIdx.x * (W::NM * 4) * 256 + tid;
        v4f add[W::NM];
        // ---- stages 0..2: Conv1d(64 -> 64, k5) + GroupNorm + Mish at L = 26 ----
#pragma clang loop unroll(disable)
        for (int s = 0; s < 3; ++s) {
            const ChainStage& st = p.st[s];
            v4f acc[8][W::NM];
            wino_layer<W>(Y, acc, rw, ldsb, aoff, wofs, i16, uq, st.ufrag, wvoff, wsoff);
            out_transform<W::NM>(acc, *reinterpret_cast<const v4f*>(st.bias + n4), Y);
            if (s < 2) uprime<W, W::TWO_PHASE ? 0 : 2>(uq, p.st[s + 1].ufrag, wvoff, wsoff);
            const v4f tb = (p.tbias && st.cb_off >= 0) ? *reinterpret_cast<const v4f*>(p.tbias + st.cb_off + n4) : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < W::NM; ++m) {
                add[m] = tb;
                if (st.cb_off >= 0) add[m] += *reinterpret_cast<const v4f*>(p.cbias + (size_t)(b0 + rw.al[m]) * p.cb_stride + st.cb_off + n4);
            }
            v4f rres[W::NM][4];                          // residual rows, requested ahead of the GroupNorm passes that cover their latency
            if (st.res_kind == CHAIN_RES_TENSOR) {
                const int rbytes = AG * 26 * 64 * 4;     // rows past the agent's end: out of range, read 0
                const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st.res + (size_t)b0 * 26 * 64), 0, rbytes, 0x00020000);
#pragma unroll
                for (int m = 0; m < W::NM; ++m)
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int pos = 4 * rw.tl[m] + o;
                        rres[m][o] = bload16(rsr, (rw.lv[m] && pos < 26) ? ((rw.al[m] * 26 + pos) * 64 + n4) * 4 : rbytes, 0);
                    }
            } else if (st.res_kind == CHAIN_RES_KEPT) {
#pragma unroll
                for (int m = 0; m < W::NM; ++m)
#pragma unroll
                    for (int o = 0; o < 4; ++o) rres[m][o] = keep[(m * 4 + o) * 256];
            }
            gn_mish_rows<W>(Y, rw, *reinterpret_cast<const v4f*>(st.gamma + n4), *reinterpret_cast<const v4f*>(st.beta + n4), add, gnw, i16, kk);
            if (st.res_kind == CHAIN_RES_TENSOR || st.res_kind == CHAIN_RES_KEPT) {
#pragma unroll
                for (int m = 0; m < W::NM; ++m)
#pragma unroll
                    for (int o = 0; o < 4; ++o) Y[m][o] += rres[m][o];
            }
            if (st.keep) {
#pragma unroll
                for (int m = 0; m < W::NM; ++m)
#pragma unroll
                    for (int o = 0; o < 4; ++o) keep[(m * 4 + o) * 256] = Y[m][o];
            }
        }

        // ---- ConvTranspose1d(64 -> 64, k4, s2, p1): out[2j] = x[j-1] W3 + x[j] W1, out[2j+1] = x[j] W2 + x[j+1] W0 (conv_block.hip), from the
        //      L = 26 image straight into the L = 52 (agent, tile) layout: outputs 0 | 1 of tile t are the parities of j = 2 t, outputs 2 | 3
        //      those of j = 2 t + 1 -- eight (input row, tap) products over four input rows 2 t - 1 .. 2 t + 2 ----
        __syncthreads();                                 // V is dead
        zero_halo<G>(lds, tid, 256);
        to_image_rows<W, G>(Y, rw, lds, n4);
        __syncthreads();
    }
    {
        const __amdgpu_buffer_rsrc_t rse = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.up_even.wfrag), 0, 4 * 2 * 4 * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.up_odd.wfrag), 0, 4 * 2 * 4 * 1024, 0x00020000);
        int xoff[V::NM];                                 // byte address of (agent, row 2 t - 1 + 2 halo rows, channel 4 kk)
#pragma unroll
        for (int m = 0; m < V::NM; ++m) xoff[m] = (rv.al[m] * G::ASTR + (1 + 2 * rv.tl[m]) * G::KCP + 4 * kk) * 4;
#pragma unroll
        for (int m = 0; m < V::NM; ++m)
#pragma unroll
            for (int o = 0; o < 4; ++o) Yf[m][o] = v4f{0.f, 0.f, 0.f, 0.f};
        auto wl = [&](const __amdgpu_buffer_rsrc_t r, const int g, const int t) { return bload16(r, lane * 16, ((g * 2 + t) * 4 + wave) * 1024); };
        v4f fe0 = wl(rse, 0, 0), fe1 = wl(rse, 0, 1), fo0 = wl(rso, 0, 0), fo1 = wl(rso, 0, 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const v4f e0 = fe0, e1 = fe1, o0 = fo0, o1 = fo1;
            if (g + 1 < 4) { fe0 = wl(rse, g + 1, 0); fe1 = wl(rse, g + 1, 1); fo0 = wl(rso, g + 1, 0); fo1 = wl(rso, g + 1, 1); }
#pragma unroll
            for (int m = 0; m < V::NM; ++m) {
                v4f xf[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) xf[d] = *reinterpret_cast<const v4f*>(ldsb + xoff[m] + d * (G::KCP * 4) + g * 64);
                auto mac = [&](v4f& y, const v4f w, const v4f x) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) y = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], x[e], y, 0, 0, 0);
                };
                mac(Yf[m][0], e0, xf[0]); mac(Yf[m][0], e1, xf[1]);      // out[4t]     = x[2t-1] W3 + x[2t]   W1
                mac(Yf[m][1], o0, xf[1]); mac(Yf[m][1], o1, xf[2]);      // out[4t + 1] = x[2t]   W2 + x[2t+1] W0
                mac(Yf[m][2], e0, xf[1]); mac(Yf[m][2], e1, xf[2]);      // out[4t + 2] = x[2t]   W3 + x[2t+1] W1
                mac(Yf[m][3], o0, xf[2]); mac(Yf[m][3], o1, xf[3]);      // out[4t + 3] = x[2t+1] W2 + x[2t+2] W0
            }
        }
        const v4f be = *reinterpret_cast<const v4f*>(p.up_even.bias + n4), bo = *reinterpret_cast<const v4f*>(p.up_odd.bias + n4);
        const v4f zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < V::NM; ++m) {
            Yf[m][0] = rv.lv[m] ? Yf[m][0] + be : zero; Yf[m][1] = rv.lv[m] ? Yf[m][1] + bo : zero;
            Yf[m][2] = rv.lv[m] ? Yf[m][2] + be : zero; Yf[m][3] = rv.lv[m] ? Yf[m][3] + bo : zero;
        }
    }

    // ---- final_conv.0: Conv1d(64 -> 64, k5) + GroupNorm + Mish at L = 52, Winograd form ----
    {
        v4f uq[V::WD];
        uprime<V, V::TWO_PHASE ? 0 : 2>(uq, p.fin.ufrag, wvoff, wsoff);
        v4f acc[8][V::NM];
        wino_layer<V>(Yf, acc, rv, ldsb, aoff, wofs, i16, uq, p.fin.ufrag, wvoff, wsoff);
        out_transform<V::NM>(acc, *reinterpret_cast<const v4f*>(p.fin.bias + n4), Yf);
        v4f add[V::NM];
#pragma unroll
        for (int m = 0; m < V::NM; ++m) add[m] = v4f{0.f, 0.f, 0.f, 0.f};
        gn_mish_rows<V>(Yf, rv, *reinterpret_cast<const v4f*>(p.fin.gamma + n4), *reinterpret_cast<const v4f*>(p.fin.beta + n4), add, gnw, i16, kk);
        __syncthreads();                                 // V is dead
        zero_halo<H>(lds, tid, 256);
        to_image_rows<V, H>(Yf, rv, lds, n4);
        __syncthreads();
    }

    // ---- final_conv.1: Conv1d(64 -> 4, k1) + the step's DDPM update, as conv_chain.hip's tail (and compiled like it: the update must
    //      round as head_kernel's does) ----
    {
#pragma clang fp contract(fast)
        const int n16 = i16, q = kk;
        const int hrow = H::frag0(n16);
        constexpr int HSTEP = H::RPT * H::KCP * 4;
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
                    const int pos = H::RPT * m + (4 * q + r) / AG;      // H::pos with a run-time M-tile
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

template <int AG> constexpr size_t tail_wino_lds() { return sizeof(float) * (v_floats<AG>() + kGnFloats); }

template <int AG>
static hipError_t launch_chain_tail_wino_inst(const ChainTailArgs& a, int b_pad, hipStream_t s) {
    static unsigned long long attr_done = 0;      // one bit per device (a second handle on another device sets it there too)
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(chain_tail_wino_kernel<AG>), 160 * 1024, &attr_done); e != hipSuccess) return e;
    if (b_pad % AG) return hipErrorInvalidValue;
    hipLaunchKernelGGL(chain_tail_wino_kernel<AG>, dim3(b_pad / AG), dim3(256), tail_wino_lds<AG>(), s, a);
    return hipGetLastError();
}
hipError_t launch_chain_tail_wino(const ChainTailArgs& a, int b_pad, int agents_per_tile, hipStream_t s) {
    if (!a.st[0].ufrag || !a.st[1].ufrag || !a.st[2].ufrag || !a.fin.ufrag) return hipErrorInvalidValue;
    return agents_per_tile == 4 ? launch_chain_tail_wino_inst<4>(a, b_pad, s)
         : agents_per_tile == 2 ? launch_chain_tail_wino_inst<2>(a, b_pad, s) : launch_chain_tail_wino_inst<1>(a, b_pad, s);
}
// three Winograd layers at L = 26, the transposed conv (8 products x 16 k-steps per M-tile of the L = 52 layout), final_conv.0 in Winograd form,
// final_conv.1 (16 per M-tile of the image, shared between the four waves)
template <int AG> static constexpr double tail_wino_mfma() {
    return 3 * 128.0 * WGeo<26, AG>::NM + 128.0 * WGeo<52, AG>::NM + 128.0 * WGeo<52, AG>::NM + 4.0 * Geo<52, AG>::NMT;
}
double chain_tail_wino_exec_flop(int b_pad, int agents_per_tile) {
    return (agents_per_tile == 4 ? tail_wino_mfma<4>() : agents_per_tile == 2 ? tail_wino_mfma<2>() : tail_wino_mfma<1>()) * 4 * 2048.0 * (b_pad / agents_per_tile);
}

}  // namespace cld
