// wino1d_edge.hip -- Conv1d(k = 5, pad 2) + GroupNorm(8) + Mish [+ cond / time vector] [+ residual] of TemporalMapUnet's L = 13 and
// L = 26 levels (reference: Conv1dBlock, src/tbsim/models/diffuser_helpers.py:34-67, inside ResidualTemporalMapBlockConcat,
// src/tbsim/models/temporal.py:18-60) by Winograd F(4, 5) on the part of the sequence that fills whole tiles and by the DIRECT form
// on what is left over.
//
// wino1d_kernels.hip covers L = 13 with four tiles of four outputs: 16 computed, 13 used -- the fourth tile spends 8 multiplies per channel
// pair on ONE output whose taps 3 and 4 fall on the zero padding.  Here a row of the GEMMs is not an (agent, tile) but an agent (L = 13)
// or half of one (L = 26), and what a workgroup issues per row and 4 input channels is
//     L = 13:  3 tiles x 8 xi   (outputs 0 .. 11)                      + 3 taps of output 12           = 27 MFMAs (32 there)
//     L = 26:  3 tiles x 8 xi   (outputs 12 h .. 12 h + 11, h = 0, 1)  + 4 taps of output 24 + h       = 28 per half (32 there, 8 of
//                                                                                                        every 64 rows idle)
// -- 2.41x resp. 2.32x fewer multiplies than the direct form (2.03x there), and the leftover output is as exact as the direct form makes
// it.  The taps of the direct column accumulate into ONE accumulator, so a wave holds 25 of them (100 registers) instead of 32.
//
// Kernel: a workgroup owns 16 rows (16 agents at L = 13; 8 agents x 2 halves at L = 26) x 64 output channels.  M-tile m = 0..2 of the
// 8 transform-domain products is tile m of every row, M-tile 3 the rows' direct column; wave w holds the accumulators of channels
// 16 w .. 16 w + 15; filters are the MFMA's A operand, rows its B operand: a lane ends up with four consecutive channels of ALL 13 outputs
// of its row, so the output transform, the GroupNorm sums over positions and the affine / Mish / store run without a lane exchange (the
// channel lanes of a group meet by permlane swaps, the two halves of an agent by one DPP swap, the two waves of a 32-channel group
// through 2 x 64 floats of LDS).  Staging: wave m < 3 fetches and transforms tile m of the 16 rows (thread = row x four channels of the
// 16-channel chunk: eight 16-byte loads, B^T, eight LDS stores), wave 3 copies the direct column's 3 / 4 input rows.  V images: (24 + 4)
// items of [16 rows][16 channels] (64-byte rows, 16-byte slots permuted as in wino1d_kernels.hip), two images; U and the raw taps come
// from L2 in MFMA fragment order -- 12 planes per 16-channel chunk: xi 0 .. 7, taps 0 .. 3 (cld_api.hip, ConvLayer::ufrag_edge) -- four
// planes ahead.  No dead outputs and no idle rows: every output a lane computes is stored.
#include "wino1d_common.h"

// (wino1d_kernels.hip) a row's result must not depend on its place in the workgroup: no implicit contraction in this file
#pragma clang fp contract(off)

namespace cld {

namespace {

template <int L_, int CIN_, int CS_, int COUT_>
struct E1Geo {
    static constexpr int L = L_, CIN = CIN_, CS = CS_, COUT = COUT_;
    static constexpr int AG = L == 13 ? 16 : 8;        // agents per workgroup (16 rows)
    static constexpr int NT = L == 13 ? 3 : 4;         // taps of the direct column that meet data
    static constexpr int NP = 8 + NT;                  // weight planes in use per chunk, of the 12 stored
    static constexpr int NIT = 24 + NT;                // (xi, tile) items + taps of a V image
    static constexpr int KC = 16, NCH = CIN / KC, NC1 = CS / KC, NCB = COUT / 64, NTN = COUT / 16;
    static constexpr int GS = COUT / 8;                // GroupNorm group: 32 channels = two waves, 16 = one, 8 = half of one
    static constexpr int VB = 28 * 256;                // floats per V image
    static constexpr int XCH = 2 * 4 * 16;             // [pass][wave][row]
    static constexpr size_t LDS_BYTES = (2 * VB + XCH) * sizeof(float);
    static_assert(L == 13 || L == 26, "12 outputs by tiles + one direct column per row");
    static_assert(NCH % 2 == 0 && (CS == CIN || 2 * CS == CIN), "chunk pairs are unrolled; one source or two equal ones");
    static_assert(GS == 32 || GS == 16 || GS == 8, "a group is two waves' channels, one wave's or half of one's");
    static_assert(NCB == 1 || NCB == 2 || NCB == 4, "XCD-aware id mapping");
};

}  // namespace

// KS = 1: four waves, two workgroups per CU.  KS = 2 (launches of about one item per CU, where a second workgroup per CU does not exist): eight
// waves -- waves 4 .. 7 run the second half of the input channels of the same item on V images of their own, so that every SIMD has two
// waves and neither the staging nor the weight traffic is duplicated (the half items of wino1d_kernels.hip fetch every plane twice); their
// accumulators meet those of waves 0 .. 3 through the LDS the images leave free after the last chunk, and waves 0 .. 3 run the epilogue alone.
template <int L, int CIN, int CS, int COUT, int KS>
__global__ __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) void wino1d_edge_kernel(const ConvArgs p, const int b_pad, const int xcd_map) {
    typedef E1Geo<L, CIN, CS, COUT> G;
    extern __shared__ __attribute__((aligned(16))) float lds1[];
    constexpr int VB = G::VB;
    float* xch = lds1 + 2 * KS * VB;
    const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, kk = lane >> 4;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3, kh = KS == 1 ? 0 : wave8 >> 2;          // role within the item (staging tile / output-channel tile), half of the input channels
    static_assert(KS == 1 || KS == 2, "one or two halves of the input channels");
    static_assert((G::NCH / KS) % 2 == 0, "chunk pairs are unrolled within a half");
    constexpr int NCK = G::NCH / KS;                      // chunks per half
    const int c_first = kh * NCK;
    int cb, grp;
    {
        const int e = blockIdx.x;
        if (xcd_map) {
            cb = (e >> 3) % G::NCB;
            grp = (e / (8 * G::NCB)) * 8 + (e & 7);
        } else {
            cb = e % G::NCB;
            grp = e / G::NCB;
        }
    }
    const int b0 = grp * G::AG;
    W1STAMP(0);
    W1STAMP_RT(8);

    // ---- staging role: wave m < 3: tile m of row i (inputs p0 + 4 m - 2 ..), wave 3: the direct column (inputs pd - 2 ..); four
    //      channels of the chunk per thread ----
    const int rr = (tid >> 2) & 15, cq = tid & 3;
    const int total_bytes = b_pad * L * CS * 4;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1), 0, total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(CS == CIN ? p.x1 : p.x2), 0, total_bytes, 0x00020000);
    int voff[8];
    {
        const int a = L == 13 ? rr : rr >> 1, hh = L == 13 ? 0 : rr & 1;
        const int first = wave < 3 ? 12 * hh + 4 * wave - 2 : (L == 13 ? 12 : 24 + hh) - 2;
        const int cnt = wave < 3 ? 8 : G::NT;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pos = first + i;
            voff[i] = (i < cnt && pos >= 0 && pos < L) ? (((b0 + a) * L + pos) * CS + 4 * cq) * 4 : total_bytes;      // the zero padding: out of range reads 0
        }
    }
    v4f d[8];
    auto patch_load = [&](const int i, const int c) {
        if (CS == CIN || c < G::NC1) d[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsx, voff[i], c * (G::KC * 4), 0));
        else d[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsx2, voff[i], (c - G::NC1) * (G::KC * 4), 0));
    };
    const int wofs = rr * 16 + ((((rr >> 2) & 3) ^ hsw1(cq)) << 2);
    // B^T d in four pieces (xi pairs share their even / odd halves), each stored as it is formed; wave 3 stores its rows as they are
    auto transform_piece = [&](const int k, const int buf) {
        float* vb = lds1 + (2 * kh + buf) * VB + wofs;
        if (wave == 3) {
            if (k < G::NT) *reinterpret_cast<v4f*>(vb + (24 + k) * 256) = d[k];
            return;
        }
        vb += wave * 256;
        auto st = [&](const int xi, const v4f v) { *reinterpret_cast<v4f*>(vb + xi * (3 * 256)) = v; };
        if (k == 0) {
            const v4f e = fma4(d[4], -4.25f, d[2] + d[6]), o = fma4(d[3], -4.25f, d[1] + d[5]);
            st(1, e + o); st(2, e - o);
        } else if (k == 1) {
            const v4f e = fma4(d[2], 0.25f, fma4(d[4], -1.25f, d[6])), o = fma4(d[1], 0.5f, fma4(d[3], -2.5f, 2.0f * d[5]));
            st(3, e + o); st(4, e - o);
        } else if (k == 2) {
            const v4f e = fma4(d[2], 4.0f, fma4(d[4], -5.0f, d[6])), o = fma4(d[1], 2.0f, fma4(d[3], -2.5f, 0.5f * d[5]));
            st(5, e + o); st(6, e - o);
        } else {
            st(0, fma4(d[2] - d[4], 5.25f, d[6] - d[0]));
            st(7, fma4(d[3] - d[5], 5.25f, d[7] - d[1]));
        }
    };

    // ---- MFMA role: lane (i16, kk) of wave w: row i16, channels 4 kk .. 4 kk + 3 of the chunk, output channels 16 w .. ----
    const char* ldsb = reinterpret_cast<const char*>(lds1);
    const int abase = (i16 * 16 + ((((i16 >> 2) & 3) ^ hsw1(kk)) << 2)) * 4;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wfrag), 0, G::NCH * 12 * G::NTN * 1024, 0x00020000);
    const int wvoff = lane * 16;
    const int wsoff = (cb * 4 + wave) * 1024;
    auto wload = [&](const int plane) {      // plane = chunk * 12 + (xi | 8 + tap); past the end: out of range, reads 0, never used
        return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff, plane * (G::NTN * 1024) + wsoff, 0));
    };

    const int n4 = cb * 64 + 16 * wave + 4 * kk;         // epilogue: this lane's four output channels
    const v4f bias = *reinterpret_cast<const v4f*>(p.bias + n4);
    v4f acc[8][3], accd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int xi = 0; xi < 8; ++xi)
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[xi][m] = v4f{0.f, 0.f, 0.f, 0.f};
    v4f bq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = wload(c_first * 12 + i);

#pragma unroll
    for (int i = 0; i < 8; ++i) patch_load(i, c_first);
#pragma unroll
    for (int k = 0; k < 4; ++k) transform_piece(k, 0);
    __syncthreads();
    W1STAMP(1);

    // one chunk: 8 xi x 3 tiles + NT taps, 4 MFMAs (k-steps) each.  V fragments run two items ahead of their MFMAs (a rolling window of
    // three), weight planes four ahead (ring slot = plane & 3; plane 11 of the L = 13 shapes is a slot that is skipped); the next chunk's
    // rows are requested during the first eight items and transformed during xi = 4 .. 7
    auto mfma_block = [&](const int buf, const int c, const bool stage) {
        const int bo = (2 * kh + buf) * (VB * 4);
        auto frag = [&](const int it) { return *reinterpret_cast<const v4f*>(ldsb + abase + bo + it * 1024); };
        v4f ar[3];
        ar[0] = frag(0);
        ar[1] = frag(1);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const v4f bcur = bq[j & 3];
            {
                const int nx = j + 4, npl = nx % 12;
                if (npl < G::NP) bq[j & 3] = wload((c + nx / 12) * 12 + npl);
            }
            const int nm = j < 8 ? 3 : (j < G::NP ? 1 : 0);
#pragma unroll
            for (int m = 0; m < nm; ++m) {
                const int it = j < 8 ? 3 * j + m : 24 + (j - 8);
                if (it + 2 < G::NIT) ar[(it + 2) % 3] = frag(it + 2);
                if (stage && it < 8) patch_load(it, c + 1);
                if (stage && j >= 4 && j < 8 && m == 1) transform_piece(j - 4, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                if (j < 8) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j & 7][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[e], ar[it % 3][e], acc[j & 7][m], 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) accd = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[e], ar[it % 3][e], accd, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

#pragma clang loop unroll(disable)
    for (int c = c_first; c < c_first + NCK; c += 2) {
        mfma_block(0, c, true);
        __syncthreads();
        const bool more = c + 2 < c_first + NCK;
        mfma_block(1, c + 1, more);
        __syncthreads();
        if (c == 0) W1STAMP(5);
    }
    W1STAMP(2);
    if constexpr (KS == 2) {
        // the two halves' accumulators: waves 4 .. 7 leave theirs in LDS (the images are dead: the loop ended on a barrier), [wave][tile][lane] x 16 bytes
        float* red = lds1 + (wave * 25 * 64 + lane) * 4;
        if (kh == 1) {
#pragma unroll
            for (int xi = 0; xi < 8; ++xi)
#pragma unroll
                for (int m = 0; m < 3; ++m) *reinterpret_cast<v4f*>(red + (3 * xi + m) * 256) = acc[xi][m];
            *reinterpret_cast<v4f*>(red + 24 * 256) = accd;
        }
        __syncthreads();
        if (kh == 0) {      // (three tiles at a time: read all at once, the 25 partners would need 100 registers next to the accumulators)
#pragma unroll
            for (int xi = 0; xi < 8; ++xi) {
#pragma unroll
                for (int m = 0; m < 3; ++m) acc[xi][m] += *reinterpret_cast<const v4f*>(red + (3 * xi + m) * 256);
                __builtin_amdgcn_sched_barrier(0);
            }
            accd += *reinterpret_cast<const v4f*>(red + 24 * 256);
        }
    }
  constexpr bool PAIR = G::GS == 32;
  if (kh == 0) {

    // ---- epilogue.  Lane: channels n4 .. n4 + 3 of row i16: outputs p0 + 4 m + o (m < 3, o < 4) and pd ----
    v4f Y[13];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const v4f p12 = acc[1][m] + acc[2][m], m12 = acc[1][m] - acc[2][m];
        const v4f p34 = acc[3][m] + acc[4][m], m34 = acc[3][m] - acc[4][m];
        const v4f p56 = acc[5][m] + acc[6][m], m56 = acc[5][m] - acc[6][m];
        Y[4 * m + 0] = ((acc[0][m] + p12) + (p34 + p56)) + bias;
        Y[4 * m + 1] = fma4(m56, 0.5f, fma4(m34, 2.0f, m12)) + bias;
        Y[4 * m + 2] = fma4(p56, 0.25f, fma4(p34, 4.0f, p12)) + bias;
        Y[4 * m + 3] = (fma4(m56, 0.125f, fma4(m34, 8.0f, m12)) + acc[7][m]) + bias;
    }
    Y[12] = accd + bias;
    // every operand of the second half of the epilogue is requested now, behind the output transform (wino1d_kernels.hip) -- and not before it:
    // the residual loads sit in a conditional block of their own, which the compiler otherwise places in front of the transform (the eight-wave
    // instances spilled 40 .. 56 registers that way); a compiler-level memory fence that consumes Y pins the order
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int o = 0; o < 13; ++o) asm volatile("" : "+v"(Y[o]) : : "memory");
    const int ybytes = b_pad * L * COUT * 4;
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.y), 0, ybytes, 0x00020000);
    const bool has_res = p.res != nullptr;
    const int b = b0 + (L == 13 ? i16 : i16 >> 1), hh = L == 13 ? 0 : i16 & 1;
    // (the distances between a row's outputs go into the VECTOR offset: with them in the scalar offset operand of the 16-byte stores, whose
    //  data registers the next output reuses, lanes 12 .. 15 of every 16 stored the next output's values -- measured on gfx950)
    const int obase = ((b * L + 12 * hh) * COUT + n4) * 4;                        // outputs 12 hh + 0 .. 11
    const int odir = ((b * L + (L == 13 ? 12 : 24 + hh)) * COUT + n4) * 4;
    v4f rv[13];
    if (has_res) {
#pragma unroll
        for (int o = 0; o < 12; ++o) rv[o] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsr, obase + o * (COUT * 4), 0, 0));
        rv[12] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsr, odir, 0, 0));
    }
    v4f cbv = {0.f, 0.f, 0.f, 0.f};
    if (p.cbias) cbv = *reinterpret_cast<const v4f*>(p.cbias + (size_t)b * p.cb_stride + n4);
    const v4f gam = *reinterpret_cast<const v4f*>(p.gamma + n4), bet = *reinterpret_cast<const v4f*>(p.beta + n4);
    const v4f tb = p.tbias ? *reinterpret_cast<const v4f*>(p.tbias + n4) : v4f{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    // GroupNorm(GS channels x L rows per agent, eps 1e-5, biased variance, two passes; diffuser_helpers.py:61): a lane's partial sum
    // (4 channels x 13 outputs) -> the group's total: channel lanes by permlane swaps (16 lanes apart: the other channel quad of an
    // 8-channel group; 32 apart: the rest of the wave's 16 channels), the other half of the agent at L = 26 by a DPP swap of neighbouring
    // lanes, the other wave of a 32-channel group through LDS
    const float inv = 1.0f / (float)(G::GS * L);
    auto group_total = [&](float sv, float* scratch) {
        const auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, sv), __builtin_bit_cast(unsigned, sv), false, false);
        const unsigned a16 = r16[0], b16 = r16[1];      // (scalars first: __builtin_bit_cast of a vector ELEMENT reads element 0)
        sv = __builtin_bit_cast(float, a16) + __builtin_bit_cast(float, b16);
        if (G::GS >= 16) {
            const auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, sv), __builtin_bit_cast(unsigned, sv), false, false);
            const unsigned a32 = r32[0], b32 = r32[1];
            sv = __builtin_bit_cast(float, a32) + __builtin_bit_cast(float, b32);
        }
        if (L == 26) sv += W1_DPP(sv, 0xB1);       // quad_perm:[1,0,3,2]: the agent's other half
        if (PAIR) {
            if (kk == 0) scratch[wave * 16 + i16] = sv;
            __syncthreads();
            sv += scratch[(wave ^ 1) * 16 + i16];
        }
        return sv;
    };
    auto lo2 = [](const v4f v) { return __builtin_shufflevector(v, v, 0, 1); };
    auto hi2 = [](const v4f v) { return __builtin_shufflevector(v, v, 2, 3); };
    float mean, s2;
    {
        v2f sv = {0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 13; ++o) {
            sv += lo2(Y[o]);
            sv += hi2(Y[o]);
        }
        mean = group_total(sv[0] + sv[1], xch) * inv;
        const v2f mm = {mean, mean};
        v2f q = {0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 13; ++o) {
            const v2f d0 = lo2(Y[o]) - mm, d1 = hi2(Y[o]) - mm;
            q = __builtin_elementwise_fma(d0, d0, q);
            q = __builtin_elementwise_fma(d1, d1, q);
        }
        s2 = group_total(q[0] + q[1], xch + G::XCH / 2);
    }
    W1STAMP(3);
    {
        // (Y - mean) sc + beta = Y sc + (beta - mean sc)
        const v4f sc = (1.0f / sqrtf(s2 * inv + 1e-5f)) * gam;
        const v4f sh = __builtin_elementwise_fma(sc, v4f{-mean, -mean, -mean, -mean}, bet);
        const v4f add = tb + cbv;
#pragma unroll
        for (int o = 0; o < 13; ++o) {
            v4f v = mish4(__builtin_elementwise_fma(Y[o], sc, sh)) + add;
            if (has_res) v += rv[o];
            if (o < 12) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), rsy, obase + o * (COUT * 4), 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), rsy, odir, 0, 0);
        }
    }
    W1STAMP(4);
    W1STAMP_RT(9);
  } else if (PAIR) {      // waves 4 .. 7: the two barriers of the epilogue's statistics passes
    __syncthreads();
    __syncthreads();
  }
}

long wino1d_edge_row_planes(int l_in, int b_pad) { return l_in == 13 ? 27L * b_pad : 56L * b_pad; }      // (GEMM rows x planes) of a launch: 27 per agent, 2 x 28 per agent

template <int L, int CIN, int CS, int COUT>
static hipError_t launch_wino1d_edge_inst(const ConvArgs& a, int b_pad, bool k_split, hipStream_t s) {
    typedef E1Geo<L, CIN, CS, COUT> G;
    auto kern = wino1d_edge_kernel<L, CIN, CS, COUT, 1>;
    auto kern2 = wino1d_edge_kernel<L, CIN, CS, COUT, 2>;
    constexpr size_t lds2 = (4 * G::VB + G::XCH) * sizeof(float);
    static unsigned long long attr_done = 0, attr_done2 = 0;      // one bit per device
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, &attr_done); e != hipSuccess) return e;
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(kern2), (int)lds2, &attr_done2); e != hipSuccess) return e;
    if ((long)b_pad * L * CS * 4 >= (1L << 31) || (long)b_pad * L * COUT * 4 >= (1L << 31)) return hipErrorInvalidValue;      // byte offsets are 32-bit
    const int groups = b_pad / G::AG;
    if (k_split) hipLaunchKernelGGL(kern2, dim3(groups * G::NCB), dim3(512), lds2, s, a, b_pad, groups % 8 == 0 ? 1 : 0);
    else hipLaunchKernelGGL(kern, dim3(groups * G::NCB), dim3(256), G::LDS_BYTES, s, a, b_pad, groups % 8 == 0 ? 1 : 0);
    return hipGetLastError();
}

// a.wfrag: the 12-plane fragments (ConvLayer::ufrag_edge)
hipError_t launch_wino1d_edge(const ConvArgs& a, int l_in, int b_pad, bool k_split, hipStream_t s) {
    if (b_pad < 16 || b_pad % 16 || a.res4_x || a.c1_real != a.c1_pad || (a.c2 != 0) != (a.x2 != nullptr)) return hipErrorInvalidValue;
#define X(L, CIN, CS, COUT) \
    if (l_in == L && a.c1_real == CS && a.c1_real + a.c2 == CIN && a.c_out == COUT) return launch_wino1d_edge_inst<L, CIN, CS, COUT>(a, b_pad, k_split, s);
    CLD_WINO1D_INSTANCES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace cld
