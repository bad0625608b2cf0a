// wino1d_kernels.hip -- Conv1d(k = 5, pad 2) + GroupNorm(8) + Mish [+ cond / time vector] [+ residual] of TemporalMapUnet's
// 256-channel level (reference: Conv1dBlock, src/tbsim/models/diffuser_helpers.py:34-67, inside ResidualTemporalMapBlockConcat,
// src/tbsim/models/temporal.py:18-60; the seven 256 -> 256 launches at L = 13 of a U-Net evaluation, 50 % of its FLOPs) by
// Winograd / Cook-Toom minimal filtering F(4, 5).
//
// conv_block.hip runs these launches at the clock-adjusted ceiling of the fp32 MFMA pipe: what is left to take out is
// the arithmetic.  Four outputs of a 5-tap correlation need 8 multiplies instead of 20 when the filter and the 8-row input tile
// d = x[4t - 2 .. 4t + 5] are taken to the points {0, +-1, +-2, +-1/2, inf}:
//     y[4t .. 4t + 3] = A^T [ (G g) (.) (B^T d) ]
//     B^T = [ -1  0  21/4   0   -21/4   0    1  0        A^T = [ 1 1  1 1  1  1    1   0
//              0  1   1   -17/4 -17/4   1    1  0                0 1 -1 2 -2 1/2 -1/2  0
//              0 -1   1    17/4 -17/4  -1    1  0                0 1  1 4  4 1/4  1/4  0
//              0 1/2 1/4   -5/2  -5/4   2    1  0                0 1 -1 8 -8 1/8 -1/8  1 ]
//              0 -1/2 1/4   5/2  -5/4  -2    1  0
//              0  2   4    -5/2  -5    1/2   1  0
//              0 -2   4     5/2  -5   -1/2   1  0
//              0 -1   0    21/4   0   -21/4  0  1 ]
// (every entry exact in fp32; G g is formed in double at cld_finalize).  The sum over input channels moves inside the element-wise
// product: 8 GEMMs  M_xi[row][n] = sum_c V_xi[row][c] U_xi[c][n]  over rows = (agent, tile).  L = 13 is four tiles per agent (16
// outputs, three discarded), so a launch issues 8 x 4 MFMA k-steps per agent and channel pair where the direct form issues
// 5 x 13: 2.03x fewer.  Rounding: 9e-7 of max|y| against fp64 on unit-variance data, the direct form 6e-7 (one-dimensional transforms
// do not square the constants the way F(4x4, 3x3) does).
//
// Kernel (the structure of wino_kernels.hip): a workgroup owns 16 agents = 64 rows = four M-tiles x 64 output channels x the 8 xi;
// wave w holds the 32 accumulators of channels 16 w .. 16 w + 15; the filters are the MFMA's A operand and the rows its B operand, so
// a lane ends up with four consecutive channels of one (agent, tile).  Per 16-channel chunk a thread fetches the 8 input rows of one
// (agent, tile) x four channels (eight 16-byte loads, issued two blocks' worth of latency ahead), applies B^T in the shadow of the
// MFMAs and writes V[xi][row][16 channels] into one of two LDS images (64-byte rows, 16-byte slots permuted so the four 16-lane groups
// of a ds_read_b128 touch every bank once).  U comes from L2 in MFMA fragment order (pack_conv_weights with xi as the tap), four items
// ahead.  Epilogue in registers: A^T, conv bias, two-pass GroupNorm statistics (a group = 32 channels = two waves: DPP + permlane
// sums inside the wave, 2 x 64 floats of LDS between the two), affine, Mish, vectors, residual, 16-byte stores.
// Workgroup id -> (agent group, channel block) keeps the four channel blocks of an agent group on one XCD (they stage the same rows).
#include "wino1d_common.h"

#ifndef WINO1D_HALVES_BELOW
#define WINO1D_HALVES_BELOW 512
#endif
#ifndef CLD_STORE_AUX
#define CLD_STORE_AUX 16
#endif

// A row's result must not depend on its place in the workgroup (tests: a shuffled batch reproduces its rows bit for bit).  The epilogue
// is unrolled over the four M-tiles, and left to itself the compiler fuses a multiply-add in one unrolled copy and not in another
// (scalar there, packed here): no implicit contraction in this file -- every fused multiply-add below is written as one.
#pragma clang fp contract(off)

namespace cld {

namespace {

// CIN input channels in one tensor (CS == CIN) or in two of CS = CIN / 2 channels each (torch.cat of the skip, temporal.py:167: a second
// buffer descriptor); COUT output channels in blocks of 64 per workgroup
template <int L_, int CIN_, int CS_, int COUT_>
struct W1Geo {
    static constexpr int L = L_, CIN = CIN_, CS = CS_, COUT = COUT_;
    static constexpr int TPA = (L + 3) / 4;            // tiles per agent: 4 at L = 13, 7 at L = 26
    static constexpr int AG = TPA == 4 ? 16 : 8;       // agents per workgroup: 64 rows = four M-tiles (at L = 26 the last 8 rows are idle)
    static constexpr int ROWS = AG * TPA;
    static constexpr int KC = 16, NCH = CIN / KC, NC1 = CS / KC, NCB = COUT / 64, NTN = COUT / 16;
    static constexpr int GS = COUT / 8;                // GroupNorm group: 32 channels = two waves, 16 = one, 8 = half of one
    static constexpr int VBUF = 8 * 64 * KC;           // floats per V image
    static constexpr int XCH = 2 * 4 * 2 * 64;         // exchange scratch behind the images: [pass][wave][group of the wave][row]
    static constexpr size_t LDS_BYTES = (2 * VBUF + XCH) * sizeof(float);
    static_assert(TPA == 4 || TPA == 7, "L = 13 or 26");
    static_assert(NCH % 2 == 0 && (CS == CIN || 2 * CS == CIN), "chunk pairs are unrolled; one source or two equal ones");
    static_assert(GS == 32 || GS == 16 || GS == 8, "a group is two waves' channels, one wave's or half of one's");
    static_assert(GS != 32 || TPA == 4, "the two-wave exchange is written for the quad layout");
    static_assert(NCB == 1 || NCB == 2 || NCB == 4, "XCD-aware id mapping");
};

}  // namespace

// One item: NM M-tiles (4 = a whole agent group, 2 = half of one) x the 64 output channels of block cb, agents b0 ..
template <int L, int CIN, int CS, int COUT, int NM>
__device__ __forceinline__ void wino1d_item(const ConvArgs& p, const int b_pad, const int cb, const int b0, float* lds1) {
    typedef W1Geo<L, CIN, CS, COUT> G;
    constexpr int ROWS = G::ROWS * NM / 4;             // live rows of the item (at L = 26 the rest of its last M-tile idles)
    constexpr int VB = 8 * 16 * NM * G::KC;             // floats per V image of this item: 8 xi x 16 NM rows x 16 channels (a half item's images are half as large)
    float* xch = lds1 + 2 * VB;                        // GroupNorm sums meet here: two waves of a 32-channel group; the rows of an agent at L = 26
    const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    W1STAMP(0);
    W1STAMP_RT(8);
#ifdef CLD_STAMPS
    if (p.stamps && tid == 0) {      // where the dispatcher put this workgroup: HW_ID (wave slot, SIMD, CU, SH, SE) and XCC_ID
        p.stamps[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        p.stamps[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif

    // ---- staging role: row rs = (agent rs / TPA, tile rs % TPA); a whole item: 64 rows x four channels of the chunk per thread (16-byte
    //      loads); a half item: 32 rows x TWO channels (8-byte loads), so that all four waves stage there as well -- with the 16-byte map
    //      waves 2, 3 of a half item had nothing to stage and waited at every chunk barrier for waves 0, 1 ----
    constexpr int SW = NM == 4 ? 4 : 2;                // channels per staging thread
    typedef float vS __attribute__((ext_vector_type(SW)));
    const int rs = NM == 4 ? tid >> 2 : tid >> 3, cq = NM == 4 ? tid & 3 : tid & 7;
    const int total_bytes = b_pad * L * CS * 4;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1), 0, total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(CS == CIN ? p.x1 : p.x2), 0, total_bytes, 0x00020000);
    int voff[8];
    {
        const int a = rs / G::TPA, t = rs % G::TPA;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pos = 4 * t - 2 + i;
            voff[i] = (rs < ROWS && pos >= 0 && pos < L) ? (((b0 + a) * L + pos) * CS + SW * cq) * 4 : total_bytes;      // the zero padding: out of range reads 0
        }
    }
    vS d[8];
    auto ld = [&](const __amdgpu_buffer_rsrc_t r, const int vo, const int so) {
        if constexpr (SW == 4) return __builtin_bit_cast(vS, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
        else return __builtin_bit_cast(vS, __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0));
    };
    auto patch_load = [&](const int i, const int c) {
        if (CS == CIN || c < G::NC1) d[i] = ld(rsx, voff[i], c * (G::KC * 4));
        else d[i] = ld(rsx2, voff[i], (c - G::NC1) * (G::KC * 4));
    };
    const int c4 = NM == 4 ? cq : cq >> 1;              // the 16-byte slot of the row
    const int wofs = rs * 16 + ((((rs >> 2) & 3) ^ hsw1(c4)) << 2) + (NM == 4 ? 0 : (cq & 1) * 2);
    auto fmaS = [](const vS a, const float sc, const vS b) {
        vS sv;
#pragma unroll
        for (int j = 0; j < SW; ++j) sv[j] = sc;
        return __builtin_elementwise_fma(a, sv, b);
    };
    // B^T d in four pieces (xi pairs share their even / odd halves), each stored as it is formed
    auto transform_piece = [&](const int k, const int buf) {
        float* vb = lds1 + buf * VB + wofs;
        auto st = [&](const int xi, const vS v) { *reinterpret_cast<vS*>(vb + xi * (16 * NM * 16)) = v; };
        if (k == 0) {
            const vS e = fmaS(d[4], -4.25f, d[2] + d[6]), o = fmaS(d[3], -4.25f, d[1] + d[5]);
            st(1, e + o); st(2, e - o);
        } else if (k == 1) {
            const vS e = fmaS(d[2], 0.25f, fmaS(d[4], -1.25f, d[6])), o = fmaS(d[1], 0.5f, fmaS(d[3], -2.5f, 2.0f * d[5]));
            st(3, e + o); st(4, e - o);
        } else if (k == 2) {
            const vS e = fmaS(d[2], 4.0f, fmaS(d[4], -5.0f, d[6])), o = fmaS(d[1], 2.0f, fmaS(d[3], -2.5f, 0.5f * d[5]));
            st(5, e + o); st(6, e - o);
        } else {
            st(0, fmaS(d[2] - d[4], 5.25f, d[6] - d[0]));
            st(7, fmaS(d[3] - d[5], 5.25f, d[7] - d[1]));
        }
    };

    // ---- MFMA role: lane (i16, kk) of wave w: rows 16 m + i16, channels 4 kk .. 4 kk + 3 of the chunk, output channels 16 w .. ----
    const char* ldsb = reinterpret_cast<const char*>(lds1);
    const int abase = (i16 * 16 + ((((i16 >> 2) & 3) ^ hsw1(kk)) << 2)) * 4;
    const int nitems = G::NCH * 8;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wfrag), 0, nitems * G::NTN * 1024, 0x00020000);
    const int wvoff = lane * 16;
    const int wsoff = (cb * 4 + wave) * 1024;
    auto wload = [&](int item) {          // item = chunk * 8 + xi; past the end: out of range, reads 0, never used
        return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff, item * (G::NTN * 1024) + wsoff, 0));
    };

    const int n4 = cb * 64 + 16 * wave + 4 * kk;         // epilogue: this lane's four output channels
    const v4f bias = *reinterpret_cast<const v4f*>(p.bias + n4);      // (requested here: the output transform would wait for it)
    v4f acc[8][NM];
#pragma unroll
    for (int xi = 0; xi < 8; ++xi)
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[xi][m] = v4f{0.f, 0.f, 0.f, 0.f};
    v4f bq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = wload(i);

#pragma unroll
    for (int i = 0; i < 8; ++i) patch_load(i, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) transform_piece(k, 0);
    __syncthreads();
    W1STAMP(1);

    // one chunk: 8 xi x NM M-tiles x 4 MFMAs.  Fragments run two (xi, M-tile) items ahead of their MFMAs (a rolling window of three);
    // the next chunk's rows are requested during xi = 0, 1 and transformed during xi = 4 .. 7
    auto mfma_block = [&](const int buf, const int c, const bool stage) {
        const int bo = buf * (VB * 4);
        auto frag = [&](const int it) { return *reinterpret_cast<const v4f*>(ldsb + abase + bo + it * 1024); };      // (xi, M-tile) = it / NM, it % NM: images of 16 NM rows
        v4f ar[3];
        ar[0] = frag(0);
        ar[1] = frag(1);
#pragma unroll
        for (int xi = 0; xi < 8; ++xi) {
            const v4f bcur = bq[xi & 3];
            bq[xi & 3] = wload(c * 8 + xi + 4);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int it = NM * xi + m;
                if (it + 2 < 8 * NM) ar[(it + 2) % 3] = frag(it + 2);
                if (stage && it < 8) patch_load(it, c + 1);
                if (stage && xi >= 4 && m == 1) transform_piece(xi - 4, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[xi][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[e], ar[it % 3][e], acc[xi][m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

#pragma clang loop unroll(disable)
    for (int c = 0; c < G::NCH; c += 2) {
        mfma_block(0, c, true);
        __syncthreads();
        const bool more = c + 2 < G::NCH;
        mfma_block(1, c + 1, more);
        __syncthreads();
        if (c == 0) W1STAMP(5);
    }
    W1STAMP(2);

    // ---- epilogue.  Lane: channels n4 .. n4 + 3; M-tile m: row 16 m + i16 = (agent row / TPA, tile t = row % TPA), outputs at 4 t + o ----
    int al[NM], tl[NM];                                   // agent (within the workgroup) and tile of this lane's row of M-tile m
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int r = 16 * m + i16;
        al[m] = G::TPA == 4 ? r >> 2 : r / G::TPA;
        tl[m] = G::TPA == 4 ? r & 3 : r % G::TPA;
    }
    v4f Y[NM][4];                                         // [m][o]
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
    // Every operand of the second half of the epilogue is requested NOW, behind the output transform: residual rows, per-agent and
    // per-step vectors, GroupNorm affine.  The two statistics passes below (two workgroup barriers in the 32-channel-group case) cover
    // their latency; requested where they are used -- per M-tile, in front of their first use -- each M-tile of the epilogue waited out
    // a trip to L2 / HBM with both workgroups of every CU doing the same at the same moment (5 exposed round trips per workgroup).
    // Output and residual go through buffer descriptors: one 32-bit offset per (M-tile, lane), the four outputs of a tile at immediate
    // distances, a dead output (past the end of the agent, an idle row) at an out-of-range offset -- dropped by the range check: no
    // 64-bit address arithmetic and no branch per store
    __builtin_amdgcn_sched_barrier(0);                    // (not before the accumulators are dead: 128 + 92 registers would not fit)
    const int ybytes = b_pad * L * COUT * 4;
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.y), 0, ybytes, 0x00020000);
    const bool has_res = p.res != nullptr;
    int ooff[NM][4];
    v4f rv[NM][4], cbv[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int b = b0 + al[m];
        const bool rowok = ROWS == 16 * NM || 16 * m + i16 < ROWS;
        const int obase = (((b * L + 4 * tl[m]) * COUT) + n4) * 4;
#pragma unroll
        for (int o = 0; o < 4; ++o) ooff[m][o] = (rowok && 4 * tl[m] + o < L) ? obase + o * (COUT * 4) : ybytes;
        if (has_res) {
#pragma unroll
            for (int o = 0; o < 4; ++o) rv[m][o] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsr, ooff[m][o], 0, 0));
        }
        cbv[m] = v4f{0.f, 0.f, 0.f, 0.f};
        if (p.cbias && rowok) cbv[m] = *reinterpret_cast<const v4f*>(p.cbias + (size_t)b * p.cb_stride + n4);
    }
    const v4f gam = *reinterpret_cast<const v4f*>(p.gamma + n4), bet = *reinterpret_cast<const v4f*>(p.beta + n4);
    const v4f tb = p.tbias ? *reinterpret_cast<const v4f*>(p.tbias + n4) : v4f{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    // GroupNorm(GS channels x L rows per agent, eps 1e-5, biased variance, two passes; diffuser_helpers.py:61).  group_totals() turns a
    // lane's partial sums (its 4 channels x its row's outputs) into the group's totals:
    //   L = 13: an agent's four tiles are a quad of lanes -> DPP + permlane sums; a 32-channel group's other half lives in wave w ^ 1;
    //   L = 26: the channel lanes by permlane sums, then the seven rows of the agent through LDS (wave-private: one wave holds whole
    //           groups there), added in a fixed order.
    constexpr bool PAIR = G::GS == 32;
    const float inv = 1.0f / (float)(G::GS * L);
    auto group_totals = [&](float (&v)[NM], float* scratch) {
        if constexpr (G::TPA == 4) {
#pragma unroll
            for (int m = 0; m < NM; ++m) v[m] = agent_sum(v[m]);
            if (PAIR) {
                if (kk == 0 && (i16 & 3) == 0) {
#pragma unroll
                    for (int m = 0; m < NM; ++m) scratch[wave * 16 + 4 * m + (i16 >> 2)] = v[m];
                }
                __syncthreads();
#pragma unroll
                for (int m = 0; m < NM; ++m) v[m] += scratch[(wave ^ 1) * 16 + 4 * m + (i16 >> 2)];
            }
        } else {
            const int gsel = G::GS == 8 ? kk >> 1 : 0;
            float* rowsum = scratch + (wave * 2 + gsel) * 64;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                float sv = v[m];
                const auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, sv), __builtin_bit_cast(unsigned, sv), false, false);
                const unsigned a16 = r16[0], b16 = r16[1];
                sv = __builtin_bit_cast(float, a16) + __builtin_bit_cast(float, b16);
                if (G::GS == 16) {
                    const auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, sv), __builtin_bit_cast(unsigned, sv), false, false);
                    const unsigned a32 = r32[0], b32 = r32[1];
                    sv = __builtin_bit_cast(float, a32) + __builtin_bit_cast(float, b32);
                }
                if ((kk & (G::GS == 8 ? 1 : 3)) == 0) rowsum[16 * m + i16] = sv;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the wave's own LDS writes have landed (a wave's LDS operations complete in order)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                float tot = 0.f;
                const int r0 = al[m] * G::TPA;
#pragma unroll
                for (int j = 0; j < G::TPA; ++j) tot += rowsum[(r0 + j) & 63];
                v[m] = tot;
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    // The arithmetic below runs on register PAIRS (v_pk_add / mul / fma_f32): with both workgroups of a CU in their epilogues the phase is
    // bound by VALU issue (scripts/ubench/mfma_covalu.hip).  Outputs past the agent's end exist only in an agent's last tile (L = 13: its
    // outputs 1..3; L = 26: 2, 3): they are zeroed by a factor per row instead of a select per value, and an idle row needs no guard at all
    // -- its sums are never read and its stores fall outside the buffer's range.
    constexpr int NDEAD = 4 * G::TPA - L;
    auto lo2 = [](const v4f v) { return __builtin_shufflevector(v, v, 0, 1); };
    auto hi2 = [](const v4f v) { return __builtin_shufflevector(v, v, 2, 3); };
    auto cat2 = [](const v2f a, const v2f b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3); };
    float tailf[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) tailf[m] = tl[m] == G::TPA - 1 ? 0.f : 1.f;
    float mean[NM], s2[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        v2f sv = {0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o >= 4 - NDEAD) Y[m][o] = cat2(lo2(Y[m][o]) * tailf[m], hi2(Y[m][o]) * tailf[m]);
            sv += lo2(Y[m][o]);
            sv += hi2(Y[m][o]);
        }
        mean[m] = sv[0] + sv[1];
    }
    group_totals(mean, xch);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        mean[m] *= inv;
        const v2f mm = {mean[m], mean[m]};
        v2f q = {0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            v2f d0 = lo2(Y[m][o]) - mm, d1 = hi2(Y[m][o]) - mm;
            if (o >= 4 - NDEAD) { d0 *= tailf[m]; d1 *= tailf[m]; }
            q = __builtin_elementwise_fma(d0, d0, q);
            q = __builtin_elementwise_fma(d1, d1, q);
        }
        s2[m] = q[0] + q[1];
    }
    group_totals(s2, xch + G::XCH / 2);
    W1STAMP(3);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        // (Y - mean) sc + beta = Y sc + (beta - mean sc)
        const v4f sc = (1.0f / sqrtf(s2[m] * inv + 1e-5f)) * gam;
        const v4f sh = __builtin_elementwise_fma(sc, v4f{-mean[m], -mean[m], -mean[m], -mean[m]}, bet);
        const v4f add = tb + cbv[m];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            v4f v = mish4(__builtin_elementwise_fma(Y[m][o], sc, sh)) + add;
            if (has_res) v += rv[m][o];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), rsy, ooff[m][o], 0, 0);
        }
    }
    W1STAMP(4);
    W1STAMP_RT(9);
}

// Workgroup id -> item.  Whole items (a 16-agent group at L = 13, 8 agents at L = 26) when there are at least 512 of them -- two per CU;
// below that every group is split into two half items (two M-tiles each: twice the U traffic per MFMA, but a second wave on every SIMD:
// at 1,024 rows 256 whole items leave every CU with one workgroup).  ids x, x + 8, x + 16, x + 24 (whole) resp. 2 x the same (halves) are
// the channel blocks of one agent group: one XCD, back to back.
// Two instantiations per layer shape: whole items (128 accumulator registers: two workgroups per CU) and half items, which need half the
// accumulators and half the LDS (36 KB) -- WINO1D_HALF_WGS workgroups per CU.
#ifndef WINO1D_HALF_WGS
#define WINO1D_HALF_WGS 2
#endif
template <int L, int CIN, int CS, int COUT, int HALVES>
__global__ __launch_bounds__(256, HALVES ? WINO1D_HALF_WGS : 2) void wino1d_conv_kernel(const ConvArgs p, const int b_pad, const int xcd_map) {
    typedef W1Geo<L, CIN, CS, COUT> G;
    extern __shared__ __attribute__((aligned(16))) float lds1[];
    constexpr int halves = HALVES;
    const int e = halves ? blockIdx.x >> 1 : blockIdx.x, half = halves ? (int)(blockIdx.x & 1) : -1;
    int cb, grp;
    if (xcd_map) {
        cb = (e >> 3) % G::NCB;
        grp = (e / (8 * G::NCB)) * 8 + (e & 7);
    } else {
        cb = e % G::NCB;
        grp = e / G::NCB;
    }
    if constexpr (!HALVES) wino1d_item<L, CIN, CS, COUT, 4>(p, b_pad, cb, grp * G::AG, lds1);
    else wino1d_item<L, CIN, CS, COUT, 2>(p, b_pad, cb, grp * G::AG + half * (G::AG / 2), lds1);
}

bool wino1d_supported(int l_in, int c1, int c2, int c_out) {
#define X(L, CIN, CS, COUT) \
    if (l_in == L && c1 == CS && c1 + c2 == CIN && c_out == COUT) return true;
    CLD_WINO1D_INSTANCES(X)
#undef X
    return false;
}

long wino1d_gemm_rows(int l_in, int b_pad) { return l_in == 13 ? 4L * b_pad : 8L * b_pad; }      // 16 / 8 agents per 64-row item

// Whole items (a 16-agent group at L = 13, 8 agents at L = 26) when they fill generations of 512 (two workgroups per CU); half items
// below one generation and when the last generation would be at most half full (752 whole items are two rounds, 1,504 halves one and a
// half).  Whole items run in wino1d_edge.hip's form (the ragged end of the sequence direct: 27 / 28 MFMAs per row where the items of
// this file issue 32), half items here.
#ifndef WINO1D_EDGE
#define WINO1D_EDGE 1
#endif
static bool takes_halves(int l_in, int c_out, int b_pad) {
    const int nfull = b_pad / (l_in == 13 ? 16 : 8) * (c_out / 64);
    const int tail = nfull % WINO1D_HALVES_BELOW;
    return nfull < WINO1D_HALVES_BELOW || (tail != 0 && tail <= WINO1D_HALVES_BELOW / 2);
}
// ... except that a launch of about one whole item per CU runs whole items of EIGHT waves (wino1d_edge.hip, KS = 2: the two halves of the
// input channels on four waves each): two waves per SIMD without fetching any weight plane or staging any row twice
#ifndef WINO1D_KSPLIT_MIN
#define WINO1D_KSPLIT_MIN 192
#endif
static bool takes_ksplit(int l_in, int c_out, int b_pad) {
    const int nfull = b_pad / (l_in == 13 ? 16 : 8) * (c_out / 64);
    return nfull >= WINO1D_KSPLIT_MIN && nfull <= 256;
}
// 0: half items, 1: whole items, 2: whole items of eight waves
static int item_form_of(int l_in, int c_out, int b_pad, int forced) {
    if (!WINO1D_EDGE) return takes_halves(l_in, c_out, b_pad) ? 0 : 3;
    if (forced) return forced;
    if (takes_ksplit(l_in, c_out, b_pad)) return 2;
    return takes_halves(l_in, c_out, b_pad) ? 0 : 1;
}
long wino1d_row_planes(int l_in, int c_out, int b_pad, int item_form) {
    const int f = item_form_of(l_in, c_out, b_pad, item_form);
    return f == 1 || f == 2 ? wino1d_edge_row_planes(l_in, b_pad) : 8L * wino1d_gemm_rows(l_in, b_pad);
}

template <int L, int CIN, int CS, int COUT>
static hipError_t launch_wino1d_inst(const ConvArgs& a, int b_pad, int item_form, hipStream_t s) {
    typedef W1Geo<L, CIN, CS, COUT> G;
    const int groups = b_pad / G::AG, nfull = groups * G::NCB;
    const int form = item_form_of(L, COUT, b_pad, item_form);
    const bool halves = form == 0;
    if (form == 1 || form == 2) {
        ConvArgs e = a;
        e.wfrag = a.wfrag_edge;
        return e.wfrag ? launch_wino1d_edge(e, L, b_pad, form == 2, s) : hipErrorInvalidValue;
    }
    auto kern_whole = wino1d_conv_kernel<L, CIN, CS, COUT, 0>;
    auto kern_half = wino1d_conv_kernel<L, CIN, CS, COUT, 1>;
    constexpr size_t lds_half = (2 * (G::VBUF / 2) + G::XCH) * sizeof(float);
    static unsigned long long attr_done = 0, attr_done_h = 0;      // one bit per device (a second handle on another device sets it there too)
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(kern_whole), (int)G::LDS_BYTES, &attr_done); e != hipSuccess) return e;
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(kern_half), (int)lds_half, &attr_done_h); e != hipSuccess) return e;
    if ((long)b_pad * L * CS * 4 >= (1L << 31) || (long)b_pad * L * COUT * 4 >= (1L << 31)) return hipErrorInvalidValue;      // byte offsets are 32-bit
    if (halves) hipLaunchKernelGGL(kern_half, dim3(nfull << 1), dim3(256), lds_half, s, a, b_pad, groups % 8 == 0 ? 1 : 0);
    else hipLaunchKernelGGL(kern_whole, dim3(nfull), dim3(256), G::LDS_BYTES, s, a, b_pad, groups % 8 == 0 ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_wino1d(const ConvArgs& a, int l_in, int b_pad, int item_form, hipStream_t s) {
    if (b_pad < 16 || b_pad % 16 || a.res4_x || a.c1_real != a.c1_pad || (a.c2 != 0) != (a.x2 != nullptr)) return hipErrorInvalidValue;
#define X(L, CIN, CS, COUT) \
    if (l_in == L && a.c1_real == CS && a.c1_real + a.c2 == CIN && a.c_out == COUT) return launch_wino1d_inst<L, CIN, CS, COUT>(a, b_pad, item_form, s);
    CLD_WINO1D_INSTANCES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace cld
