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
// Work decomposition (see cld_kernels.h): a workgroup of NWN*KS waves owns 208 GEMM
// rows (whole agents) x NT = 16*NWN output channels; wave (nw, ks) owns 13 M-tiles
// x 1 N-tile and every KS-th 16-channel group of each K chunk.  A workgroup always
// holds complete GroupNorm groups (all rows of an agent x whole channel groups), so
// GroupNorm + Mish are fused into the epilogue with no cross-workgroup reduction.
// Measured (scripts/ubench/mfma_issue.hip): with ONE wave per SIMD the MFMA /
// ds_read / global_load mix of this loop tops out at ~83 % of the fp32-MFMA rate,
// with TWO at ~89 %; the launcher therefore picks (NWN, KS) per layer and batch so
// that every SIMD holds two waves (two 4-wave workgroups or one 8-wave workgroup
// per CU), and register use is capped at 256 (launch bound 2 waves/SIMD).
//
//   A operand  : activations, channels-last in HBM, staged per K chunk (KC input
//                channels) into a double-buffered LDS image whose rows carry a
//                2-row zero halo between agents, so the 5 taps are 5 shifted
//                ds_read_b128 of the same image and need no boundary tests.
//   B operand  : weights pre-packed on the host in MFMA fragment order; each
//                lane fetches its fragment with ONE coalesced buffer_load_dwordx4 (scalar
//                offset, loop-invariant per-lane offset) per (tap, 16-channel group) --
//                weights never touch LDS.
//   k order    : lane (i, kk) holds channels 4kk..4kk+3 of a 16-channel group and
//                feeds them to 4 successive MFMAs; A and B use the same
//                permutation, so the sum over k is unchanged.
#include "cld_kernels.h"

#ifndef CLD_STORE_AUX
#define CLD_STORE_AUX 16      // cache policy of the output stores: 16 = sc1 (write-through; see the epilogue), 0 = plain, 2 = nt -- A/B builds only
#endif

namespace cld {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
// floats of one A image + its dump row (must match conv_body: ABUFP)
constexpr int conv_image_floats(int l_in, int lm, int stride, int kc, int ain, int hm) {
    const int ag = (208 >> hm) / lm;
    const bool tmap = ain == 0 && hm == 0 && stride == 1 && kc == 32 && l_in == lm && (lm == 13 || lm == 26 || lm == 52);
    const int aex = tmap ? (lm == 26 ? 16 : 0) : ((ain == 0 && stride == 1 && kc == 32) ? 48 : 0);
    return (ag * (l_in + 2) + 2 + 1) * (kc + 8) + ag * aex;
}


__device__ __forceinline__ float mish_f(float x) {
    // x * tanh(softplus(x)) == x * n / (n + 2), n = e^x (e^x + 2): one exp, no cancellation.
    // v_exp_f32 / v_rcp_f32 are 1-ulp instructions: relative error ~2e-7, far inside the parity bar.  The reciprocal is the
    // bare v_rcp_f32 (__builtin_amdgcn_rcpf): __frcp_rn / 1.0f / x compile to the ten-instruction correctly-rounded division
    // (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup), which doubled the VALU work of every epilogue until round 2.
    const float e = __expf(fminf(x, 30.0f));
    const float n = e * (e + 2.0f);
    return x * n * __builtin_amdgcn_rcpf(n + 2.0f);
}

#ifdef CLD_STAMPS
// diagnostic build: in-kernel cycle stamps (never compiled into the shipped library)
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (p.stamps && tid == 0) {                                                                \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            p.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = t_;               \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define STAMP_RT(k)                                                                                \
    do {                                                                                           \
        if (p.stamps && tid == 0) {                                                                \
            unsigned long long t_;                                                                 \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            p.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = t_;               \
        }                                                                                          \
    } while (0)
#else
#define STAMP(k) do {} while (0)
#define STAMP_RT(k) do {} while (0)
#endif

template <int W> struct VecT;
template <> struct VecT<4> { typedef v4f type; };
template <> struct VecT<2> { typedef v2f type; };
template <> struct VecT<1> { typedef float type; };
template <int W> __device__ __forceinline__ float vget(const typename VecT<W>::type& v, int e) { return v[e]; }
template <> __device__ __forceinline__ float vget<1>(const float& v, int) { return v; }
template <int W> __device__ __forceinline__ void vset(typename VecT<W>::type& v, int e, float x) { v[e] = x; }
template <> __device__ __forceinline__ void vset<1>(float& v, int, float x) { v = x; }

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f buf_load16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    // buffer_load_dwordx4 v, voff, rsrc, soff offen: the per-lane part of the address is a loop-invariant VGPR
    // and everything that changes per chunk / iteration is a scalar -- no vector ALU work per load.
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// ---- S22 activation format (f16x2-split precision mode) ------------------------------------------------------
// A value a is kept as hi = fp16(a), lo = fp16(a - hi): 22 mantissa bits in the same 4 bytes as an fp32.  In HBM and
// in the LDS image every 8-channel block of a row is stored as [8 x hi][8 x lo] (16 B + 16 B), so an MFMA A fragment
// (8 consecutive channels per lane for v_mfma_f32_16x16x32_f16) is one ds_read_b128 per plane with no permute.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void s22_encode(const float (&v)[4], h4& hi, h4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float c = fminf(fmaxf(v[e], -65504.0f), 65504.0f);     // saturate instead of producing inf
        const _Float16 h = (_Float16)c;
        hi[e] = h;
        lo[e] = (_Float16)(c - (float)h);
    }
}

// PADC: the input is the 4-channel latent (first conv of the U-Net, k = 5).  Its K is folded over (tap, channel): 20 values =
// five MFMA k-steps per M-tile instead of the 40 a zero-padded 32-channel chunk walked tap by tap would take -- lane (row, kk)
// reads the four channels of row + kk (taps 0..3, one ds_read_b128, four MFMAs) and channel kk of row + 4 (tap 4, one MFMA).
// AIN / AOUT: activation format of the input / of the output and residual (0 = fp32, 1 = S22).  AIN = 1 selects the
// split-precision loop: products hi*hi + hi*lo + lo*hi on the fp16 MFMA (weights pre-split and pre-scaled on the
// host), fp32 accumulation -- measured indistinguishable from the exact-fp32 loop on this network
// (tests/tools/emulate_split.py) at ~3.9x its in-loop rate (scripts/ubench/mfma_issue.hip, V5).
template <int L_IN, int LM, int STRIDE, int NTAPS, int KC, int NWN, int KS, int EPI, int GS, int OSTR, int PADC, int AIN, int AOUT, int HM>
__device__ __forceinline__ void conv_body(const ConvArgs& p, float* lds, const int bx, const int by) {
    // HM = 3: eighth-height tiles, 26 rows = 1.6 M-tiles (2 / 1 agents at L = 13 / 26): the small-batch tile -- at 64 agents
    // the quarter-height launch of a 256-channel layer is 128 workgroups on 256 CUs, each wave a serial chain of 640 MFMAs
    // (9.7 us of the launch's 16.4); the eighth-height launch spreads the same work over every CU
    // HM = 1: half-height tiles -- 104 GEMM rows (8 / 4 / 2 agents at L = 13 / 26 / 52) = 6.5 M-tiles, the last one half
    // empty (its upper 8 rows are computed on row 0's operands and never read back); twice the workgroups of a full tile,
    // taken when a full-height launch would leave the chip under two workgroups per CU
    constexpr int MT = 208 >> HM;                       // HM = 2: quarter-height tiles, 52 rows = 3.25 M-tiles (4 / 2 / 1 agents)
    constexpr int NMT = (MT + 15) / 16;
    constexpr bool SPLIT = AIN == 1;
    static_assert(!SPLIT || HM == 0, "half- / quarter-height tiles: exact-fp32 loop only");
    static_assert(!SPLIT || (KS == 1 && KC % 32 == 0 && PADC == 0), "split-precision loop: whole 32-channel MFMA groups, no K split");
    constexpr int MG = KC / 32;            // split mode: 32-channel MFMA groups per chunk
    constexpr int NTHR = 64 * NWN * KS;
    constexpr int AG = MT / LM;            // agents per workgroup
    constexpr int LP = L_IN + 2;           // LDS rows per agent (2 halo rows shared with the neighbour)
    constexpr int AROWS = AG * LP + 2;
    constexpr int KCP = KC + 8;            // padded LDS row (floats): strides 40 / 72 make the b128 fragment reads of 16 consecutive rows conflict-free
    // ... of 16 CONSECUTIVE LDS rows.  An M-tile's 16 GEMM rows cross agent boundaries, where the LDS row jumps by the 2 halo
    // rows: with row stride 40 floats that shifts the bank pattern by a quarter turn and two of the four lane groups of every
    // ds_read_b128 see 2-way conflicts (46 % of the LDS-array cycles of the dominant kernel were conflicts, round 1).  Two
    // fixes, both free of extra LDS traffic:
    //   TMAP: a full-height tile holds AG = 16 / 8 / 4 agents x 13 / 26 / 52 rows = 13 M-tiles of 16: M-tile m takes rows
    //         RPT m .. RPT m + RPT - 1 (RPT = 16 / AG = 1 / 2 / 4) of EVERY agent, lane i = (agent i % AG, row RPT m + i / AG).
    //         Consecutive lanes are then a whole agent block apart; with the block length 15 rows (L = 13), 28 rows + 16 floats
    //         (L = 26) or 54 rows (L = 52) the 16 lanes of every read group land on 16 different 16-byte bank slots --
    //         conflict-free (brute-forced over all tiles and lane groups), and the fragment address becomes AFFINE in m:
    //         one VGPR + an immediate per M-tile instead of 13 VGPRs.
    //   AEX : the partial-height tiles keep consecutive rows per tile; there every agent's block of rows is followed by 48
    //         extra floats, which makes the jump at an agent boundary (2 rows + 48 floats = 32 sixteen-byte slots) a whole
    //         number of bank turns.
    constexpr int RPT = 16 / (AG > 16 ? 16 : AG);      // rows of one agent per M-tile under TMAP
    constexpr bool TMAP = !SPLIT && HM == 0 && STRIDE == 1 && KC == 32 && L_IN == LM && AG * RPT == 16 && LM % RPT == 0 && NMT * RPT == LM;
    constexpr int AEX = TMAP ? (LM == 26 ? 16 : 0) : ((!SPLIT && STRIDE == 1 && KC == 32) ? 48 : 0);
    constexpr int ASTR = LP * KCP + AEX;   // floats from one agent's first row to the next agent's
    constexpr int ABUF = AROWS * KCP + AG * AEX;      // floats per A image
    constexpr int ABUFP = ABUF + KCP;      // + one dump row for the staging pieces past the tile
    static_assert(ABUFP == conv_image_floats(L_IN, LM, STRIDE, KC, AIN, HM), "launcher and kernel disagree on the LDS image size");
    constexpr int NT = 16 * NWN;
    constexpr int OP = NT + 4;             // padded row of the output tile
    constexpr int NKG = KC / 16;           // 16-channel groups per chunk
    constexpr int KGW = NKG / KS;          // groups per wave per chunk
    static_assert(KGW >= 1 && KGW * KS == NKG, "K split must divide the chunk");
    static_assert(AG * LM == MT, "whole agents per tile");
    constexpr int IN_ROWS = AG * L_IN;
    constexpr int PPR = KC / 4;            // 16-byte pieces per staged row
    constexpr int NPC = IN_ROWS * PPR;
    constexpr int NPIECE = (NPC + NTHR - 1) / NTHR;
    constexpr int NIT = (AIN == 1) ? NTAPS * MG : NTAPS * KGW;   // iterations per chunk per wave: (tap, 16-channel group); split mode: (tap, 32-channel MFMA group)
    static_assert(NTHR % PPR == 0, "a thread keeps one channel piece");

    const int tid = threadIdx.x;
    STAMP(0);
    STAMP_RT(8);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = wave % NWN;
    const int ks = wave / NWN;
    const int b0 = bx * AG;
    const int ntile_g = by * NWN + nw;
    const int ntn = p.c_out >> 4;
    const int nchunk = (p.c1_pad + p.c2) / KC;

    // ---- staging map: piece i of this thread -> (byte offset in the tile's rows, LDS offset) ----
    // Every piece loads and stores unconditionally (no divergent branches, so the compiler can count
    // its vmcnt waits): pieces past the tile read this thread's piece 0 again and land in a dump row
    // behind the image.  Sources are buffer descriptors based at this workgroup's first row; the chunk's
    // channel offset goes into the scalar offset of the load.
    const int pc4 = (tid % PPR) * 4;
    const int stride1 = p.c1_real;                      // == p.c2 when a second source exists (host-checked)
    const bool real = !PADC || pc4 < p.c1_real;
    // A thread's pieces sit NTHR / PPR rows apart, so one VGPR offset serves them all and the row step rides in the
    // scalar offset of the load (the full-height tiles are at the 256-VGPR cap: per-piece offsets were being spilled
    // and reloaded inside the chunk loop, ahead of the prefetch they address).  The last piece keeps a VGPR of its
    // own: it is the only one that can fall past the tile, and the scalar offset is not part of the range check.
    constexpr int RSTEP = NTHR / PPR;
    const int voff0 = ((tid / PPR) * stride1 + (real ? pc4 : 0)) * 4;
    const int pstep = RSTEP * stride1 * 4;
    int voff_last;
    {
        const int idx = tid + NTHR * (NPIECE - 1);
        const int r = (idx < NPC) ? idx / PPR : tid / PPR;
        voff_last = (r * stride1 + (real ? pc4 : 0)) * 4;
    }
    const size_t tile_floats = (size_t)IN_ROWS * stride1;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x1) + (size_t)bx * tile_floats, 0, (int)(tile_floats * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x2 ? p.x2 : p.x1) + (size_t)bx * tile_floats, 0, (int)(tile_floats * 4), 0x00020000);
    // Small tiles (HM >= 2: 4 or 2 M-tiles per wave) are latency-bound, not issue-bound: a chunk's MFMAs last 0.5-1k cycles,
    // less than one trip to L2, so their activations are fetched TWO chunks ahead (two register sets, by chunk parity) and
    // their weight fragments a whole chunk pair ahead (WD below); the throughput tiles keep one set and two fragments (they
    // sit at the 256-register cap and have 1.7k cycles of MFMAs per iteration to hide behind).
    constexpr bool DEEP = HM >= 2 && !SPLIT && !PADC && STRIDE == 1;
    constexpr int NSET = DEEP ? 2 : 1;
    v4f st[NSET][NPIECE];
    auto load_chunk = [&](int c, int set = 0) {
        const int cv = c * KC;                          // virtual input channel of this chunk
        if (cv < p.c1_pad) {                            // wave-uniform: which source feeds the chunk
#pragma unroll
            for (int i = 0; i < NPIECE; ++i) {
                const v4f v = (i == NPIECE - 1) ? buf_load16(rs1, voff_last, PADC ? 0 : cv * 4)
                                                : buf_load16(rs1, voff0, (PADC ? 0 : cv * 4) + i * pstep);
                st[set][i] = real ? v : v4f{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int i = 0; i < NPIECE; ++i)
                st[set][i] = (i == NPIECE - 1) ? buf_load16(rs2, voff_last, (cv - p.c1_pad) * 4)
                                               : buf_load16(rs2, voff0, (cv - p.c1_pad) * 4 + i * pstep);
        }
    };

    // B fragments: slab (16-channel group kgg, tap t) holds [ntn][64 lanes][4]; see pack_conv_weights
    const int ngrp = (p.c1_pad + p.c2) >> 4;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wfrag), 0, PADC ? ntn * 2048 : ngrp * NTAPS * ntn * 1024, 0x00020000);
    const int wlane = lane * 16;
    auto wload = [&](int c, int it) {                   // `it` may run past the chunk
        const int cc = c + it / NIT, ii = it % NIT;
        const int t = ii / KGW;
        const int kgg = cc * NKG + ks + KS * (ii % KGW);
        return buf_load16(rsw, wlane, ((kgg * NTAPS + t) * ntn + ntile_g) * 1024);
    };

    // split mode: slab (32-channel chunk c, tap t, N tile) = [hi: 64 lanes x 8 fp16][lo: 64 lanes x 8 fp16] = 2 KiB
    auto wslab = [&](int c, int it) {                   // iteration it of chunk c = (tap it / MG, group it % MG)
        const int cc = c + it / NIT, ii = it % NIT;
        return (((cc * MG + ii % MG) * NTAPS + ii / MG) * ntn + ntile_g) * 2048;
    };
    auto wload_hi = [&](int c, int it) { return buf_load16(rsw, wlane, wslab(c, it)); };
    auto wload_lo = [&](int c, int it) { return buf_load16(rsw, wlane, wslab(c, it) + 1024); };

    // ---- prologue: chunk 0 and the first weight fragments go out first; the LDS-side address arithmetic
    //      (divisions by the row counts), the accumulator clear and the halo zeroing run under their latency ----
    load_chunk(0);
    if (DEEP && 1 < nchunk) load_chunk(1, 1);
    // B fragments run two (tap, group) iterations ahead of the MFMAs that consume them (WD for the small tiles)
    // (PADC: the whole layer is two fragments per N tile -- taps 0..3 x 4 channels, and tap 4; see pack_latent_conv_weights)
    v4f bq0 = PADC ? buf_load16(rsw, lane * 32, ntile_g * 2048) : (SPLIT ? wload_hi(0, 0) : wload(0, 0));
    v4f bq1 = PADC ? buf_load16(rsw, lane * 32 + 16, ntile_g * 2048)
                   : ((1 / NIT < nchunk) ? (SPLIT ? wload_hi(0, 1) : wload(0, 1)) : bq0);
    v4f bl0 = bq0, bl1 = bq0;                           // lo planes of the two queued weight fragments (split mode)
    if (SPLIT) {
        bl0 = wload_lo(0, 0);
        if (1 / NIT < nchunk) bl1 = wload_lo(0, 1);
    }
    __builtin_amdgcn_sched_barrier(0);                  // keep the loads ahead of the arithmetic below
    int soff[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int idx = tid + NTHR * i;
        const bool ok = idx < NPC;
        const int r = ok ? idx / PPR : tid / PPR;
        const int a = r / L_IN;
        const int l = r - a * L_IN;
        // split mode: odd LDS rows keep the hi / lo planes of every 8-channel block swapped, so the two lanes of a
        // 16-lane read group that land on one 32-byte slot (rows 5 apart mod 8) take different 16-byte halves
        const int pcs = SPLIT ? (pc4 ^ (((2 + a * LP + l) & 1) << 2)) : pc4;
        soff[i] = ok ? a * ASTR + (2 + l) * KCP + pcs : ABUF + pc4;
    }
    int aoff[NMT];
    int aoffy[SPLIT ? NMT : 1];                         // split mode: aoff = plane at +0/+16 for even taps, aoffy = the other plane
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
        int r = 16 * m + (lane & 15);
        if (HM && r >= MT) r = 0;                       // dummy rows of the half-empty last M-tile
        const int a = TMAP ? (lane & 15) % AG : r / LM;
        const int j = TMAP ? RPT * m + (lane & 15) / AG : r - a * LM;
        if (SPLIT) {
            const int row = 2 + a * LP + STRIDE * j + p.off0;      // LDS row of tap 0
            const int blk = row * KCP * 4 + 32 * (lane >> 4);      // bytes: this lane's 8-channel block
            aoff[m] = blk + ((row & 1) ? 16 : 0);                  // hi plane on even taps, lo plane on odd taps
            aoffy[m] = blk + ((row & 1) ? 0 : 16);
        } else {
            aoff[m] = PADC ? (a * ASTR + (2 + j + p.off0 + (lane >> 4)) * KCP) * 4       // channels 0..3 of the row of tap kk = lane >> 4
                           : (a * ASTR + (2 + STRIDE * j + p.off0) * KCP + 4 * (lane >> 4) + 16 * ks) * 4;
        }
    }
    v4f acc[NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) acc[m] = v4f{0.f, 0.f, 0.f, 0.f};
    auto store_chunk = [&](int buf, int set = 0) {
        float* A = lds + buf * ABUFP;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) *reinterpret_cast<v4f*>(A + soff[i]) = st[set][i];
    };
    // zero the gaps: the two leading rows, and behind every agent its two halo rows (+ the AEX floats)
    constexpr int GQ = (2 * KCP + AEX) / 4;             // 16-byte pieces per gap
    for (int i = tid; i < 2 * (AG + 1) * GQ; i += NTHR) {
        const int q = i % GQ, g = (i / GQ) % (AG + 1), buf = i / (GQ * (AG + 1));      // gap index: 0 = leading rows, g > 0 = behind agent g-1
        if (g == 0 && q >= 2 * KCP / 4) continue;
        const int at = g == 0 ? 0 : (g - 1) * ASTR + (2 + L_IN) * KCP;
        *reinterpret_cast<v4f*>(lds + buf * ABUFP + at + q * 4) = v4f{0.f, 0.f, 0.f, 0.f};
    }
    store_chunk(0);
    __syncthreads();

    STAMP(1);
    // K loop.  Per chunk each wave runs NIT (tap, group) iterations of 52 MFMAs.  The A fragments of
    // iteration it+1 are read from LDS while iteration it's MFMAs issue: one ds_read_b128 behind every
    // 4th MFMA, pinned with sched_barrier so the reads are never bunched at the end of an iteration.
    // The next chunk's image is written (and the workgroup barrier taken) before the LAST iteration of
    // the current chunk, so that iteration prefetches the next chunk's first fragments: no fragment-load
    // bubble at chunk boundaries either.  WAR: the image written at (c, NIT-2) was last read at
    // (c-1, NIT-2), i.e. before the barrier of chunk c-1.
    constexpr bool LEAN = STRIDE == 2 && !SPLIT;   // stride-2 tiles stage twice the rows (13 pieces/thread): keep ONE
                                                   // fragment buffer there and let the partner wave cover the LDS latency
    constexpr bool XPF = NIT >= 2 && !LEAN;        // cross-chunk fragment prefetch
    constexpr int WIT = XPF ? NIT - 2 : 0;         // iteration after which the next image is written
    constexpr int CUNR = 2;     // chunk pairs: image index and fragment-buffer parity are compile-time, so every
                                // LDS fragment address is one loop-invariant VGPR + an immediate offset
    const char* ldsb = reinterpret_cast<const char*>(lds);
    if constexpr (PADC) {
        static_assert(NTAPS == 5 && !SPLIT && STRIDE == 1 && KS <= 2, "the folded loop is the latent's k = 5 convolution");
        // with a K split the first rank takes taps 0..3 and the last rank tap 4 (KS = 1: one wave does both)
        if (ks == 0) {
#pragma unroll
            for (int m = 0; m < NMT; ++m) {
                const v4f a4 = *reinterpret_cast<const v4f*>(ldsb + aoff[m]);
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[sidx], bq0[sidx], acc[m], 0, 0, 0);
            }
        }
        if (ks == KS - 1) {
            const int t4 = ((4 - (lane >> 4)) * KCP + (lane >> 4)) * 4;      // from (row + kk, channel 0) to (row + 4, channel kk)
#pragma unroll
            for (int m = 0; m < NMT; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(*reinterpret_cast<const float*>(ldsb + aoff[m] + t4), bq1[0], acc[m], 0, 0, 0);
        }
    } else if constexpr (SPLIT) {
        // Split-precision loop: one iteration = one tap of a 32-channel chunk; per M-tile two ds_read_b128 (hi / lo
        // plane, for the NEXT iteration) and three fp16 MFMAs (K = 32).  Same chunk pipeline as the fp32 loop.
        v4f ah[2][NMT], al[2][NMT];
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
            ah[0][m] = *reinterpret_cast<const v4f*>(ldsb + aoff[m]);
            al[0][m] = *reinterpret_cast<const v4f*>(ldsb + aoffy[m]);
        }
        for (int c0 = 0; c0 < nchunk; c0 += CUNR) {
#pragma unroll
            for (int cu = 0; cu < CUNR; ++cu) {
                const int c = c0 + cu;
                if (c >= nchunk) break;
                const int par = XPF ? (cu * NIT) & 1 : 0;
                const int abase = cu * ABUFP * 4;
                const int anext = (cu ^ 1) * ABUFP * 4;
                const bool more = (c + 1 < nchunk);
                if (!XPF && c > 0) {
#pragma unroll
                    for (int m = 0; m < NMT; ++m) {
                        ah[0][m] = *reinterpret_cast<const v4f*>(ldsb + aoff[m] + abase);
                        al[0][m] = *reinterpret_cast<const v4f*>(ldsb + aoffy[m] + abase);
                    }
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int cur = (it + par) & 1;
                    const h8 bh = __builtin_bit_cast(h8, bq0), bl = __builtin_bit_cast(h8, bl0);
                    bq0 = bq1; bl0 = bl1;
                    if (c + (it + 2) / NIT < nchunk) { bq1 = wload_hi(c, it + 2); bl1 = wload_lo(c, it + 2); }
                    if (it == 0 && more) load_chunk(c + 1);
                    const bool in_chunk = it + 1 < NIT;
                    const bool fetch = in_chunk || (XPF && more);
                    const int tn = in_chunk ? (it + 1) / MG : 0, gn = in_chunk ? (it + 1) % MG : 0;   // next iteration's tap / group
                    const int src = (in_chunk ? abase : anext) + tn * KCP * 4 + gn * 128;
#pragma unroll
                    for (int g = 0; g < NMT; ++g) {
                        if (fetch) {       // the planes trade places on odd taps (row parity flips with the tap offset)
                            ah[cur ^ 1][g] = *reinterpret_cast<const v4f*>(ldsb + ((tn & 1) ? aoffy[g] : aoff[g]) + src);
                            al[cur ^ 1][g] = *reinterpret_cast<const v4f*>(ldsb + ((tn & 1) ? aoff[g] : aoffy[g]) + src);
                        }
                        const h8 xh = __builtin_bit_cast(h8, ah[cur][g]), xl = __builtin_bit_cast(h8, al[cur][g]);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bh, acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bl, acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, bh, acc[g], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (it == WIT && more) {
                        store_chunk(cu ^ 1);
                        __syncthreads();
                    }
                }
            }
        }
    } else {
    v4f af[2][NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) af[0][m] = *reinterpret_cast<const v4f*>(ldsb + aoff[m]);
    // weight-fragment ring: slot (iteration index within the chunk pair) % WD; WD divides 2 NIT, so the slot of every
    // iteration of the unrolled pair is a compile-time constant.  WD = 2 is the two-deep queue of the throughput tiles.
    constexpr int WD = DEEP ? 2 * NIT : 2;
    v4f bq[WD];
    bq[0] = bq0; bq[1 % WD] = bq1;
#pragma unroll
    for (int i = 2; i < WD; ++i) bq[i] = (i / NIT < nchunk) ? wload(0, i) : bq0;

    for (int c0 = 0; c0 < nchunk; c0 += CUNR) {
#pragma unroll
        for (int cu = 0; cu < CUNR; ++cu) {
            const int c = c0 + cu;
            if (c >= nchunk) break;
            const int par = XPF ? (cu * NIT) & 1 : 0;  // fragment buffer holding this chunk's iteration 0
            const int abase = cu * ABUFP * 4;          // bytes; chunk c lives in image c & 1 == cu
            const int anext = (cu ^ 1) * ABUFP * 4;
            const bool more = (c + 1 < nchunk);
            if (!XPF && !LEAN && c > 0) {
#pragma unroll
                for (int m = 0; m < NMT; ++m) af[0][m] = *reinterpret_cast<const v4f*>(ldsb + aoff[m] + abase);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int cur = LEAN ? 0 : (it + par) & 1;
                if (LEAN && !(c == 0 && it == 0)) {
#pragma unroll
                    for (int m = 0; m < NMT; ++m)
                        af[0][m] = *reinterpret_cast<const v4f*>(ldsb + aoff[m] + abase + ((it / KGW) * KCP + 16 * KS * (it % KGW)) * 4);
                }
                const int slot = (cu * NIT + it) % WD;
                const v4f bcur = bq[slot];
                if (c + (it + WD) / NIT < nchunk) bq[slot] = wload(c, it + WD);
                if (DEEP) { if (it == 0 && c + 2 < nchunk) load_chunk(c + 2, cu); }      // two chunks ahead, into the set chunk c came from
                else if (it == 0 && more) load_chunk(c + 1);      // next chunk's activations: in flight under this chunk's MFMAs
                const bool in_chunk = it + 1 < NIT;
                const bool fetch = !LEAN && (in_chunk || (XPF && more));
                const int src = in_chunk ? abase + (((it + 1) / KGW) * KCP + 16 * KS * ((it + 1) % KGW)) * 4 : anext;
#pragma unroll
                for (int g = 0; g < NMT; ++g) {
                    if (fetch) af[cur ^ 1][g] = *reinterpret_cast<const v4f*>(ldsb + aoff[g] + src);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int idx = 4 * g + q, sidx = idx / NMT, m = idx % NMT;
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][m][sidx], bcur[sidx], acc[m], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (it == (LEAN ? NIT - 1 : WIT) && more) {
                    store_chunk(cu ^ 1, DEEP ? (cu ^ 1) : 0);
                    __syncthreads();
                }
            }
#ifdef CLD_STAMPS
            if (c < 3) STAMP(10 + c);
#endif
        }
    }
    }   // exact-fp32 loop
    STAMP(2);

    // ---- epilogue mapping: TPP lanes per (agent, group); every lane handles float4 channel vectors of NV rows
    //      (16-byte LDS reads, residual loads and stores in both tilings; the last row slot is empty for some
    //      lanes when the rows of an agent do not divide evenly) ----
    constexpr int VW = 4;
    constexpr int NG = NT / GS;
    constexpr int PAIRS = AG * NG;
    constexpr int TPP = NTHR / PAIRS;
    constexpr int VPR = GS / VW;                     // vectors per row of a group
    constexpr int RPI = TPP / VPR;                   // rows covered per iteration by the lanes of a pair
    constexpr int NV = (LM + RPI - 1) / RPI;         // row slots per lane: 13 (tiling A) or 7 (tiling B)
    constexpr bool RAGGED = NV * RPI != LM;
    static_assert(PAIRS * TPP == NTHR && VPR * RPI == TPP && VPR >= 1, "epilogue mapping");
    typedef typename VecT<VW>::type vec_t;

    const int pair = tid / TPP, q = tid % TPP;
    const int a = pair / NG, g = pair % NG;
    const int ch = g * GS + (q % VPR) * VW;          // channel within the tile
    const int n = by * NT + ch;                      // global output channel
    const int jr = q / VPR;
    const bool last_ok = !RAGGED || (RPI * (NV - 1) + jr < LM);      // does this lane own a row in the last slot?
    const size_t obase = ((size_t)(b0 + a) * p.ly + (OSTR * jr + p.orow0)) * p.c_out + n;
    const size_t ostep = (size_t)OSTR * RPI * p.c_out;

    // epilogue operands are fetched now, so their latency hides under the tile exchange below
    const vec_t bias = *reinterpret_cast<const vec_t*>(p.bias + n);
    vec_t gam = bias, bet = bias, cbv = bias, tbv = bias;
    if (EPI == EPI_GN_MISH) {
        gam = *reinterpret_cast<const vec_t*>(p.gamma + n);
        bet = *reinterpret_cast<const vec_t*>(p.beta + n);
        if (p.cbias) cbv = *reinterpret_cast<const vec_t*>(p.cbias + (size_t)(b0 + a) * p.cb_stride + n);
        if (p.tbias) tbv = *reinterpret_cast<const vec_t*>(p.tbias + n);
    }
    // S22 tensors: the 4 channels of a vector sit at (fp32 byte offset) - 2 * (n & 7) in the hi plane, + 16 in the lo plane
    const int s22_adj = AOUT == 1 ? -2 * (n & 7) : 0;
    vec_t rv[NV];
    if (p.res4_x) {
        // the first block's residual: residual_conv = Conv1d(4 -> 64, k = 1) of the block input (temporal.py:32-34), the latent
        // itself -- 16 FMAs per output vector here instead of a [B,52,64] tensor written and read back through HBM
        v4f w4[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) w4[e] = *reinterpret_cast<const v4f*>(p.res4_w + (size_t)(n + e) * 4);
        const vec_t b4 = *reinterpret_cast<const vec_t*>(p.res4_b + n);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = (i < NV - 1 || last_ok) ? RPI * i + jr : jr;
            const v4f x4 = *reinterpret_cast<const v4f*>(p.res4_x + ((size_t)(b0 + a) * LM + j) * 4);
#pragma unroll
            for (int e = 0; e < VW; ++e)
                vset<VW>(rv[i], e, vget<VW>(b4, e) + w4[e][0] * x4[0] + w4[e][1] * x4[1] + w4[e][2] * x4[2] + w4[e][3] * x4[3]);
        }
    } else if (p.res) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (!(i < NV - 1 || last_ok)) { rv[i] = vec_t{0.f, 0.f, 0.f, 0.f}; continue; }
            if (AOUT == 1) {
                // lanes q (channels 8k..8k+3) and q^1 (8k+4..8k+7) share one 8-channel block: the even lane loads its
                // 16-byte hi plane, the odd lane the lo plane, and they swap halves so each ends up with hi4 + lo4
                const char* rp = reinterpret_cast<const char*>(p.res + obase + i * ostep) + s22_adj + ((n & 4) ? 8 : 0);
                const h8 pl = *reinterpret_cast<const h8*>(rp);          // even lane: hi[0..7]; odd lane: lo[0..7]
                const u4 pw = __builtin_bit_cast(u4, pl);
                // what the partner needs from me: even lane gives hi[4..7] (words 2,3); odd lane gives lo[0..3] (words 0,1)
                const unsigned g0 = (n & 4) ? pw[0] : pw[2], g1 = (n & 4) ? pw[1] : pw[3];
                const unsigned r0 = __shfl_xor((int)g0, 1), r1 = __shfl_xor((int)g1, 1);
                u2 hw, lw;
                if (n & 4) { hw = u2{r0, r1}; lw = u2{pw[2], pw[3]}; }   // odd: partner sent hi[4..7]; own lo[4..7]
                else       { hw = u2{pw[0], pw[1]}; lw = u2{r0, r1}; }   // even: own hi[0..3]; partner sent lo[0..3]
                const h4 rh = __builtin_bit_cast(h4, hw), rl = __builtin_bit_cast(h4, lw);
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[i][e] = (float)rh[e] + (float)rl[e];
            } else {
                rv[i] = *reinterpret_cast<const vec_t*>(p.res + obase + i * ostep);
            }
        }
    }

    __syncthreads();               // every wave is done reading the images before the output tile overwrites them
    STAMP(3);

    // ---- accumulators -> LDS tiles [KS][208][OP] (aliasing the A images): every K-split rank stores its own
    //      partial tile, one barrier, and the epilogue threads add the KS partials while reading ----
    float* O = lds;
    constexpr int OTILE = 16 * NMT * OP;               // (= MT rows for full tiles; the half tile's last M-tile has 8 dummy rows)
    static_assert((size_t)KS * OTILE * 4 <= 160 * 1024, "partial output tiles must fit in LDS (the launcher sizes the allocation)");
    {
        const int col = nw * 16 + (lane & 15);
        const int rb = 4 * (lane >> 4);
        float* Ok = O + ks * OTILE;
#pragma unroll
        for (int m = 0; m < NMT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)      // accumulator row rb + r of M-tile m is GEMM row (agent, position) = ...
                Ok[(TMAP ? ((rb + r) % AG) * LM + RPT * m + (rb + r) / AG : 16 * m + rb + r) * OP + col] = acc[m][r];
    }
    __syncthreads();
    STAMP(4);

    float v[NV][VW];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = (i < NV - 1 || last_ok) ? RPI * i + jr : jr;      // an empty slot re-reads row jr and is masked below
        const vec_t o = *reinterpret_cast<const vec_t*>(O + (a * LM + j) * OP + ch);
#pragma unroll
        for (int e = 0; e < VW; ++e)        // split mode: the weights were scaled by 1 / wscale_inv (a power of two) before splitting
            v[i][e] = SPLIT ? vget<VW>(o, e) * p.wscale_inv + vget<VW>(bias, e) : vget<VW>(o, e) + vget<VW>(bias, e);
#pragma unroll
        for (int k = 1; k < KS; ++k) {      // K-split partials, added in rank order
            const vec_t ok = *reinterpret_cast<const vec_t*>(O + k * OTILE + (a * LM + j) * OP + ch);
#pragma unroll
            for (int e = 0; e < VW; ++e) v[i][e] += vget<VW>(ok, e);
        }
    }

    if (EPI == EPI_GN_MISH) {
        // GroupNorm over (GS channels x LM rows) of one agent: two-pass, biased variance, eps 1e-5
        // (torch.nn.GroupNorm as used in diffuser_helpers.py:61).  The TPP lanes of an (agent, group) pair are one wave or less,
        // except in the eighth-height tiles (HM = 3), where a pair spreads over TPP / 64 waves: those add their wave sums through
        // a few floats of LDS behind the output tiles.
        constexpr int TPW = TPP > 64 ? 64 : TPP;         // lanes of a pair inside one wave
        float* xch = lds + KS * OTILE;                   // [2][waves] (the launcher allocates 64 floats behind the tiles)
        auto pair_sum = [&](float x, int slot) {
#pragma unroll
            for (int o = 1; o < TPW; o <<= 1) x += __shfl_xor(x, o);
            if constexpr (TPP > 64) {
                if (lane == 0) xch[slot * 8 + wave] = x;
                __syncthreads();
                constexpr int WPP = TPP / 64;            // waves per pair
                const int w0 = (wave / WPP) * WPP;
                x = 0.f;
#pragma unroll
                for (int k = 0; k < WPP; ++k) x += xch[slot * 8 + w0 + k];
            }
            return x;
        };
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VW; ++e) s += (i < NV - 1 || last_ok) ? v[i][e] : 0.f;
        s = pair_sum(s, 0);
        const float mean = s * (1.0f / (float)(GS * LM));
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VW; ++e) { const float d = v[i][e] - mean; ss += (i < NV - 1 || last_ok) ? d * d : 0.f; }
        ss = pair_sum(ss, 1);
        const float rstd = 1.0f / sqrtf(ss * (1.0f / (float)(GS * LM)) + 1e-5f);
        float add[VW], sc[VW], sh[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            add[e] = (p.cbias ? vget<VW>(cbv, e) : 0.f) + (p.tbias ? vget<VW>(tbv, e) : 0.f);
            sc[e] = rstd * vget<VW>(gam, e);
            sh[e] = vget<VW>(bet, e);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VW; ++e) v[i][e] = mish_f((v[i][e] - mean) * sc[e] + sh[e]) + add[e];
    }

    STAMP(5);
    // Output rows go out as write-through (sc1) 16-byte stores: the tensor is read next by other CUs after a kernel
    // boundary anyway, and lines left dirty in L2 are written back AT the boundary (~B / 6 TB/s on top of the 1.4 us);
    // measured +1.5 % end to end over plain stores at B = 1,024 (non-temporal stores: no change).
    const size_t ybase = (size_t)b0 * p.ly * p.c_out;
    const __amdgpu_buffer_rsrc_t rsy =
        __builtin_amdgcn_make_buffer_rsrc(p.y + ybase, 0, (int)((size_t)AG * p.ly * p.c_out * 4), 0x00020000);
    const int yoff = (int)((obase - ybase) * 4), ystep = (int)(ostep * 4);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        vec_t o;
#pragma unroll
        for (int e = 0; e < VW; ++e) vset<VW>(o, e, (p.res || p.res4_x) ? v[i][e] + vget<VW>(rv[i], e) : v[i][e]);
        if (i < NV - 1 || last_ok) {
            if (AOUT == 1) {
                float ov[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = o[e];
                h4 hh, ll;
                s22_encode(ov, hh, ll);
                // even lane stores the block's 16-byte hi plane (own hi4 | partner's hi4), odd lane the lo plane
                const u2 hw = __builtin_bit_cast(u2, hh), lw = __builtin_bit_cast(u2, ll);
                const unsigned g0 = (n & 4) ? hw[0] : lw[0], g1 = (n & 4) ? hw[1] : lw[1];
                const unsigned r0 = __shfl_xor((int)g0, 1), r1 = __shfl_xor((int)g1, 1);
                const u4 pw = (n & 4) ? u4{r0, r1, lw[0], lw[1]} : u4{hw[0], hw[1], r0, r1};
                __builtin_amdgcn_raw_buffer_store_b128(pw, rsy, yoff + s22_adj + ((n & 4) ? 8 : 0) + i * ystep, 0, CLD_STORE_AUX);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o), rsy, yoff + i * ystep, 0, /*aux: sc1*/ CLD_STORE_AUX);
            }
        }
    }
    STAMP(6);
    STAMP_RT(9);
}

// Workgroup -> tile.  The NB = c_out / NT workgroups that own the N tiles of one row block stage the SAME activation rows.
// Launched as a (row blocks, N tiles) grid they are a whole row of the grid apart in dispatch order and never meet in a cache:
// at 4,096 rows the dominant kernel fetched 207 MB per launch for 54.5 MB of input (round-2 PMC, profiles/r02/traffic.json
// before this change).  With a 1-D grid and this mapping they get block ids L, L + 8, L + 16, ...: consecutive in dispatch
// order AND -- blocks are dealt round-robin over the 8 XCDs -- on the same XCD, so three of the four reads of a row block
// hit that XCD's L2.  Placement is a speed assumption only; any placement computes the same result.
// Measured (interleaved A/B of whole U-Net evaluations, one process): -0.7 % at 4,096 rows per launch, where a launch is two
// or more generations of workgroups; +0.3 % at 1,024 / 2,048 rows, where every workgroup of the launch is resident at once and
// the four readers of a row block are better off spread over four L2s -- so the launcher asks for it (ConvArgs::xcd_map) only
// when the grid exceeds the resident workgroup slots.
__device__ __forceinline__ void tile_of_block(int L, int nbx, int nb, int xcd_map, int& bx, int& by) {
    if (!xcd_map) { bx = L % nbx; by = L / nbx; return; }      // row blocks fastest, N tiles slowest
    const int full = (nbx >> 3) << 3;                  // row blocks covered by whole groups of 8
    if (L < full * nb) {
        by = (L >> 3) % nb;
        bx = (L / (8 * nb)) * 8 + (L & 7);
    } else {
        const int r = nbx - full, l2 = L - full * nb;
        bx = full + l2 % r;
        by = l2 / r;
    }
}

// register budget: 256 (two waves per SIMD) for the exact-fp32 loop, which needs the partner wave to hide its non-MFMA
// issue; 512 for the split-precision loop, which keeps two full fragment sets (hi + lo) in flight and is not MFMA-bound
template <int L_IN, int LM, int STRIDE, int NTAPS, int KC, int NWN, int KS, int EPI, int GS, int OSTR, int PADC, int AIN, int AOUT, int HM>
__global__ __launch_bounds__(64 * NWN * KS, AIN == 1 ? 1 : 2) void conv_block_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nb = p.c_out / (16 * NWN);
    int bx, by;
    tile_of_block(blockIdx.x, gridDim.x / nb, nb, p.xcd_map, bx, by);
    conv_body<L_IN, LM, STRIDE, NTAPS, KC, NWN, KS, EPI, GS, OSTR, PADC, AIN, AOUT, HM>(p, lds, bx, by);
}

// Two launches that do not depend on each other and share a grid shape, merged into one: blockIdx.z picks the
// role.  Used for a residual block's first conv (k5 + GroupNorm + Mish) next to its 1x1 residual projection
// (both read only the block input), and for the two output-parity halves of a transposed conv.  Saves the
// dependent-launch boundary (~1.4 us each, 7 per U-Net evaluation); the per-workgroup prologue/epilogue cost is
// NOT saved -- the second role's workgroups still form their own generation on each CU.
struct ConvPairArgs { ConvArgs a, b; };
template <int L_IN, int LM, int KC, int NWN, int KS, int GS, int OSTR, int PADC,
          int NTAPS_A, int EPI_A, int NTAPS_B, int EPI_B, int AIN, int AOUT, int HM>
__global__ __launch_bounds__(64 * NWN * KS, AIN == 1 ? 1 : 2) void conv_pair_kernel(const ConvPairArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (interleaving the two roles along x, alone or in XCD-balanced groups of 8, measured 9 % slower end to end)
    const int nb = p.a.c_out / (16 * NWN);
    int bx, by;
    tile_of_block(blockIdx.x, gridDim.x / nb, nb, p.a.xcd_map, bx, by);
    if (blockIdx.z == 0) conv_body<L_IN, LM, 1, NTAPS_A, KC, NWN, KS, EPI_A, GS, OSTR, PADC, AIN, AOUT, HM>(p.a, lds, bx, by);
    else                 conv_body<L_IN, LM, 1, NTAPS_B, KC, NWN, KS, EPI_B, GS, OSTR, PADC, AIN, AOUT, HM>(p.b, lds, bx, by);
}

// ---------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------
// Experiments only (cld_debug_lds_floor): a lower bound on the dynamic LDS a launch asks for, to steer how many
// workgroups of which stream can share a CU.  0 = off.
static size_t g_lds_floor = 0;
void set_lds_floor(size_t bytes) { g_lds_floor = bytes; }
static inline size_t lds_request(size_t need) { return g_lds_floor > need ? (g_lds_floor < 160 * 1024 ? g_lds_floor : 160 * 1024) : need; }

template <int L_IN, int LM, int STRIDE, int NTAPS, int KC, int NWN, int KS, int EPI, int GS, int OSTR, int PADC, int AIN, int AOUT, int HM>
static hipError_t launch_inst(const ConvArgs& a, int b_pad, hipStream_t s) {
    constexpr int AG = (208 >> HM) / LM;
    constexpr int ABUF = conv_image_floats(L_IN, LM, STRIDE, KC, AIN, HM);   // image + dump row
    constexpr int OTILE = KS * 16 * (((208 >> HM) + 15) / 16) * (16 * NWN + 4);   // the epilogue's partial output tiles alias the A images
    constexpr size_t lds_bytes = sizeof(float) * ((size_t)(2 * ABUF > OTILE ? 2 * ABUF : OTILE) + 64);   // + the GroupNorm exchange slots of the eighth-height tiles
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    auto kern = conv_block_kernel<L_IN, LM, STRIDE, NTAPS, KC, NWN, KS, EPI, GS, OSTR, PADC, AIN, AOUT, HM>;
    static unsigned long long attr_done = 0;      // one bit per device (a second handle on another device sets it there too)
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(kern), 160 * 1024, &attr_done); e != hipSuccess) return e;
    dim3 grid((b_pad / AG) * (a.c_out / (16 * NWN)), 1, 1);
    ConvArgs aa = a;
    aa.xcd_map = (int)grid.x > 256 * (NWN * KS == 8 ? 1 : 2);        // more workgroups than resident slots (tile_of_block)
    hipLaunchKernelGGL(kern, grid, dim3(64 * NWN * KS), lds_request(lds_bytes), s, aa);
    return hipGetLastError();
}

template <int L_IN, int LM, int KC, int NWN, int KS, int GS, int OSTR, int PADC, int NTAPS_A, int EPI_A, int NTAPS_B, int EPI_B, int AIN, int AOUT, int HM>
static hipError_t launch_pair_inst(const ConvArgs& a, const ConvArgs& b, int b_pad, hipStream_t s) {
    constexpr int AG = (208 >> HM) / LM;
    constexpr int ABUF = conv_image_floats(L_IN, LM, 1, KC, AIN, HM);
    constexpr int OTILE = KS * 16 * (((208 >> HM) + 15) / 16) * (16 * NWN + 4);
    constexpr size_t lds_bytes = sizeof(float) * ((size_t)(2 * ABUF > OTILE ? 2 * ABUF : OTILE) + 64);
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    auto kern = conv_pair_kernel<L_IN, LM, KC, NWN, KS, GS, OSTR, PADC, NTAPS_A, EPI_A, NTAPS_B, EPI_B, AIN, AOUT, HM>;
    static unsigned long long attr_done = 0;      // one bit per device (a second handle on another device sets it there too)
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(kern), 160 * 1024, &attr_done); e != hipSuccess) return e;
    ConvPairArgs pa{a, b};
    dim3 grid((b_pad / AG) * (a.c_out / (16 * NWN)), 1, 2);
    pa.a.xcd_map = pa.b.xcd_map = 2 * (int)grid.x > 256 * (NWN * KS == 8 ? 1 : 2);
    hipLaunchKernelGGL(kern, grid, dim3(64 * NWN * KS), lds_request(lds_bytes), s, pa);
    return hipGetLastError();
}

// pairs: (L_IN, LM, KC, NWN, KS, GS, OSTR, PADC, NTAPS_A, EPI_A, NTAPS_B, EPI_B, AIN, AOUT), stride 1 on both sides
#define CLD_PAIR_INSTANCES(X)                                         \
    X(26, 26, 64, 2, 4, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 3) \
    X(13, 13, 64, 2, 4, 32, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 3) \
    X(13, 13, 64, 2, 4, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 3) \
    X(26, 26, 64, 2, 4, 8, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 3) \
    X(13, 13, 64, 2, 4, 16, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 0, 0, 3) \
    X(26, 26, 32, 4, 1, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 0) \
    X(26, 26, 32, 2, 2, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 0) \
    X(26, 26, 32, 2, 2, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 1) \
    X(26, 26, 32, 2, 2, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 2) \
    X(26, 26, 32, 2, 2, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 3) \
    X(13, 13, 32, 4, 1, 32, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 0) \
    X(13, 13, 32, 2, 2, 32, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 0) \
    X(13, 13, 32, 2, 2, 32, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 1) \
    X(13, 13, 32, 2, 2, 32, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 2) \
    X(13, 13, 32, 2, 2, 32, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 3) \
    X(13, 13, 32, 4, 1, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 0) \
    X(13, 13, 32, 2, 2, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 0) \
    X(13, 13, 32, 2, 2, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 1) \
    X(13, 13, 32, 2, 2, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 2) \
    X(13, 13, 32, 2, 2, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 3) \
    X(26, 26, 32, 4, 1, 8, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 0) \
    X(26, 26, 32, 2, 2, 8, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 0) \
    X(26, 26, 32, 2, 2, 8, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 1) \
    X(26, 26, 32, 2, 2, 8, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 2) \
    X(26, 26, 32, 2, 2, 8, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 0, 0, 3) \
    X(13, 13, 32, 4, 1, 16, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 0, 0, 0) \
    X(13, 13, 32, 2, 2, 16, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 0, 0, 0) \
    X(13, 13, 32, 2, 2, 16, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 0, 0, 1) \
    X(13, 13, 32, 2, 2, 16, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 0, 0, 2) \
    X(13, 13, 32, 2, 2, 16, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 0, 0, 3) \
    X(26, 26, 32, 4, 1, 8, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 0, 0, 0) \
    X(26, 26, 32, 2, 2, 8, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 0, 0, 0) \
    X(26, 26, 32, 2, 2, 8, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 0, 0, 1) \
    X(26, 26, 32, 2, 2, 8, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 0, 0, 2) \
    X(26, 26, 32, 2, 2, 8, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 0, 0, 3) \
    X(26, 26, 64, 4, 1, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 1, 1, 0) \
    X(13, 13, 64, 4, 1, 32, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 1, 1, 0) \
    X(13, 13, 64, 4, 1, 16, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 1, 1, 0) \
    X(26, 26, 64, 4, 1, 8, 1, 0, 5, EPI_GN_MISH, 1, EPI_BIAS, 1, 1, 0) \
    X(13, 13, 64, 4, 1, 16, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 1, 1, 0) \
    X(26, 26, 64, 4, 1, 8, 2, 0, 2, EPI_BIAS, 2, EPI_BIAS, 1, 1, 0)

static inline bool pair_is(const ConvGeom& a, const ConvGeom& b, int l_in, int lm, int kc, int nwn, int ks, int gs, int ostr,
                           int padc, int ntaps_a, int epi_a, int ntaps_b, int epi_b, int ain, int aout, int half) {
    auto common = [&](const ConvGeom& g) {
        return g.l_in == l_in && g.lm == lm && g.stride == 1 && g.kc == kc && g.nwn == nwn && g.ks == ks && g.gs == gs &&
               g.ostr == ostr && g.padc == padc && g.ain == ain && g.aout == aout && g.half == half;
    };
    return common(a) && common(b) && a.ntaps == ntaps_a && a.epi == epi_a && b.ntaps == ntaps_b && b.epi == epi_b;
}

bool conv_pair_supported(const ConvGeom& a, const ConvGeom& b) {
#define X(c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15) if (pair_is(a, b, c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15)) return true;
    CLD_PAIR_INSTANCES(X)
#undef X
    return false;
}

hipError_t launch_conv_pair(const ConvGeom& ga, const ConvArgs& a, const ConvGeom& gb, const ConvArgs& b, int b_pad, hipStream_t s) {
    if (a.c_out != b.c_out) return hipErrorInvalidValue;
#define X(c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15) \
    if (pair_is(ga, gb, c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15)) \
        return launch_pair_inst<c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12, c13, c14, c15>(a, b, b_pad, s);
    CLD_PAIR_INSTANCES(X)
#undef X
    return hipErrorInvalidValue;
}

// (L_IN, LM, STRIDE, NTAPS, KC, NWN, KS, EPI, GS, OSTR, PADC, AIN, AOUT)
//   tilings: A = (KC 32, NWN 4, KS 1) 64 columns, 4 waves   -- large batches (>= 2 workgroups per CU anyway)
//            B = (KC 32, NWN 2, KS 2) 32 columns, 4 waves   -- twice the workgroups of A
//            C = (KC 32, NWN 4, KS 2) 64 columns, 8 waves   -- the 256-channel k5 blocks at 1,024..2,047 agents (one per CU)
//            D = (KC 64, NWN 2, KS 4) 32 columns, 8 waves, eighth-height tiles only -- small batches whose launches are at most
//                one workgroup per CU: the second wave per SIMD covers the chunk barriers and LDS latencies a lone wave exposes
//   (the template also supports KC 64 / NWN 2 / KS 4, measured slower than B, and NWN 8 / KS 1 -- 128 columns in an
//    8-wave workgroup -- measured equal to A at 2,048 and 4,096 agents; neither is built)
#define CLD_CONV_INSTANCES(X)                            \
    X(26, 26, 1, 5, 64, 2, 4, EPI_GN_MISH, 16, 1, 0, 0, 0, 3) \
    X(13, 13, 1, 5, 64, 2, 4, EPI_GN_MISH, 32, 1, 0, 0, 0, 3) \
    X(13, 13, 1, 5, 64, 2, 4, EPI_GN_MISH, 16, 1, 0, 0, 0, 3) \
    X(26, 26, 1, 5, 64, 2, 4, EPI_GN_MISH, 8, 1, 0, 0, 0, 3) \
    X(26, 26, 1, 1, 64, 2, 4, EPI_BIAS, 16, 1, 0, 0, 0, 3) \
    X(13, 13, 1, 1, 64, 2, 4, EPI_BIAS, 32, 1, 0, 0, 0, 3) \
    X(13, 13, 1, 1, 64, 2, 4, EPI_BIAS, 16, 1, 0, 0, 0, 3) \
    X(26, 26, 1, 1, 64, 2, 4, EPI_BIAS, 8, 1, 0, 0, 0, 3) \
    X(13, 13, 1, 2, 64, 2, 4, EPI_BIAS, 16, 2, 0, 0, 0, 3) \
    X(52, 52, 1, 5, 32, 4, 1, EPI_GN_MISH, 8, 1, 1, 0, 0, 0) \
    X(52, 52, 1, 5, 32, 2, 2, EPI_GN_MISH, 8, 1, 1, 0, 0, 0) \
    X(52, 52, 1, 5, 32, 2, 2, EPI_GN_MISH, 8, 1, 1, 0, 0, 1) \
    X(52, 52, 1, 5, 32, 2, 2, EPI_GN_MISH, 8, 1, 1, 0, 0, 2) \
    X(52, 52, 1, 5, 32, 4, 1, EPI_GN_MISH, 8, 1, 0, 0, 0, 0) \
    X(52, 52, 1, 5, 32, 2, 2, EPI_GN_MISH, 8, 1, 0, 0, 0, 0) \
    X(52, 52, 1, 5, 32, 2, 2, EPI_GN_MISH, 8, 1, 0, 0, 0, 1) \
    X(52, 52, 1, 5, 32, 2, 2, EPI_GN_MISH, 8, 1, 0, 0, 0, 2) \
    X(26, 26, 1, 5, 32, 4, 1, EPI_GN_MISH, 16, 1, 0, 0, 0, 0) \
    X(26, 26, 1, 5, 32, 2, 2, EPI_GN_MISH, 16, 1, 0, 0, 0, 0) \
    X(26, 26, 1, 5, 32, 2, 2, EPI_GN_MISH, 16, 1, 0, 0, 0, 1) \
    X(26, 26, 1, 5, 32, 2, 2, EPI_GN_MISH, 16, 1, 0, 0, 0, 2) \
    X(26, 26, 1, 5, 32, 2, 2, EPI_GN_MISH, 16, 1, 0, 0, 0, 3) \
    X(13, 13, 1, 5, 32, 4, 1, EPI_GN_MISH, 32, 1, 0, 0, 0, 0) \
    X(13, 13, 1, 5, 32, 2, 2, EPI_GN_MISH, 32, 1, 0, 0, 0, 0) \
    X(13, 13, 1, 5, 32, 2, 2, EPI_GN_MISH, 32, 1, 0, 0, 0, 1) \
    X(13, 13, 1, 5, 32, 2, 2, EPI_GN_MISH, 32, 1, 0, 0, 0, 2) \
    X(13, 13, 1, 5, 32, 2, 2, EPI_GN_MISH, 32, 1, 0, 0, 0, 3) \
    X(13, 13, 1, 5, 32, 4, 2, EPI_GN_MISH, 32, 1, 0, 0, 0, 0) \
    X(26, 26, 1, 5, 32, 4, 2, EPI_GN_MISH, 16, 1, 0, 0, 0, 0) \
    X(52, 52, 1, 5, 32, 4, 2, EPI_GN_MISH, 8, 1, 0, 0, 0, 0) \
    X(13, 13, 1, 5, 32, 4, 1, EPI_GN_MISH, 16, 1, 0, 0, 0, 0) \
    X(13, 13, 1, 5, 32, 2, 2, EPI_GN_MISH, 16, 1, 0, 0, 0, 0) \
    X(13, 13, 1, 5, 32, 2, 2, EPI_GN_MISH, 16, 1, 0, 0, 0, 1) \
    X(13, 13, 1, 5, 32, 2, 2, EPI_GN_MISH, 16, 1, 0, 0, 0, 2) \
    X(13, 13, 1, 5, 32, 2, 2, EPI_GN_MISH, 16, 1, 0, 0, 0, 3) \
    X(26, 26, 1, 5, 32, 4, 1, EPI_GN_MISH, 8, 1, 0, 0, 0, 0) \
    X(26, 26, 1, 5, 32, 2, 2, EPI_GN_MISH, 8, 1, 0, 0, 0, 0) \
    X(26, 26, 1, 5, 32, 2, 2, EPI_GN_MISH, 8, 1, 0, 0, 0, 1) \
    X(26, 26, 1, 5, 32, 2, 2, EPI_GN_MISH, 8, 1, 0, 0, 0, 2) \
    X(26, 26, 1, 5, 32, 2, 2, EPI_GN_MISH, 8, 1, 0, 0, 0, 3) \
    X(52, 52, 1, 1, 32, 4, 1, EPI_BIAS, 8, 1, 0, 0, 0, 0) \
    X(52, 52, 1, 1, 32, 2, 2, EPI_BIAS, 8, 1, 0, 0, 0, 0) \
    X(52, 52, 1, 1, 32, 2, 2, EPI_BIAS, 8, 1, 0, 0, 0, 1) \
    X(52, 52, 1, 1, 32, 2, 2, EPI_BIAS, 8, 1, 0, 0, 0, 2) \
    X(26, 26, 1, 1, 32, 4, 1, EPI_BIAS, 16, 1, 0, 0, 0, 0) \
    X(26, 26, 1, 1, 32, 2, 2, EPI_BIAS, 16, 1, 0, 0, 0, 0) \
    X(26, 26, 1, 1, 32, 2, 2, EPI_BIAS, 16, 1, 0, 0, 0, 1) \
    X(26, 26, 1, 1, 32, 2, 2, EPI_BIAS, 16, 1, 0, 0, 0, 2) \
    X(26, 26, 1, 1, 32, 2, 2, EPI_BIAS, 16, 1, 0, 0, 0, 3) \
    X(13, 13, 1, 1, 32, 4, 1, EPI_BIAS, 32, 1, 0, 0, 0, 0) \
    X(13, 13, 1, 1, 32, 2, 2, EPI_BIAS, 32, 1, 0, 0, 0, 0) \
    X(13, 13, 1, 1, 32, 2, 2, EPI_BIAS, 32, 1, 0, 0, 0, 1) \
    X(13, 13, 1, 1, 32, 2, 2, EPI_BIAS, 32, 1, 0, 0, 0, 2) \
    X(13, 13, 1, 1, 32, 2, 2, EPI_BIAS, 32, 1, 0, 0, 0, 3) \
    X(13, 13, 1, 1, 32, 4, 1, EPI_BIAS, 16, 1, 0, 0, 0, 0) \
    X(13, 13, 1, 1, 32, 2, 2, EPI_BIAS, 16, 1, 0, 0, 0, 0) \
    X(13, 13, 1, 1, 32, 2, 2, EPI_BIAS, 16, 1, 0, 0, 0, 1) \
    X(13, 13, 1, 1, 32, 2, 2, EPI_BIAS, 16, 1, 0, 0, 0, 2) \
    X(13, 13, 1, 1, 32, 2, 2, EPI_BIAS, 16, 1, 0, 0, 0, 3) \
    X(26, 26, 1, 1, 32, 4, 1, EPI_BIAS, 8, 1, 0, 0, 0, 0) \
    X(26, 26, 1, 1, 32, 2, 2, EPI_BIAS, 8, 1, 0, 0, 0, 0) \
    X(26, 26, 1, 1, 32, 2, 2, EPI_BIAS, 8, 1, 0, 0, 0, 1) \
    X(26, 26, 1, 1, 32, 2, 2, EPI_BIAS, 8, 1, 0, 0, 0, 2) \
    X(26, 26, 1, 1, 32, 2, 2, EPI_BIAS, 8, 1, 0, 0, 0, 3) \
    X(52, 26, 2, 3, 32, 2, 2, EPI_BIAS, 8, 1, 0, 0, 0, 0) \
    X(52, 26, 2, 3, 32, 2, 2, EPI_BIAS, 8, 1, 0, 0, 0, 1) \
    X(52, 26, 2, 3, 32, 2, 2, EPI_BIAS, 8, 1, 0, 0, 0, 2) \
    X(26, 13, 2, 3, 32, 2, 2, EPI_BIAS, 16, 1, 0, 0, 0, 0) \
    X(26, 13, 2, 3, 32, 2, 2, EPI_BIAS, 16, 1, 0, 0, 0, 1) \
    X(26, 13, 2, 3, 32, 2, 2, EPI_BIAS, 16, 1, 0, 0, 0, 2) \
    X(26, 13, 2, 3, 32, 2, 2, EPI_BIAS, 16, 1, 0, 0, 0, 3) \
    X(13, 13, 1, 2, 32, 4, 1, EPI_BIAS, 16, 2, 0, 0, 0, 0) \
    X(13, 13, 1, 2, 32, 2, 2, EPI_BIAS, 16, 2, 0, 0, 0, 0) \
    X(13, 13, 1, 2, 32, 2, 2, EPI_BIAS, 16, 2, 0, 0, 0, 1) \
    X(13, 13, 1, 2, 32, 2, 2, EPI_BIAS, 16, 2, 0, 0, 0, 2) \
    X(13, 13, 1, 2, 32, 2, 2, EPI_BIAS, 16, 2, 0, 0, 0, 3) \
    X(26, 26, 1, 2, 32, 4, 1, EPI_BIAS, 8, 2, 0, 0, 0, 0) \
    X(26, 26, 1, 2, 32, 2, 2, EPI_BIAS, 8, 2, 0, 0, 0, 0) \
    X(26, 26, 1, 2, 32, 2, 2, EPI_BIAS, 8, 2, 0, 0, 0, 1) \
    X(26, 26, 1, 2, 32, 2, 2, EPI_BIAS, 8, 2, 0, 0, 0, 2) \
    X(26, 26, 1, 2, 32, 2, 2, EPI_BIAS, 8, 2, 0, 0, 0, 3) \
    X(52, 52, 1, 5, 32, 4, 1, EPI_GN_MISH, 8, 1, 1, 0, 1, 0) \
    X(52, 52, 1, 5, 64, 4, 1, EPI_GN_MISH, 8, 1, 0, 1, 1, 0) \
    X(52, 52, 1, 5, 64, 4, 1, EPI_GN_MISH, 8, 1, 0, 1, 0, 0) \
    X(26, 26, 1, 5, 64, 4, 1, EPI_GN_MISH, 16, 1, 0, 1, 1, 0) \
    X(13, 13, 1, 5, 64, 4, 1, EPI_GN_MISH, 32, 1, 0, 1, 1, 0) \
    X(13, 13, 1, 5, 64, 4, 1, EPI_GN_MISH, 16, 1, 0, 1, 1, 0) \
    X(26, 26, 1, 5, 64, 4, 1, EPI_GN_MISH, 8, 1, 0, 1, 1, 0) \
    X(26, 26, 1, 1, 64, 4, 1, EPI_BIAS, 16, 1, 0, 1, 1, 0) \
    X(13, 13, 1, 1, 64, 4, 1, EPI_BIAS, 32, 1, 0, 1, 1, 0) \
    X(13, 13, 1, 1, 64, 4, 1, EPI_BIAS, 16, 1, 0, 1, 1, 0) \
    X(26, 26, 1, 1, 64, 4, 1, EPI_BIAS, 8, 1, 0, 1, 1, 0) \
    X(52, 26, 2, 3, 32, 4, 1, EPI_BIAS, 8, 1, 0, 1, 1, 0) \
    X(26, 13, 2, 3, 32, 4, 1, EPI_BIAS, 16, 1, 0, 1, 1, 0) \
    X(13, 13, 1, 2, 64, 4, 1, EPI_BIAS, 16, 2, 0, 1, 1, 0) \
    X(26, 26, 1, 2, 64, 4, 1, EPI_BIAS, 8, 2, 0, 1, 1, 0)

static inline bool geom_is(const ConvGeom& g, int l_in, int lm, int stride, int ntaps, int kc, int nwn, int ks,
                           int epi, int gs, int ostr, int padc, int ain, int aout, int half) {
    return g.l_in == l_in && g.lm == lm && g.stride == stride && g.ntaps == ntaps && g.kc == kc &&
           g.nwn == nwn && g.ks == ks && g.epi == epi && g.gs == gs && g.ostr == ostr && g.padc == padc &&
           g.ain == ain && g.aout == aout && g.half == half;
}

bool conv_geom_supported(const ConvGeom& g) {
#define X(a, b, c, d, e, f, k, h, i, j, l, m, n, o) if (geom_is(g, a, b, c, d, e, f, k, h, i, j, l, m, n, o)) return true;
    CLD_CONV_INSTANCES(X)
#undef X
    return false;
}

hipError_t launch_conv(const ConvGeom& g, const ConvArgs& a, int b_pad, hipStream_t s) {
#define X(a_, b_, c_, d_, e_, f_, k_, h_, i_, j_, l_, m_, n_, o_) \
    if (geom_is(g, a_, b_, c_, d_, e_, f_, k_, h_, i_, j_, l_, m_, n_, o_)) return launch_inst<a_, b_, c_, d_, e_, f_, k_, h_, i_, j_, l_, m_, n_, o_>(a, b_pad, s);
    CLD_CONV_INSTANCES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace cld
