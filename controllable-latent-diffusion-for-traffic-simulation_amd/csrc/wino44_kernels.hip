// wino44_kernels.hip -- the thirteen 3x3 stride-1 convolutions of the ContextEncoder's ResNet-18 (64 channels @ 56x56, 128 @ 28x28, 256 @ 14x14,
// 512 @ 7x7; reference: torchvision resnet18 built at src/tbsim/models/base_models.py:559-614, called from models/context_utils.py:40-61) by
// Winograd's minimal filtering F(4x4, 3x3).
//
// wino_kernels.hip runs these layers as F(2x2, 3x3): 16 multiplies per 2x2 outputs, 2.25x fewer than the direct form.  On 4x4 tiles the same
// construction needs 36 multiplies per 16 outputs: **4x fewer than the direct form, 1.78x fewer MFMAs than F(2x2, 3x3)** where the map is a
// whole number of tiles (56x56: 14 x 14 tiles, 28x28: 7 x 7) and at 7x7 (2 x 2 tiles over 8x8, which F(2x2) covers too), 1.36x at 14x14 (4 x 4 tiles
// over 16x16).  What decides whether that is usable in fp32 is the choice of points.  With
// {0, +-1, +-2, inf} (the textbook set) or {0, +-1, +-1/2, inf} the rounding error of a 64-channel layer is 4 - 9e-6 of max|y| against
// fp64; with **{0, 1, -1, 1/2, -2, inf}** -- reciprocal pairs of opposite sign -- it is 2 - 3e-6 (the direct form: 0.7 - 1.5e-6,
// F(2x2, 3x3): 0.4e-6; numpy model of the kernel's arithmetic, 64 and 128 channels), inside the bars the encoder is held to:
//     B^T = [ 1 -3/2  -2   3/2   1   0        G = [   1      0      0         A^T = [ 1  1  1   1    1   0
//             0  -1   1/2  5/2   1   0              1/3    1/3    1/3                 0  1 -1  1/2  -2   0
//             0   1  -5/2  1/2   1   0             -1/3    1/3   -1/3                 0  1  1  1/4   4   0
//             0  -2   -1    2    1   0            -16/15  -8/15  -4/15                0  1 -1  1/8  -8   1 ]
//             0  1/2  -1  -1/2   1   0              1/15  -2/15   4/15
//             0   1  -3/2  -2   3/2  1 ]              0      0      1 ]
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A   for the 6x6 input patch d (rows / columns 4 t - 1 .. 4 t + 4) of a 4x4 output tile;
// every entry of B^T and A^T is exact in fp32, U = G g G^T is formed in double at cld_finalize.
//
// Kernel (the structure of wino_kernels.hip): a workgroup owns 16 tiles (one M-tile; tiles are a flat list over the pass's agents) x 64
// output channels x all 36 xi: wave w holds the 36 accumulators (144 registers) of channels 16 w .. 16 w + 15; filters are the MFMA's A
// operand, tiles its B operand, so a lane ends up with four consecutive channels of the 16 outputs of ONE tile and the output transform
// runs in registers.  Per 16-channel chunk a thread fetches the 6x6 patch of one (tile, channel) straight from the NHWC tensor (36 4-byte
// loads, the 16 channels of a tile are one 64-byte segment), applies d B two rows at a time on register pairs and B^T (.) column by column,
// and writes V[xi][tile][16 channels] into one of two LDS images (36 KB each; 64-byte rows, 16-byte slots permuted as in wino_kernels.hip).
// U comes from L2 in MFMA fragment order (pack_conv_weights with xi as the "tap"), four (xi, chunk) items ahead.  One barrier per chunk, two
// workgroups per CU.  What bounds it (scripts/ubench/w44_unit.hip): not the MFMAs (44 % busy) but the ~400 instructions per
// thread and chunk of fetching and transforming a patch, which only the OTHER workgroup's MFMAs can cover -- a variant with eight waves per
// workgroup (the xi split between wave pairs, two M-tiles per wave, half the weight traffic) measured the same.
#include "cld_kernels.h"

#ifndef W44_EXP
#define W44_EXP 0      // experiment builds of scripts/ubench/w44_unit.hip only: 1 no patch loads in the loop, 2 no rows pass, 4 no columns pass / LDS stores (results are wrong)
#endif


// (wino_kernels.hip) a tile's result must not depend on its place in the workgroup: no implicit contraction
#pragma clang fp contract(off)

namespace cld {

typedef float v4f __attribute__((ext_vector_type(4)));

#ifdef W44_STAMPS
// diagnostic build (scripts/ubench/w44_unit.hip): cycle stamps of thread 0 of every workgroup
__device__ unsigned long long w44_stamps[8192 * 8];
#define W44STAMP(k)                                                                                \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (tid == 0 && blockIdx.x < 8192) {                                                       \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            w44_stamps[(size_t)blockIdx.x * 8 + (k)] = t_;                                         \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#else
#define W44STAMP(k) do {} while (0)
#endif

namespace {

template <int HIN>
struct W44Geo {
    static constexpr int C = HIN == 56 ? 64 : HIN == 28 ? 128 : HIN == 14 ? 256 : 512;
    static constexpr int TH = (HIN + 3) / 4, TPA = TH * TH;      // 4x4 tiles per row / per agent: 14 / 196, 7 / 49, 4 / 16, 2 / 4
    static constexpr bool EXACT = HIN % 4 == 0;                   // 14x14 and 7x7: the last tile row / column hangs over the edge (16x16, 8x8 covered)
    static constexpr int MT = 16, KC = 16;                        // tiles per workgroup: one M-tile
    static constexpr int NCH = C / KC, NCB = C / 64, NTN = C / 16;
    static constexpr int VBUF = 36 * MT * KC;                     // floats per V image
    static constexpr size_t LDS_BYTES = 2 * VBUF * sizeof(float);
    static_assert(HIN == 56 || HIN == 28 || HIN == 14 || HIN == 7, "resnet18 feature maps");
    static_assert(NCH % 2 == 0, "chunk pairs are unrolled");
};

__device__ __forceinline__ int hsw4(int k) { return ((k & 1) * 3) ^ (k >> 1); }      // wino_kernels.hip hsw

// t = B^T d (or a row of d B: B^T applied along the other index) for six values
__device__ __forceinline__ void bt6(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5, float (&t)[6]) {
    t[0] = fmaf(d3 - d1, 1.5f, fmaf(d2, -2.0f, d0 + d4));
    t[1] = fmaf(d3, 2.5f, fmaf(d2, 0.5f, d4 - d1));
    t[2] = fmaf(d2, -2.5f, fmaf(d3, 0.5f, d1 + d4));
    t[3] = fmaf(d3 - d1, 2.0f, d4 - d2);
    t[4] = fmaf(d1 - d3, 0.5f, d4 - d2);
    t[5] = fmaf(d4 - d2, 1.5f, fmaf(d3, -2.0f, d1 + d5));
}

typedef float v2f44 __attribute__((ext_vector_type(2)));
typedef unsigned int u4v44 __attribute__((ext_vector_type(4)));
// the same for two rows at once (v_pk_fma_f32 / v_pk_add_f32: two values per issue slot)
__device__ __forceinline__ void bt6p(const v2f44 d0, const v2f44 d1, const v2f44 d2, const v2f44 d3, const v2f44 d4, const v2f44 d5, v2f44 (&t)[6]) {
    auto f = [](const v2f44 a, const float sc, const v2f44 b) { return __builtin_elementwise_fma(a, v2f44{sc, sc}, b); };
    t[0] = f(d3 - d1, 1.5f, f(d2, -2.0f, d0 + d4));
    t[1] = f(d3, 2.5f, f(d2, 0.5f, d4 - d1));
    t[2] = f(d2, -2.5f, f(d3, 0.5f, d1 + d4));
    t[3] = f(d3 - d1, 2.0f, d4 - d2);
    t[4] = f(d1 - d3, 0.5f, d4 - d2);
    t[5] = f(d4 - d2, 1.5f, f(d3, -2.0f, d1 + d5));
}

__device__ __forceinline__ v4f fma4s(const v4f a, const float s, const v4f b) { return __builtin_elementwise_fma(a, v4f{s, s, s, s}, b); }      // a s + b

}  // namespace

template <int HIN>
__global__ __launch_bounds__(256, 2) void wino44_conv_kernel(const WinoArgs p) {
    typedef W44Geo<HIN> G;
    extern __shared__ __attribute__((aligned(16))) float ldsw[];
    const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup id -> (tile group, channel block).  Ids go round the 8 XCDs: id % 8 is the XCD, and an XCD walks a CONTIGUOUS eighth of the
    // tile list (with both channel blocks of a group back to back), so that the two patch rows a tile shares with the tile above it and the
    // channel blocks' common patches are found in that XCD's L2
    const int ntiles = p.B * G::TPA, ngroups = (ntiles + G::MT - 1) / G::MT, gpx = (ngroups + 7) / 8;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int grp = xcd * gpx + idx / G::NCB, cb = idx % G::NCB;
    if (grp >= ngroups || idx / G::NCB >= gpx) return;             // (the grid is 8 x gpx x NCB)
    const int tile0 = grp * G::MT;
    W44STAMP(0);

    // ---- staging role: tile ts of the workgroup, channel c1 of the chunk ----
    const int ts = tid >> 4, c1 = tid & 15;
    const int total_bytes = p.B * HIN * HIN * G::C * 4;             // <= 256 agents per pass: < 2^31
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, total_bytes, 0x00020000);
    // Byte offsets (unsigned 32-bit) are taken from patch element (1, 1) -- the tile's first output position, always inside the image, so the
    // vector offset of every load is a valid non-negative offset by itself (the range check does not see the scalar offset).  Rows and
    // columns 1 .. 4 of a patch are always inside: their distances ride in the scalar offset (rows) and the instruction's immediate (columns);
    // row 0 / 5 and column 0 / 5 can be the zero padding and keep an offset of their own, 2^30 when they are (wino_kernels.hip: past the
    // tensor's <= 206 MB, the range check returns 0); so does every element of a tile past the end of the list
    constexpr unsigned kOut = 1u << 30;
    constexpr int ROWB = HIN * G::C * 4, COLB = G::C * 4;           // bytes between patch rows / columns
    // (14x14 / 7x7: the last tile row / column hangs over the edge, so rows / columns 3 and 4 can be padding too and get offset registers as
    //  well; 1 and 2 are inside for every tile of every map)
    unsigned vbase, vrow0, vrow3, vrow4, vrow5, coff0, coff3, coff4, coff5;
    {
        const int T = tile0 + ts;
        const int a = T / G::TPA, rem = T % G::TPA, ty = rem / G::TH, tx = rem % G::TH;
        vbase = T < ntiles ? (unsigned)((((a * HIN + 4 * ty) * HIN + 4 * tx) * G::C + c1) * 4) : kOut;
#ifdef W44_EXP_HOT
        vbase = (unsigned)(((4 * HIN + 4) * G::C + c1) * 4);      // experiment: every tile reads the same patch (always in L1 / L2)
#endif
        auto rowoff = [&](const int r) { const int iy = 4 * ty - 1 + r; return (iy >= 0 && iy < HIN) ? vbase + (unsigned)((r - 1) * ROWB) : kOut; };
        auto coloff = [&](const int c) { const int ix = 4 * tx - 1 + c; return (ix >= 0 && ix < HIN) ? (unsigned)((c - 1) * COLB) : kOut; };
        vrow0 = rowoff(0); vrow3 = rowoff(3); vrow4 = rowoff(4); vrow5 = rowoff(5);
        coff0 = coloff(0); coff3 = coloff(3); coff4 = coloff(4); coff5 = coloff(5);
    }
    v2f44 dp[3][6];                                                  // the patch in row pairs -- dp[q][c] = (d[2 q][c], d[2 q + 1][c]) --, then d B in place
    auto patch_load = [&](const int r, const int c, const int chunk) {
        const bool rreg = r == 0 || r == 5 || (!G::EXACT && r >= 3), creg = c == 0 || c == 5 || (!G::EXACT && c >= 3);
        const unsigned vr = r == 0 ? vrow0 : r == 5 ? vrow5 : (rreg && r == 3) ? vrow3 : (rreg && r == 4) ? vrow4 : vbase;
        const unsigned co = c == 0 ? coff0 : c == 5 ? coff5 : c == 3 ? coff3 : coff4;
        const unsigned vo = creg ? vr + co : vr + (unsigned)((c - 1) * COLB);
        const int so = chunk * (G::KC * 4) + (rreg ? 0 : (r - 1) * ROWB);
        dp[r >> 1][c][r & 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, (int)vo, so, 0));
    };
    const int wofs = ts * 16 + ((((ts >> 2) & 3) ^ hsw4(c1 >> 2)) << 2) + (c1 & 3);
    auto rows_pass = [&](const int q) {                              // rows 2 q, 2 q + 1 of d B, on register pairs
        v2f44 t[6];
        bt6p(dp[q][0], dp[q][1], dp[q][2], dp[q][3], dp[q][4], dp[q][5], t);
#pragma unroll
        for (int j = 0; j < 6; ++j) dp[q][j] = t[j];
    };
    auto cols_pass = [&](const int j, const int buf) {               // column j of B^T (d B), stored: xi = 6 i + j
        float t[6];
        bt6(dp[0][j][0], dp[0][j][1], dp[1][j][0], dp[1][j][1], dp[2][j][0], dp[2][j][1], t);
        float* vb = ldsw + buf * G::VBUF + wofs;
#pragma unroll
        for (int i = 0; i < 6; ++i) vb[(6 * i + j) * (G::MT * 16)] = t[i];
    };

    // ---- MFMA role: lane (i16, kk) of wave w: tile i16, channels 4 kk .. 4 kk + 3 of the chunk, N-tile w ----
    const char* ldsb = reinterpret_cast<const char*>(ldsw);
    const int abase = (i16 * 16 + ((((i16 >> 2) & 3) ^ hsw4(kk)) << 2)) * 4;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ufrag), 0, G::NCH * 36 * G::NTN * 1024, 0x00020000);
    const int wvoff = lane * 16;
    const int wsoff = (cb * 4 + wave) * 1024;
    auto wload = [&](const int item) {          // item = chunk * 36 + xi; past the end: out of range, reads 0, never used
        return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff, item * (G::NTN * 1024) + wsoff, 0));
    };

    v4f acc[36];
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) acc[xi] = v4f{0.f, 0.f, 0.f, 0.f};
    v4f bq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = wload(i);

#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) patch_load(r, c, 0);
#pragma unroll
    for (int q = 0; q < 3; ++q) rows_pass(q);
#pragma unroll
    for (int j = 0; j < 6; ++j) cols_pass(j, 0);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) patch_load(r, c, 1);
    __syncthreads();
    W44STAMP(1);

    // one chunk: 36 xi x 4 MFMAs.  V fragments run two positions ahead of their MFMAs (a rolling window of three), U fragments four.  The patch of
    // the NEXT chunk is in registers when a block starts: its rows pass (two rows at a time) runs at xi = 10, 12, 14, its columns pass (with the
    // stores into the other image) at xi = 16 .. 21, and the patch of the chunk after that is requested in ONE go at xi = 35, behind the block's last
    // weight-fragment request.  Loads return in order: a weight fragment requested behind patch loads is not usable before they have landed.  With
    // the patch requests spread over the first half of a block (two per position) every position there waited for a trip to HBM that had started
    // four positions earlier -- the block ran at the pace of the memory system (a chunk pair with staging 35k cycles, without 17k); requested in
    // one go at the end, the fragments of the next block's first four positions are already on their way and only what is left of ONE trip after a
    // barrier and four positions is exposed.
    auto mfma_block = [&](const int buf, const int c, const bool stage1, const bool stage2) {
        const int bo = buf * (G::VBUF * 4);
        auto frag = [&](const int xi) { return *reinterpret_cast<const v4f*>(ldsb + abase + bo + xi * 1024); };
        v4f ar[3];
        ar[0] = frag(0);
        ar[1] = frag(1);
#pragma unroll
        for (int xi = 0; xi < 36; ++xi) {
            const v4f bcur = bq[xi & 3];
            bq[xi & 3] = wload(c * 36 + xi + 4);
            if (xi + 2 < 36) ar[(xi + 2) % 3] = frag(xi + 2);
            if (!(W44_EXP & 2) && stage1 && xi >= 10 && xi < 16 && (xi & 1) == 0) rows_pass((xi - 10) >> 1);
            if (!(W44_EXP & 4) && stage1 && xi >= 16 && xi < 22) cols_pass(xi - 16, buf ^ 1);
            if (!(W44_EXP & 1) && stage2 && xi == 35) {      // behind the block's last weight fragment request
#pragma unroll
                for (int i = 0; i < 36; ++i) patch_load(i / 6, i % 6, c + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[e], ar[xi % 3][e], acc[xi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

#pragma clang loop unroll(disable)
    for (int c = 0; c < G::NCH; c += 2) {
        const bool more = c + 2 < G::NCH;
        mfma_block(0, c, true, more);
        __syncthreads();
        mfma_block(1, c + 1, more, more);
        __syncthreads();
        if (c == 0) W44STAMP(2);
    }
    W44STAMP(3);

    // ---- epilogue: Y = A^T M A in registers (first along j -- M A, row by row of M --, then along i), folded BatchNorm, residual, ReLU;
    //      lane = tile i16, channels n4 .. n4 + 3: sixteen 16-byte stores ----
    const int n4 = cb * 64 + 16 * wave + 4 * kk;
    v4f r[6][4];                                                     // M A: [row of M][output column]
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const v4f m0 = acc[6 * i], m1 = acc[6 * i + 1], m2 = acc[6 * i + 2], m3 = acc[6 * i + 3], m4 = acc[6 * i + 4], m5 = acc[6 * i + 5];
        const v4f p12 = m1 + m2, q12 = m1 - m2;
        r[i][0] = (m0 + p12) + (m3 + m4);
        r[i][1] = fma4s(m4, -2.0f, fma4s(m3, 0.5f, q12));
        r[i][2] = fma4s(m4, 4.0f, fma4s(m3, 0.25f, p12));
        r[i][3] = fma4s(m4, -8.0f, fma4s(m3, 0.125f, q12)) + m5;
    }
    W44STAMP(4);
    W44STAMP(5);
    const int T = tile0 + i16;
    if (T >= ntiles) return;
    const int a = T / G::TPA, rem = T % G::TPA, ty = rem / G::TH, tx = rem % G::TH;
    const bool has_res = p.res != nullptr;
    const int o0 = (((a * HIN + 4 * ty) * HIN + 4 * tx) * G::C + n4) * 4;      // byte offset of the tile's first output (the tensors are < 2^31 bytes)
    const int nrow = HIN - 4 * ty, ncol = HIN - 4 * tx;                       // outputs of the tile inside the map (>= 4 except on the ragged maps' last row / column)
    auto ooff = [&](const int oy, const int ox) {                             // an output past the edge: an offset past the tensor (the load returns 0, the store is dropped)
        const int off = o0 + oy * ROWB + ox * COLB;
        return (G::EXACT || (oy < nrow && ox < ncol)) ? off : (int)kOut;
    };
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_res ? p.res : p.y), 0, total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, total_bytes, 0x00020000);
    // the 16 residual values are requested in one go, behind M A (the accumulators are dead): one offset register, output rows in the scalar
    // offset, columns in the immediate.  (Requested per output row in front of their use they cost four trips to memory in a row.)
    v4f rv[16];
    if (has_res) {
#pragma unroll
        for (int k = 0; k < 16; ++k) rv[k] = G::EXACT ? __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsr, o0 + (k & 3) * COLB, (k >> 2) * ROWB, 0))
                               : __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsr, ooff(k >> 2, k & 3), 0, 0));
    }
    const v4f sc = *reinterpret_cast<const v4f*>(p.scale + n4), sh = *reinterpret_cast<const v4f*>(p.shift + n4);
#pragma unroll
    for (int ox = 0; ox < 4; ++ox) {
        const v4f p12 = r[1][ox] + r[2][ox], q12 = r[1][ox] - r[2][ox];
        v4f Y[4];
        Y[0] = (r[0][ox] + p12) + (r[3][ox] + r[4][ox]);
        Y[1] = fma4s(r[4][ox], -2.0f, fma4s(r[3][ox], 0.5f, q12));
        Y[2] = fma4s(r[4][ox], 4.0f, fma4s(r[3][ox], 0.25f, p12));
        Y[3] = fma4s(r[4][ox], -8.0f, fma4s(r[3][ox], 0.125f, q12)) + r[5][ox];
#pragma unroll
        for (int oy = 0; oy < 4; ++oy) {
            v4f v = __builtin_elementwise_fma(Y[oy], sc, sh);
            if (has_res) v += rv[4 * oy + ox];
            if (p.relu) v = v4f{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            // (16-byte stores keep their distances in the VECTOR offset: wino1d_edge.hip)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v44, v), rsy, ooff(oy, ox), 0, 0);
        }
    }
    W44STAMP(6);
}

template <int HIN>
static hipError_t launch_wino44_inst(const WinoArgs& a, hipStream_t s) {
    typedef W44Geo<HIN> G;
    static unsigned long long attr_done = 0;      // one bit per device
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(wino44_conv_kernel<HIN>), (int)G::LDS_BYTES, &attr_done); e != hipSuccess) return e;
    const int groups = (a.B * G::TPA + G::MT - 1) / G::MT, gpx = (groups + 7) / 8;
    hipLaunchKernelGGL(wino44_conv_kernel<HIN>, dim3(8 * gpx * G::NCB), dim3(256), G::LDS_BYTES, s, a);
    return hipGetLastError();
}

bool wino44_supported(int hin, int channels) { return (hin == 56 && channels == 64) || (hin == 28 && channels == 128) || (hin == 14 && channels == 256) || (hin == 7 && channels == 512); }

// a.ufrag: U = G g G^T of F(4x4, 3x3), 36 planes per 16-channel chunk (cld_api.hip, Conv2dLayer::ufrag44)
hipError_t launch_wino44_conv(int hin, int channels, const WinoArgs& a, hipStream_t s) {
    if (a.B < 1 || a.B > 256) return hipErrorInvalidValue;          // byte offsets are 32-bit: one pass of the encoder at a time
    if (hin == 56 && channels == 64) return launch_wino44_inst<56>(a, s);
    if (hin == 28 && channels == 128) return launch_wino44_inst<28>(a, s);
    if (hin == 14 && channels == 256) return launch_wino44_inst<14>(a, s);
    if (hin == 7 && channels == 512) return launch_wino44_inst<7>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace cld
