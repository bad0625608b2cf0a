// wino_kernels.hip -- the 3x3 stride-1 convolutions of the ContextEncoder's ResNet-18 (reference: torchvision resnet18 built at
// src/tbsim/models/base_models.py:559-614, called from models/context_utils.py:40-61) by Winograd's minimal filtering F(2x2, 3x3).
//
// 13 of the 19 convolutions behind the stem are 3x3 / stride 1 / pad 1 with C_in = C_out (64 @ 56x56, 128 @ 28x28, 256 @ 14x14,
// 512 @ 7x7); as implicit GEMMs (context_kernels.hip conv2d_kernel) they keep the fp32 MFMA pipe busy 85-98 % of the time -- what is
// left to take out is the arithmetic itself.  For a 2x2 output tile with its 4x4 input patch d and the 3x3 filter g
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A ,    B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],
//                                                 A^T = [1 1 1 0; 0 1 -1 -1]
// the sum over input channels moves inside the element-wise product: 16 independent GEMMs  M_xi[tile][k] = sum_c V_xi[tile][c] U_xi[c][k]
// (xi = position in the 4x4 transform domain), i.e. 16 multiplies per 4 outputs instead of 36 -- 2.25x fewer MFMAs (1.72x at 7x7, whose
// 4x4 tiles cover 8x8).  B^T and A^T hold 0 / +-1 only, U = G g G^T is formed in double at cld_finalize: the fp32 rounding error of the
// result equals the direct form's (3e-7 of max|y| on unit-variance inputs, both measured against fp64).
//
// Kernel: a workgroup owns 32 tiles (two 16-row M-tiles; tiles are a flat list over the pass's agents, so every launch is full
// whatever H is) x 64 output channels x all 16 xi: wave w holds the 32 accumulators (128 registers) of channels 16 w .. 16 w + 15, and the
// output transform runs on them in registers.  Per 16-channel chunk the 256 threads fetch the 4x4 patches straight from the NHWC tensor (one
// thread = one tile x two channels: 16 8-byte loads; the 4x overlap between neighbouring tiles is served by L1/L2), transform them
// (32 packed adds) and write V[xi][tile][16 channels] into one of two LDS images; the MFMA loop reads its A fragments from there with
// ds_read_b128 -- rows of 64 bytes, 16-byte slots permuted by (row / 4) ^ h(k group) so that the four 16-lane groups of a read touch
// every bank once (scripts/lds_conflicts.py wino) -- and its B fragments (U in MFMA fragment order, pack_conv_weights with xi as the
// "tap") from L2, four (xi, chunk) items ahead.  One barrier per chunk; two workgroups per CU, so one's patch loads, transform and
// epilogue run under the other's MFMAs.  Workgroup id -> (output-channel block = id % (C / 64), tile group): under round-robin XCD
// placement an XCD's L2 holds one channel block's U (<= 2 MB).
#include "cld_kernels.h"

#ifndef WINO_RING
#define WINO_RING 4
#endif
#ifndef WINO_IL
#define WINO_IL 1
#endif
#ifndef WINO_PIPE
#define WINO_PIPE 1
#endif
#ifndef WINO_SKIP
#define WINO_SKIP 0
#endif
#ifndef WINO_STAGGER
#define WINO_STAGGER 0
#endif
#ifndef WINO_EXP
#define WINO_EXP 0
#endif

// no implicit multiply-add contraction: the epilogue is unrolled over the two M-tiles, and a tile's result must not depend on which copy
// computed it (wino1d_kernels.hip)
#pragma clang fp contract(off)

namespace cld {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

namespace {

template <int HIN>
struct WinoGeo {
    static constexpr int C = HIN == 56 ? 64 : HIN == 28 ? 128 : HIN == 14 ? 256 : 512;
    static constexpr int TH = (HIN + 1) / 2, TPA = TH * TH;       // 2x2 tiles per row / per agent
    static constexpr int MT = 32, NMT = 2, KC = 16;
    static constexpr int NCH = C / KC, NCB = C / 64, NTN = C / 16;
    static constexpr int VBUF = 16 * MT * KC;                      // floats per V image
    static constexpr size_t LDS_BYTES = 2 * VBUF * sizeof(float);
    static_assert(HIN == 56 || HIN == 28 || HIN == 14 || HIN == 7, "resnet18 feature maps");
    static_assert(NCH % 2 == 0, "chunk pairs are unrolled");
};

// slot permutation of the 16-byte channel quads inside a 64-byte V row (see the header): {0, 3, 1, 2}
__device__ __forceinline__ int hsw(int k) { return ((k & 1) * 3) ^ (k >> 1); }

}  // namespace

template <int HIN>
__global__ __launch_bounds__(256, 2) void wino_conv_kernel(const WinoArgs p) {
    typedef WinoGeo<HIN> G;
    extern __shared__ __attribute__((aligned(16))) float ldsw[];
    const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, kk = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = blockIdx.x % G::NCB, tile0 = (blockIdx.x / G::NCB) * G::MT;
    const int ntiles = p.B * G::TPA;

    // ---- staging role: tile ts of the workgroup, channels 2 c2, 2 c2 + 1 of the chunk ----
    const int ts = tid >> 3, c2 = tid & 7;
    const int total_bytes = p.B * HIN * HIN * G::C * 4;             // <= 256 agents per pass: < 2^31
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, total_bytes, 0x00020000);
    // byte offset of patch element (r, c) = vbase + roff[r] + coff[c] (unsigned 32-bit); a row / column outside the image (the zero
    // padding) or a tile past the end of the list contributes 2^30 instead: one, two or three of them put the sum past the tensor's
    // <= 206 MB whatever the (possibly negative, row / column -1) base is, and the buffer's range check returns 0
    constexpr unsigned kOut = 1u << 30;
    unsigned vbase, roff[4], coff[4];
    {
        const int T = tile0 + ts;
        const int a = T / G::TPA, rem = T % G::TPA, ty = rem / G::TH, tx = rem % G::TH;
        vbase = T < ntiles ? (unsigned)((((a * HIN + 2 * ty - 1) * HIN + 2 * tx - 1) * G::C + 2 * c2) * 4) : kOut;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = 2 * ty - 1 + r, ix = 2 * tx - 1 + r;
            roff[r] = (iy >= 0 && iy < HIN) ? (unsigned)(r * HIN * G::C * 4) : kOut;
            coff[r] = (ix >= 0 && ix < HIN) ? (unsigned)(r * G::C * 4) : kOut;
        }
    }
    v2f d[16];
    auto patch_load = [&](const int i, const int c) {
        d[i] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsx, (int)(vbase + roff[i >> 2] + coff[i & 3]), c * (G::KC * 4), 0));
    };
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < 16; ++i) patch_load(i, c);
    };
    const int wofs = ts * 16 + ((((ts >> 2) & 3) ^ hsw(c2 >> 1)) << 2) + (c2 & 1) * 2;
    auto transform_store = [&](int buf) {
        float* vb = ldsw + buf * G::VBUF + wofs;
        v2f t[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t[0][c] = d[c] - d[8 + c];
            t[1][c] = d[4 + c] + d[8 + c];
            t[2][c] = d[8 + c] - d[4 + c];
            t[3][c] = d[4 + c] - d[12 + c];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<v2f*>(vb + (4 * i + 0) * (G::MT * 16)) = t[i][0] - t[i][2];
            *reinterpret_cast<v2f*>(vb + (4 * i + 1) * (G::MT * 16)) = t[i][1] + t[i][2];
            *reinterpret_cast<v2f*>(vb + (4 * i + 2) * (G::MT * 16)) = t[i][2] - t[i][1];
            *reinterpret_cast<v2f*>(vb + (4 * i + 3) * (G::MT * 16)) = t[i][1] - t[i][3];
        }
    };

    // the same transform in eight pieces (the four rows of d B, then the four rows of B^T (d B) with their stores), spread over the second half
    // of the MFMA block that precedes the image's use: the VALU / LDS work issues in the MFMAs' shadow instead of between two blocks
    v2f tt[4][4];
    auto transform_piece = [&](const int k, const int buf) {
        if (k < 4) {                      // row k of d B (needs patch row k only: the loads of rows k + 1 .. are still in flight)
            const int r = k;
            tt[r][0] = d[4 * r] - d[4 * r + 2];
            tt[r][1] = d[4 * r + 1] + d[4 * r + 2];
            tt[r][2] = d[4 * r + 2] - d[4 * r + 1];
            tt[r][3] = d[4 * r + 1] - d[4 * r + 3];
        } else {                          // row i of B^T (d B), stored
            const int i = k - 4;
            float* vb = ldsw + buf * G::VBUF + wofs;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v2f v = i == 0 ? tt[0][j] - tt[2][j] : i == 1 ? tt[1][j] + tt[2][j] : i == 2 ? tt[2][j] - tt[1][j] : tt[1][j] - tt[3][j];
                *reinterpret_cast<v2f*>(vb + (4 * i + j) * (G::MT * 16)) = v;
            }
        }
    };

    // ---- MFMA role: lane (i16, kk) of wave w: rows 16 m + i16 of the tile list, channels 4 kk .. 4 kk + 3 of the chunk, N-tile w ----
    const char* ldsb = reinterpret_cast<const char*>(ldsw);
    const int abase = (i16 * 16 + ((((i16 >> 2) & 3) ^ hsw(kk)) << 2)) * 4;
    const int nitems = G::NCH * 16;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ufrag), 0, nitems * G::NTN * 1024, 0x00020000);
    const int wvoff = lane * 16;
    const int wsoff = (cb * 4 + wave) * 1024;
    auto wload = [&](int item) {          // item = chunk * 16 + xi; past the end: out of range, reads 0, never used
        return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff, item * (G::NTN * 1024) + wsoff, 0));
    };

#if WINO_STAGGER
    // experiment: the two workgroups of a CU start half a block apart (wave slot parity from HW_ID[3:0])
    if (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1) __builtin_amdgcn_s_sleep(WINO_STAGGER);
#endif
    v4f acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) { acc[xi][0] = v4f{0.f, 0.f, 0.f, 0.f}; acc[xi][1] = v4f{0.f, 0.f, 0.f, 0.f}; }
    constexpr int WD = WINO_RING;      // B fragments in flight (items = (chunk, xi) pairs)
    v4f bq[WD];
#pragma unroll
    for (int i = 0; i < WD; ++i) bq[i] = wload(i);

    load_chunk(0);
    transform_store(0);
    __syncthreads();

    auto mfma_block = [&](const int buf, const int c, const bool stage) {
        const int bo = buf * (G::VBUF * 4);
#if WINO_IL == 4
        // two positions at a time: four independent accumulators in turn (an MFMA never waits for the one before it)
        v4f af[2][2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m) af[0][j][m] = *reinterpret_cast<const v4f*>(ldsb + abase + bo + j * 2048 + m * 1024);
#pragma unroll
        for (int xp = 0; xp < 8; ++xp) {
            const int cur = xp & 1, x0 = 2 * xp, x1 = 2 * xp + 1;
            const v4f b0 = bq[x0 % WD], b1 = bq[x1 % WD];
            bq[x0 % WD] = wload(c * 16 + x0 + WD);
            bq[x1 % WD] = wload(c * 16 + x1 + WD);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (xp + 1 < 8) af[cur ^ 1][e >> 1][e & 1] = *reinterpret_cast<const v4f*>(ldsb + abase + bo + (2 * xp + 2 + (e >> 1)) * 2048 + (e & 1) * 1024);
                __builtin_amdgcn_sched_barrier(0);
                acc[x0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[e], af[cur][0][0][e], acc[x0][0], 0, 0, 0);
                acc[x0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[e], af[cur][0][1][e], acc[x0][1], 0, 0, 0);
                acc[x1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[e], af[cur][1][0][e], acc[x1][0], 0, 0, 0);
                acc[x1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[e], af[cur][1][1][e], acc[x1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
        v4f af[2][2];
        af[0][0] = *reinterpret_cast<const v4f*>(ldsb + abase + bo);
        af[0][1] = *reinterpret_cast<const v4f*>(ldsb + abase + bo + 1024);
        if (WINO_SKIP & 2) { af[1][0] = af[0][1]; af[1][1] = af[0][0]; }
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const int cur = xi & 1;
            const v4f bcur = bq[xi % WD];
            if (!(WINO_SKIP & 1)) bq[xi % WD] = wload(c * 16 + xi + WD);
            if (WINO_PIPE && stage && xi < 8 && WINO_EXP != 1 && WINO_EXP != 3) { patch_load(2 * xi, c + 1); patch_load(2 * xi + 1, c + 1); }
            // the next position's fragments are read behind this one's MFMAs: pinned, or the compiler sinks the reads to their use
            if (xi + 1 < 16 && !(WINO_SKIP & 2)) af[cur ^ 1][0] = *reinterpret_cast<const v4f*>(ldsb + abase + bo + (xi + 1) * 2048);
            __builtin_amdgcn_sched_barrier(0);
#if WINO_IL == 2
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[e], af[cur][0][e], acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[e], af[cur][1][e], acc[xi][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (xi + 1 < 16) af[cur ^ 1][1] = *reinterpret_cast<const v4f*>(ldsb + abase + bo + (xi + 1) * 2048 + 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 2; e < 4; ++e) {
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[e], af[cur][0][e], acc[xi][0], 0, 0, 0);
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[e], af[cur][1][e], acc[xi][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[e], af[cur][0][e], acc[xi][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (xi + 1 < 16 && !(WINO_SKIP & 2)) af[cur ^ 1][1] = *reinterpret_cast<const v4f*>(ldsb + abase + bo + (xi + 1) * 2048 + 1024);
            if (WINO_PIPE && stage && xi >= 8) transform_piece(xi - 8, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[e], af[cur][1][e], acc[xi][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
#endif
    };

#pragma clang loop unroll(disable)
    for (int c = 0; c < G::NCH; c += 2) {
        if (!WINO_PIPE && WINO_EXP != 1 && WINO_EXP != 3) load_chunk(c + 1);
        if (WINO_EXP != 2) mfma_block(0, c, true);
        if ((!WINO_PIPE || WINO_EXP == 2) && WINO_EXP != 3) transform_store(1);
        if (WINO_EXP != 3) __syncthreads();
        const bool more = c + 2 < G::NCH;
        if (!WINO_PIPE && more && WINO_EXP != 1 && WINO_EXP != 3) load_chunk(c + 2);
        if (WINO_EXP != 2) mfma_block(1, c + 1, more);
        if (more && (!WINO_PIPE || WINO_EXP == 2) && WINO_EXP != 3) transform_store(0);
        if (WINO_EXP != 3) __syncthreads();
    }

    // ---- epilogue: Y = A^T M A in registers, folded BatchNorm, residual, ReLU.  The filters are the MFMA's A operand and the tiles its
    //      B operand (M^T = U^T V^T), so a lane holds FOUR CONSECUTIVE CHANNELS 16 w + 4 kk .. + 3 of ONE tile 16 m + i16: every global
    //      access of the epilogue is 16 bytes per lane ----
    const int n4 = cb * 64 + 16 * wave + 4 * kk;
    const v4f sc = *reinterpret_cast<const v4f*>(p.scale + n4), sh = *reinterpret_cast<const v4f*>(p.shift + n4);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        v4f s0[4], s1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s0[i] = acc[4 * i][m] + acc[4 * i + 1][m] + acc[4 * i + 2][m];
            s1[i] = acc[4 * i + 1][m] - acc[4 * i + 2][m] - acc[4 * i + 3][m];
        }
        v4f Y[2][2];
        Y[0][0] = s0[0] + s0[1] + s0[2];
        Y[1][0] = s0[1] - s0[2] - s0[3];
        Y[0][1] = s1[0] + s1[1] + s1[2];
        Y[1][1] = s1[1] - s1[2] - s1[3];
        const int T = tile0 + 16 * m + i16;
        if (T >= ntiles) continue;
        const int a = T / G::TPA, rem = T % G::TPA, ty = rem / G::TH, tx = rem % G::TH;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int oy = 2 * ty + dy, ox = 2 * tx + dx;
                if ((HIN & 1) && (oy >= HIN || ox >= HIN)) continue;
                const size_t o = (((size_t)a * HIN + oy) * HIN + ox) * G::C + n4;
                v4f v = __builtin_elementwise_fma(Y[dy][dx], sc, sh);
                if (p.res) v += *reinterpret_cast<const v4f*>(p.res + o);
                if (p.relu) v = v4f{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                *reinterpret_cast<v4f*>(p.y + o) = v;
            }
    }
}

template <int HIN>
static hipError_t launch_wino_inst(const WinoArgs& a, hipStream_t s) {
    typedef WinoGeo<HIN> G;
    static unsigned long long attr_done = 0;      // one bit per device (a second handle on another device sets it there too)
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(wino_conv_kernel<HIN>), (int)G::LDS_BYTES, &attr_done); e != hipSuccess) return e;
    const int groups = (a.B * G::TPA + G::MT - 1) / G::MT;
    hipLaunchKernelGGL(wino_conv_kernel<HIN>, dim3(groups * G::NCB), dim3(256), G::LDS_BYTES, s, a);
    return hipGetLastError();
}

hipError_t launch_wino_conv(int hin, int channels, const WinoArgs& a, hipStream_t s) {
    if (a.B < 1 || a.B > 256) return hipErrorInvalidValue;          // byte offsets are 32-bit: one pass of the encoder at a time
    if (hin == 56 && channels == 64) return launch_wino_inst<56>(a, s);
    if (hin == 28 && channels == 128) return launch_wino_inst<28>(a, s);
    if (hin == 14 && channels == 256) return launch_wino_inst<14>(a, s);
    if (hin == 7 && channels == 512) return launch_wino_inst<7>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace cld
