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
// whatever H is) x 64 output channels x all 16 xi: wave w holds the 32 accumulators (128 registers) of N-tile w, and the output
// transform runs on them in registers.  Per 16-channel chunk the 256 threads fetch the 4x4 patches straight from the NHWC tensor (one
// thread = one tile x two channels: 16 8-byte loads; the 4x overlap between neighbouring tiles is served by L1/L2), transform them
// (32 packed adds) and write V[xi][tile][16 channels] into one of two LDS images; the MFMA loop reads its A fragments from there with
// ds_read_b128 -- rows of 64 bytes, 16-byte slots permuted by (row / 4) ^ h(k group) so that the four 16-lane groups of a read touch
// every bank once (scripts/lds_conflicts.py wino) -- and its B fragments (U in MFMA fragment order, pack_conv_weights with xi as the
// "tap") from L2, four (xi, chunk) items ahead.  One barrier per chunk; two workgroups per CU, so one's patch loads, transform and
// epilogue run under the other's MFMAs.  Workgroup id -> (output-channel block = id % (C / 64), tile group): under round-robin XCD
// placement an XCD's L2 holds one channel block's U (<= 2 MB).
#include "cld_kernels.h"

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
    int voff[16];
    {
        const int T = tile0 + ts;
        const int a = T / G::TPA, rem = T % G::TPA, ty = rem / G::TH, tx = rem % G::TH;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int iy = 2 * ty - 1 + r, ix = 2 * tx - 1 + c;
                const bool in = T < ntiles && iy >= 0 && iy < HIN && ix >= 0 && ix < HIN;
                voff[4 * r + c] = in ? (((a * HIN + iy) * HIN + ix) * G::C + 2 * c2) * 4 : total_bytes;      // out of range reads 0
            }
    }
    v2f d[16];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsx, voff[i], c * (G::KC * 4), 0));
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

    v4f acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) { acc[xi][0] = v4f{0.f, 0.f, 0.f, 0.f}; acc[xi][1] = v4f{0.f, 0.f, 0.f, 0.f}; }
    v4f bq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bq[i] = wload(i);

    load_chunk(0);
    transform_store(0);
    __syncthreads();

    auto mfma_block = [&](const int buf, const int c) {
        const int bo = buf * (G::VBUF * 4);
        v4f af[2][2];
        af[0][0] = *reinterpret_cast<const v4f*>(ldsb + abase + bo);
        af[0][1] = *reinterpret_cast<const v4f*>(ldsb + abase + bo + 1024);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const int cur = xi & 1;
            const v4f bcur = bq[xi & 3];
            bq[xi & 3] = wload(c * 16 + xi + 4);
            // the next position's fragments are read behind this one's MFMAs: pinned, or the compiler sinks the reads to their use
            if (xi + 1 < 16) af[cur ^ 1][0] = *reinterpret_cast<const v4f*>(ldsb + abase + bo + (xi + 1) * 2048);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][0][e], bcur[e], acc[xi][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (xi + 1 < 16) af[cur ^ 1][1] = *reinterpret_cast<const v4f*>(ldsb + abase + bo + (xi + 1) * 2048 + 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][1][e], bcur[e], acc[xi][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

#pragma clang loop unroll(disable)
    for (int c = 0; c < G::NCH; c += 2) {
        load_chunk(c + 1);
        mfma_block(0, c);
        transform_store(1);
        __syncthreads();
        const bool more = c + 2 < G::NCH;
        if (more) load_chunk(c + 2);
        mfma_block(1, c + 1);
        if (more) transform_store(0);
        __syncthreads();
    }

    // ---- epilogue: Y = A^T M A in registers, folded BatchNorm, residual, ReLU; register r of a lane = tile 16 m + 4 kk + r, column n ----
    const int n = cb * 64 + 16 * wave + i16;
    const float sc = p.scale[n], sh = p.shift[n];
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
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int T = tile0 + 16 * m + 4 * kk + r;
            if (T >= ntiles) continue;
            const int a = T / G::TPA, rem = T % G::TPA, ty = rem / G::TH, tx = rem % G::TH;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int oy = 2 * ty + dy, ox = 2 * tx + dx;
                    if ((HIN & 1) && (oy >= HIN || ox >= HIN)) continue;
                    const size_t o = (((size_t)a * HIN + oy) * HIN + ox) * G::C + n;
                    float v = Y[dy][dx][r] * sc + sh;
                    if (p.res) v += p.res[o];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.y[o] = v;
                }
        }
    }
}

template <int HIN>
static hipError_t launch_wino_inst(const WinoArgs& a, hipStream_t s) {
    typedef WinoGeo<HIN> G;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv_kernel<HIN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)G::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
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
