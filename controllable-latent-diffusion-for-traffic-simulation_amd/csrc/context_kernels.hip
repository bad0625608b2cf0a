// context_kernels.hip -- ContextEncoder (reference models/context_utils.py:8-61), the producer of cond_feat:
//   map branch   : torchvision resnet18 with a 34-channel 7x7/2 stem and a 512 -> 256 fc
//                  (src/tbsim/models/base_models.py:559-614, via MapEncoder src/tbsim/models/diffuser_helpers.py:297-348)
//   state branch : MLP 4 -> 64 -> 64 -> 64 with LayerNorm + ReLU (base_models.py:21-96)
//   combine      : MLP 320 -> 320 -> 320 -> 256 -> 256 -> 256 with LayerNorm + ReLU
//
// Kernels (all exact fp32 on v_mfma_f32_16x16x4_f32):
//   stem_conv_kernel   7x7/2 conv read STRAIGHT from the NCHW raster the reference hands over (6.8 MB per agent, the
//                      one HBM-relevant read of the path): a workgroup owns 2 output rows x 112 columns x 64 channels,
//                      stages a 9 x 224 strip of one input plane at a time in LDS (coalesced 896-byte rows) and walks
//                      the 49 taps of that plane as 13 MFMA k-steps; BatchNorm (eval) + ReLU fused; NHWC out.
//   maxpool_kernel     3x3/2 on NHWC.
//   conv2d_kernel      3x3 (stride 1 / 2) and 1x1/2 convolutions on NHWC as an implicit GEMM: a workgroup owns
//                      NA agents x TR output rows x all columns (<= 224 pixels = 14 M-tiles) x 64 output channels,
//                      stages the input patch of a 16-channel chunk once in LDS and reads the 9 taps as 9 shifted
//                      ds_read_b128; weights come pre-packed in MFMA fragment order (never through LDS);
//                      BatchNorm + residual + ReLU fused.
//   context_head_kernel  avg-pool + fc + both MLPs, 4 agents per workgroup.
#include "cld_kernels.h"

namespace cld {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4f cbuf_load16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// =============================================================================================
// stem: Conv2d(34, 64, 7, stride 2, pad 3, bias=False) + BatchNorm2d(eval) + ReLU
// =============================================================================================
namespace stem {
constexpr int CIN = 34, HIN = 224, HO = 112, COUT = 64;
#ifndef STEM_TR
#define STEM_TR 2
#endif
constexpr int TR = STEM_TR;               // output rows per workgroup
constexpr int PRW = 2 * TR + 5;           // input rows of the strip: 9
constexpr int RS = 233;                   // LDS row stride (floats): odd, so taps that wrap to the next kernel row hit other banks
constexpr int SMT = TR * HO / 16;         // 14 M-tiles, 7 per output row
constexpr int NQ = 13;                    // k-steps per input plane: 49 taps padded to 52
constexpr int PIECES = PRW * (HIN / 4);   // 16-byte pieces per strip: 504
constexpr int NPIECE = (PIECES + 255) / 256;
constexpr int BUF = PRW * RS;             // floats per strip image
}  // namespace stem

// image [B,34,224,224] fp32 NCHW; wq: [34][4][4][64] float4 (plane c, k-step group qg, N tile, lane) -> the lane's B values
// of k-steps 4qg..4qg+3; scale/shift [64]: folded BatchNorm; y [B,112,112,64] NHWC.
// three workgroups per CU (<= 168 VGPRs: the plane batch is 2 deep for that) so that one workgroup's plane loads and
// barriers hide behind the MFMAs of the others: 1.44 -> 1.25 ms structured, 6.59 -> 6.42 ms dense per 256 agents
// pooled != null: MaxPool2d(3, stride 2, pad 1) (base_models.py:559-614: resnet18's maxpool) is applied here instead of by a launch of its
// own -- y is not written at all.  Pooled pixel (pr, pc) is the maximum over conv rows 2 pr - 1 .. 2 pr + 1 and columns 2 pc - 1 .. 2 pc + 1.
// A workgroup owns conv rows 2 k and 2 k + 1, a lane four consecutive columns c0 .. c0 + 3 (c0 a multiple of 4) of one channel: it takes
// the maxima it can form in registers and hands them to the pooled tensor with atomic max -- exact and order-independent; the values are
// ReLU outputs (>= 0), so their bit patterns compare like unsigned integers and the caller's zero fill is the identity.
__global__ __launch_bounds__(256, stem::TR == 2 ? 3 : 2) void stem_conv_kernel(const float* __restrict__ image, const float* __restrict__ wq,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       float* __restrict__ y, float* __restrict__ pooled, int B) {
    using namespace stem;
    __shared__ float lds[2 * BUF];
    __shared__ int plane_nz[CIN];                   // 7-bit mask per input plane: which 16-column output blocks of this strip see a non-zero value
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> (agent, strip).  Neighbouring strips of an agent share 5 of their 9 input rows; dispatched as a (strip,
    // agent) grid they land on different XCDs (blocks are dealt round-robin) and every strip is fetched from beyond L2
    // (2.3x the raster, round-1 PMC).  Here the 56 strips of an agent get block ids L, L + 8, L + 16, ...: one XCD, back to
    // back, so the shared rows hit that XCD's L2.  (A speed assumption only; any placement computes the same result.)
    int b, strip;
    {
        const int L = blockIdx.x, NS = HO / TR, full = (B >> 3) << 3;
        if (L < full * NS) { b = (L / (8 * NS)) * 8 + (L & 7); strip = (L >> 3) % NS; }
        else { const int r = B - full, l2 = L - full * NS; b = full + l2 % r; strip = l2 / r; }
    }
    const int r0 = strip * TR;                      // first output row
    const int i16 = lane & 15, kk = lane >> 4;

    // ---- staging map: piece -> (byte offset inside plane 0 of this agent's raster | out of range, LDS word) ----
    const size_t img_bytes = (size_t)CIN * HIN * HIN * 4;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(image) + (size_t)b * CIN * HIN * HIN, 0, (int)img_bytes, 0x00020000);
    int voff[NPIECE], soff[NPIECE], pmask[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int idx = tid + 256 * i;
        const bool ok = idx < PIECES;
        const int row = ok ? idx / (HIN / 4) : 0, c4 = ok ? idx % (HIN / 4) : 0;
        const int irow = 2 * r0 - 3 + row;
        const bool in = ok && irow >= 0 && irow < HIN;
        voff[i] = in ? (irow * HIN + 4 * c4) * 4 : (int)img_bytes;      // out of range -> the buffer load returns 0
        soff[i] = ok ? row * RS + 3 + 4 * c4 : -1;
        // output-column blocks (16 output columns = one M-tile per row) whose 7-wide windows can see this piece:
        // block cb reads strip columns [32 cb, 32 cb + 36]
        const int lo = 3 + 4 * c4, hi = lo + 3;
        int mk = 0;
#pragma unroll
        for (int cb = 0; cb < 7; ++cb) mk |= (32 * cb <= hi && 32 * cb + 36 >= lo) ? (1 << cb) : 0;
        pmask[i] = mk;
    }
    // zero the 3-column halos of both images once (never overwritten)
    for (int i = tid; i < 2 * PRW * 6; i += 256) {
        const int bufi = i / (PRW * 6), r = (i / 6) % PRW, c = i % 6;
        lds[bufi * BUF + r * RS + (c < 3 ? c : 224 + c)] = 0.f;
    }
    if (tid < CIN) plane_nz[tid] = 0;
    __syncthreads();

    // ---- pass 1: which planes does this strip need at all?  The history planes of the raster are almost empty (one +1 pixel
    // for the agent and a -1 per neighbour, trajdata_utils.py:123-156): the strip of such a plane is all zeros for most
    // workgroups and contributes exactly nothing.  One streaming read of the 34 strips (eight planes = sixteen 16-byte loads
    // per thread in flight, no LDS, no barrier) decides it from the registers; an empty strip costs that read and nothing
    // else -- until round 2 every plane went registers -> LDS -> barrier -> flag test, 0.62 ms of the 1.25 ms per 256 agents.
    // plane_nz[c] = 7-bit mask of the 16-column output blocks that see a non-zero value of plane c.  Exact for any input
    // (a zero window contributes +-0 to every sum); dense rasters simply keep every plane.
    {
        constexpr int G = 8;
        for (int c0 = 0; c0 < CIN; c0 += G) {
            v4f v[G][NPIECE];
#pragma unroll
            for (int k = 0; k < G; ++k)
#pragma unroll
                for (int i = 0; i < NPIECE; ++i)
                    v[k][i] = cbuf_load16(rsx, voff[i], (c0 + k < CIN ? c0 + k : CIN - 1) * (HIN * HIN * 4));
#pragma unroll
            for (int k = 0; k < G; ++k) {
                if (c0 + k >= CIN) break;
                int mk = 0;
#pragma unroll
                for (int i = 0; i < NPIECE; ++i) {
                    const v4f x = v[k][i];
                    mk |= ((x[0] != 0.f) | (x[1] != 0.f) | (x[2] != 0.f) | (x[3] != 0.f)) ? pmask[i] : 0;
                }
                if (__builtin_amdgcn_ballot_w64(mk != 0) != 0) {        // wave-uniform: most strips stop here
                    int wm = 0;
#pragma unroll
                    for (int cb = 0; cb < 7; ++cb) wm |= (__builtin_amdgcn_ballot_w64((mk >> cb) & 1) != 0) ? (1 << cb) : 0;
                    if (lane == 0) atomicOr(&plane_nz[c0 + k], wm);
                }
            }
        }
    }
    __syncthreads();

    // per-lane LDS byte offsets of the 13 k-steps: tap k = 4q + kk -> (kh, kw); pixel column 2 * i16
    int qoff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int k = 4 * q + kk;
        const int kh = k < 49 ? k / 7 : 0, kw = k < 49 ? k % 7 : 0;     // k >= 49: zero weight, any valid address
        qoff[q] = (kh * RS + kw + 2 * i16) * 4;
    }
    const v4f* wq4 = reinterpret_cast<const v4f*>(wq) + wave * 64 + lane;

    v4f acc[SMT];
#pragma unroll
    for (int m = 0; m < SMT; ++m) acc[m] = v4f{0.f, 0.f, 0.f, 0.f};

    // ---- pass 2: only the planes with something in them (their strips are L2-warm from pass 1): registers -> LDS image
    // (two images in turn: ONE barrier per plane), the next such plane's loads in flight under this plane's MFMAs ----
    auto next_plane = [&](int c) {                   // wave-uniform scan of the 34 flags
        while (c < CIN && __builtin_amdgcn_readfirstlane(plane_nz[c]) == 0) ++c;
        return c;
    };
    v4f st[NPIECE];
    auto load_plane = [&](int c) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) st[i] = cbuf_load16(rsx, voff[i], c * (HIN * HIN * 4));
    };
    const char* ldsb = reinterpret_cast<const char*>(lds);
    int c = next_plane(0), cu = 0;
    if (c < CIN) load_plane(c);
    while (c < CIN) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i)
            if (soff[i] >= 0) {
                float* d = lds + cu * BUF + soff[i];
                d[0] = st[i][0]; d[1] = st[i][1]; d[2] = st[i][2]; d[3] = st[i][3];
            }
        const int blocks = __builtin_amdgcn_readfirstlane(plane_nz[c]);
        const int cn = next_plane(c + 1);
        if (cn < CIN) load_plane(cn);
        __syncthreads();
        v4f bq[4];
#pragma unroll
        for (int qg = 0; qg < 4; ++qg) bq[qg] = wq4[(c * 4 + qg) * 256];
        const int ib = cu * BUF * 4;
        if (blocks == 0x7f) {
            // the 14 A values of k-step q+1 are read while k-step q's MFMAs issue (one ds_read_b32 behind each MFMA, pinned)
            float av[2][SMT];
#pragma unroll
            for (int m = 0; m < SMT; ++m)
                av[0][m] = *reinterpret_cast<const float*>(ldsb + qoff[0] + ib + (2 * (m / 7) * RS + 32 * (m % 7)) * 4);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int cur = q & 1;
                const float bv = bq[q >> 2][q & 3];
#pragma unroll
                for (int m = 0; m < SMT; ++m) {
                    const int imm = (2 * (m / 7) * RS + 32 * (m % 7)) * 4;
                    if (q + 1 < NQ) av[cur ^ 1][m] = *reinterpret_cast<const float*>(ldsb + qoff[q + 1 < NQ ? q + 1 : q] + ib + imm);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][m], bv, acc[m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // sparse plane: only the column blocks that see a non-zero value (every row of the block: M-tiles cb, cb + 7, ...)
#pragma unroll
            for (int cb = 0; cb < 7; ++cb) {
                if (!((blocks >> cb) & 1)) continue;
                const int imm0 = ib + (32 * cb) * 4;
                float a[TR];
#pragma unroll
                for (int r = 0; r < TR; ++r) a[r] = *reinterpret_cast<const float*>(ldsb + qoff[0] + imm0 + r * 2 * RS * 4);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float bv = bq[q >> 2][q & 3];
                    float cv[TR];
#pragma unroll
                    for (int r = 0; r < TR; ++r) cv[r] = a[r];
                    if (q + 1 < NQ) {
#pragma unroll
                        for (int r = 0; r < TR; ++r) a[r] = *reinterpret_cast<const float*>(ldsb + qoff[q + 1 < NQ ? q + 1 : q] + imm0 + r * 2 * RS * 4);
                    }
#pragma unroll
                    for (int r = 0; r < TR; ++r) acc[cb + 7 * r] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[r], bv, acc[cb + 7 * r], 0, 0, 0);
                }
            }
        }
        // the image read here (cu) is rewritten two planes on, behind the next plane's barrier: no second barrier needed
        cu ^= 1;
        c = cn;
    }

    // epilogue: folded BatchNorm + ReLU; lane holds rows 4 (lane >> 4) + r of column n = 16 wave + i16 of every M-tile
    const int n = 16 * wave + i16;
    const float sc = scale[n], sh = shift[n];
    if (pooled) {
        static_assert(TR == 2 || TR == 4, "the fused pool pairs conv rows 2 k, 2 k + 1");
        unsigned* pl = reinterpret_cast<unsigned*>(pooled) + (size_t)b * 56 * 56 * COUT + n;
        auto amax = [&](const int pr, const int pc, const float v) {
            if (pr < 56 && pc < 56) atomicMax(pl + ((size_t)pr * 56 + pc) * COUT, __builtin_bit_cast(unsigned, v));
        };
#pragma unroll
        for (int rp = 0; rp < TR / 2; ++rp)                  // row pairs (2 k, 2 k + 1) of this workgroup
#pragma unroll
            for (int mc = 0; mc < 7; ++mc) {
                const int k = r0 / 2 + rp, c0 = 16 * mc + 4 * kk;
                float e[4], o[4];                            // even row 2 k, odd row 2 k + 1
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = fmaxf(acc[mc + 7 * (2 * rp)][r] * sc + sh, 0.f);
                    o[r] = fmaxf(acc[mc + 7 * (2 * rp + 1)][r] * sc + sh, 0.f);
                }
                // horizontal pieces: pooled column c0 / 2 sees columns c0, c0 + 1 of this lane (and c0 - 1 of its neighbour), c0 / 2 + 1 sees
                // c0 + 1 .. c0 + 3, c0 / 2 + 2 sees c0 + 3
                const float oa = fmaxf(o[0], o[1]), ob = fmaxf(fmaxf(o[1], o[2]), o[3]), oc = o[3];
                const float va = fmaxf(fmaxf(e[0], e[1]), oa), vb = fmaxf(fmaxf(fmaxf(e[1], e[2]), e[3]), ob), vc = fmaxf(e[3], oc);
                amax(k, c0 / 2, va); amax(k, c0 / 2 + 1, vb); amax(k, c0 / 2 + 2, vc);                  // pooled row k: rows 2 k, 2 k + 1
                amax(k + 1, c0 / 2, oa); amax(k + 1, c0 / 2 + 1, ob); amax(k + 1, c0 / 2 + 2, oc);      // pooled row k + 1: row 2 k + 1 is its row above
            }
        return;
    }
#pragma unroll
    for (int m = 0; m < SMT; ++m) {
        const int orow = r0 + m / 7;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ocol = 16 * (m % 7) + 4 * kk + r;
            const float v = fmaxf(acc[m][r] * sc + sh, 0.f);
            y[(((size_t)b * HO + orow) * HO + ocol) * COUT + n] = v;
        }
    }
}

hipError_t launch_stem_conv(const float* image, const float* wq, const float* scale, const float* shift, float* y, float* pooled, int B,
                            hipStream_t s) {
    if (pooled) {      // the fused max-pool accumulates by atomic max: the pooled tensor starts at 0 (= the smallest ReLU output)
        hipError_t e = hipMemsetAsync(pooled, 0, (size_t)B * 56 * 56 * stem::COUT * sizeof(float), s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(stem_conv_kernel, dim3((stem::HO / stem::TR) * B), dim3(256), 0, s, image, wq, scale, shift, y, pooled, B);
    return hipGetLastError();
}

// =============================================================================================
// MaxPool2d(3, stride 2, pad 1) on NHWC [B,112,112,64] -> [B,56,56,64]; one thread = one pixel x 4 channels
// =============================================================================================
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int B) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * 56 * 56 * 16;
    if (idx >= total) return;
    const int c4 = (int)(idx & 15);
    long p = idx >> 4;
    const int ow = (int)(p % 56); p /= 56;
    const int oh = (int)(p % 56);
    const int b = (int)(p / 56);
    v4f m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dh = -1; dh <= 1; ++dh) {
        const int ih = 2 * oh + dh;
        if (ih < 0 || ih >= 112) continue;
#pragma unroll
        for (int dw = -1; dw <= 1; ++dw) {
            const int iw = 2 * ow + dw;
            if (iw < 0 || iw >= 112) continue;
            const v4f v = *reinterpret_cast<const v4f*>(x + (((size_t)b * 112 + ih) * 112 + iw) * 64 + 4 * c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
        }
    }
    *reinterpret_cast<v4f*>(y + (size_t)idx * 4) = m;
}
hipError_t launch_maxpool(const float* x, float* y, int B, hipStream_t s) {
    const long total = (long)B * 56 * 56 * 16;
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, B);
    return hipGetLastError();
}

// =============================================================================================
// conv2d on NHWC: 3x3 pad 1 (stride 1 | 2) and 1x1 stride 2, + folded BatchNorm [+ residual] [+ ReLU]
// =============================================================================================
// KH: 3 | 1; S: conv stride; HIN: input height = width; TR: output rows per tile; NA: agents per tile; NB: LDS images
template <int KH, int S, int HIN, int TR, int NA, int NB>
struct C2 {
    static constexpr int PAD = KH / 2;
    static constexpr int HO = HIN / S;
    static constexpr int SS = (KH == 1) ? S : 1;        // 1x1/2: only the sampled pixels are staged
    static constexpr int SL = (KH == 1) ? 1 : S;        // pixel stride inside the staged patch
    static constexpr int PR = (TR - 1) * SL + KH;       // patch rows / columns per agent
    static constexpr int PW = (HO - 1) * SL + KH;
    static constexpr int ROWS = NA * PR * PW;           // LDS rows (one pixel x KC channels each)
    static constexpr int KC = 16, SROW = 24;            // 16-channel chunks; row stride 24 floats: conflict-free b128 fragment reads
    static constexpr int PX = NA * TR * HO;             // output pixels per tile
    static constexpr int NMT = (PX + 15) / 16;
    static constexpr int NTAPS = KH * KH;
    static constexpr int PIECES = ROWS * (KC / 4);
    static constexpr int NPIECE = (PIECES + 255) / 256;
    static constexpr int IMG = ROWS * SROW;             // floats per LDS image
    static constexpr size_t LDS_BYTES = (size_t)NB * IMG * 4;
    static_assert(HO % TR == 0, "row tiles must divide the output height");
    static_assert(NMT <= 14 && LDS_BYTES <= 160 * 1024, "tile budget");
};

struct Conv2dArgs {
    const float* x;        // [B, HIN, HIN, cin]
    const float* wfrag;    // pack_conv_weights layout: slab (16-channel group, tap) = [cout/16][64 lanes][4]
    const float* scale;    // [cout] folded BatchNorm
    const float* shift;
    const float* res;      // [B, HO, HO, cout] or null
    float* y;              // [B, HO, HO, cout]
    int B, cin, cout, relu;
};

// The 56x56 and 28x28 layers (K = 576 / 1,152: few chunks per workgroup, so prologue and epilogue weigh most) run three
// workgroups per CU: a rolling 3-fragment window instead of a tap's worth of A fragments (<= 168 VGPRs) and, at 56x56, a
// single LDS image.  Measured per launch, 256 agents: 572 -> 501 us (56x56), 541 -> 495 us (28x28); the 14x14 / 7x7 layers
// are faster with the whole-tap prefetch and two workgroups (472 vs 521 us) and keep it.
template <int KH, int S, int HIN, int TR, int NA, int NB>
__global__ __launch_bounds__(256, HIN >= 28 ? 3 : 1) void conv2d_kernel(const Conv2dArgs p) {
    constexpr bool ROLL = HIN >= 28;
    typedef C2<KH, S, HIN, TR, NA, NB> G;
    extern __shared__ __attribute__((aligned(16))) float lds2[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NTR = G::HO / TR;
    const int a0 = (blockIdx.x / NTR) * NA;
    const int r0 = (blockIdx.x % NTR) * TR;
    const int ntile_g = blockIdx.y * 4 + wave;
    const int ntn = p.cout >> 4;
    const int nchunk = p.cin / G::KC;

    // ---- staging map ----
    const size_t agent_bytes = (size_t)HIN * HIN * p.cin * 4;
    const int oob = (int)(NA * agent_bytes);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x) + (size_t)a0 * HIN * HIN * p.cin, 0, oob, 0x00020000);
    int voff[G::NPIECE], soff[G::NPIECE];
#pragma unroll
    for (int i = 0; i < G::NPIECE; ++i) {
        const int idx = tid + 256 * i;
        const bool ok = idx < G::PIECES;
        const int lr = ok ? idx >> 2 : 0, c4 = idx & 3;
        const int a = lr / (G::PR * G::PW), rem = lr % (G::PR * G::PW);
        const int pr = rem / G::PW, pc = rem % G::PW;
        const int ir = (r0 * G::SL + pr) * G::SS - G::PAD, ic = pc * G::SS - G::PAD;
        const bool in = ok && ir >= 0 && ir < HIN && ic >= 0 && ic < HIN && (a0 + a) < p.B;
        voff[i] = in ? (int)(a * agent_bytes) + ((ir * HIN + ic) * p.cin + 4 * c4) * 4 : oob;
        soff[i] = ok ? lr * G::SROW + 4 * c4 : -1;
    }
    v4f st[G::NPIECE];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < G::NPIECE; ++i) st[i] = cbuf_load16(rsx, voff[i], c * (G::KC * 4));
    };
    auto store_chunk = [&](int bufi) {
#pragma unroll
        for (int i = 0; i < G::NPIECE; ++i)
            if (soff[i] >= 0) *reinterpret_cast<v4f*>(lds2 + bufi * G::IMG + soff[i]) = st[i];
    };

    // ---- A fragment offsets (bytes): lane (i16, kk) of M-tile m reads 4 channels of pixel 16 m + i16 ----
    const int i16 = lane & 15, kk = lane >> 4;
    int aoff[G::NMT];
#pragma unroll
    for (int m = 0; m < G::NMT; ++m) {
        int px = 16 * m + i16;
        if (px >= G::PX) px = 0;                       // ragged last M-tile: computed, never stored
        const int a = px / (TR * G::HO), rem = px % (TR * G::HO);
        const int r = rem / G::HO, w = rem % G::HO;
        aoff[m] = (((a * G::PR + r * G::SL) * G::PW + w * G::SL) * G::SROW + 4 * kk) * 4;
    }
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wfrag), 0, nchunk * G::NTAPS * ntn * 1024, 0x00020000);
    const int wlane = lane * 16;

    v4f acc[G::NMT];
#pragma unroll
    for (int m = 0; m < G::NMT; ++m) acc[m] = v4f{0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    const char* ldsb = reinterpret_cast<const char*>(lds2);
    for (int c = 0; c < nchunk; ++c) {
        const int bufi = (NB == 2) ? (c & 1) : 0;
        const bool more = c + 1 < nchunk;
        if (more) load_chunk(c + 1);
        const int boff = bufi * G::IMG * 4;
        // fragments of tap t+1 are read while tap t's MFMAs issue: one ds_read_b128 behind every 4 MFMAs, pinned with
        // sched_barrier (left alone the compiler bunches the reads in front of each tap and the MFMA pipe starts every
        // tap behind an LDS bubble); weight fragments run one tap ahead
        v4f bcur = cbuf_load16(rsw, wlane, ((c * G::NTAPS + 0) * ntn + ntile_g) * 1024);
        if constexpr (ROLL) {
            // rolling fragment window: item i = (tap i / NMT, M-tile i % NMT); the fragment of item i + 2 is read behind the
            // MFMAs of item i (8 MFMAs = 256 cycles of cover), so 3 fragments are live instead of a whole tap's worth
            constexpr int NI = G::NTAPS * G::NMT;
            auto frag = [&](int i) {
                const int t = i / G::NMT, m = i % G::NMT;
                return *reinterpret_cast<const v4f*>(ldsb + aoff[m] + boff + ((t / KH) * G::PW + (t % KH)) * G::SROW * 4);
            };
            v4f ar[3];
            ar[0] = frag(0);
            ar[1] = frag(NI > 1 ? 1 : 0);
            v4f bnext = bcur;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int t = i / G::NMT, m = i % G::NMT;
                if (m == 0 && t + 1 < G::NTAPS) bnext = cbuf_load16(rsw, wlane, ((c * G::NTAPS + t + 1) * ntn + ntile_g) * 1024);
                if (i + 2 < NI) ar[(i + 2) % 3] = frag(i + 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i % 3][e], bcur[e], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (m == G::NMT - 1) bcur = bnext;
            }
        } else {
            v4f af[2][G::NMT];
#pragma unroll
            for (int m = 0; m < G::NMT; ++m) af[0][m] = *reinterpret_cast<const v4f*>(ldsb + aoff[m] + boff);
#pragma unroll
            for (int t = 0; t < G::NTAPS; ++t) {
                const int cur = t & 1;
                const bool nxt = t + 1 < G::NTAPS;
                const v4f bnext = nxt ? cbuf_load16(rsw, wlane, ((c * G::NTAPS + t + 1) * ntn + ntile_g) * 1024) : bcur;
                const int toff = (((t + 1) / KH) * G::PW + ((t + 1) % KH)) * G::SROW * 4;
#pragma unroll
                for (int m = 0; m < G::NMT; ++m) {
                    if (nxt) af[cur ^ 1][m] = *reinterpret_cast<const v4f*>(ldsb + aoff[m] + boff + toff);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cur][m][e], bcur[e], acc[m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                bcur = bnext;
            }
        }
        if (NB == 1) __syncthreads();                  // everyone is done reading before the single image is rewritten
        if (more) store_chunk(NB == 2 ? (bufi ^ 1) : 0);
        __syncthreads();
    }

    // ---- epilogue: folded BatchNorm, residual, ReLU; lane holds rows 4 kk + r, column n ----
    const int n = blockIdx.y * 64 + 16 * wave + i16;
    const float sc = p.scale[n], sh = p.shift[n];
#pragma unroll
    for (int m = 0; m < G::NMT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int px = 16 * m + 4 * kk + r;
            if (px >= G::PX) continue;
            const int a = px / (TR * G::HO), rem = px % (TR * G::HO);
            if (a0 + a >= p.B) continue;
            const int orow = r0 + rem / G::HO, ocol = rem % G::HO;
            const size_t o = (((size_t)(a0 + a) * G::HO + orow) * G::HO + ocol) * p.cout + n;
            float v = acc[m][r] * sc + sh;
            if (p.res) v += p.res[o];
            if (p.relu) v = fmaxf(v, 0.f);
            p.y[o] = v;
        }
    }
}

template <int KH, int S, int HIN, int TR, int NA, int NB>
static hipError_t launch_conv2d_inst(const Conv2dArgs& a, hipStream_t s) {
    typedef C2<KH, S, HIN, TR, NA, NB> G;
    auto kern = conv2d_kernel<KH, S, HIN, TR, NA, NB>;
    static unsigned long long attr_done = 0;      // one bit per device (a second handle on another device sets it there too)
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(kern), (int)G::LDS_BYTES, &attr_done); e != hipSuccess) return e;
    const int groups = (a.B + NA - 1) / NA;
    dim3 grid(groups * (G::HO / TR), a.cout / 64, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), G::LDS_BYTES, s, a);
    return hipGetLastError();
}

// (KH, S, HIN, TR, NA, NB): the ten conv shapes of resnet18 behind the stem
#define CLD_CONV2D_INSTANCES(X) \
    X(3, 1, 56, 4, 1, 1)        \
    X(3, 2, 56, 4, 1, 1)        \
    X(1, 2, 56, 7, 1, 2)        \
    X(3, 1, 28, 7, 1, 2)        \
    X(3, 2, 28, 7, 1, 1)        \
    X(1, 2, 28, 14, 1, 2)       \
    X(3, 1, 14, 14, 1, 2)       \
    X(3, 2, 14, 7, 2, 1)        \
    X(1, 2, 14, 7, 4, 2)        \
    X(3, 1, 7, 7, 4, 2)

hipError_t launch_conv2d(int kh, int stride, int hin, const float* x, const float* wfrag, const float* scale,
                         const float* shift, const float* res, float* y, int B, int cin, int cout, int relu, hipStream_t s) {
    if (cin % 16 || cout % 64 || B < 1) return hipErrorInvalidValue;
    Conv2dArgs a{x, wfrag, scale, shift, res, y, B, cin, cout, relu};
#define X(KH, S, HIN, TR, NA, NB) \
    if (kh == KH && stride == S && hin == HIN) return launch_conv2d_inst<KH, S, HIN, TR, NA, NB>(a, s);
    CLD_CONV2D_INSTANCES(X)
#undef X
    return hipErrorInvalidValue;
}

// =============================================================================================
// head: avg-pool(7x7) -> fc(512 -> 256) ; state MLP ; concat ; combine MLP.  4 agents per workgroup.
// =============================================================================================
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// LayerNorm (eps 1e-5, biased variance, two-pass) + ReLU over `n` values per agent held as v[g][slot] by thread tid at
// index tid + 256 * slot; red: LDS scratch [4 agents][4 waves]
template <int NSLOT>
__device__ __forceinline__ void ln_relu_4(float (&v)[4][NSLOT], int n, const float* gamma, const float* beta, float* red,
                                          int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    float mean[4], rstd[4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NSLOT; ++k) {
                const bool ok = tid + 256 * k < n;
                const float d = pass == 0 ? v[g][k] : (v[g][k] - mean[g]);
                s += ok ? (pass == 0 ? d : d * d) : 0.f;
            }
            s = wave_sum(s);
            if (lane == 0) red[g * 4 + wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float t = (red[g * 4 + 0] + red[g * 4 + 1] + red[g * 4 + 2] + red[g * 4 + 3]) / (float)n;
            if (pass == 0) mean[g] = t; else rstd[g] = 1.0f / sqrtf(t + 1e-5f);
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int j = tid + 256 * k;
        if (j < n) {
            const float ga = gamma[j], be = beta[j];
#pragma unroll
            for (int g = 0; g < 4; ++g) v[g][k] = fmaxf((v[g][k] - mean[g]) * rstd[g] * ga + be, 0.f);
        }
    }
}

// out[g][j] = b[j] + sum_k in[g][k] * wt[k][j]  for j = tid + 256 * slot < n_out; wt is the TRANSPOSED weight [n_in][n_out]
template <int NSLOT>
__device__ __forceinline__ void linear_4(const float* in /*LDS [4][stride]*/, int stride, int n_in, const float* wt,
                                         const float* bias, int n_out, float (&v)[4][NSLOT], int tid) {
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int j = tid + 256 * k;
        const bool ok = j < n_out;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (ok) {
            // the kernel is ONE workgroup's chain of nine such products (63 workgroups at 250 agents: a single generation), and a product is
            // n_in dependent steps of (weight from L2, four FMAs): sixteen weights are requested at a time so that a step does not wait out a
            // trip to L2 each (same order of additions: results unchanged)
            int i = 0;
            for (; i + 16 <= n_in; i += 16) {
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = wt[(size_t)(i + u) * n_out + j];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    a0 = fmaf(in[i + u], w[u], a0);
                    a1 = fmaf(in[stride + i + u], w[u], a1);
                    a2 = fmaf(in[2 * stride + i + u], w[u], a2);
                    a3 = fmaf(in[3 * stride + i + u], w[u], a3);
                }
            }
            for (; i < n_in; ++i) {
                const float w = wt[(size_t)i * n_out + j];
                a0 = fmaf(in[i], w, a0);
                a1 = fmaf(in[stride + i], w, a1);
                a2 = fmaf(in[2 * stride + i], w, a2);
                a3 = fmaf(in[3 * stride + i], w, a3);
            }
            const float bj = bias[j];
            a0 += bj; a1 += bj; a2 += bj; a3 += bj;
        }
        v[0][k] = a0; v[1][k] = a1; v[2][k] = a2; v[3][k] = a3;
    }
}

template <int NSLOT>
__device__ __forceinline__ void put_4(float* dst /*LDS [4][stride]*/, int stride, int off, int n, const float (&v)[4][NSLOT], int tid) {
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int j = tid + 256 * k;
        if (j < n) {
#pragma unroll
            for (int g = 0; g < 4; ++g) dst[g * stride + off + j] = v[g][k];
        }
    }
}

__global__ __launch_bounds__(256) void context_head_kernel(const ContextHeadArgs a) {
    __shared__ float pooled[4][512];
    __shared__ float bufA[4][320];
    __shared__ float bufB[4][320];
    __shared__ float red[16];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * 4;

    // avg-pool over the 49 pixels of layer4's output [B,7,7,512] (adaptive_avg_pool2d((1,1)))
    if (!a.map_feat_in) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int b = b0 + g;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ch = tid + 256 * k;
                float s = 0.f;
                if (b < a.B) {
                    float fv[49];                   // (all 49 requested before the first add: same order of additions)
#pragma unroll
                    for (int px = 0; px < 49; ++px) fv[px] = a.feat[((size_t)b * 49 + px) * 512 + ch];
#pragma unroll
                    for (int px = 0; px < 49; ++px) s += fv[px];
                }
                pooled[g][ch] = s * (1.0f / 49.0f);
            }
        }
    }
    // current states -> bufB[g][0..3]
    if (tid < 16) {
        const int g = tid >> 2, b = b0 + g;
        bufB[g][tid & 3] = (b < a.B) ? a.curr_states[(size_t)b * 4 + (tid & 3)] : 0.f;
    }
    __syncthreads();

    // map branch: fc 512 -> 256 (the 'map_model.fc' node: no output activation), or a map feature handed in
    if (a.map_feat_in) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int b = (b0 + g < a.B) ? b0 + g : a.B - 1;
            bufA[g][64 + tid] = a.map_feat_in[(size_t)b * a.map_feat_stride + tid];
        }
    } else {
        float v[4][1];
        linear_4<1>(&pooled[0][0], 512, 512, a.fc_wt, a.fc_b, 256, v, tid);
        put_4<1>(&bufA[0][0], 320, 64, 256, v, tid);
        if (a.map_feat_out) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (b0 + g < a.B) a.map_feat_out[(size_t)(b0 + g) * 256 + tid] = v[g][0];
        }
    }
    // state branch: 4 -> 64 (LN, ReLU) -> 64 (LN, ReLU) -> 64
    {
        float v[4][1];
        linear_4<1>(&bufB[0][0], 320, 4, a.s_wt[0], a.s_b[0], 64, v, tid);
        ln_relu_4<1>(v, 64, a.s_g[0], a.s_be[0], red, tid);
        __syncthreads();
        put_4<1>(&bufB[0][0], 320, 0, 64, v, tid);
        __syncthreads();
        linear_4<1>(&bufB[0][0], 320, 64, a.s_wt[1], a.s_b[1], 64, v, tid);
        ln_relu_4<1>(v, 64, a.s_g[1], a.s_be[1], red, tid);
        __syncthreads();
        put_4<1>(&bufB[0][0], 320, 0, 64, v, tid);
        __syncthreads();
        linear_4<1>(&bufB[0][0], 320, 64, a.s_wt[2], a.s_b[2], 64, v, tid);
        put_4<1>(&bufA[0][0], 320, 0, 64, v, tid);          // cat([state_feat, map_feat]) -> bufA[g][0..319]
    }
    __syncthreads();

    // combine MLP: 320 -> 320 -> 320 -> 256 -> 256 (each LN + ReLU) -> 256
    float (*src)[320] = bufA;
    float (*dst)[320] = bufB;
    const int dims[6] = {320, 320, 320, 256, 256, 256};
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        float v[4][2];
        linear_4<2>(&src[0][0], 320, dims[l], a.c_wt[l], a.c_b[l], dims[l + 1], v, tid);
        if (l < 4) {
            ln_relu_4<2>(v, dims[l + 1], a.c_g[l], a.c_be[l], red, tid);
            put_4<2>(&dst[0][0], 320, 0, dims[l + 1], v, tid);
            __syncthreads();
            float (*t)[320] = src; src = dst; dst = t;
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (b0 + g < a.B) a.cond_out[(size_t)(b0 + g) * 256 + tid] = v[g][0];
        }
    }
}

hipError_t launch_context_head(const ContextHeadArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(context_head_kernel, dim3((a.B + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace cld
