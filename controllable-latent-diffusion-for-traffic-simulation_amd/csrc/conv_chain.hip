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
#include "chain_common.h"

namespace cld {

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
    static unsigned long long attr_done = 0;      // one bit per device (a second handle on another device sets it there too)
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(chain_head_kernel<AG>), 160 * 1024, &attr_done); e != hipSuccess) return e;
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
    static unsigned long long attr_done = 0;      // one bit per device (a second handle on another device sets it there too)
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(chain_tail_kernel<AG>), 160 * 1024, &attr_done); e != hipSuccess) return e;
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
