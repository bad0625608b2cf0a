// cld_api.hip -- the C-ABI of libcld_hip (include/cld.h): handle, weight ingestion and
// re-layout, the per-step launch plan of the denoiser, and the sampling loop.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/cld.h"
#include "cld_kernels.h"

using namespace cld;

namespace {

constexpr int T = 52, D = 4, COND = 256, TE = 32, NCB = 1792, ACT = 3328;   // ACT floats/agent/buffer
constexpr int NBUF = 8;

struct BlockDef { const char* name; int cin, cout, L; };
// the 12 residual blocks in execution order (temporal.py:84-115,148-167)
const BlockDef kBlocks[12] = {
    {"model.downs.0.0", 4, 64, 52},   {"model.downs.0.1", 64, 64, 52},
    {"model.downs.1.0", 64, 128, 26}, {"model.downs.1.1", 128, 128, 26},
    {"model.downs.2.0", 128, 256, 13}, {"model.downs.2.1", 256, 256, 13},
    {"model.mid_block1", 256, 256, 13}, {"model.mid_block2", 256, 256, 13},
    {"model.ups.0.0", 512, 128, 13},  {"model.ups.0.1", 128, 128, 13},
    {"model.ups.1.0", 256, 64, 26},   {"model.ups.1.1", 64, 64, 26},
};

struct ConvLayer {
    ConvGeom g{};        // kc / nwn / ks are filled per launch by pick_tiling()
    bool has_a = false, has_b = false;   // which tilings have a kernel instance
    float *wfrag = nullptr, *bias = nullptr, *gamma = nullptr, *beta = nullptr;
    float* ufrag_edge = nullptr;   // the same with the raw taps 0..3 behind them, 12 planes per chunk (wino1d_edge.hip: whole-item launches)
    float* ufrag = nullptr;   // Winograd-domain filters G g of a k5 layer that has a Winograd form (exact-fp32 handles): as a launch of its own
                              // (wino1d_kernels.hip: wino_launch) or inside the layer chains (chain_wino.hip: the 64 -> 64 layers at L = 52 / 26)
    bool wino_launch = false;
    int c_out = 0, c1_real = 0, c1_pad = 0, c2 = 0, ly = 0, off0 = 0, orow0 = 0;
    int cb_off = -1;    // offset into the 1792-wide cond/time bias rows, -1 = none
};

struct ResBlock { ConvLayer c0, c1, res; bool has_res = false; };
constexpr long kProfStride = 10;
constexpr int kChainSmall = 944;      // launch sets of at most this many rows take the one-agent chain tiles (chain_tile): measured per U-Net
                                      // evaluation, one- vs four-agent tiles: 245 / 320 us at 64 rows, 630 / 676 at 512, 916 / 928 at 768,
                                      // 1,120 / 1,096 at 1,008, 1,081 / 1,052 at 1,024, 2,004 / 1,939 at 2,048
const char* const kResnet = "context_encoder.map_encoder.encoder_heads.map_model.";

}  // namespace

struct cld_handle_s {
    cld_config cfg{};
    int stride = 1;                                  // DmModel.stride (dm_model.py:25,119): the loop visits i = ..., 2 stride, stride, 0
    int precision = CLD_PRECISION_F32;               // cfg.precision
    int force_kernel[6] = {0, 0, 0, 0, 0, 0};           // cld_debug_force_kernel: formulation of the guide / decode / encode kernels, of the U-Net's layer chains and of the ContextEncoder's 3x3 convolutions (0 = the library's choice)
    std::string err;
    std::map<std::string, std::vector<float>> w;     // host copies keyed by reference state_dict name
    std::map<std::string, size_t> expect;            // name -> numel
    bool finalized = false, has_decoder = false, has_unet = false;
    std::vector<void*> dev_allocs;
    // schedule (host, fp32 as in dm_model.py:29-56)
    std::vector<float> x_t_cof, noise_cof, plvc, sqrt_acp, sqrt_1m_acp, sqrt_recip_acp, sqrt_recipm1_acp;
    float* qs_tab = nullptr;            // device [2][n_timesteps]: sqrt(alphas_cumprod), sqrt(1 - alphas_cumprod) (q_sample, dm_model.py:91-96)
    // device-side model
    ResBlock blocks[12];
    ConvLayer down[2], upT[2][2], final_cb;
    float *wc = nullptr, *cbias_b = nullptr, *tb = nullptr, *head_w = nullptr, *head_b = nullptr, *res4_w = nullptr, *res4_b = nullptr;
    float* head_wfrag = nullptr;        // final_conv.1 as an MFMA N tile (the tail chain of conv_chain.hip)
    const HeadArgs* fuse_upd = nullptr; // set by the caller of run_unet: a DDPM update the tail chain may apply itself (plain steps, no CFG combine)
    bool upd_fused = false;             // ... and whether the last run_unet did
    bool eps_in_buf7 = false;           // the last run_unet left the noise prediction [b_pad,52,4] in buf[7] (chains) instead of final_conv.0's activations
    DecoderWeights dec{};
    EncoderWeights enc{};
    bool has_encoder = false;
    // ContextEncoder (optional): stem, 19 NHWC convolutions, head
    struct Conv2dLayer { float *wfrag = nullptr, *ufrag = nullptr, *ufrag44 = nullptr, *scale = nullptr, *shift = nullptr; int kh = 3, stride = 1, hin = 56, cin = 64, cout = 64; };   // ufrag: the Winograd-domain filters of a 3x3 / stride-1 layer (wino_kernels.hip)
    bool has_context = false;
    float *stem_w = nullptr, *stem_scale = nullptr, *stem_shift = nullptr;
    Conv2dLayer rn_conv[4][2][2], rn_ds[4];
    ContextHeadArgs ctx_head{};
    DynParams dyn{};
    // optional HIP-event timing of the dominant conv kernel instance (k5 GN+Mish block -> 256 channels, L = 13)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;     // pairs (start, stop)
    size_t prof_used = 0;
    double prof_flop = 0.0;              // algorithmic FLOP of the timed launches
    double prof_exec_flop = 0.0;         // FLOP their MFMAs executed in the form each launch took (Winograd F(4, 5): 8 x 4 k-steps per agent
                                         // and channel pair instead of 5 x 13)
    // the same two counts over ALL launches of the most recent U-Net evaluation (cld_profile_read_executed)
    double eval_alg_flop = 0.0, eval_exec_flop = 0.0;
    int eval_launches = 0;
    // diagnostic (-DCLD_STAMPS builds): launch index within a U-Net evaluation that receives the stamp buffer
    unsigned long long* stamp_buf = nullptr;
    int stamp_layer = -1, launch_counter = 0;
    size_t lds_floor = 0;                // experiments (cld_debug_lds_floor)
    long eval_counter = 0;               // U-Net evaluations since profile_enable: every kProfStride-th one is timed
};

namespace {

int fail(cld_handle h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}
#define HIPCK(h, expr)                                                                           \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return fail(h, CLD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));     \
    } while (0)

float* upload(cld_handle h, const std::vector<float>& v, hipStream_t s, hipError_t* err) {
    float* d = nullptr;
    *err = hipMalloc(reinterpret_cast<void**>(&d), v.size() * sizeof(float));
    if (*err != hipSuccess) return nullptr;
    h->dev_allocs.push_back(d);
    *err = hipMemcpyAsync(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice, s);
    return d;
}

void add_expect(cld_handle h) {
    auto& e = h->expect;
    auto lin = [&](const std::string& p, int o, int i) { e[p + ".weight"] = (size_t)o * i; e[p + ".bias"] = o; };
    auto conv = [&](const std::string& p, int o, int i, int k) { e[p + ".weight"] = (size_t)o * i * k; e[p + ".bias"] = o; };
    auto gn = [&](const std::string& p, int c) { e[p + ".weight"] = c; e[p + ".bias"] = c; };
    lin("model.time_mlp.1", 128, 32);
    lin("model.time_mlp.3", 32, 128);
    for (const auto& b : kBlocks) {
        const std::string p = b.name;
        lin(p + ".time_mlp.1", b.cout, COND + TE);
        conv(p + ".blocks.0.block.0", b.cout, b.cin, 5);
        gn(p + ".blocks.0.block.2", b.cout);
        conv(p + ".blocks.1.block.0", b.cout, b.cout, 5);
        gn(p + ".blocks.1.block.2", b.cout);
        if (b.cin != b.cout) conv(p + ".residual_conv", b.cout, b.cin, 1);
    }
    conv("model.downs.0.2.conv", 64, 64, 3);
    conv("model.downs.1.2.conv", 128, 128, 3);
    conv("model.ups.0.2.conv", 128, 128, 4);
    conv("model.ups.1.2.conv", 64, 64, 4);
    conv("model.final_conv.0.block.0", 64, 64, 5);
    gn("model.final_conv.0.block.2", 64);
    conv("model.final_conv.1", 4, 64, 1);
    // decoder (lstm_vae.py:28-43)
    e["lstm_dec.lstm.weight_ih_l0"] = 256 * 4;  e["lstm_dec.lstm.weight_hh_l0"] = 256 * 64;
    e["lstm_dec.lstm.bias_ih_l0"] = 256;        e["lstm_dec.lstm.bias_hh_l0"] = 256;
    e["lstm_dec.lstm.weight_ih_l1"] = 256 * 64; e["lstm_dec.lstm.weight_hh_l1"] = 256 * 64;
    e["lstm_dec.lstm.bias_ih_l1"] = 256;        e["lstm_dec.lstm.bias_hh_l1"] = 256;
    e["lstm_dec.cond2hidden.weight"] = 64 * 256; e["lstm_dec.cond2hidden.bias"] = 64;
    e["lstm_dec.hid2act.weight"] = 2 * 64;       e["lstm_dec.hid2act.bias"] = 2;
    // encoder + latent heads (lstm_vae.py:6-19,82-83)
    e["lstm_enc.lstm.weight_ih_l0"] = 256 * 6;  e["lstm_enc.lstm.weight_hh_l0"] = 256 * 64;
    e["lstm_enc.lstm.bias_ih_l0"] = 256;        e["lstm_enc.lstm.bias_hh_l0"] = 256;
    e["lstm_enc.lstm.weight_ih_l1"] = 256 * 64; e["lstm_enc.lstm.weight_hh_l1"] = 256 * 64;
    e["lstm_enc.lstm.bias_ih_l1"] = 256;        e["lstm_enc.lstm.bias_hh_l1"] = 256;
    e["lstm_enc.cond2hidden.weight"] = 64 * 256; e["lstm_enc.cond2hidden.bias"] = 64;
    e["mu.weight"] = 4 * 64;     e["mu.bias"] = 4;
    e["logvar.weight"] = 4 * 64; e["logvar.bias"] = 4;
    // ContextEncoder (models/context_utils.py:8-38): two base_models.MLP (`_model` Sequential: Linear, LayerNorm, ReLU, ...)
    // and torchvision resnet18 under map_encoder.encoder_heads.map_model (base_models.py:559-614)
    auto mlp = [&](const std::string& p, int d_in, std::initializer_list<int> hidden, int d_out) {
        int i = 0, d = d_in;
        for (int hd : hidden) {
            lin(p + "._model." + std::to_string(i), hd, d);
            gn(p + "._model." + std::to_string(i + 1), hd);        // LayerNorm weight / bias
            i += 3; d = hd;
        }
        lin(p + "._model." + std::to_string(i), d_out, d);
    };
    auto bn = [&](const std::string& p, int c) {
        e[p + ".weight"] = c; e[p + ".bias"] = c; e[p + ".running_mean"] = c; e[p + ".running_var"] = c;
    };
    mlp("context_encoder.agent_state_encoder", 4, {64, 64}, 64);
    mlp("context_encoder.process_cond_mlp", 320, {320, 320, 256, 256}, 256);
    const std::string r = kResnet;
    e[r + "conv1.weight"] = (size_t)64 * 34 * 49;
    bn(r + "bn1", 64);
    int cin = 64;
    for (int li = 1; li <= 4; ++li) {
        const int c = 32 << li;
        for (int b = 0; b < 2; ++b) {
            const std::string p = r + "layer" + std::to_string(li) + "." + std::to_string(b);
            e[p + ".conv1.weight"] = (size_t)c * (b == 0 ? cin : c) * 9;
            bn(p + ".bn1", c);
            e[p + ".conv2.weight"] = (size_t)c * c * 9;
            bn(p + ".bn2", c);
            if (b == 0 && cin != c) { e[p + ".downsample.0.weight"] = (size_t)c * cin; bn(p + ".downsample.1", c); }
        }
        cin = c;
    }
    lin(r + "fc", 256, 512);
}

// dm_model.py:29-56 + diffuser_helpers.py:451-462, same op order in fp32
void build_schedule(cld_handle h) {
    const int n = h->cfg.n_timesteps;
    std::vector<double> ac(n + 1);
    const double s = 0.008;
    const int steps = n + 1;
    for (int i = 0; i <= n; ++i) {
        const double x = (double)steps * (double)i / (double)(steps - 1);      // np.linspace(0, steps, steps)
        const double c = std::cos(((x / steps) + s) / (1 + s) * M_PI * 0.5);
        ac[i] = c * c;
    }
    const double a0 = ac[0];
    for (auto& v : ac) v /= a0;
    std::vector<float> betas(n), alphas(n), acp(n), acp_prev(n);
    for (int i = 0; i < n; ++i) {
        double b = 1.0 - ac[i + 1] / ac[i];
        b = b < 0 ? 0 : (b > 0.999 ? 0.999 : b);
        betas[i] = (float)b;
        alphas[i] = 1.0f - betas[i];
    }
    // torch.cumprod on a CPU float tensor accumulates in DOUBLE and rounds every element to float (at::acc_type<float, false>):
    // a sequential float product differs from the reference's buffer by up to 3 ulp at n = 100, which the cancellation in
    // alphas - alphas_cumprod * alphas (noise_cof) then amplifies
    double run = 1.0;
    for (int i = 0; i < n; ++i) { run *= (double)alphas[i]; acp[i] = (float)run; acp_prev[i] = i ? acp[i - 1] : 1.0f; }
    h->x_t_cof.resize(n); h->noise_cof.resize(n); h->plvc.resize(n); h->sqrt_acp.resize(n); h->sqrt_1m_acp.resize(n);
    h->sqrt_recip_acp.resize(n); h->sqrt_recipm1_acp.resize(n);
    for (int i = 0; i < n; ++i) {
        h->sqrt_acp[i] = std::sqrt(acp[i]);                 // dm_model.py:36-37
        h->sqrt_1m_acp[i] = std::sqrt(1.0f - acp[i]);
        h->sqrt_recip_acp[i] = std::sqrt(1.0f / acp[i]);             // dm_model.py:40-41 (registered there, read by upstream's predict_start_from_noise)
        h->sqrt_recipm1_acp[i] = std::sqrt(1.0f / acp[i] - 1.0f);
        const float pv = betas[i] * (1.0f - acp_prev[i]) / (1.0f - acp[i]);
        h->plvc[i] = (float)std::log((double)(pv < 1e-20f ? 1e-20f : pv));      // correctly rounded: equals torch.log on every entry of the n = 10 / 50 / 100 tables
        h->x_t_cof[i] = std::sqrt(1.0f / alphas[i]);
        h->noise_cof[i] = betas[i] / std::sqrt(alphas[i] - acp[i] * alphas[i]);
    }
}

double mish_d(double x) { return x * std::tanh(std::log1p(std::exp(x))); }

// MFMA-fragment weight packing.  Slab (16-channel group kgg, tap t) holds, for every 16-column N tile nt
// and lane, the 4 values W[co = 16 nt + (lane & 15)][ci = 16 kgg + 4 (lane >> 4) + s][tap t]; the layout
// does not depend on the K-chunk size, so every tiling of a layer reads the same buffer.
template <class F>
std::vector<float> pack_conv_weights(F&& wget, int c_out, int cin_virtual, int ntaps) {
    const int ngrp = cin_virtual / 16, ntn = c_out / 16;
    std::vector<float> out((size_t)ngrp * ntaps * ntn * 256);
    size_t o = 0;
    for (int kgg = 0; kgg < ngrp; ++kgg)
        for (int t = 0; t < ntaps; ++t)
            for (int nt = 0; nt < ntn; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s)
                        out[o++] = wget(16 * nt + (lane & 15), 16 * kgg + 4 * (lane >> 4) + s, t);
    return out;
}

// The latent's convolution (4 -> 64 channels, k = 5): K folded over (tap, channel).  Per N tile and lane (n, kk) eight floats:
// [0..3] = W[co][channel s][tap kk], s = 0..3 (the B values of the four MFMAs that walk taps 0..3) and [4] = W[co][channel kk][tap 4].
template <class F>
std::vector<float> pack_latent_conv_weights(F&& wget, int c_out) {
    const int ntn = c_out / 16;
    std::vector<float> out((size_t)ntn * 64 * 8, 0.f);
    for (int nt = 0; nt < ntn; ++nt)
        for (int lane = 0; lane < 64; ++lane) {
            float* o = &out[((size_t)nt * 64 + lane) * 8];
            const int co = 16 * nt + (lane & 15), kk = lane >> 4;
            for (int s = 0; s < 4; ++s) o[s] = wget(co, s, kk);
            o[4] = wget(co, kk, 4);
        }
    return out;
}

// Split-precision packing (CLD_PRECISION_F16X2).  Weights are scaled by kWScale (a power of two that lifts the lo parts
// out of the fp16 subnormal range; undone exactly in the epilogue) and split into hi = fp16(w), lo = fp16(w - hi).
// Slab (32-channel chunk c, tap t, N tile nt) = [hi: 64 lanes x 8 fp16][lo: 64 lanes x 8 fp16]; lane l holds
// W[co = 16 nt + (l & 15)][ci = 32 c + 8 (l >> 4) + j][tap t], j = 0..7 -- the B operand of v_mfma_f32_16x16x32_f16.
constexpr float kWScale = 64.0f;
template <class F>
std::vector<float> pack_conv_weights_split(F&& wget, int c_out, int cin_virtual, int ntaps) {
    const int nchunk = cin_virtual / 32, ntn = c_out / 16;
    std::vector<_Float16> out((size_t)nchunk * ntaps * ntn * 1024);
    size_t o = 0;
    for (int c = 0; c < nchunk; ++c)
        for (int t = 0; t < ntaps; ++t)
            for (int nt = 0; nt < ntn; ++nt) {
                for (int plane = 0; plane < 2; ++plane)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const float w = wget(16 * nt + (lane & 15), 32 * c + 8 * (lane >> 4) + j, t) * kWScale;
                            const _Float16 hi = (_Float16)w;
                            out[o++] = plane == 0 ? hi : (_Float16)(w - (float)hi);
                        }
            }
    std::vector<float> bytes(out.size() / 2);
    std::memcpy(bytes.data(), out.data(), out.size() * sizeof(_Float16));
    return bytes;
}

// Tiling policy: the MFMA/LDS/global-load mix of the conv loop reaches ~83 % of the fp32-MFMA rate with one
// wave per SIMD and ~89 % with two (scripts/ubench/mfma_issue.hip), so take the 64-column tile when it still
// puts two waves on every SIMD of the chip (256 CUs x 4 SIMDs x 2 = 2,048 waves) and the 32-column tile with a
// 2-way K split (twice the workgroups, two per CU) below that.  Measured at B = 1,024: B 813k vs A 716k
// step.agent/s; an 8-wave variant (32 columns, 4-way K split) was slower than B and is not built; the 8-wave
// variant C (64 columns, 2-way K split) is taken for the widest layers where it fills the chip (below).
inline bool c1c2_mult_of_64(const ConvLayer& l) { return l.c1_pad % 64 == 0 && l.c2 % 64 == 0; }     // 64-channel chunks never straddle the two sources

bool pick_tiling(const ConvLayer& l, int b_pad, ConvGeom* g, int roles = 1 /* layers sharing the launch (conv_pair_kernel: 2) */) {
    *g = l.g;
    const long waves_a = (long)(b_pad / (MT / l.g.lm)) * (l.c_out / 64) * 4;
    auto set = [&](int kc, int nwn, int ks) { g->kc = kc; g->nwn = nwn; g->ks = ks; return true; };
    if (l.g.ain == 1)      // split-precision loop: 64-column tiling only (no K split); 64-channel chunks where the images fit
        return l.has_a ? set(l.g.stride == 2 ? 32 : 64, 4, 1) : false;
#ifdef CLD_EXPERIMENTS      // -DCLD_EXPERIMENTS builds only (A/B scripts under scripts/): the shipped library reads no environment variable
    static const char* force = getenv("CLD_TILING");            // A / B / C
#else
    constexpr const char* force = nullptr;
#endif
    if (force && force[0] == 'C' && l.c_out == 256 && l.g.ntaps == 5 && l.g.stride == 1) return set(32, 4, 2);   // 8 waves: 64 columns x 2-way K split
    if (force && force[0] == 'A' && l.has_a) return set(32, 4, 1);
    if (force && force[0] == 'B' && l.has_b) return set(32, 2, 2);
    const bool skip_a = force && force[0] == 'b';      // experiments: 32-column tiles (with CLD_TILING_HALF's height) at every batch size
    if (l.has_a && ((waves_a >= 2048 && !skip_a) || !l.has_b)) return set(32, 4, 1);
    // tiling C for the 256-channel k5 blocks between 1,024 and 2,047 agents: 8-wave workgroups (64 columns x 2-way K split),
    // one per CU -- the same two waves per SIMD as B with half the A-image staging per MFMA (+1.4 % end to end at B = 1,024).
    // One C workgroup is two B workgroups' work on a CU, so in the units of the tile-height model below it costs
    // ceil(wgs_c / 256) * 2 * 13 / 0.92 / 1.014; it competes with the B heights on that cost (B = 1,536: 384 C workgroups are
    // 1.5 rounds of the chip, 768 B workgroups exactly 3 half-rounds).
    double cost_c = 1e30;
    {
        ConvGeom c = l.g; c.kc = 32; c.nwn = 4; c.ks = 2;
        const long wgs_c = (long)(b_pad / (MT / l.g.lm)) * (l.c_out / 64);
#ifdef CLD_EXPERIMENTS
        static const char* tc = getenv("CLD_TILING_C");       // "0" = never, "all" = every layer shape that has an instance
#else
        constexpr const char* tc = nullptr;
#endif
        const bool widest_only = !(tc && tc[0] == 'a');
        if ((!force || skip_a) && !(tc && tc[0] == '0') && l.g.ain == 0 && wgs_c >= 256 && (!widest_only || l.c_out == 256) && conv_geom_supported(c)) {
            cost_c = (double)((wgs_c + 255) / 256) * 2 * 13 / 0.92 / 1.014;
            if (!l.has_b) return set(32, 4, 2);     // (64-channel chunks with this tiling: 918k vs 949k step.agent/s, not built)
        }
    }
    if (l.has_b) {
        set(32, 2, 2);
        // Tile height (HM = 0 / 1 / 2: 208 / 104 / 52 GEMM rows = 13 / 7 / 4 M-tiles per wave): shorter tiles multiply the
        // workgroups and shorten each one's serial MFMA chain, at the price of a partly empty last M-tile.  Cost model (fits
        // the measured sweep B = 8 .. 2,048): a CU holds two workgroups at a time, so a launch takes ceil(n / 256) rounds of
        // NMT M-tiles, at ~0.85 of the MFMA rate when a CU only ever sees one wave per SIMD and ~0.92 otherwise; near-ties go
        // to the taller tile from 1,024 agents (narrow layers there: 945k vs 935k step.agent/s) and to the shorter one below
        // (B = 512: 817k vs 795k).  Sample latency 74 -> 31 ms at B <= 128;
        // throughput x2.4 at B = 64 (208k), x1.9 at 256 (650k), x1.2 at 512 (817k), x1.14 at 768.
#ifdef CLD_EXPERIMENTS
        static const char* th = getenv("CLD_TILING_HALF");        // "0" full tiles only, "1" / "2" force that height
#else
        constexpr const char* th = nullptr;
#endif
        const long wgs_b = (long)(b_pad / (MT / l.g.lm)) * (l.c_out / 32);
        int best = 0;
        double best_cost = 1e30;
        for (int hm = 0; hm <= 3; ++hm) {
            ConvGeom tg = *g; tg.half = hm;
            if (!conv_geom_supported(tg)) continue;
            if (th && th[0] != '0' + hm && (th[0] == '0' || th[0] == '1' || th[0] == '2' || th[0] == '3')) continue;
            const long n = wgs_b << hm, rounds = (n + 255) / 256;
            const int nmt = ((208 >> hm) + 15) / 16;
            double cost = (double)rounds * nmt / (rounds == 1 ? 0.85 : 0.92);
            // the full-height stride-2 tiles stage twice the rows (13 pieces per thread in flight) and still spill: measured
            // per layer at B = 1,024 (scripts/tile_height_ab.sh) 17.7 / 14.7 / 16.1 us (26 -> 13) and 11.6 / 9.9 / 10.2 us (52 -> 26)
            if (l.g.stride == 2 && hm == 0) cost *= 1.15;
            if (cost < best_cost * (b_pad >= 1024 ? 0.97 : 1.0)) { best_cost = cost; best = hm; }      // near-ties: taller tile from 1,024 agents
        }
        if (cost_c <= best_cost) return set(32, 4, 2);
        g->half = best;
        if (best == 3 && (wgs_b << 3) * roles <= 256 && c1c2_mult_of_64(l)) {     // tiling D: at most one workgroup per CU -> give it a second wave per SIMD
            ConvGeom d = *g; d.kc = 64; d.nwn = 2; d.ks = 4;
            if (conv_geom_supported(d)) return set(64, 2, 4);
        }
        return true;
    }
    return false;
}

struct Ws {
    float *xw, *xtmp, *meanb, *cb, *buf[NBUF], *guide;
    float *gcur, *adam_m, *adam_v;      // multi-step guidance: current iterate and Adam moments [b_pad,52,4] each
    float *col_traj, *col_grad;         // collision term: decoded plans and d total / d plans [b_pad,52,6] each
    float* col_act;                     // ... and the decoder's scaled actions [b_pad,52,2] of the forward launch of a split guidance step
};
size_t ws_floats(int b_pad) {
    return (size_t)b_pad * (3 * T * D + NCB + (size_t)NBUF * ACT) + guide_scratch_floats(b_pad) + (size_t)b_pad * (3 * T * D + 2 * T * 6 + T * 2);
}
Ws carve(void* ws, int b_pad) {
    Ws w;
    float* p = static_cast<float*>(ws);
    w.xw = p; p += (size_t)b_pad * T * D;
    w.xtmp = p; p += (size_t)b_pad * T * D;
    w.meanb = p; p += (size_t)b_pad * T * D;
    w.cb = p; p += (size_t)b_pad * NCB;
    for (int i = 0; i < NBUF; ++i) { w.buf[i] = p; p += (size_t)b_pad * ACT; }
    w.guide = p; p += guide_scratch_floats(b_pad);      // activations kept by the guidance kernel's LSTM forward
    w.gcur = p; p += (size_t)b_pad * T * D;
    w.adam_m = p; p += (size_t)b_pad * T * D;
    w.adam_v = p; p += (size_t)b_pad * T * D;
    w.col_traj = p; p += (size_t)b_pad * T * 6;
    w.col_grad = p; p += (size_t)b_pad * T * 6;
    w.col_act = p;
    return w;
}
inline int pad16(int b) { return (b + 15) / 16 * 16; }

ConvArgs make_args(cld_handle h, const ConvLayer& l, const float* x1, const float* x2, float* y, const float* res,
                   const float* cb, const float* tb_row) {
    ConvArgs a{};
    a.x1 = x1; a.x2 = x2; a.c1_real = l.c1_real; a.c1_pad = l.c1_pad; a.c2 = l.c2;
    a.wfrag = l.wfrag; a.bias = l.bias; a.gamma = l.gamma; a.beta = l.beta;
    if (l.cb_off >= 0) { a.cbias = cb + l.cb_off; a.cb_stride = NCB; a.tbias = tb_row ? tb_row + l.cb_off : nullptr; }
    a.res = res; a.y = y; a.c_out = l.c_out; a.ly = l.ly; a.off0 = l.off0; a.orow0 = l.orow0;
    a.wscale_inv = l.g.ain == 1 ? 1.0f / kWScale : 1.0f;
    a.stamps = nullptr;
    (void)h;
    return a;
}

// One conv launch; launches of the dominant kernel shape (k5 + GroupNorm + Mish block producing 256 channels at L = 13) are
// bracketed by HIP events in every kProfStride-th U-Net evaluation while profiling is on: an event pair costs ~2 us of stream
// time, and bracketing all 800 such launches of a 100-step sample call slowed the timed region itself by 5 %.
constexpr int kWino1dMinRows = 384;       // launch sets of at least this many rows take the Winograd form of the k5 layers at L = 13 / 26 (100-step sample,
                                          // direct / Winograd: 38.7 / 48.0 ms at 256 rows, 60.1 / 51.8 at 384, 60.7 / 53.3 at 512, 89.5 / 70.8 at 768, 102.3 / 75.1 at
                                          // 1,024, 189.8 / 134.3 at 2,048; launches of fewer than 512 whole items run as half items, wino1d_kernels.hip; below 384
                                          // rows a launch is a few workgroups' serial MFMA chain and the direct form's small tiles spread it wider)
// the rule itself, a function of the layer shape, the rows of the launch set and what a test forced (cld_debug_conv5_form exposes it)
bool conv5_takes_winograd(int l_in, int c1, int c2, int c_out, long b_pad, int forced) {
    if (!wino1d_supported(l_in, c1, c2, c_out) || forced == CLD_FORM_DIRECT) return false;
    const long widest = c1 > c_out ? c1 : c_out;      // the Winograd kernel addresses its tensors with 32-bit byte offsets (per source tensor)
    if (b_pad * l_in * widest * 4 >= (1L << 31)) return false;
    return forced == CLD_FORM_WINOGRAD || forced == CLD_FORM_WINOGRAD_WHOLE || forced == CLD_FORM_WINOGRAD_KSPLIT || b_pad >= kWino1dMinRows;
}
// what a test forced about the items of the Winograd launches: 0 = by size, 1 = whole items, 2 = whole items of eight waves
int wino_item_form(cld_handle h) {
    const int f = h->force_kernel[CLD_KERNEL_CONV5];
    return f == CLD_FORM_WINOGRAD_WHOLE ? 1 : f == CLD_FORM_WINOGRAD_KSPLIT ? 2 : 0;
}
bool use_wino1d(cld_handle h, const ConvLayer& l, int b_pad) {
    if (!l.ufrag || !l.wino_launch) return false;
    return conv5_takes_winograd(l.g.l_in, l.c1_real, l.c2, l.c_out, b_pad, h->force_kernel[CLD_KERNEL_CONV5]);
}
// FLOP a conv launch stands for (2 x output rows x K x N over the real channels) and FLOP its MFMAs execute in the form
// the launch takes: the direct form pads the 4-channel latent's K to (tap, channel) = 20; the Winograd form runs 8 GEMMs over
// wino1d_gemm_rows() rows (64 per item, idle rows of the L = 26 items included) x C_in x C_out
void count_flop(cld_handle h, const ConvLayer& l, int b_pad, double* alg, double* exec) {
    const double cin = (double)(l.c1_real + l.c2);
    *alg = 2.0 * (double)b_pad * l.g.lm * (l.g.ntaps * cin) * l.c_out;
    if (use_wino1d(h, l, b_pad)) *exec = 2.0 * (double)wino1d_row_planes(l.g.l_in, l.c_out, b_pad, wino_item_form(h)) * cin * l.c_out;
    else *exec = 2.0 * (double)b_pad * l.g.lm * (l.g.padc ? 20.0 : (double)(l.g.ntaps * (l.c1_pad + l.c2))) * l.c_out;
}
void count_launch(cld_handle h, const ConvLayer& l, int b_pad) {
    double alg, exec;
    count_flop(h, l, b_pad, &alg, &exec);
    h->eval_alg_flop += alg;
    h->eval_exec_flop += exec;
    h->eval_launches++;
}
hipError_t launch_one(cld_handle h, const ConvLayer& l, const ConvGeom& g, const ConvArgs& a, int b_pad, hipStream_t s) {
    count_launch(h, l, b_pad);
    if (use_wino1d(h, l, b_pad)) {
        ConvArgs w = a;
        w.wfrag = l.ufrag;
        w.wfrag_edge = l.ufrag_edge;
        return launch_wino1d(w, l.g.l_in, b_pad, wino_item_form(h), s);
    }
    return launch_conv(g, a, b_pad, s);
}
hipError_t launch_maybe_timed(cld_handle h, const ConvLayer& l, const ConvGeom& g, const ConvArgs& a, int b_pad, hipStream_t s) {
    const bool timed = h->prof_on && g.l_in == 13 && g.ntaps == 5 && g.epi == EPI_GN_MISH && l.c_out == 256 && l.c1_real == 256 &&
                       (h->eval_counter % kProfStride) == 0;
    if (!timed) return launch_one(h, l, g, a, b_pad, s);
    if (h->prof_used + 2 > h->prof_ev.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t ev;
            hipError_t e = hipEventCreate(&ev);
            if (e != hipSuccess) return e;
            h->prof_ev.push_back(ev);
        }
    }
    hipError_t e = hipEventRecord(h->prof_ev[h->prof_used], s);
    if (e != hipSuccess) return e;
    e = launch_one(h, l, g, a, b_pad, s);
    if (e != hipSuccess) return e;
    e = hipEventRecord(h->prof_ev[h->prof_used + 1], s);
    h->prof_used += 2;
    double alg, exec;
    count_flop(h, l, b_pad, &alg, &exec);
    h->prof_flop += alg;
    h->prof_exec_flop += exec;
    return e;
}

// two independent layers with one grid shape -> one launch (conv_pair_kernel); falls back to two launches
hipError_t run_pair(cld_handle h, const ConvLayer& la, const ConvArgs& aa, const ConvLayer& lb, const ConvArgs& ab,
                    int b_pad, hipStream_t s) {
    ConvGeom ga, gb;
    if (!pick_tiling(la, b_pad, &ga, 2) || !pick_tiling(lb, b_pad, &gb, 2)) return hipErrorInvalidValue;
#ifdef CLD_EXPERIMENTS      // diagnostics (tests/tools/debug_split.py): stop a U-Net evaluation after N launches
    static const int stop_after = getenv("CLD_DEBUG_STOP") ? atoi(getenv("CLD_DEBUG_STOP")) : 1 << 30;
    if (h->launch_counter >= stop_after) return hipSuccess;
#endif
    h->launch_counter += 2;
    if (use_wino1d(h, la, b_pad) || use_wino1d(h, lb, b_pad)) {      // the k5 conv in its Winograd form; the 1x1 projection beside it stays a direct launch
        hipError_t e = launch_maybe_timed(h, la, ga, aa, b_pad, s);
        return e != hipSuccess ? e : launch_maybe_timed(h, lb, gb, ab, b_pad, s);
    }
    if (ga.nwn == gb.nwn && ga.ks == gb.ks && conv_pair_supported(ga, gb)) {
        count_launch(h, la, b_pad);
        count_launch(h, lb, b_pad);
        h->eval_launches--;          // one launch for the two layers
        return launch_conv_pair(ga, aa, gb, ab, b_pad, s);
    }
    hipError_t e = launch_maybe_timed(h, la, ga, aa, b_pad, s);
    return e != hipSuccess ? e : launch_maybe_timed(h, lb, gb, ab, b_pad, s);
}

hipError_t run_args(cld_handle h, const ConvLayer& l, ConvArgs a, int b_pad, hipStream_t s);
hipError_t run_conv(cld_handle h, const ConvLayer& l, const float* x1, const float* x2, float* y, const float* res,
                    const float* cb, const float* tb_row, int b_pad, hipStream_t s) {
    return run_args(h, l, make_args(h, l, x1, x2, y, res, cb, tb_row), b_pad, s);
}
hipError_t run_args(cld_handle h, const ConvLayer& l, ConvArgs a, int b_pad, hipStream_t s) {
#ifdef CLD_EXPERIMENTS
    static const int stop_after = getenv("CLD_DEBUG_STOP") ? atoi(getenv("CLD_DEBUG_STOP")) : 1 << 30;
    if (h->launch_counter >= stop_after) return hipSuccess;
#endif
    a.stamps = (h->stamp_buf && h->launch_counter == h->stamp_layer) ? h->stamp_buf : nullptr;
    h->launch_counter++;
    ConvGeom g;
    if (!pick_tiling(l, b_pad, &g)) return hipErrorInvalidValue;
    return launch_maybe_timed(h, l, g, a, b_pad, s);
}

// The 64-channel levels as LDS-resident layer chains (conv_chain.hip): exact-fp32 handles only (the split-precision mode keeps
// one launch per layer).
bool use_chains(cld_handle h, int b_pad) {
    if (h->precision != CLD_PRECISION_F32) return false;
    const int f = h->force_kernel[CLD_KERNEL_UNET];
    if (f == CLD_FORM_LAYERS) return false;
    if (f == CLD_FORM_CHAIN || f == CLD_FORM_CHAIN_TILE1 || f == CLD_FORM_CHAIN_TILE4 || f == CLD_FORM_CHAIN_WINO || f == CLD_FORM_CHAIN_WINO2 || f == CLD_FORM_CHAIN_WINO1) return true;
    (void)b_pad;
    return true;              // measured faster than one launch per layer at every batch size (profiles/r03/chain_check.txt)
}
// agents per chain workgroup: 4 (13 M-tiles per wave at L = 52: the throughput tile) or, for the small batches whose time is
// the LENGTH of the dependent launch sequence, 1 (a quarter of the serial MFMA chain per stage, four times the workgroups)
int chain_tile(cld_handle h, int b_pad) {
    const int f = h->force_kernel[CLD_KERNEL_UNET];
    if (f == CLD_FORM_CHAIN_TILE1) return 1;
    if (f == CLD_FORM_CHAIN_TILE4 || f == CLD_FORM_CHAIN_WINO) return 4;
    return b_pad <= kChainSmall ? 1 : 4;
}
// the chains run their 64 -> 64 k5 layers in Winograd F(4, 5) form (chain_wino.hip) unless a test forces the direct form of the k5 layers
// (CLD_KERNEL_CONV5) or of the chains (CLD_FORM_CHAIN_TILE1 / CLD_FORM_CHAIN_TILE4)
// agents per Winograd chain tile (chain_wino.hip): 4 (two workgroups per CU), 2 or 1 (three per CU)
// (per U-Net evaluation, one / two / four agents: 240 / 262 / 310 us at 64 rows, 536 / 539 / 586 at 512, 700 / 709 / 729 at 768, 768 / 749 / 767 at 1,024,
//  1,100 / 1,080 / 1,119 at 1,536, 1,344 / 1,314 / 1,314 at 2,048, 2,571 / 2,502 / 2,510 at 4,096; the direct one-agent tiles: 246 / 550 / 732 / 810)
int chain_wino_tile(cld_handle h, int b_pad) {
    const int f = h->force_kernel[CLD_KERNEL_UNET];
    if (f == CLD_FORM_CHAIN_WINO) return 4;
    if (f == CLD_FORM_CHAIN_WINO2) return 2;
    if (f == CLD_FORM_CHAIN_WINO1) return 1;
    return b_pad <= kChainSmall ? 1 : 2;
}
bool chain_wino(cld_handle h, int b_pad) {
    const int f = h->force_kernel[CLD_KERNEL_UNET];
    if (f == CLD_FORM_CHAIN_WINO || f == CLD_FORM_CHAIN_WINO2 || f == CLD_FORM_CHAIN_WINO1) return true;
    if (f == CLD_FORM_CHAIN_TILE1 || f == CLD_FORM_CHAIN_TILE4) return false;
    return h->force_kernel[CLD_KERNEL_CONV5] != CLD_FORM_DIRECT;
}

// One U-Net evaluation (temporal.py:122-180) on the padded latent `x` [b_pad,52,4]; leaves the
// final_conv.0 activations [b_pad,52,64] in w.buf[7].
hipError_t run_unet(cld_handle h, const Ws& w, const float* x, int t_idx, int b_pad, hipStream_t s) {
    const float* tbr = t_idx >= 0 ? h->tb + (size_t)t_idx * NCB : nullptr;     // < 0: per-agent timesteps, time bias folded into w.cb
    float* const* b = w.buf;
    h->launch_counter = 0;
    h->eval_counter++;
    h->eval_alg_flop = h->eval_exec_flop = 0.0;
    h->eval_launches = 0;
    set_lds_floor(h->lds_floor);
    hipError_t e;
#define RC(...) do { e = run_conv(h, __VA_ARGS__, w.cb, tbr, b_pad, s); if (e != hipSuccess) return e; } while (0)
    auto resblock = [&](const ResBlock& rb, const float* in1, const float* in2, float* out) -> hipError_t {
        const float* r = in1;           // identity residual reads the block input
        if (rb.has_res) {               // first conv and 1x1 projection both read only the block input: one launch
            e = run_pair(h, rb.c0, make_args(h, rb.c0, in1, in2, b[1], nullptr, w.cb, tbr),
                         rb.res, make_args(h, rb.res, in1, in2, b[0], nullptr, w.cb, tbr), b_pad, s);
            if (e != hipSuccess) return e;
            r = b[0];
        } else {
            RC(rb.c0, in1, in2, b[1], nullptr);
        }
        RC(rb.c1, b[1], nullptr, out, r);
        return hipSuccess;
    };
#define RB(...) do { e = resblock(__VA_ARGS__); if (e != hipSuccess) return e; } while (0)
    if (use_chains(h, b_pad)) {
        // downs.0 (five layers at 64 channels x 52 rows) as one launch, the tile resident in LDS (conv_chain.hip)
        ChainHeadArgs ca{};
        ca.x = x;
        auto stage = [&](const ConvLayer& l, int res_kind, int keep) {
            ChainStage st{};
            st.wfrag = l.wfrag; st.ufrag = l.ufrag; st.bias = l.bias; st.gamma = l.gamma; st.beta = l.beta; st.cb_off = l.cb_off;
            st.res_kind = res_kind; st.keep = keep;
            return st;
        };
        ca.st[0] = stage(h->blocks[0].c0, CHAIN_RES_NONE, 0);
        ca.st[1] = stage(h->blocks[0].c1, CHAIN_RES_LATENT, 1);
        ca.st[2] = stage(h->blocks[1].c0, CHAIN_RES_NONE, 0);
        ca.st[3] = stage(h->blocks[1].c1, CHAIN_RES_KEPT, 0);
        ca.st[4] = stage(h->down[0], CHAIN_RES_NONE, 0);
        ca.res4_w = h->res4_w; ca.res4_b = h->res4_b;
        ca.cbias = w.cb; ca.cb_stride = NCB; ca.tbias = tbr;
        // spill slots of the chain: b[2] -- and, with the one-agent tiles (4 M-tiles of 16 rows for 52: 4,096 floats per agent instead of
        // 3,328), the first 768 floats per agent of b[3], which follows it in the workspace; both are dead until the up path writes them
        ca.keep = b[2]; ca.y = b[6];
        ca.stamps = (h->stamp_buf && h->stamp_layer == 0) ? h->stamp_buf : nullptr;
        h->launch_counter += 5;
        h->eval_alg_flop += 2.0 * b_pad * 52.0 * 64 * (20 + 3 * 320) + 2.0 * b_pad * 26.0 * 64 * 192 + 2.0 * b_pad * 52.0 * 64 * 4;
        const bool cw = chain_wino(h, b_pad);
        h->eval_exec_flop += cw ? chain_head_wino_exec_flop(b_pad, chain_wino_tile(h, b_pad)) : chain_head_exec_flop(b_pad, chain_tile(h, b_pad));
        h->eval_launches++;
        e = cw ? launch_chain_head_wino(ca, b_pad, chain_wino_tile(h, b_pad), s) : launch_chain_head(ca, b_pad, chain_tile(h, b_pad), s);
        if (e != hipSuccess) return e;
    } else {
    {   // block 0: conv(4 -> 64) | conv(64 -> 64) + residual_conv(x), the 1x1 projection of the latent evaluated in the epilogue
        RC(h->blocks[0].c0, x, nullptr, b[1], nullptr);
        ConvArgs a1 = make_args(h, h->blocks[0].c1, b[1], nullptr, b[2], nullptr, w.cb, tbr);
        a1.res4_x = x; a1.res4_w = h->res4_w; a1.res4_b = h->res4_b;
        e = run_args(h, h->blocks[0].c1, a1, b_pad, s);
        if (e != hipSuccess) return e;
    }
    RB(h->blocks[1], b[2], nullptr, b[3]);
    RC(h->down[0], b[3], nullptr, b[6], nullptr);
    }
    RB(h->blocks[2], b[6], nullptr, b[2]);
    RB(h->blocks[3], b[2], nullptr, b[4]);            // skip 128@26
    RC(h->down[1], b[4], nullptr, b[6], nullptr);
    RB(h->blocks[4], b[6], nullptr, b[2]);
    RB(h->blocks[5], b[2], nullptr, b[5]);            // skip 256@13
    RB(h->blocks[6], b[5], nullptr, b[2]);
    RB(h->blocks[7], b[2], nullptr, b[3]);
    RB(h->blocks[8], b[3], b[5], b[2]);               // cat(x, skip) 512@13 -> 128@13
    RB(h->blocks[9], b[2], nullptr, b[6]);
    e = run_pair(h, h->upT[0][0], make_args(h, h->upT[0][0], b[6], nullptr, b[3], nullptr, w.cb, tbr),
                 h->upT[0][1], make_args(h, h->upT[0][1], b[6], nullptr, b[3], nullptr, w.cb, tbr), b_pad, s);   // 128@26
    if (e != hipSuccess) return e;
    h->eps_in_buf7 = false;
    h->upd_fused = false;
    if (use_chains(h, b_pad)) {
        // ups.1.0's first conv + residual projection as one pair launch (K = 256 x 5 from HBM), then everything behind it --
        // ups.1.0's second conv, ups.1.1, the transposed conv, final_conv.0 and final_conv.1 -- as one launch (conv_chain.hip)
        const ResBlock& rb = h->blocks[10];
        e = run_pair(h, rb.c0, make_args(h, rb.c0, b[3], b[4], b[1], nullptr, w.cb, tbr),
                     rb.res, make_args(h, rb.res, b[3], b[4], b[0], nullptr, w.cb, tbr), b_pad, s);
        if (e != hipSuccess) return e;
        auto stage = [&](const ConvLayer& l, int res_kind, int keep, const float* res) {
            ChainStage st{};
            st.wfrag = l.wfrag; st.ufrag = l.ufrag; st.bias = l.bias; st.gamma = l.gamma; st.beta = l.beta; st.cb_off = l.cb_off;
            st.res_kind = res_kind; st.keep = keep; st.res = res;
            return st;
        };
        ChainTailArgs ct{};
        ct.x = b[1];
        ct.st[0] = stage(rb.c1, CHAIN_RES_TENSOR, 1, b[0]);
        ct.st[1] = stage(h->blocks[11].c0, CHAIN_RES_NONE, 0, nullptr);
        ct.st[2] = stage(h->blocks[11].c1, CHAIN_RES_KEPT, 0, nullptr);
        ct.up_even = stage(h->upT[1][0], CHAIN_RES_NONE, 0, nullptr);
        ct.up_odd = stage(h->upT[1][1], CHAIN_RES_NONE, 0, nullptr);
        ct.fin = stage(h->final_cb, CHAIN_RES_NONE, 0, nullptr);
        ct.head_wfrag = h->head_wfrag; ct.head_b = h->head_b;
        ct.cbias = w.cb; ct.cb_stride = NCB; ct.tbias = tbr;
        ct.keep = b[2]; ct.eps = b[7];
        h->upd_fused = false;
        if (const HeadArgs* u = h->fuse_upd) {           // the step's DDPM update in the same launch
            ct.eps = nullptr;
            ct.upd_x = u->x; ct.upd_z = u->z; ct.upd_mean_out = u->mean_out; ct.upd_x_out = u->x_out;
            ct.xc = u->xc; ct.nc = u->nc; ct.sg = u->sg; ct.B = u->B; ct.seed = u->seed; ct.step_salt = u->step_salt;
            h->upd_fused = true;
        }
        h->launch_counter += 6;
        h->eval_alg_flop += 2.0 * b_pad * 26.0 * 64 * (3 * 320) + 2.0 * b_pad * 52.0 * 64 * 128 + 2.0 * b_pad * 52.0 * 64 * 320 + 2.0 * b_pad * 52.0 * 4 * 64;
        const bool cw = chain_wino(h, b_pad);
        h->eval_exec_flop += cw ? chain_tail_wino_exec_flop(b_pad, chain_wino_tile(h, b_pad)) : chain_tail_exec_flop(b_pad, chain_tile(h, b_pad));
        h->eval_launches++;
        e = cw ? launch_chain_tail_wino(ct, b_pad, chain_wino_tile(h, b_pad), s) : launch_chain_tail(ct, b_pad, chain_tile(h, b_pad), s);
        if (e != hipSuccess) return e;
        h->eps_in_buf7 = true;
    } else {
    RB(h->blocks[10], b[3], b[4], b[2]);              // cat(x, skip) 256@26 -> 64@26
    RB(h->blocks[11], b[2], nullptr, b[6]);
    e = run_pair(h, h->upT[1][0], make_args(h, h->upT[1][0], b[6], nullptr, b[3], nullptr, w.cb, tbr),
                 h->upT[1][1], make_args(h, h->upT[1][1], b[6], nullptr, b[3], nullptr, w.cb, tbr), b_pad, s);   // 64@52
    if (e != hipSuccess) return e;
    RC(h->final_cb, b[3], nullptr, b[7], nullptr);
    }
#undef RB
#undef RC
    return hipSuccess;
}

// what the head kernel reads after run_unet: final_conv.0's activations, or the noise prediction itself when the tail chain ran
inline void head_source(cld_handle h, const Ws& w, HeadArgs& a) {
    if (h->eps_in_buf7) a.eps_in = w.buf[7];
    else a.f = w.buf[7];
}

// loop iterations of the sampler: len(range(0, n_timesteps, stride)) (dm_model.py:119)
inline int loop_steps(cld_handle h) { return (h->cfg.n_timesteps + h->stride - 1) / h->stride; }

const std::vector<float>* getw(cld_handle h, const std::string& k) {
    auto it = h->w.find(k);
    return it == h->w.end() ? nullptr : &it->second;
}

}  // namespace

// =============================================================================================
extern "C" {

const char* cld_version(void) { return "libcld_hip 0.1.0 gfx950 mfma_f32_16x16x4 / f16x2-split mfma_f32_16x16x32_f16"; }

int cld_get_precision(cld_handle h) { return h ? h->precision : CLD_ERR_ARG; }

void cld_default_config(cld_config* c) {
    if (!c) return;
    c->horizon = 52; c->latent_dim = 4; c->cond_dim = 256; c->base_dim = 32;
    c->dim_mults[0] = 2; c->dim_mults[1] = 4; c->dim_mults[2] = 8;
    c->hidden = 64; c->n_timesteps = 100; c->step_time = 0.1f;
    c->acce_bound[0] = -10.f; c->acce_bound[1] = 8.f;
    c->v_bound[0] = -10.f; c->v_bound[1] = 30.f;
    c->max_steer = 0.5f; c->max_yawvel = 6.283185307179586f;
    const float mean[6] = {13.162f, -0.13891f, 5.0223f, -0.0046415f, -0.0080072f, -0.0013546f};
    const float stdv[6] = {13.0717f, 2.2462f, 3.6187f, 0.2210f, 2.5770f, 0.0840f};
    for (int i = 0; i < 6; ++i) { c->norm_mean[i] = mean[i]; c->norm_std[i] = stdv[i]; }
    c->precision = CLD_PRECISION_F32;
}

int cld_create(const cld_config* cfg, cld_handle* out) {
    if (!cfg || !out) return CLD_ERR_ARG;
    *out = nullptr;
    if (cfg->horizon != 52 || cfg->latent_dim != 4 || cfg->cond_dim != 256 || cfg->base_dim != 32 ||
        cfg->dim_mults[0] != 2 || cfg->dim_mults[1] != 4 || cfg->dim_mults[2] != 8 || cfg->hidden != 64 ||
        cfg->n_timesteps < 1 || cfg->n_timesteps > 4096)
        return CLD_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return CLD_ERR_HIP;
    cld_handle h = new cld_handle_s();
    h->cfg = *cfg;
    h->precision = cfg->precision;
    if (h->precision != CLD_PRECISION_F32 && h->precision != CLD_PRECISION_F16X2) { delete h; return CLD_ERR_ARG; }
    add_expect(h);
    build_schedule(h);
    DynParams& d = h->dyn;
    d.dt = cfg->step_time; d.acc_lo = cfg->acce_bound[0]; d.acc_hi = cfg->acce_bound[1];
    d.v_lo = cfg->v_bound[0]; d.v_hi = cfg->v_bound[1]; d.max_steer = cfg->max_steer; d.max_yawvel = cfg->max_yawvel;
    for (int i = 0; i < 6; ++i) { d.mean[i] = cfg->norm_mean[i]; d.std[i] = cfg->norm_std[i]; }
    *out = h;
    return CLD_OK;
}

int cld_debug_stamps(cld_handle h, void* buf, int32_t layer) {
    if (!h) return CLD_ERR_ARG;
    h->stamp_buf = static_cast<unsigned long long*>(buf);
    h->stamp_layer = layer;
    return CLD_OK;
}

int cld_debug_guide_stamps(void* out_host) {
    if (!out_host) return CLD_ERR_ARG;
    read_guide_stamps(static_cast<unsigned long long*>(out_host));
    return CLD_OK;
}

int cld_set_stride(cld_handle h, int32_t stride) {
    if (!h || stride < 1 || stride > h->cfg.n_timesteps) return fail(h, CLD_ERR_ARG, "cld_set_stride: stride out of range");
    h->stride = stride;
    return CLD_OK;
}

int cld_debug_lds_floor(cld_handle h, size_t bytes) {
    if (!h) return CLD_ERR_ARG;
    h->lds_floor = bytes;
    return CLD_OK;
}

int cld_debug_force_kernel(cld_handle h, int32_t which, int32_t form) {
    if (!h || which < 0 || which > 5 || form < 0 || form > (which == CLD_KERNEL_GUIDE || which == CLD_KERNEL_CONTEXT ? 3 : which == CLD_KERNEL_CONV5 ? 4 : (which == CLD_KERNEL_UNET ? 7 : 2))) return fail(h, CLD_ERR_ARG, "cld_debug_force_kernel: bad argument");
    h->force_kernel[which] = form;
    return CLD_OK;
}

int cld_debug_conv5_form(int32_t l_in, int32_t c1, int32_t c2, int32_t c_out, int64_t rows, int32_t forced_form) {
    if (l_in < 1 || c1 < 1 || c2 < 0 || c_out < 1 || rows < 0 || forced_form < 0 || forced_form > 4) return CLD_ERR_ARG;
    return conv5_takes_winograd(l_in, c1, c2, c_out, (long)((rows + 15) / 16 * 16), forced_form) ? CLD_FORM_WINOGRAD : CLD_FORM_DIRECT;
}

int cld_profile_enable(cld_handle h, int32_t on) {
    if (!h) return CLD_ERR_ARG;
    h->prof_on = on != 0;
    h->prof_used = 0;
    h->eval_counter = -1;
    h->prof_flop = 0.0;
    h->prof_exec_flop = 0.0;
    return CLD_OK;
}

int cld_profile_read(cld_handle h, double* total_ms, int64_t* launches, double* total_flop) {
    if (!h) return CLD_ERR_ARG;
    double ms = 0.0;
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
        HIPCK(h, hipEventSynchronize(h->prof_ev[i + 1]));
        float t = 0.f;
        HIPCK(h, hipEventElapsedTime(&t, h->prof_ev[i], h->prof_ev[i + 1]));
        ms += t;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = (int64_t)(h->prof_used / 2);
    if (total_flop) *total_flop = h->prof_flop;
    return CLD_OK;
}

int cld_profile_read_executed(cld_handle h, double* timed_executed_flop, double* eval_algorithmic_flop, double* eval_executed_flop,
                              int32_t* eval_launches) {
    if (!h) return CLD_ERR_ARG;
    if (timed_executed_flop) *timed_executed_flop = h->prof_exec_flop;
    if (eval_algorithmic_flop) *eval_algorithmic_flop = h->eval_alg_flop;
    if (eval_executed_flop) *eval_executed_flop = h->eval_exec_flop;
    if (eval_launches) *eval_launches = h->eval_launches;
    return CLD_OK;
}

int cld_destroy(cld_handle h) {
    if (!h) return CLD_ERR_ARG;
    for (hipEvent_t ev : h->prof_ev) (void)hipEventDestroy(ev);
    for (void* p : h->dev_allocs) (void)hipFree(p);
    delete h;
    return CLD_OK;
}

const char* cld_last_error(cld_handle h) { return h ? h->err.c_str() : "null handle"; }

int cld_load_weight(cld_handle h, const char* name, const float* data, size_t numel) {
    if (!h || !name || !data) return fail(h, CLD_ERR_ARG, "cld_load_weight: null argument");
    if (h->finalized) return fail(h, CLD_ERR_STATE, "cld_load_weight: handle already finalized");
    std::string k = name;
    for (const char* pre : {"dm.", "vae.", "lstmvae."})
        if (k.rfind(pre, 0) == 0) k = k.substr(std::strlen(pre));
    static const char* sched[] = {"betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                                  "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
                                  "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                                  "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
                                  "x_t_cof", "noise_cof"};
    for (const char* sname : sched)
        if (k == sname) return CLD_OK;     // rebuilt from n_timesteps by cld_create
    if (k.size() > 20 && k.compare(k.size() - 20, 20, ".num_batches_tracked") == 0) return CLD_OK;   // BatchNorm counter: unused in eval
    auto it = h->expect.find(k);
    if (it == h->expect.end()) return fail(h, CLD_ERR_ARG, "cld_load_weight: unknown key '" + k + "'");
    if (it->second != numel)
        return fail(h, CLD_ERR_ARG, "cld_load_weight: '" + k + "' expects " + std::to_string(it->second) +
                                        " values, got " + std::to_string(numel));
    h->w[k].assign(data, data + numel);
    return CLD_OK;
}

int cld_finalize(cld_handle h, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (h->finalized) return fail(h, CLD_ERR_STATE, "cld_finalize: already finalized");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // The U-Net is optional as a whole (a VAE-only or ContextEncoder-only handle is legal: VaeModel without a DmModel), never in part
    size_t unet_have = 0, unet_want = 0;
    std::string unet_missing;
    for (const auto& kv : h->expect)
        if (kv.first.rfind("model.", 0) == 0) {
            ++unet_want;
            if (h->w.count(kv.first)) ++unet_have;
            else if (unet_missing.empty()) unet_missing = kv.first;
        }
    if (unet_have != 0 && unet_have != unet_want) return fail(h, CLD_ERR_STATE, "cld_finalize: missing weight '" + unet_missing + "'");
    h->has_unet = unet_have == unet_want;
    hipError_t e = hipSuccess;
#define UP(dst, vec) do { dst = upload(h, vec, s, &e); if (e != hipSuccess) return fail(h, CLD_ERR_HIP, std::string("upload: ") + hipGetErrorString(e)); } while (0)

    // ---- conv layers -------------------------------------------------------------------
    auto build_unet = [&]() -> int {
    auto make_conv = [&](ConvLayer& l, const std::string& wname, int c_out, int c1_real, int c2, int L_in, int lm,
                         int stride, int ntaps, const int* tapk, bool transposed, int off0, int orow0, int ostr,
                         int ly, int epi, const std::string& gn_name, bool in_f32 = false, bool out_f32 = false) -> int {
        const bool split = h->precision == CLD_PRECISION_F16X2;
        const int ain = (split && !in_f32) ? 1 : 0, aout = (split && !out_f32) ? 1 : 0;
        const std::vector<float>& W = *getw(h, wname + ".weight");
        const int c1_pad = (c1_real + 31) / 32 * 32;     // the 4-channel latent is padded to one 32-channel chunk
        const int cin_real = c1_real + c2;
        const int kw = (int)(W.size() / ((size_t)c_out * cin_real));
        auto wget = [&](int co, int civ, int t) -> float {
            int ci;
            if (civ < c1_pad) { if (civ >= c1_real) return 0.f; ci = civ; }
            else ci = c1_real + (civ - c1_pad);
            const int k = tapk[t];
            return transposed ? W[((size_t)ci * c_out + co) * kw + k]     // ConvTranspose1d [C_in, C_out, k]
                              : W[((size_t)co * cin_real + ci) * kw + k]; // Conv1d [C_out, C_in, k]
        };
        std::vector<float> packed = c1_real < 32 ? pack_latent_conv_weights(wget, c_out)
                                    : ain       ? pack_conv_weights_split(wget, c_out, c1_pad + c2, ntaps)
                                                : pack_conv_weights(wget, c_out, c1_pad + c2, ntaps);
        UP(l.wfrag, packed);
        const bool chain_k5 = (L_in == 52 || L_in == 26) && c1_real == 64 && c2 == 0 && c_out == 64;
        l.wino_launch = wino1d_supported(L_in, c1_real, c2, c_out);
        if (!split && !transposed && stride == 1 && ntaps == 5 && epi == EPI_GN_MISH && (l.wino_launch || chain_k5)) {
            // F(4, 5) at the points {0, +-1, +-2, +-1/2, inf}: U[xi][ci][co] = sum_k G[xi][k] w[co][ci][k], in double
            static const double Gm[8][5] = {{-1, 0, 0, 0, 0},
                                            {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
                                            {-2.0 / 9, 2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},
                                            {1.0 / 90, 1.0 / 45, 2.0 / 45, 4.0 / 45, 8.0 / 45},
                                            {1.0 / 90, -1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45},
                                            {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45},
                                            {32.0 / 45, -16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45},
                                            {0, 0, 0, 0, 1}};
            auto uget = [&](int co, int ci, int xi) -> float {
                double u = 0.0;
                for (int k = 0; k < 5; ++k) u += Gm[xi][k] * (double)W[((size_t)co * cin_real + ci) * kw + k];
                return (float)u;
            };
            std::vector<float> upacked = pack_conv_weights(uget, c_out, cin_real, 8);
            UP(l.ufrag, upacked);
            if (l.wino_launch) {      // wino1d_edge.hip: 12 planes per 16-channel chunk -- xi 0 .. 7, then the raw taps 0 .. 3 of the direct column
                auto eget = [&](int co, int ci, int pl) -> float { return pl < 8 ? uget(co, ci, pl) : W[((size_t)co * cin_real + ci) * kw + (pl - 8)]; };
                std::vector<float> epacked = pack_conv_weights(eget, c_out, cin_real, 12);
                UP(l.ufrag_edge, epacked);
            }
        }
        UP(l.bias, *getw(h, wname + ".bias"));
        if (epi == EPI_GN_MISH) {
            UP(l.gamma, *getw(h, gn_name + ".weight"));
            UP(l.beta, *getw(h, gn_name + ".bias"));
        }
        l.c_out = c_out; l.c1_real = c1_real; l.c1_pad = c1_pad; l.c2 = c2; l.ly = ly; l.off0 = off0; l.orow0 = orow0;
        l.g = ConvGeom{L_in, lm, stride, ntaps, 32, 4, 1, epi, c_out / 8, ostr, c1_real < 32 ? 1 : 0, ain, aout, 0};
        if (c2 > 0 && c2 != c1_real) return fail(h, CLD_ERR_ARG, "cld_finalize: concatenated sources must have equal channel counts");
        ConvGeom t = l.g;
        if (ain == 1 && stride == 1) t.kc = 64;
        l.has_a = conv_geom_supported(t);
        t.nwn = 2; t.ks = 2;
        l.has_b = conv_geom_supported(t);
        if (!l.has_a && !l.has_b)
            return fail(h, CLD_ERR_ARG, "cld_finalize: no kernel instance for layer '" + wname + "'");
        return CLD_OK;
    };
    static const int k5[5] = {0, 1, 2, 3, 4}, k3[3] = {0, 1, 2}, k1[1] = {0};
    static const int kT_even[2] = {3, 1}, kT_odd[2] = {2, 0};   // out[2j] = x[j-1] W3 + x[j] W1 ; out[2j+1] = x[j] W2 + x[j+1] W0
    int cb_off = 0, rc;
    for (int i = 0; i < 12; ++i) {
        const BlockDef& bd = kBlocks[i];
        ResBlock& rb = h->blocks[i];
        const std::string p = bd.name;
        const bool cat = (i == 8 || i == 10);
        const int c1 = cat ? bd.cin / 2 : bd.cin, c2 = cat ? bd.cin / 2 : 0;
        const bool latent_in = (i == 0);     // the 4-channel latent stays fp32 (its range is unbounded): exact-fp32 loop
        if ((rc = make_conv(rb.c0, p + ".blocks.0.block.0", bd.cout, c1, c2, bd.L, bd.L, 1, 5, k5, false, -2, 0, 1, bd.L,
                            EPI_GN_MISH, p + ".blocks.0.block.2", latent_in)) != CLD_OK) return rc;
        rb.c0.cb_off = cb_off;
        if ((rc = make_conv(rb.c1, p + ".blocks.1.block.0", bd.cout, bd.cout, 0, bd.L, bd.L, 1, 5, k5, false, -2, 0, 1,
                            bd.L, EPI_GN_MISH, p + ".blocks.1.block.2")) != CLD_OK) return rc;
        rb.has_res = bd.cin != bd.cout && !latent_in;
        if (rb.has_res)
            if ((rc = make_conv(rb.res, p + ".residual_conv", bd.cout, c1, c2, bd.L, bd.L, 1, 1, k1, false, 0, 0, 1, bd.L,
                                EPI_BIAS, "", latent_in)) != CLD_OK) return rc;
        if (latent_in) {      // the first block's 1x1 residual projection of the 4-channel latent runs inside its second conv's epilogue
            UP(h->res4_w, *getw(h, p + ".residual_conv.weight"));      // [64, 4, 1]
            UP(h->res4_b, *getw(h, p + ".residual_conv.bias"));
        }
        cb_off += bd.cout;
    }
    if ((rc = make_conv(h->down[0], "model.downs.0.2.conv", 64, 64, 0, 52, 26, 2, 3, k3, false, -1, 0, 1, 26, EPI_BIAS, ""))) return rc;
    if ((rc = make_conv(h->down[1], "model.downs.1.2.conv", 128, 128, 0, 26, 13, 2, 3, k3, false, -1, 0, 1, 13, EPI_BIAS, ""))) return rc;
    for (int u = 0; u < 2; ++u) {
        const std::string p = u == 0 ? "model.ups.0.2.conv" : "model.ups.1.2.conv";
        const int c = u == 0 ? 128 : 64, L = u == 0 ? 13 : 26;
        if ((rc = make_conv(h->upT[u][0], p, c, c, 0, L, L, 1, 2, kT_even, true, -1, 0, 2, 2 * L, EPI_BIAS, ""))) return rc;
        if ((rc = make_conv(h->upT[u][1], p, c, c, 0, L, L, 1, 2, kT_odd, true, 0, 1, 2, 2 * L, EPI_BIAS, ""))) return rc;
    }
    if ((rc = make_conv(h->final_cb, "model.final_conv.0.block.0", 64, 64, 0, 52, 52, 1, 5, k5, false, -2, 0, 1, 52,
                        EPI_GN_MISH, "model.final_conv.0.block.2", false, /*out_f32: the head kernel reads fp32*/ true))) return rc;

    // ---- cond half of every block's time_mlp Linear, concatenated: wc [1792][256], bias [1792] ----
    // ---- time half folded with the timestep embedding into a table tb [n_timesteps][1792]       ----
    {
        std::vector<float> wc((size_t)NCB * COND), bb(NCB);
        std::vector<float> wt((size_t)NCB * TE);
        int off = 0;
        for (const auto& bd : kBlocks) {
            const std::vector<float>& W = *getw(h, std::string(bd.name) + ".time_mlp.1.weight");   // [cout, 288]: [0:32] time, [32:288] cond
            const std::vector<float>& Bv = *getw(h, std::string(bd.name) + ".time_mlp.1.bias");
            for (int n = 0; n < bd.cout; ++n) {
                for (int k = 0; k < TE; ++k) wt[(size_t)(off + n) * TE + k] = W[(size_t)n * (TE + COND) + k];
                for (int k = 0; k < COND; ++k) wc[(size_t)(off + n) * COND + k] = W[(size_t)n * (TE + COND) + TE + k];
                bb[off + n] = Bv[n];
            }
            off += bd.cout;
        }
        UP(h->wc, wc);
        UP(h->cbias_b, bb);
        const std::vector<float>& W1 = *getw(h, "model.time_mlp.1.weight");   // [128,32]
        const std::vector<float>& B1 = *getw(h, "model.time_mlp.1.bias");
        const std::vector<float>& W3 = *getw(h, "model.time_mlp.3.weight");   // [32,128]
        const std::vector<float>& B3 = *getw(h, "model.time_mlp.3.bias");
        const int n = h->cfg.n_timesteps;
        std::vector<float> tb((size_t)n * NCB);
        const int half = TE / 2;
        const float negc = (float)(-(std::log(10000.0) / (half - 1)));
        for (int t = 0; t < n; ++t) {
            float emb[TE];
            for (int k = 0; k < half; ++k) {              // diffuser_helpers.py:25-32, fp32 like the reference
                const float wk = std::exp((float)k * negc);
                const float ph = (float)t * wk;
                emb[k] = std::sin(ph);
                emb[half + k] = std::cos(ph);
            }
            double h1[128], te[TE];
            for (int j = 0; j < 128; ++j) {
                double a = B1[j];
                for (int k = 0; k < TE; ++k) a += (double)W1[j * TE + k] * emb[k];
                h1[j] = mish_d((double)(float)a);
            }
            for (int j = 0; j < TE; ++j) {
                double a = B3[j];
                for (int k = 0; k < 128; ++k) a += (double)W3[j * 128 + k] * (double)(float)h1[k];
                te[j] = mish_d((double)(float)a);          // the block's Mish on the [t_emb | cond] vector
            }
            for (int nn = 0; nn < NCB; ++nn) {
                double a = 0;
                for (int k = 0; k < TE; ++k) a += (double)wt[(size_t)nn * TE + k] * (double)(float)te[k];
                tb[(size_t)t * NCB + nn] = (float)a;
            }
        }
        UP(h->tb, tb);
    }
    {
        std::vector<float> qs(h->sqrt_acp);
        qs.insert(qs.end(), h->sqrt_1m_acp.begin(), h->sqrt_1m_acp.end());
        UP(h->qs_tab, qs);
    }
    UP(h->head_w, *getw(h, "model.final_conv.1.weight"));
    UP(h->head_b, *getw(h, "model.final_conv.1.bias"));
    {   // final_conv.1 [4,64,1] as one 16-column N tile of the chain kernel (columns 4..15 zero)
        const std::vector<float>& W = *getw(h, "model.final_conv.1.weight");
        UP(h->head_wfrag, pack_conv_weights([&](int co, int ci, int) { return co < 4 ? W[(size_t)co * 64 + ci] : 0.f; }, 16, 64, 1));
    }
    return CLD_OK;
    };
    if (h->has_unet) {
        const int rcu = build_unet();
        if (rcu != CLD_OK) return rcu;
    }
    int rc = CLD_OK;

    // ---- decoder (optional) ---------------------------------------------------------------
    h->has_decoder = true;
    for (const auto& kv : h->expect)
        if (kv.first.rfind("lstm_dec.", 0) == 0 && !h->w.count(kv.first)) h->has_decoder = false;
    if (h->has_decoder) {
        float* tmp;
        UP(tmp, *getw(h, "lstm_dec.lstm.weight_ih_l0")); h->dec.w_ih0 = tmp;
        UP(tmp, *getw(h, "lstm_dec.lstm.weight_hh_l0")); h->dec.w_hh0 = tmp;
        UP(tmp, *getw(h, "lstm_dec.lstm.weight_ih_l1")); h->dec.w_ih1 = tmp;
        UP(tmp, *getw(h, "lstm_dec.lstm.weight_hh_l1")); h->dec.w_hh1 = tmp;
        std::vector<float> b0(256), b1(256);
        for (int i = 0; i < 256; ++i) {
            b0[i] = (*getw(h, "lstm_dec.lstm.bias_ih_l0"))[i] + (*getw(h, "lstm_dec.lstm.bias_hh_l0"))[i];
            b1[i] = (*getw(h, "lstm_dec.lstm.bias_ih_l1"))[i] + (*getw(h, "lstm_dec.lstm.bias_hh_l1"))[i];
        }
        UP(tmp, b0); h->dec.b0 = tmp;
        UP(tmp, b1); h->dec.b1 = tmp;
        UP(tmp, *getw(h, "lstm_dec.cond2hidden.weight")); h->dec.w_c2h = tmp;
        UP(tmp, *getw(h, "lstm_dec.cond2hidden.bias")); h->dec.b_c2h = tmp;
        UP(tmp, *getw(h, "lstm_dec.hid2act.weight")); h->dec.w_h2a = tmp;
        UP(tmp, *getw(h, "lstm_dec.hid2act.bias")); h->dec.b_h2a = tmp;
        {   // B fragments of the backward products of guide_mfma8_kernel (guide_kernels.hip): wave wv owns units 8 wv .. 8 wv + 7;
            // tile 0 = layer 1: column n < 8 -> W_hh1[col][8 wv + n], n >= 8 -> W_ih1[col][8 wv + n - 8];
            // tile 1 = layer 0: column n < 8 -> W_hh0[col][8 wv + n], 8 <= n < 12 -> W_ih0[col][n - 8], else 0;
            // k-step (j, e) of lane (n, rb) is gate column col = 16 j + 4 rb + e.  One coalesced float4 per (tile, j) and lane.
            const std::vector<float>&hh1 = *getw(h, "lstm_dec.lstm.weight_hh_l1"), &ih1 = *getw(h, "lstm_dec.lstm.weight_ih_l1"),
                                    &hh0 = *getw(h, "lstm_dec.lstm.weight_hh_l0"), &ih0 = *getw(h, "lstm_dec.lstm.weight_ih_l0");
            std::vector<float> g((size_t)8 * 2 * 16 * 64 * 4);
            size_t o = 0;
            for (int wv = 0; wv < 8; ++wv)
                for (int tile = 0; tile < 2; ++tile)
                    for (int j = 0; j < 16; ++j)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int n = lane & 15, rb = lane >> 4, hi = n >> 3, m = n & 7, col = 16 * j + 4 * rb + e;
                                float v;
                                if (tile == 0) v = (hi ? ih1 : hh1)[(size_t)col * 64 + 8 * wv + m];
                                else v = hi ? (m < 4 ? ih0[(size_t)col * 4 + m] : 0.f) : hh0[(size_t)col * 64 + 8 * wv + m];
                                g[o++] = v;
                            }
            UP(tmp, g); h->dec.gfrag = tmp;
            // B operands of the backward products of guide_quad_kernel: waves wv and wv + 4 own units 16 wv .. 16 wv + 15; lane = 4 block + column,
            // block = (K half kh << 3) | (product m << 2) | unit quad ub; k-step (j, e) is gate column col = 128 kh + 4 j + e;
            // layer 1 (first): m = 0 -> W_hh1[col][16 wv + 4 ub + row], m = 1 -> W_ih1[col][same];
            // layer 0: m = 0 -> W_hh0[col][same], m = 1 and ub = 0 -> W_ih0[col][latent channel row], else 0
            std::vector<float> gq((size_t)4 * 2 * 32 * 64 * 4);
            o = 0;
            for (int wv = 0; wv < 4; ++wv)
                for (int layer = 1; layer >= 0; --layer)
                    for (int j = 0; j < 32; ++j)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int row = lane & 3, blk = lane >> 2, kh = blk >> 3, m = (blk >> 2) & 1, ub = blk & 3;
                                const int col = 128 * kh + 4 * j + e, unit = 16 * wv + 4 * ub + row;
                                float v;
                                if (layer == 1) v = (m ? ih1 : hh1)[(size_t)col * 64 + unit];
                                else v = m ? (ub == 0 ? ih0[(size_t)col * 4 + row] : 0.f) : hh0[(size_t)col * 64 + unit];
                                gq[o++] = v;
                            }
            UP(tmp, gq); h->dec.gqfrag = tmp;
        }
    }
    // ---- encoder (optional) ---------------------------------------------------------------
    h->has_encoder = true;
    for (const auto& kv : h->expect)
        if ((kv.first.rfind("lstm_enc.", 0) == 0 || kv.first.rfind("mu.", 0) == 0 || kv.first.rfind("logvar.", 0) == 0) &&
            !h->w.count(kv.first))
            h->has_encoder = false;
    if (h->has_encoder) {
        float* tmp;
        UP(tmp, *getw(h, "lstm_enc.lstm.weight_ih_l0")); h->enc.w_ih0 = tmp;
        UP(tmp, *getw(h, "lstm_enc.lstm.weight_hh_l0")); h->enc.w_hh0 = tmp;
        UP(tmp, *getw(h, "lstm_enc.lstm.weight_ih_l1")); h->enc.w_ih1 = tmp;
        UP(tmp, *getw(h, "lstm_enc.lstm.weight_hh_l1")); h->enc.w_hh1 = tmp;
        std::vector<float> b0(256), b1(256);
        for (int i = 0; i < 256; ++i) {
            b0[i] = (*getw(h, "lstm_enc.lstm.bias_ih_l0"))[i] + (*getw(h, "lstm_enc.lstm.bias_hh_l0"))[i];
            b1[i] = (*getw(h, "lstm_enc.lstm.bias_ih_l1"))[i] + (*getw(h, "lstm_enc.lstm.bias_hh_l1"))[i];
        }
        UP(tmp, b0); h->enc.b0 = tmp;
        UP(tmp, b1); h->enc.b1 = tmp;
        UP(tmp, *getw(h, "lstm_enc.cond2hidden.weight")); h->enc.w_c2h = tmp;
        UP(tmp, *getw(h, "lstm_enc.cond2hidden.bias")); h->enc.b_c2h = tmp;
        UP(tmp, *getw(h, "mu.weight")); h->enc.w_mu = tmp;
        UP(tmp, *getw(h, "mu.bias")); h->enc.b_mu = tmp;
        UP(tmp, *getw(h, "logvar.weight")); h->enc.w_lv = tmp;
        UP(tmp, *getw(h, "logvar.bias")); h->enc.b_lv = tmp;
    }
    // ---- ContextEncoder (optional) -----------------------------------------------------------
    h->has_context = true;
    for (const auto& kv : h->expect)
        if (kv.first.rfind("context_encoder.", 0) == 0 && !h->w.count(kv.first)) h->has_context = false;
    if (h->has_context) {
        // eval-mode BatchNorm2d folded to y = x * scale + shift (eps 1e-5, torchvision resnet.py norm_layer default)
        auto fold_bn = [&](const std::string& p, float** scale, float** shift) -> int {
            const std::vector<float>&g = *getw(h, p + ".weight"), &b = *getw(h, p + ".bias"), &m = *getw(h, p + ".running_mean"),
                                    &v = *getw(h, p + ".running_var");
            std::vector<float> sc(g.size()), sh(g.size());
            for (size_t i = 0; i < g.size(); ++i) {
                const double k = (double)g[i] / std::sqrt((double)v[i] + 1e-5);
                sc[i] = (float)k;
                sh[i] = (float)((double)b[i] - (double)m[i] * k);
            }
            UP(*scale, sc);
            UP(*shift, sh);
            return CLD_OK;
        };
        const std::string r = kResnet;
        {   // stem: per input plane c the 49 taps are 13 MFMA k-steps (k = 4 q + (lane >> 4)); lane holds 4 k-steps per float4
            const std::vector<float>& W = *getw(h, r + "conv1.weight");     // [64][34][7][7]
            std::vector<float> wq((size_t)34 * 4 * 4 * 64 * 4);
            size_t o = 0;
            for (int c = 0; c < 34; ++c)
                for (int qg = 0; qg < 4; ++qg)
                    for (int nt = 0; nt < 4; ++nt)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int sidx = 0; sidx < 4; ++sidx) {
                                const int k = 4 * (4 * qg + sidx) + (lane >> 4);
                                wq[o++] = k < 49 ? W[((size_t)(16 * nt + (lane & 15)) * 34 + c) * 49 + k] : 0.f;
                            }
            UP(h->stem_w, wq);
            if ((rc = fold_bn(r + "bn1", &h->stem_scale, &h->stem_shift)) != CLD_OK) return rc;
        }
        auto make2d = [&](cld_handle_s::Conv2dLayer& l, const std::string& wname, const std::string& bnname, int kh, int stride,
                          int hin, int cin, int cout) -> int {
            const std::vector<float>& W = *getw(h, wname);                  // [cout][cin][kh][kh]
            auto wget = [&](int co, int ci, int t) -> float { return W[((size_t)co * cin + ci) * kh * kh + t]; };
            std::vector<float> packed = pack_conv_weights(wget, cout, cin, kh * kh);
            UP(l.wfrag, packed);
            l.kh = kh; l.stride = stride; l.hin = hin; l.cin = cin; l.cout = cout;
            if (kh == 3 && stride == 1 && cin == cout) {
                // Winograd F(2x2, 3x3): U[xi = 4 i + j][ci][co] = (G g G^T)[i][j], G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], formed in double
                static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
                std::vector<float> U((size_t)16 * cin * cout);
                for (int co = 0; co < cout; ++co)
                    for (int ci = 0; ci < cin; ++ci) {
                        const float* g = &W[((size_t)co * cin + ci) * 9];
                        double t[4][3];
                        for (int i = 0; i < 4; ++i)
                            for (int b = 0; b < 3; ++b) t[i][b] = Gm[i][0] * g[b] + Gm[i][1] * g[3 + b] + Gm[i][2] * g[6 + b];
                        for (int i = 0; i < 4; ++i)
                            for (int j = 0; j < 4; ++j)
                                U[((size_t)(4 * i + j) * cin + ci) * cout + co] = (float)(t[i][0] * Gm[j][0] + t[i][1] * Gm[j][1] + t[i][2] * Gm[j][2]);
                    }
                auto uget = [&](int co, int ci, int xi) -> float { return U[((size_t)xi * cin + ci) * cout + co]; };
                std::vector<float> upacked = pack_conv_weights(uget, cout, cin, 16);
                UP(l.ufrag, upacked);
                if (wino44_supported(hin, cout)) {
                    // Winograd F(4x4, 3x3) at the points {0, 1, -1, 1/2, -2, inf} (wino44_kernels.hip): U[xi = 6 i + j] = (G g G^T)[i][j], in double
                    static const double G6[6][3] = {{1, 0, 0}, {1.0 / 3, 1.0 / 3, 1.0 / 3}, {-1.0 / 3, 1.0 / 3, -1.0 / 3},
                                                    {-16.0 / 15, -8.0 / 15, -4.0 / 15}, {1.0 / 15, -2.0 / 15, 4.0 / 15}, {0, 0, 1}};
                    std::vector<float> U6((size_t)36 * cin * cout);
                    for (int co = 0; co < cout; ++co)
                        for (int ci = 0; ci < cin; ++ci) {
                            const float* g = &W[((size_t)co * cin + ci) * 9];
                            double t[6][3];
                            for (int i = 0; i < 6; ++i)
                                for (int bb = 0; bb < 3; ++bb) t[i][bb] = G6[i][0] * g[bb] + G6[i][1] * g[3 + bb] + G6[i][2] * g[6 + bb];
                            for (int i = 0; i < 6; ++i)
                                for (int j = 0; j < 6; ++j)
                                    U6[((size_t)(6 * i + j) * cin + ci) * cout + co] = (float)(t[i][0] * G6[j][0] + t[i][1] * G6[j][1] + t[i][2] * G6[j][2]);
                        }
                    auto u6get = [&](int co, int ci, int xi) -> float { return U6[((size_t)xi * cin + ci) * cout + co]; };
                    std::vector<float> u6packed = pack_conv_weights(u6get, cout, cin, 36);
                    UP(l.ufrag44, u6packed);
                }
            }
            return fold_bn(bnname, &l.scale, &l.shift);
        };
        int cin = 64, hin = 56;
        for (int li = 1; li <= 4; ++li) {
            const int c = 32 << li;
            for (int b = 0; b < 2; ++b) {
                const std::string p = r + "layer" + std::to_string(li) + "." + std::to_string(b);
                const int stride = (b == 0 && li > 1) ? 2 : 1;
                const int hout = hin / stride;
                if ((rc = make2d(h->rn_conv[li - 1][b][0], p + ".conv1.weight", p + ".bn1", 3, stride, hin, b == 0 ? cin : c, c))) return rc;
                if ((rc = make2d(h->rn_conv[li - 1][b][1], p + ".conv2.weight", p + ".bn2", 3, 1, hout, c, c))) return rc;
                if (b == 0 && cin != c)
                    if ((rc = make2d(h->rn_ds[li - 1], p + ".downsample.0.weight", p + ".downsample.1", 1, 2, hin, cin, c))) return rc;
                hin = hout;
            }
            cin = c;
        }
        // head: transposed Linear weights [in][out]
        auto upT = [&](const std::string& wname, const float** dst) -> int {
            const std::vector<float>& W = *getw(h, wname);
            const std::string bname = wname.substr(0, wname.size() - 6) + "bias";
            const size_t n_out = getw(h, bname)->size(), n_in = W.size() / n_out;
            std::vector<float> t(W.size());
            for (size_t o = 0; o < n_out; ++o)
                for (size_t i = 0; i < n_in; ++i) t[i * n_out + o] = W[o * n_in + i];
            float* d; UP(d, t); *dst = d;
            return CLD_OK;
        };
        auto up1 = [&](const std::string& name, const float** dst) -> int { float* d; UP(d, *getw(h, name)); *dst = d; return CLD_OK; };
        ContextHeadArgs& a = h->ctx_head;
        if ((rc = upT(r + "fc.weight", &a.fc_wt)) || (rc = up1(r + "fc.bias", &a.fc_b))) return rc;
        const std::string sp = "context_encoder.agent_state_encoder._model.", cp = "context_encoder.process_cond_mlp._model.";
        for (int i = 0; i < 3; ++i) {
            if ((rc = upT(sp + std::to_string(3 * i) + ".weight", &a.s_wt[i])) || (rc = up1(sp + std::to_string(3 * i) + ".bias", &a.s_b[i]))) return rc;
            if (i < 2 && ((rc = up1(sp + std::to_string(3 * i + 1) + ".weight", &a.s_g[i])) || (rc = up1(sp + std::to_string(3 * i + 1) + ".bias", &a.s_be[i])))) return rc;
        }
        for (int i = 0; i < 5; ++i) {
            if ((rc = upT(cp + std::to_string(3 * i) + ".weight", &a.c_wt[i])) || (rc = up1(cp + std::to_string(3 * i) + ".bias", &a.c_b[i]))) return rc;
            if (i < 4 && ((rc = up1(cp + std::to_string(3 * i + 1) + ".weight", &a.c_g[i])) || (rc = up1(cp + std::to_string(3 * i + 1) + ".bias", &a.c_be[i])))) return rc;
        }
    }
#undef UP
    HIPCK(h, hipStreamSynchronize(s));     // host staging vectors die with this scope
    h->w.clear();
    h->finalized = true;
    return CLD_OK;
}

size_t cld_workspace_bytes(cld_handle h, int32_t B) {
    if (!h || B < 1) return 0;
    return ws_floats(pad16(B)) * sizeof(float);
}

int cld_get_schedule(cld_handle h, float* x_t_cof, float* noise_cof, float* post_log_var) {
    if (!h) return CLD_ERR_ARG;
    const size_t n = h->x_t_cof.size() * sizeof(float);
    if (x_t_cof) std::memcpy(x_t_cof, h->x_t_cof.data(), n);
    if (noise_cof) std::memcpy(noise_cof, h->noise_cof.data(), n);
    if (post_log_var) std::memcpy(post_log_var, h->plvc.data(), n);
    return CLD_OK;
}

static int check_common(cld_handle h, const char* fn, int B, int t_idx, const void* ws, size_t ws_bytes) {
    if (!h) return CLD_ERR_ARG;
    if (!h->finalized) return fail(h, CLD_ERR_STATE, std::string(fn) + ": weights not finalized");
    if (!h->has_unet && std::string(fn) != "cld_guidance_step") return fail(h, CLD_ERR_STATE, std::string(fn) + ": U-Net weights (model.*) not loaded");
    if (B < 1) return fail(h, CLD_ERR_ARG, std::string(fn) + ": B < 1");
    if (t_idx < 0 || t_idx >= h->cfg.n_timesteps) return fail(h, CLD_ERR_ARG, std::string(fn) + ": timestep out of range");
    if (!ws || ws_bytes < cld_workspace_bytes(h, B)) return fail(h, CLD_ERR_WORKSPACE, std::string(fn) + ": workspace too small");
    if (reinterpret_cast<uintptr_t>(ws) % 16) return fail(h, CLD_ERR_ARG, std::string(fn) + ": workspace must be 16-byte aligned");
    return CLD_OK;
}

int cld_unet_forward(cld_handle h, const float* x, const float* cond, int32_t t_idx, float* eps, int32_t B,
                     void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common(h, "cld_unet_forward", B, t_idx, workspace, workspace_bytes);
    if (rc) return rc;
    if (!x || !cond || !eps) return fail(h, CLD_ERR_ARG, "cld_unet_forward: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int bp = pad16(B);
    Ws w = carve(workspace, bp);
    HIPCK(h, launch_pack_latent(x, w.xw, B, bp, s));
    HIPCK(h, launch_cond_bias(cond, h->wc, h->cbias_b, w.cb, B, bp, NCB, s));
    HIPCK(h, run_unet(h, w, w.xw, t_idx, bp, s));
    HeadArgs a{};
    head_source(h, w, a); a.w = h->head_w; a.b = h->head_b; a.x = w.xw; a.eps_out = eps; a.B = B; a.b_pad = bp;
    HIPCK(h, launch_head(a, s));
    return CLD_OK;
}

int cld_unet_forward_t(cld_handle h, const float* x, const float* cond, const int32_t* t_idx, float* eps, int32_t B,
                       void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common(h, "cld_unet_forward_t", B, 0, workspace, workspace_bytes);
    if (rc) return rc;
    if (!x || !cond || !t_idx || !eps) return fail(h, CLD_ERR_ARG, "cld_unet_forward_t: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int bp = pad16(B);
    Ws w = carve(workspace, bp);
    HIPCK(h, launch_pack_latent(x, w.xw, B, bp, s));
    HIPCK(h, launch_cond_bias(cond, h->wc, h->cbias_b, w.cb, B, bp, NCB, s));
    HIPCK(h, launch_add_time_bias(w.cb, h->tb, t_idx, h->cfg.n_timesteps, B, NCB, s));
    HIPCK(h, run_unet(h, w, w.xw, -1, bp, s));
    HeadArgs a{};
    head_source(h, w, a); a.w = h->head_w; a.b = h->head_b; a.x = w.xw; a.eps_out = eps; a.B = B; a.b_pad = bp;
    HIPCK(h, launch_head(a, s));
    return CLD_OK;
}

int cld_denoise_loss(cld_handle h, const float* z0, const float* noise, const float* cond, const int32_t* t_idx, float* z_noisy,
                     float* mse, int32_t B, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common(h, "cld_denoise_loss", B, 0, workspace, workspace_bytes);
    if (rc) return rc;
    if (!z0 || !noise || !t_idx || (!mse && !z_noisy) || (mse && !cond)) return fail(h, CLD_ERR_ARG, "cld_denoise_loss: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int bp = pad16(B);
    Ws w = carve(workspace, bp);
    // z_t = sqrt(acp[t]) z0 + sqrt(1 - acp[t]) noise  (q_sample, dm_model.py:91-96) straight into the padded latent buffer
    HIPCK(h, launch_q_sample(z0, noise, t_idx, h->qs_tab, h->cfg.n_timesteps, w.xw, z_noisy, B, bp, s));
    if (!mse) return CLD_OK;               // q_sample alone (dm_model.py:91-96): no U-Net evaluation
    HIPCK(h, launch_cond_bias(cond, h->wc, h->cbias_b, w.cb, B, bp, NCB, s));
    HIPCK(h, launch_add_time_bias(w.cb, h->tb, t_idx, h->cfg.n_timesteps, B, NCB, s));
    HIPCK(h, run_unet(h, w, w.xw, -1, bp, s));
    HeadArgs a{};
    head_source(h, w, a); a.w = h->head_w; a.b = h->head_b; a.x = w.xw; a.eps_out = w.xtmp; a.B = B; a.b_pad = bp;
    HIPCK(h, launch_head(a, s));
    HIPCK(h, launch_mse_rows(noise, w.xtmp, mse, B, s));
    return CLD_OK;
}

int cld_ddpm_step(cld_handle h, const float* x, const float* cond, int32_t t_idx, const float* z, float* x_next,
                  float* mean, float* sigma_host, int32_t B, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common(h, "cld_ddpm_step", B, t_idx, workspace, workspace_bytes);
    if (rc) return rc;
    if (!x || !cond || (!z && x_next && t_idx != 0)) return fail(h, CLD_ERR_ARG, "cld_ddpm_step: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int bp = pad16(B);
    Ws w = carve(workspace, bp);
    const float sigma = std::exp(0.5f * h->plvc[t_idx]);
    if (sigma_host) *sigma_host = sigma;
    HIPCK(h, launch_pack_latent(x, w.xw, B, bp, s));
    HIPCK(h, launch_cond_bias(cond, h->wc, h->cbias_b, w.cb, B, bp, NCB, s));
    HIPCK(h, run_unet(h, w, w.xw, t_idx, bp, s));
    HeadArgs a{};
    head_source(h, w, a); a.w = h->head_w; a.b = h->head_b; a.x = w.xw; a.z = z; a.B = B; a.b_pad = bp;
    a.mean_out = w.meanb; a.x_out = w.xtmp;
    a.xc = h->x_t_cof[t_idx]; a.nc = h->noise_cof[t_idx];
    a.sg = (t_idx == 0) ? 0.f : sigma;       // nonzero_mask, dm_model.py:151
    HIPCK(h, launch_head(a, s));
    if (x_next) HIPCK(h, launch_unpack(w.xtmp, x_next, B, s));
    if (mean) HIPCK(h, launch_unpack(w.meanb, mean, B, s));
    return CLD_OK;
}

static int check_collision(cld_handle h, const char* fn, const cld_collision* c, int B) {
    if (!c->extent || !c->world_from_agent || !c->curr_speed || !c->scene_start) return fail(h, CLD_ERR_ARG, std::string(fn) + ": collision term: null pointer");
    if (c->num_scenes < 1 || c->num_samp < 1 || B % c->num_samp) return fail(h, CLD_ERR_ARG, std::string(fn) + ": collision term: B must be agents x num_samp, num_scenes >= 1");
    if (c->num_disks < 1 || c->num_disks > 8) return fail(h, CLD_ERR_ARG, std::string(fn) + ": collision term: num_disks must be 1..8");
    if (c->max_scene_agents < 1 || c->max_scene_agents > 150) return fail(h, CLD_ERR_ARG, std::string(fn) + ": collision term: scenes of 1..150 agents");
    return CLD_OK;
}

static int check_map_collision(cld_handle h, const char* fn, const cld_map_collision* c, int B) {
    if (!c->extent || !c->raster_from_agent || !c->drivable_map || !c->curr_speed || !c->scene_start)
        return fail(h, CLD_ERR_ARG, std::string(fn) + ": map collision term: null pointer");
    if (c->num_scenes < 1 || c->num_samp < 1 || B % c->num_samp || c->H < 1 || c->W < 1)
        return fail(h, CLD_ERR_ARG, std::string(fn) + ": map collision term: B must be agents x num_samp, a map of H x W >= 1");
    if (c->num_points_l < 1 || c->num_points_w < 1 || c->num_points_l * c->num_points_w > 256)
        return fail(h, CLD_ERR_ARG, std::string(fn) + ": map collision term: 1..256 sample points per box");
    return CLD_OK;
}
static MapCollisionArgs map_args(const cld_map_collision* c, const float* traj, const float* grad_in, float* loss, float* grad) {
    MapCollisionArgs a{};
    a.traj = traj; a.extent = c->extent; a.raster_from_agent = c->raster_from_agent; a.drivable_map = c->drivable_map;
    a.curr_speed = c->curr_speed; a.scene_start = c->scene_start; a.scene_weight = c->scene_weight; a.grad_in = grad_in;
    a.loss = loss; a.grad = grad; a.num_scenes = c->num_scenes; a.num_samp = c->num_samp; a.H = c->H; a.W = c->W;
    a.num_points_l = c->num_points_l; a.num_points_w = c->num_points_w; a.decay_rate = c->decay_rate; a.moving_speed_th = c->moving_speed_th;
    return a;
}

static int check_guidance(cld_handle h, const char* fn, const cld_guidance* gd, int B) {
    if (!h->has_decoder) return fail(h, CLD_ERR_STATE, std::string(fn) + ": guidance needs the decoder weights");
    if (!gd->curr_states || (!gd->target_speed && !gd->speed_limit_scale && !gd->acc_limit_scale && !gd->target_pos_scale && !gd->ext_grad && !gd->collision && !gd->map_collision))
        return fail(h, CLD_ERR_ARG, std::string(fn) + ": guidance needs curr_states and at least one loss term");
    if (gd->collision) {
        const int rcc = check_collision(h, fn, gd->collision, B);
        if (rcc) return rcc;
    }
    if (gd->map_collision) {
        const int rcc = check_map_collision(h, fn, gd->map_collision, B);
        if (rcc) return rcc;
    }
    if (gd->grad_steps < 0 || gd->final_grad_steps < 0 || gd->grad_steps > 64 || gd->final_grad_steps > 64)
        return fail(h, CLD_ERR_ARG, std::string(fn) + ": grad_steps out of range (0..64)");
    if (gd->optimizer != CLD_GUIDE_ADAM && gd->optimizer != CLD_GUIDE_SGD) return fail(h, CLD_ERR_ARG, std::string(fn) + ": unknown optimizer");
    if (gd->apply_output && gd->final_optimizer != CLD_GUIDE_ADAM && gd->final_optimizer != CLD_GUIDE_SGD)
        return fail(h, CLD_ERR_ARG, std::string(fn) + ": unknown optimizer for the output step");
    if (gd->target_pos_scale && (!gd->target_pos || !gd->target_time)) return fail(h, CLD_ERR_ARG, std::string(fn) + ": target_pos_scale needs target_pos and target_time");
    return CLD_OK;
}

// The guidance step(s) of one denoising iteration on the posterior mean `mean0` (upstream perturb(), guidance_loss.py:2221-2282):
// grad_steps optimiser steps, each one guidance-kernel launch (decoder forward + roll-out + loss gradients + BPTT + update) on the
// current iterate; with a collision term each step first decodes the iterate and runs the AgentCollisionLoss kernel, whose
// gradient w.r.t. the plans enters the guidance kernel as ext_grad.  The last step adds sigma z and writes x_out (/ x_out2).
// `intermediate`: a t > 0 step (lr / perturb_th / optimizer / grad_steps) or the output step (final_*).
static int run_guidance(cld_handle h, const Ws& w, const cld_guidance* gd, int B, bool intermediate, float sigma_t, float sigma_noise,
                        const float* mean0, const float* cond, const float* z, uint64_t seed, unsigned long long salt,
                        float* mean_out, float* x_out, float* x_out2, float* grad_out, hipStream_t s) {
    GuideArgs g{};
    g.cond = cond; g.curr_states = gd->curr_states; g.target_speed = gd->target_speed; g.loss_scale = gd->loss_scale;
    g.speed_limit_scale = gd->speed_limit_scale; g.acc_limit_scale = gd->acc_limit_scale;
    g.speed_limit = gd->speed_limit; g.acc_limit = gd->acc_limit;
    g.target_pos = gd->target_pos; g.target_time = gd->target_time; g.target_pos_scale = gd->target_pos_scale;
    g.ext_grad = gd->ext_grad;
    // t = 0 (apply_guidance_output): the step's own optimiser settings, and no noise behind it (nonzero_mask, diffuser.py:929)
    const float lr_in = intermediate ? gd->lr : gd->final_lr, th_in = intermediate ? gd->perturb_th : gd->final_perturb_th;
    g.scratch = w.guide; g.lr = lr_in > 0.f ? lr_in : sigma_t; g.perturb_th = th_in > 0.f ? th_in : (th_in == 0.f ? sigma_t : -1.f);
    g.optimizer = intermediate ? gd->optimizer : gd->final_optimizer; g.B = B; g.seed = seed; g.step_salt = salt;
    const int steps_in = intermediate ? gd->grad_steps : gd->final_grad_steps;
    const int steps = steps_in > 1 ? steps_in : 1;
    g.opt_steps = steps; g.mean0 = mean0; g.adam_m = w.adam_m; g.adam_v = w.adam_v;
    for (int k = 1; k <= steps; ++k) {
        const bool last = k == steps;
        g.opt_step = k;
        g.mean = k == 1 ? mean0 : w.gcur;
        if (gd->collision || gd->map_collision) {    // the scene / map terms are functions of the decoded plans of the current iterate
            if (B >= 256 && guide_forward_available(B, h->force_kernel[CLD_KERNEL_GUIDE]) && h->force_kernel[CLD_KERNEL_DECODE] == FORM_AUTO) {
                // (from the batch size at which cld_decode itself would take its 16-agent MFMA form, ~0.3 ms; below that the
                //  one-agent-per-workgroup decoder is the shorter chain: 64 agents 579 vs 616 us per collision-guided step)
                // the guidance kernel's own forward sweep as the decoder (8 agents per workgroup: the whole chip at 2,048 agents), then
                // the O(T) roll-out.  Up to 2,048 agents every agent group has a workgroup and a scratch slot of its own in that launch: the
                // guidance kernel behind the loss kernels then runs its BACKWARD half only (GuideArgs::act_in), from the activations the
                // forward launch kept -- the sweep is not repeated (a collision-guided step 711 -> 6xx us at 2,048 agents)
                GuideArgs gf{};
                gf.mean = g.mean; gf.cond = cond; gf.curr_states = gd->curr_states; gf.scratch = w.guide; gf.B = B; gf.act_out = w.col_act;
                HIPCK(h, launch_guide_forward(h->dec, h->dyn, gf, s));
                HIPCK(h, launch_action_to_state(h->dyn, w.col_act, gd->curr_states, w.col_traj, B, 1, 1, s));
                g.act_in = guide_split_available(B, h->force_kernel[CLD_KERNEL_GUIDE]) && h->force_kernel[CLD_KERNEL_GUIDE] == FORM_AUTO ? w.col_act : nullptr;
            } else {
                HIPCK(h, launch_decode(h->dec, h->dyn, g.mean, cond, gd->curr_states, nullptr, w.col_traj, B, 1, s, h->force_kernel[CLD_KERNEL_DECODE]));
            }
        }
        if (gd->collision) {
            const cld_collision* c = gd->collision;
            CollisionArgs ca{};
            ca.traj = w.col_traj; ca.extent = c->extent; ca.world_from_agent = c->world_from_agent; ca.curr_speed = c->curr_speed;
            ca.scene_start = c->scene_start; ca.scene_weight = c->scene_weight; ca.guided = c->guided; ca.excluded = c->excluded; ca.max_scene_agents = c->max_scene_agents; ca.grad_in = gd->ext_grad;
            ca.grad = w.col_grad; ca.B_agents = B / c->num_samp; ca.num_scenes = c->num_scenes; ca.num_samp = c->num_samp;
            ca.num_disks = c->num_disks; ca.buffer_dist = c->buffer_dist; ca.decay_rate = c->decay_rate; ca.moving_speed_th = c->moving_speed_th;
            HIPCK(h, launch_agent_collision(ca, s));
            g.ext_grad = w.col_grad;
        }
        if (gd->map_collision) {                     // adds to whatever gradient is there already (caller's ext_grad, agent collisions)
            HIPCK(h, launch_map_collision(map_args(gd->map_collision, w.col_traj, gd->collision ? w.col_grad : gd->ext_grad, nullptr, w.col_grad), B, s));
            g.ext_grad = w.col_grad;
        }
        g.z = last ? z : nullptr;
        g.sigma = last ? sigma_noise : 0.f;
        g.mean_out = last ? mean_out : w.gcur;           // in place from step 2 on: a workgroup reads its agents' rows before it writes them
        g.x_out = last ? x_out : nullptr;
        g.x_out2 = last ? x_out2 : nullptr;
        g.grad_out = k == 1 ? grad_out : nullptr;        // dL/dmean of the FIRST step (what the single-step call reports)
        HIPCK(h, launch_guide(h->dec, h->dyn, g, s, h->force_kernel[CLD_KERNEL_GUIDE]));
    }
    return CLD_OK;
}

// One iteration of the ancestral loop at timestep i on the latent in w.xw (CFG: rows [bp, 2 bp) hold the same latent, w.cb the
// conditional / unconditional bias rows): U-Net, head (noise prediction -> posterior mean -> + sigma z), optional guidance step on
// the mean.  Leaves x_{i-1} in w.xw (both halves) and, at i == 0 or on a guided step, the unguided posterior mean in w.meanb.
// `z` is this step's noise slab (NULL: on-device generator keyed by (seed, salt)); mean_guided / grad (optional, [B,52,4]) receive
// the guided mean and dL/dmean of a guided step.
static int sample_iteration(cld_handle h, const Ws& w, bool cfg, int B, int bp, int i, const float* z, uint64_t seed,
                            unsigned long long salt, const float* cond, float guidance_w, const cld_guidance* gd,
                            float* mean_guided, float* grad, bool want_mean, hipStream_t s) {
    const int bpn = cfg ? 2 * bp : bp;
    float* x_hi = w.xw + (size_t)bp * T * D;
    const float sigma = std::exp(0.5f * h->plvc[i]);
    // upstream defaults: apply_guidance_intermediate on, apply_guidance_output off (diffuser.py:876-881, scene_edit_config.py:84-85)
    const bool guide = gd && (i > 0 ? !gd->no_intermediate : gd->apply_output != 0);
    HeadArgs a{};
    a.w = h->head_w; a.b = h->head_b; a.x = w.xw; a.B = B; a.b_pad = bp;
    a.z = z;
    a.seed = seed; a.step_salt = salt;
    a.xc = h->x_t_cof[i]; a.nc = h->noise_cof[i];
    a.sg = (i == 0) ? 0.f : sigma;           // nonzero_mask, dm_model.py:151
    if (guide) {
        a.mean_out = w.meanb;                            // the guidance kernel perturbs the mean and adds the noise
        if (gd->guide_clean) {                           // ... or the model's clean prediction x0_hat (diffuser.py:866-873, :710-719)
            a.xc = h->sqrt_recip_acp[i]; a.nc = h->sqrt_recipm1_acp[i];
        }
    } else {
        a.x_out = w.xw;                                  // in place: each thread rewrites the row it read
        a.x_out2 = cfg ? x_hi : nullptr;
        a.mean_out = (i == 0 || want_mean) ? w.meanb : nullptr;
    }
    h->fuse_upd = cfg ? nullptr : &a;                    // without the CFG combine the tail chain applies the update itself
    const hipError_t eu = run_unet(h, w, w.xw, i, bpn, s);
    h->fuse_upd = nullptr;
    HIPCK(h, eu);
    if (!h->upd_fused) {
        head_source(h, w, a);
        if (cfg) {
            if (a.eps_in) a.eps_in_uncond = a.eps_in + (size_t)bp * T * D;
            else a.f_uncond = w.buf[7] + (size_t)bp * T * 64;
            a.cfg_w = guidance_w;
        }
        HIPCK(h, launch_head(a, s));
    }
    if (guide) {
        // a guided t = 0 step (apply_output): x0 IS the guided mean, and log_prob_final is taken around it (w.xtmp), not around the
        // unguided mean it was stepped away from -- sigma_0 = 1e-10 would turn that step into ~ -1e18
        int rcg = run_guidance(h, w, gd, B, i > 0, sigma, a.sg, w.meanb, cond, a.z, seed, salt, mean_guided ? mean_guided : (i == 0 ? w.xtmp : nullptr),
                               w.xw, cfg ? x_hi : nullptr, grad, s);
        if (rcg) return rcg;
    }
    return CLD_OK;
}

static int sample_impl(cld_handle h, const char* fn, const float* x_T, const float* noise, const float* cond,
                       const float* non_cond, float guidance_w, const cld_guidance* gd, int32_t steps, float* x0, float* x1,
                       float* logp, int32_t B, uint64_t seed, void* workspace, size_t workspace_bytes, void* stream) {
    const bool cfg = non_cond != nullptr;
    int rc = check_common(h, fn, cfg ? 2 * pad16(B) : B, 0, workspace, workspace_bytes);
    if (rc) return rc;
    if (!x_T || !cond) return fail(h, CLD_ERR_ARG, std::string(fn) + ": null pointer");
    if (steps != loop_steps(h))
        return fail(h, CLD_ERR_ARG, std::string(fn) + ": steps must equal len(range(0, n_timesteps, stride)) = " + std::to_string(loop_steps(h)));
    if (gd && (rc = check_guidance(h, fn, gd, B)) != CLD_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // CFG: one 2B-agent batch per step: rows [0, bp) carry cond_feat, rows [bp, 2bp) the unconditional features,
    // both halves the same latent; the head combines the two noise predictions and rewrites both halves.
    const int bp = pad16(B), bpn = cfg ? 2 * bp : bp;
    Ws w = carve(workspace, bpn);
    float* x_hi = w.xw + (size_t)bp * T * D;
    HIPCK(h, launch_pack_latent(x_T, w.xw, B, bp, s));
    HIPCK(h, launch_cond_bias(cond, h->wc, h->cbias_b, w.cb, B, bp, NCB, s));
    if (cfg) {
        HIPCK(h, launch_pack_latent(x_T, x_hi, B, bp, s));
        HIPCK(h, launch_cond_bias(non_cond, h->wc, h->cbias_b, w.cb + (size_t)bp * NCB, B, bp, NCB, s));
    }
    for (int it = 0; it < steps; ++it) {
        const int i = (steps - 1 - it) * h->stride;
        rc = sample_iteration(h, w, cfg, B, bp, i, noise ? noise + (size_t)it * B * T * D : nullptr, seed, (unsigned long long)it, cond,
                              guidance_w, gd, nullptr, nullptr, false, s);
        if (rc) return rc;
        if (i == 1 && x1) HIPCK(h, launch_unpack(w.xw, x1, B, s));
        if (i == 0) {
            if (x0) HIPCK(h, launch_unpack(w.xw, x0, B, s));
            if (logp) HIPCK(h, launch_logprob(w.xw, (gd && gd->apply_output) ? w.xtmp : w.meanb, std::exp(0.5f * h->plvc[0]), logp, B, s));
        }
    }
    return CLD_OK;
}

int cld_sample_step(cld_handle h, const float* x_t, const float* cond, const float* non_cond, float guidance_w,
                    const cld_guidance* gd, int32_t t_idx, const float* z, float* x_next, float* mean, float* mean_guided,
                    float* grad, float* sigma_host, int32_t B, void* workspace, size_t workspace_bytes, void* stream) {
    const bool cfg = non_cond != nullptr;
    int rc = check_common(h, "cld_sample_step", cfg ? 2 * pad16(B) : B, t_idx, workspace, workspace_bytes);
    if (rc) return rc;
    if (!x_t || !cond || (!z && t_idx != 0)) return fail(h, CLD_ERR_ARG, "cld_sample_step: null pointer");
    if (gd && (rc = check_guidance(h, "cld_sample_step", gd, B)) != CLD_OK) return rc;
    if (sigma_host) *sigma_host = std::exp(0.5f * h->plvc[t_idx]);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int bp = pad16(B), bpn = cfg ? 2 * bp : bp;
    Ws w = carve(workspace, bpn);
    HIPCK(h, launch_pack_latent(x_t, w.xw, B, bp, s));
    HIPCK(h, launch_cond_bias(cond, h->wc, h->cbias_b, w.cb, B, bp, NCB, s));
    if (cfg) {
        HIPCK(h, launch_pack_latent(x_t, w.xw + (size_t)bp * T * D, B, bp, s));
        HIPCK(h, launch_cond_bias(non_cond, h->wc, h->cbias_b, w.cb + (size_t)bp * NCB, B, bp, NCB, s));
    }
    rc = sample_iteration(h, w, cfg, B, bp, t_idx, z, 0, 0, cond, guidance_w, gd, mean_guided, grad, mean != nullptr, s);
    if (rc) return rc;
    if (x_next) HIPCK(h, launch_unpack(w.xw, x_next, B, s));
    if (mean) HIPCK(h, launch_unpack(w.meanb, mean, B, s));
    return CLD_OK;
}

int cld_sample(cld_handle h, const float* x_T, const float* noise, const float* cond, int32_t steps, float* x0,
               float* x1, float* logp, int32_t B, uint64_t seed, void* workspace, size_t workspace_bytes, void* stream) {
    return sample_impl(h, "cld_sample", x_T, noise, cond, nullptr, 0.f, nullptr, steps, x0, x1, logp, B, seed, workspace,
                       workspace_bytes, stream);
}

int cld_sample_cfg(cld_handle h, const float* x_T, const float* noise, const float* cond, const float* non_cond,
                   float guidance_w, int32_t steps, float* x0, float* x1, float* logp, int32_t B, uint64_t seed,
                   void* workspace, size_t workspace_bytes, void* stream) {
    if (h && !non_cond) return fail(h, CLD_ERR_ARG, "cld_sample_cfg: null pointer");
    return sample_impl(h, "cld_sample_cfg", x_T, noise, cond, non_cond, guidance_w, nullptr, steps, x0, x1, logp, B, seed,
                       workspace, workspace_bytes, stream);
}

int cld_sample_guided(cld_handle h, const float* x_T, const float* noise, const float* cond, const float* non_cond,
                      float guidance_w, const cld_guidance* guidance, int32_t steps, float* x0, float* x1, float* logp,
                      int32_t B, uint64_t seed, void* workspace, size_t workspace_bytes, void* stream) {
    if (h && !guidance) return fail(h, CLD_ERR_ARG, "cld_sample_guided: null guidance");
    return sample_impl(h, "cld_sample_guided", x_T, noise, cond, non_cond, guidance_w, guidance, steps, x0, x1, logp, B, seed,
                       workspace, workspace_bytes, stream);
}

int cld_guidance_step(cld_handle h, const float* mean, const float* cond, const cld_guidance* gd, float sigma, const float* z,
                      float* mean_guided, float* x_next, float* grad, int32_t B, void* workspace, size_t workspace_bytes,
                      void* stream) {
    int rc = check_common(h, "cld_guidance_step", B, 0, workspace, workspace_bytes);
    if (rc) return rc;
    if (!h->has_decoder) return fail(h, CLD_ERR_STATE, "cld_guidance_step: decoder weights not loaded");
    if (!mean || !cond || !gd || (x_next && sigma != 0.f && !z)) return fail(h, CLD_ERR_ARG, "cld_guidance_step: null pointer");
    if ((rc = check_guidance(h, "cld_guidance_step", gd, B)) != CLD_OK) return rc;
    Ws w = carve(workspace, pad16(B));
    return run_guidance(h, w, gd, B, true, sigma, sigma, mean, cond, z, 0, 0, mean_guided, x_next, nullptr, grad, static_cast<hipStream_t>(stream));
}

int cld_guidance_losses(cld_handle h, const float* traj, const cld_guidance* gd, float* losses, int32_t B, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!traj || !gd || !losses || B < 1) return fail(h, CLD_ERR_ARG, "cld_guidance_losses: bad argument");
    if (gd->target_pos_scale && (!gd->target_pos || !gd->target_time)) return fail(h, CLD_ERR_ARG, "cld_guidance_losses: target_pos_scale needs target_pos and target_time");
    GuideArgs g{};
    g.target_speed = gd->target_speed; g.loss_scale = gd->loss_scale;
    g.speed_limit_scale = gd->speed_limit_scale; g.acc_limit_scale = gd->acc_limit_scale;
    g.speed_limit = gd->speed_limit; g.acc_limit = gd->acc_limit;
    g.target_pos = gd->target_pos; g.target_time = gd->target_time; g.target_pos_scale = gd->target_pos_scale;
    g.B = B;
    HIPCK(h, launch_guide_losses(g, traj, losses, static_cast<hipStream_t>(stream)));
    return CLD_OK;
}

int cld_log_prob(cld_handle h, const float* x_t, const float* x_tm1, const float* cond, int32_t t_idx, float* out,
                 int32_t M, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_common(h, "cld_log_prob", M, t_idx, workspace, workspace_bytes);
    if (rc) return rc;
    if (!x_t || !x_tm1 || !cond || !out) return fail(h, CLD_ERR_ARG, "cld_log_prob: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int bp = pad16(M);
    Ws w = carve(workspace, bp);
    HIPCK(h, launch_pack_latent(x_t, w.xw, M, bp, s));
    HIPCK(h, launch_cond_bias(cond, h->wc, h->cbias_b, w.cb, M, bp, NCB, s));
    HIPCK(h, run_unet(h, w, w.xw, t_idx, bp, s));
    HeadArgs a{};
    head_source(h, w, a); a.w = h->head_w; a.b = h->head_b; a.x = w.xw; a.B = M; a.b_pad = bp;
    a.mean_out = w.meanb; a.xc = h->x_t_cof[t_idx]; a.nc = h->noise_cof[t_idx];
    HIPCK(h, launch_head(a, s));
    HIPCK(h, launch_logprob(x_tm1, w.meanb, std::exp(0.5f * h->plvc[t_idx]), out, M, s));
    return CLD_OK;
}

int cld_lstm_decode(cld_handle h, const float* z, const float* cond, float* act, int32_t B, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!h->finalized || !h->has_decoder) return fail(h, CLD_ERR_STATE, "cld_lstm_decode: decoder weights not loaded");
    if (!z || !cond || !act || B < 1) return fail(h, CLD_ERR_ARG, "cld_lstm_decode: bad argument");
    HIPCK(h, launch_decode(h->dec, h->dyn, z, cond, nullptr, act, nullptr, B, 1, static_cast<hipStream_t>(stream), h->force_kernel[CLD_KERNEL_DECODE]));
    return CLD_OK;
}

int cld_action_to_state(cld_handle h, const float* act, const float* curr_states, float* traj, int32_t B,
                        int32_t scaled_input, int32_t descaled_output, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!act || !curr_states || !traj || B < 1) return fail(h, CLD_ERR_ARG, "cld_action_to_state: bad argument");
    HIPCK(h, launch_action_to_state(h->dyn, act, curr_states, traj, B, scaled_input, descaled_output,
                                    static_cast<hipStream_t>(stream)));
    return CLD_OK;
}

int cld_decode(cld_handle h, const float* z, const float* cond, const float* curr_states, float* traj, float* act_out,
               int32_t B, int32_t descaled_output, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!h->finalized || !h->has_decoder) return fail(h, CLD_ERR_STATE, "cld_decode: decoder weights not loaded");
    if (!z || !cond || !curr_states || !traj || B < 1) return fail(h, CLD_ERR_ARG, "cld_decode: bad argument");
    HIPCK(h, launch_decode(h->dec, h->dyn, z, cond, curr_states, act_out, traj, B, descaled_output,
                           static_cast<hipStream_t>(stream), h->force_kernel[CLD_KERNEL_DECODE]));
    return CLD_OK;
}

int cld_traj2z(cld_handle h, const float* x6_scaled, const float* cond, const float* noise, float* z, float* mu,
               float* logvar, int32_t B, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!h->finalized || !h->has_encoder) return fail(h, CLD_ERR_STATE, "cld_traj2z: encoder weights not loaded");
    if (!x6_scaled || !cond || B < 1 || (!z && !mu && !logvar)) return fail(h, CLD_ERR_ARG, "cld_traj2z: bad argument");
    HIPCK(h, launch_encode(h->enc, x6_scaled, cond, noise, z, mu, logvar, B, static_cast<hipStream_t>(stream), h->force_kernel[CLD_KERNEL_ENCODE]));
    return CLD_OK;
}

int cld_vae_loss(cld_handle h, const float* x6_scaled, const float* act_out, const float* mu, const float* logvar, float beta,
                 float* out3, int32_t B, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!x6_scaled || !act_out || !mu || !logvar || !out3 || B < 1) return fail(h, CLD_ERR_ARG, "cld_vae_loss: bad argument");
    if (!workspace || workspace_bytes < (size_t)B * 2 * sizeof(float)) return fail(h, CLD_ERR_WORKSPACE, "cld_vae_loss: workspace too small");
    HIPCK(h, launch_vae_loss(x6_scaled, act_out, mu, logvar, beta, static_cast<float*>(workspace), out3, B,
                             static_cast<hipStream_t>(stream)));
    return CLD_OK;
}

int cld_state_to_state_and_action(cld_handle h, const float* positions, const float* yaws, const float* curr_speed,
                                  float* out6, int32_t B, int32_t scaled_output, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!positions || !yaws || !curr_speed || !out6 || B < 1)
        return fail(h, CLD_ERR_ARG, "cld_state_to_state_and_action: bad argument");
    HIPCK(h, launch_state_to_state_action(h->dyn, positions, yaws, curr_speed, out6, B, scaled_output,
                                          static_cast<hipStream_t>(stream)));
    return CLD_OK;
}

// ---- ContextEncoder ---------------------------------------------------------------------------------------
namespace {
constexpr int kCtxChunk = 256;                               // agents per pass: bounds the activation scratch
constexpr size_t kStemFloats = (size_t)112 * 112 * 64;       // per agent
constexpr size_t kActFloats = (size_t)56 * 56 * 64;          // largest post-pool activation per agent

// Agents per pass of the ContextEncoder.  The Winograd kernels (wino44_kernels.hip, 4x4 tiles) run equal workgroups of 16 tiles x 64 channels, two
// to a CU: n * (196 | 49 | 16 | 4) / 16 * (C / 64) at 56x56 / 28x28 / 14x14 / 7x7 -- at n = 256 that is 6.125 / 3.06 / 2 / 1 generations of 512; a last
// generation that is 6 % full costs a whole one.  The pass size is the one that minimises the modelled generation count of the whole batch (a
// generation of the four launch kinds lasts 1.5 / 2.7 / 4.3 / 8.5 units -- measured: 36 / 65 / 103 / 202 us --), with a fixed cost per pass for its
// 27 launches.  (With CLD_FORM_WINOGRAD_F2 the counts of wino_kernels.hip differ; the pass size is a speed choice only.)
int context_pass_size(int B) {
    if (B <= kCtxChunk) return B;
    auto cost = [](int n) {
        auto gens = [&](int tiles_per_agent, int ncb) { return (long)((n * tiles_per_agent + 15) / 16 * ncb + 511) / 512; };
        return 4 * gens(196, 1) * 15 + 3 * gens(49, 2) * 27 + 3 * gens(16, 4) * 43 + 3 * gens(4, 8) * 85 + 60;
    };
    int best = kCtxChunk;
    long best_cost = -1;
    for (int p = kCtxChunk; p >= 128; --p) {
        long c = (long)(B / p) * cost(p) + (B % p ? cost(B % p) : 0);
        if (best_cost < 0 || c < best_cost) { best_cost = c; best = p; }
    }
    return best;
}
}
size_t cld_context_workspace_bytes(cld_handle h, int32_t B) {
    if (!h || B < 1) return 0;
    const size_t cb = (size_t)(B < kCtxChunk ? B : kCtxChunk);
    return cb * (kStemFloats + 3 * kActFloats) * sizeof(float);
}

int cld_context_encode(cld_handle h, const float* image, const float* curr_states, float* cond_feat, float* map_feat,
                       int32_t B, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!h->finalized || !h->has_context) return fail(h, CLD_ERR_STATE, "cld_context_encode: context_encoder weights not loaded");
    if (!image || !curr_states || !cond_feat || B < 1) return fail(h, CLD_ERR_ARG, "cld_context_encode: bad argument");
    if (!workspace || workspace_bytes < cld_context_workspace_bytes(h, B))
        return fail(h, CLD_ERR_WORKSPACE, "cld_context_encode: workspace too small");
    if (reinterpret_cast<uintptr_t>(workspace) % 16 || reinterpret_cast<uintptr_t>(image) % 16)
        return fail(h, CLD_ERR_ARG, "cld_context_encode: image and workspace must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int cb = B < kCtxChunk ? B : kCtxChunk;
    float* y1 = static_cast<float*>(workspace);
    float* buf[3];
    for (int i = 0; i < 3; ++i) buf[i] = y1 + (size_t)cb * kStemFloats + (size_t)i * cb * kActFloats;
    const bool wino = h->force_kernel[CLD_KERNEL_CONTEXT] != CLD_FORM_DIRECT;
    const bool wino44 = h->force_kernel[CLD_KERNEL_CONTEXT] != CLD_FORM_WINOGRAD_F2;      // F(4x4, 3x3) where the map is whole 4x4 tiles (56x56, 28x28)
    auto run = [&](const cld_handle_s::Conv2dLayer& l, const float* x, const float* res, float* y, int relu, int n) {
        if (wino && wino44 && l.ufrag44) return launch_wino44_conv(l.hin, l.cout, WinoArgs{x, l.ufrag44, l.scale, l.shift, res, y, n, relu}, s);
        if (wino && l.ufrag) return launch_wino_conv(l.hin, l.cout, WinoArgs{x, l.ufrag, l.scale, l.shift, res, y, n, relu}, s);
        return launch_conv2d(l.kh, l.stride, l.hin, x, l.wfrag, l.scale, l.shift, res, y, n, l.cin, l.cout, relu, s);
    };
    const int pass = wino ? context_pass_size(B) : cb;
    for (int b0 = 0; b0 < B; b0 += pass) {
        const int n = (B - b0) < pass ? (B - b0) : pass;
        if (h->force_kernel[CLD_KERNEL_CONTEXT] == CLD_FORM_DIRECT) {      // the direct form of the encoder: stem and max-pool as two launches
            HIPCK(h, launch_stem_conv(image + (size_t)b0 * 34 * 224 * 224, h->stem_w, h->stem_scale, h->stem_shift, y1, nullptr, n, s));
            HIPCK(h, launch_maxpool(y1, buf[0], n, s));
        } else {                                                          // max-pool inside the stem's epilogue: [n,112,112,64] is never written
            HIPCK(h, launch_stem_conv(image + (size_t)b0 * 34 * 224 * 224, h->stem_w, h->stem_scale, h->stem_shift, nullptr, buf[0], n, s));
        }
        int xi = 0;                                            // buffer holding the current block input
        for (int li = 0; li < 4; ++li)
            for (int b = 0; b < 2; ++b) {                      // BasicBlock: relu(bn2(conv2(relu(bn1(conv1 x)))) + identity)
                const int ti = (xi + 1) % 3, yi = (xi + 2) % 3;
                HIPCK(h, run(h->rn_conv[li][b][0], buf[xi], nullptr, buf[ti], 1, n));
                if (b == 0 && li > 0) {                        // identity = bn(conv1x1/2 x); x's buffer is free afterwards
                    HIPCK(h, run(h->rn_ds[li], buf[xi], nullptr, buf[yi], 0, n));
                    HIPCK(h, run(h->rn_conv[li][b][1], buf[ti], buf[yi], buf[xi], 1, n));
                } else {
                    HIPCK(h, run(h->rn_conv[li][b][1], buf[ti], buf[xi], buf[yi], 1, n));
                    xi = yi;
                }
            }
        ContextHeadArgs a = h->ctx_head;
        a.feat = buf[xi]; a.curr_states = curr_states + (size_t)b0 * 4; a.cond_out = cond_feat + (size_t)b0 * 256;
        a.map_feat_out = map_feat ? map_feat + (size_t)b0 * 256 : nullptr; a.B = n;
        HIPCK(h, launch_context_head(a, s));
    }
    return CLD_OK;
}

int cld_context_combine(cld_handle h, const float* map_feat, int32_t broadcast, const float* curr_states, float* cond_feat,
                        int32_t B, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!h->finalized || !h->has_context) return fail(h, CLD_ERR_STATE, "cld_context_combine: context_encoder weights not loaded");
    if (!map_feat || !curr_states || !cond_feat || B < 1) return fail(h, CLD_ERR_ARG, "cld_context_combine: bad argument");
    ContextHeadArgs a = h->ctx_head;
    a.feat = nullptr; a.map_feat_in = map_feat; a.map_feat_stride = broadcast ? 0 : 256;
    a.curr_states = curr_states; a.cond_out = cond_feat; a.map_feat_out = nullptr; a.B = B;
    HIPCK(h, launch_context_head(a, static_cast<hipStream_t>(stream)));
    return CLD_OK;
}

int cld_compute_reward(cld_handle h, const float* traj, const float* traj_scaled, const float* raster_from_agent,
                       const uint8_t* drivable_map, int32_t H, int32_t W, const float* other_pos, const uint8_t* other_avail,
                       int32_t S, int32_t T_other, float collision_thresh, float* reward, float* offroad, float* collision,
                       int32_t B, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!traj || !raster_from_agent || !drivable_map || B < 1 || H < 1 || W < 1 || S < 0 || T_other < 0 ||
        (S > 0 && T_other > 0 && (!other_pos || !other_avail)) || (!reward && !offroad && !collision))
        return fail(h, CLD_ERR_ARG, "cld_compute_reward: bad argument");
    RewardArgs a{};
    a.traj = traj; a.traj_scaled = traj_scaled; a.raster_from_agent = raster_from_agent; a.drivable_map = drivable_map;
    a.other_pos = other_pos; a.other_avail = other_avail; a.reward = reward; a.offroad = offroad; a.collision = collision;
    a.collision_thresh = collision_thresh; a.B = B; a.H = H; a.W = W; a.S = S; a.To = T_other < T ? T_other : T;
    HIPCK(h, launch_reward(a, static_cast<hipStream_t>(stream)));
    return CLD_OK;
}

int cld_agent_collision(cld_handle h, const float* traj, const cld_collision* c, const float* grad_in, float* loss, float* grad,
                        int32_t B, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!traj || !c || B < 1 || (!loss && !grad)) return fail(h, CLD_ERR_ARG, "cld_agent_collision: bad argument");
    int rc = check_collision(h, "cld_agent_collision", c, B);
    if (rc) return rc;
    CollisionArgs ca{};
    ca.traj = traj; ca.extent = c->extent; ca.world_from_agent = c->world_from_agent; ca.curr_speed = c->curr_speed;
    ca.scene_start = c->scene_start; ca.scene_weight = c->scene_weight; ca.guided = c->guided; ca.excluded = c->excluded; ca.max_scene_agents = c->max_scene_agents; ca.grad_in = grad_in;
    ca.loss = loss; ca.grad = grad; ca.B_agents = B / c->num_samp; ca.num_scenes = c->num_scenes; ca.num_samp = c->num_samp;
    ca.num_disks = c->num_disks; ca.buffer_dist = c->buffer_dist; ca.decay_rate = c->decay_rate; ca.moving_speed_th = c->moving_speed_th;
    HIPCK(h, launch_agent_collision(ca, static_cast<hipStream_t>(stream)));
    return CLD_OK;
}

int cld_map_collision_loss(cld_handle h, const float* traj, const cld_map_collision* c, const float* grad_in, float* loss,
                           float* grad, int32_t B, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!traj || !c || B < 1 || (!loss && !grad)) return fail(h, CLD_ERR_ARG, "cld_map_collision_loss: bad argument");
    int rc = check_map_collision(h, "cld_map_collision_loss", c, B);
    if (rc) return rc;
    HIPCK(h, launch_map_collision(map_args(c, traj, grad_in, loss, grad), B, static_cast<hipStream_t>(stream)));
    return CLD_OK;
}

int cld_world_step(cld_handle h, const float* traj, const float* centroid, const float* yaw, int32_t k, float* world,
                   float* next_curr_states, int32_t B, void* stream) {
    if (!h) return CLD_ERR_ARG;
    if (!traj || !centroid || !yaw || !world || B < 1 || k < 0 || k >= T) return fail(h, CLD_ERR_ARG, "cld_world_step: bad argument");
    HIPCK(h, launch_world_step(traj, centroid, yaw, k, world, next_curr_states, B, static_cast<hipStream_t>(stream)));
    return CLD_OK;
}

}  // extern "C"
