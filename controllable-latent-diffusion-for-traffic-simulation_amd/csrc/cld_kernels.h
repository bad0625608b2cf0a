// Internal kernel interfaces of libcld_hip (not part of the public C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cld {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel AND device: `done_mask` is the launcher's static, one bit per device
// ordinal (a process-wide flag left the attribute unset on a second device)
inline hipError_t set_max_lds_once(const void* fn, int bytes, unsigned long long* done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((*done_mask >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) *done_mask |= 1ull << dev;
    return hipSuccess;
}

// ---------------------------------------------------------------------------
// Activation layout in HBM: channels-last [B_pad, L, C] fp32, B_pad = B rounded
// up to 16 agents.  All three U-Net resolutions hold 3,328 floats per agent at
// their widest (64x52 = 128x26 = 256x13).
//
// One conv workgroup owns MT = 208 output rows (16 / 8 / 4 whole agents at
// L = 13 / 26 / 52) x NT output channels and runs the implicit GEMM
//     out[r, n] = sum_{tap, ci} X[in_row(r) + tap, ci] * W[tap][ci][n]
// on v_mfma_f32_16x16x4_f32, 13 M-tiles x 1 N-tile per wave.  Two tilings exist per layer shape
// (conv_block.hip: A = 64 columns / 4 waves, B = 32 columns / 4 waves with a 2-way K split);
// pick_tiling() chooses by batch size so that every SIMD of the chip holds two waves.
// ---------------------------------------------------------------------------
constexpr int MT = 208;
constexpr int NMT = 13;

enum { EPI_BIAS = 0, EPI_GN_MISH = 1 };

struct ConvArgs {
    // sources: virtual input channels = [x1 (c1_pad, real c1_real) | x2 (c2)]
    const float* x1;
    const float* x2;
    int c1_real;     // channels actually present in x1 (row stride of x1)
    int c1_pad;      // c1_real rounded up to the K-chunk; chunks beyond c1_real read zeros
    int c2;          // channels of x2 (0 if none)
    const float* wfrag;   // MFMA-fragment-ordered weights (see pack_conv_weights)
    const float* wfrag_edge;   // Winograd launches: the 12-plane fragments of wino1d_edge.hip (xi 0..7 | taps 0..3), or null
    const float* bias;    // [c_out] conv bias
    const float* gamma;   // [c_out] GroupNorm weight (EPI_GN_MISH)
    const float* beta;    // [c_out] GroupNorm bias
    const float* cbias;   // per-agent vector added after Mish: cbias[b*cb_stride + n] (or null)
    int cb_stride;
    const float* tbias;   // per-step vector added after Mish: tbias[n] (or null)
    const float* res;     // residual tensor, same layout as y (or null)
    // residual computed in the epilogue from a 4-channel tensor instead (the first block: residual_conv of the latent):
    const float* res4_x;  // [B_pad, ly, 4] (or null)
    const float* res4_w;  // [c_out][4] Conv1d(4 -> c_out, k = 1) weight
    const float* res4_b;  // [c_out]
    float* y;             // output [B_pad, ly, c_out]
    int c_out;
    int ly;               // rows per agent of the OUTPUT tensor
    int off0;             // input row of tap 0 relative to STRIDE*j
    int orow0;            // output row = OSTR*j + orow0
    int xcd_map;          // set by the launcher: XCD-aware workgroup -> tile mapping (conv_block.hip tile_of_block)
    float wscale_inv;     // split-precision mode: 1 / (power-of-two scale applied to the weights before splitting)
    unsigned long long* stamps;   // diagnostic builds (-DCLD_STAMPS) only: 16 u64 per workgroup; null otherwise
};

// A launcher picks the template instance from the geometry; returns hipError_t.
struct ConvGeom {
    int l_in;     // input rows per agent
    int lm;       // GEMM rows per agent (output positions computed per agent)
    int stride;   // input row step per output position
    int ntaps;
    int kc;       // K chunk (channels staged per step): 32 or 64
    int nwn;      // N-tiles (16 columns) per workgroup
    int ks;       // K split across waves; the workgroup has nwn * ks waves
    int epi;
    int gs;       // GroupNorm group size (channels) = c_out / 8
    int ostr;     // output row stride (2 for the transposed conv halves)
    int padc;     // 1: the input has fewer real channels than one K chunk (the 4-channel latent)
    int ain;      // activation format of the input: 0 = fp32, 1 = S22 (fp16 hi/lo planes, split-precision loop)
    int aout;     // activation format of the output and of the residual
    int half;     // tile height HM: 0 = 208 GEMM rows per workgroup, 1 = 104, 2 = 52, 3 = 26 (exact-fp32 tilings B / D only; conv_block.hip)
};
hipError_t launch_conv(const ConvGeom& g, const ConvArgs& a, int b_pad, hipStream_t s);
bool conv_geom_supported(const ConvGeom& g);
// two independent convolutions with the same grid shape in ONE launch (blockIdx.z picks the role)
hipError_t launch_conv_pair(const ConvGeom& ga, const ConvArgs& a, const ConvGeom& gb, const ConvArgs& b, int b_pad, hipStream_t s);
bool conv_pair_supported(const ConvGeom& a, const ConvGeom& b);
// Conv1d(k5, pad 2) + GroupNorm + Mish [+ vectors] [+ residual] by Winograd F(4, 5) (wino1d_kernels.hip): the 256 -> 256 launches at
// L = 13 (256 -> 256, 128 -> 128, 128 -> 256, cat(256, 256) -> 128); a.wfrag = G g in pack_conv_weights layout with the 8 transform points as taps; exact-fp32 activations only
bool wino1d_supported(int l_in, int c1, int c2, int c_out);      // c1 | c2: channels of the first | second (concatenated) source
hipError_t launch_wino1d(const ConvArgs& a, int l_in, int b_pad, int item_form, hipStream_t s);      // item_form: 0 by size; 1 whole items, 2 whole items of eight waves, at every size (tests)
long wino1d_row_planes(int l_in, int c_out, int b_pad, int item_form);   // (GEMM rows x planes) a launch of that shape and size runs: x 2 C_in C_out = the FLOP its MFMAs execute
hipError_t launch_wino1d_edge(const ConvArgs& a, int l_in, int b_pad, bool k_split, hipStream_t s);      // wino1d_edge.hip; a.wfrag = the 12-plane fragments; k_split: eight waves per item
long wino1d_edge_row_planes(int l_in, int b_pad);
long wino1d_gemm_rows(int l_in, int b_pad);     // GEMM rows (64 per item, idle ones included) a launch runs its 8 transform-domain products over
void set_lds_floor(size_t bytes);      // experiments only: minimum dynamic LDS per conv launch (0 = off)

// ---------------------------------------------------------------------------
// layer chains (conv_chain.hip): runs of 64-channel layers in one launch, the tile resident in LDS from layer to layer
// ---------------------------------------------------------------------------
enum { CHAIN_RES_NONE = 0, CHAIN_RES_LATENT = 1 /* residual_conv(latent), evaluated in the epilogue */,
       CHAIN_RES_KEPT = 2 /* the block input kept by an earlier stage */, CHAIN_RES_TENSOR = 3 /* [b_pad, L, 64] in HBM */ };
struct ChainStage {
    const float* wfrag;   // pack_conv_weights layout (pack_latent_conv_weights for the latent's conv)
    const float* ufrag;   // k5 layers: the Winograd-domain filters G g (pack_conv_weights layout, the 8 transform points as taps), or null
    const float* bias;    // [64]
    const float* gamma;   // GroupNorm weight / bias (null for bias-only stages)
    const float* beta;
    const float* res;     // CHAIN_RES_TENSOR
    int cb_off;           // >= 0: columns of the time / cond vectors added after Mish; -1 none
    int res_kind;
    int keep;             // 1: this stage's output is a later stage's CHAIN_RES_KEPT residual
};
struct ChainHeadArgs {    // downs.0.0 + downs.0.1 + downs.0.2: latent [b_pad,52,4] -> y [b_pad,26,64]
    const float* x;
    ChainStage st[5];     // conv 4->64 | conv 64->64 (+ residual_conv(latent), kept) | conv | conv (+ kept) | Conv1d k3 s2
    const float* res4_w;  // residual_conv of block 0: [64][4], [64]
    const float* res4_b;
    const float* cbias;   // per-agent vectors [b_pad][cb_stride]
    int cb_stride;
    const float* tbias;   // per-step vector, or null (per-agent timesteps: folded into cbias)
    float* keep;          // b_pad * 3328 floats of per-thread spill (one activation buffer)
    float* y;
    unsigned long long* stamps;   // diagnostic builds (-DCLD_STAMPS) only: 16 u64 per workgroup; null otherwise
};
hipError_t launch_chain_head(const ChainHeadArgs& a, int b_pad, int agents_per_tile /* 4 | 1 */, hipStream_t s);
struct ChainTailArgs {    // ups.1.0's second conv + ups.1.1 + ups.1.2 + final_conv: x [b_pad,26,64] -> eps [b_pad,52,4]
    const float* x;       // output of ups.1.0's first conv
    ChainStage st[3];     // conv (+ residual tensor = residual_conv of the block input, kept) | conv | conv (+ kept)
    ChainStage up_even, up_odd;   // ConvTranspose1d as two 2-tap parity convolutions (wfrag, bias)
    ChainStage fin;       // final_conv.0
    const float* head_wfrag;      // final_conv.1 [4,64,1] as one 16-column N tile in pack_conv_weights layout (columns 4..15 zero)
    const float* head_b;          // [4]
    const float* cbias;
    int cb_stride;
    const float* tbias;
    float* keep;          // b_pad * 1792 floats of per-thread spill
    float* eps;           // [b_pad,52,4] noise prediction, or null when the update below consumes it
    // the DDPM update on the noise prediction, in the same launch (plain sampling steps: no CFG combine): mean = xc x - nc eps;
    // x' = mean + sg z (dm_model.py:144-163); upd_x null = off.  upd_x_out may alias upd_x (every element is read, then written,
    // by one lane).  Rows >= B are padding: no noise.
    const float* upd_x;   // [b_pad,52,4] the step's input latent
    const float* upd_z;   // [B,52,4] noise, or null (on-device generator keyed by seed / step_salt)
    float* upd_mean_out;  // [b_pad,52,4] or null
    float* upd_x_out;     // [b_pad,52,4] or null
    float xc, nc, sg;
    int B;
    unsigned long long seed, step_salt;
};
hipError_t launch_chain_tail(const ChainTailArgs& a, int b_pad, int agents_per_tile /* 4 | 1 */, hipStream_t s);
// the same two launches with their 64 -> 64 k5 layers in Winograd F(4, 5) form (chain_wino.hip; four-agent tiles; every k5 stage needs ufrag)
hipError_t launch_chain_head_wino(const ChainHeadArgs& a, int b_pad, int agents_per_tile /* 4 | 2 | 1 */, hipStream_t s);
hipError_t launch_chain_tail_wino(const ChainTailArgs& a, int b_pad, int agents_per_tile /* 4 | 2 | 1 */, hipStream_t s);
double chain_head_wino_exec_flop(int b_pad, int agents_per_tile);
double chain_tail_wino_exec_flop(int b_pad, int agents_per_tile);
// FLOP the MFMAs of a chain launch execute (2,048 per v_mfma_f32_16x16x4_f32, padded M-tiles and N columns included)
double chain_head_exec_flop(int b_pad, int agents_per_tile);
double chain_tail_exec_flop(int b_pad, int agents_per_tile);

// ---------------------------------------------------------------------------
// small kernels (misc_kernels.hip)
// ---------------------------------------------------------------------------
// x [B,52,4] -> xw [B_pad,52,4] (rows >= B zero-filled)
hipError_t launch_pack_latent(const float* x, float* xw, int B, int b_pad, hipStream_t s);
// cb[b, n] = bias[n] + sum_k mish(cond[b,k]) * wc[n,k]   (b < B; pad rows = 0) ; wc is [ncb][256]
hipError_t launch_cond_bias(const float* cond, const float* wc, const float* bias, float* cb,
                            int B, int b_pad, int ncb, hipStream_t s);
// per-agent timesteps: cb[b, :] += tb[t_idx[b], :]  (time half of every block's Linear(Mish([t_emb | cond])), temporal.py:146)
hipError_t launch_add_time_bias(float* cb, const float* tb, const int* t_idx, int n_timesteps, int B, int ncb, hipStream_t s);
// q_sample (dm_model.py:91-96): xw[b] = qs[t_b] * z0[b] + qs[n + t_b] * noise[b] (pad rows zero); optional copy to z_noisy [B,52,4]
hipError_t launch_q_sample(const float* z0, const float* noise, const int* t_idx, const float* qs, int n_timesteps, float* xw,
                           float* z_noisy, int B, int b_pad, hipStream_t s);
// VaeModel.compute_vae_loss forward: x6 [B,52,6] scaled input, act [B,52,2], mu / lv [B,52,4] -> out = (loss, recon, kld);
// part: 2 B floats of scratch
hipError_t launch_vae_loss(const float* x6, const float* act, const float* mu, const float* lv, float beta, float* part, float* out,
                           int B, hipStream_t s);
// out[b] = mean over (52 x 4) of (a - b)^2
hipError_t launch_mse_rows(const float* a, const float* b, float* out, int B, hipStream_t s);
// head: eps = W f + b (64 -> 4); mean = xc*x - nc*eps; x' = mean + sg*z
struct HeadArgs {
    const float* eps_in; // [B_pad,52,4]: the noise prediction itself (the tail chain of conv_chain.hip has applied final_conv.1), or null
    const float* eps_in_uncond;   // its unconditional half in CFG mode
    const float* f;      // [B_pad,52,64] (eps_in == null)
    const float* w;      // [4,64]
    const float* b;      // [4]
    const float* x;      // [B_pad,52,4] current latent
    const float* z;      // [B,52,4] noise or null
    float* eps_out;      // [B,52,4] or null
    float* mean_out;     // [B_pad,52,4] or null
    float* x_out;        // [B_pad,52,4] or null (may alias x)
    float xc, nc, sg;
    float cfg_w;         // classifier-free guidance: eps = (1 + w) eps_cond - w eps_uncond when f_uncond != null
    const float* f_uncond;   // [B_pad,52,64] final-block activations of the unconditional pass, or null
    float* x_out2;       // second copy of x' (the unconditional half of the next step's 2B batch), or null
    unsigned long long seed;   // on-device RNG (z == null and sg != 0)
    unsigned long long step_salt;
    int B, b_pad;
};
hipError_t launch_head(const HeadArgs& a, hipStream_t s);
// out[b] = mean_{t,d} log N(xq[b]; mean[b], sigma)
hipError_t launch_logprob(const float* xq, const float* mean, float sigma, float* out, int B, hipStream_t s);
// copy [B,52,4] rows out of a padded buffer
hipError_t launch_unpack(const float* xw, float* x, int B, hipStream_t s);

struct DecoderWeights {   // device pointers, reference layouts
    const float *w_ih0, *w_hh0, *b0;   // [256,4] [256,64] [256] (b_ih + b_hh)
    const float *w_ih1, *w_hh1, *b1;   // [256,64] [256,64] [256]
    const float *w_c2h, *b_c2h;        // [64,256] [64]
    const float *w_h2a, *b_h2a;        // [2,64] [2]
    // re-layouts made at cld_finalize for the MFMA guidance kernel (null when the decoder is absent):
    const float* gfrag;                // [wave 8][tile 2][k-group 16][lane 64][4]: B fragments of the transposed (backward) products
    const float* gqfrag;               // [unit group 4][layer 2][k-group 32][lane 64][4]: B operands of guide_quad_kernel's backward products
};
struct DynParams {
    float dt, acc_lo, acc_hi, v_lo, v_hi, max_steer, max_yawvel;
    float mean[6], std[6];
};
// Kernel formulation of the three recurrent kernels (decode / encode / guide): picked by batch size unless a test forces one
// through cld_debug_force_kernel.
enum { FORM_AUTO = 0, FORM_VALU = 1, FORM_MFMA = 2, FORM_MFMA_QUAD = 3 /* guide kernel only: 8 agents per workgroup on the 4x4x1 MFMA */ };
// z [B,52,4], cond [B,256] -> act [B,52,2] (optional) ; if cs != null also traj [B,52,6]
hipError_t launch_decode(const DecoderWeights& w, const DynParams& d, const float* z, const float* cond,
                         const float* cs, float* act, float* traj, int B, int descaled_output, hipStream_t s, int form = FORM_AUTO);
hipError_t launch_action_to_state(const DynParams& d, const float* act, const float* cs, float* traj,
                                  int B, int scaled_input, int descaled_output, hipStream_t s);


// ---- VAE encoder (models/vae/lstm_vae.py:6-26,87-99) ---------------------------------------------------
struct EncoderWeights {   // device pointers, reference layouts
    const float *w_ih0, *w_hh0, *b0;   // [256,6] [256,64] [256] (b_ih + b_hh)
    const float *w_ih1, *w_hh1, *b1;   // [256,64] [256,64] [256]
    const float *w_c2h, *b_c2h;        // [64,256] [64]
    const float *w_mu, *b_mu;          // [4,64] [4]
    const float *w_lv, *b_lv;          // [4,64] [4]
};
// x6 [B,52,6] (scaled state+action), cond [B,256], noise [B,52,4] or null -> z, mu, logvar [B,52,4] (any may be null)
hipError_t launch_encode(const EncoderWeights& w, const float* x6, const float* cond, const float* noise,
                         float* z, float* mu, float* logvar, int B, hipStream_t s, int form = FORM_AUTO);
// positions [B,52,2], yaws [B,52,1], curr_speed [B] -> [B,52,6] = (x, y, v, yaw, acc, yaw-rate), optionally scaled
hipError_t launch_state_to_state_action(const DynParams& d, const float* pos, const float* yaw, const float* speed,
                                        float* out6, int B, int scaled_output, hipStream_t s);

// closed-loop world update (src/tbsim/envs/env_trajdata.py:452-468): state k of the planned trajectory, given in the
// agent frame at planning time, placed in the world: xy' = p_k @ [[c, s], [-s, c]] + centroid, h' = yaw + yaw_k.
// traj [B,52,6]; centroid [B,2]; yaw [B]; out world [B,3] = (x, y, h) and next curr_states [B,4] = (0, 0, v_k, 0).
hipError_t launch_world_step(const float* traj, const float* centroid, const float* yaw, int k, float* world,
                             float* next_cs, int B, hipStream_t s);

// ---- ContextEncoder (models/context_utils.py:8-61; context_kernels.hip) ------------------------------------
// stem: image [B,34,224,224] NCHW -> y [B,112,112,64] NHWC = ReLU(BN(conv 7x7/2)); wq: packed by pack_stem_weights
// pooled != null: MaxPool2d(3, 2, 1) fused (y unused): pooled [B,56,56,64] NHWC, zero-filled by the launcher, completed by atomic max
hipError_t launch_stem_conv(const float* image, const float* wq, const float* scale, const float* shift, float* y, float* pooled, int B,
                            hipStream_t s);
// MaxPool2d(3, 2, 1): [B,112,112,64] -> [B,56,56,64] (NHWC)
hipError_t launch_maxpool(const float* x, float* y, int B, hipStream_t s);
// NHWC conv (kh = 3: pad 1, stride 1 | 2; kh = 1: stride 2) + folded BatchNorm [+ residual] [+ ReLU];
// hin = input height = width in {56, 28, 14, 7}; wfrag in pack_conv_weights layout (tap = kh * KW + kw)
hipError_t launch_conv2d(int kh, int stride, int hin, const float* x, const float* wfrag, const float* scale,
                         const float* shift, const float* res, float* y, int B, int cin, int cout, int relu, hipStream_t s);
// 3x3 / stride 1 / pad 1 convolution with C_in = C_out by Winograd F(2x2, 3x3) (wino_kernels.hip) + folded BatchNorm [+ residual]
// [+ ReLU]; ufrag: U = G g G^T in pack_conv_weights layout with the 16 transform-domain positions as "taps"; B <= 256
struct WinoArgs {
    const float* x;        // [B, H, H, C] NHWC
    const float* ufrag;
    const float* scale;    // [C] folded BatchNorm
    const float* shift;
    const float* res;      // [B, H, H, C] or null
    float* y;              // [B, H, H, C]
    int B, relu;
};
hipError_t launch_wino_conv(int hin, int channels, const WinoArgs& a, hipStream_t s);
// the same layers at 56x56 / 28x28 by F(4x4, 3x3) (wino44_kernels.hip): a.ufrag = the 36-plane fragments (Conv2dLayer::ufrag44)
bool wino44_supported(int hin, int channels);
hipError_t launch_wino44_conv(int hin, int channels, const WinoArgs& a, hipStream_t s);
struct ContextHeadArgs {
    const float* feat;          // [B,7,7,512] NHWC, layer4 output
    const float* curr_states;   // [B,4]
    float* cond_out;            // [B,256]
    float* map_feat_out;        // [B,256] or null (diagnostic tap: the fc output)
    const float* map_feat_in;   // or null: skip pool + fc and take the map feature from here, row b at b * map_feat_stride
    int map_feat_stride;        // 256, or 0 to broadcast one row to every agent
    int B;
    const float *fc_wt, *fc_b;  // transposed weights [in][out] throughout
    const float *s_wt[3], *s_b[3], *s_g[2], *s_be[2];     // agent_state_encoder: 4 -> 64 -> 64 -> 64
    const float *c_wt[5], *c_b[5], *c_g[4], *c_be[4];     // process_cond_mlp: 320 -> 320 -> 320 -> 256 -> 256 -> 256
};
hipError_t launch_context_head(const ContextHeadArgs& a, hipStream_t s);

// ---- sampling-time guidance (guide_kernels.hip; upstream diffuser.py:844-929, guidance_loss.py:219-254,2221-2282) ----
struct GuideArgs {
    const float* mean;          // [>=B,52,4] posterior mean of this step
    const float* cond;          // [B,256]
    const float* curr_states;   // [B,4]
    const float* target_speed;  // [B,52] or null (term off)
    const float* loss_scale;    // [B] or null: d(total loss)/d(sum_t |v_t - target_t|) per agent; null -> 1/52
    const float* speed_limit_scale;   // [B] or null (term off): weight of sum_t relu(|v_t| - speed_limit)
    const float* acc_limit_scale;     // [B] or null (term off): weight of sum_t relu(|acc_t| - acc_limit)
    float speed_limit, acc_limit;
    const float* target_pos;          // [B,2] waypoint in the agent frame, or null
    const int* target_time;           // [B] index (0..51) of the trajectory state that should hit it
    const float* target_pos_scale;    // [B] or null (term off): weight of |pos[target_time] - target_pos|
    const float* ext_grad;            // [B,52,6] or null: dL/dtraj of a loss evaluated elsewhere on the descaled trajectory
    const float* z;             // [B,52,4] N(0,1) draw or null (on-device generator)
    float* mean_out;            // guided mean [B,52,4] or null
    float* x_out;               // guided mean + sigma z, [>=B,52,4] or null
    float* x_out2;              // second copy (the unconditional half in CFG mode) or null
    float* grad_out;            // dL/dmean [B,52,4] or null (diagnostic / tests)
    float* act_out;             // launch_guide_forward only: scaled decoder actions [B,52,2]; the kernel stops behind its forward sweep
    const float* act_in;        // launch_guide with guide_split_available(): the actions a launch_guide_forward of the SAME mean, scratch and B wrote --
                                // the kernel then runs its backward half only, from the activations that launch kept in `scratch`
    float* scratch;             // guide_scratch_floats(B) floats
    float lr, perturb_th, sigma;
    int optimizer;              // 0 = Adam, 1 = SGD
    // optimiser steps beyond the first (upstream grad_steps > 1: one torch.optim.Adam / SGD per perturb() call, its state carried
    // across the steps): step opt_step (1-based) of opt_steps; `mean` is then the CURRENT iterate, mean0 the posterior mean the
    // clip is taken around (null: mean), adam_m / adam_v [B,52,4] the moments (read when opt_step > 1, written when more steps follow)
    int opt_step, opt_steps;
    const float* mean0;
    float* adam_m;
    float* adam_v;
    int B;
    unsigned long long seed, step_salt;
};
size_t guide_scratch_floats(int B);
void read_guide_stamps(unsigned long long* out);     // -DCLD_STAMPS builds: 8 shader-clock stamps per workgroup (256 workgroups); else a no-op
hipError_t launch_guide(const DecoderWeights& w, const DynParams& d, const GuideArgs& a, hipStream_t s, int form = FORM_AUTO);
bool guide_forward_available(int B, int form = FORM_AUTO);
bool guide_split_available(int B, int form = FORM_AUTO);      // ... and every agent group has a workgroup (and a scratch slot) of its own: forward and backward may be two launches
hipError_t launch_guide_forward(const DecoderWeights& w, const DynParams& d, const GuideArgs& a, hipStream_t s);

// per-agent values of the built-in guidance losses on a decoded trajectory (include/cld.h cld_guidance_losses); uses the loss
// fields of GuideArgs (target_speed, loss_scale, speed/acc limits, waypoint) and B
hipError_t launch_guide_losses(const GuideArgs& a, const float* traj, float* losses, hipStream_t s);

// upstream's AgentCollisionLoss + its gradient w.r.t. the decoded plans (collision_kernels.hip; guidance_loss.py:442-630)
struct CollisionArgs {
    const float* traj;              // [B_agents * num_samp, 52, 6] descaled plans, sample-minor
    const float* extent;            // [B_agents, 3] length, width, height
    const float* world_from_agent;  // [B_agents, 3, 3]
    const float* curr_speed;        // [B_agents]
    const int* scene_start;         // [num_scenes + 1] agent offsets of the scenes
    const float* scene_weight;      // [num_scenes] weight of the scene's agent_collision config (0: not guided) or null (1 everywhere)
    const unsigned char* guided;    // [B_agents] or null: the config's `agents` subset
    const unsigned char* excluded;  // [B_agents] or null: upstream's `excluded_agents` -- a pair whose agents are BOTH flagged is not penalised (:586-593)
    const float* grad_in;           // [B_agents * num_samp, 52, 6] or null: added to the output gradient
    float* loss;                    // [B_agents * num_samp] per-agent values (unweighted, as upstream files them) or null
    float* grad;                    // [B_agents * num_samp, 52, 6] d total / d traj, or null
    int B_agents, num_scenes, num_samp, num_disks;
    int max_scene_agents;           // what the launch sized its LDS and grid for: a scene with more agents (scene_start is device data the host
                                    // cannot check) gets NaN values and a gradient of grad_in (or 0) instead of an overrun
    float buffer_dist, decay_rate, moving_speed_th;
};
hipError_t launch_agent_collision(const CollisionArgs& a, hipStream_t s);

// upstream's MapCollisionLoss + its gradient w.r.t. the decoded plans (collision_kernels.hip; guidance_loss.py:717-875)
struct MapCollisionArgs {
    const float* traj;               // [rows = B_agents * num_samp, 52, 6]
    const float* extent;             // [B_agents, 3]
    const float* raster_from_agent;  // [B_agents, 3, 3]
    const unsigned char* drivable_map;   // [B_agents, H, W], != 0 = drivable
    const float* curr_speed;         // [B_agents]
    const int* scene_start;          // [num_scenes + 1]
    const float* scene_weight;       // [num_scenes] or null
    const float* grad_in;            // [rows, 52, 6] or null (may alias grad)
    float* loss;                     // [rows] or null
    float* grad;                     // [rows, 52, 6] or null
    int num_scenes, num_samp, H, W, num_points_l, num_points_w;
    float decay_rate, moving_speed_th;
};
hipError_t launch_map_collision(const MapCollisionArgs& a, int rows, hipStream_t s);

// PPO reward (models/rl/criticmodel.py:7-64)
struct RewardArgs {
    const float* traj;                // [B,52,6] descaled (x, y, v, yaw, acc, yaw-rate), agent frame
    const float* traj_scaled;         // [B,52,6] scaled (jerk term) or null
    const float* raster_from_agent;   // [B,3,3]
    const unsigned char* drivable_map;   // [B,H,W] bool
    const float* other_pos;           // [B,S,To,2]
    const unsigned char* other_avail; // [B,S,To] bool
    float* reward; float* offroad; float* collision;   // [B] each, any may be null
    float collision_thresh;
    int B, H, W, S, To;
};
hipError_t launch_reward(const RewardArgs& a, hipStream_t s);

#ifdef __HIPCC__
// counter-based N(0,1) for throughput runs without caller noise (splitmix64 -> Box-Muller); row = one (agent, step) float4
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float u01(unsigned long long bits) {
    return ((float)(bits >> 40) + 0.5f) * (1.0f / 16777216.0f);
}
typedef float rng_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rng_v4f normal4(unsigned long long seed, unsigned long long step_salt, unsigned row) {
    const unsigned long long k = splitmix64(seed ^ splitmix64(step_salt * 0x100000000ull + row));
    const unsigned long long r0 = splitmix64(k), r1 = splitmix64(k + 1), r2 = splitmix64(k + 2), r3 = splitmix64(k + 3);
    const float m0 = sqrtf(-2.0f * logf(u01(r0))), m1 = sqrtf(-2.0f * logf(u01(r2)));
    float s0, c0, s1, c1;
    sincosf(6.283185307179586f * u01(r1), &s0, &c0);
    sincosf(6.283185307179586f * u01(r3), &s1, &c1);
    return rng_v4f{m0 * c0, m0 * s0, m1 * c1, m1 * s1};
}
#endif

}  // namespace cld
