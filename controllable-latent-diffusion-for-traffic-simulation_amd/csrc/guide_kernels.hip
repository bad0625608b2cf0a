// guide_kernels.hip -- sampling-time guidance of the latent posterior mean.
//
// CLD's DmModel has no sampling-time gradient; the definition is the vendored upstream
//   DiffuserModel.p_sample            src/tbsim/models/diffuser.py:844-929
//   PerturbationGuidance.perturb      src/tbsim/utils/guidance_loss.py:2221-2282   (its `decoder` hook, :2259-2261, is
//                                     where a latent model plugs its decoder in)
//   TargetSpeedLoss                   src/tbsim/utils/guidance_loss.py:219-254
// i.e. for every denoising step t > 0: decode the posterior mean mu (LSTM decoder -> descale -> unicycle roll-out),
// loss = weight * mean_agents mean_t |v_t - v_target_t|, ONE optimiser step on mu (Adam: delta = -lr g / (|g| + 1e-8);
// SGD: delta = -lr g), then x_{t-1} = mu' + sigma_t z.  Upstream also means to clip delta to +-perturb_th (sigma_t when
// None), but its perturb() computes the delta between two names of the SAME tensor (x_guidance = x_initial,
// guidance_loss.py:2239, then :2275-2278), so the clip never changes anything; a negative perturb_th reproduces that.
//
// guide_kernel: one 256-thread workgroup per agent.  Forward = the decode kernel's LSTM (thread r owns gate row r of
// both layers, weights in registers) with every gate activation and cell state written to an L2-resident scratch
// (133 KB per agent); the speed chain v_k = clip(v_0 + dt * sum clip(acc_j)) and the loss gradient are a 52-step scan;
// backward = BPTT through both layers with the TRANSPOSED recurrent matrices also register-resident (thread (j, part)
// holds 64 rows of column j), so no weight is re-read during the 52 steps.
#include "cld_kernels.h"

namespace cld {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_g(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int GT = 52;
constexpr int G_GATES = GT * 2 * 256;        // floats: post-activation gates [t][layer][256]
constexpr int G_CELLS = GT * 2 * 64;         // floats: cell states [t][layer][64]

__global__ __launch_bounds__(256) void guide_kernel(const DecoderWeights w, const DynParams d, const GuideArgs a) {
    __shared__ __attribute__((aligned(16))) float h0[64], h1[64], c0[64], c1[64], gates[256], zin[208], act[104];
    __shared__ __attribute__((aligned(16))) float condm[256];
    __shared__ float dact[GT];               // dL / d(scaled acceleration output)
    __shared__ float dgl[256];               // gate gradients of the layer being processed
    __shared__ float part[3][4][64];         // partial transposed products
    __shared__ float rec1[64], rec0[64], dh0l1[64], dc1n[64], dc0n[64];
    __shared__ float dz[208];
    const int r = threadIdx.x;
    const int gate = r >> 6;
    const int j = r & 63, pt = r >> 6;

    // forward weights: row r of each matrix
    float wi0[4], wh0[64], wi1[64], wh1[64];
#pragma unroll
    for (int k = 0; k < 4; ++k) wi0[k] = w.w_ih0[r * 4 + k];
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        wh0[k] = w.w_hh0[r * 64 + k];
        wi1[k] = w.w_ih1[r * 64 + k];
        wh1[k] = w.w_hh1[r * 64 + k];
    }
    // backward weights: column j, rows 64 pt .. 64 pt + 63
    float th1[64], ti1[64], th0[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        th1[k] = w.w_hh1[(64 * pt + k) * 64 + j];
        ti1[k] = w.w_ih1[(64 * pt + k) * 64 + j];
        th0[k] = w.w_hh0[(64 * pt + k) * 64 + j];
    }
    const float bias0 = w.b0[r], bias1 = w.b1[r];
    const float wa0 = (r < 64) ? w.w_h2a[r] : 0.f;          // d act[:, 0] / d h1[r]

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        float* sg = a.scratch + (size_t)blockIdx.x * (G_GATES + G_CELLS);
        float* sc = sg + G_GATES;
        condm[r] = a.cond[(size_t)b * 256 + r];
        if (r < 208) zin[r] = a.mean[(size_t)b * 208 + r];
        __syncthreads();
        if (r < 64) {
            float s = w.b_c2h[r];
            const float* wr = w.w_c2h + r * 256;
            for (int k = 0; k < 256; ++k) s = fmaf(condm[k], wr[k], s);
            h0[r] = s; h1[r] = s; c0[r] = 0.f; c1[r] = 0.f;
        }
        __syncthreads();
        // ---------------- forward (lstm_vae.py:44-52), activations kept ----------------
        for (int t = 0; t < GT; ++t) {
            float g = bias0;
#pragma unroll
            for (int k = 0; k < 4; ++k) g = fmaf(zin[4 * t + k], wi0[k], g);
#pragma unroll
            for (int k = 0; k < 64; ++k) g = fmaf(h0[k], wh0[k], g);
            g = (gate == 2) ? tanhf(g) : sigmoid_g(g);
            gates[r] = g;
            sg[(t * 2 + 0) * 256 + r] = g;
            __syncthreads();
            if (r < 64) {
                const float c = gates[64 + r] * c0[r] + gates[r] * gates[128 + r];
                c0[r] = c;
                sc[(t * 2 + 0) * 64 + r] = c;
                h0[r] = gates[192 + r] * tanhf(c);
            }
            __syncthreads();
            g = bias1;
#pragma unroll
            for (int k = 0; k < 64; ++k) g = fmaf(h0[k], wi1[k], g);
#pragma unroll
            for (int k = 0; k < 64; ++k) g = fmaf(h1[k], wh1[k], g);
            g = (gate == 2) ? tanhf(g) : sigmoid_g(g);
            gates[r] = g;
            sg[(t * 2 + 1) * 256 + r] = g;
            __syncthreads();
            if (r < 64) {
                const float c = gates[64 + r] * c1[r] + gates[r] * gates[128 + r];
                c1[r] = c;
                sc[(t * 2 + 1) * 64 + r] = c;
                h1[r] = gates[192 + r] * tanhf(c);
            }
            __syncthreads();
            if (r == 0) {   // hid2act, acceleration channel only (the speed loss does not see the yaw rate)
                float s = w.b_h2a[0];
                for (int k = 0; k < 64; ++k) s = fmaf(h1[k], w.w_h2a[k], s);
                act[t] = s;
            }
        }
        __syncthreads();
        // ---------------- speed chain + loss gradient (diffuser_helpers.py:573-600; guidance_loss.py:229-254) ----------------
        if (r == 0) {
            const float* cs = a.curr_states + (size_t)b * 4;
            const float* tgt = a.target_speed + (size_t)b * GT;
            const float scale = a.loss_scale ? a.loss_scale[b] : (1.0f / (float)GT);
            float v_raw = cs[2];
            float gv[GT];
            bool aok[GT];
            for (int t = 0; t < GT; ++t) {
                const float acc = act[t] * d.std[4] + d.mean[4];
                aok[t] = acc >= d.acc_lo && acc <= d.acc_hi;                 // clamp passes the gradient on [lo, hi]
                v_raw += fminf(fmaxf(acc, d.acc_lo), d.acc_hi) * d.dt;
                const bool vok = v_raw >= d.v_lo && v_raw <= d.v_hi;
                const float v = fminf(fmaxf(v_raw, d.v_lo), d.v_hi);
                const float df = v - tgt[t];
                const float sgn = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);   // d|x|/dx, 0 at 0 (and for NaN targets: nan_to_num)
                gv[t] = vok ? scale * sgn : 0.f;
            }
            float run = 0.f;
            for (int t = GT - 1; t >= 0; --t) {                               // v_k depends on every acc_j, j <= k
                run += gv[t];
                dact[t] = aok[t] ? run * d.dt * d.std[4] : 0.f;
            }
        }
        if (r < 64) { rec1[r] = 0.f; rec0[r] = 0.f; dc1n[r] = 0.f; dc0n[r] = 0.f; }
        __syncthreads();
        // ---------------- backward through time ----------------
        for (int t = GT - 1; t >= 0; --t) {
            // layer 1 gate gradients
            if (r < 64) {
                const float* gt = sg + (t * 2 + 1) * 256;
                const float ig = gt[r], fg = gt[64 + r], gg = gt[128 + r], og = gt[192 + r];
                const float c = sc[(t * 2 + 1) * 64 + r];
                const float cp = t > 0 ? sc[((t - 1) * 2 + 1) * 64 + r] : 0.f;
                const float tc = tanhf(c);
                const float dh = wa0 * dact[t] + rec1[r];
                const float dc = dh * og * (1.f - tc * tc) + dc1n[r];
                dgl[r] = dc * gg * ig * (1.f - ig);
                dgl[64 + r] = dc * cp * fg * (1.f - fg);
                dgl[128 + r] = dc * ig * (1.f - gg * gg);
                dgl[192 + r] = dh * tc * og * (1.f - og);
                dc1n[r] = dc * fg;
            }
            __syncthreads();
            {   // W_hh1^T dg1 (recurrent, for t-1) and W_ih1^T dg1 (into layer 0's h at t)
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int k = 0; k < 64; ++k) {
                    const float g = dgl[64 * pt + k];
                    s1 = fmaf(th1[k], g, s1);
                    s2 = fmaf(ti1[k], g, s2);
                }
                part[0][pt][j] = s1;
                part[1][pt][j] = s2;
            }
            __syncthreads();
            if (r < 64) {
                rec1[r] = part[0][0][r] + part[0][1][r] + part[0][2][r] + part[0][3][r];
                dh0l1[r] = part[1][0][r] + part[1][1][r] + part[1][2][r] + part[1][3][r];
            }
            __syncthreads();
            // layer 0 gate gradients
            if (r < 64) {
                const float* gt = sg + (t * 2 + 0) * 256;
                const float ig = gt[r], fg = gt[64 + r], gg = gt[128 + r], og = gt[192 + r];
                const float c = sc[(t * 2 + 0) * 64 + r];
                const float cp = t > 0 ? sc[((t - 1) * 2 + 0) * 64 + r] : 0.f;
                const float tc = tanhf(c);
                const float dh = dh0l1[r] + rec0[r];
                const float dc = dh * og * (1.f - tc * tc) + dc0n[r];
                dgl[r] = dc * gg * ig * (1.f - ig);
                dgl[64 + r] = dc * cp * fg * (1.f - fg);
                dgl[128 + r] = dc * ig * (1.f - gg * gg);
                dgl[192 + r] = dh * tc * og * (1.f - og);
                dc0n[r] = dc * fg;
            }
            __syncthreads();
            {   // W_hh0^T dg0 (recurrent) ; W_ih0^T dg0 = dL/dz_t (4 values: one wave each, lanes stride the 256 rows)
                float s1 = 0.f;
#pragma unroll
                for (int k = 0; k < 64; ++k) s1 = fmaf(th0[k], dgl[64 * pt + k], s1);
                part[2][pt][j] = s1;
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) s = fmaf(w.w_ih0[(j + 64 * q) * 4 + pt], dgl[j + 64 * q], s);
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
                if (j == 0) dz[4 * t + pt] = s;
            }
            __syncthreads();
            if (r < 64) rec0[r] = part[2][0][r] + part[2][1][r] + part[2][2][r] + part[2][3][r];
            __syncthreads();
        }
        // ---------------- one optimiser step on the mean, clipped; then the ancestral noise ----------------
        if (r < 208) {
            const float g = dz[r];
            float delta = (a.optimizer == 0) ? -a.lr * g / (fabsf(g) + 1e-8f) : -a.lr * g;      // Adam's first step | SGD
            if (a.perturb_th >= 0.f) delta = fminf(fmaxf(delta, -a.perturb_th), a.perturb_th);
            const float mu = zin[r] + delta;
            if (a.grad_out) a.grad_out[(size_t)b * 208 + r] = g;
            if (a.mean_out) a.mean_out[(size_t)b * 208 + r] = mu;
            if (a.x_out) {
                float zz = 0.f;
                if (a.sigma != 0.f) zz = a.z ? a.z[(size_t)b * 208 + r] : normal4(a.seed, a.step_salt, (unsigned)(b * 52 + (r >> 2)))[r & 3];
                const float xn = mu + a.sigma * zz;
                a.x_out[(size_t)b * 208 + r] = xn;
                if (a.x_out2) a.x_out2[(size_t)b * 208 + r] = xn;
            }
        }
        __syncthreads();
    }
}

size_t guide_scratch_floats(int B) {
    const int grid = B < 1024 ? B : 1024;
    return (size_t)grid * (G_GATES + G_CELLS);
}

hipError_t launch_guide(const DecoderWeights& w, const DynParams& d, const GuideArgs& a, hipStream_t s) {
    const int grid = a.B < 1024 ? a.B : 1024;
    hipLaunchKernelGGL(guide_kernel, dim3(grid), dim3(256), 0, s, w, d, a);
    return hipGetLastError();
}

}  // namespace cld

// =============================================================================================
// PPO reward of the reference (models/rl/criticmodel.py:7-64,88-145), one wave per agent, lane t = timestep t:
//   offroad   : trajectory point -> raster pixel (transform_points_tensor :101-112: p' = R[:2,:2] p + R[:2,2]), round half
//               to even (torch.round), clamp to the map, -1 per timestep on a non-drivable pixel              (:13-29)
//   collision : -1 per (other agent, timestep < T_other) closer than the threshold and available              (:42-64)
//   jerk      : 0.1 * mean_t |acc_{t+1} - acc_t| / dt on the SCALED acceleration channel                      (:33-37)
// =============================================================================================
namespace cld {

__global__ __launch_bounds__(64) void reward_kernel(const RewardArgs a) {
    const int b = blockIdx.x, t = threadIdx.x;
    float off = 0.f, col = 0.f, jerk = 0.f;
    if (t < 52) {
        const float* p = a.traj + ((size_t)b * 52 + t) * 6;
        const float x = p[0], y = p[1];
        const float* R = a.raster_from_agent + (size_t)b * 9;
        const float rx = x * R[0] + y * R[1] + R[2];          // bmm(points, R^T[:2,:2]) + R^T[2,:2]
        const float ry = x * R[3] + y * R[4] + R[5];
        long cx = (long)rintf(rx), cy = (long)rintf(ry);
        cx = cx < 0 ? 0 : (cx > a.W - 1 ? a.W - 1 : cx);
        cy = cy < 0 ? 0 : (cy > a.H - 1 ? a.H - 1 : cy);
        off = a.drivable_map[((size_t)b * a.H + cy) * a.W + cx] ? 0.f : -1.f;
        if (t < a.To) {
            for (int s = 0; s < a.S; ++s) {
                const size_t o = ((size_t)b * a.S + s) * a.To + t;
                const float dx = x - a.other_pos[2 * o], dy = y - a.other_pos[2 * o + 1];
                if (sqrtf(dx * dx + dy * dy) < a.collision_thresh && a.other_avail[o]) col -= 1.f;
            }
        }
        if (t < 51 && a.traj_scaled) {
            const float* q = a.traj_scaled + ((size_t)b * 52 + t) * 6;
            jerk = fabsf((q[6 + 4] - q[4]) / 0.1f);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { off += __shfl_xor(off, o); col += __shfl_xor(col, o); jerk += __shfl_xor(jerk, o); }
    if (t == 0) {
        const float jp = jerk * (1.0f / 51.0f);
        if (a.offroad) a.offroad[b] = off;
        if (a.collision) a.collision[b] = col;
        if (a.reward) a.reward[b] = off + col - jp * 0.1f;
    }
}

hipError_t launch_reward(const RewardArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(reward_kernel, dim3(a.B), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace cld
