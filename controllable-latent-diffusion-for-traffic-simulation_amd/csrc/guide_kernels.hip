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
constexpr int GNA = 2;                       // agents per workgroup: the register-resident weights are reused across them
                                             // and their 4 independent FMA chains fill the VALU pipeline

// g[ag] += sum_k src[ag][off + k] * wgt[k], k < 64, for the GNA agents of a workgroup.  The LDS operands are fetched as
// float4 one block ahead and pinned there (sched_barrier): left alone, the compiler hoists all 64 x GNA loads to the top,
// which together with the register-resident weights overflows the register file into scratch.
template <int STRIDE>
__device__ __forceinline__ void mv64(float (&g)[GNA], const float* src, int off, const float (&wgt)[64]) {
    v4f cur[GNA], nxt[GNA];
#pragma unroll
    for (int ag = 0; ag < GNA; ++ag) cur[ag] = *reinterpret_cast<const v4f*>(src + ag * STRIDE + off);
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
        if (k4 + 1 < 16) {
#pragma unroll
            for (int ag = 0; ag < GNA; ++ag) nxt[ag] = *reinterpret_cast<const v4f*>(src + ag * STRIDE + off + 4 * (k4 + 1));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ag = 0; ag < GNA; ++ag) g[ag] = fmaf(cur[ag][e], wgt[4 * k4 + e], g[ag]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ag = 0; ag < GNA; ++ag) cur[ag] = nxt[ag];
    }
}

__global__ __launch_bounds__(256) void guide_kernel(const DecoderWeights w, const DynParams d, const GuideArgs a) {
    constexpr int NA = GNA;
    __shared__ __attribute__((aligned(16))) float h0[NA][64], h1[NA][64], c0[NA][64], c1[NA][64], gates[NA][256], zin[NA][208];
    __shared__ __attribute__((aligned(16))) float condm[NA][256];
    __shared__ float act[NA][GT];            // scaled acceleration output of the decoder
    __shared__ float dact[NA][GT];           // dL / d(scaled acceleration output)
    __shared__ __attribute__((aligned(16))) float dgl[NA][256];           // gate gradients of the layer being processed
    __shared__ float part[3][NA][4][64];     // partial transposed products
    __shared__ float rec1[NA][64], rec0[NA][64], dh0l1[NA][64], dc1n[NA][64], dc0n[NA][64];
    __shared__ float dz[NA][208];
    const int r = threadIdx.x;
    const int gate = r >> 6;
    const int j = r & 63, pt = r >> 6;       // matvec phases: (column, row block); per-cell phases: unit j of agents pt, pt + 4, ...

    const float bias0 = w.b0[r], bias1 = w.b1[r];
    const float wa0 = w.w_h2a[j];                            // d act[:, 0] / d h1[j]
    const float bh2a = w.b_h2a[0];

    const int ngroups = (a.B + NA - 1) / NA;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int b0 = grp * NA;
        float* sgb = a.scratch + (size_t)blockIdx.x * NA * (G_GATES + G_CELLS);
        auto SG = [&](int ag) { return sgb + (size_t)ag * (G_GATES + G_CELLS); };
        auto SC = [&](int ag) { return sgb + (size_t)ag * (G_GATES + G_CELLS) + G_GATES; };
        auto agent = [&](int ag) { return (b0 + ag < a.B) ? b0 + ag : a.B - 1; };     // tail slots replay the last agent; never stored
#pragma unroll
        for (int ag = 0; ag < NA; ++ag) {
            condm[ag][r] = a.cond[(size_t)agent(ag) * 256 + r];
            if (r < 208) zin[ag][r] = a.mean[(size_t)agent(ag) * 208 + r];
        }
        __syncthreads();
        for (int ag = pt; ag < NA; ag += 4) {   // h0 = cond2hidden(cond): thread (unit j, agent ag)
            float s = w.b_c2h[j];
            const float* wr = w.w_c2h + j * 256;
            for (int k = 0; k < 256; ++k) s = fmaf(condm[ag][k], wr[k], s);
            h0[ag][j] = s; h1[ag][j] = s; c0[ag][j] = 0.f; c1[ag][j] = 0.f;
        }
        __syncthreads();
        // ---------------- forward (lstm_vae.py:44-52), activations kept ----------------
        {
        // forward weights: row r of each matrix, register-resident for the 52 steps (re-read from L2 per agent group:
        // forward and backward sets together do not fit the register file)
        float wi0[4], wh0[64], wi1[64], wh1[64];
        int rr = r;
        asm volatile("" : "+v"(rr));          // opaque per iteration: keeps these loads inside the group loop (not hoisted next to the backward set)
#pragma unroll
        for (int k = 0; k < 4; ++k) wi0[k] = w.w_ih0[rr * 4 + k];
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            wh0[k] = w.w_hh0[rr * 64 + k];
            wi1[k] = w.w_ih1[rr * 64 + k];
            wh1[k] = w.w_hh1[rr * 64 + k];
        }
        for (int t = 0; t < GT; ++t) {
            float g[NA];
#pragma unroll
            for (int ag = 0; ag < NA; ++ag) g[ag] = bias0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int ag = 0; ag < NA; ++ag) g[ag] = fmaf(zin[ag][4 * t + k], wi0[k], g[ag]);
            mv64<64>(g, &h0[0][0], 0, wh0);
#pragma unroll
            for (int ag = 0; ag < NA; ++ag) {
                const float v = (gate == 2) ? tanhf(g[ag]) : sigmoid_g(g[ag]);
                gates[ag][r] = v;
                SG(ag)[(t * 2 + 0) * 256 + r] = v;
            }
            __syncthreads();
            for (int ag = pt; ag < NA; ag += 4) {
                const float c = gates[ag][64 + j] * c0[ag][j] + gates[ag][j] * gates[ag][128 + j];
                c0[ag][j] = c;
                SC(ag)[(t * 2 + 0) * 64 + j] = c;
                h0[ag][j] = gates[ag][192 + j] * tanhf(c);
            }
            __syncthreads();
#pragma unroll
            for (int ag = 0; ag < NA; ++ag) g[ag] = bias1;
            mv64<64>(g, &h0[0][0], 0, wi1);
            mv64<64>(g, &h1[0][0], 0, wh1);
#pragma unroll
            for (int ag = 0; ag < NA; ++ag) {
                const float v = (gate == 2) ? tanhf(g[ag]) : sigmoid_g(g[ag]);
                gates[ag][r] = v;
                SG(ag)[(t * 2 + 1) * 256 + r] = v;
            }
            __syncthreads();
            for (int ag = pt; ag < NA; ag += 4) {
                const float c = gates[ag][64 + j] * c1[ag][j] + gates[ag][j] * gates[ag][128 + j];
                c1[ag][j] = c;
                SC(ag)[(t * 2 + 1) * 64 + j] = c;
                h1[ag][j] = gates[ag][192 + j] * tanhf(c);
            }
            __syncthreads();
            for (int ag = pt; ag < NA; ag += 4) {   // hid2act, acceleration channel only (the speed loss does not see the yaw
                float s = h1[ag][j] * wa0;          // rate): one wave per agent, lane j holds unit j
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
                if (j == 0) act[ag][t] = s + bh2a;
            }
        }
        }
        __syncthreads();
        // ---------------- speed chain + loss gradient (diffuser_helpers.py:573-600; guidance_loss.py:229-254) ----------------
        if (r < NA) {
            const int b = agent(r);
            const float* cs = a.curr_states + (size_t)b * 4;
            const float* tgt = a.target_speed + (size_t)b * GT;
            const float scale = a.loss_scale ? a.loss_scale[b] : (1.0f / (float)GT);
            float v_raw = cs[2];
            for (int t = 0; t < GT; ++t) {
                const float acc = act[r][t] * d.std[4] + d.mean[4];
                v_raw += fminf(fmaxf(acc, d.acc_lo), d.acc_hi) * d.dt;
                const bool vok = v_raw >= d.v_lo && v_raw <= d.v_hi;         // clamp passes the gradient on [lo, hi]
                const float v = fminf(fmaxf(v_raw, d.v_lo), d.v_hi);
                const float df = v - tgt[t];
                const float sgn = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);   // d|x|/dx, 0 at 0 (and for NaN targets: nan_to_num)
                dact[r][t] = vok ? scale * sgn : 0.f;                        // dL/dv_t for now
            }
            float run = 0.f;
            for (int t = GT - 1; t >= 0; --t) {                               // v_k depends on every acc_j, j <= k
                run += dact[r][t];
                const float acc = act[r][t] * d.std[4] + d.mean[4];
                dact[r][t] = (acc >= d.acc_lo && acc <= d.acc_hi) ? run * d.dt * d.std[4] : 0.f;
            }
        }
        for (int ag = pt; ag < NA; ag += 4) { rec1[ag][j] = 0.f; rec0[ag][j] = 0.f; dc1n[ag][j] = 0.f; dc0n[ag][j] = 0.f; }
        __syncthreads();
        // ---------------- backward through time: thread (unit j, agent pt) owns one cell ----------------
        // transposed recurrent weights: column j, rows 64 pt .. 64 pt + 63
        float th1[64], ti1[64], th0[64];
        int jj = j;
        asm volatile("" : "+v"(jj));
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            th1[k] = w.w_hh1[(64 * pt + k) * 64 + jj];
            ti1[k] = w.w_ih1[(64 * pt + k) * 64 + jj];
            th0[k] = w.w_hh0[(64 * pt + k) * 64 + jj];
        }
        // the kept activations of step t are fetched from the L2-resident scratch one step ahead of their use
        constexpr int NCELL = (NA + 3) / 4;          // cells (agents) per thread in the per-cell phases
        float pg[NCELL][2][4], pc[NCELL][2], pcp[NCELL][2];
        auto fetch = [&](int t) {
#pragma unroll
            for (int q = 0; q < NCELL; ++q) {
                const int ag = pt + 4 * q;
                if (ag < NA) {
#pragma unroll
                    for (int l = 0; l < 2; ++l) {
                        const float* gt = SG(ag) + (t * 2 + l) * 256;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pg[q][l][e] = gt[64 * e + j];
                        pc[q][l] = SC(ag)[(t * 2 + l) * 64 + j];
                        pcp[q][l] = t > 0 ? SC(ag)[((t - 1) * 2 + l) * 64 + j] : 0.f;
                    }
                }
            }
        };
        fetch(GT - 1);
        for (int t = GT - 1; t >= 0; --t) {
            float cg[NCELL][2][4], cc[NCELL][2], ccp[NCELL][2];
#pragma unroll
            for (int q = 0; q < NCELL; ++q)
#pragma unroll
                for (int l = 0; l < 2; ++l) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) cg[q][l][e] = pg[q][l][e];
                    cc[q][l] = pc[q][l]; ccp[q][l] = pcp[q][l];
                }
            if (t > 0) fetch(t - 1);
#pragma unroll
            for (int q = 0; q < NCELL; ++q) {   // layer 1 gate gradients
                const int ag = pt + 4 * q;
                if (ag < NA) {
                    const float ig = cg[q][1][0], fg = cg[q][1][1], gg = cg[q][1][2], og = cg[q][1][3];
                    const float c = cc[q][1], cp = ccp[q][1];
                    const float tc = tanhf(c);
                    const float dh = wa0 * dact[ag][t] + rec1[ag][j];
                    const float dc = dh * og * (1.f - tc * tc) + dc1n[ag][j];
                    dgl[ag][j] = dc * gg * ig * (1.f - ig);
                    dgl[ag][64 + j] = dc * cp * fg * (1.f - fg);
                    dgl[ag][128 + j] = dc * ig * (1.f - gg * gg);
                    dgl[ag][192 + j] = dh * tc * og * (1.f - og);
                    dc1n[ag][j] = dc * fg;
                }
            }
            __syncthreads();
            {   // W_hh1^T dg1 (recurrent, for t-1) and W_ih1^T dg1 (into layer 0's h at t): thread (column j, row block pt)
                float s1[NA], s2[NA];
#pragma unroll
                for (int ag = 0; ag < NA; ++ag) { s1[ag] = 0.f; s2[ag] = 0.f; }
                mv64<256>(s1, &dgl[0][0], 64 * pt, th1);
                mv64<256>(s2, &dgl[0][0], 64 * pt, ti1);
#pragma unroll
                for (int ag = 0; ag < NA; ++ag) { part[0][ag][pt][j] = s1[ag]; part[1][ag][pt][j] = s2[ag]; }
            }
            __syncthreads();
            for (int ag = pt; ag < NA; ag += 4) {
                rec1[ag][j] = part[0][ag][0][j] + part[0][ag][1][j] + part[0][ag][2][j] + part[0][ag][3][j];
                dh0l1[ag][j] = part[1][ag][0][j] + part[1][ag][1][j] + part[1][ag][2][j] + part[1][ag][3][j];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NCELL; ++q) {   // layer 0 gate gradients
                const int ag = pt + 4 * q;
                if (ag < NA) {
                    const float ig = cg[q][0][0], fg = cg[q][0][1], gg = cg[q][0][2], og = cg[q][0][3];
                    const float c = cc[q][0], cp = ccp[q][0];
                    const float tc = tanhf(c);
                    const float dh = dh0l1[ag][j] + rec0[ag][j];
                    const float dc = dh * og * (1.f - tc * tc) + dc0n[ag][j];
                    dgl[ag][j] = dc * gg * ig * (1.f - ig);
                    dgl[ag][64 + j] = dc * cp * fg * (1.f - fg);
                    dgl[ag][128 + j] = dc * ig * (1.f - gg * gg);
                    dgl[ag][192 + j] = dh * tc * og * (1.f - og);
                    dc0n[ag][j] = dc * fg;
                }
            }
            __syncthreads();
            {   // W_hh0^T dg0 (recurrent) ; W_ih0^T dg0 = dL/dz_t (wave pt reduces latent channel pt of every agent)
                float s1[NA];
#pragma unroll
                for (int ag = 0; ag < NA; ++ag) s1[ag] = 0.f;
                mv64<256>(s1, &dgl[0][0], 64 * pt, th0);
#pragma unroll
                for (int ag = 0; ag < NA; ++ag) part[2][ag][pt][j] = s1[ag];
                float wz[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) wz[q] = w.w_ih0[(j + 64 * q) * 4 + pt];
#pragma unroll
                for (int ag = 0; ag < NA; ++ag) {
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) s = fmaf(wz[q], dgl[ag][j + 64 * q], s);
#pragma unroll
                    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
                    if (j == 0) dz[ag][4 * t + pt] = s;
                }
            }
            __syncthreads();
            for (int ag = pt; ag < NA; ag += 4) rec0[ag][j] = part[2][ag][0][j] + part[2][ag][1][j] + part[2][ag][2][j] + part[2][ag][3][j];
            __syncthreads();
        }
        // ---------------- one optimiser step on the mean (clipped if asked); then the ancestral noise ----------------
        for (int ag = 0; ag < NA; ++ag) {
            const int b = b0 + ag;
            if (b >= a.B || r >= 208) continue;
            const float g = dz[ag][r];
            float delta = (a.optimizer == 0) ? -a.lr * g / (fabsf(g) + 1e-8f) : -a.lr * g;      // Adam's first step | SGD
            if (a.perturb_th >= 0.f) delta = fminf(fmaxf(delta, -a.perturb_th), a.perturb_th);
            const float mu = zin[ag][r] + delta;
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

static int guide_grid(int B) {
    const int groups = (B + GNA - 1) / GNA;
    return groups < 512 ? groups : 512;
}
size_t guide_scratch_floats(int B) { return (size_t)guide_grid(B) * GNA * (G_GATES + G_CELLS); }

hipError_t launch_guide(const DecoderWeights& w, const DynParams& d, const GuideArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(guide_kernel, dim3(guide_grid(a.B)), dim3(256), 0, s, w, d, a);
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
