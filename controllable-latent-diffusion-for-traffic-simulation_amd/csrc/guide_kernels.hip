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
// Unicycle roll-out + loss gradient of ONE agent (forward: diffuser_helpers.py:541-639 'parallel'; losses: guidance_loss.py):
//   acc_t = act0_t * std4 + mean4 ; w_t = act1_t * std5 + mean5
//   v_k   = clip(v_0 + dt * sum_{j<k} clip(acc_j)), k = 0..52 ; vbar_k = (v_k + v_{k+1}) / 2
//   yb_k  = max(min(0.5 |v_k|, max_yawvel / max(|v_k|, 0.1)), 0.1) ; wc_k = clip(w_k, +-yb_k)
//   th_k  = yaw_0 + dt * sum_{j<k} wc_j ; x_{k+1} = x_0 + dt * sum_{j<=k} vbar_j cos th_j (y: sin)
//   L = s_ts * sum_t |v_{t+1} - target_t|                       TargetSpeedLoss      :219-254
//     + s_sl * sum_t relu(|v_{t+1}| - speed_limit)               SpeedLimitLoss       :1509-1538
//     + s_al * sum_t relu(|acc_t| - acc_limit)                   AccLimitLoss         :1444-1467 (unclipped descaled action)
//     + s_tp * |(x, y)_{T*+1} - target_pos|                      TargetPosAtTimeLoss  :632-670   (target_time = T* >= 0)
//       or s_tp * mean_{t >= m} softmin_t(dist) dist_t^2           TargetPosLoss        :672-716   (target_time = -(m + 1) < 0)
//     + sum_t <ext_grad[b, t, :], (x, y, v, yaw, acc, yaw-rate)_t>      any loss computed elsewhere on the decoded trajectory
//       (ext_grad = dL/dtraj of the descaled [B,52,6] trajectory: this makes the kernel the vector-Jacobian product of
//        decoder + roll-out, so every upstream guidance loss that is torch code on the trajectory can drive it)
// act0 / act1 [t * st] in; dact0 / dact1 [t] = dL / d act out.  min / max / clamp pass gradients like torch (clamp: on
// [lo, hi]; an active bound of the yaw-rate clip routes the gradient into yb and from there into v_k).
// sc: 6 x 54 floats of per-agent scratch (v_k, th_k, dL/dv_k, clip mask of v_k, dL/dx_k, dL/dy_k).
__device__ __forceinline__ void chain_grad(const DynParams& d, const GuideArgs& a, int b, const float* act0, const float* act1, int st,
                                           float* dact0, float* dact1, float* sc) {
    float* vk = sc; float* th = sc + 54; float* gv = sc + 108; float* gth = sc + 162;   // gth: clip mask of v_k (1 = inside the bounds)
    float* gxk = sc + 216; float* gyk = sc + 270;                                        // direct dL/dx_{k+1}, dL/dy_{k+1} (positions first)
    const float* cs = a.curr_states + (size_t)b * 4;
    const float* tgt = a.target_speed ? a.target_speed + (size_t)b * GT : nullptr;
    const float s_ts = tgt ? (a.loss_scale ? a.loss_scale[b] : (1.0f / (float)GT)) : 0.f;
    const float s_sl = a.speed_limit_scale ? a.speed_limit_scale[b] : 0.f;
    const float s_al = a.acc_limit_scale ? a.acc_limit_scale[b] : 0.f;
    const float s_tp = a.target_pos_scale ? a.target_pos_scale[b] : 0.f;
    const float* eg = a.ext_grad ? a.ext_grad + (size_t)b * GT * 6 : nullptr;
    const bool pos = s_tp != 0.f || eg != nullptr;
    // ---- forward ----
    float v_raw = cs[2], x = cs[0], y = cs[1], yaw = cs[3];
    vk[0] = fminf(fmaxf(v_raw, d.v_lo), d.v_hi);
    th[0] = yaw;
    gv[0] = 0.f;
    int tstar = 0;
    if (s_tp != 0.f) { tstar = a.target_time[b]; tstar = tstar > GT - 1 ? GT - 1 : tstar; }    // (target_* are null without the term)
    for (int t = 0; t < GT; ++t) {
        const float acc = act0[t * st] * d.std[4] + d.mean[4];
        v_raw += fminf(fmaxf(acc, d.acc_lo), d.acc_hi) * d.dt;
        const bool vok = v_raw >= d.v_lo && v_raw <= d.v_hi;
        const float v = fminf(fmaxf(v_raw, d.v_lo), d.v_hi);
        vk[t + 1] = v;
        float g = 0.f;
        if (tgt) {
            const float df = v - tgt[t];
            g += s_ts * ((df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f));        // d|x|/dx, 0 at 0 (and for NaN targets: nan_to_num)
        }
        if (s_sl != 0.f && fabsf(v) - a.speed_limit > 0.f) g += s_sl * ((v > 0.f) ? 1.f : -1.f);
        if (eg) g += eg[t * 6 + 2];
        gv[t + 1] = g;                          // direct loss terms on v_{t+1}; the clip mask is applied in the backward sweep
        gth[t + 1] = vok ? 1.f : 0.f;           // (borrowed until the backward sweep: clip mask of v_{t+1})
        if (pos) {
            const float av = fabsf(vk[t]);
            const float yb = fmaxf(fminf(d.max_steer * av, d.max_yawvel / fmaxf(av, 0.1f)), 0.1f);
            const float wr = act1[t * st] * d.std[5] + d.mean[5];
            const float wc = fmaxf(fminf(wr, yb), -yb);
            const float vbar = 0.5f * (vk[t] + v);
            x += vbar * cosf(yaw) * d.dt;
            y += vbar * sinf(yaw) * d.dt;
            yaw += wc * d.dt;
            th[t + 1] = yaw;
            gxk[t] = x; gyk[t] = y;             // positions for now
        }
    }
    if (pos && s_tp == 0.f) {
        for (int t = 0; t < GT; ++t) { gxk[t] = 0.f; gyk[t] = 0.f; }
    } else if (pos) {      // positions -> direct position gradients
        const float wx = a.target_pos[2 * b], wy = a.target_pos[2 * b + 1];
        if (tstar >= 0) {                       // hit the waypoint AT step tstar: d|e|/de (torch.norm: 0 at 0)
            const float ex = gxk[tstar] - wx, ey = gyk[tstar] - wy;
            const float nrm = sqrtf(ex * ex + ey * ey);
            for (int t = 0; t < GT; ++t) { gxk[t] = 0.f; gyk[t] = 0.f; }
            if (nrm > 0.f) { gxk[tstar] = s_tp * ex / nrm; gyk[tstar] = s_tp * ey / nrm; }
        } else {                                // hit it at SOME step >= m: L = mean_t w_t dist_t^2, w = softmin(dist)
            int m = -tstar - 1;
            m = m > GT - 1 ? GT - 1 : m;
            float dmin = 3.4e38f;
            for (int t = m; t < GT; ++t) {
                const float ex = gxk[t] - wx, ey = gyk[t] - wy;
                dmin = fminf(dmin, sqrtf(ex * ex + ey * ey));
            }
            float z = 0.f, S = 0.f;
            for (int t = m; t < GT; ++t) {
                const float ex = gxk[t] - wx, ey = gyk[t] - wy;
                const float dd = sqrtf(ex * ex + ey * ey), e = expf(-(dd - dmin));
                z += e; S += e * dd * dd;
            }
            S /= z;
            const float inv = s_tp / (float)(GT - m);
            for (int t = 0; t < GT; ++t) {
                if (t < m) { gxk[t] = 0.f; gyk[t] = 0.f; continue; }
                const float ex = gxk[t] - wx, ey = gyk[t] - wy;
                const float dd = sqrtf(ex * ex + ey * ey), wgt = expf(-(dd - dmin)) / z;
                // d/dp_t [ sum_s w_s dist_s^2 ] = w_t (2 e_t + (S - dist_t^2) e_t / dist_t)
                const float k = inv * wgt * (2.f + (dd > 0.f ? (S - dd * dd) / dd : 0.f));
                gxk[t] = k * ex; gyk[t] = k * ey;
            }
        }
    }
    // ---- backward ----
    float gx = 0.f, gy = 0.f;                   // dL/dx_{k+1}, dL/dy_{k+1} summed over k >= current step (suffix sums)
    float g_th_suffix = 0.f;                    // sum_{m > k} dL/dth_m
    float run_v = 0.f;                          // sum_{k > j} dL/dv_raw_k
    float d_vbar_next = 0.f;                    // dL/dvbar_{k+1}
    for (int k = GT - 1; k >= 0; --k) {
        float d_w = 0.f, d_vk_from_yb = 0.f, d_vbar = 0.f;
        if (pos) {
            gx += gxk[k]; gy += gyk[k];
            if (eg) { gx += eg[k * 6 + 0]; gy += eg[k * 6 + 1]; g_th_suffix += eg[k * 6 + 3]; }     // direct terms on x_{k+1}, y_{k+1}, th_{k+1}
            float sn, cn;
            sincosf(th[k], &sn, &cn);
            const float vbar = 0.5f * (vk[k] + vk[k + 1]);
            d_vbar = d.dt * (gx * cn + gy * sn);
            const float d_thk = d.dt * vbar * (-gx * sn + gy * cn);              // dL/dth_k through the positions
            // th_m for m > k depends on wc_k
            const float d_wc = d.dt * g_th_suffix;
            const float av = fabsf(vk[k]);
            const float ya = d.max_steer * av, ybb = d.max_yawvel / fmaxf(av, 0.1f);
            const float yb = fmaxf(fminf(ya, ybb), 0.1f);
            const float wr = act1[k * st] * d.std[5] + d.mean[5];
            float d_yb = 0.f;
            if (wr > yb) d_yb = d_wc; else if (wr < -yb) d_yb = -d_wc; else d_w = d_wc;
            if (eg) d_w += eg[k * 6 + 5];                                          // the trajectory carries the unclipped descaled yaw rate
            if (d_yb != 0.f && fminf(ya, ybb) > 0.1f) {                           // the 0.1 floor is not active
                const float dyb_dav = (ya < ybb) ? d.max_steer : ((av > 0.1f) ? -d.max_yawvel / (av * av) : 0.f);
                d_vk_from_yb = d_yb * dyb_dav * ((vk[k] > 0.f) ? 1.f : ((vk[k] < 0.f) ? -1.f : 0.f));
            }
            g_th_suffix += d_thk;               // th_k joins the suffix for steps below k
        }
        // v_{k+1}: direct terms + both averages it enters (vbar_k and vbar_{k+1})
        const float d_vk1 = (gv[k + 1] + 0.5f * (d_vbar + d_vbar_next)) * gth[k + 1];      // gth: clip mask of v_{k+1}
        // (the yaw-bound path of v_{k+1}, as v_prev of step k+1, was added into gv[k+1] by the iteration above)
        run_v += d_vk1;
        const float acc = act0[k * st] * d.std[4] + d.mean[4];
        float g = (acc >= d.acc_lo && acc <= d.acc_hi) ? run_v * d.dt : 0.f;
        if (s_al != 0.f && fabsf(acc) - a.acc_limit > 0.f) g += s_al * ((acc > 0.f) ? 1.f : -1.f);
        if (eg) g += eg[k * 6 + 4];                                                // ... and the unclipped descaled acceleration
        dact0[k] = g * d.std[4];
        dact1[k] = d_w * d.std[5];
        gv[k] += d_vk_from_yb;                  // v_k is v_prev of step k (k >= 1: a parameter-dependent speed)
        d_vbar_next = d_vbar;
    }
}

// The same function for the AGN agents of a workgroup, by all NT threads of it.  chain_grad is one thread per agent walking the
// 52 steps twice with a sine / cosine pair, divisions and branches at every step: 55k cycles on 8 lanes while the other waves
// wait (6 % of an 8-agent guided step).  Everything in it that is elementwise in (agent, step) runs here one (agent, step)
// pair per thread; what remains sequential are six running sums over the steps, done by one lane per agent in the original
// order of additions -- so the results are those of chain_grad, sum for sum (the loss terms have kinks -- |v - target|, the
// clips -- where a re-associated sum could flip a sign).  Phases alternate, one __syncthreads between them.
// sc: CG_ROWS x 54 floats per agent.  act / dact as in chain_grad (act0 / act1 [t * st + agent], dact [agent][2][GT]).
constexpr int CG_ROWS = 16;
template <int AGN, int NT, class AgentFn>
__device__ __forceinline__ void chain_grad_group(const DynParams& d, const GuideArgs& a, AgentFn agent, const float* act0, const float* act1, int st,
                                                 float* dact, float* sc_all) {
    const int tid = threadIdx.x;
    enum { VK = 0, TH = 1, GV = 2, MASK = 3, GXK = 4, GYK = 5, ACCC = 6, WC = 7, CX = 8, CY = 9, DTH = 10, DVBAR = 11, THS = 12, DVY = 13, DVK1 = 14, RUN = 15 };
    auto row = [&](int ag, int r) { return sc_all + (ag * CG_ROWS + r) * 54; };
    // per-thread view of its (agent, step) pair(s) and of the agent's constants
    struct Consts { const float* cs; const float* tgt; const float* eg; float s_ts, s_sl, s_al, s_tp; bool pos; };
    auto consts = [&](int ag) {
        const int b = agent(ag);
        Consts c;
        c.cs = a.curr_states + (size_t)b * 4;
        c.tgt = a.target_speed ? a.target_speed + (size_t)b * GT : nullptr;
        c.s_ts = c.tgt ? (a.loss_scale ? a.loss_scale[b] : (1.0f / (float)GT)) : 0.f;
        c.s_sl = a.speed_limit_scale ? a.speed_limit_scale[b] : 0.f;
        c.s_al = a.acc_limit_scale ? a.acc_limit_scale[b] : 0.f;
        c.s_tp = a.target_pos_scale ? a.target_pos_scale[b] : 0.f;
        c.eg = a.ext_grad ? a.ext_grad + (size_t)b * GT * 6 : nullptr;
        c.pos = c.s_tp != 0.f || c.eg != nullptr;
        return c;
    };
    // ---- P1: clipped accelerations ----
    for (int i = tid; i < AGN * GT; i += NT) {
        const int ag = i / GT, t = i % GT;
        const float acc = act0[t * st + ag] * d.std[4] + d.mean[4];
        row(ag, ACCC)[t] = fminf(fmaxf(acc, d.acc_lo), d.acc_hi);
    }
    __syncthreads();
    // ---- S1: speeds (running sum) ----
    if (tid < AGN) {
        const int ag = tid;
        const float* cs = a.curr_states + (size_t)agent(ag) * 4;
        float* vk = row(ag, VK); float* mk = row(ag, MASK); const float* ac = row(ag, ACCC);
        float v_raw = cs[2];
        vk[0] = fminf(fmaxf(v_raw, d.v_lo), d.v_hi);
        for (int t = 0; t < GT; ++t) {
            v_raw += ac[t] * d.dt;
            mk[t + 1] = (v_raw >= d.v_lo && v_raw <= d.v_hi) ? 1.f : 0.f;
            vk[t + 1] = fminf(fmaxf(v_raw, d.v_lo), d.v_hi);
        }
    }
    __syncthreads();
    // ---- P2: direct loss terms on v_{t+1}; clipped yaw rates ----
    for (int i = tid; i < AGN * GT; i += NT) {
        const int ag = i / GT, t = i % GT;
        const Consts c = consts(ag);
        const float* vk = row(ag, VK);
        const float v = vk[t + 1];
        float g = 0.f;
        if (c.tgt) {
            const float df = v - c.tgt[t];
            g += c.s_ts * ((df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f));        // d|x|/dx, 0 at 0 (and for NaN targets: nan_to_num)
        }
        if (c.s_sl != 0.f && fabsf(v) - a.speed_limit > 0.f) g += c.s_sl * ((v > 0.f) ? 1.f : -1.f);
        if (c.eg) g += c.eg[t * 6 + 2];
        row(ag, GV)[t + 1] = g;
        if (t == 0) { row(ag, GV)[0] = 0.f; row(ag, DVBAR)[GT] = 0.f; row(ag, DVY)[GT] = 0.f; }
        if (c.pos) {
            const float av = fabsf(vk[t]);
            const float yb = fmaxf(fminf(d.max_steer * av, d.max_yawvel / fmaxf(av, 0.1f)), 0.1f);
            const float wr = act1[t * st + ag] * d.std[5] + d.mean[5];
            row(ag, WC)[t] = fmaxf(fminf(wr, yb), -yb);
        }
    }
    __syncthreads();
    // ---- S2: headings (running sum) ----
    if (tid < AGN) {
        const int ag = tid;
        const Consts c = consts(ag);
        if (c.pos) {
            float* th = row(ag, TH); const float* wc = row(ag, WC);
            float yaw = c.cs[3];
            th[0] = yaw;
            for (int t = 0; t < GT; ++t) { yaw += wc[t] * d.dt; th[t + 1] = yaw; }
        }
    }
    __syncthreads();
    // ---- P3: displacement terms ----
    for (int i = tid; i < AGN * GT; i += NT) {
        const int ag = i / GT, t = i % GT;
        const Consts c = consts(ag);
        if (c.pos) {
            const float* vk = row(ag, VK);
            const float vbar = 0.5f * (vk[t] + vk[t + 1]);
            const float yaw = row(ag, TH)[t];
            row(ag, CX)[t] = vbar * cosf(yaw);
            row(ag, CY)[t] = vbar * sinf(yaw);
        }
    }
    __syncthreads();
    // ---- S3: positions (running sums), the position losses on them, and the suffix sums of dL/dx, dL/dy ----
    if (tid < AGN) {
        const int ag = tid, b = agent(ag);
        const Consts c = consts(ag);
        float* gxk = row(ag, GXK); float* gyk = row(ag, GYK);
        if (c.pos) {
            const float* cx = row(ag, CX); const float* cy = row(ag, CY);
            float x = c.cs[0], y = c.cs[1];
            for (int t = 0; t < GT; ++t) { x += cx[t] * d.dt; y += cy[t] * d.dt; gxk[t] = x; gyk[t] = y; }
            if (c.s_tp == 0.f) {
                for (int t = 0; t < GT; ++t) { gxk[t] = 0.f; gyk[t] = 0.f; }
            } else {      // positions -> direct position gradients (as in chain_grad)
                int tstar = a.target_time[b]; tstar = tstar > GT - 1 ? GT - 1 : tstar;
                const float wx = a.target_pos[2 * b], wy = a.target_pos[2 * b + 1];
                if (tstar >= 0) {
                    const float ex = gxk[tstar] - wx, ey = gyk[tstar] - wy;
                    const float nrm = sqrtf(ex * ex + ey * ey);
                    for (int t = 0; t < GT; ++t) { gxk[t] = 0.f; gyk[t] = 0.f; }
                    if (nrm > 0.f) { gxk[tstar] = c.s_tp * ex / nrm; gyk[tstar] = c.s_tp * ey / nrm; }
                } else {
                    int m = -tstar - 1;
                    m = m > GT - 1 ? GT - 1 : m;
                    float dmin = 3.4e38f;
                    for (int t = m; t < GT; ++t) {
                        const float ex = gxk[t] - wx, ey = gyk[t] - wy;
                        dmin = fminf(dmin, sqrtf(ex * ex + ey * ey));
                    }
                    float z = 0.f, S = 0.f;
                    for (int t = m; t < GT; ++t) {
                        const float ex = gxk[t] - wx, ey = gyk[t] - wy;
                        const float dd = sqrtf(ex * ex + ey * ey), e = expf(-(dd - dmin));
                        z += e; S += e * dd * dd;
                    }
                    S /= z;
                    const float inv = c.s_tp / (float)(GT - m);
                    for (int t = 0; t < GT; ++t) {
                        if (t < m) { gxk[t] = 0.f; gyk[t] = 0.f; continue; }
                        const float ex = gxk[t] - wx, ey = gyk[t] - wy;
                        const float dd = sqrtf(ex * ex + ey * ey), wgt = expf(-(dd - dmin)) / z;
                        const float k = inv * wgt * (2.f + (dd > 0.f ? (S - dd * dd) / dd : 0.f));
                        gxk[t] = k * ex; gyk[t] = k * ey;
                    }
                }
            }
            float gx = 0.f, gy = 0.f;           // dL/dx_{k+1}, dL/dy_{k+1} summed over the steps >= k
            for (int k = GT - 1; k >= 0; --k) {
                gx += gxk[k]; gy += gyk[k];
                if (c.eg) { gx += c.eg[k * 6 + 0]; gy += c.eg[k * 6 + 1]; }
                gxk[k] = gx; gyk[k] = gy;
            }
        }
    }
    __syncthreads();
    // ---- P5: dL/dvbar_k and dL/dth_k through the positions ----
    for (int i = tid; i < AGN * GT; i += NT) {
        const int ag = i / GT, k = i % GT;
        const Consts c = consts(ag);
        if (c.pos) {
            const float* vk = row(ag, VK);
            float sn, cn;
            sincosf(row(ag, TH)[k], &sn, &cn);
            const float vbar = 0.5f * (vk[k] + vk[k + 1]);
            const float gx = row(ag, GXK)[k], gy = row(ag, GYK)[k];
            row(ag, DVBAR)[k] = d.dt * (gx * cn + gy * sn);
            row(ag, DTH)[k] = d.dt * vbar * (-gx * sn + gy * cn);
        } else {
            row(ag, DVBAR)[k] = 0.f;
        }
    }
    __syncthreads();
    // ---- S5: suffix sums of dL/dth ----
    if (tid < AGN) {
        const int ag = tid;
        const Consts c = consts(ag);
        if (c.pos) {
            const float* dth = row(ag, DTH); float* ths = row(ag, THS);
            float g = 0.f;                      // sum_{m > k} dL/dth_m (+ the direct terms on th_{m}, m > k)
            for (int k = GT - 1; k >= 0; --k) {
                if (c.eg) g += c.eg[k * 6 + 3];
                ths[k] = g;
                g += dth[k];
            }
        }
    }
    __syncthreads();
    // ---- P6: the yaw-rate clip: dL/d yaw-rate action, and the part of it routed into v_k through the bound ----
    for (int i = tid; i < AGN * GT; i += NT) {
        const int ag = i / GT, k = i % GT;
        const Consts c = consts(ag);
        float d_w = 0.f, d_vk_from_yb = 0.f;
        if (c.pos) {
            const float* vk = row(ag, VK);
            const float d_wc = d.dt * row(ag, THS)[k];
            const float av = fabsf(vk[k]);
            const float ya = d.max_steer * av, ybb = d.max_yawvel / fmaxf(av, 0.1f);
            const float yb = fmaxf(fminf(ya, ybb), 0.1f);
            const float wr = act1[k * st + ag] * d.std[5] + d.mean[5];
            float d_yb = 0.f;
            if (wr > yb) d_yb = d_wc; else if (wr < -yb) d_yb = -d_wc; else d_w = d_wc;
            if (c.eg) d_w += c.eg[k * 6 + 5];
            if (d_yb != 0.f && fminf(ya, ybb) > 0.1f) {
                const float dyb_dav = (ya < ybb) ? d.max_steer : ((av > 0.1f) ? -d.max_yawvel / (av * av) : 0.f);
                d_vk_from_yb = d_yb * dyb_dav * ((vk[k] > 0.f) ? 1.f : ((vk[k] < 0.f) ? -1.f : 0.f));
            }
        }
        row(ag, DVY)[k] = d_vk_from_yb;
        dact[(ag * 2 + 1) * GT + k] = d_w * d.std[5];
    }
    __syncthreads();
    // ---- P7: dL/dv_{k+1}: direct terms + the yaw-bound path + both averages it enters ----
    for (int i = tid; i < AGN * GT; i += NT) {
        const int ag = i / GT, k = i % GT;
        const float* dvb = row(ag, DVBAR);
        const float gvn = row(ag, GV)[k + 1] + row(ag, DVY)[k + 1];
        row(ag, DVK1)[k] = (gvn + 0.5f * (dvb[k] + dvb[k + 1])) * row(ag, MASK)[k + 1];
    }
    __syncthreads();
    // ---- S6: suffix sums of dL/dv_raw ----
    if (tid < AGN) {
        const int ag = tid;
        const float* dv = row(ag, DVK1); float* rn = row(ag, RUN);
        float run = 0.f;
        for (int k = GT - 1; k >= 0; --k) { run += dv[k]; rn[k] = run; }
    }
    __syncthreads();
    // ---- P8: dL/d acceleration action ----
    for (int i = tid; i < AGN * GT; i += NT) {
        const int ag = i / GT, k = i % GT;
        const Consts c = consts(ag);
        const float acc = act0[k * st + ag] * d.std[4] + d.mean[4];
        float g = (acc >= d.acc_lo && acc <= d.acc_hi) ? row(ag, RUN)[k] * d.dt : 0.f;
        if (c.s_al != 0.f && fabsf(acc) - a.acc_limit > 0.f) g += c.s_al * ((acc > 0.f) ? 1.f : -1.f);
        if (c.eg) g += c.eg[k * 6 + 4];
        dact[(ag * 2 + 0) * GT + k] = g * d.std[4];
    }
    __syncthreads();
}

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

// One optimiser step on one latent element (upstream perturb(), guidance_loss.py:2247-2278: torch.optim.Adam / SGD on x_guidance,
// then the clip around x_initial).  Step 1 of Adam is -lr g / (|g| + eps) (bias-corrected moments of a single gradient); steps
// k > 1 of a multi-step call carry torch's moments: m_k = 0.9 m + 0.1 g, v_k = 0.999 v + 0.001 g^2,
// x -= (lr / (1 - 0.9^k)) m_k / (sqrt(v_k) / sqrt(1 - 0.999^k) + 1e-8).  SGD has no state.  `cur` is the current iterate.
__device__ __forceinline__ float optimiser_step(const GuideArgs& a, float g, float cur, size_t gi) {
    float delta;
    if (a.optimizer != 0) {
        delta = -a.lr * g;
    } else if (a.opt_steps <= 1) {
        delta = -a.lr * g / (fabsf(g) + 1e-8f);
    } else {
        float m = 0.1f * g, v = 0.001f * g * g;
        if (a.opt_step > 1) { m = 0.9f * a.adam_m[gi] + m; v = 0.999f * a.adam_v[gi] + v; }
        if (a.opt_step < a.opt_steps) { a.adam_m[gi] = m; a.adam_v[gi] = v; }
        const float bc1 = 1.0f - powf(0.9f, (float)a.opt_step), bc2 = 1.0f - powf(0.999f, (float)a.opt_step);
        delta = -(a.lr / bc1) * m / (sqrtf(v) / sqrtf(bc2) + 1e-8f);
    }
    float mu = cur + delta;
    if (a.perturb_th >= 0.f) {
        const float m0 = a.mean0 ? a.mean0[gi] : cur;
        mu = m0 + fminf(fmaxf(mu - m0, -a.perturb_th), a.perturb_th);
    }
    return mu;
}


__global__ __launch_bounds__(256) void guide_kernel(const DecoderWeights w, const DynParams d, const GuideArgs a) {
    constexpr int NA = GNA;
    __shared__ __attribute__((aligned(16))) float h0[NA][64], h1[NA][64], c0[NA][64], c1[NA][64], gates[NA][256], zin[NA][208];
    __shared__ __attribute__((aligned(16))) float condm[NA][256];
    __shared__ float act[NA][2][GT];         // scaled (acceleration, yaw-rate) output of the decoder
    __shared__ float dact[NA][2][GT];        // dL / d(scaled output)
    __shared__ float chs[NA][324];           // roll-out scratch of chain_grad
    __shared__ __attribute__((aligned(16))) float dgl[NA][256];           // gate gradients of the layer being processed
    __shared__ float part[3][NA][4][64];     // partial transposed products
    __shared__ float rec1[NA][64], rec0[NA][64], dh0l1[NA][64], dc1n[NA][64], dc0n[NA][64];
    __shared__ float dz[NA][208];
    const int r = threadIdx.x;
    const int gate = r >> 6;
    const int j = r & 63, pt = r >> 6;       // matvec phases: (column, row block); per-cell phases: unit j of agents pt, pt + 4, ...

    const float bias0 = w.b0[r], bias1 = w.b1[r];
    const float wa0 = w.w_h2a[j], wa1 = w.w_h2a[64 + j];     // d act[:, c] / d h1[j]
    const float bh2a = w.b_h2a[0], bh2b = w.b_h2a[1];

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
            for (int ag = pt; ag < NA; ag += 4) {   // hid2act: one wave per agent, lane j holds unit j
                float s0 = h1[ag][j] * wa0, s1 = h1[ag][j] * wa1;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
                if (j == 0) { act[ag][0][t] = s0 + bh2a; act[ag][1][t] = s1 + bh2b; }
            }
        }
        }
        __syncthreads();
        // ---------------- speed chain + loss gradient (diffuser_helpers.py:573-600; guidance_loss.py:229-254) ----------------
        if (r < NA) chain_grad(d, a, agent(r), &act[r][0][0], &act[r][1][0], 1, &dact[r][0][0], &dact[r][1][0], &chs[r][0]);
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
                    const float dh = wa0 * dact[ag][0][t] + wa1 * dact[ag][1][t] + rec1[ag][j];
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
            const float mu = optimiser_step(a, g, zin[ag][r], (size_t)b * 208 + r);
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

__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

// =============================================================================================
// MFMA formulation: 16 agents per workgroup, the recurrent products as v_mfma_f32_16x16x4_f32 tiles
// =============================================================================================
// For a layer step the gate pre-activations of 16 agents are the GEMM [16 agents x K] x [K x 256].  Eight waves, two per SIMD;
// wave w owns hidden units 8w..8w+7 of BOTH layers, so every wave does an eighth of a step and one wave's cell update, exchange
// and barrier wait run under its SIMD neighbour's MFMAs (round 1's four-wave form, one wave per SIMD with 16 units each, spent
// half of a step in latency nothing covered: 565 against 506 us).  Only h goes through LDS (the A operand of the next product)
// -- two barriers per time step; A operands are read as float4 (4 consecutive k per lane, element e feeds MFMA e), B fragments
// use the same k permutation.
//   forward : a wave's gate columns are two N-tiles, P = [i of its 8 units | f of its 8 units] and Q = [g | o]; in the result
//             layout lane (n, rb) holds column n for agents 4rb..4rb+3, so lanes n < 8 hold (i, g) and lanes n >= 8 hold
//             (f, o) of unit n & 7.  The two half-rows trade what the other needs with one DPP row rotation by 8 each
//             (v_mov_dpp row_ror:8, VALU rate, no LDS) and split the four agents: lane n < 8 updates the cells of agents
//             4rb, 4rb+1, lane n >= 8 those of agents 4rb+2, 4rb+3 -- half the transcendental work per lane as well.
//   backward: the two transposed products of a layer share one N-tile: [W_hh1^T dG | W_ih1^T dG] for layer 1 and
//             [W_hh0^T dG | W_ih0^T dG (4 latent channels) | 0] for layer 0, K = 256 gate columns each: lane n < 8 ends up with
//             the recurrent gradient of unit n, lane n >= 8 with the gradient flowing down (layer 1) or with dL/dz_t, complete
//             (layer 0; no cross-wave partial sums), and again one rotation by 8 hands each half what it needs.
// The activations the backward sweep needs (i, f, g, o, c of every cell and step) are kept in an L2-resident scratch, each thread
// its own stream; the roll-out / loss scan (chain_grad) and the optimiser step are those of the 2-agent kernel.
namespace gm8 {
constexpr int AG = 16, HS = 68, GS = 260;
constexpr int ACTS = GT * 2 * 5 * 2 * 512;   // floats of kept activations per workgroup: [t][layer][i f g o c][q][thread]
}  // namespace gm8

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL store and load
// (vmcnt(0)); inside the time loops the only global traffic is each thread's private kept-activation stream (written in the
// forward sweep, read back by the same thread in the backward sweep), which needs no cross-thread ordering -- waiting for those
// stores to be acknowledged at two barriers per step was ~15 % of the step.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float xch8(float v) {      // the value lane ^ 8 holds (the other half of this 16-lane row)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
}

// [16 agents x 256 gate columns] x [256 x 16] on this wave's tile: row `src` of the gate-gradient tile (this lane's A row, k
// offset 4 rb applied), B fragments tb[j][e] for gate column 16 j + 4 rb + e.  Two accumulators in turn (no back-to-back
// dependent MFMAs); the A fragments are fetched four k-groups ahead and pinned there -- left alone the compiler hoists all
// sixteen ds_read_b128 (64 VGPRs) to the top and the kernel, capped at 256 registers for two waves per SIMD, spills.
__device__ __forceinline__ v4f tprod(const float* src, const float (&tb)[16][4]) {
    v4f p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
    v4f cur[4], nxt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cur[j] = *reinterpret_cast<const v4f*>(src + 16 * j);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        if (blk < 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) nxt[j] = *reinterpret_cast<const v4f*>(src + 16 * (4 * (blk + 1) + j));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[j][0], tb[4 * blk + j][0], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[j][1], tb[4 * blk + j][1], p1, 0, 0, 0);
            p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[j][2], tb[4 * blk + j][2], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[j][3], tb[4 * blk + j][3], p1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) p0[r] += p1[r];
    return p0;
}

#ifdef CLD_STAMPS
// diagnostic build only: shader-clock stamps of the phases of workgroup 0..255 (read back with cld_debug_guide_stamps)
__device__ unsigned long long g_guide_stamps[256 * 8];
#define GSTAMP(k)                                                                                  \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (tid == 0 && blockIdx.x < 256) {                                                        \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            g_guide_stamps[blockIdx.x * 8 + (k)] = t_;                                             \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
void read_guide_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_guide_stamps), sizeof(unsigned long long) * 256 * 8); }
// ... and cycle totals of the phases inside the time loops of guide_quad_kernel, printed by workgroup 0 (scripts/guide_one.py)
#define GPHASE_DECL unsigned long long ph_[16] = {0}, phl_ = 0
#define GPHASE_START do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(phl_)::"memory"); } while (0)
#define GPHASE(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); ph_[k] += t_ - phl_; phl_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define GPHASE_PRINT(n) do { if (tid == 0 && blockIdx.x == 0) for (int i_ = 0; i_ < (n); ++i_) printf("phase %2d: %8llu cycles (%6.0f per step)\n", i_, ph_[i_], (double)ph_[i_] / 52.0); } while (0)
#else
#define GPHASE_DECL do {} while (0)
#define GPHASE_START do {} while (0)
#define GPHASE(k) do {} while (0)
#define GPHASE_PRINT(n) do {} while (0)
#define GSTAMP(k) do {} while (0)
void read_guide_stamps(unsigned long long*) {}
#endif

__global__ __launch_bounds__(512) void guide_mfma8_kernel(const DecoderWeights w, const DynParams d, const GuideArgs a) {
    using namespace gm8;
    __shared__ __attribute__((aligned(16))) float hs[2][2][AG][HS];     // [layer][parity][agent][unit]
    __shared__ __attribute__((aligned(16))) float dG[2][AG][GS];        // gate gradients of layer 1 / layer 0
    __shared__ __attribute__((aligned(16))) float zin[AG][208];
    __shared__ __attribute__((aligned(16))) float condm[AG][256];
    __shared__ float actp[2][2][8][AG];      // [step parity][output][wave][agent]: per-wave partials of hid2act
    __shared__ float act[2][GT][AG];         // (acceleration, yaw-rate), scaled
    __shared__ float dact[AG][2][GT];
    __shared__ float chs[AG][324];           // roll-out scratch of chain_grad
    __shared__ float dz[AG][208];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, rb = lane >> 4;        // MFMA layouts: A lane = (row n, k rb); B lane = (col n, k rb); D lane = (col n, rows 4rb..4rb+3)
    const int hi = n >> 3, m = n & 7;
    const int u = 8 * wv + m;                       // this lane's hidden unit
    const int ra = 4 * rb + 2 * hi;                 // the first of the two agents whose cells this lane updates
    const float wa0 = w.w_h2a[u], wa1 = w.w_h2a[64 + u], bh2a = w.b_h2a[0], bh2b = w.b_h2a[1];

    const int ngroups = (a.B + AG - 1) / AG;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int b0 = grp * AG;
        GSTAMP(0);
        // kept activations go through a buffer descriptor: the per-lane part of every address is ONE VGPR (4 tid) and the
        // (step, layer, gate, row) part rides in the scalar offset -- no 64-bit vector address arithmetic per access
        const __amdgpu_buffer_rsrc_t keep = __builtin_amdgcn_make_buffer_rsrc(a.scratch + (size_t)blockIdx.x * ACTS, 0, ACTS * 4, 0x00020000);
        const int kvo = tid * 4;
        auto kput = [&](int slot, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), keep, kvo, slot * 2048, 0); };
        auto kget = [&](int slot) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(keep, kvo, slot * 2048, 0)); };
        auto agent = [&](int ag) { return (b0 + ag < a.B) ? b0 + ag : a.B - 1; };      // tail slots replay the last agent; never stored
        for (int i = tid; i < AG * 256; i += 512) condm[i >> 8][i & 255] = a.cond[(size_t)agent(i >> 8) * 256 + (i & 255)];
        for (int i = tid; i < AG * 208; i += 512) zin[i / 208][i % 208] = a.mean[(size_t)agent(i / 208) * 208 + i % 208];
        __syncthreads();
        if (wv < 4) {      // h0 = cond2hidden(cond) for both layers (lstm_vae.py:46-49): [16 agents x 256] x [256 x 64] as MFMA tiles,
                           // wave wv = units 16 wv .. 16 wv + 15; B fragments straight from the row-major weight (16 float4 per lane, one latency)
            const float* wr = w.w_c2h + (size_t)(16 * wv + n) * 256 + 4 * rb;
            asm volatile("" : "+v"(wr));
            v4f bf[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) bf[j] = *reinterpret_cast<const v4f*>(wr + 16 * j);
            const float bias = w.b_c2h[16 * wv + n];
            v4f h0a = {bias, bias, bias, bias}, h0b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v4f ca = *reinterpret_cast<const v4f*>(&condm[n][16 * j + 4 * rb]);
                h0a = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[0], bf[j][0], h0a, 0, 0, 0);
                h0b = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[1], bf[j][1], h0b, 0, 0, 0);
                h0a = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[2], bf[j][2], h0a, 0, 0, 0);
                h0b = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[3], bf[j][3], h0b, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = h0a[r] + h0b[r];
                hs[0][0][4 * rb + r][16 * wv + n] = v;
                hs[1][0][4 * rb + r][16 * wv + n] = v;
            }
        }
        float c0[2] = {0.f, 0.f}, c1[2] = {0.f, 0.f};
        __syncthreads();
        GSTAMP(1);
        // The two accumulator tiles of a layer step -> the four gate pre-activations of this lane's two (agent, unit) cells.
        // Lanes n < 8 hold P = i, Q = g, lanes n >= 8 hold P = f, Q = o (rows = agents 4rb..4rb+3); the lower half keeps
        // rows 0, 1 and the upper half rows 2, 3.
        auto gates = [&](const v4f& P, const v4f& Q, float (&ig)[2], float (&fg)[2], float (&gg)[2], float (&og)[2]) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float ownP = hi ? P[2 + q] : P[q], ownQ = hi ? Q[2 + q] : Q[q];
                const float gotP = xch8(hi ? P[q] : P[2 + q]), gotQ = xch8(hi ? Q[q] : Q[2 + q]);     // what the other half holds for MY rows
                ig[q] = hi ? gotP : ownP; fg[q] = hi ? ownP : gotP;
                gg[q] = hi ? gotQ : ownQ; og[q] = hi ? ownQ : gotQ;
            }
        };
        // ---------------- forward ----------------
        {
            // The weight pointers are made opaque INSIDE the group loop: the fragments are loop-invariant, and left visible the
            // compiler hoists their address arithmetic (and, given registers, the loads) of BOTH phases out of the loop -- at the
            // 256-register cap of two waves per SIMD that spills ~180 dwords and reloads them inside the time loop.
            const float *p_hh0 = w.w_hh0, *p_ih1 = w.w_ih1, *p_hh1 = w.w_hh1, *p_ih0 = w.w_ih0, *p_b0 = w.b0, *p_b1 = w.b1;
            asm volatile("" : "+v"(p_hh0), "+v"(p_ih1), "+v"(p_hh1), "+v"(p_ih0), "+v"(p_b0), "+v"(p_b1));
            // B fragments of tile T (0: P, 1: Q): column n -> gate 2T + hi of unit 8 wv + m; k-step (j, e) -> k = 16 j + 4 rb + e
            float f_hh0[2][4][4], f_ih1[2][4][4], f_hh1[2][4][4], f_ih0[2], fb0[2], fb1[2];
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                const int col = 64 * (2 * T + hi) + 8 * wv + m;
                f_ih0[T] = p_ih0[col * 4 + rb];
                fb0[T] = p_b0[col];
                fb1[T] = p_b1[col];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const v4f x0 = *reinterpret_cast<const v4f*>(p_hh0 + col * 64 + 16 * jj + 4 * rb);
                    const v4f x1 = *reinterpret_cast<const v4f*>(p_ih1 + col * 64 + 16 * jj + 4 * rb);
                    const v4f x2 = *reinterpret_cast<const v4f*>(p_hh1 + col * 64 + 16 * jj + 4 * rb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { f_hh0[T][jj][e] = x0[e]; f_ih1[T][jj][e] = x1[e]; f_hh1[T][jj][e] = x2[e]; }
                }
            }
            GSTAMP(2);
            for (int t = 0; t < GT; ++t) {
                const int pr = t & 1;
                if (t > 0 && tid < 2 * AG) {      // actions of step t-1: the eight per-wave partials (written before the last barrier)
                    const int o = tid >> 4, ag = tid & 15;
                    float s = o ? bh2b : bh2a;
#pragma unroll
                    for (int k = 0; k < 8; ++k) s += actp[pr ^ 1][o][k][ag];
                    act[o][t - 1][ag] = s;
                }
                // ---- layer 0: pre = b + x_t W_ih0^T + h0_{t-1} W_hh0^T ----
                v4f P = {fb0[0], fb0[0], fb0[0], fb0[0]}, Q = {fb0[1], fb0[1], fb0[1], fb0[1]};
                {
                    const float xa = zin[n][4 * t + rb];
                    P = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, f_ih0[0], P, 0, 0, 0);
                    Q = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, f_ih0[1], Q, 0, 0, 0);
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const v4f ha = *reinterpret_cast<const v4f*>(&hs[0][pr][n][16 * jj + 4 * rb]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        P = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[e], f_hh0[0][jj][e], P, 0, 0, 0);
                        Q = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[e], f_hh0[1][jj][e], Q, 0, 0, 0);
                    }
                }
                float ig[2], fg[2], gg[2], og[2];
                gates(P, Q, ig, fg, gg, og);
                int ks = (t * 2 + 0) * 10;          // slot of (step t, layer 0, gate 0, row 0); a slot = one float per thread
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float i_ = fsig(ig[q]), f_ = fsig(fg[q]), g_ = ftanh(gg[q]), o_ = fsig(og[q]);
                    const float c = f_ * c0[q] + i_ * g_;
                    c0[q] = c;
                    hs[0][pr ^ 1][ra + q][u] = o_ * ftanh(c);
                    kput(ks + 0 + q, i_); kput(ks + 2 + q, f_); kput(ks + 4 + q, g_); kput(ks + 6 + q, o_);
                    kput(ks + 8 + q, c);
                }
                lds_barrier();
                // ---- layer 1: pre = b + h0_t W_ih1^T + h1_{t-1} W_hh1^T ----
                P = v4f{fb1[0], fb1[0], fb1[0], fb1[0]};
                Q = v4f{fb1[1], fb1[1], fb1[1], fb1[1]};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const v4f ha = *reinterpret_cast<const v4f*>(&hs[0][pr ^ 1][n][16 * jj + 4 * rb]);
                    const v4f hb = *reinterpret_cast<const v4f*>(&hs[1][pr][n][16 * jj + 4 * rb]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        P = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[e], f_ih1[0][jj][e], P, 0, 0, 0);
                        Q = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[e], f_ih1[1][jj][e], Q, 0, 0, 0);
                        P = __builtin_amdgcn_mfma_f32_16x16x4f32(hb[e], f_hh1[0][jj][e], P, 0, 0, 0);
                        Q = __builtin_amdgcn_mfma_f32_16x16x4f32(hb[e], f_hh1[1][jj][e], Q, 0, 0, 0);
                    }
                }
                gates(P, Q, ig, fg, gg, og);
                ks += 10;
                float ap[2], aq[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float i_ = fsig(ig[q]), f_ = fsig(fg[q]), g_ = ftanh(gg[q]), o_ = fsig(og[q]);
                    const float c = f_ * c1[q] + i_ * g_;
                    c1[q] = c;
                    const float hn = o_ * ftanh(c);
                    hs[1][pr ^ 1][ra + q][u] = hn;
                    kput(ks + 0 + q, i_); kput(ks + 2 + q, f_); kput(ks + 4 + q, g_); kput(ks + 6 + q, o_);
                    kput(ks + 8 + q, c);
                    ap[q] = hn * wa0;                       // hid2act: partials over this wave's 8 units
                    aq[q] = hn * wa1;
                }
#pragma unroll
                for (int o = 1; o < 8; o <<= 1)
#pragma unroll
                    for (int q = 0; q < 2; ++q) { ap[q] += __shfl_xor(ap[q], o); aq[q] += __shfl_xor(aq[q], o); }
                if (m == 0) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) { actp[pr][0][wv][ra + q] = ap[q]; actp[pr][1][wv][ra + q] = aq[q]; }
                }
                lds_barrier();
            }
        }
        GSTAMP(3);
        // ---------------- speed chain + loss gradient (diffuser_helpers.py:573-600; guidance_loss.py:229-254) ----------------
        if (tid < 2 * AG) {
            const int o = tid >> 4, ag = tid & 15;
            float s = o ? bh2b : bh2a;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += actp[(GT - 1) & 1][o][k][ag];
            act[o][GT - 1][ag] = s;
        }
        __syncthreads();
        if (tid < AG)
            chain_grad(d, a, agent(tid), &act[0][0][tid], &act[1][0][tid], AG, &dact[tid][0][0], &dact[tid][1][0], &chs[tid][0]);
        __syncthreads();
        GSTAMP(4);
        // ---------------- backward through time ----------------
        {
            // B fragments of the transposed products, k-step (j, e) -> gate column col = 16 j + 4 rb + e:
            //   layer 1 tile: column n < 8 -> W_hh1[col][unit 8 wv + n], n >= 8 -> W_ih1[col][unit 8 wv + n - 8]
            //   layer 0 tile: column n < 8 -> W_hh0[col][unit 8 wv + n], 8 <= n < 12 -> W_ih0[col][latent channel n - 8], else 0
            // pre-packed at cld_finalize in exactly this order (DecoderWeights::gfrag): 32 coalesced float4 loads per lane.  The
            // pointer is opaque inside the group loop, as in the forward phase.
            const v4f* gf = reinterpret_cast<const v4f*>(w.gfrag) + (size_t)wv * (2 * 16 * 64) + lane;
            asm volatile("" : "+v"(gf));
            float t1[16][4], t0[16][4];
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const v4f x1 = gf[jj * 64], x0 = gf[(16 + jj) * 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) { t1[jj][e] = x1[e]; t0[jj][e] = x0[e]; }
            }
            float rec1[2] = {0.f, 0.f}, rec0[2] = {0.f, 0.f};
            float dc1n[2] = {0.f, 0.f}, dc0n[2] = {0.f, 0.f};
            // kept activations of one (step, layer): i f g o c of this lane's two cells + the previous cell state; fetched from
            // the L2-resident scratch one phase ahead, in flight under the MFMA block that precedes their use
            float kv1[6][2], kv0[6][2];
            auto fetch = [&](float (&kv)[6][2], int t, int layer) {
                const int ks = (t * 2 + layer) * 10;
#pragma unroll
                for (int k = 0; k < 5; ++k)
#pragma unroll
                    for (int q = 0; q < 2; ++q) kv[k][q] = kget(ks + 2 * k + q);
#pragma unroll
                for (int q = 0; q < 2; ++q) kv[5][q] = t > 0 ? kget(ks - 20 + 8 + q) : 0.f;      // the cell state of step t-1, same layer
            };
            fetch(kv1, GT - 1, 1);
            GSTAMP(5);
            for (int t = GT - 1; t >= 0; --t) {
                // ---- layer 1 gate gradients -> LDS ----
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float i_ = kv1[0][q], f_ = kv1[1][q], g_ = kv1[2][q], o_ = kv1[3][q], c = kv1[4][q], cp = kv1[5][q];
                    const float tc = ftanh(c);
                    const float dh = wa0 * dact[ra + q][0][t] + wa1 * dact[ra + q][1][t] + rec1[q];
                    const float dc = dh * o_ * (1.f - tc * tc) + dc1n[q];
                    float* row = &dG[0][ra + q][u];
                    row[0] = dc * g_ * i_ * (1.f - i_);
                    row[64] = dc * cp * f_ * (1.f - f_);
                    row[128] = dc * i_ * (1.f - g_ * g_);
                    row[192] = dh * tc * o_ * (1.f - o_);
                    dc1n[q] = dc * f_;
                }
                lds_barrier();
                fetch(kv0, t, 0);
                __builtin_amdgcn_sched_barrier(0);
                v4f pa = tprod(&dG[0][n][4 * rb], t1);
                // lanes n < 8: pa = recurrent gradient (-> rec1 of step t-1); lanes n >= 8: pa = dL/dh0_t from layer 1
                float down[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float got = xch8(hi ? pa[q] : pa[2 + q]);
                    rec1[q] = hi ? got : pa[q];
                    down[q] = hi ? pa[2 + q] : got;
                }
                // ---- layer 0 gate gradients -> LDS ----
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float i_ = kv0[0][q], f_ = kv0[1][q], g_ = kv0[2][q], o_ = kv0[3][q], c = kv0[4][q], cp = kv0[5][q];
                    const float tc = ftanh(c);
                    const float dh = down[q] + rec0[q];
                    const float dc = dh * o_ * (1.f - tc * tc) + dc0n[q];
                    float* row = &dG[1][ra + q][u];
                    row[0] = dc * g_ * i_ * (1.f - i_);
                    row[64] = dc * cp * f_ * (1.f - f_);
                    row[128] = dc * i_ * (1.f - g_ * g_);
                    row[192] = dh * tc * o_ * (1.f - o_);
                    dc0n[q] = dc * f_;
                }
                lds_barrier();
                if (t > 0) fetch(kv1, t - 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                v4f pc = tprod(&dG[1][n][4 * rb], t0);
                // lanes n < 8: pc = recurrent gradient (-> rec0 of step t-1); lanes 8 <= n < 12: pc = dL/dz_t, channel n - 8, complete
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float got = xch8(pc[2 + q]);
                    rec0[q] = hi ? got : pc[q];
                }
                if (wv == 0 && hi && m < 4) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dz[4 * rb + r][4 * t + m] = pc[r];
                }
            }
        }
        __syncthreads();
        GSTAMP(6);
        // ---------------- one optimiser step on the mean (clipped if asked); then the ancestral noise ----------------
        for (int i = tid; i < AG * 208; i += 512) {
            const int ag = i / 208, r = i % 208, b = b0 + ag;
            if (b >= a.B) continue;
            const float g = dz[ag][r];
            const float mu = optimiser_step(a, g, zin[ag][r], (size_t)b * 208 + r);
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

// =============================================================================================
// Eight agents per workgroup on the 16-block 4x4x1 fp32 MFMA: the whole chip at 2,048 agents
// =============================================================================================
// The 16x16x4 kernel needs 16 agents per workgroup (one M-tile): 2,048 agents are 128 workgroups on 128 of the 256 CUs, each
// bound by its own MFMA issue.  v_mfma_f32_4x4x1_16b_f32 computes sixteen independent 4x4 blocks with K = 1 at the same fp32
// rate (scripts/ubench/mfma4x4.hip: 8.4-9.0 cycles, the SIMD's rate whatever the number of waves), so the M granularity drops to
// 4: 8 agents = two quads per workgroup, 256 workgroups, half the MFMA work on each CU.  What shapes the kernel (measurements in
// scripts/ubench/{mfma_valu,lds_mfma}.hip and profiles/r02/guide_kernel_notes.md):
//   * this short MFMA has no shadow: every other instruction of the SIMD -- VALU (+4.6 cycles a pair), transcendental (+10),
//     LDS read, register move -- adds its issue cycles on top, from the same wave or another, so instruction COUNT is the cost;
//   * one weight register per k (K = 1 per MFMA): 196 registers forward, 256 backward for a wave that holds all of K -- one wave
//     per SIMD, half the weights in AGPRs behind a reload move per MFMA.  So K is split: waves wq and wq + 4 own the same 16
//     hidden units, each takes HALF of the K dimension of every product (100 / 128 weight registers, all VGPRs, two waves per
//     SIMD) and the cell updates of ONE agent quad (kp = wave >> 2); the partial sums for the other quad go to the partner
//     wave through LDS;
//   * the agents' values are the A operand (rows = the four agents of a quad) and are BROADCAST: cbsz / abid hand the rows of
//     one 4-lane block to all 16 blocks, so one ds_read_b128 whose lanes of block b hold k-group b feeds 16 k-groups (bsweep).
//     Every lane fetching its own copy instead moved ~100 KB per wave and step through the CU's 128 B/clk LDS port -- as long
//     as the MFMAs take.  The weights are the B operand: lane (block = unit, column = gate) -> the sweep leaves gate q of the
//     unit for the quad's four agents in a lane's four registers, and a 4 x 4 transpose across the unit's four lanes (DPP quad
//     permutes) hands lane (unit, a) all four gates of agent a: the cell update runs in registers.
// A step is two phases with one LDS barrier each -- forward: [cell updates: layer 0 of step t + 1 and layer 1 of step t] |
// [half sweeps: layer 0 of step t + 2 and layer 1 of step t + 1]; backward: [exchanges + gate gradients: layer 1 of step s - 1
// and layer 0 of step s] | [half products: layer 1 of step s - 1 and layer 0 of step s] -- the layer that depends on nothing
// but itself runs one step ahead.  Backward products: block = (K half of the lane, which product, unit quad), the K quarters
// are added across lanes l / l ^ 32 and across the wave pair, and a 4-way ds_bpermute hands every cell owner its unit's
// recurrent / downward gradient.  Kept activations as in the 16-agent kernel; the roll-out / loss scan is chain_grad_group.
#define CLD_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

// The value the lane byte_lane / 4 holds.  By-value float on purpose: __builtin_bit_cast(int, v[r]) of an ext-vector ELEMENT reads
// element 0 whatever r is (hipcc 7.2), which silently turns a four-register exchange into four copies of register 0.
__device__ __forceinline__ float bperm(int byte_lane, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_lane, __builtin_bit_cast(int, v)));
}


// K sweep of the 4x4x1 products with the AGENTS' values as the broadcast A operand (rows = the four agents of a quad).
//   X0 / X1 (quad 0 / quad 1): register c holds, in the lanes of block b, four consecutive k-steps of k-group PER c + b for
//   row (lane & 3) -- ONE ds_read_b128 per register feeds PER = 16 (CB = 4) or 8 (CB = 3: one broadcast group per K half)
//   k-groups: the MFMA's cbsz / abid fields hand block `abid`'s rows to every block of its group.  With every lane fetching its
//   own copy of the operand a wave pulled 98 + 128 KB per step through the CU's 128 B/clk LDS port -- as long as the MFMAs
//   themselves take (profiles/r02/guide_kernel_notes.md) -- and spent an issue slot per 8 MFMAs.
//   B operand = the weights wt(g) (columns; one register per k, resident).  Accumulators [quad][k parity] in turn, so no MFMA
//   waits for the one before it.  (abid is an immediate, hence the compile-time recursion over the k-groups.)
template <int G, int NG, int CB, class WF>
__device__ __forceinline__ void bsweep(const v4f* X0, const v4f* X1, WF wt, v4f& a00, v4f& a01, v4f& a10, v4f& a11) {
    if constexpr (G < NG) {
        constexpr int PER = 1 << CB, c = G / PER, ab = G % PER;
        const v4f wv4 = wt(G);
        a00 = __builtin_amdgcn_mfma_f32_4x4x1f32(X0[c][0], wv4[0], a00, CB, ab, 0); a10 = __builtin_amdgcn_mfma_f32_4x4x1f32(X1[c][0], wv4[0], a10, CB, ab, 0);
        a01 = __builtin_amdgcn_mfma_f32_4x4x1f32(X0[c][1], wv4[1], a01, CB, ab, 0); a11 = __builtin_amdgcn_mfma_f32_4x4x1f32(X1[c][1], wv4[1], a11, CB, ab, 0);
        a00 = __builtin_amdgcn_mfma_f32_4x4x1f32(X0[c][2], wv4[2], a00, CB, ab, 0); a10 = __builtin_amdgcn_mfma_f32_4x4x1f32(X1[c][2], wv4[2], a10, CB, ab, 0);
        a01 = __builtin_amdgcn_mfma_f32_4x4x1f32(X0[c][3], wv4[3], a01, CB, ab, 0); a11 = __builtin_amdgcn_mfma_f32_4x4x1f32(X1[c][3], wv4[3], a11, CB, ab, 0);
        bsweep<G + 1, NG, CB>(X0, X1, wt, a00, a01, a10, a11);
    }
}

// Sum over the 16 lanes that share (lane & 3) -- the 16 hidden units of a wave for one agent -- left in all of them: two DPP row
// rotations inside the 16-lane rows, then the gfx950 row / half swaps (v_permlane16_swap, v_permlane32_swap: pure VALU, where
// __shfl_xor by 16 and 32 goes through the LDS crossbar and costs ~100 cycles of latency each on a dependent chain).
__device__ __forceinline__ float sum_over_units(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    const auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    const unsigned a16 = r16[0], b16 = r16[1];      // (scalars first: __builtin_bit_cast of a vector ELEMENT reads element 0, see bperm)
    x = __builtin_bit_cast(float, a16) + __builtin_bit_cast(float, b16);      // rows 0 + 1 | rows 2 + 3
    const auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    const unsigned a32 = r32[0], b32 = r32[1];
    return __builtin_bit_cast(float, a32) + __builtin_bit_cast(float, b32);   // lower + upper half
}

// x + (the value lane ^ 32 holds), in both lanes: one v_permlane32_swap and one add, no LDS crossbar
__device__ __forceinline__ float add_other_half(float x) {
    const auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    const unsigned a32 = r32[0], b32 = r32[1];
    return __builtin_bit_cast(float, a32) + __builtin_bit_cast(float, b32);
}

// 4 x 4 transpose across the four lanes of a quad: out[g] = register (lane & 3) of the quad's lane g.  Two butterfly stages
// (lane bit 0 with register bit 0, then bit 1 with bit 1), each one select for what to send, one DPP quad permute, two selects.
__device__ __forceinline__ float quad_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false)); }
__device__ __forceinline__ float quad_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false)); }
__device__ __forceinline__ v4f quad_transpose(const v4f& x, bool odd1, bool odd2) {
    const float x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    const float r01 = quad_xor1(odd1 ? x0 : x1), r23 = quad_xor1(odd1 ? x2 : x3);
    const float y0 = odd1 ? r01 : x0, y1 = odd1 ? x1 : r01, y2 = odd1 ? r23 : x2, y3 = odd1 ? x3 : r23;
    const float r02 = quad_xor2(odd2 ? y0 : y2), r13 = quad_xor2(odd2 ? y1 : y3);
    return v4f{odd2 ? r02 : y0, odd2 ? r13 : y1, odd2 ? y2 : r02, odd2 ? y3 : r13};
}

namespace gq {
// row strides = 16 (mod 64 banks): the four agent rows a wave touches with one instruction -- 16 consecutive units each in the
// scalar writes, four 16-byte k-groups each in the first 16 lanes of a b128 read -- land on disjoint banks
constexpr int AG = 8, HS = 80, GS = 272;
constexpr int ACTS = GT * 2 * 5 * 512;       // floats of kept activations per workgroup: [t][layer][i f g o c][thread]
}  // namespace gq

__global__ __launch_bounds__(512) void guide_quad_kernel(const DecoderWeights w, const DynParams d, const GuideArgs a) {
    using namespace gq;
    __shared__ __attribute__((aligned(16))) float hs[2][2][AG][HS];     // [layer][parity][agent][unit]
    __shared__ __attribute__((aligned(16))) float dG[2][AG][GS];        // gate gradients of layer 1 / layer 0
    __shared__ __attribute__((aligned(16))) float zin[AG][208];
    __shared__ __attribute__((aligned(16))) float condm[AG][256];
    __shared__ __attribute__((aligned(16))) float hpart[4][AG][HS];     // cond2hidden: K-quarter partials
    __shared__ __attribute__((aligned(16))) float xch[2][8][64][4];     // [layer][wave][lane]: partial sums for the partner wave's quad
    __shared__ float actp[GT][2][4][AG];     // [step][output][unit group][agent]: partials of hid2act
    __shared__ float act[2][GT][AG];         // (acceleration, yaw-rate), scaled
    __shared__ float dact[AG][2][GT];
    __shared__ float chs[AG][CG_ROWS * 54];  // roll-out scratch of chain_grad_group
    __shared__ float dz[AG][208];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wv8 & 3, kp = wv8 >> 2;          // unit group; K half = the agent quad whose cells this wave updates
    const int ul = lane >> 2, q = lane & 3;         // MFMA lane roles: block ul; A row q (agent of a quad); B / D column q (gate of unit ul)
    const int u = 16 * wq + ul;                     // this lane's hidden unit; after the quad transpose its cell: (u, agent 4 kp + q)
    const int ao = 4 * kp + q;
    const float wa0 = w.w_h2a[u], wa1 = w.w_h2a[64 + u], bh2a = w.b_h2a[0], bh2b = w.b_h2a[1];

    const int ngroups = (a.B + AG - 1) / AG;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int b0 = grp * AG;
        GSTAMP(0);
        const __amdgpu_buffer_rsrc_t keep = __builtin_amdgcn_make_buffer_rsrc(a.scratch + (size_t)blockIdx.x * ACTS, 0, ACTS * 4, 0x00020000);
        const int kvo = tid * 4;
        auto kput = [&](int slot, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), keep, kvo, slot * 2048, 0); };
        auto kget = [&](int slot) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(keep, kvo, slot * 2048, 0)); };
        auto agent = [&](int ag) { return (b0 + ag < a.B) ? b0 + ag : a.B - 1; };      // tail slots replay the last agent; never stored
        for (int i = tid; i < AG * 256; i += 512) condm[i >> 8][i & 255] = a.cond[(size_t)agent(i >> 8) * 256 + (i & 255)];
        for (int i = tid; i < AG * 208; i += 512) zin[i / 208][i % 208] = a.mean[(size_t)agent(i / 208) * 208 + i % 208];
        __syncthreads();
        // a.act_in: the BACKWARD half of a split call -- the forward sweep of this very group ran in an earlier launch (launch_guide_forward:
        // same grid, same scratch slot, same mean), its kept activations are in the scratch and its actions in act_in: nothing of the
        // forward pass is repeated
        const bool backward_only = a.act_in != nullptr;
        float c0 = 0.f, c1 = 0.f;
        GPHASE_DECL;
        if (!backward_only) {
        if (wv8 < 4) {   // h0 = cond2hidden(cond) for both layers (lstm_vae.py:46-49): wave wv8 takes the K quarter 64 wv8 .. + 63 of all 64 units
            const float* wr = w.w_c2h + (size_t)(4 * ul + q) * 256 + 64 * wv8;
            asm volatile("" : "+v"(wr));
            v4f p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0, p2 = p0, p3 = p0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v4f wa = *reinterpret_cast<const v4f*>(wr + 4 * j);
                const v4f ca = *reinterpret_cast<const v4f*>(&condm[q][64 * wv8 + 4 * j]);
                const v4f cb = *reinterpret_cast<const v4f*>(&condm[4 + q][64 * wv8 + 4 * j]);
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    p0 = CLD_MFMA4(wa[e], ca[e], p0); p1 = CLD_MFMA4(wa[e], cb[e], p1);
                    p2 = CLD_MFMA4(wa[e + 1], ca[e + 1], p2); p3 = CLD_MFMA4(wa[e + 1], cb[e + 1], p3);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {            // D lane (unit quad ul, agent q): register r = unit 4 ul + r
                hpart[wv8][q][4 * ul + r] = p0[r] + p2[r];
                hpart[wv8][4 + q][4 * ul + r] = p1[r] + p3[r];
            }
        }
        __syncthreads();
        {
            const int ag = tid >> 6, uu = tid & 63;
            const float v = w.b_c2h[uu] + hpart[0][ag][uu] + hpart[1][ag][uu] + hpart[2][ag][uu] + hpart[3][ag][uu];
            hs[0][0][ag][uu] = v;
            hs[1][0][ag][uu] = v;
        }
        __syncthreads();
        GSTAMP(1);
        // ---------------- forward ----------------
        {
            // weights as B operands (columns = the four gates of this lane's unit): this wave's K half, one register per k
            //   layer 0 (K = 4 + 64): kp 0: x_t (4) and h0 units 0..31;  kp 1: h0 units 32..63 (its group 0 carries zero weights)
            //   layer 1 (K = 64 + 64): kp 0: W_ih1 (h0_t);               kp 1: W_hh1 (h1_{t-1})
            const size_t row = (size_t)(64 * q + u);
            const float *p0a = w.w_ih0 + row * 4, *p0b = w.w_hh0 + row * 64 + 32 * kp, *p1 = (kp ? w.w_hh1 : w.w_ih1) + row * 64, *p_b0 = w.b0 + u, *p_b1 = w.b1 + u;
            asm volatile("" : "+v"(p0a), "+v"(p0b), "+v"(p1), "+v"(p_b0), "+v"(p_b1));
            v4f W0[9], W1[16];
            W0[0] = *reinterpret_cast<const v4f*>(p0a);
            if (kp) W0[0] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) W0[1 + j] = *reinterpret_cast<const v4f*>(p0b + 4 * j);
#pragma unroll
            for (int j = 0; j < 16; ++j) W1[j] = *reinterpret_cast<const v4f*>(p1 + 4 * j);
            // biases: this lane's column is gate q of unit u; D register r = agent r of the quad; added once, by the kp 0 half
            const float bq0 = kp ? 0.f : p_b0[64 * q], bq1 = kp ? 0.f : p_b1[64 * q];
            const v4f bias0 = {bq0, bq0, bq0, bq0}, bias1 = {bq1, bq1, bq1, bq1};
            const v4f zero = {0.f, 0.f, 0.f, 0.f};
            v4f keep0, keep1;                 // this wave's half of its OWN quad's pre-activations (gate q x four agents)
            // A operands: lane (block b, row q) reads agent 4 aq + q, k-group 8 kp + (b & 7) (layer 0: of the 16 groups of h0) or
            // k-group b (layer 1: all 16 groups of the matrix this wave holds); see bsweep
            auto half0 = [&](int t) {         // layer 0, step t: reads x_t and hs[0][t & 1]
                const int pr = t & 1;
                const v4f Xx[2] = {*reinterpret_cast<const v4f*>(&zin[q][4 * t]), *reinterpret_cast<const v4f*>(&zin[4 + q][4 * t])};
                const v4f Xh[2] = {*reinterpret_cast<const v4f*>(&hs[0][pr][q][32 * kp + 4 * (ul & 7)]), *reinterpret_cast<const v4f*>(&hs[0][pr][4 + q][32 * kp + 4 * (ul & 7)])};
                v4f a00 = bias0, a01 = zero, a10 = bias0, a11 = zero;
                bsweep<0, 1, 4>(&Xx[0], &Xx[1], [&](int) { return W0[0]; }, a00, a01, a10, a11);
                bsweep<0, 8, 4>(&Xh[0], &Xh[1], [&](int g) { return W0[1 + g]; }, a00, a01, a10, a11);
                const v4f P0 = a00 + a01, P1 = a10 + a11;
                *reinterpret_cast<v4f*>(&xch[0][wv8][lane][0]) = kp ? P0 : P1;      // the quad this wave does not own
                keep0 = kp ? P1 : P0;
            };
            auto half1 = [&](int t) {         // layer 1, step t: kp 0 reads h0_t = hs[0][(t & 1) ^ 1], kp 1 reads h1_{t-1} = hs[1][t & 1]
                const float* src = &hs[0][0][0][0] + (kp ? 2 * AG * HS + (t & 1) * AG * HS : ((t & 1) ^ 1) * AG * HS);
                const v4f X[2] = {*reinterpret_cast<const v4f*>(src + q * HS + 4 * ul), *reinterpret_cast<const v4f*>(src + (4 + q) * HS + 4 * ul)};
                v4f a00 = bias1, a01 = zero, a10 = bias1, a11 = zero;
                bsweep<0, 16, 4>(&X[0], &X[1], [&](int g) { return W1[g]; }, a00, a01, a10, a11);
                const v4f P0 = a00 + a01, P1 = a10 + a11;
                *reinterpret_cast<v4f*>(&xch[1][wv8][lane][0]) = kp ? P0 : P1;
                keep1 = kp ? P1 : P0;
            };
            auto cell0 = [&](int t) {
                const v4f P = quad_transpose(keep0 + *reinterpret_cast<const v4f*>(&xch[0][wv8 ^ 4][lane][0]), q & 1, q & 2);   // -> agent q x four gates
                const float i_ = fsig(P[0]), f_ = fsig(P[1]), g_ = ftanh(P[2]), o_ = fsig(P[3]);
                const float c = f_ * c0 + i_ * g_;
                c0 = c;
                hs[0][(t & 1) ^ 1][ao][u] = o_ * ftanh(c);
                const int ks = (t * 2 + 0) * 5;      // slot of (step t, layer 0, gate 0); a slot = one float per thread
                kput(ks + 0, i_); kput(ks + 1, f_); kput(ks + 2, g_); kput(ks + 3, o_); kput(ks + 4, c);
            };
            auto cell1 = [&](int t) {
                const v4f P = quad_transpose(keep1 + *reinterpret_cast<const v4f*>(&xch[1][wv8 ^ 4][lane][0]), q & 1, q & 2);
                const float i_ = fsig(P[0]), f_ = fsig(P[1]), g_ = ftanh(P[2]), o_ = fsig(P[3]);
                const float c = f_ * c1 + i_ * g_;
                c1 = c;
                const float hn = o_ * ftanh(c);
                hs[1][(t & 1) ^ 1][ao][u] = hn;
                const int ks = (t * 2 + 1) * 5;
                kput(ks + 0, i_); kput(ks + 1, f_); kput(ks + 2, g_); kput(ks + 3, o_); kput(ks + 4, c);
                actp[t][0][wq][ao] = sum_over_units(hn * wa0);      // hid2act: partials over this wave's 16 units
                actp[t][1][wq][ao] = sum_over_units(hn * wa1);      // (the 16 lanes of a column hold the same sum)
            };
            GSTAMP(2);
            half0(0);
            lds_barrier();
            cell0(0);
            lds_barrier();
            half0(1); half1(0);
            lds_barrier();
            GPHASE_START;
            for (int t = 0; t < GT; ++t) {
                if (t + 1 < GT) cell0(t + 1);
                GPHASE(0);
                cell1(t);
                GPHASE(1);
                lds_barrier();
                GPHASE(2);
                if (t + 2 < GT) half0(t + 2);
                GPHASE(3);
                if (t + 1 < GT) half1(t + 1);
                GPHASE(4);
                lds_barrier();
                GPHASE(5);
            }
        }
        GSTAMP(3);
        // ---------------- speed chain + loss gradient (diffuser_helpers.py:573-600; guidance_loss.py:229-254) ----------------
        for (int i = tid; i < 2 * GT * AG; i += 512) {
            const int o = i / (GT * AG), t = (i / AG) % GT, ag = i % AG;
            act[o][t][ag] = (o ? bh2b : bh2a) + actp[t][o][0][ag] + actp[t][o][1][ag] + actp[t][o][2][ag] + actp[t][o][3][ag];
        }
        } else {
            for (int i = tid; i < 2 * GT * AG; i += 512) {
                const int ag = i / (2 * GT), t = (i >> 1) % GT, o = i & 1;
                act[o][t][ag] = a.act_in[((size_t)agent(ag) * GT + t) * 2 + o];
            }
        }
        __syncthreads();
        if (a.act_out) {       // forward only: the decoder's scaled actions of this group (the caller rolls them out); see launch_guide_forward
            for (int i = tid; i < 2 * GT * AG; i += 512) {
                const int ag = i / (2 * GT), t = (i >> 1) % GT, o = i & 1;
                if (b0 + ag < a.B) a.act_out[((size_t)(b0 + ag) * GT + t) * 2 + o] = act[o][t][ag];
            }
            __syncthreads();
            continue;
        }
        chain_grad_group<AG, 512>(d, a, agent, &act[0][0][0], &act[1][0][0], AG, &dact[0][0][0], &chs[0][0]);
        GSTAMP(4);
        // ---------------- backward through time ----------------
        {
            // B operands of the products, pre-packed at cld_finalize (DecoderWeights::gqfrag): lane (block = (K half kh, product m, unit
            // quad ub), column j) holds W_m[gate column kk + 128 kh][unit 16 wq + 4 ub + j] for kk = 0..127 (layer 1: m = 0 -> W_hh1,
            // m = 1 -> W_ih1; layer 0: m = 0 -> W_hh0, m = 1 / ub = 0 -> W_ih0[.][latent channel j], else 0);
            // of the 32 k-groups of a lane this wave takes groups 16 kp .. 16 kp + 15, i.e. gate columns 128 kh + 64 kp + 4 g + e --
            // the four K quarters are added across lanes l / l ^ 32 and across the two waves
            const v4f* gf = reinterpret_cast<const v4f*>(w.gqfrag) + (size_t)wq * (2 * 32 * 64) + (size_t)kp * (16 * 64) + lane;
            asm volatile("" : "+v"(gf));
            v4f t1[16], t0[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) { t1[j] = gf[j * 64]; t0[j] = gf[(32 + j) * 64]; }
            const int kh = lane >> 5;
            // owner lane (unit ul, agent q) <- result lane (kh 0, product m, unit quad ul >> 2, column ul & 3) = lane 16 m + ul, register q
            const int src_rec = ul * 4, src_dwn = src_rec + 64, rsel = q;
            auto pick = [&](const v4f& v, int byte_lane) {
                const float x0 = bperm(byte_lane, v[0]), x1 = bperm(byte_lane, v[1]), x2 = bperm(byte_lane, v[2]), x3 = bperm(byte_lane, v[3]);
                return rsel == 0 ? x0 : (rsel == 1 ? x1 : (rsel == 2 ? x2 : x3));
            };
            v4f keepS1 = {0.f, 0.f, 0.f, 0.f}, keepS0 = keepS1;
            // half product of layer `lay` (0: gate gradients dG[0] of LSTM layer 1, 1: dG[1] of LSTM layer 0): K quarters of this lane
            // and wave; the in-wave halves are added here, the partner's half in the exchange
            auto halfprod = [&](int lay, const v4f (&tw)[16], v4f& keepS) {
                // A operand: lane (kh, block b of the half, row q) reads the four columns of k-group 8 c + b of this wave's K quarter
                const float* gq0 = &dG[lay][q][128 * kh + 64 * kp + 4 * (ul & 7)];
                const float* gq1 = &dG[lay][4 + q][128 * kh + 64 * kp + 4 * (ul & 7)];
                const v4f X0[2] = {*reinterpret_cast<const v4f*>(gq0), *reinterpret_cast<const v4f*>(gq0 + 32)};
                const v4f X1[2] = {*reinterpret_cast<const v4f*>(gq1), *reinterpret_cast<const v4f*>(gq1 + 32)};
                v4f p00 = {0.f, 0.f, 0.f, 0.f}, p01 = p00, p10 = p00, p11 = p00;
                bsweep<0, 16, 3>(X0, X1, [&](int g) { return tw[g]; }, p00, p01, p10, p11);
                v4f S0 = p00 + p01, S1 = p10 + p11;
#pragma unroll
                for (int r = 0; r < 4; ++r) { S0[r] = add_other_half(S0[r]); S1[r] = add_other_half(S1[r]); }
                *reinterpret_cast<v4f*>(&xch[lay][wv8][lane][0]) = kp ? S0 : S1;
                keepS = kp ? S1 : S0;
            };
            float rec1 = 0.f, rec0 = 0.f, down = 0.f, dc1n = 0.f, dc0n = 0.f;
            float kv1[6], kv0[6];
            auto fetch = [&](float (&kv)[6], int t, int layer) {
                const int ks = (t * 2 + layer) * 5;
#pragma unroll
                for (int k = 0; k < 5; ++k) kv[k] = kget(ks + k);
                kv[5] = t > 0 ? kget(ks - 10 + 4) : 0.f;      // the cell state of step t-1, same layer; c_{-1} = 0 (t is wave-uniform)
            };
            auto grads = [&](const float (&kv)[6], float dh, float& dcn, float* row) {
                const float i_ = kv[0], f_ = kv[1], g_ = kv[2], o_ = kv[3], c = kv[4], cp = kv[5];
                const float tc = ftanh(c);
                const float dc = dh * o_ * (1.f - tc * tc) + dcn;
                row[0] = dc * g_ * i_ * (1.f - i_);
                row[64] = dc * cp * f_ * (1.f - f_);
                row[128] = dc * i_ * (1.f - g_ * g_);
                row[192] = dh * tc * o_ * (1.f - o_);
                dcn = dc * f_;
            };
            auto grad1 = [&](int s) { grads(kv1, wa0 * dact[ao][0][s] + wa1 * dact[ao][1][s] + rec1, dc1n, &dG[0][ao][u]); };
            auto grad0 = [&]() { grads(kv0, down + rec0, dc0n, &dG[1][ao][u]); };
            auto xchg1 = [&]() {      // layer-1 product of step s complete -> rec1 (for step s - 1), down (for layer 0, step s)
                const v4f S = keepS1 + *reinterpret_cast<const v4f*>(&xch[0][wv8 ^ 4][lane][0]);
                rec1 = pick(S, src_rec);
                down = pick(S, src_dwn);
            };
            auto xchg0 = [&](int s) { // layer-0 product of step s complete -> rec0 (for step s - 1), dL/dz_s
                const v4f S = keepS0 + *reinterpret_cast<const v4f*>(&xch[1][wv8 ^ 4][lane][0]);
                rec0 = pick(S, src_rec);
                // dL/dz_s, complete: result lanes (kh 0, product 1, unit quad 0, column = latent channel q) = lanes 16..19, register r = agent r
                if (wq == 0 && lane >= 16 && lane < 20) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dz[4 * kp + r][4 * s + q] = S[r];
                }
            };
            GSTAMP(5);
            // prologue: layer 1 at the last step
            fetch(kv1, GT - 1, 1);
            fetch(kv0, GT - 1, 0);
            grad1(GT - 1);
            lds_barrier();
            fetch(kv1, GT - 2, 1);
            halfprod(0, t1, keepS1);
            lds_barrier();
            GPHASE_START;
            for (int s = GT - 1; s >= 0; --s) {
                // phase A: exchanges and gate gradients
                xchg1();                                   // layer-1 product of step s
                GPHASE(6);
                if (s > 0) grad1(s - 1);
                GPHASE(7);
                if (s < GT - 1) xchg0(s + 1);              // layer-0 product of step s + 1
                GPHASE(8);
                grad0();                                   // layer 0, step s (kv0 = step s)
                GPHASE(9);
                lds_barrier();
                GPHASE(10);
                // phase B: half products
                if (s > 1) fetch(kv1, s - 2, 1);
                if (s > 0) fetch(kv0, s - 1, 0);
                if (s > 0) halfprod(0, t1, keepS1);        // layer 1, step s - 1
                GPHASE(11);
                halfprod(1, t0, keepS0);                   // layer 0, step s
                GPHASE(12);
                lds_barrier();
                GPHASE(13);
            }
            xchg0(0);
            GPHASE_PRINT(14);
        }
        GSTAMP(6);
        __syncthreads();
        // ---------------- one optimiser step on the mean (clipped if asked); then the ancestral noise ----------------
        for (int i = tid; i < AG * 208; i += 512) {
            const int ag = i / 208, r = i % 208, b = b0 + ag;
            if (b >= a.B) continue;
            const float g = dz[ag][r];
            const float mu = optimiser_step(a, g, zin[ag][r], (size_t)b * 208 + r);
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

static int guide_mfma_grid(int B) {
    const int groups = (B + gm8::AG - 1) / gm8::AG;
    return groups < 256 ? groups : 256;
}
static int guide_quad_grid(int B) {
    const int groups = (B + gq::AG - 1) / gq::AG;
    return groups < 256 ? groups : 256;
}
// Which formulation by batch size.  The MFMA kernels put one workgroup on a CU and each workgroup is bound by its own instruction
// issue, so what counts is the number of ROUNDS of workgroups over the 256 CUs: the 8-agent kernel (a round costs ~0.63 of a
// 16-agent round: 318 vs 506 us) wins whenever it needs fewer than 1.5 x the rounds of the 16-agent kernel -- up to 2,048 agents
// (one round each), not from 2,049 to 4,096 (two rounds against one), ...; a single round of the 2-agent VALU kernel
// (up to 512 agents) takes as long as a round of the 8-agent kernel (340 vs 318 us), so it only keeps the smallest batches.
// Tests force each form through cld_debug_force_kernel.
static int guide_form(int B, int form) {
    if (form != FORM_AUTO) return form;
    if (B < 64) return FORM_VALU;
    const int r8 = (B + 8 * 256 - 1) / (8 * 256), r16 = (B + 16 * 256 - 1) / (16 * 256);
    return 2 * r8 < 3 * r16 ? FORM_MFMA_QUAD : FORM_MFMA;
}
size_t guide_scratch_floats(int B) {
    const size_t valu = (size_t)guide_grid(B) * GNA * (G_GATES + G_CELLS);
    const size_t mfma = (size_t)guide_mfma_grid(B) * gm8::ACTS, quad = (size_t)guide_quad_grid(B) * gq::ACTS;
    return valu > mfma ? (valu > quad ? valu : quad) : (mfma > quad ? mfma : quad);
}

// The forward half of the 8-agent kernel as a decoder: 256 workgroups at 2,048 agents where decode_mfma_kernel (16 agents per
// workgroup) fills half the chip.  Used for the plans the scene-coupled guidance losses are evaluated on (cld_api.hip
// run_guidance), where a decode precedes every guidance-kernel launch.  False: this batch size does not take the 8-agent form.
bool guide_forward_available(int B, int form) { return guide_form(B, form) == FORM_MFMA_QUAD; }
bool guide_split_available(int B, int form) { return guide_form(B, form) == FORM_MFMA_QUAD && (B + gq::AG - 1) / gq::AG <= guide_quad_grid(B); }
hipError_t launch_guide_forward(const DecoderWeights& w, const DynParams& d, const GuideArgs& a, hipStream_t s) {
    if (!a.act_out || a.act_in) return hipErrorInvalidValue;
    hipLaunchKernelGGL(guide_quad_kernel, dim3(guide_quad_grid(a.B)), dim3(512), 0, s, w, d, a);
    return hipGetLastError();
}

hipError_t launch_guide(const DecoderWeights& w, const DynParams& d, const GuideArgs& a, hipStream_t s, int form) {
    if (a.act_in && !guide_split_available(a.B, form)) return hipErrorInvalidValue;
    switch (guide_form(a.B, form)) {
        case FORM_MFMA_QUAD: hipLaunchKernelGGL(guide_quad_kernel, dim3(guide_quad_grid(a.B)), dim3(512), 0, s, w, d, a); break;
        case FORM_MFMA: hipLaunchKernelGGL(guide_mfma8_kernel, dim3(guide_mfma_grid(a.B)), dim3(512), 0, s, w, d, a); break;
        default: hipLaunchKernelGGL(guide_kernel, dim3(guide_grid(a.B)), dim3(256), 0, s, w, d, a);
    }
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

// Values of the built-in guidance losses on decoded trajectories (upstream guide_losses, guidance_loss.py:2143-2172): one wave
// per agent, lane t = timestep t, wave reductions.
__global__ __launch_bounds__(64) void guide_loss_kernel(const GuideArgs a, const float* __restrict__ traj, float* __restrict__ losses) {
    const int b = blockIdx.x, t = threadIdx.x;
    const float nan = __builtin_nanf("");
    const bool on_ts = a.target_speed && (!a.loss_scale || a.loss_scale[b] != 0.f);
    const bool on_sl = a.speed_limit_scale && a.speed_limit_scale[b] != 0.f;
    const bool on_al = a.acc_limit_scale && a.acc_limit_scale[b] != 0.f;
    const bool on_tp = a.target_pos_scale && a.target_pos_scale[b] != 0.f;
    float ts = 0.f, sl = 0.f, al = 0.f, x = 0.f, y = 0.f;
    if (t < 52) {
        const float* p = traj + ((size_t)b * 52 + t) * 6;
        x = p[0]; y = p[1];
        const float v = p[2], acc = p[4];
        if (on_ts) { const float d = fabsf(v - a.target_speed[(size_t)b * 52 + t]); ts = (d == d) ? d : 0.f; }     // nan_to_num(nan = 0)
        if (on_sl) sl = fmaxf(fabsf(v) - a.speed_limit, 0.f);
        if (on_al) al = fmaxf(fabsf(acc) - a.acc_limit, 0.f);
    }
    float tp = 0.f;
    if (on_tp) {
        const float ex = x - a.target_pos[2 * b], ey = y - a.target_pos[2 * b + 1];
        const float dd = sqrtf(ex * ex + ey * ey);
        int tstar = a.target_time[b];
        if (tstar >= 0) {
            tstar = tstar > 51 ? 51 : tstar;
            tp = __shfl(dd, tstar);
        } else {
            int m = -tstar - 1;
            m = m > 51 ? 51 : m;
            const bool in = t >= m && t < 52;
            float dmin = in ? dd : 3.4e38f;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) dmin = fminf(dmin, __shfl_xor(dmin, o));
            const float e = in ? expf(-(dd - dmin)) : 0.f;
            float z = e, S = e * dd * dd;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { z += __shfl_xor(z, o); S += __shfl_xor(S, o); }
            tp = S / z / (float)(52 - m);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { ts += __shfl_xor(ts, o); sl += __shfl_xor(sl, o); al += __shfl_xor(al, o); }
    if (t == 0) {
        float* o = losses + (size_t)b * 4;
        o[0] = on_ts ? ts * (1.0f / 52.0f) : nan;
        o[1] = on_sl ? sl * (1.0f / 52.0f) : nan;
        o[2] = on_al ? al * (1.0f / 52.0f) : nan;
        o[3] = on_tp ? tp : nan;
    }
}

hipError_t launch_guide_losses(const GuideArgs& a, const float* traj, float* losses, hipStream_t s) {
    hipLaunchKernelGGL(guide_loss_kernel, dim3(a.B), dim3(64), 0, s, a, traj, losses);
    return hipGetLastError();
}

hipError_t launch_reward(const RewardArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(reward_kernel, dim3(a.B), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace cld
