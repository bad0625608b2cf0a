// collision_kernels.hip -- upstream's AgentCollisionLoss (src/tbsim/utils/guidance_loss.py:442-630) and its gradient w.r.t. the
// decoded plans, as one launch: the scene-coupled guidance loss of "controllable" traffic simulation, which the round-2 library
// left to caller torch code (a host round trip per denoising step).
//
// Upstream builds, for every (sample, step), the B x B x 25 table of disk-centre distances of ALL agent pairs of the batch and
// masks it down to the pairs of one scene.  Here a workgroup owns four agents of one (scene, sample): the world poses of ALL the
// scene's agents at all 52 steps sit in its LDS (16 B per pose; every workgroup of the scene re-derives them from the plans --
// 3,328 sincos / atan2 for a 64-agent scene, far cheaper than a second launch), one thread per (own agent, step) walks the other
// agents of the scene, rejects far pairs on the centre distance (exact: a disk centre lies within length / 2 - radius of the
// agent's centre) and evaluates the 25 disk pairs of the rest; an agent's 52 per-step sums stay inside one workgroup, so its
// value is added up in a fixed order (deterministic, no atomics).  A 64-agent scene is 16 workgroups, BASELINE configs[2]'s 32
// scenes 512: the first form of this kernel (one workgroup per scene) took 880 us there.  Value and gradient come out of the
// same pass: the loss is
//     value[i] = [moving_i] sum_t w_t (1 / B) sum_j pen_ij(t),  pen = 1 - d_ij / (r_i + r_j + buffer)  where d_ij <= that bound,
//     total    = sum_scenes weight_s * mean over the scene's guided agents (and samples) of value   (DiffuserGuidance, :2143-2172),
// pen is symmetric in (i, j), agents that are stationary or not guided are detached (:512-534), so
//     d total / d pose_i = [i guided and moving] * weight_s / (M_s N B) * sum_t w_t sum_j (1 + [j guided and moving]) d pen_ij / d pose_i.
// Output grad [B N, 52, 6] is dL/d(descaled trajectory) -- the `ext_grad` operand of the guidance kernels (guide_kernels.hip),
// which turn it into dL/d(latent) through roll-out and decoder.
#include "cld_kernels.h"

namespace cld {

namespace {
constexpr int TT = 52;
constexpr int kMaxDisks = 8;
constexpr int kOwn = 4;                    // agents whose (agent, step) items one workgroup evaluates: 4 x 52 = 208 of its 256 threads

// torch.linspace(-e, e, D)[a] as ATen computes it: from the start below the midpoint, from the end above it (init_disks, :481-492)
__device__ __forceinline__ float disk_centre(float e, int a, int D) {
    if (D <= 1) return -e;
    const float step = (e - (-e)) / (float)(D - 1);
    return a < D / 2 ? -e + step * (float)a : e - step * (float)(D - 1 - a);
}

__global__ __launch_bounds__(256) void agent_collision_kernel(const CollisionArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.x / p.num_samp, n = blockIdx.x % p.num_samp;
    const int a0 = p.scene_start[s], A = p.scene_start[s + 1] - a0;
    const int i0 = blockIdx.y * kOwn;                               // this workgroup's agents: i0 .. i0 + kOwn - 1 of the scene
    const int tid = threadIdx.x;
    if (A > p.max_scene_agents) {
        // the scene does not fit the launch (LDS and grid.y were sized from the host's max_scene_agents; scene_start is device data
        // the host cannot check): no partial result -- every agent of the scene gets a NaN value and the gradient passed in (or 0)
        if (blockIdx.y == 0)
            for (int idx = tid; idx < A * TT; idx += 256) {
                const int i = idx / TT, t = idx - i * TT;
                if (t == 0 && p.loss) p.loss[(size_t)(a0 + i) * p.num_samp + n] = __builtin_nanf("");
                if (p.grad) {
                    const size_t o = ((size_t)((a0 + i) * p.num_samp + n) * TT + t) * 6;
                    for (int k = 0; k < 6; ++k) p.grad[o + k] = p.grad_in ? p.grad_in[o + k] : 0.f;
                }
            }
        return;
    }
    if (i0 >= A) return;
    float4* pose = reinterpret_cast<float4*>(lds);                 // [A][52]: world x, y, cos / sin of the world heading
    float* part = lds + (size_t)A * TT * 4;                        // [kOwn][52]: sum_j pen_ij(t) (weighted) of the own agents -> their values
    float* agent = part + (size_t)kOwn * TT;                       // [A][4]: radius, half extent of the disk centres, row-active flag, moving flag
    int& n_guided = *reinterpret_cast<int*>(agent + (size_t)A * 4);   // (in the dynamic allocation: a static __shared__ next to a 160-KB dynamic request is refused)
    const float wgt = p.scene_weight ? p.scene_weight[s] : 1.0f;

    if (tid == 0) n_guided = 0;
    __syncthreads();
    for (int i = tid; i < A; i += 256) {
        const int b = a0 + i;
        const float len = p.extent[b * 3 + 0], wid = p.extent[b * 3 + 1];
        const float rad = wid * 0.5f;
        const bool guided = wgt != 0.f && (!p.guided || p.guided[b]);
        const bool moving = fabsf(p.curr_speed[b]) > p.moving_speed_th;
        agent[i * 4 + 0] = rad;
        agent[i * 4 + 1] = len * 0.5f - rad;                       // disk centres run from -this to +this along the agent's axis (:481-492)
        agent[i * 4 + 2] = (guided && moving) ? 1.f : 0.f;
        agent[i * 4 + 3] = (moving ? 1.f : 0.f) + ((p.excluded && p.excluded[b]) ? 2.f : 0.f);      // bit 0: moving; bit 1: in `excluded_agents`
        if (guided) atomicAdd(&n_guided, 1);
    }
    // world poses (geometry_utils.py:458-483): p_w = R p + t; heading = atan2 of the rotated unit vector
    for (int it = tid; it < A * TT; it += 256) {
        const int i = it / TT, t = it - i * TT, b = a0 + i;
        const float* W = p.world_from_agent + (size_t)b * 9;
        const float* x = p.traj + ((size_t)(b * p.num_samp + n) * TT + t) * 6;
        float sy, cy;
        sincosf(x[3], &sy, &cy);
        const float hx = W[0] * cy + W[1] * sy, hy = W[3] * cy + W[4] * sy;
        const float yw = atan2f(hy, hx);
        float sw, cw;
        sincosf(yw, &sw, &cw);
        pose[it] = make_float4(W[0] * x[0] + W[1] * x[1] + W[2], W[3] * x[0] + W[4] * x[1] + W[5], cw, sw);
    }
    __syncthreads();

    // normalised step weights decay^t / sum (:614-616)
    float wsum = 0.f, wp = 1.f;
    for (int t = 0; t < TT; ++t) { wsum += wp; wp *= p.decay_rate; }
    const int D = p.num_disks;
    const float inv_b = 1.0f / (float)p.B_agents;
    const float coef = (n_guided > 0 && wgt != 0.f) ? wgt / ((float)n_guided * (float)p.num_samp * (float)p.B_agents) : 0.f;

    const int own = (A - i0 < kOwn ? A - i0 : kOwn) * TT;
    for (int io = tid; io < own; io += 256) {
        const int i = i0 + io / TT, t = io % TT, b = a0 + i, it = i * TT + t;
        const float4 pi = pose[it];
        const float ri = agent[i * 4 + 0], ei = agent[i * 4 + 1], acti = agent[i * 4 + 2];
        const bool excl_i = ((int)agent[i * 4 + 3] & 2) != 0;
        const float wt = powf(p.decay_rate, (float)t) / wsum;
        float cxi[kMaxDisks];
#pragma unroll
        for (int a = 0; a < kMaxDisks; ++a) cxi[a] = disk_centre(ei, a, D);
        float pen_sum = 0.f, gx = 0.f, gy = 0.f, gyaw = 0.f;
        for (int j = 0; j < A; ++j) {
            if (j == i) continue;
            if (excl_i && ((int)agent[j * 4 + 3] & 2)) continue;   // both in excluded_agents: the pair is not penalised (:586-593)
            const float4 pj = pose[j * TT + t];
            const float rj = agent[j * 4 + 0], ej = agent[j * 4 + 1];
            const float pd = ri + rj + p.buffer_dist;
            const float dx = pi.x - pj.x, dy = pi.y - pj.y;
            const float reach = pd + ei + ej + 1e-3f;
            if (dx * dx + dy * dy > reach * reach) continue;       // no disk pair can be within pd
            float best = 3.0e38f, bdx = 0.f, bdy = 0.f, bcx = 0.f;
            for (int a = 0; a < D; ++a) {
                const float ax = dx + cxi[a] * pi.z, ay = dy + cxi[a] * pi.w;
                for (int c = 0; c < D; ++c) {
                    const float cxj = disk_centre(ej, c, D);
                    const float ex = ax - cxj * pj.z, ey = ay - cxj * pj.w;
                    const float d2 = ex * ex + ey * ey;
                    if (d2 < best) { best = d2; bdx = ex; bdy = ey; bcx = cxi[a]; }      // first minimum in (a, c) order, as torch.min
                }
            }
            const float dist = sqrtf(best);
            if (dist <= pd) {
                pen_sum += 1.0f - dist / pd;
                if (acti != 0.f && dist > 0.f) {
                    // d pen / d c_i = -(c_i - c_j) / (dist pd); this pair is in row i's value and, when j's row counts, in row j's too
                    const float k = -(1.0f + agent[j * 4 + 2]) / (dist * pd);
                    const float gcx = k * bdx, gcy = k * bdy;
                    gx += gcx; gy += gcy;
                    gyaw += bcx * (-pi.w * gcx + pi.z * gcy);      // c_i = p_i + cx (cos, sin)(heading)
                }
            }
        }
        part[io] = pen_sum * wt * inv_b;
        if (p.grad) {
            float* g = p.grad + ((size_t)(b * p.num_samp + n) * TT + t) * 6;
            const float* gi = p.grad_in ? p.grad_in + ((size_t)(b * p.num_samp + n) * TT + t) * 6 : nullptr;
            const float sc = coef * wt * acti;
            const float* W = p.world_from_agent + (size_t)b * 9;
            // back through p_w = R p + t and heading = atan2(R (cos, sin)(yaw)): d heading / d yaw = (h x dh) / |h|^2
            const float* x = p.traj + ((size_t)(b * p.num_samp + n) * TT + t) * 6;
            float sy, cy;
            sincosf(x[3], &sy, &cy);
            const float hx = W[0] * cy + W[1] * sy, hy = W[3] * cy + W[4] * sy;
            const float dhx = -W[0] * sy + W[1] * cy, dhy = -W[3] * sy + W[4] * cy;
            const float dyaw = (hx * dhy - hy * dhx) / (hx * hx + hy * hy);
            float o[6] = {sc * (W[0] * gx + W[3] * gy), sc * (W[1] * gx + W[4] * gy), 0.f, sc * gyaw * dyaw, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 6; ++k) g[k] = o[k] + (gi ? gi[k] : 0.f);
        }
    }
    __syncthreads();
    if (p.loss) {
        for (int io = tid; io < own / TT; io += 256) {
            const int i = i0 + io;
            float v = 0.f;
            for (int t = 0; t < TT; ++t) v += part[io * TT + t];   // fixed order: deterministic
            p.loss[(size_t)(a0 + i) * p.num_samp + n] = ((int)agent[i * 4 + 3] & 1) ? v : 0.f;
        }
    }
}
}  // namespace

hipError_t launch_agent_collision(const CollisionArgs& a, hipStream_t s) {
    const int max_scene_agents = a.max_scene_agents;
    if (a.num_disks < 1 || a.num_disks > kMaxDisks || a.num_scenes < 1 || a.num_samp < 1 || max_scene_agents < 1) return hipErrorInvalidValue;
    const size_t lds_bytes = ((size_t)max_scene_agents * TT * 4 + (size_t)kOwn * TT + (size_t)max_scene_agents * 4 + 4) * sizeof(float);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;          // a scene of more than ~190 agents: not built
    static unsigned long long attr_done = 0;      // one bit per device (a second handle on another device sets it there too)
    if (hipError_t e = set_max_lds_once(reinterpret_cast<const void*>(agent_collision_kernel), 160 * 1024, &attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(agent_collision_kernel, dim3(a.num_scenes * a.num_samp, (max_scene_agents + kOwn - 1) / kOwn), dim3(256), lds_bytes, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// MapCollisionLoss (src/tbsim/utils/guidance_loss.py:717-875): agents should not leave the drivable area
// ---------------------------------------------------------------------------------------------------------------------
// One workgroup of 13 waves per plan (agent x sample), one wave per time step (4 steps each: with 4 waves x 13 steps the kernel
// was the serial chain of one wave's 13 steps -- 0.58 ms for a single 64-agent scene): the num_points_lw grid of sample points of
// the agent's box at that step's pose (:731-753) goes to LDS with its off-road flag (drivable map at the truncated, clamped
// raster pixel, :797-805); steps where some but not all points are off road (:807-809) then give every off-road point
// 1 - (distance to the nearest ON-road point of the box) / (box diagonal), with the on-road point carrying the gradient and the
// off-road one detached (:833-848) -- which makes d / d pose = -(p_i - p_j) / (d diag) applied at p_i = pos + R(yaw) loc_i.
// Upstream takes the distances from torch.cdist (its matrix-multiply path at 100 points: ~1e-4 m of rounding at these
// coordinates, which also enters its backward); here they are exact differences, so values agree to rounding and gradients to
// the reference's own ~0.3 % noise.
namespace {
constexpr int kMaxPts = 256;

__device__ __forceinline__ float unit_linspace(int a, int n) {      // torch.linspace(-0.5, 0.5, n)[a]
    if (n <= 1) return -0.5f;
    const float step = 1.0f / (float)(n - 1);
    return a < n / 2 ? -0.5f + step * (float)a : 0.5f - step * (float)(n - 1 - a);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int kMapWaves = 13;                           // 13 waves x 4 steps = the 52 steps of a plan

// nearest on-road candidate(s) of one off-road sample, as a mergeable partial result: the smallest squared distance seen and, over the
// candidates within 1e-5 (relative) of it, the sums the gradient needs.  On a regular grid mirror-image candidates are equidistant in
// exact arithmetic (an isolated off-road sample between two on-road neighbours): torch.amin's backward shares the gradient evenly among
// the minima it finds equal, which for mirror images cancels the translation part.  Candidates within 1e-5 of the minimum are treated as
// that tie -- the symmetric choice, instead of whichever rounding favours.
struct Nearest {
    float best, sx, sy, sw, cnt;
    __device__ __forceinline__ void init() { best = 3.0e38f; sx = sy = sw = cnt = 0.f; }
    __device__ __forceinline__ void add(float d2, float ex, float ey, float w) {
        if (d2 > best * (1.0f + 1e-5f)) return;
        if (d2 < best * (1.0f - 1e-5f)) { sx = 0.f; sy = 0.f; sw = 0.f; cnt = 0.f; }
        best = fminf(best, d2);
        sx += ex; sy += ey; sw += w; cnt += 1.f;
    }
    __device__ __forceinline__ void merge(float ob, float ox, float oy, float ow, float oc) {      // another lane's partial result
        if (ob > best * (1.0f + 1e-5f)) return;
        if (ob < best * (1.0f - 1e-5f)) { sx = 0.f; sy = 0.f; sw = 0.f; cnt = 0.f; }
        best = fminf(best, ob);
        sx += ox; sy += oy; sw += ow; cnt += oc;
    }
};

__global__ __launch_bounds__(64 * kMapWaves) void map_collision_kernel(const MapCollisionArgs p) {
    // per wave: the step's ON-road samples and OFF-road samples (x, y in the agent frame), compacted -- the search of an
    // off-road sample walks on-road candidates only, and the off-road samples fill the lanes from 0
    __shared__ float2 on_pts[kMapWaves][kMaxPts];
    __shared__ float2 off_pts[kMapWaves][kMaxPts];
    __shared__ float part[TT];
    __shared__ float s_coef;
    const int row = blockIdx.x, b = row / p.num_samp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = p.num_points_l * p.num_points_w;
    if (tid == 0) {                                     // the agent's scene: weight / (agents of the scene x samples)
        float c = 0.f;
        for (int s = 0; s < p.num_scenes; ++s)
            if (b >= p.scene_start[s] && b < p.scene_start[s + 1]) {
                const float w = p.scene_weight ? p.scene_weight[s] : 1.0f;
                c = w / ((float)(p.scene_start[s + 1] - p.scene_start[s]) * (float)p.num_samp);
            }
        s_coef = c;
    }
    __syncthreads();
    const bool moving = fabsf(p.curr_speed[b]) > p.moving_speed_th;
    const float len = p.extent[b * 3 + 0], wid = p.extent[b * 3 + 1];
    const float inv_diag = 1.0f / sqrtf(len * len + wid * wid);
    const float* R = p.raster_from_agent + (size_t)b * 9;
    const unsigned char* dm = p.drivable_map + (size_t)b * p.H * p.W;
    float wsum = 0.f, wp = 1.f;
    for (int t = 0; t < TT; ++t) { wsum += wp; wp *= p.decay_rate; }
    for (int t = wave; t < TT; t += kMapWaves) {
        const float* x = p.traj + ((size_t)row * TT + t) * 6;
        const float px = x[0], py = x[1];
        float sy, cy;
        sincosf(x[3], &sy, &cy);
        int n_on = 0, n_off = 0;                        // (wave-uniform)
        for (int k0 = 0; k0 < P; k0 += 64) {
            const int k = k0 + lane;
            bool off = false, valid = k < P;
            float ax = 0.f, ay = 0.f;
            if (valid) {
                const float lx = unit_linspace(k / p.num_points_w, p.num_points_l) * len, wy = unit_linspace(k % p.num_points_w, p.num_points_w) * wid;
                ax = lx * cy - wy * sy + px; ay = lx * sy + wy * cy + py;
                int ix = (int)(R[0] * ax + R[1] * ay + R[2]), iy = (int)(R[3] * ax + R[4] * ay + R[5]);      // .long(): truncation toward zero
                ix = min(max(ix, 0), p.W - 1); iy = min(max(iy, 0), p.H - 1);
                off = dm[(size_t)iy * p.W + ix] == 0;
            }
            // compaction in sample order (the order the candidates are walked in decides nothing but rounding-level ties)
            const unsigned long long m_off = __ballot(valid && off), m_on = __ballot(valid && !off);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (valid && off) off_pts[wave][n_off + __popcll(m_off & below)] = make_float2(ax, ay);
            if (valid && !off) on_pts[wave][n_on + __popcll(m_on & below)] = make_float2(ax, ay);
            n_off += __popcll(m_off); n_on += __popcll(m_on);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0): the wave's own LDS writes have landed; the lists are wave-private
        __builtin_amdgcn_wave_barrier();
        float loss = 0.f, gx = 0.f, gy = 0.f, gyaw = 0.f;
        if (n_off != 0 && n_on != 0 && moving) {
            // lanes = (off-road sample jj, part): JN = n_off rounded up to a power of two (<= 64) samples side by side, the 64 / JN parts
            // of a sample take every (64 / JN)-th candidate each and merge their partial results across lanes (lane ^ JN, ^ 2 JN, ...)
            int JN = 1;
            while (JN < n_off && JN < 64) JN <<= 1;
            const int S = 64 / JN, jl = lane & (JN - 1), prt = lane / JN;
            for (int j0 = 0; j0 < n_off; j0 += JN) {
                const int j = j0 + jl;
                const bool act = j < n_off;
                const float2 pj = off_pts[wave][act ? j : 0];
                Nearest nr;
                nr.init();
                for (int i = prt; i < n_on; i += S) {
                    const float2 pi = on_pts[wave][i];
                    const float ex = pi.x - pj.x, ey = pi.y - pj.y;
                    nr.add(ex * ex + ey * ey, ex, ey, -ex * (pi.y - py) + ey * (pi.x - px));      // d p_i / d yaw = perp(p_i - pos)
                }
                for (int o = JN; o < 64; o <<= 1)
                    nr.merge(__shfl_xor(nr.best, o), __shfl_xor(nr.sx, o), __shfl_xor(nr.sy, o), __shfl_xor(nr.sw, o), __shfl_xor(nr.cnt, o));
                if (act && prt == 0) {
                    const float d = sqrtf(nr.best);
                    loss += 1.0f - d * inv_diag;
                    if (d > 0.f) {
                        const float k = -inv_diag / (d * nr.cnt);
                        gx += k * nr.sx; gy += k * nr.sy; gyaw += k * nr.sw;
                    }
                }
            }
        }
        loss = wave_sum(loss); gx = wave_sum(gx); gy = wave_sum(gy); gyaw = wave_sum(gyaw);
        __builtin_amdgcn_wave_barrier();                // nobody still reads the lists the next step overwrites
        const float wt = powf(p.decay_rate, (float)t) / wsum;
        if (lane == 0) {
            part[t] = loss * wt;
            if (p.grad) {
                float* g = p.grad + ((size_t)row * TT + t) * 6;
                const float* gi = p.grad_in ? p.grad_in + ((size_t)row * TT + t) * 6 : nullptr;
                const float sc = s_coef * wt;
                const float o[6] = {sc * gx, sc * gy, 0.f, sc * gyaw, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 6; ++k) g[k] = o[k] + (gi ? gi[k] : 0.f);
            }
        }
    }
    __syncthreads();
    if (tid == 0 && p.loss) {
        float v = 0.f;
        for (int t = 0; t < TT; ++t) v += part[t];      // fixed order: deterministic
        p.loss[row] = v;
    }
}
}  // namespace

hipError_t launch_map_collision(const MapCollisionArgs& a, int rows, hipStream_t s) {
    if (a.num_points_l < 1 || a.num_points_w < 1 || a.num_points_l * a.num_points_w > kMaxPts || a.num_samp < 1 || rows < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(map_collision_kernel, dim3(rows), dim3(64 * kMapWaves), 0, s, a);
    return hipGetLastError();
}

}  // namespace cld
