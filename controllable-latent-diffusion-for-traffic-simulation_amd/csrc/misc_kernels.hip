// misc_kernels.hip -- the small kernels around the conv blocks:
//   latent pack/unpack, per-sample cond-bias GEMV, the U-Net head (final 1x1 conv fused
//   with the DDPM update, reference models/dm/dm_model.py:144-163), log-prob
//   (dm_model.py:130-132,165-174), the LSTM-VAE decoder (models/vae/lstm_vae.py:28-52)
//   and the unicycle roll-out (src/tbsim/models/diffuser_helpers.py:541-639, 'parallel').
#include <cstdlib>

#include "cld_kernels.h"

namespace cld {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float mish_g(float x) {
    const float e = expf(fminf(x, 30.0f));
    const float n = e * (e + 2.0f);
    return x * (n / (n + 2.0f));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------
__global__ void pack_latent_kernel(const float* __restrict__ x, float* __restrict__ xw, int nreal, int ntot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // one float4 = one (b, l) row
    if (i >= ntot) return;
    v4f v = {0.f, 0.f, 0.f, 0.f};
    if (i < nreal) v = reinterpret_cast<const v4f*>(x)[i];
    reinterpret_cast<v4f*>(xw)[i] = v;
}
hipError_t launch_pack_latent(const float* x, float* xw, int B, int b_pad, hipStream_t s) {
    const int ntot = b_pad * 52, nreal = B * 52;
    hipLaunchKernelGGL(pack_latent_kernel, dim3((ntot + 255) / 256), dim3(256), 0, s, x, xw, nreal, ntot);
    return hipGetLastError();
}
hipError_t launch_unpack(const float* xw, float* x, int B, hipStream_t s) {
    return hipMemcpyAsync(x, xw, sizeof(float) * (size_t)B * 52 * 4, hipMemcpyDeviceToDevice, s);
}

// ------------------------------------------------------------------------------------------
// cb[b, n] = bias[n] + sum_k mish(cond[b, k]) * wc[n, k]       (temporal.py:21-25,146: the cond half of
// every block's Linear(Mish([t_emb | cond])) is constant over the denoising loop, so it is computed once
// per sample).  One workgroup = 8 agents x 256 outputs; mish(cond) rows live in LDS.
__global__ __launch_bounds__(256) void cond_bias_kernel(const float* __restrict__ cond, const float* __restrict__ wc,
                                                       const float* __restrict__ bias, float* __restrict__ cb,
                                                       int B, int ncb) {
    __shared__ float mc[8][256];
    const int b0 = blockIdx.x * 8;
    const int tid = threadIdx.x;
    for (int a = 0; a < 8; ++a) {
        const int b = b0 + a;
        mc[a][tid] = (b < B) ? mish_g(cond[(size_t)b * 256 + tid]) : 0.f;
    }
    __syncthreads();
    const int n = blockIdx.y * 256 + tid;
    if (n >= ncb) return;
    float acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) acc[a] = 0.f;
    const v4f* w4 = reinterpret_cast<const v4f*>(wc + (size_t)n * 256);
    for (int k4 = 0; k4 < 64; ++k4) {
        const v4f w = w4[k4];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const v4f m = *reinterpret_cast<const v4f*>(&mc[a][k4 * 4]);
            acc[a] = fmaf(m[0], w[0], acc[a]);
            acc[a] = fmaf(m[1], w[1], acc[a]);
            acc[a] = fmaf(m[2], w[2], acc[a]);
            acc[a] = fmaf(m[3], w[3], acc[a]);
        }
    }
    const float bn = bias[n];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int b = b0 + a;
        cb[(size_t)b * ncb + n] = (b < B) ? acc[a] + bn : 0.f;
    }
}
hipError_t launch_cond_bias(const float* cond, const float* wc, const float* bias, float* cb, int B, int b_pad,
                            int ncb, hipStream_t s) {
    hipLaunchKernelGGL(cond_bias_kernel, dim3(b_pad / 8, (ncb + 255) / 256), dim3(256), 0, s, cond, wc, bias, cb, B, ncb);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// per-agent timesteps (training-style calls: DmModel.compute_losses, dm_model.py:82-96, draws one t per sample)
__global__ __launch_bounds__(256) void add_time_bias_kernel(float* __restrict__ cb, const float* __restrict__ tb,
                                                           const int* __restrict__ t_idx, int n_timesteps, int ncb) {
    const int b = blockIdx.x;
    int t = t_idx[b];
    t = t < 0 ? 0 : (t >= n_timesteps ? n_timesteps - 1 : t);
    for (int i = threadIdx.x; i < ncb; i += 256) cb[(size_t)b * ncb + i] += tb[(size_t)t * ncb + i];
}
hipError_t launch_add_time_bias(float* cb, const float* tb, const int* t_idx, int n_timesteps, int B, int ncb, hipStream_t s) {
    hipLaunchKernelGGL(add_time_bias_kernel, dim3(B), dim3(256), 0, s, cb, tb, t_idx, n_timesteps, ncb);
    return hipGetLastError();
}

__global__ void q_sample_kernel(const float* __restrict__ z0, const float* __restrict__ noise, const int* __restrict__ t_idx,
                                const float* __restrict__ qs, int n_timesteps, float* __restrict__ xw, float* __restrict__ zn,
                                int nreal, int ntot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // one float4 = one (b, l) row
    if (i >= ntot) return;
    v4f v = {0.f, 0.f, 0.f, 0.f};
    if (i < nreal) {
        int t = t_idx[i / 52];
        t = t < 0 ? 0 : (t >= n_timesteps ? n_timesteps - 1 : t);
        const float ca = qs[t], cn = qs[n_timesteps + t];
        const v4f a = reinterpret_cast<const v4f*>(z0)[i], e = reinterpret_cast<const v4f*>(noise)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ca * a[k] + cn * e[k];
        if (zn) reinterpret_cast<v4f*>(zn)[i] = v;
    }
    reinterpret_cast<v4f*>(xw)[i] = v;
}
hipError_t launch_q_sample(const float* z0, const float* noise, const int* t_idx, const float* qs, int n_timesteps, float* xw,
                           float* z_noisy, int B, int b_pad, hipStream_t s) {
    const int ntot = b_pad * 52, nreal = B * 52;
    hipLaunchKernelGGL(q_sample_kernel, dim3((ntot + 255) / 256), dim3(256), 0, s, z0, noise, t_idx, qs, n_timesteps, xw, z_noisy,
                       nreal, ntot);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void mse_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int i = lane; i < 208; i += 64) {
        const float d = a[(size_t)r * 208 + i] - b[(size_t)r * 208 + i];
        s += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[r] = s * (1.0f / 208.0f);
}
hipError_t launch_mse_rows(const float* a, const float* b, float* out, int B, hipStream_t s) {
    hipLaunchKernelGGL(mse_rows_kernel, dim3(B), dim3(64), 0, s, a, b, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// VaeModel.compute_vae_loss (models/vae/vae_model.py:89-99), forward only:
//   recon = mse(input[..., 4:6], output) ; kld = -0.5 * sum(1 + logvar - mu^2 - exp(logvar)) / (B * T) ; loss = recon + beta * kld
// stage 1: one wave per agent -> part[b] = (sum of squared action errors, sum of the KLD integrand)
__global__ __launch_bounds__(64) void vae_loss_part_kernel(const float* __restrict__ x6, const float* __restrict__ act,
                                                          const float* __restrict__ mu, const float* __restrict__ lv,
                                                          float* __restrict__ part) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float se = 0.f, kl = 0.f;
    for (int i = lane; i < 104; i += 64) {
        const float d = x6[(size_t)b * 312 + (i >> 1) * 6 + 4 + (i & 1)] - act[(size_t)b * 104 + i];
        se += d * d;
    }
    for (int i = lane; i < 208; i += 64) {
        const float m = mu[(size_t)b * 208 + i], l = lv[(size_t)b * 208 + i];
        kl += 1.0f + l - m * m - expf(l);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { se += __shfl_xor(se, o); kl += __shfl_xor(kl, o); }
    if (lane == 0) { part[2 * b] = se; part[2 * b + 1] = kl; }
}
// stage 2: one workgroup sums the agents in a fixed order (deterministic) -> out = (loss, recon, kld)
__global__ __launch_bounds__(256) void vae_loss_final_kernel(const float* __restrict__ part, int B, float beta, float* __restrict__ out) {
    __shared__ float s0[256], s1[256];
    float a = 0.f, c = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) { a += part[2 * b]; c += part[2 * b + 1]; }
    s0[threadIdx.x] = a; s1[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s0[threadIdx.x] += s0[threadIdx.x + o]; s1[threadIdx.x] += s1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float recon = s0[0] / ((float)B * 104.0f), kld = -0.5f * s1[0] / ((float)B * 52.0f);
        out[0] = recon + beta * kld; out[1] = recon; out[2] = kld;
    }
}
hipError_t launch_vae_loss(const float* x6, const float* act, const float* mu, const float* lv, float beta, float* part, float* out,
                           int B, hipStream_t s) {
    hipLaunchKernelGGL(vae_loss_part_kernel, dim3(B), dim3(64), 0, s, x6, act, mu, lv, part);
    hipLaunchKernelGGL(vae_loss_final_kernel, dim3(1), dim3(256), 0, s, part, B, beta, out);
    return hipGetLastError();
}

// head: eps = W f + b (final_conv.1, temporal.py:119) ; mean = xc*x - nc*eps ; x' = mean + sg*z
// one thread per (b, l) row: reads 64 channels (256 B), writes 4 values.
// With eps_in the projection has been applied already (conv_chain.hip's tail chain): elementwise only.
__global__ __launch_bounds__(256) void head_kernel(const HeadArgs a) {
    __shared__ float w[4][64];
    __shared__ float bb[4];
    const int tid = threadIdx.x;
    if (!a.eps_in) {
        w[tid >> 6][tid & 63] = a.w[tid];
        if (tid < 4) bb[tid] = a.b[tid];
        __syncthreads();
    }
    const int row = blockIdx.x * 256 + tid;
    if (row >= a.b_pad * 52) return;
    const bool real = row < a.B * 52;
    v4f eps;
    if (a.eps_in) {
        eps = reinterpret_cast<const v4f*>(a.eps_in)[row];
        if (a.eps_in_uncond) {
            const v4f u = reinterpret_cast<const v4f*>(a.eps_in_uncond)[row];
#pragma unroll
            for (int d = 0; d < 4; ++d) eps[d] = (1.0f + a.cfg_w) * eps[d] - a.cfg_w * u[d];
        }
    } else {
    const v4f* f4 = reinterpret_cast<const v4f*>(a.f + (size_t)row * 64);
    float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k4 = 0; k4 < 16; ++k4) {
        const v4f f = f4[k4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            e[d] = fmaf(f[0], w[d][4 * k4 + 0], e[d]);
            e[d] = fmaf(f[1], w[d][4 * k4 + 1], e[d]);
            e[d] = fmaf(f[2], w[d][4 * k4 + 2], e[d]);
            e[d] = fmaf(f[3], w[d][4 * k4 + 3], e[d]);
        }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) eps[d] = e[d] + bb[d];
    if (a.f_uncond) {   // classifier-free guidance in noise space (upstream diffuser.py:787)
        const v4f* g4 = reinterpret_cast<const v4f*>(a.f_uncond + (size_t)row * 64);
        float u[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) {
            const v4f f = g4[k4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                u[d] = fmaf(f[0], w[d][4 * k4 + 0], u[d]);
                u[d] = fmaf(f[1], w[d][4 * k4 + 1], u[d]);
                u[d] = fmaf(f[2], w[d][4 * k4 + 2], u[d]);
                u[d] = fmaf(f[3], w[d][4 * k4 + 3], u[d]);
            }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) eps[d] = (1.0f + a.cfg_w) * eps[d] - a.cfg_w * (u[d] + bb[d]);
    }
    }
    if (a.eps_out && real) reinterpret_cast<v4f*>(a.eps_out)[row] = eps;
    if (!a.mean_out && !a.x_out) return;
    const v4f x = reinterpret_cast<const v4f*>(a.x)[row];
    v4f mean;
#pragma unroll
    for (int d = 0; d < 4; ++d) mean[d] = a.xc * x[d] - a.nc * eps[d];
    if (a.mean_out) reinterpret_cast<v4f*>(a.mean_out)[row] = mean;
    if (a.x_out) {
        v4f z = {0.f, 0.f, 0.f, 0.f};
        if (a.sg != 0.f && real) {
            if (a.z) {
                z = reinterpret_cast<const v4f*>(a.z)[row];
            } else {
                z = normal4(a.seed, a.step_salt, (unsigned)row);
            }
        }
        v4f xn;
#pragma unroll
        for (int d = 0; d < 4; ++d) xn[d] = mean[d] + a.sg * z[d];
        reinterpret_cast<v4f*>(a.x_out)[row] = xn;
        if (a.x_out2) reinterpret_cast<v4f*>(a.x_out2)[row] = xn;
    }
}
hipError_t launch_head(const HeadArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(head_kernel, dim3((a.b_pad * 52 + 255) / 256), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// out[b] = mean over (52 x 4) of Normal(mean, sigma).log_prob(xq)
//        = -(xq-mean)^2 / (2 sigma^2) - log(sigma) - log(sqrt(2 pi))        (torch.distributions.Normal)
__global__ __launch_bounds__(64) void logprob_kernel(const float* __restrict__ xq, const float* __restrict__ mean,
                                                     float sigma, float* __restrict__ out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float var2 = 2.0f * sigma * sigma;
    const float lc = logf(sigma) + 0.9189385332046727f;
    float s = 0.f;
    for (int i = lane; i < 208; i += 64) {
        const float d = xq[(size_t)b * 208 + i] - mean[(size_t)b * 208 + i];
        s += -(d * d) / var2 - lc;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[b] = s * (1.0f / 208.0f);
}
hipError_t launch_logprob(const float* xq, const float* mean, float sigma, float* out, int B, hipStream_t s) {
    hipLaunchKernelGGL(logprob_kernel, dim3(B), dim3(64), 0, s, xq, mean, sigma, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// unicycle roll-out of one agent, sequential over 52 steps (the reference's tril/bmm form is an
// O(T^2) way of writing these inclusive prefix sums; v is clipped AFTER the sum, 'parallel' mode).
__device__ void rollout_agent(const DynParams& d, const float* act /*[52][2] scaled or raw*/, const float* cs,
                              float* traj /*[52][6]*/, bool scaled_input, bool descaled_output) {
    float v_raw = cs[2], v_prev = fminf(fmaxf(cs[2], d.v_lo), d.v_hi);
    float yaw = cs[3], x = cs[0], y = cs[1];
    for (int t = 0; t < 52; ++t) {
        float acc = act[2 * t], yr = act[2 * t + 1];
        if (scaled_input) { acc = acc * d.std[4] + d.mean[4]; yr = yr * d.std[5] + d.mean[5]; }
        const float accc = fminf(fmaxf(acc, d.acc_lo), d.acc_hi);
        v_raw += accc * d.dt;
        const float v = fminf(fmaxf(v_raw, d.v_lo), d.v_hi);
        const float av = fabsf(v_prev);
        const float yb = fmaxf(fminf(d.max_steer * av, d.max_yawvel / fmaxf(av, 0.1f)), 0.1f);
        const float yrc = fmaxf(fminf(yr, yb), -yb);
        const float vavg = 0.5f * (v_prev + v);       // mat2 rows: 0.5*(v_{k-1} + v_k)
        x += vavg * cosf(yaw) * d.dt;
        y += vavg * sinf(yaw) * d.dt;
        yaw += yrc * d.dt;
        float o[6] = {x, y, v, yaw, acc, yr};
        if (scaled_input && !descaled_output) {
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = (o[k] - d.mean[k]) / d.std[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) traj[6 * t + k] = o[k];
        v_prev = v;
    }
}

__global__ void action_to_state_kernel(const DynParams d, const float* __restrict__ act, const float* __restrict__ cs,
                                       float* __restrict__ traj, int B, int scaled_input, int descaled_output) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    rollout_agent(d, act + (size_t)b * 104, cs + (size_t)b * 4, traj + (size_t)b * 312, scaled_input != 0,
                  descaled_output != 0);
}
hipError_t launch_action_to_state(const DynParams& d, const float* act, const float* cs, float* traj, int B,
                                  int scaled_input, int descaled_output, hipStream_t s) {
    hipLaunchKernelGGL(action_to_state_kernel, dim3((B + 63) / 64), dim3(64), 0, s, d, act, cs, traj, B, scaled_input,
                       descaled_output);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// LSTM decoder: one 256-thread workgroup per agent slot, thread r owns gate row r (PyTorch row order
// i, f, g, o) of both layers with its weights resident in registers (4 + 64 + 64 + 64 floats);
// h / c state lives in LDS; 52 sequential steps (lstm_vae.py:44-52: h0 = cond2hidden(cond) for both
// layers, c0 = 0).  Workgroups stride over agents.
__global__ __launch_bounds__(256) void decode_kernel(const DecoderWeights w, const DynParams d,
                                                     const float* __restrict__ z, const float* __restrict__ cond,
                                                     const float* __restrict__ cs, float* __restrict__ act_out,
                                                     float* __restrict__ traj, int B, int descaled_output) {
    __shared__ __attribute__((aligned(16))) float h0[64], h1[64], c0[64], c1[64], gates[256], zin[208], act[104];
    __shared__ __attribute__((aligned(16))) float condm[256];
    const int r = threadIdx.x;
    float wi0[4], wh0[64], wi1[64], wh1[64];
#pragma unroll
    for (int k = 0; k < 4; ++k) wi0[k] = w.w_ih0[r * 4 + k];
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        wh0[k] = w.w_hh0[r * 64 + k];
        wi1[k] = w.w_ih1[r * 64 + k];
        wh1[k] = w.w_hh1[r * 64 + k];
    }
    const float bias0 = w.b0[r], bias1 = w.b1[r];
    const int gate = r >> 6, u = r & 63;

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        condm[r] = cond[(size_t)b * 256 + r];
        if (r < 208) zin[r] = z[(size_t)b * 208 + r];
        __syncthreads();
        if (r < 64) {   // h0 = cond2hidden(cond)
            float s = w.b_c2h[r];
            const float* wr = w.w_c2h + r * 256;
            for (int k = 0; k < 256; ++k) s = fmaf(condm[k], wr[k], s);
            h0[r] = s; h1[r] = s; c0[r] = 0.f; c1[r] = 0.f;
        }
        __syncthreads();
        for (int t = 0; t < 52; ++t) {
            // ---- layer 0 ----
            float g = bias0;
#pragma unroll
            for (int k = 0; k < 4; ++k) g = fmaf(zin[4 * t + k], wi0[k], g);
#pragma unroll
            for (int k = 0; k < 64; ++k) g = fmaf(h0[k], wh0[k], g);
            gates[r] = (gate == 2) ? tanhf(g) : sigmoid_f(g);
            __syncthreads();
            if (r < 64) {
                const float c = gates[64 + r] * c0[r] + gates[r] * gates[128 + r];
                c0[r] = c;
                h0[r] = gates[192 + r] * tanhf(c);
            }
            __syncthreads();
            // ---- layer 1 ----
            g = bias1;
#pragma unroll
            for (int k = 0; k < 64; ++k) g = fmaf(h0[k], wi1[k], g);
#pragma unroll
            for (int k = 0; k < 64; ++k) g = fmaf(h1[k], wh1[k], g);
            gates[r] = (gate == 2) ? tanhf(g) : sigmoid_f(g);
            __syncthreads();
            if (r < 64) {
                const float c = gates[64 + r] * c1[r] + gates[r] * gates[128 + r];
                c1[r] = c;
                h1[r] = gates[192 + r] * tanhf(c);
            }
            __syncthreads();
            if (r < 2) {   // hid2act
                float s = w.b_h2a[r];
                const float* wr = w.w_h2a + r * 64;
                for (int k = 0; k < 64; ++k) s = fmaf(h1[k], wr[k], s);
                act[2 * t + r] = s;
            }
        }
        __syncthreads();
        if (act_out && r < 104) act_out[(size_t)b * 104 + r] = act[r];
        if (traj && r == 0) rollout_agent(d, act, cs + (size_t)b * 4, traj + (size_t)b * 312, true, descaled_output != 0);
        __syncthreads();
        (void)u;
    }
}
// MFMA formulation of the decoder for batches that fill it: 16 agents per workgroup, the gate pre-activations of a layer
// step as the GEMM [16 agents x K] x [K x 256] on v_mfma_f32_16x16x4_f32.  Wave w owns hidden units 16w..16w+15 and the
// four N-tiles {i, f, g, o} of those units, so the MFMA result layout (lane = (unit n, agent block rb), 4 registers = agents
// 4rb..4rb+3) holds all four gates of one (agent, unit) cell in one lane: the cell update runs in registers, only h passes
// through LDS as the next product's A operand (float4 reads: 4 consecutive k per lane, element e feeds MFMA e; the register-
// resident B fragments use the same k permutation) -- two barriers per time step, 196 MFMAs per wave per step.
// v_exp_f32 / v_rcp_f32 forms (1 ulp each): ~1e-7 absolute on the gates, far inside the decode bars; the libm forms cost as
// many cycles per step here as the 196 MFMAs do
__device__ __forceinline__ float fsig_m(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh_m(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

__global__ __launch_bounds__(256) void decode_mfma_kernel(const DecoderWeights w, const DynParams d,
                                                          const float* __restrict__ z, const float* __restrict__ cond,
                                                          const float* __restrict__ cs, float* __restrict__ act_out,
                                                          float* __restrict__ traj, int B, int descaled_output) {
    constexpr int AG = 16, HS = 68;
    __shared__ __attribute__((aligned(16))) float hs[2][2][AG][HS];     // [layer][parity][agent][unit]
    __shared__ __attribute__((aligned(16))) float zin[AG][208];
    __shared__ __attribute__((aligned(16))) float condm[AG][256];
    __shared__ float actp[2][52][4][AG];     // per-wave partials of hid2act
    __shared__ float act[AG][104];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, rb = lane >> 4;
    const int u = 16 * wv + n;
    const float wa0 = w.w_h2a[u], wa1 = w.w_h2a[64 + u];
    // B fragments: gate g, k-step (jj, e) -> W[col = 64 g + 16 wv + n][k = 16 jj + 4 rb + e]
    float f_hh0[4][4][4], f_ih1[4][4][4], f_hh1[4][4][4], f_ih0[4], fb0[4], fb1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = 64 * g + 16 * wv + n;
        f_ih0[g] = w.w_ih0[col * 4 + rb];
        fb0[g] = w.b0[col];
        fb1[g] = w.b1[col];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const v4f x0 = *reinterpret_cast<const v4f*>(w.w_hh0 + col * 64 + 16 * jj + 4 * rb);
            const v4f x1 = *reinterpret_cast<const v4f*>(w.w_ih1 + col * 64 + 16 * jj + 4 * rb);
            const v4f x2 = *reinterpret_cast<const v4f*>(w.w_hh1 + col * 64 + 16 * jj + 4 * rb);
#pragma unroll
            for (int e = 0; e < 4; ++e) { f_hh0[g][jj][e] = x0[e]; f_ih1[g][jj][e] = x1[e]; f_hh1[g][jj][e] = x2[e]; }
        }
    }
    const int ngroups = (B + AG - 1) / AG;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int b0 = grp * AG;
        auto agent = [&](int ag) { return (b0 + ag < B) ? b0 + ag : B - 1; };      // tail slots replay the last agent; never stored
        for (int i = tid; i < AG * 256; i += 256) condm[i >> 8][i & 255] = cond[(size_t)agent(i >> 8) * 256 + (i & 255)];
        for (int i = tid; i < AG * 208; i += 256) zin[i / 208][i % 208] = z[(size_t)agent(i / 208) * 208 + i % 208];
        __syncthreads();
        for (int i = tid; i < AG * 64; i += 256) {      // h0 = cond2hidden(cond) for both layers (lstm_vae.py:46-49)
            const int ag = i >> 6, uu = i & 63;
            float s = w.b_c2h[uu];
            const float* wr = w.w_c2h + uu * 256;
            for (int k = 0; k < 256; ++k) s = fmaf(condm[ag][k], wr[k], s);
            hs[0][0][ag][uu] = s;
            hs[1][0][ag][uu] = s;
        }
        float c0[4] = {0.f, 0.f, 0.f, 0.f}, c1[4] = {0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        for (int t = 0; t < 52; ++t) {
            const int pr = t & 1;
            v4f acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = v4f{fb0[g], fb0[g], fb0[g], fb0[g]};
            {
                const float xa = zin[n][4 * t + rb];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, f_ih0[g], acc[g], 0, 0, 0);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const v4f ha = *reinterpret_cast<const v4f*>(&hs[0][pr][n][16 * jj + 4 * rb]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[e], f_hh0[g][jj][e], acc[g], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ig = fsig_m(acc[0][r]), fg = fsig_m(acc[1][r]), gg = ftanh_m(acc[2][r]), og = fsig_m(acc[3][r]);
                const float c = fg * c0[r] + ig * gg;
                c0[r] = c;
                hs[0][pr ^ 1][4 * rb + r][u] = og * ftanh_m(c);
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = v4f{fb1[g], fb1[g], fb1[g], fb1[g]};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const v4f ha = *reinterpret_cast<const v4f*>(&hs[0][pr ^ 1][n][16 * jj + 4 * rb]);
                const v4f hb = *reinterpret_cast<const v4f*>(&hs[1][pr][n][16 * jj + 4 * rb]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[e], f_ih1[g][jj][e], acc[g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(hb[e], f_hh1[g][jj][e], acc[g], 0, 0, 0);
                }
            }
            float ap[4], aq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ig = fsig_m(acc[0][r]), fg = fsig_m(acc[1][r]), gg = ftanh_m(acc[2][r]), og = fsig_m(acc[3][r]);
                const float c = fg * c1[r] + ig * gg;
                c1[r] = c;
                const float hn = og * ftanh_m(c);
                hs[1][pr ^ 1][4 * rb + r][u] = hn;
                ap[r] = hn * wa0;
                aq[r] = hn * wa1;
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) { ap[r] += __shfl_xor(ap[r], o); aq[r] += __shfl_xor(aq[r], o); }
            if (n == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { actp[0][t][wv][4 * rb + r] = ap[r]; actp[1][t][wv][4 * rb + r] = aq[r]; }
            }
            __syncthreads();
        }
        for (int i = tid; i < AG * 104; i += 256) {      // hid2act: sum the four waves' partials in wave order + bias
            const int ag = i / 104, k = i % 104, t = k >> 1, c = k & 1;
            act[ag][k] = actp[c][t][0][ag] + actp[c][t][1][ag] + actp[c][t][2][ag] + actp[c][t][3][ag] + w.b_h2a[c];
        }
        __syncthreads();
        if (act_out)
            for (int i = tid; i < AG * 104; i += 256)
                if (b0 + i / 104 < B) act_out[(size_t)(b0 + i / 104) * 104 + i % 104] = act[i / 104][i % 104];
        if (traj && tid < AG && b0 + tid < B)
            rollout_agent(d, &act[tid][0], cs + (size_t)(b0 + tid) * 4, traj + (size_t)(b0 + tid) * 312, true, descaled_output != 0);
        __syncthreads();
    }
}

hipError_t launch_decode(const DecoderWeights& w, const DynParams& d, const float* z, const float* cond, const float* cs,
                         float* act, float* traj, int B, int descaled_output, hipStream_t s, int form) {
    // from 256 agents (16 workgroups of 16) the MFMA kernel is faster than one VALU workgroup per agent
    // (tests force either form through cld_debug_force_kernel)
    const bool mfma = form != FORM_AUTO ? form == FORM_MFMA : B >= 256;
    if (mfma) {
        const int groups = (B + 15) / 16;
        hipLaunchKernelGGL(decode_mfma_kernel, dim3(groups < 1024 ? groups : 1024), dim3(256), 0, s, w, d, z, cond, cs, act,
                           cs ? traj : nullptr, B, descaled_output);
        return hipGetLastError();
    }
    const int grid = B < 2048 ? B : 2048;
    hipLaunchKernelGGL(decode_kernel, dim3(grid), dim3(256), 0, s, w, d, z, cond, cs, act, cs ? traj : nullptr, B,
                       descaled_output);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------
// convert_state_to_state_and_action (src/tbsim/models/diffuser_helpers.py:685-749): finite differences of
// (x, y, yaw) under the unicycle model, positions/yaw pre-padded with zeros, speed with curr_speed.
__global__ void state_to_state_action_kernel(const DynParams d, const float* __restrict__ pos, const float* __restrict__ yaw,
                                             const float* __restrict__ speed, float* __restrict__ out, int B, int scaled) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float px = 0.f, py = 0.f, pyaw = 0.f, pv = speed[b];
    const float two_pi = 6.283185307179586f;
    for (int t = 0; t < 52; ++t) {
        const float x = pos[((size_t)b * 52 + t) * 2], y = pos[((size_t)b * 52 + t) * 2 + 1], th = yaw[(size_t)b * 52 + t];
        const float v = (x - px) / d.dt * cosf(th) + (y - py) / d.dt * sinf(th);
        const float acc = (v - pv) / d.dt;
        float df = th - pyaw + 0.5f * two_pi;                 // angle_diff: floored modulo into [-pi, pi)
        df = df - floorf(df / two_pi) * two_pi - 0.5f * two_pi;
        if (df > 3.141592653589793f) df -= two_pi;
        const float yr = df / d.dt;
        float o[6] = {x, y, v, th, acc, yr};
        if (scaled) {
#pragma unroll
            for (int k = 0; k < 6; ++k) o[k] = (o[k] - d.mean[k]) / d.std[k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) out[((size_t)b * 52 + t) * 6 + k] = o[k];
        px = x; py = y; pyaw = th; pv = v;
    }
}
hipError_t launch_state_to_state_action(const DynParams& d, const float* pos, const float* yaw, const float* speed,
                                        float* out6, int B, int scaled_output, hipStream_t s) {
    hipLaunchKernelGGL(state_to_state_action_kernel, dim3((B + 63) / 64), dim3(64), 0, s, d, pos, yaw, speed, out6, B,
                       scaled_output);
    return hipGetLastError();
}

// LSTM encoder + (mu, logvar) heads + reparametrisation (lstm_vae.py:6-26,87-99): same structure as the decoder
// kernel -- thread r owns gate row r of both layers (6 + 64 + 64 + 64 weights in registers), state in LDS.
__global__ __launch_bounds__(256) void encode_kernel(const EncoderWeights w, const float* __restrict__ x6,
                                                     const float* __restrict__ cond, const float* __restrict__ noise,
                                                     float* __restrict__ z, float* __restrict__ mu_out,
                                                     float* __restrict__ lv_out, int B) {
    __shared__ __attribute__((aligned(16))) float h0[64], h1[64], c0[64], c1[64], gates[256], xin[312], head[8 * 52];
    __shared__ __attribute__((aligned(16))) float condm[256];
    const int r = threadIdx.x;
    float wi0[6], wh0[64], wi1[64], wh1[64];
#pragma unroll
    for (int k = 0; k < 6; ++k) wi0[k] = w.w_ih0[r * 6 + k];
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        wh0[k] = w.w_hh0[r * 64 + k];
        wi1[k] = w.w_ih1[r * 64 + k];
        wh1[k] = w.w_hh1[r * 64 + k];
    }
    const float bias0 = w.b0[r], bias1 = w.b1[r];
    const int gate = r >> 6;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        condm[r] = cond[(size_t)b * 256 + r];
        for (int i = r; i < 312; i += 256) xin[i] = x6[(size_t)b * 312 + i];
        __syncthreads();
        if (r < 64) {
            float s = w.b_c2h[r];
            const float* wr = w.w_c2h + r * 256;
            for (int k = 0; k < 256; ++k) s = fmaf(condm[k], wr[k], s);
            h0[r] = s; h1[r] = s; c0[r] = 0.f; c1[r] = 0.f;
        }
        __syncthreads();
        for (int t = 0; t < 52; ++t) {
            float g = bias0;
#pragma unroll
            for (int k = 0; k < 6; ++k) g = fmaf(xin[6 * t + k], wi0[k], g);
#pragma unroll
            for (int k = 0; k < 64; ++k) g = fmaf(h0[k], wh0[k], g);
            gates[r] = (gate == 2) ? tanhf(g) : sigmoid_f(g);
            __syncthreads();
            if (r < 64) {
                const float c = gates[64 + r] * c0[r] + gates[r] * gates[128 + r];
                c0[r] = c;
                h0[r] = gates[192 + r] * tanhf(c);
            }
            __syncthreads();
            g = bias1;
#pragma unroll
            for (int k = 0; k < 64; ++k) g = fmaf(h0[k], wi1[k], g);
#pragma unroll
            for (int k = 0; k < 64; ++k) g = fmaf(h1[k], wh1[k], g);
            gates[r] = (gate == 2) ? tanhf(g) : sigmoid_f(g);
            __syncthreads();
            if (r < 64) {
                const float c = gates[64 + r] * c1[r] + gates[r] * gates[128 + r];
                c1[r] = c;
                h1[r] = gates[192 + r] * tanhf(c);
            }
            __syncthreads();
            if (r < 8) {   // mu (rows 0-3) and logvar (rows 4-7) heads on the top layer's output
                const float* wr = (r < 4 ? w.w_mu + r * 64 : w.w_lv + (r - 4) * 64);
                float s = (r < 4 ? w.b_mu[r] : w.b_lv[r - 4]);
                for (int k = 0; k < 64; ++k) s = fmaf(h1[k], wr[k], s);
                head[8 * t + r] = s;
            }
        }
        __syncthreads();
        if (r < 208) {
            const int t = r >> 2, k = r & 3;
            const float m = head[8 * t + k], lv = head[8 * t + 4 + k];
            if (mu_out) mu_out[(size_t)b * 208 + r] = m;
            if (lv_out) lv_out[(size_t)b * 208 + r] = lv;
            if (z) z[(size_t)b * 208 + r] = m + (noise ? noise[(size_t)b * 208 + r] : 0.f) * expf(0.5f * lv);
        }
        __syncthreads();
    }
}
// MFMA formulation of the encoder (same structure as decode_mfma_kernel: 16 agents per workgroup, wave w owns units
// 16w..16w+15 and their four gate N-tiles, cell update in registers); the 6-dim input takes two k-steps (dims 0-3, then 4-5
// padded with zeros); the mu / logvar heads are eight 64-long dot products per (agent, step), reduced over the 16 lanes of a
// unit block and over the four waves through a small parity-buffered LDS array.
__global__ __launch_bounds__(256) void encode_mfma_kernel(const EncoderWeights w, const float* __restrict__ x6,
                                                          const float* __restrict__ cond, const float* __restrict__ noise,
                                                          float* __restrict__ z, float* __restrict__ mu_out,
                                                          float* __restrict__ lv_out, int B) {
    constexpr int AG = 16, HS = 68;
    __shared__ __attribute__((aligned(16))) float hs[2][2][AG][HS];
    __shared__ __attribute__((aligned(16))) float xin[AG][312];
    __shared__ __attribute__((aligned(16))) float condm[AG][256];
    __shared__ float parts[2][8][4][AG];     // [step parity][head][wave][agent]
    __shared__ float head[AG][52][8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, rb = lane >> 4;
    const int u = 16 * wv + n;
    float wh[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { wh[k] = w.w_mu[k * 64 + u]; wh[4 + k] = w.w_lv[k * 64 + u]; }
    float f_hh0[4][4][4], f_ih1[4][4][4], f_hh1[4][4][4], f_ih0a[4], f_ih0b[4], fb0[4], fb1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = 64 * g + 16 * wv + n;
        f_ih0a[g] = w.w_ih0[col * 6 + rb];
        f_ih0b[g] = rb < 2 ? w.w_ih0[col * 6 + 4 + rb] : 0.f;
        fb0[g] = w.b0[col];
        fb1[g] = w.b1[col];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const v4f x0 = *reinterpret_cast<const v4f*>(w.w_hh0 + col * 64 + 16 * jj + 4 * rb);
            const v4f x1 = *reinterpret_cast<const v4f*>(w.w_ih1 + col * 64 + 16 * jj + 4 * rb);
            const v4f x2 = *reinterpret_cast<const v4f*>(w.w_hh1 + col * 64 + 16 * jj + 4 * rb);
#pragma unroll
            for (int e = 0; e < 4; ++e) { f_hh0[g][jj][e] = x0[e]; f_ih1[g][jj][e] = x1[e]; f_hh1[g][jj][e] = x2[e]; }
        }
    }
    const int ngroups = (B + AG - 1) / AG;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int b0 = grp * AG;
        auto agent = [&](int ag) { return (b0 + ag < B) ? b0 + ag : B - 1; };
        for (int i = tid; i < AG * 256; i += 256) condm[i >> 8][i & 255] = cond[(size_t)agent(i >> 8) * 256 + (i & 255)];
        for (int i = tid; i < AG * 312; i += 256) xin[i / 312][i % 312] = x6[(size_t)agent(i / 312) * 312 + i % 312];
        __syncthreads();
        for (int i = tid; i < AG * 64; i += 256) {
            const int ag = i >> 6, uu = i & 63;
            float s = w.b_c2h[uu];
            const float* wr = w.w_c2h + uu * 256;
            for (int k = 0; k < 256; ++k) s = fmaf(condm[ag][k], wr[k], s);
            hs[0][0][ag][uu] = s;
            hs[1][0][ag][uu] = s;
        }
        float c0[4] = {0.f, 0.f, 0.f, 0.f}, c1[4] = {0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        for (int t = 0; t < 52; ++t) {
            const int pr = t & 1;
            v4f acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = v4f{fb0[g], fb0[g], fb0[g], fb0[g]};
            {
                const float xa = xin[n][6 * t + rb];
                const float xb = rb < 2 ? xin[n][6 * t + 4 + rb] : 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, f_ih0a[g], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb, f_ih0b[g], acc[g], 0, 0, 0);
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const v4f ha = *reinterpret_cast<const v4f*>(&hs[0][pr][n][16 * jj + 4 * rb]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[e], f_hh0[g][jj][e], acc[g], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ig = fsig_m(acc[0][r]), fg = fsig_m(acc[1][r]), gg = ftanh_m(acc[2][r]), og = fsig_m(acc[3][r]);
                const float c = fg * c0[r] + ig * gg;
                c0[r] = c;
                hs[0][pr ^ 1][4 * rb + r][u] = og * ftanh_m(c);
            }
            __syncthreads();
            if (t > 0 && tid < AG * 8) {      // heads of step t-1: its partials are complete (barrier above) and not yet overwritten
                const int ag = tid >> 3, k = tid & 7, q = (t - 1) & 1;
                head[ag][t - 1][k] = parts[q][k][0][ag] + parts[q][k][1][ag] + parts[q][k][2][ag] + parts[q][k][3][ag] +
                                     (k < 4 ? w.b_mu[k] : w.b_lv[k - 4]);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = v4f{fb1[g], fb1[g], fb1[g], fb1[g]};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const v4f ha = *reinterpret_cast<const v4f*>(&hs[0][pr ^ 1][n][16 * jj + 4 * rb]);
                const v4f hb = *reinterpret_cast<const v4f*>(&hs[1][pr][n][16 * jj + 4 * rb]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[e], f_ih1[g][jj][e], acc[g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(hb[e], f_hh1[g][jj][e], acc[g], 0, 0, 0);
                }
            }
            float hp[8][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ig = fsig_m(acc[0][r]), fg = fsig_m(acc[1][r]), gg = ftanh_m(acc[2][r]), og = fsig_m(acc[3][r]);
                const float c = fg * c1[r] + ig * gg;
                c1[r] = c;
                const float hn = og * ftanh_m(c);
                hs[1][pr ^ 1][4 * rb + r][u] = hn;
#pragma unroll
                for (int k = 0; k < 8; ++k) hp[k][r] = hn * wh[k];
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1)
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hp[k][r] += __shfl_xor(hp[k][r], o);
            if (n == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) parts[pr][k][wv][4 * rb + r] = hp[k][r];
            }
            __syncthreads();
        }
        if (tid < AG * 8) {                   // heads of the last step
            const int ag = tid >> 3, k = tid & 7, q = 51 & 1;
            head[ag][51][k] = parts[q][k][0][ag] + parts[q][k][1][ag] + parts[q][k][2][ag] + parts[q][k][3][ag] +
                              (k < 4 ? w.b_mu[k] : w.b_lv[k - 4]);
        }
        __syncthreads();
        for (int i = tid; i < AG * 208; i += 256) {
            const int ag = i / 208, r = i % 208, t = r >> 2, k = r & 3, b = b0 + ag;
            if (b >= B) continue;
            const float m = head[ag][t][k], lv = head[ag][t][4 + k];
            if (mu_out) mu_out[(size_t)b * 208 + r] = m;
            if (lv_out) lv_out[(size_t)b * 208 + r] = lv;
            if (z) z[(size_t)b * 208 + r] = m + (noise ? noise[(size_t)b * 208 + r] : 0.f) * expf(0.5f * lv);
        }
        __syncthreads();
    }
}

hipError_t launch_encode(const EncoderWeights& w, const float* x6, const float* cond, const float* noise, float* z,
                         float* mu, float* logvar, int B, hipStream_t s, int form) {
    const bool mfma = form != FORM_AUTO ? form == FORM_MFMA : B >= 256;
    if (mfma) {
        const int groups = (B + 15) / 16;
        hipLaunchKernelGGL(encode_mfma_kernel, dim3(groups < 1024 ? groups : 1024), dim3(256), 0, s, w, x6, cond, noise, z, mu,
                           logvar, B);
        return hipGetLastError();
    }
    const int grid = B < 2048 ? B : 2048;
    hipLaunchKernelGGL(encode_kernel, dim3(grid), dim3(256), 0, s, w, x6, cond, noise, z, mu, logvar, B);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------
__global__ void world_step_kernel(const float* __restrict__ traj, const float* __restrict__ centroid,
                                  const float* __restrict__ yaw, int k, float* __restrict__ world,
                                  float* __restrict__ next_cs, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* t = traj + ((size_t)b * 52 + k) * 6;
    const float c = cosf(yaw[b]), s = sinf(yaw[b]);
    world[b * 3 + 0] = t[0] * c - t[1] * s + centroid[b * 2 + 0];     // [px, py] @ [[c, s], [-s, c]]
    world[b * 3 + 1] = t[0] * s + t[1] * c + centroid[b * 2 + 1];
    world[b * 3 + 2] = yaw[b] + t[3];
    if (next_cs) { next_cs[b * 4 + 0] = 0.f; next_cs[b * 4 + 1] = 0.f; next_cs[b * 4 + 2] = t[2]; next_cs[b * 4 + 3] = 0.f; }
}
hipError_t launch_world_step(const float* traj, const float* centroid, const float* yaw, int k, float* world,
                             float* next_cs, int B, hipStream_t s) {
    hipLaunchKernelGGL(world_step_kernel, dim3((B + 255) / 256), dim3(256), 0, s, traj, centroid, yaw, k, world, next_cs, B);
    return hipGetLastError();
}

}  // namespace cld
