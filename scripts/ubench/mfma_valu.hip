// Microbenchmark: how many independent VALU instructions fit in the shadow of a v_mfma_f32_4x4x1_16b_f32 issued from the same
// wave (one wave per SIMD)?  NV plain FMAs (or NT transcendentals) after every MFMA, four accumulators in turn.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NV, int NT>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = 1.f + lane * 1e-3f, b = 0.5f + lane * 1e-3f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.1f * i + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) x[(u * NV + v) & 7] = __builtin_fmaf(x[(u * NV + v) & 7], 1.0001f, 0.5f);
#pragma unroll
            for (int v = 0; v < NT; ++v) x[(u * NT + v) & 7] = __builtin_amdgcn_rcpf(x[(u * NT + v) & 7]);
#pragma unroll
            for (int v = 0; v < (NV + NT ? 1 : 0); ++v) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NV + NT, 0); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 1 << 22); (void)hipMalloc(&c, 4096 * 8);
    const int iters = 4000;
    auto run = [&](auto kern, const char* nm) {
        printf("%-30s", nm);
        for (int nw = 1; nw <= 4; nw *= 2) {       // waves per SIMD
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(kern, dim3(256), dim3(256 * nw), 0, 0, d, c, iters);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(256 * nw), 0, 0, d, c, iters);
            (void)hipEventRecord(e1);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> cy(256);
            (void)hipMemcpy(cy.data(), c, 256 * 8, hipMemcpyDeviceToHost);
            const double per = (double)cy[0] / (iters * 32.0);
            const double tflops = 256.0 * 4 * nw * iters * 32.0 * 512.0 / (ms * 1e-3) * 1e-12;
            printf(" | %d/SIMD: %6.2f cyc per MFMA of a wave, %5.2f of the SIMD, %6.1f TFLOP/s by the event clock", nw, per, per / nw, tflops);
        }
        printf("\n");
    };
    run(k<0, 0>, "MFMA only");
    run(k<1, 0>, "+ 1 v_fma per MFMA");
    run(k<2, 0>, "+ 2 v_fma per MFMA");
    run(k<3, 0>, "+ 3 v_fma per MFMA");
    run(k<4, 0>, "+ 4 v_fma per MFMA");
    run(k<0, 1>, "+ 1 v_rcp per MFMA");
    run(k<0, 2>, "+ 2 v_rcp per MFMA");
    run(k<1, 1>, "+ 1 v_fma + 1 v_rcp per MFMA");
    return 0;
}
