// Microbenchmark: operand / result lane layout and issue rate of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 blocks, K = 1)
// on gfx950 -- the candidate for an 8-agents-per-workgroup guidance kernel (M granularity 4 instead of 16).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma4x4 mfma4x4.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {     // out[3][64][4]
    const int l = threadIdx.x;
    v4f z = {0, 0, 0, 0};
    // (1) a = lane id, b = 1: D[l][r] shows which A lane feeds result register r of lane l
    v4f d1 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 1.0f, z, 0, 0, 0);
    // (2) a = 1, b = lane id: which B lane feeds lane l
    v4f d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)l, z, 0, 0, 0);
    // (3) cbsz = 3, abid = 0: A of block 0 broadcast to 8 blocks?
    v4f d3 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 1.0f, z, 3, 0, 0);
    for (int r = 0; r < 4; ++r) { out[(0 * 64 + l) * 4 + r] = d1[r]; out[(1 * 64 + l) * 4 + r] = d2[r]; out[(2 * 64 + l) * 4 + r] = d3[r]; }
}

template <int NACC, int CB = 0>
__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* cyc, int iters) {
    const int l = threadIdx.x & 63;
    v4f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4f{0, 0, 0, 0};
    float a = 1.0f + l * 1e-3f, b = 0.5f + l * 1e-3f;
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], CB, CB ? 1 : 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void rate16(float* out, unsigned long long* cyc, int iters) {     // the 16x16x4 reference
    const int l = threadIdx.x & 63;
    v4f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4f{0, 0, 0, 0};
    float a = 1.0f + l * 1e-3f, b = 0.5f + l * 1e-3f;
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 1 << 22); hipMalloc(&c, 4096 * 8);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(3 * 64 * 4);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    const char* names[3] = {"A lane -> D (a = lane, b = 1)", "B lane -> D (a = 1, b = lane)", "cbsz = 3 (a = lane, b = 1)"};
    for (int t = 0; t < 3; ++t) {
        printf("%s\n", names[t]);
        for (int l : {0, 1, 2, 3, 4, 5, 8, 17, 33, 63}) printf("  lane %2d: D = %5.0f %5.0f %5.0f %5.0f\n", l, h[(t * 64 + l) * 4], h[(t * 64 + l) * 4 + 1], h[(t * 64 + l) * 4 + 2], h[(t * 64 + l) * 4 + 3]);
    }
    const int iters = 2000;
    auto run = [&](auto kern, const char* nm, int nacc, double flop_per) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, c, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> cy(256);
        hipMemcpy(cy.data(), c, 256 * 8, hipMemcpyDeviceToHost);
        double per = (double)cy[0] / (iters * 8.0 * nacc);
        printf("%-22s %d accumulators: %6.2f cycles per MFMA per wave (one wave per SIMD) -> %5.1f FLOP/cycle/SIMD\n", nm, nacc, per, flop_per / per);
    };
    run(rate<1>, "4x4x1_16b", 1, 512.0); run(rate<2>, "4x4x1_16b", 2, 512.0); run(rate<4>, "4x4x1_16b", 4, 512.0); run(rate<8>, "4x4x1_16b", 8, 512.0);
    run(rate<4, 4>, "4x4x1_16b cbsz=4", 4, 512.0); run(rate<4, 3>, "4x4x1_16b cbsz=3", 4, 512.0); run(rate<8, 4>, "4x4x1_16b cbsz=4", 8, 512.0);
    run(rate16<1>, "16x16x4", 1, 2048.0); run(rate16<4>, "16x16x4", 4, 2048.0);
    return 0;
}
