// Microbenchmark: does VALU work of ANOTHER wave of the SIMD run in the shadow of v_mfma_f32_16x16x4_f32, and does VALU work of the
// SAME wave?  (a) workgroups of 8 waves, two per SIMD: waves 0..3 issue MFMAs the way the Winograd K loops do (four dependent MFMAs
// per accumulator, 16 accumulators in turn), waves 4..7 issue a stream of independent VALU instructions (NONE / v_pk_fma_f32 / v_fma_f32 /
// v_exp_f32) until the MFMA waves are done; (b) one wave per SIMD issuing NV VALU instructions after every MFMA.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_covalu mfma_covalu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND, int SWAP = 0>      // what the second wave of every SIMD does: 0 nothing, 1 v_pk_fma_f32, 2 v_fma_f32, 3 v_exp_f32, 4 MFMAs as well
__global__ __launch_bounds__(512) void co(float* out, unsigned long long* cyc, int iters) {      // SWAP: the VALU stream is the OLDER wave of the SIMD (waves 0..3)
    const int lane = threadIdx.x & 63, wave = SWAP ? ((threadIdx.x >> 6) ^ 4) : (threadIdx.x >> 6);
    float s = 0.f;
    unsigned long long t0 = 0, t1 = 0;
    unsigned long long nval = 0;
    if (wave < 4 || KIND == 4) {
        v4f acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = v4f{0, 0, 0, 0};
        const float a = 1.f + lane * 1e-3f, b = 0.5f + lane * 1e-3f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (KIND != 0) {
        v2f x[8];
        for (int i = 0; i < 8; ++i) x[i] = v2f{0.1f * i + lane, 0.2f * i + lane};
        const v2f m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
        t0 = __builtin_amdgcn_s_memtime();
        // a fixed amount of VALU work sized to outlast the MFMA waves when they run alone (iters * 64 MFMAs * 32 cycles)
        const int n = iters * 64 * 32 / (KIND == 3 ? 16 * 16 : 16 * 4) * 2;
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                if (KIND == 1) x[v & 7] = __builtin_elementwise_fma(x[v & 7], m, c);
                else if (KIND == 2) x[v & 7][0] = __builtin_fmaf(x[v & 7][0], 1.0001f, 0.5f);
                else x[v & 7][0] = __builtin_amdgcn_exp2f(x[v & 7][0]);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        nval = (unsigned long long)n * 16;
        for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) { cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + wave) * 2 + 1] = nval; }
}

template <int NV, int PK>
__global__ __launch_bounds__(256) void same(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    v4f acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4f{0, 0, 0, 0};
    const float a = 1.f + lane * 1e-3f, b = 0.5f + lane * 1e-3f;
    v2f x[8];
    for (int i = 0; i < 8; ++i) x[i] = v2f{0.1f * i + lane, 0.2f * i + lane};
    const v2f m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int j = ((u * 4 + e) * NV + v) & 7;
                    if (PK) x[j] = __builtin_elementwise_fma(x[j], m, c);
                    else x[j][0] = __builtin_fmaf(x[j][0], 1.0001f, 0.5f);
                }
                if (NV) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NV, 0); }
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// (c) as (b), but consecutive MFMAs go to DIFFERENT accumulators (four in turn): a dependent MFMA is four issues away
template <int NV>
__global__ __launch_bounds__(256) void indep(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    v4f acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4f{0, 0, 0, 0};
    const float a = 1.f + lane * 1e-3f, b = 0.5f + lane * 1e-3f;
    v2f x[8];
    for (int i = 0; i < 8; ++i) x[i] = v2f{0.1f * i + lane, 0.2f * i + lane};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[4 * u + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[4 * u + m], 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const int j = (((u * 4 + e) * 4 + m) * NV + v) & 7;
                        x[j][0] = __builtin_fmaf(x[j][0], 1.0001f, 0.5f);
                    }
                    if (NV) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, NV, 0); }
                }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 1 << 22); (void)hipMalloc(&c, 256 * 16 * 8);
    const int iters = 2000;
    auto run_co = [&](auto kern, const char* nm) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, c, iters);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, d, c, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> cy(256 * 16);
        (void)hipMemcpy(cy.data(), c, 256 * 16 * 8, hipMemcpyDeviceToHost);
        const double mf = (double)cy[0] / (iters * 64.0);
        printf("%-44s MFMA wave: %6.2f cycles per MFMA", nm, mf);
        if (cy[4 * 2 + 1]) printf("; partner wave: %6.2f cycles per VALU instruction (%llu issued)", (double)cy[4 * 2] / (double)cy[4 * 2 + 1], cy[4 * 2 + 1]);
        else if (cy[4 * 2]) printf("; partner wave (MFMAs too): %6.2f cycles per MFMA", (double)cy[4 * 2] / (iters * 64.0));
        printf("\n");
    };
    printf("(a) two waves per SIMD, the second one's work:\n");
    run_co(co<0>, "  nothing");
    run_co(co<1>, "  v_pk_fma_f32 stream");
    run_co(co<2>, "  v_fma_f32 stream");
    run_co(co<3>, "  v_exp_f32 stream");
    run_co(co<4>, "  the same MFMA loop");
    printf("(a') the same, the VALU stream being the OLDER wave of the SIMD:\n");
    run_co(co<1, 1>, "  v_pk_fma_f32 stream");
    run_co(co<2, 1>, "  v_fma_f32 stream");
    run_co(co<3, 1>, "  v_exp_f32 stream");
    auto run_same = [&](auto kern, const char* nm) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, c, iters);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, c, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> cy(256);
        (void)hipMemcpy(cy.data(), c, 256 * 8, hipMemcpyDeviceToHost);
        printf("%-44s %6.2f cycles per MFMA\n", nm, (double)cy[0] / (iters * 64.0));
    };
    printf("(b) one wave per SIMD, VALU instructions of the same wave behind every MFMA:\n");
    run_same(same<0, 0>, "  none");
    run_same(same<2, 0>, "  2 v_fma_f32");
    run_same(same<4, 0>, "  4 v_fma_f32");
    run_same(same<6, 0>, "  6 v_fma_f32");
    run_same(same<8, 0>, "  8 v_fma_f32");
    run_same(same<4, 1>, "  4 v_pk_fma_f32");
    run_same(same<6, 1>, "  6 v_pk_fma_f32");
    run_same(same<8, 1>, "  8 v_pk_fma_f32");
    printf("(c) one wave per SIMD, consecutive MFMAs on different accumulators (a dependent one four issues away), VALU behind every MFMA:\n");
    run_same(indep<0>, "  none");
    run_same(indep<2>, "  2 v_fma_f32");
    run_same(indep<4>, "  4 v_fma_f32");
    run_same(indep<6>, "  6 v_fma_f32");
    run_same(indep<8>, "  8 v_fma_f32");
    return 0;
}
