// Microbenchmark: does the LDS operand stream of the 8-agent guidance kernel limit its 4x4x1 MFMA sweeps?
// Four waves per workgroup (one per SIMD), one workgroup per CU; every wave runs K-sweeps of 32 groups: one (or two) ds_read_b128
// per four k-steps, two groups ahead of its use, 8 MFMAs per group.  Variants: no LDS reads; all 64 lanes read (4 distinct rows,
// two reads per group = the "every lane fetches its B operand" form); 8 lanes read under an exec mask (one read per group = the
// cbsz-broadcast form).  build: hipcc --offload-arch=gfx950 -O3 -o lds_mfma lds_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void sweep(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float hs[8][132];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8 * 132; i += 256) (&hs[0][0])[i] = 1e-3f * i;
    __syncthreads();
    v4f w[32];
    for (int j = 0; j < 32; ++j) w[j] = v4f{1.f + j, 2.f + lane, 3.f, 4.f};
    v4f a00 = {0, 0, 0, 0}, a01 = a00, a10 = a00, a11 = a00;
    const float* p0 = &hs[lane & 3][0];
    const float* p1 = &hs[4 + (lane & 3)][0];
    const unsigned p8 = (unsigned)(size_t)&hs[lane & 7][0];      // LDS byte address = the low 32 bits of the flat address
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        v4f c0[2], c1[2], n0[2], n1[2];
        if (MODE == 1) { for (int j = 0; j < 2; ++j) { c0[j] = *(const v4f*)(p0 + 4 * j); c1[j] = *(const v4f*)(p1 + 4 * j); } }
        else if (MODE == 2) {
            for (int j = 0; j < 2; ++j) asm volatile("s_mov_b64 exec, 0xff\n\tds_read_b128 %0, %1 offset:%2\n\ts_mov_b64 exec, -1" : "=&v"(c0[j]) : "v"(p8), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0[0]), "+v"(c0[1]));
            c1[0] = c0[0]; c1[1] = c0[1];
        } else { for (int j = 0; j < 2; ++j) { c0[j] = w[j]; c1[j] = w[j + 2]; } }
#pragma unroll
        for (int g = 0; g < 32; g += 2) {
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j) { n0[j] = *(const v4f*)(p0 + 4 * ((g + 2 + j) & 31)); n1[j] = *(const v4f*)(p1 + 4 * ((g + 2 + j) & 31)); }
            } else if (MODE == 2) {
                asm volatile("s_mov_b64 exec, 0xff\n\tds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_mov_b64 exec, -1"
                             : "=&v"(n0[0]), "=&v"(n0[1]) : "v"(p8), "n"(16 * ((0 + 2) & 31)), "n"(16 * ((0 + 3) & 31)));
            } else { n0[0] = c0[0]; n0[1] = c0[1]; n1[0] = c1[0]; n1[1] = c1[1]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const v4f x = c0[j], y = (MODE == 2) ? c0[j] : c1[j];
                a00 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[g + j][0], x[0], a00, 0, 0, 0); a10 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[g + j][0], y[0], a10, 0, 0, 0);
                a01 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[g + j][1], x[1], a01, 0, 0, 0); a11 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[g + j][1], y[1], a11, 0, 0, 0);
                a00 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[g + j][2], x[2], a00, 0, 0, 0); a10 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[g + j][2], y[2], a10, 0, 0, 0);
                a01 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[g + j][3], x[3], a01, 0, 0, 0); a11 = __builtin_amdgcn_mfma_f32_4x4x1f32(w[g + j][3], y[3], a11, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n0[0]), "+v"(n0[1]));
#pragma unroll
            for (int j = 0; j < 2; ++j) { c0[j] = n0[j]; c1[j] = (MODE == 2) ? n0[j] : n1[j]; }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    v4f s = a00 + a01 + a10 + a11;
    out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 1 << 22); (void)hipMalloc(&c, 4096 * 8);
    const int iters = 500;
    auto run = [&](auto kern, const char* nm) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, c, iters);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> cy(256);
        (void)hipMemcpy(cy.data(), c, 256 * 8, hipMemcpyDeviceToHost);
        printf("%-48s %7.2f cycles per MFMA (256 MFMAs per sweep, 4 waves per CU)\n", nm, (double)cy[0] / (iters * 256.0));
    };
    run(sweep<0>, "no LDS reads");
    run(sweep<1>, "64 lanes read, 2 x ds_read_b128 per 8 MFMAs");
    run(sweep<2>, "8 lanes read (exec mask), 1 x ds_read_b128 per 8 MFMAs");
    return 0;
}
