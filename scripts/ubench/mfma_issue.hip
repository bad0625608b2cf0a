// Microbenchmark: sustained issue rate of v_mfma_f32_16x16x4_f32 in the conv kernel's instruction mix.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int NMT = 13;

template <int VAR>
__global__ __launch_bounds__(512) void k(float* out, const float* wsrc, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 243 * 40; i += blockDim.x) lds[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    int aoff[NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
        const int r = 16 * m + (lane & 15), a = r / 13, j = r - a * 13;
        aoff[m] = (2 + a * 15 + j) * 40 + 4 * (lane >> 4);
    }
    v4f acc[NMT], af[2][NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) { acc[m] = v4f{0, 0, 0, 0}; af[0][m] = *(const v4f*)(lds + aoff[m]); af[1][m] = af[0][m]; }
    v4f b = *(const v4f*)(wsrc + lane * 4), bn = b;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 64 * 1024 + 4096, 0x00020000);
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (VAR == 2 || VAR == 3) bn = *(const v4f*)(wsrc + ((it + h + 2) & 63) * 256 + lane * 4);
            if (VAR == 4) {
                const int soff = __builtin_amdgcn_readfirstlane(((it + h + 2) & 63) * 1024);
                bn = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, soff, 0));
            }
#pragma unroll
            for (int g = 0; g < NMT; ++g) {
                if (VAR >= 1) af[h ^ 1][g] = *(const v4f*)(lds + aoff[g] + ((it + h) & 3) * 40);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = 4 * g + q, s = idx / NMT, m = idx % NMT;
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[h][m][s], b[s], acc[m], 0, 0, 0);
                }
                if (VAR != 3) __builtin_amdgcn_sched_barrier(0);
            }
            if (VAR >= 2) b = bn;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0;
#pragma unroll
    for (int m = 0; m < NMT; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
// f16x2 split: a = a_hi + a_lo (two fp16 planes), products hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16.
// One iteration = one (tap, 32-channel group): per M-tile 2 ds_read_b128 and 3 MFMAs; 3 weight fragments per iteration.
__global__ __launch_bounds__(512) void k_split(float* out, const float* wsrc, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 243 * 40 * 2; i += blockDim.x) lds[i] = 0.001f * (float)(i % 7);
    __syncthreads();
    int aoff[NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
        const int r = 16 * m + (lane & 15), a = r / 13, j = r - a * 13;
        aoff[m] = (2 + a * 15 + j) * 40 + 4 * (lane >> 4);
    }
    v4f acc[NMT], ah[2][NMT], al[2][NMT];
#pragma unroll
    for (int m = 0; m < NMT; ++m) { acc[m] = v4f{0, 0, 0, 0}; ah[0][m] = *(const v4f*)(lds + aoff[m]); al[0][m] = ah[0][m]; ah[1][m] = ah[0][m]; al[1][m] = ah[0][m]; }
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, 64 * 1024 + 4096, 0x00020000);
    v4f bh = *(const v4f*)(wsrc + lane * 4), bl = bh, bhn = bh, bln = bh;
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int soff = __builtin_amdgcn_readfirstlane(((it + h + 2) & 31) * 2048);
            bhn = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, soff, 0));
            bln = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, soff + 1024, 0));
#pragma unroll
            for (int g = 0; g < NMT; ++g) {
                ah[h ^ 1][g] = *(const v4f*)(lds + aoff[g] + ((it + h) & 3) * 40);
                al[h ^ 1][g] = *(const v4f*)(lds + 243 * 40 + aoff[g] + ((it + h) & 3) * 40);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ah[h][g]), __builtin_bit_cast(h8, bh), acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, ah[h][g]), __builtin_bit_cast(h8, bl), acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, al[h][g]), __builtin_bit_cast(h8, bh), acc[g], 0, 0, 0);
            }
            bh = bhn; bl = bln;
        }
    }
    float s = 0;
#pragma unroll
    for (int m = 0; m < NMT; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = 0;
}
void run_split(const char* name, int threads, int grid) {
    float *out, *w; unsigned long long* cyc;
    hipMalloc(&out, grid * 1024 * 4); hipMalloc(&w, 64 * 1024 * 4 + 8192); hipMalloc(&cyc, grid * 8);
    hipMemset(w, 0, 64 * 1024 * 4 + 8192);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k_split, hipFuncAttributeMaxDynamicSharedMemorySize, 243 * 40 * 8);
    hipLaunchKernelGGL(k_split, dim3(grid), dim3(threads), 243 * 40 * 8, 0, out, w, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_split, dim3(grid), dim3(threads), 243 * 40 * 8, 0, out, w, cyc, iters);
    hipEventRecord(e1, 0);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: FAILED %s\n", name, hipGetErrorString(e)); return; }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    // fp32-equivalent work: one iteration covers 32 channels for 13 M-tiles = 13 * 16*16*32 MAC
    const double tf = (double)grid * (threads / 64) * iters * 13.0 * 16 * 16 * 32 * 2 / (ms * 1e-3) / 1e12;
    printf("%-46s thr=%d grid=%d: wall %.3f ms = %.1f fp32-equivalent TFLOP/s (%.2fx the fp32-MFMA peak)\n", name, threads, grid, ms, tf, tf / 157.3);
    hipFree(out); hipFree(w); hipFree(cyc);
}

template <int VAR>
void run(const char* name, int threads, int grid) {
    float *out, *w; unsigned long long* cyc;
    hipMalloc(&out, grid * 1024 * 4); hipMalloc(&w, 64 * 256 * 4 + 4096); hipMalloc(&cyc, grid * 8);
    hipMemset(w, 0, 64 * 256 * 4 + 4096);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(threads), 243 * 40 * 4, 0, out, w, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(threads), 243 * 40 * 4, 0, out, w, cyc, iters);
    hipEventRecord(e1, 0);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)grid * (threads / 64) * iters * 52.0 * 2048.0 / (ms * 1e-3) / 1e12;
    if (e != hipSuccess) { printf("%s: FAILED %s\n", name, hipGetErrorString(e)); return; }
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-46s thr=%d grid=%d: %.2f memtime-cyc/MFMA/wave | wall %.3f ms = %.1f TFLOP/s (%.1f%% of 157.3)\n", name, threads, grid,
           (double)h[grid / 2] / iters / 52, ms, tf, 100 * tf / 157.3);
    hipFree(out); hipFree(w); hipFree(cyc);
}
int main() {
    run<0>("V0 pure MFMA, 13 accumulators", 256, 256);
    run<1>("V1 + 13 ds_read_b128 / 52 MFMA (pinned)", 256, 256);
    run<2>("V2 + B fragment global load / iter (pinned)", 256, 256);
    run<3>("V3 as V2, compiler-scheduled (no pins)", 256, 256);
    run<4>("V4 as V2 but buffer_load + SGPR offset", 256, 256);
    run<4>("V4 buffer_load, 8-wave WG", 512, 256);
    run_split("V5 f16x2 split mix, 1 wave/SIMD", 256, 256);
    run_split("V5 f16x2 split mix, 2 waves/SIMD", 512, 256);
    run<0>("V0 pure MFMA, 8-wave WG (2 waves/SIMD)", 512, 256);
    run<1>("V1 +ds_read pinned, 8-wave WG", 512, 256);
    run<2>("V2 full mix pinned, 8-wave WG", 512, 256);
    run<3>("V3 full mix unpinned, 8-wave WG", 512, 256);
    run<2>("V2 full mix pinned, 2 WGs of 4 waves per CU", 256, 512);
    run<3>("V3 full mix unpinned, 2 WGs of 4 waves per CU", 256, 512);
    return 0;
}
