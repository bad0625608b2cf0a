// What v_permlane16_swap / v_permlane32_swap (gfx950) do, lane by lane.  build: hipcc --offload-arch=gfx950 -O3 -o permlane permlane.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
    out[l] = a[0]; out[64 + l] = a[1]; out[128 + l] = b[0]; out[192 + l] = b[1];
}
__device__ __forceinline__ float sum_over_units(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    const auto r16 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    const unsigned a16 = r16[0], b16 = r16[1];      // (scalars first: __builtin_bit_cast of a vector ELEMENT reads element 0, see bperm)
    x = __builtin_bit_cast(float, a16) + __builtin_bit_cast(float, b16);
    const auto r32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    const unsigned a32 = r32[0], b32 = r32[1];
    return __builtin_bit_cast(float, a32) + __builtin_bit_cast(float, b32);
}
__global__ void k2(float* out) {
    const int l = threadIdx.x;
    out[l] = sum_over_units((float)(1 << (l >> 2)) * (1 + (l & 3)));      // unit ul -> bit ul; expect 65535 * (1 + q) in every lane
    float a = (float)l, b = (float)(2 * l);
    out[64 + l] = sum_over_units(a) + 0.f * b;
    out[128 + l] = sum_over_units(b);
}
int main() {
    {
        float* d; (void)hipMalloc(&d, 192 * 4);
        hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d);
        float h[192]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("sum_over_units(bit ul * (1 + q)):"); for (int l = 0; l < 64; l += 7) printf(" %g", h[l]); printf("   expect 65535 * (1 + (lane & 3))\n");
        printf("sum_over_units(lane):"); for (int l = 0; l < 8; ++l) printf(" %g", h[64 + l]); printf("   expect 480 + 16 q\n");
        printf("sum_over_units(2 lane):"); for (int l = 0; l < 8; ++l) printf(" %g", h[128 + l]); printf("\n");
    }
    unsigned* d; (void)hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[4] = {"permlane16_swap(old = lane, src = 100 + lane) -> [0]", "                                                  -> [1]", "permlane32_swap -> [0]", "                -> [1]"};
    for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int l = 0; l < 64; l += 8) printf(" %3u", h[r * 64 + l]); printf("   (lanes 0, 8, .., 56)\n"); }
    return 0;
}
