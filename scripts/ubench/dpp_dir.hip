// dpp_dir.hip -- which way the DPP row rotations used by chain_wino.hip move data: prints, for lanes 0, 1, 15, 16, the source
// lane of row_ror:1 (0x121) and row_ror:15 (0x12F).  Expected: row_ror:1 -> lane i reads lane i - 1 (lane 0 reads lane 15 of its row).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int l = threadIdx.x;
    out[l] = __builtin_amdgcn_update_dpp(0, l, 0x121, 0xf, 0xf, false);
    out[64 + l] = __builtin_amdgcn_update_dpp(0, l, 0x12F, 0xf, 0xf, false);
}
int main() {
    int* d;
    hipMalloc(&d, 128 * sizeof(int));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[128];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 15, 16, 17, 31, 63}) printf("lane %2d: row_ror:1 reads lane %2d, row_ror:15 reads lane %2d\n", l, h[l], h[64 + l]);
    bool ok = true;
    for (int l = 0; l < 64; ++l) ok = ok && h[l] == (l & ~15) + ((l + 15) & 15) && h[64 + l] == (l & ~15) + ((l + 1) & 15);
    printf(ok ? "DPP_DIR_OK\n" : "DPP_DIR_MISMATCH\n");
    return ok ? 0 : 1;
}
