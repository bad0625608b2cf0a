// Stand-alone timing / phase stamps of ONE wino44_kernels.hip launch (debug aid):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DW44_STAMPS -Icontrollable-latent-diffusion-for-traffic-simulation_amd/csrc -Iinclude \
//         -o scripts/ubench/w44_unit scripts/ubench/w44_unit.hip  &&  scripts/ubench/w44_unit <56|28> <agents>
#include "../../controllable-latent-diffusion-for-traffic-simulation_amd/csrc/wino44_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
    const int H = atoi(argv[1]), B = atoi(argv[2]), C = H == 56 ? 64 : 128;
    const size_t n = (size_t)B * H * H * C;
    float *x, *y, *r, *u, *sc, *sh;
    (void)hipMalloc(&x, n * 4); (void)hipMalloc(&y, n * 4); (void)hipMalloc(&r, n * 4); (void)hipMalloc(&u, (size_t)36 * C * C * 4); (void)hipMalloc(&sc, C * 4); (void)hipMalloc(&sh, C * 4);
    (void)hipMemset(x, 0, n * 4); (void)hipMemset(r, 0, n * 4); (void)hipMemset(u, 0, (size_t)36 * C * C * 4); (void)hipMemset(sc, 0, C * 4); (void)hipMemset(sh, 0, C * 4);
    cld::WinoArgs a{x, u, sc, sh, r, y, B, 1};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) {
        (void)hipEventRecord(e0, 0);
        hipError_t e = cld::launch_wino44_conv(H, C, a, 0);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %s %.1f us\n", it, hipGetErrorString(e), ms * 1e3);
    }
#ifdef W44_STAMPS
    const int nwg = std::min(8192, (B * (H / 4) * (H / 4) + 15) / 16 * (C / 64));
    std::vector<unsigned long long> st((size_t)8192 * 8);
    (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(cld::w44_stamps), st.size() * 8);
    const char* names[] = {"entry -> first image + second patch requested", "first chunk pair", "remaining chunk pairs", "M A (rows of the output transform)", "-", "A^T (M A) + BatchNorm + residual + stores"};
    double sum[6] = {0}; unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < nwg; ++w) { for (int k = 0; k < 6; ++k) sum[k] += (double)(st[w * 8 + k + 1] - st[w * 8 + k]); t0 = std::min(t0, st[w * 8]); t1 = std::max(t1, st[w * 8 + 6]); }
    printf("%d workgroups, kernel span %llu cycles (s_memtime, 100 MHz?)\n", nwg, t1 - t0);
    double life = 0; for (int k = 0; k < 6; ++k) life += sum[k] / nwg;
    for (int k = 0; k < 6; ++k) printf("   %-52s mean %9.0f\n", names[k], sum[k] / nwg);
    printf("   workgroup life mean %.0f\n", life);
#endif
    return 0;
}
