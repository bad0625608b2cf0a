// Standalone check of ONE wino1d_edge.hip launch against a CPU direct convolution + GroupNorm + Mish in double (a debug aid: it found
// the 16-byte-store / scalar-offset problem noted in DESIGN 4.10).  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -Icontrollable-latent-diffusion-for-traffic-simulation_amd/csrc -Iinclude \
//         -o scripts/ubench/edge_unit scripts/ubench/edge_unit.hip  &&  scripts/ubench/edge_unit <L> <C_in> <C_out>
#include "../../controllable-latent-diffusion-for-traffic-simulation_amd/csrc/wino1d_edge.hip"
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
template <class F>
std::vector<float> pack(F&& wget, int c_out, int cin, int ntaps) {      // cld_api.hip pack_conv_weights
    const int ngrp = cin / 16, ntn = c_out / 16;
    std::vector<float> out((size_t)ngrp * ntaps * ntn * 256);
    size_t o = 0;
    for (int kgg = 0; kgg < ngrp; ++kgg)
        for (int t = 0; t < ntaps; ++t)
            for (int nt = 0; nt < ntn; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s) out[o++] = wget(16 * nt + (lane & 15), 16 * kgg + 4 * (lane >> 4) + s, t);
    return out;
}
int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int L = atoi(argv[1]), C = atoi(argv[2]), N = atoi(argv[3]), B = 32;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd;
    std::vector<float> x((size_t)B * L * C), w((size_t)N * C * 5), bias(N), gam(N, 1.f), bet(N, 0.f);
    for (auto& v : x) v = nd(rng);
    for (auto& v : w) v = nd(rng) / std::sqrt(5.f * C);
    for (auto& v : bias) v = 0.1f * nd(rng);
    static const double Gm[8][5] = {{-1, 0, 0, 0, 0}, {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9}, {-2.0 / 9, 2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},
                                    {1.0 / 90, 1.0 / 45, 2.0 / 45, 4.0 / 45, 8.0 / 45}, {1.0 / 90, -1.0 / 45, 2.0 / 45, -4.0 / 45, 8.0 / 45},
                                    {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45}, {32.0 / 45, -16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45}, {0, 0, 0, 0, 1}};
    auto eget = [&](int co, int ci, int pl) -> float {
        if (pl >= 8) return w[((size_t)co * C + ci) * 5 + (pl - 8)];
        double u = 0;
        for (int k = 0; k < 5; ++k) u += Gm[pl][k] * w[((size_t)co * C + ci) * 5 + k];
        return (float)u;
    };
    std::vector<float> uf = pack(eget, N, C, 12);
    float *dx, *dw, *db, *dg, *dbe, *dy;
    (void)hipMalloc(&dx, x.size() * 4); (void)hipMalloc(&dw, uf.size() * 4); (void)hipMalloc(&db, N * 4); (void)hipMalloc(&dg, N * 4);
    (void)hipMalloc(&dbe, N * 4); (void)hipMalloc(&dy, (size_t)B * L * N * 4);
    (void)hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dw, uf.data(), uf.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, bias.data(), N * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dg, gam.data(), N * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dbe, bet.data(), N * 4, hipMemcpyHostToDevice);
    (void)hipMemset(dy, 0xff, (size_t)B * L * N * 4);
    cld::ConvArgs a{};
    a.x1 = dx; a.c1_real = C; a.c1_pad = C; a.wfrag = dw; a.bias = db; a.gamma = dg; a.beta = dbe; a.y = dy; a.c_out = N; a.ly = L;
    hipError_t e = cld::launch_wino1d_edge(a, L, B, false, 0);
    (void)hipDeviceSynchronize();
    printf("launch: %s / %s\n", hipGetErrorString(e), hipGetErrorString(hipGetLastError()));
    std::vector<float> y((size_t)B * L * N);
    (void)hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost);
    if (argc > 4) {      // timing mode: <L> <C_in> <C_out> <rows>: the launch with its weight planes warm in L2 / Infinity Cache vs after 1 GB of other traffic
        const int BT = atoi(argv[4]);
        float *tx, *ty, *flush;
        (void)hipMalloc(&tx, (size_t)BT * L * C * 4); (void)hipMalloc(&ty, (size_t)BT * L * N * 4); (void)hipMalloc(&flush, (size_t)1 << 30);
        (void)hipMemset(tx, 0, (size_t)BT * L * C * 4);
        cld::ConvArgs t = a; t.x1 = tx; t.y = ty;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int mode = 0; mode < 2; ++mode) {
            float tot = 0;
            for (int it = 0; it < 12; ++it) {
                if (mode == 1) (void)hipMemsetAsync(flush, it, (size_t)1 << 30, 0);
                (void)hipEventRecord(e0, 0);
                (void)cld::launch_wino1d_edge(t, L, BT, false, 0);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (it >= 2) tot += ms;
            }
            printf("%s: %.1f us per launch (%d rows)\n", mode ? "after 1 GB of other traffic" : "back to back (weights and rows warm)", tot / 10 * 1e3, BT);
        }
        return 0;
    }
    const int GS = N / 8;
    double all = 0;
    for (int b = 0; b < B; ++b) {
        std::vector<double> c((size_t)L * N);
        for (int l = 0; l < L; ++l)
            for (int n = 0; n < N; ++n) {
                double s = bias[n];
                for (int k = 0; k < 5; ++k) {
                    const int p = l + k - 2;
                    if (p < 0 || p >= L) continue;
                    for (int ci = 0; ci < C; ++ci) s += (double)w[((size_t)n * C + ci) * 5 + k] * x[((size_t)b * L + p) * C + ci];
                }
                c[(size_t)l * N + n] = s;
            }
        double worst = 0, errpos[32] = {0};
        int wl = -1, wn = -1;
        for (int g = 0; g < 8; ++g) {
            double m = 0, v = 0;
            for (int l = 0; l < L; ++l) for (int n = g * GS; n < (g + 1) * GS; ++n) m += c[(size_t)l * N + n];
            m /= GS * L;
            for (int l = 0; l < L; ++l) for (int n = g * GS; n < (g + 1) * GS; ++n) { const double d = c[(size_t)l * N + n] - m; v += d * d; }
            v /= GS * L;
            for (int l = 0; l < L; ++l)
                for (int n = g * GS; n < (g + 1) * GS; ++n) {
                    const double z = (c[(size_t)l * N + n] - m) / std::sqrt(v + 1e-5), r = z * std::tanh(std::log1p(std::exp(z)));
                    const double er = std::fabs(r - y[((size_t)b * L + l) * N + n]);
                    if (!(er <= worst)) { worst = er; wl = l; wn = n; }
                    if (!(er <= errpos[l])) errpos[l] = er;
                }
        }
        if (!(worst <= all)) all = worst;
        printf("agent %2d worst %.3e at pos %d ch %d | per pos:", b, worst, wl, wn);
        for (int l = 0; l < L; ++l) printf(" %.0e", errpos[l]);
        printf("\n");
    }
    printf("worst over the batch: %.3e\n", all);
    return all <= 1e-4 ? 0 : 1;
}
