// colbw.hip -- the balanced-panel collapse kernel timed alone, back to back (no launch gaps in the
// average), for every tuning variant of collapse_dma.hip at BASELINE config 2 (B=1024, N=200, T=500, r=8).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../dynamic_factor_models_amd/csrc colbw.hip -o colbw
#include "../../dynamic_factor_models_amd/csrc/collapse_dma.hip"
#include "../../dynamic_factor_models_amd/csrc/collapse_mfma.hip"
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
namespace dfm { int collapse_max_n(int) { return 1024; } }
int main(int argc, char** argv) {
    const int B = 1024, T = 500, N = 200, R = 8;
    double *panel, *Lam, *Rv, *bcol, *ssum, *scol; int* status;
    CK(hipMalloc(&panel, (size_t)B * T * N * 8)); CK(hipMalloc(&Lam, (size_t)B * N * R * 8)); CK(hipMalloc(&Rv, (size_t)B * N * 8));
    CK(hipMalloc(&bcol, (size_t)B * T * R * 8)); CK(hipMalloc(&ssum, (size_t)B * 16 * 8)); CK(hipMalloc(&status, 256)); CK(hipMalloc(&scol, (size_t)B * T * 8)); CK(hipMemset(scol, 0, (size_t)B * T * 8));
    std::vector<double> h((size_t)B * T * N);
    const bool rnd = getenv("COLBW_RANDOM") != nullptr;   // full-mantissa pseudo-normal data (power draw like the bench)
    unsigned long long st = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) {
        if (rnd) {
            double acc = 0;
            for (int k = 0; k < 4; ++k) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; acc += (double)(st >> 11) / 9007199254740992.0; }
            h[i] = (acc - 2.0) * 1.7320508;
        } else {
            h[i] = (double)((i * 2654435761u) % 1000) / 500.0 - 1.0;
        }
    }
    CK(hipMemcpy(panel, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(Lam, h.data(), (size_t)B * N * R * 8, hipMemcpyHostToDevice));
    for (size_t i = 0; i < (size_t)B * N; ++i) h[i] = 1.0 + (double)(i % 7) * 0.1;
    CK(hipMemcpy(Rv, h.data(), (size_t)B * N * 8, hipMemcpyHostToDevice));
    CK(hipMemset(status, 0, 256));
    dfm::CollapseArgs a; memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.N = N; a.panel = panel; a.Lam = Lam; a.Rv = Rv; a.bcol = bcol; a.ssum = ssum; a.status = status; a.scol = scol;
    {
        int nb = 0;
        for (size_t lds : {(size_t)32768, (size_t)40960, (size_t)51200, (size_t)53248, (size_t)65536}) {
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dfm::collapse_mfma_kernel<8, 25, 2, 2, 0, false>, 256, lds));
            printf("occupancy API: collapse_mfma_kernel<8,25,2,2,0> with %zu B LDS -> %d blocks/CU\n", lds, nb);
        }
        hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)dfm::collapse_mfma_kernel<8, 25, 2, 2, 0, false>));
        printf("numRegs %d sharedSizeBytes %zu maxDynamicShared %d\n", fa.numRegs, fa.sharedSizeBytes, fa.maxDynamicSharedSizeBytes);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<int> vars;
    for (int i = 1; i < argc; ++i) vars.push_back(atoi(argv[i]));
    if (vars.empty()) vars = {107, 3200, 3203, 3201, 4200, 3200, 6200};
    std::vector<double> ref((size_t)B * T * R), got((size_t)B * T * R);
    for (int v0 : vars) {
        const int v = v0 % 1000; a.wpr = v0 / 1000;   // 1000 * wpr + variant
        CK(hipMemset(bcol, 0, (size_t)B * T * R * 8));
        for (int i = 0; i < 3; ++i) CK(dfm::launch_collapse_dma(R, a, 0, v));
        CK(hipDeviceSynchronize());
        const int K = 20;
        CK(hipEventRecord(e0));
        for (int i = 0; i < K; ++i) CK(dfm::launch_collapse_dma(R, a, 0, v));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // checksum of a few collapsed rows against variant 0 (same arithmetic order inside a row)
        CK(hipMemcpy(got.data(), bcol, got.size() * 8, hipMemcpyDeviceToHost));
        if (v0 == vars[0]) ref = got;
        double d = 0; for (size_t i = 0; i < got.size(); ++i) d = fmax(d, fabs(got[i] - ref[i]));
        if (v >= 200 && v % 10 == 2) {
            std::vector<double> sc((size_t)B * T);
            CK(hipMemcpy(sc.data(), scol, sc.size() * 8, hipMemcpyDeviceToHost));
            double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int ns = a.wpr > 0 ? a.wpr : 4;
            for (int b = 0; b < B; ++b) for (int w = 0; w < ns; ++w) for (int k = 0; k < 8; ++k) m[k] += sc[(size_t)b * T + w * 8 + k] / (B * (double)ns);
            {   // wave records: concurrency = sum of wave lifetimes / kernel span; tick rate = ticks / real time
                double tmin = 1e300, tmax = 0, life = 0, ticks = 0; int n = 0;
                for (int b = 0; b < B; ++b) for (int w = 0; w < ns; ++w) {
                    const double* o = &sc[(size_t)b * T + 200 + w * 4];
                    if (o[1] <= o[0]) continue;
                    tmin = fmin(tmin, o[0]); tmax = fmax(tmax, o[1]); life += o[1] - o[0]; ticks += o[2]; ++n;
                }
                {
                    const int NBIN = 24; double bins[NBIN] = {0};
                    const double bw = (tmax - tmin) / NBIN;
                    for (int b = 0; b < B; ++b) for (int w = 0; w < ns; ++w) {
                        const double* o = &sc[(size_t)b * T + 200 + w * 4];
                        if (o[1] <= o[0]) continue;
                        for (int k = 0; k < NBIN; ++k) {
                            const double lo = tmin + k * bw, hi = lo + bw;
                            const double ov = fmin(hi, o[1]) - fmax(lo, o[0]);
                            if (ov > 0) bins[k] += ov / bw;
                        }
                    }
                    printf("   live waves per CU over time:");
                    for (int k = 0; k < NBIN; ++k) printf(" %.1f", bins[k] / 256.0);
                    printf("\n");
                }
                printf("   %d waves: kernel span %.1f us (100 MHz real-time ticks), mean wave life %.1f us, mean concurrency %.1f waves/CU, s_memtime rate %.2f GHz\n",
                       n, (tmax - tmin) / 100.0, life / n / 100.0, life / (tmax - tmin) / 256.0, ticks / life / 10.0);
            }
            printf("   per wave (s_memtime ticks): wait %.0f  read %.0f  issue %.0f  compute+store %.0f  total %.0f  blocks %.0f  prologue %.0f  epilogue %.0f\n", m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
        }
        printf("wpr %d variant %3d  %8.4f ms  %7.1f GB/s   maxdiff_vs_first %.3g\n", a.wpr, v, ms / K, (double)B * (T * N + N * R + N) * 8 / (ms / K * 1e-3) / 1e9, d);
    }
    return 0;
}
