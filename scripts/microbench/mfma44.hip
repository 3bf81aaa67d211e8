// mfma44.hip -- v_mfma_f64_4x4x4_4b_f64 on gfx950: discover which (a-lane, b-lane) products feed each
// output lane, and the issue rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_probe(double* D) {   // block lb: b = 1 in lane lb only; a = 1 + lane
    const int l = threadIdx.x, lb = blockIdx.x;
    const double a = 1.0 + l, b = (l == lb) ? 1.0 : 0.0;
    double c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    D[lb * 64 + l] = c;
}
__global__ __launch_bounds__(512) void k_rate(double* out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
    }
    double s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    if (s == 1.2345e300) out[0] = s;
}
__global__ __launch_bounds__(512) void k_rate_dep(double* out, int iters) {   // one dependent chain per wave
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, c0 = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    }
    if (c0 == 1.2345e300) out[0] = c0;
}
int main() {
    double* dD; CK(hipMalloc(&dD, 64 * 64 * 8));
    hipLaunchKernelGGL(k_probe, dim3(64), dim3(64), 0, 0, dD);
    std::vector<double> D(64 * 64);
    CK(hipMemcpy(D.data(), dD, 64 * 64 * 8, hipMemcpyDeviceToHost));
    for (int lb = 0; lb < 64; ++lb) {
        printf("b-lane %2d ->", lb);
        for (int lo = 0; lo < 64; ++lo) if (D[lb * 64 + lo] != 0.0) printf(" out%d<-a%d", lo, (int)D[lb * 64 + lo] - 1);
        printf("\n");
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000, blocks = 512;
    for (int mode = 0; mode < 2; ++mode) {
        auto launch = [&] { if (mode == 0) hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(512), 0, 0, dD, iters); else hipLaunchKernelGGL(k_rate_dep, dim3(blocks), dim3(512), 0, 0, dD, iters); };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double n = (double)blocks * 8 * iters * 8;   // wave-instructions
        printf("%s: %.3f ms, %.1f TF (256 FMA per instr), %.1f cycles/instr/SIMD at 2.4 GHz\n", mode ? "dependent chain" : "8 independent", ms, n * 512 / (ms * 1e-3) / 1e12, (ms * 1e-3 * 2.4e9) / (n / 1024));
    }
    return 0;
}
