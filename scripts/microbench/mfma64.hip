// mfma64.hip -- v_mfma_f64_16x16x4_f64 on gfx950: operand/result lane layout (checked against a host
// product) and issue rate alone and beside a VALU-only wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A, const double* B, double* D) {   // A[16][4], B[4][16] row-major, D raw [64][4]
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];
    const double b = B[(l / 16) * 16 + l % 16];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
}

template <int MODE>   // 0: mfma only, 1: valu fma only, 2: even waves mfma / odd waves valu
__global__ __launch_bounds__(512) void k_rate(double* out, int iters) {
    const int wave = threadIdx.x >> 6;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double f0 = a, f1 = b, f2 = a + b, f3 = a - b, f4 = a * 2, f5 = b * 2, f6 = a * 3, f7 = b * 3;
    const bool do_mfma = MODE == 0 || (MODE == 2 && (wave & 1) == 0);
    const bool do_valu = MODE == 1 || (MODE == 2 && (wave & 1) == 1);
    for (int i = 0; i < iters; ++i) {
        if (do_mfma) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f0 = fma(f0, a, b); f1 = fma(f1, a, b); f2 = fma(f2, a, b); f3 = fma(f3, a, b);
                f4 = fma(f4, a, b); f5 = fma(f5, a, b); f6 = fma(f6, a, b); f7 = fma(f7, a, b);
            }
        }
    }
    double s = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (s == 1.2345e300) out[0] = s;
}

int main() {
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = 1 + (i * 37 % 11); B[i] = 2 + (i * 53 % 7); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        if (D[l * 4 + v] != R[(4 * (l / 16) + v) * 16 + l % 16]) okA = 0;     // hypothesis A: i = 4*(l/16)+v
        if (D[l * 4 + v] != R[((l / 16) + 4 * v) * 16 + l % 16]) okB = 0;     // hypothesis B: i = l/16 + 4v
    }
    printf("layout: a=A[l%%16][l/16], b=B[l/16][l%%16]; D[i][l%%16] with i=4*(l/16)+v: %s ; i=l/16+4v: %s\n", okA ? "YES" : "no", okB ? "YES" : "no");
    if (!okA && !okB) { for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g | R row0: %g %g\n", l, D[l*4], D[l*4+1], D[l*4+2], D[l*4+3], R[l], R[16 + l]); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000, blocks = 256 * 2;
    auto run = [&](const char* nm, auto kern, double mf, double vf) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, dD, iters); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, dD, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double waves = (double)blocks * 8;
        printf("%-28s %8.3f ms  mfma %.1f TF  valu %.1f TF\n", nm, ms, mf * waves * iters * 4 * 2048 / (ms * 1e-3) / 1e12, vf * waves * iters * 64 * 64 * 2 / (ms * 1e-3) / 1e12);
    };
    run("mfma only (8 waves/CU x2)", k_rate<0>, 1.0, 0.0);
    run("valu fma only", k_rate<1>, 0.0, 1.0);
    run("half mfma / half valu", k_rate<2>, 0.5, 0.5);
    return 0;
}
