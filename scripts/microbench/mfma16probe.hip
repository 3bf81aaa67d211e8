// Lane layout probe of v_mfma_f64_16x16x4_f64 on gfx950: D = A B with A[i][k] = (i + 1) [k == 0], B[k][j] = 100 (j + 1) [k == 0]
// under the ASSUMED operand layout (A[i][k] in lane 16 k + i, B[k][j] in lane 16 k + j), then a second run with k == 3 only.
// Prints, for every lane and result register, the (i, j) decoded from the value.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, int ksel) {
    const int l = threadIdx.x;
    const int k = l >> 4, c = l & 15;
    const double a = (k == ksel) ? (double)(c + 1) : 0.0;
    const double b = (k == ksel) ? 100.0 * (double)(c + 1) : 0.0;
    v4 d = {0.0, 0.0, 0.0, 0.0};
    d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = d[v];
}
int main() {
    double* d; hipMalloc(&d, 64 * 4 * sizeof(double));
    double h[256];
    for (int ksel = 0; ksel < 4; ksel += 3) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, ksel);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("ksel=%d\n", ksel);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int v = 0; v < 4; ++v) {
                const long val = (long)(h[l * 4 + v] + 0.5);
                // val = (i+1) * 100 * (j+1): ambiguous factorisation; print raw and the decode assuming j = l % 16
                const long jj = (l & 15) + 1;
                printf("  v%d=%ld (i=%ld if j=l%%16)", v, val, val % (100 * jj) == 0 ? val / (100 * jj) - 1 : -1);
            }
            printf("\n");
            if (l == 17) l = 46;   // a sample of lanes is enough
        }
    }
    return 0;
}
