// Does `global_load_lds_dwordx4` accept a source address that is 8-byte but not 16-byte aligned (panel rows of an ODD number
// of doubles)?  Copies 64 x 16 bytes from src + 8 bytes into LDS and back out; prints the number of mismatching doubles.
#include <hip/hip_runtime.h>
#include <stdio.h>
using lds_ptr = __attribute__((address_space(3))) char*;
__global__ void probe(const double* src, double* out, int shift_doubles) {
    __shared__ __attribute__((aligned(16))) double buf[128];
    const int lane = threadIdx.x;
    const char* g = reinterpret_cast<const char*>(src + shift_doubles) + 16 * lane;
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)(buf));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(base) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[2 * lane] = buf[2 * lane];
    out[2 * lane + 1] = buf[2 * lane + 1];
}
int main() {
    double h[256], o[128];
    for (int i = 0; i < 256; ++i) h[i] = 1000.0 + i;
    double *d, *dout;
    hipMalloc(&d, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int shift = 0; shift <= 3; ++shift) {
        hipMemset(dout, 0, sizeof(o));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, dout, shift);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 128; ++i) bad += (o[i] != h[i + shift]);
        printf("shift %d doubles (%d bytes): %s, mismatches %d (o[0]=%.0f expect %.0f)\n", shift, 8 * shift, hipGetErrorString(e), bad, o[0], h[shift]);
    }
    return 0;
}
