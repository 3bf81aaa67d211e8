// chainlat.hip -- what a dependent operation costs a wave that is ALONE on its SIMD (gfx950), in shader cycles
// (s_memtime on both sides of a chain of N dependent operations, one wave per workgroup, one workgroup per CU, min over
// workgroups).  The sequential Kalman recursion (recursion_pair.hip / recursion_wave.hip) is such a wave: DESIGN.md quotes
// these numbers where it explains why instruction COUNT and not dependency depth bounded it.
//   fma64      v_fma_f64 chain                      cnd32     v_cndmask_b32 chain
//   rcp64      v_rcp_f64 chain                      dpp       v_mov_b32 quad_perm -> v_add_f64 chain
//   bperm      ds_bpermute_b32 x 2 -> v_add_f64     lds_rt    ds_write_b64 -> ds_read_b64 (same wave, fence only)
//   readlane   v_readlane_b32 x 2 -> v_fma_f64 with the SGPR pair
//   indep      8 independent v_fma_f64 chains interleaved (issue rate: cycles per instruction)
//   barrier4 / barrier16   ds_write_b64 -> s_barrier -> ds_read_b64 in a workgroup of 4 / 16 waves
// Build: hipcc --offload-arch=gfx950 -O3 chainlat.hip -o chainlat ; run: ./chainlat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N = 512;

__device__ __forceinline__ unsigned long long now() {          // shader-clock counter
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long now_rt() {       // constant 100 MHz counter
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ double shfl64(double v, int src) {
    int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, double seed) {
    __shared__ double buf[1024 + 64];
    const int l = threadIdx.x, lane = l & 63;
    double x = seed + 1e-3 * lane, y = 1.0 + 1e-9 * lane;
    double z[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) z[q] = x + q;
    buf[l] = x;
    __syncthreads();
    const unsigned long long r0 = now_rt();
    const unsigned long long t0 = now();
    if constexpr (MODE == 0) {
#pragma unroll 16
        for (int i = 0; i < N; ++i) x = fma(x, y, 1e-9);
    } else if constexpr (MODE == 1) {
#pragma unroll 16
        for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x) + 1.5;   // (rcp + add: subtract fma64's figure)
    } else if constexpr (MODE == 2) {
#pragma unroll 16
        for (int i = 0; i < N; ++i) x = shfl64(x, lane ^ 9) + y;
    } else if constexpr (MODE == 3) {
#pragma unroll 16
        for (int i = 0; i < N; ++i) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(x), 7), hi = __builtin_amdgcn_readlane(__double2hiint(x), 7);
            x = fma(__hiloint2double(hi, lo), 1e-9, x);
        }
    } else if constexpr (MODE == 4) {
        int a = __double2loint(x), b = lane;
#pragma unroll 16
        for (int i = 0; i < N; ++i) { a = (a & 1) ? b : a + 3; }
        x = (double)a;
    } else if constexpr (MODE == 5) {
#pragma unroll 16
        for (int i = 0; i < N; ++i) {
            int lo = __double2loint(x), hi = __double2hiint(x);
            lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
            hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
            x = x + __hiloint2double(hi, lo) * 0.5;
        }
    } else if constexpr (MODE == 6) {
        volatile double* vb = buf;
#pragma unroll 16
        for (int i = 0; i < N; ++i) {
            vb[lane] = x;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            x = vb[lane ^ 9] + y;
        }
    } else if constexpr (MODE == 7) {
#pragma unroll 4
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int q = 0; q < 8; ++q) z[q] = fma(z[q], y, 1e-9);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) x += z[q];
    } else {   // 8: exchange through LDS with a workgroup barrier
        for (int i = 0; i < N; ++i) {
            buf[l] = x;
            __syncthreads();
            x = buf[(l + 64) % blockDim.x] + y;
            __syncthreads();
        }
    }
    const unsigned long long t1 = now();
    const unsigned long long r1 = now_rt();
    if (lane == 0 && (l >> 6) == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = r1 - r0; }
    if (x == 1.2345e300) out[0] = x;
}

template <int MODE>
static void run(const char* name, int threads, double per, double* out, unsigned long long* cyc, int nb) {
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(threads), 0, 0, out, cyc, 1.25);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(threads), 0, 0, out, cyc, 1.25);
    CK(hipDeviceSynchronize());
    unsigned long long h[512], mn = ~0ull, mr = ~0ull;
    CK(hipMemcpy(h, cyc, 2 * nb * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < nb; ++i) { if (h[2 * i] < mn) mn = h[2 * i]; if (h[2 * i + 1] < mr) mr = h[2 * i + 1]; }
    printf("%-11s %7.1f s_memtime ticks = %6.1f ns per %s\n", name, (double)mn / (N * per), 10.0 * (double)mr / (N * per),
           per > 1.5 ? "instruction" : "link of the chain");
}

int main() {
    double* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 512 * sizeof(unsigned long long)));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("%s, %d CUs, clockRate %d kHz (s_memrealtime: 100 MHz; ns x clockRate = shader cycles at the top clock)\n",
           pr.name, pr.multiProcessorCount, pr.clockRate);
    const int nb = 128;
    run<0>("fma64", 64, 1, out, cyc, nb);
    run<1>("rcp64+add", 64, 1, out, cyc, nb);
    run<2>("bperm+add", 64, 1, out, cyc, nb);
    run<3>("readln+fma", 64, 1, out, cyc, nb);
    run<4>("cnd32", 64, 1, out, cyc, nb);
    run<5>("dpp+fma", 64, 1, out, cyc, nb);
    run<6>("lds_rt+add", 64, 1, out, cyc, nb);
    run<7>("indep", 64, 8, out, cyc, nb);
    run<8>("barrier4", 256, 1, out, cyc, nb);
    run<8>("barrier16", 1024, 1, out, cyc, nb);
    return 0;
}
