// icache.hip -- is the instruction cache warm at the start of a launch, and what does cold code cost next to a stream?
// One wave (workgroup 0) runs a 48 KB straight-line chain of v_fma_f64 twice in one launch (pass 1, pass 2) and the launch
// is repeated 4 times; the other 255 workgroups idle or stream 4 GiB with 16-byte loads until the chain is done.
//   pass 1 of launch >= 2 as fast as pass 2  -> the cache keeps the kernel's code across launches
//   (second half: the same with the wave reading its own code as data first -- does a warm L2 make cold code cheap?)
//   pass 1 slow in EVERY launch              -> code is fetched again at every launch (and under load every line waits
//                                                behind the stream): the first replicate of every role of the one-launch
//                                                pass (pass_fused.hip) then runs on cold code
// Build: hipcc --offload-arch=gfx950 -O3 icache.hip -o icache ; run: ./icache
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now_rt() {       // constant 100 MHz counter
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

#define F8(x, y) asm volatile("v_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1\n\t" \
                              "v_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1\n\tv_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y));
#define F64(x, y) F8(x, y) F8(x, y) F8(x, y) F8(x, y) F8(x, y) F8(x, y) F8(x, y) F8(x, y)
#define F512(x, y) F64(x, y) F64(x, y) F64(x, y) F64(x, y) F64(x, y) F64(x, y) F64(x, y) F64(x, y)
#define F4096(x, y) F512(x, y) F512(x, y) F512(x, y) F512(x, y) F512(x, y) F512(x, y) F512(x, y) F512(x, y)

__global__ __launch_bounds__(256) void k(const u4* big, size_t nbig16, int stream, unsigned* flag, unsigned long long* ticks,
                                         double* sink, double seed, int prefetch) {
    if (blockIdx.x == 0) {
        if (threadIdx.x >= 64) return;
        double x = seed, y = 0.999;
        unsigned long long t[3];
        if (prefetch) {
            // the code that follows, read as DATA (48 KB = 48 loads of 1 KB per wave, all in flight): the lines are in this
            // XCD's L2 when the instruction cache asks for them
            const char* pc = reinterpret_cast<const char*>(__builtin_amdgcn_s_getpc());
            const u4* c = reinterpret_cast<const u4*>(reinterpret_cast<size_t>(pc) & ~size_t(1023)) + threadIdx.x;
            unsigned acc = 0;
#pragma unroll
            for (int u = 0; u < 48; ++u) acc ^= __builtin_nontemporal_load(c + 64 * u)[0];
            if (acc == 0x12345u) sink[2] = acc;
        }
        t[0] = now_rt();
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            F4096(x, y)                                          // 4096 x 8 bytes of code (v_fma_f64 is VOP3) ... plus:
            F512(x, y) F512(x, y) F512(x, y) F512(x, y)          // 48 KB in all
            t[pass + 1] = now_rt();
        }
        if (threadIdx.x == 0) {
            ticks[0] = t[1] - t[0]; ticks[1] = t[2] - t[1];
            sink[0] = x;
            __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (!stream) return;
    const size_t per = nbig16 / (gridDim.x - 1);
    const u4* base = big + per * (blockIdx.x - 1);
    unsigned acc = 0;
    unsigned long long n = 0;
    for (size_t off = threadIdx.x; ; off += 256 * 8) {
        if (off + 256 * 7 >= per) off = threadIdx.x;
        u4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(base + off + 256 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][3];
        n += 8;
        if ((n & 63) == 0 && __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    if (acc == 0x12345678u) sink[1] = acc;
}

int main() {
    const size_t bigB = 4ull << 30;
    unsigned long long* ticks;
    double* sink;
    u4* big;
    unsigned* flag;
    CK(hipMalloc(&big, bigB));
    CK(hipMalloc(&ticks, 64)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&flag, 64));
    CK(hipMemset(big, 1, bigB));
    printf("6144 dependent v_fma_f64 (48 KB of code), us per pass; warm = %.1f us at 8 cycles each, 2.4 GHz\n", 6144 * 8 / 2400.0);
    for (int mode = 0; mode < 4; ++mode) {
        const int loaded = mode & 1, prefetch = mode >> 1;
        if (prefetch && !loaded) printf("-- with the code read as data first (into L2) --\n");
        for (int launch = 0; launch < 4; ++launch) {
            CK(hipMemset(flag, 0, 4));
            k<<<256, 256>>>(big, bigB / 16, loaded, flag, ticks, sink, 0.5, prefetch);
            CK(hipDeviceSynchronize());
            unsigned long long t[2];
            CK(hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost));
            printf("%s launch %d: pass 1 %8.1f us   pass 2 %8.1f us\n", loaded ? "beside a stream" : "idle           ", launch, t[0] * 0.01, t[1] * 0.01);
        }
    }
    return 0;
}
