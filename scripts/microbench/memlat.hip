// memlat.hip -- global-load round trips of ONE wave (a dependent chain of loads), alone and while the other CUs stream a
// panel-like buffer at the HBM ceiling: what the covariance / mover waves of the one-launch pass pay for their parameter
// and table loads next to the DMA stream.  Boxes of this pool that give the headline 0.43 instead of 0.57 of the HBM peak
// have identical clocks, idle chain latencies and streaming ceilings: this probe looks at the remaining suspect, the
// latency of scattered loads under load (page size / TLB reach of the allocation).
//   chase S   : 1024 dependent loads, consecutive addresses S bytes apart (wrapping in a 1 GiB buffer)
//   loaded    : the same while 255 workgroups stream 4 GiB with 16-byte loads
// Build: hipcc --offload-arch=gfx950 -O3 memlat.hip -o memlat ; run: ./memlat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int HOPS = 1024;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now_rt() {       // constant 100 MHz counter
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// workgroup 0: the chase (one wave).  The others: stream `big` (16 bytes per lane per load, 8 loads in flight) until the
// chase has finished (flag), or not at all (nstream == 0)
__global__ __launch_bounds__(256) void k(const unsigned long long* chase, unsigned long long start, const u4* big, size_t nbig16,
                                         int stream, unsigned* flag, unsigned long long* ticks, unsigned long long* sink,
                                         unsigned long long* streamed) {
    if (blockIdx.x == 0) {
        if (threadIdx.x >= 64) return;
        unsigned long long p = start;
        const unsigned long long t0 = now_rt();
        for (int i = 0; i < HOPS; ++i) p = __builtin_nontemporal_load(chase + p);
        const unsigned long long t1 = now_rt();
        if (threadIdx.x == 0) {
            ticks[0] = t1 - t0;
            sink[0] = p;
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
    if (threadIdx.x == 0) atomicAdd(streamed, n * 256 * 16);
}

int main() {
    const size_t chaseB = 1ull << 30, bigB = 4ull << 30;
    unsigned long long *chase, *ticks, *sink, *streamed;
    u4* big;
    unsigned* flag;
    CK(hipMalloc(&chase, chaseB));
    CK(hipMalloc(&big, bigB));
    CK(hipMalloc(&ticks, 64)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&flag, 64)); CK(hipMalloc(&streamed, 64));
    CK(hipMemset(big, 1, bigB));
    const size_t n = chaseB / 8;
    unsigned long long* h = (unsigned long long*)malloc(chaseB);
    const size_t strides[] = {64, 4096 + 64, 65536 + 64, (2u << 20) + 64, (32u << 20) + 64};
    printf("%-12s %14s %14s %12s\n", "stride", "idle ns/load", "loaded ns/load", "stream GB/s");
    for (size_t S : strides) {
        const size_t s8 = S / 8;
        // only the visited slots need values: HOPS x 4 of them (4 runs continue one after the other)
        size_t p = 0;
        for (int i = 0; i < HOPS * 8; ++i) { const size_t q = (p + s8) % n; h[p] = q; p = q; }
        // copy only what was written (sparse): simpler to copy everything once per stride for small HOPS -> copy slots
        p = 0;
        for (int i = 0; i < HOPS * 8; ++i) { CK(hipMemcpy(chase + p, h + p, 8, hipMemcpyHostToDevice)); p = h[p]; }
        double res[2], gbs = 0;
        for (int loaded = 0; loaded < 2; ++loaded) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(flag, 0, 4)); CK(hipMemset(streamed, 0, 8));
                // start each repetition at a different place of the chain: no line is revisited within the L2's lifetime
                size_t st = 0;
                for (int i = 0; i < HOPS * (loaded * 3 + rep); ++i) st = h[st];
                k<<<256, 256>>>(chase, st, big, bigB / 16, loaded, flag, ticks, sink, streamed);
                CK(hipDeviceSynchronize());
                unsigned long long t, sb;
                CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(&sb, streamed, 8, hipMemcpyDeviceToHost));
                const double ns = t * 10.0 / HOPS;
                if (ns < best) { best = ns; if (loaded) gbs = sb / (t * 10.0); }
            }
            res[loaded] = best;
        }
        printf("%-12zu %14.0f %14.0f %12.0f\n", S, res[0], res[1], gbs);
    }
    return 0;
}
