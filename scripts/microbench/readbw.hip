// readbw.hip -- read-bandwidth ceilings on MI355X for the collapse kernel's access pattern.
//   k_plain : grid-stride 16-B loads, register sum                      (classic streaming read)
//   k_dma   : per-wave LDS-DMA ring over a private contiguous segment   (collapse_dma's pattern)
// Build: hipcc --offload-arch=gfx950 -O3 readbw.hip -o readbw ; run: ./readbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_plain(const double2* __restrict__ p, size_t n2, double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0.0;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        const double2 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        s += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
    for (; i < n2; i += stride) { const double2 a = p[i]; s += a.x + a.y; }
    if (s == 1.2345e300) out[0] = s;
}

// each wave streams `seg` bytes (multiple of 1024) starting at base + wave_global * seg
template <int NSLOT, int LANES2>   // NSLOT 1-KiB pieces in flight per wave; LANES2: second partial piece lanes (0 = none)
__global__ __launch_bounds__(256) void k_dma(const char* __restrict__ base, size_t seg, double* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t wg = (size_t)blockIdx.x * 4 + wave;
    const char* src = base + wg * seg + 16 * lane;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem) + wave * NSLOT * 1024;
    const int npiece = (int)(seg / 1024);
    auto issue = [&](int piece, int slot) {
        int pc = piece < npiece ? piece : npiece - 1;
        const char* g = src + (size_t)pc * 1024;
        unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + slot * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    };
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) issue(s, s);
    double acc = 0.0;
    int slot = 0;
    constexpr int HALF = NSLOT / 2;
    for (int p0 = 0; p0 < npiece; p0 += HALF) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NSLOT - HALF) : "memory");
        double2 v[HALF];
#pragma unroll
        for (int u = 0; u < HALF; ++u) v[u] = *reinterpret_cast<const double2*>(smem + (wave * NSLOT + slot + u) * 1024 + 16 * lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < HALF; ++u) issue(p0 + NSLOT + u, slot + u);
#pragma unroll
        for (int u = 0; u < HALF; ++u) acc += v[u].x + v[u].y;
        slot += HALF;
        if (slot == NSLOT) slot = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 1.2345e300) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1024 * 500 * 200 * 8;   // the config-2 panel: 819.2 MB
    char* buf; double* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        const int K = 20;
        for (int i = 0; i < K; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-28s %8.4f ms  %7.1f GB/s\n", name, ms / K, bytes / (ms / K * 1e-3) / 1e9);
    };
    for (int blocks : {1024, 2048, 4096, 8192}) {
        char nm[64]; snprintf(nm, sizeof nm, "plain grid=%d", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL(k_plain, dim3(blocks), dim3(256), 0, 0, (const double2*)buf, bytes / 16, out); });
    }
    // DMA: 4096 waves (1024 blocks) each 200000 B -> use seg = 199680 (195 KiB) to stay 1-KiB aligned
    {
        const size_t seg = 199680;
        timeit("dma ring 8 KiB/wave  b=1024", [&] { hipLaunchKernelGGL((k_dma<8, 0>), dim3(1024), dim3(256), 4 * 8 * 1024, 0, buf, seg, out); });
        timeit("dma ring 16 KiB/wave b=1024", [&] { hipLaunchKernelGGL((k_dma<16, 0>), dim3(1024), dim3(256), 4 * 16 * 1024, 0, buf, seg, out); });
        CK(hipFuncSetAttribute((const void*)&k_dma<32, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        timeit("dma ring 32 KiB/wave b=1024", [&] { hipLaunchKernelGGL((k_dma<32, 0>), dim3(1024), dim3(256), 4 * 32 * 1024, 0, buf, seg, out); });
        const size_t seg2 = 99328;   // 2048 blocks x 4 waves x 97 KiB
        timeit("dma ring 8 KiB/wave  b=2048", [&] { hipLaunchKernelGGL((k_dma<8, 0>), dim3(2048), dim3(256), 4 * 8 * 1024, 0, buf, seg2, out); });
        timeit("dma ring 16 KiB/wave b=2048", [&] { hipLaunchKernelGGL((k_dma<16, 0>), dim3(2048), dim3(256), 4 * 16 * 1024, 0, buf, seg2, out); });
    }
    return 0;
}
