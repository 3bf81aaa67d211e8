// segbw.hip -- what does the LENGTH of the contiguous piece of a panel row cost an LDS-DMA stream?
// collapse_wide2_kernel (config 4: rows of N = 1000 doubles = 8000 bytes) brings a tile of 128 periods in as 32 stages of
// 128 row pieces x 256 bytes: every DRAM page of the tile is visited 8 times, a stage apart.  It streams at ~4.5 TB/s; the
// one-launch pass at C2 (whole rows of 1600 bytes, the replicate's panel one contiguous run) at ~6.  This program reads the same
// 4.1 GB with the same bytes in flight per CU (2 stages x 32 KB of 3 buffers, counted waits, one barrier per stage, 4 DMA waves per
// workgroup, one workgroup per CU, a replicate each) and only the stage SHAPE differs: rows x piece = 128 x 256 B (the kernel's),
// 64 x 512, 32 x 1024, 16 x 2048 (rows of the tile swept piece by piece), or `rows 4 x 8000` = whole rows in order.
// Build: hipcc --offload-arch=gfx950 -O3 segbw.hip -o segbw ; run: ./segbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using lds_ptr = __attribute__((address_space(3))) char*;

template <bool NT>
__device__ __forceinline__ void dma16(const void* g, unsigned dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}

constexpr int kStageB = 32768;   // bytes per stage: 8 DMAs of 1 KB for each of the 4 waves
// seg = bytes of a row piece (a power of two <= 2048 dividing the stage, or 0 = whole rows in order); rowB = bytes of a panel row;
// T rows per replicate; workgroup g reads replicate g, g + gridDim.x, ...
template <bool NT>
__global__ __launch_bounds__(256) void k(const char* __restrict__ base, int B, int T, unsigned rowB, unsigned seg, double* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)smem);
    const size_t repB = (size_t)T * rowB;
    // stage list of a replicate: seg > 0: tiles of R = kStageB / seg rows, each swept in ceil(rowB / seg) stages; seg == 0: the
    // replicate's bytes in order, kStageB at a time
    const unsigned rows = seg ? kStageB / seg : 0;
    const unsigned nch = seg ? (rowB + seg - 1) / seg : 0;
    const unsigned nstage = seg ? ((T + rows - 1) / rows) * nch : (unsigned)((repB + kStageB - 1) / kStageB);
    auto issue = [&](int b, unsigned q, int bsel) {
        const char* Xb = base + (size_t)b * repB;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned o = (unsigned)(wave * 8 + i) * 1024u + 16u * lane;       // byte of the stage
            size_t src;
            if (seg) {
                const unsigned tile = q / nch, ch = q % nch;
                unsigned row = tile * rows + o / seg, col = ch * seg + o % seg;
                row = row < (unsigned)T ? row : T - 1;
                col = col < rowB ? col : rowB - 16;
                src = (size_t)row * rowB + col;
            } else {
                src = (size_t)q * kStageB + o;
                src = src < repB ? src : repB - 16;
            }
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)bsel * kStageB + (unsigned)(wave * 8 + i) * 1024u);
            dma16<NT>(Xb + src, dst);
        }
    };
    double acc = 0.0;
    // the stream runs across replicate boundaries: a flat stage counter over this workgroup's replicates
    const int nrep = (B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const unsigned long long total = (unsigned long long)nrep * nstage;
    auto issue_flat = [&](unsigned long long f, int bsel) {
        const int b = (int)blockIdx.x + (int)(f / nstage) * (int)gridDim.x;
        issue(b, (unsigned)(f % nstage), bsel);
    };
    if (total > 0) issue_flat(0, 0);
    if (total > 1) issue_flat(1, 1);
    int bsel = 0;
    for (unsigned long long f = 0; f < total; ++f) {
        if (f + 1 < total) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *reinterpret_cast<const double*>(smem + bsel * kStageB + 8 * threadIdx.x);   // (a consumer touches the stage)
        if (f + 2 < total) issue_flat(f + 2, bsel == 0 ? 2 : bsel - 1);
        bsel = bsel == 2 ? 0 : bsel + 1;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, T = argc > 2 ? atoi(argv[2]) : 2000, N = argc > 3 ? atoi(argv[3]) : 1000;
    const unsigned rowB = 8u * N;
    const size_t bytes = (size_t)B * T * rowB;
    char* a = nullptr;
    double* sink = nullptr;
    CK(hipMalloc(reinterpret_cast<void**>(&a), bytes + 4096));
    CK(hipMalloc(reinterpret_cast<void**>(&sink), 64));
    CK(hipMemset(a, 0, bytes + 4096));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * kStageB));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * kStageB));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned segs[] = {256, 512, 1024, 2048, 0};
    printf("B %d T %d N %d: %.2f GB, one workgroup of 4 DMA waves per CU (256), 64 KB in flight per CU\n", B, T, N, bytes / 1e9);
    for (int rep = 0; rep < 2; ++rep)
        for (int nt = 0; nt < 2; ++nt)
            for (unsigned seg : segs) {
                auto launch = [&]() {
                    if (nt) hipLaunchKernelGGL(k<true>, dim3(256), dim3(256), 3 * kStageB, 0, a, B, T, rowB, seg, sink);
                    else hipLaunchKernelGGL(k<false>, dim3(256), dim3(256), 3 * kStageB, 0, a, B, T, rowB, seg, sink);
                };
                launch();
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 5; ++i) launch();
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipGetLastError());
                if (seg) printf("  stage %4u rows x %4u B%s: %7.3f ms  %6.0f GB/s\n", kStageB / seg, seg, nt ? " nt" : "   ", ms / 5, bytes / (ms / 5 * 1e-3) / 1e9);
                else printf("  whole rows in order   %s: %7.3f ms  %6.0f GB/s\n", nt ? " nt" : "   ", ms / 5, bytes / (ms / 5 * 1e-3) / 1e9);
            }
    return 0;
}
