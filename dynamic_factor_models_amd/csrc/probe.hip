// probe.hip -- dfm_hbm_probe: the streaming ceilings of THIS device, measured with the access patterns the pass uses, so
// that bench.py can put a same-box number beside the 8 TB/s spec peak ("plus a measured stream-copy
// ceiling on the same box"; boxes of the pool differ by up to 20 % on latency-bound code and a reader cannot attribute a
// roofline fraction without it).  No reference counterpart; measurement infrastructure, not part of the estimator.
//   mode 0  read-only, per-wave LDS-DMA ring (global_load_lds_dwordx4, 16 x 1 KiB in flight per wave): the collapse's pattern
//   mode 1  copy: 16-byte loads -> 16-byte stores, grid-stride (read + write bytes are both counted)
//   mode 2  write-only: 16-byte stores (the P_smooth fill's pattern)
#include "../../include/dfm_hip.h"

#include <hip/hip_runtime.h>

struct dfm_handle;
namespace dfm { int handle_device(const dfm_handle* h); }   // capi.hip

namespace {

using lds_ptr = __attribute__((address_space(3))) char*;

template <int NSLOT>
__global__ __launch_bounds__(256) void probe_read_kernel(const char* __restrict__ base, size_t seg, double* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t wg = (size_t)blockIdx.x * 4 + wave;
    const char* src = base + wg * seg + 16 * lane;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)smem) + wave * NSLOT * 1024;
    const int npiece = (int)(seg / 1024);
    auto issue = [&](int piece, int slot) {
        const int pc = piece < npiece ? piece : npiece - 1;
        const char* g = src + (size_t)pc * 1024;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + slot * 1024);
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

__global__ __launch_bounds__(256) void probe_copy_kernel(const double2* __restrict__ src, double2* __restrict__ dst, size_t n2) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        const double2 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n2; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void probe_write_kernel(double2* __restrict__ dst, size_t n2, double v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const double2 x = make_double2(v, v + 1.0);
    for (; i < n2; i += stride) dst[i] = x;
}

}  // namespace

extern "C" int dfm_hbm_probe(dfm_handle* h, size_t bytes, int mode, int iters, double* gbs_out, double* ms_out) {
    if (!gbs_out || mode < 0 || mode > 2 || iters < 1) return DFM_E_DIMS;
    if (bytes < (size_t)1 << 24) return DFM_E_DIMS;                     // below 16 MB the number says nothing about HBM
    if (!h) return DFM_E_NULL;
    {   // the HANDLE's device (a Julia / C caller with several handles, or the dfm_multi threads, may have another one current)
        const hipError_t ed = hipSetDevice(dfm::handle_device(h));
        if (ed != hipSuccess) return (int)ed;
    }
    hipStream_t st = nullptr;                                             // (its null stream: the probe runs alone, between timed regions)
    const int blocks = 1024;
    const size_t seg = (bytes / ((size_t)blocks * 4)) / 1024 * 1024;      // per wave, a multiple of 1 KiB
    const size_t used = mode == 0 ? seg * blocks * 4 : bytes / 16 * 16;
    char *a = nullptr, *b = nullptr;
    double* out = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&a), bytes + 1024);
    if (e == hipSuccess && mode == 1) e = hipMalloc(reinterpret_cast<void**>(&b), bytes + 1024);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&out), 64);
    if (e == hipSuccess) e = hipMemsetAsync(a, 0, bytes, st);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL((probe_read_kernel<16>), dim3(blocks), dim3(256), 4 * 16 * 1024, st, a, seg, out);
        else if (mode == 1) hipLaunchKernelGGL(probe_copy_kernel, dim3(4096), dim3(256), 0, st, reinterpret_cast<const double2*>(a),
                                               reinterpret_cast<double2*>(b), used / 16);
        else hipLaunchKernelGGL(probe_write_kernel, dim3(4096), dim3(256), 0, st, reinterpret_cast<double2*>(a), used / 16, 1.0);
    };
    float ms = 0.f;
    if (e == hipSuccess) {
        for (int i = 0; i < 3; ++i) launch();
        e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipEventRecord(e0, st);
        for (int i = 0; i < iters; ++i) launch();
        if (e == hipSuccess) e = hipEventRecord(e1, st);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (out) (void)hipFree(out);
    if (e != hipSuccess) return (int)e;
    const double per = (double)ms / iters;
    const double moved = (mode == 1 ? 2.0 : 1.0) * (double)used;
    *gbs_out = moved / (per * 1e-3) / 1e9;
    if (ms_out) *ms_out = per;
    return 0;
}
