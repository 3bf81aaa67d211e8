// gram_xx_wide.hip -- S = X'X of the PCA start for wide cross-sections (even N > 256; BASELINE config 4: N = 1000, T = 2000)
// on the f64 matrix pipe.
//
// gram_xx_mfma_kernel (pca.hip) keeps a replicate's whole triangle of 16 x 16 tiles in the registers of one workgroup: N <= 256.
// Beyond that the VALU kernel needed 160 ms for the 256 replicates of config 4 (1 TFLOP at 6 TFLOP/s).  Here an item is
// (replicate, pair of 128-series blocks bi <= bj): one workgroup streams the two column blocks of the panel through LDS in
// stages of 16 periods (32 row segments of 1 KB by `global_load_lds_dwordx4`, issued by four PRODUCER waves with a counted
// wait, three stage buffers) while eight CONSUMER waves (wave w = series 16 w .. 16 w + 15 of block bi) accumulate their
// 16 x 128 strip of the block: per step of 4 periods one 8-byte LDS read of A, eight of B, eight `v_mfma_f64_16x16x4`.
// Rows of a stage lie 1024 + 128 bytes apart (slot of (period k, series i) = 16 k + i mod 32: conflict-free b64 reads).
// The block and, off the diagonal, its mirror image are stored at the end: pca_kernel reads S as a full matrix.
// Reference counterpart: the svd of the standardised panel in pca_score (dfm_functions.ipynb:179-183), here through X'X.
#include <stdlib.h>

#include "dfm_kernels.h"

namespace dfm {

namespace {

using lds_char_ptr_gx = __attribute__((address_space(3))) char*;
using lds_cvd_ptr_gx = const volatile __attribute__((address_space(3))) double*;
__device__ __forceinline__ double lds_read64g(unsigned a) { return *(lds_cvd_ptr_gx)(size_t)a; }
typedef double gx_v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16gx(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

constexpr int kGxSer = 128;                               // series per block
constexpr int kGxPer = 16;                                // periods per stage
constexpr int kGxSteps = kGxPer / 4;
constexpr int kGxNBuf = 3;
constexpr unsigned kGxRowB = 1152;                        // LDS bytes between the rows of a stage (1024 + 128)
constexpr unsigned kGxBlockB = kGxPer * kGxRowB;          // one block's rows of a stage: 18432
constexpr unsigned kGxStageB = 2 * kGxBlockB;             // A rows | B rows
constexpr int kGxCompute = kGxSer / 16, kGxProducers = 4;
constexpr int kGxThreads = 64 * (kGxCompute + kGxProducers);
constexpr int kGxPerStage = 2 * kGxPer / kGxProducers;    // DMAs per producer and stage: 8

}  // namespace

__global__ __launch_bounds__(kGxThreads) void gram_xx_wide_kernel(PcaArgs a, int nb, int npair) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, T = a.T;
    const int nst = (T + kGxPer - 1) / kGxPer;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = (int)blockIdx.x / npair;
    int p = (int)blockIdx.x % npair, bi = 0;
    while (p >= nb - bi) { p -= nb - bi; ++bi; }              // pair p -> (bi, bj), bi <= bj
    const int bj = bi + p;
    const int si = bi * kGxSer, sj = bj * kGxSer;
    const double* __restrict__ X = a.panel + (size_t)b * T * N;

    // zero once: the lanes of a partial block never write their columns
    for (int e = tid; e < kGxNBuf * (int)kGxStageB / 8; e += kGxThreads) reinterpret_cast<double*>(smem)[e] = 0.0;
    __syncthreads();

    if (wave >= kGxCompute) {
        // ---- producers: rows 4 p .. 4 p + 3 of both blocks of a stage, one DMA per row (lane l = series 2 l, 2 l + 1 of the block)
        const int pw = wave - kGxCompute;
        __builtin_amdgcn_s_setprio(3);
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_gx)(smem));
        const bool acti = si + 2 * lane < N, actj = sj + 2 * lane < N;   // (lane 0 is active in both: every DMA is issued)
        auto issue = [&](int st, int bsel) {
            const unsigned sbase = lds0 + (unsigned)bsel * kGxStageB;
#pragma unroll
            for (int k = 0; k < kGxPer / kGxProducers; ++k) {
                const int row = (kGxPer / kGxProducers) * pw + k;
                int t = st * kGxPer + row;
                t = t < T ? t : T - 1;
                const char* rowp = reinterpret_cast<const char*>(X + (size_t)t * N);
                const unsigned dsti = __builtin_amdgcn_readfirstlane(sbase + (unsigned)row * kGxRowB);
                const unsigned dstj = __builtin_amdgcn_readfirstlane(sbase + kGxBlockB + (unsigned)row * kGxRowB);
                if (acti) dma16gx(rowp + (size_t)(si + 2 * lane) * 8, dsti);
                if (actj) dma16gx(rowp + (size_t)(sj + 2 * lane) * 8, dstj);
            }
        };
        issue(0, 0);
        if (nst > 1) issue(1, 1);
        int bsel = 0;
        for (int q = 0; q < nst; ++q) {
            if (q + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kGxPerStage) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                  // stage q has landed; every consumer is done with stage q - 1
            if (q + 2 < nst) issue(q + 2, bsel == 0 ? 2 : bsel - 1);
            bsel = bsel == 2 ? 0 : bsel + 1;
        }
        return;
    }

    // ---- consumers: wave w = rows 16 w .. 16 w + 15 of block bi, all 8 column tiles of block bj
    const int k4 = lane >> 4, c16 = lane & 15;
    const unsigned ldsc = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_gx)(smem));
    const unsigned a_off = (unsigned)k4 * kGxRowB + (unsigned)(16 * wave + c16) * 8u;
    const unsigned b_off = kGxBlockB + (unsigned)k4 * kGxRowB + (unsigned)c16 * 8u;
    gx_v4 acc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] = gx_v4{0.0, 0.0, 0.0, 0.0};
    int bsel = 0;
    for (int q = 0; q < nst; ++q) {
        __syncthreads();
        const unsigned stg = ldsc + (unsigned)bsel * kGxStageB;
        const int tfirst = q * kGxPer + k4;
        double av[kGxSteps], bv[kGxSteps][8];
#pragma unroll
        for (int s = 0; s < kGxSteps; ++s) {
            av[s] = lds_read64g(stg + a_off + (unsigned)s * (4 * kGxRowB));
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) bv[s][ct] = lds_read64g(stg + b_off + (unsigned)s * (4 * kGxRowB) + (unsigned)ct * 128u);
        }
#pragma unroll
        for (int s = 0; s < kGxSteps; ++s) {
            const double a_ = tfirst + 4 * s < T ? av[s] : 0.0;   // (the last stage repeats row T - 1)
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, bv[s][ct], acc[ct], 0, 0, 0);
        }
        bsel = bsel == 2 ? 0 : bsel + 1;
    }
    double* S = a.S + (size_t)b * N * N;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {                         // D[(l / 16) + 4 v][l % 16]
            const int gi = si + 16 * wave + k4 + 4 * v, gj = sj + 16 * ct + c16;
            if (gi < N && gj < N) {
                S[(size_t)gi * N + gj] = acc[ct][v];
                if (bi != bj) S[(size_t)gj * N + gi] = acc[ct][v];
            }
        }
    }
}

bool gram_xx_wide_supported(int N) { return (N & 1) == 0 && N > 256; }
hipError_t launch_gram_xx_wide(const PcaArgs& a, hipStream_t s) {
    note_kernel("gram_xx_wide_kernel");
    const int nb = (a.N + kGxSer - 1) / kGxSer, npair = nb * (nb + 1) / 2;
    const size_t lds = (size_t)kGxNBuf * kGxStageB;
    static LdsOptIn attr_done;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_xx_wide_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(gram_xx_wide_kernel, dim3((unsigned)((long long)a.B * npair)), dim3(kGxThreads), lds, s, a, nb, npair);
    return hipGetLastError();
}

}  // namespace dfm
