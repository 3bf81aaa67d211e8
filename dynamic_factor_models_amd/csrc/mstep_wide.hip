// mstep_wide.hip -- the loadings half of the M-step for Rp = 32 (and 16) on a balanced panel (BASELINE config 4: N = 1000,
// T = 2000, r = 20) as a second streaming pass over the panel on the f64 matrix pipe.
//
//     Sxf_i = sum_t x_it f_t|T  (32 padded factors)   Sxx_i = sum_t x_it^2      lam_i = S11^-1 Sxf_i,
//     R_i = (Sxx_i - 2 lam_i'Sxf_i + lam_i'S11 lam_i) / T                    (Shumway-Stoffer 1982; S11, S11^-1 from em_update_kernel)
//
// mstep_lam_kernel (mstep.hip: lane = series, the factors of a period re-read by every lane) needs 18.6 ms for the 4.1 GB of
// config 4 -- 0.22 TB/s, nine tenths of an EM iteration.  Here the contraction Sxf = X'F runs like the collapse of
// collapse_wide2.hip, with the roles of series and periods exchanged:
//   * an item = (replicate, block of 128 series); PERSISTENT workgroups (one per CU) take items from a queue per XCD (the 8
//     blocks of a replicate stay on one XCD: its 512 KB of smoothed factors are read from HBM once, then from that L2);
//   * the item streams through LDS in stages of 32 periods: 32 panel row segments of 1 KB and the 32 x 32 block of factors
//     (8 KB), all by `global_load_lds_dwordx4` issued by four PRODUCER waves (a fixed 10 per producer and stage: counted
//     `s_waitcnt vmcnt`), three stage buffers;
//   * eight CONSUMER waves, wave w = series 16 w .. 16 w + 15 of the block: per step of 4 periods one 8-byte LDS read of A
//     (A[i][k] = x[t + k][s0 + 16 w + i]) feeds `v_mfma_f64_16x16x4` (factors 0..15) and one `v_mfma_f64_4x4x4` per further 4
//     factors, and x^2 for Sxx; the accumulators (Sxf of 16 series x 32 factors: 8 doubles per lane) live in registers for
//     the whole item -- no partial sums, no second pass;
//   * LDS layouts free of bank conflicts: panel rows 1024 + 128 bytes apart (slot of (period k, series i) = 16 k + i mod 32),
//     the four factor rows of a step rotated by 128 bytes per odd period.
// mstep_finish_wide_kernel (thread = series) then applies S11^-1.  Rp = 16: one 16-wide factor tile, factor rows of 128 bytes.  Reference counterpart: the regression of x_i on the factors in
// estimate_factor_loading! (dfm_functions.ipynb:386-412), here with the smoothed moments of the EM in place of PCA factors.
#include <stdlib.h>

#include "dfm_gram.h"
#include "dfm_kernels.h"

namespace dfm {

namespace {

using lds_char_ptr_mw = __attribute__((address_space(3))) char*;
using lds_cvd_ptr_mw = const volatile __attribute__((address_space(3))) double*;
__device__ __forceinline__ double lds_read64m(unsigned a) { return *(lds_cvd_ptr_mw)(size_t)a; }
typedef double mw_v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16mw(const void* gsrc, unsigned lds_dst) {
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

// the PANEL rows (development A/B: -DDFM_MW_PANEL_MOD='" nt"'); the factor rows are re-read by the 8 series blocks of a replicate
#ifndef DFM_MW_PANEL_MOD
#define DFM_MW_PANEL_MOD ""
#endif
__device__ __forceinline__ void dma16mwp(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off" DFM_MW_PANEL_MOD "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

constexpr int kMwR = 32;
constexpr int kMwSer = 128;                              // series per item: 8 consumer waves x 16
constexpr int kMwPer = 32;                               // periods per stage
constexpr int kMwSteps = kMwPer / 4;
constexpr int kMwNBuf = 3;
constexpr unsigned kMwRowB = 1152;                       // LDS bytes between the panel rows of a stage (1024 + 128)
constexpr unsigned kMwPanelB = kMwPer * kMwRowB;         // 36864
constexpr unsigned kMwFB = kMwPer * kMwR * 8;            // 8192 (R = 32; R = 16 uses half of it)
constexpr unsigned kMwStageB = kMwPanelB + kMwFB;
constexpr int kMwCompute = kMwSer / 16, kMwProducers = 4;
constexpr int kMwThreads = 64 * (kMwCompute + kMwProducers + 1);
template <int R> struct MwGeo {
    static constexpr int FPieces = (kMwPer * R * 8 / 1024) / kMwProducers;            // 1-KB DMAs of the factor block per producer: 2 | 1
    static constexpr int PerStage = kMwPer / kMwProducers + FPieces;                  // DMAs per producer and stage: 10 | 9
};
constexpr int kMwRing = 8;

}  // namespace

template <int R, int NX>
__global__ __launch_bounds__(kMwThreads) void mstep_wide_kernel(MstepArgs a, double* __restrict__ sxf, double* __restrict__ sxx,
                                                                int* ctr, int nsb, int xcd_map, int rd) {
    // rd = width of the caller's arrays (fsm [T][rd], Sxf [N][rd]); rd < R (= 16) for the narrow states beyond mstep_mfma's ring
    static_assert((R == 32 && NX >= 1 && NX <= 4) || (R == 16 && NX == 0), "R = 32: 1..4 column groups past the first 16; R = 16: none");
    using GEO = MwGeo<R>;
    constexpr int N4 = NX < 4 ? NX : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, T = a.T, B = a.B;
    const int nst = (T + kMwPer - 1) / kMwPer;                // stages per item
    volatile int* itemq = reinterpret_cast<volatile int*>(smem + kMwNBuf * kMwStageB);
    const unsigned itemq_lds = (unsigned)(size_t)(lds_char_ptr_mw)(smem) + kMwNBuf * kMwStageB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = (int)blockIdx.x & 7;
    auto rep_of = [&](int kk) { return xcd_map ? (kk / nsb) * 8 + xcd : kk / nsb; };
    auto read_item = [&](int idx) {
        return __builtin_amdgcn_readfirstlane(*(const volatile __attribute__((address_space(3))) int*)(size_t)(itemq_lds + 4u * (unsigned)(idx & (kMwRing - 1))));
    };

    // Every role passes the SAME barriers: one after the set-up, one per stage that exists, one at the end.
    if (wave == kMwCompute + kMwProducers) {
        // ---- scheduler wave (as in collapse_wide2_kernel): item n is published before the barrier after which a producer can
        // first ask for it, half an item early
        int fetched = 0, nvalid = 0;
        bool ended = false;
        auto publish_upto = [&](int target) {
            while (fetched <= target && !ended) {
                int kk = 0;
                if (lane == 0) kk = atomicAdd(&ctr[xcd_map ? xcd : 0], 1);
                kk = __builtin_amdgcn_readfirstlane(kk);
                const bool ok = rep_of(kk) < B;
                if (lane == 0) itemq[fetched & (kMwRing - 1)] = ok ? kk : -1;
                ++fetched;
                if (ok) ++nvalid; else ended = true;
            }
        };
        publish_upto((3 + nst / 2) / nst);
        __syncthreads();
        for (int q = 0; q / nst < nvalid; ++q) {
            publish_upto((q + 3 + nst / 2) / nst);
            __syncthreads();
        }
        __syncthreads();
        return;
    }

    // no NaN bit patterns in the columns a partial block never writes: zero the buffers once
    for (int e = tid; e < kMwNBuf * (int)kMwStageB / 8; e += 64 * (kMwCompute + kMwProducers)) reinterpret_cast<double*>(smem)[e] = 0.0;
    __syncthreads();

    if (wave >= kMwCompute) {
        // ---- producer waves: all the LDS-DMA.  Producer p: panel rows (periods) 8 p .. 8 p + 7 of the stage, one DMA each (lane l
        // = series 2 l, 2 l + 1 of the block), and factor pieces 2 p, 2 p + 1 (a piece = the 4 periods of a step, 256 bytes each;
        // row h of the piece rotated by 8 h units of 16 bytes).  Lane 0 of a row DMA is always active: 10 DMAs per stage, always.
        const int pw = wave - kMwCompute;
        __builtin_amdgcn_s_setprio(3);
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_mw)(smem));
        auto issue_dma = [&](int b, int sb, int st, int bsel) {
            const char* Xb = reinterpret_cast<const char*>(a.panel + (size_t)b * T * N);
            const char* Fb = reinterpret_cast<const char*>(a.fsm + (size_t)b * T * rd);
            const unsigned sbase = lds0 + (unsigned)bsel * kMwStageB;
            const int t0 = st * kMwPer;
            const int ser = sb * kMwSer + 2 * lane;
            const bool act = ser < N;
#pragma unroll
            for (int k = 0; k < kMwPer / kMwProducers; ++k) {
                const int row = (kMwPer / kMwProducers) * pw + k;
                int t = t0 + row;
                t = t < T ? t : T - 1;
                const char* src = Xb + ((size_t)t * N + ser) * 8;
                const unsigned dst = __builtin_amdgcn_readfirstlane(sbase + (unsigned)row * kMwRowB);
                if (act) dma16mwp(src, dst);
            }
#pragma unroll
            for (int u = 0; u < GEO::FPieces; ++u) {
                const int piece = GEO::FPieces * pw + u;
                const char* src;
                if (R == 32) {                                // a piece = the 4 periods of a step, row h rotated by 8 h units
                    const int h = lane >> 4;
                    int t = t0 + 4 * piece + h;
                    t = t < T ? t : T - 1;
                    src = Fb + (size_t)t * (R * 8) + 16u * (unsigned)(((lane & 15) - 8 * h) & 15);
                } else {                                      // R = 16: the stage's factor rows (rd x 8 bytes each) as they lie, 1 KB per piece;
                    const size_t rowb = (size_t)rd * 8;       // past the block / the sample: the last 16 bytes again (those periods meet a zeroed A)
                    const size_t blk = (size_t)kMwPer * rowb, lim = (size_t)T * rowb - 16;
                    size_t o = (size_t)piece * 1024u + 16u * (unsigned)lane;
                    o = o < blk ? o : blk - 16;
                    o += (size_t)t0 * rowb;
                    src = Fb + (o < lim ? o : lim);
                }
                const unsigned dst = __builtin_amdgcn_readfirstlane(sbase + kMwPanelB + (unsigned)piece * 1024u);
                dma16mw(src, dst);
            }
        };
        int ii = 0, ist = 0, ikk = read_item(0);
        auto issue_next = [&](int bsel) {
            issue_dma(rep_of(ikk), ikk % nsb, ist, bsel);
            if (++ist == nst) { ist = 0; ikk = read_item(++ii); }
        };
        bool more = ikk >= 0, v1 = false;
        if (more) {
            issue_next(0);
            v1 = ikk >= 0;
            if (v1) issue_next(1);
        }
        int bsel = 0;
        while (more) {
            if (!v1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GEO::PerStage) : "memory");
            __syncthreads();
            const bool v2 = ikk >= 0;
            if (v2) issue_next(bsel == 0 ? 2 : bsel - 1);
            bsel = bsel == 2 ? 0 : bsel + 1;
            more = v1;
            v1 = v2;
        }
        __syncthreads();
        return;
    }

    // ---- consumer waves
    const int k4 = lane >> 4, c16 = lane & 15;
    const unsigned ldsc = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_mw)(smem));
    int ci = 0, st = 0, bsel = 0, ckk = read_item(0);
    bool more = ckk >= 0;
    mw_v4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0}, accb = {0.0, 0.0, 0.0, 0.0};
    double acc4[N4 > 0 ? N4 : 1];
#pragma unroll
    for (int x = 0; x < (N4 > 0 ? N4 : 1); ++x) acc4[x] = 0.0;
    double qs = 0.0;
    // A operand: period 4 s + k4 of the stage, series 16 w + c16 of the block
    const unsigned a_off = (unsigned)k4 * kMwRowB + (unsigned)(16 * wave + c16) * 8u;
    // B operands: factors of period 4 s + k4 (row k4 of piece s, rotated by 128 k4 bytes)
    // (R = 16: rows of 128 bytes as they lie -- slot 16 k4 + c16, no rotation needed)
    const unsigned f_row = kMwPanelB + (unsigned)k4 * (unsigned)(rd * 8);
    const unsigned b16 = R == 32 ? f_row + ((8u * c16 + 128u * k4) & 255u) : f_row + 8u * (c16 < rd ? c16 : 0);
    const bool col_ok = R == 32 || c16 < rd;
    const unsigned b16b = f_row + ((8u * (16 + c16) + 128u * k4) & 255u);
    unsigned b4[N4 > 0 ? N4 : 1];
#pragma unroll
    for (int x = 0; x < (N4 > 0 ? N4 : 1); ++x) b4[x] = f_row + ((8u * (16 + 4 * x + (lane & 3)) + 128u * k4) & 255u);

    while (more) {
        __syncthreads();                                      // stage q is ready (the producers waited for it)
        const int b = rep_of(ckk), sb = ckk % nsb;
        const int s0 = sb * kMwSer;
        const bool ser_ok = s0 + 16 * wave + c16 < N;
        const unsigned stg = ldsc + (unsigned)bsel * kMwStageB;
        const int tfirst = st * kMwPer + k4;                  // period of this lane's k in step 0
        {
            double av[kMwSteps], bv[kMwSteps], bvb[NX == 4 ? kMwSteps : 1], b4v[N4 > 0 ? N4 : 1][kMwSteps];
            auto load_step = [&](int s) {
                av[s] = lds_read64m(stg + a_off + (unsigned)s * (4 * kMwRowB));
                bv[s] = lds_read64m(stg + b16 + (unsigned)s * (4u * (unsigned)(rd * 8)));
                if (NX == 4) bvb[NX == 4 ? s : 0] = lds_read64m(stg + b16b + (unsigned)s * 1024u);
#pragma unroll
                for (int x = 0; x < N4; ++x) b4v[x][s] = lds_read64m(stg + b4[x] + (unsigned)s * 1024u);
            };
#pragma unroll
            for (int s = 0; s < 4; ++s) load_step(s);
#pragma unroll
            for (int s = 0; s < kMwSteps; ++s) {
                if (s + 4 < kMwSteps) load_step(s + 4);
                const double a_ = (ser_ok && tfirst + 4 * s < T) ? av[s] : 0.0;   // partial block / partial last stage: nothing
                const double b_ = col_ok ? bv[s] : 0.0;       // (narrow states: columns past rd of the 16-wide tile)
                if (s & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b_, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b_, acc0, 0, 0, 0);
                if (NX == 4) accb = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, bvb[NX == 4 ? s : 0], accb, 0, 0, 0);
#pragma unroll
                for (int x = 0; x < N4; ++x) acc4[x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_, b4v[x][s], acc4[x], 0, 0, 0);
                qs = fma(a_, a_, qs);
            }
        }
        bsel = bsel == 2 ? 0 : bsel + 1;
        if (++st == nst) {                                    // the item is complete: Sxf rows and Sxx of its series
            const mw_v4 accs = acc0 + acc1;
            double* out = sxf + ((size_t)b * N + s0 + 16 * wave) * rd;
#pragma unroll
            for (int v = 0; v < 4; ++v) {                     // 16x16x4: D[(l / 16) + 4 v][l % 16] -> series k4 + 4 v, factor c16
                const int row = k4 + 4 * v;
                if (s0 + 16 * wave + row < N) {
                    if (col_ok) out[(size_t)row * rd + c16] = accs[v];
                    if (NX == 4) out[(size_t)row * R + 16 + c16] = accb[v];
                }
            }
            if (R == 32 && NX < 4) {                          // 4x4x4: block (l / 4) % 4, D[row = l / 16][col = l % 4] -> series 4 blk + l / 16
                const int row = 4 * ((lane >> 2) & 3) + k4;
                if (s0 + 16 * wave + row < N) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) out[(size_t)row * R + 16 + 4 * x + (lane & 3)] = x < N4 ? acc4[x < N4 ? x : 0] : 0.0;
                }
            }
            qs += __shfl_xor(qs, 16, 64);
            qs += __shfl_xor(qs, 32, 64);
            if (k4 == 0 && ser_ok) sxx[(size_t)b * N + s0 + 16 * wave + c16] = qs;
            acc0 = mw_v4{0.0, 0.0, 0.0, 0.0}; acc1 = acc0; accb = acc0;
#pragma unroll
            for (int x = 0; x < (N4 > 0 ? N4 : 1); ++x) acc4[x] = 0.0;
            qs = 0.0;
            st = 0;
            ckk = read_item(++ci);
            more = ckk >= 0;
        }
    }
    __syncthreads();
}

// thread = series: lam_i = S11^-1 Sxf_i, R_i = (Sxx_i - 2 lam_i'Sxf_i + lam_i'S11 lam_i) / T
template <int R>
__global__ __launch_bounds__(256) void mstep_finish_wide_kernel(MstepArgs a, const double* __restrict__ sxf, const double* __restrict__ sxx) {
    __shared__ double s11[R * R], s11i[R * R];
    const int b = blockIdx.x;
    if (a.active && !a.active[b]) return;
    const int N = a.N;
    for (int e = threadIdx.x; e < R * R; e += 256) {
        s11[e] = a.S11[(size_t)b * R * R + e];
        s11i[e] = a.S11inv[(size_t)b * R * R + e];
    }
    __syncthreads();
    const int col = (int)blockIdx.y * 256 + (int)threadIdx.x;
    if (col >= N) return;
    double f[R], lam[R];
    const double* ps = sxf + ((size_t)b * N + col) * R;
#pragma unroll
    for (int k = 0; k < R; ++k) f[k] = ps[k];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) s = fma(s11i[i * R + k], f[k], s);
        lam[i] = s;
    }
    double quad = 0.0, cross = 0.0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) s = fma(s11[i * R + k], lam[k], s);
        quad = fma(lam[i], s, quad);
        cross = fma(lam[i], f[i], cross);
    }
    a.R_out[(size_t)b * N + col] = (sxx[(size_t)b * N + col] - 2.0 * cross + quad) / (double)a.T;
    double* lo = a.Lam_out + ((size_t)b * N + col) * a.lam_stride;
#pragma unroll
    for (int k = 0; k < R; ++k) lo[k] = lam[k];
}

// Rp = 16 | 32 with an even N; narrower states (computed 16 wide) where mstep_mfma's 4-KB row ring ends
bool mstep_wide_supported(int Rpad, int N) {
    if ((N & 1) != 0 || N < 2) return false;
    return Rpad == 32 || Rpad == 16 || (Rpad >= 2 && Rpad <= 8 && N * 8 > 4096);
}
// Sxf [B][N][Rp] | Sxx [B][N] | 8 queue counters
size_t mstep_wide_workspace(int B, int N, int Rpad) { return ((size_t)B * N * Rpad + (size_t)B * N) * sizeof(double) + 64; }

namespace {
template <int R, int NX>
hipError_t launch_mw(const MstepArgs& a, double* sxf, double* sxx, int* ctr, int G, size_t lds, int nsb, int xcd_map, hipStream_t s, int rd = R) {
    static LdsOptIn attr_done;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mstep_wide_kernel<R, NX>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((mstep_wide_kernel<R, NX>), dim3((unsigned)G), dim3(kMwThreads), lds, s, a, sxf, sxx, ctr, nsb, xcd_map, rd);
    return hipGetLastError();
}
}  // namespace

// r = the caller's factor count (columns r .. Rpad - 1 of the smoothed factors are zero padding)
hipError_t launch_mstep_wide(const MstepArgs& a, double* ws, int Rpad, int r, int num_cu, hipStream_t s) {
    note_kernel("mstep_wide_kernel");
    double* sxf = ws;
    double* sxx = ws + (size_t)a.B * a.N * Rpad;
    int* ctr = reinterpret_cast<int*>(sxx + (size_t)a.B * a.N);
    hipError_t e = hipMemsetAsync(ctr, 0, 32, s);
    if (e != hipSuccess) return e;
    const int nsb = (a.N + kMwSer - 1) / kMwSer;
    const size_t lds = (size_t)kMwNBuf * kMwStageB + kMwRing * sizeof(int) + 32;
    const int xcd_map = a.B >= 16;
    const long long NI = (long long)a.B * nsb;
    int G = num_cu > 0 ? num_cu : 256;
    G = (G / 8) * 8;
    if (G < 8) G = 8;
    if (!xcd_map && NI < G) G = (int)NI;
    if (Rpad <= 16) {
        e = launch_mw<16, 0>(a, sxf, sxx, ctr, G, lds, nsb, xcd_map, s, Rpad);
        if (e != hipSuccess) return e;
        const dim3 fg(a.B, (a.N + 255) / 256);
        switch (Rpad) {
            case 2: hipLaunchKernelGGL(mstep_finish_wide_kernel<2>, fg, dim3(256), 0, s, a, (const double*)sxf, (const double*)sxx); break;
            case 4: hipLaunchKernelGGL(mstep_finish_wide_kernel<4>, fg, dim3(256), 0, s, a, (const double*)sxf, (const double*)sxx); break;
            case 8: hipLaunchKernelGGL(mstep_finish_wide_kernel<8>, fg, dim3(256), 0, s, a, (const double*)sxf, (const double*)sxx); break;
            default: hipLaunchKernelGGL(mstep_finish_wide_kernel<16>, fg, dim3(256), 0, s, a, (const double*)sxf, (const double*)sxx); break;
        }
        return hipGetLastError();
    }
    const int nx = r <= 16 ? 1 : (r + 3 - 16) / 4;
    switch (nx) {
        case 1: e = launch_mw<32, 1>(a, sxf, sxx, ctr, G, lds, nsb, xcd_map, s); break;
        case 2: e = launch_mw<32, 2>(a, sxf, sxx, ctr, G, lds, nsb, xcd_map, s); break;
        case 3: e = launch_mw<32, 3>(a, sxf, sxx, ctr, G, lds, nsb, xcd_map, s); break;
        default: e = launch_mw<32, 4>(a, sxf, sxx, ctr, G, lds, nsb, xcd_map, s); break;
    }
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mstep_finish_wide_kernel<32>, dim3(a.B, (a.N + 255) / 256), dim3(256), 0, s, a, (const double*)sxf, (const double*)sxx);
    return hipGetLastError();
}

}  // namespace dfm
