// mstep_mfma.hip -- the loadings half of the EM M-step for BALANCED panels with the contraction on the fp64 matrix
// pipe: the second streaming read of the panel in an EM iteration,
//     Sxf_i = sum_t x_ti E[f_t | X]  (N x r),   Sxx_i = sum_t x_ti^2,
//     lam_i = S11^-1 Sxf_i,   R_i = (Sxx_i - 2 lam_i' Sxf_i + lam_i' S11 lam_i) / T          
// -- mstep_lam_kernel (mstep.hip) restricted to fully observed series, where every series shares S11.
// The reference's counterpart is the per-series OLS of x_i on the factors (dfm_functions.ipynb:355-362, :391-404).
//
// Same streaming skeleton as collapse_mfma.hip (one wave per period segment, LDS-DMA ring of period slots, counted
// vmcnt waits), transposed contraction: per row block of 4 periods and per step of CS series ONE
// `v_mfma_f64_4x4x4_4b_f64` with  A_blk[i = series][k = period] = x,  B_blk[k = period][j = factor] = f,  so that
// D_blk[i][j] accumulates Sxf for 4 series x 4 factors over the wave's whole segment: STEPS accumulator registers
// per lane, no store inside the loop.  The per-segment partial sums go to a workspace ([B][wpr][N][Rp] + [B][wpr][N])
// and mstep_finish_kernel (one workgroup per replicate, thread = series) adds them and solves.
#include <string.h>

#include <type_traits>

#include "dfm_gram.h"
#include "dfm_em_update.h"
#include "dfm_kernels.h"

namespace dfm {

using lds_char_ptr_ms = __attribute__((address_space(3))) char*;
__device__ __forceinline__ void dma16s(const void* gsrc, unsigned lds_dst, bool nt) {   // nt: see pass_fused.hip dma16f
    unsigned keep;
    if (nt) {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_dst)
            : "memory");
    } else {
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
}
template <int K>
__device__ __forceinline__ void wait_vms() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(K) : "memory");
}

template <int R>
struct MsGeo {
    static constexpr int NFG = (R + 3) / 4;                 // factor groups of 4
    static constexpr int FPI = NFG < 4 ? NFG : 4;
    static constexpr int NCG = 4 / FPI;                     // series groups per instruction
    static constexpr int CS = 4 * NCG;                      // series per step
};
__host__ __device__ inline unsigned ms_slot_bytes(int N) {   // as collapse_mfma.hip: 4 consecutive slots 64 B apart mod 256
    unsigned sb = (unsigned)N * 8u;
    while ((sb & 255u) != 64u && (sb & 255u) != 192u) sb += 16u;
    return sb;
}

template <int R, int STEPS, int NDR>
__global__ __launch_bounds__(256, 2) void mstep_mfma_kernel(MstepArgs a, unsigned SB, int wpr, double* part_sxf, double* part_sxx, EmUpdArgs ua,
                                                            int nfront_, int dma_nt) {
    using G = MsGeo<R>;
    constexpr int NB = 2, NS = 4 * NB, CS = G::CS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // The first nfront workgroups are not streaming workgroups: each of their waves runs the TRANSITION half of the M-step of one
    // replicate (dfm_em_update.h: sums of f f', A, Q, mu0, P0, S11^-1, the EM bookkeeping).  It depends on the E-step only, so it
    // hides behind the second panel stream instead of being a 33-us launch in front of it; mstep_finish_kernel (the next launch)
    // is its only consumer.  Streaming waves may read `active` before or after its update: a replicate that stops in this
    // iteration then computes sums nobody uses -- except in iteration 0, where the array is still UNINITIALISED until the front
    // waves have written it (every replicate is active then by definition): nfront < 0 says "do not look at it".
    const int nfront = nfront_ < 0 ? -nfront_ : nfront_;
    const bool trust_active = nfront_ >= 0;
    if ((int)blockIdx.x < nfront) {
        const int be = (int)blockIdx.x * 4 + wave;
        constexpr int kEmDoubles = (64 / R) * (R * R + 2 * R);
        em_update_wave<R>(ua, be < ua.B ? be : ua.B - 1, be < ua.B, lane, reinterpret_cast<double*>(smem) + wave * kEmDoubles);
        return;
    }
    const int gw = ((int)blockIdx.x - nfront) * 4 + wave;
    if (gw >= a.B * wpr) return;
    const int b = gw / wpr, segi = gw % wpr;
    if (trust_active && a.active && !a.active[b]) return;
    const int N = a.N, T = a.T;
    const unsigned rowB = (unsigned)N * 8u;
    const int K = lane >> 4, blk = (lane >> 2) & 3, q = lane & 3;
    const int g = blk / G::FPI, h = blk % G::FPI;

    int tq = (T + wpr - 1) / wpr;
    {
        unsigned gg = rowB & 127u;
        gg = gg == 0 ? 128u : (gg & (~gg + 1u));
        const int m = (int)(128u / gg);
        tq = ((tq + m - 1) / m) * m;
    }
    const int ta = (segi * tq < T) ? segi * tq : T;
    const int tb = (ta + tq < T) ? ta + tq : T;
    const int nrows = tb - ta;
    const int nblk = (nrows + 3) / 4;
    const char* __restrict__ seg = reinterpret_cast<const char*>(a.panel + ((size_t)b * T + ta) * N);
    const double* __restrict__ fseg = a.fsm + ((size_t)b * T + ta) * R;

    constexpr unsigned FB = 4u * R * 8u;                     // bytes of one block's factors: 4 periods x R doubles
    const unsigned ringB = NS * SB + NB * FB;                // period slots, then the factor blocks of the NB row blocks
    const char* ring = smem + (size_t)wave * ringB;
    const unsigned ring_lds =
        __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_ms)(smem)) + (unsigned)wave * ringB;
    const unsigned lane16 = 16u * lane;
    bool pact[NDR];
#pragma unroll
    for (int p = 0; p < NDR; ++p) pact[p] = lane16 + 1024u * p < rowB;
    auto issue_row = [&](int r, int slot) {
        const char* src = seg + (size_t)r * rowB + lane16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)slot * SB);
#pragma unroll
        for (int p = 0; p < NDR; ++p)
            if (pact[p]) dma16s(src + 1024 * p, dst + 1024u * p, dma_nt != 0);
    };
    // the factors of row block j (periods 4j .. 4j+3, R doubles each, contiguous) ride the same DMA stream into the
    // block's slot of the factor ring -- every load of the steady state is an LDS-DMA, so one counted vmcnt orders them
    const char* __restrict__ fbytes = reinterpret_cast<const char*>(fseg);
    auto issue_f = [&](int j, int bslot_) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + NS * SB + (unsigned)bslot_ * FB);
        const int row = 4 * j + (int)(lane16 / (8u * R));
        if (lane16 < FB && row < nrows) dma16s(fbytes + (size_t)j * FB + lane16, dst, false);   // (the factors were just written: they are wanted in cache)
    };
    int issued = 0;
    if (nrows > 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (issued < nrows) issue_row(issued, s);
            ++issued;
            if ((s & 3) == 3) issue_f(s >> 2, s >> 2);
        }
    }
    // accumulators: Sxf[series s CS + 4 g + K][factor 4 h + q] in D[s] (lane layout of the result: row = lane / 16)
    double D[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) D[s] = 0.0;
    constexpr int NQ = NDR;
    double qa[NQ][2];
#pragma unroll
    for (int j = 0; j < NQ; ++j) { qa[j][0] = 0.0; qa[j][1] = 0.0; }

    // A operand of step s: period (r0 + K), series s CS + 4 g + q (clamped to the row; rows of the result past N
    // are never stored)
    const int ser0 = 4 * g + q;
    const unsigned lane_off = (unsigned)K * SB + (unsigned)ser0 * 8u;
    const bool tail_clamp = (STEPS - 1) * CS + ser0 >= N;
    const unsigned last_off = tail_clamp ? (unsigned)K * SB + (unsigned)(N - 1) * 8u : lane_off + (unsigned)(STEPS - 1) * (CS * 8u);
    const int fcol = 4 * h + q;                                          // B operand: f[period r0 + K][fcol]
    const bool fvalid_col = fcol < R;

    auto row_block = [&](int bk, int bslot, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const int r0 = bk * 4;
        if constexpr (MODE <= 1) {
            constexpr int KW = (NB - 1) * (4 * NDR + 1);   // the row blocks issued after this one: periods + factors
            wait_vms<(KW <= 63 ? KW : 63)>();
        } else {
            wait_vms<0>();
        }
        const char* blkbase = ring + (unsigned)bslot * 4u * SB;
        const char* pa = blkbase + lane_off;
        double xa[STEPS];
#pragma unroll
        for (int s = 0; s + 1 < STEPS; ++s) xa[s] = *reinterpret_cast<const double*>(pa + s * (CS * 8));
        xa[STEPS - 1] = *reinterpret_cast<const double*>(blkbase + last_off);
        const char* pq = blkbase + lane16;
        double2 xq[4][NQ];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int j = 0; j < NQ; ++j)
                xq[rr][j] = pact[j] ? *reinterpret_cast<const double2*>(pq + (unsigned)rr * SB + 1024u * j) : make_double2(0.0, 0.0);
        const double fb = *reinterpret_cast<const double*>(ring + NS * SB + (unsigned)bslot * FB + ((unsigned)K * R + (unsigned)(fcol < R ? fcol : 0)) * 8u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (MODE == 0) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) issue_row(issued + rr, bslot * 4 + rr);
            issue_f(issued >> 2, bslot);
            issued += 4;
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (issued + rr < nrows) issue_row(issued + rr, bslot * 4 + rr);
            issue_f(issued >> 2, bslot);
            issued += 4;
        }
        const bool rowok = MODE == 0 || (r0 + K) < nrows;                 // periods past the segment: x, f <- 0 (stale slots)
        const double fbv = (rowok && fvalid_col) ? fb : 0.0;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const double x = rowok ? xa[s] : 0.0;
            D[s] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, fbv, D[s], 0, 0, 0);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            if (MODE == 0 || r0 + rr < nrows) {
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    qa[j][0] = fma(xq[rr][j].x, xq[rr][j].x, qa[j][0]);
                    qa[j][1] = fma(xq[rr][j].y, xq[rr][j].y, qa[j][1]);
                }
            }
        }
    };

    if (nrows > 0) {
        const int nmain = (nrows - 4 * NB) >= 4 ? (nrows - 4 * NB) / 4 : 0;
        int bslot = 0, bk = 0;
        for (; bk < nmain; ++bk) {
            row_block(bk, bslot, std::integral_constant<int, 0>{});
            bslot = (bslot + 1 == NB) ? 0 : bslot + 1;
        }
        if (bk < nblk && nrows >= NS) {
            row_block(bk, bslot, std::integral_constant<int, 1>{});
            bslot = (bslot + 1 == NB) ? 0 : bslot + 1;
            ++bk;
        }
        for (; bk < nblk; ++bk) {
            row_block(bk, bslot, std::integral_constant<int, 2>{});
            bslot = (bslot + 1 == NB) ? 0 : bslot + 1;
        }
    }
    wait_vms<0>();
    // partial sums of this segment: fold the duplicate... (each (series, factor) lives in exactly one lane) and store
    double* ps = part_sxf + ((size_t)b * wpr + segi) * (size_t)N * R;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int ser = s * CS + 4 * g + K;                               // D row = lane / 16
        if (ser < N && fcol < R) ps[(size_t)ser * R + fcol] = D[s];
    }
    double* px = part_sxx + ((size_t)b * wpr + segi) * (size_t)N;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int c = 2 * lane + 128 * j;
        if (c < N) px[c] = qa[j][0];
        if (c + 1 < N) px[c + 1] = qa[j][1];
    }
}

// thread = series: add the segments' partial sums, lam_i = S11^-1 Sxf_i, R_i = (Sxx_i - 2 lam_i'Sxf_i + lam_i'S11 lam_i) / T
// (8 waves per SIMD = 64 VGPRs: all 1024 workgroups of a C2 batch are resident at once -- at 2 per SIMD the kernel ran as two rounds of a
// latency chain, 29 us for a few loads and 150 flops per series)
template <int R>
__global__ __launch_bounds__(256, 8) void mstep_finish_kernel(MstepArgs a, int wpr, const double* part_sxf, const double* part_sxx) {
    __shared__ double s11[R * R], s11i[R * R];
    const int b = blockIdx.x;
    if (a.active && !a.active[b]) return;
    const int N = a.N;
    for (int e = threadIdx.x; e < R * R; e += 256) {
        s11[e] = a.S11[(size_t)b * R * R + e];
        s11i[e] = a.S11inv[(size_t)b * R * R + e];
    }
    __syncthreads();
    for (int col = threadIdx.x; col < N; col += 256) {
        asm volatile("" ::: "memory");                        // (the 2 R R matrix entries are re-read from LDS per series, not hoisted into 256 registers)
        double sxf[R], sxx = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) sxf[k] = 0.0;
        for (int w = 0; w < wpr; ++w) {
            const double* ps = part_sxf + (((size_t)b * wpr + w) * N + col) * R;
#pragma unroll
            for (int k = 0; k < R; ++k) sxf[k] += ps[k];
            sxx += part_sxx[((size_t)b * wpr + w) * N + col];
        }
        double lam[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) s = fma(s11i[i * R + k], sxf[k], s);
            lam[i] = s;
        }
        double quad = 0.0, cross = 0.0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) s = fma(s11[i * R + k], lam[k], s);
            quad = fma(lam[i], s, quad);
            cross = fma(lam[i], sxf[i], cross);
        }
        a.R_out[(size_t)b * N + col] = (sxx - 2.0 * cross + quad) / (double)a.T;
        double* lo = a.Lam_out + ((size_t)b * N + col) * a.lam_stride;
#pragma unroll
        for (int k = 0; k < R; ++k) lo[k] = lam[k];
    }
}

// ---------------------------------------------------------------------------------------------
constexpr int kMsMaxSteps = 32;

template <int R, int STEPS, int NDR>
static hipError_t launch_ms_one(const MstepArgs& a, int wpr, double* pf, double* px, hipStream_t s, const EmUpdArgs* ua) {
    const unsigned SB = ms_slot_bytes(a.N);
    size_t lds = (size_t)4 * (8 * SB + 2 * 4 * R * 8);
    if (ua && lds < (size_t)4 * (64 / R) * (R * R + 2 * R) * sizeof(double)) lds = (size_t)4 * (64 / R) * (R * R + 2 * R) * sizeof(double);   // (the front waves' scratch)
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mstep_mfma_kernel<R, STEPS, NDR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    EmUpdArgs u0;
    memset(&u0, 0, sizeof(u0));
    const int nfront = ua ? (a.B + 3) / 4 : 0;                // transition M-step waves in front of the streaming workgroups
    hipLaunchKernelGGL((mstep_mfma_kernel<R, STEPS, NDR>), dim3((a.B * wpr + 3) / 4 + nfront), dim3(256), lds, s, a, SB, wpr, pf, px, ua ? *ua : u0,
                       (ua && ua->k == 0) ? -nfront : nfront, stream_nt_hint((size_t)a.B * a.T * a.N * sizeof(double)) ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((mstep_finish_kernel<R>), dim3(a.B), dim3(256), 0, s, a, wpr, (const double*)pf, (const double*)px);
    return hipGetLastError();
}

template <int R, int S>
static hipError_t launch_ms_pick(const MstepArgs& a, int wpr, double* pf, double* px, hipStream_t s, int steps, const EmUpdArgs* ua) {
    if constexpr (S > kMsMaxSteps) {
        return hipErrorInvalidValue;
    } else {
        if (steps == S) {
            const int ndr = (a.N * 8 + 1023) / 1024;
            constexpr int lo = (MsGeo<R>::CS * (S - 1) * 8 + 8 + 1023) / 1024, hi = (MsGeo<R>::CS * S * 8 + 1023) / 1024;
            if constexpr (lo <= 1 && 1 <= hi) { if (ndr == 1) return launch_ms_one<R, S, 1>(a, wpr, pf, px, s, ua); }
            if constexpr (lo <= 2 && 2 <= hi) { if (ndr == 2) return launch_ms_one<R, S, 2>(a, wpr, pf, px, s, ua); }
            if constexpr (lo <= 3 && 3 <= hi) { if (ndr == 3) return launch_ms_one<R, S, 3>(a, wpr, pf, px, s, ua); }
            if constexpr (lo <= 4 && 4 <= hi) { if (ndr == 4) return launch_ms_one<R, S, 4>(a, wpr, pf, px, s, ua); }
            return hipErrorInvalidValue;
        }
        return launch_ms_pick<R, S + 1>(a, wpr, pf, px, s, steps, ua);
    }
}

// balanced panels, Rp in {4, 8} (the shapes of the fused E-step), even N, ceil(N / CS) <= 32, 8N <= 4096
bool mstep_mfma_supported(int Rpad, int N) {
    if ((N & 1) != 0 || N * 8 > 4096 || N < 4) return false;
    if (Rpad == 4) return (N + MsGeo<4>::CS - 1) / MsGeo<4>::CS <= kMsMaxSteps;
    if (Rpad == 8) return (N + MsGeo<8>::CS - 1) / MsGeo<8>::CS <= kMsMaxSteps;
    return false;
}
size_t mstep_mfma_workspace(int B, int N, int Rpad, int wpr) { return (size_t)B * wpr * ((size_t)N * Rpad + N) * sizeof(double); }

hipError_t launch_mstep_mfma(int Rpad, const MstepArgs& a, int wpr, double* workspace, hipStream_t s, const EmUpdArgs* ua) {
    double* pf = workspace;
    double* px = workspace + (size_t)a.B * wpr * a.N * Rpad;
    if (Rpad == 4) return launch_ms_pick<4, 1>(a, wpr, pf, px, s, (a.N + MsGeo<4>::CS - 1) / MsGeo<4>::CS, ua);
    if (Rpad == 8) return launch_ms_pick<8, 1>(a, wpr, pf, px, s, (a.N + MsGeo<8>::CS - 1) / MsGeo<8>::CS, ua);
    return hipErrorInvalidValue;
}

}  // namespace dfm
