// collapse_mfma.hip -- the balanced-panel collapse with the contraction on the fp64 matrix pipe.
//
//     b_t = sum_i lam_i x_it / R_i   (r)        sum_t s_t,  s_t = sum_i x_it^2 / R_i   
//
// Same contract as collapse_dma_kernel (collapse_dma.hip): no missing cell on this path, one workgroup of
// 4 waves per replicate, wave w owns a contiguous quarter of the periods and streams it HBM -> LDS with
// the LDS-DMA (`global_load_lds_dwordx4`).  What differs is the arithmetic.  The VALU version spends
// ~390 vector instructions per 4 periods (r fp64 FMAs per cell at 4 cycles each, a 32-value cross-lane
// transpose-reduce, selects), which keeps the SIMDs 53 % busy and, with two waves per SIMD, costs 50 us
// on top of the 140 us the stream itself takes (profiles/r01/collapse_ablation.txt).  Here 4 periods are
// one row block of `v_mfma_f64_4x4x4_4b_f64`: D_blk[i][j] += sum_k A_blk[i][k] B_blk[k][j] for 4 blocks,
//     i = period within the row block, k = 4 consecutive series, j = 4 factors,
//     the 4 blocks = (series group g) x (factor group h),
// so one instruction retires 4 periods x 4 NCG series x R factors with the reduction over series done
// inside the matrix pipe; the per-lane work left is one ds_read_b64 per step.
//
// Lane layout of the instruction (measured with scripts/microbench/mfma44.hip; lane l, K = l / 16,
// blk = (l / 4) % 4, q = l % 4):   A_blk[i = q][k = K],   B_blk[k = K][j = q],   D_blk[i = K][j = q].
//
// LDS ring: NS row slots of SB bytes per wave (one period per slot, slot = period mod NS), filled by
// ceil(8N / 1024) DMAs per period; a row block is read once every A operand has landed (counted
// `s_waitcnt vmcnt`), then its slots are re-armed two row blocks ahead.  s_t is accumulated by a second,
// duplicate-free read of the row block (lane owns 16-byte column pairs), as in the VALU kernel.
// Reference counterpart: forming Lambda' x_t in the per-period regression of x_t on Lambda
// (dfm_functions.ipynb:271-286 called from :364).
#include <string.h>

#include <type_traits>

#include "dfm_cov.h"
#include "dfm_gram.h"
#include "dfm_kernels.h"

namespace dfm {

using lds_char_ptr_m = __attribute__((address_space(3))) char*;

__device__ __forceinline__ void dma16m(const void* gsrc, unsigned lds_dst) {
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
template <int K>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(K) : "memory");
}
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

#ifdef DFM_MFMA_BENCH_ONLY      // scripts/microbench/colbw.hip: only the headline shape, with the ablation variants
constexpr bool kMfmaBench = true;
#else
constexpr bool kMfmaBench = false;
#endif

#ifdef DFM_MFMA_NO_PROGRESS_PRIO
constexpr bool kMfmaProgressPrio = false;
#else
constexpr bool kMfmaProgressPrio = true;
#endif

// Geometry of one MFMA step for padded factor dimension R.
template <int R>
struct MfmaGeo {
    static constexpr int NFG = (R + 3) / 4;                 // factor groups of 4
    static constexpr int FPI = NFG < 4 ? NFG : 4;           // factor groups per instruction
    static constexpr int NCG = 4 / FPI;                     // series groups per instruction
    static constexpr int NINST = NFG / FPI;                 // instructions per step
    static constexpr int CS = 4 * NCG;                      // series per step
};

// row-slot stride: 8N rounded so that 4 consecutive slots start 64 bytes apart modulo 256 (the A-operand
// read of a half wave touches 4 periods x 64 bytes)
__host__ __device__ inline unsigned mfma_slot_bytes(int N) {
    unsigned sb = (unsigned)N * 8u;
    while ((sb & 255u) != 64u && (sb & 255u) != 192u) sb += 16u;
    return sb;
}

// STEPS = ceil(N / CS) exactly (the B operands and the staged A operands are register arrays).
// FUSE: the first `ncov` workgroups of the grid are the covariance workgroups of the pass (cov_body of dfm_cov.h on
// wave 0, then the data-independent rows of P_smooth for its replicates); the streaming workgroups follow.  One
// launch instead of cov_kernel + pfill_kernel on a forked stream: the covariance waves are resident first by
// construction (lowest block indices) and the pass loses two cross-stream event edges (~16 us).
template <int R>
struct FusedCov {   // LDS table depth that fits the collapse launch's dynamic LDS (4 waves x 8 slots x SB)
    static constexpr int ECL = R <= 4 ? 8 : R <= 8 ? 3 : 0;
    using LY = CovLayout<R, ECL>;
    static constexpr size_t lds_bytes() { return (size_t)LY::GPW * LY::S * sizeof(double); }
};

template <int R, int STEPS, int NB, int NDR, int ABL, bool FUSE = false>
__global__ __launch_bounds__(256, ((STEPS * MfmaGeo<R>::NINST <= 25 && !FUSE) ? 3 : 2)) void collapse_mfma_kernel(CollapseArgs a, unsigned SB,
                                                                                                          FastArgs fa, int ncov) {
    if constexpr (FUSE) {
        if ((int)blockIdx.x < ncov) {
            extern __shared__ __attribute__((aligned(16))) double csm[];
            using FC = FusedCov<R>;
            const int first = (int)blockIdx.x * FC::LY::GPW;
            if (fa.Lam != nullptr) {   // Gram matrices of this workgroup's replicates (gram_kernel's work), one per wave at a time
                const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
                const int N = fa.N;
                for (int rep = wv; rep < FC::LY::GPW; rep += 4) {
                    const int bb = (first + rep < fa.B) ? first + rep : fa.B - 1;
                    const double* __restrict__ Lg = fa.Lam + (size_t)bb * N * R;
                    const double* __restrict__ Rg = fa.Rv + (size_t)bb * N;
                    double W[NDR][2][R];
                    bool own[NDR][2];
                    double ld = 0.0;
#pragma unroll
                    for (int j = 0; j < NDR; ++j)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int c = 2 * ln + 128 * j + e;
                            own[j][e] = c < N;
                            const int cc = own[j][e] ? c : N - 1;
                            const double rv = own[j][e] ? Rg[cc] : 1.0;
                            const double ri = own[j][e] ? 1.0 / rv : 0.0;
                            ld += log(rv);
#pragma unroll
                            for (int k2 = 0; k2 < R; ++k2) W[j][e][k2] = Lg[(size_t)cc * R + k2] * ri;
                        }
                    c_all<R, NDR, 0, true>(W, Lg, own, ln, const_cast<double*>(fa.Cfull) + (size_t)bb * R * R);
                    ld = wave_allsum(ld);
                    if (ln == 0) const_cast<double*>(fa.ldfull)[bb] = ld;
                }
                __syncthreads();   // Cfull / ldfull of the workgroup's replicates are written (workgroup-visible)
            }
            if (threadIdx.x < 64) {
                __builtin_amdgcn_s_setprio(3);
                cov_body<R, 0, FC::ECL>(fa, first, csm, (int)threadIdx.x);
            }
            __syncthreads();   // waves 1-3 wait here; wave 0's fill[] / PsInf stores are workgroup-visible after it
            if (fa.P_smooth) {   // the fixed-point rows of P_smooth (pfill_kernel's work): the 4 waves take replicates in turn
                const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
                const int npr = fa.r * (fa.r + 1) / 2;
                double* ps = csm + wv * ((R * (R + 1) / 2 + 3) & ~3);
                for (int rep = wv; rep < FC::LY::GPW; rep += 4) {
                    const int bb = first + rep;
                    if (bb >= fa.B) break;
                    wave_lds_sync();
                    for (int v = ln; v < npr; v += 64) {
                        int ri = 0;
                        while ((ri + 1) * (ri + 2) / 2 <= v) ++ri;
                        ps[v] = fa.PsInf[(size_t)bb * R * R + ri * R + (v - ri * (ri + 1) / 2)];
                    }
                    wave_lds_sync();
                    fill_psmooth_rows(fa, bb, ln, 64, ps);
                }
            }
            return;
        }
    }
    const int blk_first = FUSE ? ncov : 0;
    using G = MfmaGeo<R>;
    constexpr int NS = 4 * NB;                               // row slots per wave
    constexpr int CS = G::CS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned long long t_entry = ABL == 2 ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long rt_entry = ABL == 2 ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // Waves are the unit of work: wave gw of the grid takes segment gw % wpr of replicate gw / wpr, so a
    // workgroup's four waves may serve different replicates (each has its own ring and weights).  With
    // wpr = (resident waves of the chip) / B every wave slot gets one equal share and the grid is one round.
    const int wpr = a.wpr > 0 ? a.wpr : 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = ((int)blockIdx.x - blk_first) * 4 + wave;
    if (gw >= a.B * wpr) return;
    const int b = gw / wpr + a.b0;
    const int segi = gw % wpr;
    const int N = a.N, T = a.T;
    const unsigned rowB = (unsigned)N * 8u;
    const double* __restrict__ L = a.Lam + (size_t)b * N * R;
    const double* __restrict__ Rv = a.Rv + (size_t)b * N;

    // lane roles
    const int K = lane >> 4, blk = (lane >> 2) & 3, q = lane & 3;
    const int g = blk / G::FPI, h = blk % G::FPI;

    // this wave's periods [ta, tb): segment segi of the wpr equal segments of [0, T), each starting on a
    // 128-byte boundary
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

    const unsigned ringB = NS * SB;
    const char* ring = smem + (size_t)wave * ringB;
    const unsigned ring_lds =
        __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_m)(smem)) + (unsigned)wave * ringB;
    const unsigned lane16 = 16u * lane;
    bool pact[NDR];                                          // lane moves 16 bytes of piece p of a row
#pragma unroll
    for (int p = 0; p < NDR; ++p) pact[p] = lane16 + 1024u * p < rowB;

    // one period -> NDR DMAs into its slot
    auto issue_row = [&](int r, int slot) {
        const char* src = seg + (size_t)r * rowB + lane16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)slot * SB);
#pragma unroll
        for (int p = 0; p < NDR; ++p) {
            if (pact[p]) dma16m(src + 1024 * p, dst + 1024u * p);
        }
    };
    // B operands: W[c][f] = lam_cf / R_c for c = s CS + 4 g + K, f = 4 (h + FPI m) + q.  All loads first
    // (clamped indices, no branch), then the arithmetic: a load waited for one at a time would also wait
    // for the ring fill issued above (one vmcnt counter) -- ~1.5 us each under load.
    int issued = 0;                                          // periods issued (or skipped past the end)
    double Bw[STEPS][G::NINST];
    {
        double rv[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int c = s * CS + 4 * g + K;
            const int cc = c < N ? c : N - 1;
            rv[s] = Rv[cc];
#pragma unroll
            for (int m = 0; m < G::NINST; ++m) {
                const int f = 4 * (h + G::FPI * m) + q;
                Bw[s][m] = L[(size_t)cc * R + (f < R ? f : R - 1)];
            }
        }
        // the first ring fill goes out behind the parameter loads (older operations return first, so the
        // waits the compiler places before the arithmetic below do not wait for the fill)
        if (nrows > 0) {
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                if (issued < nrows) issue_row(issued, sl);
                ++issued;
            }
        }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int c = s * CS + 4 * g + K;
            const double ri = 1.0 / rv[s];
#pragma unroll
            for (int m = 0; m < G::NINST; ++m) {
                const int f = 4 * (h + G::FPI * m) + q;
                const double w = Bw[s][m] * ri;
                Bw[s][m] = (c < N && f < R) ? w : 0.0;
            }
        }
    }
    // columns of the duplicate-free pass for s_t: lane owns the 16-byte pairs lane + 64 j
    constexpr int NQ = NDR;                                  // pairs per lane: ceil(8N / 1024)
    double qa[NQ][2];
#pragma unroll
    for (int j = 0; j < NQ; ++j) { qa[j][0] = 0.0; qa[j][1] = 0.0; }

    if (nrows <= 0) {
        if (lane == 0) a.ssum[(size_t)b * kSsumSlots + segi] = 0.0;
        return;
    }

    // A operand of step s: period (r0 + q), series s CS + 4 g + K.  The last step may run past the row
    // (N not a multiple of CS): those lanes read the row's last series instead (their B operand is 0).
    const unsigned lane_off = (unsigned)q * SB + (unsigned)(4 * g + K) * 8u;
    const bool tail_clamp = (STEPS - 1) * CS + 4 * g + K >= N;
    const unsigned last_off = tail_clamp ? (unsigned)q * SB + (unsigned)(N - 1) * 8u : lane_off + (unsigned)(STEPS - 1) * (CS * 8u);

    // ABL == 2: s_memtime stamps around the phases of every row block (diagnostics; sums per wave -> scol)
    unsigned long long tw = 0, tr = 0, ti = 0, tc = 0, t_start = 0;
    auto now = [&]() -> unsigned long long { return ABL == 2 ? __builtin_amdgcn_s_memtime() : 0ull; };
    if constexpr (ABL == 2) t_start = now();

    // b_t stores are deferred by kDefer row blocks and issued right behind a re-arm, 16 bytes per lane.
    // Measured (profiles/r01/collapse_ablation.txt): the 32 MB of b_t cost ~30 us of the kernel's 170 -- NOT through
    // the wave's in-order vmcnt stream (deferring, bursting 4-8 blocks, non-temporal and 16-byte stores all change
    // nothing; stores to a few hot lines instead are free) but as write traffic mixed into the 6 TB/s read stream
    // (bulk writes at the end of each wave still cost 16 us).  Open item: stage b_t in LDS for the consumer.
    constexpr int kDefer = 1;
    double pendD[kDefer][G::NINST];
    int pendT[kDefer];                                       // period of the lane's pending value, -1: none
#pragma unroll
    for (int dq = 0; dq < kDefer; ++dq) {
        pendT[dq] = -1;
#pragma unroll
        for (int m = 0; m < G::NINST; ++m) pendD[dq][m] = 0.0;
    }
    auto store_pending = [&](int dq) {
#pragma unroll
        for (int m = 0; m < G::NINST; ++m) {
            const int f = 4 * (h + G::FPI * m) + q;
            if constexpr (R >= 2) {                          // 16-byte stores: the even-q lane writes factors (f, f + 1)
                const double hi = xor_lane<1>(pendD[dq][m]);
                if (g == 0 && pendT[dq] >= 0 && (q & 1) == 0 && f < R)
                    *reinterpret_cast<double2*>(&a.bcol[((size_t)b * T + pendT[dq]) * R + f]) = make_double2(pendD[dq][m], hi);
            } else {
                if (g == 0 && pendT[dq] >= 0 && f < R) a.bcol[((size_t)b * T + pendT[dq]) * R + f] = pendD[dq][m];
            }
        }
    };

    // One row block.  MODE 0: main loop (counted wait; every slot of the block is re-armed with a period
    // that exists).  MODE 1: the block after the main loop (counted wait still valid; the last < 4 periods
    // are issued).  MODE 2: drain (nothing left to issue; rows of the block may lie past the segment).
    auto row_block = [&](int bk, int bslot, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool REARM = MODE == 0;
        const int r0 = bk * 4;
        const unsigned long long s0 = now();
        // rows < r0 + 4 have landed once at most the operations YOUNGER than their DMAs are outstanding: one
        // wave's loads and stores retire in issue order on the single gfx9 vmcnt counter.
        if constexpr (MODE <= 1) {
            constexpr int KW = (NB - 1) * 4 * NDR;
            // younger than the DMAs this block needs (re-arm of iteration bk - NB): the deferred stores issued
            // behind the re-arms of iterations bk - NB .. bk - 1 (NB of them once the queue is primed) and the
            // re-arms of iterations bk - NB + 1 .. bk - 1
            if (bk >= kDefer + NB) wait_vm<(KW + NB * G::NINST <= 63 ? KW + NB * G::NINST : 63)>();
            else wait_vm<(KW <= 63 ? KW : 63)>();
        } else {
            wait_vm<0>();
        }
        const unsigned long long s1 = now();
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
                xq[rr][j] = pact[j] ? *reinterpret_cast<const double2*>(pq + (unsigned)rr * SB + 1024u * j)
                                    : make_double2(0.0, 0.0);
        wait_lgkm();                                         // the reads are done before the slots are re-armed
        const unsigned long long s2 = now();
        if constexpr (MODE == 0) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) issue_row(issued + rr, bslot * 4 + rr);
            issued += 4;
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (issued + rr < nrows) issue_row(issued + rr, bslot * 4 + rr);
            issued += 4;
        }
        if constexpr (ABL != 1 && ABL != 3) store_pending(kDefer - 1);   // the row block of kDefer iterations ago
        const unsigned long long s3 = now();
        if constexpr (ABL == 1) {                            // ablation: DMA + LDS reads only
            double z = 0.0;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) z += xa[s];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int j = 0; j < NQ; ++j) z += xq[rr][j].x + xq[rr][j].y;
            if (z == 1.2345e300) a.bcol[(size_t)b * T + ta + bk] = z;
        } else {
            double D[G::NINST];
#pragma unroll
            for (int m = 0; m < G::NINST; ++m) D[m] = 0.0;
#pragma unroll
            for (int s = 0; s < STEPS; ++s)
#pragma unroll
                for (int m = 0; m < G::NINST; ++m) {
                    if constexpr (ABL == 5) D[m] += xa[s];   // ablation: no MFMA
                    else D[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(xa[s], Bw[s][m], D[m], 0, 0, 0);
                }
            // s_t partial sums: rows past the end of the segment hold stale slots and are left out
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                if constexpr (ABL == 4) { if (rr > 0) continue; }   // ablation: (almost) no s_t pass
                if (REARM || r0 + rr < nrows) {              // wave-uniform
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        qa[j][0] = fma(xq[rr][j].x, xq[rr][j].x, qa[j][0]);
                        qa[j][1] = fma(xq[rr][j].y, xq[rr][j].y, qa[j][1]);
                    }
                }
            }
            // fold the series groups: D of lane (K, g, h, q) -> sum over g
#pragma unroll
            for (int m = 0; m < G::NINST; ++m) {
                if constexpr (G::NCG >= 2) D[m] += xor_lane<8>(D[m]);
                if constexpr (G::NCG == 4) D[m] += xor_lane<4>(D[m]);
            }
            // lane (K = period, g = 0, h, q) stores factor 4 (h + FPI m) + q of period r0 + K
            const int t = ta + r0 + K;
            if constexpr (ABL == 3) {                        // ablation: no b_t store
                if (D[0] == 1.2345e300) a.bcol[(size_t)b * T + ta + bk] = D[0];
            } else {
#pragma unroll
                for (int dq = kDefer - 1; dq > 0; --dq) {
                    pendT[dq] = pendT[dq - 1];
#pragma unroll
                    for (int m = 0; m < G::NINST; ++m) pendD[dq][m] = pendD[dq - 1][m];
                }
                pendT[0] = (REARM || t < tb) ? t : -1;
#pragma unroll
                for (int m = 0; m < G::NINST; ++m) pendD[0][m] = D[m];
            }
        }
        if constexpr (ABL == 2) {
            asm volatile("s_nop 0" ::: "memory");
            const unsigned long long s4 = now();
            tw += s1 - s0; tr += s2 - s1; ti += s3 - s2; tc += s4 - s3;
        }
    };

    // main loop: blocks whose re-arm rows all exist: 4 (bk + NB) + 3 < nrows
    const int nmain = (nrows - 4 * NB) >= 4 ? (nrows - 4 * NB) / 4 : 0;
    int bslot = 0;
    int bk = 0;
    for (; bk < nmain; ++bk) {
        if constexpr (kMfmaProgressPrio) {
            // every wave of the grid has the same work and starts together, but the SQ issues oldest-first, so
            // the waves of a CU drift apart and the stragglers cannot keep HBM busy alone: a wave that is
            // further along yields (priority 3 -> 0 over its segment) and the CU's waves finish together
            const int quarter = (4 * bk) / (nmain > 0 ? nmain : 1);
            if (quarter == 0) __builtin_amdgcn_s_setprio(3);
            else if (quarter == 1) __builtin_amdgcn_s_setprio(2);
            else if (quarter == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        row_block(bk, bslot, std::integral_constant<int, 0>{});
        bslot = (bslot + 1 == NB) ? 0 : bslot + 1;
    }
    if (bk < nblk && nrows >= NS) {     // the initial fill was complete: the counted wait holds once more
        row_block(bk, bslot, std::integral_constant<int, 1>{});
        bslot = (bslot + 1 == NB) ? 0 : bslot + 1;
        ++bk;
    }
    for (; bk < nblk; ++bk) {
        row_block(bk, bslot, std::integral_constant<int, 2>{});
        bslot = (bslot + 1 == NB) ? 0 : bslot + 1;
    }
    if constexpr (ABL == 2) {
        if (lane == 0 && a.scol) {
            double* o = a.scol + (size_t)b * T + segi * 8;
            o[0] = (double)tw; o[1] = (double)tr; o[2] = (double)ti; o[3] = (double)tc;
            o[4] = (double)(now() - t_start); o[5] = (double)nblk; o[6] = (double)(t_start - t_entry);
        }
    }
#pragma unroll
    for (int dq = kDefer - 1; dq >= 0; --dq) store_pending(dq);   // the last kDefer row blocks
    const unsigned long long t_loop_end = now();
    wait_vm<0>();
    // s = sum_t sum_i x_it^2 / R_i over this wave's periods
    double sp = 0.0;
    {
        double rq[NQ][2];
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = 2 * lane + 128 * j + e;
                rq[j][e] = Rv[c < N ? c : N - 1];
            }
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = 2 * lane + 128 * j + e;
                const double ri = 1.0 / rq[j][e];
                sp = fma(qa[j][e], (c < N) ? ri : 0.0, sp);
            }
    }
    sp = wave_allsum(sp);
    if (lane == 0) {
        a.ssum[(size_t)b * kSsumSlots + segi] = sp;
        if (sp != sp) atomicOr(a.status, 1);   // NaN in the panel on the balanced path
        if constexpr (ABL == 2) {
            if (a.scol) {
                a.scol[(size_t)b * T + segi * 8 + 7] = (double)(now() - t_loop_end);
                double* o = a.scol + (size_t)b * T + 200 + segi * 4;       // wave record: real-time start/end, tick span
                o[0] = (double)rt_entry; o[1] = (double)__builtin_amdgcn_s_memrealtime(); o[2] = (double)(now() - t_entry);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
constexpr bool mfma_can_fuse(int R) { return R == 4 || R == 8; }

template <int R, int STEPS, int NB, int NDR, int ABL = 0>
static hipError_t launch_mfma_one(const CollapseArgs& a, hipStream_t s) {
    const unsigned SB = mfma_slot_bytes(a.N);
    size_t lds = (size_t)4 * 4 * NB * SB;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int wpr = a.wpr > 0 ? a.wpr : 4;
    if (wpr > kSsumSlots) return hipErrorInvalidValue;
    const unsigned nstream = (unsigned)((a.B * wpr + 3) / 4);
    if constexpr (ABL == 0 && mfma_can_fuse(R)) {
        if (a.fuse_cov) {   // covariance workgroups at the front of the grid
            const FastArgs& fa = *static_cast<const FastArgs*>(a.fuse_cov);
            using FC = FusedCov<R>;
            if (FC::lds_bytes() > lds) lds = FC::lds_bytes();
            const int ncov = (a.B + FC::LY::GPW - 1) / FC::LY::GPW;
            static LdsOptIn attr_f;
            if (!attr_f && lds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&collapse_mfma_kernel<R, STEPS, NB, NDR, 0, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
                attr_f = true;
            }
            CollapseArgs ak = a;
            ak.fuse_cov = nullptr;
            hipLaunchKernelGGL((collapse_mfma_kernel<R, STEPS, NB, NDR, 0, true>), dim3(nstream + ncov), dim3(256), lds, s, ak, SB,
                               fa, ncov);
            return hipGetLastError();
        }
    }
    if (a.fuse_cov) return hipErrorInvalidValue;
#ifndef DFM_DIAG
    // production: widths whose covariance workgroups ride in this launch (R = 4 | 8) always arrive with them (capi.hip: the two-launch
    // pass); the stand-alone instantiation is reached only through switches of the diagnostics build and is not compiled here
    if constexpr (ABL == 0 && mfma_can_fuse(R)) return hipErrorInvalidValue;
    else
#endif
    {
        static LdsOptIn attr_done;
        if (!attr_done && lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&collapse_mfma_kernel<R, STEPS, NB, NDR, ABL>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        FastArgs none;
        memset(&none, 0, sizeof(none));
        hipLaunchKernelGGL((collapse_mfma_kernel<R, STEPS, NB, NDR, ABL>), dim3(nstream), dim3(256), lds, s, a, SB, none, 0);
        return hipGetLastError();
    }
}

template <int R, int STEPS, int NB, int NDR>
static hipError_t mfma_launch_abl(const CollapseArgs& a, hipStream_t s, int abl) {
    if constexpr (kMfmaBench) {
        if (abl == 1) return launch_mfma_one<R, STEPS, NB, NDR, 1>(a, s);
        if (abl == 2) return launch_mfma_one<R, STEPS, NB, NDR, 2>(a, s);
        if (abl == 3) return launch_mfma_one<R, STEPS, NB, NDR, 3>(a, s);
        if (abl == 4) return launch_mfma_one<R, STEPS, NB, NDR, 4>(a, s);
        if (abl == 5) return launch_mfma_one<R, STEPS, NB, NDR, 5>(a, s);
    }
    return launch_mfma_one<R, STEPS, NB, NDR, 0>(a, s);
}

template <int R, int STEPS>
static hipError_t launch_mfma_steps(const CollapseArgs& a, hipStream_t s, int variant) {
    const int ndr = (a.N * 8 + 1023) / 1024;
    const int nb = (kMfmaBench && (variant / 10) % 10 == 3) ? 3 : 2;
    const int abl = variant % 10;
#define DFM_M(NB_, NDR_)                                                        \
    if (nb == NB_ && ndr == NDR_)                                               \
        return mfma_launch_abl<R, STEPS, NB_, NDR_>(a, s, abl);
    // series per step CS -> N in (CS (STEPS - 1), CS STEPS] -> the possible DMA counts per period
    constexpr int lo = (MfmaGeo<R>::CS * (STEPS - 1) * 8 + 8 + 1023) / 1024, hi = (MfmaGeo<R>::CS * STEPS * 8 + 1023) / 1024;
    if constexpr (lo <= 1 && 1 <= hi) { DFM_M(2, 1) }
    if constexpr (lo <= 2 && 2 <= hi) { DFM_M(2, 2) if constexpr (kMfmaBench) { DFM_M(3, 2) } }
    if constexpr (lo <= 3 && 3 <= hi) { DFM_M(2, 3) }
    if constexpr (lo <= 4 && 4 <= hi) { DFM_M(2, 4) }
#undef DFM_M
    return hipErrorInvalidValue;
}

constexpr int kMfmaMaxSteps = 32;   // register budget: B operands + staged A operands, 2 doubles per step

bool collapse_mfma_fuses_cov(int Rpad, int N) { return collapse_mfma_supported(Rpad, N) && mfma_can_fuse(Rpad); }

// supported: Rp <= 16, even N, ceil(N / CS) <= 32, 8N <= 4096
bool collapse_mfma_supported(int Rpad, int N) {
    if ((N & 1) != 0 || N * 8 > 4096 || N < 4) return false;
    int cs;
    switch (Rpad) {
        case 2: cs = MfmaGeo<2>::CS; break;
        case 4: cs = MfmaGeo<4>::CS; break;
        case 8: cs = MfmaGeo<8>::CS; break;
        case 16: cs = MfmaGeo<16>::CS; break;
        default: return false;
    }
    return (N + cs - 1) / cs <= kMfmaMaxSteps;
}

template <int R, int S>
static hipError_t launch_mfma_pick(const CollapseArgs& a, hipStream_t s, int variant, int steps) {
    if constexpr (S > kMfmaMaxSteps) {
        return hipErrorInvalidValue;
    } else {
        if constexpr (!kMfmaBench || (R == 8 && S == 25)) {
            if (steps == S) return launch_mfma_steps<R, S>(a, s, variant);
        }
        return launch_mfma_pick<R, S + 1>(a, s, variant, steps);
    }
}

template <int R>
static hipError_t launch_mfma_r(const CollapseArgs& a, hipStream_t s, int variant) {
    const int steps = (a.N + MfmaGeo<R>::CS - 1) / MfmaGeo<R>::CS;
    return launch_mfma_pick<R, 1>(a, s, variant, steps);
}

hipError_t launch_collapse_mfma(int Rpad, const CollapseArgs& a, hipStream_t s, int variant) {
    note_kernel("collapse_mfma_kernel");
    switch (Rpad) {
        case 2: return launch_mfma_r<2>(a, s, variant);
        case 4: return launch_mfma_r<4>(a, s, variant);
        case 8: return launch_mfma_r<8>(a, s, variant);
        case 16: return launch_mfma_r<16>(a, s, variant);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dfm
