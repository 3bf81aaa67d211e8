// pass_fused.hip -- the whole balanced-panel Kalman-smoother pass in ONE launch (Rp = 8): persistent workgroups, one per
// CU, each walking its replicates b = blockIdx.x, blockIdx.x + gridDim.x, ...  Inside a workgroup the waves are
// specialised and DECOUPLED -- three software-pipeline stages that hand replicates to each other through LDS flags, never
// through a workgroup barrier:
//
//   waves 0 .. nsw-1     STREAM  the collapse of collapse_mfma.hip (LDS-DMA ring of period slots, contraction on
//                                v_mfma_f64_4x4x4): wave w streams segment w of the replicate's T periods.  b_t goes to
//                                one of two LDS arrays [T][8] instead of HBM, sum_t s_t to an LDS slot.  While the scan
//                                waves work on replicate j the stream waves are already filling the other array with
//                                replicate j + 1, so HBM never waits for the (latency-bound) scan.
//   the next ncov waves  COV     Gram matrix C = Lam' R^-1 Lam, the data-independent covariance recursion (dfm_cov8.h: one
//                                wave per replicate, element per lane), the transient rows of P_smooth, then the fixed-point
//                                rows of P_smooth (pure stores).  Nothing here depends on the panel, so these waves run
//                                AHEAD of the stream (replicates j, j + ncov, ... each); their tables go to the
//                                per-replicate workspace in global memory (L2-resident: written and read by the same CU).
//   the last 4 waves     SCAN    the time-parallel mean recursion (dfm_scan.h, the algorithm of meanscan_kernel) with
//                                b_t / w_t in LDS: f_smooth, log-likelihood.  Its internal barriers are 4-wave barriers on
//                                an LDS counter.
//
// Against the two-launch pass (fused collapse launch + meanscan_kernel): no b_t round trip through HBM (33 MB written,
// read back), no w_t scratch (33 MB + 33 MB), no second launch whose ~55 us of dependent scans ran behind the stream
// instead of beside it, and the covariance recursion no longer holds 128 wide waves resident for 125 us.
// Synchronisation: monotone counters in LDS (bt_ready[buffer], scan_done, cov_done[wave]); every wait is BOUNDED (a wait
// that does not end sets bit 2 of the status word and the whole workgroup drains).  No inter-workgroup communication.
// The reference has no counterpart (dfm_functions.ipynb:21-23 declares `Parametric` only).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "dfm_cov8.h"
#include "dfm_gram.h"
#include "dfm_kernels.h"
#include "dfm_scan.h"

// cache-policy modifiers of the streaming LDS-DMA loads (development A/B: -DDFM_DMA_MOD='" nt"', '" sc1"', ...); default: none
#ifndef DFM_DMA_MOD
#define DFM_DMA_MOD ""
#endif

namespace dfm {

namespace {

using lds_char_ptr_f = __attribute__((address_space(3))) char*;

// nt: the non-temporal hint on the panel stream.  The panel is read ONCE per pass; with the hint its lines do not displace the
// rest of the pass's working set from the 256-MB Infinity Cache -- the outputs (P_smooth, f_smooth: 180 MB per 1024 replicates),
// which the next pass or the M-step behind this one overwrites or reads again, and the covariance tables.  Measured in one
// process (profiles/r04/ab_dma_nt_*.txt): +12 % at B = 512, +13..15 % at 1024, +8 % at 1536, +6 % at 2048, +1 % at 4096, -2 % at
// 8192 (nothing of a 6.6-GB batch stays on chip, and the hint costs the L2 its write-combining of neighbouring rows): the
// launcher sets it for batches whose panels are <= 2 GiB.
__device__ __forceinline__ void dma16f(const void* gsrc, unsigned lds_dst, bool nt) {
    unsigned keep;
    if (nt) {                                                  // (wave-uniform)
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
            "global_load_lds_dwordx4 %1, off" DFM_DMA_MOD "\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(lds_dst)
            : "memory");
    }
}
template <int K>
__device__ __forceinline__ void wait_vmf() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(K) : "memory");
}
__device__ __forceinline__ void wait_lgkmf() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int kPfR = 8;
#ifndef DFM_PF_ECAP
#define DFM_PF_ECAP 8
#endif
constexpr int kPfEcap = DFM_PF_ECAP;                                  // transient covariance steps staged in LDS (later ones: global tab)
constexpr int kPfNst = stead_mats(kPfR);                    // steady Z, J, G + the carry powers (256 scan threads)
constexpr int kPfNlev = scan_levels(kPfR);
constexpr int kPfScanWaves = kScanThreads / 64;             // 4
constexpr int kPfMaxCov = 2;
// (development A/B: -DDFM_PF_DEFER=n.  Round 4, with the non-temporal panel stream: 1 -> 0.2064 ms, 2 -> 0.2044, 3 -> 0.2015, 4 -> 0.1955 at
//  B = 1024 in one process, all the same at B = 8192 -- profiles/r04/ab_pf_defer.txt.  While the panel streams HBM is the bottleneck and
//  every byte of fill costs stream time; the scan-only tail of the kernel has the bandwidth to spare.)
#ifndef DFM_PF_DEFER
#define DFM_PF_DEFER 4
#endif
constexpr int kPfDeferDefault = DFM_PF_DEFER;                          // P_smooth fills left to the stream waves' tail (replicates per workgroup)
constexpr int kPfMaxWaves = 12;                             // 3 waves per SIMD: 168 VGPRs each
constexpr int kPfMaxThreads = 64 * kPfMaxWaves;
constexpr unsigned kPfLdsLimit = 160u * 1024u;

// flags (unsigned, LDS): monotone counters
constexpr int kFBtReady = 0;      // [2] arrivals of stream waves per b_t buffer
constexpr int kFScanDone = 2;     // replicates (local index) whose scan is complete
constexpr int kFCovDone = 3;      // [kPfMaxCov] replicates finished by each covariance wave
constexpr int kFScanBar = 5;      // arrivals at the scan waves' barrier
constexpr int kFAbort = 6;
constexpr int kFTabReady = 7;     // replicates whose tables the mover has put into the LDS table set
constexpr int kFScanArrive = 8;   // scan_reg: scan waves that have finished their part of a replicate (4 per replicate)
constexpr int kFStreamWaits = 9;  // 1 while stream wave 0 waits for a b_t buffer, i.e. for the scan: the scan is then the critical path
constexpr int kFCount = 64;

// misc doubles (LDS)
constexpr int kMiscXi0 = 0, kMiscLlc = 8, kMiscE = 9, kMiscVec = 16, kMiscRed = 32, kMiscSsum = 40 /* [2][8] */,
              kMiscPs = 56 /* [36] */, kMiscWtr = 92 /* [kPfEcap][8] */, kMiscDoubles = 92 + kPfEcap * 8;

// row-slot stride: as collapse_mfma.hip (4 consecutive slots start 64 bytes apart modulo 256)
__host__ __device__ inline unsigned pf_slot_bytes(int N) {
    unsigned sb = (unsigned)N * 8u;
    while ((sb & 255u) != 64u && (sb & 255u) != 192u) sb += 16u;
    return sb;
}

__device__ __forceinline__ unsigned ld_flag(const unsigned* f) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// Bounded wait until *f >= target (counters are monotone).  false: the workgroup is draining (time-out somewhere).
__device__ __forceinline__ bool pf_wait_ge(const unsigned* f, unsigned target, unsigned* abortf) {
    unsigned spins = 0;
    for (;;) {
        if ((int)(ld_flag(f) - target) >= 0) break;
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 255u) == 0) {
            if (ld_flag(abortf) != 0) return false;
            if (spins > (1u << 23)) {                        // ~ a second: something upstream died
                __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return false;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}
// everything this wave wrote (LDS and global) is visible to the workgroup before the counter moves
__device__ __forceinline__ void pf_signal(unsigned* f, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Barrier of the kPfScanWaves scan waves on an LDS counter (s_barrier would involve the stream / covariance waves).
struct PfScanSync {
    unsigned* ctr;
    unsigned* abortf;
    unsigned gen;          // arrivals expected once every wave has passed the next barrier
    int lane;
    __device__ __forceinline__ void operator()() {
        gen += kPfScanWaves;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned spins = 0;
        while ((int)(ld_flag(ctr) - gen) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0) {
                if (ld_flag(abortf) != 0) break;
                if (spins > (1u << 23)) { __hip_atomic_store(abortf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};

// global loads that must not be served by the scalar cache or a stale L1 line (written by another wave of this CU)
__device__ __forceinline__ double ld_dev(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_dev(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace

// Rows [lo, hi) of one replicate's P_smooth (npr packed values per row) := ps[0 .. npr): fire-and-forget 16-byte stores by
// ONE wave; part `w` of `nw` equal row ranges (nw = 1: the whole range).  ps: LDS.
__device__ __forceinline__ void pf_fill_rows(double* Prep, int lo, int hi, int npr, const double* ps, int lane, int w, int nw) {
    const int per = (hi - lo + nw - 1) / nw;
    const int a0 = lo + w * per;
    const int a1 = a0 + per < hi ? a0 + per : hi;
    if (a1 <= a0) return;
    double* base = Prep + (size_t)a0 * npr;
    const unsigned n = (unsigned)(a1 - a0) * (unsigned)npr;
    const unsigned peel = ((reinterpret_cast<size_t>(base) & 15) != 0) ? 1u : 0u;
    if (peel && lane == 0) base[0] = ps[0];
    const unsigned npair = (n - peel) / 2;
    const unsigned step = 128u % (unsigned)npr;
    unsigned k = peel + 2u * lane;
    unsigned v = k % (unsigned)npr;
    for (unsigned p = lane; p < npair; p += 64) {
        const unsigned v1 = (v + 1 == (unsigned)npr) ? 0u : v + 1;
        *reinterpret_cast<double2*>(base + k) = make_double2(ps[v], ps[v1]);
        k += 128u;
        v += step;
        if (v >= (unsigned)npr) v -= (unsigned)npr;
    }
    if (((n - peel) & 1u) != 0 && lane == 0) base[n - 1] = ps[(n - 1) % (unsigned)npr];
}

// byte offsets into the dynamic LDS of pass_fused_kernel (computed by the host)
struct PfLds {
    unsigned flags;   // kFCount unsigned
    unsigned smat;    // [kPfNst + 1][64] doubles: steady matrices, then P_T        (scan staging)
    unsigned ctab;    // [kPfEcap][3][64] doubles: Z_e, J_e, G_e                     (scan staging)
    unsigned misc;    // kMiscDoubles doubles
    unsigned covws;   // ncov x kCov8ScratchDoubles doubles
    unsigned sa, sb;  // [32][8] doubles each (carry scan)
    unsigned bt;      // nbuf x [T4][8] doubles: b_t, then w_t
    unsigned bt_stride;   // doubles between the b_t buffers
    unsigned ring;    // nsw x 8 slots x SB bytes
    unsigned total;
    int nsw, ncov, nbuf;
    int nt;           // non-temporal hint on the panel stream (dma16f)
};

#if defined(DFM_DIAG) || defined(DFM_PF_FALLBACK_LDS)   // the round-2 scan: A/B against scan_reg in the diagnostics build only (DFM_SCAN_ABL bit 9); production: scan_reg, scan_seq
// ------------------------------------------------------------------------------------------------------------------
// The scan of one replicate, b_t in LDS, tables staged in LDS.  Called by the 4 scan waves (tid 0 .. 255); `sync` is their
// barrier.  meanscan_kernel's algorithm (fastpath.hip).
// ------------------------------------------------------------------------------------------------------------------
template <class Sync>
__device__ __forceinline__ void scan_lds(const FastArgs& a, int b, int tid, double* bt, const double* s_tab, const double* tab_over,
                                         const double* s_mat, const double* xi0p, const double* llcp, int E, double* s_a,
                                         double* s_b, double* s_vec, double* s_red, const double* ssum, int nseg, Sync& sync,
                                         double* pslot) {
    auto mark = [&](int k) {       // diagnostics: phase stamps of thread 0 (pslot null: off)
        if (pslot && tid == 0) pslot[k] = (double)__builtin_amdgcn_s_memrealtime();
    };
    constexpr int R = kPfR;
    constexpr int NG = kScanThreads / R;
    constexpr int NLEV = kPfNlev;
    constexpr int NST = kPfNst;
    const int c = tid / R, i = tid % R;
    const int T = a.T, r = a.r, L = a.L;
    const int ts = E - 1;
    const int nst = ts < kPfEcap ? ts : kPfEcap;              // transient steps whose tables are in LDS
    double* fout = a.f_smooth + (size_t)b * T * r;
    double xi = xi0p[i];
    const int clast = (T - 1 - ts) / L;                       // chunk that holds step T-1 (fwd) / step ts (bwd)
    const int cmax = c | (64 / R - 1);                        // last chunk handled by this wave
    const int t0f = ts + c * L;                               // forward chunk: steps t0f + j
    const int t0b = T - 1 - c * L;                            // backward chunk: steps t0b - j
    const bool fullf = (ts + (cmax + 1) * L) <= T;            // wave-uniform: no partial chunk in this wave
    const bool fullb = (T - (cmax + 1) * L) >= ts;
    double cur[kPF];
    double dot = 0.0;                     // lane part of sum_t xi_t' w_t

    // ---- forward transient: steps 0 .. ts-1 on wave 0 only (its lane groups redundantly) --------------------------
    if (ts > 0) {                                             // (uniform over the scan waves)
        if (tid < 64) {
            for (int t = 0; t < ts; ++t) {
                double Zp[R], Gp[R];
                const double* ent = t < nst ? s_tab + (size_t)t * 3 * R * R : tab_over + (size_t)t * 3 * R * R;
                load_xperm<R>(Zp, ent, i);
                load_xperm<R>(Gp, ent + 2 * R * R, i);
                const double btv = bt[t * R + i];
                const double w = matvec_x<R>(Zp, xi);
                wave_lds_sync();                              // every lane group has read b_t before group 0 overwrites it
                if (c == 0) {
                    dot = fma(xi, w, dot);
                    bt[t * R + i] = w;                        // b_t is consumed; the slot now holds w_t
                }
                xi = matvec_x<R>(Gp, xi, btv);
            }
            if (c == 0) s_vec[i] = xi;
        }
        sync();
        xi = s_vec[i];
        sync();                           // s_vec is reused for xi_T
    }
    // xi = xi_ts in every group.
    mark(1);

    // ---- steady forward scan: steps ts .. T-1; group c owns steps ts + c L + j ------------------------------------
    {
        double Gp[R], Zp[R];
        load_xperm<R>(Gp, s_mat + 2 * R * R, i);
        double dummy = 0.0;
        if (fullf) chunk_prefetch<R, true>(cur, bt, t0f, 1, L, ts, T, i);
        else chunk_prefetch<R, false>(cur, bt, t0f, 1, L, ts, T, i);
        // phase 1: chunk from a zero state
        const double e = fullf ? chunk_run<R, true, 0>(Gp, Gp, 0.0, cur, bt, t0f, 1, L, ts, T, i, nullptr, dummy, nullptr, r)
                               : chunk_run<R, false, 0>(Gp, Gp, 0.0, cur, bt, t0f, 1, L, ts, T, i, nullptr, dummy, nullptr, r);
        if (fullf) chunk_prefetch<R, true>(cur, bt, t0f, 1, L, ts, T, i);     // operands of phase 3
        else chunk_prefetch<R, false>(cur, bt, t0f, 1, L, ts, T, i);
        mark(2);
        // phase 2: true start state of chunk c
        const double s = carry_scan<R, NG>(e, xi, s_mat + 3 * R * R, c, i, s_a, s_b, true, sync);
        mark(3);
        // phase 3: re-run from the true start; emit w_t (in place of b_t), accumulate xi_t' w_t
        load_xperm<R>(Zp, s_mat, i);
        const double v = fullf ? chunk_run<R, true, 1>(Gp, Zp, s, cur, bt, t0f, 1, L, ts, T, i, bt, dot, nullptr, r)
                               : chunk_run<R, false, 1>(Gp, Zp, s, cur, bt, t0f, 1, L, ts, T, i, bt, dot, nullptr, r);
        if (c == clast) s_vec[i] = v;                         // xi_T
    }
    mark(4);
    sync();            // xi_T in LDS; every w_t of this replicate is written

    // ---- terminal + steady backward scan: steps T-1 .. ts; group c owns steps T-1 - c L - j --------------------
    double fT = 0.0;
    {
        const double xiT = s_vec[i];
        double PTp[R];
        load_xperm<R>(PTp, s_mat + NST * R * R, i);
        fT = matvec_x<R>(PTp, xiT);
        if (c == 0) {
            dot = fma(xiT, fT, dot);          // the log-likelihood needs sum xi'w + xi_T' f_T
            if (i < r) fout[(size_t)(T - 1) * r + i] = fT;
        }
    }
    double fb;   // smoothed mean at the steady/transient boundary (period ts)
    {
        double Jp[R];
        load_xperm<R>(Jp, s_mat + R * R, i);
        double dummy = 0.0;
        if (fullb) chunk_prefetch<R, true>(cur, bt, t0b, -1, L, ts, T, i);
        else chunk_prefetch<R, false>(cur, bt, t0b, -1, L, ts, T, i);
        const double e = fullb ? chunk_run<R, true, 0>(Jp, Jp, 0.0, cur, bt, t0b, -1, L, ts, T, i, nullptr, dummy, nullptr, r)
                               : chunk_run<R, false, 0>(Jp, Jp, 0.0, cur, bt, t0b, -1, L, ts, T, i, nullptr, dummy, nullptr, r);
        if (fullb) chunk_prefetch<R, true>(cur, bt, t0b, -1, L, ts, T, i);
        else chunk_prefetch<R, false>(cur, bt, t0b, -1, L, ts, T, i);
        mark(5);
        const double s = carry_scan<R, NG>(e, fT, s_mat + (size_t)(3 + NLEV) * R * R, c, i, s_a, s_b, true, sync);
        mark(6);
        const double v = fullb ? chunk_run<R, true, 2>(Jp, Jp, s, cur, bt, t0b, -1, L, ts, T, i, nullptr, dummy, fout, r)
                               : chunk_run<R, false, 2>(Jp, Jp, s, cur, bt, t0b, -1, L, ts, T, i, nullptr, dummy, fout, r);
        if (c == clast) s_vec[R + i] = v;
        sync();
        fb = s_vec[R + i];
    }

    mark(7);
    // ---- backward transient: steps ts-1 .. 0 (wave 0 only) --------------------------------------------------------
    if (tid < 64) {
        double v = fb;
        for (int t = ts - 1; t >= 0; --t) {
            double Jp[R];
            load_xperm<R>(Jp, (t < nst ? s_tab + (size_t)t * 3 * R * R : tab_over + (size_t)t * 3 * R * R) + R * R, i);
            const double wt = bt[t * R + i];
            v = matvec_x<R>(Jp, v, wt);
            if (c == 0 && t >= 1 && i < r) fout[(size_t)(t - 1) * r + i] = v;
        }
        if (a.f0s && c == 0) a.f0s[(size_t)b * R + i] = v;   // E[f_0 | X] (EM)
    }

    mark(8);
    // ---- log-likelihood ---------------------------------------------------------------------------------------
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, kWave);
    if ((tid & 63) == 0) s_red[tid >> 6] = dot;
    sync();
    if (tid == 0) {
        double d = 0.0, sq = 0.0;
#pragma unroll
        for (int w = 0; w < kScanThreads / 64; ++w) d += s_red[w];
        for (int w = 0; w < nseg; ++w) sq += ssum[w];
        a.loglik[b] = -0.5 * (llcp[0] + sq - d);
    }
}

#endif  // DFM_DIAG

// ------------------------------------------------------------------------------------------------------------------
// scan_reg (round 3): the same scan with every lane group owning ONE time range in BOTH directions, so that its b_t -> w_t
// never leave its registers, and with the cross-wave hand-overs cut from ~20 LDS-counter barriers per replicate to 2.
//
// At B = 1024 a CU has 4 replicates and the scans form a chain (the first one waits for the covariance chain, each next one
// for the previous): cov + 4 scans ~ 200 us on a fast box and more than the 4 x 47 us the stream needs on a box whose
// latency-bound code runs 20-30 % slower -- the pass then runs at the speed of this chain, not of HBM (0.44 of
// peak on the driver's box, 0.51-0.55 on others).  In-kernel stamps say where a scan's 34 us go (20 us when it runs alone):
// nine phases of 3-5 us, most of them a handful of dependent steps between two barriers on an LDS counter, each barrier
// several LDS round trips behind the stream waves' DMA writes and operand reads.  Here:
//   * group c owns steady steps ts + c L .. ts + c L + L - 1 forward AND backward: b_t is read from LDS once (phase 1),
//     phase 3 overwrites the registers with w_t, the backward runs read those registers -- a quarter of the LDS traffic and
//     no "every w_t is written" barrier between the directions; xi_T / f_T live in the group that produced them;
//   * the forward transient (E - 1 <= kPfEcap steps) is computed by EVERY group redundantly: no broadcast barrier;
//   * chunk carries: ONE barrier, then every group runs the affine maps of the chunks before it sequentially (<= 31
//     dependent 8 x 8 mat-vecs of ~50 cycles, operands prefetched) instead of a 5-level Kogge-Stone with a barrier per level;
//   * backward, the (partial) top chunk starts from its TRUE state f_T, so no power of J for a partial chunk is needed and
//     its rows are emitted in that first run;
//   * the backward transient runs in group 0 right behind its own phase 3 (it owns the lowest range);
//   * epilogue: the waves meet at an arrival counter; the LAST one adds the four partial sums in a fixed order, writes the
//     log-likelihood and releases the b_t buffer / the table set -- nobody waits at a barrier for it.
// Conditions (else scan_lds): L <= kRegL, E - 1 <= kPfEcap (all transient tables in LDS).  Same outputs to rounding (the sum
// xi_t' w_t is accumulated in another order).
// ------------------------------------------------------------------------------------------------------------------
constexpr int kRegL = 16;

template <class Sync, class Prio>
__device__ __forceinline__ void scan_reg(const FastArgs& a, int b, int tid, const double* bt, const double* s_tab, const double* s_mat,
                                         const double* xi0p, const double* llcp, int E, double* sE, double* sEb, double* s_red,
                                         const double* ssum, int nseg, Sync& sync, unsigned* arrive, unsigned arrive_last,
                                         unsigned* scan_done, unsigned done_value, double* wtr, Prio&& prio, double* pslot) {
    auto mark = [&](int k) {
        if (pslot && tid == 0) pslot[k] = (double)__builtin_amdgcn_s_memrealtime();
    };
    constexpr int R = kPfR, NLEV = kPfNlev, NST = kPfNst, LM = kRegL;
    const int c = tid / R, i = tid % R, lane = tid & 63;
    const int T = a.T, r = a.r, L = a.L;
    const int ts = E - 1;                                     // <= kPfEcap
    double* fout = a.f_smooth + (size_t)b * T * r;
    const int clast = (T - ts - 1) / L;                       // group that owns step T - 1 (its chunk may be partial)
    const int t0 = ts + c * L;
    const int Lc = (T - t0) < L ? (T - t0) : L;               // valid steps of this group (<= 0: idle group)
    const int wlo = __builtin_amdgcn_readfirstlane(c & ~7);  // first group of this wave (scalar: uniform loops below)
    double dot = 0.0;

    // ---- forward transient, every group redundantly: xi_ts; group 0 keeps w_t for the backward transient -----------
    double xi = xi0p[i];
#pragma unroll
    for (int t = 0; t < kPfEcap; ++t) {
        if (t < ts) {                                         // (uniform)
            double Zp[R], Gp[R];
            const double* ent = s_tab + (size_t)t * 3 * R * R;
            load_xperm<R>(Zp, ent, i);
            load_xperm<R>(Gp, ent + 2 * R * R, i);
            const double btv = bt[t * R + i];
            const double w = matvec_x<R>(Zp, xi);
            if (c == 0) {
                dot = fma(xi, w, dot);
                wtr[t * R + i] = w;                           // (LDS; read back by the same lanes in the backward transient)
            }
            xi = matvec_x<R>(Gp, xi, btv);
        }
    }
    mark(1);

    // ---- phase 1: b_t of the group's range into registers; chunk run from a zero state ------------------------------
    double u[LM];
#pragma unroll
    for (int j = 0; j < LM; ++j) {
        u[j] = 0.0;
        if (j < L) {                                          // (uniform)
            const bool ok = j < Lc;
            const int tc = ok ? t0 + j : T - 1;               // a row that exists; selected away below
            const double x = bt[tc * R + i];
            u[j] = ok ? x : 0.0;
        }
    }
    double Gp[R];
    load_xperm<R>(Gp, s_mat + 2 * R * R, i);
    {
        double e = 0.0;
#pragma unroll
        for (int j = 0; j < LM; ++j) {
            if (j < L) {
                const double vn = matvec_x<R>(Gp, e, u[j]);
                e = (j < Lc) ? vn : e;
            }
        }
        sE[c * R + i] = e;
    }
    mark(2);
    sync();                                                   // (1 of 2) the chunk end states are in LDS
    prio();                                                   // (is the stream waiting for this scan by now?)
    // ---- state entering the group's chunk: the chunks before it, one after the other ---------------------------------
    double v = xi;
    {
        double ML[R];
        load_xperm<R>(ML, s_mat + 3 * R * R, i);              // G^L
        for (int k0 = 0; k0 <= wlo; k0 += 8) {                // (uniform: blocks of 8 chunks up to this wave's own)
            double ek[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ek[q] = sE[(k0 + q) * R + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double vn = matvec_x<R>(ML, v, ek[q]);
                v = (k0 + q < c) ? vn : v;
            }
        }
    }
    mark(3);
    // ---- phase 3: re-run from the true state; w_t replaces b_t in the registers ---------------------------------------
    {
        double Zp[R];
        load_xperm<R>(Zp, s_mat, i);
#pragma unroll
        for (int j = 0; j < LM; ++j) {
            if (j < L) {
                const bool ok = j < Lc;
                const double w = matvec_x<R>(Zp, v);
                const double vn = matvec_x<R>(Gp, v, u[j]);
                dot = ok ? fma(v, w, dot) : dot;
                u[j] = w;
                v = ok ? vn : v;
            }
        }
    }
    mark(4);
    // ---- terminal (group clast: v = xi_T) + backward run of the chunk -------------------------------------------------
    double Jp[R];
    load_xperm<R>(Jp, s_mat + R * R, i);
    double eb;                                                // state leaving the chunk downwards
    {
        double PTp[R];
        load_xperm<R>(PTp, s_mat + NST * R * R, i);
        const double fT = matvec_x<R>(PTp, v);                // meaningful in group clast only
        const bool top = (c == clast);
        if (top) {
            dot = fma(v, fT, dot);                            // the log-likelihood needs sum xi'w + xi_T' f_T
            if (i < r) fout[(size_t)(T - 1) * r + i] = fT;
        }
        eb = top ? fT : 0.0;                                  // the top chunk starts from its true state, the others from zero
#pragma unroll
        for (int j = LM - 1; j >= 0; --j) {
            if (j < L) {
                const bool ok = j < Lc;
                const double vn = matvec_x<R>(Jp, eb, u[j]);
                eb = ok ? vn : eb;
                const int t = t0 + j;
                if (top && ok && t >= 1 && i < r) fout[(size_t)(t - 1) * r + i] = eb;   // f of period t -> row t - 1
            }
        }
        sEb[c * R + i] = eb;
    }
    mark(5);
    sync();                                                   // (2 of 2)
    prio();
    // ---- backward carries: the true state leaving the top chunk, then the zero-state runs of the chunks in between -----
    double vb = 0.0;
    if (wlo < clast) {                                        // (uniform) this wave has a group below the top chunk
        double JL[R];
        load_xperm<R>(JL, s_mat + (size_t)(3 + NLEV) * R * R, i);   // J^L
        vb = sEb[clast * R + i];
        for (int k0 = clast - 1; k0 > wlo; k0 -= 8) {         // chunks k0, k0 - 1, .. (those above c apply)
            double ek[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int k = k0 - q; ek[q] = sEb[(k > 0 ? k : 0) * R + i]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double vn = matvec_x<R>(JL, vb, ek[q]);
                vb = (k0 - q > c) ? vn : vb;
            }
        }
    }
    mark(6);
    // ---- backward phase 3 (groups below the top chunk; all of them hold full chunks) -------------------------------
    {
        const bool below = c < clast;
#pragma unroll
        for (int j = LM - 1; j >= 0; --j) {
            if (j < L) {
                vb = matvec_x<R>(Jp, vb, u[j]);
                const int t = t0 + j;
                if (below && t >= 1 && i < r) fout[(size_t)(t - 1) * r + i] = vb;
            }
        }
        vb = below ? vb : eb;                                 // group 0 == top chunk (short panels): its first run was the true one
    }
    mark(7);
    // ---- backward transient: steps ts-1 .. 0, group 0 (it owns the lowest range; the other groups' values are unused) --
    if (tid < 64) {                                           // (wave 0)
        double vt = vb;
#pragma unroll
        for (int t = kPfEcap - 1; t >= 0; --t) {
            if (t < ts) {
                double Jt[R];
                load_xperm<R>(Jt, s_tab + (size_t)t * 3 * R * R + R * R, i);
                vt = matvec_x<R>(Jt, vt, wtr[t * R + (tid & 7)]);
                if (c == 0 && t >= 1 && i < r) fout[(size_t)(t - 1) * r + i] = vt;
            }
        }
        if (a.f0s && c == 0) {                                // E[f_0 | X] (EM)
            int ii = tid & 7;
            asm volatile("" : "+v"(ii));                      // address formed here: a pointer kept live across the scan is spilled
            a.f0s[(size_t)b * R + ii] = vt;
        }
    }
    mark(8);
    // ---- log-likelihood + release: the last wave to arrive does both ----------------------------------------------
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, kWave);
    if (lane == 0) s_red[tid >> 6] = dot;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    unsigned before = 0;
    if (lane == 0) before = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    before = __builtin_amdgcn_readfirstlane(before);
    if (before + 1u == arrive_last) {                         // every scan wave is done with bt / the table set / ssum
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane == 0) {
            double d = 0.0, sq = 0.0;
#pragma unroll
            for (int w = 0; w < kPfScanWaves; ++w) d += s_red[w];
            for (int w = 0; w < nseg; ++w) sq += ssum[w];
            a.loglik[b] = -0.5 * (llcp[0] + sq - d);
            __hip_atomic_store(scan_done, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// scan_seq: the plain sequential mean recursion by scan wave 0 (its eight lane groups redundantly, group 0 stores) -- the
// fallback for a replicate whose Riccati transient is longer than the kPfEcap steps scan_reg keeps in LDS (E - 1 > kPfEcap:
// slowly converging covariances; entries past the staged ones come from the workspace table).  ~4x the time of scan_reg for
// that replicate (T dependent steps instead of 2 L + the carries), ~2 KB of code instead of the 36 KB of the round-2 scan_lds
// that used to serve this case inside the production kernel.  The other scan waves only take part in the arrival count.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void scan_seq(const FastArgs& a, int b, int tid, double* bt, const double* s_tab, const double* tab_over,
                                         const double* s_mat, const double* xi0p, const double* llcp, int E, const double* ssum, int nseg,
                                         unsigned* arrive, unsigned arrive_last, unsigned* scan_done, unsigned done_value) {
    constexpr int R = kPfR, NST = kPfNst;
    const int lane = tid & 63, i = tid % R, c = tid / R;
    const int T = a.T, r = a.r;
    const int ts = E - 1;
    const int nst = ts < kPfEcap ? ts : kPfEcap;
    double dot = 0.0;
    if (tid < 64) {
        double* fout = a.f_smooth + (size_t)b * T * r;
        double xi = xi0p[i];
        double Zs[R], Gs[R];
        load_xperm<R>(Zs, s_mat, i);
        load_xperm<R>(Gs, s_mat + 2 * R * R, i);
        for (int t = 0; t < T; ++t) {
            double Zp[R], Gp[R];
            if (t < ts) {                                       // (uniform) transient entry: LDS, or the workspace table
                const double* ent = t < nst ? s_tab + (size_t)t * 3 * R * R : tab_over + (size_t)t * 3 * R * R;
                load_xperm<R>(Zp, ent, i);
                load_xperm<R>(Gp, ent + 2 * R * R, i);
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k) { Zp[k] = Zs[k]; Gp[k] = Gs[k]; }
            }
            const double btv = bt[t * R + i];
            const double w = matvec_x<R>(Zp, xi);
            wave_lds_sync();                                    // every group has read b_t before group 0 overwrites it with w_t
            if (c == 0) { dot = fma(xi, w, dot); bt[t * R + i] = w; }
            xi = matvec_x<R>(Gp, xi, btv);
        }
        double PTp[R];
        load_xperm<R>(PTp, s_mat + NST * R * R, i);
        double v = matvec_x<R>(PTp, xi);                         // f_T
        if (c == 0) {
            dot = fma(xi, v, dot);
            if (i < r) fout[(size_t)(T - 1) * r + i] = v;
        }
        wave_lds_sync();
        double Js[R];
        load_xperm<R>(Js, s_mat + R * R, i);
        for (int t = T - 1; t >= 0; --t) {
            double Jp[R];
            if (t < ts) load_xperm<R>(Jp, (t < nst ? s_tab + (size_t)t * 3 * R * R : tab_over + (size_t)t * 3 * R * R) + R * R, i);
            else {
#pragma unroll
                for (int k = 0; k < R; ++k) Jp[k] = Js[k];
            }
            v = matvec_x<R>(Jp, v, bt[t * R + i]);
            if (c == 0 && t >= 1 && i < r) fout[(size_t)(t - 1) * r + i] = v;
        }
        if (a.f0s && c == 0) a.f0s[(size_t)b * R + i] = v;      // E[f_0 | X] (EM)
        dot = (c == 0) ? dot : 0.0;
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, kWave);
        if (lane == 0) {
            double sq = 0.0;
            for (int w = 0; w < nseg; ++w) sq += ssum[w];
            a.loglik[b] = -0.5 * (llcp[0] + sq - dot);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    unsigned before = 0;
    if (lane == 0) before = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    before = __builtin_amdgcn_readfirstlane(before);
    if (before + 1u == arrive_last && lane == 0)                 // (waves 1-3 arrive at once; whoever is last releases bt / the table set)
        __hip_atomic_store(scan_done, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ------------------------------------------------------------------------------------------------------------------
// Gram matrix C = Lam' R^-1 Lam of one replicate by ONE wave on the fp64 matrix pipe, in the collapse's own operand layout:
// lane (K, g, h, q) holds W[c][4 h + q] = lam_c,4h+q / R_c for the series c = 8 s + 4 g + K (the B operands of the stream
// waves).  Rows i = 4 p + (0..3) of Lam' play the part of the 4 periods of a row block: the A operand of block p is
// lam_c,4p+q -- the lane's own raw loading when p == h, its partner's (lane ^ 4) otherwise.  2 x STEPS MFMAs instead of
// ~400 VALU instructions and two 18-value transpose-reductions; C (not symmetrised) into Cs[8][8].  Returns sum_i log R_i.
// ------------------------------------------------------------------------------------------------------------------
template <int STEPS, int NQ>
__device__ __forceinline__ double gram_mfma8(const double* __restrict__ Lg, const double* __restrict__ Rg, int N, int lane,
                                             double* Cs, double* tstamp = nullptr) {
    constexpr int R = 8, CS = 8;
    const int K = lane >> 4, blk = (lane >> 2) & 3, q = lane & 3;
    const int g = blk >> 1, h = blk & 1;
    const bool tail_clamp = (STEPS - 1) * CS + 4 * g + K >= N;
    const int clast = tail_clamp ? N - 1 : (STEPS - 1) * CS + 4 * g + K;
    const double* __restrict__ Lq = Lg + (4 * g + K) * R + 4 * h + q;
    const double* __restrict__ Rq = Rg + (4 * g + K);
    double raw[STEPS], rv[STEPS];
#pragma unroll
    for (int s = 0; s + 1 < STEPS; ++s) {
        raw[s] = Lq[s * CS * R];
        rv[s] = Rq[s * CS];
    }
    raw[STEPS - 1] = Lg[(size_t)clast * R + 4 * h + q];
    rv[STEPS - 1] = Rg[clast];
    double rl[NQ][2];                                            // the lane's own series (2 l, 2 l + 1) + 128 jq: log R
#pragma unroll
    for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 2 * lane + 128 * jq + e;
            rl[jq][e] = Rg[c < N ? c : N - 1];
        }
    // every load is in flight before anything is consumed: under register pressure the scheduler otherwise pairs each load
    // with its use (a full round trip per step -- 5 to 8 us each beside the streaming waves)
    __builtin_amdgcn_sched_barrier(0);
    if (tstamp) {                                                // diagnostics: when the batch of loads has landed
        wait_vmf<0>();
        if (lane == 0) *tstamp = (double)__builtin_amdgcn_s_memrealtime();
    }
    double ld = 0.0;
#pragma unroll
    for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 2 * lane + 128 * jq + e;
            ld += log(c < N ? rl[jq][e] : 1.0);                  // unconditional call: no exec-masked block per series
        }
    double D0 = 0.0, D1 = 0.0;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const double w = (s == STEPS - 1 && tail_clamp) ? 0.0 : raw[s] * fast_rcp(rv[s]);   // the stream waves' weights exactly
        const double other = xor_lane<4>(raw[s]);
        const double a0 = h == 0 ? raw[s] : other;           // lam_c,q
        const double a1 = h == 0 ? other : raw[s];           // lam_c,4+q
        D0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, w, D0, 0, 0, 0);
        D1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, w, D1, 0, 0, 0);
    }
    D0 += xor_lane<8>(D0);                                       // fold the two series groups
    D1 += xor_lane<8>(D1);
    if (g == 0) {                                                // lane (K, h, q): C[4 p + K][4 h + q]
        Cs[K * R + 4 * h + q] = D0;
        Cs[(4 + K) * R + 4 * h + q] = D1;
    }
    return wave_allsum(ld);
}

// ------------------------------------------------------------------------------------------------------------------
template <int STEPS, int NDR>
__global__ __launch_bounds__(kPfMaxThreads) void pass_fused_kernel(CollapseArgs a, FastArgs fa, unsigned SB, PfLds ly) {
    constexpr int R = kPfR;
    constexpr int NB = 2, NS = 4 * NB;                       // row blocks / row slots of a wave's ring
    constexpr int CS = 8;                                    // series per MFMA step at R = 8: 2 series groups x 2 factor groups
    constexpr int NQ = NDR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid_wg = threadIdx.x;
    const int lane = tid_wg & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid_wg >> 6);
    const int nsw = ly.nsw, ncov = ly.ncov, nbuf = ly.nbuf;
    const bool dma_nt = ly.nt != 0;
    unsigned* flags = reinterpret_cast<unsigned*>(smem + ly.flags);
    double* misc = reinterpret_cast<double*>(smem + ly.misc);
    double* bt0 = reinterpret_cast<double*>(smem + ly.bt);
    const int N = a.N, T = a.T, B = a.B;
    const int G = (int)gridDim.x;
    const int nrep_wg = (B - (int)blockIdx.x + G - 1) / G;   // replicates of this workgroup
    // P_smooth fills deferred to the tail (see the mover)
#ifdef DFM_DIAG   // DFM_SCAN_ABL bits 12-14 = count + 1
    const int ndefer = (fa.P_smooth == nullptr) ? 0 : (((fa.abl >> 12) & 7) ? ((fa.abl >> 12) & 7) - 1 : kPfDeferDefault);
#else
    const int ndefer = (fa.P_smooth == nullptr) ? 0 : kPfDeferDefault;
#endif

    if (tid_wg < kFCount) flags[tid_wg] = 0u;
    __syncthreads();                                         // the only workgroup barrier of the kernel

#ifdef DFM_DIAG
    // DFM_SCAN_ABL bit 8: s_memrealtime stamps (10 ns ticks) of the phases of every replicate into scol[b][0..31]
    const bool prof = (fa.abl & 256) != 0 && a.scol != nullptr && T >= 40;
    auto stamp = [&](int bb, int slot) {
        if (prof && lane == 0) a.scol[(size_t)bb * T + slot] = (double)__builtin_amdgcn_s_memrealtime();
    };

    // (diagnostics) where the hardware put this wave: HW_REG_HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SE [15:13])
    if (prof) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        for (int bb = (int)blockIdx.x; bb < B; bb += G) {
            if (lane == 0) a.scol[(size_t)bb * T + 40 + wave] = (double)hw;
        }
    }
#else   // the production kernel carries no stamp code
    constexpr bool prof = false;
    auto stamp = [](int, int) {};
#endif

    if (wave < nsw) {
        // ================= STREAM: segment [ta, tb) of every replicate -> bt[buf][t][0..7], ssum[buf][wave] ==========
        // The wave's work is ONE sequence of row blocks (4 periods each) over all its replicates: the ring is re-armed
        // with the next replicate's first rows while the last rows of the current one are consumed, so the DMA stream
        // never drains at a replicate boundary.  Every block issues exactly 4 x NDR DMA instructions (rows past the
        // segment repeat its last row), which keeps the counted vmcnt wait valid across boundaries.
        const unsigned rowB = (unsigned)N * 8u;
        // lane roles of v_mfma_f64_4x4x4 (collapse_mfma.hip)
        const int K = lane >> 4, blk = (lane >> 2) & 3, q = lane & 3;
        const int g = blk >> 1, h = blk & 1;
        int tq = (T + nsw - 1) / nsw;
        {
            unsigned gg = rowB & 127u;
            gg = gg == 0 ? 128u : (gg & (~gg + 1u));
            const int m = (int)(128u / gg);
            tq = ((tq + m - 1) / m) * m;                         // segments start on 128-byte boundaries
        }
        const int ta = (wave * tq < T) ? wave * tq : T;
        const int tb = (ta + tq < T) ? ta + tq : T;
        const int nrows = tb - ta;
        const int nblk = (nrows + 3) / 4;
        const unsigned ringB = NS * SB;
        const char* ring = smem + ly.ring + (size_t)wave * ringB;
        const unsigned ring_lds =
            __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_f)(smem)) + ly.ring + (unsigned)wave * ringB;
        const unsigned lane16 = 16u * lane;
        bool pact[NDR];                                          // lane moves 16 bytes of piece p of a row
#pragma unroll
        for (int p = 0; p < NDR; ++p) pact[p] = lane16 + 1024u * p < rowB;
        const unsigned lane_off = (unsigned)q * SB + (unsigned)(4 * g + K) * 8u;
        const bool tail_clamp = (STEPS - 1) * CS + 4 * g + K >= N;
        const unsigned last_off = tail_clamp ? (unsigned)q * SB + (unsigned)(N - 1) * 8u : lane_off + (unsigned)(STEPS - 1) * (CS * 8u);
        const int nrep = (B - (int)blockIdx.x + G - 1) / G;      // replicates of this workgroup
        const int gtot = nrep * nblk;                            // row blocks of this wave

        double Bw[STEPS];                                        // B operands: lam_cf / R_c for c = s CS + 4 g + K, f = 4 h + q
        double rown[NQ][2];                                      // 1 / R of the lane's own 16-byte column pairs (s_t pass)
        // block k of local replicate jj into ring block `bslot`: 4 rows x NDR DMA instructions, always
        auto issue_block = [&](int jj, int k, int bslot) {
            const char* seg = reinterpret_cast<const char*>(a.panel + ((size_t)((int)blockIdx.x + jj * G) * T + ta) * N);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                int ri = 4 * k + rr;
                ri = ri < nrows ? ri : nrows - 1;
                const char* src = seg + (size_t)ri * rowB + lane16;
                const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(bslot * 4 + rr) * SB);
#pragma unroll
                for (int p = 0; p < NDR; ++p) {
                    if (pact[p]) dma16f(src + 1024 * p, dst + 1024u * p, dma_nt);
                }
            }
        };
        // weights of replicate bb: loads off one base pointer with constant offsets (no per-load address registers),
        // reciprocal by v_rcp_f64 + two Newton steps.  Ends with vmcnt(0): the ring blocks in flight have landed too.
        auto load_weights = [&](int bb) {
            const double* __restrict__ Lq = a.Lam + (size_t)bb * N * R + (4 * g + K) * R + 4 * h + q;
            const double* __restrict__ Rq = a.Rv + (size_t)bb * N + (4 * g + K);
            const int clast = tail_clamp ? N - 1 : (STEPS - 1) * CS + 4 * g + K;
            double rv[STEPS];
#pragma unroll
            for (int s = 0; s + 1 < STEPS; ++s) {
                Bw[s] = Lq[s * CS * R];
                rv[s] = Rq[s * CS];
            }
            Bw[STEPS - 1] = a.Lam[(size_t)bb * N * R + (size_t)clast * R + 4 * h + q];
            rv[STEPS - 1] = a.Rv[(size_t)bb * N + clast];
            const double* __restrict__ Rg = a.Rv + (size_t)bb * N;
#pragma unroll
            for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int c = 2 * lane + 128 * jq + e;
                    rown[jq][e] = Rg[c < N ? c : N - 1];
                }
            __builtin_amdgcn_sched_barrier(0);                    // all loads in flight before the first use
            wait_vmf<0>();
#pragma unroll
            for (int s = 0; s + 1 < STEPS; ++s) Bw[s] *= fast_rcp(rv[s]);
            Bw[STEPS - 1] = tail_clamp ? 0.0 : Bw[STEPS - 1] * fast_rcp(rv[STEPS - 1]);
#pragma unroll
            for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int c = 2 * lane + 128 * jq + e;
                    rown[jq][e] = (c < N) ? fast_rcp(rown[jq][e]) : 0.0;
                }
        };

        if (gtot > 0) {
            // prologue: the first NB blocks
            int ij = 0, ik = 0;                                  // next block to issue
            auto advance_issue = [&]() { ++ik; if (ik == nblk) { ik = 0; ++ij; } };
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (u < gtot) { issue_block(ij, ik, u); advance_issue(); }
            }
            int bslot = 0;
            double qa[NQ][2];
            double* bt = bt0;
            int buf = 0;
            int gidx = 0;
            for (int j = 0; j < nrep; ++j) {
                const int b = (int)blockIdx.x + j * G;
                if (wave == 0) stamp(b, 0);
                load_weights(b);                                  // (drains the vm counter)
                buf = nbuf == 2 ? (j & 1) : 0;
                bt = bt0 + (size_t)buf * ly.bt_stride;
                // the b_t buffer is free once the scan of the replicate that used it last is complete
                if (j >= nbuf) {
                    const bool must_wait = wave == 0 && (int)(ld_flag(flags + kFScanDone) - (unsigned)(j - nbuf + 1)) < 0;
                    if (must_wait && lane == 0) __hip_atomic_store(flags + kFStreamWaits, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (!pf_wait_ge(flags + kFScanDone, (unsigned)(j - nbuf + 1), flags + kFAbort)) { wait_vmf<0>(); return; }
                    if (must_wait && lane == 0) __hip_atomic_store(flags + kFStreamWaits, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (wave == 0) stamp(b, 1);
                stamp(b, 32 + wave);                              // per-wave segment start / end (diagnostics)
#pragma unroll
                for (int jq = 0; jq < NQ; ++jq) { qa[jq][0] = 0.0; qa[jq][1] = 0.0; }
                for (int k = 0; k < nblk; ++k, ++gidx) {
                    const int r0 = k * 4;
                    const bool full = r0 + 4 <= nrows;           // (wave-uniform)
                    // rows of this block have landed once at most the DMAs of the NB - 1 YOUNGER blocks are outstanding
                    // (one in-order vmcnt counter per wave); at the tail of the sequence there are fewer younger blocks
                    if (gidx + NB - 1 < gtot) wait_vmf<((NB - 1) * 4 * NDR <= 63 ? (NB - 1) * 4 * NDR : 63)>();
                    else wait_vmf<0>();
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
                        for (int jq = 0; jq < NQ; ++jq)
                            xq[rr][jq] = pact[jq] ? *reinterpret_cast<const double2*>(pq + (unsigned)rr * SB + 1024u * jq)
                                                  : make_double2(0.0, 0.0);
                    wait_lgkmf();                                        // the reads are done before the slots are re-armed
                    if (ij < nrep) { issue_block(ij, ik, bslot); advance_issue(); }
                    double D = 0.0, D2 = 0.0;                            // two accumulators: half the dependent MFMA chain
#pragma unroll
                    for (int s = 0; s < STEPS; ++s) {
                        if ((s & 1) == 0) D = __builtin_amdgcn_mfma_f64_4x4x4f64(xa[s], Bw[s], D, 0, 0, 0);
                        else D2 = __builtin_amdgcn_mfma_f64_4x4x4f64(xa[s], Bw[s], D2, 0, 0, 0);
                    }
                    D += D2;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        if (full || r0 + rr < nrows) {                   // wave-uniform: rows past the segment repeat its last row
#pragma unroll
                            for (int jq = 0; jq < NQ; ++jq) {
                                qa[jq][0] = fma(xq[rr][jq].x, xq[rr][jq].x, qa[jq][0]);
                                qa[jq][1] = fma(xq[rr][jq].y, xq[rr][jq].y, qa[jq][1]);
                            }
                        }
                    }
                    D += xor_lane<8>(D);                                 // fold the two series groups
                    // lane (K = period, g = 0, h, q) holds factor 4 h + q of period r0 + K: the even-q lane stores (f, f + 1)
                    const double hi = xor_lane<1>(D);
                    const int t = ta + r0 + K;
                    if (g == 0 && (q & 1) == 0 && t < tb)
                        *reinterpret_cast<double2*>(&bt[(size_t)t * R + 4 * h + q]) = make_double2(D, hi);
                    bslot = (bslot + 1 == NB) ? 0 : bslot + 1;
                }
                double sp = 0.0;
#pragma unroll
                for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
                    for (int e = 0; e < 2; ++e) sp = fma(qa[jq][e], rown[jq][e], sp);
                sp = wave_allsum(sp);
                if (lane == 0) {
                    misc[kMiscSsum + buf * 8 + wave] = sp;
                    if (sp != sp) atomicOr(a.status, 1);             // NaN in the panel on the balanced path
                }
                // this wave's part of b_t is in LDS.  (The release waits for the LDS writes only in effect: the DMAs in
                // flight belong to the next replicate, so no full vmcnt wait here -- an explicit lgkmcnt wait instead.)
                wait_lgkmf();
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) __hip_atomic_fetch_add(flags + kFBtReady + buf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (wave == 0) stamp(b, 2);
                if (wave == nsw - 1) stamp(b, 3);
                stamp(b, 24 + wave);
            }
            // ---- the deferred P_smooth fills: the panel is through, HBM is idle but for the last scan's outputs ----------
            if (ndefer > 0) {
                wait_vmf<0>();
                const int nact = (T + tq - 1) / tq;               // stream waves that own periods (they all get here)
                const int npr = fa.r * (fa.r + 1) / 2;
                double* ps = reinterpret_cast<double*>(const_cast<char*>(ring));   // this wave's ring is free now
                for (int j = (nrep - ndefer > 0 ? nrep - ndefer : 0); j < nrep; ++j) {
                    const int b = (int)blockIdx.x + j * G;
                    // (the covariance wave of replicate j published these long ago; the mover has raised tab_ready past j)
                    if (!pf_wait_ge(flags + kFTabReady, (unsigned)(j + 1), flags + kFAbort)) break;
                    const int flo = __builtin_amdgcn_readfirstlane(ld_dev(fa.fill + 2 * b));
                    const int fhi = __builtin_amdgcn_readfirstlane(ld_dev(fa.fill + 2 * b + 1));
                    if (fhi <= flo) continue;
                    int ri = 0;                                   // packed (caller's r) copy of P_s,inf
                    const int lv = lane < npr ? lane : 0;
                    while ((ri + 1) * (ri + 2) / 2 <= lv) ++ri;
                    const double psv = ld_dev(fa.PsInf + (size_t)b * 64 + ri * R + (lv - ri * (ri + 1) / 2));
                    wave_lds_sync();
                    if (lane < npr) ps[lane] = psv;
                    wave_lds_sync();
                    pf_fill_rows(fa.P_smooth + (size_t)b * T * npr, flo, fhi, npr, ps, lane, wave, nact);
                }
            }
        } else {
            // a wave without periods (T shorter than the segments): it only keeps the arrival counts
            for (int j = 0; j < nrep; ++j) {
                const int buf = nbuf == 2 ? (j & 1) : 0;
                if (j >= nbuf) {
                    if (!pf_wait_ge(flags + kFScanDone, (unsigned)(j - nbuf + 1), flags + kFAbort)) return;
                }
                if (lane == 0) {
                    misc[kMiscSsum + buf * 8 + wave] = 0.0;
                }
                pf_signal(flags + kFBtReady + buf, lane);
            }
        }
    } else if (wave < nsw + ncov) {
        // ================= COV: Gram matrix, covariance recursion, transient rows of P_smooth -- ahead of the stream ====
        const int cw = wave - nsw;
        double* ws = reinterpret_cast<double*>(smem + ly.covws) + (size_t)cw * kCov8ScratchDoubles;
        double* Cs = ws + 5 * kCov8TileDoubles;                  // Gram matrix
        // a latency chain (ahead of the stream after the first replicates).  (Round 3: dropping to priority 0 after the
        // workgroup's first replicate -- the only one whose chain somebody waits for -- measured neutral: the stream's first
        // two rounds are slower because the covariance waves' work shares the CU, not because of their priority.)
        __builtin_amdgcn_s_setprio(2);
        for (int b = (int)blockIdx.x + cw * G; b < B; b += ncov * G) {
            stamp(b, 6);
            const double ld = gram_mfma8<STEPS, NDR>(fa.Lam + (size_t)b * N * R, fa.Rv + (size_t)b * N, N, lane, Cs,
                                                     prof ? a.scol + (size_t)b * T + 36 : nullptr);   // (null in the production build)
            wave_lds_sync();
            const double Cel = 0.5 * (Cs[lane] + Cs[(lane & 7) * 8 + (lane >> 3)]);   // exactly symmetric
            wave_lds_sync();
            __builtin_amdgcn_sched_barrier(0);
            Cov8Dst o;
            o.tab = fa.tab + (size_t)b * T * 3 * 64; o.tab_cap = T; o.tab_over = o.tab;
            o.stead = fa.stead + (size_t)b * kPfNst * 64;
            o.PT = fa.PT + (size_t)b * 64; o.xi0 = fa.xi0 + (size_t)b * 8; o.llc = fa.llc + b; o.E = fa.E + b;
            o.fill = fa.fill + 2 * b; o.PsInf = fa.PsInf + (size_t)b * 64;
            o.SP11 = fa.SP11 ? fa.SP11 + (size_t)b * 64 : nullptr;
            o.SU = fa.SP11 ? fa.SU + (size_t)b * 64 : nullptr;
            o.P0s = fa.SP11 ? fa.P0s + (size_t)b * 64 : nullptr;
            stamp(b, 7);                                          // Gram done
            cov_wave8<kPfNlev>(fa, b, Cel, ld, ws, o, lane);
            wave_lds_sync();
            stamp(b, 8);                                          // covariance recursion done
            // the tables are complete: the scan of this replicate may start (its waves also write the fixed-point rows
            // of P_smooth -- 144 KB of pure stores that would hold this latency chain up for 15 us)
            pf_signal(flags + kFCovDone + cw, lane);
        }
    } else if (wave == nsw + ncov) {
        // ================= MOVER: tables of replicate j from the workspace into LDS, one replicate ahead of the scan =====
        // Under the streaming load a global round trip of this CU takes 5-8 us (its requests queue behind the DMA loads), and
        // the scan group needed two of them per replicate on its critical path.  This wave takes them off it: as soon as a
        // covariance wave has published replicate j it loads the tables into REGISTERS (one matrix element per lane: 32
        // matrices = 64 VGPRs), waits for the scan of replicate j - 1 to release the single LDS table set, writes it and
        // raises tab_ready.  It also writes the fixed-point rows of P_smooth (144 KB of pure stores per replicate).
        double* s_mat = reinterpret_cast<double*>(smem + ly.smat);
        double* s_tab = reinterpret_cast<double*>(smem + ly.ctab);
        double* s_ps = misc + kMiscPs;
        const int nfix0 = kPfEcap < T ? kPfEcap : T;
        __builtin_amdgcn_s_setprio(1);
        int b = blockIdx.x;
        for (int j = 0; b < B; b += G, ++j) {
            if (!pf_wait_ge(flags + kFCovDone + (j % ncov), (unsigned)(j / ncov + 1), flags + kFAbort)) break;
            stamp(b, 4);
            const double* stead = fa.stead + (size_t)b * kPfNst * 64;
            const double* tab = fa.tab + (size_t)b * T * 3 * 64;
            double ms[kPfNst + 1], mt[kPfEcap * 3];
#pragma unroll
            for (int k = 0; k < kPfNst; ++k) ms[k] = stead[k * 64 + lane];
            ms[kPfNst] = fa.PT[(size_t)b * 64 + lane];
#pragma unroll
            for (int k = 0; k < kPfEcap * 3; ++k) mt[k] = (k < nfix0 * 3) ? tab[k * 64 + lane] : 0.0;
            // one batch: every load is issued before the first one is consumed (a round trip is 5-8 us here)
            const int Ev = ld_dev(fa.E + b);
            // xi0 [8] then llc: adjacent in neither array, so two loads under lane predicates folded into one select
            const double xiv = ld_dev(fa.xi0 + (size_t)b * R + (lane & 7));
            const double llv = ld_dev(fa.llc + b);
            const int npr = fa.r * (fa.r + 1) / 2;
            int flo = 0, fhi = 0;
            double psv = 0.0;
            if (fa.P_smooth) {
                flo = ld_dev(fa.fill + 2 * b);
                fhi = ld_dev(fa.fill + 2 * b + 1);
                int ri = 0;                                       // packed (caller's r) copy of P_s,inf
                const int lv = lane < npr ? lane : 0;
                while ((ri + 1) * (ri + 2) / 2 <= lv) ++ri;
                psv = ld_dev(fa.PsInf + (size_t)b * 64 + ri * R + (lv - ri * (ri + 1) / 2));
            }
            __builtin_amdgcn_sched_barrier(0);
            const int E = __builtin_amdgcn_readfirstlane(Ev);
            const double xl = lane < R ? xiv : llv;
            const int fill_lo = __builtin_amdgcn_readfirstlane(flo), fill_hi = __builtin_amdgcn_readfirstlane(fhi);
            // the LDS table set is free once the scan of the previous replicate is complete
            if (j >= 1) {
                if (!pf_wait_ge(flags + kFScanDone, (unsigned)j, flags + kFAbort)) break;
            }
#pragma unroll
            for (int k = 0; k <= kPfNst; ++k) s_mat[k * 64 + lane] = ms[k];
#pragma unroll
            for (int k = 0; k < kPfEcap * 3; ++k) s_tab[k * 64 + lane] = mt[k];
            if (lane <= R) misc[kMiscXi0 + lane] = xl;            // xi0 [8], then llc
            if (lane == 0) misc[kMiscE] = (double)E;
            pf_signal(flags + kFTabReady, lane);
            stamp(b, 5);
            // rows [lo, hi) of P_smooth equal the backward fixed point: fire-and-forget 16-byte stores.  The fills of the
            // workgroup's LAST replicates are left to the stream waves, who write them once their last row has been issued:
            // HBM is the bottleneck while the panel streams and idle during the scan-only tail of the kernel
            if (fill_hi > fill_lo && j < nrep_wg - ndefer) {
                if (lane < npr) s_ps[lane] = psv;
                wave_lds_sync();
                pf_fill_rows(fa.P_smooth + (size_t)b * T * npr, fill_lo, fill_hi, npr, s_ps, lane, 0, 1);
                wave_lds_sync();
            }
        }
    } else if (wave < nsw + ncov + 1 + kPfScanWaves) {
        // ================= SCAN: the mean recursion of every replicate, one behind the stream ==========================
        const int tid = tid_wg - 64 * (nsw + ncov + 1);
        double* s_mat = reinterpret_cast<double*>(smem + ly.smat);
        double* s_tab = reinterpret_cast<double*>(smem + ly.ctab);
        double* s_a = reinterpret_cast<double*>(smem + ly.sa);
        double* s_b = reinterpret_cast<double*>(smem + ly.sb);
        PfScanSync sync{flags + kFScanBar, flags + kFAbort, 0u, lane};
#ifdef DFM_DIAG
        const bool use_reg = fa.L <= kRegL && (fa.abl & 512) == 0;    // DFM_SCAN_ABL bit 9: the round-2 scan (A/B)
        const int prio_mode = (fa.abl & 1024) ? 1 : (fa.abl & 2048) ? 2 : 0;   // bit 10: always 3; bit 11: always 0
#else
#ifndef DFM_PF_PRIO_MODE
#define DFM_PF_PRIO_MODE 0
#endif
        constexpr int prio_mode = DFM_PF_PRIO_MODE;               // (development A/B: 1 = always 3, 2 = always 0; the launcher admits L <= kRegL only: pass_fused_supported)
#endif
        // Priority.  The scan is the pipeline's latency chain only while somebody waits for it: the first replicate of the
        // workgroup (everything behind it waits), or a stream that has run out of b_t buffers.  Otherwise it runs BESIDE a
        // stream that is the bottleneck, and at high priority its bursts delay the stream waves' DMA issue (measured at
        // B = 8192: -4 % with the faster scan at priority 3 throughout).
        auto set_prio = [&](int j) {
            const bool hot = prio_mode == 1 || (prio_mode == 0 && (j == 0 || ld_flag(flags + kFStreamWaits) != 0)) ||
                             (prio_mode == 3 && ld_flag(flags + kFStreamWaits) != 0) || (prio_mode == 4 && j == 0);
            if (hot) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(0);
        };
        __builtin_amdgcn_s_setprio(3);
        int b = blockIdx.x;
        for (int j = 0; b < B; b += G, ++j) {
            const int buf = nbuf == 2 ? (j & 1) : 0;
            double* bt = bt0 + (size_t)buf * ly.bt_stride;
            if (tid == 0) stamp(b, 10);
            if (!pf_wait_ge(flags + kFTabReady, (unsigned)(j + 1), flags + kFAbort)) break;
            if (tid == 0) stamp(b, 11);
            if (!pf_wait_ge(flags + kFBtReady + buf, (unsigned)(nsw * (j / nbuf + 1)), flags + kFAbort)) break;
            if (tid == 0) stamp(b, 12);
            set_prio(j);
            const int E = __builtin_amdgcn_readfirstlane((int)misc[kMiscE]);
#ifdef DFM_DIAG
            if (!use_reg) {
                scan_lds(fa, b, tid, bt, s_tab, fa.tab + (size_t)b * T * 3 * 64, s_mat, misc + kMiscXi0, misc + kMiscLlc, E, s_a, s_b,
                         misc + kMiscVec, misc + kMiscRed, misc + kMiscSsum + buf * 8, nsw, sync,
                         (prof && tid == 0) ? a.scol + (size_t)b * T + 13 : nullptr);
                sync();                       // the scan is done with bt / the table set / ssum
                if (lane == 0) __hip_atomic_fetch_add(flags + kFScanArrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (tid == 0) {
                    __hip_atomic_store(flags + kFScanDone, (unsigned)(j + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    stamp(b, 22);
                }
                continue;
            }
            double* const pslot = (prof && tid == 0) ? a.scol + (size_t)b * T + 13 : nullptr;
#else
            double* const pslot = nullptr;
#endif
            // (the arrival counter counts 4 per replicate on EVERY path, so they can alternate between replicates)
            if (E - 1 <= kPfEcap) {
                scan_reg(fa, b, tid, bt, s_tab, s_mat, misc + kMiscXi0, misc + kMiscLlc, E, s_a, s_b, misc + kMiscRed,
                         misc + kMiscSsum + buf * 8, nsw, sync, flags + kFScanArrive, (unsigned)(kPfScanWaves * (j + 1)),
                         flags + kFScanDone, (unsigned)(j + 1), misc + kMiscWtr, [&]() { set_prio(j); }, pslot);
            } else {                          // a Riccati transient longer than the staged tables: the sequential fallback
#ifdef DFM_PF_FALLBACK_LDS                    // (development A/B: the round-2 scan as the fallback)
                scan_lds(fa, b, tid, bt, s_tab, fa.tab + (size_t)b * T * 3 * 64, s_mat, misc + kMiscXi0, misc + kMiscLlc, E, s_a, s_b,
                         misc + kMiscVec, misc + kMiscRed, misc + kMiscSsum + buf * 8, nsw, sync, nullptr);
                sync();
                if (lane == 0) __hip_atomic_fetch_add(flags + kFScanArrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (tid == 0) __hip_atomic_store(flags + kFScanDone, (unsigned)(j + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                continue;
#endif
                scan_seq(fa, b, tid, bt, s_tab, fa.tab + (size_t)b * T * 3 * 64, s_mat, misc + kMiscXi0, misc + kMiscLlc, E,
                         misc + kMiscSsum + buf * 8, nsw, flags + kFScanArrive, (unsigned)(kPfScanWaves * (j + 1)),
                         flags + kFScanDone, (unsigned)(j + 1));
            }
            if (tid == 0) stamp(b, 22);
        }
    }
    if (lane == 0 && ld_flag(flags + kFAbort) != 0) atomicOr(a.status, 4);   // a bounded wait ran out
}

// ------------------------------------------------------------------------------------------------------------------
static PfLds pf_layout(int T, int N, int nsw, int ncov, int nbuf) {
    PfLds l;
    unsigned off = 0;
    auto take = [&](unsigned bytes) { const unsigned at = off; off += (bytes + 255u) & ~255u; return at; };
    l.flags = take(kFCount * 4);
    l.smat = take((kPfNst + 1) * 64 * 8);
    l.ctab = take(kPfEcap * 3 * 64 * 8);
    l.misc = take(kMiscDoubles * 8);
    l.covws = take((unsigned)ncov * kCov8ScratchDoubles * 8);
    l.sa = take(32 * 8 * 8);
    l.sb = take(32 * 8 * 8);
    const unsigned btb = (unsigned)((T + 3) / 4 * 4) * 8 * 8;
    l.bt_stride = ((btb + 255u) & ~255u) / 8;
    l.bt = take(l.bt_stride * 8 * (unsigned)nbuf);
    l.ring = take((unsigned)nsw * 8u * pf_slot_bytes(N));
    l.total = off;
    l.nsw = nsw; l.ncov = ncov; l.nbuf = nbuf; l.nt = 0;
    return l;
}

// Stream waves, covariance waves and b_t buffers that fit 160 KB of LDS and 12 waves: double-buffered b_t with as many
// stream waves as fit (at least 3), else one buffer (the stream then waits for the scan of the previous replicate).
static PfLds pf_pick(int T, int N, int want_nsw, int want_ncov) {
    int ncov = want_ncov > 0 ? want_ncov : kPfMaxCov;
    if (ncov > kPfMaxCov) ncov = kPfMaxCov;
    int cap = kPfMaxWaves - kPfScanWaves - 1 - ncov;        // one mover wave
    // default: one stream wave per SIMD -- a second one on a SIMD is starved by the oldest-first arbitration (its segment
    // ends 20 us after the others') and the per-CU streaming rate is the same with 4 rings as with 5 (measured: B = 1024
    // 0.233 vs 0.245 ms, B = 8192 1.62 vs 1.68 ms)
    if (want_nsw <= 0) want_nsw = 4;
    if (want_nsw < cap) cap = want_nsw;
    while (cap > 1 && T / cap < 8) --cap;                     // keep segments a few row blocks long
    for (int nbuf = 2; nbuf >= 1; --nbuf) {
        for (int nsw = cap; nsw >= (nbuf == 2 ? (cap < 3 ? cap : 3) : 1); --nsw) {
            const PfLds l = pf_layout(T, N, nsw, ncov, nbuf);
            if (l.total <= kPfLdsLimit) return l;
        }
    }
    PfLds none = pf_layout(T, N, 1, ncov, 1);
    return none;
}

int pass_fused_pick_nsw(int T, int N, int want) { return pf_pick(T, N, want, 0).nsw; }

// Rp = 8, the shapes of the MFMA collapse (even N, 8N <= 4096, ceil(N / 8) <= 32 steps), T below the LDS limit
bool pass_fused_supported(int Rpad, int T, int N) {
    if (Rpad != 8 || !collapse_mfma_supported(8, N)) return false;
    if (T < 2) return false;
#ifndef DFM_DIAG
    // scan_reg keeps a lane group's chunk of L periods in registers (L <= kRegL: T <= 512); longer panels take the two-launch
    // pass (collapse_mfma_kernel + meanscan_kernel).  The diagnostics build still has the round-2 scan_lds for them.
    if (fast_chunk_len(8, T) > kRegL) return false;
#endif
    return pf_pick(T, N, 0, 0).total <= kPfLdsLimit;
}

template <int STEPS, int NDR>
static hipError_t launch_pf_one(const CollapseArgs& a, const FastArgs& fa, int nsw, int ncov, int num_cu, hipStream_t s) {
    PfLds ly = pf_pick(a.T, a.N, nsw, ncov);
    if (ly.total > kPfLdsLimit) return hipErrorInvalidValue;
    ly.nt = stream_nt_hint((size_t)a.B * a.T * a.N * sizeof(double)) ? 1 : 0;
    static LdsOptIn attr_done;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_fused_kernel<STEPS, NDR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPfLdsLimit);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int grid = a.B < num_cu ? a.B : num_cu;
    const int nwaves = ly.nsw + ly.ncov + 1 + kPfScanWaves;
    hipLaunchKernelGGL((pass_fused_kernel<STEPS, NDR>), dim3(grid), dim3(64 * nwaves), ly.total, s, a, fa, pf_slot_bytes(a.N), ly);
    return hipGetLastError();
}

template <int S>
static hipError_t launch_pf_pick(const CollapseArgs& a, const FastArgs& fa, int nsw, int ncov, int num_cu, hipStream_t s, int steps) {
    if constexpr (S > 32) {
        return hipErrorInvalidValue;
    } else {
#ifdef DFM_PF_ONLY            // development builds: one instantiation (ISA inspection in seconds instead of minutes)
        if constexpr (S == DFM_PF_ONLY)
#endif
        if (steps == S) {
            const int ndr = (a.N * 8 + 1023) / 1024;
            constexpr int lo = (8 * (S - 1) * 8 + 8 + 1023) / 1024, hi = (8 * S * 8 + 1023) / 1024;
            if constexpr (lo <= 1 && 1 <= hi) { if (ndr == 1) return launch_pf_one<S, 1>(a, fa, nsw, ncov, num_cu, s); }
            if constexpr (lo <= 2 && 2 <= hi) { if (ndr == 2) return launch_pf_one<S, 2>(a, fa, nsw, ncov, num_cu, s); }
            if constexpr (lo <= 3 && 3 <= hi) { if (ndr == 3) return launch_pf_one<S, 3>(a, fa, nsw, ncov, num_cu, s); }
            if constexpr (lo <= 4 && 4 <= hi) { if (ndr == 4) return launch_pf_one<S, 4>(a, fa, nsw, ncov, num_cu, s); }
            return hipErrorInvalidValue;
        }
        return launch_pf_pick<S + 1>(a, fa, nsw, ncov, num_cu, s, steps);
    }
}

hipError_t launch_pass_fused(const CollapseArgs& a, const FastArgs& fa, int nsw, int ncov, int num_cu, hipStream_t s) {
    return launch_pf_pick<1>(a, fa, nsw, ncov, num_cu, s, (a.N + 7) / 8);
}

// ------------------------------------------------------------------------------------------------------------------
// cov_wave_kernel: dfm_cov8.h's one-wave-per-replicate covariance recursion as a drop-in for cov_kernel (same global
// outputs) -- DFM_COV_WAVE=1 on the separate-launch path; diagnostics and A/B of the recursion itself.
// ------------------------------------------------------------------------------------------------------------------
#ifdef DFM_DIAG
__global__ __launch_bounds__(256) void cov_wave_kernel(FastArgs a) {
    __shared__ __attribute__((aligned(16))) double wsm[4 * kCov8ScratchDoubles];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wv;
    if (b >= a.B) return;
    __builtin_amdgcn_s_setprio(3);
    double* ws = wsm + wv * kCov8ScratchDoubles;
    const double Cel = a.Cfull[(size_t)b * 64 + lane];
    Cov8Dst o;
    o.tab = a.tab + (size_t)b * a.T * 3 * 64; o.tab_cap = a.T; o.tab_over = o.tab;
    o.stead = a.stead + (size_t)b * stead_mats(8) * 64;
    o.PT = a.PT + (size_t)b * 64; o.xi0 = a.xi0 + (size_t)b * 8; o.llc = a.llc + b; o.E = a.E + b; o.fill = a.fill + 2 * b;
    o.PsInf = a.PsInf + (size_t)b * 64;
    o.SP11 = a.SP11 ? a.SP11 + (size_t)b * 64 : nullptr;
    o.SU = a.SP11 ? a.SU + (size_t)b * 64 : nullptr;
    o.P0s = a.SP11 ? a.P0s + (size_t)b * 64 : nullptr;
    cov_wave8<scan_levels(8)>(a, b, Cel, a.ldfull[b], ws, o, lane);
}
#endif  // DFM_DIAG

// ------------------------------------------------------------------------------------------------------------------
// cov_grid_kernel: the same element-per-thread covariance recursion for states 9..16 / 17..32 wide -- a workgroup of
// R x R threads per replicate (dfm_grid.h: what crosses waves goes through small LDS buffers, one barrier per exchange).
// cov_kernel (fastpath.hip) gives a replicate R lanes with a matrix ROW per lane: at R = 32 that is 64 VGPRs per matrix,
// spills, and 1.2 ms for the ~40 dependent 32 x 32 operations of BASELINE config 4 -- longer than its streaming collapse
// (1.13 ms), which it also slowed down to 1.86 ms by sitting beside it on a third of the CUs at priority 3.
// ------------------------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(R * R) void cov_grid_kernel(FastArgs a) {
    extern __shared__ __attribute__((aligned(16))) double gsm[];
    constexpr int RR = R * R, RT = R * kTileStride<R>;
    const int b = blockIdx.x, l = threadIdx.x;
    Grid<R> G;
    G.l = l; G.i = l / R; G.j = l % R;
    G.prow = gsm + 4 * RT;
    G.red = G.prow + kGridProw<R>;
    G.tt = G.red + 2 * (RR / 64) * R;
    __builtin_amdgcn_s_setprio(3);
    const double* Cf = a.Cfull + (size_t)b * RR;
    const double Cel = 0.5 * (Cf[l] + Cf[G.j * R + G.i]);     // exactly symmetric (the matrix-pipe Gram is symmetric to rounding)
    Cov8Dst o;
    o.tab = a.tab + (size_t)b * a.T * 3 * RR; o.tab_cap = a.T; o.tab_over = o.tab;
    o.stead = a.stead + (size_t)b * stead_mats(R) * RR;
    o.PT = a.PT + (size_t)b * RR; o.xi0 = a.xi0 + (size_t)b * R; o.llc = a.llc + b; o.E = a.E + b; o.fill = a.fill + 2 * b;
    o.PsInf = a.PsInf + (size_t)b * RR;
    o.SP11 = a.SP11 ? a.SP11 + (size_t)b * RR : nullptr;
    o.SU = a.SP11 ? a.SU + (size_t)b * RR : nullptr;
    o.P0s = a.SP11 ? a.P0s + (size_t)b * RR : nullptr;
    cov_grid<R, scan_levels(R), 4>(a, b, Cel, a.ldfull[b], gsm, o, G);
}

template <int R>
static hipError_t launch_cov_grid_r(const FastArgs& a, hipStream_t s) {
    constexpr int RR = R * R, RT = R * kTileStride<R>;
    const size_t lds = (size_t)(4 * RT + kGridProw<R> + 2 * (RR / 64) * R + 2 * RT) * sizeof(double);
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cov_grid_kernel<R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((cov_grid_kernel<R>), dim3(a.B), dim3(RR), lds, s, a);
    return hipGetLastError();
}
bool cov_grid_supported(int Rpad) { return Rpad == 16 || Rpad == 32; }
hipError_t launch_cov_grid(int Rpad, const FastArgs& a, hipStream_t s) {
    note_kernel("cov_grid_kernel");
    if (Rpad == 16) return launch_cov_grid_r<16>(a, s);
    if (Rpad == 32) return launch_cov_grid_r<32>(a, s);
    return hipErrorInvalidValue;
}

hipError_t launch_cov_wave(const FastArgs& a, hipStream_t s) {
#ifdef DFM_DIAG
    note_kernel("cov_wave_kernel");
    hipLaunchKernelGGL(cov_wave_kernel, dim3((a.B + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
#else
    (void)a; (void)s;
    return hipErrorInvalidValue;                                  // (DFM_COV_WAVE is a switch of the diagnostics build)
#endif
}

}  // namespace dfm
