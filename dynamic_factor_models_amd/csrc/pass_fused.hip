// pass_fused.hip -- the whole balanced-panel Kalman-smoother pass in ONE launch (Rp = 8): persistent workgroups, one per
// CU, each walking its replicates b = blockIdx.x, blockIdx.x + gridDim.x, ...; inside a workgroup the waves are
// specialised:
//
//   waves 0 .. nsw-1  STREAM   the collapse of collapse_mfma.hip (LDS-DMA ring of period slots, contraction on
//                              v_mfma_f64_4x4x4): wave w streams segment w of the replicate's T periods.  b_t goes to an
//                              LDS array [T][8] instead of HBM, sum_t s_t to an LDS slot.
//   the last wave     COV      Gram matrix C = Lam' R^-1 Lam, the data-independent covariance recursion (dfm_cov8.h: one
//                              wave, element per lane) into LDS tables, the transient rows of P_smooth, then the
//                              fixed-point rows of P_smooth (pure stores) -- all beside the stream of the SAME replicate.
//   waves 0 .. 3      SCAN     after a workgroup barrier: the time-parallel mean recursion (dfm_scan.h, the code of
//                              meanscan_kernel) with b_t / w_t in LDS: f_smooth, log-likelihood.  Before it starts, the
//                              stream waves have already issued the first ring fill of the NEXT replicate, so the DMA
//                              engine keeps HBM busy while the (latency-bound, ~5 us) scan runs.
//
// Against the two-launch pass (fused collapse launch + meanscan_kernel): no b_t round trip through HBM (33 MB written,
// read back), no w_t scratch (33 MB + 33 MB), no second launch whose ~55 us of dependent scans ran behind the stream
// instead of beside it, and the covariance recursion no longer holds 128 wide waves resident for 125 us.  HBM traffic is
// the algorithmic minimum of SURVEY 8(d): every input read once, every output written once.
// Synchronisation is two workgroup barriers per replicate -- no inter-workgroup communication, no spin-wait.
// The reference has no counterpart (dfm_functions.ipynb:21-23 declares `Parametric` only).
#include <string.h>

#include <type_traits>

#include "dfm_cov8.h"
#include "dfm_gram.h"
#include "dfm_kernels.h"
#include "dfm_scan.h"

namespace dfm {

namespace {

using lds_char_ptr_f = __attribute__((address_space(3))) char*;

__device__ __forceinline__ void dma16f(const void* gsrc, unsigned lds_dst) {
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
__device__ __forceinline__ void wait_vmf() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(K) : "memory");
}
__device__ __forceinline__ void wait_lgkmf() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int kPfR = 8;
constexpr int kPfEcap = 8;                                  // transient covariance steps kept in LDS (later ones: global tab)
constexpr int kPfNst = stead_mats(kPfR);                    // steady Z, J, G + the carry powers (256 scan threads)
constexpr int kPfNlev = scan_levels(kPfR);
constexpr int kPfMaxThreads = 512;

// row-slot stride: as collapse_mfma.hip (4 consecutive slots start 64 bytes apart modulo 256)
__host__ __device__ inline unsigned pf_slot_bytes(int N) {
    unsigned sb = (unsigned)N * 8u;
    while ((sb & 255u) != 64u && (sb & 255u) != 192u) sb += 16u;
    return sb;
}

}  // namespace

// byte offsets into the dynamic LDS of pass_fused_kernel (computed by the host)
struct PfLds {
    unsigned smat;    // [kPfNst + 1][64] doubles: steady matrices, then P_T
    unsigned ctab;    // [kPfEcap][3][64] doubles: Z_e, J_e, G_e
    unsigned misc;    // xi0 [8] | llc [1] | pad [7] | PsInf [64] | ps packed [40] | vec [16] | red [8] | ssum [16] | ints [8]
    unsigned covws;   // kCov8ScratchDoubles doubles
    unsigned sa, sb;  // [32][8] doubles each (carry scan)
    unsigned bt;      // [T4][8] doubles: b_t, then w_t
    unsigned ring;    // nsw x 8 slots x SB bytes
    unsigned total;
};
constexpr int kMiscXi0 = 0, kMiscLlc = 8, kMiscPsInf = 16, kMiscPs = 80, kMiscVec = 120, kMiscRed = 136, kMiscSsum = 144,
              kMiscInts = 160, kMiscDoubles = 164;

// ------------------------------------------------------------------------------------------------------------------
// The scan of one replicate, operands in LDS.  Called by EVERY wave of the workgroup (the barriers inside are workgroup
// barriers); threads with act = false (tid >= 256) only keep the barrier count.  meanscan_kernel's algorithm (fastpath.hip).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void scan_lds(const FastArgs& a, int b, int tid, bool act, double* bt, const double* s_tab,
                                         const double* tab_over, const double* s_mat, const double* xi0p, const double* llcp,
                                         int E, double* s_a, double* s_b, double* s_vec, double* s_red, const double* ssum,
                                         int nseg, double* pslot) {
    auto mark = [&](int k) {       // diagnostics: phase stamps of thread 0 (pslot null: off)
        if (pslot && tid == 0) { pslot[k] = (double)__builtin_amdgcn_s_memrealtime(); pslot[k + 10] = (double)__builtin_amdgcn_s_memtime(); }
    };
    mark(10);
    constexpr int R = kPfR;
    constexpr int NG = kScanThreads / R;
    constexpr int NLEV = kPfNlev;
    constexpr int NST = kPfNst;
    const int c = act ? tid / R : 0, i = tid % R;
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
    if (ts > 0) {                                             // (workgroup-uniform)
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
        __syncthreads();
        xi = s_vec[i];
        __syncthreads();                  // s_vec is reused for xi_T
    }
    // xi = xi_ts in every group.
    mark(11);

    // ---- steady forward scan: steps ts .. T-1; group c owns steps ts + c L + j ------------------------------------
    {
        double Gp[R], Zp[R];
        load_xperm<R>(Gp, s_mat + 2 * R * R, i);
        double dummy = 0.0;
        double e = 0.0;
        if (act) {
            if (fullf) chunk_prefetch<R, true>(cur, bt, t0f, 1, L, ts, T, i);
            else chunk_prefetch<R, false>(cur, bt, t0f, 1, L, ts, T, i);
            // phase 1: chunk from a zero state
            e = fullf ? chunk_run<R, true, 0>(Gp, Gp, 0.0, cur, bt, t0f, 1, L, ts, T, i, nullptr, dummy, nullptr, r)
                      : chunk_run<R, false, 0>(Gp, Gp, 0.0, cur, bt, t0f, 1, L, ts, T, i, nullptr, dummy, nullptr, r);
            if (fullf) chunk_prefetch<R, true>(cur, bt, t0f, 1, L, ts, T, i);     // operands of phase 3
            else chunk_prefetch<R, false>(cur, bt, t0f, 1, L, ts, T, i);
        }
        mark(12);
        // phase 2: true start state of chunk c
        const double s = carry_scan<R, NG>(e, xi, s_mat + 3 * R * R, c, i, s_a, s_b, act);
        mark(13);
        // phase 3: re-run from the true start; emit w_t (in place of b_t), accumulate xi_t' w_t
        if (act) {
            load_xperm<R>(Zp, s_mat, i);
            const double v = fullf ? chunk_run<R, true, 1>(Gp, Zp, s, cur, bt, t0f, 1, L, ts, T, i, bt, dot, nullptr, r)
                                   : chunk_run<R, false, 1>(Gp, Zp, s, cur, bt, t0f, 1, L, ts, T, i, bt, dot, nullptr, r);
            if (c == clast) s_vec[i] = v;                     // xi_T
        }
    }
    mark(14);
    __syncthreads();   // xi_T in LDS; every w_t of this replicate is written

    // ---- terminal + steady backward scan: steps T-1 .. ts; group c owns steps T-1 - c L - j --------------------
    double fT = 0.0;
    {
        const double xiT = s_vec[i];
        double PTp[R];
        load_xperm<R>(PTp, s_mat + NST * R * R, i);
        fT = matvec_x<R>(PTp, xiT);
        if (act && c == 0) {
            dot = fma(xiT, fT, dot);          // the log-likelihood needs sum xi'w + xi_T' f_T
            if (i < r) fout[(size_t)(T - 1) * r + i] = fT;
        }
    }
    double fb;   // smoothed mean at the steady/transient boundary (period ts)
    {
        double Jp[R];
        load_xperm<R>(Jp, s_mat + R * R, i);
        double dummy = 0.0;
        double e = 0.0;
        if (act) {
            if (fullb) chunk_prefetch<R, true>(cur, bt, t0b, -1, L, ts, T, i);
            else chunk_prefetch<R, false>(cur, bt, t0b, -1, L, ts, T, i);
            e = fullb ? chunk_run<R, true, 0>(Jp, Jp, 0.0, cur, bt, t0b, -1, L, ts, T, i, nullptr, dummy, nullptr, r)
                      : chunk_run<R, false, 0>(Jp, Jp, 0.0, cur, bt, t0b, -1, L, ts, T, i, nullptr, dummy, nullptr, r);
            if (fullb) chunk_prefetch<R, true>(cur, bt, t0b, -1, L, ts, T, i);
            else chunk_prefetch<R, false>(cur, bt, t0b, -1, L, ts, T, i);
        }
        mark(15);
        const double s = carry_scan<R, NG>(e, fT, s_mat + (size_t)(3 + NLEV) * R * R, c, i, s_a, s_b, act);
        mark(16);
        if (act) {
            const double v = fullb ? chunk_run<R, true, 2>(Jp, Jp, s, cur, bt, t0b, -1, L, ts, T, i, nullptr, dummy, fout, r)
                                   : chunk_run<R, false, 2>(Jp, Jp, s, cur, bt, t0b, -1, L, ts, T, i, nullptr, dummy, fout, r);
            if (c == clast) s_vec[R + i] = v;
        }
        __syncthreads();
        fb = s_vec[R + i];
    }

    mark(17);
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

    mark(18);
    // ---- log-likelihood ---------------------------------------------------------------------------------------
    if (!act) dot = 0.0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, kWave);
    if (act && (tid & 63) == 0) s_red[tid >> 6] = dot;
    __syncthreads();
    if (tid == 0) {
        double d = 0.0, sq = 0.0;
#pragma unroll
        for (int w = 0; w < kScanThreads / 64; ++w) d += s_red[w];
        for (int w = 0; w < nseg; ++w) sq += ssum[w];
        a.loglik[b] = -0.5 * (llcp[0] + sq - d);
    }
}

// ------------------------------------------------------------------------------------------------------------------
template <int STEPS, int NDR>
__global__ __launch_bounds__(kPfMaxThreads, 2) void pass_fused_kernel(CollapseArgs a, FastArgs fa, unsigned SB, int nsw, PfLds ly) {
    constexpr int R = kPfR;
    constexpr int NB = 2, NS = 4 * NB;                       // row blocks / row slots of a wave's ring
    constexpr int CS = 8;                                    // series per MFMA step at R = 8: 2 series groups x 2 factor groups
    constexpr int NQ = NDR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_stream = wave < nsw;
    const bool is_cov = wave == (int)(blockDim.x >> 6) - 1;   // the last wave (the workgroup has max(nsw + 1, 4) waves: the scan needs 4)
    double* s_mat = reinterpret_cast<double*>(smem + ly.smat);
    double* s_tab = reinterpret_cast<double*>(smem + ly.ctab);
    double* misc = reinterpret_cast<double*>(smem + ly.misc);
    double* covws = reinterpret_cast<double*>(smem + ly.covws);
    double* s_a = reinterpret_cast<double*>(smem + ly.sa);
    double* s_b = reinterpret_cast<double*>(smem + ly.sb);
    double* bt = reinterpret_cast<double*>(smem + ly.bt);
    int* ints = reinterpret_cast<int*>(misc + kMiscInts);    // E, fill_lo, fill_hi
    const int N = a.N, T = a.T, B = a.B;
    const unsigned rowB = (unsigned)N * 8u;

    // ---- stream-wave constants: lane roles of v_mfma_f64_4x4x4 (collapse_mfma.hip), segment of this wave ---------
    const int K = lane >> 4, blk = (lane >> 2) & 3, q = lane & 3;
    const int g = blk >> 1, h = blk & 1;
    int tq = (T + nsw - 1) / nsw;
    {
        unsigned gg = rowB & 127u;
        gg = gg == 0 ? 128u : (gg & (~gg + 1u));
        const int m = (int)(128u / gg);
        tq = ((tq + m - 1) / m) * m;                         // segments start on 128-byte boundaries
    }
    const int sw = is_stream ? wave : 0;
    const int ta = (sw * tq < T) ? sw * tq : T;
    const int tb = (ta + tq < T) ? ta + tq : T;
    const int nrows = is_stream ? tb - ta : 0;
    const int nblk = (nrows + 3) / 4;
    const unsigned ringB = NS * SB;
    const char* ring = smem + ly.ring + (size_t)sw * ringB;
    const unsigned ring_lds =
        __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_f)(smem)) + ly.ring + (unsigned)sw * ringB;
    const unsigned lane16 = 16u * lane;
    bool pact[NDR];                                          // lane moves 16 bytes of piece p of a row
#pragma unroll
    for (int p = 0; p < NDR; ++p) pact[p] = lane16 + 1024u * p < rowB;
    const unsigned lane_off = (unsigned)q * SB + (unsigned)(4 * g + K) * 8u;
    const bool tail_clamp = (STEPS - 1) * CS + 4 * g + K >= N;
    const unsigned last_off = tail_clamp ? (unsigned)q * SB + (unsigned)(N - 1) * 8u : lane_off + (unsigned)(STEPS - 1) * (CS * 8u);

    double Bw[STEPS];                                        // B operands: lam_cf / R_c for c = s CS + 4 g + K, f = 4 h + q
    double rown[NQ][2];                                      // 1 / R of the lane's own 16-byte column pairs (s_t pass)
    auto issue_row = [&](const char* seg, int rr, int slot) {
        const char* src = seg + (size_t)rr * rowB + lane16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)slot * SB);
#pragma unroll
        for (int p = 0; p < NDR; ++p) {
            if (pact[p]) dma16f(src + 1024 * p, dst + 1024u * p);
        }
    };
    // first ring fill + weights of replicate bb (stream waves): the fill first, the weights behind it -- a counted wait on
    // the fill later also covers everything older
    auto prepare = [&](int bb) {
        const char* seg = reinterpret_cast<const char*>(a.panel + ((size_t)bb * T + ta) * N);
#pragma unroll
        for (int sl = 0; sl < NS; ++sl)
            if (sl < nrows) issue_row(seg, sl, sl);
        const double* __restrict__ Lg = a.Lam + (size_t)bb * N * R;
        const double* __restrict__ Rg = a.Rv + (size_t)bb * N;
        double rv[STEPS];
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int c = s * CS + 4 * g + K;
            const int cc = c < N ? c : N - 1;
            rv[s] = Rg[cc];
            Bw[s] = Lg[(size_t)cc * R + 4 * h + q];
        }
#pragma unroll
        for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = 2 * lane + 128 * jq + e;
                rown[jq][e] = Rg[c < N ? c : N - 1];
            }
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int c = s * CS + 4 * g + K;
            Bw[s] = (c < N) ? Bw[s] * (1.0 / rv[s]) : 0.0;
        }
#pragma unroll
        for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = 2 * lane + 128 * jq + e;
                rown[jq][e] = (c < N) ? 1.0 / rown[jq][e] : 0.0;
            }
    };

    int b = blockIdx.x;
    if (is_stream && b < B && nrows > 0) prepare(b);

    // DFM_SCAN_ABL bit 8: s_memrealtime stamps (10 ns ticks) of the phases of every replicate into scol[b][0..15]
    const bool prof = (fa.abl & 256) != 0 && a.scol != nullptr && T >= 16;
    auto stamp = [&](int bb, int slot) {
        if (prof && lane == 0) a.scol[(size_t)bb * T + slot] = (double)__builtin_amdgcn_s_memrealtime();
    };
    for (; b < B; b += gridDim.x) {
        const int bn = b + (int)gridDim.x;
        if (wave == 0) stamp(b, 0);                               // iteration starts
        if (is_stream) {
            // ================= STREAM: segment [ta, tb) of replicate b -> bt[t][0..7], ssum[wave] =================
            double qa[NQ][2];
#pragma unroll
            for (int jq = 0; jq < NQ; ++jq) { qa[jq][0] = 0.0; qa[jq][1] = 0.0; }
            if (nrows > 0) {
                const char* seg = reinterpret_cast<const char*>(a.panel + ((size_t)b * T + ta) * N);
                int issued = NS;
                // One row block.  MODE 0: main loop (counted wait; the slots of the block are re-armed with periods that
                // exist).  MODE 1: the block after the main loop (counted wait still valid; the last < 4 periods are
                // issued).  MODE 2: drain.
                auto row_block = [&](int bk, int bslot, auto mode_tag) {
                    constexpr int MODE = decltype(mode_tag)::value;
                    const int r0 = bk * 4;
                    // rows < r0 + 4 have landed once at most the operations YOUNGER than their DMAs are outstanding (one
                    // in-order vmcnt counter per wave): the re-arms of the NB - 1 row blocks in between
                    if constexpr (MODE <= 1) wait_vmf<((NB - 1) * 4 * NDR <= 63 ? (NB - 1) * 4 * NDR : 63)>();
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
                    if constexpr (MODE == 0) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) issue_row(seg, issued + rr, bslot * 4 + rr);
                        issued += 4;
                    } else if constexpr (MODE == 1) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            if (issued + rr < nrows) issue_row(seg, issued + rr, bslot * 4 + rr);
                        issued += 4;
                    }
                    double D = 0.0;
#pragma unroll
                    for (int s = 0; s < STEPS; ++s) D = __builtin_amdgcn_mfma_f64_4x4x4f64(xa[s], Bw[s], D, 0, 0, 0);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        if (MODE == 0 || r0 + rr < nrows) {              // wave-uniform: rows past the segment hold stale slots
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
                    if (g == 0 && (q & 1) == 0 && (MODE == 0 || t < tb))
                        *reinterpret_cast<double2*>(&bt[(size_t)t * R + 4 * h + q]) = make_double2(D, hi);
                };
                const int nmain = (nrows - 4 * NB) >= 4 ? (nrows - 4 * NB) / 4 : 0;
                int bslot = 0, bk = 0;
                for (; bk < nmain; ++bk) {
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
                wait_vmf<0>();
            }
            double sp = 0.0;
#pragma unroll
            for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
                for (int e = 0; e < 2; ++e) sp = fma(qa[jq][e], rown[jq][e], sp);
            sp = wave_allsum(sp);
            if (lane == 0) {
                misc[kMiscSsum + wave] = nrows > 0 ? sp : 0.0;
                if (sp != sp) atomicOr(a.status, 1);             // NaN in the panel on the balanced path
            }
            if (wave == 0) stamp(b, 1);                           // wave 0's segment is done
            if (wave == nsw - 1) stamp(b, 5);                     // the last stream wave's segment is done
            // the NEXT replicate's first ring fill and weights go out now: they land while the scan below runs
            if (bn < B && nrows > 0) prepare(bn);
        } else if (is_cov) {
            // ================= COV: Gram matrix, covariance recursion, P_smooth ====================================
            __builtin_amdgcn_s_setprio(2);
            stamp(b, 6);
            double* Cs = covws + 5 * kCov8TileDoubles;
            const double ld = gram_wave8<NDR>(fa.Lam + (size_t)b * N * R, fa.Rv + (size_t)b * N, N, lane, Cs);
            wave_lds_sync();
            const double Cel = Cs[lane];
            Cov8Dst o;
            o.tab = s_tab; o.tab_cap = kPfEcap; o.tab_over = fa.tab + (size_t)b * T * 3 * 64;
            o.stead = s_mat; o.PT = s_mat + kPfNst * 64;
            o.xi0 = misc + kMiscXi0; o.llc = misc + kMiscLlc; o.E = ints; o.fill = ints + 1; o.PsInf = misc + kMiscPsInf;
            o.SP11 = fa.SP11 ? fa.SP11 + (size_t)b * 64 : nullptr;
            o.SU = fa.SP11 ? fa.SU + (size_t)b * 64 : nullptr;
            o.P0s = fa.SP11 ? fa.P0s + (size_t)b * 64 : nullptr;
            stamp(b, 7);                                          // Gram done
            cov_wave8<kPfNlev>(fa, b, Cel, ld, covws, o, lane);
            wave_lds_sync();
            stamp(b, 8);                                          // covariance recursion done
            if (fa.SP11) fa.PT[(size_t)b * 64 + lane] = o.PT[lane];   // EM: em_update_kernel reads P_T from global memory
            if (fa.P_smooth) {      // rows [lo, hi) of P_smooth equal the backward fixed point: pure 16-byte stores
                const int npr = fa.r * (fa.r + 1) / 2;
                double* ps = misc + kMiscPs;
                if (lane < npr) {
                    int ri = 0;
                    while ((ri + 1) * (ri + 2) / 2 <= lane) ++ri;
                    ps[lane] = o.PsInf[ri * R + (lane - ri * (ri + 1) / 2)];
                }
                wave_lds_sync();
                const int lo = ints[1], hi = ints[2];
                if (hi > lo) {
                    double* base = fa.P_smooth + ((size_t)b * T + lo) * npr;
                    const unsigned n = (unsigned)(hi - lo) * (unsigned)npr;
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
            }
            __builtin_amdgcn_s_setprio(0);
            stamp(b, 9);                                          // P_smooth fill issued
        }
        __syncthreads();            // (A) b_t, sum s_t and the covariance tables of replicate b are in LDS
        if (wave == 0) stamp(b, 2);                               // past barrier A
        scan_lds(fa, b, tid, tid < kScanThreads, bt, s_tab, fa.tab + (size_t)b * T * 3 * 64, s_mat, misc + kMiscXi0,
                 misc + kMiscLlc, ints[0], s_a, s_b, misc + kMiscVec, misc + kMiscRed, misc + kMiscSsum, nsw,
                 (prof && T >= 32) ? a.scol + (size_t)b * T : nullptr);
        if (wave == 0) stamp(b, 3);                               // scan done (wave 0)
        __syncthreads();            // (B) the scan is done with bt / the tables: the next replicate may overwrite them
        if (wave == 0) stamp(b, 4);
    }
}

// ------------------------------------------------------------------------------------------------------------------
static PfLds pf_layout(int T, int N, int nsw) {
    PfLds l;
    unsigned off = 0;
    auto take = [&](unsigned bytes) { const unsigned at = off; off += (bytes + 255u) & ~255u; return at; };
    l.smat = take((kPfNst + 1) * 64 * 8);
    l.ctab = take(kPfEcap * 3 * 64 * 8);
    l.misc = take(kMiscDoubles * 8);
    l.covws = take(kCov8ScratchDoubles * 8);
    l.sa = take(32 * 8 * 8);
    l.sb = take(32 * 8 * 8);
    l.bt = take((unsigned)((T + 3) / 4 * 4) * 8 * 8);
    l.ring = take((unsigned)nsw * 8u * pf_slot_bytes(N));
    l.total = off;
    return l;
}

int pass_fused_pick_nsw(int T, int N, int want) {
    int nsw = want > 0 ? want : 7;
    if (nsw > 7) nsw = 7;
    while (nsw > 1 && T / nsw < 8) --nsw;                     // keep segments a few row blocks long
    while (nsw > 1 && pf_layout(T, N, nsw).total > 160u * 1024u) --nsw;
    return nsw;
}

// Rp = 8, the shapes of the MFMA collapse (even N, 8N <= 4096, ceil(N / 8) <= 32 steps), T below the int-index and LDS limits
bool pass_fused_supported(int Rpad, int T, int N) {
    if (Rpad != 8 || !collapse_mfma_supported(8, N)) return false;
    if (T < 2) return false;
    return pf_layout(T, N, 1).total <= 160u * 1024u;
}

template <int STEPS, int NDR>
static hipError_t launch_pf_one(const CollapseArgs& a, const FastArgs& fa, int nsw, int num_cu, hipStream_t s) {
    const PfLds ly = pf_layout(a.T, a.N, nsw);
    if (ly.total > 160u * 1024u) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_fused_kernel<STEPS, NDR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int grid = a.B < num_cu ? a.B : num_cu;
    const int nwaves = nsw + 1 > kScanThreads / 64 ? nsw + 1 : kScanThreads / 64;
    hipLaunchKernelGGL((pass_fused_kernel<STEPS, NDR>), dim3(grid), dim3(64 * nwaves), ly.total, s, a, fa, pf_slot_bytes(a.N),
                       nsw, ly);
    return hipGetLastError();
}

template <int S>
static hipError_t launch_pf_pick(const CollapseArgs& a, const FastArgs& fa, int nsw, int num_cu, hipStream_t s, int steps) {
    if constexpr (S > 32) {
        return hipErrorInvalidValue;
    } else {
        if (steps == S) {
            const int ndr = (a.N * 8 + 1023) / 1024;
            constexpr int lo = (8 * (S - 1) * 8 + 8 + 1023) / 1024, hi = (8 * S * 8 + 1023) / 1024;
            if constexpr (lo <= 1 && 1 <= hi) { if (ndr == 1) return launch_pf_one<S, 1>(a, fa, nsw, num_cu, s); }
            if constexpr (lo <= 2 && 2 <= hi) { if (ndr == 2) return launch_pf_one<S, 2>(a, fa, nsw, num_cu, s); }
            if constexpr (lo <= 3 && 3 <= hi) { if (ndr == 3) return launch_pf_one<S, 3>(a, fa, nsw, num_cu, s); }
            if constexpr (lo <= 4 && 4 <= hi) { if (ndr == 4) return launch_pf_one<S, 4>(a, fa, nsw, num_cu, s); }
            return hipErrorInvalidValue;
        }
        return launch_pf_pick<S + 1>(a, fa, nsw, num_cu, s, steps);
    }
}

hipError_t launch_pass_fused(const CollapseArgs& a, const FastArgs& fa, int nsw, int num_cu, hipStream_t s) {
    return launch_pf_pick<1>(a, fa, nsw, num_cu, s, (a.N + 7) / 8);
}

// ------------------------------------------------------------------------------------------------------------------
// cov_wave_kernel: dfm_cov8.h's one-wave-per-replicate covariance recursion as a drop-in for cov_kernel (same global
// outputs) -- DFM_COV_WAVE=1 on the separate-launch path; diagnostics and A/B of the recursion itself.
// ------------------------------------------------------------------------------------------------------------------
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

hipError_t launch_cov_wave(const FastArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(cov_wave_kernel, dim3((a.B + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace dfm
