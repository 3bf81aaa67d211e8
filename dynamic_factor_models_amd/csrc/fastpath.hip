// fastpath.hip -- Kalman-smoother recursion for BALANCED panels, parallel in time.
//
// With no missing cell C_t = Lam' R^-1 Lam is the same every period, so the covariance half of the
// information-form recursion (recursion.hip header; oracle/info_form.py) does not depend on the
// data and reaches its Riccati fixed point after E steps (a handful when the cross-section is
// informative).  The pass is split accordingly:
//
//   cov_kernel       (data-independent, sequential, O(E) steps; runs beside the streaming collapse)
//       forward   Z_e, J_e, G_e = Psi' Z_e  for the E distinct steps; step t uses entry min(t, E-1)
//       terminal  P_T, log-determinants
//       backward  P_s,t = Z_t + J_t P_s,t+1 J_t' : distinct near t = T and for t < E-1, one fixed
//                 point P_s,inf in between (rows of P_smooth in that range are filled by meanscan)
//       powers    G^L, J^L of the steady matrices for the chunk carries of meanscan
//   meanscan_kernel  (the means; one workgroup per replicate, 256/R lane groups = time chunks)
//       xi_{t+1} = G_t xi_t + b_t,  w_t = Z_t xi_t           forward  (t < E-1 sequential, then
//       f_t      = w_t + J_t f_{t+1}                         backward  a 3-phase chunked scan)
//     Steady steps form a linear time-invariant recurrence: every chunk of L steps first runs from a
//     zero state (phase 1), the chunk carries are chained with G^L / J^L (phase 2), and the chunk is
//     re-run from its true start state (phase 3) -- sequential depth 2L + T/L instead of T.
//
// r x r matrix-vector products inside a lane group use no LDS: lane i holds M[i][i^s], s = 0..R-1,
// and fetches x[i^s] with DPP row operations (xor_lane in dfm_device.h).
// The reference has no counterpart (dfm_functions.ipynb:21-23 declares `Parametric` only).
#include <stdlib.h>

#include "dfm_cov.h"
#include "dfm_scan.h"

namespace dfm {

// cov workgroup: 4 independent waves.  Four-wave workgroups put the 224-VGPR covariance waves on a quarter of
// the CUs a one-wave workgroup would touch, which is what the streaming collapse beside them loses.
// minimum waves per SIMD the register allocation of cov_kernel must allow (3 = at most 168 VGPRs, the slot of one
// collapse wave, was measured: spills slow the chain by 20 us and the collapse beside it gains nothing)
#ifndef DFM_COV_MIN_WAVES
#define DFM_COV_MIN_WAVES 2
#endif

// ================================================================================================
// cov_kernel: cov_body (dfm_cov.h), one wave per workgroup
// ================================================================================================
template <int R, int CPL2>
__global__ __launch_bounds__(cov_threads(R), DFM_COV_MIN_WAVES) void cov_kernel(FastArgs a) {
    using LY = CovLayout<R>;
    __builtin_amdgcn_s_setprio(3);   // a latency chain everything waits for, beside the streaming collapse
    extern __shared__ __attribute__((aligned(16))) double smem[];
    cov_body<R, CPL2, LY::ECL>(a, (int)blockIdx.x * LY::GPW, smem, (int)threadIdx.x);
}

// ================================================================================================
// meanscan_kernel
// ================================================================================================
// The same fill as its own launch: it only needs cov_kernel's outputs, so it can run behind cov_kernel beside
// the streaming collapse instead of inside the scan that everything waits for.
template <int R>
__global__ __launch_bounds__(256) void pfill_kernel(FastArgs a) {
    __shared__ double s_ps[R * (R + 1) / 2];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int npr = a.r * (a.r + 1) / 2;
    for (int v = tid; v < npr; v += 256) {
        int ri = 0;
        while ((ri + 1) * (ri + 2) / 2 <= v) ++ri;
        s_ps[v] = a.PsInf[(size_t)b * R * R + ri * R + (v - ri * (ri + 1) / 2)];
    }
    __syncthreads();
    // gridDim.y slices of the row range: one workgroup per replicate cannot saturate the write bandwidth when a replicate's
    // P_smooth is megabytes (config 4: 3.4 MB each, 256 replicates -- 1.03 ms; 8 slices: the stores of 2048 workgroups)
    // ... and every WAVE walks its own contiguous quarter of the slice: four waves interleaved 1 KB at a time wrote at 4.1 TB/s,
    // a wave per contiguous segment writes at 5.5-5.9 (scripts/microbench/storebw.hip, profiles/r04/microbench_storebw.txt)
    const int lo = a.fill[2 * b], hi = a.fill[2 * b + 1];
    const int ns = 4 * (int)gridDim.y, sl = 4 * (int)blockIdx.y + (tid >> 6);
    const long long span = hi > lo ? hi - lo : 0;
    fill_psmooth_range(a, b, tid & 63, 64, s_ps, lo + (int)(span * sl / ns), lo + (int)(span * (sl + 1) / ns));
}

template <int R>
struct ScanLds {   // dynamic LDS of meanscan_kernel, in doubles
    static constexpr int ST = scan_threads(R);
    static constexpr int NG = ST / R;
    static constexpr int NM = stead_mats(R) + 1;                  // steady matrices + P_T
    static constexpr int oMat = 0;
    static constexpr int oTab = oMat + NM * R * R;                // ecap(R) x (Z, J, G)
    static constexpr int oB0 = oTab + ecap(R) * 3 * R * R;        // b_t, then w_t, of the staged transient steps
    static constexpr int oA = oB0 + ecap(R) * R;
    static constexpr int oB = oA + NG * R;
    static constexpr int oVec = oB + NG * R;
    static constexpr int oPs = oVec + 2 * R;
    static constexpr int oRed = oPs + R * (R + 1) / 2;
    static constexpr int total = oRed + ST / 64;
    static constexpr size_t bytes() { return (size_t)total * sizeof(double); }
};

template <int R>
__global__ __launch_bounds__(scan_threads(R)) void meanscan_kernel(FastArgs a) {
    using LY = ScanLds<R>;
    constexpr int ST = scan_threads(R);
    constexpr int NG = LY::NG;
    constexpr int NLEV = scan_levels(R);
    constexpr int NST = stead_mats(R);
    extern __shared__ __attribute__((aligned(16))) double dsm[];
    double* s_mat = dsm + LY::oMat;      // Z, J, G, G-powers, J-powers, P_T   (row-major R x R each)
    double* s_tab = dsm + LY::oTab;
    double* s_b0 = dsm + LY::oB0;
    double* s_a = dsm + LY::oA;
    double* s_b = dsm + LY::oB;
    double* s_vec = dsm + LY::oVec;      // xi_T | f at the steady/transient boundary
    double* s_ps = dsm + LY::oPs;
    double* s_red = dsm + LY::oRed;
    const int b = blockIdx.x + a.b0;
    const int tid = threadIdx.x;
    const int c = tid / R, i = tid % R;
    const int T = a.T, r = a.r, L = a.L;
    const int E = a.E[b];
    const int ts = E - 1;
    const int nst = ts < ecap(R) ? ts : ecap(R);              // transient steps staged in LDS
    const double* bcol = a.bcol + (size_t)b * T * R;
    double* wtab = a.wtab + (size_t)b * (a.wrep ? a.wrep : (size_t)T * R);
    const double* tab = a.tab + (size_t)b * T * 3 * R * R;
    const double* stead = a.stead + (size_t)b * NST * R * R;
    double* fout = a.f_smooth + (size_t)b * T * r;
    const int npr = r * (r + 1) / 2;

    // ---- one round trip to global memory for everything the sequential parts will touch ------------
    for (int k = tid; k < NST * R * R; k += ST) s_mat[k] = stead[k];
    for (int k = tid; k < R * R; k += ST) s_mat[NST * R * R + k] = a.PT[(size_t)b * R * R + k];
    {   // a fixed count, so these loads do not wait for E (entries past E are never used)
        const int nfix = ecap(R) < T ? ecap(R) : T;
        for (int k = tid; k < nfix * 3 * R * R; k += ST) s_tab[k] = tab[k];
        for (int k = tid; k < nfix * R; k += ST) s_b0[k] = bcol[k];
    }
    if (a.P_smooth) {
        for (int v = tid; v < npr; v += ST) {       // packed (caller's r) copy of P_s,inf
            int ri = 0;
            while ((ri + 1) * (ri + 2) / 2 <= v) ++ri;
            s_ps[v] = a.PsInf[(size_t)b * R * R + ri * R + (v - ri * (ri + 1) / 2)];
        }
    }
    double xi = a.xi0[(size_t)b * R + i];
    const int clast = (T - 1 - ts) / L;                       // chunk that holds step T-1 (fwd) / step ts (bwd)
    const int cmax = c | (64 / R - 1);                        // last chunk handled by this wave
    const int t0f = ts + c * L;                               // forward chunk: steps t0f + j
    const int t0b = T - 1 - c * L;                            // backward chunk: steps t0b - j
    const bool fullf = (ts + (cmax + 1) * L) <= T;            // wave-uniform: no partial chunk in this wave
    const bool fullb = (T - (cmax + 1) * L) >= ts;
    double cur[kPF];
    if (fullf) chunk_prefetch<R, true>(cur, bcol, t0f, 1, L, ts, T, i);
    else chunk_prefetch<R, false>(cur, bcol, t0f, 1, L, ts, T, i);
    __syncthreads();

    // ---- P_smooth rows inside the fixed-point range: fire-and-forget stores (unless pfill_kernel wrote them)
    if (a.P_smooth && !(a.abl & 1)) fill_psmooth_rows(a, b, tid, ST, s_ps);
    if (a.abl & 2) return;
    if (a.abl & 4) return;

    // ---- forward transient: steps 0 .. ts-1 on wave 0 only (its lane groups redundantly); the other
    // waves wait at the barrier, leaving their SIMDs to the co-resident workgroups
    double dot = 0.0;                     // lane part of sum_t xi_t' w_t
    if (ts > 0) {
        if (tid < 64) {
            for (int t = 0; t < ts; ++t) {
                double Zp[R], Gp[R];
                const bool st = t < nst;
                const double* ent = st ? s_tab + (size_t)t * 3 * R * R : tab + (size_t)t * 3 * R * R;
                load_xperm<R>(Zp, ent, i);
                load_xperm<R>(Gp, ent + 2 * R * R, i);
                const double bt = st ? s_b0[t * R + i] : bcol[(size_t)t * R + i];
                const double w = matvec_x<R>(Zp, xi);
                if (c == 0) {
                    dot = fma(xi, w, dot);
                    if (st) s_b0[t * R + i] = w;              // b_t is consumed; the slot now holds w_t
                    else wtab[(size_t)t * R + i] = w;
                }
                xi = matvec_x<R>(Gp, xi, bt);
            }
            if (c == 0) s_vec[i] = xi;
        }
        __syncthreads();
        xi = s_vec[i];
        __syncthreads();                  // s_vec is reused for xi_T
    }
    // xi = xi_ts in every group.
    if (a.abl & 8) { if (xi == 1.2345e300) fout[0] = xi; return; }

    // ---- steady forward scan: steps ts .. T-1; group c owns steps ts + c L + j ---------------------
    {
        double Gp[R], Zp[R];
        load_xperm<R>(Gp, s_mat + 2 * R * R, i);
        double dummy = 0.0;
        // phase 1: chunk from a zero state
        const double e = fullf ? chunk_run<R, true, 0>(Gp, Gp, 0.0, cur, bcol, t0f, 1, L, ts, T, i, nullptr, dummy, nullptr, r)
                               : chunk_run<R, false, 0>(Gp, Gp, 0.0, cur, bcol, t0f, 1, L, ts, T, i, nullptr, dummy, nullptr, r);
        if (a.abl & 16) { if (e == 1.2345e300) fout[0] = e; return; }
        if (fullf) chunk_prefetch<R, true>(cur, bcol, t0f, 1, L, ts, T, i);     // operands of phase 3, in flight during phase 2
        else chunk_prefetch<R, false>(cur, bcol, t0f, 1, L, ts, T, i);
        // phase 2: true start state of chunk c
        const double s = carry_scan<R, NG>(e, xi, s_mat + 3 * R * R, c, i, s_a, s_b);
        if (a.abl & 32) { if (s == 1.2345e300) fout[0] = s; return; }
        // phase 3: re-run from the true start; emit w_t, accumulate xi_t' w_t
        load_xperm<R>(Zp, s_mat, i);
        const double v = fullf ? chunk_run<R, true, 1>(Gp, Zp, s, cur, bcol, t0f, 1, L, ts, T, i, wtab, dot, nullptr, r)
                               : chunk_run<R, false, 1>(Gp, Zp, s, cur, bcol, t0f, 1, L, ts, T, i, wtab, dot, nullptr, r);
        if (c == clast) s_vec[i] = v;                         // xi_T
    }
    if (a.abl & 64) { if (dot == 1.2345e300) fout[0] = dot; return; }
    __syncthreads();   // xi_T in LDS; every w_t of this replicate is written (workgroup-visible)
    if (fullb) chunk_prefetch<R, true>(cur, wtab, t0b, -1, L, ts, T, i);
    else chunk_prefetch<R, false>(cur, wtab, t0b, -1, L, ts, T, i);

    // ---- terminal ------------------------------------------------------------------------------
    const double xiT = s_vec[i];
    double fT;
    {
        double PTp[R];
        load_xperm<R>(PTp, s_mat + NST * R * R, i);
        fT = matvec_x<R>(PTp, xiT);
    }
    if (c == 0) {
        dot = fma(xiT, fT, dot);          // the log-likelihood needs sum xi'w + xi_T' f_T
        if (i < r) fout[(size_t)(T - 1) * r + i] = fT;
    }

    // ---- steady backward scan: steps T-1 .. ts; group c owns steps T-1 - c L - j -------------------
    double fb;   // smoothed mean at the steady/transient boundary (period ts)
    {
        double Jp[R];
        load_xperm<R>(Jp, s_mat + R * R, i);
        double dummy = 0.0;
        const double e = fullb ? chunk_run<R, true, 0>(Jp, Jp, 0.0, cur, wtab, t0b, -1, L, ts, T, i, nullptr, dummy, nullptr, r)
                               : chunk_run<R, false, 0>(Jp, Jp, 0.0, cur, wtab, t0b, -1, L, ts, T, i, nullptr, dummy, nullptr, r);
        if (fullb) chunk_prefetch<R, true>(cur, wtab, t0b, -1, L, ts, T, i);
        else chunk_prefetch<R, false>(cur, wtab, t0b, -1, L, ts, T, i);
        const double s = carry_scan<R, NG>(e, fT, s_mat + (size_t)(3 + NLEV) * R * R, c, i, s_a, s_b);
        const double v = fullb ? chunk_run<R, true, 2>(Jp, Jp, s, cur, wtab, t0b, -1, L, ts, T, i, nullptr, dummy, fout, r)
                               : chunk_run<R, false, 2>(Jp, Jp, s, cur, wtab, t0b, -1, L, ts, T, i, nullptr, dummy, fout, r);
        if (c == clast) s_vec[R + i] = v;
        __syncthreads();
        fb = s_vec[R + i];
    }

    // ---- backward transient: steps ts-1 .. 0 (wave 0 only) -----------------------------------------
    if (tid < 64) {
        double v = fb;
        for (int t = ts - 1; t >= 0; --t) {
            double Jp[R];
            const bool st = t < nst;
            load_xperm<R>(Jp, (st ? s_tab + (size_t)t * 3 * R * R : tab + (size_t)t * 3 * R * R) + R * R, i);
            const double wt = st ? s_b0[t * R + i] : wtab[(size_t)t * R + i];
            v = matvec_x<R>(Jp, v, wt);
            if (c == 0 && t >= 1 && i < r) fout[(size_t)(t - 1) * r + i] = v;
        }
        if (a.f0s && c == 0) a.f0s[(size_t)b * R + i] = v;   // E[f_0 | X] (EM)
    }

    // ---- log-likelihood ----------------------------------------------------------------------------
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, kWave);
    const int wave = tid >> 6;
    if ((tid & 63) == 0) s_red[wave] = dot;
    __syncthreads();
    if (tid == 0) {
        double d = 0.0, sq = 0.0;
#pragma unroll
        for (int w = 0; w < ST / 64; ++w) d += s_red[w];
        if (a.ntile > 0) {
            for (int w = 0; w < a.ntile; ++w) sq += a.scol[(size_t)b * T + w];
        } else {
            for (int w = 0; w < a.nseg; ++w) sq += a.ssum[(size_t)b * kSsumSlots + w];
        }
        a.loglik[b] = -0.5 * (a.llc[b] + sq - d);
    }
}


// ================================================================================================
// em_update_kernel: the transition half of the M-step on the balanced fast path
// ================================================================================================
// Sufficient statistics from the smoother output (means: f_smooth in the padded layout, f0s; covariance sums
// SP11, SU, P0s, P_T from cov_kernel):
//     S11 = sum_{t=1..T} E[f_t f_t' | X],   S10 = sum_{t=1..T} E[f_t f_{t-1}' | X],   S00 = sum_{t=0..T-1} E[f_t f_t' | X]
// then  A = S10 S00^-1,  Q = sym(S11 - A S10') / T,  mu0 = f_0|T,  P0 = sym(P_0|T)  (Shumway-Stoffer 1982), the
// S11, S11^-1 the loadings step needs, and the per-replicate EM bookkeeping -- exactly the epilogue of
// recursion_kernel (recursion.hip), which does the same for panels with missing cells.
// One lane group of R lanes per replicate (lane i = row i), 64 / R replicates per wave.
__host__ __device__ constexpr int em_update_waves(int R) { return R >= 16 ? 4 : 1; }
template <int R>
__global__ __launch_bounds__(64 * em_update_waves(R)) void em_update_kernel(EmUpdArgs a) {
    // NW waves per replicate (1 for Rp <= 8; 4 for the wide states, where one wave walked T / 2 periods of 64 dependent FMAs per
    // lane: 0.79 ms per EM iteration of config 4): lane (i = l % R, slice = NW-wave x l / R) sums row i of f_t f_t' and
    // f_t f_{t-1}' over the periods t = slice (mod NW 64 / R) -- independent loads -- then the slices are folded with
    // xor-shuffles, the waves through LDS (every wave adds the NW partial sums in the same order), and every slice runs the
    // (tiny) epilogue redundantly on its own LDS region; slice 0 of wave 0 writes.
    constexpr int GPW = 64 / R;
    constexpr int NW = em_update_waves(R);
    extern __shared__ __attribute__((aligned(16))) double ems[];
    double* Xs = ems;                                        // [NW * GPW][R * R + 2 R]
    double* red = ems + NW * GPW * (R * R + 2 * R);          // NW > 1: [NW][2][R * R]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / R, i = lane % R;
    const int b = blockIdx.x;
    const bool live = (g == 0 && wave == 0);
    double* X = Xs + (wave * GPW + g) * (R * R + 2 * R);
    const int T = a.T;
    const size_t o = (size_t)b * R * R + (size_t)i * R;
    const double* __restrict__ f = a.fsm + (size_t)b * T * R;
    const double* __restrict__ f0 = a.f0s + (size_t)b * R;

    double M11[R], M10[R];
#pragma unroll
    for (int j = 0; j < R; ++j) { M11[j] = 0.0; M10[j] = 0.0; }
#pragma unroll 2
    for (int t = wave * GPW + g; t < T; t += NW * GPW) {
        double cur[R], prev[R];
        const double* pp = (t == 0) ? f0 : f + (size_t)(t - 1) * R;
#pragma unroll
        for (int j = 0; j < R; ++j) { cur[j] = f[(size_t)t * R + j]; prev[j] = pp[j]; }
        const double ci = f[(size_t)t * R + i];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            M11[j] = fma(ci, cur[j], M11[j]);
            M10[j] = fma(ci, prev[j], M10[j]);
        }
    }
#pragma unroll
    for (int off = R; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            M11[j] += __shfl_xor(M11[j], off, kWave);
            M10[j] += __shfl_xor(M10[j], off, kWave);
        }
    }
    if constexpr (NW > 1) {
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < R; ++j) { red[(wave * 2 + 0) * R * R + i * R + j] = M11[j]; red[(wave * 2 + 1) * R * R + i * R + j] = M10[j]; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < R; ++j) {
            double s1 = 0.0, s0 = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { s1 += red[(w * 2 + 0) * R * R + i * R + j]; s0 += red[(w * 2 + 1) * R * R + i * R + j]; }
            M11[j] = s1; M10[j] = s0;
        }
    }
    const double f0i = f0[i];
    const double fTi = f[(size_t)(T - 1) * R + i];
    double fT[R];
#pragma unroll
    for (int j = 0; j < R; ++j) fT[j] = f[(size_t)(T - 1) * R + j];
    double S11[R], S10[R], S00[R], P0s[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        S11[j] = a.SP11[o + j] + M11[j];
        S10[j] = a.SU[o + j] + M10[j];
        P0s[j] = a.P0s[o + j];
        S00[j] = S11[j] - fma(fTi, fT[j], a.PT[o + j]) + fma(f0i, a.f0s[(size_t)b * R + j], P0s[j]);
    }
    // EM bookkeeping (oracle/kalman_oracle.py em()): record ll_k; stop WITHOUT applying this M-step when the
    // relative improvement over ll_{k-1} is below tol
    bool em_apply = true;
    if (a.active) {
        const double ll = a.loglik[b];
        const bool was = a.k == 0 ? true : (a.active[b] != 0);
        bool go = was;
        if (was && a.k >= 1 && a.tol > 0.0) {
            const double llp = a.ll_path[(size_t)b * a.max_iter + a.k - 1];
            go = !((ll - llp) / (0.5 * (fabs(ll) + fabs(llp))) < a.tol);
        }
        em_apply = go;
        __syncthreads();                                     // every lane has read active / ll_path
        if (live && i == 0) {
            if (was) { a.ll_path[(size_t)b * a.max_iter + a.k] = ll; a.iters[b] = a.k + 1; }
            a.active[b] = go ? 1 : 0;
        }
    }
    double inv[R], An[R], tmp[R], Qn[R], P0n[R];
#pragma unroll
    for (int j = 0; j < R; ++j) inv[j] = S00[j];
    (void)gj_inverse<R>(inv, X, i);
    __syncthreads();
    store_row<R>(X, i, inv);
    __syncthreads();
    mm_rows<R>(An, S10, X);                                  // A row i
    __syncthreads();
    store_row<R>(X, i, S10);
    __syncthreads();
    mm_rowsT<R>(tmp, An, X);                                 // (A S10')[i][:]
#pragma unroll
    for (int j = 0; j < R; ++j) Qn[j] = (S11[j] - tmp[j]) / (double)T;
    __syncthreads();
    store_row<R>(X, i, Qn);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; ++j) Qn[j] = 0.5 * (Qn[j] + X[j * R + i]);
    __syncthreads();
    store_row<R>(X, i, P0s);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; ++j) P0n[j] = 0.5 * (P0s[j] + X[j * R + i]);
#pragma unroll
    for (int j = 0; j < R; ++j) inv[j] = S11[j];
    (void)gj_inverse<R>(inv, X, i);
    if (live) {
#pragma unroll
        for (int j = 0; j < R; ++j) { a.S11[o + j] = S11[j]; a.S11inv[o + j] = inv[j]; }
        if (em_apply) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                a.A_out[o + j] = An[j];
                a.Q_out[o + j] = Qn[j];
                a.P0_out[o + j] = P0n[j];
            }
            a.mu0_out[(size_t)b * R + i] = f0i;
        }
    }
}

template <int R>
static hipError_t launch_em_update_r(const EmUpdArgs& a, hipStream_t s) {
    constexpr int NW = em_update_waves(R), GPW = 64 / R;
    const size_t lds = ((size_t)NW * GPW * (R * R + 2 * R) + (NW > 1 ? (size_t)NW * 2 * R * R : 0)) * sizeof(double);
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&em_update_kernel<R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((em_update_kernel<R>), dim3(a.B), dim3(64 * NW), lds, s, a);
    return hipGetLastError();
}
hipError_t launch_em_update(int Rpad, const EmUpdArgs& a, hipStream_t s) {
    static const bool old_wide = [] { const char* v = diag_env("DFM_EM_UPDATE_OLD"); return v && atoi(v) != 0; }();
    if (!old_wide && em_update_grid_supported(Rpad)) return launch_em_update_grid(Rpad, a, s);   // em_update_grid.hip
    switch (Rpad) {
        case 2: return launch_em_update_r<2>(a, s);
        case 4: return launch_em_update_r<4>(a, s);
        case 8: return launch_em_update_r<8>(a, s);
        case 16: return launch_em_update_r<16>(a, s);
        case 32: return launch_em_update_r<32>(a, s);
        default: return hipErrorInvalidValue;
    }
}

// ================================================================================================
template <int R, int CPL2>
static hipError_t launch_cov_rc(const FastArgs& a, hipStream_t s) {
    using LY = CovLayout<R>;
    constexpr int kCovThreads = cov_threads(R);
    constexpr int per_wg = LY::GPW * (kCovThreads / 64);
    const int grid = (a.B + per_wg - 1) / per_wg;
    static LdsOptIn attr_done;
    if (!attr_done && LY::lds_bytes() > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cov_kernel<R, CPL2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((cov_kernel<R, CPL2>), dim3(grid), dim3(kCovThreads), LY::lds_bytes(), s, a);
    return hipGetLastError();
}
// fused Gram (a.Lam != nullptr) for the series tilings whose weights fit the register file next to the
// covariance state; otherwise Cfull / ldfull come from gram_kernel
template <int R>
static hipError_t launch_cov_r(const FastArgs& a, hipStream_t s) {
    if (a.Lam != nullptr) {
        if (a.N <= 128) return launch_cov_rc<R, 1>(a, s);
        if constexpr (R <= 16) {
            if (a.N <= 256) return launch_cov_rc<R, 2>(a, s);
        }
        return hipErrorInvalidValue;
    }
    return launch_cov_rc<R, 0>(a, s);
}
bool cov_fuses_gram(int Rpad, int N) { return N <= 128 || (Rpad <= 16 && N <= 256); }
template <int R>
static hipError_t launch_scan_r(const FastArgs& a, hipStream_t s) {
    const size_t lds = ScanLds<R>::bytes();
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&meanscan_kernel<R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((meanscan_kernel<R>), dim3(a.B), dim3(scan_threads(R)), lds, s, a);
    return hipGetLastError();
}

int fast_stead_mats(int Rpad) { return stead_mats(Rpad); }
int fast_scan_groups(int Rpad) { return scan_groups(Rpad); }

int fast_chunk_len(int Rpad, int T) {
    const int ng = scan_groups(Rpad);
    int L = 1;
    while (L * ng < T) L <<= 1;
    return L;
}

hipError_t launch_cov(int Rpad, const FastArgs& a, hipStream_t s) {
    static const bool rows_only = [] { const char* v = diag_env("DFM_COV_ROWS"); return v && atoi(v) != 0; }();
    if (!rows_only && cov_tile_supported(Rpad, a, a.rstate > 0 ? a.rstate : Rpad)) return launch_cov_tile(a, a.rstate > 0 ? a.rstate : Rpad, s);
    if (!rows_only && a.Lam == nullptr && cov_grid_supported(Rpad)) return launch_cov_grid(Rpad, a, s);
    switch (Rpad) {
        case 2: return launch_cov_r<2>(a, s);
        case 4: return launch_cov_r<4>(a, s);
        case 8: return launch_cov_r<8>(a, s);
        case 16: return launch_cov_r<16>(a, s);
        case 32: return launch_cov_r<32>(a, s);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_pfill(int Rpad, const FastArgs& a, hipStream_t s) {
    if (!a.P_smooth) return hipSuccess;
    // slices per replicate: enough workgroups to fill the chip, each with at least ~64 KB of rows
    const long long bytes = (long long)a.T * (a.r * (a.r + 1) / 2) * 8;
    int ns = (int)((2048 + a.B - 1) / a.B);
    while (ns > 1 && bytes / ns < 65536) --ns;
    if (ns < 1) ns = 1;
    const dim3 grid(a.B, ns);
    switch (Rpad) {
        case 2: hipLaunchKernelGGL((pfill_kernel<2>), grid, dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((pfill_kernel<4>), grid, dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL((pfill_kernel<8>), grid, dim3(256), 0, s, a); break;
        case 16: hipLaunchKernelGGL((pfill_kernel<16>), grid, dim3(256), 0, s, a); break;
        case 32: hipLaunchKernelGGL((pfill_kernel<32>), grid, dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_meanscan(int Rpad, const FastArgs& a, hipStream_t s) {
    switch (Rpad) {
        case 2: return launch_scan_r<2>(a, s);
        case 4: return launch_scan_r<4>(a, s);
        case 8: return launch_scan_r<8>(a, s);
        case 16:
        case 32: return launch_meanscan_mfma(Rpad, a, s);  // scan_mfma32.hip (128 chunks on the matrix pipe; meanscan_kernel's lane groups hold 16)
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dfm
