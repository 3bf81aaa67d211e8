// dfm_cov.h -- the data-independent covariance half of the balanced fast path as a device function (see
// fastpath.hip for the algorithm): forward Riccati steps to the fixed point, terminal P_T and log-determinants,
// backward steps, powers of the steady matrices for the scan's chunk carries.
#pragma once
#include "dfm_gram.h"
#include "dfm_kernels.h"
#include "dfm_smallmat.h"

namespace dfm {

constexpr double kLog2PiF = 1.8378770664093454835606594728112;
constexpr int kScanThreads = 256;   // meanscan workgroup: 256 / R lane groups = time chunks
__host__ __device__ constexpr int cov_threads(int R) { (void)R; return 64; }
// levels of the carry scan over the 256 / R chunks, and matrices kept per replicate in `stead`:
// Z, J, G, then G^(L 2^k) and J^(L 2^k), k = 0 .. levels-1
// Rp = 32: 512 threads = 16 time chunks (8 chunks of 256 periods made the scan of config 4 a 0.65-ms chain of 1024 dependent
// 32 x 32 matrix-vector steps; 32 chunks would need 14 staged 8-KB matrices: over the LDS budget)
__host__ __device__ constexpr int scan_threads(int R) { return R >= 32 ? 512 : kScanThreads; }
// time chunks of the steady scan.  Rp = 16, 32 run on the matrix pipe (scan_mfma32.hip), 16 chunks = columns per wave: 8 waves
// at Rp = 32; 4 at Rp = 16 (148 VGPRs: three 4-wave workgroups share a CU, an 8-wave one would have it alone)
__host__ __device__ constexpr int scan_groups(int R) { return R >= 32 ? 128 : R == 16 ? 64 : scan_threads(R) / R; }
__host__ __device__ constexpr int scan_levels(int R) { int n = 0; while ((1 << n) < scan_groups(R)) ++n; return n; }
__host__ __device__ constexpr int stead_mats(int R) { return 3 + 2 * scan_levels(R); }

// ================================================================================================
// cov_kernel
// ================================================================================================
template <int R, int ECL_ = (R <= 8 ? 8 : R <= 16 ? 2 : 0)>
struct CovLayout {
    static constexpr int GPW = 64 / R;
    // per group: X, PSI, JS (exchange / operands), K0..K2 (forward: Q^-1, Phi, C; backward: sum P_s, sum U), V0
    // + TB: Z_e, J_e of the first ECL covariance steps (the backward pass re-reads them; beside the streaming
    // collapse a global round trip costs ~5 us)
    static constexpr int ECL = ECL_;
    static constexpr int kRaw = 6 * R * R + R + ECL * 2 * R * R;
    static constexpr int S = ((kRaw + 3) / 4) * 4 + 2;
    static constexpr size_t lds_bytes() { return (size_t)(cov_threads(R) / 64) * GPW * S * sizeof(double); }
};

// CPL2 > 0: the Gram matrices C = Lam' R^-1 Lam and sum log R of the wave's replicates are computed here first,
// one after the other by the whole wave (lane l owns series {2l, 2l+1} + 128 j, j < CPL2), instead of by a
// separate gram_kernel launch: beside the streaming collapse, which fills every CU, a second dependent launch
// on the side stream waits ~150 us for free registers.
// cov_body: ONE wave, replicates wave_first .. wave_first + 64/R - 1, LDS region wsm (CovLayout<R, ECL>::S doubles
// per lane group).  Called by cov_kernel (fastpath.hip) and by the covariance workgroups at the front of the fused
// collapse launch (collapse_mfma.hip).
template <int R, int CPL2, int ECL>
__device__ __forceinline__ void cov_body(const FastArgs& a, int wave_first, double* wsm, int lane) {
    using LY = CovLayout<R, ECL>;
    constexpr int GPW = LY::GPW;
    const int g = lane / R, i = lane % R;
    const int T = a.T, r = a.r;
    int b = wave_first + g;
    const bool live = b < a.B;
    if (!live) b = a.B - 1;

    double* X = wsm + (size_t)g * LY::S;
    double* PSI = X + R * R;
    double* JS = PSI + R * R;
    double* K0 = JS + R * R;       // rows owned by lane i: K0[i*R + j]
    double* K1 = K0 + R * R;
    double* K2 = K1 + R * R;
    double* V0 = K2 + R * R;
    double* TB = V0 + R;           // [ECL][2][R][R]: rows i of Z_e, J_e (lane i reads back only what it wrote)

    const size_t mo = (size_t)b * R * R + (size_t)i * R;     // row i of a [B][R][R] array
    double* tab = a.tab + (size_t)b * T * 3 * R * R;
    double ldfull_b = 0.0;
    if constexpr (CPL2 > 0) {
        // C = Lam' R^-1 Lam, row i in lane i of the replicate's lane group: every series is read by the whole
        // group (same address: one request), lane i accumulates (lam_ci / R_c) lam_c.  All loads independent.
        const int N = a.N;
        const double* __restrict__ L = a.Lam + (size_t)b * N * R;
        const double* __restrict__ Rv = a.Rv + (size_t)b * N;
        double crow[R];
#pragma unroll
        for (int j = 0; j < R; ++j) crow[j] = 0.0;
        constexpr int UN = 4;
        for (int c0 = 0; c0 < N; c0 += UN) {
            double lam[UN][R], rvv[UN], li[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int c = (c0 + u < N) ? c0 + u : N - 1;
                rvv[u] = Rv[c];
                li[u] = L[(size_t)c * R + i];
#pragma unroll
                for (int j = 0; j < R; ++j) lam[u][j] = L[(size_t)c * R + j];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const double wi = (c0 + u < N) ? li[u] / rvv[u] : 0.0;
#pragma unroll
                for (int j = 0; j < R; ++j) crow[j] = fma(wi, lam[u][j], crow[j]);
            }
        }
        double ld = 0.0;                                     // sum log R: series i, i + R, ... then over the group
        for (int c = i; c < N; c += R) ld += log(Rv[c]);
#pragma unroll
        for (int off = 1; off < R; off <<= 1) ld += __shfl_xor(ld, off, kWave);
        ldfull_b = ld;
#pragma unroll
        for (int j = 0; j < R; ++j) K2[i * R + j] = crow[j];
    }
    double detQ, detP0, q0;
    double PsiT[R], Omf[R];
    {
        double Arow[R], Qi[R], Phi[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            Arow[j] = a.A[mo + j];
            Qi[j] = a.Q[mo + j];
            Omf[j] = a.P0[mo + j];
            if constexpr (CPL2 == 0) K2[i * R + j] = a.Cfull[mo + j];
        }
        const double mu0i = a.mu0[(size_t)b * R + i];
        detQ = gj_inverse<R, true>(Qi, X, i);
        detP0 = gj_inverse<R, true>(Omf, X, i);                    // Omf = P0^-1
        wave_lds_sync();
        store_row<R>(X, i, Arow);
        wave_lds_sync();
        mm_rows<R>(PsiT, Qi, X);                             // Psi' = Qi A (row i)
#pragma unroll
        for (int j = 0; j < R; ++j) PSI[j * R + i] = PsiT[j];  // PSI = Psi (rows)
        wave_lds_sync();
        {
            double prow[R];
#pragma unroll
            for (int k = 0; k < R; ++k) prow[k] = PSI[i * R + k];
            mm_rows<R>(Phi, prow, X);                        // Phi = Psi A
        }
        store_row<R>(K0, i, Qi);
        store_row<R>(K1, i, Phi);
        V0[i] = mu0i;
        wave_lds_sync();
        const double xi0 = dot_vec<R>(Omf, V0);              // xi_0 = P0^-1 mu0
        q0 = mu0i * xi0;                                     // lane part of mu0' P0^-1 mu0
        if (live) a.xi0[(size_t)b * R + i] = xi0;
        wave_lds_sync();
    }

    // ---------------- forward covariance steps until the fixed point ----------------------------
    double sum_ldz = 0.0, ldz_last = 0.0;
    int E = 0;
    {
        bool done = false;
        for (int e = 0;; ++e) {
            double Z[R], Jr[R];
#pragma unroll
            for (int j = 0; j < R; ++j) Z[j] = Omf[j] + K1[i * R + j];
            const double detM = gj_inverse<R, true>(Z, X, i);
            const double ldz = -log(detM);
            mm_rows<R>(Jr, Z, PSI);                          // J = Z Psi
            wave_lds_sync();
            store_row<R>(X, i, Jr);
            wave_lds_sync();
            double Omf_new[R];
            {
                double tmp[R];
                mm_rows<R>(tmp, PsiT, X);                    // Psi' J
#pragma unroll
                for (int j = 0; j < R; ++j) Omf_new[j] = (K0[i * R + j] - tmp[j]) + K2[i * R + j];   // Om_p + C
            }
            bool same = true;
#pragma unroll
            for (int j = 0; j < R; ++j) same = same && close_enough(Omf_new[j], Omf[j]);
            V0[i] = same ? 1.0 : 0.0;
            wave_lds_sync();
            bool gsame = true;
#pragma unroll
            for (int k = 0; k < R; ++k) gsame = gsame && (V0[k] != 0.0);
            store_row<R>(X, i, Z);
            wave_lds_sync();
            if (!done) {
                double G[R];
                mm_rows<R>(G, PsiT, X);                      // G = Psi' Z
                if (live) {
                    double* te = tab + (size_t)e * 3 * R * R + (size_t)i * R;
#pragma unroll
                    for (int j = 0; j < R; ++j) { te[j] = Z[j]; te[R * R + j] = Jr[j]; te[2 * R * R + j] = G[j]; }
                }
                if (e < LY::ECL) {
#pragma unroll
                    for (int j = 0; j < R; ++j) { TB[(e * 2) * R * R + i * R + j] = Z[j]; TB[(e * 2 + 1) * R * R + i * R + j] = Jr[j]; }
                }
                E = e + 1;
                sum_ldz += ldz;
                ldz_last = ldz;
#pragma unroll
                for (int j = 0; j < R; ++j) Omf[j] = Omf_new[j];
                if (gsame || e + 1 >= T) done = true;
            }
            if (__all(done)) break;
        }
    }
    sum_ldz += (double)(T - E) * ldz_last;
    const int ts = E - 1;                                     // first steady step

    // ---------------- terminal ---------------------------------------------------------------------
    double Ps[R];
#pragma unroll
    for (int j = 0; j < R; ++j) Ps[j] = Omf[j];
    const double detOmT = gj_inverse<R, true>(Ps, X, i);           // P_T
    wave_lds_sync();
    V0[i] = q0;
    wave_lds_sync();
    {
        double qs = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) qs += V0[k];
        q0 = qs;
    }
    wave_lds_sync();
    if (live) {
#pragma unroll
        for (int j = 0; j < R; ++j) a.PT[mo + j] = Ps[j];
        if (i == 0) {
            const double LD = log(detOmT) + log(detP0) + (double)T * log(detQ) - sum_ldz;
            a.llc[b] = (double)a.N * (double)T * kLog2PiF + (double)T * (CPL2 > 0 ? ldfull_b : a.ldfull[b]) + LD + q0;
            a.E[b] = E;
        }
    }

    // ---------------- backward covariance steps ----------------------------------------------------
    const int npr = r * (r + 1) / 2;
    auto emit = [&](int trow, const double (&P)[R]) {
        if (!live || i >= r || a.P_smooth == nullptr) return;
        double* po = a.P_smooth + ((size_t)b * T + trow) * npr + i * (i + 1) / 2;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (j <= i) po[j] = P[j];
    };
    emit(T - 1, Ps);
    double* SPl = K0;   // sum over periods 1..T of P_s   (row i owned by lane i)
    double* SUl = K1;   // sum over steps 0..T-1 of U_t = Cov(f_{t+1}, f_t | X)
#pragma unroll
    for (int j = 0; j < R; ++j) { SPl[i * R + j] = Ps[j]; SUl[i * R + j] = 0.0; }
    int fill_lo = 0, fill_hi = 0;
    int t = T - 1;
    int cur_e = -2;                                           // entry whose Z, J are loaded
    double Zc[R], Jc[R];
#pragma unroll
    for (int j = 0; j < R; ++j) { Zc[j] = 0.0; Jc[j] = 0.0; }
    while (true) {
        const bool act = t >= 0;
        const int e = act ? (t < ts ? t : ts) : ts;
        if (e != cur_e) {                                     // (group-uniform: t, ts are per group)
            if (e < LY::ECL) {
#pragma unroll
                for (int j = 0; j < R; ++j) { Zc[j] = TB[(e * 2) * R * R + i * R + j]; Jc[j] = TB[(e * 2 + 1) * R * R + i * R + j]; }
            } else {
                const double* te = tab + (size_t)e * 3 * R * R + (size_t)i * R;
#pragma unroll
                for (int j = 0; j < R; ++j) { Zc[j] = te[j]; Jc[j] = te[R * R + j]; }
            }
            cur_e = e;
        }
        wave_lds_sync();
        store_row<R>(JS, i, Jc);
        wave_lds_sync();
        double U[R], Psn[R];
        mm_rowsT<R>(U, Ps, JS);                              // U = P_s J' = Cov(f_{t+1}, f_t | X)
        store_row<R>(X, i, U);
        wave_lds_sync();
        {
            double tmp[R];
            mm_rows<R>(tmp, Jc, X);                          // J U
#pragma unroll
            for (int j = 0; j < R; ++j) Psn[j] = Zc[j] + tmp[j];
        }
        bool same = true;
#pragma unroll
        for (int j = 0; j < R; ++j) same = same && close_enough(Psn[j], Ps[j]);
        V0[i] = same ? 1.0 : 0.0;
        wave_lds_sync();
        bool gsame = true;
#pragma unroll
        for (int k = 0; k < R; ++k) gsame = gsame && (V0[k] != 0.0);
        if (act) {
            const bool skip = (e == ts && t > ts && gsame);   // steps t-1 .. ts repeat this (U, P_s)
            const int plo = ts >= 1 ? ts : 1;                // periods plo .. t-1 carry P_s,inf
            const double cu = skip ? (double)(t - ts + 1) : 1.0;
            const double cp = (t >= 1 ? 1.0 : 0.0) + (skip ? (double)(t - plo) : 0.0);
#pragma unroll
            for (int j = 0; j < R; ++j) {
                SUl[i * R + j] = fma(cu, U[j], SUl[i * R + j]);
                SPl[i * R + j] = fma(cp, Psn[j], SPl[i * R + j]);
            }
            if (t >= 1) emit(t - 1, Psn);
            if (live && (t == 0 || (skip && ts == 0)) && a.P0s) {
#pragma unroll
                for (int j = 0; j < R; ++j) a.P0s[mo + j] = Psn[j];
            }
            if (skip) {
                if (live) {
#pragma unroll
                    for (int j = 0; j < R; ++j) a.PsInf[mo + j] = Psn[j];
                }
                fill_lo = plo - 1;
                fill_hi = t - 1;
                t = ts - 1;
            } else {
                t -= 1;
            }
#pragma unroll
            for (int j = 0; j < R; ++j) Ps[j] = Psn[j];
        }
        if (__all(t < 0)) break;
    }
    if (live) {
        if (i == 0) { a.fill[2 * b] = fill_lo; a.fill[2 * b + 1] = fill_hi; }
        if (a.SP11) {
#pragma unroll
            for (int j = 0; j < R; ++j) { a.SP11[mo + j] = SPl[i * R + j]; a.SU[mo + j] = SUl[i * R + j]; }
        }
    }

    // ---------------- steady Z, J, G and the powers G^(L 2^k), J^(L 2^k) for the chunk carries --------
    {
        constexpr int NLEV = scan_levels(R);
        const double* te = tab + (size_t)ts * 3 * R * R + (size_t)i * R;
        double* st = a.stead + (size_t)b * stead_mats(R) * R * R + (size_t)i * R;
        double M[R], Ms[2][R];                               // one batch of loads: a single round trip
#pragma unroll
        for (int j = 0; j < R; ++j) { M[j] = te[j]; Ms[0][j] = te[2 * R * R + j]; Ms[1][j] = te[R * R + j]; }
        if (live) {
#pragma unroll
            for (int j = 0; j < R; ++j) st[j] = M[j];
        }
#pragma unroll
        for (int which = 0; which < 2; ++which) {            // 0: G -> slot 2, powers 3..;  1: J -> slot 1, powers 3+NLEV..
            const int src = which == 0 ? 2 : 1, dst = which == 0 ? 3 : 3 + NLEV;
#pragma unroll
            for (int j = 0; j < R; ++j) M[j] = Ms[which][j];
            if (live) {
#pragma unroll
                for (int j = 0; j < R; ++j) st[src * R * R + j] = M[j];
            }
            auto square = [&]() {
                double tmp[R];
                wave_lds_sync();
                store_row<R>(X, i, M);
                wave_lds_sync();
                mm_rows<R>(tmp, M, X);
#pragma unroll
                for (int j = 0; j < R; ++j) M[j] = tmp[j];
            };
            for (int l = 1; l < a.L; l <<= 1) square();       // M^L
#pragma unroll 1
            for (int k = 0; k < NLEV; ++k) {
                if (live) {
#pragma unroll
                    for (int j = 0; j < R; ++j) st[(dst + k) * R * R + j] = M[j];
                }
                if (k + 1 < NLEV) square();
            }
        }
    }
}

// Rows [lo, hi) of P_smooth equal the backward fixed point P_s,inf: element k of the range is s_ps[k % npr]
// (packed lower triangle in the caller's r).  Plain 16-byte stores, nothing waits for them.
__device__ __forceinline__ void fill_psmooth_range(const FastArgs& a, int b, int tid, int nthreads, const double* s_ps, int lo,
                                                   int hi) {
    const int npr = a.r * (a.r + 1) / 2;
    if (hi <= lo) return;
    double* base = a.P_smooth + ((size_t)b * a.T + lo) * npr;
    const unsigned n = (unsigned)(hi - lo) * (unsigned)npr;
    const unsigned peel = ((reinterpret_cast<size_t>(base) & 15) != 0) ? 1u : 0u;   // to 16-byte alignment
    if (peel && tid == 0) base[0] = s_ps[0];
    const unsigned npair = (n - peel) / 2;
    const unsigned step = (2u * nthreads) % (unsigned)npr;
    unsigned k = peel + 2u * tid;
    unsigned v = k % (unsigned)npr;
    for (unsigned p = tid; p < npair; p += nthreads) {
        const unsigned v1 = (v + 1 == (unsigned)npr) ? 0u : v + 1;
        *reinterpret_cast<double2*>(base + k) = make_double2(s_ps[v], s_ps[v1]);
        k += 2u * nthreads;
        v += step;
        if (v >= (unsigned)npr) v -= (unsigned)npr;
    }
    if (((n - peel) & 1u) != 0 && tid == 0) base[n - 1] = s_ps[(n - 1) % (unsigned)npr];
}
__device__ __forceinline__ void fill_psmooth_rows(const FastArgs& a, int b, int tid, int nthreads, const double* s_ps) {
    fill_psmooth_range(a, b, tid, nthreads, s_ps, a.fill[2 * b], a.fill[2 * b + 1]);
}


}  // namespace dfm
