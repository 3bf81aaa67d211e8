// dfm_cov8.h -- the data-independent covariance half of the balanced fast path (see fastpath.hip / dfm_cov.h for the
// algorithm) with ONE WAVE PER REPLICATE at Rp = 8: lane l = 8 i + j holds ELEMENT (i, j) of every 8 x 8 matrix
// (dfm_grid.h: sweep-operator inverse over the LDS crossbar, products through 640-byte LDS tiles).
//
// cov_body (dfm_cov.h) gives a replicate 8 lanes (lane = matrix row): 8 replicates per wave, ~75 us alone and ~125 us
// beside the streaming collapse for a chain of ~35 dependent 8 x 8 operations.  That is fine while the covariance
// workgroups only have to finish before the scan LAUNCH, but the one-launch pass (pass_fused.hip) runs the covariance
// recursion of a replicate beside the ~35 us its own panel takes to stream through one CU: it needs the chain in ~10 us.
// Element per lane: an inverse is 4 block pivots of ~15 FMAs per lane instead of 8 pivots of 8; a product is 8 LDS reads
// + 8 FMAs per lane instead of 64.
//
// Outputs are exactly cov_body's (same tables, same bookkeeping of the fixed points, same P_smooth rows), written through
// Cov8Dst so that they can live in global memory (cov_wave_kernel: drop-in for cov_kernel) or in LDS (pass_fused.hip).
// The reference has no counterpart (dfm_functions.ipynb:21-23 declares `Parametric` only).
#pragma once
#include "dfm_cov.h"
#include "dfm_grid.h"

namespace dfm {

constexpr int kCov8TileDoubles = 8 * kTileStride<8>;              // one staged 8 x 8 matrix (rows 10 doubles apart)
constexpr int kCov8ScratchDoubles = 5 * kCov8TileDoubles + 64;    // L0, L1, LP (Psi' rows), LJ, spare + the Gram matrix
constexpr int kCov8Keep = 8;                                      // transient steps whose Z_e, J_e stay in registers

struct Cov8Dst {                 // every matrix row-major [8][8] = indexed by the lane
    double* tab;                 // entries e < tab_cap: tab + e * 3 * 64 = {Z_e, J_e, G_e}
    int tab_cap;
    double* tab_over;            // entries e >= tab_cap: tab_over + e * 3 * 64 (global [T][3][64]); may equal tab
    double* stead;               // [stead_mats][64]: Z, J, G, G^(L 2^k), J^(L 2^k)
    double* PT;                  // [64]
    double* xi0;                 // [8]
    double* llc;                 // [1]
    int* E;                      // [1]
    int* fill;                   // [2]
    double* PsInf;               // [64]
    double* SP11; double* SU; double* P0s;   // EM covariance sums [64] each, or null
};

// R x R threads (one wave at R = 8, a workgroup of 256 / 1024 threads at R = 16 / 32), replicate b.  G: the thread grid with
// l, i, j (and, R >= 16, its LDS exchange buffers) set.  Cel = element (i, j) of C = Lam' R^-1 Lam, ldfull = sum_i log R_i.
// wsm: 4 tiles of R x kTileStride<R> doubles of LDS private to the replicate.  NLEV = levels of the scan's carry tree;
// KEEP = transient steps whose Z_e, J_e stay in registers.  Matrices of Cov8Dst: row-major [R][R], element l = R i + j.
template <int R, int NLEV, int KEEP>
__device__ __forceinline__ void cov_grid(const FastArgs& a, int b, double Cel, double ldfull, double* wsm, const Cov8Dst& o,
                                         Grid<R>& G) {
    constexpr int TS = kTileStride<R>, RR = R * R, RT = R * kTileStride<R>;
    const int lane = G.l, i = G.i, j = G.j;                  // ("lane" = thread of the grid)
    double* L0 = wsm;
    double* L1 = L0 + RT;
    double* LPT = L1 + RT;                   // rows of Psi' = Q^-1 A   (constant)
    double* LJ = LPT + RT;                   // rows of the current J_e (backward sweep)
    const int T = a.T, r = a.r;
    const bool diag = (i == j);
    const size_t mo = (size_t)b * RR + lane;

    const double Ael = a.A[mo];
    double Qi = a.Q[mo];
    double Omf = a.P0[mo];
    const double mu0c = a.mu0[(size_t)b * R + j];               // column-distributed
    const double detQ = G.sweep_inverse(Qi);                    // Qi = Q^-1
    const double detP0 = G.sweep_inverse(Omf);                  // Om_f,0 = P0^-1
    // Psi' = Qi A:  (Qi A)_ij = row i of Qi . row j of A'
    L0[TS * i + j] = Qi;
    L1[TS * j + i] = Ael;                                       // A'
    G.sync();
    const double PsiT = dot_rows<R>(L0, L1, i, j);
    G.sync();
    LPT[TS * i + j] = PsiT;
    L0[TS * j + i] = PsiT;                                      // rows of Psi = columns of Psi'
    G.sync();
    const double Phi = dot_rows<R>(L0, L1, i, j);               // Phi = Psi A = A' Qi A
    double q0;
    {
        const double xi0r = G.sum_j(Omf * mu0c);                // xi_0 = P0^-1 mu0 (row-distributed: lane (i, .) holds xi0_i)
        if (j == 0) o.xi0[i] = xi0r;
        double part = diag ? mu0c * xi0r : 0.0;                 // mu0_i xi0_i on the diagonal lanes
        part = G.sum_j(part);
        q0 = G.sum_i(part);                                     // mu0' P0^-1 mu0 on every lane
    }
    G.sync();

    // ---------------- forward covariance steps until the fixed point -------------------------------------------
    LogProd detprod;                                            // prod over the E distinct steps of det(Om_f + Phi)
    double detM_last = 1.0;
    int E = 0;
    // Z_e, J_e of the first KEEP steps stay in registers for the backward sweep (one element per lane each): read
    // back from the table they cost a global round trip per distinct step -- 5 to 8 us each beside streaming waves
    double Zk[KEEP], Jk[KEEP];
#pragma unroll
    for (int u = 0; u < KEEP; ++u) { Zk[u] = 0.0; Jk[u] = 0.0; }
    double Zlast = 0.0, Jlast = 0.0, Glast = 0.0;               // entry E - 1 = the steady matrices
    for (int e = 0;; ++e) {
        double Z = Omf + Phi;
        const double detM = G.sweep_inverse(Z);                 // Z = (Om_f + Phi)^-1
        L0[TS * i + j] = Z;
        G.sync();
        const double Jr = dot_rows<R>(L0, LPT, i, j);           // J = Z Psi:  row i of Z . row j of Psi'
        L1[TS * j + i] = Jr;                                    // J' rows
        G.sync();
        const double tmp = dot_rows<R>(LPT, L1, i, j);          // Psi' J
        const double Gm = dot_rows<R>(LPT, L0, i, j);           // G = Psi' Z   (Z symmetric to rounding)
        const double Omf_new = (Qi - tmp) + Cel;                // Om_p + C
        const bool gsame = G.all_true(close_enough(Omf_new, Omf));
        double* te = (e < o.tab_cap ? o.tab : o.tab_over) + (size_t)e * 3 * RR;
        te[lane] = Z; te[RR + lane] = Jr; te[2 * RR + lane] = Gm;
#pragma unroll
        for (int u = 0; u < KEEP; ++u) {
            Zk[u] = (u == e) ? Z : Zk[u];
            Jk[u] = (u == e) ? Jr : Jk[u];
        }
        Zlast = Z; Jlast = Jr; Glast = Gm;
        E = e + 1;
        detprod.mul(detM);
        detM_last = detM;
        Omf = Omf_new;
        G.sync();                                               // L0 / L1 are free again
        if (gsame || e + 1 >= T) break;
    }
    const int ts = E - 1;                                       // first steady step

    // ---------------- terminal -------------------------------------------------------------------------------
    double Ps = Omf;
    const double detOmT = G.sweep_inverse(Ps);                  // P_T
    o.PT[lane] = Ps;
    if (lane == 0) {
        const double sum_ldz = -(detprod.log_value() + (double)(T - E) * log(detM_last));
        const double LD = log(detOmT) + log(detP0) + (double)T * log(detQ) - sum_ldz;
        o.llc[0] = (double)a.N * (double)T * kLog2PiF + (double)T * ldfull + LD + q0;
        o.E[0] = E;
    }

    // ---------------- backward covariance steps --------------------------------------------------------------
    const int npr = r * (r + 1) / 2;
    const bool pout = a.P_smooth != nullptr && i < r && j <= i;
    double* prow = a.P_smooth ? a.P_smooth + (size_t)b * T * npr + i * (i + 1) / 2 + j : nullptr;
    auto emit = [&](int trow, double P) { if (pout) prow[(size_t)trow * npr] = P; };
    emit(T - 1, Ps);
    double SP = Ps, SU = 0.0;       // sum over periods 1..T of P_s; sum over steps 0..T-1 of U_t = Cov(f_{t+1}, f_t | X)
    int fill_lo = 0, fill_hi = 0;
    int t = T - 1, cur_e = -2;
    double Zc = 0.0, Jc = 0.0;
    while (t >= 0) {
        const int e = t < ts ? t : ts;
        if (e != cur_e) {                                       // wave-uniform
            if (e == ts) {
                Zc = Zlast; Jc = Jlast;
            } else if (e < KEEP) {
#pragma unroll
                for (int u = 0; u < KEEP; ++u) {
                    Zc = (u == e) ? Zk[u] : Zc;
                    Jc = (u == e) ? Jk[u] : Jc;
                }
            } else {
                const double* te = (e < o.tab_cap ? o.tab : o.tab_over) + (size_t)e * 3 * RR;
                Zc = te[lane]; Jc = te[RR + lane];
            }
            cur_e = e;
            LJ[TS * i + j] = Jc;
        }
        L0[TS * i + j] = Ps;
        G.sync();
        const double U = dot_rows<R>(L0, LJ, i, j);             // U = P_s J'
        L1[TS * j + i] = U;                                     // U' rows
        G.sync();
        const double Psn = Zc + dot_rows<R>(LJ, L1, i, j);      // Z + J U
        const bool gsame = G.all_true(close_enough(Psn, Ps));
        const bool skip = (e == ts && t > ts && gsame);         // steps t-1 .. ts repeat this (U, P_s)
        const int plo = ts >= 1 ? ts : 1;                       // periods plo .. t-1 carry P_s,inf
        const double cu = skip ? (double)(t - ts + 1) : 1.0;
        const double cp = (t >= 1 ? 1.0 : 0.0) + (skip ? (double)(t - plo) : 0.0);
        SU = fma(cu, U, SU);
        SP = fma(cp, Psn, SP);
        if (t >= 1) emit(t - 1, Psn);
        if ((t == 0 || (skip && ts == 0)) && o.P0s) o.P0s[lane] = Psn;
        if (skip) {
            o.PsInf[lane] = Psn;
            fill_lo = plo - 1;
            fill_hi = t - 1;
            t = ts - 1;
        } else {
            t -= 1;
        }
        Ps = Psn;
        G.sync();                                               // L0 / L1 / LJ are free again
    }
    if (lane == 0) { o.fill[0] = fill_lo; o.fill[1] = fill_hi; }
    if (o.SP11) { o.SP11[lane] = SP; o.SU[lane] = SU; }

    // ---------------- steady Z, J, G and the powers G^(L 2^k), J^(L 2^k) for the chunk carries ------------------
    {
        const double Zs = Zlast, Js = Jlast, Gs = Glast;
        o.stead[lane] = Zs;
        o.stead[2 * RR + lane] = Gs;
        o.stead[1 * RR + lane] = Js;
        // the two power chains are independent: squared side by side (G through L0 / L1, J through LPT / LJ -- both free
        // now), half the dependent exchanges of one chain after the other
        double MG = Gs, MJ = Js;
        auto square2 = [&]() {
            L0[TS * i + j] = MG;
            L1[TS * j + i] = MG;
            LPT[TS * i + j] = MJ;
            LJ[TS * j + i] = MJ;
            G.sync();
            MG = dot_rows<R>(L0, L1, i, j);
            MJ = dot_rows<R>(LPT, LJ, i, j);
            G.sync();
        };
        for (int l = 1; l < a.L; l <<= 1) square2();             // M^L
#pragma unroll 1
        for (int k = 0; k < NLEV; ++k) {
            o.stead[(3 + k) * RR + lane] = MG;                   // G^(L 2^k)
            o.stead[(3 + NLEV + k) * RR + lane] = MJ;            // J^(L 2^k)
            if (k + 1 < NLEV) square2();
        }
    }
}

// ONE wave (64 lanes) per replicate at R = 8 (pass_fused.hip, cov_wave_kernel).  wsm: kCov8ScratchDoubles doubles of LDS.
template <int NLEV>
__device__ __forceinline__ void cov_wave8(const FastArgs& a, int b, double Cel, double ldfull, double* wsm, const Cov8Dst& o,
                                          int lane) {
    Grid<8> G;
    G.l = lane; G.i = lane >> 3; G.j = lane & 7; G.prow = nullptr; G.red = nullptr; G.tt = nullptr;
    cov_grid<8, NLEV, kCov8Keep>(a, b, Cel, ldfull, wsm, o, G);
}

// Gram matrix C = Lam' R^-1 Lam and sum log R of replicate b by ONE wave (lane l owns series {2l, 2l+1} + 128 q, q < NDR),
// C into the 64 doubles at Cs (LDS, row-major).  Returns sum log R on every lane.
template <int NDR>
__device__ __forceinline__ double gram_wave8(const double* __restrict__ Lg, const double* __restrict__ Rg, int N, int lane,
                                             double* Cs) {
    constexpr int R = 8;
    double W[NDR][2][R];
    bool own[NDR][2];
    double ld = 0.0;
#pragma unroll
    for (int q = 0; q < NDR; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 2 * lane + 128 * q + e;
            own[q][e] = c < N;
            const int cc = own[q][e] ? c : N - 1;
            const double rv = own[q][e] ? Rg[cc] : 1.0;
            const double ri = own[q][e] ? 1.0 / rv : 0.0;
            ld += log(rv);
#pragma unroll
            for (int k = 0; k < R; ++k) W[q][e][k] = Lg[(size_t)cc * R + k] * ri;
        }
    // the 36 packed entries in two halves: 18 partial sums live at a time instead of 36 (register budget of the
    // 12-wave workgroup of pass_fused_kernel)
    c_chunk<R, NDR, 0, 18, true>(W, Lg, own, lane, Cs);
    c_chunk<R, NDR, 18, 18, true>(W, Lg, own, lane, Cs);
    return wave_allsum(ld);
}

}  // namespace dfm
