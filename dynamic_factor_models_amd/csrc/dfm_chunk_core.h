// dfm_chunk_core.h -- the per-LANE algebra of recursion_chunk.hip: one lane owns one time chunk of one replicate and runs the
// information-form filter step and the Z-smoother step (oracle/info_form.py) on its OWN 8 x 8 matrices, held as 36 packed
// doubles (lower triangle, row-major) in its registers.  No cross-lane traffic: a product with one of the replicate's constant
// matrices (K = Q^-1 A and K') is an FMA whose scalar operand is a ROW of that matrix handed in by the caller's row source (on
// the GPU: 16 SGPRs filled by one scalar load; every lane of the wave works on the same replicate).
// The file is plain C++ so that the same text runs on the host lane by lane (tests/test_chunk_core_cpu.py drives it through
// tests/chunk_core_host.cpp against oracle/kalman_oracle.py).  Reference counterpart: none (dfm_functions.ipynb:21-23 declares
// `Parametric` only).
#pragma once

#if defined(__HIPCC__)
#define DFM_CK __host__ __device__ __forceinline__
#else
#include <cmath>
#define DFM_CK inline
#endif

namespace dfm {
namespace chunk {

constexpr int R = 8, NP = 36;

DFM_CK constexpr int pidx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

struct Row8 {
    double v[R];
};
// values a row fetch has to wait for (a scheduling fence on the GPU, ignored on the host): n <= 9
struct Deps {
    double v[R + 1];
    int n;
};

// m <- -(m^-1) for a symmetric positive definite m, by the symmetric sweep operator over all 8 pivots (no pivoting: the pivots of
// an SPD matrix are positive); returns det m.  rcp(d) = 1 / d for a positive normal d.
template <class Rcp>
DFM_CK double sweep8(double (&m)[NP], Rcp rcp) {
    double det = 1.0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const double d = m[pidx(k, k)];
        const double p = rcp(d);
        det *= d;
        double col[R], t[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            col[i] = m[pidx(i, k)];
            t[i] = col[i] * p;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (i == k) continue;
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                if (j == k) continue;
                m[pidx(i, j)] = fma(-t[i], col[j], m[pidx(i, j)]);
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i)
            if (i != k) m[pidx(i, k)] = t[i];
        m[pidx(k, k)] = -p;
    }
    return det;
}

// One forward step.  In: m = Om_f,t + Phi (packed), xi = xi_t, the period's collapsed observation through `obs`: obs.ready()
// is called once before the first read (on the GPU: the wait for the period's LDS-DMA), then obs.c(p) = C_t[packed p], obs.b(i) = b_t[i].
// Out: m = Om_f,t+1 + Phi, xi = xi_t+1; emit(zn, w) is called once with zn = -Z_t (packed) and w = w_t = Z_t xi_t (what the
// backward sweep reads back); det = det(Om_f,t + Phi), xw = xi_t' w_t.
//   Z = (Om_f + Phi)^-1;  J = Z K';  Om_f' + Phi = (Q^-1 + Phi) + C_t - K J;  w = Z xi;  xi' = K w + b_t
// krow(i, dep) = row i of K, qrow(i, dep) = row i of Q^-1 + Phi (entries 0 .. i are read); dep: see below.
template <class Obs, class KRows, class QRows, class Rcp, class Emit>
DFM_CK void fwd_step(double (&m)[NP], double (&xi)[R], Obs obs, double& det, double& xw,
                     KRows krow, QRows qrow, Rcp rcp, Emit emit) {
    det = sweep8(m, rcp);                                         // m = -Z
    double w[R];
    double dot = 0.0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < R; ++q) s = fma(m[pidx(i, q)], xi[q], s);
        w[i] = -s;
        dot = fma(xi[i], w[i], dot);
    }
    xw = dot;
    emit(m, w);
    // Row fetches are ORDERED along the arithmetic (second argument of the row source): the fetch of row i + 1 waits for the
    // first result computed with row i (so the wait for row i has already happened and does not also wait for row i + 1) and for
    // ALL results of row i - 1 (so it cannot run ahead: two rows in flight at most -- the 24 rows of a step fetched at once
    // would spill the scalar register file into VGPR lanes, ~800 lane reads per step).
    double jn[R][R];                                              // -J = (-Z) K'
    Row8 kr = krow(0, Deps{{w[0]}, 1});
#pragma unroll
    for (int cc = 0; cc < R; ++cc) {
        Row8 nx = kr;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < R; ++q) s = fma(m[pidx(k, q)], kr.v[q], s);
            jn[k][cc] = s;
            if (k == 0) {                                          // (after the last column: row 0 again, for the loop below)
                Deps d{{s}, 1};
                if (cc > 0) {
#pragma unroll
                    for (int e = 0; e < R; ++e) d.v[1 + e] = jn[e][cc - 1];
                    d.n = R + 1;
                }
                nx = krow(cc + 1 < R ? cc + 1 : 0, d);
            }
        }
        kr = nx;
    }
    Deps dq{{0.0}, R};
#pragma unroll
    for (int e = 0; e < R; ++e) dq.v[e] = jn[e][R - 1];
    Row8 qr = qrow(0, dq);
    obs.ready();
#pragma unroll
    for (int i = 0; i < R; ++i) {
        Row8 nk = kr, nq = qr;
        double s = obs.b(i);
#pragma unroll
        for (int k = 0; k < R; ++k) s = fma(kr.v[k], w[k], s);
        xi[i] = s;
        if (i + 1 < R) {
            Deps d{{s}, 1};
            if (i > 0) {
#pragma unroll
                for (int e = 0; e < i; ++e) d.v[1 + e] = m[pidx(i - 1, e)];
                d.n = 1 + i;
            }
            nk = krow(i + 1, d);
            nq = qrow(i + 1, Deps{{s}, 1});
        }
#pragma unroll
        for (int cc = 0; cc <= i; ++cc) {
            double acc = qr.v[cc] + obs.c(pidx(i, cc));
#pragma unroll
            for (int k = 0; k < R; ++k) acc = fma(kr.v[k], jn[k][cc], acc);
            m[pidx(i, cc)] = acc;
        }
        kr = nk; qr = nq;
    }
}

// One backward step: state (P, f) = (P_t+1|T, f_t+1|T) -> (P_t|T, f_t|T) with the forward sweep's -Z_t and w_t through `zw`:
// zw.ready() once before the first read (GPU: the wait for the LDS-DMA of the period's table row), zw.z(p) = -Z_t[packed p],
// zw.w(i) = w_t[i].
//   P_t = Z + J P J' = Z + Z (K' P K) Z;   f_t = w + J f = w + Z (K' f);   U_t = Cov(f_t+1, f_t | X) = P J' = (P K) Z
// ktrow(i, dep) = row i of K'.  acc.s10(k, n, v): v = U_t[k][n] + f_t+1[k] f_t[n];  acc.s11(p, v): v = (P_t + f_t f_t')[packed p]
// (the EM's sufficient statistics; formed only in lanes whose accumulator says want10() / want11(): the step counts -- a lane's
// warm-up steps may run on garbage, which must not reach a sum even multiplied by zero).  Acc::on == false: never.
template <class ZW, class KTRows, class Acc>
DFM_CK void bwd_step(double (&P)[NP], double (&f)[R], ZW zw, KTRows ktrow, Acc acc) {
    double T1[R][R], y[R];                                        // T1 = P K,  y = K' f
    Row8 kt = ktrow(0, Deps{{f[0]}, 1});
#pragma unroll
    for (int j = 0; j < R; ++j) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < R; ++q) s = fma(kt.v[q], f[q], s);
        y[j] = s;
        Deps d{{s}, 1};
        if (j > 0) {
#pragma unroll
            for (int e = 0; e < R; ++e) d.v[1 + e] = T1[e][j - 1];
            d.n = R + 1;
        }
        const Row8 nx = ktrow(j + 1 < R ? j + 1 : R - 1, d);       // (after the last column: row 7, where the G loop starts)
#pragma unroll
        for (int k = 0; k < R; ++k) {
            double u = 0.0;
#pragma unroll
            for (int q = 0; q < R; ++q) u = fma(P[pidx(k, q)], kt.v[q], u);
            T1[k][j] = u;
        }
        kt = nx;
    }
    double fn[R];
    if constexpr (Acc::on) {                                      // EM: f_t and U_t while T1 lives; -Z_t straight from the table (not held)
        zw.ready();
#pragma unroll
        for (int i = 0; i < R; ++i) {
            double s = zw.w(i);
#pragma unroll
            for (int q = 0; q < R; ++q) s = fma(-zw.z(pidx(i, q)), y[q], s);
            fn[i] = s;
        }
        if (acc.want10()) {
#pragma unroll
            for (int n = 0; n < R; ++n) {
                double u[R];
#pragma unroll
                for (int k = 0; k < R; ++k) u[k] = f[k] * fn[n];
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const double z = zw.z(pidx(j, n));
#pragma unroll
                    for (int k = 0; k < R; ++k) u[k] = fma(-T1[k][j], z, u[k]);
                }
#pragma unroll
                for (int k = 0; k < R; ++k) acc.s10(k, n, u[k]);
            }
        }
    }
    double G[NP];                                                 // G = K' P K (P is dead from here: G may take its registers)
#pragma unroll
    for (int i = R - 1; i >= 0; --i) {                            // long rows first: their FMAs cover the next row's fetch
        Row8 nx = kt;
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) s = fma(kt.v[k], T1[k][j], s);
            G[pidx(i, j)] = s;
            if (j == 0 && i > 0) {
                Deps d{{s}, 1};
                if (i + 1 < R) {
#pragma unroll
                    for (int e = 0; e < R; ++e) d.v[1 + e] = G[pidx(i + 1, e)];
                    d.n = R + 1;
                }
                nx = ktrow(i - 1, d);
            }
        }
        kt = nx;
    }
    double zn[NP];                                                // T1 is dead: -Z_t moves into registers
    if constexpr (!Acc::on) zw.ready();
#pragma unroll
    for (int k = 0; k < NP; ++k) zn[k] = zw.z(k);
    if constexpr (!Acc::on) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            double s = zw.w(i);
#pragma unroll
            for (int q = 0; q < R; ++q) s = fma(-zn[pidx(i, q)], y[q], s);
            fn[i] = s;
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        double t2[R];                                             // column j of G (-Z)
#pragma unroll
        for (int k = 0; k < R; ++k) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < R; ++q) s = fma(G[pidx(k, q)], zn[pidx(q, j)], s);
            t2[k] = s;
        }
#pragma unroll
        for (int i = j; i < R; ++i) {
            double s = -zn[pidx(i, j)];
#pragma unroll
            for (int k = 0; k < R; ++k) s = fma(zn[pidx(i, k)], t2[k], s);
            P[pidx(i, j)] = s;
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) f[i] = fn[i];
    if constexpr (Acc::on) {
        if (acc.want11()) {
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) acc.s11(pidx(i, j), fma(fn[i], fn[j], P[pidx(i, j)]));
        }
    }
}

struct NoAcc {
    static constexpr bool on = false;
    DFM_CK bool want10() const { return false; }
    DFM_CK bool want11() const { return false; }
    DFM_CK void s10(int, int, double) const {}
    DFM_CK void s11(int, double) const {}
};

// The chunk boundary check: the state a lane holds after its warm-up against the state its neighbour holds at the end of its own
// chunk, ELEMENT BY ELEMENT -- the largest |a - b| over the 36 packed matrix entries against tol x the largest |entry| of either
// matrix, and the same for the 8-vector (matrix and vector separately: their scales differ).  A bound, not a projection: every
// entry of the state a chunk starts from is within tol x scale of the sequential recursion's.  (Until round 5 the lanes compared
// two weighted sums of the entries: an error orthogonal to the weight vectors passed.)  NaN-safe: anything not provably close
// fails -- `nn` carries a NaN from any entry (fmax would drop it).
struct Gap {
    double dm, sm, dx, sx, nn;
};
DFM_CK void gap_mat(Gap& g, double a, double b) {
    const double d = fabs(a - b);
    g.dm = d > g.dm ? d : g.dm;
    const double s = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
    g.sm = s > g.sm ? s : g.sm;
    g.nn += d;
}
DFM_CK void gap_vec(Gap& g, double a, double b) {
    const double d = fabs(a - b);
    g.dx = d > g.dx ? d : g.dx;
    const double s = fabs(a) > fabs(b) ? fabs(a) : fabs(b);
    g.sx = s > g.sx ? s : g.sx;
    g.nn += d;
}
DFM_CK bool gap_close(const Gap& g, double tol) { return g.dm <= tol * g.sm && g.dx <= tol * g.sx && g.nn == g.nn && g.nn <= 1.7e308; }
DFM_CK Gap state_gap(const double (&m)[NP], const double (&x)[R], const double (&m2)[NP], const double (&x2)[R]) {
    Gap g{0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NP; ++k) gap_mat(g, m[k], m2[k]);
#pragma unroll
    for (int i = 0; i < R; ++i) gap_vec(g, x[i], x2[i]);
    return g;
}

}  // namespace chunk
}  // namespace dfm
