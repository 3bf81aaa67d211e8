/* oracle/dfm_oracle.c -- plain C (fp64) restatement of the Kalman filter / RTS smoother / EM
 * M-step for the dynamic factor model  x_t = Lam f_t + eps_t,  f_t = A f_{t-1} + eta_t.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path and the "cpu_baseline" leg of bench.py
 * (kind = "port").  The product library (libdfmhip.so) never links or calls this file.
 *
 * PARITY UNPINNED BY THE REFERENCE: QuantEcon/dynamic_factor_models declares `Parametric`
 * (dfm_functions.ipynb:21-23) and the state-space matrices M, Q, G (dfm_functions.ipynb:30-34,
 * 477-492) but has no filter/smoother/EM code.  The equations restated here are the published
 * ones (Shumway & Stoffer 1982; Banbura & Modugno 2014 for missing cells; collapsed observation
 * vector as in Jungbacker & Koopman 2015) -- SURVEY.md Appendix B.  The file is pinned against
 * oracle/kalman_oracle.py (itself pinned by brute-force Gaussian conditioning) in
 * tests/test_oracle_kalman.py.
 *
 * Layout: panel x[t*N + i] (NaN = missing); Lam[i*r + k]; A, Q, P0 row-major r x r;
 * packed symmetric output: lower triangle row-major, idx(i,j) = i(i+1)/2 + j (j <= i).
 *
 * Build:  make -C oracle      (gcc -O2 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RMAX 64
static const double LOG2PI = 1.8378770664093454835606594728112;

/* ---- tiny dense helpers (row-major, n <= RMAX) ------------------------------------------- */
static void mm(int n, const double* X, const double* Y, double* Z) { /* Z = X Y */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double a = 0.0;
            for (int k = 0; k < n; ++k) a += X[i * n + k] * Y[k * n + j];
            Z[i * n + j] = a;
        }
}
static void mmt(int n, const double* X, const double* Y, double* Z) { /* Z = X Y' */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double a = 0.0;
            for (int k = 0; k < n; ++k) a += X[i * n + k] * Y[j * n + k];
            Z[i * n + j] = a;
        }
}
static void mv(int n, const double* X, const double* v, double* y) {
    for (int i = 0; i < n; ++i) {
        double a = 0.0;
        for (int k = 0; k < n; ++k) a += X[i * n + k] * v[k];
        y[i] = a;
    }
}
static void symmetrize(int n, double* X) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) {
            double a = 0.5 * (X[i * n + j] + X[j * n + i]);
            X[i * n + j] = a; X[j * n + i] = a;
        }
}
/* LU with partial pivoting of the n x n matrix M (destroyed); solves M X = B for nb right-hand
 * sides stored as columns of the row-major n x nb matrix B (overwritten by X).  Returns
 * log|det M| through *logdet (may be NULL).  0 on success, -1 if singular. */
static int lu_solve(int n, double* M, double* B, int nb, double* logdet) {
    double ld = 0.0;
    for (int k = 0; k < n; ++k) {
        int p = k; double big = fabs(M[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(M[i * n + k]) > big) { big = fabs(M[i * n + k]); p = i; }
        if (big == 0.0 || big != big) return -1;
        if (p != k) {
            for (int j = 0; j < n; ++j) { double t = M[k * n + j]; M[k * n + j] = M[p * n + j]; M[p * n + j] = t; }
            for (int j = 0; j < nb; ++j) { double t = B[k * nb + j]; B[k * nb + j] = B[p * nb + j]; B[p * nb + j] = t; }
        }
        ld += log(fabs(M[k * n + k]));
        double inv = 1.0 / M[k * n + k];
        for (int i = k + 1; i < n; ++i) {
            double f = M[i * n + k] * inv;
            if (f == 0.0) continue;
            for (int j = k + 1; j < n; ++j) M[i * n + j] -= f * M[k * n + j];
            for (int j = 0; j < nb; ++j) B[i * nb + j] -= f * B[k * nb + j];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        double inv = 1.0 / M[k * n + k];
        for (int j = 0; j < nb; ++j) {
            double a = B[k * nb + j];
            for (int i = k + 1; i < n; ++i) a -= M[k * n + i] * B[i * nb + j];
            B[k * nb + j] = a * inv;
        }
    }
    if (logdet) *logdet = ld;
    return 0;
}

/* ---- one replicate: filter + RTS smoother ------------------------------------------------- */
/* Work arrays for the smoother are allocated per call: (T+1) * (2 r + 2 r^2) doubles.
 * Outputs (any may be NULL except loglik): f_smooth[T*r], P_smooth_packed[T*r(r+1)/2],
 * f0_s[r], P0_s[r*r], P_lag[T*r*r] = Cov(f_t, f_{t-1} | X) for t = 1..T.
 * Returns 0, or -1 on bad dims / allocation failure, -2 on a singular system. */
int dfm_oracle_ks_pass(int T, int N, int r, const double* x, const double* Lam, const double* R,
                       const double* A, const double* Q, const double* mu0, const double* P0,
                       double* f_smooth, double* P_smooth_packed, double* loglik, double* f0_s,
                       double* P0_s, double* P_lag) {
    if (T < 1 || N < 1 || r < 1 || r > RMAX) return -1;
    const int rr = r * r, np = r * (r + 1) / 2;
    double* fp_all = (double*)malloc(sizeof(double) * (size_t)T * r);
    double* ff_all = (double*)malloc(sizeof(double) * (size_t)T * r);
    double* Pp_all = (double*)malloc(sizeof(double) * (size_t)T * rr);
    double* Pf_all = (double*)malloc(sizeof(double) * (size_t)T * rr);
    double* Rinv = (double*)malloc(sizeof(double) * N);
    double* logR = (double*)malloc(sizeof(double) * N);
    if (!fp_all || !ff_all || !Pp_all || !Pf_all || !Rinv || !logR) {
        free(fp_all); free(ff_all); free(Pp_all); free(Pf_all); free(Rinv); free(logR);
        return -1;
    }
    for (int i = 0; i < N; ++i) { Rinv[i] = 1.0 / R[i]; logR[i] = log(R[i]); }
    double C[RMAX * RMAX], D[RMAX * RMAX], W[RMAX * RMAX], W2[RMAX * RMAX];
    double b[RMAX], u[RMAX], ff[RMAX], fp[RMAX], tmp[RMAX];
    double Pf[RMAX * RMAX], Pp[RMAX * RMAX];
    memcpy(ff, mu0, sizeof(double) * r);
    memcpy(Pf, P0, sizeof(double) * rr);
    double ll = 0.0;
    int rc = 0;
    for (int t = 0; t < T && rc == 0; ++t) {
        /* predict */
        mv(r, A, ff, fp);
        mm(r, A, Pf, W); mmt(r, W, A, Pp);
        for (int k = 0; k < rr; ++k) Pp[k] += Q[k];
        symmetrize(r, Pp);
        /* collapse row t */
        const double* xt = x + (size_t)t * N;
        double s = 0.0, ld = 0.0; int n = 0;
        memset(C, 0, sizeof(double) * rr); memset(b, 0, sizeof(double) * r);
        for (int i = 0; i < N; ++i) {
            double xi = xt[i];
            if (xi != xi) continue;
            const double* li = Lam + (size_t)i * r;
            double w = Rinv[i];
            ++n; ld += logR[i]; s += xi * xi * w;
            for (int k = 0; k < r; ++k) {
                b[k] += li[k] * xi * w;
                double lw = li[k] * w;
                for (int j = 0; j <= k; ++j) C[k * r + j] += lw * li[j];
            }
        }
        for (int k = 0; k < r; ++k) for (int j = 0; j < k; ++j) C[j * r + k] = C[k * r + j];
        /* update: D = I + C Pp ; Pf = Pp D^{-1}  <=>  D' Pf' = Pp' */
        mm(r, C, Pp, D);
        for (int k = 0; k < r; ++k) D[k * r + k] += 1.0;
        for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) { W[i * r + j] = D[j * r + i]; W2[i * r + j] = Pp[j * r + i]; }
        double logdetD;
        if (lu_solve(r, W, W2, r, &logdetD)) { rc = -2; break; }
        for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) Pf[i * r + j] = W2[j * r + i];
        symmetrize(r, Pf);
        mv(r, C, fp, tmp);
        for (int k = 0; k < r; ++k) u[k] = b[k] - tmp[k];
        mv(r, Pf, u, tmp);
        double quad = s, uPu = 0.0, fpb = 0.0, fCf = 0.0;
        for (int k = 0; k < r; ++k) { ff[k] = fp[k] + tmp[k]; uPu += u[k] * tmp[k]; fpb += fp[k] * b[k]; }
        mv(r, C, fp, tmp);
        for (int k = 0; k < r; ++k) fCf += fp[k] * tmp[k];
        quad += -2.0 * fpb + fCf - uPu;
        ll += -0.5 * (n * LOG2PI + ld + logdetD + quad);
        memcpy(fp_all + (size_t)t * r, fp, sizeof(double) * r);
        memcpy(ff_all + (size_t)t * r, ff, sizeof(double) * r);
        memcpy(Pp_all + (size_t)t * rr, Pp, sizeof(double) * rr);
        memcpy(Pf_all + (size_t)t * rr, Pf, sizeof(double) * rr);
    }
    *loglik = ll;
    /* RTS backward sweep */
    if (rc == 0) {
        double fs[RMAX], Ps[RMAX * RMAX], J[RMAX * RMAX], dP[RMAX * RMAX], d[RMAX];
        memcpy(fs, ff_all + (size_t)(T - 1) * r, sizeof(double) * r);
        memcpy(Ps, Pf_all + (size_t)(T - 1) * rr, sizeof(double) * rr);
        for (int t = T - 1; t >= 0; --t) {
            if (f_smooth) memcpy(f_smooth + (size_t)t * r, fs, sizeof(double) * r);
            if (P_smooth_packed)
                for (int i = 0, k = 0; i < r; ++i) for (int j = 0; j <= i; ++j, ++k)
                    P_smooth_packed[(size_t)t * np + k] = Ps[i * r + j];
            /* step to index t-1 (t-1 == -1 is the initial state f_0 with moments mu0, P0) */
            const double* Pf_prev = (t > 0) ? Pf_all + (size_t)(t - 1) * rr : P0;
            const double* ff_prev = (t > 0) ? ff_all + (size_t)(t - 1) * r : mu0;
            const double* Pp_t = Pp_all + (size_t)t * rr;
            const double* fp_t = fp_all + (size_t)t * r;
            /* J = Pf_prev A' Pp_t^{-1}:  Pp_t J' = A Pf_prev */
            memcpy(W, Pp_t, sizeof(double) * rr);
            mm(r, A, Pf_prev, W2);
            if (lu_solve(r, W, W2, r, NULL)) { rc = -2; break; }
            for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) J[i * r + j] = W2[j * r + i];
            if (P_lag) mmt(r, Ps, J, P_lag + (size_t)t * rr);          /* Cov(f_t, f_{t-1}|X) = Ps_t J' */
            for (int k = 0; k < r; ++k) d[k] = fs[k] - fp_t[k];
            mv(r, J, d, tmp);
            for (int k = 0; k < rr; ++k) dP[k] = Ps[k] - Pp_t[k];
            mm(r, J, dP, W); mmt(r, W, J, W2);
            for (int k = 0; k < r; ++k) fs[k] = ff_prev[k] + tmp[k];
            for (int k = 0; k < rr; ++k) Ps[k] = Pf_prev[k] + W2[k];
            symmetrize(r, Ps);
        }
        if (rc == 0) {
            if (f0_s) memcpy(f0_s, fs, sizeof(double) * r);
            if (P0_s) memcpy(P0_s, Ps, sizeof(double) * rr);
        }
    }
    free(fp_all); free(ff_all); free(Pp_all); free(Pf_all); free(Rinv); free(logR);
    return rc;
}

/* ---- batch driver (OpenMP over replicates) ------------------------------------------------- */
/* panel[B][T][N], Lam[B][N][r], R[B][N], A/Q/P0[B][r][r], mu0[B][r]; outputs f_smooth[B][T][r],
 * P_smooth[B][T][r(r+1)/2] (may be NULL), loglik[B].  nthreads <= 0: OpenMP default. */
int dfm_oracle_ks_pass_batch(int B, int T, int N, int r, const double* panel, const double* Lam,
                             const double* R, const double* A, const double* Q, const double* mu0,
                             const double* P0, double* f_smooth, double* P_smooth, double* loglik,
                             int nthreads) {
    int rc_all = 0;
    const size_t np = (size_t)r * (r + 1) / 2;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        int rc = dfm_oracle_ks_pass(T, N, r, panel + (size_t)b * T * N, Lam + (size_t)b * N * r,
                                    R + (size_t)b * N, A + (size_t)b * r * r, Q + (size_t)b * r * r,
                                    mu0 + (size_t)b * r, P0 + (size_t)b * r * r,
                                    f_smooth ? f_smooth + (size_t)b * T * r : NULL,
                                    P_smooth ? P_smooth + (size_t)b * T * np : NULL, loglik + b,
                                    NULL, NULL, NULL);
        if (rc) {
#pragma omp critical
            rc_all = rc;
        }
    }
    return rc_all;
}

int dfm_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- one EM iteration for one replicate ------------------------------- */
/* Parameters are updated IN PLACE; *loglik receives the log-likelihood at the parameters that
 * were passed in.  Handles missing cells (per-series normal equations).  0 / -1 / -2 as above. */
int dfm_oracle_em_step(int T, int N, int r, const double* x, double* Lam, double* R, double* A,
                       double* Q, double* mu0, double* P0, double* loglik) {
    const int rr = r * r, np = r * (r + 1) / 2;
    double* fs = (double*)malloc(sizeof(double) * (size_t)T * r);
    double* Psp = (double*)malloc(sizeof(double) * (size_t)T * np);
    double* Pl = (double*)malloc(sizeof(double) * (size_t)T * rr);
    double* Ef = (double*)malloc(sizeof(double) * (size_t)T * rr);
    if (!fs || !Psp || !Pl || !Ef) { free(fs); free(Psp); free(Pl); free(Ef); return -1; }
    double f0[RMAX], P0s[RMAX * RMAX];
    int rc = dfm_oracle_ks_pass(T, N, r, x, Lam, R, A, Q, mu0, P0, fs, Psp, loglik, f0, P0s, Pl);
    if (rc) { free(fs); free(Psp); free(Pl); free(Ef); return rc; }
    double S11[RMAX * RMAX], S00[RMAX * RMAX], S10[RMAX * RMAX], W[RMAX * RMAX], W2[RMAX * RMAX];
    memset(S11, 0, sizeof(S11)); memset(S10, 0, sizeof(S10));
    for (int t = 0; t < T; ++t) {
        const double* f = fs + (size_t)t * r;
        const double* fprev = t > 0 ? fs + (size_t)(t - 1) * r : f0;
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < r; ++j) {
                int hi = i > j ? i : j, lo = i > j ? j : i;
                double e = f[i] * f[j] + Psp[(size_t)t * np + hi * (hi + 1) / 2 + lo];
                Ef[(size_t)t * rr + i * r + j] = e;
                S11[i * r + j] += e;
                S10[i * r + j] += f[i] * fprev[j] + Pl[(size_t)t * rr + i * r + j];
            }
    }
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < r; ++j)
            S00[i * r + j] = S11[i * r + j] - Ef[(size_t)(T - 1) * rr + i * r + j] + f0[i] * f0[j] + P0s[i * r + j];
    /* A = S10 S00^{-1}  <=>  S00' A' = S10' */
    for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) { W[i * r + j] = S00[j * r + i]; W2[i * r + j] = S10[j * r + i]; }
    if (lu_solve(r, W, W2, r, NULL)) rc = -2;
    if (rc == 0) {
        for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) A[i * r + j] = W2[j * r + i];
        mmt(r, A, S10, W);                                     /* A S10' */
        for (int k = 0; k < rr; ++k) Q[k] = (S11[k] - W[k]) / T;
        symmetrize(r, Q);
        for (int i = 0; i < N && rc == 0; ++i) {
            double Sff[RMAX * RMAX], sxf[RMAX], rhs[RMAX], sxx = 0.0; int Ti = 0;
            memset(Sff, 0, sizeof(double) * rr); memset(sxf, 0, sizeof(double) * r);
            for (int t = 0; t < T; ++t) {
                double xi = x[(size_t)t * N + i];
                if (xi != xi) continue;
                ++Ti; sxx += xi * xi;
                for (int k = 0; k < r; ++k) sxf[k] += xi * fs[(size_t)t * r + k];
                for (int k = 0; k < rr; ++k) Sff[k] += Ef[(size_t)t * rr + k];
            }
            memcpy(W, Sff, sizeof(double) * rr); memcpy(rhs, sxf, sizeof(double) * r);
            if (lu_solve(r, W, rhs, 1, NULL)) { rc = -2; break; }
            double q1 = 0.0, q2 = 0.0;
            mv(r, Sff, rhs, W2);
            for (int k = 0; k < r; ++k) { Lam[(size_t)i * r + k] = rhs[k]; q1 += rhs[k] * sxf[k]; q2 += rhs[k] * W2[k]; }
            R[i] = (sxx - 2.0 * q1 + q2) / Ti;
        }
        memcpy(mu0, f0, sizeof(double) * r);
        memcpy(P0, P0s, sizeof(double) * rr);
    }
    free(fs); free(Psp); free(Pl); free(Ef);
    return rc;
}
