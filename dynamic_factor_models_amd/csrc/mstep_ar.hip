// mstep_ar.hip -- the series block of the ECM iteration for the parametric DFM with AR(q) idiosyncratic terms
// (oracle/ar_oracle.py em_step_ar, steps (2)-(4)):
//
//     x_it = lam_i' f_t + e_it,   e_it = rho_i1 e_i,t-1 + .. + rho_iq e_i,t-q + eps_it,   eps_it ~ N(0, sig2_i)
//
// rho / sig2 play the part of the reference's `uar_coef` / `uar_ser`^2 (dfm_functions.ipynb:305-311, 405-412), which the
// reference estimates once from the loading-regression residuals; here they are re-estimated jointly with the loadings
// from the smoothed moments of the companion state z_t = (f_t, .., f_{t-m+1}), m = max(p, q + 1), of the
// quasi-differenced model (capi.hip: ar_em_run).  With a_i = (1, -rho_i1, .., -rho_iq):
//   (2) loadings given rho:   lam_i = [sum_t E g_it g_it']^-1 sum_t x~_it E g_it,   g_it = sum_l a_il f_{t-l},
//                             x~_it = sum_l a_il x_i,t-l, t over the periods where x_it and its q lags are observed;
//   (3) rho given the NEW loadings: u_itl = x_i,t-l - lam_i' f_{t-l},  U_i = sum_t E[u_it u_it'],
//                             rho_i = U_i[1:,1:]^-1 U_i[1:,0];
//   (4) sig2_i = a_i' U_i a_i / n_i.
// One thread per series, two sweeps over the periods (the second needs the first one's loadings).  The smoothed moments of
// a period are the same for every series: their addresses are wave-uniform, so they travel through the scalar unit.
// The reference has no counterpart of the joint estimation (dfm_functions.ipynb:21-23 declares `Parametric` only).
#include "dfm_kernels.h"

namespace dfm {

namespace {

// In-place Cholesky solve of the leading n x n system M x = y (M SPD, lower triangle used); static indexing only.
template <int NMAX>
__device__ __forceinline__ bool chol_solve_reg(double (&M)[NMAX][NMAX], double (&y)[NMAX], int n) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        if (j < n) {
            double d = M[j][j];
#pragma unroll
            for (int k = 0; k < NMAX; ++k)
                if (k < j) d -= M[j][k] * M[j][k];
            ok = ok && (d > 0.0);
            d = sqrt(d > 0.0 ? d : 1.0);
            M[j][j] = d;
#pragma unroll
            for (int i = 0; i < NMAX; ++i) {
                if (i > j && i < n) {
                    double s = M[i][j];
#pragma unroll
                    for (int k = 0; k < NMAX; ++k)
                        if (k < j) s -= M[i][k] * M[j][k];
                    M[i][j] = s / d;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        if (i < n) {
            double s = y[i];
#pragma unroll
            for (int k = 0; k < NMAX; ++k)
                if (k < i) s -= M[i][k] * y[k];
            y[i] = s / M[i][i];
        }
    }
#pragma unroll
    for (int ii = NMAX - 1; ii >= 0; --ii) {
        if (ii < n) {
            double s = y[ii];
#pragma unroll
            for (int k = 0; k < NMAX; ++k)
                if (k > ii && k < n) s -= M[k][ii] * y[k];
            y[ii] = s / M[ii][ii];
        }
    }
    return ok;
}

}  // namespace

// R = the model's number of factors (exact), Q1 = q + 1.
template <int R, int Q1>
__global__ __launch_bounds__(256) void mstep_ar_kernel(ArMstepArgs a) {
    constexpr int Q = Q1 - 1;
    const int b = blockIdx.y;
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (a.active && a.active[b] == 0) return;                  // converged replicate: parameters stay
    const int T = a.T, N = a.N, Rk = a.Rk;
    const int Tq = T - Q;
    const bool live = i < N;
    const int ii = live ? i : N - 1;
    const size_t npk = (size_t)Rk * (Rk + 1) / 2;
    const double* __restrict__ xb = a.panel + (size_t)b * T * N;
    const double* __restrict__ zb = a.zsm + (size_t)b * Tq * Rk;
    const double* __restrict__ Pb = a.Psm + (size_t)b * Tq * npk;
    auto pk = [](int u, int v) { return u >= v ? u * (u + 1) / 2 + v : v * (v + 1) / 2 + u; };   // packed lower, symmetric

    double lam[R], av[Q1];
#pragma unroll
    for (int c = 0; c < R; ++c) lam[c] = a.Lam[((size_t)b * N + ii) * R + c];
    av[0] = 1.0;
#pragma unroll
    for (int l = 1; l < Q1; ++l) av[l] = -a.rho[((size_t)b * N + ii) * Q + (l - 1)];

    // ---- sweep 1: loadings given rho ----------------------------------------------------------------------------
    double LH[R][R], RH[R];
#pragma unroll
    for (int c = 0; c < R; ++c) {
        RH[c] = 0.0;
#pragma unroll
        for (int d = 0; d < R; ++d) LH[c][d] = 0.0;
    }
    int n = 0;
    double X[Q1];                                              // X[l] = x_{i, t + q - l}
#pragma unroll
    for (int l = 0; l < Q1; ++l) X[l] = 0.0;
#pragma unroll
    for (int l = 1; l < Q1; ++l) X[l - 1] = xb[(size_t)(Q - l) * N + ii];   // rows q-1 .. 0 -> X[0] .. X[q-1] (shifted below)
    for (int t = 0; t < Tq; ++t) {
#pragma unroll
        for (int l = Q1 - 1; l >= 1; --l) X[l] = X[l - 1];
        X[0] = xb[(size_t)(t + Q) * N + ii];
        bool ok = true;
        double xt = 0.0;
#pragma unroll
        for (int l = 0; l < Q1; ++l) { ok = ok && (X[l] == X[l]); xt = fma(av[l], X[l], xt); }
        const double* __restrict__ zt = zb + (size_t)t * Rk;   // (wave-uniform addresses from here on)
        const double* __restrict__ Pt = Pb + (size_t)t * npk;
        double g[R];
#pragma unroll
        for (int c = 0; c < R; ++c) {
            double s = 0.0;
#pragma unroll
            for (int l = 0; l < Q1; ++l) s = fma(av[l], zt[l * R + c], s);
            g[c] = s;
        }
        if (ok) {
            ++n;
#pragma unroll
            for (int c = 0; c < R; ++c) {
                RH[c] = fma(xt, g[c], RH[c]);
#pragma unroll
                for (int d = 0; d <= c; ++d) {
                    double s = g[c] * g[d];
#pragma unroll
                    for (int l = 0; l < Q1; ++l)
#pragma unroll
                        for (int l2 = 0; l2 < Q1; ++l2) s = fma(av[l] * av[l2], Pt[pk(l * R + c, l2 * R + d)], s);
                    LH[c][d] += s;
                }
            }
        }
    }
    const bool enough = n >= R + Q + 1;                       // as the oracle: fewer quasi-differenced cells -> series left as is
    {
#pragma unroll
        for (int c = 0; c < R; ++c)
#pragma unroll
            for (int d = c + 1; d < R; ++d) LH[c][d] = LH[d][c];
        const bool pd = chol_solve_reg<R>(LH, RH, R);
        if (enough && pd) {
#pragma unroll
            for (int c = 0; c < R; ++c) lam[c] = RH[c];
        }
    }

    // ---- sweep 2: rho and sig2 given the new loadings ---------------------------------------------------------------
    double U[Q1][Q1];
#pragma unroll
    for (int l = 0; l < Q1; ++l)
#pragma unroll
        for (int l2 = 0; l2 < Q1; ++l2) U[l][l2] = 0.0;
    double lp[R][R];                                          // lam_c lam_d
#pragma unroll
    for (int c = 0; c < R; ++c)
#pragma unroll
        for (int d = 0; d < R; ++d) lp[c][d] = lam[c] * lam[d];
#pragma unroll
    for (int l = 0; l < Q1; ++l) X[l] = 0.0;
#pragma unroll
    for (int l = 1; l < Q1; ++l) X[l - 1] = xb[(size_t)(Q - l) * N + ii];
    for (int t = 0; t < Tq; ++t) {
#pragma unroll
        for (int l = Q1 - 1; l >= 1; --l) X[l] = X[l - 1];
        X[0] = xb[(size_t)(t + Q) * N + ii];
        bool ok = true;
#pragma unroll
        for (int l = 0; l < Q1; ++l) ok = ok && (X[l] == X[l]);
        const double* __restrict__ zt = zb + (size_t)t * Rk;
        const double* __restrict__ Pt = Pb + (size_t)t * npk;
        if (ok) {
            double u[Q1];
#pragma unroll
            for (int l = 0; l < Q1; ++l) {
                double s = X[l];
#pragma unroll
                for (int c = 0; c < R; ++c) s = fma(-lam[c], zt[l * R + c], s);
                u[l] = s;
            }
#pragma unroll
            for (int l = 0; l < Q1; ++l)
#pragma unroll
                for (int l2 = 0; l2 <= l; ++l2) {
                    double s = u[l] * u[l2];
#pragma unroll
                    for (int c = 0; c < R; ++c)
#pragma unroll
                        for (int d = 0; d < R; ++d) s = fma(lp[c][d], Pt[pk(l * R + c, l2 * R + d)], s);
                    U[l][l2] += s;
                }
        }
    }
    if (!(live && enough)) return;
    double rho_new[Q > 0 ? Q : 1];
    double sig = U[0][0];
    if constexpr (Q > 0) {
        double M[Q][Q], y[Q];
#pragma unroll
        for (int l = 0; l < Q; ++l) {
            y[l] = U[l + 1][0];
#pragma unroll
            for (int l2 = 0; l2 < Q; ++l2) M[l][l2] = l2 <= l ? U[l + 1][l2 + 1] : U[l2 + 1][l + 1];
        }
        const bool pd = chol_solve_reg<Q>(M, y, Q);
#pragma unroll
        for (int l = 0; l < Q; ++l) rho_new[l] = pd ? y[l] : -av[l + 1];
        // a' U a with a = (1, -rho)
        double s = U[0][0];
#pragma unroll
        for (int l = 0; l < Q; ++l) s = fma(-2.0 * rho_new[l], U[l + 1][0], s);
#pragma unroll
        for (int l = 0; l < Q; ++l)
#pragma unroll
            for (int l2 = 0; l2 < Q; ++l2) s = fma(rho_new[l] * rho_new[l2], l2 <= l ? U[l + 1][l2 + 1] : U[l2 + 1][l + 1], s);
        sig = s;
#pragma unroll
        for (int l = 0; l < Q; ++l) a.rho[((size_t)b * N + i) * Q + l] = rho_new[l];
    }
#pragma unroll
    for (int c = 0; c < R; ++c) a.Lam[((size_t)b * N + i) * R + c] = lam[c];
    a.sig2[(size_t)b * N + i] = sig / (double)n;
}

template <int R, int Q1>
static hipError_t launch_ar_rq(const ArMstepArgs& a, hipStream_t s) {
    if constexpr (R * Q1 > 32) {
        return hipErrorInvalidValue;
    } else {
        hipLaunchKernelGGL((mstep_ar_kernel<R, Q1>), dim3((a.N + 255) / 256, a.B), dim3(256), 0, s, a);
        return hipGetLastError();
    }
}
template <int R>
static hipError_t launch_ar_r(const ArMstepArgs& a, hipStream_t s) {
    switch (a.q) {
        case 0: return launch_ar_rq<R, 1>(a, s);
        case 1: return launch_ar_rq<R, 2>(a, s);
        case 2: return launch_ar_rq<R, 3>(a, s);
        case 3: return launch_ar_rq<R, 4>(a, s);
        case 4: return launch_ar_rq<R, 5>(a, s);
        default: return hipErrorInvalidValue;
    }
}
bool mstep_ar_supported(int r, int q) { return r >= 1 && r <= 8 && q >= 0 && q <= 4 && r * (q + 1) <= 32; }
hipError_t launch_mstep_ar(const ArMstepArgs& a, hipStream_t s) {
    note_kernel("mstep_ar_kernel");
    switch (a.r) {
        case 1: return launch_ar_r<1>(a, s);
        case 2: return launch_ar_r<2>(a, s);
        case 3: return launch_ar_r<3>(a, s);
        case 4: return launch_ar_r<4>(a, s);
        case 5: return launch_ar_r<5>(a, s);
        case 6: return launch_ar_r<6>(a, s);
        case 7: return launch_ar_r<7>(a, s);
        case 8: return launch_ar_r<8>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dfm
