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
// Round 6: all three steps from per-series moments that do not depend on the series' parameters, the moments as products on the
// f64 matrix pipe (ar_moments_kernel), then four lanes per series for the solves (ar_solve_kernel); the two-sweep kernel with a
// thread per series (mstep_ar_kernel: 4.3 ms per 1024 replicates at the Stock-Watson shape) is kept in the diagnostics library.
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

#ifdef DFM_DIAG   // (the two-sweep kernel: diagnostics library only, DFM_AR_MSTEP_OLD=1)
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

#endif

// ---- round 6: the same CM-steps from per-series MOMENTS, the moments as matrix products ----------------------------------------
// Both sweeps of mstep_ar_kernel re-derive, per series and period, sums that do not depend on the series' parameters:
//     W_i   = sum_{t ok} E[z_t z_t'] (leading k1 = (q + 1) r states, packed),   ZX_i[l'] = sum_{t ok} x_{i,t+q-l'} z_t,
//     XX_i  = sum_{t ok} x_{i,t+q-l} x_{i,t+q-l'},   n_i            (t ok: x_it and its q lags observed)
// with which  E[g g'] = sum_{l,m} a_l a_m W_i[(l,.),(m,.)],  g'x~ = sum_{l,l'} a_l a_l' ZX_i[l'][(l,.)]  and
//     U_i[l][m] = XX_i[l][m] - lam' ZX_i[m][(l,.)] - lam' ZX_i[l][(m,.)] + lam' W_i[(l,.),(m,.)] lam
// -- 800 FMAs per series and period become 25 matrix instructions per 16 series and 4 periods.  ar_moments_kernel: one workgroup
// per (replicate, 16 SGW series), the panel's columns of the block staged through LDS in chunks of 128 periods, A operands = the
// ok mask (W tiles) or the masked lag l' of the panel (ZX tiles), B operands = the rows of V_t = [vec(E z z') | z] (mmw_vec_kernel)
// straight from L2, a step ahead; the tiles dealt to the four waves statically (a body per wave index).  OUT is stored
// series-fastest so that ar_solve_kernel (a thread per series: the r x r and q x q solves of the old kernel) reads it coalesced.
namespace {

typedef double ar_v4 __attribute__((ext_vector_type(4)));

template <int R, int Q1>
struct ArGeo {
    static constexpr int K1 = R * Q1, NPR = K1 * (K1 + 1) / 2, NTM = (NPR + 15) / 16, NZF = (K1 + 15) / 16;
    static constexpr int TT = NTM + Q1 * NZF, TPW = (TT + 3) / 4;
    static constexpr int SGW = TPW <= 6 ? 3 : (TPW <= 9 ? 2 : 1);
    static constexpr int NXX = Q1 * (Q1 + 1) / 2;
};
constexpr int kArTC = 128;                                     // periods per staged chunk of the panel block
constexpr int kArPF = 3;                                       // B operands in flight: steps ahead

template <int R, int Q1, int W>
__device__ __forceinline__ void ar_mom_body(const ArMstepArgs& a, const double* __restrict__ V, double* __restrict__ OUT,
                                            double* __restrict__ SM, int ntm16, int VW, int Ns, double* Xs) {
    using G = ArGeo<R, Q1>;
    constexpr int Q = Q1 - 1, TPW = G::TPW, SGW = G::SGW, SW = 16 * SGW;
    constexpr int NT = (G::TT - W * TPW) < TPW ? ((G::TT - W * TPW) > 0 ? (G::TT - W * TPW) : 0) : TPW;   // tiles of this wave
    const int b = blockIdx.y, i0 = (int)blockIdx.x * SW;
    const int T = a.T, N = a.N, Tq = T - Q;
    const int tid = threadIdx.x, lane = tid & 63, k4 = lane >> 4, c16 = lane & 15;
    const double* xb = a.panel + (size_t)b * T * N;
    const double* Vb = V + (size_t)b * Tq * VW;
    ar_v4 acc[SGW][NT > 0 ? NT : 1];
#pragma unroll
    for (int g = 0; g < SGW; ++g)
#pragma unroll
        for (int x = 0; x < (NT > 0 ? NT : 1); ++x) acc[g][x] = ar_v4{0.0, 0.0, 0.0, 0.0};
    double xx[SGW][G::NXX], nn[SGW];
#pragma unroll
    for (int g = 0; g < SGW; ++g) {
        nn[g] = 0.0;
#pragma unroll
        for (int e = 0; e < G::NXX; ++e) xx[g][e] = 0.0;
    }
    // column (in V) of tile slot x of this wave: W tile d -> 16 d; ZX tile (l', z) -> ntm16 + 16 z
    auto vcol = [&](int x) { const int tile = W * TPW + x; return tile < G::NTM ? 16 * tile : ntm16 + 16 * ((tile - G::NTM) % G::NZF); };
    for (int t0 = 0; t0 < Tq; t0 += kArTC) {
        const int nrow = (Tq - t0 < kArTC ? Tq - t0 : kArTC) + Q;          // panel rows t0 .. t0 + nrow - 1 of the block's columns
        __syncthreads();
        for (int e = tid; e < nrow * SW; e += 256) {
            const int rr = e / SW, cc = e - rr * SW;
            Xs[e] = (i0 + cc < N) ? xb[(size_t)(t0 + rr) * N + i0 + cc] : __builtin_nan("");
        }
        __syncthreads();
        const int nks = ((Tq - t0 < kArTC ? Tq - t0 : kArTC) + 3) / 4;
        // B operands kArPF steps ahead (a step is ~0.25 us of matrix pipe: one step ahead does not cover a trip to L2)
        double bq[kArPF][NT > 0 ? NT : 1];
        auto ldb = [&](int s, double (&dst)[NT > 0 ? NT : 1]) {
            int t = t0 + 4 * s + k4; t = t < Tq ? t : Tq - 1;
#pragma unroll
            for (int x = 0; x < NT; ++x) dst[x] = Vb[(size_t)t * VW + vcol(x) + c16];
        };
#pragma unroll
        for (int d = 0; d < kArPF; ++d) ldb(d, bq[d]);
        for (int s = 0; s < nks; ++s) {
            double bv[NT > 0 ? NT : 1];
#pragma unroll
            for (int x = 0; x < NT; ++x) bv[x] = bq[0][x];
#pragma unroll
            for (int d = 0; d + 1 < kArPF; ++d)
#pragma unroll
                for (int x = 0; x < NT; ++x) bq[d][x] = bq[d + 1][x];
            ldb(s + kArPF, bq[kArPF - 1]);                                   // (past the chunk's end: clamped rows, never used)
            const int tl = 4 * s + k4;                                      // period within the chunk; lag l' sits Q - l' rows further
            const bool tv = t0 + tl < Tq;
#pragma unroll
            for (int g = 0; g < SGW; ++g) {
                double X[Q1];
                bool ok = tv;
#pragma unroll
                for (int l = 0; l < Q1; ++l) {
                    X[l] = tv ? Xs[(size_t)(tl + Q - l) * SW + 16 * g + c16] : 0.0;
                    ok = ok && (X[l] == X[l]);
                }
                const double am = ok ? 1.0 : 0.0;
#pragma unroll
                for (int l = 0; l < Q1; ++l) X[l] = ok ? X[l] : 0.0;
#pragma unroll
                for (int x = 0; x < NT; ++x) {
                    const int tile = W * TPW + x;
                    const double av = tile < G::NTM ? am : X[(tile - G::NTM) / G::NZF];
                    acc[g][x] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[x], acc[g][x], 0, 0, 0);
                }
                if (W == 3) {                                                // (the wave with the fewest tiles keeps the scalar sums)
                    nn[g] += am;
#pragma unroll
                    for (int l = 0; l < Q1; ++l)
#pragma unroll
                        for (int l2 = 0; l2 <= l; ++l2) xx[g][l * (l + 1) / 2 + l2] = fma(X[l], X[l2], xx[g][l * (l + 1) / 2 + l2]);
                }
            }
        }
    }
    // 16x16x4 D[(lane / 16) + 4 v][lane % 16]: series 16 g + k4 + 4 v of the block, column c16 of the tile; OUT[b][column][series]
#pragma unroll
    for (int g = 0; g < SGW; ++g)
#pragma unroll
        for (int x = 0; x < NT; ++x) {
            const int tile = W * TPW + x;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = i0 + 16 * g + k4 + 4 * v;
                if (i < N) OUT[((size_t)b * (16 * G::TT) + 16 * tile + c16) * Ns + i] = acc[g][x][v];
            }
        }
    if (W == 3) {
#pragma unroll
        for (int g = 0; g < SGW; ++g) {
            const int i = i0 + 16 * g + c16;
            double n = nn[g];
            n += __shfl_xor(n, 16, 64); n += __shfl_xor(n, 32, 64);
#pragma unroll
            for (int e = 0; e < G::NXX; ++e) {
                double v = xx[g][e];
                v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                if (k4 == 0 && i < N) SM[((size_t)b * 16 + e) * Ns + i] = v;
            }
            if (k4 == 0 && i < N) SM[((size_t)b * 16 + 15) * Ns + i] = n;
        }
    }
}

}  // namespace

template <int R, int Q1>
__global__ __launch_bounds__(256) void ar_moments_kernel(ArMstepArgs a, const double* __restrict__ V, double* __restrict__ OUT,
                                                         double* __restrict__ SM, int ntm16, int VW, int Ns) {
    extern __shared__ __attribute__((aligned(16))) double ar_xs[];
    if (a.active && a.active[blockIdx.y] == 0) return;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (w == 0) ar_mom_body<R, Q1, 0>(a, V, OUT, SM, ntm16, VW, Ns, ar_xs);
    else if (w == 1) ar_mom_body<R, Q1, 1>(a, V, OUT, SM, ntm16, VW, Ns, ar_xs);
    else if (w == 2) ar_mom_body<R, Q1, 2>(a, V, OUT, SM, ntm16, VW, Ns, ar_xs);
    else ar_mom_body<R, Q1, 3>(a, V, OUT, SM, ntm16, VW, Ns, ar_xs);
}

// four lanes per series: steps (2)-(4) from the moments (the solves and the rules of mstep_ar_kernel).  The (l, l') lag pairs of the
// contractions are dealt to the four lanes (a thread per series left three waves per SIMD to hide ~900 loads each: 0.60 ms), the
// partial sums meet by two shuffles, the small solves run in all four.
template <int R, int Q1>
__global__ __launch_bounds__(256) void ar_solve_kernel(ArMstepArgs a, const double* __restrict__ OUT, const double* __restrict__ SM, int Ns) {
    using G = ArGeo<R, Q1>;
    constexpr int Q = Q1 - 1, NE = Q1 * Q1, NU = Q1 * (Q1 + 1) / 2, NUL = (NU + 3) / 4;
    const int b = blockIdx.y;
    const int lane4 = (int)threadIdx.x & 3;
    const int i = (int)blockIdx.x * 64 + ((int)threadIdx.x >> 2);
    if (a.active && a.active[b] == 0) return;
    const int N = a.N;
    const bool live = i < N;
    const int ii = live ? i : N - 1;
    const double* Wc = OUT + (size_t)b * (16 * G::TT) * Ns + ii;                 // column c of the series: Wc[c * Ns]
    const double* Sc = SM + (size_t)b * 16 * Ns + ii;
    auto pk = [](int u, int v) { return u >= v ? u * (u + 1) / 2 + v : v * (v + 1) / 2 + u; };
    auto zx = [&](int lp, int u) { return Wc[(size_t)(16 * (G::NTM + lp * G::NZF) + u) * Ns]; };   // ZX[l'][u]
    double lam[R], av[Q1];
#pragma unroll
    for (int c = 0; c < R; ++c) lam[c] = a.Lam[((size_t)b * N + ii) * R + c];
    av[0] = 1.0;
#pragma unroll
    for (int l = 1; l < Q1; ++l) av[l] = -a.rho[((size_t)b * N + ii) * Q + (l - 1)];
    auto avd = [&](int l) {                                    // av[l] for a lane-dependent l
        double v = av[0];
#pragma unroll
        for (int k = 1; k < Q1; ++k) v = l == k ? av[k] : v;
        return v;
    };
    const int n = (int)(Sc[(size_t)15 * Ns] + 0.5);
    const bool enough = n >= R + Q + 1;                       // as the oracle: fewer quasi-differenced cells -> series left as is
    // (2) loadings given rho: lane p takes the pairs e = l Q1 + l' = p (mod 4)
    double LH[R][R], RH[R];
#pragma unroll
    for (int c = 0; c < R; ++c) {
        RH[c] = 0.0;
#pragma unroll
        for (int d = 0; d < R; ++d) LH[c][d] = 0.0;
    }
    for (int e = lane4; e < NE; e += 4) {
        const int l = e / Q1, l2 = e - l * Q1;
        const double w = avd(l) * avd(l2);
#pragma unroll
        for (int c = 0; c < R; ++c) {
#pragma unroll
            for (int d = 0; d <= c; ++d) LH[c][d] = fma(w, Wc[(size_t)pk(l * R + c, l2 * R + d) * Ns], LH[c][d]);
        }
#pragma unroll
        for (int c = 0; c < R; ++c) RH[c] = fma(w, zx(l2, l * R + c), RH[c]);
    }
#pragma unroll
    for (int c = 0; c < R; ++c) {
        RH[c] += __shfl_xor(RH[c], 1, 64); RH[c] += __shfl_xor(RH[c], 2, 64);
#pragma unroll
        for (int d = 0; d <= c; ++d) { LH[c][d] += __shfl_xor(LH[c][d], 1, 64); LH[c][d] += __shfl_xor(LH[c][d], 2, 64); }
    }
#pragma unroll
    for (int c = 0; c < R; ++c)
#pragma unroll
        for (int d = c + 1; d < R; ++d) LH[c][d] = LH[d][c];
    {
        const bool pd = chol_solve_reg<R>(LH, RH, R);
        if (pd) {
#pragma unroll
            for (int c = 0; c < R; ++c) lam[c] = RH[c];
        }
    }
    // (3), (4) rho and sig2 given the new loadings: lane p takes the entries (l >= l') number p, p + 4, ..
    double mine[NUL];
#pragma unroll
    for (int k = 0; k < NUL; ++k) {
        int e = 4 * k + lane4;
        e = e < NU ? e : NU - 1;                               // (a duplicate: never read)
        int l = 0;
        while ((l + 1) * (l + 2) / 2 <= e) ++l;
        const int l2 = e - l * (l + 1) / 2;
        double s = Sc[(size_t)e * Ns];
#pragma unroll
        for (int c = 0; c < R; ++c) {
            s = fma(-lam[c], zx(l2, l * R + c), s);
            s = fma(-lam[c], zx(l, l2 * R + c), s);
        }
#pragma unroll
        for (int c = 0; c < R; ++c)
#pragma unroll
            for (int d = 0; d < R; ++d) s = fma(lam[c] * lam[d], Wc[(size_t)pk(l * R + c, l2 * R + d) * Ns], s);
        mine[k] = s;
    }
    double U[Q1][Q1];
    const int grp = ((int)threadIdx.x & 63) & ~3;
#pragma unroll
    for (int l = 0; l < Q1; ++l)
#pragma unroll
        for (int l2 = 0; l2 <= l; ++l2) {
            const int e = l * (l + 1) / 2 + l2;
            U[l][l2] = __shfl(mine[e >> 2], grp | (e & 3), 64);
        }
    if (!(live && enough) || lane4 != 0) return;
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

// V | OUT | SM (doubles) of the moment form; Ns = N rounded up to 16
static size_t ar_ws_doubles(int B, int T, int N, int r, int q, int Rk, size_t* oV = nullptr, size_t* oOUT = nullptr, int* pVW = nullptr,
                            int* pntm16 = nullptr, int* pNs = nullptr, int* pTT = nullptr) {
    const int k1 = r * (q + 1), npr = k1 * (k1 + 1) / 2, ntm = (npr + 15) / 16, nzf = (k1 + 15) / 16;
    const int VW = 16 * ntm + 16 * ((Rk + 15) / 16), Ns = (N + 15) & ~15, TT = ntm + (q + 1) * nzf;
    const size_t nV = (size_t)B * (T - q) * VW, nOUT = (size_t)B * 16 * TT * Ns, nSM = (size_t)B * 16 * Ns;
    if (oV) *oV = nV;
    if (oOUT) *oOUT = nOUT;
    if (pVW) *pVW = VW;
    if (pntm16) *pntm16 = 16 * ntm;
    if (pNs) *pNs = Ns;
    if (pTT) *pTT = TT;
    return nV + nOUT + nSM;
}
size_t mstep_ar_workspace(int B, int T, int N, int r, int q, int Rk) { return ar_ws_doubles(B, T, N, r, q, Rk) * sizeof(double) + 256; }

template <int R, int Q1>
static hipError_t launch_ar_rq(const ArMstepArgs& a, double* ws, hipStream_t s) {
    if constexpr (R * Q1 > 32) {
        return hipErrorInvalidValue;
    } else {
#ifdef DFM_DIAG
        const char* ov = diag_env("DFM_AR_MSTEP_OLD");        // (read per launch: the A/B test switches it inside one process)
        const bool old_form = ov && atoi(ov) != 0;
        if (old_form) {
            hipLaunchKernelGGL((mstep_ar_kernel<R, Q1>), dim3((a.N + 255) / 256, a.B), dim3(256), 0, s, a);
            return hipGetLastError();
        }
#endif
        if (ws == nullptr) return hipErrorInvalidValue;
        using G = ArGeo<R, Q1>;
        size_t nV = 0, nOUT = 0;
        int VW = 0, ntm16 = 0, Ns = 0, TT = 0;
        ar_ws_doubles(a.B, a.T, a.N, a.r, a.q, a.Rk, &nV, &nOUT, &VW, &ntm16, &Ns, &TT);
        double *V = ws, *OUT = V + nV, *SM = OUT + nOUT;
        hipError_t e = launch_mmw_vec(a.zsm, a.Psm, a.active, a.B, a.T - a.q, G::K1, a.Rk, ntm16, VW, V, s);
        if (e != hipSuccess) return e;
        constexpr int SW = 16 * G::SGW;
        const size_t lds = (size_t)(kArTC + Q1 - 1) * SW * sizeof(double);
        hipLaunchKernelGGL((ar_moments_kernel<R, Q1>), dim3((a.N + SW - 1) / SW, a.B), dim3(256), lds, s, a, (const double*)V, OUT, SM, ntm16, VW, Ns);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        hipLaunchKernelGGL((ar_solve_kernel<R, Q1>), dim3((a.N + 63) / 64, a.B), dim3(256), 0, s, a, (const double*)OUT, (const double*)SM, Ns);
        return hipGetLastError();
    }
}
template <int R>
static hipError_t launch_ar_r(const ArMstepArgs& a, double* ws, hipStream_t s) {
    switch (a.q) {
        case 0: return launch_ar_rq<R, 1>(a, ws, s);
        case 1: return launch_ar_rq<R, 2>(a, ws, s);
        case 2: return launch_ar_rq<R, 3>(a, ws, s);
        case 3: return launch_ar_rq<R, 4>(a, ws, s);
        case 4: return launch_ar_rq<R, 5>(a, ws, s);
        default: return hipErrorInvalidValue;
    }
}
bool mstep_ar_supported(int r, int q) { return r >= 1 && r <= 8 && q >= 0 && q <= 4 && r * (q + 1) <= 32; }
// ws: mstep_ar_workspace bytes
hipError_t launch_mstep_ar(const ArMstepArgs& a, double* ws, hipStream_t s) {
    note_kernel("mstep_ar_kernel");
    switch (a.r) {
        case 1: return launch_ar_r<1>(a, ws, s);
        case 2: return launch_ar_r<2>(a, ws, s);
        case 3: return launch_ar_r<3>(a, ws, s);
        case 4: return launch_ar_r<4>(a, ws, s);
        case 5: return launch_ar_r<5>(a, ws, s);
        case 6: return launch_ar_r<6>(a, ws, s);
        case 7: return launch_ar_r<7>(a, ws, s);
        case 8: return launch_ar_r<8>(a, ws, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dfm
