// breaks.hip -- batched Chow statistics with HAC covariance: the reference's
// `compute_chow` / `regress_hac` / `hac` / `form_hscrc` / `form_kernel` (dfm_functions.ipynb:832-977) and, through the
// problem list, `compute_qlr` (:1019-1047: the maximum over ~0.7 T break dates, plain and HAC) -- 207 series x ~157 break
// dates x 2 bandwidths per number of factors in the driver's Table 4 (Stock_Watson.ipynb:1064-1120).
//
// Problem p = (series s_p, break date tau_p, bandwidth q_p):  regress y on W = [X, X D], D_t = 1 for t >= tau
// (0-based: the first tau rows are "before"), beta = (W'W)^-1 W'y, u = y - W beta, z_t = w_t u_t,
//     v = sum_t z_t z_t' + sum_{l=1..q} (1 - l/(q+1)) sum_t (z_t z_{t+l}' + z_{t+l} z_t'),   V = (W'W)^-1 v (W'W)^-1,
//     chow = gamma' V[k:, k:]^-1 gamma,   gamma = beta[k:].
// One lane group of R = 2k (padded) lanes per problem: lane m owns regressor m -- row m of W'W, of v, of V; rows are
// exchanged through LDS; the last q + 1 score vectors z_t live in an LDS ring.  y and X of a series are shared by
// all its problems (L2 / L1 hits).
// Synchronisation: a lane group (R <= 16 lanes) never spans two waves and its LDS region is private, so every exchange
// is fenced at WAVE level (wave_lds_sync, gj_inverse<R, true>): the problems of one workgroup belong to series of
// different lengths T, and a workgroup barrier inside the `t < T` loops would be executed a different number of times
// by different waves.
#include "dfm_kernels.h"
#include "dfm_smallmat.h"

namespace dfm {

constexpr int kChowThreads = 256;
constexpr int kChowMaxQ = 15;

template <int R>
__global__ __launch_bounds__(kChowThreads) void chow_kernel(ChowArgs a) {
    constexpr int NG = kChowThreads / R;
    constexpr int GS = 2 * R + R + (kChowMaxQ + 1) * R + 2 * R * R;
    __shared__ double sm[NG * GS];
    const int tid = threadIdx.x;
    const int grp = tid / R, m = tid % R;
    double* Xg = sm + grp * GS;              // [2R] Gauss-Jordan exchange
    double* hb = Xg + 2 * R;                 // [R] vector exchange
    double* zr = hb + R;                     // [kChowMaxQ + 1][R] ring of score vectors
    double* M1 = zr + (kChowMaxQ + 1) * R;   // [R][R]
    double* M2 = M1 + R * R;                 // [R][R]
    const int p = blockIdx.x * NG + grp;
    const bool act = p < a.P;
    const int pp = act ? p : a.P - 1;
    const int s = a.prob_series[pp], tau = a.prob_break[pp], q = a.prob_q[pp];
    const int k = a.k, K2 = 2 * k;
    const int T = a.Tlen[s];
    const double* __restrict__ y = a.y + (size_t)s * a.Tmax;
    const double* __restrict__ X = a.X + (size_t)s * a.Tmax * k;
    const int mk = m < k ? m : m - k;        // column of X behind regressor m
    const bool isD = m >= k;                 // interaction regressor
    auto wreg = [&](int t, int n) -> double {   // regressor n of period t
        if (n >= K2) return 0.0;
        const double x = X[(size_t)t * k + (n < k ? n : n - k)];
        return (n >= k && t < tau) ? 0.0 : x;
    };

    // normal equations
    double G[R];
#pragma unroll
    for (int j = 0; j < R; ++j) G[j] = 0.0;
    double h = 0.0;
    for (int t = 0; t < T; ++t) {
        const double wm = (m < K2) ? ((isD && t < tau) ? 0.0 : X[(size_t)t * k + mk]) : 0.0;
        h = fma(wm, y[t], h);
#pragma unroll
        for (int j = 0; j < R; ++j) G[j] = fma(wm, wreg(t, j), G[j]);
    }
    if (m >= K2) {
#pragma unroll
        for (int j = 0; j < R; ++j) G[j] = (j == m) ? 1.0 : 0.0;
    }
    const double dk = equilibrate_rows<R, true>(G, Xg, m);
    gj_inverse<R, true>(G, Xg, m);                 // G <- (D W'W D)^-1 row m
    wave_lds_sync();
    hb[m] = h * dk;
    wave_lds_sync();
    double beta = 0.0;
#pragma unroll
    for (int j = 0; j < R; ++j) beta = fma(G[j], hb[j], beta);
    beta *= dk;
    wave_lds_sync();
    hb[m] = beta;                            // coefficients, for the residuals
    // (W'W)^-1 = D Ginv D: keep it in M2 for the sandwich
    Xg[m] = dk;
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < R; ++j) M2[m * R + j] = G[j] * dk * Xg[j];
    // scores and their lagged products
    double v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = 0.0;
    const double qp1 = (double)(q + 1);
    for (int t = 0; t < T; ++t) {
        double fit = 0.0;
        for (int n = 0; n < K2; ++n) fit = fma(wreg(t, n), hb[n], fit);
        const double u = y[t] - fit;
        const double wm = (m < K2) ? ((isD && t < tau) ? 0.0 : X[(size_t)t * k + mk]) : 0.0;
        const double zt = wm * u;
        double* cur = zr + (t % (kChowMaxQ + 1)) * R;
        wave_lds_sync();
        cur[m] = zt;
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = fma(zt, cur[j], v[j]);                  // lag 0
        for (int l = 1; l <= q && l <= t; ++l) {
            const double* old = zr + ((t - l) % (kChowMaxQ + 1)) * R;
            const double kw = 1.0 - (double)l / qp1;
            const double zo = old[m];
#pragma unroll
            for (int j = 0; j < R; ++j) v[j] = fma(kw, fma(zt, old[j], zo * cur[j]), v[j]);   // z_t z_{t-l}' + z_{t-l} z_t'
        }
    }
    // V = (W'W)^-1 v (W'W)^-1
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < R; ++j) M1[m * R + j] = v[j];
    wave_lds_sync();
    double t1[R], Vr[R];
    {
        double gi[R];
#pragma unroll
        for (int j = 0; j < R; ++j) gi[j] = M2[m * R + j];
        mm_rows<R>(t1, gi, M1);              // row m of (W'W)^-1 v
    }
    mm_rows<R>(Vr, t1, M2);                  // row m of V
    // chow = gamma' V22^-1 gamma: invert blockdiag(I, V22)
    double B[R];
#pragma unroll
    for (int j = 0; j < R; ++j) B[j] = (m >= k && m < K2 && j >= k && j < K2) ? Vr[j] : ((j == m) ? 1.0 : 0.0);
    const double d2 = equilibrate_rows<R, true>(B, Xg, m);
    gj_inverse<R, true>(B, Xg, m);
    wave_lds_sync();
    const double gam = (m >= k && m < K2) ? beta : 0.0;
    hb[m] = gam * d2;
    wave_lds_sync();
    double w2 = 0.0;
#pragma unroll
    for (int j = 0; j < R; ++j) w2 = fma(B[j], hb[j], w2);
    double part = gam * w2 * d2;
#pragma unroll
    for (int off = 1; off < R; off <<= 1) part += __shfl_xor(part, off, kWave);
    if (act && m == 0) a.chow[p] = part;
}

template <int R>
static hipError_t launch_chow_r(const ChowArgs& a, hipStream_t s) {
    constexpr int NG = kChowThreads / R;
    hipLaunchKernelGGL((chow_kernel<R>), dim3((a.P + NG - 1) / NG), dim3(kChowThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_chow(const ChowArgs& a, hipStream_t s) {
    const int K2 = 2 * a.k;
    if (K2 > 16 || a.P < 1) return hipErrorInvalidValue;   // k <= 8 regressors (the driver uses 4 and 8 factors)
    if (K2 <= 2) return launch_chow_r<2>(a, s);
    if (K2 <= 4) return launch_chow_r<4>(a, s);
    if (K2 <= 8) return launch_chow_r<8>(a, s);
    return launch_chow_r<16>(a, s);
}

}  // namespace dfm
