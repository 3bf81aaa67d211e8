// boot.hip -- wild-bootstrap impulse-response bands of the factor VAR (BASELINE config 5).
//
// The reference estimates the factor VAR (`estimate_var!`, dfm_functions.ipynb:444-468), fills the
// state-space matrices (`fill_matrices!`, :477-492: companion M, selector Q, G = lower Cholesky of the
// residual covariance) and computes impulse responses by powering the companion matrix
// (`impulse_response`, :793-816: irf[:, t, k] = Q M^t G[:, k]).  It has no bootstrap; the draw loop defined
// here (and restated in oracle/boot_oracle.py) is the standard recursive-design wild bootstrap:
//     e*_t = s_t e_t  (s_t = +-1 Rademacher, one sign per period),   y*_t = c + sum_l A_l y*_{t-l} + e*_t,
//     y*_t = y_t for the first p periods;  re-estimate the VAR on y*  ->  M*, G*  ->  irf*.
// One lane group (R = 1 + ns p regressors padded to a power of two) per draw: lane k owns regressor k -- row k
// of X'X and of X'Y, row k of the coefficient matrix -- the bootstrap series lives in the group's LDS, the
// normal equations are inverted by the Gauss-Jordan of dfm_smallmat.h, the IRF recursion keeps state m in
// lane m + 1.  quantile_kernel then sorts the draws of every (variable, horizon, shock) in LDS (bitonic) and
// picks the nearest-rank order statistics.
#include "dfm_kernels.h"
#include "dfm_philox.h"
#include "dfm_smallmat.h"

namespace dfm {

constexpr int kBootThreads = 256;
constexpr int kBootMaxNs = 8;

template <int R>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = 1; off < R; off <<= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

template <int R>
__global__ __launch_bounds__(kBootThreads) void var_boot_kernel(BootArgs a) {
    constexpr int NG = kBootThreads / R;
    constexpr int NS = kBootMaxNs;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int T = a.T, ns = a.ns, p = a.p, H = a.H;
    const int K = 1 + ns * p;
    const int tid = threadIdx.x;
    const int grp = tid / R, k = tid % R;
    const int GS = T * ns + K * ns + 2 * R;
    double* ys = sm + (size_t)grp * GS;      // [T][ns] bootstrap series
    double* hb = ys + T * ns;                // [K][ns] exchange (X'Y, then the coefficients)
    double* Xg = hb + K * ns;                // [2R] Gauss-Jordan exchange
    const int d = blockIdx.x * NG + grp;
    const bool act = d < a.B;
    const int dd = act ? d : a.B - 1;

    double bk[NS];                           // row k of the point estimate (regressor k -> every equation)
#pragma unroll
    for (int j = 0; j < NS; ++j) bk[j] = (k < K && j < ns) ? a.betahat[(size_t)k * ns + j] : 0.0;
    if (k < ns)
        for (int t = 0; t < p; ++t) ys[t * ns + k] = a.y[(size_t)t * ns + k];
    // regressor k of period t: 1, y*_{t-1}, ..., y*_{t-p}  (dfm_functions.ipynb:446-451)
    auto xreg = [&](int t, int m) -> double {
        if (m == 0) return 1.0;
        if (m >= K) return 0.0;
        return ys[(t - 1 - (m - 1) / ns) * ns + (m - 1) % ns];
    };

    double Grow[R];
#pragma unroll
    for (int j = 0; j < R; ++j) Grow[j] = 0.0;
    double hrow[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) hrow[j] = 0.0;
    for (int t = p; t < T; ++t) {
        wave_lds_sync();
        const double xk = xreg(t, k);
        double sgn;
        if (a.signs) {
            sgn = a.signs[(size_t)dd * T + t];
        } else {
            uint32_t o[4];
            Philox::block(a.seed, (uint64_t)t, (uint64_t)(a.first_draw + dd), o);
            sgn = (o[0] & 1u) ? 1.0 : -1.0;
        }
        double yn[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            yn[j] = 0.0;
            if (j < ns) yn[j] = group_sum<R>(bk[j] * xk) + sgn * a.resid[(size_t)t * ns + j];
        }
#pragma unroll
        for (int j = 0; j < NS; ++j)
            if (j == k && j < ns) ys[t * ns + j] = yn[j];
#pragma unroll
        for (int m = 0; m < R; ++m) Grow[m] = fma(xk, xreg(t, m), Grow[m]);
#pragma unroll
        for (int j = 0; j < NS; ++j) hrow[j] = fma(xk, yn[j], hrow[j]);
    }
    wave_lds_sync();
    if (k >= K) {
#pragma unroll
        for (int j = 0; j < R; ++j) Grow[j] = (j == k) ? 1.0 : 0.0;
    }
    const double dk = equilibrate_rows<R>(Grow, Xg, k);     // the constant next to the levels: scale first
    gj_inverse<R>(Grow, Xg, k);
    __syncthreads();
    if (k < K)
        for (int j = 0; j < ns; ++j) hb[k * ns + j] = hrow[j] * dk;
    __syncthreads();
    double bst[NS];                          // row k of the re-estimated coefficient matrix
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        double s = 0.0;
        if (j < ns && k < K) {
#pragma unroll
            for (int m = 0; m < R; ++m)
                if (m < K) s = fma(Grow[m], hb[m * ns + j], s);
        }
        bst[j] = s * dk;
    }
    __syncthreads();
    if (act && a.beta_out && k < K)
        for (int j = 0; j < ns; ++j) a.beta_out[((size_t)d * K + k) * ns + j] = bst[j];

    // residual covariance (dfm_functions.ipynb:460-463), every lane redundantly
    double S[NS][NS];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j) S[i][j] = 0.0;
    for (int t = p; t < T; ++t) {
        const double xk = xreg(t, k);
        double e[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) e[j] = (j < ns) ? ys[t * ns + j] - group_sum<R>(bst[j] * xk) : 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) S[i][j] = fma(e[i], e[j], S[i][j]);
    }
    const double dof = (double)(T - p - K);
    // lower Cholesky factor in place (fill_matrices!: G = cholesky(seps).U')
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            if (i < ns) {
                double s = S[i][j] / dof;
                if (j == 0 && i == 0) { /* first pivot */ }
#pragma unroll
                for (int q = 0; q < j; ++q) s -= S[i][q] * S[j][q];
                S[i][j] = (i == j) ? sqrt(s) : s / S[j][j];
            }
        }
    // impulse responses: state m in lane m + 1 (dfm_functions.ipynb:793-816)
    const int lane = tid & 63;
    for (int s = 0; s < ns; ++s) {
        double x = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (k == i + 1 && i < ns) x = (i >= s) ? S[i][s] : 0.0;        // G[:, s]: column s of the lower factor
        for (int h = 0; h < H; ++h) {
            if (act && k >= 1 && k <= ns) a.irf[(((size_t)d * ns + (k - 1)) * H + h) * ns + s] = x;
            double top[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) top[i] = (i < ns) ? group_sum<R>((k >= 1 && k < K) ? bst[i] * x : 0.0) : 0.0;
            const double sh = __shfl(x, lane - ns, kWave);                  // state m - ns (same group: k > ns)
            double xn = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i)
                if (k == i + 1 && i < ns) xn = top[i];
            if (k > ns && k < K) xn = sh;
            x = xn;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Nearest-rank quantiles over the draws: series s (one workgroup) = x[0..B)[s]; out[q][s] = the ceil(q B)-th
// smallest draw (numpy.quantile(..., method="inverted_cdf")).  NaN draws sort last.
__global__ __launch_bounds__(256) void quantile_kernel(QuantArgs a) {
    extern __shared__ __attribute__((aligned(16))) double v[];
    const int s = blockIdx.x, tid = threadIdx.x;
    int n2 = 1;
    while (n2 < a.B) n2 <<= 1;
    for (int i = tid; i < n2; i += 256) {
        double x = i < a.B ? a.x[(size_t)i * a.S + s] : __builtin_huge_val();
        v[i] = (x == x) ? x : __builtin_huge_val();
    }
    __syncthreads();
    for (int len = 2; len <= n2; len <<= 1)
        for (int st = len >> 1; st >= 1; st >>= 1) {
            for (int i = tid; i < n2; i += 256) {
                const int j = i ^ st;
                if (j > i) {
                    const bool up = (i & len) == 0;
                    const double x = v[i], y = v[j];
                    if ((x > y) == up) { v[i] = y; v[j] = x; }
                }
            }
            __syncthreads();
        }
    for (int q = tid; q < a.nq; q += 256) {
        int idx = (int)ceil(a.q[q] * (double)a.B) - 1;
        idx = idx < 0 ? 0 : (idx >= a.B ? a.B - 1 : idx);
        a.out[(size_t)q * a.S + s] = v[idx];
    }
}

template <int R>
static hipError_t launch_boot_r(const BootArgs& a, hipStream_t s) {
    constexpr int NG = kBootThreads / R;
    const int K = 1 + a.ns * a.p;
    const size_t lds = (size_t)NG * ((size_t)a.T * a.ns + (size_t)K * a.ns + 2 * R) * sizeof(double);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static LdsOptIn attr_done;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&var_boot_kernel<R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((var_boot_kernel<R>), dim3((a.B + NG - 1) / NG), dim3(kBootThreads), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_var_boot(const BootArgs& a, hipStream_t s) {
    const int K = 1 + a.ns * a.p;
    if (a.ns > kBootMaxNs || K > 64) return hipErrorInvalidValue;
    if (K <= 8) return launch_boot_r<8>(a, s);
    if (K <= 16) return launch_boot_r<16>(a, s);
    if (K <= 32) return launch_boot_r<32>(a, s);
    return launch_boot_r<64>(a, s);
}

hipError_t launch_quantiles(const QuantArgs& a, hipStream_t s) {
    int n2 = 1;
    while (n2 < a.B) n2 <<= 1;
    const size_t lds = (size_t)n2 * sizeof(double);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static LdsOptIn attr_done;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&quantile_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(quantile_kernel, dim3(a.S), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace dfm
