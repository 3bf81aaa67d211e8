// als.hip -- the reference's NON-parametric estimator on the GPU: batched alternating least squares with
// missing cells (`estimate_factor!`, dfm_functions.ipynb:328-382) and the batched complete-case OLS it is
// made of (`ols_skipmissing`, :242-286; used on its own by `estimate_factor_loading!` :391-415, `uar`
// :305-311 and `estimate_var!` :444-468).  SURVEY.md section 8(f1): the independent runs issued by
// `estimate_factor_numbers` / `amengual_watson_test` (:698-768; 77-195 runs per table), bootstrap draws and
// Monte-Carlo replicates are the batch axis -- one workgroup per run.
//
// One sweep (dfm_functions.ipynb:352-366), on the standardised window z (T x N, NaN = missing):
//   loadings | factors : for every series i with >= nt_min observed periods, lam_i = argmin over the
//                        observed t of sum (z_ti - lam' f_t)^2          (:355-362; others stay undefined)
//   factors | loadings : for every period t, f_t = argmin over the series observed in t that have loadings
//                        of sum (z_ti - lam_i' f)^2, with the NEW loadings (:364-365); SSR = sum of those
//                        squared residuals (:366)
//   stop when |SSR_old - SSR| < tol T N, SSR_old = 0 before the first sweep (:349-353, :367-368)
//
// Mapping (wave64): a group of R lanes (R = r padded to 2..32) owns one regression: lane k accumulates row k
// of the normal matrix and entry k of the right-hand side over the observed cells, the R x R system is
// inverted in place by the Gauss-Jordan of dfm_smallmat.h (rows exchanged through LDS) and the solution is
// a row-times-vector product.  Padded factors get a unit diagonal and a zero right-hand side, so their
// coefficients are exactly 0 and problems with different numbers of factors share one kernel.  Factors and
// loadings live in LDS for the whole run; the panel is re-read from L2 every sweep.
// The reference solves each regression by Householder QR (`X\y`, :205-210); the normal equations agree
// with it to ~1e-12 on these well-conditioned problems (tests/test_oracle_sw.py::test_solvers_agree).
#include "dfm_kernels.h"
#include "dfm_smallmat.h"

namespace dfm {

constexpr int kAlsThreads = 256;

template <int R>
struct AlsLds {   // doubles
    static constexpr int NG = kAlsThreads / R;
    static __host__ __device__ size_t doubles(int T, int N) {
        return (size_t)T * R + (size_t)N * R + (size_t)N + (size_t)NG * 2 * R + (size_t)NG * R + 16;
    }
};

// Inverse-based solve of the group's R x R system: lane k holds row k of G (in: normal matrix, out: its
// inverse) and h_k; returns x_k = sum_j Ginv[k][j] h_j.  Every thread of the workgroup must call it.
template <int R>
__device__ __forceinline__ double group_solve(double (&Grow)[R], double hk, double* Xg, double* hb, int k) {
    gj_inverse<R>(Grow, Xg, k);
    __syncthreads();
    hb[k] = hk;
    __syncthreads();
    double x = 0.0;
#pragma unroll
    for (int j = 0; j < R; ++j) x = fma(Grow[j], hb[j], x);
    return x;
}

template <int R>
__global__ __launch_bounds__(kAlsThreads) void als_kernel(AlsArgs a) {
    using LY = AlsLds<R>;
    constexpr int NG = LY::NG;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int T = a.T, N = a.N;
    double* fs = sm;                         // [T][R] factors (padded columns 0)
    double* lam = fs + (size_t)T * R;        // [N][R] loadings (rows of series without loadings: 0, good = 0)
    double* good = lam + (size_t)N * R;      // [N] 1.0 / 0.0
    double* Xall = good + N;                 // [NG][2R] Gauss-Jordan exchange
    double* hball = Xall + NG * 2 * R;       // [NG][R]
    double* red = hball + NG * R;            // [16]
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int grp = tid / R, k = tid % R;
    double* Xg = Xall + grp * 2 * R;
    double* hb = hball + grp * R;
    const int rb = a.r_each ? a.r_each[b] : a.rmax;          // factors of this run (<= rmax <= R)
    const double* __restrict__ z = a.z + (size_t)b * a.z_stride;
    double* Fg = a.F + (size_t)b * T * a.rmax;
    double* Lg = a.Lam + (size_t)b * N * a.rmax;

    for (int e = tid; e < T * R; e += kAlsThreads) {
        const int t = e / R, j = e % R;
        fs[e] = j < rb ? Fg[(size_t)t * a.rmax + j] : 0.0;
    }
    __syncthreads();

    // loadings given factors, for the series chunk [i0, i0 + NG); also used for the R2 epilogue
    auto lambda_step = [&]() {
        for (int i0 = 0; i0 < N; i0 += NG) {
            const int i = i0 + grp;
            const bool act = i < N;
            double Grow[R];
#pragma unroll
            for (int j = 0; j < R; ++j) Grow[j] = 0.0;
            double hk = 0.0;
            int cnt = 0;
            if (act) {
#pragma unroll 4
                for (int t = 0; t < T; ++t) {
                    const double zv = z[(size_t)t * N + i];
                    const bool obs = zv == zv;
                    const double zz = obs ? zv : 0.0;
                    const double fk = obs ? fs[t * R + k] : 0.0;
                    cnt += obs ? 1 : 0;
                    hk = fma(zz, fk, hk);
#pragma unroll
                    for (int j = 0; j < R; ++j) Grow[j] = fma(fk, fs[t * R + j], Grow[j]);
                }
            }
            const bool ok = act && cnt >= a.nt_min;
            if (!ok) {                                       // keep the elimination finite
#pragma unroll
                for (int j = 0; j < R; ++j) Grow[j] = (j == k) ? 1.0 : 0.0;
                hk = 0.0;
            } else if (k >= rb) {
#pragma unroll
                for (int j = 0; j < R; ++j) Grow[j] = (j == k) ? 1.0 : 0.0;   // padded factor: coefficient 0
                hk = 0.0;
            } else {
#pragma unroll
                for (int j = 0; j < R; ++j) Grow[j] = (j >= rb) ? 0.0 : Grow[j];
            }
            const double x = group_solve<R>(Grow, hk, Xg, hb, k);
            if (act) {
                lam[i * R + k] = ok ? x : 0.0;
                if (k == 0) good[i] = ok ? 1.0 : 0.0;
            }
            __syncthreads();
        }
    };

    double ssr = 0.0, ssr_old = 0.0;
    int it = 0;
    const double thresh = a.tol * (double)T * (double)N;
    for (it = 1; it <= a.max_iter; ++it) {
        ssr_old = ssr;
        lambda_step();
        // factors given loadings, period chunk [t0, t0 + NG), then the residuals of those periods
        double ssr_part = 0.0;
        for (int t0 = 0; t0 < T; t0 += NG) {
            const int t = t0 + grp;
            const bool act = t < T;
            double Grow[R];
#pragma unroll
            for (int j = 0; j < R; ++j) Grow[j] = 0.0;
            double hk = 0.0;
            if (act) {
                const double* zr = z + (size_t)t * N;
#pragma unroll 4
                for (int i = 0; i < N; ++i) {
                    const double zv = zr[i];
                    const bool use = (zv == zv) && good[i] != 0.0;
                    const double zz = use ? zv : 0.0;
                    const double lk = use ? lam[i * R + k] : 0.0;
                    hk = fma(zz, lk, hk);
#pragma unroll
                    for (int j = 0; j < R; ++j) Grow[j] = fma(lk, lam[i * R + j], Grow[j]);
                }
            }
            if (!act || k >= rb) {
#pragma unroll
                for (int j = 0; j < R; ++j) Grow[j] = (j == k) ? 1.0 : 0.0;
                hk = 0.0;
            } else {
#pragma unroll
                for (int j = 0; j < R; ++j) Grow[j] = (j >= rb) ? 0.0 : Grow[j];
            }
            const double x = group_solve<R>(Grow, hk, Xg, hb, k);
            if (act) fs[t * R + k] = x;
            __syncthreads();
            if (act) {                                       // residuals: lane k takes series k, k + R, ...
                const double* zr = z + (size_t)t * N;
                for (int i = k; i < N; i += R) {
                    const double zv = zr[i];
                    if ((zv == zv) && good[i] != 0.0) {
                        double fit = 0.0;
#pragma unroll
                        for (int j = 0; j < R; ++j) fit = fma(lam[i * R + j], fs[t * R + j], fit);
                        const double e = zv - fit;
                        ssr_part = fma(e, e, ssr_part);
                    }
                }
            }
        }
        // SSR of the sweep: the same value in every thread
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ssr_part += __shfl_xor(ssr_part, off, kWave);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = ssr_part;
        __syncthreads();
        ssr = 0.0;
#pragma unroll
        for (int w = 0; w < kAlsThreads / 64; ++w) ssr += red[w];
        if (tid == 0 && a.ssr_path && it <= a.path_cap) a.ssr_path[(size_t)b * a.path_cap + it - 1] = ssr;
        if (!(fabs(ssr_old - ssr) >= thresh)) break;         // dfm_functions.ipynb:367-368
    }
    if (it > a.max_iter) it = a.max_iter;
    __syncthreads();

    // results: factors, loadings of the last sweep (NaN where the reference leaves them undefined)
    for (int e = tid; e < T * a.rmax; e += kAlsThreads) {
        const int t = e / a.rmax, j = e % a.rmax;
        Fg[e] = j < rb ? fs[t * R + j] : nan("");
    }
    for (int e = tid; e < N * a.rmax; e += kAlsThreads) {
        const int i = e / a.rmax, j = e % a.rmax;
        Lg[e] = (good[i] != 0.0 && j < rb) ? lam[i * R + j] : nan("");
    }
    if (tid == 0) {
        a.iters[b] = it;
        a.ssr[b] = ssr;
        if (a.ssr_path)
            for (int q = it; q < a.path_cap; ++q) a.ssr_path[(size_t)b * a.path_cap + q] = nan("");
    }
    // R2 of every included series on the final factors (dfm_functions.ipynb:372-380, compute_r2 :565-569)
    if (a.R2) {
        __syncthreads();
        lambda_step();                                       // lam <- OLS on the final factors
        for (int i = tid; i < N; i += kAlsThreads) {
            double out = nan("");
            if (good[i] != 0.0) {
                double s1 = 0.0; int cnt = 0;
                for (int t = 0; t < T; ++t) { const double zv = z[(size_t)t * N + i]; if (zv == zv) { s1 += zv; ++cnt; } }
                const double mean = s1 / cnt;
                double tss = 0.0, ee = 0.0;
                for (int t = 0; t < T; ++t) {
                    const double zv = z[(size_t)t * N + i];
                    if (zv == zv) {
                        double fit = 0.0;
#pragma unroll
                        for (int j = 0; j < R; ++j) fit = fma(lam[i * R + j], fs[t * R + j], fit);
                        const double e = zv - fit, d = zv - mean;
                        ee = fma(e, e, ee);
                        tss = fma(d, d, tss);
                    }
                }
                out = 1.0 - ee / tss;
            }
            a.R2[(size_t)b * N + i] = out;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Batched complete-case OLS: problem p regresses y_p (T) on X_p (T x K), dropping the rows where y or any
// regressor is NaN (dfm_functions.ipynb:242-252).  One lane group of R >= K lanes per problem.
template <int R>
__global__ __launch_bounds__(kAlsThreads) void ols_kernel(OlsArgs a) {
    constexpr int NG = kAlsThreads / R;
    __shared__ double Xall[NG * 2 * R];
    __shared__ double hball[NG * R];
    const int tid = threadIdx.x;
    const int grp = tid / R, k = tid % R;
    double* Xg = Xall + grp * 2 * R;
    double* hb = hball + grp * R;
    const int p = blockIdx.x * NG + grp;
    const bool act = p < a.P;
    const int pp = act ? p : a.P - 1;
    const int T = a.T, K = a.K;
    const double* __restrict__ X = a.X + (size_t)pp * a.x_stride;
    const double* __restrict__ y = a.y + (size_t)pp * a.y_stride;
    double Grow[R];
#pragma unroll
    for (int j = 0; j < R; ++j) Grow[j] = 0.0;
    double hk = 0.0;
    int cnt = 0;
    for (int t = 0; t < T; ++t) {
        const double yv = y[(size_t)t * a.y_inc];
        double xr[R];
        bool ok = yv == yv;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            xr[j] = j < K ? X[(size_t)t * K + j] : 0.0;
            ok = ok && (xr[j] == xr[j]);
        }
        if (ok) {
            const double xk = k < K ? X[(size_t)t * K + k] : 0.0;
            ++cnt;
            hk = fma(yv, xk, hk);
#pragma unroll
            for (int j = 0; j < R; ++j) Grow[j] = fma(xk, xr[j], Grow[j]);
        }
    }
    const bool solve = act && cnt >= a.nt_min && cnt >= K;
    if (!solve || k >= K) {
#pragma unroll
        for (int j = 0; j < R; ++j) Grow[j] = (j == k) ? 1.0 : 0.0;
        hk = 0.0;
    }
    const double dk = equilibrate_rows<R>(Grow, Xg, k);      // raw-unit regressors (a constant next to levels)
    const double bk = dk * group_solve<R>(Grow, hk * dk, Xg, hb, k);
    __syncthreads();
    hb[k] = bk;                                              // the coefficient vector, for the residual pass
    __syncthreads();
    if (act) {
        if (k < K) a.beta[(size_t)p * K + k] = solve ? bk : nan("");
        // residuals: lane k takes rows k, k + R, ...
        double ss = 0.0, sy = 0.0, syy = 0.0;
        for (int t = k; t < T; t += R) {
            const double yv = y[(size_t)t * a.y_inc];
            bool ok = yv == yv;
            double fit = 0.0;
            for (int j = 0; j < K; ++j) {
                const double xv = X[(size_t)t * K + j];
                ok = ok && (xv == xv);
                fit = fma(xv, hb[j], fit);
            }
            const double e = yv - fit;
            if (ok && solve) { ss = fma(e, e, ss); sy += yv; syy = fma(yv, yv, syy); }
            if (a.resid) a.resid[(size_t)p * T + t] = (ok && solve) ? e : nan("");
        }
#pragma unroll
        for (int off = 1; off < R; off <<= 1) {
            ss += __shfl_xor(ss, off, kWave);
            sy += __shfl_xor(sy, off, kWave);
            syy += __shfl_xor(syy, off, kWave);
        }
        if (k == 0) {
            a.ssr[p] = solve ? ss : nan("");
            a.nobs[p] = cnt;
            if (a.tss) a.tss[p] = solve ? syy - sy * sy / cnt : nan("");   // sum (y - ybar)^2 over the used rows
        }
    }
}

// ---------------------------------------------------------------------------------------------
// `standardize_data` (dfm_functions.ipynb:501-509) for B panels: per series, mean and POPULATION standard
// deviation over the observed cells (two passes, as the reference), z = (x - mean) / sd in place; NaN stays NaN.
// One workgroup per panel, thread = series (coalesced along i), HBM-bound: 3 reads + 1 write of the panel.
__global__ __launch_bounds__(256) void standardize_kernel(int T, int N, double* panel, double* mean_out, double* sd_out) {
    const int b = blockIdx.x;
    double* x = panel + (size_t)b * T * N;
    for (int i = threadIdx.x; i < N; i += 256) {
        double s = 0.0; int n = 0;
        for (int t = 0; t < T; ++t) { const double v = x[(size_t)t * N + i]; if (v == v) { s += v; ++n; } }
        const double mu = s / n;
        double ss = 0.0;
        for (int t = 0; t < T; ++t) { const double v = x[(size_t)t * N + i]; if (v == v) { const double d = v - mu; ss = fma(d, d, ss); } }
        const double sd = sqrt(ss / n);
        for (int t = 0; t < T; ++t) { const double v = x[(size_t)t * N + i]; x[(size_t)t * N + i] = (v - mu) / sd; }
        if (mean_out) mean_out[(size_t)b * N + i] = mu;
        if (sd_out) sd_out[(size_t)b * N + i] = sd;
    }
}
hipError_t launch_standardize(int B, int T, int N, double* panel, double* mean_out, double* sd_out, hipStream_t s) {
    hipLaunchKernelGGL(standardize_kernel, dim3(B), dim3(256), 0, s, T, N, panel, mean_out, sd_out);
    return hipGetLastError();
}

template <int R>
static hipError_t launch_als_r(const AlsArgs& a, hipStream_t s) {
    const size_t lds = AlsLds<R>::doubles(a.T, a.N) * sizeof(double);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static LdsOptIn attr_done;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&als_kernel<R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((als_kernel<R>), dim3(a.B), dim3(kAlsThreads), lds, s, a);
    return hipGetLastError();
}

bool als_fits(int Rpad, int T, int N) {
    size_t d = 0;
    switch (Rpad) {
        case 2: d = AlsLds<2>::doubles(T, N); break;
        case 4: d = AlsLds<4>::doubles(T, N); break;
        case 8: d = AlsLds<8>::doubles(T, N); break;
        case 16: d = AlsLds<16>::doubles(T, N); break;
        case 32: d = AlsLds<32>::doubles(T, N); break;
        default: return false;
    }
    return d * sizeof(double) <= 160 * 1024;
}

hipError_t launch_als(int Rpad, const AlsArgs& a, hipStream_t s) {
    switch (Rpad) {
        case 2: return launch_als_r<2>(a, s);
        case 4: return launch_als_r<4>(a, s);
        case 8: return launch_als_r<8>(a, s);
        case 16: return launch_als_r<16>(a, s);
        case 32: return launch_als_r<32>(a, s);
        default: return hipErrorInvalidValue;
    }
}

template <int R>
static hipError_t launch_ols_r(const OlsArgs& a, hipStream_t s) {
    constexpr int NG = kAlsThreads / R;
    hipLaunchKernelGGL((ols_kernel<R>), dim3((a.P + NG - 1) / NG), dim3(kAlsThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_ols(int Rpad, const OlsArgs& a, hipStream_t s) {
    switch (Rpad) {
        case 2: return launch_ols_r<2>(a, s);
        case 4: return launch_ols_r<4>(a, s);
        case 8: return launch_ols_r<8>(a, s);
        case 16: return launch_ols_r<16>(a, s);
        case 32: return launch_ols_r<32>(a, s);
        case 64: return launch_ols_r<64>(a, s);     // one wave per problem (8-factor VAR(4): 33 regressors)
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dfm
