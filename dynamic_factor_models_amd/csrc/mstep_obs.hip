// mstep_obs.hip -- the parametric model with OBSERVED factors (oracle/obs_oracle.py):
//
//     x_it = lam_o,i' g_t + lam_u,i' f_t + e_it,   e_it ~ N(0, R_i),     f_t = A f_{t-1} + eta_t
//
// with g_t (r_o columns) known regressors: the reference's `factor` matrix carries nfac_o observed columns in front of the
// nfac_u estimated ones (dfm_functions.ipynb:89-146; `lambda[:, nfac_o+1:end]` are the unobserved loadings, :364), but its own
// estimator is non-functional for nfac_o > 0 (:358-359, :371) and it has no parametric estimator (:21-23), so
// the semantics are this file's (documented in the oracle).  Per EM iteration (capi.hip: obs_em_run):
//   obs_residual_kernel   y = x - Lam_o g for the E-step (the ordinary smoother pass with Lam_u) and the padded Lam_u;
//   mstep_obs_kernel      one thread per series: lam_i = [sum_t E z z']^-1 sum_t x_it E z_t over its observed periods,
//                         z = (g, f), E z z' = z^ z^' + blockdiag(0, P_t); R_i = expected residual variance.  The moments of
//                         a period are the same for every series (wave-uniform addresses: scalar loads), the solve is an
//                         in-register Cholesky.  A niche path (a handful of observed factors, FAVAR): written for clarity,
//                         not tuned -- it reads the panel once more per iteration with a stride of N.
#include "dfm_grid.h"
#include "dfm_kernels.h"

namespace dfm {

namespace {

__global__ __launch_bounds__(256) void obs_residual_kernel(ObsArgs a, double* __restrict__ y, double* __restrict__ LamP) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int T = a.T, N = a.N, ro = a.ro, re = a.ro + a.ru, Rl = a.Rl;
    const size_t n = (size_t)a.B * T * N;
    if (tid < n) {
        const int i = (int)(tid % N);
        const size_t bt = tid / N;
        const size_t b = bt / T;
        const double* lam = a.Lam + (b * N + i) * re;
        const double* g = a.G + bt * ro;
        double v = a.panel[tid];
        for (int k = 0; k < ro; ++k) v = fma(-lam[k], g[k], v);
        y[tid] = v;
    }
    const size_t nl = (size_t)a.B * N * Rl;
    if (tid < nl) {
        const int c = (int)(tid % Rl);
        const size_t bn = tid / Rl;
        LamP[tid] = c < a.ru ? a.Lam[bn * re + ro + c] : 0.0;
    }
}

// in-place Cholesky solve of the leading n x n system M x = y (lower triangle of M used); static indexing only
template <int NMAX>
__device__ __forceinline__ bool chol_solve_obs(double (&M)[NMAX][NMAX], double (&y)[NMAX], int n) {
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

// RE = r_o + r_u (exact)
template <int RE>
__global__ __launch_bounds__(256) void mstep_obs_kernel(ObsArgs a) {
    const int b = blockIdx.y;
    const int i = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (a.active && a.active[b] == 0) return;                  // converged replicate: parameters stay
    const int T = a.T, N = a.N, ro = a.ro, Rl = a.Rl;
    const bool live = i < N;
    const int ii = live ? i : N - 1;
    const size_t npk = (size_t)Rl * (Rl + 1) / 2;
    const double* __restrict__ xb = a.panel + (size_t)b * T * N;
    const double* __restrict__ gb = a.G + (size_t)b * T * ro;
    const double* __restrict__ fb = a.fsm + (size_t)b * T * Rl;
    const double* __restrict__ Pb = a.Psm + (size_t)b * T * npk;
    double M[RE][RE], rhs[RE];
#pragma unroll
    for (int c = 0; c < RE; ++c) {
        rhs[c] = 0.0;
#pragma unroll
        for (int d = 0; d < RE; ++d) M[c][d] = 0.0;
    }
    double sxx = 0.0;
    int n = 0;
    for (int t = 0; t < T; ++t) {
        const double x = xb[(size_t)t * N + ii];
        const double* __restrict__ gt = gb + (size_t)t * ro;   // (wave-uniform addresses from here on)
        const double* __restrict__ ft = fb + (size_t)t * Rl;
        const double* __restrict__ Pt = Pb + (size_t)t * npk;
        double z[RE];
#pragma unroll
        for (int c = 0; c < RE; ++c) z[c] = c < ro ? gt[c < ro ? c : 0] : ft[c - ro];
        if (x == x) {
            ++n;
            sxx = fma(x, x, sxx);
#pragma unroll
            for (int c = 0; c < RE; ++c) {
                rhs[c] = fma(x, z[c], rhs[c]);
#pragma unroll
                for (int d = 0; d <= c; ++d) {
                    double s = z[c] * z[d];
                    if (c >= ro && d >= ro) s += Pt[(c - ro) * (c - ro + 1) / 2 + (d - ro)];   // Var(f_t | X), packed lower
                    M[c][d] += s;
                }
            }
        }
    }
    if (!live || n < RE + 1) return;                           // (as the oracle: too few cells for the joint regression)
    double lam[RE], Mc[RE][RE];
#pragma unroll
    for (int c = 0; c < RE; ++c) {
        lam[c] = rhs[c];
#pragma unroll
        for (int d = 0; d < RE; ++d) Mc[c][d] = d <= c ? M[c][d] : M[d][c];
    }
    if (!chol_solve_obs<RE>(Mc, lam, RE)) return;              // not positive definite: the series keeps its parameters
    double fit = 0.0;                                          // sum x^2 - 2 lam' rhs + lam' M lam
#pragma unroll
    for (int c = 0; c < RE; ++c) {
        double ml = 0.0;
#pragma unroll
        for (int d = 0; d < RE; ++d) ml = fma(d <= c ? M[c][d] : M[d][c], lam[d], ml);
        fit = fma(lam[c], ml - 2.0 * rhs[c], fit);
    }
#pragma unroll
    for (int c = 0; c < RE; ++c) a.Lam[((size_t)b * N + i) * RE + c] = lam[c];
    a.R[(size_t)b * N + i] = (sxx + fit) / (double)n;
}

template <int RE>
hipError_t launch_obs_re(const ObsArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((mstep_obs_kernel<RE>), dim3((a.N + 255) / 256, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- r_o + r_u = 9 .. 32: augmented moments for the ordinary loadings step (mstep.hip) -----------------------------------
// z_t = (g_t, f_t, 0 ..),  Var z_t = blockdiag(0, P_t, I): the identity on the padding keeps sum_t E z z' positive definite (T on
// its diagonal, as the padded state of an ordinary pass) and the padded loadings at exactly zero.
__global__ __launch_bounds__(256) void obs_augment_kernel(ObsArgs a, int Re, double* __restrict__ z, double* __restrict__ Vz,
                                                         double* __restrict__ LamAug) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int T = a.T, N = a.N, ro = a.ro, ru = a.ru, re = a.ro + a.ru, Rl = a.Rl;
    const int NPe = Re * (Re + 1) / 2;
    const size_t npk = (size_t)Rl * (Rl + 1) / 2;
    const size_t nz = (size_t)a.B * T * Re, nv = (size_t)a.B * T * NPe, nl = (size_t)a.B * N * Re;
    if (tid < nz) {
        const int c = (int)(tid % Re);
        const size_t bt = tid / Re;
        z[tid] = c < ro ? a.G[bt * ro + c] : c < re ? a.fsm[bt * Rl + (c - ro)] : 0.0;
    }
    if (tid < nv) {
        const int v = (int)(tid % NPe);
        const size_t bt = tid / NPe;
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= v) ++i;
        const int j = v - i * (i + 1) / 2;                    // j <= i
        double x = 0.0;
        if (i >= re) x = (i == j) ? 1.0 : 0.0;
        else if (j >= ro) x = a.Psm[bt * npk + (size_t)(i - ro) * (i - ro + 1) / 2 + (j - ro)];
        Vz[tid] = x;
    }
    if (tid < nl) {
        const int c = (int)(tid % Re);
        const size_t bn = tid / Re;
        LamAug[tid] = c < re ? a.Lam[bn * re + c] : 0.0;
    }
    (void)ru;
}

// sum_t E z_t z_t' and its inverse: one workgroup of R x R threads per replicate (element per thread, Grid<R>)
template <int R>
__global__ __launch_bounds__(R * R) void obs_moments_kernel(int T, const double* __restrict__ z, const double* __restrict__ Vz,
                                                            double* __restrict__ S11, double* __restrict__ S11inv) {
    constexpr int RR = R * R, NP = R * (R + 1) / 2, TS = kTileStride<R>, RT = R * TS;
    __shared__ __attribute__((aligned(16))) double wsm[kGridProw<R> + 2 * (RR / 64) * R + 2 * RT];
    Grid<R> G;
    G.prow = wsm;
    G.red = G.prow + kGridProw<R>;
    G.tt = G.red + 2 * (RR / 64) * R;
    const int l = threadIdx.x, i = l / R, j = l % R;
    G.l = l; G.i = i; G.j = j;
    const int b = blockIdx.x;
    const double* zb = z + (size_t)b * T * R;
    const double* vb = Vz + (size_t)b * T * NP;
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    const int pk = hi * (hi + 1) / 2 + lo;
    double s = 0.0;
    for (int t = 0; t < T; ++t) s += fma(zb[(size_t)t * R + hi], zb[(size_t)t * R + lo], vb[(size_t)t * NP + pk]);   // (symmetric by construction)
    S11[(size_t)b * RR + l] = s;
    double inv = s;
    (void)G.sweep_inverse(inv);
    S11inv[(size_t)b * RR + l] = inv;
}

}  // namespace

bool mstep_obs_supported(int ro, int ru) { return ro >= 1 && ru >= 1 && ro + ru <= 8; }
bool mstep_obs_wide_supported(int ro, int ru) { return ro >= 1 && ru >= 1 && ro + ru > 8 && ro + ru <= 32; }
int mstep_obs_wide_width(int ro, int ru) { return ro + ru <= 16 ? 16 : 32; }

hipError_t launch_obs_augment(const ObsArgs& a, int Re, double* z, double* Vz, double* LamAug, double* S11, double* S11inv, hipStream_t s) {
    note_kernel("obs_augment_kernel");
    const size_t NPe = (size_t)Re * (Re + 1) / 2;
    size_t n = (size_t)a.B * a.T * NPe;
    const size_t nl = (size_t)a.B * a.N * Re;
    if (nl > n) n = nl;
    hipLaunchKernelGGL(obs_augment_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, Re, z, Vz, LamAug);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (Re == 16) hipLaunchKernelGGL(obs_moments_kernel<16>, dim3(a.B), dim3(256), 0, s, a.T, z, Vz, S11, S11inv);
    else if (Re == 32) hipLaunchKernelGGL(obs_moments_kernel<32>, dim3(a.B), dim3(1024), 0, s, a.T, z, Vz, S11, S11inv);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_mstep_obs(const ObsArgs& a, hipStream_t s) {
    note_kernel("mstep_obs_kernel");
    switch (a.ro + a.ru) {
        case 2: return launch_obs_re<2>(a, s);
        case 3: return launch_obs_re<3>(a, s);
        case 4: return launch_obs_re<4>(a, s);
        case 5: return launch_obs_re<5>(a, s);
        case 6: return launch_obs_re<6>(a, s);
        case 7: return launch_obs_re<7>(a, s);
        case 8: return launch_obs_re<8>(a, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_obs_residual(const ObsArgs& a, double* y, double* LamP, hipStream_t s) {
    note_kernel("obs_residual_kernel");
    const size_t n1 = (size_t)a.B * a.T * a.N, n2 = (size_t)a.B * a.N * a.Rl;
    const size_t n = n1 > n2 ? n1 : n2;
    hipLaunchKernelGGL(obs_residual_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, y, LamP);
    return hipGetLastError();
}

}  // namespace dfm
