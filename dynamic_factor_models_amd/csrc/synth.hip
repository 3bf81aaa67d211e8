// synth.hip -- synthetic replicates drawn on the device (SURVEY.md §8(d) DGP; the reference has no RNG
// and no Monte-Carlo loop -- this is the generator the benchmark / bootstrap harness needs so that
// B x T x N panels never cross PCIe):
//     lam_ij ~ N(0,1);  A = diag(linspace(.5,.9,r));  Q = I - A A';  R_i ~ U(.5,1.5);  f_0 ~ N(0,I)
//     f_t = A f_{t-1} + eta_t;  x_t = Lam f_t + sqrt(R) .* eps_t;  columns standardised exactly as
//     standardize_data (dfm_functions.ipynb:501-509: mean and population s.d.), parameters rescaled to the
//     standardised panel;  optionally iid missing cells (NaN) with probability missing_prob.
// Counter-based generator (Philox4x32-10, Salmon et al. 2011) keyed by (seed, replicate): every number is a
// pure function of (seed, replicate, stream, index), so results do not depend on the launch geometry.
#include "dfm_kernels.h"
#include "dfm_philox.h"

namespace dfm {

// uniform in (0,1) with 53 random bits
__device__ __forceinline__ double u01(uint32_t a, uint32_t b) {
    const uint64_t x = ((uint64_t)a << 32) | b;
    return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
// two independent N(0,1) per counter (Box-Muller on two 53-bit uniforms)
__device__ __forceinline__ void normal2(uint64_t key, uint64_t stream, uint64_t idx, double& z0, double& z1) {
    uint32_t o[4];
    Philox::block(key, idx, stream, o);
    const double u = u01(o[0], o[1]), v = u01(o[2], o[3]);
    const double rad = sqrt(-2.0 * log(u));
    double sn, cs;
    sincospi(2.0 * v, &sn, &cs);
    z0 = rad * cs; z1 = rad * sn;
}
__device__ __forceinline__ double uniform1(uint64_t key, uint64_t stream, uint64_t idx) {
    uint32_t o[4];
    Philox::block(key, idx, stream, o);
    return u01(o[0], o[1]);
}

constexpr int kSynThreads = 256;
enum : uint64_t { kStrLam = 1, kStrR = 2, kStrF = 3, kStrEps = 4, kStrMiss = 5 };

__global__ __launch_bounds__(kSynThreads) void synth_kernel(SynthArgs a) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int N = a.N, T = a.T, r = a.r;
    const uint64_t key = a.seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(a.first_replicate + b + 1));
    double* X = a.panel + (size_t)b * T * N;
    double* Lam = a.Lam + (size_t)b * N * r;
    double* Rv = a.R + (size_t)b * N;
    double* F = a.fscratch + (size_t)b * (T + 1) * r;        // f_1 .. f_T (row t-1) -- scratch

    // loadings and idiosyncratic variances
    for (int idx = tid; idx < (N * r + 1) / 2; idx += kSynThreads) {
        double z0, z1;
        normal2(key, kStrLam, idx, z0, z1);
        Lam[2 * idx] = z0;
        if (2 * idx + 1 < N * r) Lam[2 * idx + 1] = z1;
    }
    for (int i = tid; i < N; i += kSynThreads) Rv[i] = 0.5 + uniform1(key, kStrR, i);
    // factor paths: independent AR(1) per factor, unit unconditional variance
    if (tid < r) {
        const double ak = r > 1 ? 0.5 + 0.4 * (double)tid / (double)(r - 1) : 0.5;
        const double qk = sqrt(1.0 - ak * ak);
        double z0, z1;
        normal2(key, kStrF, (uint64_t)tid, z0, z1);
        double f = z0;                                       // f_0 ~ N(0,1)
        for (int t = 0; t < T; t += 2) {
            normal2(key, kStrF, (uint64_t)r + (uint64_t)(t / 2) * r + tid, z0, z1);
            f = ak * f + qk * z0;
            F[(size_t)t * r + tid] = f;
            if (t + 1 < T) {
                f = ak * f + qk * z1;
                F[(size_t)(t + 1) * r + tid] = f;
            }
        }
    }
    __syncthreads();
    // x_ti = lam_i' f_t + sqrt(R_i) eps_ti     (two cells per counter)
    const size_t ncell = (size_t)T * N;
    for (size_t p = tid; p < (ncell + 1) / 2; p += kSynThreads) {
        double z[2];
        normal2(key, kStrEps, p, z[0], z[1]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const size_t cell = 2 * p + e;
            if (cell < ncell) {
                const int t = (int)(cell / N), i = (int)(cell % N);
                double s = 0.0;
                for (int k = 0; k < r; ++k) s = fma(Lam[(size_t)i * r + k], F[(size_t)t * r + k], s);
                X[cell] = s + sqrt(Rv[i]) * z[e];
            }
        }
    }
    __syncthreads();
    // standardise columns (mean, population s.d.), rescale the parameters, punch the missing cells
    for (int i = tid; i < N; i += kSynThreads) {
        double m = 0.0;
        for (int t = 0; t < T; ++t) m += X[(size_t)t * N + i];
        m /= (double)T;
        double v = 0.0;
        for (int t = 0; t < T; ++t) { const double d = X[(size_t)t * N + i] - m; v = fma(d, d, v); }
        const double sd = sqrt(v / (double)T);
        const double inv = 1.0 / sd;
        for (int t = 0; t < T; ++t) {
            double x = (X[(size_t)t * N + i] - m) * inv;
            if (a.missing_prob > 0.0 && uniform1(key, kStrMiss, (uint64_t)t * N + i) < a.missing_prob)
                x = __longlong_as_double(0x7FF8000000000000ll);
            X[(size_t)t * N + i] = x;
        }
        for (int k = 0; k < r; ++k) Lam[(size_t)i * r + k] *= inv;
        Rv[i] *= inv * inv;
    }
    // transition parameters of the DGP
    for (int idx = tid; idx < r * r; idx += kSynThreads) {
        const int i = idx / r, j = idx % r;
        const double ak = r > 1 ? 0.5 + 0.4 * (double)i / (double)(r - 1) : 0.5;
        a.A[(size_t)b * r * r + idx] = (i == j) ? ak : 0.0;
        a.Q[(size_t)b * r * r + idx] = (i == j) ? 1.0 - ak * ak : 0.0;
        a.P0[(size_t)b * r * r + idx] = (i == j) ? 1.0 : 0.0;
    }
    for (int i = tid; i < r; i += kSynThreads) a.mu0[(size_t)b * r + i] = 0.0;
}

hipError_t launch_synth(const SynthArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(synth_kernel, dim3(a.B), dim3(kSynThreads), 0, s, a);
    return hipGetLastError();
}

}  // namespace dfm
