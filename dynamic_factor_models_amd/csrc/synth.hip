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
constexpr int kSynTile = 32;                                 // periods per tile of the cell / standardisation kernels
enum : uint64_t { kStrLam = 1, kStrR = 2, kStrF = 3, kStrEps = 4, kStrMiss = 5 };

// Round 4: three launches instead of one workgroup per replicate.  The round-1 kernel gave a replicate ONE workgroup of 256
// threads (256 workgroups for config 4: one wave per SIMD), re-read the loadings from L2 for every cell and made three
// column-strided passes over the panel to standardise it: 85 GB fetched to write 10.8 GB at config 4, 44 ms.
// Now: (1) parameters and factor paths, one workgroup per replicate (tiny); (2) cells: a thread owns a column PAIR for a tile of
// 32 periods -- its two rows of loadings in registers, the tile's factors in LDS, one Philox counter per pair of cells as
// before -- writes the raw cells once and leaves per-tile column sums (sum x, sum x^2); (3) standardisation: every workgroup
// reduces the tiles' sums for its columns (mean, population s.d. = sqrt(sum x^2 / T - mean^2): the columns have mean ~ 0, so
// the one-pass form loses nothing), rescales its tile in place and punches the missing cells; the workgroups of tile 0 rescale
// the parameters.  HBM: panel written once, read once, written once.  Every number is still a pure function of
// (seed, replicate, stream, index): results do not depend on the launch geometry (oracle/synth_oracle.py restates them).

__global__ __launch_bounds__(kSynThreads) void synth_params_kernel(SynthArgs a) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int N = a.N, T = a.T, r = a.r;
    const uint64_t key = a.seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(a.first_replicate + b + 1));
    double* Lam = a.Lam + (size_t)b * N * r;
    double* Rv = a.R + (size_t)b * N;
    double* F = a.fscratch + (size_t)b * (T + 1) * r;        // f_1 .. f_T (row t-1) -- scratch
    for (int idx = tid; idx < (N * r + 1) / 2; idx += kSynThreads) {   // loadings (raw; rescaled by the standardisation kernel)
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

// x_ti = lam_i' f_t + sqrt(R_i) eps_ti for the tile's periods and the thread's two columns (two cells per counter: cell c and
// c + 1 share counter c / 2 when c is even)
__global__ __launch_bounds__(kSynThreads) void synth_cells_kernel(SynthArgs a, int ntile, int zoff) {
    __shared__ double Fs[kSynTile * 32];
    const int b = (int)blockIdx.z + zoff, tile = blockIdx.y;
    const int tid = threadIdx.x;
    const int N = a.N, T = a.T, r = a.r;
    const uint64_t key = a.seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(a.first_replicate + b + 1));
    const int t0 = tile * kSynTile, t1 = t0 + kSynTile < T ? t0 + kSynTile : T;
    const double* F = a.fscratch + (size_t)b * (T + 1) * r;
    for (int e = tid; e < (t1 - t0) * r; e += kSynThreads) Fs[e] = F[(size_t)t0 * r + e];
    __syncthreads();
    const int i0 = 2 * ((int)blockIdx.x * kSynThreads + tid);
    if (i0 >= N) return;
    const bool two = i0 + 1 < N;
    const double* Lam = a.Lam + (size_t)b * N * r;
    double l0[32], l1[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        l0[k] = k < r ? Lam[(size_t)i0 * r + k] : 0.0;
        l1[k] = (k < r && two) ? Lam[(size_t)(i0 + 1) * r + k] : 0.0;
    }
    const double s0 = sqrt(a.R[(size_t)b * N + i0]), s1 = two ? sqrt(a.R[(size_t)b * N + i0 + 1]) : 0.0;
    double* X = a.panel + (size_t)b * T * N;
    double sum0 = 0.0, sq0 = 0.0, sum1 = 0.0, sq1 = 0.0;
    for (int t = t0; t < t1; ++t) {
        const size_t c = (size_t)t * N + i0;
        double za, zb, e0, e1;
        normal2(key, kStrEps, c >> 1, za, zb);
        if ((c & 1) == 0) { e0 = za; e1 = zb; }
        else {                                                // (odd N, odd row: the pair straddles two counters)
            e0 = zb;
            double zc, zd;
            normal2(key, kStrEps, (c + 1) >> 1, zc, zd);
            e1 = zc;
        }
        const double* f = Fs + (t - t0) * r;
        double x0 = 0.0, x1 = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (k < r) { const double fk = f[k]; x0 = fma(l0[k], fk, x0); x1 = fma(l1[k], fk, x1); }
        }
        x0 += s0 * e0; x1 += s1 * e1;
        X[c] = x0;
        if (two) X[c + 1] = x1;
        sum0 += x0; sq0 = fma(x0, x0, sq0);
        sum1 += x1; sq1 = fma(x1, x1, sq1);
    }
    double* cs = a.colstats + ((size_t)b * ntile + tile) * 2 * N;
    cs[i0] = sum0; cs[N + i0] = sq0;
    if (two) { cs[i0 + 1] = sum1; cs[N + i0 + 1] = sq1; }
}

// standardise the tile's rows of the thread's two columns (mean, population s.d.), punch the missing cells; tile 0 also rescales
// the parameters to the standardised panel
__global__ __launch_bounds__(kSynThreads) void synth_standardize_kernel(SynthArgs a, int ntile, int zoff) {
    const int b = (int)blockIdx.z + zoff, tile = blockIdx.y;
    const int tid = threadIdx.x;
    const int N = a.N, T = a.T, r = a.r;
    const int i0 = 2 * ((int)blockIdx.x * kSynThreads + tid);
    if (i0 >= N) return;
    const bool two = i0 + 1 < N;
    const uint64_t key = a.seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(a.first_replicate + b + 1));
    const double* cs = a.colstats + (size_t)b * ntile * 2 * N;
    double sum0 = 0.0, sq0 = 0.0, sum1 = 0.0, sq1 = 0.0;
    for (int q = 0; q < ntile; ++q) {
        const double* c = cs + (size_t)q * 2 * N;
        sum0 += c[i0]; sq0 += c[N + i0];
        if (two) { sum1 += c[i0 + 1]; sq1 += c[N + i0 + 1]; }
    }
    const double m0 = sum0 / (double)T, m1 = sum1 / (double)T;
    const double inv0 = 1.0 / sqrt(sq0 / (double)T - m0 * m0);
    const double inv1 = two ? 1.0 / sqrt(sq1 / (double)T - m1 * m1) : 0.0;
    double* X = a.panel + (size_t)b * T * N;
    const int t0 = tile * kSynTile, t1 = t0 + kSynTile < T ? t0 + kSynTile : T;
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
    for (int t = t0; t < t1; ++t) {
        const size_t c = (size_t)t * N + i0;
        double x0 = (X[c] - m0) * inv0;
        if (a.missing_prob > 0.0 && uniform1(key, kStrMiss, c) < a.missing_prob) x0 = qnan;
        X[c] = x0;
        if (two) {
            double x1 = (X[c + 1] - m1) * inv1;
            if (a.missing_prob > 0.0 && uniform1(key, kStrMiss, c + 1) < a.missing_prob) x1 = qnan;
            X[c + 1] = x1;
        }
    }
    if (tile == 0) {
        double* Lam = a.Lam + (size_t)b * N * r;
        double* Rv = a.R + (size_t)b * N;
        for (int k = 0; k < r; ++k) {
            Lam[(size_t)i0 * r + k] *= inv0;
            if (two) Lam[(size_t)(i0 + 1) * r + k] *= inv1;
        }
        Rv[i0] *= inv0 * inv0;
        if (two) Rv[i0 + 1] *= inv1 * inv1;
    }
}

int synth_tiles(int T) { return (T + kSynTile - 1) / kSynTile; }

hipError_t launch_synth(const SynthArgs& a, hipStream_t s) {
    note_kernel("synth_cells_kernel");
    const int ntile = synth_tiles(a.T);
    const int npair = (a.N + 1) / 2;
    hipLaunchKernelGGL(synth_params_kernel, dim3(a.B), dim3(kSynThreads), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // the batch rides on gridDim.z (<= 65535): larger batches in slices, each launch checked
    for (int z0 = 0; z0 < a.B; z0 += 65535) {
        const int nz = a.B - z0 < 65535 ? a.B - z0 : 65535;
        const dim3 grid((npair + kSynThreads - 1) / kSynThreads, ntile, nz);
        hipLaunchKernelGGL(synth_cells_kernel, grid, dim3(kSynThreads), 0, s, a, ntile, z0);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    for (int z0 = 0; z0 < a.B; z0 += 65535) {
        const int nz = a.B - z0 < 65535 ? a.B - z0 : 65535;
        const dim3 grid((npair + kSynThreads - 1) / kSynThreads, ntile, nz);
        hipLaunchKernelGGL(synth_standardize_kernel, grid, dim3(kSynThreads), 0, s, a, ntile, z0);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace dfm
