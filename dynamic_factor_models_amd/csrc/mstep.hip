// mstep.hip -- EM M-step for the observation equation: new loadings Lam and idiosyncratic
// variances R from the smoothed moments (Banbura & Modugno 2014 for missing
// cells).  Second (and last) streaming read of the panel in an EM iteration.
//
//   per series i over its observed periods T_i:
//     Sff_i = sum_t E[f_t f_t' | X]          (= S11 - sum over the periods where x_ti is missing)
//     Sxf_i = sum_t x_ti E[f_t | X],  Sxx_i = sum_t x_ti^2
//     lam_i = Sff_i^-1 Sxf_i,   R_i = (Sxx_i - 2 lam_i'Sxf_i + lam_i'Sff_i lam_i) / |T_i|
// The reference's counterpart of this step is the per-series OLS of x_i on the factors over the
// series' complete cases (dfm_functions.ipynb:355-362 and :391-404).
//
// Mapping: one workgroup (256 lanes) per replicate, LANE = SERIES (column i = tid + 256 j): a row of
// the panel is read by the 4 waves as 4 x 512 contiguous bytes, E[f_t] / Var[f_t] are wave-uniform
// loads, and every per-series sum lives in the registers of the lane that owns the series -- no
// cross-lane reduction, no atomics, bit-reproducible.  The (A, Q, mu0, P0) half of the M-step is the
// epilogue of recursion_kernel, which also leaves S11 and S11^-1 for this kernel.
#include "dfm_kernels.h"

namespace dfm {

// Solve the SPD system S x = rhs (S packed lower, row-major) by Cholesky.  R <= 8: fully unrolled
// (compile-time indices, everything in registers).  Larger R: plain loops over a private array
// (only series with missing cells take this path).
// Returns false when a pivot is not positive (S not positive definite to working precision: the caller keeps the series'
// parameters, as mstep_obs_kernel and mmw_finish_kernel do); x is then unspecified.
template <int R>
__device__ __forceinline__ bool chol_solve_packed(const double (&S)[R * (R + 1) / 2], const double (&rhs)[R],
                                                  double (&x)[R]) {
    double L[R * (R + 1) / 2];
    double y[R];
    bool ok = true;
    if constexpr (R <= 8) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                double s = S[i * (i + 1) / 2 + j];
#pragma unroll
                for (int k = 0; k < j; ++k) s = fma(-L[i * (i + 1) / 2 + k], L[j * (j + 1) / 2 + k], s);
                if (j == i) { ok = ok && (s > 0.0); L[i * (i + 1) / 2 + j] = sqrt(s > 0.0 ? s : 1.0); }
                else L[i * (i + 1) / 2 + j] = s / L[j * (j + 1) / 2 + j];
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            double s = rhs[i];
#pragma unroll
            for (int k = 0; k < i; ++k) s = fma(-L[i * (i + 1) / 2 + k], y[k], s);
            y[i] = s / L[i * (i + 1) / 2 + i];
        }
#pragma unroll
        for (int i = R - 1; i >= 0; --i) {
            double s = y[i];
#pragma unroll
            for (int k = i + 1; k < R; ++k) s = fma(-L[k * (k + 1) / 2 + i], x[k], s);
            x[i] = s / L[i * (i + 1) / 2 + i];
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < R; ++i) {
#pragma unroll 1
            for (int j = 0; j <= i; ++j) {
                double s = S[i * (i + 1) / 2 + j];
#pragma unroll 1
                for (int k = 0; k < j; ++k) s = fma(-L[i * (i + 1) / 2 + k], L[j * (j + 1) / 2 + k], s);
                if (j == i) { ok = ok && (s > 0.0); L[i * (i + 1) / 2 + j] = sqrt(s > 0.0 ? s : 1.0); }
                else L[i * (i + 1) / 2 + j] = s / L[j * (j + 1) / 2 + j];
            }
        }
#pragma unroll 1
        for (int i = 0; i < R; ++i) {
            double s = rhs[i];
#pragma unroll 1
            for (int k = 0; k < i; ++k) s = fma(-L[i * (i + 1) / 2 + k], y[k], s);
            y[i] = s / L[i * (i + 1) / 2 + i];
        }
#pragma unroll 1
        for (int i = R - 1; i >= 0; --i) {
            double s = y[i];
#pragma unroll 1
            for (int k = i + 1; k < R; ++k) s = fma(-L[k * (k + 1) / 2 + i], x[k], s);
            x[i] = s / L[i * (i + 1) / 2 + i];
        }
    }
    return ok;
}

// MX (round 6; R <= 8, N <= 256, register accumulators): the masked moment update D_i += m_it E_t -- 36 multiply-adds per lane and
// period for a 0 / 1 weight, three quarters of what the kernel issued (~123 instructions per period and wave, bound by its issue rate:
// 0.51 ms per 1024 replicates at the C2 shape) -- is a matrix product, D (series x entries) += M (series x periods) E (periods x
// entries), and goes to the matrix pipe, which the kernel left idle: per block of 4 periods a wave issues 4 x ET `v_mfma_f64_16x16x4`
// (its 64 series as 4 row tiles, the NP packed entries as ET = ceil(NP / 16) column tiles; 0 / 1 operands are exact in fp64, so the sums
// are the same numbers in another order); the NP mod 16 entries beyond the full tiles stay with the VALU.  The A operands (the lanes' own NaN tests, as bytes) and the B operands (E_t, already in
// the LDS tile) arrive by 4 + ET LDS reads per block.  The VALU keeps Sxf, Sxx and the counts.
typedef double mx_v4 __attribute__((ext_vector_type(4)));
template <int R, int CPL, bool REGD, bool MX = false>
__global__ __launch_bounds__(256, (MX ? 2 : 1)) void mstep_lam_kernel(MstepArgs a) {
    constexpr int NP = R * (R + 1) / 2;
    constexpr int ET = MX ? NP / 16 : 1;                      // FULL column tiles of 16 entries go to the matrix pipe; the NP % 16 entries left (R = 8: 32..35)
    constexpr int NVX = MX ? NP - 16 * ET : 0;                // stay on the VALU -- the fp64 matrix peak of gfx950 equals its vector peak: a quarter-full tile would cost the pipe a full one
    static_assert(!MX || (REGD && CPL == 1), "the matrix-pipe form keeps its sums in registers, one series per lane");
    const int b = blockIdx.x;
    if (a.active && !a.active[b]) return;
    const int tid = threadIdx.x;
    const int T = a.T, N = a.N, r = a.r;
    const double* __restrict__ X = a.panel + (size_t)b * T * N;
    const double* __restrict__ F = a.fsm + (size_t)b * T * R;
    const double* __restrict__ PS = a.Psm + (size_t)b * T * NP;

    double sxf[CPL][R], sxx[CPL];
    int ti[CPL];
    double dm[REGD ? CPL : 1][REGD ? NP : 1];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        sxx[j] = 0.0;
        ti[j] = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) sxf[j][k] = 0.0;
        if constexpr (REGD) {
#pragma unroll
            for (int v = 0; v < NP; ++v) dm[j][v] = 0.0;
        }
    }
    double* dmg = a.Dmiss ? a.Dmiss + (size_t)b * N * NP : nullptr;   // global accumulators (zeroed by the host)

    // Periods are taken in groups of UN.  The panel rows of group g+1 and its smoothed moments (E[f_t], Var[f_t]:
    // wave-uniform, R + NP doubles a period) are in flight while group g is processed: rows into registers, moments
    // through a double-buffered LDS tile that all four waves read back as broadcasts -- a period no longer waits for
    // its own (scalar) loads.
#ifndef DFM_MSTEP_UN8
#define DFM_MSTEP_UN8 8
#endif
    constexpr int UN = (R <= 8) ? DFM_MSTEP_UN8 : 4;          // (development A/B: -DDFM_MSTEP_UN8=n; 16 and 24 periods ahead are SLOWER -- 0.574 / 0.598 / 0.945 ms: the kernel is bound by its masked moment updates, not by bytes in flight)
    constexpr int PER = R + NP;                               // moments per period
    constexpr int NLD = (UN * PER + 255) / 256;               // loads per thread and group
    __shared__ double mom[2][UN * PER];
    __shared__ unsigned char mkS[MX ? UN * 256 : 1];          // MX: the group's NaN tests, [period][series]
    __shared__ double Dl[MX ? 64 * (NP + 1) : 1];             // MX: a wave's sums on their way back to the lanes that own the series
    static_assert(!MX || UN % 4 == 0, "blocks of 4 periods");
    mx_v4 macc[MX ? 4 : 1][MX ? ET : 1];
#pragma unroll
    for (int st = 0; st < (MX ? 4 : 1); ++st)
#pragma unroll
        for (int et = 0; et < (MX ? ET : 1); ++et) macc[st][et] = mx_v4{0.0, 0.0, 0.0, 0.0};
    const int mlane = tid & 63, mk4 = mlane >> 4, mc16 = mlane & 15, mwave = tid >> 6;
    const int ngroups = (T + UN - 1) / UN;
    double xn[UN][CPL], mn[NLD];
    auto issue = [&](int g) {
        const int t0 = g * UN;
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int t = (t0 + u < T) ? t0 + u : T - 1;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int col = tid + 256 * j;
                xn[u][j] = (col < N) ? X[(size_t)t * N + col] : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int e = tid + 256 * q;                      // [0, UN R): means, then covariances
            double v = 0.0;
            if (e < UN * R) {
                const size_t at = (size_t)t0 * R + e;
                if (at < (size_t)T * R) v = F[at];
            } else if (e < UN * PER) {
                const size_t at = (size_t)t0 * NP + (e - UN * R);
                if (at < (size_t)T * NP) v = PS[at];
            }
            mn[q] = v;
        }
    };
    issue(0);
    for (int g = 0; g < ngroups; ++g) {
        const int t0 = g * UN;
        double* mb = mom[g & 1];
        double xv[UN][CPL];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int j = 0; j < CPL; ++j) xv[u][j] = xn[u][j];
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int e = tid + 256 * q;
            if (e < UN * PER) mb[e] = mn[q];
        }
        // (workgroup barriers that order LDS traffic only: __syncthreads() also drains the vector-memory counter, i.e. it would
        // wait for the rows of the next group that are requested in between)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (g + 1 < ngroups) issue(g + 1);
        // E_t = f_t f_t' + P_t once per period, in place of P_t in the tile: it is the same for every series (every lane of every
        // wave used to recompute the 36 products for each of its missing cells -- half of the masked update's instructions)
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int e = tid + 256 * q - UN * R;
            if (e >= 0 && e < UN * NP) {
                const int u = e / NP, v = e - u * NP;
                int i = 0;
                while ((i + 1) * (i + 2) / 2 <= v) ++i;
                const int jj = v - i * (i + 1) / 2;
                mb[UN * R + e] = fma(mb[u * R + i], mb[u * R + jj], mb[UN * R + e]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if constexpr (MX) {
            // the masked update of the group's UN periods on the matrix pipe (see the head of the kernel); periods past the sample: 0
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const double x = xv[u][0];
                mkS[u * 256 + tid] = (t0 + u < T && tid < N && x != x) ? 1 : 0;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wave reads back its own columns: LDS serves a wave in order)
#pragma unroll
            for (int blk = 0; blk < UN / 4; ++blk) {
                const int u = 4 * blk + mk4;
                double av[4], bv[ET];
#pragma unroll
                for (int st = 0; st < 4; ++st) av[st] = (double)mkS[u * 256 + 64 * mwave + 16 * st + mc16];
#pragma unroll
                for (int et = 0; et < ET; ++et) {
                    const int e = 16 * et + mc16;
                    bv[et] = e < NP ? mb[UN * R + u * NP + e] : 0.0;
                }
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int et = 0; et < ET; ++et)
                        macc[st][et] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st], bv[et], macc[st][et], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int t = t0 + u;
            if (t >= T) break;
            double f[R];
#pragma unroll
            for (int k = 0; k < R; ++k) f[k] = mb[u * R + k];
            // E_t: REGD (accumulators in registers, R <= 8) reads the period's 36 entries in ONE batch in front of the update and
            // adds them by a multiply-add with the lane's 0 / 1 mask: no divergent region, no wait per LDS read
            double ef[(REGD && !MX) ? NP : 1];
            if constexpr (REGD && !MX) {
                const double* pv = mb + UN * R + u * NP;
#pragma unroll
                for (int v = 0; v < NP; ++v) ef[v] = pv[v];
            }
            bool miss_any = false;
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int col = tid + 256 * j;
                const double x = xv[u][j];
                const bool ok = (x == x);
                const double xz = ok ? x : 0.0;
                if (col < N) {
                    ti[j] += ok ? 1 : 0;
                    miss_any = miss_any || !ok;
                }
                sxx[j] = fma(xz, xz, sxx[j]);
#pragma unroll
                for (int k = 0; k < R; ++k) sxf[j][k] = fma(xz, f[k], sxf[j][k]);
            }
            if constexpr (MX) {
                if constexpr (NVX > 0) {                       // the entries beyond the full tiles: masked multiply-adds as before
                    const double x = xv[u][0];
                    const double mk = (tid < N && x != x) ? 1.0 : 0.0;
#pragma unroll
                    for (int v = 0; v < NVX; ++v) dm[0][16 * ET + v] = fma(mk, mb[UN * R + u * NP + 16 * ET + v], dm[0][16 * ET + v]);
                }
            } else if constexpr (REGD) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const int col = tid + 256 * j;
                    const double x = xv[u][j];
                    const double mk = (col < N && x != x) ? 1.0 : 0.0;
#pragma unroll
                    for (int v = 0; v < NP; ++v) dm[j][v] = fma(mk, ef[v], dm[j][v]);
                }
            } else if (__any(miss_any)) {   // wave-uniform: somebody in this wave lacks period t
                const double* pv = mb + UN * R + u * NP;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const int col = tid + 256 * j;
                    const double x = xv[u][j];
                    if (col < N && x != x) {
#pragma unroll
                        for (int v = 0; v < NP; ++v) dmg[(size_t)col * NP + v] += pv[v];   // E_t (see above)
                    }
                }
            }
        }
    }

    if constexpr (MX) {
        // the sums leave the accumulator layout -- lane (K, j), register v of tile (st, et) = series 16 st + K + 4 v of the wave, entry
        // 16 et + j -- for the lane that owns the series, one wave at a time through Dl
        for (int w = 0; w < 4; ++w) {
            if (mwave == w) {
#pragma unroll
                for (int st = 0; st < 4; ++st)
#pragma unroll
                    for (int et = 0; et < ET; ++et)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int e = 16 * et + mc16;
                            if (e < NP) Dl[(16 * st + mk4 + 4 * v) * (NP + 1) + e] = macc[st][et][v];
                        }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int v = 0; v < 16 * ET; ++v) dm[0][v] = Dl[mlane * (NP + 1) + v];
            }
            __syncthreads();
        }
    }

    // per-series solve
    const double* S11 = a.S11 + (size_t)b * R * R;
    const double* S11inv = a.S11inv + (size_t)b * R * R;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int col = tid + 256 * j;
        if (col >= N) continue;
        if (ti[j] < (a.min_cells > 1 ? a.min_cells : 1)) continue;   // a series without a single observed cell (or fewer than the caller's minimum) keeps its parameters (as mmw_finish_kernel, mstep_obs_kernel)
        double lam[R];
        double quad;   // lam' Sff lam
        if (ti[j] == T) {   // fully observed series: shared inverse
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < R; ++k) s = fma(S11inv[i * R + k], sxf[j][k], s);
                lam[i] = s;
            }
            quad = 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < R; ++k) s = fma(S11[i * R + k], lam[k], s);
                quad = fma(lam[i], s, quad);
            }
        } else {
            double Sff[NP];
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int jj = 0; jj <= i; ++jj) {
                    const int v = i * (i + 1) / 2 + jj;
                    double d;
                    if constexpr (REGD) d = dm[j][v];
                    else d = dmg[(size_t)col * NP + v];
                    Sff[v] = 0.5 * (S11[i * R + jj] + S11[jj * R + i]) - d;
                }
            if (!chol_solve_packed<R>(Sff, sxf[j], lam)) continue;   // not positive definite: the series keeps its parameters
            quad = 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const int hi = i > k ? i : k, lo = i > k ? k : i;
                    s = fma(Sff[hi * (hi + 1) / 2 + lo], lam[k], s);
                }
                quad = fma(lam[i], s, quad);
            }
        }
        double cross = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) cross = fma(lam[k], sxf[j][k], cross);
        const double Rn = (sxx[j] - 2.0 * cross + quad) / (double)ti[j];
        a.R_out[(size_t)b * N + col] = Rn;
        double* lo = a.Lam_out + ((size_t)b * N + col) * R;
#pragma unroll
        for (int k = 0; k < R; ++k) lo[k] = lam[k];
    }
    (void)r;
}

template <int R>
static hipError_t launch_m(const MstepArgs& a, hipStream_t s) {
    constexpr bool small = (R <= 8);
    if (a.N <= 256) {
        if (a.Dmiss) hipLaunchKernelGGL((mstep_lam_kernel<R, 1, false>), dim3(a.B), dim3(256), 0, s, a);
        else if constexpr (R == 8) hipLaunchKernelGGL((mstep_lam_kernel<R, 1, true, true>), dim3(a.B), dim3(256), 0, s, a);
        else if constexpr (small) hipLaunchKernelGGL((mstep_lam_kernel<R, 1, true>), dim3(a.B), dim3(256), 0, s, a);
        else return hipErrorInvalidValue;
    } else if (a.N <= 512) {
        if (!a.Dmiss) return hipErrorInvalidValue;
        hipLaunchKernelGGL((mstep_lam_kernel<R, 2, false>), dim3(a.B), dim3(256), 0, s, a);
    } else if (a.N <= 1024) {
        if (!a.Dmiss) return hipErrorInvalidValue;
        hipLaunchKernelGGL((mstep_lam_kernel<R, 4, false>), dim3(a.B), dim3(256), 0, s, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Dmiss (global per-series accumulators, zeroed by the caller) is required unless R <= 8 and N <= 256.
bool mstep_needs_dmiss(int Rpad, int N) { return !(Rpad <= 8 && N <= 256); }

hipError_t launch_mstep_lam(int Rpad, const MstepArgs& a, hipStream_t s) {
    switch (Rpad) {
        case 2: return launch_m<2>(a, s);
        case 4: return launch_m<4>(a, s);
        case 8: return launch_m<8>(a, s);
        case 16: return launch_m<16>(a, s);
        case 32: return launch_m<32>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dfm
