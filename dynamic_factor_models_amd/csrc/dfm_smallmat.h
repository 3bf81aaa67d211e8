// dfm_smallmat.h -- r x r dense helpers for the "lane i owns row i" layout of the recursion kernels:
// a group of R lanes holds one matrix, rows other lanes need are exchanged through a per-group LDS
// slot (one wave per workgroup, so __syncthreads() is a wave-level fence).
#pragma once
#include "dfm_device.h"

namespace dfm {

constexpr double kSteadyTol = 4.5e-16;  // ~2 ulp: successive Om_f / P_s this close are "equal"

template <int R>
__device__ __forceinline__ void store_row(double* M, int i, const double (&row)[R]) {
#pragma unroll
    for (int j = 0; j < R; ++j) M[i * R + j] = row[j];
}
// out[j] = sum_k own[k] * M[k][j]
template <int R>
__device__ __forceinline__ void mm_rows(double (&out)[R], const double (&own)[R], const double* M) {
#pragma unroll
    for (int j = 0; j < R; ++j) out[j] = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const double a = own[k];
#pragma unroll
        for (int j = 0; j < R; ++j) out[j] = fma(a, M[k * R + j], out[j]);
    }
}
// out[j] = sum_k own[k] * M[j][k]
template <int R>
__device__ __forceinline__ void mm_rowsT(double (&out)[R], const double (&own)[R], const double* M) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) s = fma(own[k], M[j * R + k], s);
        out[j] = s;
    }
}
template <int R>
__device__ __forceinline__ double dot_vec(const double (&own)[R], const double* v) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k + 1 < R; k += 2) {
        s0 = fma(own[k], v[k], s0);
        s1 = fma(own[k + 1], v[k + 1], s1);
    }
    return s0 + s1;
}

// In-place Gauss-Jordan inverse (no pivoting; SPD input).  Lane i holds row i in m; sweep k broadcasts
// row k through the two R-double LDS rows at X (double-buffered: one barrier per sweep).
// Returns det(input) = product of pivots.  Every lane of the group gets the same value.
// Exchange fence for LDS traffic between the lanes of ONE wave (no s_barrier): a wave's DS operations are
// processed in issue order, so only the compiler has to be kept from moving loads above stores.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool WAVE>
__device__ __forceinline__ void group_sync() {
    if constexpr (WAVE) wave_lds_sync(); else __syncthreads();
}

// WAVE = true: the lane groups exchanging through X all live in one wave of a multi-wave workgroup
template <int R, bool WAVE = false>
__device__ __forceinline__ double gj_inverse(double (&m)[R], double* X, int i) {
    double det = 1.0;
    group_sync<WAVE>();
#pragma unroll
    for (int k = 0; k < R; ++k) {
        double* buf = X + (k & 1) * R;
        if (i == k) {
#pragma unroll
            for (int j = 0; j < R; ++j) buf[j] = m[j];
        }
        group_sync<WAVE>();
        double q[R];
#pragma unroll
        for (int j = 0; j < R; ++j) q[j] = buf[j];
        const double piv = q[k];
        const double d = 1.0 / piv;
        det *= piv;
        const double c = m[k];
        const bool me = (i == k);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (j == k) {
                m[j] = me ? d : -c * d;
            } else {
                const double qj = q[j] * d;
                m[j] = me ? qj : fma(-c, qj, m[j]);
            }
        }
    }
    return det;
}

// Symmetric diagonal scaling of a normal matrix held one row per lane: row <- d_k row d, d_k = G_kk^-1/2
// (G_kk > 0), exchanged through the group's R-double LDS row `dx`.  Returns d_k; the caller scales its
// right-hand side entry by d_k before, and its solution entry by d_k after, the solve.  Regressors on very
// different scales (a constant next to levels) otherwise cost the unpivoted elimination its accuracy.
template <int R, bool WAVE = false>
__device__ __forceinline__ double equilibrate_rows(double (&m)[R], double* dx, int i) {
    double dk = 1.0;
#pragma unroll
    for (int j = 0; j < R; ++j)
        if (j == i) dk = m[j] > 0.0 ? 1.0 / sqrt(m[j]) : 1.0;
    group_sync<WAVE>();
    dx[i] = dk;
    group_sync<WAVE>();
#pragma unroll
    for (int j = 0; j < R; ++j) m[j] *= dk * dx[j];
    return dk;
}

__device__ __forceinline__ bool close_enough(double a, double b) {
    return fabs(a - b) <= kSteadyTol * fabs(b);
}

}  // namespace dfm
