// dfm_scan.h -- building blocks of the time-parallel mean recursion of the balanced fast path (fastpath.hip header):
// r x r matrix-vector products inside a lane group with DPP row permutations (lane i holds M[i][i ^ s]), chunked runs of
// the linear recurrence v <- M v + u_t, and the Kogge-Stone scan of the chunk carries.  Shared by meanscan_kernel
// (fastpath.hip: operands in global memory) and pass_fused_kernel (pass_fused.hip: operands in LDS).
#pragma once
#include "dfm_cov.h"

namespace dfm {

// sum_s Mp[s] * x[lane ^ s]  with Mp[s] = M[i][i ^ s]
template <int R, int S>
struct MatVecX {
    static __device__ __forceinline__ void run(const double (&Mp)[R], double x, double& a0, double& a1) {
        if constexpr (S < R) {
            const double xs = xor_lane<S>(x);
            if constexpr ((S & 1) != 0) a1 = fma(Mp[S], xs, a1);
            else a0 = fma(Mp[S], xs, a0);
            MatVecX<R, S + 1>::run(Mp, x, a0, a1);
        }
    }
};
template <int R>
__device__ __forceinline__ double matvec_x(const double (&Mp)[R], double x, double add = 0.0) {
    double a0 = add, a1 = 0.0;
    MatVecX<R, 0>::run(Mp, x, a0, a1);
    return a0 + a1;
}
template <int R>
__device__ __forceinline__ void load_xperm(double (&Mp)[R], const double* M, int i) {
#pragma unroll
    for (int s = 0; s < R; ++s) Mp[s] = M[i * R + (i ^ s)];
}

constexpr int kPF = 8;    // chain steps per prefetch block
// transient covariance steps staged in LDS (later ones are read from global memory)
__host__ __device__ constexpr int ecap(int R) { return R <= 8 ? 8 : R <= 16 ? 4 : 2; }

// First prefetch block of a chunk: operands u_t, t = t0 + dir * j, j < kPF.
template <int R, bool FULL>
__device__ __forceinline__ void chunk_prefetch(double (&cur)[kPF], const double* src, int t0, int dir, int L, int tlo,
                                               int thi, int i) {
#pragma unroll
    for (int u = 0; u < kPF; ++u) {
        const int t = t0 + dir * u;
        if constexpr (FULL) {
            cur[u] = (u < L) ? src[(size_t)t * R + i] : 0.0;
        } else {   // branch-free: read a row that exists, select afterwards (no exec-masked block per operand)
            const bool ok = u < L && t >= tlo && t < thi;
            const int tc = t < tlo ? tlo : (t >= thi ? thi - 1 : t);
            const double x = src[(size_t)tc * R + i];
            cur[u] = ok ? x : 0.0;
        }
    }
}

// One chunk of a linear recurrence v <- M v + u_t over L steps (t = t0 + dir * j), operands u_t from
// `src` ([t][R] doubles; first block already in `cur`, later ones prefetched kPF steps ahead).
// FULL: every step of every lane group in this wave is valid (no predicates).  EMIT = 0: nothing;
// 1: forward phase 3 (also w_t = Z v -> wout, dot += v.w); 2: backward phase 3 (f of period t -> f_smooth
// row t-1).
template <int R, bool FULL, int EMIT>
__device__ __forceinline__ double chunk_run(const double (&Mp)[R], const double (&Zp)[R], double v, double (&cur)[kPF],
                                            const double* src, int t0, int dir, int L, int tlo, int thi, int i,
                                            double* wout, double& dot, double* fout, int r) {
    auto valid = [&](int j) { const int t = t0 + dir * j; return j < L && t >= tlo && t < thi; };
    double nxt[kPF];
    for (int j0 = 0; j0 < L; j0 += kPF) {
#pragma unroll
        for (int u = 0; u < kPF; ++u) {
            const int j = j0 + kPF + u;
            const int t = t0 + dir * j;
            if constexpr (FULL) {
                nxt[u] = (j < L) ? src[(size_t)t * R + i] : 0.0;
            } else {
                const int tc = t < tlo ? tlo : (t >= thi ? thi - 1 : t);
                const double x = src[(size_t)tc * R + i];
                nxt[u] = valid(j) ? x : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < kPF; ++u) {
            const int j = j0 + u;
            const int t = t0 + dir * j;
            if (j < L) {                                     // wave-uniform (L is a kernel argument)
                const bool ok = FULL ? true : valid(j);
                if constexpr (EMIT == 1) {
                    const double w = matvec_x<R>(Zp, v);
                    const double vn = matvec_x<R>(Mp, v, cur[u]);
                    if (ok) {
                        dot = fma(v, w, dot);
                        wout[(size_t)t * R + i] = w;
                    }
                    v = ok ? vn : v;
                } else {
                    const double vn = matvec_x<R>(Mp, v, cur[u]);
                    v = ok ? vn : v;
                    if constexpr (EMIT == 2) {
                        if (ok && t >= 1 && i < r) fout[(size_t)(t - 1) * r + i] = v;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kPF; ++u) cur[u] = nxt[u];
    }
    return v;
}

// Start state of every chunk from the chunk-local end states e_c (zero-state runs): an inclusive
// Kogge-Stone scan of the affine maps x -> M^L x + e_c over the NG chunks, level k using M^(L 2^k)
// (pw: the NLEV power matrices, in LDS).  `head` (the true state entering chunk 0) is folded into e_0.
// Returns the state entering chunk c.
// act = false: a thread outside the NG chunks (pass_fused_kernel's idle waves) -- it executes the barriers only.
struct BlockSync {                                            // the whole workgroup scans
    __device__ __forceinline__ void operator()() const { __syncthreads(); }
};
// `sync`: barrier of the threads that take part in the scan (pass_fused_kernel: four of its waves, on an LDS counter)
template <int R, int NG, class Sync = BlockSync>
__device__ __forceinline__ double carry_scan(double e, double head, const double* pw, int c, int i, double* sA,
                                             double* sB, bool act = true, Sync&& sync = Sync()) {
    constexpr int NLEV = scan_levels(R);
    double Mp[R];
    load_xperm<R>(Mp, pw, i);                                // M^L
    const double h = matvec_x<R>(Mp, head);
    double v = (c == 0) ? e + h : e;
    double* cur = sA;
    double* oth = sB;
    sync();                                                   // previous users of sA / sB are done
#pragma unroll 1
    for (int k = 0; k < NLEV; ++k) {
        if (act) cur[c * R + i] = v;
        sync();
        const int d = 1 << k;
        const double left = (act && c >= d) ? cur[(c - d) * R + i] : 0.0;
        if (k > 0) load_xperm<R>(Mp, pw + (size_t)k * R * R, i);
        v = matvec_x<R>(Mp, left, v);                         // lanes with c < d add M * 0
        double* t_ = cur; cur = oth; oth = t_;
    }
    if (act) cur[c * R + i] = v;
    sync();
    return (c == 0 || !act) ? head : cur[(c - 1) * R + i];
}

}  // namespace dfm
