// dfm_grid.h -- element-per-thread algebra on an R x R grid of threads (thread l = R i + j holds ELEMENT (i, j) of every
// R x R matrix): symmetric sweep-operator inverse with 2 x 2 block pivots, products through small LDS tiles, row / column
// reductions and transposes.  R = 8: one wave, everything in registers / the LDS crossbar.  R = 16 / 32: a workgroup, what
// crosses waves goes through small LDS buffers with one s_barrier per exchange.  Shared by recursion_wave.hip (sequential
// Kalman recursion, one replicate per wave / workgroup) and dfm_cov8.h (the covariance half of the balanced fast path with
// one WAVE per replicate).
#pragma once
#include "dfm_smallmat.h"

namespace dfm {
namespace {

__device__ __forceinline__ double uniform_lane(double v, int src) {   // src wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// 1 / x for a positive, normal x: v_rcp_f64 refined by two Newton steps (no scaling / fix-up: pivots of an SPD matrix)
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
// the same to one rounding less of latency: r (1 + e + e^2), e = 1 - x r (cubic: 2^-23 -> 2^-69) -- three dependent FMAs
__device__ __forceinline__ double fast_rcp3(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}
// running log of a product of positive numbers without a log per factor: mantissa product + exponent sum
struct LogProd {
    double m = 1.0;
    int e = 0;
    __device__ __forceinline__ void mul(double x) {
        m *= x;
        e += __builtin_amdgcn_frexp_exp(m);
        m = __builtin_amdgcn_frexp_mant(m);
    }
    __device__ __forceinline__ double log_value() const { return log(m) + (double)e * 0.69314718055994530942; }
};

// LDS tiles hold R x R matrices with rows R + 2 doubles apart: the 16 lanes of one pass of a ds_read_b128 then read 16
// different rows from 16 disjoint bank groups (at a stride of R doubles they collide 2-way at R = 8, 8-way at 16, 16-way
// at 32), and so do the transposed 8-byte stores.
template <int R>
constexpr int kTileStride = R + 2;
template <int R>
constexpr int kGridProw = 2 * (4 * R + 4);          // doubles behind Grid<R>::prow

// Cross-thread plumbing of one replicate's R x R element grid (thread l = R i + j).  R = 8: one wave, everything stays in
// registers / the LDS crossbar.  R = 16: four waves of a workgroup; what crosses waves goes through small LDS buffers
// with ONE s_barrier per exchange (buffers alternate, so the next exchange's writes cannot overtake this one's reads).
template <int R>
struct Grid {
    double* prow;   // [2][4 R + 4]  the two pivot rows of a block sweep, raw and scaled, + D^-1  (R >= 16): kGridProw<R> doubles
    double* red;    // [2][R][NW]  per-wave column partial sums    (R >= 16; NW = R R / 64 waves)
    double* tt;     // [2][R][R]   transposes                      (R = 16)
    int pr = 0, pt = 0;
    int l, i, j;

    __device__ __forceinline__ void sync() const {
        if constexpr (R == 8) wave_lds_sync(); else __syncthreads();
    }
    __device__ __forceinline__ bool all_true(bool v) const {
        if constexpr (R == 8) return __all(v); else return __syncthreads_and(v) != 0;
    }
    // over the R lanes of a row group (a DPP row or half-row): every one of them gets the total
    __device__ __forceinline__ double sum_j(double v) const {
        v += xor_lane<1>(v);
        v += xor_lane<2>(v);
        v += xor_lane<4>(v);
        if constexpr (R >= 16) v += xor_lane<8>(v);
        if constexpr (R == 32) {       // a row is two DPP rows
            double a = v, b = v;
            swap_rows16(a, b);
            v = a + b;
        }
        return v;
    }
    // over the R row groups: threads with the same j
    __device__ __forceinline__ double sum_i(double v) {
        if constexpr (R == 8) v += xor_lane<8>(v);
        if constexpr (R <= 16) {
            double a = v, b = v;
            swap_rows16(a, b);        // a = rows {0,0,2,2}, b = rows {1,1,3,3} of v
            v = a + b;
        }
        {
            double a = v, b = v;
            swap_halves32(a, b);      // a = low half twice, b = high half twice
            v = a + b;
        }
        if constexpr (R >= 16) {      // one partial per wave and column -> LDS, one barrier, everybody adds them up
            constexpr int NW = R * R / 64;
            double* rb = red + (pr ^= 1) * NW * R;
            if ((l & 63) < R) rb[j * NW + (l >> 6)] = v;
            __syncthreads();
            const double2* r2 = reinterpret_cast<const double2*>(rb + j * NW);
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int q = 0; q < NW / 2; ++q) { const double2 u = r2[q]; s0 += u.x; s1 += u.y; }
            v = s0 + s1;
        }
        return v;
    }
    __device__ __forceinline__ double transposed(double v) {
        if constexpr (R == 8) return __shfl(v, 8 * j + i, 64);
        else {
            double* tb = tt + (pt ^= 1) * R * kTileStride<R>;
            tb[kTileStride<R> * j + i] = v;
            __syncthreads();
            return tb[kTileStride<R> * i + j];
        }
    }
    // In-place inverse of a symmetric positive definite R x R matrix (element per thread) by the symmetric sweep
    // operator (Beaton) with 2 x 2 BLOCK pivots: the sweeps are one dependent chain of cross-thread exchanges, and a
    // block pivot {k, k+1} needs one exchange (both pivot rows at once) where two scalar pivots need two.  For rows /
    // columns outside the block  B_ij = A_ij - a_i' D^-1 a_j  (a_i = A[i][k], A[i][k+1]; D the pivot block), inside
    // B_Kj = D^-1 A_Kj and B_KK = -D^-1; after all blocks the register holds -M^-1.  Returns det M = product of det D.
    __device__ __forceinline__ double sweep_inverse(double& m) {
        double det = 1.0;
        const double negI = (i == j) ? -1.0 : 0.0;
        (void)negI;
#pragma unroll (R == 8 ? 4 : 1)       // R >= 16: a real loop (code size)
        for (int k = 0; k < R; k += 2) {
            double nm;
            if constexpr (R <= 16) {
                // B = A - a_i' adj(D) a_j / det D for every element: the exchange reads a COPY of the matrix whose pivot block is
                // -I, so a_K = -e_K arrives folded (with A_Kj = A_iK = 0 the same expression yields D^-1 A_Kj, (D^-1 A_Ki)' and
                // -D^-1) -- two selects per pivot instead of the fourteen that picked among five results, and the chain
                // pivots -> det -> reciprocal runs beside exchange -> a_i' adj(D) a_j: one FMA joins them.  A wave alone on its
                // SIMD is ISSUE-bound here (every wave64 instruction is 4+ cycles): 58 -> 38 instructions per pivot.
                const bool iK = (i >> 1) == (k >> 1), jK = (j >> 1) == (k >> 1);
                const double mc = (iK && jK) ? negI : m;
                double qj0, qj1, qi0, qi1, p00, p01, p11;
                if constexpr (R == 8) {
                    qj0 = __shfl(mc, 8 * k + j, 64);
                    qj1 = __shfl(mc, 8 * k + 8 + j, 64);
                    qi0 = __shfl(mc, 8 * k + i, 64);
                    qi1 = __shfl(mc, 8 * k + 8 + i, 64);
                    p00 = uniform_lane(m, 9 * k);
                    p01 = uniform_lane(m, 9 * k + 1);
                    p11 = uniform_lane(m, 9 * k + 9);
                } else {
                    // four waves: the two pivot rows (and the raw pivot block) through LDS, one barrier per pivot
                    double* pb = prow + ((k >> 1) & 1) * (4 * R + 4);
                    if (iK) {
                        pb[(i - k) * R + j] = mc;
                        if (jK) pb[2 * R + 2 * (i - k) + (j - k)] = m;
                    }
                    __syncthreads();
                    qj0 = pb[j]; qj1 = pb[R + j]; qi0 = pb[i]; qi1 = pb[R + i];
                    p00 = pb[2 * R]; p01 = pb[2 * R + 1]; p11 = pb[2 * R + 3];
                }
                const double dd = fma(p00, p11, -p01 * p01);
                const double rd = fast_rcp3(dd);
                det *= dd;
                const double m0 = (iK || jK) ? 0.0 : m;
                const double u0 = fma(p11, qj0, -p01 * qj1), u1 = fma(p00, qj1, -p01 * qj0);
                nm = fma(-fma(qi0, u0, qi1 * u1), rd, m0);
            } else {
                // R = 32 (16 waves).  The pivot rows k, k + 1 are R x 2 consecutive lanes of ONE wave: that wave alone inverts the pivot
                // block and scales its two rows (T = D^-1 A_K, lane exchanges only), publishes raw rows, scaled rows and D^-1,
                // and after the barrier every thread needs two FMAs and a few selects.  (Every thread used to redo the block
                // inverse and the scaling of its row's and column's pivot entries: ~50 instructions x 1024 threads per pivot --
                // the sweep was issue-bound at ~1260 cycles per pivot on a 16-wave workgroup, 56 % of a sequential step.)
                const bool ik0 = i == k, ik1 = i == k + 1, jk0 = j == k, jk1 = j == k + 1;
                double* pb = prow + ((k >> 1) & 1) * (4 * R + 4);       // raw0 | raw1 | t0 | t1 | e00 e01 e11 det
                if (ik0 || ik1) {
                    const int base = (k * R) & 63;                      // lane of element (k, 0) in this wave
                    const double p00 = __shfl(m, base + k, 64), p01 = __shfl(m, base + k + 1, 64), p11 = __shfl(m, base + R + k + 1, 64);
                    const double other = __shfl_xor(m, R, 64);          // the same column of the other pivot row
                    const double qj0 = ik0 ? m : other, qj1 = ik0 ? other : m;
                    const double dd = fma(p00, p11, -p01 * p01);
                    const double rd = fast_rcp(dd);
                    const double e00 = p11 * rd, e01 = -p01 * rd, e11 = p00 * rd;
                    const double tj0 = fma(e00, qj0, e01 * qj1), tj1 = fma(e01, qj0, e11 * qj1);
                    pb[(ik0 ? 0 : R) + j] = m;
                    pb[(ik0 ? 2 * R : 3 * R) + j] = ik0 ? tj0 : tj1;
                    if (ik0 && j == 0) { pb[4 * R] = e00; pb[4 * R + 1] = e01; pb[4 * R + 2] = e11; pb[4 * R + 3] = dd; }
                }
                __syncthreads();
                const double qi0 = pb[i], qi1 = pb[R + i];
                const double tj0 = pb[2 * R + j], tj1 = pb[3 * R + j];
                const double ti0 = pb[2 * R + i], ti1 = pb[3 * R + i];   // (the matrix stays symmetric through the sweeps)
                det *= pb[4 * R + 3];
                nm = m - fma(qi0, tj0, qi1 * tj1);
                nm = ik0 ? tj0 : nm;
                nm = ik1 ? tj1 : nm;
                nm = jk0 ? ti0 : nm;
                nm = jk1 ? ti1 : nm;
                if ((ik0 || ik1) && (jk0 || jk1)) {
                    const double e00 = pb[4 * R], e01 = pb[4 * R + 1], e11 = pb[4 * R + 2];
                    nm = -(ik0 ? (jk0 ? e00 : e01) : (jk0 ? e01 : e11));
                }
            }
            m = nm;
        }
        m = -m;
        return det;
    }
};
// sum_k X[i][k] Y[j][k], X and Y staged row-major in LDS
template <int R>
__device__ __forceinline__ double dot_rows(const double* xs, const double* ys, int i, int j) {
    const double2* a = reinterpret_cast<const double2*>(xs + kTileStride<R> * i);
    const double2* b = reinterpret_cast<const double2*>(ys + kTileStride<R> * j);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll (R >= 32 ? 4 : R / 2)   // R = 32: 64 operands in flight would not fit the 128-VGPR budget of a 16-wave workgroup
    for (int q = 0; q < R / 2; ++q) {
        const double2 u = a[q], v = b[q];
        s0 = fma(u.x, v.x, s0);
        s1 = fma(u.y, v.y, s1);
    }
    return s0 + s1;
}


}  // namespace
}  // namespace dfm
