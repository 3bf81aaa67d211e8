// dfm_gram.h -- device helpers shared by the collapse kernels: per-lane column ownership
// (lane l owns panel columns {2l, 2l+1} + 128 j) and the wave-wide Gram reduction
// C = sum_c m_c lam_c lam_c' / R_c over the lanes' own columns.
#pragma once
#include "dfm_device.h"

namespace dfm {

__host__ __device__ constexpr int tri_row(int v) {
    int k = 0;
    while ((k + 1) * (k + 2) / 2 <= v) ++k;
    return k;
}

// Partial (own-columns) sums of the packed lower triangle entries [V0, V0+CNT) of
// sum_c m_c W[c][k] lam_c[k'], then wave reduction; canonical lanes store to out[V0 + idx].
template <int R, int CPL2, int V0, int CNT, bool FULL>
__device__ __forceinline__ void c_chunk(const double (&W)[CPL2][2][R], const double* __restrict__ L,
                                        const bool (&m)[CPL2][2], int lane, double* out) {
    double pc[CNT];
#pragma unroll
    for (int v = 0; v < CNT; ++v) pc[v] = 0.0;
#pragma unroll
    for (int j = 0; j < CPL2; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (m[j][e]) {
                const int c = 2 * lane + 128 * j + e;
                const double* lc = L + (size_t)c * R;
                double lam[R];
#pragma unroll
                for (int k = 0; k < R; ++k) lam[k] = lc[k];
#pragma unroll
                for (int v = 0; v < CNT; ++v) {
                    const int k = tri_row(V0 + v);
                    const int kp = V0 + v - k * (k + 1) / 2;
                    pc[v] = fma(W[j][e][k], lam[kp], pc[v]);
                }
            }
        }
    wave_transpose_reduce<CNT>(pc, lane);
    bool canon;
    const int idx = reduce_index<CNT>(lane, canon);
    if (canon) {
        const int v = V0 + idx;
        if constexpr (FULL) {  // full symmetric r x r output
            const int k = tri_row(v), kp = v - k * (k + 1) / 2;
            out[k * R + kp] = pc[0];
            out[kp * R + k] = pc[0];
        } else {
            out[v] = pc[0];
        }
    }
}

template <int R, int CPL2, int V0, bool FULL>
__device__ __forceinline__ void c_all(const double (&W)[CPL2][2][R], const double* __restrict__ L,
                                      const bool (&m)[CPL2][2], int lane, double* out) {
    constexpr int NP = R * (R + 1) / 2;
    if constexpr (V0 < NP) {
        constexpr int CNT = (NP - V0) < 64 ? (NP - V0) : 64;
        c_chunk<R, CPL2, V0, CNT, FULL>(W, L, m, lane, out);
        c_all<R, CPL2, V0 + CNT, FULL>(W, L, m, lane, out);
    }
}

__device__ __forceinline__ double wave_allsum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

}  // namespace dfm
