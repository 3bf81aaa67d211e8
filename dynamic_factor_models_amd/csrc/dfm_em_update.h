// dfm_em_update.h -- the transition half of the M-step on the balanced fast path for Rp <= 8 as a DEVICE FUNCTION of one wave
// (em_update_kernel's algorithm, fastpath.hip: one lane group of R lanes per replicate row, 64 / R time slices per wave), so that
// it can run as extra workgroups at the front of the loadings step's streaming launch (mstep_mfma.hip) instead of as its own
// 33-us launch between the E-step and the second panel stream: nothing in it depends on the panel, and the streaming launch
// leaves plenty of idle issue slots.  Sufficient statistics, A = S10 S00^-1, Q = sym(S11 - A S10') / T, mu0, P0, S11^-1 and the
// per-replicate EM bookkeeping exactly as em_update_kernel / the epilogue of recursion_kernel (Shumway-Stoffer 1982).
#pragma once
#include "dfm_kernels.h"
#include "dfm_smallmat.h"

namespace dfm {

// X: this wave's LDS scratch, (64 / R) x (R R + 2 R) doubles.  valid = false: a tail wave (replicate index clamped by the caller)
// that takes part in nothing but the arithmetic.
template <int R>
__device__ __forceinline__ void em_update_wave(const EmUpdArgs& a, int b, bool valid, int lane, double* Xs) {
    constexpr int GPW = 64 / R;
    const int g = lane / R, i = lane % R;
    const bool live = valid && g == 0;
    double* X = Xs + g * (R * R + 2 * R);
    const int T = a.T;
    const size_t o = (size_t)b * R * R + (size_t)i * R;
    const double* __restrict__ f = a.fsm + (size_t)b * T * R;
    const double* __restrict__ f0 = a.f0s + (size_t)b * R;
    double M11[R], M10[R];
#pragma unroll
    for (int j = 0; j < R; ++j) { M11[j] = 0.0; M10[j] = 0.0; }
#pragma unroll 2
    for (int t = g; t < T; t += GPW) {
        double cur[R], prev[R];
        const double* pp = (t == 0) ? f0 : f + (size_t)(t - 1) * R;
#pragma unroll
        for (int j = 0; j < R; ++j) { cur[j] = f[(size_t)t * R + j]; prev[j] = pp[j]; }
        const double ci = f[(size_t)t * R + i];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            M11[j] = fma(ci, cur[j], M11[j]);
            M10[j] = fma(ci, prev[j], M10[j]);
        }
    }
#pragma unroll
    for (int off = R; off < 64; off <<= 1) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            M11[j] += __shfl_xor(M11[j], off, kWave);
            M10[j] += __shfl_xor(M10[j], off, kWave);
        }
    }
    const double f0i = f0[i];
    const double fTi = f[(size_t)(T - 1) * R + i];
    double S11[R], S10[R], S00[R], P0s[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const double fTj = f[(size_t)(T - 1) * R + j];
        S11[j] = a.SP11[o + j] + M11[j];
        S10[j] = a.SU[o + j] + M10[j];
        P0s[j] = a.P0s[o + j];
        S00[j] = S11[j] - fma(fTi, fTj, a.PT[o + j]) + fma(f0i, f0[j], P0s[j]);
    }
    bool em_apply = true;
    if (a.active) {                                          // EM bookkeeping (oracle/kalman_oracle.py em())
        const double ll = a.loglik[b];
        const bool was = a.k == 0 ? true : (a.active[b] != 0);
        bool go = was;
        if (was && a.k >= 1 && a.tol > 0.0) {
            const double llp = a.ll_path[(size_t)b * a.max_iter + a.k - 1];
            go = !((ll - llp) / (0.5 * (fabs(ll) + fabs(llp))) < a.tol);
        }
        em_apply = go;
        wave_lds_sync();                                     // (every lane has read active / ll_path)
        __builtin_amdgcn_s_waitcnt(0);
        if (live && i == 0) {
            if (was) { a.ll_path[(size_t)b * a.max_iter + a.k] = ll; a.iters[b] = a.k + 1; }
            a.active[b] = go ? 1 : 0;
        }
    }
    double inv[R], An[R], tmp[R], Qn[R], P0n[R];
#pragma unroll
    for (int j = 0; j < R; ++j) inv[j] = S00[j];
    (void)gj_inverse<R, true>(inv, X, i);
    wave_lds_sync();
    store_row<R>(X, i, inv);
    wave_lds_sync();
    mm_rows<R>(An, S10, X);                                  // A row i
    wave_lds_sync();
    store_row<R>(X, i, S10);
    wave_lds_sync();
    mm_rowsT<R>(tmp, An, X);                                 // (A S10')[i][:]
#pragma unroll
    for (int j = 0; j < R; ++j) Qn[j] = (S11[j] - tmp[j]) / (double)T;
    wave_lds_sync();
    store_row<R>(X, i, Qn);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < R; ++j) Qn[j] = 0.5 * (Qn[j] + X[j * R + i]);
    wave_lds_sync();
    store_row<R>(X, i, P0s);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < R; ++j) P0n[j] = 0.5 * (P0s[j] + X[j * R + i]);
#pragma unroll
    for (int j = 0; j < R; ++j) inv[j] = S11[j];
    (void)gj_inverse<R, true>(inv, X, i);
    if (live) {
#pragma unroll
        for (int j = 0; j < R; ++j) { a.S11[o + j] = S11[j]; a.S11inv[o + j] = inv[j]; }
        if (em_apply) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                a.A_out[o + j] = An[j];
                a.Q_out[o + j] = Qn[j];
                a.P0_out[o + j] = P0n[j];
            }
            a.mu0_out[(size_t)b * R + i] = f0i;
        }
    }
}

}  // namespace dfm
