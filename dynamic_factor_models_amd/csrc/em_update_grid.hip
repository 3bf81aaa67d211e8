// em_update_grid.hip -- the transition half of the M-step on the balanced fast path for the wide states (Rp = 16, 32):
// em_update_kernel's contract (fastpath.hip) with an ELEMENT of every Rp x Rp matrix per thread.
//
//     S11 = sum_{t=1..T} E[f_t f_t' | X],  S10 = sum_{t=1..T} E[f_t f_{t-1}' | X],  S00 = sum_{t=0..T-1} E[f_t f_t' | X]
//     A = S10 S00^-1,  Q = sym(S11 - A S10') / T,  mu0 = f_0|T,  P0 = sym(P_0|T),  S11^-1 for the loadings step
//
// em_update_kernel gives a replicate lane groups of Rp lanes (lane = matrix row): at Rp = 32 its two Gauss-Jordan inversions
// and three products hold 5 rows of 32 doubles per lane -- 512 VGPRs and 357 spilled: 0.40 ms per EM iteration of config 4
// (0.20 ms at Rp = 16), the largest piece of the iteration after the two panel reads.  Here:
//   * sum_t f_t f_t' and sum_t f_t f_{t-1}' on the matrix pipe: the (Rp / 16)^2 tiles x 4 slices of the periods = one wave each
//     (v_mfma_f64_16x16x4 with A[i][k] = f_{t+k}[16 it + i], B[k][j] = f_{t+k}[16 jt + j] resp. f_{t+k-1}), slices summed in LDS;
//   * the inversions by the symmetric sweep operator with 2 x 2 block pivots and the products through LDS tiles of dfm_grid.h
//     (the element-per-thread algebra of the covariance kernels).
// Reference counterpart: none (the reference has no EM); oracle: oracle/kalman_oracle.py em_step.
#include <stdlib.h>

#include "dfm_kernels.h"
#include "dfm_smallmat.h"
#include "dfm_grid.h"

namespace dfm {

namespace {
typedef double eu_v4 __attribute__((ext_vector_type(4)));
}

template <int R>
__global__ __launch_bounds__(R * R) void em_update_grid_kernel(EmUpdArgs a) {
    constexpr int RR = R * R, TS = kTileStride<R>, RT = R * TS;
    constexpr int NW = RR / 64, NTILE = (R / 16) * (R / 16), NSL = NW / NTILE;   // waves, 16 x 16 tiles, period slices (4)
    extern __shared__ __attribute__((aligned(16))) double ews[];
    double* L0 = ews;
    double* L1 = L0 + RT;
    Grid<R> G;
    G.prow = L1 + RT;
    G.red = G.prow + kGridProw<R>;
    G.tt = G.red + 2 * (RR / 64) * R;
    double* part = G.tt + 2 * RT;                             // [2][NW][4][64]
    const int l = threadIdx.x, lane = l & 63, wave = l >> 6;
    const int i = l / R, j = l % R;
    G.l = l; G.i = i; G.j = j;
    const int b = blockIdx.x;
    const int T = a.T;
    const double* __restrict__ f = a.fsm + (size_t)b * T * R;
    const double* __restrict__ f0 = a.f0s + (size_t)b * R;
    const size_t o = (size_t)b * RR + (size_t)i * R + j;

    // ---- sum_t f_t f_t' and sum_t f_t f_{t-1}' (t = 0 .. T-1; f_{-1} = f_0|T): tile (it, jt) over the periods of slice sl
    {
        const int tile = wave % NTILE, sl = wave / NTILE;
        const int it = tile / (R / 16), jt = tile % (R / 16);
        const int k4 = lane >> 4, c16 = lane & 15;
        const int steps = (T + 3) / 4, sps = (steps + NSL - 1) / NSL;
        const int s_lo = sl * sps, s_hi = (s_lo + sps < steps) ? s_lo + sps : steps;
        eu_v4 a11 = {0.0, 0.0, 0.0, 0.0}, a10 = {0.0, 0.0, 0.0, 0.0};
        for (int s0 = s_lo; s0 < s_hi; s0 += 8) {
            double av[8], bv[8], pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = 4 * (s0 + u) + k4;
                const int tc = t < T ? t : T - 1;
                av[u] = f[(size_t)tc * R + 16 * it + c16];
                bv[u] = f[(size_t)tc * R + 16 * jt + c16];
                pv[u] = tc == 0 ? f0[16 * jt + c16] : f[(size_t)(tc - 1) * R + 16 * jt + c16];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = 4 * (s0 + u) + k4;
                const double x = (t < T && s0 + u < s_hi) ? av[u] : 0.0;
                a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, bv[u], a11, 0, 0, 0);
                a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, pv[u], a10, 0, 0, 0);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            part[((0 * NW + wave) * 4 + v) * 64 + lane] = a11[v];
            part[((1 * NW + wave) * 4 + v) * 64 + lane] = a10[v];
        }
    }
    __syncthreads();
    double M11 = 0.0, M10 = 0.0;
    {   // element (i, j): tile (i / 16, j / 16), D[(lane / 16) + 4 v][lane % 16]
        const int tile = (i >> 4) * (R / 16) + (j >> 4), ri = i & 15, cj = j & 15;
        const int src = 16 * (ri & 3) + cj, v = ri >> 2;
#pragma unroll
        for (int q = 0; q < NSL; ++q) {
            M11 += part[((0 * NW + q * NTILE + tile) * 4 + v) * 64 + src];
            M10 += part[((1 * NW + q * NTILE + tile) * 4 + v) * 64 + src];
        }
    }
    const double f0i = f0[i], f0j = f0[j];
    const double fTi = f[(size_t)(T - 1) * R + i], fTj = f[(size_t)(T - 1) * R + j];
    const double S11 = a.SP11[o] + M11;
    const double S10 = a.SU[o] + M10;
    const double P0s = a.P0s[o];
    const double S00 = S11 - fma(fTi, fTj, a.PT[o]) + fma(f0i, f0j, P0s);

    // EM bookkeeping (oracle/kalman_oracle.py em()): record ll_k; stop WITHOUT applying this M-step when the relative
    // improvement over ll_{k-1} is below tol
    bool em_apply = true;
    if (a.active) {
        const double ll = a.loglik[b];
        const bool was = a.k == 0 ? true : (a.active[b] != 0);
        bool go = was;
        if (was && a.k >= 1 && a.tol > 0.0) {
            const double llp = a.ll_path[(size_t)b * a.max_iter + a.k - 1];
            go = !((ll - llp) / (0.5 * (fabs(ll) + fabs(llp))) < a.tol);
        }
        em_apply = go;
        __syncthreads();                                     // every thread has read active / ll_path
        if (l == 0) {
            if (was) { a.ll_path[(size_t)b * a.max_iter + a.k] = ll; a.iters[b] = a.k + 1; }
            a.active[b] = go ? 1 : 0;
        }
    }

    // A = S10 S00^-1
    double inv = S00;
    (void)G.sweep_inverse(inv);
    L0[TS * i + j] = S10;
    L1[TS * i + j] = inv;                                     // symmetric: row j = column j
    __syncthreads();
    const double An = dot_rows<R>(L0, L1, i, j);
    __syncthreads();
    L1[TS * i + j] = An;
    __syncthreads();
    const double AS = dot_rows<R>(L1, L0, i, j);              // (A S10')[i][j] = sum_k A[i][k] S10[j][k]
    const double Qr = (S11 - AS) / (double)T;
    const double Qn = 0.5 * (Qr + G.transposed(Qr));
    const double P0n = 0.5 * (P0s + G.transposed(P0s));
    double inv11 = S11;
    (void)G.sweep_inverse(inv11);
    a.S11[o] = S11;
    a.S11inv[o] = inv11;
    if (em_apply) {
        a.A_out[o] = An;
        a.Q_out[o] = Qn;
        a.P0_out[o] = P0n;
        if (j == 0) a.mu0_out[(size_t)b * R + i] = f0i;
    }
}

namespace {
template <int R>
hipError_t launch_eug(const EmUpdArgs& a, hipStream_t s) {
    constexpr int RR = R * R, RT = R * kTileStride<R>, NW = RR / 64;
    const size_t lds = (size_t)(2 * RT + kGridProw<R> + 2 * (RR / 64) * R + 2 * RT + 2 * NW * 4 * 64) * sizeof(double);
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&em_update_grid_kernel<R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((em_update_grid_kernel<R>), dim3(a.B), dim3(RR), lds, s, a);
    return hipGetLastError();
}
}  // namespace

bool em_update_grid_supported(int Rpad) { return Rpad == 16 || Rpad == 32; }
hipError_t launch_em_update_grid(int Rpad, const EmUpdArgs& a, hipStream_t s) {
    note_kernel("em_update_grid_kernel");
    return Rpad == 32 ? launch_eug<32>(a, s) : Rpad == 16 ? launch_eug<16>(a, s) : hipErrorInvalidValue;
}

}  // namespace dfm
