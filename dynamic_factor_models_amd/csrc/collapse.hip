// collapse.hip -- streaming "collapse" of the N-dimensional observation rows onto the r factors.
//
// For replicate b and period t (Jungbacker & Koopman 2015), over observed cells:
//     b_t = sum_i lam_i x_it / R_i   (r)        s_t = sum_i x_it^2 / R_i
//     n_t = #observed                           ld_t = sum_i log R_i          C_t = sum_i lam_i lam_i' / R_i
// This is the only kernel that touches the panel: one coalesced pass, HBM-bound (8 N T bytes per
// replicate, ~2 (r+1) flops per 8 bytes).  The reference's analogue is the per-period complete-case
// regression of x_t on Lambda (dfm_functions.ipynb:271-286 called from :364), which also first forms
// the normal equations Lambda_t' x_t over the observed series of period t.
//
// Mapping (wave64): one workgroup (4 waves) per replicate; lane l owns panel columns
// {2l, 2l+1} + 128 j (16-byte loads, a wave reads 1 KiB contiguous per instruction) and keeps
// W[c][k] = lam_ck / R_c for its columns in registers for the whole replicate; each wave takes row
// blocks of RB periods, accumulates RB (r+1) per-lane partial sums and folds them across the 64
// lanes with the transpose-reduce of dfm_device.h (~1 shuffle per value instead of 6).
// Rows with a NaN take a slow path that also emits n_t, ld_t and the packed C_t.
#include "dfm_gram.h"
#include "dfm_kernels.h"

namespace dfm {

// KR <= R (round 6): the columns k >= KR of the loadings are zero padding (a 20-wide companion state in the 32-wide layout: the AR
// idiosyncratic model) -- their sums are not formed (b_t: zeros; C_t: the packed entries of rows >= KR are zeros), which is 0.4 of the
// multiply-adds and, more to the point, of the registers: the 32-wide instantiation holds 256 VGPRs + 210 AGPRs at one wave per SIMD
template <int R, int CPL2, int RB, int KR = R>
__global__ __launch_bounds__(256) void collapse_kernel(CollapseArgs a) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int N = a.N, T = a.T;
    constexpr int NP = R * (R + 1) / 2;
    const double* __restrict__ X = a.panel + (size_t)b * T * N;
    const double* __restrict__ L = a.Lam + (size_t)b * N * R;
    const double* __restrict__ Rv = a.Rv + (size_t)b * N;

    double W[CPL2][2][KR];
    double Ri[CPL2][2];
    bool own[CPL2][2];
#pragma unroll
    for (int j = 0; j < CPL2; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 2 * lane + 128 * j + e;
            own[j][e] = c < N;
            const double ri = own[j][e] ? 1.0 / Rv[c] : 0.0;
            Ri[j][e] = ri;
#pragma unroll
            for (int k = 0; k < KR; ++k) W[j][e][k] = own[j][e] ? L[(size_t)c * R + k] * ri : 0.0;
        }

    // LDS tables for the rows with missing cells: loadings, 1/R, log R per series, the packed full-row C, its log det term
    extern __shared__ __attribute__((aligned(16))) double csm[];
    double* LamS = csm;                       // [N][R]
    double* rinvS = LamS + (size_t)N * R;     // [N]
    double* logRS = rinvS + N;                // [N]
    double* CfS = logRS + N;                  // [NP] packed lower triangle of C over all series, then ldfull
    for (int q = threadIdx.x; q < N * R; q += 256) LamS[q] = L[q];
    for (int q = threadIdx.x; q < N; q += 256) { const double rv = Rv[q]; rinvS[q] = 1.0 / rv; logRS[q] = log(rv); }
    __syncthreads();
    if (wave == 0) {  // per-replicate constants for the balanced rows
        if constexpr (KR == R) {
            c_all<R, CPL2, 0, true>(W, L, own, lane, a.Cfull + (size_t)b * R * R);
            c_all<R, CPL2, 0, false>(W, L, own, lane, CfS);
        } else {
            // (the real columns only, from the LDS tables: a lane per packed entry, a loop over the series -- once per replicate)
            double* cf = a.Cfull + (size_t)b * R * R;
            for (int v = lane; v < NP; v += 64) {
                int kk = 0;
                while ((kk + 1) * (kk + 2) / 2 <= v) ++kk;
                const int kp = v - kk * (kk + 1) / 2;
                double s = 0.0;
                if (kk < KR) {
                    for (int c = 0; c < N; ++c) s = fma(LamS[(size_t)c * R + kk] * rinvS[c], LamS[(size_t)c * R + kp], s);
                }
                CfS[v] = s;
                cf[kk * R + kp] = s;
                cf[kp * R + kk] = s;
            }
        }
        double ld = 0.0;
#pragma unroll
        for (int j = 0; j < CPL2; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (own[j][e]) ld += logRS[2 * lane + 128 * j + e];
        ld = wave_allsum(ld);
        if (lane == 0) { a.ldfull[b] = ld; CfS[NP] = ld; }
    }
    __syncthreads();
    // packed entries of C_t this lane produces on the slow path: v = lane + 64 q -> LDS offsets of lam_k, lam_k'
    constexpr int NPK = KR * (KR + 1) / 2;
    constexpr int EPL = (NPK + 63) / 64;
    int ek[EPL], ekp[EPL];
#pragma unroll
    for (int q = 0; q < EPL; ++q) {
        int v = lane + 64 * q;
        v = v < NPK ? v : NPK - 1;
        int k = 0;
        while ((k + 1) * (k + 2) / 2 <= v) ++k;
        ek[q] = k;
        ekp[q] = v - k * (k + 1) / 2;
    }

    constexpr int NV = RB * (KR + 1);
    bool canon;
    const int myidx = reduce_index<NV>(lane, canon);
    const int my_rr = myidx / (KR + 1), my_k = myidx % (KR + 1);
    const bool vec2 = (N & 1) == 0;
    const int nrb = (T + RB - 1) / RB;

    // the wave's NEXT row block is in flight while the current one is processed
    double xq[RB][CPL2][2];
    auto fetch = [&](int rb) {
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            int t = rb * RB + rr;
            t = t < T ? t : T - 1;
            const double* xr = X + (size_t)t * N;
#pragma unroll
            for (int j = 0; j < CPL2; ++j) {
                const int c0 = 2 * lane + 128 * j;
                double x0 = 0.0, x1 = 0.0;
                if (vec2) {
                    if (c0 < N) {
                        const double2 xx = *reinterpret_cast<const double2*>(xr + c0);
                        x0 = xx.x;
                        x1 = xx.y;
                    }
                } else {
                    if (c0 < N) x0 = xr[c0];
                    if (c0 + 1 < N) x1 = xr[c0 + 1];
                }
                xq[rr][j][0] = x0;
                xq[rr][j][1] = x1;
            }
        }
    };
    constexpr bool PREF = CPL2 <= 2;          // wider cross-sections have no registers to spare for a second block
    if (PREF && wave < nrb) fetch(wave);
    for (int rb = wave; rb < nrb; rb += 4) {
        const int t0 = rb * RB;
        if (!PREF) fetch(rb);
        double xc[RB][CPL2][2];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
            for (int j = 0; j < CPL2; ++j) { xc[rr][j][0] = xq[rr][j][0]; xc[rr][j][1] = xq[rr][j][1]; }
        if (PREF && rb + 4 < nrb) fetch(rb + 4);
        double acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = 0.0;
        bool nanrow[RB];
        unsigned nanbits[RB];                    // bit 2 j + e: this lane's cell (j, e) of the row is missing
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            bool anynan = false;
            unsigned nb = 0;
#pragma unroll
            for (int j = 0; j < CPL2; ++j) {
                double x0 = xc[rr][j][0], x1 = xc[rr][j][1];
                const bool n0 = x0 != x0, n1 = x1 != x1;
                anynan = anynan || n0 || n1;
                nb |= (n0 ? 1u : 0u) << (2 * j) | (n1 ? 2u : 0u) << (2 * j);
                x0 = n0 ? 0.0 : x0;
                x1 = n1 ? 0.0 : x1;
#pragma unroll
                for (int k = 0; k < KR; ++k) {
                    acc[rr * (KR + 1) + k] = fma(W[j][0][k], x0, acc[rr * (KR + 1) + k]);
                    acc[rr * (KR + 1) + k] = fma(W[j][1][k], x1, acc[rr * (KR + 1) + k]);
                }
                acc[rr * (KR + 1) + KR] = fma(x0 * Ri[j][0], x0, acc[rr * (KR + 1) + KR]);
                acc[rr * (KR + 1) + KR] = fma(x1 * Ri[j][1], x1, acc[rr * (KR + 1) + KR]);
            }
            nanrow[rr] = anynan;
            nanbits[rr] = nb;
        }
        wave_transpose_reduce<NV>(acc, lane);
        unsigned nanmask = 0;  // bit rr set: period t0+rr has a missing cell somewhere in the row
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) nanmask |= __any(nanrow[rr]) ? (1u << rr) : 0u;
        {
            const int t = t0 + my_rr;
            if (canon && t < T) {
                if (my_k < KR) a.bcol[((size_t)b * T + t) * R + my_k] = acc[0];
                else {
                    if constexpr (KR < R) {
#pragma unroll
                        for (int kk = KR; kk < R; ++kk) a.bcol[((size_t)b * T + t) * R + kk] = 0.0;
                    }
                    a.scol[(size_t)b * T + t] = acc[0];
                    if (((nanmask >> my_rr) & 1u) == 0) a.nobs[(size_t)b * T + t] = N;
                }
            }
        }
        // slow path: periods with missing cells also need n_t, ld_t, C_t
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int t = t0 + rr;
            if (t < T && ((nanmask >> rr) & 1u)) {
                if (a.Ct == nullptr) {   // caller promised a balanced panel: flag the error, keep the
                    if (lane == 0) {     // recursion on valid memory (results of this call are void)
                        atomicOr(a.status, 1);
                        a.nobs[(size_t)b * T + t] = N;
                    }
                    continue;
                }
                // C_t = C_full - sum over the MISSING series (or the plain sum over the observed ones when those
                // are fewer): a wave-uniform loop over the set bits of the per-slot ballots, every lane accumulating
                // its own packed entries from the LDS tables -- no per-lane Gram partials, no 64-lane reduction
                unsigned long long mm[CPL2][2];
                int nmiss = 0;
#pragma unroll
                for (int j = 0; j < CPL2; ++j)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        mm[j][e] = __ballot((nanbits[rr] >> (2 * j + e)) & 1u);
                        nmiss += __popcll(mm[j][e]);
                    }
                const bool comp = 2 * nmiss <= N;          // complement form
                if (!comp) {
#pragma unroll
                    for (int j = 0; j < CPL2; ++j)
#pragma unroll
                        for (int e = 0; e < 2; ++e) mm[j][e] = __ballot(own[j][e] && !((nanbits[rr] >> (2 * j + e)) & 1u));
                }
                double cacc[EPL], ldacc = 0.0;
#pragma unroll
                for (int q = 0; q < EPL; ++q) cacc[q] = 0.0;
#pragma unroll
                for (int j = 0; j < CPL2; ++j)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        unsigned long long bits = mm[j][e];
                        while (bits) {
                            const int l = __ffsll((long long)bits) - 1;
                            bits &= bits - 1;
                            const int c = 2 * l + 128 * j + e;
                            const double* lc = LamS + (size_t)c * R;
                            const double ri = rinvS[c];
                            ldacc += logRS[c];
#pragma unroll
                            for (int q = 0; q < EPL; ++q) cacc[q] = fma(lc[ek[q]] * ri, lc[ekp[q]], cacc[q]);
                        }
                    }
                if (lane == 0) {
                    a.nobs[(size_t)b * T + t] = N - nmiss;
                    a.ldrow[(size_t)b * T + t] = comp ? CfS[NP] - ldacc : ldacc;
                }
                double* ct = a.Ct + ((size_t)b * T + t) * NP;
#pragma unroll
                for (int q = 0; q < EPL; ++q) {
                    const int v = lane + 64 * q;
                    if (v < NPK) ct[v] = comp ? CfS[v] - cacc[q] : cacc[q];
                }
                if constexpr (KR < R) {
                    for (int v = NPK + lane; v < NP; v += 64) ct[v] = 0.0;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <int R, int CPL2, int RB, int KR = R>
static hipError_t launch_one(const CollapseArgs& a, hipStream_t s) {
    const size_t lds = ((size_t)a.N * R + 2 * (size_t)a.N + R * (R + 1) / 2 + 1) * sizeof(double);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&collapse_kernel<R, CPL2, RB, KR>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((collapse_kernel<R, CPL2, RB, KR>), dim3(a.B), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int R>
static hipError_t launch_r(const CollapseArgs& a, hipStream_t s) {
    // register budget: CPL2*2*R doubles of W + RB*(R+1) accumulators
    constexpr int RB = (R <= 4) ? 8 : (R <= 8) ? 4 : (R <= 16) ? 2 : 1;
    if constexpr (R == 32) {                                     // a narrower real state in the 32-wide layout (CollapseArgs::kreal)
        if (a.kreal > 0 && a.kreal <= 20) return a.N <= 128 ? launch_one<R, 1, RB, 20>(a, s) : a.N <= 256 ? launch_one<R, 2, RB, 20>(a, s) : hipErrorInvalidValue;
        if (a.kreal > 0 && a.kreal <= 24) return a.N <= 128 ? launch_one<R, 1, RB, 24>(a, s) : a.N <= 256 ? launch_one<R, 2, RB, 24>(a, s) : hipErrorInvalidValue;
    }
    if (a.N <= 128) return launch_one<R, 1, RB>(a, s);
    if (a.N <= 256) return launch_one<R, 2, RB>(a, s);
    if constexpr (R <= 16) {
        if (a.N <= 512) return launch_one<R, 4, RB>(a, s);
    }
    if constexpr (R <= 8) {
        if (a.N <= 1024) return launch_one<R, 8, RB>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_collapse(int Rpad, const CollapseArgs& a, hipStream_t s) {
    note_kernel("collapse_kernel");
    switch (Rpad) {
        case 2: return launch_r<2>(a, s);
        case 4: return launch_r<4>(a, s);
        case 8: return launch_r<8>(a, s);
        case 16: return launch_r<16>(a, s);
        case 32: return launch_r<32>(a, s);
        default: return hipErrorInvalidValue;
    }
}

int collapse_max_n(int Rpad) { return Rpad <= 8 ? 1024 : Rpad <= 16 ? 512 : 256; }

}  // namespace dfm
