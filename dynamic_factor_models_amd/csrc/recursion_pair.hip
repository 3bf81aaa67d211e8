// recursion_pair.hip -- recursion_wave_kernel<8> (information-form filter + "Z-smoother", one replicate per wave, lane
// l = 8 i + j = ELEMENT (i, j) of every 8 x 8 matrix) with the replicate's work split over TWO waves of one workgroup:
//
//   wave 0, the COVARIANCE wave: Z_t = (Om_f,t + Phi)^-1, J_t = Z_t K', Om_p,t+1 = Q^-1 - K J_t, the log determinants, the
//           table of distinct (Z_e, J_e), then P_t|T = Z_t + J_t P_t+1|T J_t' backwards, sum P, sum Cov(f_t+1, f_t);
//   wave 1, the MEAN wave: w_t = Z_t xi_t, xi_t+1 = K w_t + b_t, the quadratic forms and the per-period constants of the
//           likelihood, then f_t|T = w_t + J_t f_t+1|T backwards, sum f f', sum f_t+1 f_t'.
//
// Why: a wave alone on its SIMD is bound by what it ISSUES (every wave64 instruction occupies the 16-lane SIMD for four or
// more cycles) plus the latencies of its own chain of exchanges -- ~410 instructions and ~8 LDS / crossbar round trips per
// period at B = 1024, where every SIMD of the chip holds exactly one replicate.  The covariances never read the means, so
// the mean wave takes 110 of the 410 instructions off the chain and fills the covariance wave's waits on the same SIMDs.
// Forward the mean wave runs one chunk of 8 periods behind (Z_t through a double-buffered LDS ring, one s_barrier per
// chunk); backward the two are independent (both read the (Z_e, J_e) table) and meet once at the end.
// The four 8 x 8 products of a period (J = Z K', K J forward; P J', Z + J U backward) run on v_mfma_f64_4x4x4 with the
// matrices in its D layout -- operands by one ds_bpermute / a DPP row_ror:8, no LDS tiles (scripts/microbench/chainlat.hip:
// an LDS write -> read round trip costs a lone wave 122 cycles, a ds_bpermute pair 77, every f64 instruction ~6).
// Same inputs, scratch tables and outputs as recursion_wave_kernel<8, false>; results equal to rounding (sums are split).
// Reference counterpart: none (dfm_functions.ipynb:21-23 declares `Parametric` only); oracle: oracle/kalman_oracle.py.
#include <stdlib.h>
#include <type_traits>
#include "dfm_kernels.h"
#include "dfm_smallmat.h"
#include "dfm_grid.h"

namespace dfm {

namespace {
constexpr double kLog2PiP = 1.8378770664093454835606594728112;
constexpr int kPairChunk = 8;
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter, i.e. it would
// wait for the prefetch of the next chunk at every chunk
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
}  // namespace

__global__ __launch_bounds__(128, 2) void recursion_pair_kernel(RecursionArgs a) {
    constexpr int R = 8, RR = 64, CHW = kPairChunk;
    constexpr int TS = kTileStride<R>, RT = R * TS;
    extern __shared__ __attribute__((aligned(16))) double psm[];
    double* LK = psm;                // K = Q^-1 A rows (constant)
    double* L0 = LK + RT;
    double* L1 = L0 + RT;
    double* LJ = L1 + RT;            // J rows (backward sweep)
    double* ring = LJ + RT;          // [2][CHW][64]  Z_t of a chunk, covariance wave -> mean wave
    double* xch = ring + 2 * CHW * RR;   // [10][64]  exchanges: 0-1 prologue, 2-4 terminal, 5-9 final
    int* eidxS = reinterpret_cast<int*>(xch + 10 * RR);   // [T] table entry of forward step t
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // 0: covariances, 1: means
    const int i = lane >> 3, j = lane & 7;
    Grid<R> G;
    G.l = lane; G.i = i; G.j = j;
    const int T = a.T, N = a.N, r = a.r;
    const int b = blockIdx.x;
    if (a.only_if && a.only_if[b] == 0) return;              // (block-uniform) the replicate was done by recursion_chunk_kernel
    const bool diag = (i == j);

    const int Rc = a.Rc > 0 ? a.Rc : R;
    const int NPc = Rc * (Rc + 1) / 2;
    const bool inC = i < Rc && j < Rc;
    const double* bcol = a.bcol + (size_t)b * T * Rc;
    const double* scol = a.scol + (size_t)b * T;
    const int* nobs = a.nobs + (size_t)b * T;
    const double* ldrow = a.ldrow + (size_t)b * T;
    const double ldfull = a.ldfull[b];
    double* ZJ = a.ZJtab + (size_t)b * (T + 1) * 2 * RR;
    double* wtab = a.wtab + (size_t)b * T * R;
    const int pk = (i >= j) ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i;
    const int nchunks = (T + CHW - 1) / CHW;
    const int npr = r * (r + 1) / 2;
    const int rl = a.rl > 0 ? a.rl : R;
    const bool inL = i < rl && j < rl;
    const bool em = a.S11 != nullptr;

#ifdef DFM_PAIR_PROF
    const unsigned long long pt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long pt1 = 0, pt2 = 0;
#define PSTAMP(v) v = __builtin_amdgcn_s_memrealtime()
#else
#define PSTAMP(v) do {} while (0)
#endif
    if (role == 0) {
        // =========================================== covariance wave ==============================================
        const double Ael = a.A[(size_t)b * RR + lane];
        double Qi = a.Q[(size_t)b * RR + lane];
        const double Cf = inC ? a.Cfull[(size_t)b * Rc * Rc + i * Rc + j] : 0.0;
        double Omf = a.P0[(size_t)b * RR + lane];
        const double mu0c = a.mu0[(size_t)b * R + j];
        const double detQ = G.sweep_inverse(Qi);
        const double detP0 = G.sweep_inverse(Omf);                 // Om_f,0 = P0^-1
        L0[TS * i + j] = Qi;
        L1[TS * j + i] = Ael;                                      // A'
        G.sync();
        const double K = dot_rows<R>(L0, L1, i, j);                // K = Qi A
        G.sync();
        LK[TS * i + j] = K;
        L0[TS * j + i] = K;
        G.sync();
        const double KT = L0[TS * i + j];
        const double Phi = dot_rows<R>(L0, L1, i, j);              // Phi = K' A
        G.sync();
        const double xi0r = G.sum_j(Omf * mu0c);                   // xi_0 = P0^-1 mu0, row-distributed
        const double q0_part = diag ? mu0c * xi0r : 0.0;
        xch[0 * RR + lane] = KT;
        xch[1 * RR + lane] = G.transposed(xi0r);                   // column-distributed
        __syncthreads();                                           // (P) constants for the mean wave

        // D layout of v_mfma_f64_4x4x4 (described at the backward sweep)
        const int dI = (lane >> 3) & 1, dJ = (lane >> 2) & 1, lo2 = lane & 3, hi2 = lane >> 4;
        const int di = 4 * dI + hi2, dj = lane & 7;
        const int eD = 8 * di + dj;                                // row-major index of this lane's element
        const int srcA0 = hi2 | (dI << 3) | (lo2 << 4), srcA1 = srcA0 | 4;   // lanes of X[4 dI + lo2][4 K + hi2], K = 0, 1
        const int srcT0 = hi2 | (dJ << 3) | (lo2 << 4), srcT1 = srcT0 | 4;   // lanes of X[4 dJ + lo2][4 K + hi2]: B operand of X'
        const bool hiHalf = (lane & 8) != 0;
        auto ror8 = [&](double v) {                               // value of lane l ^ 8
            int lo = __double2loint(v), hi = __double2hiint(v);
            lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xF, 0xF, false);
            hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xF, 0xF, false);
            return __hiloint2double(hi, lo);
        };
        // the two products of a step on the matrix pipe (see the backward sweep for the D layout): operands of the constant K are
        // taken once; Z (element layout 8 i + j) gives its A operand by one ds_bpermute, K J comes back by another
        const int srcZ0 = 8 * (4 * dI + lo2) + hi2, srcZ1 = srcZ0 + 4;        // lanes (8 i + j layout) of X[4 dI + lo2][4 K + hi2]
        const double kB0 = __shfl(K, 8 * (4 * dJ + lo2) + hi2, 64), kB1 = __shfl(K, 8 * (4 * dJ + lo2) + 4 + hi2, 64);   // B operand of K'
        const double kA0 = __shfl(K, srcZ0, 64), kA1 = __shfl(K, srcZ1, 64);  // A operand of K
        const int lDij = j | ((i >> 2) << 3) | ((i & 3) << 4);                // D-layout lane of element (i, j)
        double Jd = 0.0;                                                      // J_t in the D layout (stored at its row-major index)
        auto products = [&](double Zel) {                                     // J = Z K' (kept in Jd); returns K J in the 8 i + j layout
            const double zA0 = __shfl(Zel, srcZ0, 64), zA1 = __shfl(Zel, srcZ1, 64);
            double jd = __builtin_amdgcn_mfma_f64_4x4x4f64(zA0, kB0, 0.0, 0, 0, 0);
            jd = __builtin_amdgcn_mfma_f64_4x4x4f64(zA1, kB1, jd, 0, 0, 0);
            Jd = jd;
            const double jx = ror8(jd);
            const double jB0 = hiHalf ? jx : jd, jB1 = hiHalf ? jd : jx;
            double kj = __builtin_amdgcn_mfma_f64_4x4x4f64(kA0, jB0, 0.0, 0, 0, 0);
            kj = __builtin_amdgcn_mfma_f64_4x4x4f64(kA1, jB1, kj, 0, 0, 0);
            return __shfl(kj, lDij, 64);
        };

        // ---- forward
        double cc[CHW], nc_[CHW];
        int cn[CHW], nn_[CHW];
        auto issue_fwd = [&](int c) {
#pragma unroll
            for (int s = 0; s < CHW; ++s) {
                int t = c * CHW + s;
                t = t < T ? t : T - 1;
                nn_[s] = nobs[t];
                nc_[s] = (a.Ct && inC) ? a.Ct[((size_t)b * T + t) * NPc + pk] : 0.0;
            }
        };
        double Z = 0.0, Jr = 0.0, Omp = 0.0, detM_cur = 1.0;
        LogProd detprod;
        int e = -1;
        bool need_cov = true;
        double zb[CHW], jb[CHW];
        int eb[CHW];
#pragma unroll
        for (int s = 0; s < CHW; ++s) { zb[s] = 0.0; jb[s] = 0.0; eb[s] = -1; }
        auto flush_fwd = [&]() {
#pragma unroll
            for (int s = 0; s < CHW; ++s) {
                if (eb[s] >= 0) {
                    ZJ[((size_t)eb[s] * 2 + 0) * RR + lane] = zb[s];
                    ZJ[((size_t)eb[s] * 2 + 1) * RR + eD] = jb[s];   // (J_e is held in the D layout: lane -> element eD)
                }
                eb[s] = -1;
            }
        };
        issue_fwd(0);
        for (int c = 0; c < nchunks; ++c) {
#pragma unroll
            for (int s = 0; s < CHW; ++s) { cn[s] = nn_[s]; cc[s] = nc_[s]; }
            if (c > 0) flush_fwd();
            if (c + 1 < nchunks) issue_fwd(c + 1);
            double* rb = ring + (c & 1) * CHW * RR;
            const int smax = (T - c * CHW) < CHW ? (T - c * CHW) : CHW;
            // DENSE: every period of the chunk has a missing cell, so every step is a new covariance step whatever the
            // steady-state test says -- the chunk runs without that test (a compare, a ballot and a branch per step on the
            // chain) and as ONE basic block, which lets the scheduler put a step's tail under the next step's first exchange
            double qcp[CHW];
            auto step = [&](auto dense_tag, int s) {
                constexpr bool DENSE = decltype(dense_tag)::value;
                const int t = c * CHW + s;
                if constexpr (DENSE) {
                    // Z_t = (Om_p,t + C_t + Phi)^-1 with  Om_p,t+1 + C_t+1 + Phi = (Qi + C_t+1 + Phi) - K J_t : the constant part is
                    // summed beside the chain (qcp), the chain sees one subtraction between the product and the next sweep
                    detM_cur = G.sweep_inverse(Z);
                    const double kj = products(Z);                 // J = Z K' (Jd), K J
                    ++e;
                    zb[s] = Z; jb[s] = Jd; eb[s] = e;
                    rb[s * RR + lane] = Z;
                    detprod.mul(detM_cur);
                    if (s + 1 < CHW) Z = qcp[s + 1] - kj;
                    else { Omp = Qi - kj; Omf = Omp + cc[s]; }
                    return;
                }
                const bool computed = need_cov;
                const double Omf_used = Omf;
                if (computed) {  // wave-uniform
                    Z = Omf + Phi;
                    detM_cur = G.sweep_inverse(Z);
                    Omp = Qi - products(Z);                        // J = Z K' (Jd), Om_p = Qi - K J
                    ++e;
                    zb[s] = Z; jb[s] = Jd; eb[s] = e;
                }
                rb[s * RR + lane] = Z;
                if (lane == 0) eidxS[t] = e;
                detprod.mul(detM_cur);
                const bool full = (cn[s] == N);
                const double Omf_new = Omp + (full ? Cf : cc[s]);
                if (computed) {
                    const bool same = full && close_enough(Omf_new, Omf_used);
                    need_cov = !G.all_true(same);
                } else {
                    need_cov = !full;
                }
                Omf = Omf_new;
            };
            bool dense = smax == CHW;
#pragma unroll
            for (int s = 0; s < CHW; ++s) dense = dense && (cn[s] != N);
            if (dense && need_cov) {
#pragma unroll
                for (int s = 0; s < CHW; ++s) qcp[s] = Qi + (cc[s > 0 ? s - 1 : 0] + Phi);   // qcp[s]: for the step after s - 1
                if (lane < CHW) eidxS[c * CHW + lane] = e + 1 + lane;
                Z = Omf + Phi;
#pragma unroll
                for (int s = 0; s < CHW; ++s) step(std::true_type{}, s);
            } else {
#pragma unroll
                for (int s = 0; s < CHW; ++s)
                    if (s < smax) step(std::false_type{}, s);
            }
            lds_barrier();                                         // (F_c) chunk c of the ring is complete
        }
        flush_fwd();
        PSTAMP(pt1);

        // ---- terminal
        double Ps = Omf;
        const double detOmT = G.sweep_inverse(Ps);                 // P_T
        xch[2 * RR + lane] = Ps;
        __syncthreads();                                           // (T1) P_T out, xi_T and the mean wave's sums in
        const double xi = xch[3 * RR + lane];
        const double fs_r = G.sum_j(Ps * xi);                      // f_T, row-distributed
        bool em_apply = true;
        {
            const double part = diag ? q0_part - xi * fs_r : 0.0;
            const double qd = G.sum_i(G.sum_j(part)) - xch[4 * RR + 0];
            const double ssum = xch[4 * RR + 1], nsum = xch[4 * RR + 2], ldsum = xch[4 * RR + 3];
            const double LD = log(detOmT) + log(detP0) + (double)T * log(detQ) + detprod.log_value();
            const double ll = -0.5 * (nsum * kLog2PiP + ldsum + LD + ssum + qd);
            if (lane == 0) {
                a.loglik[b] = ll;
                if (a.ncov) a.ncov[b] = e + 1;
            }
            if (a.active) {   // EM bookkeeping, as recursion_kernel
                const bool was = a.k == 0 ? true : (a.active[b] != 0);
                bool go = was;
                if (was && a.k >= 1 && a.tol > 0.0) {
                    const double llp = a.ll_path[(size_t)b * a.max_iter + a.k - 1];
                    go = !((ll - llp) / (0.5 * (fabs(ll) + fabs(llp))) < a.tol);
                }
                em_apply = go;
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    if (was) { a.ll_path[(size_t)b * a.max_iter + a.k] = ll; a.iters[b] = a.k + 1; }
                    a.active[b] = go ? 1 : 0;
                }
            }
        }

        // ---- backward: P_t|T = Z_t + J_t P_t+1|T J_t', sum P, sum U (U = P_t+1|T J_t' = Cov(f_t+1, f_t | X)) on the matrix pipe.
        // The matrices move to the D layout of v_mfma_f64_4x4x4 (four 4 x 4 blocks: lane l = element (4 dI + (l >> 4),
        // 4 dJ + (l & 3)) of block (dI, dJ) = bits 3, 2 of l), in which an 8 x 8 product is two MFMAs whose result lands where
        // the next product wants it: the A operand X[4 dI + (l & 3)][4 K + (l >> 4)] is one ds_bpermute of X, the B operand
        // Y[4 K + (l >> 4)][4 dJ + (l & 3)] is Y of lane l with bit 3 := K (one DPP row_ror:8 and a select).  The operands of
        // J_t do not depend on the chain.  (The LDS-tile products cost ~50 instructions and three LDS round trips per step.)
        const bool inLd = di < rl && dj < rl;
        auto emitP = [&](int trow, double P) {
            if (di >= r) return;
            if (a.P_smooth && dj <= di) a.P_smooth[((size_t)b * T + trow) * npr + di * (di + 1) / 2 + dj] = inLd ? P : (di == dj ? 1.0 : 0.0);
        };
        const double PsT = Ps;
        double PsD = __shfl(Ps, eD, 64);
        double S11c = PsD, S10c = 0.0, U = 0.0;
        double jA0 = 0.0, jA1 = 0.0, jT0 = 0.0, jT1 = 0.0;
        double zc[CHW], zn[CHW], jc[CHW], jn[CHW];
        int ec[CHW], en[CHW];
        auto issue_bwd = [&](int c) {
#pragma unroll
            for (int s = 0; s < CHW; ++s) {
                int t = c * CHW + s;
                t = t < T ? t : T - 1;
                const int ee = eidxS[t];
                en[s] = ee;
                zn[s] = ZJ[((size_t)ee * 2 + 0) * RR + eD];
                jn[s] = ZJ[((size_t)ee * 2 + 1) * RR + eD];
            }
        };
        double pb[CHW + 1];
        int tb[CHW + 1];
#pragma unroll
        for (int s = 0; s <= CHW; ++s) { pb[s] = 0.0; tb[s] = -1; }
        pb[CHW] = PsD; tb[CHW] = T - 1;
        auto flush_bwd = [&]() {
#pragma unroll
            for (int s = 0; s <= CHW; ++s) {
                if (tb[s] >= 0) emitP(tb[s], pb[s]);
                tb[s] = -1;
            }
        };
        int e_prev = -1;
        bool need_b = true;
        issue_bwd(nchunks - 1);
        for (int c = nchunks - 1; c >= 0; --c) {
#pragma unroll
            for (int s = 0; s < CHW; ++s) { zc[s] = zn[s]; jc[s] = jn[s]; ec[s] = en[s]; }
            flush_bwd();
            if (c - 1 >= 0) issue_bwd(c - 1);
            const int smax = (T - c * CHW) < CHW ? (T - c * CHW) : CHW;
            // DENSE: eight distinct table entries -- every step takes its own J and recomputes P (always valid: the test only
            // skips steps that would reproduce their input)
            auto step = [&](auto dense_tag, int s) {
                constexpr bool DENSE = decltype(dense_tag)::value;
                const int t = c * CHW + s;
                const bool changed = DENSE || ec[s] != e_prev;     // wave-uniform
                if (changed) {
                    Z = zc[s];
                    e_prev = ec[s];
                    jA0 = __shfl(jc[s], srcA0, 64); jA1 = __shfl(jc[s], srcA1, 64);
                    jT0 = __shfl(jc[s], srcT0, 64); jT1 = __shfl(jc[s], srcT1, 64);
                }
                if (DENSE || need_b || changed) {  // wave-uniform
                    const double pA0 = __shfl(PsD, srcA0, 64), pA1 = __shfl(PsD, srcA1, 64);
                    U = __builtin_amdgcn_mfma_f64_4x4x4f64(pA0, jT0, 0.0, 0, 0, 0);
                    U = __builtin_amdgcn_mfma_f64_4x4x4f64(pA1, jT1, U, 0, 0, 0);          // U = P_s J'
                    const double Ux = ror8(U);
                    const double uB0 = hiHalf ? Ux : U, uB1 = hiHalf ? U : Ux;
                    double pn_ = __builtin_amdgcn_mfma_f64_4x4x4f64(jA0, uB0, Z, 0, 0, 0);
                    pn_ = __builtin_amdgcn_mfma_f64_4x4x4f64(jA1, uB1, pn_, 0, 0, 0);      // Z + J U
                    if constexpr (!DENSE) {
                        const bool same = close_enough(pn_, PsD);
                        need_b = !G.all_true(same);
                    }
                    PsD = pn_;
                }
                if (em) {
                    S10c += U;
                    if (t > 0) S11c += PsD;
                }
                if (t > 0) { pb[s] = PsD; tb[s] = t - 1; }
            };
            if (smax == CHW && ec[CHW - 1] - ec[0] == CHW - 1) {
#pragma unroll
                for (int s = CHW - 1; s >= 0; --s) step(std::true_type{}, s);
                need_b = true;
            } else {
#pragma unroll
                for (int s = CHW - 1; s >= 0; --s)
                    if (s < smax) step(std::false_type{}, s);
            }
        }
        {   // back to the element layout of the forward sweep (lane 8 i + j) for the epilogue
            const int lD = j | ((i >> 2) << 3) | ((i & 3) << 4);
            Ps = __shfl(PsD, lD, 64);
            S11c = __shfl(S11c, lD, 64);
            S10c = __shfl(S10c, lD, 64);
        }
        flush_bwd();
        PSTAMP(pt2);
#ifdef DFM_PAIR_PROF
        if (lane == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const unsigned cu = (hw >> 8) & 15, se = (hw >> 13) & 7, simd = (hw >> 4) & 3;
            if ((xcc & 15) == 0 && se == 0 && cu < 2)
                printf("PAIRPROF cov b=%d xcc %u se %u cu %u simd %u: fwd %llu bwd %llu (10 ns ticks)\n", b, xcc & 15, se, cu, simd, pt1 - pt0, pt2 - pt1);
        }
#endif
        __syncthreads();                                           // (E1) the mean wave's sums
        if (em) {
            const double S11m = xch[5 * RR + lane], S10m = xch[6 * RR + lane];
            const double f0r = xch[7 * RR + lane], f0c = xch[8 * RR + lane], fTfT = xch[9 * RR + lane];
            const size_t o = (size_t)b * RR + lane;
            const double S11 = S11c + S11m, S10 = S10c + S10m;
            const double S00 = S11 - (PsT + fTfT) + fma(f0r, f0c, Ps);
            const bool narrow = a.rl > 0;
            if (!narrow) a.S11[o] = S11;
            a.S10[o] = S10;
            a.S00[o] = S00;
            a.P0s[o] = Ps;
            if (j == 0) a.f0s[(size_t)b * R + i] = f0r;
            if (a.A_out) {
                double inv = S00;
                double S10m_ = S10;
                if (a.kdim > 0 && a.ka > 0) {
                    if (i >= a.ka || j >= a.ka) inv = (i == j) ? 1.0 : 0.0;
                    if (j >= a.ka) S10m_ = 0.0;
                }
                (void)G.sweep_inverse(inv);
                G.sync();
                L0[TS * i + j] = S10m_;
                L1[TS * i + j] = inv;
                G.sync();
                const double An = dot_rows<R>(L0, L1, i, j);
                G.sync();
                L1[TS * i + j] = An;
                G.sync();
                double Qn = (S11 - dot_rows<R>(L1, L0, i, j)) / (double)T;
                Qn = 0.5 * (Qn + G.transposed(Qn));
                double Aout = An;
                if (a.kdim > 0) {
                    const int kd = a.kdim;
                    const int rb_ = a.kb > 0 ? a.kb : rl;
                    if (i >= rb_ && i < kd) Aout = (j == i - rb_) ? 1.0 : 0.0;
                    if ((i >= rb_ && i < kd) || (j >= rb_ && j < kd)) Qn = 0.0;
                }
                const double P0n = 0.5 * (Ps + G.transposed(Ps));
                double inv2 = S11;
                if (narrow) {
                    if (!inL) inv2 = (i == j) ? (double)T : 0.0;
                    if (inC) a.S11[(size_t)b * Rc * Rc + i * Rc + j] = inv2;
                }
                (void)G.sweep_inverse(inv2);
                if (narrow) { if (inC) a.S11inv[(size_t)b * Rc * Rc + i * Rc + j] = inv2; }
                else a.S11inv[o] = inv2;
                if (em_apply) {
                    a.A_out[o] = Aout;
                    a.Q_out[o] = Qn;
                    a.P0_out[o] = P0n;
                    if (j == 0) a.mu0_out[(size_t)b * R + i] = f0r;
                }
            }
        }
        return;
    }

    // =============================================== mean wave =====================================================
    __syncthreads();                                               // (P)
    const double KT = xch[0 * RR + lane];
    double xi = xch[1 * RR + lane];                                // column-distributed
    double cb[CHW], cs[CHW], cl[CHW], nb_[CHW], ns_[CHW], nl_[CHW];
    int cn[CHW], nn_[CHW];
    auto issue_fwd = [&](int c) {
#pragma unroll
        for (int s = 0; s < CHW; ++s) {
            int t = c * CHW + s;
            t = t < T ? t : T - 1;
            nb_[s] = j < Rc ? bcol[(size_t)t * Rc + j] : 0.0;
            ns_[s] = scol[t];
            nn_[s] = nobs[t];
            nl_[s] = ldrow[t];
        }
    };
    double sum_xw = 0.0, ssum = 0.0, nsum = 0.0, ldsum = 0.0;
    double wb[CHW];
#pragma unroll
    for (int s = 0; s < CHW; ++s) wb[s] = 0.0;
    auto flush_fwd = [&](int c) {
#pragma unroll
        for (int s = 0; s < CHW; ++s) {
            const int t = c * CHW + s;
            if (t < T && j == 0) wtab[(size_t)t * R + i] = wb[s];
        }
    };
    issue_fwd(0);
    for (int c = 0; c < nchunks; ++c) {
#pragma unroll
        for (int s = 0; s < CHW; ++s) { cb[s] = nb_[s]; cs[s] = ns_[s]; cn[s] = nn_[s]; cl[s] = nl_[s]; }
        lds_barrier();                                             // (F_c)
        if (c > 0) flush_fwd(c - 1);
        if (c + 1 < nchunks) issue_fwd(c + 1);
        const double* rb = ring + (c & 1) * CHW * RR;
        double zs[CHW];
#pragma unroll
        for (int s = 0; s < CHW; ++s) zs[s] = rb[s * RR + lane];
        const int smax = (T - c * CHW) < CHW ? (T - c * CHW) : CHW;
#pragma unroll
        for (int s = 0; s < CHW; ++s) {
            if (s < smax) {
                const double w = G.sum_j(zs[s] * xi);              // w = Z xi, row-distributed
                sum_xw = fma(xi, w, sum_xw);                       // (x_j w_i in every lane: the diagonal is picked at the end)
                wb[s] = w;
                xi = G.sum_i(KT * w) + cb[s];                      // xi <- K w + b_t, column-distributed
                ssum += cs[s];
                nsum += (double)cn[s];
                ldsum += (cn[s] == N) ? ldfull : cl[s];
            }
        }
    }
    flush_fwd(nchunks - 1);
    PSTAMP(pt1);
    {
        const double sxw = G.sum_i(G.sum_j(diag ? sum_xw : 0.0));
        xch[3 * RR + lane] = xi;
        if (lane == 0) { xch[4 * RR + 0] = sxw; xch[4 * RR + 1] = ssum; xch[4 * RR + 2] = nsum; xch[4 * RR + 3] = ldsum; }
    }
    __syncthreads();                                               // (T1)
    double fs_r = G.sum_j(xch[2 * RR + lane] * xi);                // f_T = P_T xi_T
    double fs_c = G.transposed(fs_r);
    const double fTfT = fs_r * fs_c;
    double S11m = fTfT, S10m = 0.0;

    // ---- backward: f_t|T = w_t + J_t f_t+1|T in both distributions (two independent reductions, no transpose on the chain)
    auto emitF = [&](int trow, double f_row) {
        if (i >= r) return;
        if (j == 0) a.f_smooth[((size_t)b * T + trow) * r + i] = i < rl ? f_row : 0.0;
    };
    double wc[CHW], wn[CHW], wcc[CHW], wnc[CHW], jc[CHW], jn[CHW], jtc[CHW], jtn[CHW];
    auto issue_bwd = [&](int c) {
#pragma unroll
        for (int s = 0; s < CHW; ++s) {
            int t = c * CHW + s;
            t = t < T ? t : T - 1;
            wn[s] = wtab[(size_t)t * R + i];
            wnc[s] = wtab[(size_t)t * R + j];
            const int ee = eidxS[t];
            jn[s] = ZJ[((size_t)ee * 2 + 1) * RR + lane];
            jtn[s] = ZJ[((size_t)ee * 2 + 1) * RR + 8 * j + i];
        }
    };
    double fb[CHW + 1];
    int tb[CHW + 1];
#pragma unroll
    for (int s = 0; s <= CHW; ++s) { fb[s] = 0.0; tb[s] = -1; }
    fb[CHW] = fs_r; tb[CHW] = T - 1;
    auto flush_bwd = [&]() {
#pragma unroll
        for (int s = 0; s <= CHW; ++s) {
            if (tb[s] >= 0) emitF(tb[s], fb[s]);
            tb[s] = -1;
        }
    };
    issue_bwd(nchunks - 1);
    for (int c = nchunks - 1; c >= 0; --c) {
#pragma unroll
        for (int s = 0; s < CHW; ++s) { wc[s] = wn[s]; wcc[s] = wnc[s]; jc[s] = jn[s]; jtc[s] = jtn[s]; }
        flush_bwd();
        if (c - 1 >= 0) issue_bwd(c - 1);
        const int smax = (T - c * CHW) < CHW ? (T - c * CHW) : CHW;
#pragma unroll
        for (int s = CHW - 1; s >= 0; --s) {
            if (s < smax) {
                const int t = c * CHW + s;
                const double fnew = wc[s] + G.sum_j(jc[s] * fs_c);
                const double fnew_c = wcc[s] + G.sum_i(jtc[s] * fs_r);
                const double fprev_r = fs_r;
                fs_r = fnew;
                fs_c = fnew_c;
                if (em) {
                    S10m = fma(fprev_r, fs_c, S10m);               // f_{t+1} f_t'
                    if (t > 0) S11m = fma(fs_r, fs_c, S11m);
                }
                if (t > 0) { fb[s] = fs_r; tb[s] = t - 1; }
            }
        }
    }
    flush_bwd();
    PSTAMP(pt2);
#ifdef DFM_PAIR_PROF
    if (lane == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const unsigned cu = (hw >> 8) & 15, se = (hw >> 13) & 7, simd = (hw >> 4) & 3;
        if ((xcc & 15) == 0 && se == 0 && cu < 2)
            printf("PAIRPROF mean b=%d xcc %u se %u cu %u simd %u: fwd %llu bwd %llu (10 ns ticks)\n", b, xcc & 15, se, cu, simd, pt1 - pt0, pt2 - pt1);
    }
#endif
    xch[5 * RR + lane] = S11m;
    xch[6 * RR + lane] = S10m;
    xch[7 * RR + lane] = fs_r;                                     // f_0|T
    xch[8 * RR + lane] = fs_c;
    xch[9 * RR + lane] = fTfT;
    __syncthreads();                                               // (E1)
}

static size_t pair_lds_bytes(int T) {
    constexpr size_t RT = (size_t)8 * kTileStride<8>;
    return (4 * RT + 2 * kPairChunk * 64 + 10 * 64) * sizeof(double) + (size_t)T * sizeof(int);
}

// Rp = 8, information form, batches of at most a.pair_bmax replicates (default: one replicate per SIMD -- two waves share a
// SIMD then; beyond that recursion_wave_kernel<8> with chunks of 4 already runs two replicates per SIMD and the split would
// only add barriers).
bool recursion_pair_supported(const RecursionArgs& a) {
    if (a.cov || a.B > a.pair_bmax) return false;
    if (a.rl != 0 && a.Rc == 0) return false;
    return pair_lds_bytes(a.T) <= 38 * 1024;                        // four workgroups per CU
}

hipError_t launch_recursion_pair(const RecursionArgs& a, hipStream_t s) {
    note_kernel("recursion_pair_kernel");
    hipLaunchKernelGGL(recursion_pair_kernel, dim3(a.B), dim3(128), pair_lds_bytes(a.T), s, a);
    return hipGetLastError();
}

}  // namespace dfm
