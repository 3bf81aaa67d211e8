// recursion_wave.hip -- recursion_kernel's algebra (recursion.hip: information-form filter + "Z-smoother", EM statistics
// and transition M-step) with ONE WAVE PER REPLICATE for Rp = 8 (state 5..8 wide; narrower states are padded to it), and
// the same element-per-thread layout on a 256-thread workgroup per replicate for Rp = 16 (Grid<16>).
//
// recursion_kernel gives a replicate r lanes (lane i = row i): at B = 1024 that is 128 waves on 1024 SIMDs, each walking
// 500 dependent periods at ~6 us a period (an r x r Gauss-Jordan through LDS, four r x r products of 64 FMAs per lane).
// Here lane l = 8 i + j holds ELEMENT (i, j) of every 8 x 8 matrix:
//   * inverse: symmetric sweep operator, pivot row / column by ds_bpermute (no LDS memory), pivot by v_readlane;
//     one FMA per lane per sweep instead of eight;
//   * products: both operands staged row-wise in a 512-byte LDS tile, 8 x ds_read_b128 + 8 FMAs per lane;
//   * matrix-vector products: vectors live "column-distributed" (lane (i,j) holds x_j) or "row-distributed" (x_i);
//     M x is one multiply and a 3-step DPP reduction over j (row-distributed result), M' x the same over i
//     (column-distributed result), so the mean recursion alternates forms and never transposes;
//   * 1024 waves instead of 128: every SIMD of the chip walks a replicate.
// Same inputs, scratch tables (Z_e, J_e row-major, w_t) and outputs as recursion_kernel; same memoisation of repeated
// covariance steps.  The reference has no counterpart (dfm_functions.ipynb:21-23 declares `Parametric` only).
#include <stdlib.h>
#include "dfm_kernels.h"
#include "dfm_smallmat.h"
#include "dfm_grid.h"

namespace dfm {

#ifdef DFM_WAVE_PROF
#define TICK(var) do { var -= (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define TOCK(var) do { var += (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TICK(var) do {} while (0)
#define TOCK(var) do {} while (0)
#endif

namespace {

constexpr double kLog2PiW = 1.8378770664093454835606594728112;
// periods per prefetch chunk at Rp = 8: 8 when the batch leaves at most one wave per SIMD (the deeper prefetch hides more of
// the chain's memory latency: 0.86 vs 1.07 ms per 1024 replicates), 4 beyond that -- 212 instead of 310 registers, so TWO
// waves share a SIMD and fill each other's waits (B = 2048: 1.26 ms with chunks of 4, 1.72 ms in two rounds of chunks of 8)

}  // namespace

// COV = true: covariance-form forward step (recursion.hip, COV): Q may be singular (companion states, DFM_F_SINGULAR_Q)
template <int R, bool COV, int CH8 = 8>
#ifndef DFM_WG16_WAVES
#define DFM_WG16_WAVES 2
#endif
// R = 16: a workgroup is one wave on each SIMD of a CU and a chain of barrier-separated exchanges -- latency-bound; the
// register budget is capped so that several workgroups share a CU and fill each other's waits.
__global__ __launch_bounds__(R * R, (R == 16 ? DFM_WG16_WAVES : 1)) void recursion_wave_kernel(RecursionArgs a) {   // R = 32: 16 waves, 128 VGPRs
    constexpr int RR = R * R;
    constexpr int CHW = R == 32 ? 2 : R == 16 ? 4 : CH8;
    extern __shared__ __attribute__((aligned(16))) double wsm[];
    double* LK = wsm;            // K = Q^-1 A, rows (constant; COV: A rows)
    constexpr int TS = kTileStride<R>, RT = R * TS;           // tile row stride, tile size (doubles)
    double* L0 = LK + RT;
    double* L1 = L0 + RT;
    double* LJ = L1 + RT;        // J rows (backward sweep)
    Grid<R> G;
    G.prow = LJ + RT;            // (R = 8: unused, zero bytes reserved)
    G.red = G.prow + (R >= 16 ? kGridProw<R> : 0);
    G.tt = G.red + (R >= 16 ? 2 * (RR / 64) * R : 0);
    int* eidxS = reinterpret_cast<int*>(G.tt + (R >= 16 ? 2 * RT : 0));   // [T] covariance-table entry of forward step t
    const int lane = threadIdx.x;
    const int i = lane / R, j = lane % R;
    G.l = lane; G.i = i; G.j = j;
    const int T = a.T, N = a.N, r = a.r;
    long long p_inv = 0, p_mm = 0, p_mean = 0, p_bcov = 0, p_bmean = 0, p_tot = 0, p_stage = 0;
    (void)p_inv; (void)p_mm; (void)p_mean; (void)p_bcov; (void)p_bmean; (void)p_tot; (void)p_stage;
#ifdef DFM_WAVE_PROF
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    unsigned long long t_fwd_end = 0, t_bwd_start = 0;
#endif
    const int b = blockIdx.x;
    if (a.only_if && a.only_if[b] == 0) return;              // (block-uniform) the replicate was done by recursion_chunk_kernel
    const bool diag = (i == j);

    const int Rc = a.Rc > 0 ? a.Rc : R;                      // width of the collapsed observations (state padded beyond it)
    const int NPc = Rc * (Rc + 1) / 2;
    const bool inC = i < Rc && j < Rc;
    const double* bcol = a.bcol + (size_t)b * T * Rc;
    const double* scol = a.scol + (size_t)b * T;
    const int* nobs = a.nobs + (size_t)b * T;
    const double* ldrow = a.ldrow + (size_t)b * T;
    const double ldfull = a.ldfull[b];
    double* ZJ = a.ZJtab + (size_t)b * (T + 1) * 2 * R * R;
    double* wtab = a.wtab + (size_t)b * T * R;
    const int pk = (i >= j) ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i;   // packed lower-triangle index of (i, j)

    // ---------------- prologue: constants --------------------------------------------------------------------
    const double Ael = a.A[(size_t)b * RR + lane];
    double Qi = a.Q[(size_t)b * RR + lane];
    const double Cf = inC ? a.Cfull[(size_t)b * Rc * Rc + i * Rc + j] : 0.0;
    double Omf = a.P0[(size_t)b * RR + lane];
    const double mu0c = a.mu0[(size_t)b * R + j];            // column-distributed
    // COV: Qi stays Q (never inverted), Omf holds P_f (P0 to start), xi holds m_f (mu0 to start), LK holds A rows
    double detQ = 1.0, detP0 = 1.0, K = 0.0, KT = 0.0, Phi = 0.0, q0_part = 0.0, xi = mu0c;
    if constexpr (!COV) {
        detQ = G.sweep_inverse(Qi);
        detP0 = G.sweep_inverse(Omf);                    // Om_f,0 = P0^-1
        // K = Qi A:  K_ij = sum_k Qi[i][k] A[k][j] = row i of Qi . row j of A'
        L0[TS * i + j] = Qi;
        L1[TS * j + i] = Ael;                                 // A'
        G.sync();
        K = dot_rows<R>(L0, L1, i, j);
        G.sync();
        LK[TS * i + j] = K;
        L0[TS * j + i] = K;                                   // K' rows = K columns
        G.sync();
        KT = L0[TS * i + j];                                       // K_ji
        // Phi = K' A = A' Qi A:  Phi_ij = sum_k K[k][i] A[k][j] = row i of K' . row j of A'
        Phi = dot_rows<R>(L0, L1, i, j);
        G.sync();
        xi = G.sum_j(Omf * mu0c);                         // xi_0 = P0^-1 mu0, row-distributed
        q0_part = diag ? mu0c * xi : 0.0;
        xi = G.transposed(xi);                           // column-distributed from here on
    } else {
        LK[TS * i + j] = Ael;
        G.sync();
    }
    (void)K;

    // ---------------- forward sweep ---------------------------------------------------------------------------
    const int nchunks = (T + CHW - 1) / CHW;
    double cb[CHW], cs[CHW], cl[CHW], cc[CHW], nb_[CHW], ns_[CHW], nl_[CHW], nc_[CHW];
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
            nc_[s] = (a.Ct && inC) ? a.Ct[((size_t)b * T + t) * NPc + pk] : 0.0;
        }
    };
    auto take_fwd = [&]() {
#pragma unroll
        for (int s = 0; s < CHW; ++s) { cb[s] = nb_[s]; cs[s] = ns_[s]; cn[s] = nn_[s]; cl[s] = nl_[s]; cc[s] = nc_[s]; }
    };

    double Z = 0.0, Jr = 0.0, Omp = 0.0, Gm = 0.0, mf_r = 0.0, detP_cur = 1.0;
    double detM_cur = 1.0, sum_xw = 0.0, ssum = 0.0, nsum = 0.0, ldsum = 0.0;
    LogProd detprod;                                           // prod_t det(Om_f,t + Phi)
    int e = -1;
    bool need_cov = true;
    // Loads and stores share the in-order vmcnt counter on gfx9: a store issued just before a wait for prefetched loads
    // would expose its full latency.  So a chunk's outputs are buffered in registers and flushed at the top of the NEXT
    // chunk, right before that chunk issues the prefetch of the one after: every wait then covers operations that are a
    // whole chunk old.
    double zb[CHW], jb[CHW], wb[CHW];
    int eb[CHW];
#pragma unroll
    for (int s = 0; s < CHW; ++s) { zb[s] = 0.0; jb[s] = 0.0; wb[s] = 0.0; eb[s] = -1; }
    auto flush_fwd = [&](int c) {                              // outputs of chunk c
#pragma unroll
        for (int s = 0; s < CHW; ++s) {
            const int t = c * CHW + s;
            if (t < T) {
                if (eb[s] >= 0) {
                    ZJ[((size_t)eb[s] * 2 + 0) * RR + lane] = zb[s];
                    ZJ[((size_t)eb[s] * 2 + 1) * RR + lane] = jb[s];
                }
                if constexpr (COV) { if (i == 0) wtab[(size_t)t * R + j] = wb[s]; }   // w_t column-distributed there
                else { if (j == 0) wtab[(size_t)t * R + i] = wb[s]; }
            }
        }
    };
    issue_fwd(0);
    for (int c = 0; c < nchunks; ++c) {
        take_fwd();
        if (c > 0) flush_fwd(c - 1);
        if (c + 1 < nchunks) issue_fwd(c + 1);
        const int smax = (T - c * CHW) < CHW ? (T - c * CHW) : CHW;
#pragma unroll
        for (int s = 0; s < CHW; ++s) {
            if (s < smax) {
                const int t = c * CHW + s;
                eb[s] = -1;
                if constexpr (COV) {
                    const bool full = (cn[s] == N);
                    const double Crow = full ? Cf : cc[s];
                    if (need_cov || !full) {  // wave-uniform; need_cov == false: the last computed step had a full row and reproduced its P_f
                        G.sync();
                        L0[TS * i + j] = Omf;                            // P_f rows (symmetric)
                        G.sync();
                        const double AP = dot_rows<R>(LK, L0, i, j);  // A P_f
                        G.sync();
                        L1[TS * i + j] = AP;
                        L0[TS * j + i] = AP;                        // (A P_f)' rows = its columns
                        G.sync();
                        Omp = dot_rows<R>(L1, LK, i, j) + Qi;         // P_p = A P_f A' + Q
                        detP_cur = G.sweep_inverse(Omp);       // Om_p = P_p^-1
                        G.sync();
                        LJ[TS * i + j] = Omp;
                        G.sync();
                        Gm = dot_rows<R>(LJ, L0, i, j);               // G = Om_p A P_f
                        Jr = dot_rows<R>(LJ, L0, j, i);               // J = G' = P_f A' Om_p
                        G.sync();
                        L1[TS * i + j] = Jr;
                        G.sync();
                        Z = Omf - dot_rows<R>(L1, L0, i, j);          // Z = P_f - J A P_f
                        double Pn = Omp + Crow;
                        detM_cur = G.sweep_inverse(Pn);        // P_f' = (Om_p + C_t)^-1
                        // (a row with missing cells is a new step whatever the test says: no ballot / barrier for it)
                        const bool steady = full && G.all_true(close_enough(Pn, Omf));
                        Omf = Pn;
                        need_cov = !steady;
                        ++e;
                        zb[s] = Z; jb[s] = Jr; eb[s] = e;
                    }
                    if (lane == 0) eidxS[t] = e;
                    // m_p = A m_f, w = m_f - J m_p, m_f' = P_f' (Om_p m_p + b_t)
                    const double mp = G.sum_j(Ael * xi);                    // row-distributed
                    wb[s] = xi - G.sum_i(Gm * mp);                          // column-distributed
                    const double y = G.sum_i(Omp * mp) + cb[s];             // column-distributed (Om_p symmetric)
                    const double cm = G.sum_i(Crow * mp);
                    mf_r = G.sum_j(Omf * y);                                // row-distributed
                    sum_xw += diag ? fma(cb[s], mp, (cb[s] - cm) * mf_r) : 0.0;   // quad_t = s_t - sum_i (b_i m_p,i + u_i m_f,i)
                    detprod.mul(detM_cur);                                     // log det(I + C_t P_p) = log det(Om_p + C_t) + log det P_p
                    detprod.mul(detP_cur);
                    xi = G.transposed(mf_r);
                    ssum += cs[s];
                    nsum += (double)cn[s];
                    ldsum += full ? ldfull : cl[s];
                    continue;
                }
                const bool computed = need_cov;
                const double Omf_used = Omf;
                if (need_cov) {  // wave-uniform
                    Z = Omf + Phi;
                    TICK(p_inv);
                    detM_cur = G.sweep_inverse(Z);
                    TOCK(p_inv);
                    TICK(p_mm);
                    G.sync();
                    L0[TS * i + j] = Z;
                    G.sync();
                    Jr = dot_rows<R>(L0, LK, i, j);               // J = Z K'
                    L1[TS * j + i] = Jr;                        // J' rows = J columns
                    G.sync();
                    Omp = Qi - dot_rows<R>(LK, L1, i, j);         // Om_p = Qi - K J
                    ++e;
                    zb[s] = Z; jb[s] = Jr; eb[s] = e;
                    TOCK(p_mm);
                }
                TICK(p_mean);
                if (lane == 0) eidxS[t] = e;
                // mean recursion: w = Z xi (row-distributed), xi <- K w + b_t (column-distributed)
                const double w = G.sum_j(Z * xi);
                sum_xw += diag ? xi * w : 0.0;
                detprod.mul(detM_cur);
                wb[s] = w;
                xi = G.sum_i(KT * w) + cb[s];
                ssum += cs[s];
                const double nt = (double)cn[s];
                nsum += nt;
                const bool full = (cn[s] == N);
                ldsum += full ? ldfull : cl[s];                // ldrow is only written for rows with NaN
                const double Omf_new = Omp + (full ? Cf : cc[s]);
                if (computed) {
                    need_cov = !(full && G.all_true(close_enough(Omf_new, Omf_used)));   // (no ballot / barrier for a row with missing cells)
                } else {
                    need_cov = !full;
                }
                Omf = Omf_new;
                TOCK(p_mean);
            }
        }
    }
    flush_fwd(nchunks - 1);
#ifdef DFM_WAVE_PROF
    t_fwd_end = __builtin_amdgcn_s_memtime();
#endif

    // ---------------- terminal: P_T = Om_f^-1, f_T = P_T xi, log-likelihood ------------------------------------
    bool em_apply = true;
    double Ps = Omf;
    double detOmT = 1.0, fs_r = mf_r;                          // COV: P_T = P_f, f_T = m_f as they stand
    if constexpr (!COV) {
        detOmT = G.sweep_inverse(Ps);
        fs_r = G.sum_j(Ps * xi);                            // f_T, row-distributed
    }
    {
        double part = COV ? -sum_xw : (diag ? q0_part - xi * fs_r : 0.0) - sum_xw;
        const double qd = G.sum_i(G.sum_j(part));
        const double LD = COV ? detprod.log_value()
                              : log(detOmT) + log(detP0) + (double)T * log(detQ) + detprod.log_value();   // sum_ldz = -log prod
        const double ll = -0.5 * (nsum * kLog2PiW + ldsum + LD + ssum + qd);
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

    // ---------------- backward sweep ----------------------------------------------------------------------------
    const int npr = r * (r + 1) / 2;
    const int rl = a.rl > 0 ? a.rl : R;      // observation loads on the first rl state components; the rest of the output
    const bool inL = i < rl && j < rl;       // layout is padding for the loadings step (mean 0, identity covariance)
    auto emit = [&](int trow, double P, double f_row) {        // smoothed moments of period trow + 1
        if (i >= r) return;
        if (j == 0) a.f_smooth[((size_t)b * T + trow) * r + i] = i < rl ? f_row : 0.0;
        if (a.P_smooth && j <= i) a.P_smooth[((size_t)b * T + trow) * npr + i * (i + 1) / 2 + j] = inL ? P : (i == j ? 1.0 : 0.0);
    };
    double fs_c = G.transposed(fs_r);
    double Jt = 0.0;                                           // J_ji
    const bool em = a.S11 != nullptr;
    const double termT = fma(fs_r, fs_c, Ps);                  // E[f_T f_T'] element
    double S11 = termT, S10 = 0.0, U = 0.0;

    // per chunk: w_t in both distributions and the table entry (Z, J) of every step, addressed through the LDS index
    // the forward sweep left; outputs buffered and flushed like the forward sweep's
    double wc[CHW], wn[CHW], wcc[CHW], wnc[CHW], zc[CHW], zn[CHW], jc[CHW], jn[CHW];
    int ec[CHW], en[CHW];
    auto issue_bwd = [&](int c) {
#pragma unroll
        for (int s = 0; s < CHW; ++s) {
            int t = c * CHW + s;
            t = t < T ? t : T - 1;
            wn[s] = wtab[(size_t)t * R + i];                   // row-distributed
            wnc[s] = wtab[(size_t)t * R + j];                  // column-distributed
            const int ee = eidxS[t];
            en[s] = ee;
            zn[s] = ZJ[((size_t)ee * 2 + 0) * RR + lane];
            jn[s] = ZJ[((size_t)ee * 2 + 1) * RR + lane];
        }
    };
    double pb[CHW + 1], fb[CHW + 1];                           // smoothed moments to emit: rows c CHW - 1 + s, and T - 1 first
    int tb[CHW + 1];
#pragma unroll
    for (int s = 0; s <= CHW; ++s) { pb[s] = 0.0; fb[s] = 0.0; tb[s] = -1; }
    pb[CHW] = Ps; fb[CHW] = fs_r; tb[CHW] = T - 1;
    auto flush_bwd = [&]() {
#pragma unroll
        for (int s = 0; s <= CHW; ++s) {
            if (tb[s] >= 0) emit(tb[s], pb[s], fb[s]);
            tb[s] = -1;
        }
    };
#ifdef DFM_WAVE_PROF
    t_bwd_start = __builtin_amdgcn_s_memtime();
#endif
    G.sync();                                           // eidxS complete
    int e_prev = -1;                                           // entry staged in LJ
    bool need_b = true;
    issue_bwd(nchunks - 1);
    for (int c = nchunks - 1; c >= 0; --c) {
#pragma unroll
        for (int s = 0; s < CHW; ++s) { wc[s] = wn[s]; wcc[s] = wnc[s]; zc[s] = zn[s]; jc[s] = jn[s]; ec[s] = en[s]; }
        flush_bwd();
#ifdef DFM_WAVE_PROF
    p_tot = (long long)(__builtin_amdgcn_s_memtime() - t_start);
    if (b == 5 && lane == 0)
        printf("wave prof: fwd span %lld bwd span %lld | (memtime ticks): total %lld  fwd: inverse %lld products %lld mean %lld | bwd: stage %lld cov %lld mean %lld  (T=%d)\n",
               (long long)(t_fwd_end - t_start), p_tot - (long long)(t_bwd_start - t_start), p_tot, p_inv, p_mm, p_mean, p_stage, p_bcov, p_bmean, T);
#endif
        if (c - 1 >= 0) issue_bwd(c - 1);
        const int smax = (T - c * CHW) < CHW ? (T - c * CHW) : CHW;
#pragma unroll
        for (int s = CHW - 1; s >= 0; --s) {
            if (s < smax) {
                const int t = c * CHW + s;       // step t: from period t+1 to period t (t = 0: initial state)
                const bool changed = ec[s] != e_prev;          // wave-uniform
                TICK(p_stage);
                if (changed) {
                    Z = zc[s]; Jr = jc[s];
                    e_prev = ec[s];
                    G.sync();
                    LJ[TS * i + j] = Jr;
                    G.sync();
                    Jt = LJ[TS * j + i];
                }
                TOCK(p_stage);
                TICK(p_bcov);
                if (need_b || changed) {  // wave-uniform
                    G.sync();
                    L0[TS * i + j] = Ps;
                    G.sync();
                    U = dot_rows<R>(L0, LJ, i, j);                // U = P_s J' = Cov(f_{t+1}, f_t | X)
                    L1[TS * j + i] = U;                         // U' rows = U columns
                    G.sync();
                    const double pn_ = Z + dot_rows<R>(LJ, L1, i, j);   // Z + J U
                    // the test only matters when the NEXT step reuses this table entry (wave-uniform, known from the prefetch)
                    const int e_next = s > 0 ? ec[s - 1] : en[CHW - 1];
                    const bool reuse = (s > 0 || c > 0) && e_next == ec[s];
                    need_b = reuse ? !G.all_true(close_enough(pn_, Ps)) : true;
                    Ps = pn_;
                }
                TOCK(p_bcov);
                TICK(p_bmean);
                // f_t = w_t + J f_{t+1} in both distributions, by two independent reductions (no transpose on the chain)
                const double fnew = wc[s] + G.sum_j(Jr * fs_c);
                const double fnew_c = wcc[s] + G.sum_i(Jt * fs_r);
                const double fprev_r = fs_r;
                fs_r = fnew;
                fs_c = fnew_c;
                if (em) {
                    S10 += fma(fprev_r, fs_c, U);              // E[f_{t+1} f_t']
                    if (t > 0) S11 += fma(fs_r, fs_c, Ps);
                }
                if (t > 0) { pb[s] = Ps; fb[s] = fs_r; tb[s] = t - 1; }
                TOCK(p_bmean);
            }
        }
    }
    flush_bwd();
    // now fs / Ps are the smoothed moments of the initial state f_0
    if (em) {
        const size_t o = (size_t)b * RR + lane;
        const double S00 = S11 - termT + fma(fs_r, fs_c, Ps);
        // a.rl > 0: the loadings step sees the first Rc components only -- S11 / S11^-1 go out in its [Rc][Rc] layout
        const bool narrow = a.rl > 0;
        if (!narrow) a.S11[o] = S11;
        a.S10[o] = S10;
        a.S00[o] = S00;
        a.P0s[o] = Ps;
        if (j == 0) a.f0s[(size_t)b * R + i] = fs_r;
        if (a.A_out) {
            // A = S10 S00^-1 ;  Q = sym(S11 - A S10') / T ;  mu0 = f_0|T ;  P0 = sym(P_0|T) ;  S11^-1
            double inv = S00;
            double S10m = S10;
            if (a.kdim > 0 && a.ka > 0) {   // VAR(p) inside a wider state: A = S10[:, :ka] S00[:ka, :ka]^-1, zero beyond
                if (i >= a.ka || j >= a.ka) inv = (i == j) ? 1.0 : 0.0;
                if (j >= a.ka) S10m = 0.0;
            }
            (void)G.sweep_inverse(inv);
            G.sync();
            L0[TS * i + j] = S10m;
            L1[TS * i + j] = inv;                                    // symmetric: rows = columns
            G.sync();
            const double An = dot_rows<R>(L0, L1, i, j);
            G.sync();
            L1[TS * i + j] = An;
            G.sync();
            double Qn = (S11 - dot_rows<R>(L1, L0, i, j)) / (double)T;   // (A S10')_ij = row i of A . row j of S10
            Qn = 0.5 * (Qn + G.transposed(Qn));
            double Aout = An;
            if (a.kdim > 0) {   // companion state: only [A_1 .. A_p] and the innovation covariance of f_t are free
                const int kd = a.kdim;
                const int rb = a.kb > 0 ? a.kb : rl;               // block size of the companion state
                if (i >= rb && i < kd) Aout = (j == i - rb) ? 1.0 : 0.0;
                if ((i >= rb && i < kd) || (j >= rb && j < kd)) Qn = 0.0;
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
                if (j == 0) a.mu0_out[(size_t)b * R + i] = fs_r;
            }
        }
    }
}

template <int R>
static size_t wave_lds_bytes(int T) {
    constexpr size_t RT = (size_t)R * kTileStride<R>;
    const size_t extra = R >= 16 ? (kGridProw<R> + 2 * (R * R / 64) * R + 2 * RT) : 0;
    return (4 * RT + extra) * sizeof(double) + (size_t)T * sizeof(int);
}

bool recursion_wave_supported(int Rpad, const RecursionArgs& a) {
    if (a.rl != 0 && a.Rc == 0) return false;
    constexpr size_t cap = 150 * 1024;                            // LDS: tiles + 4 bytes per period
    // Rp = 16 (state 9..16 wide: r = 4 factors with VAR(4) dynamics): a 256-thread workgroup per replicate
    if (Rpad == 16) return wave_lds_bytes<16>(a.T) <= cap;
    // Rp = 32 (17..32: r = 8 factors with VAR(4) dynamics, AR(4) idiosyncratic terms at r = 4): 1024 threads per replicate
    if (Rpad == 32) return wave_lds_bytes<32>(a.T) <= cap;
    // Rp = 8, batch size: the lane-group kernel packs 8 replicates in a wave and stays at its ~3.3 ms latency floor up to
    // B ~ 8192; a wave per replicate costs 0.86 ms per 1024 replicates while a SIMD holds one wave (chunks of 8) and 0.63 ms
    // per 1024 once two share a SIMD (chunks of 4: 2.4 ms at B = 4096) -- it wins up to ~6000.
    static const int bmax = [] { const char* v = diag_env("DFM_WAVE_BMAX"); return v ? atoi(v) : 6144; }();
    if (a.Rc == 0 && a.B > bmax) return false;
    return Rpad == 8 && wave_lds_bytes<8>(a.T) <= 60 * 1024;
}

template <int R, bool COV, int CH8>
static hipError_t launch_wave_cov_ch(const RecursionArgs& a, hipStream_t s) {
    const size_t lds = wave_lds_bytes<R>(a.T);
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        // (the kernel also has 256 bytes of static LDS: 160 KB of dynamic LDS on top would be refused -- every Rp >= 16 panel
        // with T > ~1000 failed here with "invalid argument"; recursion_wave_supported caps the request at 150 KB)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&recursion_wave_kernel<R, COV, CH8>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((recursion_wave_kernel<R, COV, CH8>), dim3(a.B), dim3(R * R), lds, s, a);
    return hipGetLastError();
}
template <int R, bool COV>
static hipError_t launch_wave_cov(const RecursionArgs& a, hipStream_t s) {
    if constexpr (R == 8) {
        static const int simds = [] {
            int dev = 0; hipDeviceProp_t pr;
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 1024;
            return pr.multiProcessorCount * 4;
        }();
        if (a.B > simds) return launch_wave_cov_ch<R, COV, 4>(a, s);
    }
    return launch_wave_cov_ch<R, COV, 8>(a, s);
}

template <int R>
static hipError_t launch_wave(const RecursionArgs& a, hipStream_t s) {
    return a.cov ? launch_wave_cov<R, true>(a, s) : launch_wave_cov<R, false>(a, s);
}

// the sequential kernel for the replicates recursion_chunk_kernel handed back (a.only_if): one wave per replicate at any batch size
bool recursion_wave8_fits(int T) { return wave_lds_bytes<8>(T) <= 60 * 1024; }
// (up to one replicate per SIMD the covariance-wave + mean-wave pair is the faster sequential kernel: 0.62 against 0.88 ms per 1024
// replicates of 500 periods -- on the real Stock-Watson window EVERY replicate comes back here)
hipError_t launch_recursion_wave8_fallback(const RecursionArgs& a, hipStream_t s) {
    if (recursion_pair_supported(a)) return launch_recursion_pair(a, s);
    note_kernel("recursion_wave_kernel");
    return launch_wave<8>(a, s);
}

hipError_t launch_recursion_wave(const RecursionArgs& a, hipStream_t s, int Rpad) {
    if (Rpad == 8 && recursion_pair_supported(a)) return launch_recursion_pair(a, s);
    note_kernel("recursion_wave_kernel");
    return Rpad == 32 ? launch_wave<32>(a, s) : Rpad == 16 ? launch_wave<16>(a, s) : launch_wave<8>(a, s);
}

}  // namespace dfm
