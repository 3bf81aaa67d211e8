// recursion.hip -- the sequential part of one Kalman-smoother pass, batched over replicates.
//
// Algebra (pinned in oracle/info_form.py against the covariance-form oracle): information-form
// filter + "Z-smoother"; ONE r x r SPD inversion per period serves filter and smoother.
//   constants   Qi = Q^-1, Psi = A' Qi, Phi = A' Qi A;  Om_f,0 = P0^-1, xi_0 = P0^-1 mu0
//   forward t = 0..T-1 (step t consumes panel row t = period t+1):
//       Z = (Om_f + Phi)^-1,  J = Z Psi,  Om_p = Qi - Psi' J,  w = Z xi,
//       xi <- Psi' w + b_t,   Om_f <- Om_p + C_t
//   terminal    P_T = Om_f^-1,  f_T = P_T xi
//   backward    P_s <- Z + J (P_s J')',  f_s <- w + J f_s      (P_s J' = Cov(f_t+1, f_t | X))
//   log-lik     telescoped sums of log det Z_t and xi_t' w_t (see oracle/info_form.py)
// The reference has no counterpart (dfm_functions.ipynb:21-23 declares `Parametric` only); its
// state-space notation is dfm_functions.ipynb:30-34.
//
// Mapping (wave64): a group of R lanes owns one replicate, lane i keeps ROW i of every r x r
// matrix in registers; 64/R replicates per wave, one wave per workgroup.  Rows that other lanes need
// (pivot row of a Gauss-Jordan sweep, operand of a product, a state vector) are exchanged through a
// per-group LDS slot and read back with same-address (broadcast) ds_read_b128.
// Covariance steps are memoised: when Om_f repeats (balanced stretch of the panel: Riccati fixed
// point reached in a handful of periods) Z, J, Om_p are reused and only the O(r^2) mean recursion
// runs; same for P_s going backward.  Distinct (Z, J) pairs go to a scratch table in HBM; an LDS
// bitmask remembers at which periods a new pair was made.
#include "dfm_kernels.h"
#include "dfm_smallmat.h"

namespace dfm {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;
constexpr int CH = 8;                 // periods per prefetch chunk

template <int R>
struct RecLayout {
    static constexpr int GPW = 64 / R;                       // replicates per wave
    // per-group doubles: X (exchange), PSI, JS, V0, V1, ring[2][CH][R+3]
    static constexpr int kRing = 2 * CH * (R + 3);
    static constexpr int kRaw = 3 * R * R + 2 * R + kRing;
    static constexpr int S = ((kRaw + 3) / 4) * 4 + 2;       // S/2 odd: groups land on distinct 16-B bank slots
    static __host__ __device__ size_t lds_bytes(int T) {
        return (size_t)GPW * S * sizeof(double) + (size_t)((T + 31) / 32 + 1) * sizeof(unsigned);
    }
};

// COV = true: the same sweeps with the forward covariance step in COVARIANCE form,
//       P_p = A P_f A' + Q,  Om_p = P_p^-1,  J = P_f A' Om_p,  Z = P_f - J A P_f,  w = (I - J A) m_f,
//       P_f <- (Om_p + C_t)^-1,  m_f <- P_f (Om_p A m_f + b_t),   log det(I + C_t P_p) = log det(Om_p + C_t) + log det P_p
// which never inverts Q: the state innovation covariance may be singular (companion form of VAR(p) factor dynamics,
// SURVEY.md §8 f3; dfm_functions.ipynb:477-492 builds that companion for the reference's factor VAR).  Costs two
// inversions and four products per distinct covariance step instead of one and two.  The collapsed observations may
// then be narrower than the state (a.Rc: loadings only on the first a.rl state components).
template <int R, bool COV>
__global__ __launch_bounds__(64) void recursion_kernel(RecursionArgs a) {
    using LY = RecLayout<R>;
    constexpr int GPW = LY::GPW;
    constexpr int NPp = R * (R + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const int g = lane / R, i = lane % R;
    const int T = a.T, N = a.N, r = a.r;
    int b = blockIdx.x * GPW + g;
    const bool live = b < a.B;
    if (!live) b = a.B - 1;

    double* X = smem + (size_t)g * LY::S;
    double* PSI = X + R * R;
    double* JS = PSI + R * R;
    double* V0 = JS + R * R;
    double* V1 = V0 + R;
    double* RING = V1 + R;  // [2][CH][R+3]: b[0..R), s, n, ld
    unsigned* cmask = reinterpret_cast<unsigned*>(smem + (size_t)GPW * LY::S);
    const int nwords = (T + 31) / 32 + 1;
    for (int w = lane; w < nwords; w += 64) cmask[w] = 0u;

    const double* Ab = a.A + (size_t)b * R * R;
    const double* Qb = a.Q + (size_t)b * R * R;
    const double* P0b = a.P0 + (size_t)b * R * R;
    const int Rc = a.Rc > 0 ? a.Rc : R;                     // width of the collapsed observations (<= R)
    const int NPc = Rc * (Rc + 1) / 2;
    const double* bcol = a.bcol + (size_t)b * T * Rc;
    const double* scol = a.scol + (size_t)b * T;
    const int* nobs = a.nobs + (size_t)b * T;
    const double* ldrow = a.ldrow + (size_t)b * T;
    const double ldfull = a.ldfull[b];
    double* ZJ = a.ZJtab + (size_t)b * (T + 1) * 2 * R * R;
    double* wtab = a.wtab + (size_t)b * T * R;

    // ---------------- prologue: constants --------------------------------------------------------
    // COV: Qi holds Q (never inverted), Omf holds P_f (P0 to start), xi holds m_f (mu0 to start), PSI holds A
    double Arow[R], Qi[R], PsiT[R], Phi[R], Cf[R], Omf[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        Arow[j] = Ab[i * R + j];
        Qi[j] = Qb[i * R + j];
        Cf[j] = (i < Rc && j < Rc) ? a.Cfull[(size_t)b * Rc * Rc + i * Rc + j] : 0.0;
        Omf[j] = P0b[i * R + j];
        PsiT[j] = 0.0; Phi[j] = 0.0;
    }
    const double mu0i = a.mu0[(size_t)b * R + i];
    double detQ = 1.0, detP0 = 1.0, xi = mu0i, q0_part = 0.0;
    if constexpr (!COV) {
        detQ = gj_inverse<R>(Qi, X, i);
        detP0 = gj_inverse<R>(Omf, X, i);                  // Omf = P0^-1 = Om_f,0
        __syncthreads();
        store_row<R>(X, i, Arow);                          // X = A rows
        __syncthreads();
        mm_rows<R>(PsiT, Qi, X);                           // Psi' = Qi A   (row i)
#pragma unroll
        for (int j = 0; j < R; ++j) PSI[j * R + i] = PsiT[j];  // PSI = Psi rows (transpose of Psi')
        __syncthreads();
        {
            double prow[R];
#pragma unroll
            for (int k = 0; k < R; ++k) prow[k] = PSI[i * R + k];
            mm_rows<R>(Phi, prow, X);                      // Phi = Psi A
        }
        V0[i] = mu0i;
        __syncthreads();
        xi = dot_vec<R>(Omf, V0);                          // xi_0 = P0^-1 mu0
        q0_part = mu0i * xi;
    } else {
        __syncthreads();
        store_row<R>(PSI, i, Arow);                        // PSI = A rows for the whole sweep
        __syncthreads();
    }

    // ---------------- forward sweep --------------------------------------------------------------
    const int nchunks = (T + CH - 1) / CH;
    constexpr int SPL = (CH + R - 1) / R;                  // scalar-loading steps per lane
    double pb[CH], ps[SPL], pl[SPL];
    int pn[SPL];
    auto issue_fwd = [&](int c) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            int t = c * CH + s;
            t = t < T ? t : T - 1;
            pb[s] = i < Rc ? bcol[(size_t)t * Rc + i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
            const int s = i + q * R;
            int t = c * CH + s;
            t = t < T ? t : T - 1;
            if (s < CH) { ps[q] = scol[t]; pn[q] = nobs[t]; pl[q] = ldrow[t]; }
        }
    };
    auto commit_fwd = [&](int slot) {
        double* ring = RING + slot * CH * (R + 3);
#pragma unroll
        for (int s = 0; s < CH; ++s) ring[s * (R + 3) + i] = pb[s];
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
            const int s = i + q * R;
            if (s < CH) {
                ring[s * (R + 3) + R] = ps[q];
                ring[s * (R + 3) + R + 1] = (double)pn[q];
                ring[s * (R + 3) + R + 2] = (pn[q] == N) ? ldfull : pl[q];   // ldrow is only written for rows with NaN
            }
        }
    };

    double Z[R], Jr[R], Omp[R];
#pragma unroll
    for (int j = 0; j < R; ++j) { Z[j] = 0.0; Jr[j] = 0.0; Omp[j] = 0.0; }
    double ldz_cur = 0.0, sum_ldz = 0.0, sum_xw = 0.0, ssum = 0.0, nsum = 0.0, ldsum = 0.0;
    int e = -1;
    bool need_cov = true;

    double ctn[R];                                         // row i of C_t, one period ahead of the recursion
#pragma unroll
    for (int j = 0; j < R; ++j) ctn[j] = 0.0;
    if (a.Ct) {
        const double* ct = a.Ct + (size_t)b * T * NPc;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int hi = i > j ? i : j, lo = i > j ? j : i;
            ctn[j] = hi < Rc ? ct[hi * (hi + 1) / 2 + lo] : 0.0;
        }
    }
    issue_fwd(0);
    __syncthreads();
    commit_fwd(0);
    if (nchunks > 1) issue_fwd(1);
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const double* ring = RING + (c & 1) * CH * (R + 3);
        const int smax = (T - c * CH) < CH ? (T - c * CH) : CH;
        for (int s = 0; s < smax; ++s) {
            const int t = c * CH + s;
            if constexpr (COV) {
                const double* rs = ring + s * (R + 3);
                const double nt = rs[R + 1];
                const bool full = (nt == (double)N);
                double Crow[R];                            // row i of this period's C_t
#pragma unroll
                for (int j = 0; j < R; ++j) Crow[j] = full ? Cf[j] : ctn[j];
                if (a.Ct) {   // C_{t+1}: in flight during this step
                    const int tn = t + 1 < T ? t + 1 : T - 1;
                    const double* ct = a.Ct + ((size_t)b * T + tn) * NPc;
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        const int hi = i > j ? i : j, lo = i > j ? j : i;
                        ctn[j] = hi < Rc ? ct[hi * (hi + 1) / 2 + lo] : 0.0;
                    }
                }
                // need_cov == false: the last computed step had a full row and reproduced its own P_f (fixed point)
                const bool compute = !__all(!need_cov && full);
                if (compute) {  // wave-uniform
                    double AP[R], tmp[R];
                    __syncthreads();
                    store_row<R>(X, i, Omf);               // X = P_f
                    __syncthreads();
                    mm_rows<R>(AP, Arow, X);               // A P_f      (row i)
                    mm_rowsT<R>(tmp, AP, PSI);             // A P_f A'
#pragma unroll
                    for (int j = 0; j < R; ++j) Omp[j] = tmp[j] + Qi[j];   // P_p
                    const double detPp = gj_inverse<R>(Omp, X, i);          // Om_p = P_p^-1
                    __syncthreads();
                    store_row<R>(JS, i, AP);
                    __syncthreads();
                    mm_rows<R>(tmp, Omp, JS);              // G = Om_p A P_f
                    store_row<R>(X, i, tmp);
                    __syncthreads();
#pragma unroll
                    for (int j = 0; j < R; ++j) Jr[j] = X[j * R + i];       // J = G'
                    mm_rows<R>(tmp, Jr, JS);               // J A P_f
                    bool same = full;
                    double Pn[R];
#pragma unroll
                    for (int j = 0; j < R; ++j) { Z[j] = Omf[j] - tmp[j]; Pn[j] = Omp[j] + Crow[j]; }
                    const double detOf = gj_inverse<R>(Pn, X, i);           // P_f' = (Om_p + C_t)^-1
                    ldz_cur = log(detOf) + log(detPp);     // log det(I + C_t P_p)
#pragma unroll
                    for (int j = 0; j < R; ++j) { same = same && close_enough(Pn[j], Omf[j]); Omf[j] = Pn[j]; }
                    need_cov = !__all(same);
                    ++e;
                    {
                        double* zt = ZJ + ((size_t)e * 2 + 0) * R * R + i * R;
                        double* jt = ZJ + ((size_t)e * 2 + 1) * R * R + i * R;
#pragma unroll
                        for (int j = 0; j < R; ++j) { zt[j] = Z[j]; jt[j] = Jr[j]; }
                    }
                    if (lane == 0) cmask[t >> 5] |= 1u << (t & 31);
                }
                // mean recursion: m_p = A m_f, w = m_f - J m_p, m_f' = P_f' (Om_p m_p + b_t)
                __syncthreads();
                V0[i] = xi;
                __syncthreads();
                const double mp = dot_vec<R>(Arow, V0);
                V1[i] = mp;
                __syncthreads();
                const double w = xi - dot_vec<R>(Jr, V1);
                const double bi = rs[i];
                const double y = dot_vec<R>(Omp, V1) + bi;
                const double cm = dot_vec<R>(Crow, V1);
                wtab[(size_t)t * R + i] = w;
                V0[i] = y;
                __syncthreads();
                const double mfn = dot_vec<R>(Omf, V0);
                sum_xw = fma(bi, mp, fma(bi - cm, mfn, sum_xw));   // quad_t = s_t - sum_i (b_i m_p,i + u_i m_f,i)
                sum_ldz += ldz_cur;
                ssum += rs[R];
                nsum += nt;
                ldsum += rs[R + 2];
                xi = mfn;
                continue;
            }
            const bool computed = need_cov;
            double Omf_used[R];
            if (need_cov) {  // wave-uniform
#pragma unroll
                for (int j = 0; j < R; ++j) { Omf_used[j] = Omf[j]; Z[j] = Omf[j] + Phi[j]; }
                const double detM = gj_inverse<R>(Z, X, i);
                ldz_cur = -log(detM);                      // log det Z
                mm_rows<R>(Jr, Z, PSI);                    // J = Z Psi
                __syncthreads();
                store_row<R>(X, i, Jr);
                __syncthreads();
                double tmp[R];
                mm_rows<R>(tmp, PsiT, X);                  // Psi' J
#pragma unroll
                for (int j = 0; j < R; ++j) Omp[j] = Qi[j] - tmp[j];
                ++e;
                {   // (groups past the end of the batch duplicate replicate B-1: same values, same address)
                    double* zt = ZJ + ((size_t)e * 2 + 0) * R * R + i * R;
                    double* jt = ZJ + ((size_t)e * 2 + 1) * R * R + i * R;
#pragma unroll
                    for (int j = 0; j < R; ++j) { zt[j] = Z[j]; jt[j] = Jr[j]; }
                }
                if (lane == 0) cmask[t >> 5] |= 1u << (t & 31);
            }
            // mean recursion
            V0[i] = xi;
            __syncthreads();
            const double w = dot_vec<R>(Z, V0);
            sum_xw = fma(xi, w, sum_xw);
            sum_ldz += ldz_cur;
            wtab[(size_t)t * R + i] = w;
            V1[i] = w;
            __syncthreads();
            const double xip = dot_vec<R>(PsiT, V1);
            const double* rs = ring + s * (R + 3);
            xi = xip + rs[i];
            ssum += rs[R];
            const double nt = rs[R + 1];
            nsum += nt;
            ldsum += rs[R + 2];
            const bool full = (nt == (double)N);
            double Omf_new[R];
            if (full) {
#pragma unroll
                for (int j = 0; j < R; ++j) Omf_new[j] = Omp[j] + Cf[j];
            } else {  // row i of the packed C_t of this period (fetched one period ahead: ctn)
#pragma unroll
                for (int j = 0; j < R; ++j) Omf_new[j] = Omp[j] + ctn[j];
            }
            if (a.Ct) {   // C_{t+1} for the next period: in flight during the rest of this step and the next inversion
                const int tn = t + 1 < T ? t + 1 : T - 1;
                const double* ct = a.Ct + ((size_t)b * T + tn) * NPc;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const int hi = i > j ? i : j, lo = i > j ? j : i;
                    ctn[j] = hi < Rc ? ct[hi * (hi + 1) / 2 + lo] : 0.0;
                }
            }
            if (computed) {
                bool same = full;
#pragma unroll
                for (int j = 0; j < R; ++j) same = same && close_enough(Omf_new[j], Omf_used[j]);
                need_cov = !__all(same);
            } else {
                need_cov = __any(!full);
            }
#pragma unroll
            for (int j = 0; j < R; ++j) Omf[j] = Omf_new[j];
        }
        __syncthreads();
        if (c + 1 < nchunks) commit_fwd((c + 1) & 1);
        if (c + 2 < nchunks) issue_fwd(c + 2);
        __syncthreads();
    }

    // ---------------- terminal: P_T = Om_f^-1, f_T = P_T xi, log-likelihood ------------------------
    bool em_apply = true;
    double Ps[R];
#pragma unroll
    for (int j = 0; j < R; ++j) Ps[j] = Omf[j];
    double detOmT = 1.0, fs = xi;                          // COV: P_T = P_f, f_T = m_f as they stand
    if constexpr (!COV) {
        detOmT = gj_inverse<R>(Ps, X, i);
        __syncthreads();
        V0[i] = xi;
        __syncthreads();
        fs = dot_vec<R>(Ps, V0);
    }
    {
        // lane part of mu0'P0^-1mu0 - xi_T'f_T - sum xi'w   (COV: of -sum (b'm_p + u'm_f))
        const double part = COV ? -sum_xw : q0_part - xi * fs - sum_xw;
        __syncthreads();
        V1[i] = part;
        __syncthreads();
        double qd = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) qd += V1[k];
        const double LD = COV ? sum_ldz : log(detOmT) + log(detP0) + (double)T * log(detQ) - sum_ldz;
        const double ll = -0.5 * (nsum * kLog2Pi + ldsum + LD + ssum + qd);
        if (live && i == 0) {
            a.loglik[b] = ll;
            if (a.ncov) a.ncov[b] = e + 1;
        }
        // EM bookkeeping (oracle/kalman_oracle.py em()): record ll_k; stop WITHOUT applying this
        // M-step when the relative improvement over ll_{k-1} is below tol.
        if (a.active) {
            const bool was = a.k == 0 ? true : (a.active[b] != 0);
            bool go = was;
            if (was && a.k >= 1 && a.tol > 0.0) {
                const double llp = a.ll_path[(size_t)b * a.max_iter + a.k - 1];
                go = !((ll - llp) / (0.5 * (fabs(ll) + fabs(llp))) < a.tol);
            }
            em_apply = go;
            if (live && i == 0) {
                if (was) { a.ll_path[(size_t)b * a.max_iter + a.k] = ll; a.iters[b] = a.k + 1; }
                a.active[b] = go ? 1 : 0;
            }
        }
    }

    // ---------------- backward sweep -------------------------------------------------------------
    const int npr = r * (r + 1) / 2;
    // a.rl > 0: only the first rl state components are factors of the observation equation; components rl..r-1 of
    // the output layout are padding for the loadings step (mean 0, identity covariance, like padded factors)
    const int rl = a.rl > 0 ? a.rl : R;
    auto emit = [&](int trow, const double (&P)[R], double f) {   // smoothed moments of period trow+1
        if (!live || i >= r) return;
        a.f_smooth[((size_t)b * T + trow) * r + i] = i < rl ? f : 0.0;
        if (a.P_smooth) {
            double* po = a.P_smooth + ((size_t)b * T + trow) * npr + i * (i + 1) / 2;
#pragma unroll
            for (int j = 0; j < R; ++j)
                if (j <= i) po[j] = i < rl ? P[j] : (j == i ? 1.0 : 0.0);
        }
    };
    emit(T - 1, Ps, fs);

    // gathered f_s (period t+1) for the products below
    double fvec[R];
    __syncthreads();
    V0[i] = fs;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < R; ++k) fvec[k] = V0[k];

    const bool em = a.S11 != nullptr;
    double S11[R], S10[R], termT[R], U[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        termT[j] = fma(fs, fvec[j], Ps[j]);                // E[f_T f_T'] row i
        S11[j] = termT[j];
        S10[j] = 0.0;
        U[j] = 0.0;
    }

    double pw[CH];
    auto issue_bwd = [&](int c) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            int t = c * CH + s;
            t = t < T ? t : T - 1;
            pw[s] = wtab[(size_t)t * R + i];
        }
    };
    auto commit_bwd = [&](int slot) {
        double* ring = RING + slot * CH * (R + 3);
#pragma unroll
        for (int s = 0; s < CH; ++s) ring[s * (R + 3) + i] = pw[s];
    };
    // Z/J of the current entry; the next older entry is prefetched into Zn/Jn
    double Zn[R], Jn[R];
    auto load_entry = [&](int ee, double (&zz)[R], double (&jj)[R]) {
        const int ec = ee < 0 ? 0 : ee;
        const double* zt = ZJ + ((size_t)ec * 2 + 0) * R * R + i * R;
        const double* jt = ZJ + ((size_t)ec * 2 + 1) * R * R + i * R;
#pragma unroll
        for (int j = 0; j < R; ++j) { zz[j] = zt[j]; jj[j] = jt[j]; }
    };
    // after the forward sweep Z, Jr hold entry e (the last one made) -- still in registers
    int e_cur = e;
    load_entry(e_cur - 1, Zn, Jn);
    __syncthreads();
    store_row<R>(JS, i, Jr);
    bool need_b = true;

    issue_bwd(nchunks - 1);
    __syncthreads();
    commit_bwd((nchunks - 1) & 1);
    if (nchunks > 1) issue_bwd(nchunks - 2);
    __syncthreads();

    for (int c = nchunks - 1; c >= 0; --c) {
        const double* ring = RING + (c & 1) * CH * (R + 3);
        const int smax = (T - c * CH) < CH ? (T - c * CH) : CH;
        for (int s = smax - 1; s >= 0; --s) {
            const int t = c * CH + s;          // step t: from period t+1 to period t (t = 0: initial state)
            // entry of step t: e_cur is the entry of the step processed before (t+1); it changes iff a
            // new pair was made at step t+1
            bool changed = false;
            if (t + 1 < T && ((cmask[(t + 1) >> 5] >> ((t + 1) & 31)) & 1u)) {
                --e_cur;
#pragma unroll
                for (int j = 0; j < R; ++j) { Z[j] = Zn[j]; Jr[j] = Jn[j]; }
                load_entry(e_cur - 1, Zn, Jn);
                __syncthreads();
                store_row<R>(JS, i, Jr);
                __syncthreads();
                changed = true;
            }
            if (need_b || changed) {  // wave-uniform
                __syncthreads();
                mm_rowsT<R>(U, Ps, JS);                    // U = P_s J'  = Cov(f_{t+1}, f_t | X)
                store_row<R>(X, i, U);
                __syncthreads();
                double tmp[R];
                mm_rows<R>(tmp, Jr, X);                    // J U
                bool same = true;
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const double pn_ = Z[j] + tmp[j];
                    same = same && close_enough(pn_, Ps[j]);
                    Ps[j] = pn_;
                }
                need_b = !__all(same);
            }
            const double fnew = ring[s * (R + 3) + i] + dot_vec<R>(Jr, fvec);   // w_t + J f_{t+1}
            if (em) {
#pragma unroll
                for (int j = 0; j < R; ++j) S10[j] += U[j];          // + f_{t+1} f_t' added below
            }
            const double fprev_i = fs;                     // f_{t+1}[i]
            fs = fnew;
            __syncthreads();
            V0[i] = fs;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < R; ++k) fvec[k] = V0[k];
            if (em) {
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    S10[j] = fma(fprev_i, fvec[j], S10[j]);
                    if (t > 0) S11[j] += fma(fs, fvec[j], Ps[j]);
                }
            }
            if (t > 0) emit(t - 1, Ps, fs);
        }
        __syncthreads();
        if (c - 1 >= 0) commit_bwd((c - 1) & 1);
        if (c - 2 >= 0) issue_bwd(c - 2);
        __syncthreads();
    }
    // now fs / Ps / fvec are the smoothed moments of the initial state f_0
    if (em) {
        const size_t o = (size_t)b * R * R + (size_t)i * R;
        double S00[R];
#pragma unroll
        for (int j = 0; j < R; ++j) S00[j] = S11[j] - termT[j] + fma(fs, fvec[j], Ps[j]);
        if (live) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (a.rl == 0) a.S11[o + j] = S11[j];      // (rl > 0: written below in the loadings layout)
                a.S10[o + j] = S10[j];
                a.S00[o + j] = S00[j];
                a.P0s[o + j] = Ps[j];
            }
            a.f0s[(size_t)b * R + i] = fs;
        }
        if (a.A_out) {
            // A = S10 S00^-1 ;  Q = sym(S11 - A S10') / T ;  mu0 = f_0|T ;  P0 = sym(P_0|T) ;  S11^-1
            double inv[R], An[R], tmp[R], Qn[R], P0n[R];
            const int kd = a.kdim;                         // > 0: companion state (f_t, .., f_{t-p+1}) of width kd
#pragma unroll
            for (int j = 0; j < R; ++j) inv[j] = S00[j];
            double S10m[R];
#pragma unroll
            for (int j = 0; j < R; ++j) S10m[j] = S10[j];
            if (kd > 0 && a.ka > 0) {   // VAR(p) inside a wider state: A = S10[:, :ka] S00[:ka, :ka]^-1, zero beyond
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    if (i >= a.ka || j >= a.ka) inv[j] = (i == j) ? 1.0 : 0.0;
                    if (j >= a.ka) S10m[j] = 0.0;
                }
            }
            (void)gj_inverse<R>(inv, X, i);
            __syncthreads();
            store_row<R>(X, i, inv);
            __syncthreads();
            mm_rows<R>(An, S10m, X);                       // A row i
            __syncthreads();
            store_row<R>(X, i, S10m);
            __syncthreads();
            mm_rowsT<R>(tmp, An, X);                       // (A S10')[i][:]
#pragma unroll
            for (int j = 0; j < R; ++j) Qn[j] = (S11[j] - tmp[j]) / (double)T;
            __syncthreads();
            store_row<R>(X, i, Qn);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < R; ++j) Qn[j] = 0.5 * (Qn[j] + X[j * R + i]);
            if (kd > 0) {   // only [A_1 .. A_p] and the innovation covariance of f_t are free (dfm_functions.ipynb:477-492)
                const int rb = a.kb > 0 ? a.kb : rl;               // block size of the companion state
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    if (i >= rb && i < kd) An[j] = (j == i - rb) ? 1.0 : 0.0;
                    if ((i >= rb && i < kd) || (j >= rb && j < kd)) Qn[j] = 0.0;
                }
            }
            __syncthreads();
            store_row<R>(X, i, Ps);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < R; ++j) P0n[j] = 0.5 * (Ps[j] + X[j * R + i]);
#pragma unroll
            for (int j = 0; j < R; ++j) inv[j] = S11[j];
            if (a.rl > 0) {   // loadings step sees the factor block only: S11 <- [S11[:rl,:rl] 0; 0 T I] in the Rc layout
#pragma unroll
                for (int j = 0; j < R; ++j)
                    if (i >= rl || j >= rl) inv[j] = (i == j) ? (double)T : 0.0;
                if (live && i < Rc) {
#pragma unroll
                    for (int j = 0; j < R; ++j)
                        if (j < Rc) a.S11[(size_t)b * Rc * Rc + (size_t)i * Rc + j] = inv[j];
                }
            }
            (void)gj_inverse<R>(inv, X, i);
            if (live) {
                if (a.rl > 0) {
                    if (i < Rc) {
#pragma unroll
                        for (int j = 0; j < R; ++j)
                            if (j < Rc) a.S11inv[(size_t)b * Rc * Rc + (size_t)i * Rc + j] = inv[j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < R; ++j) a.S11inv[o + j] = inv[j];
                }
                if (em_apply) {
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        a.A_out[o + j] = An[j];
                        a.Q_out[o + j] = Qn[j];
                        a.P0_out[o + j] = P0n[j];
                    }
                    a.mu0_out[(size_t)b * R + i] = fs;
                }
            }
        }
    }
}

template <int R, bool COV>
static hipError_t launch_rec(const RecursionArgs& a, hipStream_t s) {
    using LY = RecLayout<R>;
    const int grid = (a.B + LY::GPW - 1) / LY::GPW;
    const size_t lds = LY::lds_bytes(a.T);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&recursion_kernel<R, COV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((recursion_kernel<R, COV>), dim3(grid), dim3(64), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_recursion(int Rpad, const RecursionArgs& a, hipStream_t s) {
    // Rp = 8, information form: the time-chunked recursion, then the sequential kernel for the replicates whose chunk boundaries
    // did not agree (normally none: the launch then costs its blocks' early exit)
    if (a.wave && recursion_chunk_supported(Rpad, a)) {
        hipError_t e = launch_recursion_chunk(a, s);
        if (e != hipSuccess) return e;
        if (a.chunk_obs_ready) {
            e = launch_chunk_unbridge(a, s);
            if (e != hipSuccess) return e;
        }
        RecursionArgs f = a;
        f.only_if = a.chunk_fail;
        e = launch_recursion_wave8_fallback(f, s);
        note_kernel("recursion_chunk_kernel");
        return e;
    }
    if (a.wave && recursion_comp_supported(Rpad, a)) return launch_recursion_comp(Rpad, a, s);
    if (a.wave && recursion_mbf16_supported(Rpad, a)) return launch_recursion_mbf16(a, s);
    if (a.wave && recursion_tile_supported(Rpad, a)) return launch_recursion_tile(a, s);
    if (a.wave && recursion_wave_supported(Rpad, a)) return launch_recursion_wave(a, s, Rpad);
    if (a.cov) {
        switch (Rpad) {
            case 2: return launch_rec<2, true>(a, s);
            case 4: return launch_rec<4, true>(a, s);
            case 8: return launch_rec<8, true>(a, s);
            case 16: return launch_rec<16, true>(a, s);
            case 32: return launch_rec<32, true>(a, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (Rpad) {
        case 2: return launch_rec<2, false>(a, s);
        case 4: return launch_rec<4, false>(a, s);
        case 8: return launch_rec<8, false>(a, s);
        case 16: return launch_rec<16, false>(a, s);
        case 32: return launch_rec<32, false>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dfm
