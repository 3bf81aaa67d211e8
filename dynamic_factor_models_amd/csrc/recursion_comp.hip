// recursion_comp.hip -- the smoother pass of a COMPANION state s_t = (f_t, .., f_{t-m+1}) (blocks of 4: k = 4 m <= 32 state components;
// VAR(p) factor dynamics, dfm_functions.ipynb:477-492, and AR idiosyncratic terms by quasi-differencing, :305-311 / :405-412, whose
// collapsed observation loads on every block) in INFORMATION form, as the block elimination of the block-banded posterior precision:
// ONE WAVE per replicate, every matrix as 16 x 16 accumulator tiles of `v_mfma_f64_16x16x4`, and NO k x k inversion per period -- only
// the 4 x 4 pivot of the block that leaves the state.  (scripts/dbg/r06/companion_emul.py is the NumPy model,
// tests/test_companion_model_cpu.py pins it to oracle/varp_oracle.py and oracle/ar_oracle.py at 1e-14.)
//
//   filter    posterior of s_t: (Om_f, xi_f).  Joint precision of (f_{t+1}, s_t) after f_{t+1} = Phi s_t + eta, eta ~ N(0, Q):
//             [[Qi, -Qi Phi], [-Phi' Qi, M]],  M = Om_f + Phi' Qi Phi.  s_{t+1} = (f_{t+1}, a) keeps a = s_t[:k-4] and drops the oldest
//             block d = s_t[k-4:]: a Schur complement on M_dd,
//                 Om_p = Base - U M_dd^-1 U',  Base = [[Qi, -Qi Phi_a], [., M_aa]] (= M moved one block down-right under a constant
//                 block row / column),  U = [-Qi Phi_d; M_ad],     xi_p = [0; xi_a] - U M_dd^-1 xi_d;
//             the observation then ADDS: Om_f' = Om_p + C_{t+1}, xi_f' = xi_p + b_{t+1} -- whatever its rank.
//   smoother  d | s_{t+1}, X ~ N(g + G s_{t+1}, M_dd^-1),  G = -M_dd^-1 U',  g = M_dd^-1 xi_d: the smoothed moments of s_t are those of
//             s_{t+1} moved one block up-left plus one new block row / column (G V, G V G' + M_dd^-1);
//             Cov(s_{t+1}, s_t | X) = [V[:, 4:], V G'] for the EM sums.
//   likelihood  -2 ll = sum_t (n_t log 2 pi + ld_t + s_t) + mu0' P0^-1 mu0 + log det P0 + T log det Q + sum_t log det M_dd,t
//               + log det Om_f,T - sum_t xi_d' M_dd^-1 xi_d - xi_T' Om_f,T^-1 xi_T     (the eliminations' pivots and the last marginal).
// Only P0^-1 at the start and Om_f,T^-1 at the end are k x k inversions: once per replicate, by an in-LDS Gauss-Jordan of the wave.
//
// Layout (lane l: k4 = l / 16, c = l % 16): tile (ti, tj) register v = M[16 ti + k4 + 4 v][16 tj + c].  A block is 4 rows = ONE register
// index and 4 columns = 4 lanes, so "move one block" is register renaming + a DPP row shift (row_shr:4 / row_shl:12 across tile
// columns); the rank-4 terms are one matrix instruction per tile (k = 4), with a 4 x 16 row strip (M_dd^-1 U') straight from the
// previous instruction as the A operand (same trick as recursion_mbf16.hip).  The one transposition a step needs -- a 4 x k strip from
// "row" to "column" form -- goes through LDS (one wave: served in order, no barrier).
// recursion_mbf16.hip (rank-4 covariance update + Bryson-Frazier smoother; generic transition matrix) stays the route of 16-wide
// companion states whose blocks are narrower than 4 (r <= 3); recursion_wave_kernel<32, COV> that of 32-wide ones.
// Reference counterpart: none (dfm_functions.ipynb:21-23 declares `Parametric` only).
#include <stdlib.h>

#include "dfm_kernels.h"
#include "dfm_smallmat.h"
#include "dfm_grid.h"

namespace dfm {

namespace {

typedef double c16 __attribute__((ext_vector_type(4)));
constexpr double kLog2PiC4 = 1.8378770664093454835606594728112;
constexpr int kDppShr4 = 0x114, kDppShl12 = 0x10C, kDppShl4 = 0x104, kDppShr12 = 0x11C;   // row_shr:n = lane l <- l - n, row_shl:n = lane l <- l + n (0 outside the row)

__device__ __forceinline__ c16 cm1(double a, double b, c16 acc) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); }
__device__ __forceinline__ double crd(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double crow16(double v) { v += xor_lane<1>(v); v += xor_lane<2>(v); v += xor_lane<4>(v); v += xor_lane<8>(v); return v; }
__device__ __forceinline__ double ccol4(double v) { v += xor_lane<16>(v); v += xor_lane<32>(v); return v; }
// (the operands are PRVALUES -- unary plus: `q == 0 ? x[0] : x[1]` on lvalues is an lvalue, clang emits a select of ADDRESSES and one load,
// the array then stays in scratch memory behind a dynamic offset, and every pick is a scratch round trip behind an s_waitcnt vmcnt(0))
__device__ __forceinline__ double cpick(const double (&x)[4], int q) {
    const double a0 = +x[0], a1 = +x[1], a2 = +x[2], a3 = +x[3];
    return q == 0 ? +a0 : q == 1 ? +a1 : q == 2 ? +a2 : +a3;
}
__device__ __forceinline__ double cpick44(const double (&W)[4][4], int i, int j) {   // W[i][j], i and j run-time (selects, no indexing)
    const double r0 = cpick(W[0], j), r1 = cpick(W[1], j), r2 = cpick(W[2], j), r3 = cpick(W[3], j);
    return i == 0 ? +r0 : i == 1 ? +r1 : i == 2 ? +r2 : +r3;
}
__device__ __forceinline__ void csqrt(double x, double& s, double& r) {   // sqrt and 1 / sqrt of a positive normal x (recursion_mbf16.hip)
    r = __builtin_amdgcn_rsq(x);
    r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
    r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
    s = x * r;
    s = fma(0.5 * r, fma(-s, s, x), s);
}
// inverse of a symmetric positive definite 4 x 4 matrix (wave-uniform values, every lane the same): Cholesky, triangular inverse,
// Li' Li.  Returns det; ok = false when a pivot is not positive.
__device__ __forceinline__ double inv4(const double (&S)[4][4], double (&Inv)[4][4], bool& ok) {
    double L[4][4], iL[4], Li[4][4];
    double det = 1.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double d = S[j][j];
#pragma unroll
        for (int q = 0; q < j; ++q) d = fma(-L[j][q], L[j][q], d);
        ok = ok && (d > 0.0);
        det *= (d > 0.0 ? d : 1.0);
        double ljj;
        csqrt(d > 0.0 ? d : 1.0, ljj, iL[j]);
        L[j][j] = ljj;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            double s = S[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q) s = fma(-L[i][q], L[j][q], s);
            L[i][j] = s * iL[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                 // Li = L^-1 (lower), column by column
        Li[j][j] = iL[j];
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            double s = 0.0;
#pragma unroll
            for (int q = j; q < i; ++q) s = fma(-L[i][q], Li[q][j], s);
            Li[i][j] = s * iL[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {                            // S^-1 = Li' Li
            double s = 0.0;
#pragma unroll
            for (int q = i; q < 4; ++q) s = fma(Li[q][i], Li[q][j], s);
            Inv[i][j] = s; Inv[j][i] = s;
        }
    return det;
}

constexpr int kInvLd = 33;                                        // row stride (doubles) of the in-LDS k x k scratch matrix
// In-place inverse of the symmetric positive definite k x k matrix S (LDS, row stride kInvLd) by Gauss-Jordan without pivoting, the
// wave's 64 lanes over the k^2 elements; `det` collects the pivots.  Once or twice per replicate: speed is irrelevant.
__device__ __forceinline__ bool lds_spd_inverse(double* S, int k, int lane, LogProd& det) {
    bool ok = true;
    for (int p = 0; p < k; ++p) {
        const double piv = S[p * kInvLd + p];
        ok = ok && (piv > 0.0);
        det.mul(piv > 0.0 ? piv : 1.0);
        const double ip = 1.0 / piv;
        double nv[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int e = lane + 64 * n;
            nv[n] = 0.0;
            if (e < k * k) {
                const int i = e / k, j = e - i * k;
                const double rp = S[p * kInvLd + j], cp = S[i * kInvLd + p], cur = S[i * kInvLd + j];
                nv[n] = (i == p) ? (j == p ? ip : rp * ip) : (j == p ? -cp * ip : fma(-cp * ip, rp, cur));
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int e = lane + 64 * n;
            if (e < k * k) { const int i = e / k, j = e - i * k; S[i * kInvLd + j] = nv[n]; }
        }
        wave_lds_sync();
    }
    return ok;
}

}  // namespace

// NT = 1: state padded to 16, NT = 2: to 32.  RCN: the collapse kernels' observation layout -- true: 4 wide (VAR(p): the observation
// loads on f_t alone), false: as wide as the padded state (AR idiosyncratic terms)
template <int NT, bool RCN>
__global__ __launch_bounds__(64) void recursion_comp_kernel(RecursionArgs a) {
    constexpr int R = 16 * NT, RR = R * R, kSlot = 2 * RR;
    constexpr int Rc = RCN ? 4 : R, NPc = Rc * (Rc + 1) / 2;
    __shared__ double sPhi[4 * 32], sQP[4 * 32], sXi[2][32], sM[2][32], sG[4 * 32], sGV[4 * 32], sS[32 * kInvLd], sB[32];
    const int lane = threadIdx.x, b = blockIdx.x;
    const int k4 = lane >> 4, c = lane & 15;
    const int T = a.T, N = a.N;
    const int k = a.kdim, ka = k - 4, db = k / 4 - 1, td = db >> 2, vd = db & 3;
    const double* bcol = a.bcol + (size_t)b * T * Rc;
    const double* scol = a.scol + (size_t)b * T;
    const int* nobs = a.nobs + (size_t)b * T;
    const double* ldrow = a.ldrow + (size_t)b * T;
    const bool haveCt = a.Ct != nullptr;
    const double* Ctb = (haveCt ? a.Ct : a.Cfull) + (haveCt ? (size_t)b * T * NPc : 0);
    const double* Cfb = a.Cfull + (size_t)b * Rc * Rc;
    const double ldfull = a.ldfull[b];
    double* slot0 = a.ZJtab + (size_t)b * (T + 1) * kSlot;
    const bool em = a.S11 != nullptr;
    const c16 zero = {0.0, 0.0, 0.0, 0.0};
    const double* Ag = a.A + (size_t)b * RR;
    const double* Qg = a.Q + (size_t)b * RR;
    bool okall = true;
    // element (i, j) of this lane in tile (ti, tj), register v
    auto ri = [&](int ti, int v) { return 16 * ti + k4 + 4 * v; };
    auto cj = [&](int tj) { return 16 * tj + c; };

    // ---- constants ------------------------------------------------------------------------------------------------------------
    double Qi[4][4], ldQ;
    {
        double Q4[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) Q4[i][j] = 0.5 * (Qg[i * R + j] + Qg[j * R + i]);
        const double dq = inv4(Q4, Qi, okall);
        ldQ = log(dq);
    }
    for (int e = lane; e < 4 * 32; e += 64) {                      // Phi [4][32] (the free rows of the transition), QP = Qi Phi
        const int aa = e >> 5, j = e & 31;
        sPhi[e] = (j < k) ? Ag[aa * R + j] : 0.0;
    }
    wave_lds_sync();
    for (int e = lane; e < 4 * 32; e += 64) {
        const int aa = e >> 5, j = e & 31;
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) s = fma(cpick44(Qi, aa, q), sPhi[q * 32 + j], s);
        sQP[e] = s;
    }
    wave_lds_sync();
    c16 PQP[NT][NT];                                               // Phi' Qi Phi
    double cbr[NT], cbc[NT][4];                                    // Base's constant block row / block column (see the head of the file)
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = ri(ti, v), j = cj(tj);
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) s = fma(sPhi[q * 32 + i], sQP[q * 32 + j], s);
                PQP[ti][tj][v] = (i < k && j < k) ? s : 0.0;
            }
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
        const int j = cj(tj);
        cbr[tj] = j < 4 ? cpick44(Qi, k4, j & 3) : (j < k ? -sQP[k4 * 32 + j - 4] : 0.0);
    }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = ri(ti, v);
            cbc[ti][v] = (c < 4 && i >= 4 && i < k) ? -sQP[c * 32 + i - 4] : 0.0;
        }
    const double uc0 = c < 4 ? -sQP[c * 32 + 4 * db + k4] : 0.0;   // U[c][k4] of the constant block: -(Qi Phi)[:, d]

    // ---- Om_0 = P0^-1, xi_0 = Om_0 mu0 ----------------------------------------------------------------------------------------
    LogProd detP0;
    {
        const double* Pg = a.P0 + (size_t)b * RR;
        for (int e = lane; e < k * k; e += 64) { const int i = e / k, j = e - i * k; sS[i * kInvLd + j] = 0.5 * (Pg[i * R + j] + Pg[j * R + i]); }
        wave_lds_sync();
        okall = lds_spd_inverse(sS, k, lane, detP0) && okall;
    }
    c16 Om[NT][NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = ri(ti, v), j = cj(tj);
                Om[ti][tj][v] = (i < k && j < k) ? sS[i * kInvLd + j] : 0.0;
            }
    double q0;
    {
        const double* mug = a.mu0 + (size_t)b * R;
        double xi_l = 0.0, qq = 0.0;
        if (lane < k) {
            for (int j = 0; j < k; ++j) xi_l = fma(sS[lane * kInvLd + j], mug[j], xi_l);
            qq = xi_l * mug[lane];
        }
        if (lane < 32) sXi[0][lane] = lane < k ? xi_l : 0.0;
        qq = crow16(qq); qq = ccol4(qq);
        q0 = qq;                                                   // mu0' P0^-1 mu0
    }
    wave_lds_sync();
    // the replicate's full collapsed observation (periods without a missing cell)
    c16 Cfl[NT][NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = ri(ti, v), j = cj(tj);
                Cfl[ti][tj][v] = (i < Rc && j < Rc && i < k && j < k) ? Cfb[i * Rc + j] : 0.0;
            }

    // M_dd^-1 [c][k4] in the lanes of the first four columns (the A operand of "M_dd^-1 padded")
    auto wsel = [&](const double (&W)[4][4]) -> double {
        const double r0 = cpick(W[0], k4), r1 = cpick(W[1], k4), r2 = cpick(W[2], k4), r3 = cpick(W[3], k4);
        const double x = c == 0 ? +r0 : c == 1 ? +r1 : c == 2 ? +r2 : +r3;
        return c < 4 ? x : 0.0;
    };

    // =================================================== forward ===========================================================
    LogProd detM;
    double acc = 0.0;                                              // sum_t (n log 2 pi + ld + s) - sum_t xi_d' M_dd^-1 xi_d
    int cur = 0;
    // the period's collapsed observation a step ahead (vector loads: see recursion_mbf16.hip)
    struct ObsIn { int nt; double s, ld, bl; c16 C[NT][NT]; };
    auto fetch_obs = [&](int t) {
        ObsIn o;
        t = t < T ? t : T - 1;
        o.nt = nobs[t]; o.s = scol[t]; o.ld = ldrow[t];
        o.bl = (lane < Rc && lane < k) ? bcol[(size_t)t * Rc + lane] : 0.0;   // b_t[lane]
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = ri(ti, v), j = cj(tj);
                    const int hi = i > j ? i : j, lo = i > j ? j : i;
                    const bool in = haveCt && hi < Rc && hi < k && (RCN ? (ti == 0 && tj == 0 && v == 0) : true);
                    o.C[ti][tj][v] = in ? Ctb[(size_t)t * NPc + hi * (hi + 1) / 2 + lo] : 0.0;
                }
        return o;
    };
    ObsIn nxt = fetch_obs(0);
    for (int t = 0; t < T; ++t) {
        const ObsIn ob = nxt;
        nxt = fetch_obs(t + 1);
        const double* xi = sXi[cur];
        double* xin = sXi[cur ^ 1];
        // M = Om_f + Phi' Qi Phi; its pivot block M_dd and the strip of the d rows
        c16 M[NT][NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                for (int v = 0; v < 4; ++v) M[ti][tj][v] = Om[ti][tj][v] + PQP[ti][tj][v];
        double drow[NT];                                           // M[4 db + k4][16 t + c]
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            double x = M[0][tj][0];
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int v = 0; v < 4; ++v) x = (ti == td && v == vd) ? M[ti][tj][v] : x;
            drow[tj] = x;
        }
        double Mdd[4][4], Mdi[4][4], xd[4], z[4];
        {
            double dsel = drow[0];
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) dsel = tj == td ? drow[tj] : dsel;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    const double pv = crd(dsel, 16 * i + 4 * vd + j);
                    Mdd[i][j] = pv; Mdd[j][i] = pv;
                }
        }
#ifdef DFM_DIAG
        if (a.tile_nc & 2) {                                       // DFM_COMP_ABL bit 1 (timing only, WRONG results): no 4 x 4 inversion
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mdi[i][j] = i == j ? 1.0 / Mdd[i][i] : 0.0;
        } else
#endif
        detM.mul(inv4(Mdd, Mdi, okall));
#pragma unroll
        for (int q = 0; q < 4; ++q) xd[q] = xi[4 * db + q];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) s = fma(Mdi[i][q], xd[q], s);
            z[i] = s;
        }
        acc += (double)ob.nt * kLog2PiC4 + ((ob.nt == N) ? ldfull : ob.ld) + ob.s - (xd[0] * z[0] + xd[1] * z[1] + xd[2] * z[2] + xd[3] * z[3]);
        // U in column-operand form: ucol[t] = U[16 t + c][k4]  (the d strip moved 4 columns right under the constant block)
        double ucol[NT];
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const double x = dpp_mov<kDppShr4>(drow[tj]);
            const double y = tj > 0 ? dpp_mov<kDppShl12>(drow[tj > 0 ? tj - 1 : 0]) : uc0;
            const double u = c >= 4 ? x : y;
            ucol[tj] = cj(tj) < k ? u : 0.0;
        }
        // R = M_dd^-1 U' (row strip), G = -R for the smoother
        const double ws = wsel(Mdi);
        double rrow[NT];
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) rrow[tj] = cm1(ws, ucol[tj], zero)[0];
        // Om_p = Base - (U M_dd^-1) U';  Om_f' = Om_p + C
        const bool full = ob.nt == N || !haveCt;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                c16 base;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double val;
                    if (ti == 0 && v == 0) val = cbr[tj];
                    else {
                        const double s0 = v > 0 ? M[ti][tj][v > 0 ? v - 1 : 0] : M[ti > 0 ? ti - 1 : 0][tj][3];
                        const double s1 = tj > 0 ? (v > 0 ? M[ti][tj > 0 ? tj - 1 : 0][v > 0 ? v - 1 : 0] : M[ti > 0 ? ti - 1 : 0][tj > 0 ? tj - 1 : 0][3]) : 0.0;
                        const double x = dpp_mov<kDppShr4>(s0);
                        const double y = tj > 0 ? dpp_mov<kDppShl12>(s1) : cbc[ti][v];
                        val = c >= 4 ? x : y;
                    }
                    base[v] = (ri(ti, v) < k && cj(tj) < k) ? val : 0.0;
                }
                const c16 op = cm1(-rrow[ti], ucol[tj], base);
#pragma unroll
                for (int v = 0; v < 4; ++v) Om[ti][tj][v] = op[v] + (full ? Cfl[ti][tj][v] : ob.C[ti][tj][v]);
            }
        // xi_p = [0; xi_a] - U z;  xi_f' = xi_p + b
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const double uz = ccol4(ucol[tj] * cpick(z, k4));
            const int i = cj(tj);
            if (k4 == 0) xin[i] = (i < k) ? ((i >= 4 ? xi[i - 4] : 0.0) - uz) : 0.0;
        }
        wave_lds_sync();
        if (lane < Rc && lane < k) xin[lane] += ob.bl;
        // what the backward sweep reads back: the strip G = -R, g = z, M_dd^-1
        {
            double* sl = slot0 + (size_t)t * kSlot;
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) sl[64 * tj + lane] = -rrow[tj];
            if (c < 4) sl[128 + 4 * c + k4] = ws;                 // M_dd^-1 row-major
            if (c == 4) sl[144 + k4] = cpick(z, k4);
        }
        cur ^= 1;
        wave_lds_sync();
    }
    // ---- the last marginal: V_T = Om_f,T^-1, m_T = V_T xi_T -------------------------------------------------------------------
    LogProd detOT;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = ri(ti, v), j = cj(tj);
                if (i < k && j < k) sS[i * kInvLd + j] = Om[ti][tj][v];
            }
    wave_lds_sync();
    okall = lds_spd_inverse(sS, k, lane, detOT) && okall;
    c16 V[NT][NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = ri(ti, v), j = cj(tj);
                V[ti][tj][v] = (i < k && j < k) ? 0.5 * (sS[i * kInvLd + j] + sS[j * kInvLd + i]) : 0.0;
            }
    int mc = 0;
    double qT;
    {
        const double* xi = sXi[cur];
        double ml = 0.0, qq = 0.0;
        if (lane < k) {
            for (int j = 0; j < k; ++j) ml = fma(sS[lane * kInvLd + j], xi[j], ml);
            qq = ml * xi[lane];
        }
        if (lane < 32) sM[0][lane] = lane < k ? ml : 0.0;
        qq = crow16(qq); qq = ccol4(qq);
        qT = qq;                                                   // xi_T' Om_T^-1 xi_T
    }
    wave_lds_sync();
    const double ll = okall ? -0.5 * (acc + q0 + detP0.log_value() + (double)T * ldQ + detM.log_value() + detOT.log_value() - qT) : __builtin_nan("");
    if (lane == 0) {
        a.loglik[b] = ll;
        if (a.ncov) a.ncov[b] = T;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");        // the slots are read back by this wave (other lanes)

#ifdef DFM_DIAG
    if (a.tile_nc & 1) return;                                     // DFM_COMP_ABL bit 0 (timing only): the forward sweep alone
#endif
    // =================================================== backward ==========================================================
    const int r = a.r, rl = a.rl > 0 ? a.rl : R, npr = r * (r + 1) / 2;
    c16 S11[NT][NT], S10[NT][NT], VT[NT][NT];
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) { S11[ti][tj] = zero; S10[ti][tj] = zero; VT[ti][tj] = zero; }
    double mTrow[NT][4], mTcol[NT];
    struct SlotIn { double g[NT], mdr[4], z[4]; };                // (row k4 of M_dd^-1: the only one this lane uses -- 9 values a period, not 21)
    auto fetch_slot = [&](int t) {
        SlotIn o;
        const double* sl = slot0 + (size_t)(t > 0 ? t : 0) * kSlot;
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) o.g[tj] = sl[64 * tj + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) o.mdr[q] = sl[128 + 4 * k4 + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) o.z[q] = sl[144 + q];
        return o;
    };
    SlotIn snx = fetch_slot(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        const SlotIn sc = snx;
        snx = fetch_slot(t - 1);
        const double* m = sM[mc];
        double* mn = sM[mc ^ 1];
        // ---- the smoothed moments of period t: outputs, S11
        double mrow[NT][4], mcol[NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            mcol[ti] = m[cj(ti)];
#pragma unroll
            for (int v = 0; v < 4; ++v) mrow[ti][v] = m[ri(ti, v)];
        }
        // (VAR(p): the outputs are the first block, r <= 4 -- row i = k4 of tile row 0, register 0; the other registers' rows start at 4)
#pragma unroll
        for (int ti = 0; ti < (RCN ? 1 : NT); ++ti)
#pragma unroll
            for (int v = 0; v < (RCN ? 1 : 4); ++v) {
                const int i = ri(ti, v);
#ifdef DFM_DIAG
                if (a.tile_nc & 4) continue;                       // DFM_COMP_ABL bit 2 (timing only): no output stores
#endif
                if (i < r) {
                    if (c == 0) a.f_smooth[((size_t)b * T + t) * r + i] = (i < rl && i < k) ? mrow[ti][v] : 0.0;
                    if (a.P_smooth) {
#pragma unroll
                        for (int tj = 0; tj <= ti; ++tj) {
                            const int j = cj(tj);
                            if (j <= i)
                                a.P_smooth[((size_t)b * T + t) * npr + i * (i + 1) / 2 + j] =
                                    (i < rl && j < rl && i < k) ? V[ti][tj][v] : (i == j ? 1.0 : 0.0);
                        }
                    }
                }
            }
        if (em) {
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = 0; tj < NT; ++tj)
#pragma unroll
                    for (int v = 0; v < 4; ++v) S11[ti][tj][v] += fma(mrow[ti][v], mcol[tj], V[ti][tj][v]);
            if (t == T - 1) {
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    mTcol[ti] = mcol[ti];
#pragma unroll
                    for (int tj = 0; tj < NT; ++tj) VT[ti][tj] = V[ti][tj];
#pragma unroll
                    for (int v = 0; v < 4; ++v) mTrow[ti][v] = mrow[ti][v];
                }
            }
        }
        // ---- G in column-operand form (LDS transposition): gcs[tk][s] = G[c][16 tk + 4 s + k4] for the lanes of columns 0..3
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) sG[k4 * 32 + cj(tj)] = sc.g[tj];
        wave_lds_sync();
        double gcs[NT][4];
#pragma unroll
        for (int tk = 0; tk < NT; ++tk)
#pragma unroll
            for (int s = 0; s < 4; ++s) {                          // (unconditional read of a valid row, then the select: no branch around the load)
                const double gx = sG[(c & 3) * 32 + 16 * tk + 4 * s + k4];
                gcs[tk][s] = c < 4 ? gx : 0.0;
            }
        // GV (row strip): gv[tj] = (G V)[k4][16 tj + c]
        double gv[NT];
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            c16 ac = zero;
#pragma unroll
            for (int tk = 0; tk < NT; ++tk)
#pragma unroll
                for (int s = 0; s < 4; ++s) ac = cm1(gcs[tk][s], V[tk][tj][s], ac);
            gv[tj] = ac[0];
        }
        // md = g + G m;  Vdd = M_dd^-1 + (G V) G'
        double md[4], vdd[4];                                      // vdd[q'] = Vdd[k4][q'] in every lane of row group k4
        {
            double gm = 0.0;
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) gm = fma(sc.g[tj], mcol[tj], gm);
            gm = crow16(gm);
#pragma unroll
            for (int q = 0; q < 4; ++q) md[q] = sc.z[q] + crd(gm, 16 * q);
            double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
                const double g0 = sc.g[tj], g1 = xor_lane<16>(g0), g2 = xor_lane<32>(g0), g3 = xor_lane<48>(g0);   // G[k4 ^ x][j]
                const double gq[4] = {g0, g1, g2, g3};
#pragma unroll
                for (int qp = 0; qp < 4; ++qp) part[qp] = fma(gv[tj], cpick(gq, qp ^ k4), part[qp]);
            }
#pragma unroll
            for (int qp = 0; qp < 4; ++qp) {
                vdd[qp] = crow16(part[qp]) + sc.mdr[qp];
            }
        }
        // ---- GV in column form (LDS transposition): cs0[ti][v] = GV[q'][i], cs1 = GV[q'][i + 4], q' = c - 4 vd, in the lanes of block column db
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) sGV[k4 * 32 + cj(tj)] = gv[tj];
        wave_lds_sync();
        const int qp_l = (c >> 2) == vd ? (c & 3) : -1;            // this lane's column within block column db (tile column td), or none
        // (what does not depend on the register index, outside the register loop; LDS reads unconditional at a valid address, then selected:
        // a read under a lane-divergent condition is a branch region of its own -- the loop was bound by what it issues)
        const double* gvrow = sGV + (c & 3) * 32;
        const double md_l = cpick(md, c & 3);
        double mprev_t[NT];
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const int j = cj(tj);
            const double mj4 = m[j + 4 < 32 ? j + 4 : 31];
            mprev_t[tj] = j < ka ? mj4 : ((tj == td && qp_l >= 0) ? md_l : 0.0);
        }
        // S10 += Cov(s_t, s_{t-1} | X) + m_t m_{t-1}',  Cov = [V[:, 4:], V G']
        // new state: V moved one block up-left, block row / column db from GV, block (db, db) = Vdd;  m' = (m[4:], md)
        c16 Vn[NT][NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = ri(ti, v), j = cj(tj);
                    // columns moved left by one block: Vl[i][j] = V[i][j + 4];  rows too: Vul[i][j] = V[i + 4][j + 4]
                    const double a0 = V[ti][tj][v], a1 = tj + 1 < NT ? V[ti][tj + 1 < NT ? tj + 1 : tj][v] : 0.0;
                    // (the DPP moves OUTSIDE the selects: inside a divergent branch their source lanes would be disabled and read as 0)
                    const double a0s = dpp_mov<kDppShl4>(a0), a1s = tj + 1 < NT ? dpp_mov<kDppShr12>(a1) : 0.0;
                    const double vl = c < 12 ? a0s : a1s;
                    const double b0 = v < 3 ? V[ti][tj][v < 3 ? v + 1 : 3] : (ti + 1 < NT ? V[ti + 1 < NT ? ti + 1 : ti][tj][0] : 0.0);
                    const double b1 = tj + 1 < NT ? (v < 3 ? V[ti][tj + 1 < NT ? tj + 1 : tj][v < 3 ? v + 1 : 3]
                                                           : (ti + 1 < NT ? V[ti + 1 < NT ? ti + 1 : ti][tj + 1 < NT ? tj + 1 : tj][0] : 0.0)) : 0.0;
                    const double b0s = dpp_mov<kDppShl4>(b0), b1s = tj + 1 < NT ? dpp_mov<kDppShr12>(b1) : 0.0;
                    const double vul = c < 12 ? b0s : b1s;
                    const bool incol = tj == td && qp_l >= 0;      // column j = ka + q' of the new block
                    // V, G V and m are EXACT zeros outside the k x k state (every one of them is built by selects with a literal 0.0), so
                    // V[i][j + 4], V[i + 4][j + 4] vanish by themselves where j + 4 >= k or i + 4 >= k -- in particular in the lanes of block
                    // column db -- and the tests `i < k && j < k`, `j < ka`, `i < ka && j < ka` of the first version only repeated that:
                    // Cov = V[:, 4:] + (G V)' placed in block column db, the new V likewise
                    const double g0x = gvrow[i], g1x = gvrow[i + 4 < 32 ? i + 4 : 31];
                    const double cs0 = incol ? g0x : 0.0;          // (G V)[q'][i]: 0 for i >= k by itself
                    const double cs1 = (incol && i < ka) ? g1x : 0.0;   // (G V)[q'][i + 4]: the row ends at 16 NT, the test stays
                    if (em) S10[ti][tj][v] += fma(mrow[ti][v], mprev_t[tj], vl + cs0);
                    Vn[ti][tj][v] = vul + cs1;
                }
            }
        // block row db: Vn[4 db + k4][j] = GV[k4][j + 4] (j < ka), Vdd in the block itself
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
            const double g0 = gv[tj], g1 = tj + 1 < NT ? gv[tj + 1 < NT ? tj + 1 : tj] : 0.0;
            const double g0s = dpp_mov<kDppShl4>(g0), g1s = tj + 1 < NT ? dpp_mov<kDppShr12>(g1) : 0.0;
            const double gl = c < 12 ? g0s : g1s;
            const int j = cj(tj);
            double rowv = j < ka ? gl : 0.0;
            if (tj == td && qp_l >= 0) rowv = cpick(vdd, qp_l);
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (ti == td && v == vd) Vn[ti][tj][v] = rowv;
        }
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) V[ti][tj] = Vn[ti][tj];
        {
            const double ml4 = m[(lane & 31) + 4 < 32 ? (lane & 31) + 4 : 31];
            const double mdl = cpick(md, (lane - ka) & 3);
            if (lane < 32) mn[lane] = lane < ka ? ml4 : (lane < k ? mdl : 0.0);
        }
        mc ^= 1;
        wave_lds_sync();
    }
    if (!em) return;
    // ---- V, m are now the smoothed moments of the initial state; the sums for the epilogue kernel -----------------------------------
    const double* m0 = sM[mc];
    double* S11full = slot0 + (size_t)T * kSlot;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = ri(ti, v), j = cj(tj);
                const size_t o = (size_t)b * RR + (size_t)i * R + j;
                const bool in = i < k && j < k;
                const bool pd = i >= k && i == j;                  // padding states: independent unit-variance noise
                const double e00 = in ? fma(m0[i], m0[j], V[ti][tj][v]) : 0.0;
                const double eTT = in ? fma(mTrow[ti][v], mTcol[tj], VT[ti][tj][v]) : 0.0;
                S11full[i * R + j] = in ? S11[ti][tj][v] : (pd ? (double)T : 0.0);
                a.S10[o] = in ? S10[ti][tj][v] : 0.0;
                a.S00[o] = in ? S11[ti][tj][v] - eTT + e00 : (pd ? (double)T : 0.0);
                a.P0s[o] = in ? V[ti][tj][v] : (pd ? 1.0 : 0.0);
                if (c == 0 && tj == 0) a.f0s[(size_t)b * R + i] = i < k ? m0[i] : 0.0;
            }
}

// Companion states in blocks of 4 (RecursionArgs::kdim = 4 m, 8 <= kdim <= 32), covariance-form routes of the library:
//   VAR(p) factor dynamics (dfm_*_varp_*): observation 4 wide on the first block (Rc = rl = 4, kb = 0), Rp = 16 | 32;
//   AR idiosyncratic terms (dfm_*_ar_*): observation on every block, as wide as the padded state (Rc = 0, kb = 4).
bool recursion_comp_supported(int Rpad, const RecursionArgs& a) {
    static const bool off = [] { const char* v = diag_env("DFM_NO_COMP"); return v && atoi(v) != 0; }();
    if (off || (Rpad != 16 && Rpad != 32) || !a.cov || a.kdim < 8 || a.kdim > Rpad || (a.kdim & 3) != 0) return false;
    if (a.kdim <= Rpad / 2 && Rpad == 32) return false;           // (a 16-wide state is not planned at Rp = 32)
    const bool var = a.Rc == 4 && a.rl == 4 && a.kb == 0;
    const bool ar = a.Rc == 0 && a.rl == 0 && a.kb == 4;
    if (!var && !ar) return false;
    if (a.r > Rpad || a.ZJtab == nullptr) return false;
    if (a.S11 && !a.A_out) return false;
    if (a.qsing) return false;                                    // the r x r innovation block is inverted here
    return true;
}

hipError_t launch_recursion_comp(int Rpad, const RecursionArgs& a0, hipStream_t s) {
    note_kernel("recursion_comp_kernel");
    RecursionArgs a = a0;
    static const int abl = [] { const char* v = diag_env("DFM_COMP_ABL"); return v ? atoi(v) : 0; }();   // diagnostics build: timing ablations
    a.tile_nc = abl;
    const bool var = a.Rc == 4;
    if (Rpad == 16) {
        if (var) hipLaunchKernelGGL((recursion_comp_kernel<1, true>), dim3(a.B), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((recursion_comp_kernel<1, false>), dim3(a.B), dim3(64), 0, s, a);
    } else {
        if (var) hipLaunchKernelGGL((recursion_comp_kernel<2, true>), dim3(a.B), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((recursion_comp_kernel<2, false>), dim3(a.B), dim3(64), 0, s, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.S11 == nullptr) return e;
    return launch_cov_epilogue(Rpad, a, s);
}

}  // namespace dfm
