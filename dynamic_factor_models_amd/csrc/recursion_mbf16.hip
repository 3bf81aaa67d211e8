// recursion_mbf16.hip -- the smoother pass of a model whose innovation covariance is SINGULAR (the companion state of VAR(p) factor
// dynamics, dfm_functions.ipynb:477-492: k = r p <= 16 state components, the observation loading on the first rc <= 4) with ONE WAVE
// PER REPLICATE, every 16 x 16 matrix as one accumulator tile of `v_mfma_f64_16x16x4` (four doubles per lane) and NO k x k inversion.
//
// recursion_wave_kernel<16, COV> (recursion_wave.hip) runs the textbook covariance-form filter + RTS smoother with an element of every
// matrix per thread of a 256-thread workgroup: per period two 16-pivot sweep inversions (P_p^-1 for the smoother gain, (P_p^-1 + C_t)^-1
// for the update) and seven products through LDS tiles behind 4-wave barriers -- ~11 000 cycles per period, 256 VGPRs + scratch, two
// workgroups per CU: 4.2 ms per EM iteration of 1024 Stock-Watson-shaped replicates (r = 4, p = 4), the route of the reference's
// DEFAULT model (n_factorlag = 4).  Here (scripts/dbg/r06/mbf_emul.py is the NumPy model, tests/test_mbf_model_cpu.py pins it to
// oracle/varp_oracle.py):
//   forward   P_p = A (P_f A') + Q;  the collapsed observation (b_t, C_t on the first rc components) enters by a RANK-rc update
//             P_f' = P_p - P1 W P1',  W = (I + C P11)^-1 C = C - Y G^-1 Y'  with  P11 = L L', Y = C L, G = I + L' C L  (two 4 x 4 Cholesky
//             factors, done redundantly by every lane on wave-uniform values; W and log det(I + C P11) = log det G are well defined for a
//             singular or zero C_t: periods with few or no observed cells need no special case);
//   backward  the modified Bryson-Frazier recursion (de Jong 1989; Durbin & Koopman 2012, sec. 4.4-4.7) on the adjoint pair (r_t, N_t):
//             r_{t-1} = E u_t + L_t' r_t,  N_{t-1} = E W_t E' + L_t' N_t L_t,  L_t = A (I - P1 W E'),  u_t = (I + C P11)^-1 (b_t - C m_p1);
//             s_t|T = m_p + P_p r_{t-1},  V_t = P_p - P_p N_{t-1} P_p,  Cov(s_t, s_{t-1} | X) = (I - P_p,t N_{t-1}) A P_f,t-1.
// Operand layouts of the instruction (lane l: k4 = l / 16, c = l % 16): D[v] = M[k4 + 4 v][c]; a matrix in this layout IS the B operand
// (right factor) of a product, and the A operand (left factor) of its TRANSPOSE -- so symmetric matrices serve on both sides, the constant
// A is kept in both layouts, and every product of the recursion is arranged so that its left factor is symmetric, constant, or the
// transpose of something just computed (no cross-lane transposition anywhere).  Rank-4 terms are ONE instruction (k = 4).
// Vectors live in LDS (one wave: its DS operations are served in order, no barrier).
// The EM epilogue (A, Q, mu0, P0, S11^-1, bookkeeping: companion constraints and all) is recursion_wave_kernel<16, COV>'s text with an
// element per thread, as its own small launch on the sums this kernel leaves (cov_epilogue_kernel below).
// Reference counterpart: none (dfm_functions.ipynb:21-23 declares `Parametric` only); oracle: oracle/varp_oracle.py, oracle/ar_oracle.py.
#include <stdlib.h>

#include "dfm_kernels.h"
#include "dfm_smallmat.h"
#include "dfm_grid.h"

namespace dfm {

namespace {

typedef double m16 __attribute__((ext_vector_type(4)));
constexpr int K16 = 16, KK16 = 256;
constexpr double kLog2PiM = 1.8378770664093454835606594728112;
// per-period scratch (the replicate's ZJtab slot of 512 doubles): [0, 256) P_p, lane-major (lane l: doubles 4 l .. 4 l + 3);
// [256, 272) W row-major; [272, 276) u; [288, 304) m_p.  Slot T: [0, 256) S11 (full 16 x 16, row-major) for the epilogue kernel.
constexpr int kSlot = 2 * KK16, kOffW = 256, kOffU = 272, kOffMp = 288;

__device__ __forceinline__ m16 mm4(const m16& a, const m16& b, m16 acc) {   // acc + Left Right (a: A operand of Left, b: Right in D layout)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ m16 mm1(double a, double b, m16 acc) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); }
__device__ __forceinline__ double rdlane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum16(double v) {          // sum over the 16 lanes of a row group, in every lane of it
    v += xor_lane<1>(v); v += xor_lane<2>(v); v += xor_lane<4>(v); v += xor_lane<8>(v);
    return v;
}
__device__ __forceinline__ double col_sum4(double v) {           // sum over the 4 row groups (lanes c, c + 16, c + 32, c + 48)
    v += xor_lane<16>(v); v += xor_lane<32>(v);
    return v;
}
// sqrt(x) and 1 / sqrt(x) of a positive, normal x: v_rsq_f64 refined by two Newton steps, the root by one correction (a pivot of an
// SPD matrix: no scaling, no fix-up -- the IEEE sqrt + division pair is ~35 instructions on a chain that is nothing but such pivots)
__device__ __forceinline__ void fast_sqrt_rsqrt(double x, double& s, double& r) {
    r = __builtin_amdgcn_rsq(x);
    r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
    r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
    s = x * r;
    s = fma(0.5 * r, fma(-s, s, x), s);
}
// (PRVALUE operands -- unary plus: a conditional on array lvalues is an lvalue, i.e. a select of addresses + one load: the array stays in
// scratch memory and every pick is a scratch round trip behind s_waitcnt vmcnt(0); recursion_comp.hip cpick)
__device__ __forceinline__ double pick4(const double (&x)[4], int q) {
    const double a0 = +x[0], a1 = +x[1], a2 = +x[2], a3 = +x[3];
    return q == 0 ? +a0 : q == 1 ? +a1 : q == 2 ? +a2 : +a3;
}

// the wave-uniform 4 x 4 algebra of one period (see the head of the file).  In: P11 (symmetric, full), C (symmetric, full), e0 = b - C mp1.
// Out: W (symmetric, full), u, det G.  false: a pivot was not positive (P11 is not positive definite to working precision).
struct Upd4 {
    double W[4][4], u[4], detG;
};
__device__ __forceinline__ bool update4(const double (&P)[4][4], const double (&C)[4][4], const double (&e)[4], Upd4& o) {
    double L[4][4], Y[4][4], G[4][4], Lg[4][4], Z[4][4], ig[4];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { L[i][j] = 0.0; Lg[i][j] = 0.0; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                 // P11 = L L'
        double d = P[j][j];
#pragma unroll
        for (int q = 0; q < j; ++q) d = fma(-L[j][q], L[j][q], d);
        ok = ok && (d > 0.0);
        double ljj, inv;
        fast_sqrt_rsqrt(d > 0.0 ? d : 1.0, ljj, inv);
        L[j][j] = ljj;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            double s = P[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q) s = fma(-L[i][q], L[j][q], s);
            L[i][j] = s * inv;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {                             // Y = C L
            double s = 0.0;
#pragma unroll
            for (int q = j; q < 4; ++q) s = fma(C[i][q], L[q][j], s);
            Y[i][j] = s;
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {                            // G = I + L' Y  (symmetric: lower triangle)
            double s = i == j ? 1.0 : 0.0;
#pragma unroll
            for (int q = i; q < 4; ++q) s = fma(L[q][i], Y[q][j], s);
            G[i][j] = s; G[j][i] = s;
        }
    double det = 1.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                                 // G = Lg Lg'
        double d = G[j][j];
#pragma unroll
        for (int q = 0; q < j; ++q) d = fma(-Lg[j][q], Lg[j][q], d);
        ok = ok && (d > 0.0);
        det *= (d > 0.0 ? d : 1.0);
        double ljj, inv;
        fast_sqrt_rsqrt(d > 0.0 ? d : 1.0, ljj, inv);
        ig[j] = inv;
        Lg[j][j] = ljj;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            double s = G[i][j];
#pragma unroll
            for (int q = 0; q < j; ++q) s = fma(-Lg[i][q], Lg[j][q], s);
            Lg[i][j] = s * inv;
        }
    }
    o.detG = det;
#pragma unroll
    for (int j = 0; j < 4; ++j)                                   // Z = Lg^-1 Y'  (column j: the row j of Y)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double s = Y[j][i];
#pragma unroll
            for (int q = 0; q < i; ++q) s = fma(-Lg[i][q], Z[q][j], s);
            Z[i][j] = s * ig[i];
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {                            // W = C - Z' Z
            double s = C[i][j];
#pragma unroll
            for (int q = 0; q < 4; ++q) s = fma(-Z[q][i], Z[q][j], s);
            o.W[i][j] = s; o.W[j][i] = s;
        }
    double h[4], g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                 // h = L' e
        double s = 0.0;
#pragma unroll
        for (int q = i; q < 4; ++q) s = fma(L[q][i], e[q], s);
        h[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                 // g = Lg^-1 h
        double s = h[i];
#pragma unroll
        for (int q = 0; q < i; ++q) s = fma(-Lg[i][q], g[q], s);
        g[i] = s * ig[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                 // u = e - Z' g
        double s = e[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) s = fma(-Z[q][i], g[q], s);
        o.u[i] = s;
    }
    return ok;
}

}  // namespace

template <int RC>
__global__ __launch_bounds__(64) void recursion_mbf16_kernel(RecursionArgs a) {
    __shared__ double vmf[K16], vmp[K16], vr[K16], vs[K16], vy[K16];
    const int lane = threadIdx.x, b = blockIdx.x;
    const int k4 = lane >> 4, c = lane & 15;
    const int T = a.T, N = a.N;
    constexpr int Rc = RC, NPc = Rc * (Rc + 1) / 2;
    // (wave-uniform, but deliberately VECTOR loads: scalar loads share the LDS counter, so every LDS exchange of a step would wait for
    // the next period's prefetch -- measured 1.57 -> 2.17 ms per 1024 EM iterations with s_load)
    const double* bcol = a.bcol + (size_t)b * T * Rc;
    const double* scol = a.scol + (size_t)b * T;
    const int* nobs = a.nobs + (size_t)b * T;
    const double* ldrow = a.ldrow + (size_t)b * T;
    const bool haveCt = a.Ct != nullptr;
    const double* Ctb = (haveCt ? a.Ct : a.Cfull) + (haveCt ? (size_t)b * T * NPc : 0);   // (never read without a C_t array)
    const double ldfull = a.ldfull[b];
    double* slot0 = a.ZJtab + (size_t)b * (T + 1) * kSlot;
    const bool em = a.S11 != nullptr;

    // ---- constants: A in both operand layouts, Q; the replicate's full Gram matrix (periods without a missing cell) ---------------
    m16 dA, aA, dQ, pf;
    {
        const double* Ag = a.A + (size_t)b * KK16;
        const double* Qg = a.Q + (size_t)b * KK16;
        const double* Pg = a.P0 + (size_t)b * KK16;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            dA[v] = Ag[(k4 + 4 * v) * K16 + c];
            aA[v] = Ag[c * K16 + 4 * v + k4];
            dQ[v] = Qg[(k4 + 4 * v) * K16 + c];
            pf[v] = 0.5 * (Pg[(k4 + 4 * v) * K16 + c] + Pg[c * K16 + k4 + 4 * v]);
        }
    }
    const m16 P0m = pf;
    double Cf[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Cf[i][j] = (i < Rc && j < Rc) ? a.Cfull[(size_t)b * Rc * Rc + i * Rc + j] : 0.0;
    if (lane < K16) vmf[lane] = a.mu0[(size_t)b * K16 + lane];
    wave_lds_sync();
    // W[c][k4] for the lanes of the first four columns, 0 elsewhere: the A operand of "W padded" and (W is symmetric) its B operand
    auto wsel = [&](const double (&W)[4][4]) -> double {
        double r0 = pick4(W[0], k4), r1 = pick4(W[1], k4), r2 = pick4(W[2], k4), r3 = pick4(W[3], k4);
        const double x = c == 0 ? r0 : c == 1 ? r1 : c == 2 ? r2 : r3;
        return c < 4 ? x : 0.0;
    };
    const m16 zero = {0.0, 0.0, 0.0, 0.0};

    // =================================================== forward ===========================================================
    LogProd detprod;
    double qsum = 0.0, ssum = 0.0, nsum = 0.0, ldsum = 0.0;
    bool okall = true;
    // the collapsed observation of a period (wave-uniform) is requested one step ahead: with one wave per SIMD nothing else hides a
    // round trip to memory, and the step's first dependent use is ~500 cycles after its top
    struct ObsIn { int nt; double b[4], cp[10], s, ld; };
    auto fetch_obs = [&](int t) {
        ObsIn o;
        t = t < T ? t : T - 1;
        o.nt = nobs[t];
        o.s = scol[t];
        o.ld = ldrow[t];
#pragma unroll
        for (int i = 0; i < 4; ++i) o.b[i] = i < Rc ? bcol[(size_t)t * Rc + i] : 0.0;
#pragma unroll
        for (int q = 0; q < 10; ++q) o.cp[q] = (haveCt && q < NPc) ? Ctb[(size_t)t * NPc + q] : 0.0;
        return o;
    };
    ObsIn nxt = fetch_obs(0);
    for (int t = 0; t < T; ++t) {
        const ObsIn cur = nxt;
        nxt = fetch_obs(t + 1);
        const int nt = cur.nt;
        const bool full = nt == N || !haveCt;
        double C[4][4], bt[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bt[i] = cur.b[i];
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const double cv = (i < Rc) ? (full ? Cf[i][j] : cur.cp[i * (i + 1) / 2 + j]) : 0.0;
                C[i][j] = cv; C[j][i] = cv;
            }
        }
        // P_p = A (P_f A') + Q;  m_p = A m_f
        const m16 X = mm4(pf, aA, zero);
        const m16 pp = mm4(aA, X, dQ);
        {
            const double mfc = vmf[c];
            double y[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) y[v] = row_sum16(dA[v] * mfc);
            if (c == 0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) vmp[k4 + 4 * v] = y[v];
            }
        }
        wave_lds_sync();
        double P11[4][4], mp1[4], e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mp1[i] = vmp[i];
#pragma unroll
            for (int j = 0; j <= i; ++j) {
                const double pv = rdlane(pp[0], 16 * i + j);
                P11[i][j] = pv; P11[j][i] = pv;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double s = bt[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) s = fma(-C[i][q], mp1[q], s);
            e[i] = s;
        }
        Upd4 up;
        okall = update4(P11, C, e, up) && okall;
        // P_f = P_p - P1 W P1'  (two rank-4 instructions);  m_f = m_p + P1 u
        const double ws = wsel(up.W);
        const m16 WP = mm1(ws, pp[0], zero);                      // rows 0..3: W P1'
        pf = mm1(-WP[0], pp[0], pp);
        {
            const double pu = col_sum4(pp[0] * pick4(up.u, k4));   // (P1 u)[c]
            if (k4 == 0) vmf[c] = vmp[c] + pu;
        }
        // log-likelihood terms (wave-uniform)
        {
            double quad = cur.s;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double mf1 = mp1[i];
#pragma unroll
                for (int q = 0; q < 4; ++q) mf1 = fma(P11[i][q], up.u[q], mf1);
                quad -= fma(bt[i], mp1[i], e[i] * mf1);
            }
            qsum += quad;
            ssum += 0.0;
            nsum += (double)nt;
            ldsum += (nt == N) ? ldfull : cur.ld;
            detprod.mul(up.detG);
        }
        // what the backward sweep reads back
        {
            double* sl = slot0 + (size_t)t * kSlot;
            *reinterpret_cast<m16*>(sl + 4 * lane) = pp;
            if (c < 4) sl[kOffW + 4 * c + k4] = ws;
            if (c == 4) sl[kOffU + k4] = pick4(up.u, k4);
            if (lane < K16) sl[kOffMp + lane] = vmp[lane];
        }
        wave_lds_sync();
    }
    const double ll = okall ? -0.5 * (nsum * kLog2PiM + ldsum + detprod.log_value() + qsum + ssum) : __builtin_nan("");
    if (lane == 0) {
        a.loglik[b] = ll;
        if (a.ncov) a.ncov[b] = T;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");        // the slots are read back by this wave (other lanes)

    // =================================================== backward ==========================================================
    const int r = a.r, rl = a.rl > 0 ? a.rl : K16, npr = r * (r + 1) / 2;
    m16 Nn = zero, ImNPn = zero, S11 = zero, S10 = zero, VT = zero;
    double snext[4] = {0.0, 0.0, 0.0, 0.0}, sTrow[4] = {0.0, 0.0, 0.0, 0.0}, sTc = 0.0;
    if (lane < K16) vr[lane] = 0.0;
    wave_lds_sync();
    struct SlotIn { m16 pp; double W[16], u[4], ws, mpr[4]; };
    auto fetch_slot = [&](int t) {
        SlotIn o;
        const double* sl = slot0 + (size_t)(t > 0 ? t : 0) * kSlot;
        const double* su = sl;
        o.pp = *reinterpret_cast<const m16*>(sl + 4 * lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) o.W[q] = su[kOffW + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) { o.u[q] = su[kOffU + q]; o.mpr[q] = sl[kOffMp + k4 + 4 * q]; }
        o.ws = c < 4 ? sl[kOffW + 4 * c + k4] : 0.0;
        return o;
    };
    SlotIn snx = fetch_slot(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        const SlotIn sc_ = snx;
        snx = fetch_slot(t - 1);                                   // (the slot of the next step: its round trip under this step's products)
        const m16 pp = sc_.pp;
        double W[4][4], u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u[i] = sc_.u[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) W[i][j] = sc_.W[4 * i + j];
        }
        const double ws = sc_.ws;
        // r_{t-1} = E u + (I - E W P1') A' r
        {
            double y = 0.0;
#pragma unroll
            for (int v = 0; v < 4; ++v) y = fma(dA[v], vr[k4 + 4 * v], y);
            y = col_sum4(y);                                       // (A' r)[c]
            const double z = row_sum16(pp[0] * y);                 // (P1' A' r)[k4], in every lane of row group k4
            double zq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) zq[q] = rdlane(z, 16 * q);
            double corr = 0.0;
            if (c < 4) {
                const double w0 = c == 0 ? +W[0][0] : c == 1 ? +W[1][0] : c == 2 ? +W[2][0] : +W[3][0];
                const double w1 = c == 0 ? +W[0][1] : c == 1 ? +W[1][1] : c == 2 ? +W[2][1] : +W[3][1];
                const double w2 = c == 0 ? +W[0][2] : c == 1 ? +W[1][2] : c == 2 ? +W[2][2] : +W[3][2];
                const double w3 = c == 0 ? +W[0][3] : c == 1 ? +W[1][3] : c == 2 ? +W[2][3] : +W[3][3];
                corr = (c == 0 ? +u[0] : c == 1 ? +u[1] : c == 2 ? +u[2] : +u[3]) - (w0 * zq[0] + w1 * zq[1] + w2 * zq[2] + w3 * zq[3]);
            }
            wave_lds_sync();                                       // (every lane has read the old r)
            if (k4 == 0) vr[c] = y + corr;
        }
        // N_{t-1} = E W E' + D' (A' N A) D,  D = I - P1 W E'
        const m16 NA = mm4(Nn, dA, zero);
        const m16 M = mm4(NA, dA, zero);
        const m16 PW = mm1(pp[0], ws, zero);                       // P1 W (columns 0..3)
        m16 Tm = M;
        {
            const m16 MPW = mm4(M, PW, zero);
#pragma unroll
            for (int v = 0; v < 4; ++v) Tm[v] -= MPW[v];
        }
        const m16 PT = mm4(pp, Tm, zero);
        const m16 WPT = mm1(ws, PT[0], zero);                      // rows 0..3: W P1' T
        Nn = Tm;
        Nn[0] = Nn[0] - WPT[0] + ws;                               // (+ E W E': ws is W[c][k4] = W[k4][c] in the lanes of columns 0..3, 0 elsewhere)
        wave_lds_sync();
        // s_t|T = m_p + P_p r,  V_t = P_p - P_p N P_p
        const m16 NP = mm4(Nn, pp, zero);
        m16 V = pp;
        {
            const m16 PNP = mm4(NP, pp, zero);
#pragma unroll
            for (int v = 0; v < 4; ++v) V[v] -= PNP[v];
        }
        double srow[4];
        {
            const double rc_ = vr[c];
#pragma unroll
            for (int v = 0; v < 4; ++v) srow[v] = row_sum16(pp[v] * rc_);
            wave_lds_sync();
            if (c == 0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) vs[k4 + 4 * v] = sc_.mpr[v] + srow[v];
            }
            wave_lds_sync();
#pragma unroll
            for (int v = 0; v < 4; ++v) srow[v] = vs[k4 + 4 * v];
        }
        const double sc = vs[c];
        // outputs of period t (0-based): the first r components, those beyond rl as padding (mean 0, identity covariance)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = k4 + 4 * v;
            if (i < r) {
                if (c == 0) a.f_smooth[((size_t)b * T + t) * r + i] = i < rl ? srow[v] : 0.0;
                if (a.P_smooth && c <= i) a.P_smooth[((size_t)b * T + t) * npr + i * (i + 1) / 2 + c] = (i < rl && c < rl) ? V[v] : (i == c ? 1.0 : 0.0);
            }
        }
        if (em) {
            if (t == T - 1) {
                VT = V; sTc = sc;
#pragma unroll
                for (int v = 0; v < 4; ++v) sTrow[v] = srow[v];
            } else {
                // Cov(s_{t+1}, s_t | X) = (I - P_p,t+1 N_t) A P_f,t  with  P_f,t = P_p - P1 W P1'
                const m16 WP = mm1(ws, pp[0], zero);
                const m16 pft = mm1(-WP[0], pp[0], pp);
                const m16 APf = mm4(aA, pft, zero);
                const m16 Cv = mm4(ImNPn, APf, zero);
#pragma unroll
                for (int v = 0; v < 4; ++v) S10[v] += fma(snext[v], sc, Cv[v]);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                S11[v] += fma(srow[v], sc, V[v]);
                ImNPn[v] = ((k4 + 4 * v) == c ? 1.0 : 0.0) - NP[v];
                snext[v] = srow[v];
            }
        }
    }
    if (!em) return;
    // ---- the initial state: s_1 = A s_0 + eta, nothing observed at t = 0 -----------------------------------------------------------
    {
        double y = 0.0;
#pragma unroll
        for (int v = 0; v < 4; ++v) y = fma(dA[v], vr[k4 + 4 * v], y);
        y = col_sum4(y);                                           // r_0' = A' r
        wave_lds_sync();
        if (k4 == 0) vy[c] = y;
        wave_lds_sync();
    }
    const m16 NA0 = mm4(Nn, dA, zero);
    const m16 N0 = mm4(NA0, dA, zero);                             // A' N A
    const m16 NP0 = mm4(N0, P0m, zero);
    m16 P0s = P0m;
    {
        const m16 PNP = mm4(NP0, P0m, zero);
#pragma unroll
        for (int v = 0; v < 4; ++v) P0s[v] -= PNP[v];
    }
    double f0row[4];
    {
        const double yc = vy[c];
#pragma unroll
        for (int v = 0; v < 4; ++v) f0row[v] = row_sum16(P0m[v] * yc);
        wave_lds_sync();
        if (c == 0) {
#pragma unroll
            for (int v = 0; v < 4; ++v) vs[k4 + 4 * v] = a.mu0[(size_t)b * K16 + k4 + 4 * v] + f0row[v];
        }
        wave_lds_sync();
#pragma unroll
        for (int v = 0; v < 4; ++v) f0row[v] = vs[k4 + 4 * v];
    }
    const double f0c = vs[c];
    {
        const m16 AP0 = mm4(aA, P0m, zero);                        // L_0 P_0 = A P0
        const m16 Cv = mm4(ImNPn, AP0, zero);                      // Cov(s_1, s_0 | X)
#pragma unroll
        for (int v = 0; v < 4; ++v) S10[v] += fma(snext[v], f0c, Cv[v]);
    }
    double* S11full = slot0 + (size_t)T * kSlot;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int i = k4 + 4 * v;
        const size_t o = (size_t)b * KK16 + (size_t)i * K16 + c;
        const double e00 = fma(f0row[v], f0c, P0s[v]);
        S11full[i * K16 + c] = S11[v];
        a.S10[o] = S10[v];
        a.S00[o] = S11[v] - fma(sTrow[v], sTc, VT[v]) + e00;
        a.P0s[o] = P0s[v];
        if (c == 0) a.f0s[(size_t)b * K16 + i] = f0row[v];
    }
}

// ---- the EM epilogue: recursion_wave_kernel<16, COV>'s, element (i, j) of every 16 x 16 matrix per thread, on the sums the pass left ------
// EM bookkeeping (log-likelihood path, iteration counts, who keeps iterating), A = S10 S00^-1 with the companion constraints
// (RecursionArgs::kdim / ka / kb), Q = sym(S11 - A S10') / T, mu0, P0, S11 / S11^-1 in the loadings step's layout.
template <int R>
__global__ __launch_bounds__(R * R) void cov_epilogue_kernel(RecursionArgs a) {
    constexpr int RR = R * R, TS = kTileStride<R>, RT = R * TS;
    constexpr int kSlotE = 2 * RR;
    extern __shared__ __attribute__((aligned(16))) double esm[];
    double* L0 = esm;
    double* L1 = L0 + RT;
    Grid<R> G;
    G.prow = L1 + RT;
    G.red = G.prow + kGridProw<R>;
    G.tt = G.red + 2 * (RR / 64) * R;
    const int lane = threadIdx.x, b = blockIdx.x;
    const int i = lane / R, j = lane % R;
    G.l = lane; G.i = i; G.j = j;
    const int T = a.T;
    const int Rc = a.Rc > 0 ? a.Rc : R;
    const bool inC = i < Rc && j < Rc;
    const int rl = a.rl > 0 ? a.rl : R;
    const bool inL = i < rl && j < rl;
    const size_t o = (size_t)b * RR + lane;
    const double ll = a.loglik[b];
    bool em_apply = true;
    if (a.active) {
        const bool was = a.k == 0 ? true : (a.active[b] != 0);
        bool go = was;
        if (was && a.k >= 1 && a.tol > 0.0) {
            const double llp = a.ll_path[(size_t)b * a.max_iter + a.k - 1];
            go = !((ll - llp) / (0.5 * (fabs(ll) + fabs(llp))) < a.tol);
        }
        em_apply = go;
        __syncthreads();
        if (lane == 0) {
            if (was) { a.ll_path[(size_t)b * a.max_iter + a.k] = ll; a.iters[b] = a.k + 1; }
            a.active[b] = go ? 1 : 0;
        }
    }
    const double S11 = a.ZJtab[(size_t)b * (T + 1) * kSlotE + (size_t)T * kSlotE + lane];
    const double S10 = a.S10[o], S00 = a.S00[o], Ps = a.P0s[o];
    const double fs_r = a.f0s[(size_t)b * R + i];
    const bool narrow = a.rl > 0;
    if (!narrow) a.S11[o] = S11;
    if (!a.A_out) return;
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

template <int R>
static hipError_t launch_cov_epilogue_r(const RecursionArgs& a, hipStream_t s) {
    constexpr size_t RT = (size_t)R * kTileStride<R>;
    const size_t lds = (2 * RT + kGridProw<R> + 2 * (R * R / 64) * R + 2 * RT) * sizeof(double);
    hipLaunchKernelGGL(cov_epilogue_kernel<R>, dim3(a.B), dim3(R * R), lds, s, a);
    return hipGetLastError();
}
// the sums of a pass (S11 in the last slot of ZJtab, S10 / S00 / P0s / f0s in their arrays, the log-likelihood) -> the M-step
hipError_t launch_cov_epilogue(int Rpad, const RecursionArgs& a, hipStream_t s) {
    return Rpad == 16 ? launch_cov_epilogue_r<16>(a, s) : Rpad == 32 ? launch_cov_epilogue_r<32>(a, s) : hipErrorInvalidValue;
}

// Rp = 16, covariance form, observation on the first rc <= 4 state components in the collapse kernels' narrow layout (VAR(p) factor
// dynamics: dfm_*_varp_*); the EM epilogue is recursion_wave_kernel<16, COV>'s (RecursionArgs::sums_ready)
bool recursion_mbf16_supported(int Rpad, const RecursionArgs& a) {
    static const bool off = [] { const char* v = diag_env("DFM_NO_MBF16"); return v && atoi(v) != 0; }();
    if (off || Rpad != 16 || !a.cov || a.kb != 0) return false;
    if (a.Rc != 2 && a.Rc != 4) return false;
    if (a.rl < 1 || a.rl > a.Rc) return false;
    if (a.r > 16 || a.ZJtab == nullptr) return false;
    if (a.S11 && !a.A_out) return false;                           // (sums without the M-step: not a path the library takes)
    return true;
}

hipError_t launch_recursion_mbf16(const RecursionArgs& a, hipStream_t s) {
    note_kernel("recursion_mbf16_kernel");
    if (a.Rc == 4) hipLaunchKernelGGL(recursion_mbf16_kernel<4>, dim3(a.B), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(recursion_mbf16_kernel<2>, dim3(a.B), dim3(64), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.S11 == nullptr) return e;
    return launch_cov_epilogue(16, a, s);
}

}  // namespace dfm
