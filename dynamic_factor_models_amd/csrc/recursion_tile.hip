// recursion_tile.hip -- the sequential Kalman recursion for wide states (17 <= r <= 31, padded to 32: BASELINE config 4 with
// missing cells) on the f64 matrix pipe: ONE workgroup of FOUR waves per replicate, every 32 x 32 matrix held as the four
// 16 x 16 accumulator tiles of v_mfma_f64_16x16x4 -- wave w = 2 I + J owns tile (I, J); lane (q = l / 16, c = l % 16) holds
// element (16 I + q + 4 v, 16 J + c) in register v: the "tile layout" (TL).
//
// recursion_wave_kernel<32> gives the same replicate 1024 threads with an ELEMENT per thread: 17.8 us per period, 61 % of the
// forward sweep in the sweep inverse (16 block pivots x (a 16-wave barrier + the pivot wave's chain + 112 wave-level LDS reads)),
// 0.003 of the HBM roofline.  What the tile layout buys:
//   * products: with both operands in TL the instruction computes Y'X -- the A operand of k-step s is register s of Y's tile
//     (kb, I) AS IT STANDS, the B operand register s of X's tile (kb, J).  The recursion is arranged so that every product has
//     that form:  J = Z'K',  J' = (K')'Z,  K J = (K')'J  forward;  U = P'J',  J U = (J')'U  backward  (Z, P symmetric; K' constant).
//     A product is <= 8 MFMAs per wave; tiles of other waves come through an 8-KB LDS buffer (ds_*_b128, lane-contiguous);
//   * inverse: symmetric sweep operator with 4 x 4 BLOCK pivots.  The pivot rows k0 .. k0 + 3 over column block J are register
//     (k0 % 16) / 4 of tile (k0 / 16, J), lane (q, c) = row k0 + q -- exactly the B-operand layout, and (by symmetry) the
//     A-operand layout of the pivot COLUMNS.  With the pivot block replaced by -I in the published rows R~, one rank-4 update
//     tile -= R~' (D^-1 R~) -- ONE MFMA per wave -- yields D^-1 A_Kj, its transpose and -D^-1 in place (the trick of
//     Grid::sweep_inverse, dfm_grid.h).  One barrier of 4 waves per pivot; only ceil(r / 4) pivots and k-steps are executed:
//     the identity padding beyond r is never touched (r = 20: 5 of 8, what an Rp = 24 instantiation would have bought);
//   * matrix-vector products ride in column 31 (padding for r <= 31) of the products:  Z'[K' | xi] = [J | w],
//     (K')'[J | w] = [K J | K w],  (J')'[U | f+] = [J U | J f+] -- no separate mean recursion, no lane reductions.
// The algebra is recursion_wave_kernel's (information form, Z-smoother); scripts/dbg/tile_emul.py is a
// lane-level NumPy model of this file that reproduces oracle/kalman_oracle.py.  Every period is a new covariance step (no
// memoisation: at these widths a row without a missing cell is the exception).  EM: sums of P_t and U_t are accumulated in TL;
// sum f f' terms are three more products over time (MFMA k = 4 periods) in the epilogue; the transition M-step is
// tile_mstep_kernel (Grid<32>, the epilogue of recursion_wave_kernel on the sums this kernel leaves in the workspace).
// The reference has no counterpart (dfm_functions.ipynb:21-23 declares `Parametric` only).
#include <stdlib.h>

#include "dfm_cov.h"
#include "dfm_gram.h"
#include "dfm_grid.h"
#include "dfm_kernels.h"
#include "dfm_smallmat.h"

namespace dfm {

// -DDFM_TILE_PROF (development builds): s_memrealtime spans (10 ns ticks) of replicate 5, printed by thread 0
#ifdef DFM_TILE_PROF
#define RT_NOW() ((long long)__builtin_amdgcn_s_memrealtime())
#define RT_TICK(var) do { var -= RT_NOW(); } while (0)
#define RT_TOCK(var) do { var += RT_NOW(); } while (0)
#else
#define RT_NOW() 0ll
#define RT_TICK(var) do {} while (0)
#define RT_TOCK(var) do {} while (0)
#endif

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kRt = 32;
constexpr int kRtTile = 256;                       // doubles of one tile: [2 register pairs][64 lanes] double2
constexpr int kRtBufA = 0;                         // exchange 1: Z | P_s | prologue operands        (4 tiles)
constexpr int kRtBufB = kRtBufA + 4 * kRtTile;     // exchange 2: [J | w] | [U | f+]                  (4 tiles)
constexpr int kRtXk = kRtBufB + 4 * kRtTile;       // K' with column 31 := xi_t  (B operand of J = Z'[K' | xi])
constexpr int kRtPraw = kRtXk + 4 * kRtTile;       // [2][4][32] published pivot rows (pivot block := -I)
constexpr int kRtPD = kRtPraw + 2 * 128;           // [2][16]    raw pivot block
constexpr int kRtRed = kRtPD + 32;                 // [64]
constexpr int kRtLds = kRtRed + 64;
constexpr double kLog2PiT = 1.8378770664093454835606594728112;
constexpr int kRtCh = 2;                           // periods per prefetch chunk

struct RtCtx {
    double* sm;
    int lane, w, I, J, q, c;
    int pp;                                        // pivot exchanges so far (buffer parity)
};

// Barrier of the workgroup's four waves that waits for this wave's LDS traffic ONLY.  __syncthreads() also drains vmcnt: every
// one of the ~9 barriers of a period would then wait for the table stores (8 KB per wave and period) and for the prefetch of the
// next chunk -- HBM round trips inside a chain whose steps are a few hundred cycles (the lesson of recursion_pair.hip).
__device__ __forceinline__ void rt_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void st_tile(double* buf, int tile, int lane, const v4d& m) {
    double2* p = reinterpret_cast<double2*>(buf + tile * kRtTile);
    p[lane] = make_double2(m[0], m[1]);
    p[64 + lane] = make_double2(m[2], m[3]);
}
__device__ __forceinline__ v4d ld_tile(const double* buf, int tile, int lane) {
    const double2* p = reinterpret_cast<const double2*>(buf + tile * kRtTile);
    const double2 a = p[lane], b = p[64 + lane];
    v4d m;
    m[0] = a.x; m[1] = a.y; m[2] = b.x; m[3] = b.y;
    return m;
}
// the same register-pair layout in global memory (the (Z, J') table): 1-KB coalesced accesses
__device__ __forceinline__ void st_tile_g(double* tab, int tile, int lane, const v4d& m) {
    double2* p = reinterpret_cast<double2*>(tab + (size_t)tile * kRtTile);
    p[lane] = make_double2(m[0], m[1]);
    p[64 + lane] = make_double2(m[2], m[3]);
}
__device__ __forceinline__ v4d ld_tile_g(const double* tab, int tile, int lane) {
    const double2* p = reinterpret_cast<const double2*>(tab + (size_t)tile * kRtTile);
    const double2 a = p[lane], b = p[64 + lane];
    v4d m;
    m[0] = a.x; m[1] = a.y; m[2] = b.x; m[3] = b.y;
    return m;
}

// acc + Y'X restricted to the first nks k-steps (rows 4 nks .. 31 of Y and X are padding that contributes nothing):
// Y0 / Y1 = Y's tiles (0, I) / (1, I), X0 / X1 = X's tiles (0, J) / (1, J)
__device__ __forceinline__ v4d mm_tn(const v4d& Y0, const v4d& Y1, const v4d& X0, const v4d& X1, int nks, v4d acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Y0[s], X0[s], acc, 0, 0, 0);   // (r > 16: the first tile row is full)
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (4 + s < nks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Y1[s], X1[s], acc, 0, 0, 0);           // (wave-uniform)
    return acc;
}

// In-place inverse of the leading 4 npiv x 4 npiv block of a symmetric positive definite matrix in TL (the rest -- identity
// padding, zero cross terms -- is left alone).  Returns the determinant.  One barrier per pivot block.
__device__ __forceinline__ double rt_sweep_inverse(RtCtx& x, v4d& m, int npiv) {
    double det = 1.0;
    const int lane = x.lane, I = x.I, J = x.J, q = x.q, c = x.c;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        if (p < npiv) {                                          // (wave-uniform)
            const int Ik = p >> 2, vk = p & 3, ck = 4 * (p & 3);
            double* praw = x.sm + kRtPraw + (x.pp & 1) * 128;
            double* pD = x.sm + kRtPD + (x.pp & 1) * 16;
            ++x.pp;
            const bool inblk = (J == Ik) && (c >= ck) && (c < ck + 4);
            if (I == Ik) {                                       // this wave holds pivot rows k0 .. k0 + 3 over its column block
                double val = m[vk];
                if (inblk) {
                    pD[q * 4 + (c - ck)] = val;
                    val = (c - ck == q) ? -1.0 : 0.0;
                }
                praw[q * 32 + 16 * J + c] = val;
            }
            rt_barrier();
            // D = L diag(d) L' (unit lower L) -- every lane, redundantly (the values are wave-uniform) -- and T~ = D^-1 R~ for the
            // lane's column by two substitutions.  No explicit D^-1: the pivot columns of R~ are -I, so the same solve leaves -D^-1
            // in the pivot block.  (A first version inverted D by 2 x 2 blocks and multiplied: ~90 fp64 instructions on a chain
            // that issues one instruction per ~6 cycles -- 156 instructions per pivot, 60 % of the forward sweep; this is ~55.)
            const double2* d2 = reinterpret_cast<const double2*>(pD);
            const double2 r0a = d2[0], r0b = d2[1], r1a = d2[2], r1b = d2[3], r2b = d2[5], r3b = d2[7];
            const int col = 16 * J + c;
            const double p0 = praw[col], p1 = praw[32 + col], p2 = praw[64 + col], p3 = praw[96 + col];
            const double aop = praw[q * 32 + 16 * I + c];        // A operand R~[q][16 I + c]
            const double D00 = r0a.x, D10 = r0a.y, D20 = r0b.x, D30 = r0b.y, D11 = r1a.y, D21 = r1b.x, D31 = r1b.y;
            const double D22 = r2b.x, D32 = r2b.y, D33 = r3b.y;
            const double i0 = fast_rcp3(D00);
            const double l10 = D10 * i0, l20 = D20 * i0, l30 = D30 * i0;
            const double e1 = fma(-l10, D10, D11);
            const double i1 = fast_rcp3(e1);
            const double u21 = fma(-l20, D10, D21), u31 = fma(-l30, D10, D31);
            const double l21 = u21 * i1, l31 = u31 * i1;
            const double e2 = fma(-l21, u21, fma(-l20, D20, D22));
            const double i2 = fast_rcp3(e2);
            const double u32 = fma(-l31, u21, fma(-l30, D20, D32));
            const double l32 = u32 * i2;
            const double e3 = fma(-l32, u32, fma(-l31, u31, fma(-l30, D30, D33)));
            const double i3 = fast_rcp3(e3);
            det *= (D00 * e1) * (e2 * e3);
            const double y1 = fma(-l10, p0, p1);
            const double y2 = fma(-l21, y1, fma(-l20, p0, p2));
            const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, p0, p3)));
            const double tq3 = y3 * i3;
            const double tq2 = fma(-l32, tq3, y2 * i2);
            const double tq1 = fma(-l31, tq3, fma(-l21, tq2, y1 * i1));
            const double tq0 = fma(-l30, tq3, fma(-l20, tq2, fma(-l10, tq1, p0 * i0)));
            const double tq = (q & 2) ? ((q & 1) ? tq3 : tq2) : ((q & 1) ? tq1 : tq0);
            v4d acc = m;
            if (I == Ik) acc[vk] = 0.0;                           // pivot rows (a whole register of this wave) ...
            if (inblk) { acc[0] = 0.0; acc[1] = 0.0; acc[2] = 0.0; acc[3] = 0.0; }   // ... and pivot columns start from zero
            m = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, -tq, acc, 0, 0, 0);
        }
    }
    const int lim = 4 * npiv;
    const bool cin = 16 * J + c < lim;
#pragma unroll
    for (int v = 0; v < 4; ++v) m[v] = (cin && 16 * I + q + 4 * v < lim) ? -m[v] : m[v];
    return det;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Time chunks (round 5).  One workgroup per replicate leaves the machine idle twice over: B = 256 replicates put one wave on each
// SIMD, and that wave is a chain of dependent MFMAs, LDS exchanges and 4-wave barriers (B = 512 takes 13.3 ms where B = 256 takes
// 10.8: a second workgroup per CU runs in the first one's bubbles).  CH = true cuts a replicate's T periods into tile_nc chunks of
// tile_lc periods, one workgroup each -- the scheme of recursion_chunk.hip at workgroup granularity:
//   * forward: chunk c > 0 starts tile_w periods early from a guess (Om_f = Q^-1 + C, xi = 0) -- the filter forgets its start at
//     the rate of its closed loop -- and every chunk but the last runs tile_w periods PAST its end (exact: a continuation), into a
//     private table, so that
//   * backward: it can start tile_w periods late from a guess (P = Z, f = w of the last extra period) and arrive at its own end
//     with the smoothed moments, forgotten likewise;
//   * nothing is assumed: the state each chunk ENTERS its own periods with (forward at its first period, backward at its last)
//     is compared with the exact one its neighbour LEAVES there (tile_chunk_finish_kernel, chunk_tol relative to the largest
//     entry); a replicate with one boundary off is run again by the sequential instantiation (only_if = chunk_fail), which
//     overwrites everything the chunks wrote;
//   * per-chunk parts of the log-likelihood and of the EM sums meet in tile_chunk_finish_kernel, which also keeps the EM
//     bookkeeping and the three sums over f_smooth (they span the chunks).
// Scratch per (replicate, chunk): the private table [tile_w][2][32][32] + [tile_w][32], four boundary states, the parts.
// ------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kTkBst = kRt * kRt + kRt;            // a boundary state: matrix (register-pair layout) + vector
constexpr int kTkPart = 16;                        // log-likelihood part of a chunk (one double used)
constexpr int kTkWmax = 32, kTkNCmax = 16;
__host__ __device__ inline size_t tk_slot_doubles(int W) {
    return (size_t)W * (2 * kRt * kRt + kRt) + 4 * kTkBst + kTkPart + 3 * kRt * kRt;
}

// sum_t f_t f_t', sum_t f_t f_t-1', sum_t f_t-1 f_t-1' (periods 1 .. T; f_0 in f0sh) as products over time: k = 4 periods per
// MFMA, operands straight from f_smooth (agent-scope loads: in the sequential kernel this CU wrote the rows itself)
// the same three products for a kernel that runs AFTER the one that wrote F (tile_chunk_finish_kernel): plain loads, the operands of
// four steps in flight together (one step at a time every step waited for its own trip to L2: 0.34 ms per config-4 EM iteration with
// one workgroup per replicate).  The matrix instructions keep their order: the sums are the same numbers.
__device__ __forceinline__ void tile_gsums_ahead(const double* __restrict__ F, const double* f0sh, int T, int r, int I, int J, int q, int c,
                                                 v4d& G11, v4d& G10, v4d& G00) {
    const int ci = 16 * I + c, cj = 16 * J + c;
    const bool oki = ci < r, okj = cj < r;
    const double f0i = f0sh[ci], f0j = f0sh[cj];
    constexpr int U = 4;
    for (int k0 = 0; k0 < T; k0 += 4 * U) {
        double fi[U], fj[U], pi[U], pj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = k0 + 4 * u + q;
            const bool in = t < T;
            fi[u] = (in && oki) ? F[(size_t)t * r + ci] : 0.0;
            fj[u] = (in && okj) ? F[(size_t)t * r + cj] : 0.0;
            pi[u] = 0.0; pj[u] = 0.0;
            if (in) {
                pi[u] = t == 0 ? f0i : (oki ? F[(size_t)(t - 1) * r + ci] : 0.0);
                pj[u] = t == 0 ? f0j : (okj ? F[(size_t)(t - 1) * r + cj] : 0.0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k0 + 4 * u < T) {
                G11 = __builtin_amdgcn_mfma_f64_16x16x4f64(fi[u], fj[u], G11, 0, 0, 0);
                G10 = __builtin_amdgcn_mfma_f64_16x16x4f64(fi[u], pj[u], G10, 0, 0, 0);
                G00 = __builtin_amdgcn_mfma_f64_16x16x4f64(pi[u], pj[u], G00, 0, 0, 0);
            }
        }
    }
}
__device__ __forceinline__ void tile_gsums(const double* F, const double* f0sh, int T, int r, int I, int J, int q, int c,
                                           v4d& G11, v4d& G10, v4d& G00) {
    const int ci = 16 * I + c, cj = 16 * J + c;                  // this lane's column of F as A operand (I) and as B operand (J)
    const bool oki = ci < r, okj = cj < r;
    const double f0i = f0sh[ci], f0j = f0sh[cj];
    for (int k0 = 0; k0 < T; k0 += 4) {
        const int t = k0 + q;                                    // F row t = period t + 1; "previous" = row t - 1, or f_0
        const bool in = t < T;
        const double fi = (in && oki) ? __hip_atomic_load(F + (size_t)t * r + ci, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        const double fj = (in && okj) ? __hip_atomic_load(F + (size_t)t * r + cj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        double pi = 0.0, pj = 0.0;
        if (in) {
            pi = t == 0 ? f0i : (oki ? __hip_atomic_load(F + (size_t)(t - 1) * r + ci, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0);
            pj = t == 0 ? f0j : (okj ? __hip_atomic_load(F + (size_t)(t - 1) * r + cj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0);
        }
        G11 = __builtin_amdgcn_mfma_f64_16x16x4f64(fi, fj, G11, 0, 0, 0);
        G10 = __builtin_amdgcn_mfma_f64_16x16x4f64(fi, pj, G10, 0, 0, 0);
        G00 = __builtin_amdgcn_mfma_f64_16x16x4f64(pi, pj, G00, 0, 0, 0);
    }
}
}  // namespace

template <bool CH>
__global__ __launch_bounds__(256, CH ? 2 : 1) void recursion_tile_kernel(RecursionArgs a) {
    constexpr int R = kRt, RR = R * R;
    __shared__ __attribute__((aligned(16))) double sm[kRtLds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    RtCtx x;
    x.sm = sm; x.lane = lane; x.w = w; x.I = w >> 1; x.J = w & 1; x.q = lane >> 4; x.c = lane & 15; x.pp = 0;
    const int I = x.I, J = x.J, q = x.q, c = x.c;
    const int NC = CH ? a.tile_nc : 1;
    const int b = CH ? (int)blockIdx.x % a.B : (int)blockIdx.x;
    const int ck = CH ? (int)blockIdx.x / a.B : 0;            // this workgroup's chunk
    if (!CH && a.only_if != nullptr && a.only_if[b] == 0) return;   // (the sequential run of the replicates a chunk boundary rejected)
    const int T = a.T, N = a.N, r = a.r;                      // r: width of the OUTPUT layout (the caller's r, or 32 inside EM)
    const bool first = !CH || ck == 0, lastc = !CH || ck == NC - 1;
    const int Wk = CH ? a.tile_w : 0;
    const int s0 = CH ? ck * a.tile_lc : 0;                    // own periods s0 .. e0 - 1 (steps; the last chunk adds the terminal step T)
    const int e0 = lastc ? T : s0 + a.tile_lc;
    const int tb = first ? 0 : s0 - Wk;                        // forward steps tb .. te - 1 (tb, s0 even: register sets go by parity)
    const int te = lastc ? T + 1 : e0 + Wk;
    const int rs = a.rstate;                                   // the model's state width, 17 .. 31
    const int npiv = (rs + 3) >> 2, nks = npiv;
    const int col = 16 * J + c;
    const bool c31 = (J == 1) && (c == 15);                    // this lane holds column 31: the mean vectors
    int rowv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) rowv[v] = 16 * I + q + 4 * v;
    double* bufA = sm + kRtBufA;
    double* bufB = sm + kRtBufB;
    double* Xk = sm + kRtXk;
    const int tY0 = I, tY1 = 2 + I, tX0 = J, tX1 = 2 + J;      // tiles (0, I), (1, I), (0, J), (1, J)
    long long p_inv = 0, p_p1 = 0, p_p2 = 0; (void)p_inv; (void)p_p1; (void)p_p2;
    const long long t_start = RT_NOW(); (void)t_start;
    // column 31 of an LDS tile buffer, rows of this wave's row block: element v of lane (q, 15) of tile (I, 1)
    auto put_c31 = [&](double* buf, const double (&vec)[4]) {
        if (c31) {
            double2* p = reinterpret_cast<double2*>(buf + (2 * I + 1) * kRtTile);
            p[lane] = make_double2(vec[0], vec[1]);
            p[64 + lane] = make_double2(vec[2], vec[3]);
        }
    };

    const double* bcol = a.bcol + (size_t)b * T * R;
    const double* scol = a.scol + (size_t)b * T;
    const int* nobs = a.nobs + (size_t)b * T;
    const double* ldrow = a.ldrow + (size_t)b * T;
    const double ldfull = a.ldfull[b];
    double* ZJ = a.ZJtab + (size_t)b * (T + 1) * 2 * RR;       // per period: Z (4 tiles), J' (4 tiles), register-pair layout
    double* wtab = a.wtab + (size_t)b * T * R;
    // chunk scratch: the table of the extra forward steps e0 .. e0 + Wk - 1, boundary states, parts
    double* const slot = CH ? a.tile_scr + ((size_t)b * NC + ck) * tk_slot_doubles(Wk) : nullptr;
    double* const xZJ = slot;
    double* const xw = CH ? slot + (size_t)Wk * 2 * RR : nullptr;
    double* const bst = CH ? xw + (size_t)Wk * R : nullptr;
    double* const part = CH ? bst + 4 * kTkBst : nullptr;
    double* const sums = CH ? part + kTkPart : nullptr;
    auto tab_ent = [&](int t) -> double* { return (CH && t >= e0) ? xZJ + (size_t)(t - e0) * 2 * RR : ZJ + (size_t)t * 2 * RR; };
    auto w_ent = [&](int t) -> double* { return (CH && t >= e0) ? xw + (size_t)(t - e0) * R : wtab + (size_t)t * R; };
    auto put_state = [&](int k, const v4d& M, const double (&vec)[4]) {   // boundary state k: 0 / 1 forward entry / exit, 2 / 3 backward
        double* d = bst + (size_t)k * kTkBst;
        st_tile_g(d, w, lane, M);
        if (c31) {
#pragma unroll
            for (int v = 0; v < 4; ++v) d[RR + rowv[v]] = vec[v];
        }
    };
    // C_t rows: the packed leading ct_r x ct_r block (ct_miss_wide2_kernel writes that for this kernel: the padding beyond it carries
    // no loadings, its entries are Cfull's), or the full packed 32 x 32 layout of the other collapse kernels (ct_r = 0)
    const int ctr = a.ct_r > 0 ? a.ct_r : R;
    const int NPc = ctr * (ctr + 1) / 2;
    int pk[4];
    bool cin[4];                                               // the lane's element (i, j) lies inside the block
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int i = rowv[v], j = col;
        cin[v] = i < ctr && j < ctr;
        pk[v] = cin[v] ? ((i >= j) ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i) : 0;
    }

    // ---------------- prologue: Qi = Q^-1, Om_f,0 = P0^-1, K' = A'Qi, Phi = A'Qi A, xi_0 = P0^-1 mu0 -----------------------
    v4d Qi, Omf, Ael, Cf, zero4;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const size_t o = (size_t)b * RR + rowv[v] * R + col;
        Qi[v] = a.Q[o]; Omf[v] = a.P0[o]; Ael[v] = a.A[o]; Cf[v] = a.Cfull[o];
        zero4[v] = 0.0;
    }
    double mu0v[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) mu0v[v] = a.mu0[(size_t)b * R + rowv[v]];
    const double detQ = rt_sweep_inverse(x, Qi, npiv);
    const double detP0 = rt_sweep_inverse(x, Omf, npiv);
    st_tile(bufA, w, lane, Ael);
    st_tile(bufB, w, lane, Qi);
    rt_barrier();
    v4d KtY0, KtY1, Phi;
    {
        const v4d A0i = ld_tile(bufA, tY0, lane), A1i = ld_tile(bufA, tY1, lane);   // A as Y: tiles (kb, I)
        const v4d A0j = ld_tile(bufA, tX0, lane), A1j = ld_tile(bufA, tX1, lane);   // A as X: tiles (kb, J)
        const v4d Q0i = ld_tile(bufB, tY0, lane), Q1i = ld_tile(bufB, tY1, lane);
        const v4d Q0j = ld_tile(bufB, tX0, lane), Q1j = ld_tile(bufB, tX1, lane);
        const v4d Kt = mm_tn(A0i, A1i, Q0j, Q1j, nks, zero4);                       // K' = A'Qi
        const v4d Km = mm_tn(Q0i, Q1i, A0j, A1j, nks, zero4);                       // K  = Qi A
        st_tile(Xk, w, lane, Kt);
        rt_barrier();                                                             // (bufA / bufB are read; Xk is complete)
        st_tile(bufB, w, lane, Km);
        rt_barrier();
        const v4d K0i = ld_tile(bufB, tY0, lane), K1i = ld_tile(bufB, tY1, lane);
        Phi = mm_tn(K0i, K1i, A0j, A1j, nks, zero4);                                // Phi = K'A
        KtY0 = ld_tile(Xk, tY0, lane); KtY1 = ld_tile(Xk, tY1, lane);               // K' as Y: constant for the whole kernel
        st_tile(bufA, w, lane, Omf);
        rt_barrier();                                                             // (K' tiles are read before column 31 is patched)
    }
    put_c31(Xk, mu0v);
    rt_barrier();
    double xi[4];                                              // (column-31 lanes) xi_t, rows of this wave's row block
    double qacc = 0.0;                                         // (column-31 lanes) mu0'xi_0 - xi_T'f_T - sum_t xi_t'w_t
    {
        const v4d Y0 = ld_tile(bufA, tY0, lane), Y1 = ld_tile(bufA, tY1, lane);
        const v4d X0 = ld_tile(Xk, tX0, lane), X1 = ld_tile(Xk, tX1, lane);
        const v4d x0 = mm_tn(Y0, Y1, X0, X1, 8, zero4);       // column 31: Om_f,0 mu0
#pragma unroll
        for (int v = 0; v < 4; ++v) { xi[v] = x0[v]; qacc = fma(mu0v[v], x0[v], qacc); }
    }
    if (CH && !first) {                                        // (uniform) a later chunk: the guess its warm-up periods forget
        qacc = 0.0;
#pragma unroll
        for (int v = 0; v < 4; ++v) { xi[v] = 0.0; Omf[v] = Qi[v] + Cf[v]; }
    }
    rt_barrier();                                           // (every wave has read Xk before the next patch)
    put_c31(Xk, xi);

    // ---------------- forward sweep (t = T: the terminal inverse P_T = Om_f,T^-1 and f_T = P_T xi_T) ----------------------
    // Operands of period t live in register set t & 1 and are RE-LOADED for period t + 2 right behind their last use (they are
    // consumed at the very end of a period): two periods of lead, no staging copies (a first version prefetched a chunk into a
    // second set and copied it over: 22 v_mov per two periods on a chain that is bound by its instruction count).
    static_assert(kRtCh == 2, "register sets alternate by the parity of the period");
    double cb[kRtCh][4], cc[kRtCh][4], cs[kRtCh], cl[kRtCh];
    int cn[kRtCh];
    const double* Ctb = a.Ct ? a.Ct + (size_t)b * T * NPc : nullptr;
    auto load_fwd = [&](int s, int t) {
        t = t < T ? t : T - 1;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            cb[s][v] = bcol[(size_t)t * R + rowv[v]];
            cc[s][v] = Ctb ? Ctb[(size_t)t * NPc + pk[v]] : 0.0;
        }
        cs[s] = scol[t]; cn[s] = nobs[t]; cl[s] = ldrow[t];
    };
    double ssum = 0.0, nsum = 0.0, ldsum = 0.0;
    LogProd detprod;
    v4d Ps = zero4;
    double fs[4] = {0.0, 0.0, 0.0, 0.0};
    double detOmT = 1.0;
    load_fwd(0, tb);
    load_fwd(1, tb + 1);
    for (int ch = tb / kRtCh; ch * kRtCh < te; ++ch) {
#pragma unroll
        for (int s = 0; s < kRtCh; ++s) {
            const int t = ch * kRtCh + s;
            if (t < te) {                                        // (uniform)
                const bool last = (t == T);
                const bool own = !CH || (t >= s0 && t < e0);     // (uniform) a period of this chunk: it counts, its table entry is THE entry
                if (CH) {
                    if (!first && t == s0) put_state(0, Omf, xi);   // what the warm-up arrived at ...
                    if (!lastc && t == e0) put_state(1, Omf, xi);   // ... is checked against the exact state the chunk before leaves
                }
                v4d Z;
#pragma unroll
                for (int v = 0; v < 4; ++v) Z[v] = last ? Omf[v] : Omf[v] + Phi[v];
                RT_TICK(p_inv);
                const double dM = rt_sweep_inverse(x, Z, npiv);
                RT_TOCK(p_inv);
                RT_TICK(p_p1);
                st_tile(bufA, w, lane, Z);
                rt_barrier();                                 // E1: Z in LDS; column 31 of Xk holds xi_t
                const v4d ZY0 = ld_tile(bufA, tY0, lane), ZY1 = ld_tile(bufA, tY1, lane);
                const v4d X0 = ld_tile(Xk, tX0, lane), X1 = ld_tile(Xk, tX1, lane);
                const v4d Jaug = mm_tn(ZY0, ZY1, X0, X1, nks, zero4);          // [J | w] = Z'[K' | xi]
                if (last) {
                    RT_TOCK(p_p1);
                    detOmT = dM;
                    Ps = Z;
#pragma unroll
                    for (int v = 0; v < 4; ++v) { fs[v] = Jaug[v]; qacc = fma(-xi[v], Jaug[v], qacc); }   // f_T = P_T xi_T (column 31)
                } else {
                    if (own) detprod.mul(dM);
                    const v4d ZX0 = ld_tile(bufA, tX0, lane), ZX1 = ld_tile(bufA, tX1, lane);
                    const v4d Jt = mm_tn(KtY0, KtY1, ZX0, ZX1, nks, zero4);    // J' = K Z
                    st_tile(bufB, w, lane, Jaug);
                    if (!CH || t >= s0) {                        // (uniform; warm-up periods leave no entry)
                        double* ent = tab_ent(t);
                        st_tile_g(ent, w, lane, Z);
                        st_tile_g(ent + RR, w, lane, Jt);
                        if (c31) {
                            double* we = w_ent(t);
#pragma unroll
                            for (int v = 0; v < 4; ++v) we[rowv[v]] = Jaug[v];
                        }
                    }
                    RT_TOCK(p_p1);
                    RT_TICK(p_p2);
                    rt_barrier();                             // E2: [J | w] in LDS
                    const v4d JX0 = ld_tile(bufB, tX0, lane), JX1 = ld_tile(bufB, tX1, lane);
                    const v4d prod = mm_tn(KtY0, KtY1, JX0, JX1, nks, zero4);  // K [J | w]
                    const bool full = (cn[s] == N);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const double omp = c31 ? Qi[v] : Qi[v] - prod[v];     // column 31 is padding: Qi's own entries
                        Omf[v] = omp + ((full || !cin[v]) ? Cf[v] : cc[s][v]);
                    }
                    if (c31) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            if (own) qacc = fma(-xi[v], Jaug[v], qacc);   // xi_t'w_t
                            xi[v] = prod[v] + cb[s][v];          // xi_t+1 = K w_t + b_t
                        }
                    }
                    put_c31(Xk, xi);                             // (read again after the next barrier at the earliest)
                    if (own) {
                        ssum += cs[s];
                        nsum += (double)cn[s];
                        ldsum += full ? ldfull : cl[s];
                    }
                    load_fwd(s, t + 2);                          // (this set's operands are consumed)
                    RT_TOCK(p_p2);
                }
            }
        }
    }

    const long long t_fwd = RT_NOW(); (void)t_fwd;
    // ---------------- log-likelihood, EM bookkeeping ------------------------------------------------------------------
    {
        if (c31) sm[kRtRed + 4 * I + q] = qacc;
        rt_barrier();
        if (tid == 0) {
            double qd = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) qd += sm[kRtRed + k];
            if (CH) {
                // this chunk's part of  n log 2 pi + sum log det R_t + log dets + sum s_t + quadratic terms  (tile_chunk_finish_kernel adds them up)
                double LD = detprod.log_value();
                if (first) LD += log(detP0) + (double)T * log(detQ);
                if (lastc) LD += log(detOmT);
                part[0] = nsum * kLog2PiT + ldsum + LD + ssum + qd;
            } else {
                const double LD = log(detOmT) + log(detP0) + (double)T * log(detQ) + detprod.log_value();
                const double ll = -0.5 * (nsum * kLog2PiT + ldsum + LD + ssum + qd);
                a.loglik[b] = ll;
                if (a.ncov) a.ncov[b] = T;
                if (a.active) {                                  // EM bookkeeping, as recursion_kernel
                    const bool was = a.k == 0 ? true : (a.active[b] != 0);
                    bool go = was;
                    if (was && a.k >= 1 && a.tol > 0.0) {
                        const double llp = a.ll_path[(size_t)b * a.max_iter + a.k - 1];
                        go = !((ll - llp) / (0.5 * (fabs(ll) + fabs(llp))) < a.tol);
                    }
                    if (was) { a.ll_path[(size_t)b * a.max_iter + a.k] = ll; a.iters[b] = a.k + 1; }
                    a.active[b] = go ? 1 : 0;
                }
            }
        }
    }

    // ---------------- backward sweep -------------------------------------------------------------------------------------
    const int npr = r * (r + 1) / 2;
    int poff[4];                                               // packed offset of the lane's element (i, col), -1: not in the output
#pragma unroll
    for (int v = 0; v < 4; ++v) poff[v] = (rowv[v] < r && col <= rowv[v]) ? rowv[v] * (rowv[v] + 1) / 2 + col : -1;
    double* const fsm = a.f_smooth + (size_t)b * T * r;
    double* const Psm = a.P_smooth ? a.P_smooth + (size_t)b * T * npr : nullptr;
    auto emit = [&](int trow, const v4d& P, const double (&f)[4]) {   // smoothed moments of period trow + 1
        if (c31) {
            double* fr = fsm + (size_t)trow * r;
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (rowv[v] < r) fr[rowv[v]] = f[v];
        }
        if (Psm) {
            double* pr = Psm + (size_t)trow * npr;
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (poff[v] >= 0) pr[poff[v]] = P[v];
        }
    };
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // (one workgroup = one CU = one write-through L1: no agent-scope L2 write-back)
    __syncthreads();                                           // (table and w_t stores of the forward sweep are visible: one CU, one L1)
    if (CH && !lastc) {                                        // (uniform) the guess the backward warm-up forgets: (Z, w) of the last extra period
        Ps = ld_tile_g(xZJ + (size_t)(Wk - 1) * 2 * RR, w, lane);
#pragma unroll
        for (int v = 0; v < 4; ++v) fs[v] = xw[(size_t)(Wk - 1) * R + rowv[v]];
    } else {
        emit(T - 1, Ps, fs);
    }
    const bool em = a.S11 != nullptr;
    const v4d PT = Ps;
    v4d SP = lastc ? Ps : zero4, SU = zero4;                   // sum_t P_t (periods 1 .. T), sum of the lag-one covariances: this chunk's part
    // (Z, J') of step t and w_t in register set t & 1, re-loaded for step t - 2 behind their last use: the B operands of the
    // first product early in the step, Z / the A operands / w_t behind the second
    v4d zc[kRtCh], jx0[kRtCh], jx1[kRtCh], jy0[kRtCh], jy1[kRtCh];
    double wc[kRtCh][4];
    auto load_bwd_x = [&](int s, int t) {
        t = t > s0 ? t : s0;
        const double* ent = tab_ent(t) + RR;
        jx0[s] = ld_tile_g(ent, tX0, lane); jx1[s] = ld_tile_g(ent, tX1, lane);
    };
    auto load_bwd_y = [&](int s, int t) {
        t = t > s0 ? t : s0;
        const double* ent = tab_ent(t);
        zc[s] = ld_tile_g(ent, w, lane);
        jy0[s] = ld_tile_g(ent + RR, tY0, lane); jy1[s] = ld_tile_g(ent + RR, tY1, lane);
        const double* we = w_ent(t);
#pragma unroll
        for (int v = 0; v < 4; ++v) wc[s][v] = we[rowv[v]];
    };
    const int tl = lastc ? T - 1 : e0 + Wk - 1;                // backward steps tl .. s0
    load_bwd_x(tl & 1, tl); load_bwd_y(tl & 1, tl);
    load_bwd_x((tl - 1) & 1, tl - 1); load_bwd_y((tl - 1) & 1, tl - 1);   // (one step only: that step again, unused)
    for (int ch = tl / kRtCh; ch * kRtCh >= s0; --ch) {
#pragma unroll
        for (int s = kRtCh - 1; s >= 0; --s) {
            const int t = ch * kRtCh + s;                        // step t: from period t + 1 to period t (t = 0: the initial state)
            if (t <= tl && t >= s0) {                            // (uniform)
                const bool own = !CH || t < e0;
                st_tile(bufA, w, lane, Ps);
                rt_barrier();                                 // E3
                const v4d PY0 = ld_tile(bufA, tY0, lane), PY1 = ld_tile(bufA, tY1, lane);
                const v4d U = mm_tn(PY0, PY1, jx0[s], jx1[s], nks, zero4);      // U = P_s J' = Cov(f_t+1, f_t | X)
                load_bwd_x(s, t - 2);
                v4d Uaug = U;
                if (c31) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) Uaug[v] = fs[v];                // column 31 := f_t+1
                }
                st_tile(bufB, w, lane, Uaug);
                rt_barrier();                                 // E4
                const v4d UX0 = ld_tile(bufB, tX0, lane), UX1 = ld_tile(bufB, tX1, lane);
                v4d Pn = mm_tn(jy0[s], jy1[s], UX0, UX1, nks, zc[s]);           // Z + J [U | f_t+1]
                if (c31) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        fs[v] = wc[s][v] + (Pn[v] - zc[s][v]);                  // f_t = w_t + J f_t+1
                        Pn[v] = zc[s][v];                                       // column 31 is padding: Z's own entries
                    }
                }
                Ps = Pn;
                load_bwd_y(s, t - 2);
                if (em && own) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) { SU[v] += U[v]; if (t > 0) SP[v] += Pn[v]; }
                }
                if (own && t > 0) emit(t - 1, Ps, fs);
                if (CH && !lastc && t == e0) put_state(2, Ps, fs);   // what the backward warm-up arrived at (period e0) ...
            }
        }
    }
    if (CH && !first) put_state(3, Ps, fs);                    // ... is checked against the exact moments the chunk behind leaves (its period s0)
#ifdef DFM_TILE_PROF
    if (b == 5 && tid == 0)
        printf("TILEPROF T=%d (10 ns ticks): total %lld  forward %lld (inverse %lld, Z exchange + 2 products %lld, J exchange + product + update %lld)  backward %lld\n",
               T, RT_NOW() - t_start, t_fwd - t_start, p_inv, p_p1, p_p2, RT_NOW() - t_fwd);
#endif
    if (!em) return;

    if (CH) {                                                  // this chunk's parts of the sums; the rest is tile_chunk_finish_kernel's
        st_tile_g(sums, w, lane, SP);
        st_tile_g(sums + RR, w, lane, SU);
        if (lastc) st_tile_g(sums + 2 * RR, w, lane, PT);
        if (first) {
#pragma unroll
            for (int v = 0; v < 4; ++v) a.P0s[(size_t)b * RR + rowv[v] * R + col] = Ps[v];
            if (c31) {
#pragma unroll
                for (int v = 0; v < 4; ++v) a.f0s[(size_t)b * R + rowv[v]] = fs[v];
            }
        }
        return;
    }
    // ---------------- EM sums: S11 = sum E[f_t f_t'], S10 = sum E[f_t f_t-1'], S00 (periods 1 .. T; f_0 = fs, P_0 = Ps) ----
    if (c31) {
#pragma unroll
        for (int v = 0; v < 4; ++v) sm[kRtRed + rowv[v]] = fs[v];               // f_0 (32 doubles)
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // (one workgroup = one CU = one write-through L1: no agent-scope L2 write-back)                                           // (f_smooth rows written by the column-31 lanes are read by every wave)
    __syncthreads();
    v4d G11 = zero4, G10 = zero4, G00 = zero4;
    tile_gsums(a.f_smooth + (size_t)b * T * r, sm + kRtRed, T, r, I, J, q, c, G11, G10, G00);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const size_t o = (size_t)b * RR + rowv[v] * R + col;
        a.S11[o] = SP[v] + G11[v];
        a.S10[o] = SU[v] + G10[v];
        a.S00[o] = (SP[v] - PT[v] + Ps[v]) + G00[v];
        a.P0s[o] = Ps[v];
    }
    if (c31) {
#pragma unroll
        for (int v = 0; v < 4; ++v) a.f0s[(size_t)b * R + rowv[v]] = fs[v];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// recursion_tile1_kernel (round 6): the same recursion, the same chunks, the same scratch -- ONE WAVE per (replicate, chunk) that holds
// ALL FOUR tiles of every matrix (t[I][J]: 16 registers per lane).  What recursion_tile_kernel pays for its four waves: nine 4-wave
// barriers and four LDS exchanges per period around 30 matrix instructions per wave -- 6.9 us per period of a workgroup, a third of the
// matrix pipe busy with two workgroups per CU -- and a table of (Z, J') in the tile layout, 16 KB written and 16 KB read back per period:
// 17 GB per config-4 pass.  Here:
//   * a D-layout matrix IS the B operand of a product and the A operand of its transpose, tile by tile, as it stands: no exchange, no
//     barrier.  What is left in LDS: the pivot rows of the block sweep (512 bytes per pivot, the wave's own in-order LDS queue; the pivot
//     block itself comes out of the registers by v_readlane), two constant matrices that are only ever used element-wise (Phi, Qi) and one
//     stage for the rows that arrive by LDS-DMA -- 19.8 KB per wave at r <= 20;
//   * the table entry of a period is the PACKED lower triangle of the leading block of Z (210 doubles at r = 20); the backward step
//     rebuilds J' = K Z (20 matrix instructions) and the padding is the identity: 2 x 1.7 KB of traffic per period instead of 2 x 16 KB;
//   * C_t, b_t (forward) and Z, w_t (backward) are fetched a period ahead by LDS-DMA (no registers in flight), the smoothed moments of a
//     step are stored at the start of the next one, the three scalars of a period by per-lane loads: every wait of the chain finds only
//     traffic a period old.
// 120 (forward 60, backward 60) matrix instructions per period and wave, four times as many independent chains per CU: 7.16 -> 4.8 ms per
// 256 config-4 passes.  What bounds it now is the block sweep: 0.7 us per pivot block of ~170 dependent scalar-chain instructions
// (LDL' of the 4 x 4 pivot, two substitutions) that one wave per SIMD cannot hide; a second wave per SIMD needs <= 256 registers and
// spills (DFM_T1_OCC=2: measured slower).  Batches too small to put a wave on every SIMD keep recursion_tile_kernel.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct M32 { v4d t[2][2]; };                       // t[I][J]: lane (q, c), register v = element (16 I + q + 4 v, 16 J + c)

constexpr int kT1Praw = 0;                         // [32 columns][4 pivot rows]
constexpr int kT1Phi = 128;                        // two matrices in the register-pair tile layout
constexpr int kT1Qi = kT1Phi + 4 * kRtTile;
constexpr int kT1Cf = kT1Qi + 4 * kRtTile;         // (end of the constants: the stage starts here)

// the wave's own LDS writes are visible to its later reads (one in-order queue): only the compiler must not move them
__device__ __forceinline__ void t1_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void t1_ld_g(const double* tab, int lane, M32& m) {
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J) m.t[I][J] = ld_tile_g(tab, 2 * I + J, lane);
}
__device__ __forceinline__ void t1_st_g(double* tab, int lane, const M32& m) {
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J) st_tile_g(tab, 2 * I + J, lane, m.t[I][J]);
}
__device__ __forceinline__ void t1_ld_s(const double* buf, int lane, M32& m) {
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J) m.t[I][J] = ld_tile(buf, 2 * I + J, lane);
}
__device__ __forceinline__ void t1_st_s(double* buf, int lane, const M32& m) {
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J) st_tile(buf, 2 * I + J, lane, m.t[I][J]);
}

// out = init + Y'X over the first nks k-steps: tile (I, J) takes register s of Y's tile (kb, I) as the A operand and register s of
// X's tile (kb, J) as the B operand -- the four accumulators are independent chains of the matrix pipe
__device__ __forceinline__ void t1_mm(const M32& Y, const M32& X, int nks, const M32& init, M32& out) {
    v4d a00 = init.t[0][0], a01 = init.t[0][1], a10 = init.t[1][0], a11 = init.t[1][1];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[0][0][s], X.t[0][0][s], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[0][0][s], X.t[0][1][s], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[0][1][s], X.t[0][0][s], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[0][1][s], X.t[0][1][s], a11, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (4 + s < nks) {                                       // (wave-uniform)
            a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[1][0][s], X.t[1][0][s], a00, 0, 0, 0);
            a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[1][0][s], X.t[1][1][s], a01, 0, 0, 0);
            a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[1][1][s], X.t[1][0][s], a10, 0, 0, 0);
            a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[1][1][s], X.t[1][1][s], a11, 0, 0, 0);
        }
    out.t[0][0] = a00; out.t[0][1] = a01; out.t[1][0] = a10; out.t[1][1] = a11;
}

// out = Y'X (accumulators start from the inline constant 0: no registers of zeros)
__device__ __forceinline__ void t1_mm0(const M32& Y, const M32& X, int nks, M32& out) {
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    v4d a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[0][0][0], X.t[0][0][0], z, 0, 0, 0);
    v4d a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[0][0][0], X.t[0][1][0], z, 0, 0, 0);
    v4d a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[0][1][0], X.t[0][0][0], z, 0, 0, 0);
    v4d a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[0][1][0], X.t[0][1][0], z, 0, 0, 0);
#pragma unroll
    for (int s = 1; s < 4; ++s) {
        a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[0][0][s], X.t[0][0][s], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[0][0][s], X.t[0][1][s], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[0][1][s], X.t[0][0][s], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[0][1][s], X.t[0][1][s], a11, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (4 + s < nks) {                                       // (wave-uniform)
            a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[1][0][s], X.t[1][0][s], a00, 0, 0, 0);
            a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[1][0][s], X.t[1][1][s], a01, 0, 0, 0);
            a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[1][1][s], X.t[1][0][s], a10, 0, 0, 0);
            a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(Y.t[1][1][s], X.t[1][1][s], a11, 0, 0, 0);
        }
    out.t[0][0] = a00; out.t[0][1] = a01; out.t[1][0] = a10; out.t[1][1] = a11;
}

__device__ __forceinline__ double t1_readlane(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

// rt_sweep_inverse in one wave: the pivot block D comes out of the registers by v_readlane (the pivot is unrolled: constant lanes),
// the four pivot rows of the lane's two columns through 512 bytes of LDS, the A operand is the lane's own published value
__device__ __forceinline__ double t1_sweep_inverse(double* praw, M32& m, int npiv, int q, int c) {
    double det = 1.0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        if (p < npiv) {                                          // (wave-uniform)
            const int Ik = p >> 2, vk = p & 3, ck = 4 * (p & 3);
            const bool inblk = (c >= ck) && (c < ck + 4);        // (in tile column Ik)
            double val[2];
            val[0] = m.t[Ik][0][vk];
            val[1] = m.t[Ik][1][vk];
            const double vd = val[Ik];
            const double D00 = t1_readlane(vd, ck), D10 = t1_readlane(vd, 16 + ck), D20 = t1_readlane(vd, 32 + ck), D30 = t1_readlane(vd, 48 + ck);
            const double D11 = t1_readlane(vd, 16 + ck + 1), D21 = t1_readlane(vd, 32 + ck + 1), D31 = t1_readlane(vd, 48 + ck + 1);
            const double D22 = t1_readlane(vd, 32 + ck + 2), D32 = t1_readlane(vd, 48 + ck + 2), D33 = t1_readlane(vd, 48 + ck + 3);
            if (inblk) val[Ik] = (c - ck == q) ? -1.0 : 0.0;
            praw[c * 4 + q] = val[0];
            praw[(16 + c) * 4 + q] = val[1];
            t1_lds_fence();
            const double2* pr = reinterpret_cast<const double2*>(praw);
            const double2 pa0 = pr[c * 2], pb0 = pr[c * 2 + 1], pa1 = pr[(16 + c) * 2], pb1 = pr[(16 + c) * 2 + 1];
            const double i0 = fast_rcp3(D00);
            const double l10 = D10 * i0, l20 = D20 * i0, l30 = D30 * i0;
            const double e1 = fma(-l10, D10, D11);
            const double i1 = fast_rcp3(e1);
            const double u21 = fma(-l20, D10, D21), u31 = fma(-l30, D10, D31);
            const double l21 = u21 * i1, l31 = u31 * i1;
            const double e2 = fma(-l21, u21, fma(-l20, D20, D22));
            const double i2 = fast_rcp3(e2);
            const double u32 = fma(-l31, u21, fma(-l30, D20, D32));
            const double l32 = u32 * i2;
            const double e3 = fma(-l32, u32, fma(-l31, u31, fma(-l30, D30, D33)));
            const double i3 = fast_rcp3(e3);
            det *= (D00 * e1) * (e2 * e3);
            double tq[2];
#pragma unroll
            for (int J = 0; J < 2; ++J) {
                const double p0 = J ? pa1.x : pa0.x, p1 = J ? pa1.y : pa0.y, p2 = J ? pb1.x : pb0.x, p3 = J ? pb1.y : pb0.y;
                const double y1 = fma(-l10, p0, p1);
                const double y2 = fma(-l21, y1, fma(-l20, p0, p2));
                const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, p0, p3)));
                const double tq3 = y3 * i3;
                const double tq2 = fma(-l32, tq3, y2 * i2);
                const double tq1 = fma(-l31, tq3, fma(-l21, tq2, y1 * i1));
                const double tq0 = fma(-l30, tq3, fma(-l20, tq2, fma(-l10, tq1, p0 * i0)));
                tq[J] = -((q & 2) ? ((q & 1) ? tq3 : tq2) : ((q & 1) ? tq1 : tq0));
            }
            t1_lds_fence();                                      // (the rows are read before the next pivot publishes its own)
            m.t[Ik][0][vk] = 0.0;                                // pivot rows ...
            m.t[Ik][1][vk] = 0.0;
#pragma unroll
            for (int I = 0; I < 2; ++I)
                if (inblk) { m.t[I][Ik][0] = 0.0; m.t[I][Ik][1] = 0.0; m.t[I][Ik][2] = 0.0; m.t[I][Ik][3] = 0.0; }   // ... and pivot columns start from zero
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J) m.t[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[I], tq[J], m.t[I][J], 0, 0, 0);
        }
    }
    const int lim = 4 * npiv;
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) m.t[I][J][v] = (16 * J + c < lim && 16 * I + q + 4 * v < lim) ? -m.t[I][J][v] : m.t[I][J][v];
    return det;
}
}  // namespace

namespace {
using t1_lds_cptr = __attribute__((address_space(3))) char*;
// One LDS-DMA instruction: lane l copies 16 bytes from gbase + voff to LDS byte lds_dst + 16 l -- nothing lands in a VGPR until the
// arithmetic asks for it.  gbase, lds_dst wave-uniform.  t1_dma16: lanes 0 .. 15 only (a 256-byte row).
__device__ __forceinline__ void t1_dma(const void* gbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void t1_dma16(const void* gbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    unsigned long long ex;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_mov_b64 exec, 0xffff\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void t1_wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr int kT1QiC = kT1Qi;                      // Qi, plus Cfull where the lane's element lies outside the C_t block
// LDS of a wave: the pivot rows, Phi, QiC, and ONE stage [rows area: a packed C_t / Z row in whole KBs of DMA (nR of them) | a 32-vector
// (b_t / w_t) | the zero, the one] -- 19.8 KB at r <= 20: eight waves per CU, two per SIMD (the chain of a period is latency, not issue)
constexpr int kT1Stg = kT1Cf;
__host__ __device__ inline int t1_rows_kb(int ct_r, int rstate) {
    const int ctr = ct_r > 0 ? ct_r : kRt, rb = 4 * ((rstate + 3) / 4);
    const int nC = (ctr * (ctr + 1) / 2 * 8 + 1023) >> 10, nZ = (rb * (rb + 1) / 2 * 8 + 1023) >> 10;
    return nC > nZ ? nC : nZ;
}
__host__ __device__ inline size_t t1_lds_bytes(int ct_r, int rstate) { return (size_t)(kT1Stg + t1_rows_kb(ct_r, rstate) * 128 + 32 + 8) * sizeof(double); }
}  // namespace

// keeps the address arithmetic of a rarely taken branch inside the branch (hoisted out of the period loop it would sit in registers)
template <typename P>
__device__ __forceinline__ P* t1_here(P* p) { asm volatile("" : "+s"(p)); return p; }

#ifndef DFM_T1_OCC
#define DFM_T1_OCC 1        // waves per SIMD the pass instantiation is compiled for (2: 256 registers -- measured slower: spills)
#endif
template <bool EM>
__global__ __launch_bounds__(64, (EM ? 1 : DFM_T1_OCC)) void recursion_tile1_kernel(RecursionArgs a) {
    constexpr int R = kRt, RR = R * R;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lane = threadIdx.x, q = lane >> 4, c = lane & 15;
    const int NC = a.tile_nc;
    const int b = (int)blockIdx.x % a.B;
    const int ck = (int)blockIdx.x / a.B;                     // this wave's chunk
    const int T = a.T, N = a.N, r = a.r;
    const bool first = ck == 0, lastc = ck == NC - 1;
    const int Wk = a.tile_w;
    const int s0 = ck * a.tile_lc;
    const int e0 = lastc ? T : s0 + a.tile_lc;
    const int tb = first ? 0 : s0 - Wk;
    const int te = lastc ? T + 1 : e0 + Wk;
    const int rs = a.rstate;
    const int npiv = (rs + 3) >> 2, nks = npiv;
    const bool c31 = (c == 15);                                // column 31 lives in the tiles (., 1) of these lanes
    const int stgB = t1_rows_kb(a.ct_r, rs) * 128, stgZ = stgB + 32, stgOne = stgZ + 1;   // the stage: rows | 32-vector | zero, one
    double* praw = sm + kT1Praw;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(t1_lds_cptr)(reinterpret_cast<char*>(sm)));
    int zero_v;                                                // a zero the compiler cannot see through: per-lane (vector) loads of
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));            // wave-uniform scalars -- their waits are vmcnt's, not the LDS queue's

    const double* bcol = a.bcol + (size_t)b * T * R;
    const double* scol = a.scol + (size_t)b * T;
    const int* nobs = a.nobs + (size_t)b * T;
    const double* ldrow = a.ldrow + (size_t)b * T;
    double* ZJ = a.ZJtab + (size_t)b * (T + 1) * 2 * RR;
    double* wtab = a.wtab + (size_t)b * T * R;
    double* const slot = a.tile_scr + ((size_t)b * NC + ck) * tk_slot_doubles(Wk);
    double* const xZJ = slot;
    double* const xw = slot + (size_t)Wk * 2 * RR;
    double* const bst = xw + (size_t)Wk * R;
    double* const part = bst + 4 * kTkBst;
    double* const sums = part + kTkPart;
    auto tab_ent = [&](int t) -> double* { return t >= e0 ? xZJ + (size_t)(t - e0) * 2 * RR : ZJ + (size_t)t * 2 * RR; };
    auto w_ent = [&](int t) -> double* { return t >= e0 ? xw + (size_t)(t - e0) * R : wtab + (size_t)t * R; };
    auto put_state = [&](int k, const M32& M, const double (&vec)[2][4]) {
        double* d = t1_here(bst + (size_t)k * kTkBst);
        t1_st_g(d, lane, M);
        if (c31) {
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int v = 0; v < 4; ++v) d[RR + 16 * I + q + 4 * v] = vec[I][v];
        }
    };
    const int ctr = a.ct_r > 0 ? a.ct_r : R;
    const int NPc = ctr * (ctr + 1) / 2;                       // (even for every ctr the library uses: rows are whole 16-byte units)
    const unsigned rowB = (unsigned)NPc * 8u;
    const int nC = (int)((rowB + 1023u) >> 10);                // DMA instructions per C_t row
    int pkB[2][2][4];                                          // byte offset of the lane's element in a stage (the zero outside the block)
    unsigned cinm = 0;                                         // bit 8 I + 4 J + v: the element lies inside the C_t block
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = 16 * I + q + 4 * v, j = 16 * J + c;
                const bool in = i < ctr && j < ctr;
                if (in) cinm |= 1u << (8 * I + 4 * J + v);
                pkB[I][J][v] = in ? 8 * ((i >= j) ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i) : 8 * stgZ;
            }

    // The table entry of a period is the PACKED lower triangle of the leading rb x rb block of Z (rb = 4 npiv: 210 doubles at r = 20 where
    // the tile layout of (Z, J') took 2048) -- the rest of Z is the identity padding and J' = K Z is rebuilt by the backward step: an entry
    // written and read back is 2 x 1.7 KB of HBM traffic instead of 2 x 16 KB (the kernel was bound by that traffic: 17 GB per pass).
    const int rb = 4 * npiv;
    const unsigned rowZB = (unsigned)(rb * (rb + 1) / 2) * 8u;
    const int nZ = (int)((rowZB + 1023u) >> 10);
    auto zoff = [&](int I, int J, int v, bool gather) -> int {
        const int i = 16 * I + q + 4 * v, j = 16 * J + c;
        const bool in = i < rb && j < rb;
        (void)gather;
        return in ? 8 * ((i >= j) ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i) : 8 * (i == j ? stgOne : stgZ);
    };
    // Z (tile layout) -> its packed row at `ent`: every lane stores the elements it holds of the block's lower triangle where they belong
    // (12 scattered 8-byte stores; 1.7 KB per period -- and Z is dead right behind the product that uses it, not a period's update later)
    auto put_packed = [&](double* ent, const M32& Zm) {
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 16 * I + q + 4 * v, j = 16 * J + c;
                    if (i < rb && j <= i) ent[i * (i + 1) / 2 + j] = Zm.t[I][J][v];
                }
    };
    // the packed row of an entry (+ the 32-vector at vsrc) by LDS-DMA into stage (byte address dst)
    auto dma_packed = [&](const double* ent, const double* vsrc, unsigned dst) {
        for (int k = 0; k < nZ; ++k) {                           // (uniform)
            unsigned vo = (unsigned)lane * 16u + ((unsigned)k << 10);
            vo = vo < rowZB - 16u ? vo : rowZB - 16u;
            t1_dma(ent, vo, dst + ((unsigned)k << 10));
        }
        t1_dma16(vsrc, (unsigned)(lane & 15) * 16u, dst + 8u * (unsigned)stgB);
    };
    auto get_packed = [&](const double* stg, M32& Zm) {
        const char* sb = reinterpret_cast<const char*>(stg);
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J < 2; ++J)
#pragma unroll
                for (int v = 0; v < 4; ++v) Zm.t[I][J][v] = *reinterpret_cast<const double*>(sb + zoff(I, J, v, true));
    };

    // ---------------- prologue ---------------------------------------------------------------------------------------------
    M32 Omf, Kt;
    double xi[2][4];                                           // (column-31 lanes) xi_t
    double qacc = 0.0;
    double ldet0 = 0.0;                                        // log det P0 + T log det Q (the first chunk's part)
    {
        M32 Qi, Ael, Cf, Km, Phi;
        double mu0v[2][4];
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J < 2; ++J)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const size_t o = (size_t)b * RR + (16 * I + q + 4 * v) * R + 16 * J + c;
                    Qi.t[I][J][v] = a.Q[o]; Omf.t[I][J][v] = a.P0[o]; Ael.t[I][J][v] = a.A[o]; Cf.t[I][J][v] = a.Cfull[o];
                }
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int v = 0; v < 4; ++v) mu0v[I][v] = a.mu0[(size_t)b * R + 16 * I + q + 4 * v];
        double dets[2];
#pragma unroll 1
        for (int k = 0; k < 2; ++k) {                            // Q^-1, then P0^-1: one copy of the sweep's code
            M32 m = k ? Omf : Qi;
            dets[k] = t1_sweep_inverse(praw, m, npiv, q, c);
            if (k) Omf = m; else Qi = m;
        }
        ldet0 = log(dets[1]) + (double)T * log(dets[0]);
        t1_mm0(Ael, Qi, nks, Kt);                                // K' = A'Qi
        t1_mm0(Qi, Ael, nks, Km);                                // K  = Qi A
        t1_mm0(Km, Ael, nks, Phi);                               // Phi = K'A
        t1_st_s(sm + kT1Phi, lane, Phi);
        if (first) {
            M32 X = Kt, x0;
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int v = 0; v < 4; ++v) X.t[I][1][v] = c31 ? mu0v[I][v] : Kt.t[I][1][v];
            t1_mm0(Omf, X, 8, x0);                               // column 31: Om_f,0 mu0
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int v = 0; v < 4; ++v) { xi[I][v] = x0.t[I][1][v]; qacc = fma(mu0v[I][v], x0.t[I][1][v], qacc); }
        } else {                                                 // (uniform) a later chunk: the guess its warm-up periods forget
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int v = 0; v < 4; ++v) Omf.t[I][J][v] = Qi.t[I][J][v] + Cf.t[I][J][v];
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int v = 0; v < 4; ++v) xi[I][v] = 0.0;
        }
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J < 2; ++J)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (!((cinm >> (8 * I + 4 * J + v)) & 1u)) Qi.t[I][J][v] += Cf.t[I][J][v];
        t1_st_s(sm + kT1QiC, lane, Qi);
        if (lane < 2) sm[kT1Stg + stgZ + lane] = (double)lane;   // the zero, the one
        t1_lds_fence();
    }

    // ---------------- forward sweep ----------------------------------------------------------------------------------------
    // C_t and b_t of period t arrive in the stage by LDS-DMA, issued a period ahead right behind the last use of the stage (the packing of
    // Z); the three scalars of a period by per-lane loads a period ahead.
    const double* Ctb = a.Ct ? a.Ct + (size_t)b * T * NPc : nullptr;
    double cs_n, cl_n;
    int cn_n;
    auto issue_fwd = [&](int t) {
        t = t < T ? t : T - 1;
        const unsigned dst = lds0 + 8u * (unsigned)kT1Stg;
        if (Ctb) {
            const double* src = Ctb + (size_t)t * NPc;
            for (int k = 0; k < nC; ++k) {                       // (uniform)
                unsigned vo = (unsigned)lane * 16u + ((unsigned)k << 10);
                vo = vo < rowB - 16u ? vo : rowB - 16u;
                t1_dma(src, vo, dst + ((unsigned)k << 10));
            }
        }
        t1_dma16(bcol + (size_t)t * R, (unsigned)(lane & 15) * 16u, dst + 8u * (unsigned)stgB);
    };
    auto load_scal = [&](int t) {
        t = t < T ? t : T - 1;
        cs_n = scol[t + zero_v]; cn_n = nobs[t + zero_v]; cl_n = ldrow[t + zero_v];
    };
    double ssum = 0.0, nsum = 0.0, ldsum = 0.0;
    LogProd detprod;
    long long p_inv = 0, p_p1 = 0, p_p2 = 0, p_p3 = 0, q_w = 0, q_g = 0, q_u = 0, q_r = 0;
    (void)p_inv; (void)p_p1; (void)p_p2; (void)p_p3; (void)q_w; (void)q_g; (void)q_u; (void)q_r;
    const long long t_start = RT_NOW(); (void)t_start;
    issue_fwd(tb);
    load_scal(tb);
#pragma unroll 1
    for (int t = tb; t < te; ++t) {
        const bool last = (t == T);
        const bool own = t >= s0 && t < e0;
        if (!first && t == s0) put_state(0, Omf, xi);
        if (!lastc && t == e0) put_state(1, Omf, xi);
        const double cs = cs_n, cl = cl_n;                       // (loaded a period ago)
        const int cn = cn_n;
        load_scal(t + 1);
        M32 Z;
        {
            M32 Phi;
            t1_ld_s(sm + kT1Phi, lane, Phi);
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int v = 0; v < 4; ++v) Z.t[I][J][v] = last ? Omf.t[I][J][v] : Omf.t[I][J][v] + Phi.t[I][J][v];
        }
        RT_TICK(p_inv);
        const double dM = t1_sweep_inverse(praw, Z, npiv, q, c);
        RT_TOCK(p_inv);
        RT_TICK(p_p1);
        M32 Jaug;
        {
            M32 X = Kt;
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int v = 0; v < 4; ++v) X.t[I][1][v] = c31 ? xi[I][v] : Kt.t[I][1][v];
            t1_mm0(Z, X, nks, Jaug);                             // [J | w] = Z'[K' | xi]
        }
        if (!last && t >= s0) put_packed(tab_ent(t), Z);         // (uniform; warm-up periods leave no entry)
        if (last) RT_TOCK(p_p1);
        if (last) {                                              // the terminal step: P_T = Z, f_T = column 31 -- through the table's spare
            double* ent = t1_here(ZJ + (size_t)T * 2 * RR);      // entry T (the backward sweep starts there; nothing live across the loop)
            put_packed(ent, Z);
            if (c31) {
#pragma unroll
                for (int I = 0; I < 2; ++I)
#pragma unroll
                    for (int v = 0; v < 4; ++v) { ent[RR + 16 * I + q + 4 * v] = Jaug.t[I][1][v]; qacc = fma(-xi[I][v], Jaug.t[I][1][v], qacc); }
            }
            detprod.mul(dM);                                     // det Om_f,T
        } else {
            if (own) detprod.mul(dM);
            M32 prod;
            t1_mm0(Kt, Jaug, nks, prod);                         // K [J | w]
            RT_TOCK(p_p1);
            RT_TICK(p_p2);
            t1_wait_vm();                                        // the stage holds period t (issued a period ago)
            const bool full = (cn == N);
            const double* stg = sm + kT1Stg;
            M32 QiL;
            t1_ld_s(sm + kT1QiC, lane, QiL);
            if (full || Ctb == nullptr) {                        // (uniform, rare) no missing cell: C_t = Cfull, and no row was written
                const double* Cfg = t1_here(a.Cfull + (size_t)b * RR);   // (from memory: a stall per such period, no LDS for the rare case)
#pragma unroll
                for (int I = 0; I < 2; ++I)
#pragma unroll
                    for (int J = 0; J < 2; ++J)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const bool in = (cinm >> (8 * I + 4 * J + v)) & 1u;
                            const double omp = (J == 1 && c31) ? QiL.t[I][J][v] : QiL.t[I][J][v] - prod.t[I][J][v];
                            Omf.t[I][J][v] = omp + (in ? Cfg[(16 * I + q + 4 * v) * R + 16 * J + c] : 0.0);
                        }
            } else {
#pragma unroll
                for (int I = 0; I < 2; ++I)
#pragma unroll
                    for (int J = 0; J < 2; ++J)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const double cv = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(stg) + pkB[I][J][v]);
                            const double omp = (J == 1 && c31) ? QiL.t[I][J][v] : QiL.t[I][J][v] - prod.t[I][J][v];
                            Omf.t[I][J][v] = omp + cv;
                        }
            }
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double bv = stg[stgB + 16 * I + q + 4 * v];
                    if (own) qacc = fma(-xi[I][v], Jaug.t[I][1][v], qacc);
                    xi[I][v] = prod.t[I][1][v] + bv;
                }
            if (own) {
                ssum += cs;
                nsum += (double)cn;
                ldsum += full ? a.ldfull[b] : cl;
            }
            t1_lds_fence();                                      // (the stage is read)
            RT_TOCK(p_p2);
            RT_TICK(p_p3);
            if (t >= s0) {                                       // (uniform; warm-up periods leave no entry)
                if (c31) {
                    double* we = w_ent(t);
#pragma unroll
                    for (int I = 0; I < 2; ++I)
#pragma unroll
                        for (int v = 0; v < 4; ++v) we[16 * I + q + 4 * v] = Jaug.t[I][1][v];
                }
            }
            issue_fwd(t + 1);                                    // re-arm the stage
            RT_TOCK(p_p3);
        }
    }

    const long long t_fwd = RT_NOW(); (void)t_fwd;
    // ---------------- this chunk's part of the log-likelihood ---------------------------------------------------------------
    {
        double qd = c31 ? qacc : 0.0;
        qd += __shfl_xor(qd, 16, 64);
        qd += __shfl_xor(qd, 32, 64);                            // lanes c = 15: the sum over q
        if (lane == 15) {
            double LD = detprod.log_value();                     // (the last chunk's includes det Om_f,T)
            if (first) LD += ldet0;
            part[0] = nsum * kLog2PiT + ldsum + LD + ssum + qd;
        }
    }

    // ---------------- backward sweep ---------------------------------------------------------------------------------------
    // Z (packed) and w_t of step t arrive in the stage by LDS-DMA, issued a step ahead right behind the gather of the step before; the
    // smoothed moments a step leaves are stored at the start of the NEXT step, behind its wait: every wait finds only traffic a step old
    const int npr = r * (r + 1) / 2;
    int poff[2][2][4];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = 16 * I + q + 4 * v, j = 16 * J + c;
                poff[I][J][v] = (i < r && j <= i) ? i * (i + 1) / 2 + j : -1;
            }
    double* const fsm = a.f_smooth + (size_t)b * T * r;
    double* const Psm = a.P_smooth ? a.P_smooth + (size_t)b * T * npr : nullptr;
    auto emit = [&](int trow, const M32& P, const double (&f)[2][4]) {
        if (c31) {
            double* fr = fsm + (size_t)trow * r;
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (16 * I + q + 4 * v < r) fr[16 * I + q + 4 * v] = f[I][v];
        }
        if (Psm) {
            double* pr = Psm + (size_t)trow * npr;
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J <= I; ++J)                     // (tile (0, 1) lies above the diagonal)
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (poff[I][J][v] >= 0) pr[poff[I][J][v]] = P.t[I][J][v];
        }
    };
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // (the wave reads back its own table and w_t stores: one CU, one write-through L1)
    t1_lds_fence();
    auto issue_bwd = [&](int t) {
        t = t > s0 ? t : s0;
        dma_packed(tab_ent(t), w_ent(t), lds0 + 8u * (unsigned)kT1Stg);
    };
    const int tl = lastc ? T - 1 : e0 + Wk - 1;                // backward steps tl .. s0
    M32 Ps;
    double fs[2][4];
    {   // the start: (P_T, f_T) from entry T, or the guess the backward warm-up forgets: (Z, w) of the last extra period
        const double* ent = lastc ? ZJ + (size_t)T * 2 * RR : xZJ + (size_t)(Wk - 1) * 2 * RR;
        const double* fv = lastc ? ent + RR : xw + (size_t)(Wk - 1) * R;
        dma_packed(ent, fv, lds0 + 8u * (unsigned)kT1Stg);
        t1_wait_vm();
        const double* stg = sm + kT1Stg;
        get_packed(stg, Ps);
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int v = 0; v < 4; ++v) fs[I][v] = stg[stgB + 16 * I + q + 4 * v];
        t1_lds_fence();
        issue_bwd(tl);
    }
    M32 SP, SU;
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) { SP.t[I][J][v] = (EM && lastc) ? Ps.t[I][J][v] : 0.0; SU.t[I][J][v] = 0.0; }
    if (EM && lastc) t1_st_g(sums + 2 * RR, lane, Ps);         // P_T
    int pend = lastc ? T - 1 : -1;                             // row the moments in (Ps, fs) belong to and still have to be stored; -1: none
#pragma unroll 1
    for (int t = tl; t >= s0; --t) {
        const bool own = t < e0;
        RT_TICK(q_w);
        t1_wait_vm();                                            // the stage holds step t
        const double* stg = sm + kT1Stg;
        M32 zc, jt, U, Pn;
        double wv[2][4];
        get_packed(stg, zc);
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int v = 0; v < 4; ++v) wv[I][v] = stg[stgB + 16 * I + q + 4 * v];
        t1_lds_fence();                                          // (the stage is read: re-arm it)
        issue_bwd(t - 1);
        if (pend >= 0) emit(pend, Ps, fs);
        RT_TOCK(q_w);
        RT_TICK(q_g);
        t1_mm0(Kt, zc, nks, jt);                                 // J' = K Z (the table holds Z only)
        RT_TOCK(q_g);
        RT_TICK(q_u);
        t1_mm0(Ps, jt, nks, U);                                  // U = P_s J' = Cov(f_t+1, f_t | X)
        {
            M32 Uaug = U;
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int v = 0; v < 4; ++v) Uaug.t[I][1][v] = c31 ? fs[I][v] : U.t[I][1][v];
            t1_mm(jt, Uaug, nks, zc, Pn);                        // Z + J [U | f_t+1]   ((J')' = J: the same registers as the A operand)
        }
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (c31) {
                    fs[I][v] = wv[I][v] + (Pn.t[I][1][v] - zc.t[I][1][v]);
                    Pn.t[I][1][v] = zc.t[I][1][v];
                }
            }
        Ps = Pn;
        RT_TOCK(q_u);
        if (EM && own) {
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int v = 0; v < 4; ++v) { SU.t[I][J][v] += U.t[I][J][v]; if (t > 0) SP.t[I][J][v] += Pn.t[I][J][v]; }
        }
        pend = (own && t > 0) ? t - 1 : -1;
        if (!lastc && t == e0) put_state(2, Ps, fs);
    }
    if (pend >= 0) emit(pend, Ps, fs);
    if (!first) put_state(3, Ps, fs);
#ifdef DFM_TILE_PROF
    if (b == 5 && ck == 1 && lane == 0)
        printf("TILE1PROF steps %d (10 ns ticks): total %lld  forward %lld (inverse %lld, 2 products %lld, wait + update %lld, pack + store + issue %lld)  backward %lld (wait + issue + emit %lld, gather + J' %lld, U + P %lld)\n",
               te - tb, RT_NOW() - t_start, t_fwd - t_start, p_inv, p_p1, p_p2, p_p3, RT_NOW() - t_fwd, q_w, q_g, q_u);
#endif
    if (!EM) return;
    t1_st_g(sums, lane, SP);
    t1_st_g(sums + RR, lane, SU);
    if (first) {
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J < 2; ++J)
#pragma unroll
                for (int v = 0; v < 4; ++v) a.P0s[(size_t)b * RR + (16 * I + q + 4 * v) * R + 16 * J + c] = Ps.t[I][J][v];
        if (c31) {
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int v = 0; v < 4; ++v) a.f0s[(size_t)b * R + 16 * I + q + 4 * v] = fs[I][v];
        }
    }
}

// The meeting point of a replicate's chunks: boundary checks (chunk_fail), the log-likelihood and the EM bookkeeping, the EM sums.
__global__ __launch_bounds__(256) void tile_chunk_finish_kernel(RecursionArgs a, double tol) {
    constexpr int R = kRt, RR = R * R;
    __shared__ double red[4][4];
    __shared__ double f0sh[R];
    __shared__ int failS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int I = w >> 1, J = w & 1, q = lane >> 4, c = lane & 15;
    const int b = blockIdx.x, NC = a.tile_nc, Wk = a.tile_w, T = a.T, r = a.r;
    const size_t sd = tk_slot_doubles(Wk);
    const double* base = a.tile_scr + (size_t)b * NC * sd;
    auto bst = [&](int ck, int k) { return base + (size_t)ck * sd + (size_t)Wk * (2 * RR + R) + (size_t)k * kTkBst; };
    if (tid == 0) failS = 0;
    __syncthreads();
    for (int ck = 1; ck < NC; ++ck) {
#pragma unroll 1
        for (int dir = 0; dir < 2; ++dir) {
            // forward: the entry state of chunk ck against the exit state of chunk ck - 1; backward: the entry state of chunk ck - 1
            // against the exit state of chunk ck
            const double* got = dir == 0 ? bst(ck, 0) : bst(ck - 1, 2);
            const double* ref = dir == 0 ? bst(ck - 1, 1) : bst(ck, 3);
            double dm = 0.0, rm = 0.0, dv = 0.0, rv = 0.0;
            bool bad = false;
            for (int i = tid; i < kTkBst; i += 256) {
                const double g = got[i], e = ref[i], d = fabs(g - e);
                bad = bad || !(d == d);
                if (i < RR) { dm = fmax(dm, d); rm = fmax(rm, fabs(e)); } else { dv = fmax(dv, d); rv = fmax(rv, fabs(e)); }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                dm = fmax(dm, __shfl_xor(dm, o, 64)); rm = fmax(rm, __shfl_xor(rm, o, 64));
                dv = fmax(dv, __shfl_xor(dv, o, 64)); rv = fmax(rv, __shfl_xor(rv, o, 64));
            }
            if (bad) atomicOr(&failS, 1);
            __syncthreads();                                     // (the last round's reads of red are done)
            if (lane == 0) { red[w][0] = dm; red[w][1] = rm; red[w][2] = dv; red[w][3] = rv; }
            __syncthreads();
            if (tid == 0) {
                double m[4];
                for (int k = 0; k < 4; ++k) m[k] = fmax(fmax(red[0][k], red[1][k]), fmax(red[2][k], red[3][k]));
                // (the mean vector against the larger of its own scale and 1: a state near zero is not a disagreement)
                if (!(m[0] <= tol * m[1]) || !(m[2] <= tol * fmax(m[3], 1.0))) failS = 1;
            }
        }
    }
    __syncthreads();
    const bool fail = failS != 0;
    if (tid == 0) a.chunk_fail[b] = fail ? 1 : 0;
    if (fail) return;                                          // (uniform) the sequential kernel runs this replicate again
    if (tid == 0) {
        double tot = 0.0;
        for (int ck = 0; ck < NC; ++ck) tot += (bst(ck, 0) + 4 * kTkBst)[0];
        const double ll = -0.5 * tot;
        a.loglik[b] = ll;
        if (a.ncov) a.ncov[b] = T;
        if (a.active) {                                          // EM bookkeeping, as recursion_kernel
            const bool was = a.k == 0 ? true : (a.active[b] != 0);
            bool go = was;
            if (was && a.k >= 1 && a.tol > 0.0) {
                const double llp = a.ll_path[(size_t)b * a.max_iter + a.k - 1];
                go = !((ll - llp) / (0.5 * (fabs(ll) + fabs(llp))) < a.tol);
            }
            if (was) { a.ll_path[(size_t)b * a.max_iter + a.k] = ll; a.iters[b] = a.k + 1; }
            a.active[b] = go ? 1 : 0;
        }
    }
    if (a.S11 == nullptr) return;
    // EM sums: the chunks' parts of sum P_t and sum U_t, plus the three products over f_smooth
    if (tid < R) f0sh[tid] = a.f0s[(size_t)b * R + tid];
    __syncthreads();
    v4d SP, SU, G11, G10, G00;
#pragma unroll
    for (int v = 0; v < 4; ++v) { SP[v] = 0.0; SU[v] = 0.0; G11[v] = 0.0; G10[v] = 0.0; G00[v] = 0.0; }
    for (int ck = 0; ck < NC; ++ck) {
        const double* sums = bst(ck, 0) + 4 * kTkBst + kTkPart;
        const v4d p = ld_tile_g(sums, w, lane), u = ld_tile_g(sums + RR, w, lane);
#pragma unroll
        for (int v = 0; v < 4; ++v) { SP[v] += p[v]; SU[v] += u[v]; }
    }
    const v4d PT = ld_tile_g(bst(NC - 1, 0) + 4 * kTkBst + kTkPart + 2 * RR, w, lane);
    tile_gsums_ahead(a.f_smooth + (size_t)b * T * r, f0sh, T, r, I, J, q, c, G11, G10, G00);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const size_t o = (size_t)b * RR + (16 * I + q + 4 * v) * R + 16 * J + c;
        a.S11[o] = SP[v] + G11[v];
        a.S10[o] = SU[v] + G10[v];
        a.S00[o] = (SP[v] - PT[v] + a.P0s[o]) + G00[v];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The transition M-step on the sums recursion_tile_kernel leaves in the workspace -- the epilogue of recursion_wave_kernel<32>
// (element per thread, Grid<32>):  A = S10 S00^-1,  Q = sym(S11 - A S10') / T,  mu0 = f_0|T,  P0 = sym(P_0|T),  S11^-1.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024, 1) void tile_mstep_kernel(RecursionArgs a) {
    constexpr int R = kRt, RR = R * R;
    constexpr int TS = kTileStride<R>, RT = R * TS;
    __shared__ __attribute__((aligned(16))) double wsm[2 * RT + kGridProw<R> + 2 * (RR / 64) * R + 2 * RT];
    double* L0 = wsm;
    double* L1 = L0 + RT;
    Grid<R> G;
    G.prow = L1 + RT;
    G.red = G.prow + kGridProw<R>;
    G.tt = G.red + 2 * (RR / 64) * R;
    const int lane = threadIdx.x, i = lane / R, j = lane % R;
    G.l = lane; G.i = i; G.j = j;
    const int b = blockIdx.x;
    const size_t o = (size_t)b * RR + lane;
    const double S11 = a.S11[o], S10 = a.S10[o], S00 = a.S00[o], Ps = a.P0s[o];
    const bool em_apply = a.active ? (a.active[b] != 0) : true;
    double inv = S00;
    (void)G.sweep_inverse(inv);
    G.sync();
    L0[TS * i + j] = S10;
    L1[TS * i + j] = inv;                                      // symmetric: rows = columns
    G.sync();
    const double An = dot_rows<R>(L0, L1, i, j);
    G.sync();
    L1[TS * i + j] = An;
    G.sync();
    double Qn = (S11 - dot_rows<R>(L1, L0, i, j)) / (double)a.T;   // (A S10')_ij = row i of A . row j of S10
    Qn = 0.5 * (Qn + G.transposed(Qn));
    const double P0n = 0.5 * (Ps + G.transposed(Ps));
    double inv2 = S11;
    (void)G.sweep_inverse(inv2);
    a.S11inv[o] = inv2;
    if (em_apply) {
        a.A_out[o] = An;
        a.Q_out[o] = Qn;
        a.P0_out[o] = P0n;
        if (j == 0) a.mu0_out[(size_t)b * R + i] = a.f0s[(size_t)b * R + i];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// cov_tile_kernel: the data-independent covariance half of the BALANCED fast path at Rp = 32 (dfm_cov8.h cov_grid: the forward
// steps to the fixed point, terminal, backward steps, carry powers -- same tables, same bookkeeping) in the tile layout: four
// waves per replicate instead of the 1024 threads of cov_grid_kernel<32>.  That kernel (128 VGPRs x 16 waves) fills a CU, so
// the streaming collapse of the pass cannot start beside it: its 0.19 ms at BASELINE config 4 are serial time in front of the
// 0.89 ms of the collapse, ~50 dependent 32 x 32 operations each behind 16-wave barriers.  Here a product is <= 8 MFMAs per
// wave behind a 4-wave barrier and an inverse executes ceil(r / 4) block pivots.
//   forward   Z = (Om_f + Phi)^-1,  J = Z K',  G = K Z (= J'),  Om_f <- Qi - K J + C       (K' = A'Qi, Phi = K'A)
//   backward  U = P_s J' = P'G,  P_s <- Z + J U = Z + G'U
//   powers    (G^2, J^2) <- (J'G, G'J): the pair stays each other's transpose, every product has the form Y'X
// Matrices leave in the row-major [32][32] layout of the tables (4 rows x 128 bytes per store instruction).
// ------------------------------------------------------------------------------------------------------------------
namespace {
// AND over the workgroup's 256 threads (one barrier; flag slots alternate so that the next call cannot overtake this one's reads)
__device__ __forceinline__ bool rt_all(RtCtx& x, bool pred, int& par) {
    const bool wv = __all(pred ? 1 : 0) != 0;
    double* fl = x.sm + kRtRed + 8 * (par & 1);
    par ^= 1;
    if (x.lane == 0) fl[x.w] = wv ? 1.0 : 0.0;
    rt_barrier();
    return fl[0] != 0.0 && fl[1] != 0.0 && fl[2] != 0.0 && fl[3] != 0.0;
}
// element (i, j) of a matrix st_tile() left in an LDS tile buffer
__device__ __forceinline__ double tl_elem(const double* buf, int i, int j) {
    const int ri = i & 15, v = ri >> 2, qq = ri & 3;
    return buf[(2 * (i >> 4) + (j >> 4)) * kRtTile + (v >> 1) * 128 + (qq * 16 + (j & 15)) * 2 + (v & 1)];
}
}  // namespace

__global__ __launch_bounds__(256, 1) void cov_tile_kernel(FastArgs a, int rstate) {
    constexpr int R = kRt, RR = R * R, NLEV = scan_levels(R), KEEP = 4;
    __shared__ __attribute__((aligned(16))) double sm[kRtLds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    RtCtx x;
    x.sm = sm; x.lane = lane; x.w = w; x.I = w >> 1; x.J = w & 1; x.q = lane >> 4; x.c = lane & 15; x.pp = 0;
    const int I = x.I, J = x.J, q = x.q, c = x.c;
    const int b = blockIdx.x;
    const int T = a.T, r = a.r;
    const int npiv = (rstate + 3) >> 2, nks = npiv;
    const int col = 16 * J + c;
    int rowv[4], rm[4];                                        // rows of the lane's elements; their row-major offsets
#pragma unroll
    for (int v = 0; v < 4; ++v) { rowv[v] = 16 * I + q + 4 * v; rm[v] = rowv[v] * R + col; }
    double* bufA = sm + kRtBufA;
    double* bufB = sm + kRtBufB;
    double* Xk = sm + kRtXk;
    const int tY0 = I, tY1 = 2 + I, tX0 = J, tX1 = 2 + J;
    auto put_rm = [&](double* dst, const v4d& m) {             // TL -> row-major [32][32]
#pragma unroll
        for (int v = 0; v < 4; ++v) dst[rm[v]] = m[v];
    };
    int apar = 0;

    // ---------------- prologue: Qi = Q^-1, Om_f,0 = P0^-1, K' = A'Qi, Phi = K'A, xi_0 = P0^-1 mu0 ------------------------------
    v4d Qi, Omf, Ael, Cel, zero4;
    {
        const double* Cf = a.Cfull + (size_t)b * RR;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const size_t o = (size_t)b * RR + rm[v];
            Qi[v] = a.Q[o]; Omf[v] = a.P0[o]; Ael[v] = a.A[o];
            Cel[v] = 0.5 * (Cf[rm[v]] + Cf[col * R + rowv[v]]);  // exactly symmetric (the matrix-pipe Gram is symmetric to rounding)
            zero4[v] = 0.0;
        }
    }
    const double mu0t = tid < R ? a.mu0[(size_t)b * R + tid] : 0.0;
    const double detQ = rt_sweep_inverse(x, Qi, npiv);
    const double detP0 = rt_sweep_inverse(x, Omf, npiv);
    st_tile(bufA, w, lane, Ael);
    st_tile(bufB, w, lane, Qi);
    if (tid < R) sm[kRtRed + 16 + tid] = mu0t;
    rt_barrier();
    v4d KtY0, KtY1, Phi;
    {
        const v4d A0i = ld_tile(bufA, tY0, lane), A1i = ld_tile(bufA, tY1, lane);
        const v4d A0j = ld_tile(bufA, tX0, lane), A1j = ld_tile(bufA, tX1, lane);
        const v4d Q0i = ld_tile(bufB, tY0, lane), Q1i = ld_tile(bufB, tY1, lane);
        const v4d Q0j = ld_tile(bufB, tX0, lane), Q1j = ld_tile(bufB, tX1, lane);
        const v4d Kt = mm_tn(A0i, A1i, Q0j, Q1j, nks, zero4);                       // K' = A'Qi
        const v4d Km = mm_tn(Q0i, Q1i, A0j, A1j, nks, zero4);                       // K  = Qi A
        st_tile(Xk, w, lane, Kt);
        rt_barrier();
        st_tile(bufB, w, lane, Km);
        rt_barrier();
        const v4d K0i = ld_tile(bufB, tY0, lane), K1i = ld_tile(bufB, tY1, lane);
        Phi = mm_tn(K0i, K1i, A0j, A1j, nks, zero4);                                // Phi = K'A
        KtY0 = ld_tile(Xk, tY0, lane); KtY1 = ld_tile(Xk, tY1, lane);               // K' as Y: constant
        st_tile(bufA, w, lane, Omf);
        rt_barrier();
    }
    double q0 = 0.0;
    if (w == 0) {                                              // xi_0 = Om_f,0 mu0 and mu0'xi_0: one row per lane (lanes 32.. idle)
        double xr = 0.0;
        if (lane < R) {
            for (int j = 0; j < R; ++j) xr = fma(tl_elem(bufA, lane, j), sm[kRtRed + 16 + j], xr);
            a.xi0[(size_t)b * R + lane] = xr;
        }
        q0 = wave_allsum(lane < R ? xr * mu0t : 0.0);
    }

    // ---------------- forward covariance steps until the fixed point ---------------------------------------------------------
    LogProd detprod;
    double detM_last = 1.0;
    int E = 0;
    v4d Zk[KEEP], Gk[KEEP];                                    // own tiles of Z_e and of J_e' = G_e of the first steps (backward sweep)
#pragma unroll
    for (int u = 0; u < KEEP; ++u) { Zk[u] = zero4; Gk[u] = zero4; }
    v4d Zlast = zero4, Jlast = zero4, Glast = zero4;
    double* tabb = a.tab + (size_t)b * T * 3 * RR;
    for (int e = 0;; ++e) {
        v4d Z;
#pragma unroll
        for (int v = 0; v < 4; ++v) Z[v] = Omf[v] + Phi[v];
        const double detM = rt_sweep_inverse(x, Z, npiv);       // (its barriers: every wave is done with bufA / bufB of the last step)
        st_tile(bufA, w, lane, Z);
        rt_barrier();
        const v4d ZY0 = ld_tile(bufA, tY0, lane), ZY1 = ld_tile(bufA, tY1, lane);
        const v4d ZX0 = ld_tile(bufA, tX0, lane), ZX1 = ld_tile(bufA, tX1, lane);
        const v4d X0 = ld_tile(Xk, tX0, lane), X1 = ld_tile(Xk, tX1, lane);
        const v4d Jm = mm_tn(ZY0, ZY1, X0, X1, nks, zero4);    // J = Z K'
        const v4d Gm = mm_tn(KtY0, KtY1, ZX0, ZX1, nks, zero4);   // G = K Z
        st_tile(bufB, w, lane, Jm);
        rt_barrier();
        const v4d JX0 = ld_tile(bufB, tX0, lane), JX1 = ld_tile(bufB, tX1, lane);
        const v4d KJ = mm_tn(KtY0, KtY1, JX0, JX1, nks, zero4);
        v4d Omn;
        bool same = true;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            Omn[v] = (Qi[v] - KJ[v]) + Cel[v];
            same = same && close_enough(Omn[v], Omf[v]);
        }
        const bool gsame = rt_all(x, same, apar);
        double* te = tabb + (size_t)e * 3 * RR;
        put_rm(te, Z); put_rm(te + RR, Jm); put_rm(te + 2 * RR, Gm);
#pragma unroll
        for (int u = 0; u < KEEP; ++u)
            if (u == e) { Zk[u] = Z; Gk[u] = Gm; }              // (uniform)
        Zlast = Z; Jlast = Jm; Glast = Gm;
        E = e + 1;
        detprod.mul(detM);
        detM_last = detM;
        Omf = Omn;
        if (gsame || e + 1 >= T) break;
    }
    const int ts = E - 1;

    // ---------------- terminal ---------------------------------------------------------------------------------------------
    v4d Ps = Omf;
    const double detOmT = rt_sweep_inverse(x, Ps, npiv);
    put_rm(a.PT + (size_t)b * RR, Ps);
    if (tid == 0) {
        const double sum_ldz = -(detprod.log_value() + (double)(T - E) * log(detM_last));
        const double LD = log(detOmT) + log(detP0) + (double)T * log(detQ) - sum_ldz;
        a.llc[b] = (double)a.N * (double)T * kLog2PiF + (double)T * a.ldfull[b] + LD + q0;
        a.E[b] = E;
    }

    // ---------------- backward covariance steps ----------------------------------------------------------------------------
    const int npr = r * (r + 1) / 2;
    int poff[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) poff[v] = (a.P_smooth && rowv[v] < r && col <= rowv[v]) ? rowv[v] * (rowv[v] + 1) / 2 + col : -1;
    double* Psm = a.P_smooth ? a.P_smooth + (size_t)b * T * npr : nullptr;
    auto emit = [&](int trow, const v4d& P) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (poff[v] >= 0) Psm[(size_t)trow * npr + poff[v]] = P[v];
    };
    emit(T - 1, Ps);
    v4d SP = Ps, SU = zero4;
    int fill_lo = 0, fill_hi = 0;
    int t = T - 1, cur_e = -2;
    v4d Zc = zero4, GX0 = zero4, GX1 = zero4, GY0 = zero4, GY1 = zero4;
    while (t >= 0) {
        const int e = t < ts ? t : ts;
        const bool reload = e != cur_e;                          // (uniform)
        if (reload) {
            v4d Gc = zero4;
            if (e == ts) {
                Zc = Zlast; Gc = Glast;
            } else if (e < KEEP) {
#pragma unroll
                for (int u = 0; u < KEEP; ++u)
                    if (u == e) { Zc = Zk[u]; Gc = Gk[u]; }
            } else {
                const double* te = tabb + (size_t)e * 3 * RR;
#pragma unroll
                for (int v = 0; v < 4; ++v) { Zc[v] = te[rm[v]]; Gc[v] = te[2 * RR + rm[v]]; }
            }
            cur_e = e;
            st_tile(Xk, w, lane, Gc);                            // (K' is done with)
        }
        st_tile(bufA, w, lane, Ps);
        rt_barrier();
        if (reload) {
            GX0 = ld_tile(Xk, tX0, lane); GX1 = ld_tile(Xk, tX1, lane);
            GY0 = ld_tile(Xk, tY0, lane); GY1 = ld_tile(Xk, tY1, lane);
        }
        const v4d PY0 = ld_tile(bufA, tY0, lane), PY1 = ld_tile(bufA, tY1, lane);
        const v4d U = mm_tn(PY0, PY1, GX0, GX1, nks, zero4);   // U = P_s J'
        st_tile(bufB, w, lane, U);
        rt_barrier();
        const v4d UX0 = ld_tile(bufB, tX0, lane), UX1 = ld_tile(bufB, tX1, lane);
        const v4d Psn = mm_tn(GY0, GY1, UX0, UX1, nks, Zc);    // Z + J U
        bool same = true;
#pragma unroll
        for (int v = 0; v < 4; ++v) same = same && close_enough(Psn[v], Ps[v]);
        const bool gsame = rt_all(x, same, apar);
        const bool skip = (e == ts && t > ts && gsame);          // steps t-1 .. ts repeat this (U, P_s)
        const int plo = ts >= 1 ? ts : 1;                        // periods plo .. t-1 carry P_s,inf
        const double cu = skip ? (double)(t - ts + 1) : 1.0;
        const double cp = (t >= 1 ? 1.0 : 0.0) + (skip ? (double)(t - plo) : 0.0);
#pragma unroll
        for (int v = 0; v < 4; ++v) { SU[v] = fma(cu, U[v], SU[v]); SP[v] = fma(cp, Psn[v], SP[v]); }
        if (t >= 1) emit(t - 1, Psn);
        if ((t == 0 || (skip && ts == 0)) && a.SP11) put_rm(a.P0s + (size_t)b * RR, Psn);
        if (skip) {
            put_rm(a.PsInf + (size_t)b * RR, Psn);
            fill_lo = plo - 1;
            fill_hi = t - 1;
            t = ts - 1;
        } else {
            t -= 1;
        }
        Ps = Psn;
    }
    if (tid == 0) { a.fill[2 * b] = fill_lo; a.fill[2 * b + 1] = fill_hi; }
    if (a.SP11) { put_rm(a.SP11 + (size_t)b * RR, SP); put_rm(a.SU + (size_t)b * RR, SU); }

    // ---------------- steady Z, J, G and the powers G^(L 2^k), J^(L 2^k) for the chunk carries ----------------------------------
    {
        double* st = a.stead + (size_t)b * stead_mats(R) * RR;
        put_rm(st, Zlast); put_rm(st + RR, Jlast); put_rm(st + 2 * RR, Glast);
        v4d MG = Glast, MJ = Jlast;
        auto square2 = [&]() {
            st_tile(bufA, w, lane, MG);
            st_tile(bufB, w, lane, MJ);
            rt_barrier();
            const v4d gY0 = ld_tile(bufA, tY0, lane), gY1 = ld_tile(bufA, tY1, lane), gX0 = ld_tile(bufA, tX0, lane), gX1 = ld_tile(bufA, tX1, lane);
            const v4d jY0 = ld_tile(bufB, tY0, lane), jY1 = ld_tile(bufB, tY1, lane), jX0 = ld_tile(bufB, tX0, lane), jX1 = ld_tile(bufB, tX1, lane);
            MG = mm_tn(jY0, jY1, gX0, gX1, nks, zero4);        // J'G = G G
            MJ = mm_tn(gY0, gY1, jX0, jX1, nks, zero4);        // G'J = J J
            rt_barrier();                                      // (the buffers are free again)
        };
        for (int l = 1; l < a.L; l <<= 1) square2();             // M^L
#pragma unroll 1
        for (int k = 0; k < NLEV; ++k) {
            put_rm(st + (size_t)(3 + k) * RR, MG);
            put_rm(st + (size_t)(3 + NLEV + k) * RR, MJ);
            if (k + 1 < NLEV) square2();
        }
    }
}

bool cov_tile_supported(int Rpad, const FastArgs& a, int rstate) {
    static const bool off = [] { const char* v = diag_env("DFM_NO_COV_TILE"); return v && atoi(v) != 0; }();
    return !off && Rpad == 32 && a.Lam == nullptr && rstate >= 1 && rstate <= 32 && a.T >= 1;
}
hipError_t launch_cov_tile(const FastArgs& a, int rstate, hipStream_t s) {
    note_kernel("cov_tile_kernel");
    hipLaunchKernelGGL(cov_tile_kernel, dim3(a.B), dim3(256), 0, s, a, rstate);
    return hipGetLastError();
}

// Rp = 32, information form, the plain factor model (loadings as wide as the state), 17 <= state width <= 31 (column 31 must
// be padding).  DFM_NO_TILE=1: recursion_wave_kernel<32> instead (A/B, diagnostics).
bool recursion_tile_supported(int Rpad, const RecursionArgs& a) {
    static const bool off = [] { const char* v = diag_env("DFM_NO_TILE"); return v && atoi(v) != 0; }();
    if (off || Rpad != 32 || a.cov || a.Rc != 0 || a.rl != 0 || a.kdim != 0) return false;
    if (a.rstate < 17 || a.rstate > 31 || a.T < 1) return false;
    if (a.S11 && !a.A_out) return false;                       // (sums without the M-step: not a path the library takes)
    return true;
}

// Chunks per replicate for this launch (1: the sequential kernel), their length and warm-up.  tile_nc: 0 = as many as give every
// CU two workgroups, 1 = never, n = that many (development / tests); chunks are at least 4 warm-ups long and of even length.
int recursion_tile_chunks(const RecursionArgs& a, int* lc_out, int* w_out) {
    int W = a.tile_w > 0 ? a.tile_w : 16;
    W = (W + 1) & ~1;
    if (W > kTkWmax) W = kTkWmax;
    if (w_out) *w_out = W;
    if (lc_out) *lc_out = a.T;
    if (a.tile_scr == nullptr || a.chunk_fail == nullptr || a.tile_nc == 1 || a.B < 1) return 1;
    const int cu = a.num_cu > 0 ? a.num_cu : 256;
    int want = a.tile_nc > 1 ? a.tile_nc : (2 * cu) / a.B;
    if (want > kTkNCmax) want = kTkNCmax;
    for (; want > 1; --want) {
        const int lc = 2 * ((a.T + 2 * want - 1) / (2 * want));
        const int lastlen = a.T - (want - 1) * lc;
        if (lc < 4 * W || lastlen < W + 2) continue;
        if ((size_t)a.B * want * tk_slot_doubles(W) * sizeof(double) > a.tile_scr_bytes) continue;
        if (lc_out) *lc_out = lc;
        return want;
    }
    return 1;
}
// workspace for the chunks of a batch: sized for the automatic choice on a device of up to 512 CUs (two workgroups each) and for forced
// counts on small batches; nothing for batches that fill such a device with one workgroup per replicate (they run sequentially)
size_t recursion_tile_scratch_bytes(int B, int T) {
    (void)T;
    if (B > 1024) return 0;                                    // (recursion_tile1_kernel: one chunk per replicate up to 1024 replicates)
    const size_t slots = (size_t)B * kTkNCmax < 2048 ? (size_t)B * kTkNCmax : 2048;   // (two waves per SIMD of a 256-CU device)
    return slots * tk_slot_doubles(kTkWmax) * sizeof(double);
}

// recursion_tile1_kernel's plan: one wave per (replicate, chunk), as many chunks as put a wave on every SIMD (tile_nc = 0) or the forced
// count; a single chunk is allowed (a batch that fills the SIMDs by itself).  0: not this kernel (no scratch, tile_nc = 1, DFM_NO_TILE1).
static int tile1_chunks(const RecursionArgs& a, int* lc_out, int* w_out) {
    static const bool off = [] { const char* v = diag_env("DFM_NO_TILE1"); return v && atoi(v) != 0; }();
    if (off || a.tile_scr == nullptr || a.chunk_fail == nullptr || a.tile_nc == 1 || a.B < 1) return 0;
    int W = a.tile_w > 0 ? a.tile_w : 16;
    W = (W + 1) & ~1;
    if (W > kTkWmax) W = kTkWmax;
    *w_out = W;
    const int cu = a.num_cu > 0 ? a.num_cu : 256;
    // waves a CU holds: by LDS (160 KB), two per SIMD for the pass (256 registers), one with the EM sums
    int wcu = (int)((size_t)160 * 1024 / t1_lds_bytes(a.ct_r, a.rstate));
    const int wmax = a.S11 ? 4 : 4 * DFM_T1_OCC;
    wcu = wcu > wmax ? wmax : (wcu < 1 ? 1 : wcu);
    int want = a.tile_nc > 1 ? a.tile_nc : (wcu * cu + a.B - 1) / a.B;
    if (a.tile_nc <= 0 && want > kTkNCmax) return 0;         // (a batch whose chunks cannot put a wave on every SIMD: four waves per chunk)
    if (want > kTkNCmax) want = kTkNCmax;
    for (; want >= 1; --want) {
        const int lc = want == 1 ? a.T : 2 * ((a.T + 2 * want - 1) / (2 * want));
        const int lastlen = a.T - (want - 1) * lc;
        if (want > 1 && (lc < 4 * W || lastlen < W + 2)) continue;
        if ((size_t)a.B * want * tk_slot_doubles(W) * sizeof(double) > a.tile_scr_bytes) continue;
        *lc_out = lc;
        return want;
    }
    return 0;
}

// true: the launch goes through tile_chunk_finish_kernel, which writes chunk_fail[b] for every replicate (dfm_chunk_fallbacks counts them)
bool recursion_tile_writes_fail(const RecursionArgs& a) {
    int lc = 0, W = 0;
    return tile1_chunks(a, &lc, &W) > 0 || recursion_tile_chunks(a, nullptr, nullptr) > 1;
}

hipError_t launch_recursion_tile(const RecursionArgs& a, hipStream_t s) {
    note_kernel("recursion_tile_kernel");
    int lc = a.T, W = 16;
    hipError_t e;
    if (const int n1 = tile1_chunks(a, &lc, &W)) {
        note_kernel("recursion_tile1_kernel");
        RecursionArgs c = a;
        c.tile_nc = n1; c.tile_lc = lc; c.tile_w = W;
        const unsigned lds = (unsigned)t1_lds_bytes(a.ct_r, a.rstate);
        if (a.S11) hipLaunchKernelGGL(recursion_tile1_kernel<true>, dim3((unsigned)(a.B * n1)), dim3(64), lds, s, c);
        else hipLaunchKernelGGL(recursion_tile1_kernel<false>, dim3((unsigned)(a.B * n1)), dim3(64), lds, s, c);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        double ctol = a.chunk_tol > 0.0 ? a.chunk_tol : 1e-10;
        if (a.S11 != nullptr && a.tol > 0.0 && 1e-2 * a.tol < ctol) ctol = 1e-2 * a.tol;
        hipLaunchKernelGGL(tile_chunk_finish_kernel, dim3(a.B), dim3(256), 0, s, c, ctol);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (n1 > 1) {                                           // replicates with a boundary off (normally none: the blocks exit)
            RecursionArgs f = a;
            f.only_if = a.chunk_fail;
            hipLaunchKernelGGL(recursion_tile_kernel<false>, dim3(a.B), dim3(256), 0, s, f);
        }
        e = hipGetLastError();
        if (e != hipSuccess || !a.S11) return e;
        hipLaunchKernelGGL(tile_mstep_kernel, dim3(a.B), dim3(1024), 0, s, a);
        return hipGetLastError();
    }
    const int nc = recursion_tile_chunks(a, &lc, &W);
    if (nc > 1) {
        RecursionArgs c = a;
        c.tile_nc = nc; c.tile_lc = lc; c.tile_w = W;
        hipLaunchKernelGGL(recursion_tile_kernel<true>, dim3((unsigned)(a.B * nc)), dim3(256), 0, s, c);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        double ctol = a.chunk_tol > 0.0 ? a.chunk_tol : 1e-10;
        if (a.S11 != nullptr && a.tol > 0.0 && 1e-2 * a.tol < ctol) ctol = 1e-2 * a.tol;   // (EM stop rule: see launch_recursion_chunk)
        hipLaunchKernelGGL(tile_chunk_finish_kernel, dim3(a.B), dim3(256), 0, s, c, ctol);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        RecursionArgs f = a;                                    // replicates with a boundary off (normally none: the blocks exit)
        f.only_if = a.chunk_fail;
        hipLaunchKernelGGL(recursion_tile_kernel<false>, dim3(a.B), dim3(256), 0, s, f);
    } else {
        RecursionArgs f = a;
        f.only_if = nullptr;
        hipLaunchKernelGGL(recursion_tile_kernel<false>, dim3(a.B), dim3(256), 0, s, f);
    }
    e = hipGetLastError();
    if (e != hipSuccess || !a.S11) return e;
    hipLaunchKernelGGL(tile_mstep_kernel, dim3(a.B), dim3(1024), 0, s, a);
    return hipGetLastError();
}

}  // namespace dfm
