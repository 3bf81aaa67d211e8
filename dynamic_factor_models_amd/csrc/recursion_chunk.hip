// recursion_chunk.hip -- the sequential recursion of a panel WITH missing cells (Rp = 8, information form: oracle/info_form.py), cut in
// TIME: a replicate's T periods are 64 chunks of L = ceil(T / 64) periods, chunk j belongs to LANE j of the replicate's one wave, and
// every lane runs the plain filter / smoother recursion on its own 8 x 8 matrices in its own registers (dfm_chunk_core.h): no
// cross-lane exchange, no LDS tile, no table look-up on the chain -- 36 packed doubles per matrix, products with the replicate's
// constant K = Q^-1 A as FMAs with a scalar operand (rows of K in SGPRs, scalar loads from a table chunk_prep_kernel wrote).
//
// Why it is allowed: the filter forgets.  Om_f,t depends on Om_f,t-W through a product of W contractions (with N series loading
// on r factors, (I + C P)^-1 is O(r / N) per period), so a lane that starts W periods before its chunk from a GUESS (forward:
// Om_p = Q^-1, xi = b; backward: P = 0, f = 0) holds the exact state to rounding when its chunk begins.  That is CHECKED, not assumed:
// the state a lane holds after its warm-up is compared ELEMENT BY ELEMENT (dfm_chunk_core.h state_gap / gap_close: largest difference
// against chunk_tol x the largest entry, matrix and vector separately) with the state its neighbour holds at the end of its own
// chunk -- forward and backward -- and a replicate with one boundary off raises chunk_fail[b]: launch_recursion then runs it on
// the sequential kernel (RecursionArgs::only_if).  The warm-up states wait for the comparison in two extra slots of the smoother
// table (44 doubles per lane, lane-contiguous like the table's rows; the neighbour reads them back after its last step).
// Lane 0 starts from the exact initial state, the top lane from the exact terminal state, so every chunk starts within chunk_tol
// (relative, max-norm) of the state the sequential recursion holds there; on the C2 shape with 10 % missing cells the boundaries
// agree to 1e-13 with W = 8 (tests/test_chunk_core_cpu.py has the lane-level NumPy model of this file against the oracle).
//
// Cost: L + W steps per lane instead of T per replicate -- 16 forward + 16 backward steps at T = 500 where recursion_pair_kernel
// runs 500 + 500 on a chain of cross-lane exchanges, each step ~1400 / ~1900 full-rate fp64 instructions on all 64 lanes
// (recursion_pair: ~410 instructions per period at one element per lane).  Scratch: -Z_t and w_t (44 doubles per period,
// chunk-major so that lanes are contiguous: 180 KB per replicate instead of the 1 MB (Z, J) table).
// EM (template EM): the lane adds U_t = Cov(f_t+1, f_t) + f_t+1 f_t' and P_t + f_t f_t' of its counted periods into LDS
// accumulators (ds_add_f64, two lanes per slot); the wave then finishes S11 / S10 / S00 and the transition M-step in the
// element-per-lane layout exactly as recursion_pair_kernel does.
// Reference counterpart: none (dfm_functions.ipynb:21-23 declares `Parametric` only); oracle: oracle/kalman_oracle.py.
#include <stdlib.h>
#include "dfm_kernels.h"
#include "dfm_smallmat.h"
#include "dfm_grid.h"
#include "dfm_chunk_core.h"
#include "dfm_ctbuild.h"

// development ablations (scripts/dbg/r05/abl_chunk.sh builds one library per value; results are WRONG for any value but 0):
// 1 no output stores, 2 no row fetches (one scalar for every constant), 4 no LDS-DMA, 8 no table stores, 16 print shader-clock vs wall ticks
#ifndef DFM_CK_ABL
#define DFM_CK_ABL 0
#endif

namespace dfm {

namespace {

constexpr double kLog2PiC = 1.8378770664093454835606594728112;
constexpr int kCstStride = 320;     // doubles per replicate in chunk_cst
constexpr int kOffK = 0, kOffKT = 64, kOffQPhi = 128, kOffPhi = 192, kOffM0 = 256, kOffXi0 = 292, kOffLdc = 300, kOffQ0 = 301;
constexpr int kTermStride = 96;     // P_T (36) f_T (8) P_0 (36) f_0 (8)
constexpr int kScrRows = 22;        // double2 rows per period: 18 of -Z, 4 of w
constexpr int kAccSlots = 16;       // LDS accumulator slots per statistic (four lanes per slot)

using cdp = const double __attribute__((address_space(4)))*;

__device__ __forceinline__ cdp as_const(const double* p) { return (cdp)(unsigned long long)p; }
// a fresh, opaque copy of a uniform pointer: loads through it cannot be merged with (or hoisted to) loads through another copy,
// so a row of K lives in SGPRs only while it is used (all 64 + 64 + 64 constants at once would spill the scalar file)
__device__ __forceinline__ cdp launder(cdp p) {
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    return (cdp)v;
}
// ... and not before the values of `d` exist (the row fetches of a step are ordered along its arithmetic: dfm_chunk_core.h).
// Empty volatile asm statements keep their order: each "uses" one value, the last one produces the pointer.
__device__ __forceinline__ cdp launder_after(cdp p, const chunk::Deps& d) {
#pragma unroll
    for (int e = 0; e < chunk::R + 1; ++e)
        if (e < d.n) asm volatile("" : : "v"(d.v[e]));
    return launder(p);
}
struct RowSrc {
    cdp base;
    __device__ __forceinline__ chunk::Row8 operator()(int i, const chunk::Deps& d) const {
        chunk::Row8 r;
#if DFM_CK_ABL & 2
        const double c = base[0];
#pragma unroll
        for (int k = 0; k < 8; ++k) r.v[k] = c;
        return r;
#endif
        const cdp p = launder_after(base + 8 * i, d);
#pragma unroll
        for (int k = 0; k < 8; ++k) r.v[k] = p[k];
        return r;
    }
};
struct RcpDev {
    __device__ __forceinline__ double operator()(double d) const { return fast_rcp(d); }
};

__device__ __forceinline__ double wave_sum64(double v) {
    v += xor_lane<1>(v);
    v += xor_lane<2>(v);
    v += xor_lane<4>(v);
    v += xor_lane<8>(v);
    v += xor_lane<16>(v);
    v += xor_lane<32>(v);
    return v;
}
// a lane's state after its warm-up -> one slot of the smoother table (rows of 64 double2, lane = chunk: 22 coalesced stores), and
// the comparison of lane `src`'s stored state with the caller's own (the stores are visible: a wave_mem_fence() lies between)
__device__ __forceinline__ void save_state(double2* slot, int lane, const double (&m)[chunk::NP], const double (&x)[chunk::R]) {
    double2* dst = slot + lane;
#pragma unroll
    for (int k = 0; k < chunk::NP / 2; ++k) dst[k * 64] = make_double2(m[2 * k], m[2 * k + 1]);
#pragma unroll
    for (int k = 0; k < chunk::R / 2; ++k) dst[(chunk::NP / 2 + k) * 64] = make_double2(x[2 * k], x[2 * k + 1]);
}
__device__ __forceinline__ bool saved_state_close(const double2* slot, int src, const double (&m)[chunk::NP], const double (&x)[chunk::R],
                                                  double tol) {
    const double2* p = slot + src;
    chunk::Gap g{0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < chunk::NP / 2; ++k) {
        const double2 v = p[k * 64];
        chunk::gap_mat(g, m[2 * k], v.x);
        chunk::gap_mat(g, m[2 * k + 1], v.y);
    }
#pragma unroll
    for (int k = 0; k < chunk::R / 2; ++k) {
        const double2 v = p[(chunk::NP / 2 + k) * 64];
        chunk::gap_vec(g, x[2 * k], v.x);
        chunk::gap_vec(g, x[2 * k + 1], v.y);
    }
    return chunk::gap_close(g, tol);
}

// EM accumulators in LDS: statistic q, slot = lane & (kAccSlots - 1)
struct LdsAcc {
    static constexpr bool on = true;
    double* s10p;      // [64][kAccSlots]
    double* s11p;      // [36][kAccSlots]
    int slot;
    bool on10, on11;   // the step counts (lanes on garbage -- warm-up, beyond the sample -- add nothing)
    __device__ __forceinline__ bool want10() const { return on10; }
    __device__ __forceinline__ bool want11() const { return on11; }
    __device__ __forceinline__ void s10(int k, int n, double v) const { atomicAdd(&s10p[(8 * k + n) * kAccSlots + slot], v); }
    __device__ __forceinline__ void s11(int p, double v) const { atomicAdd(&s11p[p * kAccSlots + slot], v); }
};

// ---- one period's row of a chunk-major table (rows of 64 double2, lane = chunk) into the wave's LDS stage by LDS-DMA: nothing
// lands in a VGPR until the arithmetic asks for it (the 46 doubles of a period held beside the step's 100 live matrix entries
// cost ~500 VGPR <-> AGPR moves per step).  NROWS loads of 16 bytes per lane; lds_dst, base wave-uniform.
using lds_cptr = __attribute__((address_space(3))) char*;
#define DFM_CK_DMA_ROW                                   \
    "s_nop 0\n\t"                                        \
    "global_load_lds_dwordx4 %1, %2\n\t"                 \
    "v_add_u32 %1, 0x400, %1\n\t"                        \
    "s_add_u32 m0, m0, 0x400\n\t"
#define DFM_CK_DMA_ROW4 DFM_CK_DMA_ROW DFM_CK_DMA_ROW DFM_CK_DMA_ROW DFM_CK_DMA_ROW
template <int NROWS>
__device__ __forceinline__ void dma_rows(const double2* base, unsigned voff, unsigned lds_dst) {
    static_assert(NROWS == 22 || NROWS == 23, "table rows per period");
#if DFM_CK_ABL & 4
    return;
#endif
    unsigned keep;
    if constexpr (NROWS == 23) {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            DFM_CK_DMA_ROW4 DFM_CK_DMA_ROW4 DFM_CK_DMA_ROW4 DFM_CK_DMA_ROW4 DFM_CK_DMA_ROW4 DFM_CK_DMA_ROW DFM_CK_DMA_ROW DFM_CK_DMA_ROW
            "s_mov_b32 m0, %0"
            : "=&s"(keep), "+v"(voff)
            : "s"(base), "s"(lds_dst)
            : "memory", "scc");
    } else {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            DFM_CK_DMA_ROW4 DFM_CK_DMA_ROW4 DFM_CK_DMA_ROW4 DFM_CK_DMA_ROW4 DFM_CK_DMA_ROW4 DFM_CK_DMA_ROW DFM_CK_DMA_ROW
            "s_mov_b32 m0, %0"
            : "=&s"(keep), "+v"(voff)
            : "s"(base), "s"(lds_dst)
            : "memory", "scc");
    }
}
// What the wave stores to global memory (the smoother table, the warm-up states, the terminal states) is read back by the SAME wave --
// by other lanes, through the same CU's write-through L1.  The stores have to be complete and ordered before the loads, nothing more:
// a workgroup-scope fence (s_waitcnt).  __threadfence() is agent scope: buffer_wbl2 + buffer_inv -- the XCD's L2 writes back every dirty
// line it holds (all the outputs the batch has just stored) once per fence and replicate.
__device__ __forceinline__ void wave_mem_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lds_reads() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// the observation stage: 23 rows of 64 double2 (lane = chunk), as the table has them: doubles 0..35 of a lane's period C_t (packed),
// 36..43 b_t, 44 s_t, 45 n_t log 2 pi + log det R_t
struct StageObs {
    const double* st;   // stage + 2 * lane
    __device__ __forceinline__ void ready() const { wait_dma(); }
    __device__ __forceinline__ double at(int e) const { return st[(e >> 1) * 128 + (e & 1)]; }
    __device__ __forceinline__ double c(int p) const { return at(p); }
    __device__ __forceinline__ double b(int i) const { return at(chunk::NP + i); }
};
struct StageZw {
    const double* st;
    __device__ __forceinline__ void ready() const { wait_dma(); }
    __device__ __forceinline__ double z(int p) const { return st[(p >> 1) * 128 + (p & 1)]; }
    __device__ __forceinline__ double w(int i) const { return st[((chunk::NP + i) >> 1) * 128 + (i & 1)]; }
};

__device__ __forceinline__ int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

}  // namespace

// ---- per-replicate constants ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void chunk_prep_kernel(RecursionArgs a) {
    constexpr int R = 8, TS = kTileStride<R>, RT = R * TS;
    __shared__ double tiles[2 * RT];
    double* L0 = tiles;
    double* L1 = tiles + RT;
    const int lane = threadIdx.x, b = blockIdx.x;
    const int i = lane >> 3, j = lane & 7;
    Grid<R> G;
    G.l = lane; G.i = i; G.j = j;
    const double Ael = a.A[(size_t)b * 64 + lane];
    double Qi = a.Q[(size_t)b * 64 + lane];
    double Om0 = a.P0[(size_t)b * 64 + lane];
    const double mu0c = a.mu0[(size_t)b * R + j];
    const double detQ = G.sweep_inverse(Qi);
    const double detP0 = G.sweep_inverse(Om0);
    L0[TS * i + j] = Qi;
    L1[TS * j + i] = Ael;                                          // A'
    G.sync();
    const double K = dot_rows<R>(L0, L1, i, j);                    // K = Q^-1 A
    G.sync();
    L0[TS * j + i] = K;                                            // K'
    G.sync();
    const double Phi = dot_rows<R>(L0, L1, i, j);                  // Phi = K' A
    const double xi0r = G.sum_j(Om0 * mu0c);                       // xi_0 = P0^-1 mu0 (row-distributed)
    const double q0 = G.sum_i(G.sum_j(i == j ? mu0c * xi0r : 0.0));
    double* cst = a.chunk_cst + (size_t)b * kCstStride;
    cst[kOffK + lane] = K;
    cst[kOffKT + 8 * j + i] = K;
    cst[kOffQPhi + lane] = Qi + Phi;
    cst[kOffPhi + lane] = Phi;
    if (j <= i) cst[kOffM0 + chunk::pidx(i, j)] = Om0 + Phi;
    if (j == 0) cst[kOffXi0 + i] = xi0r;
    if (lane == 0) {
        cst[kOffLdc] = log(detP0) + (double)a.T * log(detQ);
        cst[kOffQ0] = q0;
    }
}

// ---- the observation table: obs[b][slot][23][lane] double2, period t = L lane + slot -- doubles 0..35 C_t (8 x 8 packed, zero beyond
// the RC x RC block), 36..43 b_t, 44 s_t, 45 n_t log 2 pi + sum of log R over the observed cells.  Chunk-major like the smoother table:
// the pass's LDS-DMA of a step is 23 contiguous KBs, and the kernels below store whole KBs.  EVERY slot of all 64 lanes is written:
// beyond the sample a benign row (the replicate's full Gram matrix, zeros) -- lanes step through such periods uncounted.
//
// From the collapse kernels' per-period arrays (bcol, scol, nobs, ldrow; C_t only where a cell is missing, else the replicate's
// Cfull): collapse_kernel's shapes (N > 224, collapsed observations 2 / 4 wide).  One wave per (replicate, slot), lane = chunk.
template <int RC>
__global__ __launch_bounds__(64) void chunk_bridge_kernel(RecursionArgs a) {
    using namespace chunk;
    constexpr int NPC = RC * (RC + 1) / 2;
    const int T = a.T, L = a.chunk_L;
    const int b = blockIdx.x / L, slot = blockIdx.x - b * L, lane = threadIdx.x;
    const int t = L * lane + slot;
    const bool in = t < T;
    const size_t bt = (size_t)b * T + (in ? t : 0);
    const int n = in ? a.nobs[bt] : a.N;
    const bool full = n == a.N || a.Ct == nullptr;
    double v[2 * kObsRows];
#pragma unroll
    for (int k = 0; k < 2 * kObsRows; ++k) v[k] = 0.0;
#pragma unroll
    for (int i = 0; i < RC; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j)
            v[pidx(i, j)] = full ? a.Cfull[(size_t)b * RC * RC + i * RC + j] : a.Ct[bt * NPC + pidx(i, j)];
    if (in) {
#pragma unroll
        for (int i = 0; i < RC; ++i) v[NP + i] = a.bcol[bt * RC + i];
        v[NP + R] = a.scol[bt];
        v[NP + R + 1] = (double)n * kLog2PiC + (n == a.N ? a.ldfull[b] : a.ldrow[bt]);
    }
    double2* dst = reinterpret_cast<double2*>(a.chunk_obs) + ((size_t)b * L + slot) * kObsRows * 64 + lane;
#pragma unroll
    for (int k = 0; k < kObsRows; ++k) dst[k * 64] = make_double2(v[2 * k], v[2 * k + 1]);
}

// ---- ... and back, for the replicates the pass hands to the sequential kernel when the collapse wrote rows and masks only
// (collapse_miss_kernel's table mode): bcol, scol, C_t of every period; nobs = 0 and ldrow = n_t log 2 pi + sum log R make the sequential kernels'
// "n log 2 pi + sum of log R" come out right without n_t (they read C_t / ldrow of every period whose nobs differs from N).
// Without a C_t array (the caller promised a balanced panel) the periods are all alike: nobs = N, Cfull and ldfull from period 0.
// RC: width of the per-period arrays the sequential kernel reads (RecursionArgs::Rc: 2 / 4 when the loadings are narrower than the
// 8-wide state -- the leading RC x RC block of the packed C_t is its first RC (RC + 1) / 2 entries; the padding block is zero)
template <int RC>
__global__ __launch_bounds__(64) void chunk_unbridge_kernel(RecursionArgs a, double* bcol, double* scol, int* nobs, double* ldrow, double* Ct,
                                                          double* Cfull, double* ldfull) {
    using namespace chunk;
    constexpr int NPC = RC * (RC + 1) / 2;
    const int T = a.T, L = a.chunk_L;
    const int b = blockIdx.x / L, slot = blockIdx.x - b * L, lane = threadIdx.x;
    if (a.chunk_fail && a.chunk_fail[b] == 0) return;          // (no flags: every replicate -- the companion models' collapse, capi.hip comp_table)
    const int t = L * lane + slot;
    if (t >= T) return;
    const size_t bt = (size_t)b * T + t;
    const double2* src = reinterpret_cast<const double2*>(a.chunk_obs) + ((size_t)b * L + slot) * kObsRows * 64 + lane;
    double v[2 * kObsRows];
#pragma unroll
    for (int k = 0; k < kObsRows; ++k) { const double2 u = src[k * 64]; v[2 * k] = u.x; v[2 * k + 1] = u.y; }
#pragma unroll
    for (int i = 0; i < RC; ++i) bcol[bt * RC + i] = v[NP + i];
    scol[bt] = v[NP + R];
    if (Ct) {
        nobs[bt] = 0;                                          // (never equal to N >= 1: the period reads its own C_t / ldrow; adds 0 to the n-sum)
        ldrow[bt] = v[NP + R + 1];
#pragma unroll
        for (int k = 0; k < NPC; ++k) Ct[bt * NPC + k] = v[k];
    } else {
        nobs[bt] = a.N;
        if (t == 0) {
            ldfull[b] = v[NP + R + 1] - (double)a.N * kLog2PiC;
#pragma unroll
            for (int i = 0; i < RC; ++i)
#pragma unroll
                for (int j = 0; j < RC; ++j) Cfull[(size_t)b * RC * RC + i * RC + j] = v[pidx(i, j)];
        }
    }
}

// ---- the pass ---------------------------------------------------------------------------------------------------------------
// OUT8: outputs are full 8-wide rows (r = 8, rl = 0 | 8)
template <bool EM, bool OUT8>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void recursion_chunk_kernel(RecursionArgs a) {
    using namespace chunk;
    extern __shared__ __attribute__((aligned(16))) double csm[];
    const int lane = threadIdx.x, b = blockIdx.x;
    const int T = a.T;
    const int L = a.chunk_L, W = a.chunk_W, NS = L + W;
    const int c0 = L * lane;
    const int jtop = (T - 1) / L;
    const cdp cst = as_const(a.chunk_cst + (size_t)b * kCstStride);
    const RowSrc krow{cst + kOffK}, ktrow{cst + kOffKT}, qrow{cst + kOffQPhi};
    const double tol = a.chunk_tol;
    const double2* obs = reinterpret_cast<const double2*>(a.chunk_obs) + (size_t)b * L * kObsRows * 64;   // [slot][23][lane]
    double2* scr = reinterpret_cast<double2*>(a.chunk_scr) + (size_t)b * (L + 2) * kScrRows * 64;
    double2* sav_f = scr + (size_t)L * kScrRows * 64;            // the states after the warm-ups (forward, backward): slots L, L + 1
    double2* sav_b = sav_f + kScrRows * 64;
    double* term = a.chunk_term + (size_t)b * kTermStride;
    const int r = a.r;
    const int rl = a.rl > 0 ? a.rl : R;
    const int npr = r * (r + 1) / 2;
    // A replicate whose filter forgets too slowly for the warm-up fails its boundary check in EVERY iteration of an EM run (the real
    // Stock-Watson window: all of them, whatever the warm-up up to 20 periods -- weakly loaded factors forget over tens of periods).  The
    // count of CONSECUTIVE failures survives in the handle's workspace between the iterations of a run (k = 0 clears it; a first
    // iteration from a rough start may fail where the later ones pass): after three in a row a replicate leaves at once and is the
    // sequential kernel's, instead of spending 16 + 16 steps per lane on a result that is thrown away; every 8th iteration it is tried again.
    if (a.chunk_skip) {
        if (a.k == 0) { if (lane == 0) a.chunk_skip[b] = 0; }
        else if (a.chunk_skip[b] >= 3 && (a.k & 7) != 0) { if (lane == 0) a.chunk_fail[b] = 1; return; }
    }

    // LDS: the stage (64 x 368 bytes), then (EM) the accumulators [64 + 36][kAccSlots]
    // and two 8 x 8 tiles of the epilogue
    double* stage = csm;
    const double* stl = stage + 2 * lane;                          // (both tables' stage: rows of 64 double2, lane = chunk)
    const double* sto = stl;
    const unsigned stage_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cptr)(reinterpret_cast<char*>(stage)));
    double* acc10 = csm + 2 * 64 * kObsRows;
    double* acc11 = acc10 + 64 * kAccSlots;
    if constexpr (EM) {
        for (int e = lane; e < 100 * kAccSlots; e += 64) acc10[e] = 0.0;
        wave_lds_sync();
    }

    // the observation rows of the periods c0 + d of all lanes (d may be negative: slot d mod L of the lanes below; ct_build / the bridge
    // fill the table for every slot of all 64 lanes -- benign rows beyond the sample) -> stage
    auto issue_obs = [&](int d) {
        const int sh = floor_div(d, L), slot = d - sh * L;
        int src = lane + sh;
        src = src < 0 ? 0 : (src > 63 ? 63 : src);
        dma_rows<kObsRows>(obs + (size_t)slot * kObsRows * 64, (unsigned)src * 16u, stage_lds);
    };
    auto store_row = [&](int trow, const double (&P)[NP], const double (&f)[R]) {
#if DFM_CK_ABL & 1
        if (a.chunk_tol < 0.0)
#endif
        if constexpr (OUT8) {
            if (a.P_smooth) {
                double2* dst = reinterpret_cast<double2*>(a.P_smooth + ((size_t)b * T + trow) * NP);
#pragma unroll
                for (int k = 0; k < NP / 2; ++k) dst[k] = make_double2(P[2 * k], P[2 * k + 1]);
            }
            double2* df = reinterpret_cast<double2*>(a.f_smooth + ((size_t)b * T + trow) * R);
#pragma unroll
            for (int k = 0; k < R / 2; ++k) df[k] = make_double2(f[2 * k], f[2 * k + 1]);
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (i >= r) break;
                a.f_smooth[((size_t)b * T + trow) * r + i] = i < rl ? f[i] : 0.0;
                if (a.P_smooth) {
#pragma unroll
                    for (int j = 0; j <= i; ++j)
                        a.P_smooth[((size_t)b * T + trow) * npr + pidx(i, j)] = (i < rl && j < rl) ? P[pidx(i, j)] : (i == j ? 1.0 : 0.0);
                }
            }
        }
    };

#if DFM_CK_ABL & 16
    const long long ck0 = clock64(), wk0 = wall_clock64();
#endif
    // =================================================== forward ===========================================================
    double m[NP], xi[R];
    {   // guess at the start of the window: a step "with J = 0" on the data of the period before it
        issue_obs(-W - 1);
        const StageObs ob{sto};
        ob.ready();
        const cdp qf = launder(cst + kOffQPhi);
#pragma unroll
        for (int i = 0; i < R; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) m[pidx(i, j)] = qf[8 * i + j] + ob.c(pidx(i, j));
            xi[i] = ob.b(i);
        }
        wait_lds_reads();
        issue_obs(-W);
    }
    LogProd lp;
    double sxw = 0.0, ssum = 0.0, ldnsum = 0.0;
    double ldT = 0.0, xfT = 0.0;
    const int ucap = W + (T - 1) % L;
    for (int u = 0; u < NS; ++u) {
        const int t = c0 - W + u;
        if (u <= W && (W - u) % L == 0) {                          // (wave-uniform) some lane's window reaches the initial state here
            const int jr = (W - u) / L;
            const cdp m0 = launder(cst + kOffM0);
            if (lane == jr) {
#pragma unroll
                for (int k = 0; k < NP; ++k) m[k] = m0[k];
#pragma unroll
                for (int i = 0; i < R; ++i) xi[i] = m0[kOffXi0 - kOffM0 + i];
            }
        }
        if (u == W) save_state(sav_f, lane, m, xi);
        const bool counted = u >= W && t < T;
        double det, xw;
        fwd_step(m, xi, StageObs{sto}, det, xw, krow, qrow, RcpDev{}, [&](const double (&zn)[NP], const double (&w)[R]) {
            if (counted && !((DFM_CK_ABL & 8) && tol >= 0.0)) {
                double2* dst = scr + (size_t)(u - W) * kScrRows * 64 + lane;
#pragma unroll
                for (int k = 0; k < NP / 2; ++k) dst[k * 64] = make_double2(zn[2 * k], zn[2 * k + 1]);
#pragma unroll
                for (int k = 0; k < R / 2; ++k) dst[(NP / 2 + k) * 64] = make_double2(w[2 * k], w[2 * k + 1]);
            }
        });
        {
            const double st_s = StageObs{sto}.at(NP + R), st_l = StageObs{sto}.at(NP + R + 1);
            lp.mul(counted ? det : 1.0);
            sxw += counted ? xw : 0.0;
            ssum += counted ? st_s : 0.0;
            ldnsum += counted ? st_l : 0.0;
        }
        wait_lds_reads();                                          // the stage is re-armed: its reads are done
        if (u + 1 < NS) issue_obs(u + 1 - W);
        if (u == ucap) {                                           // (wave-uniform) the top lane has just taken period T - 1
            const cdp ph = launder(cst + kOffPhi);
            double om[NP];
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) om[pidx(i, j)] = m[pidx(i, j)] - ph[8 * i + j];
            const double detT = sweep8(om, RcpDev{});              // om = -P_T
            double PT[NP], fT[R];
#pragma unroll
            for (int k = 0; k < NP; ++k) PT[k] = -om[k];
            double dot = 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < R; ++q) s = fma(PT[pidx(i, q)], xi[q], s);
                fT[i] = s;
                dot = fma(xi[i], s, dot);
            }
            if (lane == jtop) {
                ldT = log(detT);
                xfT = dot;
#pragma unroll
                for (int k = 0; k < NP; ++k) term[k] = PT[k];
#pragma unroll
                for (int i = 0; i < R; ++i) term[NP + i] = fT[i];
                store_row(T - 1, PT, fT);
            }
        }
    }
    // log-likelihood (oracle/info_form.py): the lanes' counted periods add up
    double ll;
    {
        const double ldz = wave_sum64(lp.log_value());
        const double sx = wave_sum64(sxw), ss = wave_sum64(ssum), ls = wave_sum64(ldnsum);
        const double ldTt = wave_sum64(ldT), xfTt = wave_sum64(xfT);
        const double LD = ldTt + cst[kOffLdc] + ldz;
        const double qd = cst[kOffQ0] - xfTt - sx;
        ll = -0.5 * (ls + LD + ss + qd);
    }
    wave_mem_fence();                                               // the table, the warm-up states and the terminal state are read back below
    // boundary check, forward: the state lane + 1 started its chunk from against the state this lane leaves its own chunk with
    bool ok = lane + 1 > jtop || saved_state_close(sav_f, lane < 63 ? lane + 1 : 63, m, xi, tol);
    if (!(__all(ok) != 0 && ll == ll)) {                           // (wave-uniform) a forward boundary is off: no backward sweep for nothing
        // off by four decades and more: the filter forgets too slowly for this warm-up whatever the next iterations do to the parameters
        // (the real Stock-Watson window: 4e-3 against 1e-10) -- counted as three failures at once, the replicate goes straight to the
        // sequential kernel from the next iteration on (retried every 8th, as after three ordinary failures in a row)
        const bool near = lane + 1 > jtop || saved_state_close(sav_f, lane < 63 ? lane + 1 : 63, m, xi, 1e4 * tol);
        const int inc = (__all(near) != 0) ? 1 : 3;
        if (lane == 0) { a.chunk_fail[b] = 1; if (a.chunk_skip) a.chunk_skip[b] = (a.k == 0 ? 0 : a.chunk_skip[b]) + inc; }
        return;
    }

    // =================================================== backward ==========================================================
    double P[NP], f[R];
#pragma unroll
    for (int k = 0; k < NP; ++k) P[k] = 0.0;
#pragma unroll
    for (int i = 0; i < R; ++i) f[i] = 0.0;
    auto issue_zw = [&](int d) {                                   // period c0 + d: slot d % L of the lane d / L above
        const int sh = d / L, slot = d - sh * L;
        const int src = lane + sh < 64 ? lane + sh : 63;
        dma_rows<kScrRows>(scr + (size_t)slot * kScrRows * 64, (unsigned)src * 16u, stage_lds);
    };
    wait_lds_reads();
    issue_zw(NS - 1);
    for (int u = 0; u < NS; ++u) {
        const int d = NS - 1 - u;
        const int t = c0 + d;                                      // the step that takes state t + 1 to state t
        {
            const int q = T - NS + u;                              // (wave-uniform) lane q / L takes period T - 1 now: exact terminal state
            if (q >= 0 && q % L == 0) {
                const int jr = q / L;
                if (lane == jr) {
#pragma unroll
                    for (int k = 0; k < NP; ++k) P[k] = term[k];
#pragma unroll
                    for (int i = 0; i < R; ++i) f[i] = term[NP + i];
                }
            }
        }
        if (u == W) save_state(sav_b, lane, P, f);
        const bool counted = u >= W && t < T;
        if constexpr (EM) {
            LdsAcc acc{acc10, acc11, lane & (kAccSlots - 1), counted, counted && t >= 1};
            bwd_step(P, f, StageZw{stl}, ktrow, acc);
        } else {
            bwd_step(P, f, StageZw{stl}, ktrow, NoAcc{});
        }
        wait_lds_reads();
        if (u + 1 < NS) issue_zw(d - 1);
        if (counted) {
            if (t >= 1) store_row(t - 1, P, f);
            else {
#pragma unroll
                for (int k = 0; k < NP; ++k) term[44 + k] = P[k];
#pragma unroll
                for (int i = 0; i < R; ++i) term[44 + NP + i] = f[i];
            }
        }
    }
    wave_mem_fence();
    // ... backward: the state lane - 1 started its chunk from (period L * lane) against the state this lane arrives there with
    ok = ok && (lane == 0 || L * lane >= T || saved_state_close(sav_b, lane > 0 ? lane - 1 : 0, P, f, tol));
#if DFM_CK_ABL & 16
    if (lane == 0 && (b == 0 || b == 777)) printf("CKCLK b=%d shader cycles %lld wall ticks %lld\n", b, clock64() - ck0, wall_clock64() - wk0);
#endif
    const bool good = __all(ok) != 0 && ll == ll;                  // (a NaN log-likelihood also goes to the sequential kernel)
    if (lane == 0) { a.chunk_fail[b] = good ? 0 : 1; if (a.chunk_skip) a.chunk_skip[b] = good ? 0 : (a.k == 0 ? 0 : a.chunk_skip[b]) + 1; }
    if (!good) return;
    if (lane == 0) {
        a.loglik[b] = ll;
        if (a.ncov) a.ncov[b] = T;
    }
    if constexpr (!EM) return;

    // =================================================== EM epilogue =======================================================
    // element-per-lane layout (lane = 8 i + j), as recursion_pair_kernel's: sufficient statistics, EM bookkeeping, transition M-step
    if constexpr (EM) {
        constexpr int TS = kTileStride<R>, RT = R * TS, RR = 64;
        double* L0 = acc10 + 100 * kAccSlots;
        double* L1 = L0 + RT;
        bool em_apply = true;
        if (a.active) {
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
        wave_mem_fence();                                           // state 0 (term[44..]) and the LDS accumulators
        wave_lds_sync();
        const int i = lane >> 3, j = lane & 7;
        Grid<R> G;
        G.l = lane; G.i = i; G.j = j;
        const int Rc = a.Rc > 0 ? a.Rc : R;
        const bool inC = i < Rc && j < Rc;
        const bool inL = i < rl && j < rl;
        double S10 = 0.0, S11 = 0.0;
        {
            const int pk = pidx(i, j);
#pragma unroll 4
            for (int s = 0; s < kAccSlots; ++s) {
                const int ss = (s + lane) & (kAccSlots - 1);       // staggered: the lanes of a read hit different banks
                S10 += acc10[lane * kAccSlots + ss];
                S11 += acc11[pk * kAccSlots + ss];
            }
        }
        const double PsT = term[pidx(i, j)], Ps = term[44 + pidx(i, j)];
        const double fTi = term[NP + i], fTj = term[NP + j];
        const double f0r = term[44 + NP + i], f0c = term[44 + NP + j];
        const double fTfT = fTi * fTj;
        S11 += PsT + fTfT;                                         // state T
        const double S00 = S11 - (PsT + fTfT) + fma(f0r, f0c, Ps);
        const size_t o = (size_t)b * RR + lane;
        const bool narrow = a.rl > 0;
        if (!narrow) a.S11[o] = S11;
        a.S10[o] = S10;
        a.S00[o] = S00;
        a.P0s[o] = Ps;
        if (j == 0) a.f0s[(size_t)b * R + i] = f0r;
        if (a.A_out) {
            double inv = S00;
            (void)G.sweep_inverse(inv);
            G.sync();
            L0[TS * i + j] = S10;
            L1[TS * i + j] = inv;
            G.sync();
            const double An = dot_rows<R>(L0, L1, i, j);
            G.sync();
            L1[TS * i + j] = An;
            G.sync();
            double Qn = (S11 - dot_rows<R>(L1, L0, i, j)) / (double)T;
            Qn = 0.5 * (Qn + G.transposed(Qn));
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
                a.A_out[o] = An;
                a.Q_out[o] = Qn;
                a.P0_out[o] = P0n;
                if (j == 0) a.mu0_out[(size_t)b * R + i] = f0r;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
namespace {
// (the route switches DFM_NO_CHUNK, DFM_CHUNK_W, DFM_CHUNK_TOL belong to the HANDLE -- capi.hip reads them at dfm_create and hands them
// over in RecursionArgs::chunk_W / chunk_tol, or leaves chunk_scr null -- so that one process can hold contexts on either kernel)
constexpr int kChunkWarmDefault = 8;
constexpr double kChunkTolDefault = 1e-10;
int chunk_warm(const RecursionArgs& a) { return a.chunk_W > 0 ? (a.chunk_W > 64 ? 64 : a.chunk_W) : kChunkWarmDefault; }
size_t chunk_lds_bytes(bool em) {
    return (size_t)(2 * 64 * kObsRows) * sizeof(double) + (em ? (size_t)(100 * kAccSlots + 2 * 8 * kTileStride<8>) * sizeof(double) : 0);
}
}  // namespace

int recursion_chunk_len(int T) {
    const int L = (T + 63) / 64;
    return L < 4 ? 4 : L;
}
size_t recursion_chunk_scratch_bytes(int B, int T) {
    return (size_t)B * (recursion_chunk_len(T) + 2) * kScrRows * 64 * sizeof(double2);     // (+ 2 slots: the states after the warm-ups)
}
size_t recursion_chunk_obs_bytes(int B, int T) {
    return (size_t)B * recursion_chunk_len(T) * kObsRows * 64 * sizeof(double2);   // [B][L][23][64] double2: every slot of all 64 lanes
}
size_t recursion_chunk_rows_bytes(int B, int T) { return (size_t)B * T * 14 * sizeof(double); }   // collapse_miss_kernel's 112-byte rows

// Plain factor model at Rp = 8 in information form (no companion state: the EM epilogue here has no shift rows), collapsed
// observations 2, 4 or 8 wide; a sample long enough for two lanes.  The sequential kernel must be able to take a replicate back.
bool recursion_chunk_supported(int Rpad, const RecursionArgs& a) {
    if (Rpad != 8 || a.cov || a.kdim != 0 || a.kb != 0 || a.ka != 0 || a.ct_r != 0) return false;
    if (!a.chunk_scr || !a.chunk_cst || !a.chunk_term || !a.chunk_fail || !a.chunk_obs) return false;
    if ((size_t)a.B * recursion_chunk_len(a.T) > 0x7fffffffu) return false;   // (grid of the bridge kernels)
    if (a.rl != 0 && a.Rc == 0) return false;
    const int Rc = a.Rc > 0 ? a.Rc : 8;
    if (Rc != 2 && Rc != 4 && Rc != 8) return false;
    if (a.rl != 0 && a.rl != Rc) return false;
    if (!recursion_wave8_fits(a.T)) return false;
    const int L = recursion_chunk_len(a.T);
    return a.T >= 2 * (L + chunk_warm(a));
}

template <bool EM>
static hipError_t launch_chunk_em(const RecursionArgs& a, hipStream_t s) {
    const bool out8 = a.r == 8 && (a.rl == 0 || a.rl == 8);
    const size_t lds = chunk_lds_bytes(EM);
    if (out8) hipLaunchKernelGGL((recursion_chunk_kernel<EM, true>), dim3(a.B), dim3(64), lds, s, a);
    else hipLaunchKernelGGL((recursion_chunk_kernel<EM, false>), dim3(a.B), dim3(64), lds, s, a);
    return hipGetLastError();
}

// a.chunk_obs_ready != 0: the collapse kernel wrote the chunk-major table itself
hipError_t launch_recursion_chunk(const RecursionArgs& a0, hipStream_t s) {
    note_kernel("recursion_chunk_kernel");
    RecursionArgs a = a0;
    a.chunk_L = recursion_chunk_len(a.T);
    a.chunk_W = chunk_warm(a0);
    a.chunk_tol = a0.chunk_tol > 0.0 ? a0.chunk_tol : kChunkTolDefault;
    // EM: the stop rule (ll_k - ll_k-1) / avg < tol is evaluated on this kernel's log-likelihood, which equals the sequential
    // recursion's to ~chunk_tol only -- keep the boundary tolerance two decades under the caller's so that chunk noise cannot decide it
    if (a0.S11 != nullptr && a0.tol > 0.0 && 1e-2 * a0.tol < a.chunk_tol) a.chunk_tol = 1e-2 * a0.tol;
    hipLaunchKernelGGL(chunk_prep_kernel, dim3(a.B), dim3(64), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (!a.chunk_obs_ready) {                                       // (else collapse_miss_kernel wrote the table itself: dfm_ctbuild.h)
        const int Rc = a.Rc > 0 ? a.Rc : 8;
        const dim3 grid((unsigned)a.B * (unsigned)a.chunk_L);
        if (Rc == 8) hipLaunchKernelGGL(chunk_bridge_kernel<8>, grid, dim3(64), 0, s, a);
        else if (Rc == 4) hipLaunchKernelGGL(chunk_bridge_kernel<4>, grid, dim3(64), 0, s, a);
        else hipLaunchKernelGGL(chunk_bridge_kernel<2>, grid, dim3(64), 0, s, a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return a.S11 != nullptr ? launch_chunk_em<true>(a, s) : launch_chunk_em<false>(a, s);
}

// the per-period rows of the replicates with chunk_fail set, for the sequential kernel (the table came from collapse_gemm_kernel)
hipError_t launch_chunk_unbridge(const RecursionArgs& a0, hipStream_t s) {
    RecursionArgs a = a0;
    a.chunk_L = recursion_chunk_len(a.T);
    const int Rc = a.Rc > 0 ? a.Rc : 8;
    const dim3 grid((unsigned)a.B * (unsigned)a.chunk_L);
#define DFM_UNBRIDGE(RC_)                                                                                                                   \
    hipLaunchKernelGGL(chunk_unbridge_kernel<RC_>, grid, dim3(64), 0, s, a, const_cast<double*>(a.bcol), const_cast<double*>(a.scol),      \
                       const_cast<int*>(a.nobs), const_cast<double*>(a.ldrow), const_cast<double*>(a.Ct), const_cast<double*>(a.Cfull),     \
                       const_cast<double*>(a.ldfull))
    if (Rc == 8) DFM_UNBRIDGE(8);
    else if (Rc == 4) DFM_UNBRIDGE(4);
    else DFM_UNBRIDGE(2);
#undef DFM_UNBRIDGE
    return hipGetLastError();
}

}  // namespace dfm
