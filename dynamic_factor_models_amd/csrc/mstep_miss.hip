// mstep_miss.hip -- the loadings half of the M-step for panels WITH missing cells on the f64 matrix pipe
// (Banbura & Modugno 2014; reference counterpart: the per-series OLS over each series' complete cases,
// dfm_functions.ipynb:355-362 and :391-404).
//
//   per series i, with m_it = 1 where x_it is missing and xz_it = x_it with NaN -> 0:
//     D_i   = sum_t m_it vec(E_t)        E_t = f_t|T f_t|T' + P_t|T (packed lower, r (r + 1) / 2 values)
//     Sxf_i = sum_t xz_it f_t|T          Sxx_i = sum_t xz_it^2        n_i = sum_t (1 - m_it)
//     Sff_i = S11 - D_i,  lam_i = Sff_i^-1 Sxf_i,  R_i = (Sxx_i - lam_i' Sxf_i) / n_i
//
// mstep_lam_kernel (mstep.hip: lane = series) adds vec(E_t) into a per-series accumulator for every missing cell -- 528
// uncoalesced read-modify-writes of global memory per cell at Rp = 32: 337 ms per EM iteration at BASELINE config 4 with 10 %
// of the cells missing.  Here both contractions are ONE product per replicate,
//     [D | Sxf] (N x (np + r))  =  [M | XZ]' (N x T, two A operands from the same panel element)  x  V (T x (np + r)),
// V_t = [vec(E_t) | f_t] written once per iteration by mmw_vec_kernel (it is read back from L2 by the 8 series blocks of a
// replicate, which run on the same XCD at the same pace), on `v_mfma_f64_16x16x4`:
//   * an item = (replicate, block of 16 SG series); workgroups of 8 waves, wave = (series group sg, column group cg), holding
//     its 16 x 16 accumulator tiles in registers for the whole item (r = 20: 14 tiles of D + 2 of Sxf = 64 doubles per lane);
//   * the item streams through LDS in stages of 8 periods, three stage buffers, all by `global_load_lds_dwordx4` with a
//     counted `s_waitcnt vmcnt` (every wave issues the same number of DMAs per stage); rows 128 bytes (mod 256) apart;
//   * the mask costs nothing extra: it is the A operand (1.0 | 0.0) of the D tiles, as xz is the A operand of the Sxf tiles.
// mmw_solve_kernel (a lane per row of a series' normal matrix, 64 / r series per wave) then solves by symmetric elimination.
// Flops at config 4: 2 x 256 x 1000 x 2000 x 256 = 2.6e11 = 3.3 ms at the fp64 matrix peak.
#include <stdlib.h>

#include "dfm_device.h"
#include "dfm_kernels.h"

namespace dfm {

namespace {

using lds_char_ptr_mm = __attribute__((address_space(3))) char*;
using lds_cvd_ptr_mm = const __attribute__((address_space(3))) double*;   // (not volatile: a "memory" clobber follows every barrier)
__device__ __forceinline__ double lds_read64mm(unsigned a) { return *(lds_cvd_ptr_mm)(size_t)a; }
typedef double mm_v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16mm(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

constexpr int kMmWaves = 8;
// s_waitcnt vmcnt(n) for a wave-uniform n known only at run time (the immediate must be a constant): n in 0 .. 12
__device__ __forceinline__ void wait_vm_upto(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // (never less patient than needed)
    }
}
constexpr unsigned kMmLdsMax = 156 * 1024;               // stage buffers of a workgroup

struct MmGeo {
    int r, Rp, npr;                                      // caller's factor count, padded width, r (r + 1) / 2
    int ntm, ntf, tt;                                    // tiles of D, of Sxf, total (16 columns each)
    int cg, sg, ser;                                     // column groups, series groups, series per item
    int tpw;                                             // tiles per wave (upper bound; the last column group may hold fewer)
    int nv;                                              // 1-KB DMAs per row of V
    unsigned vrowB, vstride, pstride, panelB, stageB;    // bytes: a row of V, its LDS stride, panel row stride, panel block, stage
    int U;                                               // DMAs per wave and stage
    int kp, nbuf;                                        // periods per stage, stage buffers (2: one stage ahead, 3: two)
};

// kp_want: 0 = the deepest stage that fits twice (32, then 16 periods), else 8 periods in three buffers
MmGeo mm_geo(int Rp, int r, int kp_want = 0, int nbuf_want = 0) {
    MmGeo g;
    g.r = r; g.Rp = Rp; g.npr = r * (r + 1) / 2;
    g.ntm = (g.npr + 15) / 16;
    g.ntf = (Rp + 15) / 16;
    g.tt = g.ntm + g.ntf;
    g.cg = g.tt > 21 ? 2 : 1;
    g.sg = kMmWaves / g.cg;
    g.ser = 16 * g.sg;
    g.tpw = (g.tt + g.cg - 1) / g.cg;
    g.vrowB = (unsigned)g.tt * 128u;
    g.nv = (int)((g.vrowB + 1023u) / 1024u);
    g.vstride = g.vrowB + ((g.tt & 1) ? 0u : 128u);      // = 128 (mod 256): the four periods of a step on distinct banks
    g.pstride = (unsigned)g.ser * 8u + 128u;
    g.kp = 8; g.nbuf = 3;
    for (int kp : {32, 16}) {
        if ((kp_want != 0 && kp_want != kp) || g.tpw > 18) continue;      // (21 tile slots + a deep stage spill)
        // narrow states (4 tile slots, 95 VGPRs): 16 periods, so that TWO workgroups share a CU -- the stages are too short to
        // hide the DMA latency one stage ahead (C2 shape: 0.68 ms against 0.77 with 32 periods and one workgroup)
        if (kp_want == 0 && kp == 32 && g.tpw <= 4) continue;
        if (2u * (unsigned)kp * (g.pstride + g.vstride) <= kMmLdsMax) { g.kp = kp; g.nbuf = 2; break; }
    }
    if (g.nbuf == 3 && nbuf_want == 4 && 4u * 8u * (g.pstride + g.vstride) <= kMmLdsMax) g.nbuf = 4;   // (diagnostics: DFM_MM_NBUF)
    g.panelB = (unsigned)g.kp * g.pstride;
    g.stageB = g.panelB + (unsigned)g.kp * g.vstride;
    g.U = (g.kp * (1 + g.nv) + kMmWaves - 1) / kMmWaves;
    return g;
}

}  // namespace

// V[b][t][0 .. 16 ntm) = vec(E_t) (npr values, then zeros), V[b][t][16 ntm .. 16 tt) = f_t (Rp values, then zeros)
__global__ __launch_bounds__(256) void mmw_vec_kernel(MstepArgs a, double* __restrict__ V, int r, int Rp, int ntm16, int tt16) {
    const int b = blockIdx.x;
    if (a.active && !a.active[b]) return;
    const int T = a.T, npr = r * (r + 1) / 2, npp = Rp * (Rp + 1) / 2;
    const int t0 = (int)blockIdx.y * 16;
    __shared__ double fs[16][32];
    const int tid = threadIdx.x;
    for (int e = tid; e < 16 * Rp; e += 256) {
        const int u = e / Rp, k = e % Rp, t = t0 + u;
        fs[u][k] = t < T ? a.fsm[((size_t)b * T + t) * Rp + k] : 0.0;
    }
    __syncthreads();
    for (int c = tid; c < tt16; c += 256) {
        int i = 0, j = 0, kind = 0;                      // 0: zero, 1: E_ij, 2: f_i
        if (c < npr) {
            while ((i + 1) * (i + 2) / 2 <= c) ++i;
            j = c - i * (i + 1) / 2;
            kind = 1;
        } else if (c >= ntm16 && c - ntm16 < Rp) {
            i = c - ntm16;
            kind = 2;
        }
        for (int u = 0; u < 16; ++u) {
            const int t = t0 + u;
            if (t >= T) break;
            double v = 0.0;
            if (kind == 1) v = fma(fs[u][i], fs[u][j], a.Psm[((size_t)b * T + t) * npp + c]);
            else if (kind == 2) v = fs[u][i];
            V[((size_t)b * T + t) * tt16 + c] = v;
        }
    }
}


// ---- the LIST form of D for panels with FEW missing cells --------------------------------------------------------------------
// The dense product spends 14 of its 16 tiles (r = 20) multiplying V by a mask that is 90 % zeros.  Here the mask is never an
// operand: a wave owns 16 series of the item's 128; per stage of 16 periods it takes the NaN pattern of its series as ballots
// (lane = period x 4 series: four LDS reads give sixteen 16-bit masks in SGPRs) and, series by series, walks the SET bits only:
// one row of V (vec(E_t), 1 680 bytes at r = 20) read from the LDS stage by the whole wave -- lane l holds columns 128 k + 2 l,
// + 1 of the row in `ds_read_b128` number k -- and added into that series' accumulators (2 NR doubles per lane and series: 128
// VGPRs at r = 20).  The accumulator index is static (the walk is unrolled over the 16 series), the loop over the bits scalar
// (s_ff1 / s_and), the next row's read in flight under the current row's adds.  Sxf / Sxx / n stay a dense product on the
// matrix pipe (A = the panel with NaN -> 0, B = f_t: NTF tiles), issued before the walk so that it runs under it.
// Cost: one LDS row read per MISSING cell (LDS-bound: 16 clocks per cell and CU at r = 20) instead of 14 matrix instructions
// per 64 cells: config 4 with 10 % missing 6.2 -> see DESIGN 8.9.  The time grows with the share of missing cells, the dense
// form's does not: mm_share_kernel samples the panel (64 periods x 8 replicates) and BOTH kernels are launched -- each reads the
// two counters and the one that is not chosen exits at once (no host round trip).
constexpr int kMlKP = 16, kMlSer = 128, kMlMaxU = 4, kMlWaves = 16;

typedef double mm_v2 __attribute__((ext_vector_type(2)));
using lds_cv2_ptr_mm = const __attribute__((address_space(3))) mm_v2*;
__device__ __forceinline__ mm_v2 lds_read128mm(unsigned a) { return *(lds_cv2_ptr_mm)(size_t)a; }

__device__ __forceinline__ bool list_route(const unsigned* share, unsigned thr_pct) {
    return (unsigned long long)share[0] * 100ull <= (unsigned long long)share[1] * (unsigned long long)thr_pct;
}

struct MlGeo {
    int ntm16, tt16;                                     // column of f_t in a row of V / OUT, row length (doubles)
    int nv;                                              // 1-KB DMAs per row of V
    unsigned vrowB, vstride, pstride, panelB, stageB;
    int U, nbuf;
};

// share[0] += NaN cells, share[1] += cells of the sample: block (x, y) = period x T / gridDim.x of replicate y B / gridDim.y
__global__ __launch_bounds__(256) void mm_share_kernel(const double* __restrict__ panel, int B, int T, int N, unsigned* __restrict__ share) {
    const int t = (int)(((long long)blockIdx.x * T) / gridDim.x), b = (int)(((long long)blockIdx.y * B) / gridDim.y);
    const double* row = panel + ((size_t)b * T + t) * N;
    unsigned c = 0;
    for (int i = threadIdx.x; i < N; i += 256) { const double x = row[i]; c += (x != x) ? 1u : 0u; }
    __shared__ unsigned red[4];
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&share[0], red[0] + red[1] + red[2] + red[3]);
        atomicAdd(&share[1], (unsigned)N);
    }
}

template <int NR, int NTF>
__global__ __launch_bounds__(64 * kMlWaves) void mstep_miss_list_kernel(MstepArgs a, const double* __restrict__ V, double* __restrict__ OUT,
                                                                       double* __restrict__ sxx, double* __restrict__ cnt, MlGeo g, int nsb,
                                                                       const unsigned* __restrict__ share, unsigned thr_pct) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (share && !list_route(share, thr_pct)) return;
    constexpr int KP = kMlKP;
    const int N = a.N, T = a.T, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k4 = lane >> 4, c16 = lane & 15;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_mm)(smem));
    const int nst = (T + KP - 1) / KP;
    const int tt16 = g.tt16;
    const int ND = KP * (1 + g.nv);
    const int U = g.U, NBUF = g.nbuf;
    const bool xmap = B >= 16;                                // (items as in mstep_miss_kernel: a replicate's blocks side by side on one L2)
    const int xcd = xmap ? (int)blockIdx.x & 7 : 0, slot = xmap ? (int)blockIdx.x >> 3 : (int)blockIdx.x;
    const int nslot = xmap ? (int)gridDim.x >> 3 : (int)gridDim.x, xstep = xmap ? 8 : 1;
    const int nrep_x = xmap ? (B - xcd + 7) / 8 : B;
    // the wave's DMAs of a stage (the same for every stage and item): LDS offset within the stage buffer | period << 16 | piece << 24
    // (piece 0 = the panel row, k = the k-th KB of the row of V) -- worked out once (a division per DMA and stage otherwise)
    unsigned cdU[kMlMaxU];
#pragma unroll
    for (int u = 0; u < kMlMaxU; ++u) {
        int d = wave + kMlWaves * u;
        d = d < ND ? d : ND - 1;                              // (a duplicate of the last DMA keeps the count equal)
        const int per = d / (1 + g.nv), pc = d % (1 + g.nv);
        const unsigned dst = pc == 0 ? (unsigned)per * g.pstride : g.panelB + (unsigned)per * g.vstride + 1024u * (unsigned)(pc - 1);
        cdU[u] = __builtin_amdgcn_readfirstlane(dst | ((unsigned)per << 16) | ((unsigned)pc << 24));
    }
    const size_t rowXB = (size_t)N * 8;
    // roles of the wave: the walk over series 8 wave .. + 7 of the item; of the dense part, periods 2 half .. + 1 (mod 4) of the
    // 16 series of tile dt (two waves share a tile: their partial sums meet in LDS at the end of the item)
    const int dt = wave >> 1, half = wave & 1;

    for (int e = tid; e < (int)((NBUF * g.stageB + 1024u * NR) / 8); e += 64 * kMlWaves) reinterpret_cast<double*>(smem)[e] = 0.0;
    __syncthreads();

    for (int q = slot; q < nrep_x * nsb; q += nslot) {
        const int b = xcd + xstep * (q / nsb), sb = q % nsb;
        if (a.active && !a.active[b]) continue;
        const int s0 = sb * kMlSer;
        const char* Xl = reinterpret_cast<const char*>(a.panel + (size_t)b * T * N) + (size_t)(s0 + 2 * lane) * 8;
        const char* Vl = reinterpret_cast<const char*>(V + (size_t)b * T * tt16) + 16 * lane;
        const bool okP = s0 + 2 * lane < N;
        auto issue_stage = [&](int st, int bsel) {
            const unsigned sbase = lds0 + (unsigned)bsel * g.stageB;
#pragma unroll
            for (int u = 0; u < kMlMaxU; ++u) {
                if (u < U) {
                    int t = st * KP + (int)((cdU[u] >> 16) & 255u);
                    t = t < T ? t : T - 1;
                    const unsigned dst = sbase + (cdU[u] & 0xffffu);
                    const unsigned pc = cdU[u] >> 24;
                    if (pc == 0) {
                        if (okP) dma16mm(Xl + (size_t)t * rowXB, dst);
                    } else {
                        const unsigned o = 1024u * (pc - 1u);
                        if (o + 16u * (unsigned)lane < g.vrowB) dma16mm(Vl + ((size_t)t * g.vrowB + o), dst);
                    }
                }
            }
        };
        mm_v2 acc[8][NR];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < NR; ++k) acc[j][k] = mm_v2{0.0, 0.0};
        mm_v4 accf[NTF];
#pragma unroll
        for (int x = 0; x < NTF; ++x) accf[x] = mm_v4{0.0, 0.0, 0.0, 0.0};
        double qs = 0.0, nc = 0.0;
        const int sw = s0 + 8 * wave;                         // the walk's first series
        const int sd = s0 + 16 * dt;                          // the dense tile's first series
        const bool ser_ok = sd + c16 < N;
        const unsigned a_off = (unsigned)(8 * half + k4) * g.pstride + (unsigned)(16 * dt + c16) * 8u;
        const unsigned b_off = g.panelB + (unsigned)(8 * half + k4) * g.vstride + (unsigned)(g.ntm16 + c16) * 8u;
        const unsigned m_off = (unsigned)c16 * g.pstride + (unsigned)(8 * wave + k4) * 8u;   // masks: lane = (period c16, series k4 + 4 q)

        for (int q0 = 0; q0 < NBUF - 1; ++q0)
            if (q0 < nst) issue_stage(q0, q0);
        int bsel = 0;
        for (int st = 0; st < nst; ++st) {
            {
                const int younger = (nst - 1 - st < NBUF - 2) ? nst - 1 - st : NBUF - 2;
                wait_vm_upto(younger * U);
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (st + NBUF - 1 < nst) issue_stage(st + NBUF - 1, bsel == 0 ? NBUF - 1 : bsel - 1);
            const unsigned stg = lds0 + (unsigned)bsel * g.stageB;
            // the stage's operands in flight together: the NaN patterns' cells, the dense part's two steps
            double mx[2], xr[2], bvv[2][NTF];
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) mx[qq] = lds_read64mm(stg + m_off + (unsigned)qq * 32u);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                xr[s] = lds_read64mm(stg + a_off + (unsigned)s * 4u * g.pstride);
#pragma unroll
                for (int x = 0; x < NTF; ++x) bvv[s][x] = lds_read64mm(stg + b_off + (unsigned)s * 4u * g.vstride + 128u * (unsigned)x);
            }
            // the NaN patterns of the wave's 8 series over the stage's 16 periods: bits 16 k4 + period of bal[q] = series 4 q + k4
            unsigned long long bal[2];
            {
                const bool per_ok = st * KP + c16 < T;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) bal[qq] = __ballot((int)per_ok & (int)(sw + 4 * qq + k4 < N) & (int)(mx[qq] != mx[qq]));
            }
            // Sxf, Sxx, n: the dense part on the matrix pipe
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bool ok = (bool)((int)ser_ok & (int)(st * KP + 8 * half + 4 * s + k4 < T) & (int)(xr[s] == xr[s]));
                const double xk = ok ? xr[s] : 0.0;
                qs = fma(xk, xk, qs);
                nc += ok ? 1.0 : 0.0;
#pragma unroll
                for (int x = 0; x < NTF; ++x) accf[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(xk, bvv[s][x], accf[x], 0, 0, 0);
            }
            // D: the rows of V at the missing cells, series by series (static accumulators), bit by bit (a scalar loop), the next
            // row's read in flight under the current row's adds; the other waves of the SIMD cover the rest of the latency
            const unsigned vb = stg + g.panelB + 16u * (unsigned)lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned m = (unsigned)(bal[j >> 2] >> (16 * (j & 3))) & 0xffffu;
                if (m) {
                    mm_v2 cur[NR];
                    unsigned ad = vb + (unsigned)__builtin_ctz(m) * g.vstride;
                    m &= m - 1u;
#pragma unroll
                    for (int k = 0; k < NR; ++k) cur[k] = lds_read128mm(ad + 1024u * (unsigned)k);
                    while (m) {
                        mm_v2 nx[NR];
                        ad = vb + (unsigned)__builtin_ctz(m) * g.vstride;
                        m &= m - 1u;
#pragma unroll
                        for (int k = 0; k < NR; ++k) nx[k] = lds_read128mm(ad + 1024u * (unsigned)k);
#pragma unroll
                        for (int k = 0; k < NR; ++k) { acc[j][k] += cur[k]; cur[k] = nx[k]; }
                    }
#pragma unroll
                    for (int k = 0; k < NR; ++k) acc[j][k] += cur[k];
                }
            }
            bsel = bsel == NBUF - 1 ? 0 : bsel + 1;
        }
        // D: lane l holds columns 128 k + 2 l, + 1 of series sw + j
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (sw + j < N) {
                double* orow = OUT + ((size_t)b * N + sw + j) * tt16;
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    const int col = 128 * k + 2 * lane;
                    if (col < g.ntm16) *reinterpret_cast<mm_v2*>(orow + col) = acc[j][k];
                }
            }
        }
        // the dense part: the odd wave's partial sums to the even wave through LDS (the stage buffers are free behind the barrier)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        double* xch = reinterpret_cast<double*>(smem) + (size_t)dt * (4 * NTF + 2) * 64 + lane;
        if (half == 1) {
#pragma unroll
            for (int x = 0; x < NTF; ++x)
#pragma unroll
                for (int v = 0; v < 4; ++v) xch[(4 * x + v) * 64] = accf[x][v];
            xch[4 * NTF * 64] = qs;
            xch[(4 * NTF + 1) * 64] = nc;
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int x = 0; x < NTF; ++x)
#pragma unroll
                for (int v = 0; v < 4; ++v) accf[x][v] += xch[(4 * x + v) * 64];
            qs += xch[4 * NTF * 64];
            nc += xch[(4 * NTF + 1) * 64];
            // Sxf: 16x16x4 D[(l / 16) + 4 v][l % 16] -> series k4 + 4 v, column c16 of the tile
            double* out = OUT + ((size_t)b * N + sd) * tt16 + g.ntm16 + c16;
#pragma unroll
            for (int x = 0; x < NTF; ++x) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = k4 + 4 * v;
                    if (sd + row < N) out[(size_t)row * tt16 + 16 * x] = accf[x][v];
                }
            }
            qs += __shfl_xor(qs, 16, 64); qs += __shfl_xor(qs, 32, 64);
            nc += __shfl_xor(nc, 16, 64); nc += __shfl_xor(nc, 32, 64);
            if (k4 == 0 && ser_ok) {
                sxx[(size_t)b * N + sd + c16] = qs;
                cnt[(size_t)b * N + sd + c16] = nc;
            }
        }
        __syncthreads();                                      // the exchange area is read before the next item's DMAs land on it
    }
}

// OUT[b][series][16 tt]: D (16 ntm columns) then Sxf (16 ntf columns); sxx, cnt [b][series]
template <int TPW, int KP, int NBUF>
__global__ __launch_bounds__(64 * kMmWaves) void mstep_miss_kernel(MstepArgs a, const double* __restrict__ V, double* __restrict__ OUT,
                                                                  double* __restrict__ sxx, double* __restrict__ cnt, MmGeo g, int nsb,
                                                                  const unsigned* __restrict__ share, unsigned thr_pct) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (share && list_route(share, thr_pct)) return;      // (few missing cells: mstep_miss_list_kernel has done this launch's work)
    const int N = a.N, T = a.T, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sgi = wave % g.sg, cgi = wave / g.sg;
    const int tile0 = cgi * g.tpw;
    const int ntile = (g.tt - tile0 < g.tpw) ? g.tt - tile0 : g.tpw;   // tiles of this wave
    const int k4 = lane >> 4, c16 = lane & 15;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_mm)(smem));
    const int nst = (T + KP - 1) / KP;
    const int tt16 = g.tt * 16;
    const int ND = KP * (1 + g.nv);
    const int U = g.U;                                        // DMAs per wave and stage (the same for every wave: counted waits)
    // items: XCD x (= blockIdx.x & 7 under the round-robin dispatch) owns the replicates b = x (mod 8); its workgroups take
    // (replicate, series block) pairs in order, so the blocks of a replicate run side by side on one L2
    // (small batches: plain round robin over all workgroups)
    const bool xmap = B >= 16;
    const int xcd = xmap ? (int)blockIdx.x & 7 : 0, slot = xmap ? (int)blockIdx.x >> 3 : (int)blockIdx.x;
    const int nslot = xmap ? (int)gridDim.x >> 3 : (int)gridDim.x, xstep = xmap ? 8 : 1;
    const int nrep_x = xmap ? (B - xcd + 7) / 8 : B;

    // no NaN bit patterns in LDS bytes no DMA writes (columns of a partial series block)
    for (int e = tid; e < (int)(NBUF * g.stageB / 8); e += 64 * kMmWaves) reinterpret_cast<double*>(smem)[e] = 0.0;
    __syncthreads();

    for (int q = slot; q < nrep_x * nsb; q += nslot) {
        const int b = xcd + xstep * (q / nsb), sb = q % nsb;
        if (a.active && !a.active[b]) continue;
        const int s0 = sb * g.ser;
        const char* Xb = reinterpret_cast<const char*>(a.panel + (size_t)b * T * N);
        const char* Vb = reinterpret_cast<const char*>(V + (size_t)b * T * tt16);
        auto issue_stage = [&](int st, int bsel) {
            const unsigned sbase = lds0 + (unsigned)bsel * g.stageB;
            for (int u = 0; u < U; ++u) {
                int d = wave + kMmWaves * u;
                d = d < ND ? d : ND - 1;                          // (a duplicate of the last DMA keeps the count equal)
                const int per = d / (1 + g.nv), piece = d % (1 + g.nv);
                int t = st * KP + per;
                t = t < T ? t : T - 1;
                if (piece == 0) {
                    const int ser = s0 + 2 * lane;
                    const unsigned dst = __builtin_amdgcn_readfirstlane(sbase + (unsigned)per * g.pstride);
                    if (2 * lane < g.ser && ser < N) dma16mm(Xb + ((size_t)t * N + ser) * 8, dst);
                } else {
                    const unsigned o = 1024u * (unsigned)(piece - 1) + 16u * (unsigned)lane;
                    const unsigned dst = __builtin_amdgcn_readfirstlane(sbase + g.panelB + (unsigned)per * g.vstride + 1024u * (unsigned)(piece - 1));
                    if (o < g.vrowB) dma16mm(Vb + (size_t)t * g.vrowB + o, dst);
                }
            }
        };
        mm_v4 acc[TPW];
#pragma unroll
        for (int x = 0; x < TPW; ++x) acc[x] = mm_v4{0.0, 0.0, 0.0, 0.0};
        double qs = 0.0, nc = 0.0;
        const bool ser_ok = s0 + 16 * sgi + c16 < N;
        const unsigned a_off = (unsigned)k4 * g.pstride + (unsigned)(16 * sgi + c16) * 8u;
        const unsigned b_off = g.panelB + (unsigned)k4 * g.vstride + (unsigned)(16 * tile0 + c16) * 8u;

        unsigned xoff[TPW];                                   // byte offset of tile slot x in a row of V
#pragma unroll
        for (int x = 0; x < TPW; ++x) xoff[x] = 128u * (unsigned)(x < ntile ? x : ntile - 1);

#pragma unroll
        for (int q0 = 0; q0 < NBUF - 1; ++q0)
            if (q0 < nst) issue_stage(q0, q0);                // NBUF - 1 stages ahead
        int bsel = 0;
        for (int st = 0; st < nst; ++st) {
            // this wave's DMAs of stage st have landed: at most those of the NBUF - 2 younger stages are still in flight
            {
                const int younger = (nst - 1 - st < NBUF - 2) ? nst - 1 - st : NBUF - 2;
                wait_vm_upto(younger * U);
            }
            __builtin_amdgcn_s_barrier();                     // ... everybody's have, and the buffer read one stage ago is free
            asm volatile("" ::: "memory");
            if (st + NBUF - 1 < nst) issue_stage(st + NBUF - 1, bsel == 0 ? NBUF - 1 : bsel - 1);
            const unsigned stg = lds0 + (unsigned)bsel * g.stageB;
            const int nm = g.ntm - tile0;                     // tiles x < nm of this wave belong to D (A = the mask), the others to Sxf (A = xz)
            double mk[KP / 4], xk[KP / 4];
#pragma unroll
            for (int s = 0; s < KP / 4; ++s) {
                const double xr = lds_read64mm(stg + a_off + (unsigned)s * 4u * g.pstride);
                const bool valid = ser_ok && (st * KP + 4 * s + k4 < T);
                const bool ok = xr == xr;
                mk[s] = (valid && !ok) ? 1.0 : 0.0;
                xk[s] = (valid && ok) ? xr : 0.0;
                qs = fma(xk[s], xk[s], qs);
                nc += (valid && ok) ? 1.0 : 0.0;
            }
            // every tile slot of the instantiation runs (slots past the wave's last tile repeat it and are not stored): no
            // branches, the B operands of a step in flight before its first MFMA
#pragma unroll
            for (int s = 0; s < KP / 4; ++s) {
                double bv[TPW];
#pragma unroll
                for (int x = 0; x < TPW; ++x) bv[x] = lds_read64mm(stg + b_off + (unsigned)s * 4u * g.vstride + xoff[x]);
#pragma unroll
                for (int x = 0; x < TPW; ++x)
                    acc[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(x < nm ? mk[s] : xk[s], bv[x], acc[x], 0, 0, 0);
            }
            bsel = bsel == NBUF - 1 ? 0 : bsel + 1;
        }
        // the item is complete: 16x16x4 D[(l / 16) + 4 v][l % 16] -> series k4 + 4 v of the group, column c16 of the tile
        double* out = OUT + ((size_t)b * N + s0 + 16 * sgi) * tt16 + 16 * tile0 + c16;
#pragma unroll
        for (int x = 0; x < TPW; ++x) {
            if (x < ntile) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = k4 + 4 * v;
                    if (s0 + 16 * sgi + row < N) out[(size_t)row * tt16 + 16 * x] = acc[x][v];
                }
            }
        }
        if (cgi == 0) {
            qs += __shfl_xor(qs, 16, 64); qs += __shfl_xor(qs, 32, 64);
            nc += __shfl_xor(nc, 16, 64); nc += __shfl_xor(nc, 32, 64);
            if (k4 == 0 && ser_ok) {
                sxx[(size_t)b * N + s0 + 16 * sgi + c16] = qs;
                cnt[(size_t)b * N + s0 + 16 * sgi + c16] = nc;
            }
        }
        __builtin_amdgcn_s_barrier();                         // every wave is done with the stage buffers before the next item's DMAs
        asm volatile("" ::: "memory");
    }
}

#ifdef DFM_DIAG   // (the LDS-column solve: diagnostics library only, DFM_MM_FINISH=1)
// thread = series: packed Sff_i = S11 - D_i in the thread's own LDS column, Cholesky in place, lam_i, R_i.  256 threads load
// the block's rows (one contiguous piece of OUT); the first ns of them solve (ns = series per block = the LDS column count)
__global__ __launch_bounds__(256) void mmw_finish_kernel(MstepArgs a, const double* __restrict__ OUT, const double* __restrict__ sxx,
                                                         const double* __restrict__ cnt, int r, int Rp, int ntm16, int tt16, int ns) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    if (a.active && !a.active[b]) return;
    const int tid = threadIdx.x;
    const int N = a.N, npr = r * (r + 1) / 2;
    const int nsi = ns + 1;                                    // (odd column stride: the transposing stores below spread over the banks)
    double* S = reinterpret_cast<double*>(smem);              // [npr + r][ns + 1]: packed Sff, then the right-hand side
    double* s11 = S + (size_t)(npr + r) * nsi;                // [npr] packed, symmetrised
    for (int v = tid; v < npr; v += 256) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= v) ++i;
        const int j = v - i * (i + 1) / 2;
        s11[v] = 0.5 * (a.S11[(size_t)b * Rp * Rp + i * Rp + j] + a.S11[(size_t)b * Rp * Rp + j * Rp + i]);
    }
    __syncthreads();
    const int s0 = (int)blockIdx.y * ns;
    const int nrow = (N - s0 < ns) ? N - s0 : ns;
    const double* src = OUT + ((size_t)b * N + s0) * tt16;
    const int ne = nrow * tt16;
#pragma unroll 8
    for (int e = tid; e < ne; e += 256) {
        const int row = e / tt16, c = e % tt16;
        const double v = src[e];
        if (c < npr) S[(size_t)c * nsi + row] = s11[c] - v;
        else if (c >= ntm16 && c - ntm16 < r) S[(size_t)(npr + c - ntm16) * nsi + row] = v;
    }
    for (int c = tid; c < npr + r; c += 256) S[(size_t)c * nsi + ns] = 0.0;   // the padding column (stride ns + 1): where idle threads work
    __syncthreads();
    // Cholesky and the two substitutions per series, in the series' LDS column, by ALL threads of the block: thread (si = tid % ns,
    // w = tid / ns) works on the rows i = w (mod 256 / ns) of series si in the right-looking (outer-product) form -- the updates of a
    // step are independent, three barriers per column.  (One thread per series walking the row-by-row form was ONE chain of dependent
    // LDS round trips with one wave per CU busy: 1.94 ms per config-4 EM iteration, a third of the dense product's time.)
    const int si = tid % ns, w = tid / ns, nw = 256 / ns;
    const int sic = si < nrow ? si : ns;                       // (series past N of a partial block: the zeroed padding column -- never a live
                                                               // series' column: its owner's read-modify-writes must not be repeated by an alias)
    double* L = S + sic;
    double* y = S + (size_t)npr * nsi + sic;
#define LL(i, j) L[((i) * ((i) + 1) / 2 + (j)) * nsi]
    // a series with fewer than r + 1 observed cells, or whose normal matrix is not positive definite, keeps its parameters
    // (mstep_obs_kernel's rule; an all-missing series -- e.g. the one capi.hip appends to an odd N -- has S = 0, cnt = 0)
    bool ok = true;
    for (int j = 0; j < r; ++j) {
        const double d = LL(j, j);                             // (every worker of the series reads it before worker 0 overwrites it)
        ok = ok && (d > 0.0);
        const double ljj = sqrt(d > 0.0 ? d : 1.0);
        const double inv = 1.0 / ljj;
        __syncthreads();
        if (w == 0) LL(j, j) = ljj;
        for (int i = j + 1 + w; i < r; i += nw) LL(i, j) *= inv;   // column j below the diagonal
        __syncthreads();
        for (int i = j + 1 + w; i < r; i += nw) {                // trailing update: S[i][k] -= L[i][j] L[k][j], j < k <= i
            const double lij = LL(i, j);
            double* row = &LL(i, j + 1);
            const double* colj = &LL(j + 1, j);
            int kj = (j + 1) * (j + 2) / 2 + j;                  // packed index of (k, j), advancing by k + 1
            (void)colj;
            for (int k = j + 1; k <= i; k += 4) {                // four independent elements at a time
                double av[4], bv[4];
                int kk = kj;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool in = k + u <= i;
                    av[u] = row[(in ? k + u - (j + 1) : 0) * nsi];
                    bv[u] = L[(in ? kk : kj) * nsi];
                    kk += k + u + 1;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(av[u]), "+v"(bv[u]) : : "memory");
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k + u <= i) row[(k + u - (j + 1)) * nsi] = fma(-lij, bv[u], av[u]);
                kj = kk;
            }
        }
        __syncthreads();                                       // (the next column's diagonal element is complete)
    }
    double yy = 0.0;
    for (int j = 0; j < r; ++j) {                             // L y = Sxf, column by column
        const double yj = y[j * nsi] / LL(j, j);
        yy = fma(yj, yj, yy);                                 // lam' Sff lam = lam' Sxf = y'y
        __syncthreads();
        if (w == 0) y[j * nsi] = yj;
        for (int i = j + 1 + w; i < r; i += nw) y[i * nsi] = fma(-LL(i, j), yj, y[i * nsi]);
        __syncthreads();
    }
    for (int j = r - 1; j >= 0; --j) {                        // L' lam = y, column by column from the last
        const double xj = y[j * nsi] / LL(j, j);
        __syncthreads();
        if (w == 0) y[j * nsi] = xj;
        for (int i = w; i < j; i += nw) y[i * nsi] = fma(-LL(j, i), xj, y[i * nsi]);
        __syncthreads();
    }
#undef LL
    if (w != 0 || si >= nrow) return;
    const int col = s0 + si;
    const double nobs_i = cnt[(size_t)b * N + col];
    // a series without a single observed cell keeps its parameters; so does one with fewer cells than the caller's minimum (the
    // observed-factor joint regression asks for r_o + r_u + 1; the plain EM for 1: sum E[f f'] includes P_t and is positive definite)
    if (!ok || nobs_i < (double)(a.min_cells > 1 ? a.min_cells : 1)) return;
    a.R_out[(size_t)b * N + col] = (sxx[(size_t)b * N + col] - yy) / nobs_i;
    double* lo = a.Lam_out + ((size_t)b * N + col) * a.lam_stride;
    for (int k = 0; k < Rp; ++k) lo[k] = k < r ? y[k * nsi] : 0.0;
}

#endif

// mmw_solve_kernel -- the same solve with a LANE per ROW: the r lanes of a group hold the rows of one series' normal matrix
// Sff_i = S11 - D_i (row i in the registers of lane i, static indices throughout), 64 / r series per wave.  Symmetric elimination
// without pivoting (LDL' by rows: for pivot j every later row takes its multiple of row j, broadcast by `ds_bpermute`) leaves row i of
// the upper factor in lane i; the back-substitution needs exactly that row and the solved components broadcast one by one -- no
// transpose, no LDS matrix, no barrier (mmw_finish_kernel: 140 workgroup barriers around dependent LDS read-modify-writes per block
// of 32 series: 1.02 ms per config-4 iteration).  The wave's rows of OUT pass through LDS once (coalesced in, gathered out).
template <int RMAX>
__global__ __launch_bounds__(256, 4) void mmw_solve_kernel(MstepArgs a, const double* __restrict__ OUT, const double* __restrict__ sxx,
                                                        const double* __restrict__ cnt, int r, int Rp, int ntm16, int tt16) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    if (a.active && !a.active[b]) return;
    const int N = a.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int spw = 64 / r;                                   // series per wave
    const int g = lane / r, i = lane - g * r;                 // group of the lane, its row
    const bool lane_ok = g < spw;
    const int base = g * r;
    const int s_first = ((int)blockIdx.y * 4 + wave) * spw;   // the wave's first series
    double* s11 = reinterpret_cast<double*>(smem);             // [Rp][Rp]: the replicate's S11 (every lane gathers 2 r entries of it)
    for (int e = threadIdx.x; e < Rp * Rp; e += 256) s11[e] = a.S11[(size_t)b * Rp * Rp + e];
    __syncthreads();
    if (s_first >= N) return;
    const int ns = (N - s_first < spw) ? N - s_first : spw;
    double* rows = s11 + Rp * Rp + (size_t)wave * spw * tt16;
    {
        const double2* src = reinterpret_cast<const double2*>(OUT + ((size_t)b * N + s_first) * tt16);
        double2* dst = reinterpret_cast<double2*>(rows);
        const int n2 = ns * tt16 / 2;
        for (int e = lane; e < n2; e += 64) dst[e] = src[e];
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the wave's own LDS stores are visible to its loads)
    const bool live = lane_ok && g < ns;
    const int gs = live ? g : 0;                               // (idle lanes work on a copy of group 0's rows; they never store)
    const int ii = i < r ? i : 0;
    const double* row = rows + (size_t)gs * tt16;
    double A[RMAX];
#pragma unroll
    for (int k = 0; k < RMAX; ++k) {
        A[k] = 0.0;
        if (k < r) {
            const int hi = ii > k ? ii : k, lo = ii > k ? k : ii;
            A[k] = 0.5 * (s11[ii * Rp + k] + s11[k * Rp + ii]) - row[hi * (hi + 1) / 2 + lo];
        }
    }
    const double sxf = row[ntm16 + ii];
    double bb = sxf, rdiag = 1.0;                              // rdiag: 1 / U[i][i], kept by lane i at its own pivot step
    bool ok = true;
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
        if (j < r) {                                           // (wave-uniform)
            const int src = base + j;
            const double d = __shfl(A[j], src, 64);            // the pivot
            ok = ok && (d > 0.0);
            const double rinv = 1.0 / (d > 0.0 ? d : 1.0);
            if (ii == j) rdiag = rinv;
            const double m = (ii > j) ? A[j] * rinv : 0.0;
            const double bj = __shfl(bb, src, 64);
            bb = fma(-m, bj, bb);
#pragma unroll
            for (int k = j + 1; k < RMAX; ++k) {
                if (k < r) {
                    const double pk = __shfl(A[k], src, 64);
                    A[k] = fma(-m, pk, A[k]);
                }
            }
        }
    }
    // U x = y from the last row up: lane j holds U[j][j .. r); x_j from lane j, every earlier row takes its term
    double lam = 0.0;
#pragma unroll
    for (int j = RMAX - 1; j >= 0; --j) {
        if (j < r) {
            const double xj = __shfl(bb * rdiag, base + j, 64);
            if (ii == j) lam = xj;
            if (ii < j) bb = fma(-A[j], xj, bb);
        }
    }
    // lam' Sxf over the group
    const double part = lam * sxf;
    double yy = 0.0;
#pragma unroll
    for (int k = 0; k < RMAX; ++k)
        if (k < r) yy += __shfl(part, base + k, 64);
    if (!live) return;
    const int col = s_first + g;
    const double nobs_i = cnt[(size_t)b * N + col];
    // a series without a single observed cell keeps its parameters; so does one with fewer cells than the caller's minimum, or whose
    // normal matrix is not positive definite (mmw_finish_kernel's rules)
    if (!ok || nobs_i < (double)(a.min_cells > 1 ? a.min_cells : 1)) return;
    double* lo = a.Lam_out + ((size_t)b * N + col) * a.lam_stride;
    lo[i] = lam;
    if (i == 0) {
        a.R_out[(size_t)b * N + col] = (sxx[(size_t)b * N + col] - yy) / nobs_i;
        for (int k = r; k < Rp; ++k) lo[k] = 0.0;
    }
}

// Rp = 8 | 16 | 32 (r <= Rp the caller's factor count), even N (16-byte aligned series pairs)
bool mstep_miss_supported(int Rpad, int r, int N) {
    if (Rpad != 8 && Rpad != 16 && Rpad != 32) return false;
    if (r < 1 || r > Rpad) return false;
    return (N & 1) == 0 && N >= 2;
}
// V [B][T][16 tt] | OUT [B][N][16 tt] | sxx [B][N] | cnt [B][N]
size_t mstep_miss_workspace(int B, int T, int N, int Rpad, int r) {
    const MmGeo g = mm_geo(Rpad, r);
    return ((size_t)B * T * g.tt * 16 + (size_t)B * N * g.tt * 16 + 2 * (size_t)B * N) * sizeof(double) + 512;   // (+ the two counters of mm_share_kernel)
}

namespace {
template <int TPW, int KP, int NBUF>
hipError_t launch_mm(const MstepArgs& a, const double* V, double* OUT, double* sxx, double* cnt, const MmGeo& g, int G, hipStream_t s,
                     const unsigned* share, unsigned thr) {
    static LdsOptIn attr_done;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mstep_miss_kernel<TPW, KP, NBUF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int nsb = (a.N + g.ser - 1) / g.ser;
    hipLaunchKernelGGL((mstep_miss_kernel<TPW, KP, NBUF>), dim3((unsigned)G), dim3(64 * kMmWaves), (size_t)NBUF * g.stageB, s, a, V, OUT,
                       sxx, cnt, g, nsb, share, thr);
    return hipGetLastError();
}
// tile slots per wave: the next instantiation at or above the geometry's (at most 3 idle slots)
template <int KP, int NBUF>
hipError_t launch_mm_slots(const MstepArgs& a, const double* V, double* OUT, double* sxx, double* cnt, const MmGeo& g, int G, hipStream_t s,
                           const unsigned* share, unsigned thr) {
    if (g.tpw <= 4) return launch_mm<4, KP, NBUF>(a, V, OUT, sxx, cnt, g, G, s, share, thr);              // Rp = 8
    if (g.tpw <= 7) return launch_mm<7, KP, NBUF>(a, V, OUT, sxx, cnt, g, G, s, share, thr);
    if (g.tpw <= 10) return launch_mm<10, KP, NBUF>(a, V, OUT, sxx, cnt, g, G, s, share, thr);            // r = 16
    if (g.tpw <= 13) return launch_mm<13, KP, NBUF>(a, V, OUT, sxx, cnt, g, G, s, share, thr);
    if (g.tpw <= 16) return launch_mm<16, KP, NBUF>(a, V, OUT, sxx, cnt, g, G, s, share, thr);            // r = 20 (config 4)
    if (g.tpw <= 18) return launch_mm<18, KP, NBUF>(a, V, OUT, sxx, cnt, g, G, s, share, thr);            // r = 32: two column groups
    if (g.tpw <= 21) return launch_mm<21, KP, NBUF>(a, V, OUT, sxx, cnt, g, G, s, share, thr);            // r = 24
    return hipErrorInvalidValue;
}

// the list form's geometry over the SAME V / OUT layout as the dense product (mm_geo): false = not supported (NR > 2: the 16 series'
// accumulators would not fit the registers; a stage of 16 periods that does not fit the LDS twice)
bool ml_geo(const MmGeo& d, MlGeo* g, int* NR) {
    *NR = (d.npr + 127) / 128;
    if (*NR > 2) return false;                              // (r <= 22)
    g->ntm16 = d.ntm * 16; g->tt16 = d.tt * 16;
    g->vrowB = d.vrowB; g->nv = d.nv; g->vstride = d.vstride;
    g->pstride = (unsigned)kMlSer * 8u + 128u;
    g->panelB = (unsigned)kMlKP * g->pstride;
    g->stageB = g->panelB + (unsigned)kMlKP * g->vstride;
    g->U = (kMlKP * (1 + g->nv) + kMlWaves - 1) / kMlWaves;
    const unsigned slack = 1024u * (unsigned)*NR;          // (the walk reads NR KB from a row's start whatever the row's length)
    g->nbuf = 3u * g->stageB + slack <= kMmLdsMax + 2048u ? 3 : 2;
    if (g->U > kMlMaxU) return false;
    if (g->nbuf == 3 && g->U > 6) g->nbuf = 2;            // (wait_vm_upto's cases)
    return (unsigned)g->nbuf * g->stageB + slack <= kMmLdsMax + 2048u;
}
template <int NR, int NTF>
hipError_t launch_ml(const MstepArgs& a, const double* V, double* OUT, double* sxx, double* cnt, const MlGeo& g, int G, hipStream_t s,
                     const unsigned* share, unsigned thr) {
    static LdsOptIn attr_done;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mstep_miss_list_kernel<NR, NTF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int nsb = (a.N + kMlSer - 1) / kMlSer;
    hipLaunchKernelGGL((mstep_miss_list_kernel<NR, NTF>), dim3((unsigned)G), dim3(64 * kMlWaves), (size_t)g.nbuf * g.stageB + 1024u * NR, s,
                       a, V, OUT, sxx, cnt, g, nsb, share, thr);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_mmw_vec(const double* fsm, const double* Psm, const int* active, int B, int T, int r, int Rp, int ntm16, int tt16,
                          double* V, hipStream_t s) {
    MstepArgs a{};
    a.B = B; a.T = T; a.fsm = fsm; a.Psm = Psm; a.active = active;
    hipLaunchKernelGGL(mmw_vec_kernel, dim3(B, (T + 15) / 16), dim3(256), 0, s, a, V, r, Rp, ntm16, tt16);
    return hipGetLastError();
}

hipError_t launch_mstep_miss(const MstepArgs& a, double* ws, int Rpad, int r, int num_cu, hipStream_t s) {
    note_kernel("mstep_miss_kernel");
    static const int kp_want = [] { const char* v = diag_env("DFM_MM_KP"); return v ? atoi(v) : 0; }();   // diagnostics: 8 | 16 | 32
    static const int nbuf_want = [] { const char* v = diag_env("DFM_MM_NBUF"); return v ? atoi(v) : 0; }();  // diagnostics: 4 (with DFM_MM_KP=8)
    const MmGeo g = mm_geo(Rpad, r, kp_want, nbuf_want);
    const int tt16 = g.tt * 16, ntm16 = g.ntm * 16;
    double* V = ws;
    double* OUT = V + (size_t)a.B * a.T * tt16;
    double* sxx = OUT + (size_t)a.B * a.N * tt16;
    double* cnt = sxx + (size_t)a.B * a.N;
    hipLaunchKernelGGL(mmw_vec_kernel, dim3(a.B, (a.T + 15) / 16), dim3(256), 0, s, a, V, r, Rpad, ntm16, tt16);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // persistent workgroups: as many per CU as their stage buffers allow (the per-wave VGPR budget at 8 waves per workgroup:
    // 256 / 128 / 85 for 1 / 2 / 3 workgroups -- the narrow shapes need 95 or fewer).  DFM_MM_WGS overrides (diagnostics).
    static const int wgs_env = [] { const char* v = diag_env("DFM_MM_WGS"); return v ? atoi(v) : 0; }();
    int wgs = 1;
    if (g.tpw <= 4) {
        const size_t lds_wg = (size_t)g.nbuf * g.stageB + 1024;
        wgs = (int)((160 * 1024) / lds_wg);
        if (wgs > 2) wgs = 2;
        if (wgs < 1) wgs = 1;
    }
    if (wgs_env > 0) wgs = wgs_env;
    int G = (num_cu > 0 ? num_cu : 256) * wgs;
    G = (G / 8) * 8;
    if (G < 8) G = 8;
    if (g.nbuf >= 3 && g.U > 6) return hipErrorInvalidValue;
    // few missing cells: the list form (mstep_miss_list_kernel).  DFM_MSTEP_LIST: 0 = never, 1 = by the sampled share of missing cells
    // (default; both kernels are launched, one exits), 2 = always where its geometry exists
    static const int list_mode = [] { const char* v = route_env("DFM_MSTEP_LIST"); return v ? atoi(v) : 1; }();
    static const int list_pct = [] { const char* v = diag_env("DFM_MM_LIST_PCT"); return v ? atoi(v) : 40; }();
    const unsigned* share = nullptr;
    const unsigned thr = (unsigned)list_pct;
    MlGeo lg; int NR = 0;
    bool list_done = false;
    if (list_mode != 0 && ml_geo(g, &lg, &NR)) {
        unsigned* sh = reinterpret_cast<unsigned*>(cnt + (size_t)a.B * a.N);
        if (list_mode == 1) {
            e = hipMemsetAsync(sh, 0, 2 * sizeof(unsigned), s);
            if (e != hipSuccess) return e;
            const int np = a.T < 64 ? a.T : 64, nr = a.B < 8 ? a.B : 8;
            hipLaunchKernelGGL(mm_share_kernel, dim3(np, nr), dim3(256), 0, s, a.panel, a.B, a.T, a.N, sh);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            share = sh;
        }
        int Gl = ((num_cu > 0 ? num_cu : 256) / 8) * 8;
        if (Gl < 8) Gl = 8;
        const int ntf = g.ntf;
        if (NR == 1 && ntf == 1) e = launch_ml<1, 1>(a, V, OUT, sxx, cnt, lg, Gl, s, share, thr);
        else if (NR == 2 && ntf == 1) e = launch_ml<2, 1>(a, V, OUT, sxx, cnt, lg, Gl, s, share, thr);
        else if (NR == 2 && ntf == 2) e = launch_ml<2, 2>(a, V, OUT, sxx, cnt, lg, Gl, s, share, thr);
        else e = hipErrorInvalidValue;
        if (e != hipSuccess) return e;
        list_done = list_mode != 1;
    }
    if (list_done) { /* DFM_MSTEP_LIST=2: the list form has done the step */ }
    else if (g.kp == 32) e = launch_mm_slots<32, 2>(a, V, OUT, sxx, cnt, g, G, s, share, thr);
    else if (g.kp == 16) e = launch_mm_slots<16, 2>(a, V, OUT, sxx, cnt, g, G, s, share, thr);
    else if (g.nbuf == 4) e = launch_mm_slots<8, 4>(a, V, OUT, sxx, cnt, g, G, s, share, thr);
    else e = launch_mm_slots<8, 3>(a, V, OUT, sxx, cnt, g, G, s, share, thr);
    if (e != hipSuccess) return e;
#ifdef DFM_DIAG
    const char* fv = diag_env("DFM_MM_FINISH");                // (read per launch: the A/B test switches it inside one process)
    const int fin_old = fv ? atoi(fv) : 0;
    if (fin_old) {
        const int npr = r * (r + 1) / 2;
        // series per block: 64 where two blocks' packed matrices share a CU's LDS, else 32 (r = 20: 60 KB per block; one block per CU with
        // 64 series was 1.5 ms of a config-4 iteration's loadings step against 1.0 ms)
        const int nthr = ((size_t)(npr + r) * 65 + npr) * sizeof(double) <= 72 * 1024 ? 64 : 32;
        const size_t lds = ((size_t)(npr + r) * (nthr + 1) + npr) * sizeof(double);
        static LdsOptIn fin_done;
        if (!fin_done) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mmw_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            fin_done = true;
        }
        hipLaunchKernelGGL(mmw_finish_kernel, dim3(a.B, (a.N + nthr - 1) / nthr), dim3(256), lds, s, a, (const double*)OUT, (const double*)sxx,
                           (const double*)cnt, r, Rpad, ntm16, tt16, nthr);
        return hipGetLastError();
    }
#endif
    // the per-series solves: a lane per row (mmw_solve_kernel); DFM_MM_FINISH=1 (diagnostics) keeps the LDS-column kernel
    {
        const int spw = 64 / r;
        const size_t lds = ((size_t)4 * spw * tt16 + (size_t)Rpad * Rpad) * sizeof(double);
        const dim3 grid(a.B, (a.N + 4 * spw - 1) / (4 * spw));
        // (the 16-wide instantiation sends the register allocator into 4 000 spills; Rp = 16 takes the 32-wide one with 16 live columns)
        if (Rpad == 8) hipLaunchKernelGGL(mmw_solve_kernel<8>, grid, dim3(256), lds, s, a, (const double*)OUT, (const double*)sxx, (const double*)cnt, r, Rpad, ntm16, tt16);
        else hipLaunchKernelGGL(mmw_solve_kernel<32>, grid, dim3(256), lds, s, a, (const double*)OUT, (const double*)sxx, (const double*)cnt, r, Rpad, ntm16, tt16);
        return hipGetLastError();
    }
}

}  // namespace dfm
