// collapse_miss.hip -- the collapse for panels WITH missing cells (NaN), Rp = 8, on the LDS-DMA ring and the fp64 matrix pipe.
//
// For replicate b and period t, over the OBSERVED cells (Jungbacker & Koopman 2015):
//     b_t = sum_i lam_i x_it / R_i     s_t = sum_i x_it^2 / R_i     n_t = #observed     ld_t = sum_i log R_i
//     C_t = sum_i lam_i lam_i' / R_i   (packed lower triangle; only for periods with a missing cell)
// collapse_kernel (collapse.hip) streams the panel through registers with 16-byte loads (1.28 TB/s at the C2 shape with
// 10 % of the cells missing) and builds C_t by a wave-uniform loop over the missing series of the period -- data-dependent
// trip counts, LDS table reads, 0.65 ms per 1024 replicates.  Here the panel takes the path of the balanced collapse
// (collapse_mfma.hip / pass_fused.hip): `global_load_lds_dwordx4` into a ring of 8 period slots per wave, 4 periods x 8
// series x 8 factors per `v_mfma_f64_4x4x4`, with the missing cells handled from the landed rows:
//   * NaN operands are replaced by 0 before the MFMA (b_t, s_t over the observed cells); every lane keeps a bit mask of
//     the NaN among the A operands it read (its period, its 25 series);
//   * C_t = C_full - (Gram matrix of the MISSING series of the period) -- or the Gram matrix of the observed ones when
//     those are fewer: the series indices are compacted into an LDS list with ballots / prefix popcounts, and the Gram
//     matrix of the list is the matrix-pipe contraction of the balanced pass's Gram (pass_fused.hip gram_mfma8) with its
//     operands GATHERED from LDS tables of W = lam / R and R: 2 MFMAs per 8 listed series (3 steps at 10 % missing instead
//     of the 25 of the full row; a first version that masked the full-row Gram cost 0.47 ms of the 0.80: the 50 MFMAs and
//     ~150 selects / DPP moves per period did not overlap).  The partner's raw loading (other factor group) comes by one
//     DPP row shift;
//   * n_t from the ballots, s_t and ld_t = ldfull - sum over the missing cells of log R by ONE 8-value transpose-reduce per
//     row block (wave_allsum through ds_bpermute cost 0.11 ms).
// Periods without a missing cell skip all of it (wave-uniform branch): the kernel then runs at the balanced collapse's rate.
// TABLE MODE (round 6; the route of the time-chunked recursion, recursion_chunk.hip): no C_t here at all.  The per-period list work
// above cost as much as the stream (0.39 ms against 0.226 at the C2 shape, three rewrites in round 5 did not move it: two LDS round
// trips and a gathered contraction per period on the whole wave).  The kernel writes one 112-byte row per period -- b_t, s_t,
// n_t log 2 pi + sum log R, and the period's NaN bit mask as four ballots -- and, when its four waves have streamed their segments, the
// workgroup forms C_t of the replicate's periods with one LANE per period from the masks, straight into the pass's chunk-major
// observation table (dfm_ctbuild.h: the table's stores go out while the CU's other workgroup streams its panel).
// One workgroup per replicate, one wave per period segment; 2 workgroups per CU (rings 53 KB + tables 14 KB + lists).
// The reference's analogue is the per-period complete-case regression of x_t on Lambda (dfm_functions.ipynb:271-286 called
// from :364), which also forms the normal equations Lambda_t' Lambda_t and Lambda_t' x_t over the observed series.
#include <stdlib.h>
#include <string.h>

#include "dfm_ctbuild.h"
#include "dfm_gram.h"
#include "dfm_grid.h"
#include "dfm_kernels.h"

namespace dfm {

namespace {

using lds_char_ptr_x = __attribute__((address_space(3))) char*;

__device__ __forceinline__ void dma16x(const void* gsrc, unsigned lds_dst) {
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
template <int K>
__device__ __forceinline__ void wait_vmx() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(K) : "memory");
}
__device__ __forceinline__ void wait_lgkmx() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__host__ __device__ inline unsigned miss_slot_bytes(int N) {
    unsigned sb = (unsigned)N * 8u;
    while ((sb & 255u) != 64u && (sb & 255u) != 192u) sb += 16u;
    return sb;
}

// value of lane l + 4 within its row of 16 lanes (DPP row_shl:4) -- the (g, h = 1, q) partner of a lane (g, h = 0, q)
__device__ __forceinline__ double row_shl4(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x104, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x104, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// value held by lane (lane & ~3) + J of every quad (DPP quad_perm [J, J, J, J])
template <int J>
__device__ __forceinline__ unsigned quad_bcast(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, J * 0x55, 0xF, 0xF, true);
}

}  // namespace

// wave-wide sum on DPP / swizzle / permlane moves (no ds_bpermute round trips)
__device__ __forceinline__ double wave_allsum_x(double v) {
    v += xor_lane<1>(v);
    v += xor_lane<2>(v);
    v += xor_lane<4>(v);
    v += xor_lane<8>(v);
    v += xor_lane<16>(v);
    v += xor_lane<32>(v);
    return v;
}

// TAB: table mode (a.obs_chunk: one row per period for recursion_chunk.hip) as its own instantiation -- the kernel is bound by what
// it issues, and the per-period arrays' stores and tests are not compiled into it
template <int STEPS, int NDR, bool TAB>
__global__ __launch_bounds__(256, 2) void collapse_miss_kernel(CollapseArgs a, unsigned SB, int abl) {
    constexpr int R = 8, NP = 36;
    constexpr int NB = 2, NS = 4 * NB;
    constexpr int CS = 8;
    constexpr int NQ = NDR;
    constexpr int WPR = 4;                                        // one workgroup per replicate, one period segment per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int N = a.N, T = a.T;
    const unsigned rowB = (unsigned)N * 8u;
    const int K = lane >> 4, blk = (lane >> 2) & 3, q = lane & 3;
    const int g = blk >> 1, h = blk & 1;
    int tq = (T + WPR - 1) / WPR;
    {
        unsigned gg = rowB & 127u;
        gg = gg == 0 ? 128u : (gg & (~gg + 1u));
        const int m = (int)(128u / gg);
        tq = ((tq + m - 1) / m) * m;                             // segments start on 128-byte boundaries
    }
    const int ta = (wave * tq < T) ? wave * tq : T;
    const int tb = (ta + tq < T) ? ta + tq : T;
    const int nrows = tb - ta;
    const int nblk = (nrows + 3) / 4;
    const unsigned ringB = NS * SB;
    // LDS: 4 rings | Wt [N + 1][8] (row N = 0) | Rt [N + 1] | per wave: index list [N8 + 8] ints
    const int N8 = (N + 7) & ~7;
    const char* ring = smem + (size_t)wave * ringB;
    double* Wt = reinterpret_cast<double*>(smem + 4 * ringB);
    double* Rt = Wt + (size_t)(N + 1) * R;
    int* idx = reinterpret_cast<int*>(Rt + (N + 2)) + (size_t)wave * (N8 + 8);
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_x)(smem)) + (unsigned)wave * ringB;
    const unsigned lane16 = 16u * lane;
    bool pact[NDR];
#pragma unroll
    for (int p = 0; p < NDR; ++p) pact[p] = lane16 + 1024u * p < rowB;
    const unsigned lane_off = (unsigned)q * SB + (unsigned)(4 * g + K) * 8u;
    const bool tail_clamp = (STEPS - 1) * CS + 4 * g + K >= N;
    const unsigned last_off = tail_clamp ? (unsigned)q * SB + (unsigned)(N - 1) * 8u : lane_off + (unsigned)(STEPS - 1) * (CS * 8u);
    const char* seg = reinterpret_cast<const char*>(a.panel + ((size_t)b * T + ta) * N);
    // table mode (a.obs_chunk = RecursionArgs::chunk_rows instead of bcol .. ldrow / C_t): one 112-byte row per period,
    // doubles 0..7 b_t, 8 s_t, 9 n_t log 2 pi + log det R_t, 10..13 the NaN mask: bit l of word 2 jq + e = series 2 l + 128 jq + e is missing
    constexpr bool obs = TAB;
    auto obs_row = [&](int t) { return a.obs_chunk + ((size_t)b * T + t) * 14; };

    auto issue_block = [&](int k, int bslot) {                   // 4 rows x NDR DMAs, always (rows past the segment: its last row)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            int ri = 4 * k + rr;
            ri = ri < nrows ? ri : nrows - 1;
            const char* src = seg + (size_t)ri * rowB + lane16;
            const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(bslot * 4 + rr) * SB);
#pragma unroll
            for (int p = 0; p < NDR; ++p) {
                if (pact[p]) dma16x(src + 1024 * p, dst + 1024u * p);
            }
        }
    };
    if (nrows > 0) {
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (u < nblk) issue_block(u, u);
    }

    // ---- tables of the replicate in LDS: W = lam / R, R (lam = W R); row N is a zero row (padding of the index lists) ------
    const int lw = a.lam_w > 0 ? a.lam_w : R;                   // (narrow loadings: r <= 4 on the 8-wide state)
    const double* __restrict__ Lg = a.Lam + (size_t)b * N * lw;
    const double* __restrict__ Rg = a.Rv + (size_t)b * N;
    for (int e = threadIdx.x; e < (N + 1) * R; e += 256) {
        const int c = e >> 3, f = e & 7;
        Wt[e] = (c < N && f < lw) ? Lg[c * lw + f] / Rg[c] : 0.0;
    }
    for (int c = threadIdx.x; c <= N; c += 256) Rt[c] = c < N ? Rg[c] : 1.0;
    double rown[NQ][2], lr[NQ][2];
#pragma unroll
    for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 2 * lane + 128 * jq + e;
            const double x = c < N ? Rg[c] : 1.0;
            lr[jq][e] = log(x);
            rown[jq][e] = c < N ? 1.0 / x : 0.0;
        }
    double ldfull = 0.0;
#pragma unroll
    for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
        for (int e = 0; e < 2; ++e) ldfull += lr[jq][e];
    ldfull = wave_allsum_x(ldfull);
    __syncthreads();                                             // tables complete (the first ring blocks have landed too)
    double Bw[STEPS];                                            // B operands of the stream: W[c][4 h + q], c = 8 s + 4 g + K
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = s * CS + 4 * g + K;
        Bw[s] = Wt[(size_t)(c < N ? c : N) * R + 4 * h + q];
    }

    // Gram matrix sum_c lam_c lam_c' / R_c over the series listed in idx[0 .. 8 nst) (padding entries = N: zero row):
    // lane (K, g = 0, h, q): E0 = C[4 h + K][4 h + q];  h = 0: E1 = C[4 + K][q]  (the 36 packed entries of the lower triangle)
    auto list_gram = [&](int nst, double& E0, double& E1) {
        E0 = 0.0; E1 = 0.0;
        for (int s = 0; s < nst; ++s) {
            const int c = idx[s * CS + 4 * g + K];
            const double w = Wt[(size_t)c * R + 4 * h + q];
            const double own = w * Rt[c];                        // raw loading lam_c,4h+q
            const double part = row_shl4(own);                   // h = 0 lanes: lam_c,4+q of the (h = 1) partner
            E0 = __builtin_amdgcn_mfma_f64_4x4x4f64(own, w, E0, 0, 0, 0);
            E1 = __builtin_amdgcn_mfma_f64_4x4x4f64(h == 0 ? part : own, w, E1, 0, 0, 0);
        }
        E0 += xor_lane<8>(E0);                                   // fold the two series groups
        E1 += xor_lane<8>(E1);
    };
    // the full-row Gram matrix (identity list), kept per lane for the complement form
    for (int e = lane; e < N8 + 8; e += 64) idx[e] = e < N ? e : N;
    wave_lds_sync();
    double CF0, CF1;
    list_gram(N8 / CS, CF0, CF1);
    wave_lds_sync();
    const int i0 = 4 * h + K, f0 = 4 * h + q, i1 = 4 + K;
    if (wave == 0) {                                             // per-replicate constants for the periods without a missing cell
        double* full64 = a.Cfull + (size_t)b * R * R;
        if (g == 0) {
            if (q <= K) { full64[i0 * R + f0] = CF0; full64[f0 * R + i0] = CF0; }
            if (h == 0) { full64[i1 * R + q] = CF1; full64[q * R + i1] = CF1; }
        }
        if (lane == 0) a.ldfull[b] = ldfull;
    }
    if (nrows <= 0 && !TAB) return;                              // (table mode: every wave meets the barrier in front of the tail)
    unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));   // lanes below this one

    // ---- the stream ------------------------------------------------------------------------------------------------
    int bslot = 0;
    for (int k = 0; k < nblk; ++k) {
        const int r0 = k * 4;
        if (k + NB - 1 < nblk) wait_vmx<((NB - 1) * 4 * NDR <= 63 ? (NB - 1) * 4 * NDR : 63)>();
        else wait_vmx<0>();
        const char* blkbase = ring + (unsigned)bslot * 4u * SB;
        const char* pa = blkbase + lane_off;
        double xa[STEPS];
#pragma unroll
        for (int s = 0; s + 1 < STEPS; ++s) xa[s] = *reinterpret_cast<const double*>(pa + s * (CS * 8));
        xa[STEPS - 1] = *reinterpret_cast<const double*>(blkbase + last_off);
        const char* pq = blkbase + lane16;
        double2 xq[4][NQ];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int jq = 0; jq < NQ; ++jq)
                xq[rr][jq] = pact[jq] ? *reinterpret_cast<const double2*>(pq + (unsigned)rr * SB + 1024u * jq)
                                      : make_double2(0.0, 0.0);
        wait_lgkmx();                                            // the reads are done before the slots are re-armed
        if (k + NB < nblk) issue_block(k + NB, bslot);
        // b_t over the observed cells: NaN operands -> 0
        double D = 0.0, D2 = 0.0;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const double x = (xa[s] != xa[s]) ? 0.0 : xa[s];
            if ((s & 1) == 0) D = __builtin_amdgcn_mfma_f64_4x4x4f64(x, Bw[s], D, 0, 0, 0);
            else D2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, Bw[s], D2, 0, 0, 0);
        }
        D += D2;
        D += xor_lane<8>(D);
        {
            const double hi = xor_lane<1>(D);
            const int t = ta + r0 + K;
            if (g == 0 && (q & 1) == 0 && t < tb) {
                if (obs) *reinterpret_cast<double2*>(obs_row(t) + 4 * h + q) = make_double2(D, hi);
                else *reinterpret_cast<double2*>(&a.bcol[((size_t)b * T + t) * R + 4 * h + q]) = make_double2(D, hi);
            }
        }
        // s_t and the missing cells' sum of log R of the 4 periods from the duplicate-free second read of the rows: 8 values,
        // one transpose-reduce
        double red[8];
        unsigned nanbits[4];                                     // bit 2 jq + e: this lane's cell (jq, e) of the row is missing
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            double s = 0.0, l = 0.0;
            unsigned nb = 0;
#pragma unroll
            for (int jq = 0; jq < NQ; ++jq) {
                const double x0 = xq[rr][jq].x, x1 = xq[rr][jq].y;
                const bool n0 = x0 != x0, n1 = x1 != x1;
                nb |= (n0 ? 1u : 0u) << (2 * jq) | (n1 ? 2u : 0u) << (2 * jq);
                l += (n0 ? lr[jq][0] : 0.0) + (n1 ? lr[jq][1] : 0.0);
                const double y0 = n0 ? 0.0 : x0, y1 = n1 ? 0.0 : x1;
                s = fma(y0 * rown[jq][0], y0, s);
                s = fma(y1 * rown[jq][1], y1, s);
            }
            red[rr] = s; red[4 + rr] = l; nanbits[rr] = nb;
        }
        const unsigned long long anynan = __ballot((nanbits[0] | nanbits[1] | nanbits[2] | nanbits[3]) != 0u);
        wave_transpose_reduce<8>(red, lane);
        bool canon;
        const int ridx = reduce_index<8>(lane, canon);
        int nmiss_row[4] = {0, 0, 0, 0};
        if constexpr (TAB) {
            // the masks of the block's four periods: 2 NQ ballots each, ballot (rr, w = 2 jq + e) into lane 4 rr + w of (mlo, mhi);
            // lanes 0..15 store one 64-bit word each (words of a short cross-section stay 0)
            int mlo = 0, mhi = 0;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                int nm = 0;
#pragma unroll
                for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const unsigned long long bal = __ballot(((nanbits[rr] >> (2 * jq + e)) & 1u) != 0u);
                        nm += __popcll(bal);
                        if (2 * jq + e < 4) {
                            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(mlo) : "s"((unsigned)bal), "i"(4 * rr + 2 * jq + e));
                            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(mhi) : "s"((unsigned)(bal >> 32)), "i"(4 * rr + 2 * jq + e));
                        }
                    }
                nmiss_row[rr] = nm;
            }
            if (anynan != 0ull && a.Ct == nullptr && lane == 0) atomicOr(a.status, 1);   // caller promised a balanced panel: flag it
            const int tm = ta + r0 + (lane >> 2);
            if (lane < 16 && tm < tb)
                *reinterpret_cast<uint2*>(obs_row(tm) + 10 + (lane & 3)) = make_uint2((unsigned)mlo, (unsigned)mhi);
        } else
        if (anynan != 0ull && !(abl & 1)) {                      // (wave-uniform) some period of the block has a missing cell
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int t = ta + r0 + rr;
                const unsigned long long rowany = __ballot(nanbits[rr] != 0u);
                if (t >= tb) continue;                           // (wave-uniform)
                if (rowany == 0ull) continue;                    // a complete period
                if (a.Ct == nullptr && lane == 0) atomicOr(a.status, 1);   // caller promised a balanced panel: flag it
                if (a.Ct == nullptr) continue;                   // (no C_t array: keep valid memory)
                // index list of the missing series of the period (ballot compaction), or of the observed ones when fewer
                unsigned long long mb[NQ][2];
                int nmiss = 0;
#pragma unroll
                for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        mb[jq][e] = __ballot(((nanbits[rr] >> (2 * jq + e)) & 1u) != 0u);
                        nmiss += __popcll(mb[jq][e]);
                    }
                nmiss_row[rr] = nmiss;
                const bool comp = 2 * nmiss <= N;                // complement form: C_t = C_full - sum over the missing series
                int base = 0;
#pragma unroll
                for (int jq = 0; jq < NQ; ++jq)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int c = 2 * lane + 128 * jq + e;
                        unsigned long long bits = mb[jq][e];
                        if (!comp) bits = __ballot(c < N && !(((nanbits[rr] >> (2 * jq + e)) & 1u) != 0u));
                        const bool mine = ((bits >> lane) & 1ull) != 0ull;
                        if (mine) idx[base + __popcll(bits & lt_mask)] = c;
                        base += __popcll(bits);
                    }
                const int nlist = base;                          // = nmiss or N - nmiss
                const int nst = (nlist + CS - 1) / CS;
                if (lane < CS) idx[nlist + lane] = N;            // pad the last step with the zero row
                wave_lds_sync();
                double E0, E1;
                list_gram(nst, E0, E1);
                wave_lds_sync();
                if (g == 0) {
                    double* ct = a.Ct + ((size_t)b * T + t) * NP;
                    if (q <= K) ct[i0 * (i0 + 1) / 2 + f0] = comp ? CF0 - E0 : E0;
                    if (h == 0) ct[i1 * (i1 + 1) / 2 + q] = comp ? CF1 - E1 : E1;
                }
            }
        }
        // stores of the reduced values: lane `canon` with index ridx: s_t of period ridx (< 4) or the log-R sum of period ridx - 4
        {
            const int rr = ridx & 3;
            const int t = ta + r0 + rr;
            int nm = nmiss_row[0];
            nm = rr == 1 ? nmiss_row[1] : nm; nm = rr == 2 ? nmiss_row[2] : nm; nm = rr == 3 ? nmiss_row[3] : nm;
            if (canon && t < tb && obs) {                        // doubles 8, 9 of the row: s_t, n_t log 2 pi + sum of log R over the observed
                double* sl = obs_row(t) + 8;
                if (ridx < 4) sl[0] = red[0];
                else sl[1] = (double)(N - nm) * 1.8378770664093454835606594728112 + (ldfull - red[0]);
            } else if (canon && t < tb) {
                if (ridx < 4) {
                    a.scol[(size_t)b * T + t] = red[0];
                    a.nobs[(size_t)b * T + t] = N - nm;
                } else if (nm > 0) {
                    a.ldrow[(size_t)b * T + t] = ldfull - red[0];
                }
            }
        }
        bslot = (bslot + 1 == NB) ? 0 : bslot + 1;
    }
    if constexpr (TAB) {
        // ---- the tail: C_t of the replicate's periods -> the observation table (dfm_ctbuild.h).  The rows are this workgroup's own
        // stores (same CU, write-through L1): complete and visible behind the barrier.  The rings are dead: v = lam / sqrt(R) and the
        // packed full Gram matrix take their place.
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __syncthreads();
        double* Cf = reinterpret_cast<double*>(smem);
        double* V = Cf + kCtbLdsDoubles;
        for (int e = threadIdx.x; e < N * R; e += 256) {
            const int c = e >> 3;
            V[c * kCtbStride + (e & 7)] = Wt[e] * sqrt(Rt[c]);       // (lam / R) sqrt(R)
        }
        if (threadIdx.x < NP) {
            int i = 0;
            while ((i + 1) * (i + 2) / 2 <= (int)threadIdx.x) ++i;
            Cf[threadIdx.x] = a.Cfull[(size_t)b * R * R + i * R + ((int)threadIdx.x - i * (i + 1) / 2)];
        }
        __syncthreads();
        const int L = a.obs_L;
        ct_build_replicate(reinterpret_cast<const unsigned long long*>(a.obs_chunk) + (size_t)b * T * 14,
                           reinterpret_cast<double2*>(a.obs_table) + (size_t)b * L * kObsRows * 64, Cf, V, T, N, L, wave, WPR, lane);
    }
}

// ------------------------------------------------------------------------------------------------------------------
static size_t cm_lds_bytes(int N);
bool collapse_miss_supported(int Rpad, int N) { return Rpad == 8 && collapse_mfma_supported(8, N) && cm_lds_bytes(N) <= 80 * 1024; }

static size_t cm_lds_bytes(int N) {
    const unsigned SB = miss_slot_bytes(N);
    const int N8 = (N + 7) & ~7;
    return (size_t)4 * 8 * SB + ((size_t)(N + 1) * 8 + (N + 2)) * sizeof(double) + (size_t)4 * (N8 + 8) * sizeof(int);
}

template <int STEPS, int NDR, bool TAB>
static hipError_t launch_cm_tab(const CollapseArgs& a, hipStream_t s) {
    const unsigned SB = miss_slot_bytes(a.N);
    const size_t lds = cm_lds_bytes(a.N);
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&collapse_miss_kernel<STEPS, NDR, TAB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static const int abl = [] { const char* v = diag_env("DFM_CM_ABL"); return v ? atoi(v) : 0; }();   // diagnostics: bit 0 skips C_t
    hipLaunchKernelGGL((collapse_miss_kernel<STEPS, NDR, TAB>), dim3(a.B), dim3(256), lds, s, a, SB, abl);
    return hipGetLastError();
}
template <int STEPS, int NDR>
static hipError_t launch_cm_one(const CollapseArgs& a, int num_cu, hipStream_t s) {
    (void)num_cu;
    return a.obs_chunk ? launch_cm_tab<STEPS, NDR, true>(a, s) : launch_cm_tab<STEPS, NDR, false>(a, s);
}

template <int S>
static hipError_t launch_cm_pick(const CollapseArgs& a, int num_cu, hipStream_t s, int steps) {
    if constexpr (S > 32) {
        return hipErrorInvalidValue;
    } else {
        if (steps == S) {
            const int ndr = (a.N * 8 + 1023) / 1024;
            constexpr int lo = (8 * (S - 1) * 8 + 8 + 1023) / 1024, hi = (8 * S * 8 + 1023) / 1024;
            if constexpr (lo <= 1 && 1 <= hi) { if (ndr == 1) return launch_cm_one<S, 1>(a, num_cu, s); }
            if constexpr (lo <= 2 && 2 <= hi) { if (ndr == 2) return launch_cm_one<S, 2>(a, num_cu, s); }
            if constexpr (lo <= 3 && 3 <= hi) { if (ndr == 3) return launch_cm_one<S, 3>(a, num_cu, s); }
            if constexpr (lo <= 4 && 4 <= hi) { if (ndr == 4) return launch_cm_one<S, 4>(a, num_cu, s); }
            return hipErrorInvalidValue;
        }
        return launch_cm_pick<S + 1>(a, num_cu, s, steps);
    }
}

hipError_t launch_collapse_miss(const CollapseArgs& a, int num_cu, hipStream_t s) {
    note_kernel("collapse_miss_kernel");
    return launch_cm_pick<1>(a, num_cu, s, (a.N + 7) / 8);
}

}  // namespace dfm
