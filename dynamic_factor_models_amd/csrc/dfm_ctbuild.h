// dfm_ctbuild.h -- C_t of a panel with missing cells, one LANE per period, straight into the observation table of recursion_chunk.hip.
//
// collapse_miss_kernel's table mode leaves one 112-byte row per period (RecursionArgs::chunk_rows): b_t (8), s_t, n_t log 2 pi + sum
// log R, and the period's NaN bit mask in four 64-bit words -- bit l of word 2 jq + e = series 2 l + 128 jq + e is missing.  Until round
// 6 that kernel built C_t per period with its whole wave (ballots, a compacted index list in LDS, a gathered matrix-pipe contraction,
// two LDS round trips per period): as long again as its stream (0.39 ms against 0.226 at the C2 shape).  Here a lane walks the set bits
// of its own period's mask, reads v_c = lam_c / sqrt(R_c) (8 doubles, LDS table of the replicate) and adds the 36 products of v_c v_c'
// in registers: C_t = C - sum over the missing series, or the sum over the observed ones when those are fewer.  No cross-lane step; the
// 64 periods of a wave differ only in trip count.
//
// The table is chunk-major -- obs[b][slot][23][lane] double2, period t = L lane + slot -- so wave w takes the slots w, w + nw, .. with
// lane = chunk and every store of a period is a contiguous KB.  EVERY slot of all 64 lanes is written: beyond the sample a benign row
// (the replicate's full Gram matrix, zeros), because the lanes of the pass step through such periods uncounted.
// Runs as the TAIL of collapse_miss_kernel's workgroup (the replicate's rows are its own: visible after a workgroup barrier; the table
// stores -- 188 KB per replicate at the C2 shape -- go out while the CU's other workgroup streams its panel).
// Reference counterpart: the normal equations Lambda_t' Lambda_t over the observed series of the per-period regression,
// dfm_functions.ipynb:271-286 (called from :364).
#pragma once
#include <hip/hip_runtime.h>

#include "dfm_chunk_core.h"

namespace dfm {

constexpr int kObsRows = 23;        // double2 rows per period of the observation table: 18 of C_t, 4 of b_t, (s_t, n_t log 2 pi + log det R_t)
constexpr int kCtbStride = 10;      // doubles per series in the LDS table of v (80 bytes: 16-byte aligned rows that start on 16 different banks)
constexpr int kCtbLdsDoubles = 40;  // + N * kCtbStride: [36 (+4)] the full Gram matrix packed, then v

// rows: the replicate's [T][14] rows; table: the replicate's [L][23][64] double2; Cf (LDS): the full Gram matrix, packed; V (LDS): [N][kCtbStride]
__device__ __forceinline__ void ct_build_replicate(const unsigned long long* rows, double2* table, const double* Cf, const double* V,
                                                   int T, int N, int L, int wave, int nwaves, int lane) {
    using namespace chunk;
    // valid-series masks of the four words (series 2 l + 128 jq + e < N)
    unsigned long long valid[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int first = 128 * (w >> 1) + (w & 1);               // series of bit 0; bit l: first + 2 l
        const int nb = first < N ? (N - first + 1) / 2 : 0;       // bits in use
        valid[w] = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
    }
    for (int slot = wave; slot < L; slot += nwaves) {
        const int t = L * lane + slot;
        const bool in = t < T;
        double2* dst = table + (size_t)slot * kObsRows * 64 + lane;
        unsigned long long m[4];
        {   // the period's row: b_t, s_t, ld go straight through to the table (rows 18..22); the mask stays
            const double2* src = reinterpret_cast<const double2*>(rows + (size_t)(in ? t : 0) * 14);
            const double2 m01 = src[5], m23 = src[6];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const double2 q = src[k];
                dst[(NP / 2 + k) * 64] = in ? q : make_double2(0.0, 0.0);
            }
            m[0] = (unsigned long long)__double_as_longlong(m01.x); m[1] = (unsigned long long)__double_as_longlong(m01.y);
            m[2] = (unsigned long long)__double_as_longlong(m23.x); m[3] = (unsigned long long)__double_as_longlong(m23.y);
        }
        int nmiss = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            m[w] = in ? (m[w] & valid[w]) : 0ull;
            nmiss += __popcll(m[w]);
        }
        const bool comp = 2 * nmiss <= N;                          // complement form: C_t = C - sum over the missing series
        if (!comp) {
#pragma unroll
            for (int w = 0; w < 4; ++w) m[w] = ~m[w] & valid[w];
        }
        double E[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) E[k] = 0.0;
        // the lane's list, one series per trip; the NEXT series' row is asked for before the 36 products of this one (a trip is an LDS
        // round trip + 36 independent FMAs: with the read issued a trip ahead the wave waits for neither)
        int w = 0;
        unsigned long long cur = m[0];
        auto next_series = [&]() -> int {                          // -1: the list is done
            while (cur == 0ull && w < 3) { ++w; cur = w == 1 ? m[1] : (w == 2 ? m[2] : m[3]); }
            if (cur == 0ull) return -1;
            const int bit = __ffsll((long long)cur) - 1;
            cur &= cur - 1ull;
            return 2 * bit + 128 * (w >> 1) + (w & 1);
        };
        int c = next_series();
        double2 n01 = make_double2(0.0, 0.0), n23 = n01, n45 = n01, n67 = n01;
        if (c >= 0) {
            const double2* vr = reinterpret_cast<const double2*>(V + c * kCtbStride);
            n01 = vr[0]; n23 = vr[1]; n45 = vr[2]; n67 = vr[3];
        }
        while (c >= 0) {
            const double v[R] = {n01.x, n01.y, n23.x, n23.y, n45.x, n45.y, n67.x, n67.y};
            c = next_series();
            if (c >= 0) {
                const double2* vr = reinterpret_cast<const double2*>(V + c * kCtbStride);
                n01 = vr[0]; n23 = vr[1]; n45 = vr[2]; n67 = vr[3];
            }
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) E[pidx(i, j)] = fma(v[i], v[j], E[pidx(i, j)]);
        }
#pragma unroll
        for (int k = 0; k < NP / 2; ++k) {
            const double c0 = Cf[2 * k], c1 = Cf[2 * k + 1];
            dst[k * 64] = comp ? make_double2(c0 - E[2 * k], c1 - E[2 * k + 1]) : make_double2(E[2 * k], E[2 * k + 1]);
        }
    }
}

}  // namespace dfm
