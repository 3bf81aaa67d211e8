// collapse_ks.hip -- the balanced-panel collapse for wide cross-sections at Rp = 32, 17 <= r <= 20 (BASELINE config 4: N = 1000,
// T = 2000, r = 20) with the SERIES split over the waves and the weights in REGISTERS.
//
//     b_t = sum_i lam_i x_it / R_i   (factors 0..15 on v_mfma_f64_16x16x4, 16..19 on v_mfma_f64_4x4x4)      s_t = sum_i x_it^2 / R_i
//
// collapse_wide2_kernel (collapse_wide2.hip) gives every wave 16 PERIODS of a 128-period tile and streams the tile through LDS
// 32 series at a time: 128 row pieces of 256 bytes per stage, and the 32 x 32 block of W = lam / R of those series with them --
// W comes back from L2 once per tile, a fifth of the kernel's LDS-DMA traffic, and the kernel streams config 4's 4.1 GB at
// 4.6 TB/s where the same DMA ring alone reads them at 6.2 (scripts/microbench/segbw.hip, profiles/r04/microbench_segbw.txt).
// Here a stage is 16 periods x 256 series (row pieces of 2 KB), the eight consumer waves share its 16 periods and split its
// SERIES: wave w owns series 32 w .. 32 w + 31 of every stage -- the same 128 series in every 16-period group of the item, so
// their weights live in registers (40 doubles per lane: 32 B operands of the 16x16x4 steps, 8 A operands of the 4x4x4 steps,
// which carry 16 series each) and are fetched once per item of 256 periods with ordinary loads.  No W in the DMA stream, ONE 8-byte LDS read per step of 4 series (the A operand) plus a broadcast
// read of 1 / R for s_t, rows long enough for the non-temporal hint to pay (segbw: 6.5 -> 6.9 TB/s at 2-KB pieces).
// The price: the eight partial b_t (16 periods x 20 factors each) of a group meet in LDS -- 24 KB written and read per 128 KB
// of panel, folded into the first stage of the next group.
//   workgroup  persistent, one per CU: 8 consumer waves + 4 producer waves (ALL the LDS-DMA: 8 instructions per stage each, a
//              counted wait leaves the next stage in flight): two consumers and a producer on every SIMD, 168 registers a wave;
//              one barrier per stage, three stage buffers of 33 KB; the stage stream runs across groups and items without draining
//   items      256 (or fewer) periods of one replicate, dealt STATICALLY: XCD x takes replicates x, x + 8, ..., its 32
//              workgroups the items of those replicates in turn (one replicate's 1 / R, lam and panel stay in one L2)
//   LDS rows   2048 + 16 bytes apart: the 8-byte slot of (period i, series k of a step) is 2 i + k mod 32 -- an A read
//              (16 periods x 2 series per half-wave) covers the 32 slots once
// Supported: Rp = 32, 17 <= r <= 20, 256 < N <= 1024 even, no missing cells (CollapseArgs::nobs == nullptr).  Everything else
// stays with collapse_wide2_kernel.  b_t rows are CollapseArgs::bst doubles apart.  sum_t s_t per 16-period group -> scol[b][group].
// Reference counterpart: forming Lambda' x_t in the per-period regression of x_t on Lambda (dfm_functions.ipynb:271-286 called
// from :364).
#include <type_traits>

#include "dfm_gram.h"
#include "dfm_kernels.h"

namespace dfm {

namespace {

using lds_char_ptr_k = __attribute__((address_space(3))) char*;
using lds_cvd_ptr_k = const volatile __attribute__((address_space(3))) double*;
// one ds_read_b64 (never merged into ds_read2_b64, never moved relative to other volatile accesses) of LDS byte address a
__device__ __forceinline__ double ks_read64(unsigned a) { return *(lds_cvd_ptr_k)(size_t)a; }
typedef double ks_v4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void ks_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

constexpr int kKsR = 32;
constexpr int kKsG = 16;                             // periods per group
constexpr int kKsSer = 256;                          // series per stage
constexpr int kKsCons = 8, kKsProd = 4;
constexpr int kKsSteps = kKsSer / (4 * kKsCons);     // steps of 4 series per consumer wave and stage: 8 (series 32 w .. 32 w + 31 of the stage)
constexpr int kKsNG = kKsSteps / 4;                  // 16-series groups of a wave's slice (the 4x4x4 part)
constexpr int kKsMaxSt = 4;                          // stages per group: N <= 1024
constexpr unsigned kKsRowB = kKsSer * 8 + 16;        // 2064
constexpr unsigned kKsStageB = kKsG * kKsRowB;       // 33024
constexpr int kKsNBuf = 3;
constexpr int kKsThreads = 64 * (kKsCons + kKsProd);
constexpr int kKsTab = kKsMaxSt * kKsSer;            // entries of a 1 / R table
// LDS: stage buffers | partials [8 waves][6][64] doubles | two 1 / R tables
constexpr unsigned kKsRedOff = kKsNBuf * kKsStageB;
constexpr unsigned kKsTabOff = kKsRedOff + kKsCons * 6 * 64 * 8;
constexpr unsigned kKsLds = kKsTabOff + 2 * kKsTab * 8;

// the static item list of a workgroup (see the head of the file)
struct KsSched {
    int B, T, IR, IPR, xcd_map, x, y, GY, nitems;
    __device__ __forceinline__ void init(int B_, int T_, int IR_, int xcd_map_, int g, int G) {
        B = B_; T = T_; IR = IR_; xcd_map = xcd_map_;
        IPR = (T + IR - 1) / IR;
        if (xcd_map) {
            x = g & 7; y = g >> 3; GY = G >> 3;
            const int nrep = x < B ? (B - x + 7) >> 3 : 0;
            const int tot = nrep * IPR;
            nitems = tot > y ? (tot - y + GY - 1) / GY : 0;
        } else {
            x = 0; y = g; GY = G;
            const int tot = B * IPR;
            nitems = tot > y ? (tot - y + GY - 1) / GY : 0;
        }
    }
    // item j of this workgroup: replicate, first period, number of 16-period groups
    __device__ __forceinline__ void item(int j, int& b, int& t0, int& ng) const {
        const int idx = y + j * GY;
        const int m = idx / IPR, c = idx - m * IPR;
        b = xcd_map ? x + 8 * m : m;
        t0 = c * IR;
        const int t1 = t0 + IR < T ? t0 + IR : T;
        ng = (t1 - t0 + kKsG - 1) / kKsG;
    }
};

}  // namespace

template <bool NT>
__global__ __launch_bounds__(kKsThreads) void collapse_ks_kernel(CollapseArgs a, const double* __restrict__ rinvAll, int npad, int IR, int xcd_map, int abl) {
    // abl (DFM_KS_ABL, diagnostics build, WRONG results): 1 = no compute (stream only), 2 = no DMA (compute only), 4 = no 4x4x4 part
    constexpr int R = kKsR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, T = a.T, B = a.B;
    const int nst = (N + kKsSer - 1) / kKsSer;                // stages per group (2 .. 4)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_k)(smem));
    double* red = reinterpret_cast<double*>(smem + kKsRedOff);
    double* rtab = reinterpret_cast<double*>(smem + kKsTabOff);
    KsSched sc;
    sc.init(B, T, IR, xcd_map, (int)blockIdx.x, (int)gridDim.x);
    const int bst = a.bst > 0 ? a.bst : R;

    // no NaN bit patterns where the DMAs never write (the 16 bytes between rows): zero the stage buffers once
    for (int e = tid; e < (int)(kKsNBuf * kKsStageB / 8); e += kKsThreads) reinterpret_cast<double*>(smem)[e] = 0.0;
    // 1 / R of the first item's replicate (zero past N)
    if (sc.nitems > 0) {
        int b0, t00, ng0;
        sc.item(0, b0, t00, ng0);
        for (int e = tid; e < kKsTab; e += kKsThreads) rtab[e] = e < N ? rinvAll[(size_t)b0 * npad + e] : 0.0;
    }
    __syncthreads();

    if (wave >= kKsCons) {
        // ---- producer waves: rows 4 p .. 4 p + 3 of every stage, two 1-KB pieces each -- 8 DMAs per stage, always (lanes past the
        // end of a row re-read its last 16 bytes: they only ever meet a zero weight)
        const int pw = wave - kKsCons;
        __builtin_amdgcn_s_setprio(3);
        const unsigned rowB = (unsigned)N * 8u;
        int ij = 0, iu = 0, ig = 0, ib = 0, it0 = 0, ing = 0;  // issue cursor: item, group, stage
        bool have = sc.nitems > 0;
        if (have) sc.item(0, ib, it0, ing);
        auto issue = [&](int bsel) {
            const char* Xb = reinterpret_cast<const char*>(a.panel + (size_t)ib * T * N);
            const int tg = it0 + kKsG * iu;
            const unsigned sbase = lds0 + (unsigned)bsel * kKsStageB;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = 4 * pw + (k >> 1), h = k & 1;
                int t = tg + row;
                t = t < T ? t : T - 1;
                unsigned colB = (unsigned)ig * (kKsSer * 8u) + 1024u * h + 16u * lane;
                colB = colB < rowB ? colB : rowB - 16u;
                const unsigned dst = __builtin_amdgcn_readfirstlane(sbase + (unsigned)row * kKsRowB + 1024u * h);
                if (!(abl & 2)) ks_dma16<NT>(Xb + (size_t)t * rowB + colB, dst);
            }
            if (++ig == nst) {
                ig = 0;
                if (++iu == ing) {
                    iu = 0;
                    have = ++ij < sc.nitems;
                    if (have) sc.item(ij, ib, it0, ing);
                }
            }
        };
        bool more = have, v1 = false;
        if (more) {
            issue(0);
            v1 = have;
            if (v1) issue(1);
        }
        int bsel = 0;
        while (more) {
            if (v1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                  // stage q has landed; every consumer is done with stage q - 1
            const bool v2 = have;
            if (v2) issue(bsel == 0 ? 2 : bsel - 1);
            bsel = bsel == 2 ? 0 : bsel + 1;
            more = v1;
            v1 = v2;
        }
        __syncthreads();
        return;
    }

    // ---- consumer waves -------------------------------------------------------------------------------------------------
    const int k4 = lane >> 4, i16 = lane & 15;                 // A operand: period i16 of the group, series k4 of the step
    const unsigned aoff = (unsigned)i16 * kKsRowB + (unsigned)(32 * wave + k4) * 8u;
    const unsigned roff = (unsigned)(32 * wave + k4) * 8u;
    const int blk = (lane >> 2) & 3, q4 = lane & 3;            // 4x4x4 lane coordinates (K = k4)
    // B operand of the 4x4x4 steps: x[period 4 m + q][series 32 w + 16 G + 4 blk + K]
    const unsigned boff = (unsigned)q4 * kKsRowB + (unsigned)(32 * wave + 4 * blk + k4) * 8u;
    // barrier for data that travels through LDS (the producers waited for the DMAs): __syncthreads() would also wait for this
    // wave's outstanding global accesses -- the b_t stores of the last flush, the next item's 1 / R
    auto ks_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    int bsel = 0;
    // a finished group: its partials wait in `red`, folded after the next barrier
    int pend_b = -1, pend_t0 = 0;
    auto flush_pending = [&]() {
        if (pend_b < 0) return;
        double* out = a.bcol + ((size_t)pend_b * T + pend_t0) * bst;
        double t = 0.0;
        if (wave >= 6) { pend_b = -1; return; }
#pragma unroll
        for (int u = 0; u < kKsCons; ++u) t += red[(u * 6 + wave) * 64 + lane];   // wave w < 6 folds slot w of the eight partials
        if (wave < 4) {                                       // 16x16x4: D[(l / 16) + 4 v][l % 16], v = wave
            const int p = k4 + 4 * wave;
            if (pend_t0 + p < T) out[(size_t)p * bst + i16] = t;
        } else if (wave == 4) {                               // entry 16 m + 4 K + q: factor 16 + K of period 4 m + q
            const int p = 4 * (lane >> 4) + (lane & 3);
            if (pend_t0 + p < T) out[(size_t)p * bst + 16 + ((lane >> 2) & 3)] = t;
        } else {                                              // sum of s_t over the group; the padding columns 20 .. bst - 1
            t = wave_allsum(t);
            if (lane == 0) {
                a.scol[(size_t)pend_b * T + pend_t0 / kKsG] = t;
                if (t != t) atomicOr(a.status, 1);            // NaN in the panel on the balanced path
            }
            const int p = lane >> 2;
            if (pend_t0 + p < T)
                for (int f = 20 + (lane & 3); f < bst; f += 4) out[(size_t)p * bst + f] = 0.0;
        }
        pend_b = -1;
    };

    for (int j = 0; j < sc.nitems; ++j) {
        int b, t0, ng;
        sc.item(j, b, t0, ng);
        // weights of this wave's series.  Wr: B operands of the 16x16x4 steps (series 32 w + 4 s + k4 of the stage, factor i16).
        // W4n: factors 16..19 as the A operand of v_mfma_f64_4x4x4 with its four blocks on four DIFFERENT groups of 4 series -- lane
        // (K, blk, q) holds the weight of series 32 w + 16 G + 4 blk + K on factor 16 + q, so ONE register covers 16 series (as the B
        // operand -- the same four series in every block, the blocks on four period groups -- it takes one per 4 series).
        // Loads are unconditional (clamped index, the product times 0 or 1): 10 independent pairs per stage, no branches.
        double Wr[kKsMaxSt][kKsSteps], W4n[kKsMaxSt][kKsNG];
        {
            const double* Lb = a.Lam + (size_t)b * N * R;
            const double* rb = rinvAll + (size_t)b * npad;
            // stage g lies wholly inside the cross-section unless it is the last one: constant address offsets there, clamped indices
            // (and a 0 / 1 factor) only in the last stage
            auto load_stage = [&](auto gtag, auto clamp_tag) {
                constexpr int g = decltype(gtag)::value;
                constexpr bool CL = decltype(clamp_tag)::value;
                const int cA = kKsSer * g + 32 * wave + k4, cB = kKsSer * g + 32 * wave + 4 * blk + k4;
                double lv[kKsSteps + kKsNG], rv[kKsSteps + kKsNG];
#pragma unroll
                for (int s = 0; s < kKsSteps + kKsNG; ++s) {
                    const int c = s < kKsSteps ? cA + 4 * s : cB + 16 * (s - kKsSteps);
                    const int cc = CL ? (c < N ? c : N - 1) : c;
                    lv[s] = Lb[(size_t)cc * R + (s < kKsSteps ? i16 : 16 + q4)];
                    rv[s] = rb[cc];
                }
#pragma unroll
                for (int s = 0; s < kKsSteps + kKsNG; ++s) {
                    const int c = s < kKsSteps ? cA + 4 * s : cB + 16 * (s - kKsSteps);
                    const double v = (CL && c >= N) ? 0.0 : lv[s] * rv[s];
                    if (s < kKsSteps) Wr[g][s] = v; else W4n[g][s - kKsSteps] = v;
                }
                asm volatile("" ::: "memory");
            };
            auto load_g = [&](auto gtag) {
                constexpr int g = decltype(gtag)::value;
                if (g >= nst) {
#pragma unroll
                    for (int s = 0; s < kKsSteps; ++s) Wr[g][s] = 0.0;
#pragma unroll
                    for (int G = 0; G < kKsNG; ++G) W4n[g][G] = 0.0;
                } else if (kKsSer * (g + 1) <= N) load_stage(gtag, std::false_type{});
                else load_stage(gtag, std::true_type{});
            };
            load_g(std::integral_constant<int, 0>{}); load_g(std::integral_constant<int, 1>{});
            load_g(std::integral_constant<int, 2>{}); load_g(std::integral_constant<int, 3>{});
            static_assert(kKsMaxSt == 4, "load_g calls");
        }
        const bool nxt = j + 1 < sc.nitems;
        const unsigned rt = lds0 + kKsTabOff + (unsigned)(j & 1) * (kKsTab * 8u) + roff;
        for (int u = 0; u < ng; ++u) {
            const int tg = t0 + kKsG * u;
            const bool rowok = tg + i16 < T;                    // (rows past the end of the sample repeat row T - 1: they count for nothing)
            ks_v4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            double acc4m[4] = {0.0, 0.0, 0.0, 0.0}, qs = 0.0;    // acc4m[m]: lane (K, blk, q) = (factor 16 + K, period 4 m + q), partial over the block's series
            bool rowokB[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) rowokB[m] = tg + 4 * m + q4 < T;
#pragma unroll
            for (int g = 0; g < kKsMaxSt; ++g) {
                if (g < nst) {                                  // (uniform)
                    ks_barrier();                               // the stage has landed (the producers waited for it)
                    flush_pending();
                    const unsigned sb = lds0 + (unsigned)bsel * kKsStageB;
                    const unsigned st = sb + aoff;
                    const unsigned rg = rt + (unsigned)g * (kKsSer * 8u);
                    if (!(abl & 1)) {
                    // operands four steps ahead of their MFMAs; the B operands of the 4x4x4 part with the first of them
                    double av[kKsSteps], ri[kKsSteps], bx[kKsNG][4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) { av[s] = ks_read64(st + 32u * s); ri[s] = ks_read64(rg + 32u * s); }
#pragma unroll
                    for (int G = 0; G < kKsNG; ++G)
#pragma unroll
                        for (int m = 0; m < 4; ++m) bx[G][m] = ks_read64(sb + boff + 128u * G + (unsigned)(4 * m) * kKsRowB);
#pragma unroll
                    for (int s = 0; s < kKsSteps; ++s) {
                        if (s + 4 < kKsSteps) { av[s + 4] = ks_read64(st + 32u * (s + 4)); ri[s + 4] = ks_read64(rg + 32u * (s + 4)); }
                        const double a_ = rowok ? av[s] : 0.0;
                        if (s & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, Wr[g][s], acc1, 0, 0, 0);
                        else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, Wr[g][s], acc0, 0, 0, 0);
                        qs = fma(a_ * ri[s], a_, qs);
                    }
                    if (!(abl & 4)) {
#pragma unroll
                    for (int G = 0; G < kKsNG; ++G)
#pragma unroll
                        for (int m = 0; m < 4; ++m) acc4m[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(W4n[g][G], rowokB[m] ? bx[G][m] : 0.0, acc4m[m], 0, 0, 0);
                    }
                    }
                    bsel = bsel == 2 ? 0 : bsel + 1;
                    if (nxt && u == ng - 1 && g == nst - 1) {   // the last stage of the item: 1 / R of the next item's replicate into the other table
                        int bn, tn_, gn;
                        sc.item(j + 1, bn, tn_, gn);
                        double* tn = rtab + (size_t)((j + 1) & 1) * kKsTab;
                        for (int e = tid; e < kKsTab; e += 64 * kKsCons) tn[e] = e < N ? rinvAll[(size_t)bn * npad + e] : 0.0;
                    }
                }
            }
            // park the group's partials (slot 4: entry 16 m + 4 K + q = factor 16 + K of period 4 m + q, the four blocks summed)
            {
                const ks_v4 acc = acc0 + acc1;
                double* rw = red + (size_t)wave * 6 * 64;
                rw[lane] = acc[0]; rw[64 + lane] = acc[1]; rw[128 + lane] = acc[2]; rw[192 + lane] = acc[3];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    double v = acc4m[m];
                    v += __shfl_xor(v, 4, 64);
                    v += __shfl_xor(v, 8, 64);
                    if (blk == 0) rw[256 + 16 * m + 4 * k4 + q4] = v;
                }
                rw[320 + lane] = qs;
            }
            pend_b = b; pend_t0 = tg;
        }
    }
    ks_barrier();
    flush_pending();
}

// Rp = 32, 17 <= r <= 20, even 256 < N <= 1024, balanced -- and only when asked for: DFM_COLLAPSE_KS=1 (route switch).
// MEASURED SLOWER than collapse_wide2_kernel at BASELINE config 4 (profiles/r04/ab_collapse_ks_c4.txt: 1.02 ms against 0.73 net
// of the covariance kernel in front of both; its stream alone 0.77 ms = 5.3 TB/s, its compute alone 0.79 ms), so the default stays
// with that kernel.  What it established: the weights fit in registers once the 4x4x4 part takes them as its A operand, the
// result is the same to rounding (tests/test_gpu_round4.py::test_collapse_ks_route), and neither the W traffic nor the short row
// pieces are what holds the streaming collapse at 4.6 TB/s -- a stage loop of this shape does not reach the DMA ring's rate either.
bool collapse_ks_supported(int Rpad, int r, int N, bool missing) {
    static const bool on = [] { const char* v = route_env("DFM_COLLAPSE_KS"); return v && atoi(v) != 0; }();
    return on && !missing && Rpad == 32 && r >= 17 && r <= 20 && (N % 2) == 0 && N > kKsSer && N <= kKsMaxSt * kKsSer;
}
int collapse_ks_tiles(int T) { return (T + kKsG - 1) / kKsG; }

hipError_t launch_collapse_ks(const CollapseArgs& a, const double* rinv, int npad, int num_cu, hipStream_t s) {
    note_kernel("collapse_ks_kernel");
    int G = ((num_cu > 0 ? num_cu : 256) / 8) * 8;
    if (G < 8) G = 8;
    const int xcd_map = a.B >= 16;
    // periods per item: 256, fewer while the batch does not give every workgroup a few items (never below 64: the weights of an
    // item are 256 KB of L2 reads)
    int IR = 256;
    while (IR > 64 && (long long)a.B * ((a.T + IR - 1) / IR) < 4ll * G) IR >>= 1;
    static const int nt_force = [] { const char* v = route_env("DFM_DMA_NT"); return v ? (atoi(v) != 0 ? 1 : 0) : -1; }();
    const bool nt = nt_force >= 0 ? nt_force == 1 : true;
    static const int abl = [] { const char* v = diag_env("DFM_KS_ABL"); return v ? atoi(v) : 0; }();
    static const int ir_env = [] { const char* v = diag_env("DFM_KS_IR"); return v ? atoi(v) : 0; }();
    if (ir_env >= 16) IR = (ir_env / 16) * 16;
    static LdsOptIn attr_done;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&collapse_ks_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&collapse_ks_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (nt) hipLaunchKernelGGL(collapse_ks_kernel<true>, dim3((unsigned)G), dim3(kKsThreads), kKsLds, s, a, rinv, npad, IR, xcd_map, abl);
    else hipLaunchKernelGGL(collapse_ks_kernel<false>, dim3((unsigned)G), dim3(kKsThreads), kKsLds, s, a, rinv, npad, IR, xcd_map, abl);
    return hipGetLastError();
}

}  // namespace dfm
