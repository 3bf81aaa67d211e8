// collapse_wide2.hip -- the balanced-panel collapse for Rp = 32 (BASELINE config 4: N = 1000, T = 2000, r = 20) on the
// LDS-DMA path and the f64 matrix pipe (`v_mfma_f64_16x16x4` for factors 0..15, `v_mfma_f64_4x4x4` for every further 4).
//
//     b_t = sum_i lam_i x_it / R_i   (32 padded factors)        sum_t s_t,  s_t = sum_i x_it^2 / R_i
//
// collapse_wide_kernel (collapse_wide.hip) loads its A operands straight from the panel -- 4 rows x 32..64 bytes per load
// instruction -- divides by R in every step and re-reads the weights from L2 once per 16-period tile: 3.5 ms for the
// 4.1 GB of config 4 (1.17 TB/s, 0.15 of HBM peak).  Here:
//   * wide_prep_kernel (once per pass): W = lam / R and 1 / R into the workspace, C = Lam' W (32 x 32) on the matrix pipe
//     (4 waves = the 2 x 2 tiles of 16 x 16, a chain of v_mfma_f64_16x16x4 over the series each), sum log R, and the tile
//     queue counters reset;
//   * collapse_wide2_kernel: PERSISTENT workgroups, one per CU, 8 streaming waves + 1 scheduler wave.  An item is a tile of
//     128 periods x ALL series of one replicate, taken from a queue (one atomic counter per XCD) by the scheduler wave one
//     tile ahead of need -- the covariance kernel runs beside this one and holds some CUs for the first ~0.2 ms, a static
//     split would make the whole launch wait for those CUs' shares.  The tile streams through LDS in stages of 32 series:
//     128 rows x 256 bytes of the panel, the 32 x 32 block of W (8 KB) and 32 values of 1 / R, all by
//     `global_load_lds_dwordx4`, three stage buffers (two in flight while one is consumed), one barrier per stage, and the
//     stage stream runs ACROSS tile boundaries (no fill / drain per tile).  Wave w owns the 16 periods 16 w .. 16 w + 15 of
//     the tile: per step of 4 series ONE 8-byte LDS read of A (A[i][k] = x[t0 + 16 w + i][c + k]) feeds the 16x16x4 MFMA
//     (factors 0..15) and NX 4x4x4 MFMAs (factors 16 + 4 x ..: their A operand has the same lane layout), and s_t
//     (x^2 / R from the same register).  r = 20 costs 80 MFMA cycles per step instead of the 128 of two 16-wide tiles.
//   * LDS layouts are chosen so that every operand read is bank-conflict free (the first version of this kernel was LDS-bound:
//     2-way conflicts on A and B, 4-way on a separate s_t pass): see kW2GroupB and the W swizzle.
// HBM sees every panel byte once; W is read once per 128-period tile (1/4 of the panel bytes) and mostly from L2.
// Reference counterpart: forming Lambda' x_t in the per-period regression of x_t on Lambda (dfm_functions.ipynb:271-286
// called from :364).
#include <type_traits>

#include "dfm_gram.h"
#include "dfm_kernels.h"

// cache-policy modifiers of the streaming LDS-DMA loads (development A/B: -DDFM_DMA_MOD='" nt"', '" sc1"', ...); default: none
#ifndef DFM_DMA_MOD
#define DFM_DMA_MOD ""
#endif

namespace dfm {

namespace {

using lds_char_ptr_w = __attribute__((address_space(3))) char*;
using lds_cvd_ptr_w = const volatile __attribute__((address_space(3))) double*;
// one ds_read_b64 (never merged into ds_read2_b64, never moved relative to other volatile accesses) of LDS byte address a
__device__ __forceinline__ double lds_read64(unsigned a) { return *(lds_cvd_ptr_w)(size_t)a; }
typedef double w2_v4 __attribute__((ext_vector_type(4)));
typedef double w2_v2 __attribute__((ext_vector_type(2)));
typedef unsigned w2_u4 __attribute__((ext_vector_type(4)));
typedef unsigned w2_u2 __attribute__((ext_vector_type(2)));
using lds_cv2_ptr_w = const __attribute__((address_space(3))) w2_v2*;
using lds_cu4_ptr_w = const __attribute__((address_space(3))) w2_u4*;
using lds_cu2_ptr_w = const __attribute__((address_space(3))) w2_u2*;
using lds_cu1_ptr_w = const __attribute__((address_space(3))) unsigned*;

__device__ __forceinline__ void dma16w(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off" DFM_DMA_MOD "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
// the PANEL rows of collapse_wide2_kernel (development A/B: -DDFM_W2_PANEL_MOD='" nt"' puts a modifier on them alone -- the W / R
// tables of a replicate are re-read from L2 by every CU of its XCD and must stay cacheable)
#ifndef DFM_W2_PANEL_MOD
#define DFM_W2_PANEL_MOD DFM_DMA_MOD
#endif
__device__ __forceinline__ void dma16wp(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off" DFM_W2_PANEL_MOD "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
__device__ __forceinline__ void wait_all_w() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr int kW2R = 32;          // padded factors (widest; the kernels are templates in R = 16 | 32, the stage layout is R = 32's)
#ifndef DFM_W2_ROWS
#define DFM_W2_ROWS 128
#endif
// periods per tile: 128 -> 130 KB of LDS per workgroup, one per CU.  (64 -> 79 KB, two per CU, or one beside a 60-KB workgroup
// of the covariance kernel that runs next to this launch: measured slower -- 1.29 ms against 1.04 with the covariance
// kernel beside it, which itself went from 0.20 to 0.34 ms; alone 0.88 against 0.86.)
constexpr int kW2Rows = DFM_W2_ROWS;
constexpr int kW2Chunk = 32;      // series per stage (256 bytes of a panel row)
constexpr int kW2Steps = kW2Chunk / 4;
constexpr int kW2NBuf = 3;        // stage buffers
// One DMA lands FOUR rows x 256 bytes back to back (16 lanes each), so rows of a group start a whole number of bank sweeps
// apart; read side by side (A operand: 16 rows x 4 series per instruction) they would collide.  Row h of a group is
// therefore stored ROTATED by h 16-byte units (its lane u fetches piece u - h mod 16) and groups are 1024 + 64 bytes
// apart: the 8-byte slot of (group g, row h, series k) is 8 g + 2 h + k mod 32 -- each half-wave of an A read (16 rows x 2
// series) covers the 32 slots exactly once.
constexpr unsigned kW2GroupB = 1088;
constexpr unsigned kW2PanelB = (kW2Rows / 4) * kW2GroupB;          // 34816
constexpr unsigned kW2WB = kW2Chunk * kW2R * 8;                    // 8192: W block of the stage
constexpr unsigned kW2StageB = kW2PanelB + kW2WB + 2 * kW2Chunk * 8;   // + 1 / R and (panels with missing cells) log R of the stage's series
constexpr int kW2Compute = kW2Rows / 16;    // consumer waves (16 periods of the tile each)
constexpr int kW2Producers = kW2Rows / 32;  // LDS-DMA waves (32 rows + their share of W each); one more wave is the scheduler
template <int R> struct W2Geo {
    static constexpr unsigned WB = kW2Chunk * R * 8;               // bytes of the stage's W block: 8 KB (R = 32), 4 KB (R = 16)
    static constexpr int WPieces = (int)(WB / 1024) / kW2Producers;   // 1-KB DMAs of it per producer
    static constexpr int PerStage = 8 + WPieces;                   // DMAs per producer and stage (+ 1 for producer 0: 1 / R)
};
constexpr int kW2Threads = 64 * (kW2Compute + kW2Producers + 1);
constexpr int kW2Ring = 8;        // published items (ring)

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// W = lam / R, C = Lam' W, sum log R.  One workgroup of 16 waves per replicate: the (R / 16)^2 tiles of C x 16 / tiles slices of
// the series (a 4-wave version ran each tile's chain over ALL series: ~30 dependent batches of loads, 0.1 ms -- a fifth of the
// pass at r = 8 / N = 1000); the slices meet in LDS.
// rd = width of the caller's arrays (Lam [N][rd], Cfull [rd][rd]); rd < R (= 16) for the narrow states whose cross-section is
// beyond the row ring of the MFMA collapse: W is padded with zero columns, the collapse computes 16 and stores rd.
constexpr int kPrepThreads = 1024;
template <int R>
__global__ __launch_bounds__(kPrepThreads) void wide_prep_kernel(CollapseArgs a, double* Wout, double* rinv_out, double* logr_out, int npad, int* ctr, int rd, double* Vout) {
    constexpr int NW = kPrepThreads / 64, NTILE = (R / 16) * (R / 16), NSL = NW / NTILE;
    __shared__ double red[NW];
    __shared__ double part[NW * 4 * 64];                      // [wave][v][lane]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N;
    const double* __restrict__ L = a.Lam + (size_t)b * N * rd;
    const double* __restrict__ Rv = a.Rv + (size_t)b * N;
    double* W = Wout + (size_t)b * N * R;
    // 1 / R and log R ONCE per series (a first version divided every element of W by R and took every logarithm twice: 32 fp64
    // divisions and 2 logarithms per thread in front of a pass whose streaming kernel waits for W)
    double* rinv = rinv_out + (size_t)b * npad;
    double ld = 0.0;
    for (int c = tid; c < npad; c += kPrepThreads) {          // (0 past N: padding of the last stage)
        const double rv = c < N ? Rv[c] : 1.0;
        const double lg = c < N ? log(rv) : 0.0;
        rinv[c] = c < N ? 1.0 / rv : 0.0;
        logr_out[(size_t)b * npad + c] = lg;
        ld += lg;
    }
    if (b == 0 && tid < 8) ctr[tid] = 0;                      // tile queues of the collapse that follows on this stream
    ld = wave_allsum(ld);
    if (lane == 0) red[wave] = ld;
    __syncthreads();                                          // (1 / R of this replicate is visible to its workgroup)
    // W stored with the two 16-column halves of ODD series swapped (column f of series c at f ^ 16 (c & 1)): the B operand of
    // a step reads 4 consecutive series x 16 columns, and two series 256 bytes apart would meet on the same banks
    if (Wout != nullptr) {                                    // (null: the balanced Rp = 32 collapse takes lam and 1 / R themselves)
        for (int e = tid; e < N * R; e += kPrepThreads) {
            const int c = e / R;
            const int f = e % R;
            W[R == 32 ? (e ^ (16 * (c & 1))) : e] = f < rd ? L[(size_t)c * rd + f] * rinv[c] : 0.0;
        }
    }
    // V = lam / sqrt(R), plain [N][R] rows (ct_miss_dma_kernel: the deficit of a period is the sum of v_i v_i' over its missing series)
    if (Vout != nullptr) {
        double* V = Vout + (size_t)b * N * R;
        for (int e = tid; e < N * R; e += kPrepThreads) {
            const int c = e / R, f = e % R;
            V[e] = f < rd ? L[(size_t)c * rd + f] * sqrt(rinv[c]) : 0.0;
        }
    }
    // tile (it, jt) of C over the series of slice sl: C[16 it + i][16 jt + j] = sum_c Lam[c][16 it + i] Lam[c][16 jt + j] / R_c
    const int tile = wave % NTILE, sl = wave / NTILE;
    const int it = tile / (R / 16), jt = tile % (R / 16);
    const int k4 = lane >> 4, c16 = lane & 15;
    w2_v4 acc = {0.0, 0.0, 0.0, 0.0};
    const int steps = (N + 3) / 4;
    const int sps = (steps + NSL - 1) / NSL;
    const int s_lo = sl * sps, s_hi = (s_lo + sps < steps) ? s_lo + sps : steps;
    for (int s0 = s_lo; s0 < s_hi; s0 += 8) {                 // 24 loads in flight per batch (clamped addresses, select afterwards)
        double av[8], bv[8], rv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = 4 * (s0 + u) + k4;
            const int cc = c < N ? c : N - 1;
            av[u] = (16 * it + c16 < rd) ? L[(size_t)cc * rd + 16 * it + c16] : 0.0;
            bv[u] = (16 * jt + c16 < rd) ? L[(size_t)cc * rd + 16 * jt + c16] : 0.0;
            rv[u] = rinv[cc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = 4 * (s0 + u) + k4;
            const bool ok = c < N && s0 + u < s_hi;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? av[u] : 0.0, bv[u] * rv[u], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) part[(wave * 4 + v) * 64 + lane] = acc[v];
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < NW; ++w) t += red[w];
        a.ldfull[b] = t;
    }
    if (sl == 0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {                         // D[(l / 16) + 4 v][l % 16]
            double t = 0.0;
            for (int q = 0; q < NSL; ++q) t += part[((q * NTILE + tile) * 4 + v) * 64 + lane];
            if (16 * it + k4 + 4 * v < rd && 16 * jt + c16 < rd)
                a.Cfull[(size_t)b * rd * rd + (size_t)(16 * it + k4 + 4 * v) * rd + 16 * jt + c16] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Streaming collapse (see the head of the file).  NX = 4-factor groups past the first 16 (1..3), or 4 = a second 16-wide tile.
// Workgroup -> tiles: `xcd_map` (B >= 16): workgroups are dealt to the 8 XCDs round-robin, so workgroup g sits on XCD g % 8;
// XCD x takes items from queue x = tiles of replicates x, x + 8, ... in order: its CUs work on ONE replicate at a time and
// that replicate's 256 KB of W stay in that L2.  Small batches: one queue over the flat tile list.
// MODE 0: balanced panel (sum_t s_t per tile -> scol[b][tile]); 1: the same with the DFM_W2_ABL diagnostics compiled in;
// 2: panel with missing cells -- NaN operands count as 0 (b_t, s_t over the observed cells), per period s_t -> scol[b][t],
// n_t -> nobs[b][t] and, where cells are missing, log det R_t -> ldrow[b][t]; C_t of those periods: ct_miss_wide_kernel.
template <int R, int NX, int MODE>
__global__ __launch_bounds__(kW2Threads) void collapse_wide2_kernel(CollapseArgs a, const double* __restrict__ Wall,
                                                                   const double* __restrict__ rinvAll, int npad, int* ctr,
                                                                   int ntile, int xcd_map, int abl_, int rd, int lamd_) {
    constexpr bool DIAG = MODE == 1, MISS = MODE == 2;
    // lamd: the B operands are the LOADINGS themselves (DMA'd straight from CollapseArgs::Lam, the half swap of odd series applied by
    // the source addresses) and the A operand is x / R -- the product x (1 / R) the s_t term forms anyway; wide_prep then writes no W
    const bool lamd = !MISS && R == 32 && lamd_ != 0;
    static_assert((R == 32 && NX >= 1 && NX <= 4) || (R == 16 && NX == 0), "R = 32: 1..4 column groups past the first 16; R = 16: none");
    using GEO = W2Geo<R>;
    constexpr int N4 = NX < 4 ? NX : 0;                       // 4x4x4 groups
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int abl = DIAG ? abl_ : 0;                          // DFM_W2_ABL (diagnostics, wrong results): compiled out of the product kernel
    const int N = a.N, T = a.T, B = a.B;
    const int nch = (N + kW2Chunk - 1) / kW2Chunk;
    double* redS = reinterpret_cast<double*>(smem + kW2NBuf * kW2StageB);   // [2][8]: s_t partials of the waves, by tile parity
    volatile int* itemq = reinterpret_cast<volatile int*>(redS + 16);       // [kW2Ring]: published items (-1 = the queue is empty)
    const unsigned itemq_lds = (unsigned)(size_t)(lds_char_ptr_w)(smem) + kW2NBuf * kW2StageB + 16 * 8;
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(redS + 32);   // DIAG, abl & 256: [64 stages][4] of workgroup 0, wave 0
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = (int)blockIdx.x & 7;
    auto valid_item = [&](int kk) { return (xcd_map ? (kk / ntile) * 8 + xcd : kk / ntile) < B; };
    auto decode = [&](int kk, int& b, int& tile) {
        b = xcd_map ? (kk / ntile) * 8 + xcd : kk / ntile;
        tile = kk % ntile;
    };
    auto read_item = [&](int idx) {
        return __builtin_amdgcn_readfirstlane(*(const volatile __attribute__((address_space(3))) int*)(size_t)(itemq_lds + 4u * (unsigned)(idx & (kW2Ring - 1))));
    };

    // Every role passes the SAME barriers: one after the set-up, one per stage that exists, one at the end.
    if (wave == kW2Compute + kW2Producers) {
        // ---- scheduler wave: publishes item n before the barrier after which a producer can first ask for it (the issue cursor
        // enters item (q + 3) / nch after barrier q), half a tile early so that the atomic's latency is never waited for.
        int fetched = 0, nvalid = 0;
        bool ended = false;
        auto publish_upto = [&](int target) {
            while (fetched <= target && !ended) {
                int kk = 0;
                if (lane == 0) kk = atomicAdd(&ctr[xcd_map ? xcd : 0], 1);
                kk = __builtin_amdgcn_readfirstlane(kk);
                const bool ok = valid_item(kk);
                if (lane == 0) itemq[fetched & (kW2Ring - 1)] = ok ? kk : -1;
                ++fetched;
                if (ok) ++nvalid; else ended = true;
            }
        };
        publish_upto((3 + nch / 2) / nch);
        __syncthreads();
        for (int q = 0; q / nch < nvalid; ++q) {              // stage q exists (its item was published long ago)
            publish_upto((q + 3 + nch / 2) / nch);
            __syncthreads();
        }
        __syncthreads();
        return;
    }

    // no NaN bit patterns in columns / rows the DMAs of a partial stage do not write: zero the buffers once
    for (int e = tid; e < kW2NBuf * (int)kW2StageB / 8; e += 64 * (kW2Compute + kW2Producers)) reinterpret_cast<double*>(smem)[e] = 0.0;
    __syncthreads();                                          // (and the scheduler's first items are published)

    if (wave >= kW2Compute) {
        // ---- producer waves: ALL the LDS-DMA of the kernel.  A DMA instruction costs its wave 60-190 issue cycles; six per stage
        // in each streaming wave made the consumer loop issue-bound (1.3 us per stage against 0.27 us of MFMA).  Producer p
        // brings in rows 32 p .. 32 p + 31 of the stage (8 DMAs of 4 rows x 256 bytes), 2 KB of the W block, and (p = 0) the
        // stage's 1 / R: a FIXED number of DMAs per stage -- the row DMAs have lane 0 active in every stage, the others are
        // unconditional -- so "stage q has landed" is a counted wait that leaves stage q + 1 outstanding.  (A wave that skipped
        // a DMA of the partial last stage would wait for too few of the stage before it.)
        const int pw = wave - kW2Compute;
        __builtin_amdgcn_s_setprio(3);                        // few instructions, all on the critical path of the stream
        const unsigned rowB = (unsigned)N * 8u;
        const unsigned wbytes = (unsigned)N * R * 8u;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_w)(smem));
        auto issue_dma = [&](int b, int t0, int ch, int bsel) {
            if (abl & 32) return;                             // (diagnostics: compute only)
            const char* Xb = reinterpret_cast<const char*>(a.panel + (size_t)b * T * N);
            const char* Wb = lamd ? reinterpret_cast<const char*>(a.Lam + (size_t)b * N * R) : reinterpret_cast<const char*>(Wall + (size_t)b * N * R);
            const char* Rb = reinterpret_cast<const char*>(rinvAll + (size_t)b * npad);
            const unsigned sbase = lds0 + (unsigned)bsel * kW2StageB;
            const int h = lane >> 4;                          // row of the group; its lane u fetches piece (u - h) mod 16
            const unsigned colB = (unsigned)ch * (kW2Chunk * 8u) + 16u * (unsigned)((lane - h) & 15);
            const bool act = colB < rowB;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int t = t0 + 32 * pw + 4 * k + h;
                t = t < T ? t : T - 1;
                const char* src = Xb + (size_t)t * rowB + colB;
                const unsigned dst = __builtin_amdgcn_readfirstlane(sbase + (unsigned)(pw * 8 + k) * kW2GroupB);
                if (act) dma16wp(src, dst);
            }
#pragma unroll
            for (int u = 0; u < GEO::WPieces; ++u) {   // past the end of W -- the last, partial stage -- the lanes re-read its last 16 bytes; those rows only ever meet a zeroed A
                const unsigned o = (unsigned)ch * GEO::WB + (unsigned)(GEO::WPieces * pw + u) * 1024u + 16u * lane;
                const unsigned dst = __builtin_amdgcn_readfirstlane(sbase + kW2PanelB + (unsigned)(GEO::WPieces * pw + u) * 1024u);
                unsigned oo = o < wbytes ? o : wbytes - 16u;
                if (lamd) oo ^= (oo & 256u) >> 1;           // odd series (256-byte rows): 16-byte unit u <- u ^ 8, i.e. byte offset ^ 128
                dma16w(Wb + oo, dst);
            }
            if (pw == 0) {                  // 1 / R of the stage's 32 series (the table is padded with zeros to a whole stage)
                const unsigned dst = __builtin_amdgcn_readfirstlane(sbase + kW2PanelB + kW2WB);
                if (lane < 16) dma16w(Rb + (size_t)ch * (kW2Chunk * 8u) + 16u * lane, dst);
                if (MISS) {                 // ... and log R (the table follows the 1 / R table of the whole batch)
                    const char* Lb = Rb + (size_t)B * npad * 8u;
                    if (lane < 16) dma16w(Lb + (size_t)ch * (kW2Chunk * 8u) + 16u * lane, dst + kW2Chunk * 8u);
                }
            }
        };
        int ii = 0, ich = 0, ikk = read_item(0), ib = 0, itile = 0;   // issue cursor: the next stage to request
        if (ikk >= 0) decode(ikk, ib, itile);
        auto issue_next = [&](int bsel) {
            issue_dma(ib, itile * kW2Rows, ich, bsel);
            if (++ich == nch) {
                ich = 0;
                ikk = read_item(++ii);
                if (ikk >= 0) decode(ikk, ib, itile);
            }
        };
        bool more = ikk >= 0, v1 = false;                     // stage q exists; stage q + 1 exists (and has been requested)
        if (more) {
            issue_next(0);
            v1 = ikk >= 0;
            if (v1) issue_next(1);
        }
        int bsel = 0;
        while (more) {
            if (!v1) wait_all_w();
            else if (pw == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GEO::PerStage + (MISS ? 2 : 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GEO::PerStage) : "memory");
            __syncthreads();                                  // stage q has landed; every consumer is done with stage q - 1
            const bool v2 = ikk >= 0;                         // stage q + 2 exists: into the buffer stage q - 1 used
            if (v2) issue_next(bsel == 0 ? 2 : bsel - 1);
            bsel = bsel == 2 ? 0 : bsel + 1;
            more = v1;
            v1 = v2;
        }
        __syncthreads();
        return;
    }

    // ---- consumer waves: wave w owns periods 16 w .. 16 w + 15 of the tile
    int qstamp = 0;
    auto stamp = [&](int k) {
        if constexpr (DIAG) {
            if ((abl & 256) && blockIdx.x == 0 && threadIdx.x == 0 && qstamp < 64) stamps[qstamp * 4 + k] = __builtin_amdgcn_s_memrealtime();
        }
    };
    const int k4 = lane >> 4, c16 = lane & 15;
    const unsigned ldsc = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_w)(smem));
    int ci = 0, ch = 0, bsel = 0, cb = 0, ctile = 0;          // consume cursor
    bool more;
    {
        const int kk0 = read_item(0);
        more = kk0 >= 0;
        if (more) decode(kk0, cb, ctile);
    }
    typedef double w2_v4 __attribute__((ext_vector_type(4)));
    w2_v4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};   // factors 0..15: even / odd steps
    w2_v4 accb = {0.0, 0.0, 0.0, 0.0};                                // NX == 4: factors 16..31
    double acc4[N4 > 0 ? N4 : 1];
#pragma unroll
    for (int x = 0; x < (N4 > 0 ? N4 : 1); ++x) acc4[x] = 0.0;
    double qs = 0.0, nmis = 0.0, lmis = 0.0;                  // MISS: missing cells of this lane's row (count, sum of log R)
    // A operand of this lane: row 16 w + c16 of the tile -> group 4 w + c16 / 4, row h = c16 % 4 of the group; series j of the
    // stage at byte (8 j + 16 h) mod 256 of the row (only steps 6 and 7 can wrap)
    const unsigned ah = (unsigned)(c16 & 3);
    const unsigned arow = (unsigned)(4 * wave + (c16 >> 2)) * kW2GroupB + ah * 256u;
    const unsigned arot = 8u * k4 + 16u * ah;
    const unsigned a_lo = arow + arot;                        // steps 0 .. 5: + 32 s
    const unsigned a_6 = arow + ((arot + 192u) & 255u), a_7 = arow + ((arot + 224u) & 255u);
    // B operands: W[series 4 s + k4][f], odd series with their 16-column halves swapped
    const unsigned bsw = R == 32 ? 16u * (k4 & 1) : 0u;       // (R = 16: rows of 128 bytes, slot 16 k4 + c16 -- no swizzle needed)
    const unsigned b16 = (unsigned)k4 * (R * 8u) + (((unsigned)c16) ^ bsw) * 8u;             // f = c16
    const unsigned b16b = (unsigned)k4 * (R * 8u) + ((16u + c16) ^ bsw) * 8u;                // f = 16 + c16 (NX == 4)
    const unsigned b4 = (unsigned)k4 * (R * 8u) + ((16u + (lane & 3)) ^ bsw) * 8u;           // f = 16 + q (+ 4 x)

    // finished tile: its results wait in registers / redS until the next barrier has passed
    int pend_b = -1, pend_t0 = 0, pend_par = 0, pend_tile = 0;
    w2_v4 pacc = {0.0, 0.0, 0.0, 0.0}, paccb = {0.0, 0.0, 0.0, 0.0};
    double pacc4[N4 > 0 ? N4 : 1];
    auto flush_pending = [&]() {
        if (pend_b < 0) return;
        // row stride of b_t: rd (= R except for the narrow states on R = 16), or -- R = 32, balanced -- CollapseArgs::bst
        const int bs = (R == 32 && !MISS && a.bst > 0) ? a.bst : rd;
        double* out = a.bcol + ((size_t)pend_b * T + pend_t0 + 16 * wave) * bs;
#pragma unroll
        for (int v = 0; v < 4; ++v) {                         // 16x16x4: D[(l / 16) + 4 v][l % 16]
            const int row = k4 + 4 * v;
            if (pend_t0 + 16 * wave + row < T) {
                if (R == 32 || c16 < rd) out[(size_t)row * bs + c16] = pacc[v];
                if (NX == 4 && 16 + c16 < bs) out[(size_t)row * bs + 16 + c16] = paccb[v];
            }
        }
        if (R == 32 && NX < 4) {                              // 4x4x4: block (l / 4) % 4, D[row = l / 16][col = l % 4] -> period 4 blk + l / 16
            const int row = 4 * ((lane >> 2) & 3) + k4;
            if (pend_t0 + 16 * wave + row < T) {
#pragma unroll
                for (int x = 0; x < 4; ++x)                   // (the padding columns past 16 + 4 NX are zero)
                    if (16 + 4 * x < bs) out[(size_t)row * bs + 16 + 4 * x + (lane & 3)] = x < N4 ? pacc4[x < N4 ? x : 0] : 0.0;
            }
        }
        if (!MISS && tid == 0) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < kW2Compute; ++w) tot += redS[pend_par * 8 + w];
            a.scol[(size_t)pend_b * T + pend_tile] = tot;
            if (tot != tot) atomicOr(a.status, 1);            // NaN in the panel on the balanced path
        }
        pend_b = -1;
    };

    while (more) {
        stamp(0);
        __syncthreads();                                      // stage q is ready (the producers waited for it)
        stamp(1);
        flush_pending();
        stamp(2);
        const int bsel_cur = bsel;
        const int cfirst = ch * kW2Chunk + k4;                // series of this lane's k in step 0
        if (!(abl & 4)) {
            // Operands four steps ahead of their MFMAs.  The reads are VOLATILE so that the compiler neither merges pairs of them
            // into ds_read2_b64 (half the LDS rate, 32-bank groups of 16 lanes: the layouts above are conflict-free for
            // ds_read_b64's two groups of 32 lanes over 64 banks) nor sinks them back next to their use.
            double av[kW2Steps], bv[kW2Steps], ri[kW2Steps], lr[MISS ? kW2Steps : 1], bvb[NX == 4 ? kW2Steps : 1], b4v[N4 > 0 ? N4 : 1][kW2Steps];
            const unsigned st = ldsc + (unsigned)bsel_cur * kW2StageB, wo = st + kW2PanelB, ro = wo + kW2WB + 8u * k4;
            auto load_step = [&](int s) {
                av[s] = lds_read64(st + (s == 6 ? a_6 : s == 7 ? a_7 : a_lo + 32u * s));
                bv[s] = lds_read64(wo + b16 + s * (4 * R * 8));
                if (NX == 4) bvb[NX == 4 ? s : 0] = lds_read64(wo + b16b + s * (4 * R * 8));
#pragma unroll
                for (int x = 0; x < N4; ++x) b4v[x][s] = lds_read64(wo + b4 + s * (4 * R * 8) + x * 32);
                ri[s] = lds_read64(ro + 32u * s);
                if (MISS) lr[MISS ? s : 0] = lds_read64(ro + kW2Chunk * 8u + 32u * s);
            };
#pragma unroll
            for (int s = 0; s < 4; ++s) load_step(s);
#pragma unroll
            for (int s = 0; s < kW2Steps; ++s) {
                if (s + 4 < kW2Steps) load_step(s + 4);
                double a_ = (cfirst + 4 * s < N) ? av[s] : 0.0;   // the last stage may be partial: its stale columns / W rows count for nothing
                if (MISS) {                                   // a missing cell: no contribution to b_t and s_t; counted for n_t, log det R_t
                    const bool nanv = a_ != a_;
                    a_ = nanv ? 0.0 : a_;
                    nmis += nanv ? 1.0 : 0.0;
                    lmis += nanv ? lr[MISS ? s : 0] : 0.0;
                }
                const double ar = a_ * ri[s];                 // x / R
                const double am = lamd ? ar : a_;             // A operand: x / R against the loadings, x against W = lam / R
                if (s & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bv[s], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bv[s], acc0, 0, 0, 0);
                if (NX == 4) accb = __builtin_amdgcn_mfma_f64_16x16x4f64(am, bvb[NX == 4 ? s : 0], accb, 0, 0, 0);
#pragma unroll
                for (int x = 0; x < N4; ++x) acc4[x] = __builtin_amdgcn_mfma_f64_4x4x4f64(am, b4v[x][s], acc4[x], 0, 0, 0);
                qs = fma(ar, a_, qs);                         // s_t = sum_i x_it^2 / R_i from the same register
            }
        }
        stamp(3);
        ++qstamp;
        bsel = bsel == 2 ? 0 : bsel + 1;
        if (++ch == nch) {                                    // the tile is complete: park b_t and the s_t partial
            pend_b = cb; pend_t0 = ctile * kW2Rows; pend_tile = ctile; pend_par = ci & 1;
            pacc = acc0 + acc1; paccb = accb;
#pragma unroll
            for (int x = 0; x < (N4 > 0 ? N4 : 1); ++x) { pacc4[x] = acc4[x]; acc4[x] = 0.0; }
            acc0 = w2_v4{0.0, 0.0, 0.0, 0.0}; acc1 = acc0; accb = acc0;
            if (MISS) {                                       // per period: fold the four series classes k4 of the row
                qs += __shfl_xor(qs, 16, 64); qs += __shfl_xor(qs, 32, 64);
                nmis += __shfl_xor(nmis, 16, 64); nmis += __shfl_xor(nmis, 32, 64);
                lmis += __shfl_xor(lmis, 16, 64); lmis += __shfl_xor(lmis, 32, 64);
                const int t = pend_t0 + 16 * wave + c16;
                if (k4 == 0 && t < T) {
                    const size_t o = (size_t)cb * T + t;
                    a.scol[o] = qs;
                    a.nobs[o] = N - (int)nmis;
                    if (nmis > 0.0) a.ldrow[o] = a.ldfull[cb] - lmis;
                }
                nmis = 0.0; lmis = 0.0;
            } else {
                if (pend_t0 + 16 * wave + c16 >= T) qs = 0.0; // this lane's row is past the end of the sample (it repeats row T - 1)
                qs = wave_allsum(qs);
                if (lane == 0) redS[pend_par * 8 + wave] = qs;
            }
            qs = 0.0;
            ch = 0;
            const int ckk = read_item(++ci);                  // (published at least two barriers ago)
            more = ckk >= 0;
            if (more) decode(ckk, cb, ctile);
        }
    }
    __syncthreads();
    flush_pending();
    if constexpr (DIAG) {
        if ((abl & 256) && blockIdx.x == 0 && tid == 0) {     // 100 MHz ticks relative to the first stamp: barrier-begin, barrier-end, stores-end, compute-end
            for (int q = 0; q < (qstamp < 64 ? qstamp : 64); ++q)
                printf("W2STAMP %d %llu %llu %llu %llu\n", q, stamps[q * 4] - stamps[0], stamps[q * 4 + 1] - stamps[0],
                       stamps[q * 4 + 2] - stamps[0], stamps[q * 4 + 3] - stamps[0]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Panels with missing cells at Rp = 32: C_t = C - sum over the MISSING series i of lam_i lam_i' / R_i (packed lower triangle,
// 528 entries) for every period that has a missing cell.  One workgroup of 16 waves per (replicate, 16 periods): wave w owns
// period t0 + w and lane l the packed entries l, l + 64, ... (9 per lane).  A 16-bit mask per series says in which of the 16
// periods it is missing; the series stream through LDS in stages of 32 (W = lam / R and lam, 8 KB each) and a wave adds
// w_ij lam_ik of the series missing in ITS period -- one ballot per stage finds them, the work is 18 LDS reads and 9 FMAs per
// lane for one series in ten at 10 % missing.  (Thread = entry with all 16 periods per thread repeats the per-series control
// flow in 9 waves: 41 ms for config 4 against ~8 ms this way; the dense form -- (16 x N) masks times (N x 528) products on
// the matrix pipe -- is ~17 x the collapse's MFMA work.)
constexpr int kCtP = 16;
constexpr int kCtThreads = 64 * kCtP;
constexpr int kCtQ = (kW2R * (kW2R + 1) / 2 + 63) / 64;       // packed entries per lane: 9
__global__ __launch_bounds__(kCtThreads) void ct_miss_wide_kernel(CollapseArgs a, const double* __restrict__ Wall, int ntile16) {
    constexpr int R = kW2R, NP = R * (R + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, T = a.T;
    const int npad = ((N + kW2Chunk - 1) / kW2Chunk) * kW2Chunk;
    double* Ws = reinterpret_cast<double*>(smem);             // [32][32] W (stored layout: odd series with their halves swapped)
    double* Ls = Ws + kW2Chunk * R;                           // [32][32] lam
    unsigned short* mask = reinterpret_cast<unsigned short*>(Ls + kW2Chunk * R);   // [npad]
    unsigned* anyS = reinterpret_cast<unsigned*>(mask + npad);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = (int)blockIdx.x / ntile16, t0 = ((int)blockIdx.x % ntile16) * kCtP;
    const double* __restrict__ X = a.panel + (size_t)b * T * N;
    if (tid == 0) *anyS = 0u;
    __syncthreads();
    unsigned many = 0u;
    for (int i = tid; i < npad; i += kCtThreads) {
        unsigned m = 0u;
        if (i < N) {
#pragma unroll
            for (int t = 0; t < kCtP; ++t) {
                const int tt = t0 + t < T ? t0 + t : T - 1;
                const double x = X[(size_t)tt * N + i];
                m |= (x != x && t0 + t < T) ? (1u << t) : 0u;
            }
        }
        mask[i] = (unsigned short)m;
        many |= m;
    }
    if (many) atomicOr(anyS, many);
    __syncthreads();
    const unsigned tilemask = *anyS;                          // periods of the tile with a missing cell
    if (tilemask == 0u) return;
    const bool mine = (tilemask >> wave) & 1u;                // (wave-uniform) this wave's period has a missing cell
    // packed entries of this lane: v = lane + 64 q -> (j, k), k <= j; offsets into a stage row
    int oj[kCtQ], ok[kCtQ];
#pragma unroll
    for (int q = 0; q < kCtQ; ++q) {
        int v = lane + 64 * q;
        v = v < NP ? v : NP - 1;
        int j = 0;
        while ((j + 1) * (j + 2) / 2 <= v) ++j;
        oj[q] = j;
        ok[q] = v - j * (j + 1) / 2;
    }
    double E[kCtQ];
#pragma unroll
    for (int q = 0; q < kCtQ; ++q) E[q] = 0.0;
    const double* __restrict__ Wb = Wall + (size_t)b * N * R;
    const double* __restrict__ Lb = a.Lam + (size_t)b * N * R;
    const int nch = npad / kW2Chunk;
    for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();
        for (int q = tid; q < kW2Chunk * R; q += kCtThreads) {
            const int c = ch * kW2Chunk + q / R;
            Ws[q] = c < N ? Wb[(size_t)ch * kW2Chunk * R + q] : 0.0;
            Ls[q] = c < N ? Lb[(size_t)ch * kW2Chunk * R + q] : 0.0;
        }
        __syncthreads();
        if (!mine) continue;
        // the stage's series that are missing in this wave's period: one mask read per lane, one ballot, then only those
        unsigned long long bits = __ballot(lane < kW2Chunk && ((mask[ch * kW2Chunk + (lane & (kW2Chunk - 1))] >> wave) & 1u) != 0u);
#pragma unroll 1
        while (bits != 0ull) {
            const int ii = __builtin_ctzll(bits);
            bits &= bits - 1ull;
            const int sw = 16 * (ii & 1);                     // (stage start is a multiple of 32: parity of ii = parity of the series)
#pragma unroll
            for (int q = 0; q < kCtQ; ++q) E[q] = fma(Ws[ii * R + (oj[q] ^ sw)], Ls[ii * R + ok[q]], E[q]);
        }
    }
    if (mine && a.Ct == nullptr) {                            // the caller promised a balanced panel: flag it, keep valid memory
        if (lane == 0) atomicOr(a.status, 1);
    } else if (mine) {
        const int t = t0 + wave;
#pragma unroll
        for (int q = 0; q < kCtQ; ++q) {
            const int v = lane + 64 * q;
            if (v < NP) a.Ct[((size_t)b * T + t) * NP + v] = a.Cfull[(size_t)b * R * R + oj[q] * R + ok[q]] - E[q];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// ct_miss_wide2_kernel (round 3): the same C_t with (i) the stage loads of W / lam double-buffered through registers -- the
// first version waited a full L2 round trip per stage of 32 series, 32 times per tile -- and (ii) a 2-D register blocking of
// the rank-1 updates: the lower-triangular 4 x 4 blocks of the r x r matrix (r = the caller's factor count, 15 blocks at
// r = 20) are dealt to the lanes of a SLOT of 16 | 32 | 64 lanes, and the 4 | 2 | 1 slots of a wave take different missing
// series of the stage at the same time: per series and lane 4 LDS reads of 16 bytes and 16 FMAs instead of 18 reads of 8
// bytes and 9 FMAs on every lane (the packed-entry-per-lane form was LDS-bound: two operand reads per FMA).  The slots'
// partial sums meet in two butterfly steps per accumulator at the end of the tile.
// ------------------------------------------------------------------------------------------------------------------
// Round 4 (third version).  Measured on the round-3 kernel: 6.0 ms per config-4 batch -- a third of the whole pass once the
// recursion moved to the matrix pipe -- and neither arithmetic (48 M FMAs per replicate: 0.1 ms of a CU) nor HBM (4.1 GB of panel
// + 2.2 GB of output: 1.5 ms) explains it: one workgroup per CU (128 KB of LDS) walks 32 stage hand-overs per tile of 16 periods,
// each waiting ~2 us for an L2 -> LDS DMA with two stages in flight, in front of ~3 missing series per wave and stage; the mask
// build in front of the stage loop and the output behind it are serial phases of the same workgroup.  Now:
//   * PPW = 2 periods per wave (a tile = 32 periods): half the hand-overs and half the re-streaming per period;
//   * a stage = 64 series of W ONLY (16 KB, one DMA per wave): lam_i = w_i R_i with R in LDS -- the loadings are not streamed a
//     second time (L2 -> LDS traffic 16 GB -> 4 GB per batch), and a stage carries twice the series: 16 hand-overs per tile;
//   * the first stages are requested BEFORE the mask build, so their latency hides behind the panel reads;
//   * output rows are the packed leading ct_r x ct_r block when the recursion kernel reads that (a.ct_r > 0: 210 instead of 528
//     doubles per period at r = 20 -- 0.86 instead of 2.2 GB written, and read back).
template <int PPW, int NBUF>
__global__ __launch_bounds__(kCtThreads) void ct_miss_wide2_kernel(CollapseArgs a, const double* __restrict__ Wall, int ntile, int r) {
    constexpr int R = kW2R, NPfull = R * (R + 1) / 2;
    constexpr int P = kCtP * PPW;                             // periods of a tile
    constexpr int SS = 64;                                    // series per stage
    constexpr unsigned kPieceB = 1088, kStageB = 16 * kPieceB;   // a stage: 16 pieces (4 series rows of 256 bytes) 1024 + 64 bytes apart
    // NBUF stage buffers: NBUF - 1 stages in flight behind the one in use (5 where the compact output rows leave the LDS for them)
    constexpr unsigned kStagesB = (unsigned)NBUF * kStageB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, T = a.T;
    const int npad = ((N + SS - 1) / SS) * SS;
    double* Rs = reinterpret_cast<double*>(smem + kStagesB);                              // [npad] R_i
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(Rs + npad);         // [npad]: bit t = cell (t0 + t, i) is missing
    unsigned long long* anyS = mask + npad;
    const int ctr = a.ct_r > 0 ? a.ct_r : R;                  // rows / columns of the packed output block
    const int NPo = ctr * (ctr + 1) / 2;
    double* CsAll = reinterpret_cast<double*>(anyS + 2);      // [16 waves][NPo] output rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = (int)blockIdx.x / ntile, t0 = ((int)blockIdx.x % ntile) * P;
    const double* __restrict__ X = a.panel + (size_t)b * T * N;
    const char* __restrict__ Wb = reinterpret_cast<const char*>(Wall + (size_t)b * N * R);
    const int nch = npad / SS;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_w)(smem));
    const unsigned wbytes = (unsigned)N * R * 8u;
    auto issue = [&](int ch, int buf) {                        // wave w moves piece w of stage ch: ONE global_load_lds_dwordx4
        unsigned o = (unsigned)ch * (SS * R * 8u) + (unsigned)wave * 1024u + 16u * lane;
        o = o < wbytes ? o : wbytes - 16u;                    // the last stage may be partial: its rows past N are never selected
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf * kStageB + (unsigned)wave * kPieceB);
        dma16w(Wb + o, dst);
    };
#pragma unroll
    for (int u = 0; u < NBUF - 1; ++u)
        if (u < nch) issue(u, u);
    if (tid == 0) *anyS = 0ull;
    for (int i = tid; i < npad; i += kCtThreads) Rs[i] = i < N ? a.Rv[(size_t)b * N + i] : 1.0;
    __syncthreads();                                          // (drains this wave's DMAs too: counted waits start from zero below)
    unsigned long long many = 0ull;
    for (int i = tid; i < npad; i += kCtThreads) {
        unsigned long long m = 0ull;
        if (i < N) {
#pragma unroll 8
            for (int t = 0; t < P; ++t) {
                const int tt = t0 + t < T ? t0 + t : T - 1;
                const double x = X[(size_t)tt * N + i];
                m |= (x != x && t0 + t < T) ? (1ull << t) : 0ull;
            }
        }
        mask[i] = m;
        many |= m;
    }
    if (many) atomicOr(anyS, many);
    __syncthreads();
    const unsigned long long tilemask = *anyS;                // periods of the tile with a missing cell
    if (tilemask == 0ull) return;
    const int tw0 = wave * PPW;                               // this wave's periods: t0 + tw0 + pp
    const unsigned wmask = (unsigned)((tilemask >> tw0) & ((1ull << PPW) - 1ull));   // (wave-uniform) which of them have a missing cell
    const bool mine = wmask != 0u;
    // blocks: nb x nb of 4 x 4 over the r x r matrix, lower triangle; LS lanes per slot
    const int nb = (r + 3) / 4, nlt = nb * (nb + 1) / 2;
    const int LS = nlt <= 16 ? 16 : (nlt <= 32 ? 32 : 64), NS = 64 / LS;
    const int slot = lane / LS, bl = lane % LS;
    const bool act = bl < nlt;
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= (act ? bl : 0)) ++bi;
    const int bj = (act ? bl : 0) - bi * (bi + 1) / 2;
    double E[PPW][4][4];
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp)
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) E[pp][x][y] = 0.0;
    // NBUF stage buffers, counted waits (the compiler's own scoreboard put a vmcnt(0) in front of every register-staged LDS
    // write, i.e. one L2 round trip per stage); rows of different pieces start on different banks (the slots read different
    // series at the same column block).  Stages 0 .. NBUF - 2 were requested before the mask build and have landed (the
    // barriers above drain vmcnt); from then on stage ch + NBUF - 1 is requested when stage ch is taken up, so at most
    // NBUF - 2 YOUNGER requests of this wave are in flight when it needs stage ch.
    for (int ch = 0; ch < nch; ++ch) {
        if (ch >= NBUF - 1) {
            const int younger = nch - 1 - ch < NBUF - 2 ? nch - 1 - ch : NBUF - 2;
            switch (younger) {                                    // (wave-uniform; the immediate must be a constant)
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // everybody's piece has landed; every wave is done with stage ch - 1
        if (ch + NBUF - 1 < nch) issue(ch + NBUF - 1, (ch + NBUF - 1) % NBUF);   // into the buffer stage ch - 1 used (ch = 0: a fresh one)
        if (!mine) continue;
        const char* st = smem + (size_t)(ch % NBUF) * kStageB;
        const unsigned long long mrow = mask[ch * SS + lane];  // the lane's series of this stage
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) {
            unsigned long long bits = __ballot(((mrow >> (tw0 + pp)) & 1ull) != 0ull);
#pragma unroll 1
            while (bits != 0ull) {
                int my = -1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                 // the next NS missing series of the stage, one per slot
                    if (q < NS && bits != 0ull) {
                        const int ii = __builtin_ctzll(bits);
                        bits &= bits - 1ull;
                        my = (slot == q) ? ii : my;
                    }
                }
                if (act && my >= 0) {
                    const int sw = 16 * (my & 1);             // (stage start is even: parity of ii = parity of the series; odd rows of W have their halves swapped)
                    const char* row = st + (unsigned)(my >> 2) * kPieceB + (unsigned)(my & 3) * 256u;
                    const double2* wp = reinterpret_cast<const double2*>(row + 8 * ((4 * bi) ^ sw));
                    const double2* lp = reinterpret_cast<const double2*>(row + 8 * ((4 * bj) ^ sw));
                    const double ri = Rs[ch * SS + my];
                    const double2 w01 = wp[0], w23 = wp[1], l01 = lp[0], l23 = lp[1];
                    const double wv[4] = {w01.x, w01.y, w23.x, w23.y};
                    const double lv[4] = {l01.x * ri, l01.y * ri, l23.x * ri, l23.y * ri};   // lam_i = w_i R_i
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y = 0; y < 4; ++y) E[pp][x][y] = fma(wv[x], lv[y], E[pp][x][y]);
                }
            }
        }
    }
    if (!mine) return;
    const double* Cf = a.Cfull + (size_t)b * R * R;
    double* Cs = CsAll + (size_t)wave * NPo;                  // (its own LDS: slower waves still read the stage buffers)
    // rows past the blocks (padding inside the output block): no series loads on them, C_t = C -- the same for every period
    const int j0 = 4 * nb < ctr ? 4 * nb : ctr;
    for (int v = j0 * (j0 + 1) / 2 + lane; v < NPo; v += 64) {
        int jj = j0;
        while ((jj + 1) * (jj + 2) / 2 <= v) ++jj;
        Cs[v] = Cf[jj * R + (v - jj * (jj + 1) / 2)];
    }
    double cfv[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int jj = 4 * bi + x, kk = 4 * bj + y;
            cfv[x][y] = (act && kk <= jj && jj < ctr) ? Cf[jj * R + kk] : 0.0;
        }
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) {
        if (!((wmask >> pp) & 1u)) continue;                  // (wave-uniform) a period without a missing cell keeps Cfull
        // fold the slots (lanes bl, bl + LS, ...).  The period's packed C_t is assembled in LDS and leaves as whole 16-byte
        // pieces, 1 KB per store instruction: written straight from the blocks it was 16 scattered 8-byte stores per lane --
        // 270 M partial-sector writes per config-4 batch, which (not the arithmetic) then bounded the kernel.
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                double e = E[pp][x][y];
                if (NS >= 2) e += __shfl_xor(e, 32, 64);
                if (NS >= 4) e += __shfl_xor(e, 16, 64);
                const int jj = 4 * bi + x, kk = 4 * bj + y;
                if (act && slot == 0 && kk <= jj && jj < ctr) Cs[jj * (jj + 1) / 2 + kk] = cfv[x][y] - e;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (a.Ct == nullptr) {                                // the caller promised a balanced panel: flag it, keep valid memory
            if (lane == 0) atomicOr(a.status, 1);
        } else {
            double* Co = a.Ct + ((size_t)b * T + (t0 + tw0 + pp)) * NPo;      // (NPo is even: rows are 16-byte aligned)
            for (int v = 2 * lane; v < NPo; v += 128) *reinterpret_cast<double2*>(Co + v) = *reinterpret_cast<const double2*>(Cs + v);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (the row is read before the next period overwrites it)
        __builtin_amdgcn_wave_barrier();
    }
    (void)NPfull;
}

// ------------------------------------------------------------------------------------------------------------------
// ct_miss_slice_kernel (round 5).  Three per-period designs were measured at config 4 (256 x 2000 periods, 100 of 1000 series
// missing per period; profiles/r05/ab_ct_miss.txt) before this one:
//   the staged kernel above (every series through LDS per tile of 32 periods, 16-wave barriers)                      4.3 ms
//   one wave per period, rows of W / lam gathered from L2 into registers (4 x 4 blocks, four series per iteration)    3.3 ms
//   the same with the rows moved by LDS-DMA, 4 | 6 | 8 instructions in flight                                   3.16 | 3.25 | 3.94 ms
// The counters of the gather version: FETCH_SIZE = the panel once (the rows hit in L2), 135 M L2 requests (17 GB), 1.6 G L1
// accesses; its phases by s_memrealtime: 9 us for the panel row, 10-12 us for 17 gathers, whatever the number in flight -- a
// period fetches 200 cache lines of rows beside the 64 of its panel row, and the CUs take ~20 GB/s of lines each, the rate the
// streaming collapse runs at as well.  So the rows must not travel per period at all:
//   * V = lam / sqrt(R) (wide_prep_kernel): the deficit of a period is sum v_i v_i' over its missing series;
//   * V is cut into SLICES of series that fit the LDS of a CU beside the waves' buffers (r = 20: 500 series = 80 KB); one launch
//     per slice, one persistent workgroup of 16 waves per (replicate, time chunk) that copies its slice into LDS ONCE and then
//     walks its periods -- wave w the periods w, w + 16, ... -- without another workgroup-level step;
//   * per period a wave reads its part of the panel row (the next period's is in flight meanwhile), compacts the missing series
//     of the slice into a list (ballot + prefix count) and takes the rows from LDS, nothing through the L1;
//   * every lane owns a different BH x BW block of the r x r matrix and the wave takes ONE series at a time (SPT of them per trip:
//     one broadcast read of their indices, 2 SPT row reads issued together): nothing to add up across lanes at the end;
//   * the first slice writes C_t = C - E_0 for every period with a missing cell ANYWHERE (n_t < N from the collapse), the later
//     ones subtract their E_s from the stored row (asked for in front of the arithmetic).  Launches run back to back on the
//     stream: the order of the subtractions is fixed.
//   * periods are handed out by a counter in LDS (the SIMD issues its oldest wave first: dealt round-robin, the youngest of its
//     four waves took 7.4 us per period where the oldest took 4.6, and the workgroup waited for it).
// 1.73 ms for the two launches of config 4.  Without the panel read and the output an earlier state took 2.0 of 2.2 ms: the time
// is inside the CU -- the waves' dependent trips (list, index read, row reads, multiply-adds) at four waves per SIMD.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kCsWaves = 16;
// BH x BW: the block a lane owns -- 2 x 2 up to r = 20 (55 lanes), 2 x 4 up to r = 28 (56), 4 x 4 beyond (36).
template <int BH, int BW>
__global__ __launch_bounds__(64 * kCsWaves) void ct_miss_slice_kernel(CollapseArgs a, const double* __restrict__ Vall, int r, int s0, int ns, int first, int TC) {
    constexpr int R = kW2R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = a.N, T = a.T;
    const int ctr = a.ct_r > 0 ? a.ct_r : R;                  // rows / columns of the packed output block
    const int NPo = ctr * (ctr + 1) / 2, NPe = (NPo + 1) & ~1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = (int)blockIdx.x / TC, tc = (int)blockIdx.x % TC;
    const int Tc = (T + TC - 1) / TC;
    const int t_lo = tc * Tc, t_hi = t_lo + Tc < T ? t_lo + Tc : T;
    const int rb = ((r + BW - 1) / BW) * BW, recB = 8 * rb;   // a row of the slice in LDS: the leading columns of V (BH divides BW)
    // LDS: the slice + one row of zeros | packed C (first slice only) | per wave: the packed deficit row | per wave: the list
    constexpr int SPT = BH * BW == 4 ? 8 : (BH * BW == 8 ? 4 : 2);   // series per trip (registers: SPT (BH + BW) operands)
    const int nspad = ((ns + 63) / 64) * 64;
    const int lcap = nspad + 8;                               // (+ the padding of the last trip)
    char* Vs = smem;
    double* Cfp = reinterpret_cast<double*>(smem + (size_t)(ns + 1) * recB);
    double* Cs = Cfp + NPe + (size_t)wave * NPe;
    unsigned short* list = reinterpret_cast<unsigned short*>(Cfp + (size_t)(1 + kCsWaves) * NPe) + (size_t)wave * lcap;
    int* next_t = reinterpret_cast<int*>(reinterpret_cast<unsigned short*>(Cfp + (size_t)(1 + kCsWaves) * NPe) + (size_t)kCsWaves * lcap);   // (lcap is a multiple of 8: aligned)
    const unsigned list0 = (unsigned)(size_t)(lds_char_ptr_w)(reinterpret_cast<char*>(list));
    {
        const int pr = rb / 2;                                // 16-byte pieces per row
        const double* Vb = Vall + ((size_t)b * N + s0) * R;
        for (int g = tid; g < (ns + 1) * pr; g += 64 * kCsWaves) {
            const int i = g / pr, pc = g % pr;
            *reinterpret_cast<double2*>(Vs + (size_t)i * recB + 16 * pc) =
                i < ns ? *reinterpret_cast<const double2*>(Vb + (size_t)i * R + 2 * pc) : make_double2(0.0, 0.0);
        }
        const double* Cf = a.Cfull + (size_t)b * R * R;
        for (int v = tid; v < NPe; v += 64 * kCsWaves) {
            int q = 0;
            while ((q + 1) * (q + 2) / 2 <= v) ++q;
            Cfp[v] = v < NPo ? Cf[q * R + (v - q * (q + 1) / 2)] : 0.0;
        }
        for (int v = lane; v < NPe; v += 64) Cs[v] = 0.0;     // (entries outside the blocks stay zero: no loadings there)
        if (tid == 0) *next_t = t_lo + kCsWaves;              // (the first kCsWaves periods are the waves' own)
    }
    __syncthreads();                                          // the only workgroup-level step
    // this lane's block: rows BH ba .., columns BW bc .. (the blocks that meet the lower triangle, row by row)
    int ba = 0, bc = 0;
    bool act = false;
    {
        int cnt = 0;
        const int nra = (r + BH - 1) / BH;
        for (int aa = 0; aa < nra && !act; ++aa)
            for (int cc = 0; BW * cc <= BH * aa + BH - 1 && BW * cc < r; ++cc) {
                if (cnt == lane) { ba = aa; bc = cc; act = true; break; }
                ++cnt;
            }
    }
#if defined(DFM_CS_ABL) && DFM_CS_ABL == 1
    const unsigned offr = 16u * (lane & 7), offc = 16u * ((lane >> 3) & 7);     // development (wrong results): fewer lanes per address
#elif defined(DFM_CS_ABL) && DFM_CS_ABL == 2
    const unsigned offr = 0u, offc = 16u;                                       // development (wrong results): every lane the same two addresses
#else
    const unsigned offr = 8u * BH * ba, offc = 8u * BW * bc;
#endif
    const unsigned vs0 = (unsigned)(size_t)(lds_char_ptr_w)(Vs);
    const double* __restrict__ X = a.panel + (size_t)b * T * N + s0;
    const int* __restrict__ nobs = a.nobs + (size_t)b * T;
    constexpr int XU = 8;                                     // panel loads in flight per lane (512 series per batch)
    double xn[XU];
    auto load_row = [&](int t, int k0) {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = k0 + 64 * u + lane;
            xn[u] = (t < t_hi && i < ns) ? X[(size_t)t * N + i] : 0.0;
        }
    };
    load_row(t_lo + wave, 0);
    const bool even = (NPo & 1) == 0;
    // Periods are handed out by a counter in LDS, not dealt round-robin: the SIMD issues its oldest wave first, the youngest of its
    // four took 7.4 us per period where the oldest took 4.6, and with equal shares the workgroup waited for the slow ones
    // (0.92 ms per launch where the mean rate gives 0.73).
    for (int t = t_lo + wave, tn = 0; t < t_hi; t = tn) {
        // asked for in front of the arithmetic, used behind it: n_t, and the stored row a later slice subtracts from (two 16-byte
        // pieces per lane cover 256 entries; wider rows take the rest in the loop at the end)
        const int nob = nobs[t];
        double* Co = a.Ct ? a.Ct + ((size_t)b * T + t) * NPo : nullptr;
        double2 cpre[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int v = 2 * lane + 128 * k;
            cpre[k] = (!first && even && Co != nullptr && v < NPo) ? *reinterpret_cast<const double2*>(Co + v) : make_double2(0.0, 0.0);
        }
        double E[BH][BW];
#pragma unroll
        for (int x = 0; x < BH; ++x)
#pragma unroll
            for (int y = 0; y < BW; ++y) E[x][y] = 0.0;
        // ---- the period's missing series of this slice: a list in the wave's LDS (ballot + prefix count), padded with the row of zeros
        int n = 0;
        for (int k0 = 0; k0 < nspad; k0 += 64 * XU) {
            double xv[XU];
#pragma unroll
            for (int u = 0; u < XU; ++u) xv[u] = xn[u];
            if (k0 + 64 * XU < nspad) load_row(t, k0 + 64 * XU);   // (slices wider than a batch: the next batch of this row)
            else {                                            // the next period's first batch: in flight under this period's arithmetic
                int got = 0;
                if (lane == 0) got = atomicAdd(next_t, 1);
                tn = __builtin_amdgcn_readfirstlane(got);
                load_row(tn, 0);
            }
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                const bool m = xv[u] != xv[u];
                const unsigned long long bal = __ballot(m);
                if (m) list[n + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = (unsigned short)(k0 + 64 * u + lane);
                n += __popcll(bal);
            }
        }
        if (lane < SPT) list[n + lane] = (unsigned short)ns;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // ---- SPT series per trip, the same for every lane: their indices in one broadcast read, their rows in 2 SPT reads issued
        // together (the empty statements pin that order: left alone, the compiler re-uses one register set and waits per series)
        for (int e0 = 0; e0 < n; e0 += SPT) {
            unsigned idx[SPT];
            if (SPT == 8) {
                const w2_u4 L = *(lds_cu4_ptr_w)(size_t)(list0 + 2u * (unsigned)e0);
#pragma unroll
                for (int k = 0; k < 4; ++k) { idx[2 * k] = L[k] & 0xffffu; idx[2 * k + 1] = L[k] >> 16; }
            } else if (SPT == 4) {
                const w2_u2 L = *(lds_cu2_ptr_w)(size_t)(list0 + 2u * (unsigned)e0);
#pragma unroll
                for (int k = 0; k < 2; ++k) { idx[2 * k] = L[k] & 0xffffu; idx[2 * k + 1] = L[k] >> 16; }
            } else {
                const unsigned L = *(lds_cu1_ptr_w)(size_t)(list0 + 2u * (unsigned)e0);
                idx[0] = L & 0xffffu; idx[SPT - 1] = L >> 16;
            }
            double vr[SPT][BH], vc[SPT][BW];
#pragma unroll
            for (int k = 0; k < SPT; ++k) {
                const unsigned ra = idx[k] * (unsigned)recB + vs0;
#pragma unroll
                for (int x = 0; x < BH; x += 2) {
                    const w2_v2 d = *(lds_cv2_ptr_w)(size_t)(ra + offr + 8u * x);
                    vr[k][x] = d[0]; vr[k][x + 1] = d[1];
                }
#pragma unroll
                for (int y = 0; y < BW; y += 2) {
                    const w2_v2 d = *(lds_cv2_ptr_w)(size_t)(ra + offc + 8u * y);
                    vc[k][y] = d[0]; vc[k][y + 1] = d[1];
                }
            }
#pragma unroll
            for (int k = 0; k < SPT; ++k) {
#pragma unroll
                for (int x = 0; x < BH; ++x) asm volatile("" : "+v"(vr[k][x]) : : "memory");
#pragma unroll
                for (int y = 0; y < BW; ++y) asm volatile("" : "+v"(vc[k][y]) : : "memory");
#pragma unroll
                for (int x = 0; x < BH; ++x)
#pragma unroll
                    for (int y = 0; y < BW; ++y) E[x][y] = fma(vr[k][x], vc[k][y], E[x][y]);
            }
        }
        const bool anymiss = nob != N;                        // (uniform) a cell of this period is missing somewhere in the row
        if (first ? !anymiss : n == 0) continue;              // a complete period keeps Cfull; a later slice with nothing to subtract
        if (a.Ct == nullptr) {                                // the caller promised a balanced panel: flag it, keep valid memory
            if (lane == 0) atomicOr(a.status, 1);
            continue;
        }
        // ---- the packed deficit row, then C_t = C - E (first slice) or C_t -= E
#pragma unroll
        for (int x = 0; x < BH; ++x)
#pragma unroll
            for (int y = 0; y < BW; ++y) {
                const int q = BH * ba + x, kk = BW * bc + y;
                if (act && kk <= q && q < ctr) Cs[q * (q + 1) / 2 + kk] = E[x][y];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (even) {                                           // (rows are 16-byte aligned)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int v = 2 * lane + 128 * k;
                if (v < NPo) {
                    const double2 e = *reinterpret_cast<const double2*>(Cs + v);
                    const double2 c = first ? *reinterpret_cast<const double2*>(Cfp + v) : cpre[k];
                    *reinterpret_cast<double2*>(Co + v) = make_double2(c.x - e.x, c.y - e.y);
                }
            }
            for (int v = 2 * lane + 256; v < NPo; v += 128) {
                const double2 e = *reinterpret_cast<const double2*>(Cs + v);
                const double2 c = first ? *reinterpret_cast<const double2*>(Cfp + v) : *reinterpret_cast<const double2*>(Co + v);
                *reinterpret_cast<double2*>(Co + v) = make_double2(c.x - e.x, c.y - e.y);
            }
        } else {
            for (int v = lane; v < NPo; v += 64) Co[v] = (first ? Cfp[v] : Co[v]) - Cs[v];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the row is read before the next period overwrites it)
        __builtin_amdgcn_wave_barrier();
    }
}

int collapse_wide2_tiles(int T) { return (T + kW2Rows - 1) / kW2Rows; }
// Rp = 16 | 32 with an even N; narrower states (computed 16 wide) only where the row ring of the MFMA collapse ends (8 N > 4 KB)
bool collapse_wide2_supported(int Rpad, int N) {
    if ((N % 2) != 0 || N < 2) return false;
    return Rpad == 32 || Rpad == 16 || (Rpad >= 2 && Rpad <= 8 && N * 8 > 4096);
}
static int w2_compute_width(int Rpad) { return Rpad < 16 ? 16 : Rpad; }
// workspace of the collapse: W [B][N][Rk] | 1 / R [B][npad] | log R [B][npad] | 8 queue counters  (Rk = max(Rp, 16))
size_t collapse_wide2_ws_bytes(int B, int N, int Rpad) {
    const size_t npad = (size_t)((N + kW2Chunk - 1) / kW2Chunk) * kW2Chunk;
    return ((size_t)B * N * w2_compute_width(Rpad) + 2 * (size_t)B * npad) * sizeof(double) + 64;
}

// Rp = 32, balanced: the collapse takes the loadings and 1 / R themselves (collapse_wide2_kernel `lamd`), wide_prep writes no W.
// DFM_WIDE_W=1 (diagnostics build): the W table as before.
static bool wide2_lam_direct(int Rpad, bool missing) {
    static const bool off = [] { const char* v = diag_env("DFM_WIDE_W"); return v && atoi(v) != 0; }();
    return !off && Rpad == 32 && !missing;
}

namespace {
struct W2Ws { double* W; double* rinv; double* logr; int* ctr; int npad; };
W2Ws w2_ws(const CollapseArgs& a, double* ws, int Rpad) {
    W2Ws w;
    w.npad = ((a.N + kW2Chunk - 1) / kW2Chunk) * kW2Chunk;
    w.W = ws;
    w.rinv = ws + (size_t)a.B * a.N * w2_compute_width(Rpad);
    w.logr = w.rinv + (size_t)a.B * w.npad;                   // (the kernel finds it behind the 1 / R table)
    w.ctr = reinterpret_cast<int*>(w.logr + (size_t)a.B * w.npad);
    return w;
}
template <int R, int NX, int MODE>
hipError_t launch_w2v(const CollapseArgs& a, const W2Ws& w, int G, size_t lds, int ntile, int xcd_map, int abl, hipStream_t s, int rd = R, int lamd = 0) {
    static LdsOptIn attr_done;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&collapse_wide2_kernel<R, NX, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((collapse_wide2_kernel<R, NX, MODE>), dim3((unsigned)G), dim3(kW2Threads), lds, s, a, w.W, w.rinv, w.npad, w.ctr, ntile, xcd_map, abl, rd, lamd);
    return hipGetLastError();
}
template <int NX>
hipError_t launch_w2(const CollapseArgs& a, const W2Ws& w, int G, size_t lds, int ntile, int xcd_map, int abl, hipStream_t s) {
    if (a.nobs != nullptr) return launch_w2v<32, NX, 2>(a, w, G, lds, ntile, xcd_map, 0, s);   // panel with missing cells
    const int lamd = wide2_lam_direct(32, false) ? 1 : 0;
    return abl ? launch_w2v<32, NX, 1>(a, w, G, lds, ntile, xcd_map, abl, s, 32, lamd) : launch_w2v<32, NX, 0>(a, w, G, lds, ntile, xcd_map, 0, s, 32, lamd);
}
}  // namespace

hipError_t launch_wide_prep(const CollapseArgs& a, double* ws, int Rpad, hipStream_t s, int r, double* V) {
    note_kernel("wide_prep_kernel");
    const W2Ws w = w2_ws(a, ws, Rpad);
    (void)r;
    const bool ks = wide2_lam_direct(Rpad, a.nobs != nullptr);   // no W table for these
    if (Rpad <= 16) hipLaunchKernelGGL(wide_prep_kernel<16>, dim3(a.B), dim3(kPrepThreads), 0, s, a, w.W, w.rinv, w.logr, w.npad, w.ctr, Rpad, (double*)nullptr);
    else hipLaunchKernelGGL(wide_prep_kernel<32>, dim3(a.B), dim3(kPrepThreads), 0, s, a, ks ? nullptr : w.W, w.rinv, w.logr, w.npad, w.ctr, 32, V);
    return hipGetLastError();
}

// r = the caller's factor count (columns r .. Rpad - 1 of Lam are zero padding)
hipError_t launch_collapse_wide2(const CollapseArgs& a, double* ws, int Rpad, int r, int num_cu, hipStream_t s) {
    const W2Ws w = w2_ws(a, ws, Rpad);
    (void)r;
    note_kernel("collapse_wide2_kernel");
    const int ntile = collapse_wide2_tiles(a.T);
    const size_t lds = (size_t)kW2NBuf * kW2StageB + 32 * sizeof(double) + 64 * 4 * sizeof(unsigned long long);   // stages | redS, itemq | DIAG stamps
    static const int xcd_env = [] { const char* v = diag_env("DFM_WIDE_XCD"); return v ? atoi(v) : -1; }();
    const int xcd_map = xcd_env >= 0 ? (xcd_env != 0) : (a.B >= 16);
    static const int abl = [] { const char* v = diag_env("DFM_W2_ABL"); return v ? atoi(v) : 0; }();   // diagnostics (wrong results)
    // one persistent workgroup per CU (130 KB of LDS each), a multiple of 8 so that every XCD has the same number
    const long long NT = (long long)a.B * ntile;
    int G = (num_cu > 0 ? num_cu : 256) * (kW2Rows <= 64 ? 2 : 1);   // as many as fit the LDS of every CU
    G = (G / 8) * 8;
    if (G < 8) G = 8;
    if (!xcd_map && NT < G) G = (int)NT;
    if (Rpad <= 16) {
        if (a.nobs != nullptr) return hipErrorInvalidValue;  // (missing cells at Rp <= 16: collapse_kernel / collapse_miss)
        return launch_w2v<16, 0, 0>(a, w, G, lds, ntile, xcd_map, 0, s, Rpad);
    }
    const int nx = r <= 16 ? 1 : (r + 3 - 16) / 4;            // 4-factor groups past the first 16; 4 = a second 16-wide tile
    switch (nx) {
        case 1: return launch_w2<1>(a, w, G, lds, ntile, xcd_map, abl, s);
        case 2: return launch_w2<2>(a, w, G, lds, ntile, xcd_map, abl, s);
        case 3: return launch_w2<3>(a, w, G, lds, ntile, xcd_map, abl, s);
        default: return launch_w2<4>(a, w, G, lds, ntile, xcd_map, abl, s);
    }
}

// C_t of the periods with missing cells (a.Ct, packed; the other periods keep Cfull): after launch_wide_prep, beside or after
// the collapse (it reads the panel itself)
// compact C_t rows (CollapseArgs::ct_r) fit the LDS of ct_miss_wide2_kernel for this cross-section
bool ct_miss_wide_compact_ok(int N, int ct_r) {
    const size_t npad64 = (size_t)((N + 63) / 64) * 64;
    return (size_t)3 * 16 * 1088 + npad64 * (sizeof(double) + sizeof(unsigned long long)) + 16
           + (size_t)kCtP * (ct_r * (ct_r + 1) / 2) * sizeof(double) <= 160 * 1024;
}

hipError_t launch_ct_miss_wide(const CollapseArgs& a, double* ws, int r, hipStream_t s, const double* V) {
    static const int skip = [] { const char* v = diag_env("DFM_CT_SKIP"); return v ? atoi(v) : 0; }();   // diagnostics (wrong results)
    if (skip) return hipSuccess;
    const W2Ws w = w2_ws(a, ws, kW2R);
    const int ntile16 = (a.T + kCtP - 1) / kCtP;
    static const int old = [] { const char* v = diag_env("DFM_CT_OLD"); return v ? atoi(v) : 0; }();     // A/B: the round-2 kernel
    static const int staged = [] { const char* v = diag_env("DFM_CT_STAGED"); return v ? atoi(v) : 0; }();   // A/B: the round-4 kernel
    static const int noslice = [] { const char* v = diag_env("DFM_CT_NOSLICE"); return v ? atoi(v) : 0; }();   // A/B: the per-period gather kernels
    if (!old && !staged && !noslice && V != nullptr && a.nobs != nullptr) {
        // slices of V that leave room for the waves' rows: as few as fit 150 KB, of equal size, a multiple of 4 series
        const int rr = r > 0 && r <= kW2R ? r : kW2R;
        const int shape = rr <= 20 ? 0 : (rr <= 28 ? 1 : 2);  // 2 x 2 | 2 x 4 | 4 x 4 blocks per lane
        const int bw = shape == 0 ? 2 : 4;
        const int recB = 8 * (((rr + bw - 1) / bw) * bw);
        const int ctr = a.ct_r > 0 ? a.ct_r : kW2R;
        const size_t NPe = (size_t)(((ctr * (ctr + 1) / 2) + 1) & ~1);
        auto lds_for = [&](int ns) {                              // the slice + a row of zeros | C | per wave: deficit row, list
            return (size_t)(ns + 1) * recB + (1 + (size_t)kCsWaves) * NPe * sizeof(double) + (size_t)kCsWaves * ((size_t)((ns + 63) / 64) * 64 + 8) * sizeof(unsigned short) + 16;
        };
        int nsl = 1;
        while (nsl < a.N && lds_for((a.N + nsl - 1) / nsl + 3) > 150 * 1024) ++nsl;
        const int per = ((a.N + nsl - 1) / nsl + 3) & ~3;
        if (lds_for(per) <= 150 * 1024) {
            note_kernel("ct_miss_slice_kernel");
            static LdsOptIn attr_cs;
            if (!attr_cs) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ct_miss_slice_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ct_miss_slice_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ct_miss_slice_kernel<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
                attr_cs = true;
            }
            // time chunks: one workgroup per CU where the batch leaves CUs idle (each copies its slice once: chunks of >= 64 periods)
            static const int ncu = [] {                       // one workgroup per CU: time chunks per replicate = CUs / B (chunks >= 64 periods)
                int dev = 0, n = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
                return n;
            }();
            int TC = a.B >= ncu ? 1 : ncu / a.B;
            if (TC > a.T / 64) TC = a.T / 64;
            if (TC < 1) TC = 1;
            for (int s0 = 0, k = 0; s0 < a.N; s0 += per, ++k) {
                const int ns = s0 + per <= a.N ? per : a.N - s0;
                const dim3 grid((unsigned)(a.B * TC)), blk(64 * kCsWaves);
                const int fst = k == 0 ? 1 : 0;
                if (shape == 0) hipLaunchKernelGGL((ct_miss_slice_kernel<2, 2>), grid, blk, lds_for(ns), s, a, V, rr, s0, ns, fst, TC);
                else if (shape == 1) hipLaunchKernelGGL((ct_miss_slice_kernel<2, 4>), grid, blk, lds_for(ns), s, a, V, rr, s0, ns, fst, TC);
                else hipLaunchKernelGGL((ct_miss_slice_kernel<4, 4>), grid, blk, lds_for(ns), s, a, V, rr, s0, ns, fst, TC);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        }
    }
    if (!old) {
        note_kernel("ct_miss_wide2_kernel");
        static const int ppw_env = [] { const char* v = diag_env("DFM_CT_PPW"); return v ? atoi(v) : 0; }();   // A/B: 1 = the round-3 tiling
        const int ppw = ppw_env == 1 ? 1 : 2;
        const int ntile = (a.T + kCtP * ppw - 1) / (kCtP * ppw);
        const size_t npad64 = (size_t)((a.N + 63) / 64) * 64;
        const int ctr = a.ct_r > 0 ? a.ct_r : kW2R;
        auto lds_for = [&](int nb) {                              // W stages | R | masks | [16 waves] output rows
            return (size_t)nb * 16 * 1088 + npad64 * (sizeof(double) + sizeof(unsigned long long)) + 16
                   + (size_t)kCtP * (ctr * (ctr + 1) / 2) * sizeof(double);
        };
        // (five stage buffers -- four stages in flight -- measured the same 4.4-4.5 ms as three at config 4: the stage loop is bound by
        //  the waves' instruction issue, ~300 instructions per wave and stage at four waves per SIMD, not by the DMA latency)
        const int nbuf = 3;
        const size_t lds2 = lds_for(nbuf);
        if (lds2 > 160 * 1024) {
            // a very wide cross-section (16 bytes of LDS per series here): the round-2 kernel, whose masks are 2 bytes per series --
            // full rows only (capi.hip asks ct_miss_wide_compact_ok before it promises the recursion compact rows)
            if (a.ct_r > 0) return hipErrorInvalidValue;
            note_kernel("ct_miss_wide_kernel");
            const size_t lds = (size_t)2 * kW2Chunk * kW2R * sizeof(double) + (size_t)w.npad * sizeof(unsigned short) + 16;
            hipLaunchKernelGGL(ct_miss_wide_kernel, dim3((unsigned)((long long)a.B * ntile16)), dim3(kCtThreads), lds, s, a, w.W, ntile16);
            return hipGetLastError();
        }
        static LdsOptIn attr_ct;
        if (!attr_ct) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ct_miss_wide2_kernel<2, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ct_miss_wide2_kernel<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ct_miss_wide2_kernel<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_ct = true;
        }
        const int rr = r > 0 && r <= kW2R ? r : kW2R;
        const dim3 grid((unsigned)((long long)a.B * ntile));
        if (ppw == 1) hipLaunchKernelGGL((ct_miss_wide2_kernel<1, 3>), grid, dim3(kCtThreads), lds2, s, a, w.W, ntile, rr);
        else if (nbuf == 5) hipLaunchKernelGGL((ct_miss_wide2_kernel<2, 5>), grid, dim3(kCtThreads), lds2, s, a, w.W, ntile, rr);
        else hipLaunchKernelGGL((ct_miss_wide2_kernel<2, 3>), grid, dim3(kCtThreads), lds2, s, a, w.W, ntile, rr);
        return hipGetLastError();
    }
    note_kernel("ct_miss_wide_kernel");
    const size_t lds = (size_t)2 * kW2Chunk * kW2R * sizeof(double) + (size_t)w.npad * sizeof(unsigned short) + 16;
    hipLaunchKernelGGL(ct_miss_wide_kernel, dim3((unsigned)((long long)a.B * ntile16)), dim3(kCtThreads), lds, s, a, w.W, ntile16);
    return hipGetLastError();
}

}  // namespace dfm
