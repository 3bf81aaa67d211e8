// collapse_wide2.hip -- the balanced-panel collapse for Rp = 32 (BASELINE config 4: N = 1000, T = 2000, r = 20) on the
// LDS-DMA path and `v_mfma_f64_16x16x4`.
//
//     b_t = sum_i lam_i x_it / R_i   (32 padded factors)        sum_t s_t,  s_t = sum_i x_it^2 / R_i
//
// collapse_wide_kernel (collapse_wide.hip) loads its A operands straight from the panel -- 4 rows x 32..64 bytes per load
// instruction -- divides by R in every step and re-reads the weights from L2 once per 16-period tile: 3.5 ms for the
// 4.1 GB of config 4 (1.17 TB/s, 0.15 of HBM peak).  Here:
//   * wide_prep_kernel (once per pass): W = lam / R into the workspace, C = Lam' W (32 x 32) on the matrix pipe (4 waves =
//     the 2 x 2 tiles of 16 x 16, a chain of v_mfma_f64_16x16x4 over the series each), sum log R -- gram_wide_kernel's work
//     (0.31 ms of scalar-indexed VALU loops) in ~0.03 ms;
//   * collapse_wide2_kernel: one workgroup (8 waves) per tile of 64 periods x ALL series.  Panel tile AND weights stream into
//     LDS by `global_load_lds_dwordx4` in stages of 64 series (64 rows x 512 bytes + the contiguous 64 x 32 block of W),
//     double-buffered, one barrier per stage.  Wave (rt, half) accumulates row tile rt (16 periods) x both 16-factor tiles
//     over its half of the stage's steps: per step of 4 series one 8-byte LDS read of A (A[i][k] = x[t0 + 16 rt + i][c + k];
//     row pairs 1040 bytes apart), two of B (B[k][j] = W[c + k][16 ft + j]) and two MFMAs.  s_t from a duplicate-free second
//     read of the stage (thread = row x 8-series group).  HBM sees every panel byte once; W is read once per 64-period tile
//     (1/2 of the panel bytes; a first version with 32-period tiles and register-fed W read as many W bytes as panel
//     bytes -- 65 MB of weights do not stay in the 4-MB L2s while 4 GB stream through them -- and ran at 2.1 ms).
// r = 20 uses 20 of the 32 factor columns (the second factor tile is 3/4 padding): the kernel is bandwidth-bound, the matrix
// pipe has the room.  Reference counterpart: forming Lambda' x_t in the per-period regression of x_t on Lambda
// (dfm_functions.ipynb:271-286 called from :364).
#include "dfm_gram.h"
#include "dfm_kernels.h"

namespace dfm {

namespace {

using lds_char_ptr_w = __attribute__((address_space(3))) char*;
typedef double w2_v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16w(const void* gsrc, unsigned lds_dst) {
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
__device__ __forceinline__ void wait_all_w() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr int kW2R = 32;          // padded factors
constexpr int kW2Rows = 64;       // periods per workgroup tile
constexpr int kW2Chunk = 64;      // series per stage (512 bytes of a panel row)
constexpr unsigned kW2PairB = 1040;  // LDS bytes of a PAIR of panel rows (2 x 512 + 16: 16 consecutive rows start on different banks)
constexpr unsigned kW2RS = kW2PairB / 2;   // average bytes per row (sizes only)
constexpr int kW2Steps = kW2Chunk / 4;
constexpr int kW2NBuf = 3;        // stage buffers

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// W = lam / R, C = Lam' W, sum log R.  One workgroup of 4 waves per replicate.
__global__ __launch_bounds__(256) void wide_prep_kernel(CollapseArgs a, double* Wout) {
    constexpr int R = kW2R;
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N;
    const double* __restrict__ L = a.Lam + (size_t)b * N * R;
    const double* __restrict__ Rv = a.Rv + (size_t)b * N;
    double* W = Wout + (size_t)b * N * R;
    for (int e = tid; e < N * R; e += 256) W[e] = L[e] / Rv[e / R];
    double ld = 0.0;
    for (int c = tid; c < N; c += 256) ld += log(Rv[c]);
    ld = wave_allsum(ld);
    if (lane == 0) red[wave] = ld;
    __syncthreads();                                         // W of this replicate is complete (and visible to this workgroup)
    if (tid == 0) a.ldfull[b] = red[0] + red[1] + red[2] + red[3];
    // tile (it, jt) of C: C[16 it + i][16 jt + j] = sum_c Lam[c][16 it + i] W[c][16 jt + j]
    const int it = wave >> 1, jt = wave & 1;
    const int k4 = lane >> 4, c16 = lane & 15;
    w2_v4 acc = {0.0, 0.0, 0.0, 0.0};
    const int steps = (N + 3) / 4;
    for (int s0 = 0; s0 < steps; s0 += 8) {                  // 16 loads in flight per batch (clamped addresses, select afterwards)
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = 4 * (s0 + u) + k4;
            const int cc = c < N ? c : N - 1;
            av[u] = L[(size_t)cc * R + 16 * it + c16];
            bv[u] = W[(size_t)cc * R + 16 * jt + c16];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = 4 * (s0 + u) + k4;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(c < N ? av[u] : 0.0, bv[u], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v)                               // D[(l / 16) + 4 v][l % 16]
        a.Cfull[(size_t)b * R * R + (size_t)(16 * it + k4 + 4 * v) * R + 16 * jt + c16] = acc[v];
}

// ------------------------------------------------------------------------------------------------------------------
// One workgroup of 8 waves per tile of 64 periods.  Stage = 64 series: 64 panel rows x 512 bytes + the 64 x 32 block of W
// (16 KB), both by LDS-DMA, double-buffered (98 KB).  Wave rt (0..3) x half (0..1): row tile rt, the stage's steps of its
// half (8 of 16) for BOTH factor tiles -- the two halves of a row tile are summed at the end through LDS.  Per step: one
// 8-byte LDS read of A, two of B, two MFMAs.
__global__ __launch_bounds__(512) void collapse_wide2_kernel(CollapseArgs a, const double* __restrict__ Wall, int ntile) {
    constexpr int R = kW2R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr unsigned kStageB = kW2Rows * kW2RS + kW2Chunk * R * 8;                 // panel rows | W block
    double* rinvS = reinterpret_cast<double*>(smem + kW2NBuf * kStageB);             // [nch * kW2Chunk]: 1 / R of every series (0 past N)
    double* redS = rinvS + ((a.N + kW2Chunk - 1) / kW2Chunk) * kW2Chunk;             // [8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = (int)blockIdx.x / ntile, tile = (int)blockIdx.x % ntile;
    const int N = a.N, T = a.T;
    const int t0 = tile * kW2Rows;
    const int rt = wave >> 1, half = wave & 1;
    const int k4 = lane >> 4, c16 = lane & 15;
    const unsigned rowB = (unsigned)N * 8u;
    const char* Xb = reinterpret_cast<const char*>(a.panel + (size_t)b * T * N);
    const char* Wb = reinterpret_cast<const char*>(Wall + (size_t)b * N * R);
    const double* __restrict__ Rv = a.Rv + (size_t)b * N;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_w)(smem));
    const int nch = (N + kW2Chunk - 1) / kW2Chunk;
    const unsigned wbytes = (unsigned)N * R * 8u;

    // stage ch -> buffer bsel: wave w brings in panel rows 8 w .. 8 w + 7 (512 bytes each: lanes 0..31) and 2 KB of the W
    // block (2 DMAs of 1 KB: the block is contiguous in memory)
    auto issue_dma = [&](int ch, int bsel) {
        const unsigned sbase = lds0 + (unsigned)bsel * kStageB;
        const unsigned colB = (unsigned)ch * (kW2Chunk * 8u) + 16u * (lane & 31);
        const bool act = colB < rowB;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {                      // one DMA moves two rows: lanes 0..31 row 2 rr, lanes 32..63 row 2 rr + 1
            const int row = wave * 8 + 2 * rr + (lane >> 5);
            int t = t0 + row;
            t = t < T ? t : T - 1;
            const char* src = Xb + (size_t)t * rowB + colB;
            // LDS destination of lane l = base + 16 l: rows kW2RS apart need one base per row pair with the second row at +512
            // -> rows are laid out in PAIRS: pair p at p * 2 * kW2RS', row stride inside the pair 512 bytes
            const unsigned dst = __builtin_amdgcn_readfirstlane(sbase + (unsigned)(wave * 4 + rr) * kW2PairB);
            if (act) dma16w(src, dst);
        }
        const unsigned woff = (unsigned)ch * (kW2Chunk * R * 8u) + (unsigned)wave * 2048u + 16u * lane;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned o = woff + 1024u * u;
            const unsigned dst = __builtin_amdgcn_readfirstlane(sbase + kW2Rows * kW2RS + (unsigned)wave * 2048u + 1024u * u);
            if (o < wbytes) dma16w(Wb + o, dst);
        }
    };
    // no NaN bit patterns in columns / rows the DMAs of a partial stage do not write: zero the buffers once
    for (int e = tid; e < kW2NBuf * (int)kStageB / 8; e += 512) reinterpret_cast<double*>(smem)[e] = 0.0;
    __syncthreads();
    // THREE stage buffers: two stages are in flight while one is consumed (with two, the workgroup's 49 KB burst per barrier
    // left HBM idle half of the time: 2.07 ms).  Every wave issues exactly 6 DMAs per stage except in the last, partial one
    // (which has no younger stage), so "stage ch has landed" is a counted wait that leaves stage ch + 1 outstanding.
    issue_dma(0, 0);
    if (nch > 1) issue_dma(1, 1);
    for (int c = tid; c < nch * kW2Chunk; c += 512) rinvS[c] = c < N ? 1.0 / Rv[c] : 0.0;   // (no register load may sit between the
                                                                                             // DMAs of the loop: one vmcnt order)
    w2_v4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    double qs = 0.0;
    const int srow = tid >> 3, scg = tid & 7;                 // s_t pass: thread = (row 0..63, group of 8 series)
    // A operand of this lane: row 16 rt + c16 of the tile -> pair (16 rt + c16) / 2, slot (16 rt + c16) % 2
    const unsigned arow = (unsigned)((16 * rt + c16) >> 1) * kW2PairB + (unsigned)((16 * rt + c16) & 1) * 512u;
    const unsigned srowoff = (unsigned)(srow >> 1) * kW2PairB + (unsigned)(srow & 1) * 512u;
    int bsel = 0;
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else wait_all_w();
        __syncthreads();                                      // stage ch (panel rows, W block, 1 / R) is ready
        const int bnew = bsel == 0 ? 2 : bsel - 1;            // buffer of stage ch + 2 = the one stage ch - 1 used
        if (ch + 2 < nch) issue_dma(ch + 2, bnew);
        const char* stage = smem + (size_t)bsel * kStageB;
        const char* pa = stage + arow + (size_t)(half * (kW2Steps / 2) * 4 + k4) * 8;
        const char* pb = stage + kW2Rows * kW2RS + (size_t)(half * (kW2Steps / 2) * 4 + k4) * (R * 8) + (size_t)c16 * 8;
        const int cfirst = ch * kW2Chunk + half * (kW2Steps / 2) * 4 + k4;   // series of this lane's k in step 0
#pragma unroll
        for (int s = 0; s < kW2Steps / 2; ++s) {
            double av = *reinterpret_cast<const double*>(pa + s * 32);
            av = (cfirst + 4 * s < N) ? av : 0.0;             // the last stage may be partial: its stale columns / W rows count for nothing
            const double b0 = *reinterpret_cast<const double*>(pb + s * (4 * R * 8));
            const double b1 = *reinterpret_cast<const double*>(pb + s * (4 * R * 8) + 128);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1, acc1, 0, 0, 0);
        }
        {   // s_t: 8 cells of one row per thread
            const char* ps = stage + srowoff + (size_t)scg * 64;
            const double* pr = rinvS + ch * kW2Chunk + scg * 8;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double2 x = *reinterpret_cast<const double2*>(ps + u * 16);
                const double2 ri = *reinterpret_cast<const double2*>(pr + u * 2);
                qs = fma(x.x * ri.x, x.x, qs);
                qs = fma(x.y * ri.y, x.y, qs);
            }
        }
        bsel = bsel == 2 ? 0 : bsel + 1;
    }
    // the two halves of a row tile meet in LDS (the stage buffers are free now)
    __syncthreads();
    double* xch = reinterpret_cast<double*>(smem);            // [4 row tiles][2 factor tiles][4][64]
    if (half == 1) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            xch[((rt * 2 + 0) * 4 + v) * 64 + lane] = acc0[v];
            xch[((rt * 2 + 1) * 4 + v) * 64 + lane] = acc1[v];
        }
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {                         // D[(l / 16) + 4 v][l % 16]
            const int t = t0 + rt * 16 + k4 + 4 * v;
            if (t < T) {
                double* out = a.bcol + ((size_t)b * T + t) * R;
                out[c16] = acc0[v] + xch[((rt * 2 + 0) * 4 + v) * 64 + lane];
                out[16 + c16] = acc1[v] + xch[((rt * 2 + 1) * 4 + v) * 64 + lane];
            }
        }
    }
    // sum over the tile's valid periods of s_t
    if (t0 + srow >= T) qs = 0.0;
    qs = wave_allsum(qs);
    if (lane == 0) redS[wave] = qs;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += redS[w];
        a.scol[(size_t)b * T + tile] = tot;
        if (tot != tot) atomicOr(a.status, 1);               // NaN in the panel on the balanced path
    }
}

int collapse_wide2_tiles(int T) { return (T + kW2Rows - 1) / kW2Rows; }
bool collapse_wide2_supported(int Rpad, int N) { return Rpad == 32 && (N % 2) == 0 && N >= 2 && N <= 1280; }   // 1 / R table in LDS beside the three stages

hipError_t launch_wide_prep(const CollapseArgs& a, double* W, hipStream_t s) {
    hipLaunchKernelGGL(wide_prep_kernel, dim3(a.B), dim3(256), 0, s, a, W);
    return hipGetLastError();
}

hipError_t launch_collapse_wide2(const CollapseArgs& a, const double* W, hipStream_t s) {
    const int ntile = collapse_wide2_tiles(a.T);
    const size_t stage = (size_t)kW2Rows * kW2RS + (size_t)kW2Chunk * kW2R * 8;
    size_t lds = kW2NBuf * stage + (size_t)(((a.N + kW2Chunk - 1) / kW2Chunk) * kW2Chunk + 8) * sizeof(double);
    const size_t xch = (size_t)4 * 2 * 4 * 64 * sizeof(double);
    if (lds < xch) lds = xch;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&collapse_wide2_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(collapse_wide2_kernel, dim3((unsigned)((long long)a.B * ntile)), dim3(512), lds, s, a, W, ntile);
    return hipGetLastError();
}

}  // namespace dfm
