// pca.hip -- PCA initialisation of the EM loop, batched over replicates.
//
// Reference: pca_score (dfm_functions.ipynb:179-183): `_, _, V = svd(X); score = (X*V)[:, 1:nfac_u]` on
// the standardised balanced panel (:339-348), followed here by the closed-form OLS start of EM used by
// the oracle (oracle/kalman_oracle.py pca_init; Doz, Giannone & Reichlin two-step start):
//     Lam = OLS(x on F) = V_r,  R_i = mean squared residual,  A, Q = VAR(1) OLS of F (no constant,
//     divisor T-1),  mu0 = 0,  P0 = F'F / T.
// The right singular vectors of X are the eigenvectors of S = X'X, so per replicate:
//   gram_xx_kernel   S = X'X           (N x N, the one dense contraction on the path; 4x4 register tiles)
//   pca_kernel       top-r eigenpairs of S by orthogonal (subspace) iteration with a Rayleigh-Ritz
//                    projection (r x r cyclic Jacobi) -- iterated to the fp64 floor, then sign-fixed as
//                    the oracle does (largest-|.| entry of each vector positive); F = X V; the r x r
//                    moment matrices of F; the OLS / VAR solves.
// One workgroup (256 threads) per replicate in both kernels.
#include <stdlib.h>

#include "dfm_grid.h"
#include "dfm_kernels.h"

namespace dfm {

constexpr int kPcaThreads = 256;
constexpr int kPcaFastThreads = 512;   // pca_iterate_lds: 8 waves, 2 per SIMD (the 128-VGPR budget of 1024 threads spills)

// ---------------------------------------------------------------------------------------------
// S = X'X.  Thread tile 4 x 4 over the upper triangle of 4 x 4 blocks; mirrored on store.
__global__ __launch_bounds__(kPcaThreads) void gram_xx_kernel(PcaArgs a) {
    const int b = blockIdx.x;
    const int N = a.N, T = a.T;
    const double* __restrict__ X = a.panel + (size_t)b * T * N;
    double* S = a.S + (size_t)b * N * N;
    const int nb = (N + 3) / 4;                     // 4-wide blocks per side
    const int ntile = nb * (nb + 1) / 2;
    for (int tile = threadIdx.x; tile < ntile; tile += kPcaThreads) {
        // tile -> (bi <= bj) in the upper triangle, row-major over bi
        int bi = 0, rem = tile;
        while (rem >= nb - bi) { rem -= nb - bi; ++bi; }
        const int bj = bi + rem;
        const int i0 = 4 * bi, j0 = 4 * bj;
        double acc[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
        for (int t = 0; t < T; ++t) {
            const double* xr = X + (size_t)t * N;
            double xi[4], xj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xi[u] = (i0 + u < N) ? xr[i0 + u] : 0.0;
                xj[u] = (j0 + u < N) ? xr[j0 + u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = fma(xi[u], xj[v], acc[u][v]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = i0 + u, j = j0 + v;
                if (i < N && j < N) {
                    S[(size_t)i * N + j] = acc[u][v];
                    S[(size_t)j * N + i] = acc[u][v];
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// S = X'X on the fp64 matrix pipe (N <= 256).  One workgroup of 8 waves per replicate.  The panel streams through LDS in
// blocks of 32 periods (double-buffered: the next block's global loads are in flight while this one is consumed); a 16 x 16
// tile (bi, bj), bi <= bj, of S accumulates in 4 VGPR pairs of one wave as a chain of v_mfma_f64_16x16x4 over the periods
// (A = 4 periods x 16 series bi, transposed; B = the same 4 periods x 16 series bj -- both operands are 8-byte LDS reads of
// one panel row segment).  Wave w owns tiles w, w + 8, ...: at N = 200 (13 x 14 / 2 = 91 tiles) 12 accumulator tiles = 96
// VGPRs.  Arithmetic: 91 tiles x 125 MFMAs x 64 cycles per replicate and SIMD pair -- ~0.3 ms for 1024 replicates against
// the 0.2 ms the panel takes to stream; the VALU kernel above needed 3.1 ms.
// v_mfma_f64_16x16x4 lane layout (measured: scripts/microbench/mfma16probe.hip): A[i][k] in lane 16 k + i; B[k][j] in lane
// 16 k + j; D[(l / 16) + 4 v][l % 16] in register v of lane l.
constexpr int kGxThreads = 512;
constexpr int kGxPB = 32;                     // periods per LDS block
typedef double gx_v4 __attribute__((ext_vector_type(4)));

template <int MAXP>
__global__ __launch_bounds__(kGxThreads) void gram_xx_mfma_kernel(PcaArgs a) {
    extern __shared__ __attribute__((aligned(16))) double gx_lds[];
    const int b = blockIdx.x;
    const int N = a.N, T = a.T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double* __restrict__ X = a.panel + (size_t)b * T * N;
    double* S = a.S + (size_t)b * N * N;
    const int NT = (N + 15) / 16;
    const int P = NT * (NT + 1) / 2;
    const int ld = NT * 16 + 2;                                  // row stride of the LDS block (doubles): columns >= N are zero
    const int k4 = lane >> 4, c16 = lane & 15;
    // tiles of this wave
    int tbi[MAXP], tbj[MAXP];
#pragma unroll
    for (int m = 0; m < MAXP; ++m) {
        const int p = wave + 8 * m;
        int bi = 0, rem = p < P ? p : 0;
        while (rem >= NT - bi) { rem -= NT - bi; ++bi; }
        tbi[m] = bi; tbj[m] = bi + rem;
    }
    gx_v4 acc[MAXP];
#pragma unroll
    for (int m = 0; m < MAXP; ++m) acc[m] = gx_v4{0.0, 0.0, 0.0, 0.0};
    const int nblk = (T + kGxPB - 1) / kGxPB;
    const int per_thread = (kGxPB * ld + kGxThreads - 1) / kGxThreads;   // LDS doubles each thread stages per block
    // element e of a block: row e / ld, column e % ld
    auto fetch = [&](int blk, int e) -> double {
        const int rr = e / ld, cc = e - rr * ld;
        const int t = blk * kGxPB + rr;
        return (rr < kGxPB && cc < N && t < T) ? X[(size_t)t * N + cc] : 0.0;
    };
    constexpr int kStage = 8;                                    // staged doubles per thread and step
    double* buf0 = gx_lds;
    double* buf1 = gx_lds + (size_t)kGxPB * ld;
    for (int e = tid; e < kGxPB * ld; e += kGxThreads) buf0[e] = fetch(0, e);
    __syncthreads();
    for (int blk = 0; blk < nblk; ++blk) {
        double* cur = (blk & 1) ? buf1 : buf0;
        double* nxt = (blk & 1) ? buf0 : buf1;
        const bool more = blk + 1 < nblk;
        // the compute of this block, with the next block's loads issued in slices of kStage in between the tiles
        int e_next = tid;
#pragma unroll
        for (int m = 0; m < MAXP; ++m) {
            double stage[kStage];
            int es[kStage];
            if (more) {
#pragma unroll
                for (int u = 0; u < kStage; ++u) {
                    es[u] = e_next;
                    stage[u] = (e_next < kGxPB * ld) ? fetch(blk + 1, e_next) : 0.0;
                    e_next += kGxThreads;
                }
            }
            if (wave + 8 * m < P) {                              // (wave-uniform)
                const double* pa = cur + (size_t)k4 * ld + tbi[m] * 16 + c16;
                const double* pb = cur + (size_t)k4 * ld + tbj[m] * 16 + c16;
#pragma unroll
                for (int kk = 0; kk < kGxPB / 4; ++kk) {
                    const double av = pa[(size_t)kk * 4 * ld];
                    const double bv = pb[(size_t)kk * 4 * ld];
                    acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[m], 0, 0, 0);
                }
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < kStage; ++u)
                    if (es[u] < kGxPB * ld) nxt[es[u]] = stage[u];
            }
        }
        if (more) {                                              // whatever the slices above did not cover
            for (int e = e_next; e < kGxPB * ld; e += kGxThreads) nxt[e] = fetch(blk + 1, e);
        }
        __syncthreads();
    }
    (void)per_thread;
    // store: tile (bi, bj) and its mirror
#pragma unroll
    for (int m = 0; m < MAXP; ++m) {
        if (wave + 8 * m < P) {
            const int j = tbj[m] * 16 + c16;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = tbi[m] * 16 + k4 + 4 * v;
                if (i < N && j < N) {
                    S[(size_t)i * N + j] = acc[m][v];
                    S[(size_t)j * N + i] = acc[m][v];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same kernel with the panel rows brought in by LDS-DMA (even N: rows start on 16-byte boundaries).  The register-staged
// version above pays an integer division per staged element for its (row, column) -- 8 of them in front of every tile, more VALU
// time than the tile's 8 MFMAs take -- and holds the staged values across the tile.  Here wave w issues rows w, w + 8, ... of the
// NEXT block as `global_load_lds_dwordx4` (1 KB per instruction) before it starts on its tiles and waits for them at the block's
// barrier; nothing else moves the panel.  Row stride = 128 bytes (mod 256): the four k-rows of an MFMA step on distinct banks.
// Columns >= N of the buffers stay zero (zeroed once: the DMAs never write them); the rows of a partial last block are zeroed.
namespace {
using lds_char_ptr_gx = __attribute__((address_space(3))) char*;
__device__ __forceinline__ void dma16gx(const void* gsrc, unsigned lds_dst) {
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
__host__ __device__ inline int gx_dma_ld(int N) { const int NT = (N + 15) / 16; return NT * 16 + ((NT & 1) ? 0 : 16); }
}  // namespace

template <int MAXP>
__global__ __launch_bounds__(kGxThreads) void gram_xx_dma_kernel(PcaArgs a) {
    extern __shared__ __attribute__((aligned(16))) double gx_lds[];
    const int b = blockIdx.x;
    const int N = a.N, T = a.T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* __restrict__ X = reinterpret_cast<const char*>(a.panel + (size_t)b * T * N);
    double* S = a.S + (size_t)b * N * N;
    const int NT = (N + 15) / 16;
    const int P = NT * (NT + 1) / 2;
    const int ld = gx_dma_ld(N);
    const unsigned rowB = (unsigned)N * 8u, ldB = (unsigned)ld * 8u;
    const int npiece = (int)((rowB + 1023u) / 1024u);
    const int k4 = lane >> 4, c16 = lane & 15;
    int tbi[MAXP], tbj[MAXP];
#pragma unroll
    for (int m = 0; m < MAXP; ++m) {
        const int p = wave + 8 * m;
        int bi = 0, rem = p < P ? p : 0;
        while (rem >= NT - bi) { rem -= NT - bi; ++bi; }
        tbi[m] = bi; tbj[m] = bi + rem;
    }
    gx_v4 acc[MAXP];
#pragma unroll
    for (int m = 0; m < MAXP; ++m) acc[m] = gx_v4{0.0, 0.0, 0.0, 0.0};
    const int nblk = (T + kGxPB - 1) / kGxPB;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr_gx)(gx_lds));
    const unsigned bufB = (unsigned)kGxPB * ldB;
    for (int e = tid; e < 2 * kGxPB * ld; e += kGxThreads) gx_lds[e] = 0.0;
    __syncthreads();
    auto issue = [&](int blk, int sel) {                         // rows wave, wave + 8, ... of block blk into buffer sel
        const int t0 = blk * kGxPB;
        for (int rr = wave; rr < kGxPB; rr += 8) {
            const int t = t0 + rr;
            if (t < T) {                                         // (wave-uniform)
                const char* src = X + (size_t)t * rowB + 16u * (unsigned)lane;
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)sel * bufB + (unsigned)rr * ldB);
                for (int pc = 0; pc < npiece; ++pc)
                    if (16u * (unsigned)lane + 1024u * (unsigned)pc < rowB) dma16gx(src + 1024 * pc, dst + 1024u * (unsigned)pc);
            } else {                                             // past the sample: a zero row
                double* row = gx_lds + (size_t)sel * kGxPB * ld + (size_t)rr * ld;
                for (int c = lane; c < N; c += 64) row[c] = 0.0;
            }
        }
    };
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int blk = 0; blk < nblk; ++blk) {
        const double* cur = gx_lds + (size_t)(blk & 1) * kGxPB * ld;
        if (blk + 1 < nblk) issue(blk + 1, (blk + 1) & 1);       // (that buffer was last read in block blk - 1: everybody is past its barrier)
#pragma unroll
        for (int m = 0; m < MAXP; ++m) {
            if (wave + 8 * m < P) {                              // (wave-uniform)
                const double* pa = cur + (size_t)k4 * ld + tbi[m] * 16 + c16;
                const double* pb = cur + (size_t)k4 * ld + tbj[m] * 16 + c16;
                double av[kGxPB / 4], bv[kGxPB / 4];
#pragma unroll
                for (int kk = 0; kk < kGxPB / 4; ++kk) { av[kk] = pa[(size_t)kk * 4 * ld]; bv[kk] = pb[(size_t)kk * 4 * ld]; }
#pragma unroll
                for (int kk = 0; kk < kGxPB / 4; ++kk) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc[m], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's rows of the next block have landed
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < MAXP; ++m) {
        if (wave + 8 * m < P) {
            const int j = tbj[m] * 16 + c16;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = tbi[m] * 16 + k4 + 4 * v;
                if (i < N && j < N) {
                    S[(size_t)i * N + j] = acc[m][v];
                    S[(size_t)j * N + i] = acc[m][v];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// block-wide sum of NV values per thread -> every thread gets the totals (through LDS)
template <int NV, int NT = kPcaThreads>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* red /* [NT / 64][NV] */) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off, kWave);
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) red[wave * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) t += red[w * NV + k];
        v[k] = t;
    }
}

// M (r x r, row-major in LDS, leading dimension R) = A' B for tall A, B ([n][R] in global memory)
template <int R, int NT = kPcaThreads>
__device__ __forceinline__ void tall_gram(double* M, const double* A, const double* Bm, int n, int r, double* red) {
    for (int p = 0; p < r; ++p) {
        double acc[R];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = 0.0;
        for (int i = threadIdx.x; i < n; i += NT) {
            const double ap = A[(size_t)i * R + p];
#pragma unroll
            for (int q = 0; q < R; ++q) acc[q] = fma(ap, Bm[(size_t)i * R + q], acc[q]);
        }
        block_sum<R, NT>(acc, red);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < R; ++q) M[p * R + q] = acc[q];
        }
    }
    __syncthreads();
}

// In-place Cholesky of the leading r x r block of G (LDS, ld R): lower factor L; thread 0.
template <int R>
__device__ __forceinline__ void chol_lds(double* G, int r) {
    for (int j = 0; j < r; ++j) {
        double d = G[j * R + j];
        for (int k = 0; k < j; ++k) d -= G[j * R + k] * G[j * R + k];
        d = sqrt(d);
        G[j * R + j] = d;
        for (int i = j + 1; i < r; ++i) {
            double s = G[i * R + j];
            for (int k = 0; k < j; ++k) s -= G[i * R + k] * G[j * R + k];
            G[i * R + j] = s / d;
        }
    }
}

// tall_gram on the matrix pipe (generic path, 4 waves): M = A'B as 16 x 16 tiles, A'[p][k] = A[k][16 pt + p] and
// B[k][q] = Bm[k][16 qt + q] straight from global memory (128 contiguous bytes per k), 8 steps in flight.  R = 32: one tile per
// wave over all rows; R <= 16: the one tile's rows split over the waves, partial tiles summed through `part` ([NT / 64][4][64]).
// (tall_gram above re-reads B once per column p and block-sums R values r times: ~60 us per call at config 4, two calls per
// iteration and four in the tail.)
template <int R, int NT>
__device__ __forceinline__ void tall_gram_mfma(double* M, const double* A, const double* Bm, int n, double* part) {
    typedef double tg_v4 __attribute__((ext_vector_type(4)));
    constexpr int CT = R >= 16 ? R / 16 : 1, NW = NT / 64, NTILE = CT * CT, NSL = NW / NTILE;
    static_assert(NSL >= 1, "one wave per tile at least");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k4 = lane >> 4, c16 = lane & 15;
    const int tile = wave % NTILE, sl = wave / NTILE, pt = tile / CT, qt = tile % CT;
    const int steps = (n + 3) / 4, sps = (steps + NSL - 1) / NSL;
    const int s_lo = sl * sps, s_hi = s_lo + sps < steps ? s_lo + sps : steps;
    const int pa = 16 * pt + c16, pb = 16 * qt + c16;
    tg_v4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int s0 = s_lo; s0 < s_hi; s0 += 8) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = 4 * (s0 + u) + k4, cc = c < n ? c : n - 1;
            av[u] = pa < R ? A[(size_t)cc * R + pa] : 0.0;
            bv[u] = pb < R ? Bm[(size_t)cc * R + pb] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = 4 * (s0 + u) + k4 < n && s0 + u < s_hi;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? av[u] : 0.0, bv[u], acc, 0, 0, 0);
        }
    }
    if constexpr (NSL == 1) {
#pragma unroll
        for (int v = 0; v < 4; ++v) M[(16 * pt + k4 + 4 * v) * R + 16 * qt + c16] = acc[v];
        __syncthreads();
    } else {
#pragma unroll
        for (int v = 0; v < 4; ++v) part[(wave * 4 + v) * 64 + lane] = acc[v];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += part[(w * 4 + v) * 64 + lane];
                const int row = k4 + 4 * v;
                if (row < R && c16 < R) M[row * R + c16] = t;
            }
        }
        __syncthreads();
    }
}

// chol_lds with the rows of a column in parallel: the same operations in the same order per element (identical results), two
// barriers per column instead of one thread walking r^3 / 3 dependent LDS operations (0.1 ms per call at r = 20)
template <int R, int NT>
__device__ __forceinline__ void chol_lds_par(double* G, int r) {
    for (int j = 0; j < r; ++j) {
        if (threadIdx.x == 0) {
            double d = G[j * R + j];
            for (int k = 0; k < j; ++k) d -= G[j * R + k] * G[j * R + k];
            G[j * R + j] = sqrt(d);
        }
        __syncthreads();
        const double dj = G[j * R + j];
        for (int i = j + 1 + (int)threadIdx.x; i < r; i += NT) {
            double s = G[i * R + j];
            for (int k = 0; k < j; ++k) s -= G[i * R + k] * G[j * R + k];
            G[i * R + j] = s / dj;
        }
        __syncthreads();
    }
}

// Cyclic Jacobi on the symmetric r x r H (LDS, ld R); W receives the eigenvectors (columns), sorted by
// decreasing eigenvalue; ev[k] the eigenvalues.  Thread 0.
template <int R>
__device__ __forceinline__ void jacobi_lds(double* H, double* W, double* ev, int r) {
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < r; ++j) W[i * R + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < r; ++j) {
                if (i == j) dia += H[i * R + j] * H[i * R + j];
                else off += H[i * R + j] * H[i * R + j];
            }
        if (off <= 1e-32 * dia) break;
        for (int p = 0; p < r - 1; ++p)
            for (int q = p + 1; q < r; ++q) {
                const double hpq = H[p * R + q];
                if (hpq == 0.0) continue;
                const double theta = (H[q * R + q] - H[p * R + p]) / (2.0 * hpq);
                const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < r; ++k) {            // H <- H J
                    const double hkp = H[k * R + p], hkq = H[k * R + q];
                    H[k * R + p] = c * hkp - s * hkq;
                    H[k * R + q] = s * hkp + c * hkq;
                }
                for (int k = 0; k < r; ++k) {            // H <- J' H
                    const double hpk = H[p * R + k], hqk = H[q * R + k];
                    H[p * R + k] = c * hpk - s * hqk;
                    H[q * R + k] = s * hpk + c * hqk;
                }
                for (int k = 0; k < r; ++k) {            // W <- W J
                    const double wkp = W[k * R + p], wkq = W[k * R + q];
                    W[k * R + p] = c * wkp - s * wkq;
                    W[k * R + q] = s * wkp + c * wkq;
                }
            }
    }
    for (int k = 0; k < r; ++k) ev[k] = H[k * R + k];
    for (int a_ = 0; a_ < r - 1; ++a_) {                   // selection sort, descending
        int m = a_;
        for (int k = a_ + 1; k < r; ++k)
            if (ev[k] > ev[m]) m = k;
        if (m != a_) {
            const double t = ev[a_]; ev[a_] = ev[m]; ev[m] = t;
            for (int k = 0; k < r; ++k) { const double w = W[k * R + a_]; W[k * R + a_] = W[k * R + m]; W[k * R + m] = w; }
        }
    }
}

// The same decomposition by a whole WORKGROUP for the wide states (R = 16, 32), H and W in LDS: parallel ordering (R / 2 disjoint
// pairs per round), one thread per pair computes its rotation, then (row, pair) items apply H <- H J and W <- W J, then
// (pair, column) items H <- J' H -- three barriers per round instead of the single thread's ~100 k dependent LDS
// read-modify-writes per sweep (config 4: 116 ms of pca_kernel per 256 replicates).  cs: [R] doubles of LDS scratch.
template <int R, int NT>
__device__ __forceinline__ void jacobi_block(double* H, double* W, double* ev, double* cs, double* red, int r, int tid) {
    constexpr int RM = R - 1, NP = R / 2;
    // pair k of round t (round-robin: index R - 1 stays, the others move round a circle of R - 1): (t, R - 1) for k = 0, else
    // ((t + k) mod (R - 1), (t - k) mod (R - 1)); returned as x < y
    auto pair_of = [&](int k, int t, int& x, int& y) {
        const int a_ = k == 0 ? t : (t + k) % RM, b_ = k == 0 ? RM : (t - k + RM) % RM;
        x = a_ < b_ ? a_ : b_;
        y = a_ < b_ ? b_ : a_;
    };
    for (int e = tid; e < R * R; e += NT) W[e] = (e / R == e % R) ? 1.0 : 0.0;
    __syncthreads();
    for (int sweep = 0; sweep < 60; ++sweep) {
        double part[2] = {0.0, 0.0};                            // off-diagonal, diagonal sums of squares
        for (int e = tid; e < R * R; e += NT) {
            const int i = e / R, j = e % R;
            if (i < r && j < r) { const double h = H[e]; if (i == j) part[1] = fma(h, h, part[1]); else part[0] = fma(h, h, part[0]); }
        }
        block_sum<2, NT>(part, red);
        if (part[0] <= 1e-32 * part[1]) break;                  // (uniform: block_sum leaves the totals in every thread)
        for (int t = 0; t < RM; ++t) {
            if (tid < NP) {                                     // thread k: the rotation of pair k, from the H of the start of the round
                int x, y;
                pair_of(tid, t, x, y);
                const double hpq = H[x * R + y];
                const bool act = y < r && hpq != 0.0;
                const double theta = (H[y * R + y] - H[x * R + x]) / (2.0 * (act ? hpq : 1.0));
                const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cc = 1.0 / sqrt(tt * tt + 1.0);
                cs[2 * tid] = act ? cc : 1.0;
                cs[2 * tid + 1] = act ? tt * cc : 0.0;
            }
            __syncthreads();
            for (int it = tid; it < R * NP; it += NT) {         // (row i, pair k): columns x < y of H and of W
                const int i = it / NP, k = it % NP;
                int x, y;
                pair_of(k, t, x, y);
                const double c = cs[2 * k], sn = cs[2 * k + 1];
                const double hp = H[i * R + x], hq = H[i * R + y];
                H[i * R + x] = c * hp - sn * hq;
                H[i * R + y] = sn * hp + c * hq;
                const double wp = W[i * R + x], wq = W[i * R + y];
                W[i * R + x] = c * wp - sn * wq;
                W[i * R + y] = sn * wp + c * wq;
            }
            __syncthreads();
            for (int it = tid; it < R * NP; it += NT) {         // (pair k, column j): rows x < y of H
                const int j = it % R, k = it / R;
                int x, y;
                pair_of(k, t, x, y);
                const double c = cs[2 * k], sn = cs[2 * k + 1];
                const double hp = H[x * R + j], hq = H[y * R + j];
                H[x * R + j] = c * hp - sn * hq;
                H[y * R + j] = sn * hp + c * hq;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid == 0) {
        for (int k = 0; k < r; ++k) ev[k] = H[k * R + k];
        for (int a_ = 0; a_ < r - 1; ++a_) {                   // selection sort, descending
            int m = a_;
            for (int k = a_ + 1; k < r; ++k)
                if (ev[k] > ev[m]) m = k;
            if (m != a_) {
                const double t = ev[a_]; ev[a_] = ev[m]; ev[m] = t;
                for (int k = 0; k < r; ++k) { const double w = W[k * R + a_]; W[k * R + a_] = W[k * R + m]; W[k * R + m] = w; }
            }
        }
    }
    __syncthreads();
}

// Jacobi eigen-decomposition by ONE WAVE with an element of H and of W per lane (lane = R i + j, lanes >= R R idle): cross-lane
// fetches and a few FMAs per lane instead of ~100 dependent LDS read-modify-writes of one thread (the single-thread version
// took 3.5 of pca_kernel's 5.7 ms per 1024 replicates).  Same rotation formula, stopping rule and sorting as jacobi_lds; the
// ORDER of the rotations differs (parallel ordering, below) -- the decomposition it converges to is the same up to the sign
// and order conventions the caller fixes afterwards.  Called by wave 0 (all 64 lanes); H, W, ev in LDS.
template <int R>
__device__ __forceinline__ void jacobi_wave(double* H, double* W, double* ev, int r, int lane) {
    const int le = lane < R * R ? lane : 0;                     // (idle lanes shadow lane 0: valid indices, results unused)
    const int i = le / R, j = le % R;
    const bool in = lane < R * R && i < r && j < r;
    double h = in ? H[i * R + j] : 0.0;
    double w = (lane < R * R && i == j) ? 1.0 : 0.0;
    // PARALLEL ordering: a round rotates R / 2 disjoint index pairs at once (round-robin schedule: index R - 1 stays, the
    // others move round a circle of R - 1), R - 1 rounds per sweep; every lane computes the rotation of its column's pair and
    // of its row's pair itself, from the H of the start of the round (disjoint pairs: J'HJ with J the product of the R / 2
    // rotations).  The cyclic one-pair-at-a-time form was a chain of 28 x (9 dependent ds_bpermute + a divide and two square
    // roots) per sweep on one wave while seven wait: 0.24 ms of the 0.75 ms a replicate's start takes.
    constexpr int RM = R - 1;
    auto partner = [&](int x, int t) { return x == RM ? t : (x == t ? RM : (2 * t - x + 2 * RM) % RM); };
    auto rot = [&](int p, int q, double& c, double& sn) {   // rotation of pair (p < q) from the current h (lane-dependent p, q)
        const int pc = p < R ? p : 0, qc = q < R ? q : 0;
        const double hpq = __shfl(h, pc * R + qc, kWave);
        const double hpp = __shfl(h, pc * R + pc, kWave), hqq = __shfl(h, qc * R + qc, kWave);
        const bool act = q < r && hpq != 0.0;
        const double theta = (hqq - hpp) / (2.0 * (act ? hpq : 1.0));
        const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cc = 1.0 / sqrt(tt * tt + 1.0);
        c = act ? cc : 1.0;
        sn = act ? tt * cc : 0.0;
    };
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = (in && i != j) ? h * h : 0.0, dia = (in && i == j) ? h * h : 0.0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { off += __shfl_xor(off, o, kWave); dia += __shfl_xor(dia, o, kWave); }
        if (off <= 1e-32 * dia) break;
        for (int t = 0; t < (RM > 0 ? RM : 1); ++t) {
            const int pj = partner(j, t), pi = partner(i, t);
            double cc, sc, cr, sr;
            rot(j < pj ? j : pj, j < pj ? pj : j, cc, sc);      // the pair of this lane's column
            rot(i < pi ? i : pi, i < pi ? pi : i, cr, sr);      // ... and of its row
            // H <- H J : columns p and q of every row (own = h_ip for the lower index: c own - s partner; else s partner + c own)
            const double hc = __shfl(h, i * R + pj, kWave);
            h = j < pj ? cc * h - sc * hc : sc * hc + cc * h;
            // H <- J' H : rows p and q
            const double hr = __shfl(h, pi * R + j, kWave);
            h = i < pi ? cr * h - sr * hr : sr * hr + cr * h;
            // W <- W J
            const double wc = __shfl(w, i * R + pj, kWave);
            w = j < pj ? cc * w - sc * wc : sc * wc + cc * w;
        }
    }
    if (lane < R * R) { W[lane] = w; if (i == j) ev[i] = h; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
        for (int a_ = 0; a_ < r - 1; ++a_) {                   // selection sort, descending
            int m = a_;
            for (int k = a_ + 1; k < r; ++k)
                if (ev[k] > ev[m]) m = k;
            if (m != a_) {
                const double t = ev[a_]; ev[a_] = ev[m]; ev[m] = t;
                for (int k = 0; k < r; ++k) { const double ww = W[k * R + a_]; W[k * R + a_] = W[k * R + m]; W[k * R + m] = ww; }
            }
        }
    }
}

// Solve (leading r x r of) M Xs = Bs for Xs, M SPD, nb right-hand sides as columns of Bs; all LDS ld R;
// M is destroyed (Cholesky).  Thread 0.
template <int R>
__device__ __forceinline__ void spd_solve_lds(double* M, double* Bs, int r, int nb) {
    chol_lds<R>(M, r);
    for (int c = 0; c < nb; ++c) {
        for (int i = 0; i < r; ++i) {
            double s = Bs[i * R + c];
            for (int k = 0; k < i; ++k) s -= M[i * R + k] * Bs[k * R + c];
            Bs[i * R + c] = s / M[i * R + i];
        }
        for (int i = r - 1; i >= 0; --i) {
            double s = Bs[i * R + c];
            for (int k = i + 1; k < r; ++k) s -= M[k * R + i] * Bs[k * R + c];
            Bs[i * R + c] = s / M[i * R + i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The subspace iteration of pca_kernel with the basis in LDS and 512 threads (R <= 8, N <= 256): the 256-thread version
// re-read S row by row with one load in flight per thread and ran 16 two-barrier block reductions per iteration (5.7 ms per
// 1024 replicates at N = 200: ~100 us per iteration and replicate).  Here
//   Y = S V:   thread (part p of 2, row i) multiplies half of row i of S (8 independent coalesced loads in flight)
//              with the rows of V (LDS broadcast reads), the partial rows meet in LDS;
//   H = V'Y, G = Y'Y:  every row thread forms its 2 x R x R outer products, one 64-value transpose-reduce per wave and
//              matrix, 4 partial matrices per matrix meet in LDS -- one barrier for both;
//   residual, Cholesky of G (thread 0), V = Y L^-T per row.
// Same arithmetic, start and stopping rule as the generic loop; leaves the orthonormal basis in V (global), returns whether
// the residual test fired.  sH / sG: R x R LDS matrices of the caller; red: >= 32 doubles.
template <int R>
__device__ __forceinline__ bool pca_iterate_lds(const PcaArgs& a, const double* __restrict__ S, double* V, double* Y, int N,
                                                int r, double* sH, double* sG, double* red) {
    constexpr int NT = kPcaFastThreads, NPART = NT / 256, RR = R * R;
    extern __shared__ __attribute__((aligned(16))) double pl[];
    double* Vs = pl;                                           // [N][R]
    double* Ys = Vs + (size_t)N * R;                           // [N][R]
    double* part = Ys + (size_t)N * R;                         // [4 waves][2][RR] partial Gram matrices
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = tid >> 8, i = tid & 255;
    const int H4 = (N + NPART - 1) / NPART;
    const int j0 = p * H4, j1 = (j0 + H4 < N) ? j0 + H4 : N;
    const bool row = i < N;

    for (int e = tid; e < N * R; e += NT) Ys[e] = Y[e];        // the deterministic start
    __syncthreads();
    // grams[0] = A'B ... helper: G = Ys'Ys (and optionally H = Vs'Ys) into sG / sH
    auto grams = [&](bool with_h) {
        if (tid < 256) {
            double y[R], v[R];
#pragma unroll
            for (int k = 0; k < R; ++k) { y[k] = row ? Ys[(size_t)i * R + k] : 0.0; v[k] = (row && with_h) ? Vs[(size_t)i * R + k] : 0.0; }
            // R rows of R products at a time (R values: the 128-VGPR budget of a 1024-thread workgroup has no room for R x R)
            bool canon;
            const int idx = reduce_index<R>(lane, canon);
#pragma unroll
            for (int q = 0; q < R; ++q) {
                double pg[R];
#pragma unroll
                for (int k = 0; k < R; ++k) pg[k] = y[q] * y[k];
                wave_transpose_reduce<R>(pg, lane);
                if (canon) part[(wave * 2 + 0) * RR + q * R + idx] = pg[0];
                if (with_h) {
                    double ph[R];
#pragma unroll
                    for (int k = 0; k < R; ++k) ph[k] = v[q] * y[k];
                    wave_transpose_reduce<R>(ph, lane);
                    if (canon) part[(wave * 2 + 1) * RR + q * R + idx] = ph[0];
                }
            }
        }
        __syncthreads();
        if (tid < 2 * RR) {
            const int which = tid / RR, idx = tid % RR;
            const double t = part[(0 * 2 + which) * RR + idx] + part[(1 * 2 + which) * RR + idx] + part[(2 * 2 + which) * RR + idx]
                           + part[(3 * 2 + which) * RR + idx];
            if (which == 0) sG[idx] = t; else if (with_h) sH[idx] = t;
        }
        __syncthreads();
    };
    auto orthonormalise = [&]() {                               // Vs <- Ys L^-T  with  Ys'Ys = L L'  (sG holds Ys'Ys)
        if (tid == 0) chol_lds<R>(sG, r);
        __syncthreads();
        if (tid < 256 && row) {
            double y[R], v[R];
#pragma unroll
            for (int k = 0; k < R; ++k) { y[k] = Ys[(size_t)i * R + k]; v[k] = 0.0; }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (k < r) {
                    double sacc = y[k];
#pragma unroll
                    for (int m = 0; m < R; ++m)
                        if (m < k) sacc -= v[m] * sG[k * R + m];
                    v[k] = sacc / sG[k * R + k];
                }
            }
#pragma unroll
            for (int k = 0; k < R; ++k) Vs[(size_t)i * R + k] = v[k];
        }
        __syncthreads();
    };
    grams(false);
    orthonormalise();
    double best = 1e300;
    int stall = 0;
    bool converged = false;
    auto apply_S_lds = [&]() {                                  // Ys = S Vs
        {
            double acc[R];
#pragma unroll
            for (int k = 0; k < R; ++k) acc[k] = 0.0;
            if (row) {
                int j = j0;
                // (S comes from L2 every iteration -- 320 KB per replicate do not fit LDS: the batches keep 20, then 8 loads of a
                // column in flight; with 8 only, the ten dependent round trips were most of the 60 us an iteration took)
                for (; j + 20 <= j1; j += 20) {
                    double sv[20];
#pragma unroll
                    for (int u = 0; u < 20; ++u) sv[u] = S[(size_t)(j + u) * N + i];
#pragma unroll
                    for (int u = 0; u < 20; ++u) {
                        const double* vr = Vs + (size_t)(j + u) * R;
#pragma unroll
                        for (int k = 0; k < R; ++k) acc[k] = fma(sv[u], vr[k], acc[k]);
                    }
                }
                for (; j + 8 <= j1; j += 8) {
                    double sv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sv[u] = S[(size_t)(j + u) * N + i];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double* vr = Vs + (size_t)(j + u) * R;
#pragma unroll
                        for (int k = 0; k < R; ++k) acc[k] = fma(sv[u], vr[k], acc[k]);
                    }
                }
                for (; j < j1; ++j) {
                    const double sv = S[(size_t)j * N + i];
                    const double* vr = Vs + (size_t)j * R;
#pragma unroll
                    for (int k = 0; k < R; ++k) acc[k] = fma(sv, vr[k], acc[k]);
                }
            }
            // the two halves of a row meet in Ys itself (its old content is dead here): the second half's thread stores, the first
            // half's adds -- no [2][N][R] buffer of partial rows, which kept a workgroup at 51 KB of LDS (three per CU at most)
            static_assert(NPART == 2, "two half-row threads per row");
            if (row && p == 1) {
#pragma unroll
                for (int k = 0; k < R; ++k) Ys[(size_t)i * R + k] = acc[k];
            }
            __syncthreads();
            if (row && p == 0) {
#pragma unroll
                for (int k = 0; k < R; ++k) Ys[(size_t)i * R + k] = acc[k] + Ys[(size_t)i * R + k];
            }
        }
        __syncthreads();
    };
    for (int it = 0; it < a.max_iter; ++it) {
        apply_S_lds();
        grams(true);                                            // sG = Y'Y, sH = V'Y = V'S V
        // residual ||Y - V H||_F / ||Y||_F
        double pr[2] = {0.0, 0.0};
        if (tid < 256 && row) {
            double y[R], v[R];
#pragma unroll
            for (int k = 0; k < R; ++k) { y[k] = Ys[(size_t)i * R + k]; v[k] = Vs[(size_t)i * R + k]; }
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (k < r) {
                    double vh = 0.0;
#pragma unroll
                    for (int m = 0; m < R; ++m)
                        if (m < r) vh = fma(v[m], sH[m * R + k], vh);
                    const double d = y[k] - vh;
                    pr[0] = fma(d, d, pr[0]);
                    pr[1] = fma(y[k], y[k], pr[1]);
                }
            }
        }
        block_sum<2, NT>(pr, red);
        const double rel = sqrt(pr[0] / pr[1]);
        orthonormalise();
        if (rel <= 1e-14) { converged = true; break; }
        if (rel < 0.5 * best) { best = rel; stall = 0; }
        else if (++stall >= 8 && best < 1e-10) { converged = true; break; }
    }
    // H = V'S V of the FINAL basis for the Rayleigh-Ritz step of the caller, still from LDS (the caller's generic product
    // reads V from global memory: 0.31 ms per replicate, as long as the whole iteration)
    apply_S_lds();
    grams(true);
    for (int e = tid; e < N * R; e += NT) V[e] = Vs[e];
    __syncthreads();
    return converged;
}

// (the 512-thread instantiations: 4 waves per SIMD = two workgroups per CU.  At 169 VGPRs a CU held ONE workgroup, so the 1024
// replicates of config 2 ran as four rounds of a kernel whose every stage is a latency chain)
template <int R, int NT>
__global__ __launch_bounds__(NT, (NT == kPcaFastThreads ? 4 : 1)) void pca_kernel(PcaArgs a) {
    __shared__ double sH[R * R], sW[R * R], sG[R * R], sM[R * R], sev[R], sred[(NT / 64) * (R > 2 ? R : 2)], sflag[2];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int N = a.N, T = a.T, r = a.r;
    constexpr bool kFastLds = NT == kPcaFastThreads && R <= 8;      // the LDS-resident iteration (its own Gram / Cholesky code)
    __shared__ double sPart[kFastLds || R >= 32 ? 1 : (NT / 64) * 256];   // partial tiles of tall_gram_mfma (R <= 16)
    auto tgram = [&](double* M, const double* A_, const double* B_, int n) {
        if constexpr (kFastLds) tall_gram<R, NT>(M, A_, B_, n, r, sred);
        else tall_gram_mfma<R, NT>(M, A_, B_, n, sPart);
    };
    const double* __restrict__ X = a.panel + (size_t)b * T * N;
    const double* __restrict__ S = a.S + (size_t)b * N * N;
    double* V = a.V + (size_t)b * N * R;           // [N][R] current basis (columns >= r are zero)
    double* Y = a.Y + (size_t)b * N * R;
    double* F = a.F + (size_t)b * T * R;           // [T][R] scores

    // deterministic, replicate-independent start: a fixed hash of (i, k), columns >= r zero
    for (int idx = tid; idx < N * R; idx += NT) {
        const int i = idx / R, k = idx % R;
        unsigned hsh = (unsigned)(i * 73856093u) ^ (unsigned)((k + 1) * 19349663u);
        hsh ^= hsh >> 13; hsh *= 0x5bd1e995u; hsh ^= hsh >> 15;
        Y[idx] = (k < r) ? ((double)(hsh & 0xFFFFu) / 65536.0 - 0.5) : 0.0;
    }
    __syncthreads();

    auto orthonormalise = [&]() {                   // V <- Y L^-T  with  Y'Y = L L'  (Cholesky QR)
        tgram(sG, Y, Y, N);
        if constexpr (kFastLds) { if (tid == 0) chol_lds<R>(sG, r); __syncthreads(); }
        else chol_lds_par<R, NT>(sG, r);
        for (int i = tid; i < N; i += NT) {
            double y[R], v[R];
#pragma unroll
            for (int k = 0; k < R; ++k) { y[k] = Y[(size_t)i * R + k]; v[k] = 0.0; }
            for (int k = 0; k < r; ++k) {           // forward substitution on the row: v L' = y
                double s = y[k];
                for (int m = 0; m < k; ++m) s -= v[m] * sG[k * R + m];
                v[k] = s / sG[k * R + k];
            }
#pragma unroll
            for (int k = 0; k < R; ++k) V[(size_t)i * R + k] = v[k];
        }
        __syncthreads();
    };
    // Y <- S V on the matrix pipe: wave w takes the row tiles w, w + NT / 64, ...; per step of 4 columns of S one load of A
    // (S[k][row] = S[row][k]: 128 contiguous bytes per k) and one of B per 16-column tile of V, 8 steps in flight.  (The VALU
    // form -- thread = row, 32 loads of V per 32 FMAs -- took 5 ms per iteration at config 4: 13 GFLOP/s per CU.)
    typedef double pca_v4 __attribute__((ext_vector_type(4)));
    auto apply_S = [&]() {
        constexpr int CT = R >= 16 ? R / 16 : 1;
        const int lane = tid & 63, wave = tid >> 6, k4 = lane >> 4, c16 = lane & 15;
        const int nrt = (N + 15) / 16, steps = (N + 3) / 4;
        for (int rt = wave; rt < nrt; rt += NT / 64) {
            pca_v4 acc[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct] = pca_v4{0.0, 0.0, 0.0, 0.0};
            const int row = 16 * rt + c16, rowc = row < N ? row : N - 1;
            for (int s0 = 0; s0 < steps; s0 += 8) {
                double av[8], bv[CT][8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = 4 * (s0 + u) + k4, cc = c < N ? c : N - 1;
                    av[u] = S[(size_t)cc * N + rowc];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) bv[ct][u] = (16 * ct + c16 < R) ? V[(size_t)cc * R + 16 * ct + (c16 < R ? c16 : 0)] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double a_ = (4 * (s0 + u) + k4 < N) ? av[u] : 0.0;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, bv[ct][u], acc[ct], 0, 0, 0);
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int v = 0; v < 4; ++v) {                 // D[(l / 16) + 4 v][l % 16]
                    const int rr = 16 * rt + k4 + 4 * v, col = 16 * ct + c16;
                    if (rr < N && col < R) Y[(size_t)rr * R + col] = acc[ct][v];
                }
        }
        __syncthreads();
    };

    bool converged = false;
    if constexpr (NT == kPcaFastThreads && R <= 8) {
        converged = pca_iterate_lds<R>(a, S, V, Y, N, r, sH, sG, sred);
    } else {
    orthonormalise();
    // iterate until the invariant-subspace residual ||S V - V (V'S V)||_F / ||S V||_F reaches the fp64
    // floor (or stops improving): the Ritz vectors taken afterwards are then exact to roundoff / gap
    double best = 1e300;
    int stall = 0;
    for (int it = 0; it < a.max_iter; ++it) {
        apply_S();
        tgram(sH, V, Y, N);                                  // H = V'S V
        double part[2] = {0.0, 0.0};
        for (int i = tid; i < N; i += NT) {
            for (int k = 0; k < r; ++k) {
                double vh = 0.0;
                for (int m = 0; m < r; ++m) vh = fma(V[(size_t)i * R + m], sH[m * R + k], vh);
                const double y = Y[(size_t)i * R + k], d = y - vh;
                part[0] = fma(d, d, part[0]);
                part[1] = fma(y, y, part[1]);
            }
        }
        block_sum<2, NT>(part, sred);
        const double rel = sqrt(part[0] / part[1]);
        orthonormalise();
        if (rel <= 1e-14) { converged = true; break; }
        if (rel < 0.5 * best) { best = rel; stall = 0; }
        else if (++stall >= 8 && best < 1e-10) { converged = true; break; }
    }
    }
    if (a.stop_after == 1) return;
    // max_iter exhausted above the tolerance (near-degenerate spectrum at the cut: the rate is lambda_{r+1} / lambda_r):
    // the basis is NOT the reference's svd-based pca_score (dfm_functions.ipynb:179-183) -- say so instead of returning it
    if (!converged && tid == 0 && a.status) atomicOr(a.status, 2);
    // Rayleigh-Ritz: H = V'SV, H = W Theta W', V <- V W (descending), sign rule of the oracle
    if constexpr (!(NT == kPcaFastThreads && R <= 8)) {       // (the fast path left H in sH)
        apply_S();
        tgram(sH, V, Y, N);
    }
    if (tid == 0) {
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < i; ++j) { const double h = 0.5 * (sH[i * R + j] + sH[j * R + i]); sH[i * R + j] = h; sH[j * R + i] = h; }
    }
    __syncthreads();
    if constexpr (R <= 8) {
        if (tid < 64) jacobi_wave<R>(sH, sW, sev, r, tid);
    } else {
        jacobi_block<R, NT>(sH, sW, sev, sM, sred, r, tid);     // (sM: scratch for the rotations of a round, 2 doubles per pair)
    }
    __syncthreads();
    for (int i = tid; i < N; i += NT) {
        double v[R], w[R];
#pragma unroll
        for (int k = 0; k < R; ++k) { v[k] = V[(size_t)i * R + k]; w[k] = 0.0; }
        for (int k = 0; k < r; ++k) {
            double s = 0.0;
            for (int m = 0; m < r; ++m) s = fma(v[m], sW[m * R + k], s);
            w[k] = s;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) Y[(size_t)i * R + k] = w[k];    // rotated basis in Y
    }
    __syncthreads();
    if (a.stop_after == 2) return;
    // sign: largest-|.| entry of each eigenvector positive (first such entry on ties, as numpy argmax)
    for (int k = 0; k < r; ++k) {
        double best = -1.0; int bi = N;
        for (int i = tid; i < N; i += NT) {
            const double av = fabs(Y[(size_t)i * R + k]);
            if (av > best) { best = av; bi = i; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ob = __shfl_xor(best, off, kWave);
            const int oi = __shfl_xor(bi, off, kWave);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { sred[2 * (tid >> 6)] = best; sred[2 * (tid >> 6) + 1] = (double)bi; }
        __syncthreads();
        if (tid == 0) {
            double bb = sred[0]; int ii = (int)sred[1];
            for (int w = 1; w < NT / 64; ++w)
                if (sred[2 * w] > bb || (sred[2 * w] == bb && (int)sred[2 * w + 1] < ii)) { bb = sred[2 * w]; ii = (int)sred[2 * w + 1]; }
            sflag[0] = (Y[(size_t)ii * R + k] < 0.0) ? -1.0 : 1.0;
        }
        __syncthreads();
        const double sg = sflag[0];
        for (int i = tid; i < N; i += NT) V[(size_t)i * R + k] = sg * Y[(size_t)i * R + k];
        __syncthreads();
    }
    for (int idx = tid; idx < N * R; idx += NT)
        if (idx % R >= r) V[idx] = 0.0;
    __syncthreads();

    if (a.stop_after == 3) return;
    // scores F = X V  (one wave per period, lanes over series)
    if constexpr (R <= 8) {
        // the lane's series (i = lane + 64 u) keep their rows of V in registers; the R sums of a period by one transpose-reduce
        const int lane = tid & 63, wave = tid >> 6;
        constexpr int NU = 4;                                   // N <= 256
        if (N <= 64 * NU) {
            double vr[NU][R];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = lane + 64 * u;
#pragma unroll
                for (int k = 0; k < R; ++k) vr[u][k] = i < N ? V[(size_t)i * R + k] : 0.0;
            }
            bool canon;
            const int idx = reduce_index<R>(lane, canon);
            for (int t = wave; t < T; t += NT / 64) {
                double f[R];
#pragma unroll
                for (int k = 0; k < R; ++k) f[k] = 0.0;
                double x[NU];
#pragma unroll
                for (int u = 0; u < NU; ++u) { const int i = lane + 64 * u; x[u] = i < N ? X[(size_t)t * N + i] : 0.0; }
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int k = 0; k < R; ++k) f[k] = fma(x[u], vr[u][k], f[k]);
                wave_transpose_reduce<R>(f, lane);
                if (canon) F[(size_t)t * R + idx] = f[0];
            }
        } else {
            {   // scores on the matrix pipe: tile = 16 periods; A[i][k] = x[t0 + i][c + k] (32-byte pieces of 16 rows: the next three
                // steps hit the same lines), B[k][j] = V[c + k][16 ct + j].  (wave per period with 32 loads of V per series: 15 ms at config 4)
                constexpr int CT = R >= 16 ? R / 16 : 1;
                const int lane = tid & 63, wave = tid >> 6, k4 = lane >> 4, c16 = lane & 15;
                const int ntt = (T + 15) / 16, steps = (N + 3) / 4;
                for (int tt = wave; tt < ntt; tt += NT / 64) {
                    pca_v4 acc[CT];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[ct] = pca_v4{0.0, 0.0, 0.0, 0.0};
                    const int trow = 16 * tt + c16, trc = trow < T ? trow : T - 1;
                    for (int s0 = 0; s0 < steps; s0 += 8) {
                        double av[8], bv[CT][8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int c = 4 * (s0 + u) + k4, cc = c < N ? c : N - 1;
                            av[u] = X[(size_t)trc * N + cc];
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct) bv[ct][u] = (16 * ct + c16 < R) ? V[(size_t)cc * R + 16 * ct + (c16 < R ? c16 : 0)] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const double a_ = (4 * (s0 + u) + k4 < N) ? av[u] : 0.0;
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, bv[ct][u], acc[ct], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int t = 16 * tt + k4 + 4 * v, col = 16 * ct + c16;
                            if (t < T && col < R) F[(size_t)t * R + col] = acc[ct][v];
                        }
                }
            }
        }
    } else {
        {   // scores on the matrix pipe: tile = 16 periods; A[i][k] = x[t0 + i][c + k] (32-byte pieces of 16 rows: the next three
            // steps hit the same lines), B[k][j] = V[c + k][16 ct + j].  (wave per period with 32 loads of V per series: 15 ms at config 4)
            constexpr int CT = R >= 16 ? R / 16 : 1;
            const int lane = tid & 63, wave = tid >> 6, k4 = lane >> 4, c16 = lane & 15;
            const int ntt = (T + 15) / 16, steps = (N + 3) / 4;
            for (int tt = wave; tt < ntt; tt += NT / 64) {
                pca_v4 acc[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[ct] = pca_v4{0.0, 0.0, 0.0, 0.0};
                const int trow = 16 * tt + c16, trc = trow < T ? trow : T - 1;
                for (int s0 = 0; s0 < steps; s0 += 8) {
                    double av[8], bv[CT][8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = 4 * (s0 + u) + k4, cc = c < N ? c : N - 1;
                        av[u] = X[(size_t)trc * N + cc];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) bv[ct][u] = (16 * ct + c16 < R) ? V[(size_t)cc * R + 16 * ct + (c16 < R ? c16 : 0)] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double a_ = (4 * (s0 + u) + k4 < N) ? av[u] : 0.0;
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, bv[ct][u], acc[ct], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int t = 16 * tt + k4 + 4 * v, col = 16 * ct + c16;
                        if (t < T && col < R) F[(size_t)t * R + col] = acc[ct][v];
                    }
            }
        }
    }
    __syncthreads();
    if (a.stop_after == 4) return;
    if (a.factors) {
        for (int idx = tid; idx < T * r; idx += NT) {
            const int t = idx / r, k = idx % r;
            a.factors[((size_t)b * T + t) * r + k] = F[(size_t)t * R + k];
        }
    }
    // Lam = V_r;  R_i = (S_ii - sum_k theta_k V_ik^2) / T, with theta_k = ||F_k||^2 (= Ritz value)
    constexpr bool kTail8 = kFastLds && R == 8;               // the start's tail on ONE pass over the scores + one wave (below)
    __shared__ double sTail[kTail8 ? 8 * 128 + 2 * 8 * kTileStride<8> : 1];
    if constexpr (kTail8) {
        // F'F and F0'F1 in ONE pass over the scores: lane (i, j) = l / 8, l % 8 of wave w sums f_t[i] f_t[j] and f_t[i] f_t+1[j] over
        // t = w, w + 8, ...; the eight waves' partial sums meet in LDS.  (The four tall_gram calls of the first version -- F'F,
        // F0'F0, F0'F1, F1'F1, each r column passes with a block reduction -- and the single-thread solve behind them were
        // 0.52 of the kernel's 1.82 ms; F0'F0 = F'F - f_T f_T' and F1'F1 = F'F - f_1 f_1' need no pass of their own.)
        const int lane = tid & 63, wave = tid >> 6, ii = lane >> 3, jj = lane & 7;
        double g = 0.0, wv = 0.0;
        for (int t = wave; t < T; t += NT / 64) {
            const double fi = F[(size_t)t * R + ii];
            g = fma(fi, F[(size_t)t * R + jj], g);
            if (t + 1 < T) wv = fma(fi, F[(size_t)(t + 1) * R + jj], wv);
        }
        sTail[wave * 128 + lane] = g;
        sTail[wave * 128 + 64 + lane] = wv;
        __syncthreads();
        if (tid < 128) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) t += sTail[w * 128 + tid];
            if (tid < 64) sG[tid] = t; else sW[tid - 64] = t;      // sG = F'F, sW = F0'F1  ([p][q] = sum_t F[t][p] F[t+1][q])
        }
        __syncthreads();
    } else {
        tgram(sG, F, F, T);                                       // F'F
    }
    for (int i = tid; i < N; i += NT) {
        double q = 0.0;
        for (int k = 0; k < r; ++k) {
            const double v = V[(size_t)i * R + k];
            a.Lam[((size_t)b * N + i) * r + k] = v;
            // residual of series i: ||x_i||^2 - 2 x_i'F lam_i + lam_i'F'F lam_i, with F'x_i = (F'F) lam_i
            double s = 0.0;
            for (int m = 0; m < r; ++m) s = fma(sG[k * R + m], V[(size_t)i * R + m], s);
            q = fma(v, s, q);
        }
        a.Rv[(size_t)b * N + i] = (S[(size_t)i * N + i] - q) / (double)T;
    }
    if (a.stop_after == 5) return;
    if constexpr (kTail8) {
        // VAR(1) of F without constant on wave 0, an element of every 8 x 8 matrix per lane (dfm_grid.h): A' = (F0'F0)^-1 F0'F1 by the
        // symmetric sweep inverse, Q = sym(F1'F1 - A F0'F1) / (T - 1) (= e'e at the OLS solution), P0 = sym(F'F) / T, mu0 = 0
        if (tid < 64) {
            constexpr int TS = kTileStride<8>;
            double* L0 = sTail + 8 * 128;
            double* L1 = L0 + 8 * TS;
            Grid<8> G8;
            const int i = tid >> 3, j = tid & 7;
            G8.l = tid; G8.i = i; G8.j = j;
            const bool in = i < r && j < r;
            const double eye = (i == j) ? 1.0 : 0.0;
            const double Gel = sG[tid], Wel = in ? sW[tid] : 0.0;
            const double fLi = F[(size_t)(T - 1) * R + i], fLj = F[(size_t)(T - 1) * R + j];
            const double f0i = F[i], f0j = F[j];
            double Hinv = in ? fma(-fLi, fLj, Gel) : eye;              // F0'F0 = F'F - f_T f_T'  (identity on the padding)
            const double Mel = in ? fma(-f0i, f0j, Gel) : 0.0;         // F1'F1 = F'F - f_1 f_1'
            (void)G8.sweep_inverse(Hinv);
            wave_lds_sync();
            L0[TS * j + i] = Wel;                                      // W' rows
            L1[TS * i + j] = Hinv;                                     // (symmetric)
            wave_lds_sync();
            const double An = dot_rows<8>(L0, L1, i, j);               // A[i][j] = sum_k W[k][i] Hinv[k][j]
            wave_lds_sync();
            L1[TS * i + j] = An;
            wave_lds_sync();
            double Qn = (Mel - dot_rows<8>(L1, L0, i, j)) / (double)(T - 1);   // (A W)[i][j] = sum_k A[i][k] W[k][j]
            Qn = 0.5 * (Qn + G8.transposed(Qn));
            const double P0n = 0.5 * (Gel + G8.transposed(Gel)) / (double)T;
            if (in) {
                a.A[(size_t)b * r * r + i * r + j] = An;
                a.Q[(size_t)b * r * r + i * r + j] = Qn;
                a.P0[(size_t)b * r * r + i * r + j] = P0n;
            }
            if (i == 0 && j < r) a.mu0[(size_t)b * r + j] = 0.0;
        }
        return;
    }
    // VAR(1) of F without constant: A = (F0'F0)^-1 F0'F1 (transposed), Q = e'e / (T-1)
    tgram(sH, F, F, T - 1);                                           // F0'F0
    tgram(sW, F, F + R, T - 1);                                       // F0'F1   ([p][q] = sum_t F[t][p] F[t+1][q])
    tgram(sM, F + R, F + R, T - 1);                                   // F1'F1
    if (tid == 0) {
        double* Ao = a.A + (size_t)b * r * r;
        double* Qo = a.Q + (size_t)b * r * r;
        double* P0o = a.P0 + (size_t)b * r * r;
        for (int i = 0; i < r; ++i) {
            a.mu0[(size_t)b * r + i] = 0.0;
            for (int j = 0; j < r; ++j) P0o[i * r + j] = 0.5 * (sG[i * R + j] + sG[j * R + i]) / (double)T;
        }
        // keep F0'F1 in sev-free storage: copy to sG (F'F no longer needed)
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < r; ++j) sG[i * R + j] = sW[i * R + j];
        spd_solve_lds<R>(sH, sW, r, r);                           // sW <- (F0'F0)^-1 F0'F1 = A'   ([p][q]: A[q][p])
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < r; ++j) Ao[i * r + j] = sW[j * R + i];
        // e'e = F1'F1 - A (F0'F1) - (F0'F1)' A' + A (F0'F0) A' = F1'F1 - A (F0'F1)   at the OLS solution
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < r; ++j) {
                double s = sM[i * R + j];
                for (int k = 0; k < r; ++k) s -= sW[k * R + i] * sG[k * R + j];   // A[i][k] (F0'F1)[k][j]
                sH[i * R + j] = s / (double)(T - 1);
            }
        for (int i = 0; i < r; ++i)
            for (int j = 0; j < r; ++j) Qo[i * r + j] = 0.5 * (sH[i * R + j] + sH[j * R + i]);
    }
}

// ---------------------------------------------------------------------------------------------
template <int MAXP>
static hipError_t launch_gx_dma(const PcaArgs& a, hipStream_t s) {
    const size_t lds = (size_t)2 * kGxPB * gx_dma_ld(a.N) * sizeof(double);
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_xx_dma_kernel<MAXP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((gram_xx_dma_kernel<MAXP>), dim3(a.B), dim3(kGxThreads), lds, s, a);
    return hipGetLastError();
}

template <int MAXP>
static hipError_t launch_gx_mfma(const PcaArgs& a, hipStream_t s) {
    static const int no_dma = [] { const char* v = diag_env("DFM_GRAM_XX_NO_DMA"); return v ? atoi(v) : 0; }();   // A/B: the register-staged kernel
    if ((a.N & 1) == 0 && !no_dma) return launch_gx_dma<MAXP>(a, s);
    const int NT = (a.N + 15) / 16;
    const size_t lds = (size_t)2 * kGxPB * (NT * 16 + 2) * sizeof(double);
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_xx_mfma_kernel<MAXP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((gram_xx_mfma_kernel<MAXP>), dim3(a.B), dim3(kGxThreads), lds, s, a);
    return hipGetLastError();
}

// variant: 0 = matrix pipe where the shape allows it (N <= 256), 1 = the VALU kernel (DFM_GRAM_XX_VALU=1; N > 256)
hipError_t launch_gram_xx(const PcaArgs& a, hipStream_t s, int variant) {
    const int NT = (a.N + 15) / 16;
    const int per_wave = (NT * (NT + 1) / 2 + 7) / 8;
    if (variant == 0 && a.N <= 256 && per_wave <= 17) note_kernel((a.N & 1) == 0 ? "gram_xx_dma_kernel" : "gram_xx_mfma_kernel");
    if (variant == 0 && a.N <= 256) {
        if (per_wave <= 4) return launch_gx_mfma<4>(a, s);
        if (per_wave <= 8) return launch_gx_mfma<8>(a, s);
        if (per_wave <= 12) return launch_gx_mfma<12>(a, s);
        if (per_wave <= 17) return launch_gx_mfma<17>(a, s);
    }
    if (variant == 0 && gram_xx_wide_supported(a.N)) return launch_gram_xx_wide(a, s);   // even N > 256: gram_xx_wide.hip
    hipLaunchKernelGGL(gram_xx_kernel, dim3(a.B), dim3(kPcaThreads), 0, s, a);
    return hipGetLastError();
}

template <int R>
static hipError_t launch_pca_fast(const PcaArgs& a, hipStream_t s) {
    const size_t lds = ((size_t)2 * a.N * R + 8 * R * R) * sizeof(double);   // Vs, Ys, 4 x 2 partial Gram matrices
    static LdsOptIn attr_done;
    if (!attr_done && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pca_kernel<R, kPcaFastThreads>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);   // (+ the kernel's static LDS)
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((pca_kernel<R, kPcaFastThreads>), dim3(a.B), dim3(kPcaFastThreads), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_pca(int Rpad, const PcaArgs& a, hipStream_t s) {
    static const bool slow = [] { const char* v = diag_env("DFM_PCA_GENERIC"); return v && atoi(v) != 0; }();
    if (!slow && a.N <= 256 && a.N * Rpad >= 64 && Rpad <= 8) {   // the LDS-resident iteration (1024 threads per replicate)
        switch (Rpad) {
            case 2: return launch_pca_fast<2>(a, s);
            case 4: return launch_pca_fast<4>(a, s);
            case 8: return launch_pca_fast<8>(a, s);
        }
    }
    switch (Rpad) {
        case 2: hipLaunchKernelGGL((pca_kernel<2, kPcaThreads>), dim3(a.B), dim3(kPcaThreads), 0, s, a); break;
        case 4: hipLaunchKernelGGL((pca_kernel<4, kPcaThreads>), dim3(a.B), dim3(kPcaThreads), 0, s, a); break;
        case 8: hipLaunchKernelGGL((pca_kernel<8, kPcaThreads>), dim3(a.B), dim3(kPcaThreads), 0, s, a); break;
        case 16: hipLaunchKernelGGL((pca_kernel<16, kPcaThreads>), dim3(a.B), dim3(kPcaThreads), 0, s, a); break;
        case 32: hipLaunchKernelGGL((pca_kernel<32, kPcaThreads>), dim3(a.B), dim3(kPcaThreads), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace dfm
