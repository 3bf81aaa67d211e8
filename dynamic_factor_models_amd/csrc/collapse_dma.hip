// collapse_dma.hip -- the balanced-panel collapse at HBM speed: panel rows stream HBM -> LDS with
// the gfx950 LDS-DMA (`global_load_lds_dwordx4`), never passing through VGPRs, so the number of
// bytes a wave keeps in flight is set by its LDS ring (NS rows), not by its register budget.
//
//     b_t = sum_i lam_i x_it / R_i   (r)        sum_t s_t,  s_t = sum_i x_it^2 / R_i   
//
// The panel must have no missing cell on this path (a NaN makes s_t NaN, which raises status
// bit 0; panels with NaN take collapse_kernel in collapse.hip, which also emits n_t, ld_t, C_t).
// Reference counterpart: forming Lambda' x_t in the per-period regression of x_t on Lambda
// (dfm_functions.ipynb:271-286 called from :364).
//
// Mapping (wave64): one workgroup of 4 waves per replicate; wave w owns the contiguous periods
// [w T/4, (w+1) T/4) -- one contiguous byte stream, moved in full 1-KiB DMA pieces (64 lanes x 16 B)
// into a power-of-two ring regardless of where the 8N-byte rows fall.  Lane l owns columns
// {2l, 2l+1} + 128 j and keeps W[c][k] = lam_ck / R_c in registers.  The wave waits with a counted
// `s_waitcnt vmcnt` only for the pieces under the next RB rows, reads them back with conflict-free
// ds_read_b128 (address = stream offset mod ring), re-arms the freed pieces, and reduces the RB r
// per-lane partial sums across lanes with the transpose-reduce of dfm_device.h.  s_t is only ever
// needed summed over t: each lane accumulates sum_t x_it^2 for its own columns and the wave folds
// sum_i q_i / R_i once at the end (ssum[b][wave]).
//
// The companion gram_kernel computes the data-independent C = Lam' R^-1 Lam and sum log R_i per
// replicate (one wave each) for the covariance recursion.
#include "dfm_gram.h"
#include "dfm_kernels.h"

namespace dfm {

using lds_char_ptr = __attribute__((address_space(3))) char*;

// One LDS-DMA: every active lane moves 16 B from its own global address to lds_dst + 16 * lane.
// M0 carries the wave-uniform LDS byte address; it is compiler-reserved, so it is saved and
// restored inside the statement (cdna_hip_programming.md §5.7).  hipcc does not count this load.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
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
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(K) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// OPT bits (tuning, measured with scripts/microbench/colbw.hip): 1 = wave segments start on 128-byte
// boundaries (rows per wave rounded to a multiple of 128 / gcd(8N, 128)); 2 = the first ring fill is
// issued before the loading weights are fetched; 4 = no filler DMAs past the end of the segment (the
// last row blocks drain with vmcnt(0) instead).
template <int R, int CPL2, int RB, int NP, int ABL = 0, int OPT = 0>
__global__ __launch_bounds__(256) void collapse_dma_kernel(CollapseArgs a) {
    // ring of NP 1-KiB pieces per wave; a row block spans at most RB * CPL2 + 1 pieces
    static_assert((NP & (NP - 1)) == 0 && NP >= 2 * RB * CPL2, "ring: power of two, two row blocks deep");
    constexpr int KWAIT = NP - RB * CPL2 - 1;   // pieces that may still be in flight when a row block is read
    static_assert(KWAIT >= 1 && KWAIT <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x + a.b0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = a.N, T = a.T;
    const unsigned rowB = (unsigned)N * 8u;                    // N even: multiple of 16
    const double* __restrict__ L = a.Lam + (size_t)b * N * R;
    const double* __restrict__ Rv = a.Rv + (size_t)b * N;

    // this wave's periods [ta, tb): one contiguous byte stream of the panel
    int tq = (T + 3) / 4;
    if constexpr ((OPT & 1) != 0) {            // segment starts on 128-byte boundaries
        unsigned g = rowB & 127u;              // gcd(rowB, 128) for rowB a multiple of 16
        g = g == 0 ? 128u : (g & (~g + 1u));
        const int m = (int)(128u / g);
        tq = ((tq + m - 1) / m) * m;
    }
    const int ta = (wave * tq < T) ? wave * tq : T;
    const int tb = (ta + tq < T) ? ta + tq : T;
    const int nrows = tb - ta;
    const int nblk = (nrows + RB - 1) / RB;
    const unsigned segB = (unsigned)nrows * rowB;
    const int npiece = (int)((segB + 1023u) / 1024u);
    const char* __restrict__ seg = reinterpret_cast<const char*>(a.panel + ((size_t)b * T + ta) * N);

    constexpr unsigned kRingB = NP * 1024u;
    const char* ring = smem + (size_t)wave * kRingB;
    const unsigned ring_lds =
        __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_ptr)(smem)) + (unsigned)wave * kRingB;
    const unsigned lane16 = 16u * lane;
    bool act[CPL2];
#pragma unroll
    for (int j = 0; j < CPL2; ++j) act[j] = (128 * j + 2 * lane) < N;

    // piece p -> ring slot p mod NP.  Without OPT&4, pieces past the end of the segment re-load the last
    // piece: they only keep the number of in-flight DMAs constant so that one fixed vmcnt threshold is valid
    // to the end.  With OPT&4 nothing is issued past the end and the tail blocks drain with vmcnt(0).
    auto issue_piece = [&](int p) {
        if constexpr ((OPT & 4) != 0) {
            if (p >= npiece) return;
        }
        const int pc = p < npiece ? p : npiece - 1;
        const unsigned off = (unsigned)pc * 1024u + lane16;
        if (off < segB) dma16(seg + off, __builtin_amdgcn_readfirstlane(ring_lds + ((unsigned)p & (NP - 1)) * 1024u));
    };
    int issued = 0;
    if constexpr ((OPT & 2) != 0) {
        if (nrows > 0) {
#pragma unroll
            for (int s = 0; s < NP; ++s) issue_piece(issued++);
        }
    }

    double W[CPL2][2][R];
    double Ri[CPL2][2];
#pragma unroll
    for (int j = 0; j < CPL2; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 2 * lane + 128 * j + e;
            const bool own = c < N;
            const int cc = own ? c : N - 1;       // clamped: unconditional loads, no branch per element
            const double ri = own ? 1.0 / Rv[cc] : 0.0;
            Ri[j][e] = ri;
#pragma unroll
            for (int k = 0; k < R; ++k) W[j][e][k] = L[(size_t)cc * R + k] * ri;
        }
    // (OPT & 2: the weights were fetched behind the first ring fill; loads return in order, so the counted
    // wait of the first row block -- at most KWAIT of the YOUNGEST operations outstanding -- still implies
    // that every piece of the first fill has landed)

    if (nrows <= 0) {
        if (lane == 0) a.ssum[(size_t)b * kSsumSlots + wave] = 0.0;
        return;
    }
    if constexpr ((OPT & 2) == 0) {
#pragma unroll
        for (int s = 0; s < NP; ++s) issue_piece(issued++);
    }

    constexpr int NV = RB * R;
    bool canon;
    const int myidx = reduce_index<NV>(lane, canon);
    const int my_rr = myidx / R, my_k = myidx % R;
    double q[CPL2][2];                     // per-column sum of x^2 over this wave's periods
#pragma unroll
    for (int j = 0; j < CPL2; ++j) { q[j][0] = 0.0; q[j][1] = 0.0; }

    for (int blk = 0; blk < nblk; ++blk) {
        const int r0 = blk * RB;
        // all pieces below ceil((r0 + RB) rowB / 1024) have landed once at most KWAIT younger DMAs are
        // outstanding (loads complete in order; outstanding stores only make this wait longer)
        // The row-block store of b_t is a VMEM operation too and sits between the re-arm DMAs of successive
        // blocks: with OPT&8 the counted wait allows for the (at most two) stores younger than the pieces this
        // block needs -- gfx9 returns loads and stores of one wave in issue order on the single vmcnt counter.
        auto counted_wait = [&]() {
            if constexpr ((OPT & 8) != 0 && KWAIT + 2 <= 63) {
                if (blk >= 2) wait_vmcnt<KWAIT + 2>();
                else if (blk == 1) wait_vmcnt<KWAIT + 1>();
                else wait_vmcnt<KWAIT>();
            } else {
                wait_vmcnt<KWAIT>();
            }
        };
        if constexpr ((OPT & 4) != 0) {
            if (issued >= npiece) wait_vmcnt<0>();        // wave-uniform: every real piece is issued, drain
            else counted_wait();
        } else {
            counted_wait();
        }
        double2 xs[RB][CPL2];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            int r = r0 + rr;
            r = r < nrows ? r : nrows - 1;                       // clamped duplicate rows are never stored
            const unsigned rbase = (unsigned)r * rowB + lane16;
#pragma unroll
            for (int j = 0; j < CPL2; ++j) {
                const unsigned o = (rbase + 1024u * j) & (kRingB - 1);
                xs[rr][j] = act[j] ? *reinterpret_cast<const double2*>(ring + o) : make_double2(0.0, 0.0);
            }
        }
        wait_lgkm0();   // the reads are done before their pieces are re-armed
        {
            int done_rows = r0 + RB;
            done_rows = done_rows < nrows ? done_rows : nrows;
            const int freed = (int)(((unsigned)done_rows * rowB) / 1024u);
            const int target = freed + NP;
            while (issued < target) issue_piece(issued++);
        }
        if constexpr (ABL == 1) {          // ablation: DMA + LDS read only
            double qq = 0.0;
#pragma unroll
            for (int rr = 0; rr < RB; ++rr)
#pragma unroll
                for (int j = 0; j < CPL2; ++j) qq += xs[rr][j].x + xs[rr][j].y;
            if (qq == 1.2345e300) a.bcol[(size_t)b * T + ta + blk] = qq;
            continue;
        }
        double acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = 0.0;
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const bool valid = (r0 + rr) < nrows;
#pragma unroll
            for (int j = 0; j < CPL2; ++j) {
                const double x0 = xs[rr][j].x, x1 = xs[rr][j].y;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    acc[rr * R + k] = fma(W[j][0][k], x0, acc[rr * R + k]);
                    acc[rr * R + k] = fma(W[j][1][k], x1, acc[rr * R + k]);
                }
                if (valid) {
                    q[j][0] = fma(x0, x0, q[j][0]);
                    q[j][1] = fma(x1, x1, q[j][1]);
                }
            }
        }
        if constexpr (ABL == 2) {          // ablation: no cross-lane reduce
            double qq = 0.0;
#pragma unroll
            for (int v = 0; v < NV; ++v) qq += acc[v];
            if (qq == 1.2345e300) a.bcol[(size_t)b * T + ta + blk] = qq;
            continue;
        }
        wave_transpose_reduce<NV>(acc, lane);
        const int t = ta + r0 + my_rr;
        if constexpr (ABL == 3) {          // ablation: everything but the store
            if (acc[0] == 1.2345e300) a.bcol[(size_t)b * T + ta + blk] = acc[0];
            continue;
        }
        if (canon && t < tb) a.bcol[((size_t)b * T + t) * R + my_k] = acc[0];
    }
    wait_vmcnt<0>();   // drain the trailing DMAs before the LDS is released
    // s = sum_t sum_i x_it^2 / R_i over this wave's periods
    double sp = 0.0;
#pragma unroll
    for (int j = 0; j < CPL2; ++j) sp = fma(q[j][0], Ri[j][0], fma(q[j][1], Ri[j][1], sp));
    sp = wave_allsum(sp);
    if (lane == 0) {
        a.ssum[(size_t)b * kSsumSlots + wave] = sp;
        if (sp != sp) atomicOr(a.status, 1);   // NaN in the panel on the balanced path
    }
}

// ---------------------------------------------------------------------------------------------
// C = Lam' R^-1 Lam (full symmetric r x r) and sum_i log R_i; one wave per replicate.
template <int R, int CPL2>
__global__ __launch_bounds__(64) void gram_kernel(CollapseArgs a) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int N = a.N;
    const double* __restrict__ L = a.Lam + (size_t)b * N * R;
    const double* __restrict__ Rv = a.Rv + (size_t)b * N;
    double W[CPL2][2][R];
    bool own[CPL2][2];
    double ld = 0.0;
#pragma unroll
    for (int j = 0; j < CPL2; ++j)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = 2 * lane + 128 * j + e;
            own[j][e] = c < N;
            const int cc = own[j][e] ? c : N - 1;
            const double rv = own[j][e] ? Rv[cc] : 1.0;
            const double ri = own[j][e] ? 1.0 / rv : 0.0;
            ld += log(rv);
#pragma unroll
            for (int k = 0; k < R; ++k) W[j][e][k] = L[(size_t)cc * R + k] * ri;
        }
    c_all<R, CPL2, 0, true>(W, L, own, lane, a.Cfull + (size_t)b * R * R);
    ld = wave_allsum(ld);
    if (lane == 0) a.ldfull[b] = ld;
}

// ---------------------------------------------------------------------------------------------
template <int R, int CPL2, int RB, int NP, int ABL = 0, int OPT = 0>
static hipError_t launch_dma_one(const CollapseArgs& a, hipStream_t s) {
    const size_t lds = (size_t)4 * NP * 1024;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&collapse_dma_kernel<R, CPL2, RB, NP, ABL, OPT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((collapse_dma_kernel<R, CPL2, RB, NP, ABL, OPT>), dim3(a.B), dim3(256), lds, s, a);
    return hipGetLastError();
}

constexpr int ring_pieces(int rb, int cpl2) {   // smallest power of two >= 2 * rb * cpl2 (and >= 4)
    int n = 4;
    while (n < 2 * rb * cpl2) n <<= 1;
    return n;
}

template <int R>
static hipError_t launch_dma_r(const CollapseArgs& a, hipStream_t s, int variant) {
    constexpr int RB = (R <= 4) ? 8 : (R <= 8) ? 4 : (R <= 16) ? 2 : 1;
    // defaults carry OPT = 7 (aligned segments, fill before weights, no filler DMAs): -6 % at the headline shape
    if (a.N <= 128) return launch_dma_one<R, 1, RB, ring_pieces(RB, 1), 0, 7>(a, s);
    if (a.N <= 256) {
        if constexpr (R == 8) {   // tuning variants of the headline shape (DFM_COLLAPSE_VARIANT)
            if (variant == 1) return launch_dma_one<8, 2, 4, 32>(a, s);
            if (variant == 2) return launch_dma_one<8, 2, 2, 8>(a, s);
            if (variant == 3) return launch_dma_one<8, 2, 2, 16>(a, s);
            if (variant == 10) return launch_dma_one<8, 2, 4, 16, 1>(a, s);
            if (variant == 11) return launch_dma_one<8, 2, 4, 16, 2>(a, s);
            if (variant == 12) return launch_dma_one<8, 2, 2, 8, 1>(a, s);
            if (variant == 13) return launch_dma_one<8, 2, 2, 8, 2>(a, s);
#ifdef DFM_MFMA_BENCH_ONLY
            if (variant >= 100 && variant < 116) {   // 100 + OPT: <8,2,4,16> with tuning bits
                switch (variant - 100) {
#define DFM_V(o) case o: return launch_dma_one<8, 2, 4, 16, 0, o>(a, s);
                    DFM_V(0) DFM_V(1) DFM_V(2) DFM_V(3) DFM_V(4) DFM_V(5) DFM_V(6) DFM_V(7)
                    DFM_V(8) DFM_V(9) DFM_V(10) DFM_V(11) DFM_V(12) DFM_V(13) DFM_V(14) DFM_V(15)
#undef DFM_V
                    default: break;
                }
            }
            if (variant == 20) return launch_dma_one<8, 2, 4, 16, 3, 7>(a, s);    // no store
            if (variant == 21) return launch_dma_one<8, 2, 4, 16, 2, 7>(a, s);    // no reduce, no store
            if (variant == 22) return launch_dma_one<8, 2, 4, 16, 1, 7>(a, s);    // DMA + LDS read only
            if (variant == 23) return launch_dma_one<8, 2, 2, 16, 0, 15>(a, s);
            if (variant == 24) return launch_dma_one<8, 2, 2, 8, 0, 15>(a, s);
            if (variant == 25) return launch_dma_one<8, 2, 4, 32, 0, 15>(a, s);
#endif
        }
        return launch_dma_one<R, 2, RB, ring_pieces(RB, 2), 0, 7>(a, s);
    }
    if constexpr (R <= 16) {
        if (a.N <= 512) return launch_dma_one<R, 4, RB, ring_pieces(RB, 4), 0, 7>(a, s);
    }
    if constexpr (R <= 8) {
        if (a.N <= 1024) return launch_dma_one<R, 8, RB / 2, ring_pieces(RB / 2, 8), 0, 7>(a, s);
    }
    return hipErrorInvalidValue;
}

bool collapse_dma_supported(int Rpad, int N) { return (N % 2 == 0) && N <= collapse_max_n(Rpad); }

hipError_t launch_collapse_dma(int Rpad, const CollapseArgs& a, hipStream_t s, int variant) {
    if (variant >= 200 && variant < 300 && collapse_mfma_supported(Rpad, a.N)) return launch_collapse_mfma(Rpad, a, s, variant - 200);
    note_kernel("collapse_dma_kernel");
    switch (Rpad) {
        case 2: return launch_dma_r<2>(a, s, variant);
        case 4: return launch_dma_r<4>(a, s, variant);
        case 8: return launch_dma_r<8>(a, s, variant);
        case 16: return launch_dma_r<16>(a, s, variant);
        case 32: return launch_dma_r<32>(a, s, variant);
        default: return hipErrorInvalidValue;
    }
}

template <int R>
static hipError_t launch_gram_r(const CollapseArgs& a, hipStream_t s) {
    if (a.N <= 128) hipLaunchKernelGGL((gram_kernel<R, 1>), dim3(a.B), dim3(64), 0, s, a);
    else if (a.N <= 256) hipLaunchKernelGGL((gram_kernel<R, 2>), dim3(a.B), dim3(64), 0, s, a);
    else if (R <= 16 && a.N <= 512) hipLaunchKernelGGL((gram_kernel<(R <= 16 ? R : 2), 4>), dim3(a.B), dim3(64), 0, s, a);
    else if (R <= 8 && a.N <= 1024) hipLaunchKernelGGL((gram_kernel<(R <= 8 ? R : 2), 8>), dim3(a.B), dim3(64), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

bool gram_supported(int Rpad, int N) {
    return N <= 256 || (Rpad <= 16 && N <= 512) || (Rpad <= 8 && N <= 1024);
}

hipError_t launch_gram(int Rpad, const CollapseArgs& a, hipStream_t s) {
    switch (Rpad) {
        case 2: return launch_gram_r<2>(a, s);
        case 4: return launch_gram_r<4>(a, s);
        case 8: return launch_gram_r<8>(a, s);
        case 16: return launch_gram_r<16>(a, s);
        case 32: return launch_gram_r<32>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace dfm
