// scan_mfma32.hip -- the mean recursion of the balanced fast path for Rp = 32 (BASELINE config 4) and Rp = 16 on the f64
// matrix pipe (the text below says 32; Rp = 16 is one row tile, 4 k-steps, 4 waves = 64 chunks).
//
//     xi_{t+1} = G_t xi_t + b_t,  w_t = Z_t xi_t      forward       f_t = w_t + J_t f_{t+1}      backward
//
// meanscan_kernel (fastpath.hip) runs the steady part as 16 time chunks per replicate, each a lane group of 32 lanes doing
// 32 x 32 matrix-vector products with DPP / ds_swizzle lane exchanges: ~2400 cycles per step, 4 x 125 dependent steps --
// 0.62 ms of the 1.75 ms of a config-4 pass.  The steady recurrence applies the SAME matrix to every chunk's state, so one
// step of 16 chunks is a 32 x 32 by 32 x 16 matrix product: two 16 x 16 output tiles x 8 `v_mfma_f64_16x16x4`.  And the
// result layout of that instruction IS its B-operand layout for the next step: lane (K = l / 16, j = l % 16) holds
// D[K + 4 v][j] in register v, and as a B operand it must supply B[4 s + K][j] for k-step s -- the same element for
// s = v (+ 4 per row tile).  The chunk states never leave the registers of their wave: no lane exchange, no LDS.
//
// One workgroup (8 waves) per replicate, 128 chunks of L = T / 128 steps (wave w: chunks 16 w .. 16 w + 15 = its 16
// columns): sequential depth 2 L + 7 carry levels instead of 2 x 125 + 4.
//   transient  t < E - 1: time-varying matrices, wave 0 alone with the lane-group code of dfm_scan.h (as meanscan_kernel)
//   phase 1    every chunk from a zero state (chunk 0 from the true xi): L steps of 16 MFMAs
//   carry      Kogge-Stone over the 128 chunk end states: level k multiplies by G^(L 2^k) (from cov_kernel, staged in LDS with
//              a conflict-free row stride) the states 2^k columns to the left, exchanged through LDS
//   phase 3    re-run from the true start states; w_t = Z xi_t (16 more MFMAs per step, off the dependent chain),
//              xi_t' w_t for the log-likelihood; w_t goes to the scratch table
//   backward   the same three phases with J and the operands w_t; f_t to f_smooth
// Reference counterpart: none (dfm_functions.ipynb:21-23 declares `Parametric` only); the oracle is oracle/kalman_oracle.c.
#include <stdlib.h>

#include <type_traits>

#include "dfm_cov.h"
#include "dfm_scan.h"

namespace dfm {

namespace {

constexpr int kS3PF = 4;                           // operand prefetch distance (steps) of the emitting runs
#ifndef DFM_S3_PF1
#define DFM_S3_PF1 8
#endif
constexpr int kS3PF1 = DFM_S3_PF1;                 // ... of the zero-state runs (no Z operand, no emission: registers to spare)
typedef double s3_v4 __attribute__((ext_vector_type(4)));
// workgroup barrier for data that travels through LDS only: __syncthreads() also waits for every outstanding global access
// (vmcnt(0)) -- here that would be the prefetched operands of later steps and the asynchronous staging
__device__ __forceinline__ void s3_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
using s3_lds_ptr = __attribute__((address_space(3))) char*;
// 64 lanes x 16 bytes, global -> LDS (1 KB back to back from byte address `lds_dst`), asynchronous: counted by vmcnt
__device__ __forceinline__ void s3_dma16(const void* gsrc, unsigned lds_dst) {
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

template <int R>
struct S3Geo {
    static constexpr int NIO = R / 16;                 // 16-row tiles of the state
    static constexpr int NK = R / 4;                   // k-steps of a product with an R x R matrix
    static constexpr int NC = scan_groups(R);          // chunks: 128 (R = 32), 64 (R = 16)
    static constexpr int NT = 4 * NC;                  // threads: 16 chunks per wave
    static constexpr int LEV = scan_levels(R);         // 7
    // row stride (doubles) of the staged power matrices and of the chunk-state table: the 8-byte slots of (row / column j, k),
    // j < 16, k < 2, are all different mod 32 -- conflict-free b64 reads (R = 32: 2 j + k; R = 16: 18 j + k)
    static constexpr int PS = R + 2;
    static constexpr int oPow = 0;                                   // [LEV][R][PS]
    static constexpr int oP = oPow + LEV * R * PS;                   // [128][PS] chunk states
    static constexpr int oVec = oP + NC * PS;                        // xi_ts | xi_T | f_T | f_b  (R each)
    static constexpr int oPs = oVec + 4 * R;                         // packed P_s,inf
    static constexpr int oRed = oPs + R * (R + 1) / 2;
    static constexpr int total = oRed + 8;
};

// State order.  The MFMA fixes which PHYSICAL row m = K + 4 v + 16 io of the state a lane (K, j) holds in register (io, v); it
// does not care which component of the state that row is.  Row m carries component pi(m) = 16 io + 8 (v / 2) + 2 K + v % 2: a
// register PAIR of a lane is 16 contiguous bytes of b_t / w_t / f_t, and the four K lanes of a column read 64 contiguous bytes with
// ONE 16-byte load each -- an instruction touches 16 rows x 64 bytes.  (Round 2 gave a lane 8 consecutive components: every 16-byte
// load then touched 16 rows x 4 pieces 64 bytes apart, each 64-byte sector was asked for by four different instructions, and the
// forward runs over the cold b_t sat at 3.3 us per step against 0.85 us of MFMA time.)  Only the matrices have to follow:
// A[m][n] = M[pi(m)][pi(n)].
template <int R>
__device__ __forceinline__ constexpr int s3_pi(int K, int v, int io) { return 16 * io + 8 * (v >> 1) + 2 * K + (v & 1); }
// A operand of the 16x16x4 MFMA for matrix M (row-major, row stride `ld` doubles): lane (k4 = l / 16, c16 = l % 16) holds
// physical element (16 io + c16, 4 s + k4) in A[io][s]
template <int R>
__device__ __forceinline__ void load_aop(double (&A)[R / 16][R / 4], const double* M, int ld, int k4, int c16) {
#pragma unroll
    for (int io = 0; io < R / 16; ++io)
#pragma unroll
        for (int s = 0; s < R / 4; ++s)
            A[io][s] = M[(size_t)s3_pi<R>(c16 & 3, c16 >> 2, io) * ld + s3_pi<R>(k4, s & 3, s >> 2)];
}

// Y = M X (+ Y0): X, Y in the D layout of the MFMA (X[it][v] = element (K + 4 v + 16 it, column j) of lane (K, j))
template <int R>
__device__ __forceinline__ void mm_step(const double (&A)[R / 16][R / 4], const s3_v4 (&X)[R / 16], s3_v4 (&Y)[R / 16]) {
#pragma unroll
    for (int s = 0; s < R / 4; ++s) {
#pragma unroll
        for (int io = 0; io < R / 16; ++io) Y[io] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[io][s], X[s >> 2][s & 3], Y[io], 0, 0, 0);
    }
}

}  // namespace

// (R = 16: at most 128 VGPRs, so that four of the 4-wave workgroups share a CU -- 1024 replicates in one round instead of two)
template <int R>
__global__ __launch_bounds__(S3Geo<R>::NT, (R == 16 ? 4 : 1)) void meanscan_mfma_kernel(FastArgs a) {
    using S3Lds = S3Geo<R>;
    constexpr int kS3Threads = S3Lds::NT;
    constexpr int NIO = S3Lds::NIO, kS3Lev = S3Lds::LEV, kS3PS = S3Lds::PS, kS3NC = S3Lds::NC;
    (void)kS3NC;
    extern __shared__ __attribute__((aligned(16))) double dsm[];
    double* s_pow = dsm + S3Lds::oPow;
    double* s_P = dsm + S3Lds::oP;
    double* s_vec = dsm + S3Lds::oVec;
    double* s_ps = dsm + S3Lds::oPs;
    double* s_red = dsm + S3Lds::oRed;
    const int b = blockIdx.x + a.b0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = lane >> 4, j16 = lane & 15;                 // MFMA lane coordinates
    const int ch = 16 * wave + j16;                           // this lane's chunk (its column)
    const int i32 = lane % R, cg = lane / R;                  // lane-group coordinates of the transient code (wave 0; group 0 writes)
    const int T = a.T, r = a.r, L = a.L;
    const int E = a.E[b];
    const int ts = E - 1;
    const int bst = a.bst > 0 ? a.bst : R;                    // doubles between the rows of b_t (the wide collapse drops padding columns)
    const double* bcol = a.bcol + (size_t)b * T * bst;
    double* wtab = a.wtab + (size_t)b * (a.wrep ? a.wrep : (size_t)T * R);   // natural rows (the transient steps use them)
    // steady w_t, CHUNK-MAJOR behind the T natural rows: 16-byte piece pc (= 2 io + h) of step j of wave w's 16 chunks is the
    // 1 KB [(w L + j) NPc + pc][lane] -- the forward re-run writes whole KBs, every wave its own contiguous run (interleaved
    // 64-byte pieces wrote at the 4 TB/s of grid-stride stores: scripts/microbench/storebw.hip), and the backward runs read the
    // same KBs back (their chunks are the forward ones shifted: at most two KBs per instruction, every byte used)
    constexpr int NPc = R / 8;
    const int lsh = __builtin_ctz((unsigned)L);
    const int npc = ((a.rstate > 0 ? a.rstate : R) + 7) >> 3;   // 16-byte pieces per lane and period that carry state components (the rest: padding)
    double2* wch = reinterpret_cast<double2*>(wtab + (size_t)T * R);

    const double* tab = a.tab + (size_t)b * T * 3 * R * R;
    const double* stead = a.stead + (size_t)b * stead_mats(R) * R * R;
    double* fout = a.f_smooth + (size_t)b * T * r;
    const int npr = r * (r + 1) / 2;
    const bool fpair = (r & 1) == 0 && (reinterpret_cast<size_t>(fout) & 15) == 0;   // f_t rows can be stored as 16-byte pairs
    // DFM_SCAN_ABL & 256: phase stamps of workgroup 0 -- diagnostics build only (the dynamically indexed array is 96 bytes of scratch and
    // the printf a hostcall buffer in every launch of the production kernel otherwise)
#ifdef DFM_DIAG
    unsigned long long stamps[10];
    int nstamp = 0;
    auto stamp = [&]() { if ((a.abl & 256) && blockIdx.x == 0 && tid == 0 && nstamp < 10) stamps[nstamp++] = __builtin_amdgcn_s_memrealtime(); };
#else
    auto stamp = [] {};
#endif
    stamp();

    constexpr int NW = kS3Threads / 64;
    const unsigned lds_dsm = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(s3_lds_ptr)dsm);
    // carry powers of one direction -> LDS (row stride kS3PS), ASYNCHRONOUSLY: 1-KB LDS-DMA pieces dealt to the waves; the 16 bytes
    // of lane l of piece i are LDS doubles 128 i + 2 l, 128 i + 2 l + 1 = columns (c, c + 1) of row (level k, r) -- c = R is the
    // padding pair, it re-reads the row's last pair.  The carry scan waits for them (vmcnt(0) + barrier) a whole phase later; the
    // backward powers are requested when the forward carry scan is done with the table, under the forward re-run.
    auto stage_powers = [&](int first) {
        constexpr int bytes = kS3Lev * R * kS3PS * 8;
        constexpr int npiece = (bytes + 1023) / 1024;
        const double* src0 = stead + (size_t)first * R * R;
        for (int i = wave; i < npiece; i += NW) {
            const int byte = 1024 * i + 16 * lane;
            const int e = byte >> 3;
            const int kr = e / kS3PS, col = e % kS3PS;
            const int krc = kr < kS3Lev * R ? kr : kS3Lev * R - 1;
            const double* src = src0 + (size_t)krc * R + (col < R ? col : R - 2);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dsm + (unsigned)(S3Lds::oPow * 8 + 1024 * i));
            if (byte < bytes) s3_dma16(src, dst);
        }
    };
    auto powers_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };   // (then a barrier: every wave's pieces)
    // ---- forward transient, steps 0 .. ts - 1 (time-varying Z_t, G_t: 8 KB each per step): the chain xi_(t+1) = G_t xi_t + b_t on
    // the even waves, w_t = Z_t xi_t (off the chain) on the odd ones, step t on waves 2 t and 2 t + 1 (mod NW) -- every wave
    // fetches the one matrix of its first step NOW, all at once, and the one of its next step as soon as it has used the last
    // (round 2: wave 0 alone, two dependent 8-KB fetches + two products per step, 6.4 us per step).  xi_t travels through LDS,
    // ping-pong between s_vec[0 ..) and s_vec[3 R ..) so that xi_ts ends in s_vec[0 ..).
    double Mp[R];                                             // (forward transient only; the backward one has its own: Mb)
    double btv = 0.0;
    const bool isG = (wave & 1) == 0;
    int tn = wave >> 1;                                       // the next step of this wave
    auto xi_slot = [&](int t) { return ((ts - t) & 1) ? 3 * R : 0; };
    auto preload_fwd = [&](int t) {
        if (t < ts) {
            load_xperm<R>(Mp, tab + (size_t)t * 3 * R * R + (isG ? 2 * R * R : 0), i32);
            if (isG) btv = i32 < bst ? bcol[(size_t)t * bst + i32] : 0.0;
        }
    };
    const double xi00 = a.xi0[(size_t)b * R + i32];
    preload_fwd(tn);
    stage_powers(3);                                          // (behind the transient's fetches: loads return in order)
    if (wave == 0 && cg == 0) s_vec[xi_slot(0) + i32] = xi00;
    if (a.P_smooth && !(a.abl & 1)) {
        for (int v = tid; v < npr; v += kS3Threads) {         // packed (caller's r) copy of P_s,inf
            int ri = 0;
            while ((ri + 1) * (ri + 2) / 2 <= v) ++ri;
            s_ps[v] = a.PsInf[(size_t)b * R * R + ri * R + (v - ri * (ri + 1) / 2)];
        }
        __syncthreads();
        // ---- P_smooth rows inside the fixed-point range: fire-and-forget stores (unless pfill_kernel wrote them)
        fill_psmooth_rows(a, b, tid, kS3Threads, s_ps);
    }
    if (a.abl & 2) { powers_landed(); return; }
    s3_barrier_lds();

    stamp();   // 1: staging done
    double dot = 0.0;                                         // lane part of sum_t xi_t' w_t
    for (int t = 0; t < ts; ++t) {
        if (tn == t) {                                        // (wave-uniform)
            const double xi = s_vec[xi_slot(t) + i32];
            if (isG) {
                const double xn = matvec_x<R>(Mp, xi, btv);
                if (cg == 0) s_vec[xi_slot(t + 1) + i32] = xn;
            } else {
                const double w = matvec_x<R>(Mp, xi);
                if (cg == 0) {
                    dot = fma(xi, w, dot);
                    wtab[(size_t)t * R + i32] = w;
                }
            }
            tn += NW / 2;
            preload_fwd(tn);
        }
        s3_barrier_lds();
    }

    stamp();   // 2: forward transient done
    int kn = wave;                                            // backward transient: the next step of this wave
    double Mb[R], wtv;                                        // (not initialised: nothing to keep alive across the forward scan)
    auto preload_bwd = [&](int k) {
        if (k < ts) {
            const int t = ts - 1 - k;
            load_xperm<R>(Mb, tab + (size_t)t * 3 * R * R + R * R, i32);
            wtv = wtab[(size_t)t * R + i32];
        }
    };
    const int clast = (T - 1 - ts) / L;                       // chunk that holds step T - 1 (forward) / step ts (backward)
    // One chunked scan.  dir = +1: steps t = ts + ch L + j, operands b_t, emits w_t and xi' w;  dir = -1:
    // steps t = T - 1 - ch L - j, operands w_t, emits f.  head = s_vec offset of the start vector of chunk 0,
    // tail = offset that receives the state after the last valid step.
    auto scan = [&](const double (&AM)[NIO][R / 4], const double (&AZ)[NIO][R / 4], auto fwd_tag, int head, int tail) {
        constexpr bool FWD = decltype(fwd_tag)::value;
        const int t0 = FWD ? ts + ch * L : T - 1 - ch * L;
        auto step_t = [&](int j) { return FWD ? t0 + j : t0 - j; };
        auto valid_t = [&](int j) { const int t = step_t(j); return j < L && t >= ts && t < T; };
        // operands of step j in the D layout
        auto load_u = [&](s3_v4 (&U)[NIO], int j) {
            int t = step_t(j);
            t = t < ts ? ts : (t >= T ? T - 1 : t);           // (a row that exists; the step is skipped when invalid)
            const double2* p;
            int st;                                           // stride between the lane's pieces, in 16-byte units
            if constexpr (FWD) {
                p = reinterpret_cast<const double2*>(bcol + (size_t)t * bst) + K;   // components 16 io + 8 h + 2 K (+ 1): piece pc at 4 pc + K
                st = 4;
            } else {                                          // w_t where the forward re-run put it: chunk (t - ts) / L, step (t - ts) % L
                const int d = t - ts, cf = d >> lsh, jf = d & (L - 1);   // (L is a power of two: fast_chunk_len)
                p = wch + ((size_t)((cf >> 4) * L + jf) * NPc) * 64 + 16 * K + (cf & 15);
                st = 64;
            }
#pragma unroll
            for (int io = 0; io < NIO; ++io) {                // (pieces of padding components only -- exact zeros -- are neither stored nor read)
                const double2 z2 = make_double2(0.0, 0.0);
                const double2 x = (2 * io < npc) ? p[(2 * io) * st] : z2, y = (2 * io + 1 < npc) ? p[(2 * io + 1) * st] : z2;
                U[io][0] = x.x; U[io][1] = x.y; U[io][2] = y.x; U[io][3] = y.y;
            }
        };
        auto run = [&](s3_v4 (&X)[NIO], auto emit_tag) {
            constexpr bool EMIT = decltype(emit_tag)::value;
            constexpr int PF = (EMIT || R < 32) ? kS3PF : kS3PF1;
            s3_v4 cur[PF][NIO];                                 // ring of operands, PF steps ahead (slot u refilled once consumed)
#pragma unroll
            for (int u = 0; u < PF; ++u) load_u(cur[u], u);
            for (int j0 = 0; j0 < L; j0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int j = j0 + u;
                    if (j < L) {                              // wave-uniform
                        const bool ok = valid_t(j);
                        const int t = step_t(j);
                        if constexpr (EMIT && FWD) {          // w_t = Z xi_t, xi_t' w_t
                            s3_v4 W[NIO];
#pragma unroll
                            for (int io = 0; io < NIO; ++io) W[io] = s3_v4{0.0, 0.0, 0.0, 0.0};
                            mm_step<R>(AZ, X, W);
                            if (ok) {
                                double2* q = wch + ((size_t)(wave * L + j) * NPc) * 64 + lane;
#pragma unroll
                                for (int io = 0; io < NIO; ++io) {
#pragma unroll
                                    for (int v = 0; v < 4; ++v) dot = fma(X[io][v], W[io][v], dot);
                                    if (2 * io < npc) q[(2 * io) * 64] = make_double2(W[io][0], W[io][1]);
                                    if (2 * io + 1 < npc) q[(2 * io + 1) * 64] = make_double2(W[io][2], W[io][3]);
                                }
                            }
                        }
                        s3_v4 Y[NIO];
#pragma unroll
                        for (int io = 0; io < NIO; ++io) Y[io] = cur[u][io];
                        load_u(cur[u], j + PF);
                        mm_step<R>(AM, X, Y);
#pragma unroll
                        for (int io = 0; io < NIO; ++io)
#pragma unroll
                            for (int v = 0; v < 4; ++v) X[io][v] = ok ? Y[io][v] : X[io][v];
                        if constexpr (EMIT && !FWD) {         // f of period t - 1
                            if (ok && t >= 1) {
                                double* fo = fout + (size_t)(t - 1) * r;
#pragma unroll
                                for (int io = 0; io < NIO; ++io)
#pragma unroll
                                    for (int hh = 0; hh < 2; ++hh) {
                                        const int c = s3_pi<R>(K, 2 * hh, io);      // even; the pair (c, c + 1)
                                        if (fpair) {                              // r even, rows 16-byte aligned: c < r implies c + 1 < r
                                            if (c < r) *reinterpret_cast<double2*>(fo + c) = make_double2(X[io][2 * hh], X[io][2 * hh + 1]);
                                        } else {
                                            if (c < r) fo[c] = X[io][2 * hh];
                                            if (c + 1 < r) fo[c + 1] = X[io][2 * hh + 1];
                                        }
                                    }
                            }
                        }
                    }
                }
            }
        };
        auto from_vec = [&](s3_v4 (&X)[NIO], const double* vsrc, bool take) {
#pragma unroll
            for (int io = 0; io < NIO; ++io)
#pragma unroll
                for (int v = 0; v < 4; ++v) X[io][v] = take ? vsrc[s3_pi<R>(K, v, io)] : 0.0;
        };
        // phase 1: chunk 0 from the true start (its end state then carries the head through the scan), the others from zero
        s3_v4 X[NIO];
        from_vec(X, s_vec + head, ch == 0);
        run(X, std::false_type{});
        stamp();   // 3 / 7: phase 1 done
        // carry: inclusive Kogge-Stone scan of the end states, P_c += M^(L 2^k) P_(c - 2^k)
        auto put_state = [&]() {
#pragma unroll
            for (int io = 0; io < NIO; ++io)
#pragma unroll
                for (int v = 0; v < 4; ++v) s_P[ch * kS3PS + K + 4 * v + 16 * io] = X[io][v];
        };
        put_state();
        powers_landed();
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < kS3Lev; ++k) {
            double AP[NIO][R / 4];
            load_aop<R>(AP, s_pow + (size_t)k * R * kS3PS, kS3PS, K, j16);
            const int src = ch - (1 << k);
            s3_v4 Bv[NIO];                                     // the state 2^k columns to the left, as the B operand (= D layout)
#pragma unroll
            for (int io = 0; io < NIO; ++io)
#pragma unroll
                for (int v = 0; v < 4; ++v) Bv[io][v] = src >= 0 ? s_P[(src < 0 ? 0 : src) * kS3PS + K + 4 * v + 16 * io] : 0.0;
            mm_step<R>(AP, Bv, X);
            __syncthreads();                                  // every wave has read the old states
            put_state();
            __syncthreads();
        }
        if constexpr (FWD) stage_powers(3 + kS3Lev);           // J^(L 2^k): every wave is past its last read of the table
        else preload_bwd(kn);                                  // first J_t of this wave for the backward transient (w_t, t < ts: written long ago)
        // start state of chunk c: P_(c - 1); chunk 0: the head
        if (ch == 0) from_vec(X, s_vec + head, true);
        else {
#pragma unroll
            for (int io = 0; io < NIO; ++io)
#pragma unroll
                for (int v = 0; v < 4; ++v) X[io][v] = s_P[(ch - 1) * kS3PS + K + 4 * v + 16 * io];
        }
        stamp();   // 4 / 8: carry done
        // phase 3
        run(X, std::true_type{});
        stamp();   // 5 / 9: phase 3 done
        if (ch == clast) {
#pragma unroll
            for (int io = 0; io < NIO; ++io)
#pragma unroll
                for (int v = 0; v < 4; ++v) s_vec[tail + s3_pi<R>(K, v, io)] = X[io][v];
        }
    };

    {
        double AG[NIO][R / 4], AZ[NIO][R / 4];
        load_aop<R>(AG, stead + 2 * R * R, R, K, j16);           // steady G
        load_aop<R>(AZ, stead, R, K, j16);                       // steady Z
        scan(AG, AZ, std::true_type{}, 0, R);                 // xi_ts -> ... -> xi_T
    }
    __syncthreads();   // xi_T in LDS; every w_t of this replicate is written (workgroup-visible)

    // ---- terminal -------------------------------------------------------------------------------------------------------
    if (wave == 0) {
        const double xiT = s_vec[R + i32];
        double PTp[R];
        load_xperm<R>(PTp, a.PT + (size_t)b * R * R, i32);
        const double fT = matvec_x<R>(PTp, xiT);
        if (cg == 0) {
            dot = fma(xiT, fT, dot);                          // the log-likelihood needs sum xi'w + xi_T' f_T
            if (i32 < r) fout[(size_t)(T - 1) * r + i32] = fT;
            s_vec[2 * R + i32] = fT;
        }
    }
    __syncthreads();

    stamp();   // 6: terminal done
    // ---- steady backward scan: steps T - 1 .. ts ----------------------------------------------------------------------------
    {
        double AJ[NIO][R / 4];
        load_aop<R>(AJ, stead + R * R, R, K, j16);               // steady J
        scan(AJ, AJ, std::false_type{}, 2 * R, 3 * R);
    }
    __syncthreads();

    // ---- backward transient: v <- J_t v + w_t, t = ts - 1 .. 0; step k = ts - 1 - t on wave k mod NW, whose J_t and w_t were
    // requested under the backward re-run (preload_bwd); v travels through LDS (s_vec[3 R ..) <-> s_vec[0 ..))
    {
        auto v_slot = [&](int k) { return (k & 1) ? 0 : 3 * R; };
        for (int k = 0; k < ts; ++k) {
            if (kn == k) {                                    // (wave-uniform)
                const int t = ts - 1 - k;
                const double v = matvec_x<R>(Mb, s_vec[v_slot(k) + i32], wtv);
                if (cg == 0) {
                    s_vec[v_slot(k + 1) + i32] = v;
                    if (t >= 1 && i32 < r) fout[(size_t)(t - 1) * r + i32] = v;
                    if (t == 0 && a.f0s) a.f0s[(size_t)b * R + i32] = v;    // E[f_0 | X] (EM)
                }
                kn += NW;
                preload_bwd(kn);
            }
            s3_barrier_lds();
        }
        if (ts == 0 && wave == 0 && cg == 0 && a.f0s) a.f0s[(size_t)b * R + i32] = s_vec[3 * R + i32];
    }

#ifdef DFM_DIAG
    if ((a.abl & 256) && blockIdx.x == 0 && tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        printf("S3STAMP ts=%d L=%d :", ts, L);
        for (int k = 1; k < nstamp; ++k) printf(" %llu", stamps[k] - stamps[0]);
        printf(" end %llu\n", t1 - stamps[0]);
    }
#endif
    // ---- log-likelihood ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) dot += __shfl_xor(dot, off, kWave);
    if (lane == 0) s_red[wave] = dot;
    __syncthreads();
    if (tid == 0) {
        double d = 0.0, sq = 0.0;
#pragma unroll
        for (int w = 0; w < kS3Threads / 64; ++w) d += s_red[w];
        if (a.ntile > 0) {
            for (int w = 0; w < a.ntile; ++w) sq += a.scol[(size_t)b * T + w];
        } else {
            for (int w = 0; w < a.nseg; ++w) sq += a.ssum[(size_t)b * kSsumSlots + w];
        }
        a.loglik[b] = -0.5 * (a.llc[b] + sq - d);
    }
}

namespace {
template <int R>
hipError_t launch_scan_mfma_r(const FastArgs& a, hipStream_t s) {
    if (a.wrep < (size_t)(a.T + S3Geo<R>::NC * a.L) * R) return hipErrorInvalidValue;   // (the chunk-major region of w_t: capi.hip wtab_rows)
    const size_t lds = (size_t)S3Geo<R>::total * sizeof(double);
    static LdsOptIn attr_done;
    if (!attr_done && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&meanscan_mfma_kernel<R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((meanscan_mfma_kernel<R>), dim3(a.B), dim3(S3Geo<R>::NT), lds, s, a);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_meanscan_mfma(int Rpad, const FastArgs& a, hipStream_t s) {
    note_kernel("meanscan_mfma_kernel");
    return Rpad == 32 ? launch_scan_mfma_r<32>(a, s) : Rpad == 16 ? launch_scan_mfma_r<16>(a, s) : hipErrorInvalidValue;
}

}  // namespace dfm
