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

constexpr int kS3PF = 4;                           // steps per operand prefetch block
typedef double s3_v4 __attribute__((ext_vector_type(4)));

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
// does not care which component of the state that row is.  Row m carries component pi(m) = (R / 4) K + 4 io + v: a lane's R / 4
// registers are CONSECUTIVE components -- 64 (R = 32) or 32 (R = 16) contiguous bytes of b_t / w_t / f_t per lane and period
// instead of 8-byte accesses 32 bytes apart.  Only the matrices have to follow: A[m][n] = M[pi(m)][pi(n)].
template <int R>
__device__ __forceinline__ constexpr int s3_pi(int K, int v, int io) { return (R / 4) * K + 4 * io + v; }
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
    const double* bcol = a.bcol + (size_t)b * T * R;
    double* wtab = a.wtab + (size_t)b * T * R;
    const double* tab = a.tab + (size_t)b * T * 3 * R * R;
    const double* stead = a.stead + (size_t)b * stead_mats(R) * R * R;
    double* fout = a.f_smooth + (size_t)b * T * r;
    const int npr = r * (r + 1) / 2;
    unsigned long long stamps[10];                            // DFM_SCAN_ABL & 256: phase stamps of workgroup 0 (diagnostics)
    int nstamp = 0;
    auto stamp = [&]() { if ((a.abl & 256) && blockIdx.x == 0 && tid == 0 && nstamp < 10) stamps[nstamp++] = __builtin_amdgcn_s_memrealtime(); };
    stamp();

    // carry powers of one direction -> LDS (row stride kS3PS)
    auto stage_powers = [&](int first) {
        for (int e = tid; e < kS3Lev * R * R; e += kS3Threads) {
            const int k = e / (R * R), rc = e % (R * R);
            s_pow[(size_t)k * R * kS3PS + (rc / R) * kS3PS + (rc % R)] = stead[(size_t)(first + k) * R * R + rc];
        }
    };
    stage_powers(3);
    if (a.P_smooth) {
        for (int v = tid; v < npr; v += kS3Threads) {         // packed (caller's r) copy of P_s,inf
            int ri = 0;
            while ((ri + 1) * (ri + 2) / 2 <= v) ++ri;
            s_ps[v] = a.PsInf[(size_t)b * R * R + ri * R + (v - ri * (ri + 1) / 2)];
        }
    }
    __syncthreads();
    // ---- P_smooth rows inside the fixed-point range: fire-and-forget stores (unless pfill_kernel wrote them)
    if (a.P_smooth && !(a.abl & 1)) fill_psmooth_rows(a, b, tid, kS3Threads, s_ps);
    if (a.abl & 2) return;

    stamp();   // 1: staging done
    // ---- forward transient: steps 0 .. ts - 1 on wave 0 (its two lane groups redundantly) ------------------------------
    double dot = 0.0;                                         // lane part of sum_t xi_t' w_t
    if (wave == 0) {
        double xi = a.xi0[(size_t)b * R + i32];
        for (int t = 0; t < ts; ++t) {
            double Zp[R], Gp[R];
            const double* ent = tab + (size_t)t * 3 * R * R;
            load_xperm<R>(Zp, ent, i32);
            load_xperm<R>(Gp, ent + 2 * R * R, i32);
            const double bt = bcol[(size_t)t * R + i32];
            const double w = matvec_x<R>(Zp, xi);
            if (cg == 0) {
                dot = fma(xi, w, dot);
                wtab[(size_t)t * R + i32] = w;
            }
            xi = matvec_x<R>(Gp, xi, bt);
        }
        if (cg == 0) s_vec[i32] = xi;                         // xi_ts
    }
    __syncthreads();

    stamp();   // 2: forward transient done
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
            const double2* p = reinterpret_cast<const double2*>((FWD ? bcol : wtab) + (size_t)t * R + (R / 4) * K);
#pragma unroll
            for (int io = 0; io < NIO; ++io) {
                const double2 x = p[2 * io], y = p[2 * io + 1];
                U[io][0] = x.x; U[io][1] = x.y; U[io][2] = y.x; U[io][3] = y.y;
            }
        };
        auto run = [&](s3_v4 (&X)[NIO], auto emit_tag) {
            constexpr bool EMIT = decltype(emit_tag)::value;
            s3_v4 cur[kS3PF][NIO];                              // ring of operands, kS3PF steps ahead (slot u refilled once consumed)
#pragma unroll
            for (int u = 0; u < kS3PF; ++u) load_u(cur[u], u);
            for (int j0 = 0; j0 < L; j0 += kS3PF) {
#pragma unroll
                for (int u = 0; u < kS3PF; ++u) {
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
                                double2* q = reinterpret_cast<double2*>(wtab + (size_t)t * R + (R / 4) * K);
#pragma unroll
                                for (int io = 0; io < NIO; ++io) {
#pragma unroll
                                    for (int v = 0; v < 4; ++v) dot = fma(X[io][v], W[io][v], dot);
                                    q[2 * io] = make_double2(W[io][0], W[io][1]);
                                    q[2 * io + 1] = make_double2(W[io][2], W[io][3]);
                                }
                            }
                        }
                        s3_v4 Y[NIO];
#pragma unroll
                        for (int io = 0; io < NIO; ++io) Y[io] = cur[u][io];
                        load_u(cur[u], j + kS3PF);
                        mm_step<R>(AM, X, Y);
#pragma unroll
                        for (int io = 0; io < NIO; ++io)
#pragma unroll
                            for (int v = 0; v < 4; ++v) X[io][v] = ok ? Y[io][v] : X[io][v];
                        if constexpr (EMIT && !FWD) {         // f of period t - 1
                            if (ok && t >= 1) {
#pragma unroll
                                for (int io = 0; io < NIO; ++io)
#pragma unroll
                                    for (int v = 0; v < 4; ++v) {
                                        const int row = s3_pi<R>(K, v, io);
                                        if (row < r) fout[(size_t)(t - 1) * r + row] = X[io][v];
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
    __syncthreads();   // xi_T in LDS; every w_t of this replicate is written (workgroup-visible); the G powers are done with
    stage_powers(3 + kS3Lev);                                 // J^(L 2^k)

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

    // ---- backward transient: steps ts - 1 .. 0 (wave 0) -----------------------------------------------------------------
    if (wave == 0) {
        double v = s_vec[3 * R + i32];                        // smoothed mean at the steady / transient boundary
        for (int t = ts - 1; t >= 0; --t) {
            double Jp[R];
            load_xperm<R>(Jp, tab + (size_t)t * 3 * R * R + R * R, i32);
            const double wt = wtab[(size_t)t * R + i32];
            v = matvec_x<R>(Jp, v, wt);
            if (cg == 0 && t >= 1 && i32 < r) fout[(size_t)(t - 1) * r + i32] = v;
        }
        if (a.f0s && cg == 0) a.f0s[(size_t)b * R + i32] = v;  // E[f_0 | X] (EM)
    }

    if ((a.abl & 256) && blockIdx.x == 0 && tid == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        printf("S3STAMP ts=%d L=%d :", ts, L);
        for (int k = 1; k < nstamp; ++k) printf(" %llu", stamps[k] - stamps[0]);
        printf(" end %llu\n", t1 - stamps[0]);
    }
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
