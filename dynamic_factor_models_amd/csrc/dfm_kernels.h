// dfm_kernels.h -- host-visible launch interface of the gfx950 kernels (internal to libdfmhip.so).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "dfm_device.h"

namespace dfm {

// The kernel a launcher actually dispatched to, for the per-kernel timing of dfm_profile_* (capi.hip): several launch_*
// functions choose among kernels (cov_kernel | cov_grid_kernel, recursion_kernel | recursion_wave_kernel |
// recursion_pair_kernel, ...), and bench.py's JSON must carry the names rocprofv3 prints.  thread_local: the multi-GPU
// object launches from one host thread per GPU.
extern thread_local const char* t_launched_kernel;
// Environment switches of the library, in two classes.
//   route_env: selects among PRODUCTION kernels that compute the same result to rounding (the one-launch pass or the two-launch
//              pass, one wave or a lane group per replicate, ...): what the test-suite uses to cover every fallback, and tuning
//              knobs (DFM_NUM_CU).  Ten names since round 6: DFM_FORCE_GENERAL, DFM_PASS_FUSED, DFM_NO_CHUNK, DFM_NO_PAIR,
//              DFM_MSTEP_MISS, DFM_TILE_NC, DFM_NUM_CU -- which kernel runs, never what it computes -- and DFM_CHUNK_W, DFM_TILE_W,
//              DFM_CHUNK_TOL, which set the warm-up and the boundary tolerance of the time-chunked recursions (a replicate that misses the
//              tolerance is redone sequentially: accuracy knobs).
//   diag_env:  ablations (some produce WRONG results on purpose: stream-only, compute-only, skipped stages), phase stamps, the
//              *_OLD kernels kept for A/B.  Read only by a library built with -DDFM_DIAG (`python -m
//              dynamic_factor_models_amd.build --diag`); the default libdfmhip.so ignores them.
inline const char* route_env(const char* name) { return getenv(name); }
#ifdef DFM_DIAG
inline const char* diag_env(const char* name) { return getenv(name); }
#else
inline const char* diag_env(const char*) { return nullptr; }
#endif

// Non-temporal hint on a kernel's ONE-PASS panel stream (pass_fused.hip dma16f, mstep_mfma.hip): on for batches whose panels are
// <= 2 GiB, where part of the pass's working set survives in the 256-MB Infinity Cache between launches (measured gains up to
// B = 4096 at the C2 shape, a loss at 8192).  DFM_DMA_NT=0 | 1 (diagnostics build) overrides.
inline bool stream_nt_hint(size_t panel_bytes) {
    static const int force = [] { const char* v = diag_env("DFM_DMA_NT"); return v ? (atoi(v) != 0 ? 1 : 0) : -1; }();
    if (force >= 0) return force != 0;
    return panel_bytes <= ((size_t)2 << 30);
}

inline void note_kernel(const char* name) { t_launched_kernel = name; }


// All device arrays below use the PADDED factor dimension Rp (2,4,8,16,32 >= r): parameters are
// embedded by pad_params_kernel (extra states: A = 0, Q = I, P0 = I, mu0 = 0, Lam = 0 -- independent
// unit-variance noise states that no series loads on; every determinant / quadratic form they add
// is exactly 0), so kernels are instantiated for a handful of sizes only.
constexpr int kSsumSlots = 16;

struct CollapseArgs {
    int B, T, N;
    int b0;               // collapse_dma only: first replicate of this launch (sub-batch pipelining)
    const double* panel;  // [B][T][N]
    const double* Lam;    // [B][N][Rp]
    const double* Rv;     // [B][N]
    double* bcol;         // [B][T][Rp]
    double* scol;         // [B][T]
    int* nobs;            // [B][T]
    double* ldrow;        // [B][T]      (written only where nobs < N)
    double* Ct;           // [B][T][Rp(Rp+1)/2] packed, or nullptr (written only where nobs < N)
    double* Cfull;        // [B][Rp][Rp]
    double* ldfull;       // [B]
    int* status;          // bit0: NaN met while Ct == nullptr
    double* ssum;         // [B][kSsumSlots]  balanced path only: sum_t s_t of each wave's periods (slots 0 .. nseg-1)
    const void* fuse_cov; // collapse_mfma only, HOST pointer read by the launcher (not by the kernel): FastArgs of the pass whose
                          // covariance workgroups + P_smooth fill ride at the front of this launch, or null
    int wpr;              // collapse_mfma only: waves (period segments) per replicate (0 = 4); nseg = wpr <= kSsumSlots
    int bst;              // collapse_wide2 (Rp = 32, balanced) only: > 0 = doubles between the rows of bcol (a multiple of 8, >= the state
                          // width: the padding components of b_t are exact zeros nobody needs); 0 = Rp
    int ct_r;             // ct_miss_wide2 only: > 0 = Ct rows are the packed LEADING ct_r x ct_r block (ct_r (ct_r + 1) / 2 doubles; the
                          // padding of the state carries no loadings, its entries equal Cfull's) -- what recursion_tile_kernel reads;
                          // 0 = the full Rp (Rp + 1) / 2 layout of the other recursion kernels
    double* obs_chunk;    // collapse_miss only (table mode): RecursionArgs::chunk_rows -- one 112-byte row per period: b_t (8), s_t,
                          // n_t log 2 pi + sum log R, the period's NaN bit mask (4 x 64 bits) -- written INSTEAD of bcol .. ldrow and C_t
    int kreal;            // collapse_kernel<32>: > 0 = the loadings' columns kreal .. 31 are zero padding (companion states narrower than their
                          // layout): their sums are written as zeros, not formed; 0 = all columns
    double* obs_table;    // ... and RecursionArgs::chunk_obs, the pass's observation table [B][obs_L][23][64] double2 (chunk-major, period
    int obs_L;            // t = obs_L lane + slot), which the workgroup fills from those rows when its stream is done (dfm_ctbuild.h)
    int lam_w;            // collapse_miss only: > 0 = Lam is [B][N][lam_w], lam_w < 8 (r <= 4 on the 8-wide state: columns lam_w .. 7 of the
                          // kernel's tables are zeros -- b_t, C_t come out 8 wide with a zero padding block); 0 = 8
};

struct RecursionArgs {
    int B, T, N, r;       // r = caller's factor count (<= Rp) for the outputs
    const double* A;      // [B][Rp][Rp]
    const double* Q;      // [B][Rp][Rp]
    const double* mu0;    // [B][Rp]
    const double* P0;     // [B][Rp][Rp]
    const double* bcol; const double* scol; const int* nobs; const double* ldrow;
    const double* Ct; const double* Cfull; const double* ldfull;
    // scratch
    double* ZJtab;        // [B][T+1][2][Rp][Rp]   Z_e, J_e of every distinct covariance step
    double* wtab;         // [B][T][Rp]
    int* eidx;            // [B][T]
    // outputs (caller's r): f_smooth [B][T][r], P_smooth [B][T][r(r+1)/2] or null, loglik [B]
    double* f_smooth; double* P_smooth; double* loglik;
    // EM sufficient statistics (null for a plain pass): S11,S10,S00 [B][Rp][Rp]; f0s [B][Rp]; P0s [B][Rp][Rp]
    double* S11; double* S10; double* S00; double* f0s; double* P0s;
    int* ncov;            // [B] number of distinct covariance steps (diagnostic), or null
    // EM epilogue (all null / 0 for a plain pass): new transition parameters written IN PLACE
    // (each replicate's A, Q, mu0, P0 are read only in the kernel prologue of the same group)
    double* A_out; double* Q_out; double* mu0_out; double* P0_out;   // padded [B][Rp][Rp] / [B][Rp]
    double* S11inv;       // [B][Rp][Rp]  (S11)^-1 for the loadings step
    int* active;          // [B] or null: replicate still iterating (read, then updated by the kernel)
    int* iters;           // [B] or null
    double* ll_path;      // [B][max_iter] or null
    int k, max_iter;      // current EM iteration
    double tol;
    // covariance-form recursion (Q may be singular) and companion states (recursion.hip, COV = true); all 0 otherwise
    int cov;              // 1: covariance-form forward sweep
    int Rc;               // padded width of bcol / Ct / Cfull when narrower than the state (0: same as the state)
    int rl;               // > 0: observation loads on the first rl state components only; outputs / S11 in the Rc layout
    int kdim;             // > 0: companion state of width kdim = rl * p: the M-step keeps the shift rows and Q's zero blocks
    int wave;             // 1: one wave per replicate where recursion_wave.hip supports the shape (Rp = 8, information form)
    int kb;               // > 0 with kdim: block size r of the companion state when the observation loads on MORE than the first
                          // block (AR idiosyncratic terms: rl = 0); 0: the block is rl
    int pair_bmax;        // Rp = 8, information form: batches up to this size run as a covariance wave + a mean wave per replicate
                          // (recursion_pair.hip); 0: never
    int ka;               // > 0 with kdim: only the first ka = r p columns of the transition rows are free (a VAR(p) inside a
                          // state that carries m > p lags); 0: all kdim columns
    int qsing;            // 1 with kdim: the caller passed DFM_F_SINGULAR_Q -- the r x r innovation block itself may be rank deficient:
                          // keep the kernels that never invert it (recursion_comp.hip inverts the block and is skipped)
    int ct_r;             // > 0: Ct holds the packed leading ct_r x ct_r block per period (see CollapseArgs::ct_r; recursion_tile only)
    int rstate;           // the model's state width before padding (0: unknown) -- recursion_tile.hip executes ceil(rstate / 4) of
                          // the 8 block pivots / k-steps of a 32-wide state and keeps the mean vectors in (padding) column 31
    // time-chunked recursion (recursion_chunk.hip; Rp = 8, information form): a replicate's T periods as 64 chunks of chunk_L, one per
    // lane of a wave, each lane warmed up over chunk_W periods; scratch owned by the handle's workspace
    double* chunk_scr;    // [B][chunk_L][22][64][2]  -Z_t (36 packed) and w_t (8) of every period, chunk-major (lane = chunk)
    double* chunk_cst;    // [B][320]  per-replicate constants (K, K', Q^-1 + Phi, Phi, Om_0 + Phi, xi_0, log-det and quadratic constants)
    double* chunk_term;   // [B][96]   P_T|T (36), f_T|T (8), P_0|T (36), f_0|T (8)
    double* chunk_obs;    // [B][chunk_L][23][64][2]  the observation table, chunk-major (period t = chunk_L lane + slot): C_t (36 packed), b_t (8),
                          // s_t, n_t log 2 pi + log det R_t of EVERY slot of all 64 lanes (benign rows beyond the sample)
    int chunk_obs_ready;  // 1: collapse_miss_kernel wrote the table itself (its table mode); 0: launch_recursion_chunk gathers the collapse
                          // kernels' per-period arrays
    int* chunk_fail;      // [B]       1: a chunk boundary did not agree to chunk_tol -- the replicate belongs to the sequential kernel
    int* chunk_skip;      // [B] or null: persists between the iterations of an EM run (k = 0 clears it) -- consecutive failures of the replicate's
                          // boundary check; from three on the chunked kernel leaves it to the sequential one at once (retried every 8th iteration)
    const int* only_if;   // [B] or null: the sequential kernels (recursion_wave / recursion_pair) run replicate b only if only_if[b] != 0
    int chunk_L, chunk_W;
    double chunk_tol;
    // time-chunked recursion_tile (Rp = 32, information form): tile_nc chunks of tile_lc periods per replicate, one workgroup each,
    // warmed up over tile_w periods; boundaries checked to chunk_tol, chunk_fail / only_if as above
    double* tile_scr;     // [B][tile_nc][...]  private tables of the warm-up periods, boundary states, parts of the sums (recursion_tile.hip)
    size_t tile_scr_bytes;
    int tile_nc;          // in: 0 = automatic, 1 = never, n = that many; the launcher fills in the count it uses
    int tile_lc, tile_w;
    int num_cu;
};

// Rp = 8, information form, plain factor model (no companion state): the time-chunked recursion
bool recursion_chunk_supported(int Rpad, const RecursionArgs& a);
size_t recursion_chunk_scratch_bytes(int B, int T);          // chunk_scr
size_t recursion_chunk_obs_bytes(int B, int T);              // chunk_obs
size_t recursion_chunk_rows_bytes(int B, int T);             // chunk_rows
int recursion_chunk_len(int T);                              // chunk_L for a sample of T periods
hipError_t launch_recursion_chunk(const RecursionArgs& a, hipStream_t s);
hipError_t launch_chunk_unbridge(const RecursionArgs& a, hipStream_t s);
bool recursion_wave8_fits(int T);
hipError_t launch_recursion_wave8_fallback(const RecursionArgs& a, hipStream_t s);

// Rp = 32, 17 <= rstate <= 31, information form, plain factor model: four waves per replicate, matrices as MFMA accumulator
// tiles, 4 x 4 block-pivot sweep inverse (recursion_tile.hip)
bool recursion_tile_supported(int Rpad, const RecursionArgs& a);
hipError_t launch_recursion_tile(const RecursionArgs& a, hipStream_t s);
int recursion_tile_chunks(const RecursionArgs& a, int* lc_out, int* w_out);   // chunks per replicate the launch will use (1: sequential)
bool recursion_tile_writes_fail(const RecursionArgs& a);      // the launch checks chunk boundaries / writes chunk_fail (recursion_tile1_kernel or chunks > 1)
size_t recursion_tile_scratch_bytes(int B, int T);            // tile_scr

struct MstepArgs {
    int B, T, N, r;
    const double* panel;  // [B][T][N]
    const double* fsm;    // [B][T][Rp]            smoothed means (padded layout)
    const double* Psm;    // [B][T][Rp(Rp+1)/2]    smoothed covariances, packed
    const double* S11;    // [B][Rp][Rp]
    const double* S11inv; // [B][Rp][Rp]
    double* Dmiss;        // [B][N][Rp(Rp+1)/2] zeroed scratch, or null (register accumulators)
    const int* active;    // [B] or null
    double* Lam_out;      // [B][N][lam_stride]
    double* R_out;        // [B][N]
    int lam_stride;
    int min_cells;        // mstep_lam: a series with fewer observed cells keeps its parameters (0 = 1)
};

// balanced panels at Rp = 16 | 32 (even N): the loadings step as a second streaming pass on the matrix pipe (mstep_wide.hip);
// ws = mstep_wide_workspace bytes; r = the caller's factor count
bool mstep_wide_supported(int Rpad, int N);
size_t mstep_wide_workspace(int B, int N, int Rpad);
hipError_t launch_mstep_wide(const MstepArgs& a, double* ws, int Rpad, int r, int num_cu, hipStream_t s);
// panels with missing cells, Rp = 8 | 16 | 32 (even N): D_i = sum over the series' missing periods of vec(E[f f']) and Sxf_i as one
// product per replicate on the matrix pipe, then a per-series Cholesky solve (mstep_miss.hip); r = the loadings' factor count
bool mstep_miss_supported(int Rpad, int r, int N);
size_t mstep_miss_workspace(int B, int T, int N, int Rpad, int r);
hipError_t launch_mstep_miss(const MstepArgs& a, double* ws, int Rpad, int r, int num_cu, hipStream_t s);
hipError_t launch_collapse(int Rpad, const CollapseArgs& a, hipStream_t s);
// the same contract on the LDS-DMA ring + matrix pipe (collapse_miss.hip): Rp = 8, the shapes of the MFMA collapse
bool collapse_miss_supported(int Rpad, int N);
hipError_t launch_collapse_miss(const CollapseArgs& a, int num_cu, hipStream_t s);
// Rp = 16, covariance form, observation on the first <= 4 state components (VAR(p) factor dynamics): one wave per replicate on the
// matrix pipe, rank-4 updates and the Bryson-Frazier smoother instead of 16 x 16 inversions (recursion_mbf16.hip)
bool recursion_mbf16_supported(int Rpad, const RecursionArgs& a);
hipError_t launch_recursion_mbf16(const RecursionArgs& a, hipStream_t s);
hipError_t launch_cov_epilogue(int Rpad, const RecursionArgs& a, hipStream_t s);
// companion states in blocks of 4 (VAR(p) factor dynamics at r = 4, AR idiosyncratic terms at r = 4): information form as a block
// elimination, one wave per replicate on matrix-pipe tiles, 4 x 4 pivots only (recursion_comp.hip)
bool recursion_comp_supported(int Rpad, const RecursionArgs& a);
hipError_t launch_recursion_comp(int Rpad, const RecursionArgs& a, hipStream_t s);
bool recursion_wave_supported(int Rpad, const RecursionArgs& a);
hipError_t launch_recursion_wave(const RecursionArgs& a, hipStream_t s, int Rpad);
// Rp = 8, information form, B <= ~1.5 x the SIMD count: the replicate split over a covariance wave and a mean wave (recursion_pair.hip)
bool recursion_pair_supported(const RecursionArgs& a);
hipError_t launch_recursion_pair(const RecursionArgs& a, hipStream_t s);
int collapse_max_n(int Rpad);
// balanced panels (no NaN), even N: LDS-DMA streaming collapse (writes bcol, scol only) + Gram kernel
bool collapse_dma_supported(int Rpad, int N);
hipError_t launch_collapse_dma(int Rpad, const CollapseArgs& a, hipStream_t s, int variant);
hipError_t launch_gram(int Rpad, const CollapseArgs& a, hipStream_t s);
// same contract as launch_collapse_dma, contraction on the fp64 matrix pipe (collapse_mfma.hip)
bool collapse_mfma_supported(int Rpad, int N);
bool collapse_mfma_fuses_cov(int Rpad, int N);   // launch_collapse_mfma honours CollapseArgs::fuse_cov for this shape
hipError_t launch_collapse_mfma(int Rpad, const CollapseArgs& a, hipStream_t s, int variant);
// cross-sections beyond the register tilings above (collapse_wide.hip): one wave per 16-period tile, weights
// re-read from L2 per tile; per-tile partial sums of s_t go to scol[b][tile]
bool collapse_wide_supported(int Rpad, int N);
int collapse_wide_tiles(int T);
hipError_t launch_collapse_wide(int Rpad, const CollapseArgs& a, hipStream_t s);
hipError_t launch_gram_wide(int Rpad, const CollapseArgs& a, hipStream_t s);
// Rp = 32 on the LDS-DMA path (collapse_wide2.hip): launch_wide_prep writes W = lam / R ([B][N][32] workspace), Cfull and
// ldfull (the Gram kernel's outputs); launch_collapse_wide2 then streams the panel (partials of sum_t s_t: scol[b][tile])
bool collapse_wide2_supported(int Rpad, int N);
int collapse_wide2_tiles(int T);
size_t collapse_wide2_ws_bytes(int B, int N, int Rpad);    // W [B][N][Rp] | 1 / R, log R [B][N padded to 32] | tile queue counters
// r > 0: the caller's factor count (skips the W table when launch_collapse_wide2 will not read it); V: [B][N][32] lam / sqrt(R) for
// launch_ct_miss_wide, or null
hipError_t launch_wide_prep(const CollapseArgs& a, double* ws, int Rpad, hipStream_t s, int r = 0, double* V = nullptr);
hipError_t launch_collapse_wide2(const CollapseArgs& a, double* ws, int Rpad, int r, int num_cu, hipStream_t s);
// a.nobs != nullptr selects the variant for panels with missing cells (per-period scol / nobs / ldrow); their C_t:
bool ct_miss_wide_compact_ok(int N, int ct_r);   // compact C_t rows (CollapseArgs::ct_r) are available for this cross-section
hipError_t launch_ct_miss_wide(const CollapseArgs& a, double* ws, int r, hipStream_t s, const double* V = nullptr);   // r: the caller's factor count; V: launch_wide_prep's
bool gram_supported(int Rpad, int N);       // launch_gram's register tilings
hipError_t launch_recursion(int Rpad, const RecursionArgs& a, hipStream_t s);
hipError_t launch_mstep_lam(int Rpad, const MstepArgs& a, hipStream_t s);
bool mstep_needs_dmiss(int Rpad, int N);
// balanced panels: the same M-step with the contraction on the matrix pipe (mstep_mfma.hip); workspace for the
// per-segment partial sums
bool mstep_mfma_supported(int Rpad, int N);
size_t mstep_mfma_workspace(int B, int N, int Rpad, int wpr);
struct EmUpdArgs;
// ua != nullptr (Rp <= 8): the transition M-step of every replicate runs as extra workgroups at the front of the streaming launch
// (dfm_em_update.h) instead of as em_update_kernel's own launch
hipError_t launch_mstep_mfma(int Rpad, const MstepArgs& a, int wpr, double* workspace, hipStream_t s, const EmUpdArgs* ua = nullptr);

// Series block of the ECM iteration with AR(q) idiosyncratic terms (mstep_ar.hip): loadings, AR coefficients and innovation
// variances from the smoothed moments of the companion state of the quasi-differenced model.
struct ArMstepArgs {
    int B, T, N, r, q, Rk;      // T: rows of the ORIGINAL panel (the moments cover its rows q .. T-1); Rk: padded state width
    const double* panel;        // [B][T][N], NaN = missing
    const double* zsm;          // [B][T-q][Rk]             smoothed means of z_t = (f_t, .., f_{t-m+1})
    const double* Psm;          // [B][T-q][Rk(Rk+1)/2]     smoothed covariances, packed lower
    const int* active;          // [B] or null
    double* Lam;                // [B][N][r]   in / out
    double* rho;                // [B][N][q]   in / out
    double* sig2;               // [B][N]      in / out
};
bool mstep_ar_supported(int r, int q);
size_t mstep_ar_workspace(int B, int T, int N, int r, int q, int Rk);
// ws: mstep_ar_workspace bytes (ar_moments_kernel + ar_solve_kernel)
hipError_t launch_mstep_ar(const ArMstepArgs& a, double* ws, hipStream_t s);
// V[b][t][tt16] = [vec(E f f' + P) of the leading r states (packed lower), zeros to ntm16 | f_t (Rp columns), zeros] (mstep_miss.hip)
hipError_t launch_mmw_vec(const double* fsm, const double* Psm, const int* active, int B, int T, int r, int Rp, int ntm16, int tt16,
                          double* V, hipStream_t s);

// Parametric model with OBSERVED factors (mstep_obs.hip): x_it = lam_o,i' g_t + lam_u,i' f_t + e_it with g_t
// known regressors.  E-step on the residual panel y = x - Lam_o g; loadings by one joint regression per series on z = (g, f).
struct ObsArgs {
    int B, T, N, ro, ru, Rl;    // Rl: width of the rows of fsm (padded loadings width); Psm rows: Rl (Rl + 1) / 2 packed
    const double* panel;        // [B][T][N], NaN = missing
    const double* G;            // [B][T][ro]  observed factors (no NaN)
    const double* fsm;          // [B][T][Rl]
    const double* Psm;          // [B][T][Rl(Rl+1)/2]
    const int* active;          // [B] or null
    double* Lam;                // [B][N][ro + ru]  observed-factor loadings first   in / out
    double* R;                  // [B][N]                                             in / out
};
bool mstep_obs_supported(int ro, int ru);
hipError_t launch_mstep_obs(const ObsArgs& a, hipStream_t s);
// r_o + r_u in 9 .. 32: the joint regression is the ORDINARY loadings step (launch_mstep_lam at Re = 16 | 32) on augmented
// moments z = (g, f), Var z = blockdiag(0, P) -- and the identity on the padding, as the padded state of an ordinary pass has it.
// launch_obs_augment builds z [B][T][Re], Var z packed [B][T][Re (Re + 1) / 2], the padded copy of the loadings [B][N][Re] and
// sum_t E z z' with its inverse [B][Re][Re] each.
bool mstep_obs_wide_supported(int ro, int ru);
int mstep_obs_wide_width(int ro, int ru);                      // Re
hipError_t launch_obs_augment(const ObsArgs& a, int Re, double* z, double* Vz, double* LamAug, double* S11, double* S11inv, hipStream_t s);
// y = x - Lam_o g (NaN stays NaN) and LamP [B][N][Rl] = unobserved-factor loadings, zero padded
hipError_t launch_obs_residual(const ObsArgs& a, double* y, double* LamP, hipStream_t s);

// Balanced-panel fast path (fastpath.hip): data-independent covariance steps (cov_kernel) and the
// time-parallel mean recursion (meanscan_kernel).  All matrices in the padded dimension Rp.
struct FastArgs {
    int B, T, N, r, L;          // r = caller's factor count for the outputs; L = chunk length (power of 2)
    const double* A; const double* Q; const double* mu0; const double* P0;   // [B][Rp][Rp] / [B][Rp]
    const double* Cfull; const double* ldfull;                               // gram_kernel outputs (Lam == nullptr)
    const double* Lam; const double* Rv;                                     // or: cov_kernel computes them itself
    // cov_kernel -> meanscan_kernel
    double* tab;                // [B][T][3][Rp][Rp]  Z_e, J_e, G_e of the distinct covariance steps
    int* E;                     // [B]  number of distinct steps; step t uses entry min(t, E-1)
    double* stead;              // [B][fast_stead_mats][Rp][Rp]  steady Z, J, G, G^(L 2^k), J^(L 2^k)
    double* xi0;                // [B][Rp]
    double* PT;                 // [B][Rp][Rp]   P_T|T
    double* llc;                // [B]  data-independent part of -2 loglik
    int* fill;                  // [B][2]  rows [lo, hi) of P_smooth equal to PsInf
    double* PsInf;              // [B][Rp][Rp]
    double* SP11; double* SU; double* P0s;   // EM covariance sums (or null)
    // collapse -> meanscan
    const double* bcol; const double* ssum;                                  // ssum [B][kSsumSlots]
    int nseg;                   // slots of ssum that were written
    const double* scol; int ntile;   // ntile > 0: sum_t s_t = sum of scol[b][0 .. ntile) instead (collapse_wide)
    double* wtab;               // [B][T][Rp] scratch
    size_t wrep;                // doubles per replicate of wtab (0: T Rp)
    int bst;                    // doubles between the rows of bcol (0: Rp) -- CollapseArgs::bst of the collapse that wrote them
    // outputs
    double* f_smooth; double* P_smooth; double* loglik;
    double* f0s;                // [B][Rp] E[f_0 | X] (EM) or null
    int b0;                     // meanscan: first replicate of this launch (sub-batch pipelining)
    int abl;                    // diagnostics (DFM_SCAN_ABL): bit0 skip the P_smooth fill, bit1 skip the scans
    int rstate;                 // the model's state width (<= Rp; 0 = unknown: Rp) -- cov_tile_kernel executes ceil(rstate / 4) block pivots
};
struct EmUpdArgs {               // transition M-step + EM bookkeeping after a fast-path E-step (all padded, Rp)
    int B, T;
    const double* fsm;           // [B][T][Rp] smoothed means
    const double* f0s;           // [B][Rp]    E[f_0 | X]
    const double* SP11; const double* SU; const double* P0s; const double* PT;   // cov_kernel: [B][Rp][Rp]
    const double* loglik;        // [B] log-likelihood of this E-step
    double* S11; double* S11inv; // [B][Rp][Rp] for the loadings step
    double* A_out; double* Q_out; double* mu0_out; double* P0_out;
    int* active; int* iters; double* ll_path; int k, max_iter; double tol;   // as RecursionArgs
};
hipError_t launch_em_update(int Rpad, const EmUpdArgs& a, hipStream_t s);
bool em_update_grid_supported(int Rpad);                                     // Rp = 16, 32: element-per-thread form (em_update_grid.hip)
hipError_t launch_em_update_grid(int Rpad, const EmUpdArgs& a, hipStream_t s);
hipError_t launch_cov(int Rpad, const FastArgs& a, hipStream_t s);
bool cov_fuses_gram(int Rpad, int N);   // launch_cov with a.Lam != nullptr is supported for this shape
hipError_t launch_meanscan(int Rpad, const FastArgs& a, hipStream_t s);
hipError_t launch_meanscan_mfma(int Rpad, const FastArgs& a, hipStream_t s);   // Rp = 16, 32: the steady scans on the matrix pipe (scan_mfma32.hip)
hipError_t launch_pfill(int Rpad, const FastArgs& a, hipStream_t s);   // the P_smooth fill of meanscan (then run it with abl bit 0)
// The whole balanced pass in one launch (pass_fused.hip: persistent workgroups, stream / covariance / scan waves) and the
// one-wave-per-replicate covariance recursion as a drop-in for launch_cov (Rp = 8, Cfull / ldfull from gram_kernel).
bool pass_fused_supported(int Rpad, int T, int N);
int pass_fused_pick_nsw(int T, int N, int want);            // stream waves per workgroup that fit the 160 KB of LDS (diagnostic)
// nsw / ncov: stream / covariance waves per workgroup (0 = as many as fit)
hipError_t launch_pass_fused(const CollapseArgs& a, const FastArgs& fa, int nsw, int ncov, int num_cu, hipStream_t s);
hipError_t launch_cov_wave(const FastArgs& a, hipStream_t s);
// the element-per-thread covariance recursion for Rp = 16 / 32 (a workgroup of Rp x Rp threads per replicate; Cfull / ldfull
// from the Gram kernel): what launch_cov runs for those widths unless DFM_COV_ROWS=1
bool cov_grid_supported(int Rpad);
// Rp = 32: the same recursion with four waves per replicate on the matrix pipe (recursion_tile.hip); rstate = the model's state width
bool cov_tile_supported(int Rpad, const FastArgs& a, int rstate);
hipError_t launch_cov_tile(const FastArgs& a, int rstate, hipStream_t s);
hipError_t launch_cov_grid(int Rpad, const FastArgs& a, hipStream_t s);
int fast_chunk_len(int Rpad, int T);
int fast_stead_mats(int Rpad);
int fast_scan_groups(int Rpad);   // time chunks of the mean scan

// PCA initialisation (pca.hip).  Scratch arrays use the padded factor dimension Rp; outputs the caller's r.
struct PcaArgs {
    int B, T, N, r, max_iter;
    const double* panel;        // [B][T][N], no NaN
    double* S;                  // [B][N][N]   X'X
    double* V; double* Y;       // [B][N][Rp]  basis / S V
    double* F;                  // [B][T][Rp]  scores
    double* Lam; double* Rv; double* A; double* Q; double* mu0; double* P0;   // caller's layout (r)
    double* factors;            // [B][T][r] or null
    int* status;                // bit 1 (value 2): subspace iteration stopped at max_iter above its tolerance; or null
    int stop_after;             // diagnostics (DFM_PCA_STOP): > 0: pca_kernel returns after that phase (timing of the phases)
};
hipError_t launch_gram_xx(const PcaArgs& a, hipStream_t s, int variant = 0);   // 0: matrix pipe (N <= 256; even N beyond: gram_xx_wide.hip), 1: VALU
bool gram_xx_wide_supported(int N);
hipError_t launch_gram_xx_wide(const PcaArgs& a, hipStream_t s);
hipError_t launch_pca(int Rpad, const PcaArgs& a, hipStream_t s);

// Non-parametric estimator (als.hip): batched alternating least squares and batched complete-case OLS.
struct AlsArgs {
    int B, T, N, rmax;
    const double* z;            // standardised panels, [T][N] each, NaN = missing; problem b at z + b * z_stride
    long long z_stride;         // 0: every problem reads the same panel
    const int* r_each;          // [B] factors of each problem (<= rmax), or null (all rmax)
    double* F;                  // [B][T][rmax]  in: start (PCA scores); out: factors after the last sweep
    double* Lam;                // [B][N][rmax]  out: loadings of the last sweep (NaN: series without loadings)
    int nt_min, max_iter;
    double tol;
    double* ssr_path; int path_cap;   // [B][path_cap] SSR after every sweep (NaN past the last), or null
    int* iters; double* ssr;    // [B]
    double* R2;                 // [B][N] or null
};
hipError_t launch_als(int Rpad, const AlsArgs& a, hipStream_t s);
bool als_fits(int Rpad, int T, int N);      // factors + loadings fit the 160 KB of LDS

struct OlsArgs {
    int P, T, K;                // problems, rows, regressors (K <= 32)
    const double* X; long long x_stride;          // [T][K] per problem; x_stride 0: shared regressors
    const double* y; long long y_stride, y_inc;   // y_p[t] = y[p * y_stride + t * y_inc]
    int nt_min;                 // fewer complete rows: coefficients NaN
    double* beta;               // [P][K]
    double* resid;              // [P][T] (NaN on dropped rows) or null
    double* ssr; double* tss;   // [P]; tss = sum (y - ybar)^2 over the used rows, or null
    int* nobs;                  // [P] complete rows
};
hipError_t launch_ols(int Rpad, const OlsArgs& a, hipStream_t s);
hipError_t launch_standardize(int B, int T, int N, double* panel, double* mean_out, double* sd_out, hipStream_t s);

// Wild-bootstrap impulse responses of the factor VAR and quantile bands (boot.hip).
struct BootArgs {
    int B, T, ns, p, H;         // draws, periods of the window, variables, lags, horizons
    const double* y;            // [T][ns] the VAR's data over the window (no NaN)
    const double* betahat;      // [1 + ns p][ns] point estimate (constant first)
    const double* resid;        // [T][ns] residuals (rows < p unused)
    const double* signs;        // [B][T] +-1, or null: Rademacher signs from Philox(seed; draw, period)
    uint64_t seed; int64_t first_draw;   // device-drawn signs are a function of (seed, first_draw + d, period)
    double* beta_out;           // [B][1 + ns p][ns] re-estimated coefficients, or null
    double* irf;                // [B][ns][H][ns]  irf[d][i][h][k] = (Q M^h G[:, k])_i of draw d
};
hipError_t launch_var_boot(const BootArgs& a, hipStream_t s);
struct QuantArgs {
    int B, S, nq;               // draws, series per draw, quantiles
    const double* x;            // [B][S]
    const double* q;            // [nq] in (0, 1]
    double* out;                // [nq][S]
};
hipError_t launch_quantiles(const QuantArgs& a, hipStream_t s);

// Batched Chow / QLR statistics with HAC covariance (breaks.hip).
struct ChowArgs {
    int S, Tmax, k, P;          // series, row capacity per series, regressors (before interaction), problems
    const double* y;            // [S][Tmax] complete cases of each series
    const double* X;            // [S][Tmax][k]
    const int* Tlen;            // [S] rows in use
    const int* prob_series; const int* prob_break; const int* prob_q;   // [P]
    double* chow;               // [P]
};
hipError_t launch_chow(const ChowArgs& a, hipStream_t s);

// Device-side synthetic replicates (synth.hip); all arrays in the caller's layout (r).
struct SynthArgs {
    int B, T, N, r;
    uint64_t seed; int64_t first_replicate; double missing_prob;
    double* panel; double* Lam; double* R; double* A; double* Q; double* mu0; double* P0;
    double* fscratch;           // [B][T+1][r]
    double* colstats;           // [B][synth_tiles(T)][2][N]  per-tile column sums (sum x, sum x^2) of the raw cells
};
int synth_tiles(int T);
hipError_t launch_synth(const SynthArgs& a, hipStream_t s);

}  // namespace dfm
