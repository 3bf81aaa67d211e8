/* include/dfm_hip.h -- C-ABI of libdfmhip.so: MI355X (gfx950) batched Kalman filter / RTS smoother /
 * EM for the dynamic factor model of QuantEcon/dynamic_factor_models.
 *
 * What it replaces in the reference.  The reference has NO FFI and NO parametric estimator: it
 * declares the dispatch tag `struct Parametric <: EstimationMethod end` (dfm_functions.ipynb:21-23),
 * writes the state-space form y_t = Q z_t, z_t = M z_{t-1} + G u_t (dfm_functions.ipynb:30-34,
 * matrices filled at :477-492) and implements only `estimate!(m, ::NonParametric)` (:530-543).
 * The entry points below are what a new method `estimate!(m::DFMModel, ::Parametric; ...)` binds
 * with `ccall` (binding shown in INTEGRATION.md / julia/dfm_hip.jl).  Each entry point cites the
 * reference object whose role it fills.
 *
 * Conventions
 *   - all arithmetic and storage: IEEE fp64; integers only for sizes / indices
 *   - panel[b][t][i]  (i fastest; the reference stores T x ns column-major per model,
 *                      dfm_functions.ipynb:89-111 `data`; the Julia shim permutes once)
 *     NaN = missing cell (the reference's `missing`, dfm_functions.ipynb:155-158)
 *   - Lam[b][i][k] (= `lambda`, ns x r, dfm_functions.ipynb:104), R[b][i] (idiosyncratic variance;
 *     the reference keeps `uar_ser`, :106), A[b][r][r] row-major (= VAR(1) block of `M`, :477-492),
 *     Q[b][r][r] (= `seps`, :57 / G G'), mu0[b][r], P0[b][r][r]
 *   - packed symmetric outputs: lower triangle, row-major: idx(i,j) = i(i+1)/2 + j, j <= i
 *   - every function returns an int status: 0 ok; <0 argument error (DFM_E_*); >0 = hipError_t.
 *     No C++ exception or exit() crosses the boundary; dfm_last_error() gives the text.
 *   - "_dev" entry points take DEVICE pointers and only enqueue work on the handle's stream
 *     (asynchronous; inputs must stay valid until the stream is synchronised).  The plain entry
 *     points take HOST pointers, copy in, run, copy out and synchronise (what Julia's ccall binds).
 *   - the caller owns every buffer passed in; the library owns its workspace inside the handle.
 *   - requires Q and P0 positive definite (information-form recursion), 1 <= r <= DFM_MAX_R.
 */
#ifndef DFM_HIP_H
#define DFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFM_MAX_R 32

enum {
    DFM_OK = 0,
    DFM_E_DIMS = -1,          /* B,T,N,r out of range */
    DFM_E_R_UNSUPPORTED = -2, /* r > DFM_MAX_R */
    DFM_E_NULL = -3,          /* required pointer is NULL */
    DFM_E_MISSING = -4,       /* NaN found in the panel but DFM_F_MAY_HAVE_MISSING not set */
    DFM_E_NUMERIC = -5,       /* non-finite log-likelihood in some replicate (non-PD Q/P0/...) */
    DFM_E_NO_DEVICE = -6,     /* no HIP device / extension not usable */
    DFM_E_COMM = -7           /* multi-GPU entry points: RCCL could not be loaded or a collective failed */
};

/* flags */
#define DFM_F_MAY_HAVE_MISSING 1u /* panel may contain NaN: allocate the per-period C_t workspace */
#define DFM_F_SINGULAR_Q 2u       /* Q (state innovation covariance) may be singular or ill-conditioned: run the
                                   * recursion in covariance form (never inverts Q; P0 must still be positive
                                   * definite).  Slower than the default information form; balanced panels lose the
                                   * time-parallel fast path. */

typedef struct dfm_handle dfm_handle;

/* Create a context bound to HIP device `device_id`.  `stream` is a hipStream_t the caller owns
 * (e.g. torch's current stream) or NULL for a stream created and owned by the handle. */
int dfm_create(dfm_handle** h, int device_id, void* stream);
int dfm_destroy(dfm_handle* h);
int dfm_set_stream(dfm_handle* h, void* stream);
/* dfm_synchronize: wait for the handle's stream, THEN read the handle's status word -- DFM_E_MISSING (NaN in a panel that
 * was declared balanced), DFM_E_NUMERIC (a bounded wait inside the one-launch pass ran out: outputs invalid; PCA start not
 * converged).  The word is STICKY: kernels only ever set bits, and the call that reads a non-zero value reports it once and
 * clears it -- so it covers every call enqueued since the previous check (no per-call reset: that was a fill kernel in front
 * of every pass).  The "_dev" entry points only enqueue, so this is where a device-pointer caller learns that a call went
 * wrong; the host-pointer entry points make the same check themselves -- and open a new EPOCH when they start: they read and
 * clear the word at entry, so what they report at the end is their own kernels' (a bit left by an earlier unchecked "_dev" call
 * is discarded there; dfm_last_error is left alone -- the call has not failed).  dfm_check_status is the same call under the name a reader looks for. */
int dfm_synchronize(dfm_handle* h);
int dfm_check_status(dfm_handle* h);
const char* dfm_last_error(const dfm_handle* h);
const char* dfm_version(void);

/* Per-kernel timing with HIP events recorded on the handle's stream around every kernel launch
 * (bench.py's roofline leg; no reference counterpart).  dfm_profile_enable(h, 1) clears and starts,
 * dfm_profile_read synchronises and returns the summed duration and launch count of kernel
 * `kernel_index` (0 .. until it returns DFM_E_DIMS) together with its name. */
int dfm_profile_enable(dfm_handle* h, int on);
int dfm_profile_read(dfm_handle* h, int kernel_index, char* name_out, int name_cap, double* total_ms,
                     int* launches);

/* dfm_hbm_probe: the streaming ceilings of the handle's device measured on the spot with the pass's own access patterns
 * (bench.py reports them beside the 8 TB/s spec peak; no reference counterpart).  mode 0: read-only LDS-DMA ring
 * (global_load_lds_dwordx4 -- the collapse's pattern); 1: 16-byte copy (read + write bytes both counted); 2: write-only.
 * `bytes` >= 16 MB of device memory are allocated for the call and freed; `iters` back-to-back launches are timed with HIP
 * events.  *gbs = GB/s, *ms_per_launch may be NULL. */
int dfm_hbm_probe(dfm_handle* h, size_t bytes, int mode, int iters, double* gbs, double* ms_per_launch);

/* dfm_chunk_fallbacks: diagnostics of the last pass / EM iteration with missing cells that ran on a time-chunked recursion
 * (r <= 8: csrc/recursion_chunk.hip, one chunk per lane; 17 <= r <= 31: the chunks of recursion_tile_kernel, one workgroup each,
 * csrc/recursion_tile.hip; no reference counterpart).  *n_total = replicates of that launch (0: the last call did not use one),
 * *n_failed = replicates whose chunk boundaries did not agree to the tolerance and were redone by the sequential kernel.
 * Synchronises the handle's stream. */
int dfm_chunk_fallbacks(dfm_handle* h, int* n_failed, int* n_total);

/* Bytes of device workspace the handle will hold for a pass / EM call on a (B,T,N,r) problem with these flags (for
 * capacity planning; the larger of the sequential and -- for balanced panels -- the time-parallel plan).  The VAR(p),
 * AR-idiosyncratic, PCA and synthetic-panel entry points add their own scratch on top (a quasi-differenced panel copy,
 * [B][N][N] Gram matrices, ...). */
size_t dfm_workspace_bytes(int B, int T, int N, int r, unsigned flags);

/* --- one full Kalman-smoother pass per replicate (SURVEY.md §8(d) "pass") ---------------------
 * Fills the slot of the reference's unused `Parametric` estimator: E-step at fixed parameters.
 * Outputs: f_smooth[b][t][k] = E[f_t | X] (what the reference stores in `factor`,
 * dfm_functions.ipynb:103), P_smooth[b][t][r(r+1)/2] = Var[f_t | X] packed (may be NULL),
 * loglik[b] = Gaussian log-likelihood of the replicate's observed cells. */
int dfm_ks_pass_batch_dev(dfm_handle* h, int B, int T, int N, int r, const double* panel,
                          const double* Lam, const double* R, const double* A, const double* Q,
                          const double* mu0, const double* P0, double* f_smooth, double* P_smooth,
                          double* loglik, unsigned flags);
int dfm_ks_pass_batch(dfm_handle* h, int B, int T, int N, int r, const double* panel,
                      const double* Lam, const double* R, const double* A, const double* Q,
                      const double* mu0, const double* P0, double* f_smooth, double* P_smooth,
                      double* loglik, unsigned flags);

/* --- EM (Shumway-Stoffer / Banbura-Modugno) ------------------------------------------------------
 * One EM iteration per replicate, parameters updated IN PLACE (device pointers); loglik[b] is the
 * log-likelihood at the parameters passed in.  The role of the reference's per-series OLS
 * (dfm_functions.ipynb:391-415) and factor VAR OLS (:444-468) in the non-parametric path. */
int dfm_em_step_batch_dev(dfm_handle* h, int B, int T, int N, int r, const double* panel,
                          double* Lam, double* R, double* A, double* Q, double* mu0, double* P0,
                          double* loglik, unsigned flags);
/* max_iter EM iterations; loglik_path[b][k] = log-likelihood at the parameters entering iteration
 * k (NaN for k >= iters[b]); a replicate stops after iteration k >= 1 when
 * (ll_k - ll_{k-1}) / (0.5 (|ll_k| + |ll_{k-1}|)) < tol (tol <= 0: run all iterations); iters[b] =
 * iterations run.  A replicate that stops this way keeps the parameters that ENTERED its last
 * iteration (that M-step is discarded), exactly as oracle/kalman_oracle.py em().  Afterwards
 * f_smooth/P_smooth (may be NULL) hold the smoother output of the last E-step that was run.
 * Host-pointer variant copies parameters in and out. */
int dfm_em_batch_dev(dfm_handle* h, int B, int T, int N, int r, const double* panel, double* Lam,
                     double* R, double* A, double* Q, double* mu0, double* P0, int max_iter,
                     double tol, double* loglik_path, int* iters, double* f_smooth,
                     double* P_smooth, unsigned flags);
int dfm_em_batch(dfm_handle* h, int B, int T, int N, int r, const double* panel, double* Lam,
                 double* R, double* A, double* Q, double* mu0, double* P0, int max_iter, double tol,
                 double* loglik_path, int* iters, double* f_smooth, double* P_smooth,
                 unsigned flags);

/* EM iteration number k (0-based) of max_iter with the bookkeeping of dfm_em_batch_dev kept in CALLER-owned device
 * arrays that persist between calls: loglik_path [B][max_iter] and iters [B] (both initialised by the k = 0 call) and
 * active [B] (written by every call: 1 while the replicate keeps iterating).  dfm_em_batch_dev is this call for k = 0,
 * 1, ... until no replicate of its batch is active; a multi-GPU driver (replicates sharded over the
 * GPUs, no data-path collective) calls it on every shard, all-gathers {loglik_path[:, k], active} -- north_star's
 * "single allgather at the end of each EM iteration" -- and stops when no replicate ANYWHERE is active:
 * dynamic_factor_models_amd/shard.py em_batch_sharded (one process per GPU, torch.distributed over RCCL) and
 * dfm_em_batch_multi below (one process, one host thread per GPU).  f_smooth / P_smooth (may be NULL) receive the
 * smoother output of this iteration's E-step. */
int dfm_em_iterate_batch_dev(dfm_handle* h, int B, int T, int N, int r, const double* panel, double* Lam, double* R,
                             double* A, double* Q, double* mu0, double* P0, int k, int max_iter, double tol,
                             double* loglik_path, int* iters, int* active, double* f_smooth, double* P_smooth,
                             unsigned flags);

/* --- the same operations on SEVERAL GPUs of one node from ONE process (a library-owned object with
 * `ngpu`, `device_ids`, per-GPU handles / streams / workspaces and ONE RCCL communicator; what
 * `estimate!(m, ::Parametric; nrep, ngpu)` of julia/dfm_hip.jl binds -- Julia has no torch.distributed).
 * GPU g of ngpu owns replicates [g B / ngpu, (g+1) B / ngpu) of a job (the partition of shard.py replicate_range); no
 * data-path collective.  One host thread per GPU inside a call; after every EM iteration ONE ncclAllGather of {loglik,
 * active} ([ceil(B/ngpu)][2] doubles per GPU) over xGMI gives every thread the global convergence state, and all GPUs stop
 * at the same iteration.
 *
 * dfm_multi_create: device_ids[ngpu] (NULL: 0 .. ngpu-1) must be distinct.  The communicator (ncclCommInitAll) is created
 * here, once, when ngpu > 1 or DFM_MULTI_F_FORCE_COMM is set (a 1-rank communicator: the exchange path of the EM loop then
 * runs -- and is tested -- on a single GPU); RCCL is bound with dlopen at that moment (DFM_E_COMM if it cannot be).
 * The replicates of a job live ON THE DEVICES between calls ("resident" job):
 *   dfm_multi_load   uploads host arrays (layouts of dfm_em_batch) to their owners;
 *   dfm_multi_synth  generates them where they live -- GPU g calls dfm_synth_panels_dev with first_replicate + lo_g, so
 *                    BASELINE configs[2] (65 536 replicates, 52 GB of panels) never crosses PCIe; pca_start != 0 replaces
 *                    the DGP parameters by the PCA + OLS start (dfm_pca_init_batch_dev; balanced panels only);
 *   dfm_multi_ks_pass / dfm_multi_em  run on the resident job (EM updates the resident parameters in place);
 *   dfm_multi_fetch  copies one resident array of the whole job back to the host, in global replicate order. */
typedef struct dfm_multi dfm_multi;
#define DFM_MULTI_F_FORCE_COMM 1u
enum {  /* dfm_multi_fetch `what`; element type double unless noted */
    DFM_MULTI_LAM = 0, DFM_MULTI_R, DFM_MULTI_A, DFM_MULTI_Q, DFM_MULTI_MU0, DFM_MULTI_P0,
    DFM_MULTI_F_SMOOTH,    /* [B][T][r]         of the last pass / last E-step */
    DFM_MULTI_P_SMOOTH,    /* [B][T][r(r+1)/2]  (only when the last call asked for it) */
    DFM_MULTI_LOGLIK,      /* [B]               of the last dfm_multi_ks_pass */
    DFM_MULTI_LOGLIK_PATH, /* [B][max_iter]     of the last dfm_multi_em */
    DFM_MULTI_ITERS,       /* int [B]           of the last dfm_multi_em */
    DFM_MULTI_PANEL        /* [B][T][N] */
};
int dfm_multi_create(dfm_multi** m, int ngpu, const int* device_ids, unsigned mflags, char* err, int err_cap);
int dfm_multi_destroy(dfm_multi* m);
int dfm_multi_ngpu(const dfm_multi* m);
int dfm_multi_has_comm(const dfm_multi* m);          /* 1 when the object owns an RCCL communicator */
const char* dfm_multi_last_error(const dfm_multi* m);
int dfm_multi_load(dfm_multi* m, int B, int T, int N, int r, const double* panel, const double* Lam, const double* R,
                   const double* A, const double* Q, const double* mu0, const double* P0);
int dfm_multi_synth(dfm_multi* m, uint64_t seed, int64_t first_replicate, int B, int T, int N, int r,
                    double missing_prob, int pca_start);
/* One smoother pass of every resident replicate (want_P: also P_smooth).  Synchronises every GPU. */
int dfm_multi_ks_pass(dfm_multi* m, int want_P, unsigned flags);
/* The EM loop of dfm_em_batch on the resident job: per iteration dfm_em_iterate_batch_dev on every shard, then the
 * all-gather; *iterations_run (may be NULL) = iterations every GPU ran.  want_smooth: keep f_smooth (and, with want_P,
 * P_smooth) of the last E-step resident for dfm_multi_fetch. */
int dfm_multi_em(dfm_multi* m, int max_iter, double tol, int want_smooth, int want_P, unsigned flags,
                 int* iterations_run);
int dfm_multi_fetch(dfm_multi* m, int what, void* dst);

/* Handle-less forms (round-2 ABI, kept): create + load + run + fetch + destroy in one call.  HOST pointers, layouts as
 * dfm_em_batch / dfm_ks_pass_batch; err[err_cap] (may be NULL) receives the message of the first failing GPU. */
int dfm_em_batch_multi(int ngpu, const int* device_ids, int B, int T, int N, int r, const double* panel, double* Lam,
                       double* R, double* A, double* Q, double* mu0, double* P0, int max_iter, double tol,
                       double* loglik_path, int* iters, double* f_smooth, double* P_smooth, unsigned flags,
                       int* iterations_run, char* err, int err_cap);
int dfm_ks_pass_batch_multi(int ngpu, const int* device_ids, int B, int T, int N, int r, const double* panel,
                            const double* Lam, const double* R, const double* A, const double* Q, const double* mu0,
                            const double* P0, double* f_smooth, double* P_smooth, double* loglik, unsigned flags,
                            char* err, int err_cap);

/* --- VAR(p) factor dynamics ------------------------------------------------------
 *   x_t = Lam f_t + e_t,   f_t = A_1 f_{t-1} + ... + A_p f_{t-p} + eta_t,  eta_t ~ N(0, Q)
 * the parametric model with the reference's `n_factorlag` lags (DFMModel, dfm_functions.ipynb:120-146), run in the
 * companion form its `fill_matrices!` builds for the factor VAR (dfm_functions.ipynb:477-492): state
 * z_t = (f_t, .., f_{t-p+1}), k = r p <= DFM_MAX_R, transition [A_1 .. A_p; I 0], innovation covariance [Q 0; 0 0]
 * (singular: covariance-form recursion, as with DFM_F_SINGULAR_Q).
 *   Avar [B][r][r p] = [A_1 .. A_p],  Q [B][r][r],  mu0 [B][r p], P0 [B][r p][r p] = moments of z_0 (P0 positive definite)
 * Q itself (the r x r block) positive definite, or pass DFM_F_SINGULAR_Q: at r = 4 the companion recursion eliminates the state
 * in 4 x 4 blocks and inverts that block (recursion_comp.hip; a block that is not positive definite gives a NaN log-likelihood =
 * DFM_E_NUMERIC); with the flag the kernels that never invert it run instead (slower at r = 4).  Same for dfm_*_ar_*.
 * Outputs as dfm_ks_pass_batch / dfm_em_batch, for f_t = z_t[:r].  The M-step re-estimates Lam, R, [A_1..A_p], Q,
 * mu0, P0 and keeps the companion structure (oracle/varp_oracle.py).  p = 1 is dfm_em_batch's model. */
int dfm_ks_pass_varp_batch_dev(dfm_handle* h, int B, int T, int N, int r, int p, const double* panel,
                               const double* Lam, const double* R, const double* Avar, const double* Q,
                               const double* mu0, const double* P0, double* f_smooth, double* P_smooth,
                               double* loglik, unsigned flags);
int dfm_ks_pass_varp_batch(dfm_handle* h, int B, int T, int N, int r, int p, const double* panel,
                           const double* Lam, const double* R, const double* Avar, const double* Q,
                           const double* mu0, const double* P0, double* f_smooth, double* P_smooth,
                           double* loglik, unsigned flags);
int dfm_em_varp_batch_dev(dfm_handle* h, int B, int T, int N, int r, int p, const double* panel, double* Lam,
                          double* R, double* Avar, double* Q, double* mu0, double* P0, int max_iter, double tol,
                          double* loglik_path, int* iters, double* f_smooth, double* P_smooth, unsigned flags);
int dfm_em_varp_batch(dfm_handle* h, int B, int T, int N, int r, int p, const double* panel, double* Lam,
                      double* R, double* Avar, double* Q, double* mu0, double* P0, int max_iter, double tol,
                      double* loglik_path, int* iters, double* f_smooth, double* P_smooth, unsigned flags);

/* --- AR idiosyncratic terms --------------------------------------------------------
 *   x_it = lam_i' f_t + e_it,   e_it = rho_i1 e_i,t-1 + .. + rho_iq e_i,t-q + eps_it,  eps_it ~ N(0, sig2_i)
 * with rho [B][N][q] / sig2 [B][N] in the role of the reference's uar_coef / uar_ser^2 (AR(n_uarlag) of the loading
 * regression residuals, dfm_functions.ipynb:305-311, 405-412) and VAR(p) factor dynamics as above.  The panel is
 * quasi-differenced on the device (x~_it = x_it - sum_l rho_il x_i,t-l, missing when x_it or one of its q lags is), the
 * state is z_t = (f_t, .., f_{t-m+1}), m = max(p, q + 1), r m <= DFM_MAX_R, loadings [lam_i, -rho_i1 lam_i, ..]; the
 * likelihood is conditional on the first q rows.  mu0 [B][r m], P0 [B][r m][r m]: moments of z_q (P0 positive definite).
 * Outputs for the T - q rows q+1..T: f_smooth [B][T-q][r], P_smooth [B][T-q][r(r+1)/2] or NULL, loglik [B].
 * q = 0 is dfm_ks_pass_varp_batch.  (Smoother pass only: rho / sig2 come from the reference's own estimator.) */
int dfm_ks_pass_ar_batch_dev(dfm_handle* h, int B, int T, int N, int r, int p, int q, const double* panel,
                             const double* Lam, const double* sig2, const double* rho, const double* Avar,
                             const double* Q, const double* mu0, const double* P0, double* f_smooth,
                             double* P_smooth, double* loglik, unsigned flags);
int dfm_ks_pass_ar_batch(dfm_handle* h, int B, int T, int N, int r, int p, int q, const double* panel,
                         const double* Lam, const double* sig2, const double* rho, const double* Avar,
                         const double* Q, const double* mu0, const double* P0, double* f_smooth,
                         double* P_smooth, double* loglik, unsigned flags);

/* Joint estimation of the same model: loadings, AR coefficients (the reference's `uar_coef`, dfm_functions.ipynb:305-311,
 * 405-412), innovation variances (`uar_ser`^2), [A_1 .. A_p] and Q by ECM -- per iteration one smoother pass of the
 * quasi-differenced model at the current rho, the transition step (a VAR(p) inside the state of max(p, q + 1) lags), then
 * per series the loadings given rho, rho given the new loadings, and sig2 (all from the smoothed moments of the companion
 * state).  Every conditional step maximises its block of the expected complete-data likelihood, so loglik_path
 * [B][max_iter] (the likelihood conditional on the first q rows, at the parameters ENTERING each iteration) is
 * non-decreasing; bookkeeping (tol, iters) as dfm_em_batch.  Parameters are updated in place; rho [B][N][q]; mu0 / P0:
 * moments of the state at period q.  r <= 8, q <= 4, r * max(p, q + 1) <= 32; f_smooth / P_smooth (may be NULL): the
 * T - q smoothed rows of the last E-step.  Series with fewer than r + q + 1 usable quasi-differenced cells keep their
 * loadings / rho / sig2. */
int dfm_em_ar_batch_dev(dfm_handle* h, int B, int T, int N, int r, int p, int q, const double* panel, double* Lam,
                        double* sig2, double* rho, double* Avar, double* Q, double* mu0, double* P0, int max_iter,
                        double tol, double* loglik_path, int* iters, double* f_smooth, double* P_smooth, unsigned flags);
int dfm_em_ar_batch(dfm_handle* h, int B, int T, int N, int r, int p, int q, const double* panel, double* Lam,
                    double* sig2, double* rho, double* Avar, double* Q, double* mu0, double* P0, int max_iter, double tol,
                    double* loglik_path, int* iters, double* f_smooth, double* P_smooth, unsigned flags);

/* --- OBSERVED factors ------------------------------------------------------------------------------
 *   x_it = lam_o,i' g_t + lam_u,i' f_t + e_it,   e_it ~ N(0, R_i),     f_t = A f_{t-1} + eta_t,  eta_t ~ N(0, Q)
 * The reference's DFMModel has `nfac_o` observed factors in front of the `nfac_u` estimated ones (`factor` is T x nfac_t,
 * dfm_functions.ipynb:89-146; `lambda[:, nfac_o+1:end]` are the unobserved loadings, :364) but its estimator does not work for
 * nfac_o > 0 (:358-359, :371) -- the semantics here are those its data layout implies: g_t = the observed columns of `factor`
 * are KNOWN REGRESSORS of the measurement equation (FAVAR); their joint dynamics with f_t are the reference's own second
 * stage (`estimate_var!` on the whole `factor` matrix, :444-468).  EM: E-step = the ordinary pass on x - Lam_o g; loadings =
 * one joint regression per series on (g_t, f_t) over its observed periods; A, Q, mu0, P0 as dfm_em_batch (oracle/obs_oracle.py).
 * G [B][T][r_o] (no NaN), Lam [B][N][r_o + r_u] with the OBSERVED-factor loadings first, A / Q / P0 [B][r_u][r_u], mu0 [B][r_u];
 * f_smooth [B][T][r_u], P_smooth [B][T][r_u(r_u+1)/2] (may be NULL); bookkeeping (loglik_path, iters, tol) as dfm_em_batch.
 * r_o >= 1, r_u >= 1, r_o + r_u <= 32 (up to 8: a per-series Cholesky in registers; 9 .. 32: the ordinary loadings step on the
 * moments of z = (g, f), N <= 1024).  A series with fewer than r_o + r_u + 1 observed cells keeps its loadings and variance. */
int dfm_em_obs_batch_dev(dfm_handle* h, int B, int T, int N, int r_u, int r_o, const double* panel, const double* G, double* Lam,
                         double* R, double* A, double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path,
                         int* iters, double* f_smooth, double* P_smooth, unsigned flags);
int dfm_em_obs_batch(dfm_handle* h, int B, int T, int N, int r_u, int r_o, const double* panel, const double* G, double* Lam,
                     double* R, double* A, double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path,
                     int* iters, double* f_smooth, double* P_smooth, unsigned flags);

/* --- PCA initialisation (reference: pca_score, dfm_functions.ipynb:179-183, on the standardised
 * balanced panel, :339-348) followed by the OLS start of EM: Lam = OLS(x on F), R = residual
 * variance, A/Q = VAR(1) OLS of F, mu0 = 0, P0 = F'F/T.  Balanced panels only (no NaN). */
int dfm_pca_init_batch_dev(dfm_handle* h, int B, int T, int N, int r, const double* panel,
                           double* Lam, double* R, double* A, double* Q, double* mu0, double* P0,
                           double* factors /* [B][T][r] PCA scores, may be NULL */);
int dfm_pca_init_batch(dfm_handle* h, int B, int T, int N, int r, const double* panel, double* Lam,
                       double* R, double* A, double* Q, double* mu0, double* P0, double* factors);

/* --- the reference's NON-parametric estimator, batched ---------------------------------------------
 * dfm_als_batch: `estimate_factor!` (dfm_functions.ipynb:328-382) for B independent runs -- the runs of
 * `estimate_factor_numbers` / `amengual_watson_test` (:698-768), bootstrap draws, Monte-Carlo replicates.
 * z: the standardised estimation window ([T][N] per run, NaN = missing, `standardize_data` :501-509 applied by
 * the caller); run b reads z + b * z_stride (0: all runs share one panel).  F [B][T][r]: in = starting factors
 * (`pca_score`, :179-183, columns >= r_each[b] ignored), out = factors after the last sweep.  Lam [B][N][r]:
 * loadings of the last sweep, NaN for the series the reference leaves undefined (fewer than nt_min observed
 * periods, :357).  A run stops after the sweep at which |SSR_old - SSR| < tol T N (SSR_old = 0 before the
 * first sweep: at least one sweep always runs, :349-353, :367-368) or after max_iter sweeps.
 * ssr_path [B][path_cap] (may be NULL): SSR after every sweep (`m.fes.ssr`, :366), NaN past iters[b].
 * R2 [B][N] (may be NULL): :372-380.  r_each may be NULL (every run has r factors).  r <= DFM_MAX_R and
 * (T + N) * pad(r) * 8 bytes must fit the 160 KB of LDS (DFM_E_DIMS otherwise). */
int dfm_als_batch_dev(dfm_handle* h, int B, int T, int N, int r, const double* z, long long z_stride,
                      const int* r_each, double* F, double* Lam, int nt_min, int max_iter, double tol,
                      double* ssr_path, int path_cap, int* iters, double* ssr, double* R2);
int dfm_als_batch(dfm_handle* h, int B, int T, int N, int r, const double* z, long long z_stride,
                  const int* r_each, double* F, double* Lam, int nt_min, int max_iter, double tol,
                  double* ssr_path, int path_cap, int* iters, double* ssr, double* R2);

/* dfm_standardize_batch_dev: `standardize_data` (dfm_functions.ipynb:501-509) of B panels IN PLACE (device
 * pointers): per series the mean and the POPULATION standard deviation over the observed cells; NaN stays NaN.
 * mean [B][N], sd [B][N] may be NULL. */
int dfm_standardize_batch_dev(dfm_handle* h, int B, int T, int N, double* panel, double* mean, double* sd);

/* dfm_ols_batch: P complete-case least-squares problems (`ols_skipmissing(..., Balanced())`,
 * dfm_functions.ipynb:242-252; the engine of `estimate_factor_loading!` :391-415, `uar` :305-311 and
 * `estimate_var!` :444-468).  Problem p regresses y_p[t] = y[p * y_stride + t * y_inc] on the rows of X_p =
 * X + p * x_stride ([T][K], x_stride 0 = shared regressors), dropping every row in which y or a regressor is
 * NaN.  Outputs: beta [P][K]; resid [P][T] (NaN on dropped rows; may be NULL); ssr [P]; tss [P] = sum (y -
 * ybar)^2 over the used rows (may be NULL; r2 = 1 - ssr / tss, `compute_r2` :565-569); nobs [P] = rows used.
 * Problems with fewer than max(nt_min, K) complete rows get NaN.  K <= 64. */
int dfm_ols_batch_dev(dfm_handle* h, int P, int T, int K, const double* X, long long x_stride, const double* y,
                      long long y_stride, long long y_inc, int nt_min, double* beta, double* resid, double* ssr,
                      double* tss, int* nobs);
int dfm_ols_batch(dfm_handle* h, int P, int T, int K, const double* X, long long x_stride, const double* y,
                  long long y_stride, long long y_inc, int nt_min, double* beta, double* resid, double* ssr,
                  double* tss, int* nobs);

/* --- wild-bootstrap impulse-response bands of the factor VAR (BASELINE config 5) --------------------
 * The reference stops at the point estimate: `estimate_var!` (dfm_functions.ipynb:444-468), `fill_matrices!`
 * (:477-492: companion M, selector Q, G = lower Cholesky of the residual covariance) and `impulse_response`
 * (:793-816: irf[:, t, k] = Q M^t G[:, k]).  dfm_var_bootstrap_irf runs B recursive-design wild-bootstrap
 * draws of that chain: e*_t = s_t e_t with one Rademacher sign per period, y*_t = c + sum_l A_l y*_{t-l} +
 * e*_t (y*_t = y_t for the first p periods), VAR(p) with constant re-estimated on y*, irf* from its M*, G*.
 * y [T][ns]: the VAR's data over the estimation window (no NaN); betahat [1 + ns p][ns] (constant first) and
 * resid [T][ns] (rows < p ignored): the point estimate, e.g. from dfm_ols_batch.  signs [B][T] (+1 / -1) or
 * NULL: signs drawn on the device (Philox4x32-10 keyed by seed, counter (period, first_draw + d); bit 0 of
 * word 0), a pure function of the GLOBAL draw index: a rank that owns draws [lo, hi) passes first_draw = lo.
 * Outputs: irf [B][ns][H][ns] (variable, horizon, shock); beta_out [B][1 + ns p][ns] or NULL.
 * ns <= 8, 1 + ns p <= 64.  A draw with all signs +1 reproduces the point estimate. */
int dfm_var_bootstrap_irf_dev(dfm_handle* h, int B, int T, int ns, int p, int H, const double* y,
                              const double* betahat, const double* resid, const double* signs, uint64_t seed,
                              int64_t first_draw, double* beta_out, double* irf);
int dfm_var_bootstrap_irf(dfm_handle* h, int B, int T, int ns, int p, int H, const double* y,
                          const double* betahat, const double* resid, const double* signs, uint64_t seed,
                          int64_t first_draw, double* beta_out, double* irf);
/* Nearest-rank quantiles over draws: x [B][S] -> out [nq][S], out[j][s] = the ceil(q[j] B)-th smallest of
 * x[0..B)[s] (NaN draws sort last).  B <= 16384 (the draws of one series are sorted in LDS). */
int dfm_quantile_bands_dev(dfm_handle* h, int B, int S, int nq, const double* x, const double* q, double* out);
int dfm_quantile_bands(dfm_handle* h, int B, int S, int nq, const double* x, const double* q, double* out);

/* --- structural-break statistics ------------------------------------------------------
 * dfm_chow_batch: P Chow statistics with HAC covariance -- `compute_chow` / `regress_hac` / `hac` /
 * `form_hscrc` / `form_kernel` (dfm_functions.ipynb:832-977); a `compute_qlr` (:1019-1047) is the maximum over
 * the problems of one series and bandwidth.  Series s: its complete cases y [S][Tmax], X [S][Tmax][k] (rows
 * 0 .. Tlen[s]-1 in use; k <= 8).  Problem p: series prob_series[p], break date prob_break[p] (the first
 * prob_break[p] rows are "before": D_t = 1 for t >= break, 0-based), Bartlett bandwidth prob_q[p] <= 15 (0 =
 * heteroskedasticity-robust only).  chow[p] = gamma' V22^-1 gamma of y = X beta + (X D) gamma. */
int dfm_chow_batch_dev(dfm_handle* h, int S, int Tmax, int k, const double* y, const double* X, const int* Tlen,
                       int P, const int* prob_series, const int* prob_break, const int* prob_q, double* chow);
int dfm_chow_batch(dfm_handle* h, int S, int Tmax, int k, const double* y, const double* X, const int* Tlen,
                   int P, const int* prob_series, const int* prob_break, const int* prob_q, double* chow);

/* --- synthetic replicates generated on the device (SURVEY.md §8(d) DGP; no reference
 * counterpart -- the reference has no RNG).  Writes the standardised panel and the DGP parameters
 * rescaled to it.  Counter-based generator keyed by (seed, first_replicate + b). */
int dfm_synth_panels_dev(dfm_handle* h, uint64_t seed, int64_t first_replicate, int B, int T, int N,
                         int r, double missing_prob, double* panel, double* Lam, double* R,
                         double* A, double* Q, double* mu0, double* P0);

#ifdef __cplusplus
}
#endif
#endif /* DFM_HIP_H */
