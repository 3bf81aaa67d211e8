// capi.hip -- the C-ABI of libdfmhip.so (include/dfm_hip.h): handle, workspace, entry points.
#include "../../include/dfm_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "dfm_kernels.h"

using namespace dfm;

thread_local const char* dfm::t_launched_kernel = nullptr;

struct dfm_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t side = nullptr;            // data-independent kernels (gram, cov) run beside the collapse
    hipStream_t post = nullptr;            // meanscan of sub-batch s runs here, beside the collapse of sub-batch s+1
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_post = nullptr;
    std::vector<hipEvent_t> ev_sub;        // collapse of sub-batch s done
    int subbatch = 0;                      // DFM_SUBBATCH: sub-batches per fast pass (0 = automatic)
    bool force_general = false;            // DFM_FORCE_GENERAL=1: never take the balanced fast path
    int collapse_variant = 0;              // DFM_COLLAPSE_VARIANT: 0 = automatic; 1..199 VALU kernel tunings; 200 = MFMA kernel
    int collapse_wpr = 0;                  // DFM_COLLAPSE_WPR: period segments (waves) per replicate of the MFMA collapse; 0 = automatic
    int num_cu = 256;
    int scan_abl = 0;
    bool no_side = false;                  // DFM_NO_SIDE=1: gram/cov on the main stream (diagnostics)
    bool no_pipe = false;                  // DFM_PIPE=0 (diagnostics build): large batches with missing cells as ONE batch on one stream (pipe_eligible)
    bool in_pipe = false;                  // inside a sub-batch of pipe_run: no nesting
    // The handle's status word: its own 256-byte allocation, zeroed at creation and again by whoever READS a non-zero value
    // (status_check).  It is sticky between checks -- no memset per call: that was a 5 us fill kernel in front of every pass,
    // 2 % of the headline's step.  Device-pointer callers that never check see nothing; dfm_synchronize / dfm_check_status and
    // every host-pointer entry report (and clear) whatever was raised since the last check.
    int* status_dev = nullptr;
    int discarded_status = 0;              // status bits of earlier, unchecked device-pointer calls that a host-pointer entry cleared (status_epoch)
    bool no_rec_wave = false;              // DFM_NO_RECURSION_WAVE=1: lane-group recursion_kernel also at Rp = 8 (A/B)
    int tile_nc = 0, tile_w = 0;   // DFM_TILE_NC (route): chunks per replicate on recursion_tile_kernel (0 = automatic, 1 = the sequential kernel), DFM_TILE_W: warm-up periods
    bool no_chunk = false; int chunk_w = 0; double chunk_tol = 0.0;   // DFM_NO_CHUNK=1 (route): panels with missing cells at Rp = 8 on the sequential kernels;
                                           // DFM_CHUNK_W=n, DFM_CHUNK_TOL=x (route): warm-up periods / boundary tolerance of recursion_chunk.hip (0 = its defaults: 8, 1e-10)
    int pair_bmax = -1;                    // Rp = 8: batch limit of the covariance-wave + mean-wave pair (recursion_pair.hip); -1 = one replicate per SIMD,
                                           // DFM_PAIR_BMAX=n; DFM_NO_PAIR=1 = 0 (never)
    bool no_pfill = false;                 // DFM_NO_PFILL=1: P_smooth fill inside meanscan (diagnostics)
    bool fused_gram = true;                // DFM_FUSED_GRAM=0: gram_kernel as its own launch in front of the fused collapse launch
    bool no_fuse_cov = false;              // DFM_NO_FUSE_COV=1: cov_kernel / pfill_kernel as their own launches on a forked stream
    bool no_defer_em = false;              // DFM_NO_DEFER_EM=1 (route): em_update_kernel as its own launch behind the E-step
    bool no_mstep_mfma = false;            // DFM_NO_MSTEP_MFMA=1: VALU M-step for balanced panels too (diagnostics)
    bool em_general = false;               // DFM_EM_GENERAL=1: EM of balanced panels on the general path too (diagnostics)
    bool fuse_gram = false;                // DFM_FUSE_GRAM=1: Gram matrices inside cov_kernel instead of gram_kernel (slower: its
                                           // per-series loads are dependent round trips, ~5 us each beside the collapse)
    int pass_fused = 1;                    // the balanced pass at Rp = 8 as ONE launch (pass_fused.hip); DFM_PASS_FUSED=0: two launches
    int pass_nsw = 0;                      // DFM_PASS_NSW: stream waves per workgroup of that launch (0 = automatic)
    bool wide_old = false;                 // DFM_WIDE_OLD=1: Rp = 32 balanced collapse by collapse_wide_kernel + gram_wide_kernel
    bool collapse_miss_old = false;        // DFM_COLLAPSE_MISS_OLD=1: register-streamed collapse_kernel for panels with missing cells
    bool narrow_tab_off = false;           // DFM_NARROW_TAB=0 (diagnostics build): r <= 4 on the 8-wide state through collapse_kernel<4> + chunk_bridge_kernel
    bool gram_xx_valu = false;             // DFM_GRAM_XX_VALU=1: X'X of the PCA start on the VALU kernel (diagnostics)
    int pass_ncov = 0;                     // DFM_PASS_NCOV: covariance waves per workgroup of that launch (0 = automatic)
    bool cov_wave = false;                 // DFM_COV_WAVE=1: one-wave-per-replicate covariance recursion on the separate-launch path
    void* ws = nullptr;
    size_t ws_bytes = 0;
    const int* ck_fail_dev = nullptr; int ck_fail_n = 0;   // chunk_fail of the last launch on recursion_chunk_kernel (dfm_chunk_fallbacks)
    // EM on the fast path at Rp <= 8: the transition M-step is not launched behind the E-step but handed to the loadings step's
    // streaming launch (mstep_mfma.hip runs it as extra workgroups): em_iteration sets defer_em, enqueue_pass_fast parks the
    // arguments here
    bool defer_em = false, have_deferred_em = false;
    dfm::EmUpdArgs deferred_em;
    void* odd = nullptr;                   // panel / loadings / R with one all-missing series appended (odd N beyond the tilings, odd_pad)
    size_t odd_bytes = 0;
    const double* odd_panel_src = nullptr; int odd_panel_dims[3] = {0, 0, 0};   // the panel whose padded copy h->odd holds (odd_pad keep_panel)
    std::string prof_file;                 // DFM_PF_PROF_FILE with DFM_SCAN_ABL=256: phase stamps of the fused pass
    char err[512] = {0};
    // optional per-kernel timing (bench.py roofline leg): event pairs on the launch stream
    bool profiling = false;
    struct Ev { const char* name; hipEvent_t a, b; };   // name: the kernel the launcher dispatched to (static string)
    std::vector<Ev> events;
    std::vector<const char*> prof_names;                // distinct names of `events`, in order of first launch
};

enum KernelId { K_COLLAPSE = 0, K_RECURSION, K_MSTEP_STATS, K_MSTEP_SOLVE, K_PCA, K_SYNTH, K_PAD,
                K_COLLAPSE_DMA, K_GRAM, K_COV, K_MEANSCAN, K_PFILL, K_COLLAPSE_MFMA, K_ALS, K_OLS, K_BOOT, K_QUANT, K_COLLAPSE_WIDE, K_EM_UPDATE, K_CHOW, K_MSTEP_MFMA, K_GRAM_XX, K_PASS_FUSED, K_COUNT };
static const char* const kKernelNames[K_COUNT] = {"collapse_kernel", "recursion_kernel", "mstep_lam_kernel",
                                                  "mstep_solve_kernel", "pca_kernel", "synth_kernel",
                                                  "pad_params_kernel", "collapse_dma_kernel", "gram_kernel",
                                                  "cov_kernel", "meanscan_kernel", "pfill_kernel", "collapse_mfma_kernel", "als_kernel", "ols_kernel", "var_boot_kernel", "quantile_kernel", "collapse_wide_kernel", "em_update_kernel", "chow_kernel", "mstep_mfma_kernel", "gram_xx_kernel", "pass_fused_kernel"};

namespace dfm { int handle_device(const dfm_handle* h) { return h->device; } }   // (probe.hip)

namespace {

int fail(dfm_handle* h, int code, const char* fmt, const char* detail = "") {
    if (h) snprintf(h->err, sizeof(h->err), fmt, detail);
    return code;
}
int hip_fail(dfm_handle* h, hipError_t e, const char* where) {
    if (h) snprintf(h->err, sizeof(h->err), "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}
#define HIP_TRY(h, expr)                                   \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return hip_fail(h, _e, #expr); \
    } while (0)

struct ProfScope {  // records an event pair around one kernel launch when profiling is on
    dfm_handle* h; int idx = -1; hipStream_t st;
    ProfScope(dfm_handle* h_, int kid, hipStream_t st_ = nullptr) : h(h_), st(st_ ? st_ : h_->stream) {
        if (!h->profiling) return;
        dfm_handle::Ev ev; ev.name = kKernelNames[kid];
        if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) return;
        dfm::t_launched_kernel = nullptr;
        (void)hipEventRecord(ev.a, st);
        h->events.push_back(ev);
        idx = (int)h->events.size() - 1;
    }
    ~ProfScope() {
        if (idx < 0) return;
        (void)hipEventRecord(h->events[idx].b, st);
        // the launcher says which kernel it dispatched to (rocprofv3's name); the scope's id is only the default
        if (dfm::t_launched_kernel) h->events[idx].name = dfm::t_launched_kernel;
    }
};

int pad_r(int r) { return pow2_ge(r) < 2 ? 2 : pow2_ge(r); }

struct Plan {  // byte offsets into the workspace (all 256-byte aligned)
    int Rp;
    int r = 0;                                     // the caller's factor count (columns r .. Rp - 1 of the padded Lam are zero)
    size_t LamP, AP, QP, P0P, mu0P;                // padded parameters (only used when r != Rp)
    size_t bcol, scol, ldrow, nobs, Ct, Cfull, ldfull;
    size_t ZJ, wtab, status, ncov;
    size_t ck_skip = (size_t)-1;                  // recursion_chunk.hip: replicates that failed their boundary check earlier in this EM run
    size_t ck_rows = (size_t)-1;                  // collapse_miss_kernel's rows + NaN masks (ct_build_kernel's input)
    size_t ck_scr = (size_t)-1, ck_obs = (size_t)-1, ck_cst = (size_t)-1, ck_term = (size_t)-1, ck_fail = (size_t)-1;   // recursion_chunk.hip (Rp = 8, general path)
    size_t Vwide = (size_t)-1;
    size_t tk_scr = (size_t)-1, tk_bytes = 0;   // recursion_tile.hip (Rp = 32, general path): the chunks' scratch (ck_fail is shared)
    size_t S11, S10, S00, P0s, f0s, fsm, Psm, Sxf, Sxx, Dmiss, llbuf, active;
    // balanced fast path (fastpath.hip); (size_t)-1 when the plan is for the general path
    size_t f_tab, f_E, f_stead, f_xi0, f_PT, f_llc, f_fill, f_PsInf, f_ssum;
    size_t ms_ws = (size_t)-1; int ms_wpr = 0;   // mstep_mfma partial sums (EM on the fast path)
    size_t mw_ws = (size_t)-1;                     // mstep_wide: Sxf, Sxx of the Rp = 32 loadings step (EM on the fast path)
    size_t mm_ws = (size_t)-1;                     // mstep_miss: V, [D | Sxf], Sxx, counts of the loadings step with missing cells
    size_t Wwide = (size_t)-1;                     // W = lam / R of the Rp = 32 collapse (collapse_wide2.hip)
    bool fast;
    // covariance-form recursion (DFM_F_SINGULAR_Q) and companion states (dfm_*_varp_*): see RecursionArgs
    bool cov = false; int Rc = 0, rl = 0, kdim = 0, kb = 0, ka = 0, qsing = 0;
    size_t total;
};

size_t take(size_t& off, size_t bytes) {
    const size_t at = off;
    off += (bytes + 255) & ~(size_t)255;
    return at;
}

// Sequential path with r <= 4: the state is padded to 8 so that the one-wave-per-replicate recursion (recursion_wave.hip)
// applies; collapse, loadings and the loadings M-step stay pad_r(r) wide (Plan::Rc).  DFM_NO_RECURSION_WAVE=1 turns it off.
bool g_widen_small_r = true;
// DFM_NO_CHUNK=1 (route switch; process-wide like g_widen_small_r: follows the most recently created handle): no chunk scratch /
// observation table in the plans (184 KB per replicate at T = 500 that the sequential kernels never touch)
bool g_plan_chunk = true;
// Loadings step with missing cells on the matrix pipe (mstep_miss.hip).  DFM_MSTEP_MISS: 0 = never (mstep_lam_kernel),
// 1 = where mstep_lam_kernel keeps its per-series accumulators in global memory (Rp > 8 or N > 256; default), 2 = wherever supported.
int g_mstep_miss_mode = 1;

// rows of the w_t scratch per replicate: T, plus (fast path, Rp >= 16) the chunk-major region of meanscan_mfma_kernel
static size_t wtab_rows(bool fast, int Rp, int T) {
    return (size_t)T + ((fast && Rp >= 16) ? (size_t)fast_scan_groups(Rp) * fast_chunk_len(Rp, T) : 0);
}
Plan make_plan(int B, int T, int N, int r, unsigned flags, bool em, bool fast = false) {
    Plan p;
    int Rp = pad_r(r);
    p.r = r;
    p.fast = fast;
    p.cov = (flags & DFM_F_SINGULAR_Q) != 0;
    // (one wave per replicate pays while the batch leaves SIMDs idle under the lane-group kernel: measured crossover
    // at r = 4 between B = 1024 (0.39 vs 0.60 ms) and B = 4096 (1.52 vs 0.65 ms))
    if (!fast && !p.cov && Rp < 8 && g_widen_small_r && B <= 1536) {
        p.Rc = Rp; p.rl = Rp;
        Rp = 8;
    }
    p.Rp = Rp;
    const size_t d = sizeof(double), rr = (size_t)Rp * Rp, np = (size_t)Rp * (Rp + 1) / 2;
    size_t off = 0;
    p.LamP = take(off, (size_t)B * N * Rp * d);
    p.AP = take(off, B * rr * d);
    p.QP = take(off, B * rr * d);
    p.P0P = take(off, B * rr * d);
    p.mu0P = take(off, (size_t)B * Rp * d);
    p.bcol = take(off, (size_t)B * T * Rp * d);
    p.scol = take(off, (size_t)B * T * d);
    p.ldrow = take(off, (size_t)B * T * d);
    p.nobs = take(off, (size_t)B * T * sizeof(int));
    p.Ct = (flags & DFM_F_MAY_HAVE_MISSING) ? take(off, (size_t)B * T * np * d) : (size_t)-1;
    p.Cfull = take(off, B * rr * d);
    p.ldfull = take(off, (size_t)B * d);
    p.f_tab = p.f_E = p.f_stead = p.f_xi0 = p.f_PT = p.f_llc = p.f_fill = p.f_PsInf = p.f_ssum = (size_t)-1;
    p.ZJ = (size_t)-1;
    if (fast) {
        p.f_tab = take(off, (size_t)B * T * 3 * rr * d);
        p.f_E = take(off, (size_t)B * sizeof(int));
        p.f_stead = take(off, (size_t)B * fast_stead_mats(Rp) * rr * d);
        p.f_xi0 = take(off, (size_t)B * Rp * d);
        p.f_PT = take(off, B * rr * d);
        p.f_llc = take(off, (size_t)B * d);
        p.f_fill = take(off, (size_t)B * 2 * sizeof(int));
        p.f_PsInf = take(off, B * rr * d);
        p.f_ssum = take(off, (size_t)B * kSsumSlots * d);
        // (+ room for up to 8 sub-batches laid out as batches of their own: enqueue_pass_fast)
        if (collapse_wide2_supported(Rp, N)) p.Wwide = take(off, collapse_wide2_ws_bytes(B, N, Rp) + 8 * collapse_wide2_ws_bytes(1, N, Rp));
    } else {
        p.ZJ = take(off, (size_t)B * (T + 1) * 2 * rr * d);
        if (p.Rc == 0 && Rp == 32 && N > collapse_max_n(32) && collapse_wide2_supported(32, N)) {
            p.Wwide = take(off, collapse_wide2_ws_bytes(B, N, 32));
            p.Vwide = take(off, (size_t)B * N * 32 * d);      // lam / sqrt(R): the C_t kernel's table
        }
    }
    // (fast path, Rp >= 16: the mean scan on the matrix pipe keeps the steady part of w_t in a second, chunk-major region behind the
    // T natural rows -- scan_mfma32.hip)
    p.wtab = take(off, (size_t)B * wtab_rows(fast, Rp, T) * Rp * d);
    if (!fast && !p.cov && Rp == 8 && g_plan_chunk) {
        p.ck_scr = take(off, recursion_chunk_scratch_bytes(B, T));
        p.ck_obs = take(off, recursion_chunk_obs_bytes(B, T));
        if (collapse_miss_supported(8, N)) p.ck_rows = take(off, recursion_chunk_rows_bytes(B, T));   // (also r <= 4 on the 8-wide state: CollapseArgs::lam_w)
        p.ck_cst = take(off, (size_t)B * 320 * d);
        p.ck_term = take(off, (size_t)B * 96 * d);
        p.ck_fail = take(off, (size_t)B * sizeof(int));
        p.ck_skip = take(off, (size_t)B * sizeof(int));
    }
    if (!fast && !p.cov && Rp == 32 && p.Rc == 0) {
        p.tk_bytes = recursion_tile_scratch_bytes(B, T);
        p.tk_scr = take(off, p.tk_bytes);
        p.ck_fail = take(off, (size_t)B * sizeof(int));
    }
    p.status = take(off, 256);
    p.ncov = take(off, (size_t)B * sizeof(int));
    p.S11 = p.S10 = p.S00 = p.P0s = p.f0s = p.fsm = p.Psm = p.Sxf = p.Sxx = p.Dmiss = p.llbuf = p.active = (size_t)-1;
    if (em) {
        p.S11 = take(off, B * rr * d);
        p.S10 = take(off, B * rr * d);
        p.S00 = take(off, B * rr * d);
        p.P0s = take(off, B * rr * d);
        p.f0s = take(off, (size_t)B * Rp * d);
        p.fsm = take(off, (size_t)B * T * Rp * d);
        p.Psm = take(off, (size_t)B * T * np * d);
        p.Sxf = take(off, B * rr * d);                       // S11^-1
        p.llbuf = take(off, (size_t)B * d);
        p.active = take(off, (size_t)B * sizeof(int));
        if (mstep_needs_dmiss(Rp, N)) p.Dmiss = take(off, (size_t)B * N * np * d);
        if (fast && mstep_mfma_supported(Rp, N)) {
            int w = (256 * 12) / B;
            w = w < 1 ? 1 : (w > 8 ? 8 : w);
            while (w > 1 && T / w < 8) --w;
            p.ms_wpr = w;
            p.ms_ws = take(off, mstep_mfma_workspace(B, N, Rp, w));
        }
        if (fast && mstep_wide_supported(Rp, N)) p.mw_ws = take(off, mstep_wide_workspace(B, N, Rp));
        // (sized for loadings as wide as the state: companion models narrow them after the plan is made)
        if (!fast && g_mstep_miss_mode && mstep_miss_supported(Rp, r < Rp ? r : Rp, N) &&
            (g_mstep_miss_mode == 2 || mstep_needs_dmiss(Rp, N)))
            p.mm_ws = take(off, mstep_miss_workspace(B, T, N, Rp, r < Rp ? r : Rp));
    }
    p.total = off;
    return p;
}

int ensure_ws(dfm_handle* h, size_t bytes) {
    // every entry point sizes the workspace before it uses it: whatever chunk_fail flags the last pass left in the block are about to
    // be overwritten or freed -- dfm_chunk_fallbacks must not read them (enqueue_pass sets the pointer again behind its launch)
    h->ck_fail_dev = nullptr; h->ck_fail_n = 0;
    if (bytes <= h->ws_bytes) return 0;
    if (h->ws) {
        // every stream this handle has launched on (the caller may have swapped streams with dfm_set_stream, and the
        // side / post streams of the fast path) must be done with the old block before it goes back to the allocator
        HIP_TRY(h, hipDeviceSynchronize());
        HIP_TRY(h, hipFree(h->ws));
        h->ws = nullptr;
        h->ws_bytes = 0;
    }
    HIP_TRY(h, hipMalloc(&h->ws, bytes));
    h->ws_bytes = bytes;
    return 0;
}

template <class T>
T* at(dfm_handle* h, size_t off) {
    return off == (size_t)-1 ? nullptr : reinterpret_cast<T*>(static_cast<char*>(h->ws) + off);
}

int check_dims(dfm_handle* h, int B, int T, int N, int r) {
    if (!h) return DFM_E_NULL;
    if (B < 1 || T < 1 || N < 1 || r < 1) return fail(h, DFM_E_DIMS, "B, T, N, r must be >= 1%s");
    if (r > DFM_MAX_R) return fail(h, DFM_E_R_UNSUPPORTED, "r > DFM_MAX_R (32)%s");
    return 0;
}
// panels with missing cells (and EM) go through collapse_kernel's register tiling
// plain = the factor model itself (loadings as wide as the state): at Rp = 32 cross-sections beyond the register tiling take
// the streaming collapse of config 4 in its variant for missing cells (collapse_wide2.hip)
int check_general_n(dfm_handle* h, int N, int r, bool plain = false) {
    if (plain && pad_r(r) == 32 && (collapse_wide2_supported(32, N) || collapse_wide2_supported(32, N + 1))) return 0;   // odd N: odd_pad
    if (N > collapse_max_n(pad_r(r)))
        return fail(h, DFM_E_DIMS, "N too large for this r on the path with missing cells / EM (collapse kernel register "
                                   "tiling: N <= 1024 for r <= 8, 512 for r <= 16; r > 16: 256, or any even N for the plain model)%s");
    return 0;
}

// EM: balanced panels keep the fast-path E-step at any N (the wide collapse has no register tiling); what bounds them
// is the loadings M-step (mstep_lam_kernel: lane = series, N <= 1024; BASELINE config 4 is N = 1000, r = 20).
int check_em_n(dfm_handle* h, int N, int r, unsigned flags);

// Embed caller parameters (factor dimension r) into the padded dimension Rp.
__global__ void pad_params_kernel(int B, int N, int r, int Rp, int Rl, const double* Lam, const double* A,
                                  const double* Q, const double* mu0, const double* P0, double* LamP,
                                  double* AP, double* QP, double* mu0P, double* P0P) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nl = (size_t)B * N * Rl, nm = (size_t)B * Rp * Rp, nv = (size_t)B * Rp;
    if (tid < nl) {
        const int k = tid % Rl;
        const size_t bn = tid / Rl;
        LamP[tid] = k < r ? Lam[bn * r + k] : 0.0;
    }
    if (tid < nm) {
        const int j = tid % Rp, i = (tid / Rp) % Rp;
        const size_t b = tid / ((size_t)Rp * Rp);
        const bool in = i < r && j < r;
        const double eye = (i == j) ? 1.0 : 0.0;
        AP[tid] = in ? A[(b * r + i) * r + j] : 0.0;
        QP[tid] = in ? Q[(b * r + i) * r + j] : eye;
        P0P[tid] = in ? P0[(b * r + i) * r + j] : eye;
    }
    if (tid < nv) {
        const int i = tid % Rp;
        const size_t b = tid / Rp;
        mu0P[tid] = i < r ? mu0[b * r + i] : 0.0;
    }
}

struct PaddedParams {
    const double *Lam, *A, *Q, *mu0, *P0;
};

int pad_params(dfm_handle* h, const Plan& p, int B, int N, int r, const double* Lam, const double* A,
               const double* Q, const double* mu0, const double* P0, PaddedParams* out) {
    if (r == p.Rp) {
        *out = PaddedParams{Lam, A, Q, mu0, P0};
        return 0;
    }
    const size_t n = (size_t)B * N * p.Rp > (size_t)B * p.Rp * p.Rp ? (size_t)B * N * p.Rp : (size_t)B * p.Rp * p.Rp;
    const int threads = 256;
    const unsigned blocks = (unsigned)((n + threads - 1) / threads);
    hipLaunchKernelGGL(pad_params_kernel, dim3(blocks), dim3(threads), 0, h->stream, B, N, r, p.Rp, p.Rc ? p.Rc : p.Rp, Lam, A, Q,
                       mu0, P0, at<double>(h, p.LamP), at<double>(h, p.AP), at<double>(h, p.QP),
                       at<double>(h, p.mu0P), at<double>(h, p.P0P));
    HIP_TRY(h, hipGetLastError());
    *out = PaddedParams{at<double>(h, p.LamP), at<double>(h, p.AP), at<double>(h, p.QP), at<double>(h, p.mu0P),
                        at<double>(h, p.P0P)};
    return 0;
}


// ---- odd N beyond the register tiling, panel with missing cells, r > 16 ------------------------------------------------
// The streaming collapse for missing cells (collapse_wide2.hip) and the matrix-pipe loadings step (mstep_miss.hip) move
// 16-byte series pairs.  The reference's estimator takes any cross-section (dfm_functions.ipynb:352-366 exists because panels
// are unbalanced), so an odd N gets ONE series appended: every cell missing, loadings 0, R = 1.  Its contribution to b_t, C_t,
// n_t, s_t and sum log R over the observed cells is exactly 0, so the pass equals the N-series pass; the loadings step skips
// a series without observed cells (mmw_finish_kernel), and its parameters are dropped on the way out.
__global__ void pad_last_col_kernel(size_t rows, int N, const double* src, double* dst) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= rows * (size_t)(N + 1)) return;
    const size_t row = tid / (size_t)(N + 1);
    const int i = (int)(tid % (size_t)(N + 1));
    dst[tid] = i < N ? src[row * N + i] : __builtin_nan("");
}
// dst[b][n][k] (n <= N) from src[b][n][k] (n < N), `fill` for the appended row; N1 = rows of dst per b (N + 1), or N to un-pad
__global__ void copy_series_rows_kernel(size_t nb, int Ns, int Nd, int w, double fill, const double* src, double* dst) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nb * (size_t)Nd * w) return;
    const int k = (int)(tid % w);
    const int n = (int)((tid / w) % Nd);
    const size_t b = tid / ((size_t)w * Nd);
    dst[tid] = n < Ns ? src[(b * Ns + n) * w + k] : fill;
}
struct OddPad { double *panel, *Lam, *R; };
// keep_panel: the call continues an EM run on the same panel (dfm_em_iterate_batch_dev with k > 0: one call per iteration from the
// multi-GPU drivers) -- the padded copy made at k = 0 is still in h->odd, only the loadings and variances are copied again
int odd_pad(dfm_handle* h, int B, int T, int N, int r, const double* panel, const double* Lam, const double* R, OddPad* out,
            bool keep_panel = false) {
    const size_t n_panel = (size_t)B * T * (N + 1), n_lam = (size_t)B * (N + 1) * r, n_R = (size_t)B * (N + 1);
    const size_t bytes = (n_panel + n_lam + n_R) * sizeof(double) + 768;
    if (bytes > h->odd_bytes) {
        if (h->odd) { HIP_TRY(h, hipDeviceSynchronize()); HIP_TRY(h, hipFree(h->odd)); h->odd = nullptr; h->odd_bytes = 0; }
        h->odd_panel_src = nullptr;
        HIP_TRY(h, hipMalloc(&h->odd, bytes));
        h->odd_bytes = bytes;
    }
    auto al = [](size_t n) { return (n + 31) & ~(size_t)31; };
    out->panel = static_cast<double*>(h->odd);
    out->Lam = out->panel + al(n_panel);
    out->R = out->Lam + al(n_lam);
    auto grid = [](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    const bool same = keep_panel && h->odd_panel_src == panel && h->odd_panel_dims[0] == B && h->odd_panel_dims[1] == T && h->odd_panel_dims[2] == N;
    if (!same) hipLaunchKernelGGL(pad_last_col_kernel, grid(n_panel), dim3(256), 0, h->stream, (size_t)B * T, N, panel, out->panel);
    h->odd_panel_src = panel; h->odd_panel_dims[0] = B; h->odd_panel_dims[1] = T; h->odd_panel_dims[2] = N;
    hipLaunchKernelGGL(copy_series_rows_kernel, grid(n_lam), dim3(256), 0, h->stream, (size_t)B, N, N + 1, r, 0.0, Lam, out->Lam);
    hipLaunchKernelGGL(copy_series_rows_kernel, grid(n_R), dim3(256), 0, h->stream, (size_t)B, N, N + 1, 1, 1.0, R, out->R);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

struct EmOpts {          // all-null for a plain pass
    double *A_out = nullptr, *Q_out = nullptr, *mu0_out = nullptr, *P0_out = nullptr;
    int* active = nullptr; int* iters = nullptr; double* ll_path = nullptr;
    int k = 0, max_iter = 1; double tol = 0.0;
};

// Balanced panel, even N, plain pass: may this call take the fast path (fastpath.hip)?
bool fast_eligible(const dfm_handle* h, int N, int r, unsigned flags) {
    if (h->force_general || (flags & (DFM_F_MAY_HAVE_MISSING | DFM_F_SINGULAR_Q))) return false;
    return collapse_dma_supported(pad_r(r), N) || collapse_wide_supported(pad_r(r), N);
}

bool g_odd_pad8 = true;                  // DFM_ODD_PAD8=0 (diagnostics build): odd N at states up to 8 wide stays on collapse_kernel
bool needs_odd_pad(bool fast, int N, int r, unsigned flags = 0, int B = 0) {
    if (fast) return false;
    // states up to 8 wide, panels with missing cells: collapse_miss_kernel's rows are moved 16 bytes at a time (even N).  With the
    // appended series the pass takes its table mode (+ recursion_chunk_kernel) instead of collapse_kernel + chunk_bridge_kernel:
    // the Stock-Watson window (N = 139) is such a panel.  (r <= 4 beyond 1536 replicates stays on the 4-wide lane-group kernels.)
    if (g_odd_pad8 && pad_r(r) <= 8 && (N & 1) && (flags & DFM_F_MAY_HAVE_MISSING) && !(flags & DFM_F_SINGULAR_Q) && collapse_miss_supported(8, N + 1) &&
        (pad_r(r) == 8 || (g_widen_small_r && B <= 1536)))
        return true;
    return pad_r(r) == 32 && (N & 1) && N > collapse_max_n(32) && collapse_wide2_supported(32, N + 1);
}

int check_em_n(dfm_handle* h, int N, int r, unsigned flags) {
    if (fast_eligible(h, N, r, flags) && !h->em_general) {
        if (N > 1024 && !mstep_mfma_supported(pad_r(r), N))
            return fail(h, DFM_E_DIMS, "N > 1024: the loadings M-step (one lane per series, 4 series per lane) does not cover this cross-section%s");
        return 0;
    }
    return check_general_n(h, N, r, true);
}

// ---- large batches of panels with missing cells (states 8 wide): sub-batches on two streams, two workspace slots ------------------
// The general path's workspace is 0.9 MB per replicate (table entries of every period): 7.4 GB at B = 8192, 59 GB at B = 65536, beside
// 0.8 MB of panel per replicate.  From 16 replicates per CU on, the batch runs as sub-batches of 8 per CU that alternate between the
// caller's stream and h->post, each stream with its own workspace slot: the workspace stops growing (3.7 GB), and the kernels of
// neighbouring sub-batches overlap.  What the overlap is worth, measured (B = 8192, C2 shape, 10 % missing; profiles/r06/README.md):
// nothing for the pass (3.80 ms as one batch; 4.13 / 4.01 / 3.79 / 3.84 ms with sub-batches of 512 / 1024 / 2048 / 4096 -- the collapse
// beside the recursion takes 0.43-0.57 ms per 1024 replicates instead of 0.25, the recursion 0.37-0.45 instead of 0.24: one wave of each
// on a SIMD compete for its VALU issue slots), 3 % for the EM iteration (7.86 -> 7.56-7.64 ms).  Replicates are independent: the results
// are bit for bit those of the one-batch call (tests/test_gpu_pipe.py).
// body(b0, bn): enqueue everything for replicates [b0, b0 + bn) on h->stream with h->ws as its workspace.
int pipe_sub(const dfm_handle* h) { return 8 * h->num_cu; }
bool pipe_eligible(const dfm_handle* h, int B, int N, int r, unsigned flags) {
    if (h->no_pipe || h->in_pipe || !h->post) return false;
    if (pad_r(r) != 8 || !(flags & DFM_F_MAY_HAVE_MISSING) || (flags & DFM_F_SINGULAR_Q)) return false;
    return B >= 2 * pipe_sub(h) && collapse_miss_supported(8, N) && !h->collapse_miss_old && !h->no_chunk;
}
template <class Body>
int pipe_run(dfm_handle* h, int B, size_t slot_bytes, Body body) {
    const int Bs = pipe_sub(h), S = (B + Bs - 1) / Bs;
    const size_t slot = (slot_bytes + 255) & ~(size_t)255;
    if (int rc = ensure_ws(h, 2 * slot + (size_t)B * sizeof(int))) return rc;
    char* base = static_cast<char*>(h->ws);
    int* agg = reinterpret_cast<int*>(base + 2 * slot);       // chunk_fail of every replicate (dfm_chunk_fallbacks)
    hipStream_t main = h->stream;
    HIP_TRY(h, hipEventRecord(h->ev_fork, main));
    HIP_TRY(h, hipStreamWaitEvent(h->post, h->ev_fork, 0));
    int rc = 0;
    bool have_fail = true;
    h->in_pipe = true;
    for (int s_ = 0; s_ < S && rc == 0; ++s_) {
        const int b0 = s_ * Bs, bn = (B - b0 < Bs) ? B - b0 : Bs;
        h->ws = base + (size_t)(s_ & 1) * slot;
        h->stream = (s_ & 1) ? h->post : main;
        rc = body(b0, bn);
        if (rc == 0) {
            if (h->ck_fail_dev && h->ck_fail_n == bn) {
                const hipError_t e = hipMemcpyAsync(agg + b0, h->ck_fail_dev, (size_t)bn * sizeof(int), hipMemcpyDeviceToDevice, h->stream);
                if (e != hipSuccess) rc = hip_fail(h, e, "hipMemcpyAsync(chunk_fail)");
            } else have_fail = false;
        }
    }
    h->in_pipe = false;
    h->ws = base; h->stream = main;
    (void)hipEventRecord(h->ev_post, h->post);
    (void)hipStreamWaitEvent(main, h->ev_post, 0);              // join (also after a failure: the slots are in use until then)
    h->ck_fail_dev = (rc == 0 && have_fail) ? agg : nullptr;
    h->ck_fail_n = (rc == 0 && have_fail) ? B : 0;
    return rc;
}

// gram + cov on the side stream, beside the streaming collapse on the main stream; the batch is cut
// into sub-batches so that the (latency-bound) meanscan of sub-batch s runs on a third stream beside the
// (bandwidth-bound) collapse of sub-batch s+1.
int enqueue_pass_fast(dfm_handle* h, const Plan& p, int B, int T, int N, int out_r, const double* panel,
                      const PaddedParams& pp, const double* Rv, double* f_smooth, double* P_smooth,
                      double* loglik, const EmOpts* em = nullptr) {
    CollapseArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.B = B; ca.T = T; ca.N = N;
    ca.panel = panel; ca.Lam = pp.Lam; ca.Rv = Rv;
    ca.bcol = at<double>(h, p.bcol); ca.scol = at<double>(h, p.scol); ca.ssum = at<double>(h, p.f_ssum);
    ca.Cfull = at<double>(h, p.Cfull); ca.ldfull = at<double>(h, p.ldfull); ca.status = h->status_dev;
    FastArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.B = B; fa.T = T; fa.N = N; fa.r = out_r; fa.L = fast_chunk_len(p.Rp, T);
    fa.rstate = p.r;
    fa.A = pp.A; fa.Q = pp.Q; fa.mu0 = pp.mu0; fa.P0 = pp.P0;
    fa.Cfull = ca.Cfull; fa.ldfull = ca.ldfull;
    fa.tab = at<double>(h, p.f_tab); fa.E = at<int>(h, p.f_E); fa.stead = at<double>(h, p.f_stead);
    fa.xi0 = at<double>(h, p.f_xi0); fa.PT = at<double>(h, p.f_PT); fa.llc = at<double>(h, p.f_llc);
    fa.fill = at<int>(h, p.f_fill); fa.PsInf = at<double>(h, p.f_PsInf);
    fa.bcol = ca.bcol; fa.ssum = ca.ssum; fa.wtab = at<double>(h, p.wtab); fa.wrep = wtab_rows(true, p.Rp, T) * (size_t)p.Rp;
    // collapse kernel of the balanced path: contraction on the matrix pipe where the shape allows it
    // (collapse_mfma.hip), else the VALU kernel (collapse_dma.hip); DFM_COLLAPSE_VARIANT < 200 forces the latter
    // shapes outside the register tilings (or DFM_COLLAPSE_VARIANT=198): the wide kernel
    // Rp = 16 | 32 with an even N: the streaming collapse of collapse_wide2.hip (Rp = 16 used to take the VALU kernel: 1.4-2.1 TB/s)
    const bool prefer_wide2 = !h->wide_old && h->collapse_variant == 0 && p.Wwide != (size_t)-1 && collapse_wide2_supported(p.Rp, N);
    const bool use_wide = prefer_wide2 || !collapse_dma_supported(p.Rp, N) || h->collapse_variant == 198;
    const bool use_mfma = !use_wide && collapse_mfma_supported(p.Rp, N) && (h->collapse_variant == 0 || h->collapse_variant >= 200);
    const int cvariant = use_mfma ? (h->collapse_variant >= 200 ? h->collapse_variant : 200)
                                  : (h->collapse_variant == 199 ? 0 : h->collapse_variant);   // 199: the VALU kernel's default
    const bool use_wide2 = use_wide && !h->wide_old && p.Wwide != (size_t)-1 && collapse_wide2_supported(p.Rp, N);
    if (use_wide) {   // sum_t s_t arrives as partials per tile of the collapse kernel that will run
        fa.scol = ca.scol;
        fa.ntile = !use_wide2 ? collapse_wide_tiles(T) : collapse_wide2_tiles(T);
    }
    // Rp = 32 on the streaming collapse: b_t rows 8 ceil(r / 8) doubles apart instead of 32 -- the padding components are exact zeros
    // that the mean scan (its only reader here) substitutes; at BASELINE config 4 (r = 20) a quarter of the 0.39 GB of b_t traffic
    if (use_wide2 && p.Rp == 32) ca.bst = fa.bst = 8 * ((p.r + 7) / 8);
    double* Wwide = use_wide2 ? at<double>(h, p.Wwide) : nullptr;
    // Gram matrix (+ W for the Rp = 32 collapse) and the streaming collapse of this shape, on stream `st`
    auto run_gram = [&](hipStream_t st) -> hipError_t {
        if (use_wide2) return launch_wide_prep(ca, Wwide, p.Rp, st, p.r);
        return gram_supported(p.Rp, N) ? launch_gram(p.Rp, ca, st) : launch_gram_wide(p.Rp, ca, st);
    };
    auto run_collapse = [&](const CollapseArgs& c, hipStream_t st) -> hipError_t {
        if (use_wide2) return launch_collapse_wide2(c, Wwide, p.Rp, p.r, h->num_cu, st);
        return use_wide ? launch_collapse_wide(p.Rp, c, st) : launch_collapse_dma(p.Rp, c, st, cvariant);
    };
    // MFMA collapse: as many period segments per replicate as the chip has resident wave slots for this batch
    // (3 workgroups x 4 waves on each CU), so that the launch is one balanced round
    int wpr = 4;
    if (use_mfma) {
        wpr = h->collapse_wpr > 0 ? h->collapse_wpr : (h->num_cu * 12) / B;
        if (wpr < 1) wpr = 1;
        if (wpr > 8) wpr = 8;
        while (wpr > 1 && T / wpr < 8) --wpr;     // keep segments a few row blocks long
    }
    ca.wpr = wpr;
    fa.nseg = use_mfma ? wpr : 4;
    const bool fuse_gram = cov_fuses_gram(p.Rp, N) && h->fuse_gram;
    if (fuse_gram) { fa.Lam = pp.Lam; fa.Rv = Rv; }
    fa.f_smooth = f_smooth; fa.P_smooth = P_smooth; fa.loglik = loglik;
    fa.abl = h->scan_abl;
    if (em) {   // EM: covariance sums from cov_kernel, E[f_0 | X] from meanscan (workspace slots of S10 / S00 reused)
        fa.SP11 = at<double>(h, p.S10); fa.SU = at<double>(h, p.S00); fa.P0s = at<double>(h, p.P0s);
        fa.f0s = at<double>(h, p.f0s);
    }
    // after the scan: transition M-step + bookkeeping from the sufficient statistics
    auto em_update = [&]() -> int {
        if (!em) return 0;
        EmUpdArgs ua;
        memset(&ua, 0, sizeof(ua));
        ua.B = B; ua.T = T; ua.fsm = f_smooth; ua.f0s = fa.f0s; ua.SP11 = fa.SP11; ua.SU = fa.SU; ua.P0s = fa.P0s;
        ua.PT = fa.PT; ua.loglik = loglik; ua.S11 = at<double>(h, p.S11); ua.S11inv = at<double>(h, p.Sxf);
        ua.A_out = em->A_out; ua.Q_out = em->Q_out; ua.mu0_out = em->mu0_out; ua.P0_out = em->P0_out;
        ua.active = em->active; ua.iters = em->iters; ua.ll_path = em->ll_path; ua.k = em->k; ua.max_iter = em->max_iter;
        ua.tol = em->tol;
        if (h->defer_em) { h->deferred_em = ua; h->have_deferred_em = true; return 0; }
        { ProfScope ps(h, K_EM_UPDATE); HIP_TRY(h, launch_em_update(p.Rp, ua, h->stream)); }
        return 0;
    };
    if (h->no_side) {   // diagnostics: everything in order on the main stream
        if (!fuse_gram) { ProfScope ps(h, K_GRAM); HIP_TRY(h, run_gram(h->stream)); }
        { ProfScope ps(h, K_COV); HIP_TRY(h, launch_cov(p.Rp, fa, h->stream)); }
        { ProfScope ps(h, use_wide ? K_COLLAPSE_WIDE : use_mfma ? K_COLLAPSE_MFMA : K_COLLAPSE_DMA); HIP_TRY(h, run_collapse(ca, h->stream)); }
        { ProfScope ps(h, K_MEANSCAN); HIP_TRY(h, launch_meanscan(p.Rp, fa, h->stream)); }
        return em_update();
    }
    // Measured on MI355X (profiles/r01): every cross-stream event edge costs 7-25 us, more than the
    // overlap buys at B = 1024, so the default is one sub-batch; DFM_SUBBATCH keeps the knob for big batches.
    int S = h->subbatch > 0 ? h->subbatch : 1;
    if (S > B) S = B;
    if (use_wide2) S = 1;                                     // (its workspace -- W, tile queues -- is laid out for the whole batch)
    if (S == 1 && h->pass_fused && use_mfma && pass_fused_supported(p.Rp, T, N) && h->collapse_variant == 0) {
        // ONE launch: persistent workgroups, b_t / w_t and the covariance tables never leave the chip (pass_fused.hip)
        fa.Lam = pp.Lam; fa.Rv = Rv;
        if (h->scan_abl & 256) {                                  // phase stamps of every replicate -> scol; readable through the
            ca.scol = at<double>(h, p.scol);                      // workspace dump below (diagnostics)
            if (const char* f = diag_env("DFM_PF_PROF_FILE")) h->prof_file = f;
        }
        { ProfScope ps(h, K_PASS_FUSED); HIP_TRY(h, launch_pass_fused(ca, fa, h->pass_nsw, h->pass_ncov, h->num_cu, h->stream)); }
        if ((h->scan_abl & 256) && !h->prof_file.empty()) {       // diagnostics: dump the stamps of this pass (synchronises)
            std::vector<double> st((size_t)B * T);
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            HIP_TRY(h, hipMemcpy(st.data(), ca.scol, st.size() * sizeof(double), hipMemcpyDeviceToHost));
            if (FILE* fp = fopen(h->prof_file.c_str(), "w")) {
                for (int bb = 0; bb < B; ++bb) {
                    fprintf(fp, "%d", bb);
                    for (int k = 0; k < 56; ++k) fprintf(fp, " %.0f", st[(size_t)bb * T + k]);
                    fprintf(fp, "\n");
                }
                fclose(fp);
            }
        }
        return em_update();
    }
    if (S == 1 && use_mfma && !fuse_gram && !h->no_fuse_cov && collapse_mfma_fuses_cov(p.Rp, N)) {
        // ONE stream, two launches: [Gram + covariance workgroups + P_smooth fill | streaming collapse] -> scan.
        // The covariance waves sit at the front of the collapse grid (resident first, no cross-stream events).
        if (h->fused_gram) { fa.Lam = pp.Lam; fa.Rv = Rv; }   // the covariance workgroups compute their Gram matrices themselves
        else { ProfScope ps(h, K_GRAM); HIP_TRY(h, run_gram(h->stream)); }
        ca.fuse_cov = &fa;
        { ProfScope ps(h, K_COLLAPSE_MFMA); HIP_TRY(h, launch_collapse_dma(p.Rp, ca, h->stream, cvariant)); }
        ca.fuse_cov = nullptr;
        if (P_smooth) fa.abl |= 1;
        { ProfScope ps(h, K_MEANSCAN); HIP_TRY(h, launch_meanscan(p.Rp, fa, h->stream)); }
        return em_update();
    }
    // Wide states (collapse_wide2), DFM_WIDE_SUB = n > 1 (diagnostics; default off): the batch in n sub-batches, each a complete
    // small batch of its own (own W workspace slice, own tile queues), the mean scan of sub-batch s beside the collapse of sub-batch
    // s + 1.  MEASURED SLOWER at config 4 (B = 256): 1.39 ms -> 1.56 (n = 2) -> 1.81 (n = 4).  The scan is a latency chain per
    // replicate -- 0.41 ms for 256 replicates, 0.46 ms for 128 -- so a sub-batch's scan hides nothing and the last one still runs alone.
    static const int wide_sub = [] { const char* v = diag_env("DFM_WIDE_SUB"); return v ? atoi(v) : 1; }();
    const int Sw = (use_wide2 && wide_sub > 1 && wide_sub <= 8 && B >= 32 * wide_sub) ? wide_sub : 1;
    if (S == 1 && Sw > 1) {
        while ((int)h->ev_sub.size() < Sw + 1) {
            hipEvent_t e;
            HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            h->ev_sub.push_back(e);
        }
        const int bmax = (B + Sw - 1) / Sw;
        const size_t slice = collapse_wide2_ws_bytes(bmax, N, p.Rp);
        auto sub_lo = [&](int s_) { return (int)((long long)B * s_ / Sw); };
        auto sub_args = [&](int s_) {
            const int b0 = sub_lo(s_), b1 = sub_lo(s_ + 1);
            CollapseArgs c = ca;
            c.B = b1 - b0;
            c.panel = ca.panel + (size_t)b0 * T * N; c.Lam = ca.Lam + (size_t)b0 * N * p.Rp; c.Rv = ca.Rv + (size_t)b0 * N;
            c.bcol = ca.bcol + (size_t)b0 * T * (ca.bst > 0 ? ca.bst : p.Rp); c.scol = ca.scol + (size_t)b0 * T; c.ssum = ca.ssum + (size_t)b0 * kSsumSlots;
            c.Cfull = ca.Cfull + (size_t)b0 * p.Rp * p.Rp; c.ldfull = ca.ldfull + b0;
            return c;
        };
        auto sub_ws = [&](int s_) { return reinterpret_cast<double*>(reinterpret_cast<char*>(Wwide) + (size_t)s_ * slice); };
        for (int s_ = 0; s_ < Sw; ++s_) { ProfScope ps(h, K_GRAM); HIP_TRY(h, launch_wide_prep(sub_args(s_), sub_ws(s_), p.Rp, h->stream, p.r)); }
        HIP_TRY(h, hipEventRecord(h->ev_fork, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
        { ProfScope ps(h, K_COV); HIP_TRY(h, launch_cov(p.Rp, fa, h->stream)); }      // resident before the collapse fills the CUs
        HIP_TRY(h, hipEventRecord(h->ev_sub[Sw], h->stream));                         // covariance tables done
        for (int s_ = 0; s_ < Sw; ++s_) {
            { ProfScope ps(h, K_COLLAPSE_WIDE, h->side); HIP_TRY(h, launch_collapse_wide2(sub_args(s_), sub_ws(s_), p.Rp, p.r, h->num_cu, h->side)); }
            HIP_TRY(h, hipEventRecord(h->ev_sub[s_], h->side));
        }
        const bool fill = !h->no_pfill && P_smooth;
        if (fill) {                       // 0.86 GB of stores at config 4: beside the scan of the last sub-batch, not beside the collapse
            HIP_TRY(h, hipStreamWaitEvent(h->post, h->ev_sub[Sw], 0));
            HIP_TRY(h, hipStreamWaitEvent(h->post, h->ev_sub[Sw - 1], 0));
            { ProfScope ps(h, K_PFILL, h->post); HIP_TRY(h, launch_pfill(p.Rp, fa, h->post)); }
            HIP_TRY(h, hipEventRecord(h->ev_post, h->post));
            fa.abl |= 1;
        }
        for (int s_ = 0; s_ < Sw; ++s_) {
            HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_sub[s_], 0));
            FastArgs fs = fa;
            fs.b0 = sub_lo(s_); fs.B = sub_lo(s_ + 1) - sub_lo(s_);
            { ProfScope ps(h, K_MEANSCAN); HIP_TRY(h, launch_meanscan(p.Rp, fs, h->stream)); }
        }
        if (fill) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_post, 0));
        return em_update();
    }
    if (S == 1) {
        // The covariance kernel (128 waves, 224 VGPRs each) must be resident BEFORE the streaming collapse fills
        // every CU, or it waits for the collapse to drain (measured: 285 us instead of 90).  It therefore goes
        // first on the caller's stream, and the collapse is the forked work: its queue starts ~6 us later.
        if (!fuse_gram || use_wide2) { ProfScope ps(h, K_GRAM); HIP_TRY(h, run_gram(h->stream)); }   // 12 us, alone
        HIP_TRY(h, hipEventRecord(h->ev_fork, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
        { ProfScope ps(h, K_COV); HIP_TRY(h, (h->cov_wave && p.Rp == 8 && !fuse_gram) ? launch_cov_wave(fa, h->stream) : launch_cov(p.Rp, fa, h->stream)); }
        { ProfScope ps(h, use_wide ? K_COLLAPSE_WIDE : use_mfma ? K_COLLAPSE_MFMA : K_COLLAPSE_DMA, h->side); HIP_TRY(h, run_collapse(ca, h->side)); }
        HIP_TRY(h, hipEventRecord(h->ev_join, h->side));
        const bool fill = !h->no_pfill && P_smooth;
        // Rp = 32 (config 4): the fill is 0.86 GB of stores -- beside the collapse they cost it 0.4 ms of its 1.13; beside the
        // latency-bound scan they are free.  So: cov -> [event] ; collapse (side) -> [event] ; fill on the third stream
        // after both, scan on the caller's stream after the collapse, join at the end.
        // (A trickle of these stores from a few persistent workgroups UNDER the collapse was tried: the collapse lost what the scan
        // gained -- profiles/r04/ab_pfill_trickle_c4.txt.)
        const bool fill_late = fill && use_wide2;
        if (fill && !fill_late) {         // the data-independent rows of P_smooth, beside the collapse
            ProfScope ps(h, K_PFILL);
            HIP_TRY(h, launch_pfill(p.Rp, fa, h->stream));
            fa.abl |= 1;
        }
        if (fill_late) {
            if (h->ev_sub.empty()) {
                hipEvent_t e;
                HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                h->ev_sub.push_back(e);
            }
            HIP_TRY(h, hipEventRecord(h->ev_sub[0], h->stream));            // cov_kernel's outputs
            HIP_TRY(h, hipStreamWaitEvent(h->post, h->ev_sub[0], 0));
            HIP_TRY(h, hipStreamWaitEvent(h->post, h->ev_join, 0));        // ... and not before the collapse is done
            { ProfScope ps(h, K_PFILL, h->post); HIP_TRY(h, launch_pfill(p.Rp, fa, h->post)); }
            HIP_TRY(h, hipEventRecord(h->ev_post, h->post));
            fa.abl |= 1;
        }
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
        { ProfScope ps(h, K_MEANSCAN); HIP_TRY(h, launch_meanscan(p.Rp, fa, h->stream)); }
        if (fill_late) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_post, 0));
        return em_update();
    }
    while ((int)h->ev_sub.size() < S) {
        hipEvent_t e;
        HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->ev_sub.push_back(e);
    }
    // fork: side and post start after everything already enqueued on the main stream (parameters, memsets)
    HIP_TRY(h, hipEventRecord(h->ev_fork, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
    HIP_TRY(h, hipStreamWaitEvent(h->post, h->ev_fork, 0));
    if (!fuse_gram) { ProfScope ps(h, K_GRAM, h->side); HIP_TRY(h, run_gram(h->side)); }
    { ProfScope ps(h, K_COV, h->side); HIP_TRY(h, launch_cov(p.Rp, fa, h->side)); }
    HIP_TRY(h, hipEventRecord(h->ev_join, h->side));
    HIP_TRY(h, hipStreamWaitEvent(h->post, h->ev_join, 0));
    for (int s = 0; s < S; ++s) {
        const int b0 = (int)((long long)B * s / S), b1 = (int)((long long)B * (s + 1) / S);
        CollapseArgs cs = ca;
        cs.b0 = b0; cs.B = b1 - b0;
        { ProfScope ps(h, use_mfma ? K_COLLAPSE_MFMA : K_COLLAPSE_DMA); HIP_TRY(h, launch_collapse_dma(p.Rp, cs, h->stream, cvariant)); }
        HIP_TRY(h, hipEventRecord(h->ev_sub[s], h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->post, h->ev_sub[s], 0));
        FastArgs fs = fa;
        fs.b0 = b0; fs.B = b1 - b0;
        { ProfScope ps(h, K_MEANSCAN, h->post); HIP_TRY(h, launch_meanscan(p.Rp, fs, h->post)); }
    }
    HIP_TRY(h, hipEventRecord(h->ev_post, h->post));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_post, 0));   // join
    return em_update();
}

// Enqueue collapse + recursion for already-planned workspace.  out_r = factor dimension of the
// f_smooth / P_smooth layout (caller's r for a plain pass, Rp for EM-internal buffers).
int enqueue_pass(dfm_handle* h, const Plan& p, int B, int T, int N, int out_r, const double* panel,
                 const PaddedParams& pp, const double* Rv, double* f_smooth, double* P_smooth, double* loglik,
                 const EmOpts* em) {
    if (p.fast) return enqueue_pass_fast(h, p, B, T, N, out_r, panel, pp, Rv, f_smooth, P_smooth, loglik, em);
    CollapseArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.B = B; ca.T = T; ca.N = N;
    ca.panel = panel; ca.Lam = pp.Lam; ca.Rv = Rv;
    ca.bcol = at<double>(h, p.bcol); ca.scol = at<double>(h, p.scol); ca.nobs = at<int>(h, p.nobs);
    ca.ldrow = at<double>(h, p.ldrow); ca.Ct = at<double>(h, p.Ct); ca.Cfull = at<double>(h, p.Cfull);
    ca.ldfull = at<double>(h, p.ldfull); ca.status = h->status_dev;
    ca.obs_chunk = nullptr; ca.obs_table = nullptr; ca.obs_L = 0;
    ca.kreal = (p.kdim > 0 && p.Rc == 0) ? p.kdim : 0;          // (companion state observed on every block: the AR idiosyncratic model)
    RecursionArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.B = B; ra.T = T; ra.N = N; ra.r = out_r;
    ra.rstate = p.r; ra.cov = p.cov ? 1 : 0; ra.Rc = p.Rc; ra.rl = p.rl; ra.kdim = p.kdim; ra.kb = p.kb; ra.ka = p.ka; ra.qsing = p.qsing; ra.wave = h->no_rec_wave ? 0 : 1;
    ra.pair_bmax = h->pair_bmax >= 0 ? h->pair_bmax : 4 * h->num_cu;
    ra.A = pp.A; ra.Q = pp.Q; ra.mu0 = pp.mu0; ra.P0 = pp.P0;
    ra.bcol = ca.bcol; ra.scol = ca.scol; ra.nobs = ca.nobs; ra.ldrow = ca.ldrow; ra.Ct = ca.Ct;
    ra.Cfull = ca.Cfull; ra.ldfull = ca.ldfull;
    ra.ZJtab = at<double>(h, p.ZJ); ra.wtab = at<double>(h, p.wtab); ra.eidx = nullptr;
    ra.chunk_scr = h->no_chunk ? nullptr : at<double>(h, p.ck_scr); ra.chunk_W = h->chunk_w; ra.chunk_tol = h->chunk_tol; ra.chunk_obs = at<double>(h, p.ck_obs); ra.chunk_cst = at<double>(h, p.ck_cst); ra.chunk_term = at<double>(h, p.ck_term); ra.chunk_fail = at<int>(h, p.ck_fail);
    ra.chunk_skip = at<int>(h, p.ck_skip);
    ra.tile_scr = at<double>(h, p.tk_scr); ra.tile_scr_bytes = p.tk_bytes; ra.tile_nc = h->tile_nc; ra.tile_w = h->tile_w; ra.num_cu = h->num_cu;
    ra.f_smooth = f_smooth; ra.P_smooth = P_smooth; ra.loglik = loglik;
    ra.ncov = at<int>(h, p.ncov);
    if (em) {
        ra.S11 = at<double>(h, p.S11); ra.S10 = at<double>(h, p.S10); ra.S00 = at<double>(h, p.S00);
        ra.f0s = at<double>(h, p.f0s); ra.P0s = at<double>(h, p.P0s); ra.S11inv = at<double>(h, p.Sxf);
        ra.A_out = em->A_out; ra.Q_out = em->Q_out; ra.mu0_out = em->mu0_out; ra.P0_out = em->P0_out;
        ra.active = em->active; ra.iters = em->iters; ra.ll_path = em->ll_path;
        ra.k = em->k; ra.max_iter = em->max_iter; ra.tol = em->tol;
    }
    // C_t rows: the packed leading block when recursion_tile_kernel reads them (it executes ceil(r / 4) block pivots of the 32-wide
    // state: the rest is padding whose entries equal Cfull's), the full Rp (Rp + 1) / 2 layout for the other recursion kernels
    ca.ct_r = 0;
    if (ra.wave && p.Wwide != (size_t)-1 && N > collapse_max_n(p.Rc ? p.Rc : p.Rp) && recursion_tile_supported(p.Rp, ra) &&
        ct_miss_wide_compact_ok(N, 4 * ((p.r + 3) / 4))) ca.ct_r = 4 * ((p.r + 3) / 4);
    ra.ct_r = ca.ct_r;
    {
        const int Rcol = p.Rc ? p.Rc : p.Rp;
        // the time-chunked recursion reads one table row per period (C_t, b_t, s_t, n_t log 2 pi + log det R_t): at Rp = 8 with loadings as
        // wide as the state collapse_miss_kernel writes it directly
        // (r <= 4 widened to the 8-wide state, Plan::Rc = 2 / 4: the kernel pads the loadings with zero columns in its LDS tables --
        // CollapseArgs::lam_w -- and the table is the 8-wide one the chunks read anyway; collapse_kernel<4> + chunk_bridge_kernel took
        // 0.25 ms per 1024 replicates of the Stock-Watson window where this takes 0.1)
        const bool narrow_tab = p.Rc > 0 && p.Rc < 8 && p.kdim == 0 && p.rl == p.Rc && !h->narrow_tab_off;
        const bool table = ra.wave && !h->collapse_miss_old && (p.Rc == 0 || narrow_tab) && recursion_chunk_supported(p.Rp, ra);
        // companion states (VAR(p) factor dynamics) with loadings up to 4 wide: the same table, then EVERY replicate's rows written
        // back in the layout the sequential kernels read (chunk_unbridge_kernel<Rc>: b_t, s_t, C_t of every period, nobs = 0) --
        // collapse_kernel<4> took 0.235 ms per 1024 replicates of the Stock-Watson window where these two take 0.12
        const bool comp_table = p.kdim > 0 && p.kb == 0 && p.Rc > 0 && p.Rc < 8 && p.ck_rows != (size_t)-1 && p.ck_obs != (size_t)-1 &&
                                collapse_miss_supported(8, N);
        if (comp_table) {
            ca.lam_w = p.Rc;
            ca.obs_chunk = at<double>(h, p.ck_rows);
            ca.obs_table = at<double>(h, p.ck_obs); ca.obs_L = recursion_chunk_len(T);
            { ProfScope ps(h, K_COLLAPSE); HIP_TRY(h, launch_collapse_miss(ca, h->num_cu, h->stream)); }
            RecursionArgs ua = ra;
            ua.chunk_obs = at<double>(h, p.ck_obs);
            ua.chunk_fail = nullptr;                              // (no flags: every replicate)
            HIP_TRY(h, launch_chunk_unbridge(ua, h->stream));
        } else
        if (table && p.ck_rows != (size_t)-1 && collapse_miss_supported(8, N) && (p.Rc == 0 ? Rcol == 8 : true)) {
            ca.lam_w = p.Rc;
            ca.obs_chunk = at<double>(h, p.ck_rows);
            ca.obs_table = ra.chunk_obs; ca.obs_L = recursion_chunk_len(T);
            ra.chunk_obs_ready = 1;
            ProfScope ps(h, K_COLLAPSE); HIP_TRY(h, launch_collapse_miss(ca, h->num_cu, h->stream));
        } else
        if (!h->collapse_miss_old && collapse_miss_supported(Rcol, N)) { ProfScope ps(h, K_COLLAPSE); HIP_TRY(h, launch_collapse_miss(ca, h->num_cu, h->stream)); }
        else if (p.Wwide != (size_t)-1 && N > collapse_max_n(Rcol)) {   // Rp = 32 beyond the register tiling (config 4 with missing cells)
            double* W = at<double>(h, p.Wwide);
            double* V = at<double>(h, p.Vwide);
            { ProfScope ps(h, K_GRAM); HIP_TRY(h, launch_wide_prep(ca, W, 32, h->stream, 0, V)); }
            { ProfScope ps(h, K_COLLAPSE_WIDE); HIP_TRY(h, launch_collapse_wide2(ca, W, 32, p.r, h->num_cu, h->stream)); }
            { ProfScope ps(h, K_COLLAPSE); HIP_TRY(h, launch_ct_miss_wide(ca, W, p.r, h->stream, V)); }
        } else { ProfScope ps(h, K_COLLAPSE); HIP_TRY(h, launch_collapse(Rcol, ca, h->stream)); }
    }
    h->ck_fail_dev = nullptr; h->ck_fail_n = 0;
    if (ra.wave && recursion_chunk_supported(p.Rp, ra)) { h->ck_fail_dev = ra.chunk_fail; h->ck_fail_n = B; }
    else if (ra.wave && recursion_tile_supported(p.Rp, ra) && recursion_tile_writes_fail(ra)) { h->ck_fail_dev = ra.chunk_fail; h->ck_fail_n = B; }
    { ProfScope ps(h, K_RECURSION); HIP_TRY(h, launch_recursion(p.Rp, ra, h->stream)); }
    return 0;
}

// dst[b][i][j] = src[b][i][j], i < rd, j < cd (un-padding of parameters / smoother outputs)
__global__ void copy_block_kernel(size_t nb, int rs, int cs, int rd, int cd, const double* src, double* dst) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = nb * rd * cd;
    if (tid >= n) return;
    const int j = tid % cd;
    const int i = (tid / cd) % rd;
    const size_t b = tid / ((size_t)cd * rd);
    dst[tid] = src[(b * rs + i) * cs + j];
}
int copy_block(dfm_handle* h, size_t nb, int rs, int cs, int rd, int cd, const double* src, double* dst) {
    const size_t n = nb * rd * cd;
    hipLaunchKernelGGL(copy_block_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, nb, rs, cs,
                       rd, cd, src, dst);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

// One EM iteration on PADDED, writable device parameters (Lam [B][N][Rp], A/Q/P0 [B][Rp][Rp], mu0 [B][Rp]).
int em_iteration(dfm_handle* h, const Plan& p, int B, int T, int N, const double* panel, double* LamP,
                 double* Rv, double* AP, double* QP, double* mu0P, double* P0P, double* fsm, double* Psm,
                 double* loglik, EmOpts eo) {
    const int Rp = p.Rc ? p.Rc : p.Rp;          // width of the loadings (narrower than the state for a companion model)
    PaddedParams pp{LamP, AP, QP, mu0P, P0P};
    eo.A_out = AP; eo.Q_out = QP; eo.mu0_out = mu0P; eo.P0_out = P0P;
    const bool ms_mfma = p.fast && p.ms_ws != (size_t)-1 && !h->no_mstep_mfma;
    h->defer_em = ms_mfma && p.Rp <= 8 && !h->no_defer_em;  // the transition M-step rides in the loadings step's launch
    h->have_deferred_em = false;
    const int rc_pass = enqueue_pass(h, p, B, T, N, Rp, panel, pp, Rv, fsm, Psm, loglik, &eo);
    h->defer_em = false;
    if (rc_pass) return rc_pass;
    MstepArgs ma;
    ma.B = B; ma.T = T; ma.N = N; ma.r = Rp;
    ma.panel = panel; ma.fsm = fsm; ma.Psm = Psm;
    ma.S11 = at<double>(h, p.S11); ma.S11inv = at<double>(h, p.Sxf);
    ma.Dmiss = mstep_needs_dmiss(Rp, N) ? at<double>(h, p.Dmiss) : nullptr;
    ma.active = eo.active; ma.Lam_out = LamP; ma.R_out = Rv; ma.lam_stride = Rp; ma.min_cells = 1;
    if (p.fast && p.ms_ws != (size_t)-1 && !h->no_mstep_mfma) {   // balanced panel: second panel read on the matrix pipe
        ProfScope ps(h, K_MSTEP_MFMA);
        HIP_TRY(h, launch_mstep_mfma(Rp, ma, p.ms_wpr, at<double>(h, p.ms_ws), h->stream, h->have_deferred_em ? &h->deferred_em : nullptr));
        h->have_deferred_em = false;
        return 0;
    }
    if (p.fast && p.mw_ws != (size_t)-1 && !h->no_mstep_mfma) {   // ... Rp = 32 (config 4): the same on the streaming machinery of its collapse
        ProfScope ps(h, K_MSTEP_MFMA);
        HIP_TRY(h, launch_mstep_wide(ma, at<double>(h, p.mw_ws), Rp, p.r, h->num_cu, h->stream));
        return 0;
    }
    {
        // panels with missing cells: both contractions of the loadings step as one product per replicate on the matrix pipe
        const int rl = p.Rc ? (p.rl ? p.rl : Rp) : (p.r < Rp ? p.r : Rp);     // the loadings' factor count
        if (p.mm_ws != (size_t)-1 && mstep_miss_supported(Rp, rl, N) && (g_mstep_miss_mode == 2 || mstep_needs_dmiss(Rp, N))) {
            ProfScope ps(h, K_MSTEP_STATS);
            HIP_TRY(h, launch_mstep_miss(ma, at<double>(h, p.mm_ws), Rp, rl, h->num_cu, h->stream));
            return 0;
        }
    }
    if (ma.Dmiss)
        HIP_TRY(h, hipMemsetAsync(ma.Dmiss, 0, (size_t)B * N * (Rp * (Rp + 1) / 2) * sizeof(double), h->stream));
    { ProfScope ps(h, K_MSTEP_STATS); HIP_TRY(h, launch_mstep_lam(Rp, ma, h->stream)); }
    return 0;
}

// Shared driver of dfm_em_step_batch_dev (max_iter = 1, no bookkeeping), dfm_em_batch_dev (iterations 0 .. max_iter-1,
// stops launching once no replicate of THIS batch is active) and dfm_em_iterate_batch_dev (iteration k_first only, the
// caller owns `active` and decides when to stop: the multi-GPU drivers).
int em_run(dfm_handle* h, int B, int T, int N, int r, const double* panel, double* Lam, double* R, double* A,
           double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path, int* iters,
           double* loglik_single, double* f_smooth, double* P_smooth, unsigned flags, int k_first = 0, int k_count = -1,
           int* active_ext = nullptr) {
    if (int rc = check_dims(h, B, T, N, r)) return rc;
    if (int rc = check_em_n(h, N, r, flags)) return rc;
    if (!panel || !Lam || !R || !A || !Q || !mu0 || !P0) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    if (max_iter < 1) return fail(h, DFM_E_DIMS, "max_iter must be >= 1%s");
    HIP_TRY(h, hipSetDevice(h->device));
    if (needs_odd_pad(fast_eligible(h, N, r, flags) && !h->em_general, N, r, flags, B)) {   // odd N beyond the tilings: one all-missing series appended
        OddPad o;
        if (int rc = odd_pad(h, B, T, N, r, panel, Lam, R, &o, k_first > 0)) return rc;
        // (the appended series is missing in EVERY period: the padded problem has missing cells whatever the caller said about N)
        if (int rc = em_run(h, B, T, N + 1, r, o.panel, o.Lam, o.R, A, Q, mu0, P0, max_iter, tol, loglik_path, iters, loglik_single,
                            f_smooth, P_smooth, flags | DFM_F_MAY_HAVE_MISSING, k_first, k_count, active_ext)) return rc;
        const size_t n_lam = (size_t)B * N * r, n_R = (size_t)B * N;
        hipLaunchKernelGGL(copy_series_rows_kernel, dim3((unsigned)((n_lam + 255) / 256)), dim3(256), 0, h->stream, (size_t)B, N + 1, N, r, 0.0, o.Lam, Lam);
        hipLaunchKernelGGL(copy_series_rows_kernel, dim3((unsigned)((n_R + 255) / 256)), dim3(256), 0, h->stream, (size_t)B, N + 1, N, 1, 0.0, o.R, R);
        HIP_TRY(h, hipGetLastError());
        return 0;
    }
    if (pipe_eligible(h, B, N, r, flags)) {     // sub-batches as EM runs of their own on two streams (pipe_run)
        const Plan ps = make_plan(pipe_sub(h), T, N, r, flags, true, false);
        const size_t rr = (size_t)r * r, np = (size_t)r * (r + 1) / 2;
        return pipe_run(h, B, ps.total, [&](int b0, int bn) -> int {
            return em_run(h, bn, T, N, r, panel + (size_t)b0 * T * N, Lam + (size_t)b0 * N * r, R + (size_t)b0 * N, A + b0 * rr, Q + b0 * rr,
                          mu0 + (size_t)b0 * r, P0 + b0 * rr, max_iter, tol, loglik_path ? loglik_path + (size_t)b0 * max_iter : nullptr,
                          iters ? iters + b0 : nullptr, loglik_single ? loglik_single + b0 : nullptr,
                          f_smooth ? f_smooth + (size_t)b0 * T * r : nullptr, P_smooth ? P_smooth + (size_t)b0 * T * np : nullptr, flags,
                          k_first, k_count, active_ext ? active_ext + b0 : nullptr);
        });
    }
    // balanced panels: E-step on the fast path (collapse on the matrix pipe, time-parallel scan), transition
    // M-step by em_update_kernel; panels with missing cells: recursion_kernel does both
    const Plan p = make_plan(B, T, N, r, flags, true, fast_eligible(h, N, r, flags) && !h->em_general);
    if (int rc = ensure_ws(h, p.total)) return rc;
    const int Rp = p.Rp, Rl = p.Rc ? p.Rc : p.Rp;            // state width, loadings width
    const size_t np = (size_t)r * (r + 1) / 2, npp = (size_t)Rl * (Rl + 1) / 2;
    const bool padded = (r != Rp);
    double *LamP = Lam, *AP = A, *QP = Q, *mu0P = mu0, *P0P = P0;
    if (padded) {
        PaddedParams pp;
        if (int rc = pad_params(h, p, B, N, r, Lam, A, Q, mu0, P0, &pp)) return rc;
        LamP = at<double>(h, p.LamP); AP = at<double>(h, p.AP); QP = at<double>(h, p.QP);
        mu0P = at<double>(h, p.mu0P); P0P = at<double>(h, p.P0P);
    }
    // smoother outputs of the E-steps: straight into the caller's buffers when layouts coincide
    double* fsm = (!padded && f_smooth) ? f_smooth : at<double>(h, p.fsm);
    double* Psm = (!padded && P_smooth) ? P_smooth : at<double>(h, p.Psm);
    // balanced panels with the loadings step on the matrix pipe: nothing in the EM reads the per-period smoothed covariances
    // (the M-step works from their sums, cov_kernel's SP11 / SU) -- unless the caller asked for them, the E-steps do not
    // write them (0.15 GB of stores per iteration at config 2, 0.86 GB at config 4)
    if (!P_smooth && p.fast && !h->no_mstep_mfma && (p.ms_ws != (size_t)-1 || p.mw_ws != (size_t)-1)) Psm = nullptr;
    double* llbuf = loglik_single ? loglik_single : at<double>(h, p.llbuf);
    const bool book = loglik_path != nullptr;
    int* active = book ? (active_ext ? active_ext : at<int>(h, p.active)) : nullptr;
    if (book && k_first == 0) {
        HIP_TRY(h, hipMemsetAsync(loglik_path, 0xFF, (size_t)B * max_iter * sizeof(double), h->stream));  // NaN
        HIP_TRY(h, hipMemsetAsync(iters, 0, (size_t)B * sizeof(int), h->stream));
    }
    const int k_end = k_count < 0 ? max_iter : (k_first + k_count < max_iter ? k_first + k_count : max_iter);
    std::vector<int> act_host;
    for (int k = k_first; k < k_end; ++k) {
        EmOpts eo;
        eo.active = active; eo.iters = iters; eo.ll_path = loglik_path; eo.k = k; eo.max_iter = max_iter; eo.tol = tol;
        if (int rc = em_iteration(h, p, B, T, N, panel, LamP, R, AP, QP, mu0P, P0P, fsm, Psm, llbuf, eo)) return rc;
        if (book && !active_ext && tol > 0.0 && k + 1 < k_end) {   // stop launching once every replicate has converged
            act_host.resize(B);
            HIP_TRY(h, hipMemcpyAsync(act_host.data(), active, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            bool any = false;
            for (int b = 0; b < B; ++b) any = any || act_host[b] != 0;
            if (!any) break;
        }
    }
    if (padded) {
        if (int rc = copy_block(h, (size_t)B * N, 1, Rl, 1, r, LamP, Lam)) return rc;
        if (int rc = copy_block(h, B, Rp, Rp, r, r, AP, A)) return rc;
        if (int rc = copy_block(h, B, Rp, Rp, r, r, QP, Q)) return rc;
        if (int rc = copy_block(h, B, Rp, Rp, r, r, P0P, P0)) return rc;
        if (int rc = copy_block(h, B, 1, Rp, 1, r, mu0P, mu0)) return rc;
        if (f_smooth) if (int rc = copy_block(h, (size_t)B * T, 1, Rl, 1, r, fsm, f_smooth)) return rc;
        if (P_smooth) if (int rc = copy_block(h, (size_t)B * T, 1, (int)npp, 1, (int)np, Psm, P_smooth)) return rc;
    } else {
        if (f_smooth && fsm != f_smooth)
            HIP_TRY(h, hipMemcpyAsync(f_smooth, fsm, (size_t)B * T * r * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        if (P_smooth && Psm != P_smooth)
            HIP_TRY(h, hipMemcpyAsync(P_smooth, Psm, (size_t)B * T * np * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    return 0;
}

// ---- VAR(p) factor dynamics in companion form ---------------------------------------------
// z_t = (f_t, .., f_{t-p+1}), k = r p <= 32;  M = [A_1 .. A_p; I 0],  Q_z = [Q 0; 0 0]  (dfm_functions.ipynb:477-492);
// padded to Rk x Rk with the usual identity / zero padding.  Loadings padded to Rc = pad_r(r).
__global__ void companion_pad_kernel(int B, int N, int r, int k, int ka, int Rc, int Rk, const double* Lam, const double* Avar,
                                     const double* Q, const double* mu0, const double* P0, double* LamP, double* AP,
                                     double* QP, double* mu0P, double* P0P) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nl = (size_t)B * N * Rc, nm = (size_t)B * Rk * Rk, nv = (size_t)B * Rk;
    if (Lam && tid < nl) {
        const int c = tid % Rc;
        const size_t bn = tid / Rc;
        LamP[tid] = c < r ? Lam[bn * r + c] : 0.0;
    }
    if (tid < nm) {
        const int j = tid % Rk, i = (tid / Rk) % Rk;
        const size_t b = tid / ((size_t)Rk * Rk);
        double a = 0.0, q = 0.0, p0 = 0.0;
        if (i < r && j < ka) a = Avar[(b * r + i) * ka + j];        // [A_1 .. A_p], ka = r p <= k
        else if (i >= r && i < k && j == i - r) a = 1.0;
        if (i < r && j < r) q = Q[(b * r + i) * r + j];
        else if (i >= k && i == j) q = 1.0;
        if (i < k && j < k) p0 = P0[(b * k + i) * k + j];
        else if (i >= k && i == j) p0 = 1.0;
        AP[tid] = a; QP[tid] = q; P0P[tid] = p0;
    }
    if (tid < nv) {
        const int i = tid % Rk;
        const size_t b = tid / Rk;
        mu0P[tid] = i < k ? mu0[b * k + i] : 0.0;
    }
}

int varp_run(dfm_handle* h, int B, int T, int N, int r, int nlag, const double* panel, double* Lam, double* R,
             double* Avar, double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path, int* iters,
             double* loglik_single, double* f_smooth, double* P_smooth, unsigned flags, bool em) {
    if (!h) return DFM_E_NULL;
    if (nlag < 1) return fail(h, DFM_E_DIMS, "number of factor lags must be >= 1%s");
    if (int rc = check_dims(h, B, T, N, r)) return rc;
    const int k = r * nlag;
    if (k > DFM_MAX_R) return fail(h, DFM_E_R_UNSUPPORTED, "r * p > DFM_MAX_R (32)%s");
    if (int rc = check_general_n(h, N, r)) return rc;
    if (!panel || !Lam || !R || !Avar || !Q || !mu0 || !P0) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    if (max_iter < 1) return fail(h, DFM_E_DIMS, "max_iter must be >= 1%s");
    HIP_TRY(h, hipSetDevice(h->device));
    // panels with missing cells, loadings up to 4 wide: the collapse on collapse_miss_kernel's table mode (comp_table below); odd N
    // gets one all-missing series appended for it, as in em_run (the Stock-Watson window has 139 series)
    const bool comp_tab_ok = g_odd_pad8 && !h->narrow_tab_off && !h->collapse_miss_old && pad_r(r) < 8 && (flags & DFM_F_MAY_HAVE_MISSING);
    if (comp_tab_ok && (N & 1) && collapse_miss_supported(8, N + 1)) {
        OddPad o;
        if (int rc = odd_pad(h, B, T, N, r, panel, Lam, R, &o)) return rc;
        if (int rc = varp_run(h, B, T, N + 1, r, nlag, o.panel, o.Lam, o.R, Avar, Q, mu0, P0, max_iter, tol, loglik_path, iters, loglik_single,
                              f_smooth, P_smooth, flags, em)) return rc;
        if (em) {
            const size_t n_lam = (size_t)B * N * r, n_R = (size_t)B * N;
            hipLaunchKernelGGL(copy_series_rows_kernel, dim3((unsigned)((n_lam + 255) / 256)), dim3(256), 0, h->stream, (size_t)B, N + 1, N, r, 0.0, o.Lam, Lam);
            hipLaunchKernelGGL(copy_series_rows_kernel, dim3((unsigned)((n_R + 255) / 256)), dim3(256), 0, h->stream, (size_t)B, N + 1, N, 1, 0.0, o.R, R);
            HIP_TRY(h, hipGetLastError());
        }
        return 0;
    }
    Plan p = make_plan(B, T, N, k, flags | DFM_F_SINGULAR_Q, em, false);
    p.Rc = pad_r(r); p.rl = r; p.kdim = k; p.qsing = (flags & DFM_F_SINGULAR_Q) ? 1 : 0;
    if (comp_tab_ok && collapse_miss_supported(8, N) && (size_t)B * recursion_chunk_len(T) <= 0x7fffffffu) {
        size_t off = p.total;                                 // rows + masks, the chunk-major table (comp_table)
        p.ck_rows = take(off, recursion_chunk_rows_bytes(B, T));
        p.ck_obs = take(off, recursion_chunk_obs_bytes(B, T));
        p.total = off;
    }
    if (int rc = ensure_ws(h, p.total)) return rc;
    const int Rk = p.Rp, Rc = p.Rc;
    double *LamP = at<double>(h, p.LamP), *AP = at<double>(h, p.AP), *QP = at<double>(h, p.QP),
           *mu0P = at<double>(h, p.mu0P), *P0P = at<double>(h, p.P0P);
    {
        const size_t n = (size_t)B * N * Rc > (size_t)B * Rk * Rk ? (size_t)B * N * Rc : (size_t)B * Rk * Rk;
        hipLaunchKernelGGL(companion_pad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, B, N, r, k, k,
                           Rc, Rk, Lam, Avar, Q, mu0, P0, LamP, AP, QP, mu0P, P0P);
        HIP_TRY(h, hipGetLastError());
    }
    PaddedParams pp{LamP, AP, QP, mu0P, P0P};
    if (!em)   // plain pass: smoothed moments of f_t = z_t[:r] straight into the caller's layout
        return enqueue_pass(h, p, B, T, N, r, panel, pp, R, f_smooth, P_smooth, loglik_single, nullptr);
    const size_t np = (size_t)r * (r + 1) / 2, npc = (size_t)Rc * (Rc + 1) / 2;
    double* fsm = at<double>(h, p.fsm);
    double* Psm = at<double>(h, p.Psm);
    double* llbuf = loglik_single ? loglik_single : at<double>(h, p.llbuf);
    const bool book = loglik_path != nullptr;
    int* active = book ? at<int>(h, p.active) : nullptr;
    if (book) {
        HIP_TRY(h, hipMemsetAsync(loglik_path, 0xFF, (size_t)B * max_iter * sizeof(double), h->stream));  // NaN
        HIP_TRY(h, hipMemsetAsync(iters, 0, (size_t)B * sizeof(int), h->stream));
    }
    std::vector<int> act_host;
    for (int it = 0; it < max_iter; ++it) {
        EmOpts eo;
        eo.active = active; eo.iters = iters; eo.ll_path = loglik_path; eo.k = it; eo.max_iter = max_iter; eo.tol = tol;
        if (int rc = em_iteration(h, p, B, T, N, panel, LamP, R, AP, QP, mu0P, P0P, fsm, Psm, llbuf, eo)) return rc;
        if (book && tol > 0.0 && it + 1 < max_iter) {
            act_host.resize(B);
            HIP_TRY(h, hipMemcpyAsync(act_host.data(), active, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            bool any = false;
            for (int b = 0; b < B; ++b) any = any || act_host[b] != 0;
            if (!any) break;
        }
    }
    if (int rc = copy_block(h, (size_t)B * N, 1, Rc, 1, r, LamP, Lam)) return rc;
    if (int rc = copy_block(h, B, Rk, Rk, r, k, AP, Avar)) return rc;
    if (int rc = copy_block(h, B, Rk, Rk, r, r, QP, Q)) return rc;
    if (int rc = copy_block(h, B, Rk, Rk, k, k, P0P, P0)) return rc;
    if (int rc = copy_block(h, B, 1, Rk, 1, k, mu0P, mu0)) return rc;
    if (f_smooth) if (int rc = copy_block(h, (size_t)B * T, 1, Rc, 1, r, fsm, f_smooth)) return rc;
    if (P_smooth) if (int rc = copy_block(h, (size_t)B * T, 1, (int)npc, 1, (int)np, Psm, P_smooth)) return rc;
    return 0;
}

// ---- AR idiosyncratic terms by quasi-differencing ----------------------------------------------
// x~_it = x_it - sum_l rho_il x_i,t-l (NaN when x_it or a lag is NaN);  loadings [lam_i, -rho_i1 lam_i, .., -rho_iq lam_i, 0..]
__global__ void quasi_diff_kernel(int B, int T, int N, int q, const double* x, const double* rho, double* out) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)B * (T - q) * N;
    if (tid >= n) return;
    const int i = tid % N;
    const int t = (tid / N) % (T - q);
    const size_t b = tid / ((size_t)N * (T - q));
    const double* xb = x + b * (size_t)T * N;
    double v = xb[(size_t)(t + q) * N + i];
    for (int l = 1; l <= q; ++l) v -= rho[(b * N + i) * q + (l - 1)] * xb[(size_t)(t + q - l) * N + i];
    out[tid] = v;
}
__global__ void ar_loadings_kernel(int B, int N, int r, int q, int Rk, const double* Lam, const double* rho, double* LamK) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (size_t)B * N * Rk) return;
    const int c = tid % Rk;
    const size_t bn = tid / Rk;
    const int l = c / r, cc = c % r;
    double v = 0.0;
    if (l <= q) v = (l == 0 ? 1.0 : -rho[bn * q + (l - 1)]) * Lam[bn * r + cc];
    LamK[tid] = v;
}

int ar_pass_run(dfm_handle* h, int B, int T, int N, int r, int nlag, int q, const double* panel, const double* Lam,
                const double* sig2, const double* rho, const double* Avar, const double* Q, const double* mu0,
                const double* P0, double* f_smooth, double* P_smooth, double* loglik, unsigned flags) {
    if (!h) return DFM_E_NULL;
    if (nlag < 1 || q < 0) return fail(h, DFM_E_DIMS, "need p >= 1 factor lags and q >= 0 idiosyncratic lags%s");
    if (int rc = check_dims(h, B, T, N, r)) return rc;
    if (T <= q) return fail(h, DFM_E_DIMS, "T must exceed the number of idiosyncratic lags%s");
    const int m = nlag > q + 1 ? nlag : q + 1, k = r * m;
    if (k > DFM_MAX_R) return fail(h, DFM_E_R_UNSUPPORTED, "r * max(p, q + 1) > DFM_MAX_R (32)%s");
    if (int rc = check_general_n(h, N, k)) return rc;
    if (!panel || !Lam || !sig2 || (q > 0 && !rho) || !Avar || !Q || !mu0 || !P0 || !f_smooth || !loglik)
        return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    HIP_TRY(h, hipSetDevice(h->device));
    const int Tq = T - q;
    Plan p = make_plan(B, Tq, N, k, flags | DFM_F_SINGULAR_Q, false, false);
    p.kdim = k; p.kb = r; p.ka = r * nlag;                   // (the companion structure: recursion_comp.hip's route; collapse width)
    p.qsing = (flags & DFM_F_SINGULAR_Q) ? 1 : 0;
    const size_t xoff = (p.total + 255) & ~(size_t)255;
    if (int rc = ensure_ws(h, xoff + (size_t)B * Tq * N * sizeof(double))) return rc;
    const int Rk = p.Rp;
    double *LamP = at<double>(h, p.LamP), *AP = at<double>(h, p.AP), *QP = at<double>(h, p.QP),
           *mu0P = at<double>(h, p.mu0P), *P0P = at<double>(h, p.P0P), *xq = at<double>(h, xoff);
    const double* xin = panel;
    if (q > 0) {
        const size_t n = (size_t)B * Tq * N;
        hipLaunchKernelGGL(quasi_diff_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, B, T, N, q, panel,
                           rho, xq);
        HIP_TRY(h, hipGetLastError());
        xin = xq;
    }
    {
        const size_t n = (size_t)B * N * Rk;
        hipLaunchKernelGGL(ar_loadings_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, B, N, r, q, Rk, Lam,
                           rho, LamP);
        HIP_TRY(h, hipGetLastError());
        const size_t nm = (size_t)B * Rk * Rk;
        hipLaunchKernelGGL(companion_pad_kernel, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, h->stream, B, N, r, k,
                           r * nlag, Rk, Rk, (const double*)nullptr, Avar, Q, mu0, P0, LamP, AP, QP, mu0P, P0P);
        HIP_TRY(h, hipGetLastError());
    }
    PaddedParams pp{LamP, AP, QP, mu0P, P0P};
    return enqueue_pass(h, p, B, Tq, N, r, xin, pp, sig2, f_smooth, P_smooth, loglik, nullptr);
}

// Joint estimation with AR(q) idiosyncratic terms by ECM (oracle/ar_oracle.py em_ar).  Per iteration: quasi-difference the
// panel at the current rho, loadings [lam, -rho_1 lam, ..] on the companion state, smoother pass + transition CM-step (the
// recursion kernel's epilogue: VAR(p) inside the m-lag state, RecursionArgs::ka), then the series CM-steps (mstep_ar.hip).
int ar_em_run(dfm_handle* h, int B, int T, int N, int r, int nlag, int q, const double* panel, double* Lam, double* sig2,
              double* rho, double* Avar, double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path,
              int* iters, double* f_smooth, double* P_smooth, unsigned flags) {
    if (!h) return DFM_E_NULL;
    if (nlag < 1 || q < 0) return fail(h, DFM_E_DIMS, "need p >= 1 factor lags and q >= 0 idiosyncratic lags%s");
    if (int rc = check_dims(h, B, T, N, r)) return rc;
    if (T <= q + 1) return fail(h, DFM_E_DIMS, "T must exceed the number of idiosyncratic lags by at least 2%s");
    const int m = nlag > q + 1 ? nlag : q + 1, k = r * m;
    if (k > DFM_MAX_R) return fail(h, DFM_E_R_UNSUPPORTED, "r * max(p, q + 1) > DFM_MAX_R (32)%s");
    if (!mstep_ar_supported(r, q)) return fail(h, DFM_E_R_UNSUPPORTED, "joint AR estimation needs r <= 8 and q <= 4%s");
    if (int rc = check_general_n(h, N, k)) return rc;
    if (!panel || !Lam || !sig2 || (q > 0 && !rho) || !Avar || !Q || !mu0 || !P0 || !loglik_path || !iters)
        return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    if (max_iter < 1) return fail(h, DFM_E_DIMS, "max_iter must be >= 1%s");
    HIP_TRY(h, hipSetDevice(h->device));
    const int Tq = T - q;
    Plan p = make_plan(B, Tq, N, k, flags | DFM_F_SINGULAR_Q, true, false);
    p.qsing = (flags & DFM_F_SINGULAR_Q) ? 1 : 0;
    p.kdim = k; p.kb = r; p.ka = r * nlag;                   // companion constraints; the observation loads on q + 1 blocks (rl = 0)
    const size_t xoff = (p.total + 255) & ~(size_t)255;
    const int Rk = p.Rp;
    const size_t moff = (xoff + (size_t)B * Tq * N * sizeof(double) + 255) & ~(size_t)255;   // the series CM-steps' moments (mstep_ar.hip)
    if (int rc = ensure_ws(h, moff + mstep_ar_workspace(B, T, N, r, q, Rk))) return rc;
    double *LamP = at<double>(h, p.LamP), *AP = at<double>(h, p.AP), *QP = at<double>(h, p.QP),
           *mu0P = at<double>(h, p.mu0P), *P0P = at<double>(h, p.P0P), *xq = at<double>(h, xoff), *mws = at<double>(h, moff);
    {
        const size_t nm = (size_t)B * Rk * Rk;
        hipLaunchKernelGGL(companion_pad_kernel, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, h->stream, B, N, r, k,
                           r * nlag, Rk, Rk, (const double*)nullptr, Avar, Q, mu0, P0, LamP, AP, QP, mu0P, P0P);
        HIP_TRY(h, hipGetLastError());
    }
    double* fsm = at<double>(h, p.fsm);
    double* Psm = at<double>(h, p.Psm);
    double* llbuf = at<double>(h, p.llbuf);
    int* active = at<int>(h, p.active);
    HIP_TRY(h, hipMemsetAsync(loglik_path, 0xFF, (size_t)B * max_iter * sizeof(double), h->stream));  // NaN
    HIP_TRY(h, hipMemsetAsync(iters, 0, (size_t)B * sizeof(int), h->stream));
    const size_t np = (size_t)r * (r + 1) / 2, npk = (size_t)Rk * (Rk + 1) / 2;
    PaddedParams pp{LamP, AP, QP, mu0P, P0P};
    std::vector<int> act_host;
    for (int it = 0; it < max_iter; ++it) {
        const double* xin = panel;
        if (q > 0) {
            const size_t n = (size_t)B * Tq * N;
            hipLaunchKernelGGL(quasi_diff_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, B, T, N, q, panel,
                               rho, xq);
            HIP_TRY(h, hipGetLastError());
            xin = xq;
        }
        {
            const size_t n = (size_t)B * N * Rk;
            hipLaunchKernelGGL(ar_loadings_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, B, N, r, q, Rk, Lam,
                               rho, LamP);
            HIP_TRY(h, hipGetLastError());
        }
        EmOpts eo;
        eo.A_out = AP; eo.Q_out = QP; eo.mu0_out = mu0P; eo.P0_out = P0P;
        eo.active = active; eo.iters = iters; eo.ll_path = loglik_path; eo.k = it; eo.max_iter = max_iter; eo.tol = tol;
        if (int rc = enqueue_pass(h, p, B, Tq, N, Rk, xin, pp, sig2, fsm, Psm, llbuf, &eo)) return rc;
        ArMstepArgs ma;
        ma.B = B; ma.T = T; ma.N = N; ma.r = r; ma.q = q; ma.Rk = Rk;
        ma.panel = panel; ma.zsm = fsm; ma.Psm = Psm; ma.active = active; ma.Lam = Lam; ma.rho = rho; ma.sig2 = sig2;
        { ProfScope ps(h, K_MSTEP_STATS); HIP_TRY(h, launch_mstep_ar(ma, mws, h->stream)); }
        if (tol > 0.0 && it + 1 < max_iter) {
            act_host.resize(B);
            HIP_TRY(h, hipMemcpyAsync(act_host.data(), active, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            bool any = false;
            for (int b = 0; b < B; ++b) any = any || act_host[b] != 0;
            if (!any) break;
        }
    }
    if (int rc = copy_block(h, B, Rk, Rk, r, r * nlag, AP, Avar)) return rc;
    if (int rc = copy_block(h, B, Rk, Rk, r, r, QP, Q)) return rc;
    if (int rc = copy_block(h, B, Rk, Rk, k, k, P0P, P0)) return rc;
    if (int rc = copy_block(h, B, 1, Rk, 1, k, mu0P, mu0)) return rc;
    if (f_smooth) if (int rc = copy_block(h, (size_t)B * Tq, 1, Rk, 1, r, fsm, f_smooth)) return rc;
    if (P_smooth) if (int rc = copy_block(h, (size_t)B * Tq, 1, (int)npk, 1, (int)np, Psm, P_smooth)) return rc;
    return 0;
}

// ---- observed factors (mstep_obs.hip, oracle/obs_oracle.py em_obs) ------------------------------------
// Per iteration: y = x - Lam_o g and the padded Lam_u -> the ordinary smoother pass on y with the transition M-step (the
// pass's own EM epilogue: A, Q, mu0, P0 of the unobserved block) -> the joint loadings regression on z = (g, f).
int obs_em_run(dfm_handle* h, int B, int T, int N, int ru, int ro, const double* panel, const double* G, double* Lam, double* R,
               double* A, double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path, int* iters,
               double* f_smooth, double* P_smooth, unsigned flags) {
    if (!h) return DFM_E_NULL;
    if (int rc = check_dims(h, B, T, N, ru)) return rc;
    const bool wide_obs = mstep_obs_wide_supported(ro, ru);   // r_o + r_u = 9 .. 32: the ordinary loadings step on augmented moments
    if (!mstep_obs_supported(ro, ru) && !wide_obs)
        return fail(h, DFM_E_R_UNSUPPORTED, "observed factors: need r_o >= 1, r_u >= 1 and r_o + r_u <= 32%s");
    if (int rc = check_em_n(h, N, ru, flags)) return rc;
    if (wide_obs && N > 1024)   // (the joint regression at r_o + r_u > 8 runs on mstep_lam_kernel: lane = series, N <= 1024)
        return fail(h, DFM_E_DIMS, "observed factors with r_o + r_u > 8: N <= 1024%s");
    if (!panel || !G || !Lam || !R || !A || !Q || !mu0 || !P0 || !loglik_path || !iters)
        return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    if (max_iter < 1) return fail(h, DFM_E_DIMS, "max_iter must be >= 1%s");
    HIP_TRY(h, hipSetDevice(h->device));
    const Plan p = make_plan(B, T, N, ru, flags, true, fast_eligible(h, N, ru, flags) && !h->em_general);
    const size_t yoff = (p.total + 255) & ~(size_t)255;
    // wide joint regression: z [B][T][Re] | Var z [B][T][NPe] | LamAug [B][N][Re] | S11, S11inv [B][Re][Re] | Dmiss [B][N][NPe]
    const int Re = wide_obs ? mstep_obs_wide_width(ro, ru) : 0;
    const size_t NPe = (size_t)Re * (Re + 1) / 2;
    size_t woff = (yoff + (size_t)B * T * N * sizeof(double) + 255) & ~(size_t)255, wend = woff;
    auto wtake = [&](size_t bytes) { const size_t o = wend; wend = (wend + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_z = wide_obs ? wtake((size_t)B * T * Re * sizeof(double)) : 0;
    const size_t o_v = wide_obs ? wtake((size_t)B * T * NPe * sizeof(double)) : 0;
    const size_t o_l = wide_obs ? wtake((size_t)B * N * Re * sizeof(double)) : 0;
    const size_t o_s = wide_obs ? wtake((size_t)B * Re * Re * sizeof(double)) : 0;
    const size_t o_i = wide_obs ? wtake((size_t)B * Re * Re * sizeof(double)) : 0;
    const size_t o_d = wide_obs ? wtake((size_t)B * N * NPe * sizeof(double)) : 0;
    if (int rc = ensure_ws(h, wend)) return rc;
    const int Rp = p.Rp, Rl = p.Rc ? p.Rc : p.Rp;            // state width, loadings width
    const bool padded = (ru != Rp);
    double *LamP = at<double>(h, p.LamP), *y = at<double>(h, yoff);
    double *AP = A, *QP = Q, *mu0P = mu0, *P0P = P0;
    if (padded) {
        AP = at<double>(h, p.AP); QP = at<double>(h, p.QP); mu0P = at<double>(h, p.mu0P); P0P = at<double>(h, p.P0P);
        const size_t n = (size_t)B * Rp * Rp;                 // (N = 0: the loadings are embedded by launch_obs_residual)
        hipLaunchKernelGGL(pad_params_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, B, 0, ru, Rp, Rl,
                           (const double*)nullptr, A, Q, mu0, P0, (double*)nullptr, AP, QP, mu0P, P0P);
        HIP_TRY(h, hipGetLastError());
    }
    double* fsm = at<double>(h, p.fsm);
    double* Psm = at<double>(h, p.Psm);
    double* llbuf = at<double>(h, p.llbuf);
    int* active = at<int>(h, p.active);
    HIP_TRY(h, hipMemsetAsync(loglik_path, 0xFF, (size_t)B * max_iter * sizeof(double), h->stream));  // NaN
    HIP_TRY(h, hipMemsetAsync(iters, 0, (size_t)B * sizeof(int), h->stream));
    ObsArgs oa;
    oa.B = B; oa.T = T; oa.N = N; oa.ro = ro; oa.ru = ru; oa.Rl = Rl;
    oa.panel = panel; oa.G = G; oa.fsm = fsm; oa.Psm = Psm; oa.active = active; oa.Lam = Lam; oa.R = R;
    PaddedParams pp{LamP, AP, QP, mu0P, P0P};
    std::vector<int> act_host;
    for (int it = 0; it < max_iter; ++it) {
        { ProfScope ps(h, K_PAD); HIP_TRY(h, launch_obs_residual(oa, y, LamP, h->stream)); }
        EmOpts eo;
        eo.A_out = AP; eo.Q_out = QP; eo.mu0_out = mu0P; eo.P0_out = P0P;
        eo.active = active; eo.iters = iters; eo.ll_path = loglik_path; eo.k = it; eo.max_iter = max_iter; eo.tol = tol;
        if (int rc = enqueue_pass(h, p, B, T, N, Rl, y, pp, R, fsm, Psm, llbuf, &eo)) return rc;
        if (!wide_obs) {
            ProfScope ps(h, K_MSTEP_STATS);
            HIP_TRY(h, launch_mstep_obs(oa, h->stream));
        } else {
            double *z = at<double>(h, o_z), *Vz = at<double>(h, o_v), *LamAug = at<double>(h, o_l);
            { ProfScope ps(h, K_PAD); HIP_TRY(h, launch_obs_augment(oa, Re, z, Vz, LamAug, at<double>(h, o_s), at<double>(h, o_i), h->stream)); }
            MstepArgs ma;
            ma.B = B; ma.T = T; ma.N = N; ma.r = Re;
            ma.panel = panel; ma.fsm = z; ma.Psm = Vz; ma.S11 = at<double>(h, o_s); ma.S11inv = at<double>(h, o_i);
            ma.Dmiss = at<double>(h, o_d);
            ma.active = active; ma.Lam_out = LamAug; ma.R_out = R; ma.lam_stride = Re;
            ma.min_cells = ro + ru + 1;                       // (as mstep_obs_kernel and the oracle: too few cells for the joint regression)
            HIP_TRY(h, hipMemsetAsync(ma.Dmiss, 0, (size_t)B * N * NPe * sizeof(double), h->stream));
            { ProfScope ps(h, K_MSTEP_STATS); HIP_TRY(h, launch_mstep_lam(Re, ma, h->stream)); }
            if (int rc = copy_block(h, (size_t)B * N, 1, Re, 1, ro + ru, LamAug, Lam)) return rc;   // (inactive replicates: their own values back)
        }
        if (tol > 0.0 && it + 1 < max_iter) {
            act_host.resize(B);
            HIP_TRY(h, hipMemcpyAsync(act_host.data(), active, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            bool any = false;
            for (int b = 0; b < B; ++b) any = any || act_host[b] != 0;
            if (!any) break;
        }
    }
    const size_t np = (size_t)ru * (ru + 1) / 2, npl = (size_t)Rl * (Rl + 1) / 2;
    if (padded) {
        if (int rc = copy_block(h, B, Rp, Rp, ru, ru, AP, A)) return rc;
        if (int rc = copy_block(h, B, Rp, Rp, ru, ru, QP, Q)) return rc;
        if (int rc = copy_block(h, B, Rp, Rp, ru, ru, P0P, P0)) return rc;
        if (int rc = copy_block(h, B, 1, Rp, 1, ru, mu0P, mu0)) return rc;
    }
    if (f_smooth) if (int rc = copy_block(h, (size_t)B * T, 1, Rl, 1, ru, fsm, f_smooth)) return rc;
    if (P_smooth) if (int rc = copy_block(h, (size_t)B * T, 1, (int)npl, 1, (int)np, Psm, P_smooth)) return rc;
    return 0;
}

}  // namespace

extern "C" {

const char* dfm_version(void) { return "dfmhip 0.1 (gfx950, fp64)"; }

int dfm_create(dfm_handle** out, int device_id, void* stream) {
    if (!out) return DFM_E_NULL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return DFM_E_NO_DEVICE;
    if (device_id < 0 || device_id >= ndev) return DFM_E_DIMS;
    dfm_handle* h = new (std::nothrow) dfm_handle();
    if (!h) return DFM_E_NULL;
    h->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) {
        if (stream) {
            h->stream = static_cast<hipStream_t>(stream);
        } else {
            e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
            h->own_stream = true;
        }
    }
    if (e == hipSuccess) {   // the side stream carries the short latency-bound kernels (gram, cov) the scan waits for:
        int lo = 0, hi = 0;      // highest priority, so that their workgroups are placed ahead of the streaming collapse
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->post, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_post, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->status_dev), 256);
    if (e == hipSuccess) e = hipMemset(h->status_dev, 0, 256);
    if (e != hipSuccess) {
        delete h;
        return (int)e;
    }
    if (const char* v = route_env("DFM_FORCE_GENERAL")) h->force_general = atoi(v) != 0;
    if (const char* v = diag_env("DFM_COLLAPSE_VARIANT")) h->collapse_variant = atoi(v);
    if (const char* v = diag_env("DFM_COLLAPSE_WPR")) { h->collapse_wpr = atoi(v); if (h->collapse_wpr < 0 || h->collapse_wpr > kSsumSlots) h->collapse_wpr = 0; }
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) h->num_cu = prop.multiProcessorCount; }
    if (const char* v = route_env("DFM_NUM_CU")) { if (atoi(v) > 0) h->num_cu = atoi(v); }   // diagnostics: persistent grids sized for fewer CUs
    if (const char* v = diag_env("DFM_NO_SIDE")) h->no_side = atoi(v) != 0;
    if (const char* v = diag_env("DFM_PIPE")) h->no_pipe = atoi(v) == 0;
    if (const char* v = diag_env("DFM_NO_RECURSION_WAVE")) h->no_rec_wave = atoi(v) != 0;
    if (const char* v = diag_env("DFM_PAIR_BMAX")) h->pair_bmax = atoi(v) > 0 ? atoi(v) : 0;
    if (const char* v = route_env("DFM_NO_PAIR")) { if (atoi(v) != 0) h->pair_bmax = 0; }
    g_widen_small_r = !h->no_rec_wave;      // process-wide: follows the most recently created handle
    if (const char* v = diag_env("DFM_NO_PFILL")) h->no_pfill = atoi(v) != 0;
    if (const char* v = diag_env("DFM_FUSED_GRAM")) h->fused_gram = atoi(v) != 0;
    if (const char* v = diag_env("DFM_NO_FUSE_COV")) h->no_fuse_cov = atoi(v) != 0;
    if (const char* v = diag_env("DFM_NO_MSTEP_MFMA")) h->no_mstep_mfma = atoi(v) != 0;
    if (const char* v = diag_env("DFM_NO_DEFER_EM")) h->no_defer_em = atoi(v) != 0;
    if (const char* v = diag_env("DFM_EM_GENERAL")) h->em_general = atoi(v) != 0;
    if (const char* v = diag_env("DFM_FUSE_GRAM")) h->fuse_gram = atoi(v) != 0;
    if (const char* v = diag_env("DFM_SUBBATCH")) h->subbatch = atoi(v);
    if (const char* v = diag_env("DFM_SCAN_ABL")) h->scan_abl = atoi(v);
    if (const char* v = route_env("DFM_MSTEP_MISS")) g_mstep_miss_mode = atoi(v);
    if (const char* v = route_env("DFM_PASS_FUSED")) h->pass_fused = atoi(v);
    if (const char* v = diag_env("DFM_PASS_NSW")) h->pass_nsw = atoi(v);
    if (const char* v = diag_env("DFM_PASS_NCOV")) h->pass_ncov = atoi(v);
    if (const char* v = diag_env("DFM_GRAM_XX_VALU")) h->gram_xx_valu = atoi(v) != 0;
    if (const char* v = diag_env("DFM_COLLAPSE_MISS_OLD")) h->collapse_miss_old = atoi(v) != 0;
    if (const char* v = diag_env("DFM_NARROW_TAB")) h->narrow_tab_off = atoi(v) == 0;
    g_odd_pad8 = true;                       // (process-wide like g_widen_small_r: follows the most recently created handle)
    if (const char* v = diag_env("DFM_ODD_PAD8")) g_odd_pad8 = atoi(v) != 0;
    if (const char* v = route_env("DFM_NO_CHUNK")) h->no_chunk = atoi(v) != 0;
    g_plan_chunk = !h->no_chunk;
    if (const char* v = route_env("DFM_CHUNK_W")) h->chunk_w = atoi(v) > 0 ? atoi(v) : 0;
    if (const char* v = route_env("DFM_CHUNK_TOL")) h->chunk_tol = atof(v) > 0.0 ? atof(v) : 0.0;
    if (const char* v = route_env("DFM_TILE_NC")) h->tile_nc = atoi(v) > 0 ? atoi(v) : 0;
    if (const char* v = route_env("DFM_TILE_W")) h->tile_w = atoi(v) > 0 ? atoi(v) : 0;
    if (const char* v = diag_env("DFM_WIDE_OLD")) h->wide_old = atoi(v) != 0;
    if (const char* v = diag_env("DFM_COV_WAVE")) h->cov_wave = atoi(v) != 0;
    *out = h;
    return 0;
}

int dfm_destroy(dfm_handle* h) {
    if (!h) return 0;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->side) { hipStreamSynchronize(h->side); hipStreamDestroy(h->side); }
    if (h->post) { hipStreamSynchronize(h->post); hipStreamDestroy(h->post); }
    if (h->ev_post) hipEventDestroy(h->ev_post);
    for (auto e : h->ev_sub) hipEventDestroy(e);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->ws) hipFree(h->ws);
    if (h->odd) hipFree(h->odd);
    if (h->status_dev) hipFree(h->status_dev);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return 0;
}

int dfm_set_stream(dfm_handle* h, void* stream) {
    if (!h) return DFM_E_NULL;
    hipStream_t ns = static_cast<hipStream_t>(stream);
    if (ns == h->stream) return 0;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->own_stream && h->stream) {
        hipStreamSynchronize(h->stream);
        hipStreamDestroy(h->stream);
        h->own_stream = false;
    } else {
        // The handle has ONE workspace (bcol, wtab, tab, status, ...): work enqueued on the new stream must not start
        // before the work already enqueued on the old one has finished with it.
        HIP_TRY(h, hipEventRecord(h->ev_fork, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(ns, h->ev_fork, 0));
    }
    h->stream = ns;
    return 0;
}

// (forward: defined with the entry points that use it)
static int status_check(dfm_handle* h);
// A host-pointer entry point opens a new status epoch: whatever earlier, unchecked *_dev calls left in the sticky word is
// read and cleared here, so that the check at the END of the call reports this call's own kernels only (a stale NaN / PCA /
// time-out bit used to fail the next unrelated host call, and silently triggered api.estimate's singular-Q retry).  The
// discarded bits are kept in the handle (discarded_status) but NOT written to dfm_last_error -- a caller that reads the string after
// a successful call must not find an error text there; device-pointer callers that care call dfm_check_status after their own calls.  These entries copy whole panels across PCIe -- one 4-byte read more is free.
static int status_epoch(dfm_handle* h) {
    if (!h->status_dev) return 0;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    int st = 0;
    HIP_TRY(h, hipMemcpy(&st, h->status_dev, sizeof(int), hipMemcpyDeviceToHost));
    if (st) {
        HIP_TRY(h, hipMemset(h->status_dev, 0, sizeof(int)));
        h->discarded_status |= st;                               // (not into h->err: this call has not failed)
    }
    return 0;
}

int dfm_synchronize(dfm_handle* h) {
    if (!h) return DFM_E_NULL;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return status_check(h);
}

int dfm_check_status(dfm_handle* h) { return dfm_synchronize(h); }

int dfm_chunk_fallbacks(dfm_handle* h, int* n_failed, int* n_total) {
    if (!h) return DFM_E_NULL;
    if (!n_failed || !n_total) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    *n_failed = 0; *n_total = 0;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!h->ck_fail_dev || h->ck_fail_n <= 0) return 0;
    std::vector<int> host((size_t)h->ck_fail_n);
    HIP_TRY(h, hipMemcpy(host.data(), h->ck_fail_dev, host.size() * sizeof(int), hipMemcpyDeviceToHost));
    int nf = 0;
    for (int v : host) nf += v != 0;
    *n_failed = nf; *n_total = h->ck_fail_n;
    return 0;
}

int dfm_profile_enable(dfm_handle* h, int on) {
    if (!h) return DFM_E_NULL;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (auto& ev : h->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    h->events.clear();
    h->profiling = on != 0;
    return 0;
}

int dfm_profile_read(dfm_handle* h, int kernel_index, char* name_out, int name_cap, double* total_ms,
                     int* launches) {
    if (!h) return DFM_E_NULL;
    if (kernel_index < 0) return DFM_E_DIMS;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->prof_names.clear();                                  // distinct kernel names, in order of first launch
    for (auto& ev : h->events) {
        bool seen = false;
        for (const char* n : h->prof_names) seen = seen || strcmp(n, ev.name) == 0;
        if (!seen) h->prof_names.push_back(ev.name);
    }
    if (kernel_index >= (int)h->prof_names.size()) return DFM_E_DIMS;
    const char* want = h->prof_names[kernel_index];
    double tot = 0.0; int n = 0;
    for (auto& ev : h->events)
        if (strcmp(ev.name, want) == 0) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) { tot += ms; ++n; }
        }
    if (name_out && name_cap > 0) { strncpy(name_out, want, name_cap - 1); name_out[name_cap - 1] = 0; }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return 0;
}

const char* dfm_last_error(const dfm_handle* h) { return h ? h->err : "null handle"; }

size_t dfm_workspace_bytes(int B, int T, int N, int r, unsigned flags) {
    if (B < 1 || T < 1 || N < 1 || r < 1 || r > DFM_MAX_R) return 0;
    // the larger of the two plans an entry point may take for this shape: sequential (general) path, or -- balanced
    // panels only -- the time-parallel fast path, whose `tab` ([B][T][3][Rp][Rp]) and M-step partial sums are larger
    size_t best = make_plan(B, T, N, r, flags, true, false).total;
    const int Rp = pad_r(r);
    if (!(flags & (DFM_F_MAY_HAVE_MISSING | DFM_F_SINGULAR_Q)) && (collapse_dma_supported(Rp, N) || collapse_wide_supported(Rp, N))) {
        const size_t f = make_plan(B, T, N, r, flags, true, true).total;
        if (f > best) best = f;
    }
    return best;
}

int dfm_ks_pass_batch_dev(dfm_handle* h, int B, int T, int N, int r, const double* panel, const double* Lam,
                          const double* R, const double* A, const double* Q, const double* mu0,
                          const double* P0, double* f_smooth, double* P_smooth, double* loglik,
                          unsigned flags) {
    if (int rc = check_dims(h, B, T, N, r)) return rc;
    if (!panel || !Lam || !R || !A || !Q || !mu0 || !P0 || !f_smooth || !loglik)
        return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    if (!fast_eligible(h, N, r, flags))
        if (int rc = check_general_n(h, N, r, true)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    if (needs_odd_pad(fast_eligible(h, N, r, flags), N, r, flags, B)) {  // odd N beyond the tilings: one all-missing series appended
        OddPad o;
        if (int rc = odd_pad(h, B, T, N, r, panel, Lam, R, &o)) return rc;
        // (the appended series is missing in EVERY period: the padded problem has missing cells whatever the caller said about N)
        return dfm_ks_pass_batch_dev(h, B, T, N + 1, r, o.panel, o.Lam, o.R, A, Q, mu0, P0, f_smooth, P_smooth, loglik,
                                     flags | DFM_F_MAY_HAVE_MISSING);
    }
    if (pipe_eligible(h, B, N, r, flags)) {
        const Plan ps = make_plan(pipe_sub(h), T, N, r, flags, false, false);
        const size_t rr = (size_t)r * r, np = (size_t)r * (r + 1) / 2;
        return pipe_run(h, B, ps.total, [&](int b0, int bn) -> int {
            PaddedParams pp;
            if (int rc = pad_params(h, ps, bn, N, r, Lam + (size_t)b0 * N * r, A + b0 * rr, Q + b0 * rr, mu0 + (size_t)b0 * r, P0 + b0 * rr, &pp)) return rc;
            return enqueue_pass(h, ps, bn, T, N, r, panel + (size_t)b0 * T * N, pp, R + (size_t)b0 * N, f_smooth + (size_t)b0 * T * r,
                                P_smooth ? P_smooth + (size_t)b0 * T * np : nullptr, loglik + b0, nullptr);
        });
    }
    const Plan p = make_plan(B, T, N, r, flags, false, fast_eligible(h, N, r, flags));
    if (int rc = ensure_ws(h, p.total)) return rc;
    PaddedParams pp;
    if (int rc = pad_params(h, p, B, N, r, Lam, A, Q, mu0, P0, &pp)) return rc;
    return enqueue_pass(h, p, B, T, N, r, panel, pp, R, f_smooth, P_smooth, loglik, nullptr);
}

// The status word of the last call's plan, read after the stream has been synchronised.  Bits: 1 = NaN in a panel that was
// declared balanced, 2 = the PCA start's subspace iteration did not converge, 4 = a bounded wait between the waves of the
// one-launch pass ran out (its outputs are invalid even where the log-likelihood happens to be finite).  Every synchronising
// entry point goes through here; device-pointer callers get the same check from dfm_synchronize / dfm_check_status.
static int status_check(dfm_handle* h) {
    if (!h->status_dev) return 0;
    int st = 0;
    HIP_TRY(h, hipMemcpy(&st, h->status_dev, sizeof(int), hipMemcpyDeviceToHost));
    if (st) HIP_TRY(h, hipMemset(h->status_dev, 0, sizeof(int)));      // reported once
    if (st & 4) return fail(h, DFM_E_NUMERIC, "one-launch pass: a bounded wait between its waves ran out (results invalid)%s");
    if (st & 1) return fail(h, DFM_E_MISSING, "panel contains NaN but DFM_F_MAY_HAVE_MISSING was not set%s");
    if (st & 2) return fail(h, DFM_E_NUMERIC, "PCA subspace iteration did not converge (near-degenerate spectrum at the cut)%s");
    return 0;
}
// status word + log-likelihood sanity after a synchronising call (loglik_host: stride doubles apart)
static int post_check(dfm_handle* h, const double* loglik_host, int B, size_t stride = 1) {
    if (int rc = status_check(h)) return rc;
    for (int b = 0; b < B; ++b)
        if (!isfinite(loglik_host[(size_t)b * stride])) return fail(h, DFM_E_NUMERIC, "non-finite log-likelihood (Q or P0 not positive definite?)%s");
    return 0;
}

int dfm_ks_pass_batch(dfm_handle* h, int B, int T, int N, int r, const double* panel, const double* Lam,
                      const double* R, const double* A, const double* Q, const double* mu0, const double* P0,
                      double* f_smooth, double* P_smooth, double* loglik, unsigned flags) {
    if (int rc = check_dims(h, B, T, N, r)) return rc;
    if (!panel || !Lam || !R || !A || !Q || !mu0 || !P0 || !f_smooth || !loglik)
        return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    if (int rc = status_epoch(h)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t d = sizeof(double), np = (size_t)r * (r + 1) / 2;
    const size_t n_panel = (size_t)B * T * N, n_lam = (size_t)B * N * r, n_R = (size_t)B * N,
                 n_m = (size_t)B * r * r, n_v = (size_t)B * r, n_f = (size_t)B * T * r, n_P = (size_t)B * T * np;
    const size_t total = (n_panel + n_lam + n_R + 3 * n_m + n_v + n_f + (P_smooth ? n_P : 0) + B) * d;
    double* buf = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), total));
    double* dp = buf;
    auto up = [&](const double* src, size_t n) -> double* {
        double* dst = dp;
        dp += n;
        hipMemcpyAsync(dst, src, n * d, hipMemcpyHostToDevice, h->stream);
        return dst;
    };
    double *x_d = up(panel, n_panel), *lam_d = up(Lam, n_lam), *R_d = up(R, n_R), *A_d = up(A, n_m),
           *Q_d = up(Q, n_m), *mu_d = up(mu0, n_v), *P0_d = up(P0, n_m);
    double* f_d = dp; dp += n_f;
    double* P_d = P_smooth ? dp : nullptr; if (P_smooth) dp += n_P;
    double* ll_d = dp;
    int rc = dfm_ks_pass_batch_dev(h, B, T, N, r, x_d, lam_d, R_d, A_d, Q_d, mu_d, P0_d, f_d, P_d, ll_d, flags);
    if (rc == 0) {
        hipMemcpyAsync(f_smooth, f_d, n_f * d, hipMemcpyDeviceToHost, h->stream);
        if (P_smooth) hipMemcpyAsync(P_smooth, P_d, n_P * d, hipMemcpyDeviceToHost, h->stream);
        hipMemcpyAsync(loglik, ll_d, B * d, hipMemcpyDeviceToHost, h->stream);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = hip_fail(h, e, "hipStreamSynchronize");
    }
    if (rc == 0) rc = post_check(h, loglik, B);
    hipFree(buf);
    return rc;
}


// ---- EM --------------------------------------------------------------------------------------------
int dfm_em_step_batch_dev(dfm_handle* h, int B, int T, int N, int r, const double* panel, double* Lam, double* R,
                          double* A, double* Q, double* mu0, double* P0, double* loglik, unsigned flags) {
    if (!h) return DFM_E_NULL;
    if (!loglik) return fail(h, DFM_E_NULL, "loglik is NULL%s");
    return em_run(h, B, T, N, r, panel, Lam, R, A, Q, mu0, P0, 1, 0.0, nullptr, nullptr, loglik, nullptr, nullptr, flags);
}

int dfm_em_batch_dev(dfm_handle* h, int B, int T, int N, int r, const double* panel, double* Lam, double* R,
                     double* A, double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path,
                     int* iters, double* f_smooth, double* P_smooth, unsigned flags) {
    if (!h) return DFM_E_NULL;
    if (!loglik_path || !iters) return fail(h, DFM_E_NULL, "loglik_path / iters is NULL%s");
    return em_run(h, B, T, N, r, panel, Lam, R, A, Q, mu0, P0, max_iter, tol, loglik_path, iters, nullptr, f_smooth,
                  P_smooth, flags);
}

int dfm_em_iterate_batch_dev(dfm_handle* h, int B, int T, int N, int r, const double* panel, double* Lam, double* R,
                             double* A, double* Q, double* mu0, double* P0, int k, int max_iter, double tol,
                             double* loglik_path, int* iters, int* active, double* f_smooth, double* P_smooth,
                             unsigned flags) {
    if (!h) return DFM_E_NULL;
    if (!loglik_path || !iters || !active) return fail(h, DFM_E_NULL, "loglik_path / iters / active is NULL%s");
    if (max_iter < 1 || k < 0 || k >= max_iter) return fail(h, DFM_E_DIMS, "need 0 <= k < max_iter%s");
    return em_run(h, B, T, N, r, panel, Lam, R, A, Q, mu0, P0, max_iter, tol, loglik_path, iters, nullptr, f_smooth, P_smooth,
                  flags, k, 1, active);
}

int dfm_em_batch(dfm_handle* h, int B, int T, int N, int r, const double* panel, double* Lam, double* R, double* A,
                 double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path, int* iters,
                 double* f_smooth, double* P_smooth, unsigned flags) {
    if (int rc = check_dims(h, B, T, N, r)) return rc;
    if (int rc = check_em_n(h, N, r, flags)) return rc;
    if (!panel || !Lam || !R || !A || !Q || !mu0 || !P0 || !loglik_path || !iters)
        return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    if (max_iter < 1) return fail(h, DFM_E_DIMS, "max_iter must be >= 1%s");
    if (int rc = status_epoch(h)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t d = sizeof(double), np = (size_t)r * (r + 1) / 2;
    const size_t n_panel = (size_t)B * T * N, n_lam = (size_t)B * N * r, n_R = (size_t)B * N, n_m = (size_t)B * r * r,
                 n_v = (size_t)B * r, n_f = (size_t)B * T * r, n_P = (size_t)B * T * np, n_ll = (size_t)B * max_iter;
    const size_t total = (n_panel + n_lam + n_R + 3 * n_m + n_v + n_f + n_P + n_ll) * d + (size_t)B * sizeof(int);
    double* buf = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), total));
    double* dp = buf;
    auto up = [&](const double* src, size_t n) -> double* {
        double* dst = dp; dp += n;
        (void)hipMemcpyAsync(dst, src, n * d, hipMemcpyHostToDevice, h->stream);
        return dst;
    };
    double *x_d = up(panel, n_panel), *lam_d = up(Lam, n_lam), *R_d = up(R, n_R), *A_d = up(A, n_m), *Q_d = up(Q, n_m),
           *mu_d = up(mu0, n_v), *P0_d = up(P0, n_m);
    double* f_d = dp; dp += n_f;
    double* P_d = dp; dp += n_P;
    double* ll_d = dp; dp += n_ll;
    int* it_d = reinterpret_cast<int*>(dp);
    int rc = dfm_em_batch_dev(h, B, T, N, r, x_d, lam_d, R_d, A_d, Q_d, mu_d, P0_d, max_iter, tol, ll_d, it_d,
                              f_smooth ? f_d : nullptr, P_smooth ? P_d : nullptr, flags);
    if (rc == 0) {
        auto down = [&](void* dst, const void* src, size_t bytes) { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream); };
        down(Lam, lam_d, n_lam * d); down(R, R_d, n_R * d); down(A, A_d, n_m * d); down(Q, Q_d, n_m * d);
        down(mu0, mu_d, n_v * d); down(P0, P0_d, n_m * d); down(loglik_path, ll_d, n_ll * d);
        down(iters, it_d, (size_t)B * sizeof(int));
        if (f_smooth) down(f_smooth, f_d, n_f * d);
        if (P_smooth) down(P_smooth, P_d, n_P * d);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = hip_fail(h, e, "hipStreamSynchronize");
    }
    if (rc == 0) rc = post_check(h, loglik_path, B, (size_t)max_iter);
    (void)hipFree(buf);
    return rc;
}
// ---- VAR(p) factor dynamics -------------------------------------------------------------------------
int dfm_ks_pass_varp_batch_dev(dfm_handle* h, int B, int T, int N, int r, int p, const double* panel, const double* Lam,
                               const double* R, const double* Avar, const double* Q, const double* mu0, const double* P0,
                               double* f_smooth, double* P_smooth, double* loglik, unsigned flags) {
    if (!h) return DFM_E_NULL;
    if (!f_smooth || !loglik) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    return varp_run(h, B, T, N, r, p, panel, const_cast<double*>(Lam), const_cast<double*>(R), const_cast<double*>(Avar),
                    const_cast<double*>(Q), const_cast<double*>(mu0), const_cast<double*>(P0), 1, 0.0, nullptr, nullptr,
                    loglik, f_smooth, P_smooth, flags, false);
}

int dfm_em_varp_batch_dev(dfm_handle* h, int B, int T, int N, int r, int p, const double* panel, double* Lam, double* R,
                          double* Avar, double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path,
                          int* iters, double* f_smooth, double* P_smooth, unsigned flags) {
    if (!h) return DFM_E_NULL;
    if (!loglik_path || !iters) return fail(h, DFM_E_NULL, "loglik_path / iters is NULL%s");
    return varp_run(h, B, T, N, r, p, panel, Lam, R, Avar, Q, mu0, P0, max_iter, tol, loglik_path, iters, nullptr,
                    f_smooth, P_smooth, flags, true);
}

// host-pointer variants: one device block for inputs and outputs, parameters copied in and (EM) out
static int varp_host(dfm_handle* h, int B, int T, int N, int r, int p, const double* panel, double* Lam, double* R,
                     double* Avar, double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path,
                     int* iters, double* f_smooth, double* P_smooth, double* loglik, unsigned flags, bool em) {
    if (int rc = check_dims(h, B, T, N, r)) return rc;
    if (p < 1 || r * p > DFM_MAX_R) return fail(h, DFM_E_R_UNSUPPORTED, "need 1 <= p and r * p <= DFM_MAX_R (32)%s");
    if (!panel || !Lam || !R || !Avar || !Q || !mu0 || !P0) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    if (em ? (!loglik_path || !iters) : (!f_smooth || !loglik)) return fail(h, DFM_E_NULL, "required output pointer is NULL%s");
    if (max_iter < 1) return fail(h, DFM_E_DIMS, "max_iter must be >= 1%s");
    if (int rc = status_epoch(h)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t d = sizeof(double), k = (size_t)r * p, np = (size_t)r * (r + 1) / 2;
    const size_t n_panel = (size_t)B * T * N, n_lam = (size_t)B * N * r, n_R = (size_t)B * N, n_a = (size_t)B * r * k,
                 n_q = (size_t)B * r * r, n_v = (size_t)B * k, n_p0 = (size_t)B * k * k, n_f = (size_t)B * T * r,
                 n_P = (size_t)B * T * np, n_ll = em ? (size_t)B * max_iter : (size_t)B;
    const size_t total = (n_panel + n_lam + n_R + n_a + n_q + n_v + n_p0 + n_f + n_P + n_ll) * d + (size_t)B * sizeof(int);
    double* buf = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), total));
    double* dp = buf;
    auto up = [&](const double* src, size_t n) -> double* {
        double* dst = dp; dp += n;
        (void)hipMemcpyAsync(dst, src, n * d, hipMemcpyHostToDevice, h->stream);
        return dst;
    };
    double *x_d = up(panel, n_panel), *lam_d = up(Lam, n_lam), *R_d = up(R, n_R), *A_d = up(Avar, n_a), *Q_d = up(Q, n_q),
           *mu_d = up(mu0, n_v), *P0_d = up(P0, n_p0);
    double* f_d = dp; dp += n_f;
    double* P_d = dp; dp += n_P;
    double* ll_d = dp; dp += n_ll;
    int* it_d = reinterpret_cast<int*>(dp);
    int rc = em ? dfm_em_varp_batch_dev(h, B, T, N, r, p, x_d, lam_d, R_d, A_d, Q_d, mu_d, P0_d, max_iter, tol, ll_d, it_d,
                                        f_smooth ? f_d : nullptr, P_smooth ? P_d : nullptr, flags)
                : dfm_ks_pass_varp_batch_dev(h, B, T, N, r, p, x_d, lam_d, R_d, A_d, Q_d, mu_d, P0_d, f_d,
                                             P_smooth ? P_d : nullptr, ll_d, flags);
    if (rc == 0) {
        auto down = [&](void* dst, const void* src, size_t bytes) { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream); };
        if (em) {
            down(Lam, lam_d, n_lam * d); down(R, R_d, n_R * d); down(Avar, A_d, n_a * d); down(Q, Q_d, n_q * d);
            down(mu0, mu_d, n_v * d); down(P0, P0_d, n_p0 * d); down(loglik_path, ll_d, n_ll * d);
            down(iters, it_d, (size_t)B * sizeof(int));
        } else {
            down(loglik, ll_d, n_ll * d);
        }
        if (f_smooth) down(f_smooth, f_d, n_f * d);
        if (P_smooth) down(P_smooth, P_d, n_P * d);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = hip_fail(h, e, "hipStreamSynchronize");
    }
    if (rc == 0) rc = post_check(h, em ? loglik_path : loglik, B, em ? (size_t)max_iter : (size_t)1);
    (void)hipFree(buf);
    return rc;
}

int dfm_ks_pass_varp_batch(dfm_handle* h, int B, int T, int N, int r, int p, const double* panel, const double* Lam,
                           const double* R, const double* Avar, const double* Q, const double* mu0, const double* P0,
                           double* f_smooth, double* P_smooth, double* loglik, unsigned flags) {
    return varp_host(h, B, T, N, r, p, panel, const_cast<double*>(Lam), const_cast<double*>(R), const_cast<double*>(Avar),
                     const_cast<double*>(Q), const_cast<double*>(mu0), const_cast<double*>(P0), 1, 0.0, nullptr, nullptr,
                     f_smooth, P_smooth, loglik, flags, false);
}

int dfm_em_varp_batch(dfm_handle* h, int B, int T, int N, int r, int p, const double* panel, double* Lam, double* R,
                      double* Avar, double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path,
                      int* iters, double* f_smooth, double* P_smooth, unsigned flags) {
    return varp_host(h, B, T, N, r, p, panel, Lam, R, Avar, Q, mu0, P0, max_iter, tol, loglik_path, iters, f_smooth,
                     P_smooth, nullptr, flags, true);
}

// ---- AR idiosyncratic terms -----------------------------------------------------------------------
int dfm_ks_pass_ar_batch_dev(dfm_handle* h, int B, int T, int N, int r, int p, int q, const double* panel, const double* Lam,
                             const double* sig2, const double* rho, const double* Avar, const double* Q, const double* mu0,
                             const double* P0, double* f_smooth, double* P_smooth, double* loglik, unsigned flags) {
    return ar_pass_run(h, B, T, N, r, p, q, panel, Lam, sig2, rho, Avar, Q, mu0, P0, f_smooth, P_smooth, loglik, flags);
}

int dfm_ks_pass_ar_batch(dfm_handle* h, int B, int T, int N, int r, int p, int q, const double* panel, const double* Lam,
                         const double* sig2, const double* rho, const double* Avar, const double* Q, const double* mu0,
                         const double* P0, double* f_smooth, double* P_smooth, double* loglik, unsigned flags) {
    if (int rc = check_dims(h, B, T, N, r)) return rc;
    if (p < 1 || q < 0 || T <= q) return fail(h, DFM_E_DIMS, "need p >= 1, 0 <= q < T%s");
    const int m = p > q + 1 ? p : q + 1;
    if (r * m > DFM_MAX_R) return fail(h, DFM_E_R_UNSUPPORTED, "r * max(p, q + 1) > DFM_MAX_R (32)%s");
    if (!panel || !Lam || !sig2 || (q > 0 && !rho) || !Avar || !Q || !mu0 || !P0 || !f_smooth || !loglik)
        return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    if (int rc = status_epoch(h)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t d = sizeof(double), k = (size_t)r * m, np = (size_t)r * (r + 1) / 2, Tq = (size_t)(T - q);
    const size_t n_panel = (size_t)B * T * N, n_lam = (size_t)B * N * r, n_R = (size_t)B * N, n_rho = (size_t)B * N * q,
                 n_a = (size_t)B * r * r * p, n_q = (size_t)B * r * r, n_v = (size_t)B * k, n_p0 = (size_t)B * k * k,
                 n_f = (size_t)B * Tq * r, n_P = (size_t)B * Tq * np;
    double* buf = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), (n_panel + n_lam + n_R + n_rho + n_a + n_q + n_v + n_p0 + n_f + n_P + B) * d));
    double* dp = buf;
    auto up = [&](const double* src, size_t n) -> double* {
        double* dst = dp; dp += n;
        if (n) (void)hipMemcpyAsync(dst, src, n * d, hipMemcpyHostToDevice, h->stream);
        return dst;
    };
    double *x_d = up(panel, n_panel), *lam_d = up(Lam, n_lam), *R_d = up(sig2, n_R), *rho_d = up(rho, n_rho),
           *A_d = up(Avar, n_a), *Q_d = up(Q, n_q), *mu_d = up(mu0, n_v), *P0_d = up(P0, n_p0);
    double* f_d = dp; dp += n_f;
    double* P_d = dp; dp += n_P;
    double* ll_d = dp;
    int rc = ar_pass_run(h, B, T, N, r, p, q, x_d, lam_d, R_d, rho_d, A_d, Q_d, mu_d, P0_d, f_d, P_smooth ? P_d : nullptr, ll_d,
                         flags);
    if (rc == 0) {
        (void)hipMemcpyAsync(f_smooth, f_d, n_f * d, hipMemcpyDeviceToHost, h->stream);
        if (P_smooth) (void)hipMemcpyAsync(P_smooth, P_d, n_P * d, hipMemcpyDeviceToHost, h->stream);
        (void)hipMemcpyAsync(loglik, ll_d, (size_t)B * d, hipMemcpyDeviceToHost, h->stream);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = hip_fail(h, e, "hipStreamSynchronize");
    }
    if (rc == 0) rc = post_check(h, loglik, B);
    (void)hipFree(buf);
    return rc;
}

int dfm_em_ar_batch_dev(dfm_handle* h, int B, int T, int N, int r, int p, int q, const double* panel, double* Lam, double* sig2,
                        double* rho, double* Avar, double* Q, double* mu0, double* P0, int max_iter, double tol,
                        double* loglik_path, int* iters, double* f_smooth, double* P_smooth, unsigned flags) {
    return ar_em_run(h, B, T, N, r, p, q, panel, Lam, sig2, rho, Avar, Q, mu0, P0, max_iter, tol, loglik_path, iters, f_smooth,
                     P_smooth, flags);
}

int dfm_em_ar_batch(dfm_handle* h, int B, int T, int N, int r, int p, int q, const double* panel, double* Lam, double* sig2,
                    double* rho, double* Avar, double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path,
                    int* iters, double* f_smooth, double* P_smooth, unsigned flags) {
    if (int rc = check_dims(h, B, T, N, r)) return rc;
    if (p < 1 || q < 0 || T <= q + 1) return fail(h, DFM_E_DIMS, "need p >= 1, 0 <= q < T - 1%s");
    const int m = p > q + 1 ? p : q + 1;
    if (r * m > DFM_MAX_R) return fail(h, DFM_E_R_UNSUPPORTED, "r * max(p, q + 1) > DFM_MAX_R (32)%s");
    if (!panel || !Lam || !sig2 || (q > 0 && !rho) || !Avar || !Q || !mu0 || !P0 || !loglik_path || !iters)
        return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    if (max_iter < 1) return fail(h, DFM_E_DIMS, "max_iter must be >= 1%s");
    if (int rc = status_epoch(h)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t d = sizeof(double), k = (size_t)r * m, np = (size_t)r * (r + 1) / 2, Tq = (size_t)(T - q);
    const size_t n_panel = (size_t)B * T * N, n_lam = (size_t)B * N * r, n_R = (size_t)B * N, n_rho = (size_t)B * N * q,
                 n_a = (size_t)B * r * r * p, n_q = (size_t)B * r * r, n_v = (size_t)B * k, n_p0 = (size_t)B * k * k,
                 n_f = (size_t)B * Tq * r, n_P = (size_t)B * Tq * np, n_ll = (size_t)B * max_iter;
    double* buf = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf),
                         (n_panel + n_lam + n_R + n_rho + n_a + n_q + n_v + n_p0 + n_f + n_P + n_ll) * d + (size_t)B * sizeof(int)));
    double* dp = buf;
    auto up = [&](const double* src, size_t n) -> double* {
        double* dst = dp; dp += n;
        if (n) (void)hipMemcpyAsync(dst, src, n * d, hipMemcpyHostToDevice, h->stream);
        return dst;
    };
    double *x_d = up(panel, n_panel), *lam_d = up(Lam, n_lam), *R_d = up(sig2, n_R), *rho_d = up(rho, n_rho),
           *A_d = up(Avar, n_a), *Q_d = up(Q, n_q), *mu_d = up(mu0, n_v), *P0_d = up(P0, n_p0);
    double* f_d = dp; dp += n_f;
    double* P_d = dp; dp += n_P;
    double* ll_d = dp; dp += n_ll;
    int* it_d = reinterpret_cast<int*>(dp);
    int rc = ar_em_run(h, B, T, N, r, p, q, x_d, lam_d, R_d, rho_d, A_d, Q_d, mu_d, P0_d, max_iter, tol, ll_d, it_d,
                       f_smooth ? f_d : nullptr, P_smooth ? P_d : nullptr, flags);
    if (rc == 0) {
        auto down = [&](void* dst, const void* src, size_t bytes) { if (bytes) (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream); };
        down(Lam, lam_d, n_lam * d); down(sig2, R_d, n_R * d); down(rho, rho_d, n_rho * d); down(Avar, A_d, n_a * d);
        down(Q, Q_d, n_q * d); down(mu0, mu_d, n_v * d); down(P0, P0_d, n_p0 * d); down(loglik_path, ll_d, n_ll * d);
        down(iters, it_d, (size_t)B * sizeof(int));
        if (f_smooth) down(f_smooth, f_d, n_f * d);
        if (P_smooth) down(P_smooth, P_d, n_P * d);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = hip_fail(h, e, "hipStreamSynchronize");
    }
    if (rc == 0) rc = post_check(h, loglik_path, B, (size_t)max_iter);
    (void)hipFree(buf);
    return rc;
}

// ---- observed factors ------------------------------------------------------------------------------------------------
int dfm_em_obs_batch_dev(dfm_handle* h, int B, int T, int N, int r_u, int r_o, const double* panel, const double* G, double* Lam,
                         double* R, double* A, double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path,
                         int* iters, double* f_smooth, double* P_smooth, unsigned flags) {
    return obs_em_run(h, B, T, N, r_u, r_o, panel, G, Lam, R, A, Q, mu0, P0, max_iter, tol, loglik_path, iters, f_smooth, P_smooth,
                      flags);
}

int dfm_em_obs_batch(dfm_handle* h, int B, int T, int N, int r_u, int r_o, const double* panel, const double* G, double* Lam,
                     double* R, double* A, double* Q, double* mu0, double* P0, int max_iter, double tol, double* loglik_path,
                     int* iters, double* f_smooth, double* P_smooth, unsigned flags) {
    if (int rc = check_dims(h, B, T, N, r_u)) return rc;
    if (r_o < 1 || r_o + r_u > 32) return fail(h, DFM_E_R_UNSUPPORTED, "observed factors: need r_o >= 1 and r_o + r_u <= 32%s");
    if (!panel || !G || !Lam || !R || !A || !Q || !mu0 || !P0 || !loglik_path || !iters)
        return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    if (max_iter < 1) return fail(h, DFM_E_DIMS, "max_iter must be >= 1%s");
    for (size_t k = 0; k < (size_t)B * T * r_o; ++k)
        if (G[k] != G[k]) return fail(h, DFM_E_MISSING, "observed factors must not contain NaN%s");
    if (int rc = status_epoch(h)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t d = sizeof(double), re = (size_t)r_o + r_u, np = (size_t)r_u * (r_u + 1) / 2;
    const size_t n_panel = (size_t)B * T * N, n_g = (size_t)B * T * r_o, n_lam = (size_t)B * N * re, n_R = (size_t)B * N,
                 n_m = (size_t)B * r_u * r_u, n_v = (size_t)B * r_u, n_f = (size_t)B * T * r_u, n_P = (size_t)B * T * np,
                 n_ll = (size_t)B * max_iter;
    double* buf = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), (n_panel + n_g + n_lam + n_R + 3 * n_m + n_v + n_f + n_P + n_ll) * d + (size_t)B * sizeof(int)));
    double* dp = buf;
    auto up = [&](const double* src, size_t n) -> double* {
        double* dst = dp; dp += n;
        (void)hipMemcpyAsync(dst, src, n * d, hipMemcpyHostToDevice, h->stream);
        return dst;
    };
    double *x_d = up(panel, n_panel), *g_d = up(G, n_g), *lam_d = up(Lam, n_lam), *R_d = up(R, n_R), *A_d = up(A, n_m),
           *Q_d = up(Q, n_m), *mu_d = up(mu0, n_v), *P0_d = up(P0, n_m);
    double* f_d = dp; dp += n_f;
    double* P_d = dp; dp += n_P;
    double* ll_d = dp; dp += n_ll;
    int* it_d = reinterpret_cast<int*>(dp);
    int rc = obs_em_run(h, B, T, N, r_u, r_o, x_d, g_d, lam_d, R_d, A_d, Q_d, mu_d, P0_d, max_iter, tol, ll_d, it_d,
                        f_smooth ? f_d : nullptr, P_smooth ? P_d : nullptr, flags);
    if (rc == 0) {
        auto down = [&](void* dst, const void* src, size_t bytes) { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream); };
        down(Lam, lam_d, n_lam * d); down(R, R_d, n_R * d); down(A, A_d, n_m * d); down(Q, Q_d, n_m * d);
        down(mu0, mu_d, n_v * d); down(P0, P0_d, n_m * d); down(loglik_path, ll_d, n_ll * d);
        down(iters, it_d, (size_t)B * sizeof(int));
        if (f_smooth) down(f_smooth, f_d, n_f * d);
        if (P_smooth) down(P_smooth, P_d, n_P * d);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = hip_fail(h, e, "hipStreamSynchronize");
    }
    if (rc == 0) rc = post_check(h, loglik_path, B, (size_t)max_iter);
    (void)hipFree(buf);
    return rc;
}

int dfm_pca_init_batch_dev(dfm_handle* h, int B, int T, int N, int r, const double* panel, double* Lam, double* R,
                           double* A, double* Q, double* mu0, double* P0, double* factors) {
    if (!h) return DFM_E_NULL;
    if (B < 1 || T < 2 || N < 1 || r < 1) return fail(h, DFM_E_DIMS, "B, N, r must be >= 1 and T >= 2%s");
    if (r > DFM_MAX_R) return fail(h, DFM_E_R_UNSUPPORTED, "r > DFM_MAX_R (32)%s");
    if (r > N || r > T - 1) return fail(h, DFM_E_DIMS, "r must not exceed N or T - 1%s");
    if (!panel || !Lam || !R || !A || !Q || !mu0 || !P0) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    HIP_TRY(h, hipSetDevice(h->device));
    const int Rp = pad_r(r);
    const size_t d = sizeof(double);
    size_t off = 0;
    const size_t oS = take(off, (size_t)B * N * N * d), oV = take(off, (size_t)B * N * Rp * d),
                 oY = take(off, (size_t)B * N * Rp * d), oF = take(off, (size_t)B * T * Rp * d), oSt = take(off, 256);
    if (int rc = ensure_ws(h, off)) return rc;
    PcaArgs pa;
    pa.B = B; pa.T = T; pa.N = N; pa.r = r; pa.max_iter = 4000;
    { static const int mi = [] { const char* v = diag_env("DFM_PCA_MAXIT"); return v ? atoi(v) : 0; }(); if (mi > 0) pa.max_iter = mi; }   // diagnostics
    { static const int stop = [] { const char* v = diag_env("DFM_PCA_STOP"); return v ? atoi(v) : 0; }(); pa.stop_after = stop; }
    pa.panel = panel;
    pa.S = at<double>(h, oS); pa.V = at<double>(h, oV); pa.Y = at<double>(h, oY); pa.F = at<double>(h, oF);
    pa.Lam = Lam; pa.Rv = R; pa.A = A; pa.Q = Q; pa.mu0 = mu0; pa.P0 = P0; pa.factors = factors;
    pa.status = h->status_dev;
    { ProfScope ps(h, K_GRAM_XX); HIP_TRY(h, launch_gram_xx(pa, h->stream, h->gram_xx_valu ? 1 : 0)); }
    { ProfScope ps(h, K_PCA); HIP_TRY(h, launch_pca(Rp, pa, h->stream)); }
    return 0;
}

int dfm_pca_init_batch(dfm_handle* h, int B, int T, int N, int r, const double* panel, double* Lam, double* R,
                       double* A, double* Q, double* mu0, double* P0, double* factors) {
    if (!h) return DFM_E_NULL;
    if (B < 1 || T < 2 || N < 1 || r < 1 || r > DFM_MAX_R) return fail(h, DFM_E_DIMS, "bad dimensions%s");
    if (!panel || !Lam || !R || !A || !Q || !mu0 || !P0) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    if (int rc = status_epoch(h)) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t d = sizeof(double);
    const size_t n_panel = (size_t)B * T * N, n_lam = (size_t)B * N * r, n_R = (size_t)B * N, n_m = (size_t)B * r * r,
                 n_v = (size_t)B * r, n_f = (size_t)B * T * r;
    for (size_t k = 0; k < n_panel; ++k)
        if (panel[k] != panel[k]) return fail(h, DFM_E_MISSING, "PCA initialisation needs a balanced panel (NaN found)%s");
    double* buf = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), (n_panel + n_lam + n_R + 3 * n_m + n_v + n_f) * d));
    double* x_d = buf; double* lam_d = x_d + n_panel; double* R_d = lam_d + n_lam; double* A_d = R_d + n_R;
    double* Q_d = A_d + n_m; double* P0_d = Q_d + n_m; double* mu_d = P0_d + n_m; double* f_d = mu_d + n_v;
    (void)hipMemcpyAsync(x_d, panel, n_panel * d, hipMemcpyHostToDevice, h->stream);
    int rc = dfm_pca_init_batch_dev(h, B, T, N, r, x_d, lam_d, R_d, A_d, Q_d, mu_d, P0_d, factors ? f_d : nullptr);
    if (rc == 0) {
        auto down = [&](void* dst, const void* src, size_t bytes) { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream); };
        down(Lam, lam_d, n_lam * d); down(R, R_d, n_R * d); down(A, A_d, n_m * d); down(Q, Q_d, n_m * d);
        down(mu0, mu_d, n_v * d); down(P0, P0_d, n_m * d);
        if (factors) down(factors, f_d, n_f * d);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = hip_fail(h, e, "hipStreamSynchronize");
    }
    if (rc == 0) rc = status_check(h);
    (void)hipFree(buf);
    return rc;
}
int dfm_synth_panels_dev(dfm_handle* h, uint64_t seed, int64_t first_replicate, int B, int T, int N, int r,
                         double missing_prob, double* panel, double* Lam, double* R, double* A, double* Q,
                         double* mu0, double* P0) {
    if (!h) return DFM_E_NULL;
    if (B < 1 || T < 2 || N < 1 || r < 1) return fail(h, DFM_E_DIMS, "B, N, r must be >= 1 and T >= 2%s");
    if (r > DFM_MAX_R) return fail(h, DFM_E_R_UNSUPPORTED, "r > DFM_MAX_R (32)%s");
    if (!(missing_prob >= 0.0 && missing_prob < 1.0)) return fail(h, DFM_E_DIMS, "missing_prob must be in [0, 1)%s");
    if (!panel || !Lam || !R || !A || !Q || !mu0 || !P0) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    HIP_TRY(h, hipSetDevice(h->device));
    size_t off = 0;
    const size_t oF = take(off, (size_t)B * (T + 1) * r * sizeof(double));
    const size_t oC = take(off, (size_t)B * synth_tiles(T) * 2 * N * sizeof(double));
    if (int rc = ensure_ws(h, off)) return rc;
    SynthArgs sa;
    sa.B = B; sa.T = T; sa.N = N; sa.r = r; sa.seed = seed; sa.first_replicate = first_replicate;
    sa.missing_prob = missing_prob;
    sa.panel = panel; sa.Lam = Lam; sa.R = R; sa.A = A; sa.Q = Q; sa.mu0 = mu0; sa.P0 = P0;
    sa.fscratch = at<double>(h, oF);
    sa.colstats = at<double>(h, oC);
    { ProfScope ps(h, K_SYNTH); HIP_TRY(h, launch_synth(sa, h->stream)); }
    return 0;
}

}  // extern "C"


// ---- non-parametric estimator: batched ALS and batched complete-case OLS (als.hip) ---------------------
int dfm_als_batch_dev(dfm_handle* h, int B, int T, int N, int r, const double* z, long long z_stride,
                      const int* r_each, double* F, double* Lam, int nt_min, int max_iter, double tol,
                      double* ssr_path, int path_cap, int* iters, double* ssr, double* R2) {
    if (!h) return DFM_E_NULL;
    if (B < 1 || T < 1 || N < 1 || r < 1 || max_iter < 1 || z_stride < 0 || (ssr_path && path_cap < 1))
        return fail(h, DFM_E_DIMS, "B, T, N, r, max_iter must be >= 1, z_stride >= 0%s");
    if (r > DFM_MAX_R) return fail(h, DFM_E_R_UNSUPPORTED, "r > DFM_MAX_R (32)%s");
    if (!z || !F || !Lam || !iters || !ssr) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    const int Rp = pad_r(r);
    if (!als_fits(Rp, T, N)) return fail(h, DFM_E_DIMS, "(T + N) * pad(r) doubles exceed the 160 KB of LDS%s");
    HIP_TRY(h, hipSetDevice(h->device));
    AlsArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.N = N; a.rmax = r; a.z = z; a.z_stride = z_stride; a.r_each = r_each; a.F = F; a.Lam = Lam;
    a.nt_min = nt_min; a.max_iter = max_iter; a.tol = tol; a.ssr_path = ssr_path; a.path_cap = ssr_path ? path_cap : 0;
    a.iters = iters; a.ssr = ssr; a.R2 = R2;
    { ProfScope ps(h, K_ALS); HIP_TRY(h, launch_als(Rp, a, h->stream)); }
    return 0;
}

int dfm_als_batch(dfm_handle* h, int B, int T, int N, int r, const double* z, long long z_stride, const int* r_each,
                  double* F, double* Lam, int nt_min, int max_iter, double tol, double* ssr_path, int path_cap,
                  int* iters, double* ssr, double* R2) {
    if (!h) return DFM_E_NULL;
    if (B < 1 || T < 1 || N < 1 || r < 1 || z_stride < 0) return fail(h, DFM_E_DIMS, "B, T, N, r must be >= 1%s");
    if (!z || !F || !Lam || !iters || !ssr) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t d = sizeof(double);
    const size_t n_z = z_stride == 0 ? (size_t)T * N : (size_t)(B - 1) * z_stride + (size_t)T * N;
    const size_t n_F = (size_t)B * T * r, n_L = (size_t)B * N * r, n_p = ssr_path ? (size_t)B * path_cap : 0,
                 n_R2 = R2 ? (size_t)B * N : 0;
    char* buf = nullptr;
    const size_t bytes = (n_z + n_F + n_L + n_p + n_R2 + B) * d + (size_t)2 * B * sizeof(int) + 64;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), bytes));
    double* z_d = reinterpret_cast<double*>(buf);
    double* F_d = z_d + n_z; double* L_d = F_d + n_F; double* p_d = L_d + n_L; double* R2_d = p_d + n_p;
    double* ssr_d = R2_d + n_R2;
    int* it_d = reinterpret_cast<int*>(ssr_d + B);
    int* re_d = it_d + B;
    hipMemcpyAsync(z_d, z, n_z * d, hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(F_d, F, n_F * d, hipMemcpyHostToDevice, h->stream);
    if (r_each) hipMemcpyAsync(re_d, r_each, (size_t)B * sizeof(int), hipMemcpyHostToDevice, h->stream);
    int rc = dfm_als_batch_dev(h, B, T, N, r, z_d, z_stride, r_each ? re_d : nullptr, F_d, L_d, nt_min, max_iter, tol,
                               ssr_path ? p_d : nullptr, path_cap, it_d, ssr_d, R2 ? R2_d : nullptr);
    if (rc == 0) {
        hipMemcpyAsync(F, F_d, n_F * d, hipMemcpyDeviceToHost, h->stream);
        hipMemcpyAsync(Lam, L_d, n_L * d, hipMemcpyDeviceToHost, h->stream);
        if (ssr_path) hipMemcpyAsync(ssr_path, p_d, n_p * d, hipMemcpyDeviceToHost, h->stream);
        if (R2) hipMemcpyAsync(R2, R2_d, n_R2 * d, hipMemcpyDeviceToHost, h->stream);
        hipMemcpyAsync(ssr, ssr_d, (size_t)B * d, hipMemcpyDeviceToHost, h->stream);
        hipMemcpyAsync(iters, it_d, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, h->stream);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = hip_fail(h, e, "hipStreamSynchronize");
    }
    hipFree(buf);
    return rc;
}

int dfm_ols_batch_dev(dfm_handle* h, int P, int T, int K, const double* X, long long x_stride, const double* y,
                      long long y_stride, long long y_inc, int nt_min, double* beta, double* resid, double* ssr,
                      double* tss, int* nobs) {
    if (!h) return DFM_E_NULL;
    if (P < 1 || T < 1 || K < 1 || x_stride < 0 || y_inc < 1 || y_stride < 0)
        return fail(h, DFM_E_DIMS, "P, T, K, y_inc must be >= 1, strides >= 0%s");
    if (K > 64) return fail(h, DFM_E_R_UNSUPPORTED, "K > 64 regressors%s");
    if (!X || !y || !beta || !ssr || !nobs) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    HIP_TRY(h, hipSetDevice(h->device));
    OlsArgs a;
    memset(&a, 0, sizeof(a));
    a.P = P; a.T = T; a.K = K; a.X = X; a.x_stride = x_stride; a.y = y; a.y_stride = y_stride; a.y_inc = y_inc;
    a.nt_min = nt_min; a.beta = beta; a.resid = resid; a.ssr = ssr; a.tss = tss; a.nobs = nobs;
    { ProfScope ps(h, K_OLS); HIP_TRY(h, launch_ols(K > 32 ? 64 : pad_r(K), a, h->stream)); }
    return 0;
}

int dfm_ols_batch(dfm_handle* h, int P, int T, int K, const double* X, long long x_stride, const double* y,
                  long long y_stride, long long y_inc, int nt_min, double* beta, double* resid, double* ssr,
                  double* tss, int* nobs) {
    if (!h) return DFM_E_NULL;
    if (P < 1 || T < 1 || K < 1 || x_stride < 0 || y_inc < 1 || y_stride < 0)
        return fail(h, DFM_E_DIMS, "P, T, K, y_inc must be >= 1, strides >= 0%s");
    if (!X || !y || !beta || !ssr || !nobs) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t d = sizeof(double);
    const size_t n_X = (x_stride == 0 ? 0 : (size_t)(P - 1) * x_stride) + (size_t)T * K;
    const size_t n_y = (size_t)(P - 1) * y_stride + (size_t)(T - 1) * y_inc + 1;
    const size_t n_b = (size_t)P * K, n_e = resid ? (size_t)P * T : 0;
    char* buf = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), (n_X + n_y + n_b + n_e + 2 * (size_t)P) * d + (size_t)P * sizeof(int) + 64));
    double* X_d = reinterpret_cast<double*>(buf);
    double* y_d = X_d + n_X; double* b_d = y_d + n_y; double* e_d = b_d + n_b; double* ssr_d = e_d + n_e;
    double* tss_d = ssr_d + P;
    int* n_d = reinterpret_cast<int*>(tss_d + P);
    hipMemcpyAsync(X_d, X, n_X * d, hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(y_d, y, n_y * d, hipMemcpyHostToDevice, h->stream);
    int rc = dfm_ols_batch_dev(h, P, T, K, X_d, x_stride, y_d, y_stride, y_inc, nt_min, b_d, resid ? e_d : nullptr, ssr_d,
                               tss_d, n_d);
    if (rc == 0) {
        hipMemcpyAsync(beta, b_d, n_b * d, hipMemcpyDeviceToHost, h->stream);
        if (resid) hipMemcpyAsync(resid, e_d, n_e * d, hipMemcpyDeviceToHost, h->stream);
        hipMemcpyAsync(ssr, ssr_d, (size_t)P * d, hipMemcpyDeviceToHost, h->stream);
        if (tss) hipMemcpyAsync(tss, tss_d, (size_t)P * d, hipMemcpyDeviceToHost, h->stream);
        hipMemcpyAsync(nobs, n_d, (size_t)P * sizeof(int), hipMemcpyDeviceToHost, h->stream);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = hip_fail(h, e, "hipStreamSynchronize");
    }
    hipFree(buf);
    return rc;
}


// ---- wild-bootstrap IRF bands (boot.hip) -----------------------------------------------------------------
int dfm_var_bootstrap_irf_dev(dfm_handle* h, int B, int T, int ns, int p, int H, const double* y, const double* betahat,
                              const double* resid, const double* signs, uint64_t seed, int64_t first_draw, double* beta_out,
                              double* irf) {
    if (!h) return DFM_E_NULL;
    if (B < 1 || ns < 1 || p < 1 || H < 1 || T <= p + 1 + ns * p)
        return fail(h, DFM_E_DIMS, "B, ns, p, H must be >= 1 and T > p + 1 + ns p%s");
    if (ns > 8 || 1 + ns * p > 64) return fail(h, DFM_E_R_UNSUPPORTED, "ns > 8 or 1 + ns p > 64%s");
    if (!y || !betahat || !resid || !irf) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    HIP_TRY(h, hipSetDevice(h->device));
    BootArgs a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.ns = ns; a.p = p; a.H = H; a.y = y; a.betahat = betahat; a.resid = resid; a.signs = signs;
    a.seed = seed; a.first_draw = first_draw; a.beta_out = beta_out; a.irf = irf;
    hipError_t e;
    { ProfScope ps(h, K_BOOT); e = launch_var_boot(a, h->stream); }
    if (e == hipErrorInvalidValue) return fail(h, DFM_E_DIMS, "T x ns too large for the bootstrap kernel's LDS%s");
    HIP_TRY(h, e);
    return 0;
}

int dfm_var_bootstrap_irf(dfm_handle* h, int B, int T, int ns, int p, int H, const double* y, const double* betahat,
                          const double* resid, const double* signs, uint64_t seed, int64_t first_draw, double* beta_out,
                          double* irf) {
    if (!h) return DFM_E_NULL;
    if (B < 1 || ns < 1 || p < 1 || H < 1 || T < 1) return fail(h, DFM_E_DIMS, "B, T, ns, p, H must be >= 1%s");
    if (!y || !betahat || !resid || !irf) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t d = sizeof(double), K = 1 + (size_t)ns * p;
    const size_t n_y = (size_t)T * ns, n_b = K * ns, n_s = signs ? (size_t)B * T : 0, n_bo = beta_out ? (size_t)B * K * ns : 0,
                 n_irf = (size_t)B * ns * H * ns;
    double* buf = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), (2 * n_y + n_b + n_s + n_bo + n_irf) * d));
    double *y_d = buf, *b_d = y_d + n_y, *e_d = b_d + n_b, *s_d = e_d + n_y, *bo_d = s_d + n_s, *irf_d = bo_d + n_bo;
    hipMemcpyAsync(y_d, y, n_y * d, hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(b_d, betahat, n_b * d, hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(e_d, resid, n_y * d, hipMemcpyHostToDevice, h->stream);
    if (signs) hipMemcpyAsync(s_d, signs, n_s * d, hipMemcpyHostToDevice, h->stream);
    int rc = dfm_var_bootstrap_irf_dev(h, B, T, ns, p, H, y_d, b_d, e_d, signs ? s_d : nullptr, seed, first_draw,
                                       beta_out ? bo_d : nullptr, irf_d);
    if (rc == 0) {
        hipMemcpyAsync(irf, irf_d, n_irf * d, hipMemcpyDeviceToHost, h->stream);
        if (beta_out) hipMemcpyAsync(beta_out, bo_d, n_bo * d, hipMemcpyDeviceToHost, h->stream);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = hip_fail(h, e, "hipStreamSynchronize");
    }
    hipFree(buf);
    return rc;
}

int dfm_quantile_bands_dev(dfm_handle* h, int B, int S, int nq, const double* x, const double* q, double* out) {
    if (!h) return DFM_E_NULL;
    if (B < 1 || S < 1 || nq < 1 || B > 16384) return fail(h, DFM_E_DIMS, "B in 1..16384, S, nq >= 1%s");
    if (!x || !q || !out) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    HIP_TRY(h, hipSetDevice(h->device));
    QuantArgs a;
    a.B = B; a.S = S; a.nq = nq; a.x = x; a.q = q; a.out = out;
    { ProfScope ps(h, K_QUANT); HIP_TRY(h, launch_quantiles(a, h->stream)); }
    return 0;
}

int dfm_quantile_bands(dfm_handle* h, int B, int S, int nq, const double* x, const double* q, double* out) {
    if (!h) return DFM_E_NULL;
    if (B < 1 || S < 1 || nq < 1) return fail(h, DFM_E_DIMS, "B, S, nq must be >= 1%s");
    if (!x || !q || !out) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t d = sizeof(double), n_x = (size_t)B * S, n_o = (size_t)nq * S;
    double* buf = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), (n_x + nq + n_o) * d));
    double *x_d = buf, *q_d = x_d + n_x, *o_d = q_d + nq;
    hipMemcpyAsync(x_d, x, n_x * d, hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(q_d, q, (size_t)nq * d, hipMemcpyHostToDevice, h->stream);
    int rc = dfm_quantile_bands_dev(h, B, S, nq, x_d, q_d, o_d);
    if (rc == 0) {
        hipMemcpyAsync(out, o_d, n_o * d, hipMemcpyDeviceToHost, h->stream);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = hip_fail(h, e, "hipStreamSynchronize");
    }
    hipFree(buf);
    return rc;
}


// ---- Chow / QLR statistics with HAC covariance (breaks.hip) ---------------------------------------------
int dfm_chow_batch_dev(dfm_handle* h, int S, int Tmax, int k, const double* y, const double* X, const int* Tlen, int P,
                       const int* prob_series, const int* prob_break, const int* prob_q, double* chow) {
    if (!h) return DFM_E_NULL;
    if (S < 1 || Tmax < 1 || k < 1 || P < 1) return fail(h, DFM_E_DIMS, "S, Tmax, k, P must be >= 1%s");
    if (k > 8) return fail(h, DFM_E_R_UNSUPPORTED, "k > 8 regressors%s");
    if (!y || !X || !Tlen || !prob_series || !prob_break || !prob_q || !chow)
        return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    HIP_TRY(h, hipSetDevice(h->device));
    ChowArgs a;
    a.S = S; a.Tmax = Tmax; a.k = k; a.P = P; a.y = y; a.X = X; a.Tlen = Tlen; a.prob_series = prob_series;
    a.prob_break = prob_break; a.prob_q = prob_q; a.chow = chow;
    { ProfScope ps(h, K_CHOW); HIP_TRY(h, launch_chow(a, h->stream)); }
    return 0;
}

int dfm_chow_batch(dfm_handle* h, int S, int Tmax, int k, const double* y, const double* X, const int* Tlen, int P,
                   const int* prob_series, const int* prob_break, const int* prob_q, double* chow) {
    if (!h) return DFM_E_NULL;
    if (S < 1 || Tmax < 1 || k < 1 || P < 1) return fail(h, DFM_E_DIMS, "S, Tmax, k, P must be >= 1%s");
    if (!y || !X || !Tlen || !prob_series || !prob_break || !prob_q || !chow)
        return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    for (int p = 0; p < P; ++p) {
        const int s = prob_series[p];
        if (s < 0 || s >= S || prob_q[p] < 0 || prob_q[p] > 15 || Tlen[s] < 1 || Tlen[s] > Tmax || prob_break[p] < 0 ||
            prob_break[p] > Tlen[s])
            return fail(h, DFM_E_DIMS, "problem list: series, break date or bandwidth out of range%s");
    }
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t d = sizeof(double), n_y = (size_t)S * Tmax, n_X = n_y * k;
    char* buf = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&buf), (n_y + n_X + P) * d + ((size_t)S + 3 * (size_t)P) * sizeof(int) + 64));
    double* y_d = reinterpret_cast<double*>(buf);
    double* X_d = y_d + n_y; double* c_d = X_d + n_X;
    int* T_d = reinterpret_cast<int*>(c_d + P);
    int* ps_d = T_d + S; int* pb_d = ps_d + P; int* pq_d = pb_d + P;
    hipMemcpyAsync(y_d, y, n_y * d, hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(X_d, X, n_X * d, hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(T_d, Tlen, (size_t)S * sizeof(int), hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(ps_d, prob_series, (size_t)P * sizeof(int), hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(pb_d, prob_break, (size_t)P * sizeof(int), hipMemcpyHostToDevice, h->stream);
    hipMemcpyAsync(pq_d, prob_q, (size_t)P * sizeof(int), hipMemcpyHostToDevice, h->stream);
    int rc = dfm_chow_batch_dev(h, S, Tmax, k, y_d, X_d, T_d, P, ps_d, pb_d, pq_d, c_d);
    if (rc == 0) {
        hipMemcpyAsync(chow, c_d, (size_t)P * d, hipMemcpyDeviceToHost, h->stream);
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) rc = hip_fail(h, e, "hipStreamSynchronize");
    }
    hipFree(buf);
    return rc;
}


int dfm_standardize_batch_dev(dfm_handle* h, int B, int T, int N, double* panel, double* mean, double* sd) {
    if (!h) return DFM_E_NULL;
    if (B < 1 || T < 1 || N < 1) return fail(h, DFM_E_DIMS, "B, T, N must be >= 1%s");
    if (!panel) return fail(h, DFM_E_NULL, "required pointer is NULL%s");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, launch_standardize(B, T, N, panel, mean, sd, h->stream));
    return 0;
}
