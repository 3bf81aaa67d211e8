// multi.hip -- replicate sharding over the GPUs of one node INSIDE one process: the library-owned
// multi-GPU object a Julia host -- which has no torch.distributed -- binds to run the smoother pass / the EM loop on 1..8
// MI355X.
//
// Replicates are independent (own panel, own parameters, own EM trajectory): GPU g of G owns the contiguous block
// [g B / G, (g+1) B / G) -- the partition of dynamic_factor_models_amd/shard.py replicate_range -- and no data-path
// collective exists.  The one exchange north_star prescribes is an all-gather of the per-replicate {log-likelihood,
// still-iterating} pairs at the end of every EM iteration, so that every GPU's host thread sees the GLOBAL convergence state
// and all of them stop at the same iteration.
//
// dfm_multi (round 3) owns, for its whole life: one dfm_handle + one stream per GPU, ONE RCCL communicator
// (ncclCommInitAll at creation -- not per call), and the RESIDENT job: every GPU's shard of panels, parameters and outputs
// stays in its HBM between calls, whether it was uploaded (dfm_multi_load) or generated where it lives (dfm_multi_synth:
// BASELINE configs[2] is 52 GB of panels that never cross PCIe).  A call runs one host thread per GPU (GPU 0 on the calling
// thread); ncclAllGather is issued on each GPU's own stream.
//
// RCCL is bound lazily (dlopen): single-GPU users of libdfmhip.so never need it, and a process that has already loaded an
// RCCL (PyTorch ships one under the same SONAME) shares that copy instead of getting a second one.  This file is a client
// of the C-ABI in include/dfm_hip.h (dfm_create, dfm_em_iterate_batch_dev, ...) and of the HIP runtime; its only kernel
// packs the exchange buffer.  The reference has no counterpart (single-threaded Julia on one CPU core:
// dfm_functions.ipynb:530-543; the slot it fills is the empty `Parametric` tag, :21-23).
#include "../../include/dfm_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---- the RCCL entry points used, bound at first use (rccl.h: ncclCommInitAll, ncclAllGather, ...) ----------------
typedef struct ncclComm* comm_t;
typedef int (*fn_comm_init_all)(comm_t*, int, const int*);
typedef int (*fn_comm_destroy)(comm_t);
typedef int (*fn_comm_abort)(comm_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, comm_t, hipStream_t);
typedef const char* (*fn_error_string)(int);
constexpr int kNcclDouble = 8;   // ncclFloat64 (rccl.h)

struct Rccl {
    void* so = nullptr;
    fn_comm_init_all comm_init_all = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_comm_abort comm_abort = nullptr;          // optional
    fn_all_gather all_gather = nullptr;
    fn_error_string error_string = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::string first_error;
        for (const char* n : names) {
            r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
            const char* e = dlerror();                        // ONE call: it returns the message and clears it
            if (first_error.empty()) first_error = e ? e : "?";
        }
        if (!r.so) { r.why = "cannot load RCCL: " + first_error; return; }
        r.comm_init_all = (fn_comm_init_all)dlsym(r.so, "ncclCommInitAll");
        r.comm_destroy = (fn_comm_destroy)dlsym(r.so, "ncclCommDestroy");
        r.comm_abort = (fn_comm_abort)dlsym(r.so, "ncclCommAbort");
        r.all_gather = (fn_all_gather)dlsym(r.so, "ncclAllGather");
        r.error_string = (fn_error_string)dlsym(r.so, "ncclGetErrorString");
        if (!r.comm_init_all || !r.comm_destroy || !r.all_gather || !r.error_string) {
            r.why = "RCCL loaded but ncclCommInitAll / ncclCommDestroy / ncclAllGather / ncclGetErrorString missing";
            r.so = nullptr;
        }
    });
    return r;
}

void set_err(char* err, int cap, const std::string& s) {
    if (err && cap > 0) { strncpy(err, s.c_str(), (size_t)cap - 1); err[cap - 1] = 0; }
}

struct Range { int lo, hi; };
Range shard_of(int B, int G, int g) { return Range{(int)((long long)B * g / G), (int)((long long)B * (g + 1) / G)}; }

// a device array that grows on demand and lives as long as the object
struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t need(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, n ? n : 8);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    double* d() const { return static_cast<double*>(p); }
    int* i() const { return static_cast<int*>(p); }
};

// send[b] = {loglik_path[b][k], active[b]} for this GPU's replicates (rows >= Bl of the [mx][2] block stay 0)
__global__ void pack_exchange_kernel(int Bl, int k, int max_iter, const double* ll_path, const int* active, double* send) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bl) return;
    send[2 * b] = ll_path[(size_t)b * max_iter + k];
    send[2 * b + 1] = active[b] ? 1.0 : 0.0;
}

struct Gpu {      // one GPU's part of the object
    int dev = 0;
    dfm_handle* h = nullptr;
    hipStream_t st = nullptr;
    comm_t comm = nullptr;
    // the resident job's shard
    int lo = 0, Bl = 0;
    Buf x, lam, R, A, Q, mu, P0, f, P, ll, path, it, act, send, recv;
    // per-call status
    int rc = 0;
    std::string msg;
    int ran = 0;
    int path_cap = 0;           // row length (iterations) the resident loglik_path buffer was sized for by the last em_loop that got that far
};

}  // namespace

struct dfm_multi {
    int G = 0;
    std::vector<Gpu> gpu;
    bool has_comm = false;
    // resident job
    bool loaded = false;
    int B = 0, T = 0, N = 0, r = 0, mx = 0;
    int path_iters = 0;          // max_iter of the last dfm_multi_em (row length of the resident loglik_path)
    bool have_f = false, have_P = false, have_ll = false, have_path = false;
    std::atomic<int> poisoned{0};  // a rank aborted its communicator (em_loop): the collective object is gone, every further call is refused
    std::atomic<int> failed{0};
    char err[640] = {0};
};

namespace {

int fail(dfm_multi* m, int rc, const std::string& s) {
    if (m) { strncpy(m->err, s.c_str(), sizeof(m->err) - 1); m->err[sizeof(m->err) - 1] = 0; }
    return rc;
}

template <class F>
void on_all(dfm_multi* m, F fn) {                              // GPU 0 on the calling thread, one more thread per further GPU
    std::vector<std::thread> th;
    for (int g = 1; g < m->G; ++g) th.emplace_back([&fn, g] { fn(g); });
    fn(0);
    for (auto& t : th) t.join();
}

void clear_status(dfm_multi* m) {
    m->failed.store(0);
    for (auto& g : m->gpu) { g.rc = 0; g.msg.clear(); g.ran = 0; }
}
void bail(dfm_multi* m, Gpu& g, int rc, const std::string& s) {
    if (g.rc == 0) { g.rc = rc; g.msg = s; }
    m->failed.store(1);
}
int join_status(dfm_multi* m) {
    for (int g = 0; g < m->G; ++g)
        if (m->gpu[g].rc != 0) {
            char buf[600];
            snprintf(buf, sizeof(buf), "GPU %d (device %d): %s", g, m->gpu[g].dev, m->gpu[g].msg.c_str());
            return fail(m, m->gpu[g].rc, buf);
        }
    return 0;
}

int check_devices(int ngpu, const int* device_ids, std::vector<int>& dev, std::string& why) {
    if (ngpu < 1 || ngpu > 64) { why = "ngpu must be in 1..64"; return DFM_E_DIMS; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { why = "no HIP device"; return DFM_E_NO_DEVICE; }
    dev.resize(ngpu);
    for (int g = 0; g < ngpu; ++g) {
        dev[g] = device_ids ? device_ids[g] : g;
        if (dev[g] < 0 || dev[g] >= ndev) { why = "device id out of range"; return DFM_E_DIMS; }
        for (int q = 0; q < g; ++q)
            if (dev[q] == dev[g]) { why = "device ids must be distinct"; return DFM_E_DIMS; }
    }
    return 0;
}

int check_shape(dfm_multi* m, int B, int T, int N, int r) {
    if (B < 1 || T < 1 || N < 1 || r < 1) return fail(m, DFM_E_DIMS, "B, T, N, r must be >= 1");
    if (r > DFM_MAX_R) return fail(m, DFM_E_R_UNSUPPORTED, "r > DFM_MAX_R (32)");
    return 0;
}

// (re)allocate the shard of a (B, T, N, r) job on GPU g: inputs and the always-present outputs
void shard_alloc(dfm_multi* m, int gi) {
    Gpu& g = m->gpu[gi];
    if (hipSetDevice(g.dev) != hipSuccess) return bail(m, g, DFM_E_NO_DEVICE, "hipSetDevice failed");
    const Range s = shard_of(m->B, m->G, gi);
    g.lo = s.lo; g.Bl = s.hi - s.lo;
    const size_t Bl = (size_t)g.Bl, T = m->T, N = m->N, r = m->r, d = sizeof(double);
    hipError_t e = hipSuccess;
    auto need = [&](Buf& b, size_t bytes) { if (e == hipSuccess) e = b.need(bytes); };
    need(g.x, Bl * T * N * d); need(g.lam, Bl * N * r * d); need(g.R, Bl * N * d); need(g.A, Bl * r * r * d);
    need(g.Q, Bl * r * r * d); need(g.mu, Bl * r * d); need(g.P0, Bl * r * r * d); need(g.f, Bl * T * r * d);
    need(g.ll, Bl * d); need(g.it, (Bl + 1) * sizeof(int)); need(g.act, (Bl + 1) * sizeof(int));
    need(g.send, (size_t)2 * m->mx * d); need(g.recv, (size_t)2 * m->mx * m->G * d);
    if (e != hipSuccess) return bail(m, g, DFM_E_DIMS, std::string("hipMalloc of the shard failed: ") + hipGetErrorString(e));
}

int begin_job(dfm_multi* m, int B, int T, int N, int r) {
    if (!m) return DFM_E_NULL;
    if (int rc = check_shape(m, B, T, N, r)) return rc;
    clear_status(m);
    m->loaded = false;
    m->B = B; m->T = T; m->N = N; m->r = r;
    m->mx = 0;
    for (int g = 0; g < m->G; ++g) { const Range s = shard_of(B, m->G, g); if (s.hi - s.lo > m->mx) m->mx = s.hi - s.lo; }
    m->have_f = m->have_P = m->have_ll = m->have_path = false;
    on_all(m, [&](int g) { shard_alloc(m, g); });
    return join_status(m);
}

// The EM loop of one GPU.  EVERY thread takes part in every all-gather or none does: a thread whose iteration fails keeps
// exchanging (its replicates marked inactive) until the global stop; a thread whose STREAM fails aborts its communicator
// so that the peers' collectives return instead of waiting for it for ever.
void em_loop(dfm_multi* m, int gi, int max_iter, double tol, bool want_smooth, bool want_P, unsigned flags) {
    Gpu& g = m->gpu[gi];
    if (hipSetDevice(g.dev) != hipSuccess) bail(m, g, DFM_E_NO_DEVICE, "hipSetDevice failed");
    Rccl& Rc = rccl();
    const size_t d = sizeof(double), np = (size_t)m->r * (m->r + 1) / 2;
    if (g.rc == 0) {
        hipError_t e = g.path.need((size_t)(g.Bl > 0 ? g.Bl : 1) * max_iter * d);
        if (e == hipSuccess && want_smooth && want_P) e = g.P.need((size_t)g.Bl * m->T * np * d);
        if (e != hipSuccess) bail(m, g, DFM_E_DIMS, std::string("hipMalloc failed: ") + hipGetErrorString(e));
        else g.path_cap = max_iter;
    }
    std::vector<double> gathered((size_t)2 * m->mx * m->G);
    const bool exchange = m->has_comm || tol > 0.0;          // one GPU, no communicator, no stopping rule: nothing to agree on
    int ran = 0;
    for (int it = 0; it < max_iter; ++it) {
        if (g.rc == 0 && g.Bl > 0) {
            const int rc = dfm_em_iterate_batch_dev(g.h, g.Bl, m->T, m->N, m->r, g.x.d(), g.lam.d(), g.R.d(), g.A.d(), g.Q.d(),
                                                    g.mu.d(), g.P0.d(), it, max_iter, tol, g.path.d(), g.it.i(), g.act.i(),
                                                    want_smooth ? g.f.d() : nullptr, (want_smooth && want_P) ? g.P.d() : nullptr, flags);
            if (rc != 0) bail(m, g, rc, dfm_last_error(g.h));
        }
        ++ran;
        if (!exchange) continue;
        if (g.rc == 0 && g.Bl > 0)
            hipLaunchKernelGGL(pack_exchange_kernel, dim3((g.Bl + 255) / 256), dim3(256), 0, g.st, g.Bl, it, max_iter, g.path.d(),
                               g.act.i(), g.send.d());
        else
            (void)hipMemsetAsync(g.send.d(), 0, (size_t)2 * m->mx * d, g.st);
        const double* src = g.send.d();
        if (m->has_comm) {   // THE exchange: {loglik, active} of every replicate of the job, on this GPU's stream over xGMI
            const int e = Rc.all_gather(g.send.d(), g.recv.d(), (size_t)2 * m->mx, kNcclDouble, g.comm, g.st);
            if (e != 0) bail(m, g, DFM_E_COMM, std::string("ncclAllGather: ") + Rc.error_string(e));
            src = g.recv.d();
        }
        (void)hipMemcpyAsync(gathered.data(), src, (size_t)2 * m->mx * m->G * d, hipMemcpyDeviceToHost, g.st);
        if (hipStreamSynchronize(g.st) != hipSuccess) {
            bail(m, g, DFM_E_COMM, "stream failed during the exchange");
            // the peers' collectives return once this rank's communicator is aborted; the object is unusable afterwards
            // (dfm_multi_create refuses to build a communicator without ncclCommAbort, so the pointer is there)
            if (m->has_comm && g.comm) { (void)Rc.comm_abort(g.comm); g.comm = nullptr; m->poisoned.store(1); }
            break;
        }
        bool any = false;
        for (int q = 0; q < m->G; ++q) {
            const Range t = shard_of(m->B, m->G, q);
            for (int b = 0; b < t.hi - t.lo; ++b) any = any || gathered[((size_t)q * m->mx + b) * 2 + 1] != 0.0;
        }
        if (tol > 0.0 && !any) break;                        // identical data on every thread -> identical decision
    }
    g.ran = ran;
    if (g.rc == 0 && g.Bl > 0) {
        // the status word of the last E-step + first-iteration log-likelihoods: what dfm_em_batch checks on the host
        const int rc = dfm_synchronize(g.h);
        if (rc != 0) return bail(m, g, rc, dfm_last_error(g.h));
        std::vector<double> first((size_t)g.Bl);
        if (hipMemcpy2D(first.data(), d, g.path.d(), (size_t)max_iter * d, d, (size_t)g.Bl, hipMemcpyDeviceToHost) != hipSuccess)
            return bail(m, g, DFM_E_NUMERIC, "download of the log-likelihoods failed");
        for (int b = 0; b < g.Bl; ++b)
            if (!isfinite(first[b])) return bail(m, g, DFM_E_NUMERIC, "non-finite log-likelihood (Q or P0 not positive definite?)");
    }
}

void pass_one(dfm_multi* m, int gi, bool want_P, unsigned flags) {
    Gpu& g = m->gpu[gi];
    if (g.Bl <= 0) return;
    if (hipSetDevice(g.dev) != hipSuccess) return bail(m, g, DFM_E_NO_DEVICE, "hipSetDevice failed");
    const size_t d = sizeof(double), np = (size_t)m->r * (m->r + 1) / 2;
    if (want_P && g.P.need((size_t)g.Bl * m->T * np * d) != hipSuccess) return bail(m, g, DFM_E_DIMS, "hipMalloc of P_smooth failed");
    int rc = dfm_ks_pass_batch_dev(g.h, g.Bl, m->T, m->N, m->r, g.x.d(), g.lam.d(), g.R.d(), g.A.d(), g.Q.d(), g.mu.d(), g.P0.d(),
                                   g.f.d(), want_P ? g.P.d() : nullptr, g.ll.d(), flags);
    if (rc == 0) rc = dfm_synchronize(g.h);                    // (+ the status word: NaN on the balanced path, expired waits)
    if (rc != 0) return bail(m, g, rc, dfm_last_error(g.h));
    std::vector<double> ll((size_t)g.Bl);
    if (hipMemcpy(ll.data(), g.ll.d(), (size_t)g.Bl * d, hipMemcpyDeviceToHost) != hipSuccess)
        return bail(m, g, DFM_E_NUMERIC, "download of the log-likelihoods failed");
    for (int b = 0; b < g.Bl; ++b)
        if (!isfinite(ll[b])) return bail(m, g, DFM_E_NUMERIC, "non-finite log-likelihood (Q or P0 not positive definite?)");
}

}  // namespace

extern "C" {

int dfm_multi_create(dfm_multi** out, int ngpu, const int* device_ids, unsigned mflags, char* err, int err_cap) {
    set_err(err, err_cap, "");
    if (!out) return DFM_E_NULL;
    *out = nullptr;
    std::vector<int> dev;
    std::string why;
    if (int rc = check_devices(ngpu, device_ids, dev, why)) { set_err(err, err_cap, why); return rc; }
    dfm_multi* m = new (std::nothrow) dfm_multi();
    if (!m) return DFM_E_NULL;
    m->G = ngpu;
    m->gpu.resize(ngpu);
    for (int g = 0; g < ngpu; ++g) m->gpu[g].dev = dev[g];
    on_all(m, [&](int gi) {
        Gpu& g = m->gpu[gi];
        if (hipSetDevice(g.dev) != hipSuccess) return bail(m, g, DFM_E_NO_DEVICE, "hipSetDevice failed");
        if (hipStreamCreateWithFlags(&g.st, hipStreamNonBlocking) != hipSuccess) return bail(m, g, DFM_E_NO_DEVICE, "hipStreamCreate failed");
        const int rc = dfm_create(&g.h, g.dev, g.st);         // the handle launches on the stream the collectives use
        if (rc != 0) return bail(m, g, rc, "dfm_create failed");
    });
    int rc = join_status(m);
#ifdef DFM_DIAG
    const char* env = getenv("DFM_MULTI_FORCE_COMM");      // (diagnostics build only; callers use DFM_MULTI_F_FORCE_COMM)
#else
    const char* env = nullptr;
#endif
    const bool want_comm = ngpu > 1 || (mflags & DFM_MULTI_F_FORCE_COMM) != 0 || (env && atoi(env) != 0);
    if (rc == 0 && want_comm) {
        Rccl& R = rccl();
        if (!R.so) rc = fail(m, DFM_E_COMM, R.why);
        else if (!R.comm_abort) rc = fail(m, DFM_E_COMM, "librccl has no ncclCommAbort: a failed rank could not release its peers");
        else {
            std::vector<comm_t> comms(ngpu, nullptr);
            const int e = R.comm_init_all(comms.data(), ngpu, dev.data());
            if (e != 0) rc = fail(m, DFM_E_COMM, std::string("ncclCommInitAll: ") + R.error_string(e));
            else {
                for (int g = 0; g < ngpu; ++g) m->gpu[g].comm = comms[g];
                m->has_comm = true;
            }
        }
    }
    if (rc != 0) { set_err(err, err_cap, m->err); (void)dfm_multi_destroy(m); return rc; }
    *out = m;
    return 0;
}

int dfm_multi_destroy(dfm_multi* m) {
    if (!m) return 0;
    bool any_comm = m->has_comm;
    for (auto& g : m->gpu) any_comm = any_comm || g.comm != nullptr;
    Rccl* R = any_comm ? &rccl() : nullptr;                    // single-GPU objects never load RCCL, not even here
    on_all(m, [&](int gi) {
        Gpu& g = m->gpu[gi];
        (void)hipSetDevice(g.dev);
        if (g.st) (void)hipStreamSynchronize(g.st);
        if (g.comm && R && R->so) { (void)R->comm_destroy(g.comm); g.comm = nullptr; }
        for (Buf* b : {&g.x, &g.lam, &g.R, &g.A, &g.Q, &g.mu, &g.P0, &g.f, &g.P, &g.ll, &g.path, &g.it, &g.act, &g.send, &g.recv}) b->release();
        if (g.h) { (void)dfm_destroy(g.h); g.h = nullptr; }
        if (g.st) { (void)hipStreamDestroy(g.st); g.st = nullptr; }
    });
    delete m;
    return 0;
}

int dfm_multi_ngpu(const dfm_multi* m) { return m ? m->G : 0; }
int dfm_multi_has_comm(const dfm_multi* m) { return (m && m->has_comm) ? 1 : 0; }
const char* dfm_multi_last_error(const dfm_multi* m) { return m ? m->err : "null dfm_multi"; }

int dfm_multi_load(dfm_multi* m, int B, int T, int N, int r, const double* panel, const double* Lam, const double* R,
                   const double* A, const double* Q, const double* mu0, const double* P0) {
    if (!m) return DFM_E_NULL;
    if (!panel || !Lam || !R || !A || !Q || !mu0 || !P0) return fail(m, DFM_E_NULL, "required pointer is NULL");
    if (int rc = begin_job(m, B, T, N, r)) return rc;
    on_all(m, [&](int gi) {
        Gpu& g = m->gpu[gi];
        if (g.Bl <= 0) return;
        if (hipSetDevice(g.dev) != hipSuccess) return bail(m, g, DFM_E_NO_DEVICE, "hipSetDevice failed");
        const size_t o = (size_t)g.lo, Bl = (size_t)g.Bl, d = sizeof(double), Tn = T, Nn = N, rn = r;
        auto up = [&](Buf& b, const double* src, size_t n) { (void)hipMemcpyAsync(b.p, src, n * d, hipMemcpyHostToDevice, g.st); };
        up(g.x, panel + o * Tn * Nn, Bl * Tn * Nn); up(g.lam, Lam + o * Nn * rn, Bl * Nn * rn); up(g.R, R + o * Nn, Bl * Nn);
        up(g.A, A + o * rn * rn, Bl * rn * rn); up(g.Q, Q + o * rn * rn, Bl * rn * rn); up(g.mu, mu0 + o * rn, Bl * rn);
        up(g.P0, P0 + o * rn * rn, Bl * rn * rn);
        if (hipStreamSynchronize(g.st) != hipSuccess) return bail(m, g, DFM_E_NO_DEVICE, "upload of the shard failed");
    });
    if (int rc = join_status(m)) return rc;
    m->loaded = true;
    return 0;
}

int dfm_multi_synth(dfm_multi* m, uint64_t seed, int64_t first_replicate, int B, int T, int N, int r, double missing_prob,
                    int pca_start) {
    if (!m) return DFM_E_NULL;
    if (pca_start && missing_prob > 0.0) return fail(m, DFM_E_MISSING, "the PCA start needs balanced panels (missing_prob must be 0)");
    if (int rc = begin_job(m, B, T, N, r)) return rc;
    on_all(m, [&](int gi) {
        Gpu& g = m->gpu[gi];
        if (g.Bl <= 0) return;
        if (hipSetDevice(g.dev) != hipSuccess) return bail(m, g, DFM_E_NO_DEVICE, "hipSetDevice failed");
        int rc = dfm_synth_panels_dev(g.h, seed, first_replicate + g.lo, g.Bl, T, N, r, missing_prob, g.x.d(), g.lam.d(), g.R.d(),
                                      g.A.d(), g.Q.d(), g.mu.d(), g.P0.d());
        if (rc == 0 && pca_start)
            rc = dfm_pca_init_batch_dev(g.h, g.Bl, T, N, r, g.x.d(), g.lam.d(), g.R.d(), g.A.d(), g.Q.d(), g.mu.d(), g.P0.d(), nullptr);
        if (rc == 0) rc = dfm_synchronize(g.h);
        if (rc != 0) return bail(m, g, rc, dfm_last_error(g.h));
    });
    if (int rc = join_status(m)) return rc;
    m->loaded = true;
    return 0;
}

int dfm_multi_ks_pass(dfm_multi* m, int want_P, unsigned flags) {
    if (!m) return DFM_E_NULL;
    if (!m->loaded) return fail(m, DFM_E_NULL, "no resident job: call dfm_multi_load or dfm_multi_synth first");
    if (m->poisoned.load()) return fail(m, DFM_E_COMM, "this object's communicator was aborted by an earlier failure: destroy it and create a new one");
    clear_status(m);
    on_all(m, [&](int g) { pass_one(m, g, want_P != 0, flags); });
    if (int rc = join_status(m)) return rc;
    m->have_f = true; m->have_P = want_P != 0; m->have_ll = true;
    return 0;
}

int dfm_multi_em(dfm_multi* m, int max_iter, double tol, int want_smooth, int want_P, unsigned flags, int* iterations_run) {
    if (!m) return DFM_E_NULL;
    if (!m->loaded) return fail(m, DFM_E_NULL, "no resident job: call dfm_multi_load or dfm_multi_synth first");
    if (max_iter < 1) return fail(m, DFM_E_DIMS, "max_iter must be >= 1");
    if (m->poisoned.load()) return fail(m, DFM_E_COMM, "this object's communicator was aborted by an earlier failure: destroy it and create a new one");
    clear_status(m);
    m->have_path = false;
    for (auto& g : m->gpu) g.path_cap = 0;
    on_all(m, [&](int g) { em_loop(m, g, max_iter, tol, want_smooth != 0, want_P != 0, flags); });
    if (iterations_run) *iterations_run = m->gpu[0].ran;
    m->path_iters = max_iter;
    // (also after a failure: the path says which replicate went wrong) -- but only if EVERY GPU's buffer holds rows of max_iter:
    // a GPU that bailed before its allocation may still carry the shorter buffer of an earlier call
    m->have_path = true;
    for (auto& g : m->gpu) if (g.Bl > 0 && g.path_cap < max_iter) m->have_path = false;
    if (int rc = join_status(m)) return rc;
    if (want_smooth) { m->have_f = true; m->have_P = want_P != 0; }
    return 0;
}

int dfm_multi_fetch(dfm_multi* m, int what, void* dst) {
    if (!m) return DFM_E_NULL;
    if (!dst) return fail(m, DFM_E_NULL, "dst is NULL");
    if (!m->loaded) return fail(m, DFM_E_NULL, "no resident job");
    const size_t T = m->T, N = m->N, r = m->r, np = r * (r + 1) / 2;
    size_t per = 0, el = sizeof(double);
    bool have = true;
    switch (what) {
        case DFM_MULTI_LAM: per = N * r; break;
        case DFM_MULTI_R: per = N; break;
        case DFM_MULTI_A: case DFM_MULTI_Q: case DFM_MULTI_P0: per = r * r; break;
        case DFM_MULTI_MU0: per = r; break;
        case DFM_MULTI_F_SMOOTH: per = T * r; have = m->have_f; break;
        case DFM_MULTI_P_SMOOTH: per = T * np; have = m->have_P; break;
        case DFM_MULTI_LOGLIK: per = 1; have = m->have_ll; break;
        case DFM_MULTI_LOGLIK_PATH: per = (size_t)m->path_iters; have = m->have_path; break;
        case DFM_MULTI_ITERS: per = 1; el = sizeof(int); have = m->have_path; break;
        case DFM_MULTI_PANEL: per = T * N; break;
        default: return fail(m, DFM_E_DIMS, "dfm_multi_fetch: unknown array");
    }
    if (!have) return fail(m, DFM_E_NULL, "dfm_multi_fetch: the resident job does not hold this array (no call has produced it)");
    for (int gi = 0; gi < m->G; ++gi) {
        Gpu& g = m->gpu[gi];
        if (g.Bl <= 0) continue;
        const Buf* b = nullptr;
        switch (what) {
            case DFM_MULTI_LAM: b = &g.lam; break;   case DFM_MULTI_R: b = &g.R; break;     case DFM_MULTI_A: b = &g.A; break;
            case DFM_MULTI_Q: b = &g.Q; break;       case DFM_MULTI_MU0: b = &g.mu; break;  case DFM_MULTI_P0: b = &g.P0; break;
            case DFM_MULTI_F_SMOOTH: b = &g.f; break; case DFM_MULTI_P_SMOOTH: b = &g.P; break; case DFM_MULTI_LOGLIK: b = &g.ll; break;
            case DFM_MULTI_LOGLIK_PATH: b = &g.path; break; case DFM_MULTI_ITERS: b = &g.it; break; default: b = &g.x; break;
        }
        if (!b->p) return fail(m, DFM_E_NULL, "dfm_multi_fetch: array not allocated on some GPU");
        if (hipSetDevice(g.dev) != hipSuccess) return fail(m, DFM_E_NO_DEVICE, "hipSetDevice failed");
        (void)hipStreamSynchronize(g.st);
        const hipError_t e = hipMemcpy(static_cast<char*>(dst) + (size_t)g.lo * per * el, b->p, (size_t)g.Bl * per * el, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(m, (int)e, std::string("dfm_multi_fetch: ") + hipGetErrorString(e));
    }
    return 0;
}

// ---- handle-less forms: one object per call -------------------------------------------------------------------------
int dfm_em_batch_multi(int ngpu, const int* device_ids, int B, int T, int N, int r, const double* panel, double* Lam,
                       double* R, double* A, double* Q, double* mu0, double* P0, int max_iter, double tol,
                       double* loglik_path, int* iters, double* f_smooth, double* P_smooth, unsigned flags,
                       int* iterations_run, char* err, int err_cap) {
    set_err(err, err_cap, "");
    if (B < 1 || T < 1 || N < 1 || r < 1) { set_err(err, err_cap, "B, T, N, r must be >= 1"); return DFM_E_DIMS; }
    if (r > DFM_MAX_R) { set_err(err, err_cap, "r > DFM_MAX_R (32)"); return DFM_E_R_UNSUPPORTED; }
    if (!panel || !Lam || !R || !A || !Q || !mu0 || !P0 || !loglik_path || !iters) { set_err(err, err_cap, "required pointer is NULL"); return DFM_E_NULL; }
    if (max_iter < 1) { set_err(err, err_cap, "max_iter must be >= 1"); return DFM_E_DIMS; }
    dfm_multi* m = nullptr;
    if (int rc = dfm_multi_create(&m, ngpu, device_ids, 0u, err, err_cap)) return rc;
    int rc = dfm_multi_load(m, B, T, N, r, panel, Lam, R, A, Q, mu0, P0);
    if (rc == 0) rc = dfm_multi_em(m, max_iter, tol, f_smooth != nullptr, P_smooth != nullptr, flags, iterations_run);
    const int rc_em = rc;
    if (rc == 0 || rc == DFM_E_NUMERIC) {
        // (a numeric failure still returns the paths: they say which replicate went wrong)
        int rf = dfm_multi_fetch(m, DFM_MULTI_LOGLIK_PATH, loglik_path);
        if (rf == 0) rf = dfm_multi_fetch(m, DFM_MULTI_ITERS, iters);
        if (rc == 0) rc = rf;
    }
    if (rc == 0) {
        const int what[] = {DFM_MULTI_LAM, DFM_MULTI_R, DFM_MULTI_A, DFM_MULTI_Q, DFM_MULTI_MU0, DFM_MULTI_P0};
        double* dst[] = {Lam, R, A, Q, mu0, P0};
        for (int k = 0; rc == 0 && k < 6; ++k) rc = dfm_multi_fetch(m, what[k], dst[k]);
        if (rc == 0 && f_smooth) rc = dfm_multi_fetch(m, DFM_MULTI_F_SMOOTH, f_smooth);
        if (rc == 0 && P_smooth) rc = dfm_multi_fetch(m, DFM_MULTI_P_SMOOTH, P_smooth);
    }
    if (rc != 0) set_err(err, err_cap, rc_em != 0 || rc != 0 ? dfm_multi_last_error(m) : "");
    (void)dfm_multi_destroy(m);
    return rc;
}

int dfm_ks_pass_batch_multi(int ngpu, const int* device_ids, int B, int T, int N, int r, const double* panel,
                            const double* Lam, const double* R, const double* A, const double* Q, const double* mu0,
                            const double* P0, double* f_smooth, double* P_smooth, double* loglik, unsigned flags,
                            char* err, int err_cap) {
    set_err(err, err_cap, "");
    if (B < 1 || T < 1 || N < 1 || r < 1) { set_err(err, err_cap, "B, T, N, r must be >= 1"); return DFM_E_DIMS; }
    if (r > DFM_MAX_R) { set_err(err, err_cap, "r > DFM_MAX_R (32)"); return DFM_E_R_UNSUPPORTED; }
    if (!panel || !Lam || !R || !A || !Q || !mu0 || !P0 || !f_smooth || !loglik) { set_err(err, err_cap, "required pointer is NULL"); return DFM_E_NULL; }
    dfm_multi* m = nullptr;
    if (int rc = dfm_multi_create(&m, ngpu, device_ids, 0u, err, err_cap)) return rc;
    int rc = dfm_multi_load(m, B, T, N, r, panel, Lam, R, A, Q, mu0, P0);
    if (rc == 0) rc = dfm_multi_ks_pass(m, P_smooth != nullptr, flags);
    if (rc == 0) rc = dfm_multi_fetch(m, DFM_MULTI_F_SMOOTH, f_smooth);
    if (rc == 0 && P_smooth) rc = dfm_multi_fetch(m, DFM_MULTI_P_SMOOTH, P_smooth);
    if (rc == 0) rc = dfm_multi_fetch(m, DFM_MULTI_LOGLIK, loglik);
    if (rc != 0) set_err(err, err_cap, dfm_multi_last_error(m));
    (void)dfm_multi_destroy(m);
    return rc;
}

}  // extern "C"
