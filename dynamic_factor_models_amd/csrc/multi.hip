// multi.hip -- replicate sharding over the GPUs of one node INSIDE one process: what a Julia
// host -- which has no torch.distributed -- binds to run the EM loop on 1..8 MI355X.
//
// Replicates are independent (own panel, own parameters, own EM trajectory): GPU g of G owns the contiguous block
// [g B / G, (g+1) B / G) -- the partition of dynamic_factor_models_amd/shard.py replicate_range -- and no data-path
// collective exists.  The one exchange north_star prescribes is an all-gather of the per-replicate {log-likelihood,
// still-iterating} pairs at the end of every EM iteration, so that every GPU's host thread sees the GLOBAL
// convergence state and all of them stop at the same iteration.  Here: one host thread per GPU, a library-owned RCCL
// communicator (ncclCommInitAll), ncclAllGather over xGMI on each GPU's own stream.
//
// RCCL is bound lazily (dlopen): single-GPU users of libdfmhip.so never need it, and a process that has already
// loaded an RCCL (PyTorch ships one under the same SONAME) shares that copy instead of getting a second one.
// This file is a client of the C-ABI in include/dfm_hip.h (dfm_create, dfm_em_iterate_batch_dev, ...) and of the
// HIP runtime; its only kernel packs the exchange buffer.  The reference has no counterpart (single-threaded Julia
// on one CPU core: dfm_functions.ipynb:530-543).
#include "../../include/dfm_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// ---- the five RCCL entry points used, bound at first use (rccl.h: ncclCommInitAll, ncclAllGather, ...) ----------
typedef struct ncclComm* comm_t;
typedef int (*fn_comm_init_all)(comm_t*, int, const int*);
typedef int (*fn_comm_destroy)(comm_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, comm_t, hipStream_t);
typedef const char* (*fn_error_string)(int);
constexpr int kNcclDouble = 8;   // ncclFloat64 (rccl.h)

struct Rccl {
    void* so = nullptr;
    fn_comm_init_all comm_init_all = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_error_string error_string = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (!r.so) { r.why = std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "?"); return; }
        r.comm_init_all = (fn_comm_init_all)dlsym(r.so, "ncclCommInitAll");
        r.comm_destroy = (fn_comm_destroy)dlsym(r.so, "ncclCommDestroy");
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

struct Shard { int lo, hi; };
Shard shard_of(int B, int G, int g) { return Shard{(int)((long long)B * g / G), (int)((long long)B * (g + 1) / G)}; }

// One GPU's share of a call: device block, uploads, downloads.  All sizes in doubles.
struct DevBlock {
    double* base = nullptr;
    double* cur = nullptr;
    hipStream_t st = nullptr;
    hipError_t alloc(size_t n_doubles) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), n_doubles * sizeof(double));
        cur = base;
        return e;
    }
    double* take(size_t n) { double* p = cur; cur += n; return p; }
    double* up(const double* src, size_t n) {
        double* p = take(n);
        if (n) (void)hipMemcpyAsync(p, src, n * sizeof(double), hipMemcpyHostToDevice, st);
        return p;
    }
    void down(void* dst, const void* src, size_t bytes) {
        if (bytes) (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
    }
    ~DevBlock() { if (base) (void)hipFree(base); }
};

struct Common {
    int G, B, T, N, r;
    std::vector<int> dev;
    std::vector<comm_t> comm;
    std::vector<int> rc;
    std::vector<std::string> msg;
    std::atomic<int> failed{0};   // set by a GPU thread that fails; checked between the phases (em_setup / em_loop)
};

int check_common(int ngpu, const int* device_ids, int B, int T, int N, int r, char* err, int cap, std::vector<int>& dev) {
    if (ngpu < 1 || ngpu > 64) { set_err(err, cap, "ngpu must be in 1..64"); return DFM_E_DIMS; }
    if (B < 1 || T < 1 || N < 1 || r < 1) { set_err(err, cap, "B, T, N, r must be >= 1"); return DFM_E_DIMS; }
    if (r > DFM_MAX_R) { set_err(err, cap, "r > DFM_MAX_R (32)"); return DFM_E_R_UNSUPPORTED; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_err(err, cap, "no HIP device"); return DFM_E_NO_DEVICE; }
    dev.resize(ngpu);
    for (int g = 0; g < ngpu; ++g) {
        dev[g] = device_ids ? device_ids[g] : g;
        if (dev[g] < 0 || dev[g] >= ndev) { set_err(err, cap, "device id out of range"); return DFM_E_DIMS; }
        for (int q = 0; q < g; ++q)
            if (dev[q] == dev[g]) { set_err(err, cap, "device ids must be distinct"); return DFM_E_DIMS; }
    }
    return 0;
}

int init_comms(Common& c, char* err, int cap) {
    Rccl& R = rccl();
    if (!R.so) { set_err(err, cap, R.why); return DFM_E_COMM; }
    c.comm.assign(c.G, nullptr);
    const int e = R.comm_init_all(c.comm.data(), c.G, c.dev.data());
    if (e != 0) { set_err(err, cap, std::string("ncclCommInitAll: ") + R.error_string(e)); return DFM_E_COMM; }
    return 0;
}
void destroy_comms(Common& c) {
    Rccl& R = rccl();
    for (comm_t cm : c.comm)
        if (cm && R.so) (void)R.comm_destroy(cm);
    c.comm.clear();
}

struct EmCall {
    const double* panel; double *Lam, *R, *A, *Q, *mu0, *P0;
    int max_iter; double tol; double* loglik_path; int* iters; double *f_smooth, *P_smooth; unsigned flags;
    std::vector<int> iters_run;   // per GPU: EM iterations launched (identical on every GPU by construction)
};

// send[b] = {loglik_path[b][k], active[b]} for this GPU's replicates (rows >= Bl of the [mx][2] block stay 0)
__global__ void pack_exchange_kernel(int Bl, int k, int max_iter, const double* ll_path, const int* active, double* send) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bl) return;
    send[2 * b] = ll_path[(size_t)b * max_iter + k];
    send[2 * b + 1] = active[b] ? 1.0 : 0.0;
}

struct GpuState {    // one GPU's share of a dfm_em_batch_multi call
    dfm_handle* h = nullptr;
    DevBlock d;
    int Bl = 0, mx = 0;
    double *x_d = nullptr, *lam_d = nullptr, *R_d = nullptr, *A_d = nullptr, *Q_d = nullptr, *mu_d = nullptr, *P0_d = nullptr,
           *f_d = nullptr, *P_d = nullptr, *ll_d = nullptr, *send = nullptr, *recv = nullptr;
    int *it_d = nullptr, *act_d = nullptr;
};

// phase 1 (one thread per GPU, joined before phase 2): handle, stream, device block, uploads.  A GPU that fails here
// never reaches a collective: phase 2 is skipped on every GPU.
void em_setup(Common& c, EmCall& k, int g, GpuState& s) {
    auto bail = [&](int rc, const std::string& m) { c.rc[g] = rc; c.msg[g] = m; c.failed.store(1); };
    const Shard sh = shard_of(c.B, c.G, g);
    s.Bl = sh.hi - sh.lo;
    for (int q = 0; q < c.G; ++q) { const Shard t = shard_of(c.B, c.G, q); if (t.hi - t.lo > s.mx) s.mx = t.hi - t.lo; }
    if (hipSetDevice(c.dev[g]) != hipSuccess) return bail(DFM_E_NO_DEVICE, "hipSetDevice failed");
    int rc = dfm_create(&s.h, c.dev[g], nullptr);
    if (rc != 0) return bail(rc, "dfm_create failed");
    if (hipStreamCreateWithFlags(&s.d.st, hipStreamNonBlocking) != hipSuccess) return bail(DFM_E_NO_DEVICE, "hipStreamCreate failed");
    (void)dfm_set_stream(s.h, s.d.st);
    const size_t Bl = (size_t)s.Bl, T = c.T, N = c.N, r = c.r, np = r * (r + 1) / 2, mi = (size_t)k.max_iter;
    const size_t n_panel = Bl * T * N, n_lam = Bl * N * r, n_R = Bl * N, n_m = Bl * r * r, n_v = Bl * r, n_f = Bl * T * r,
                 n_P = Bl * T * np, n_ll = Bl * mi;
    const size_t n_int = (2 * (Bl + 1) * sizeof(int) + 7) / 8;                        // iters, active
    const size_t n_x = (size_t)2 * s.mx * (1 + c.G);                                   // send [mx][2], recv [G][mx][2]
    if (s.d.alloc(n_panel + n_lam + n_R + 3 * n_m + n_v + n_f + n_P + n_ll + n_int + n_x + 16) != hipSuccess)
        return bail(DFM_E_DIMS, "hipMalloc of the shard failed");
    const size_t o = (size_t)sh.lo;
    DevBlock& d = s.d;
    s.x_d = d.up(k.panel + o * T * N, n_panel); s.lam_d = d.up(k.Lam + o * N * r, n_lam); s.R_d = d.up(k.R + o * N, n_R);
    s.A_d = d.up(k.A + o * r * r, n_m); s.Q_d = d.up(k.Q + o * r * r, n_m); s.mu_d = d.up(k.mu0 + o * r, n_v);
    s.P0_d = d.up(k.P0 + o * r * r, n_m);
    s.f_d = d.take(n_f); s.P_d = d.take(n_P); s.ll_d = d.take(n_ll);
    s.it_d = reinterpret_cast<int*>(d.take(n_int)); s.act_d = s.it_d + (Bl + 1);
    s.send = d.take((size_t)2 * s.mx); s.recv = d.take((size_t)2 * s.mx * c.G);
    (void)hipMemsetAsync(s.send, 0, (size_t)2 * s.mx * sizeof(double), d.st);
    if (hipStreamSynchronize(d.st) != hipSuccess) return bail(DFM_E_NO_DEVICE, "upload of the shard failed");
}

// phase 2: the EM loop.  EVERY thread takes part in every all-gather or none does: a thread whose iteration fails
// keeps exchanging (with its replicates marked inactive) until the global stop.
void em_loop(Common& c, EmCall& k, int g, GpuState& s) {
    auto bail = [&](int rc, const std::string& m) { if (c.rc[g] == 0) { c.rc[g] = rc; c.msg[g] = m; } c.failed.store(1); };
    if (hipSetDevice(c.dev[g]) != hipSuccess) bail(DFM_E_NO_DEVICE, "hipSetDevice failed");
    Rccl& Rc = rccl();
    DevBlock& d = s.d;
    const size_t mi = (size_t)k.max_iter;
    std::vector<double> gathered((size_t)2 * s.mx * c.G);
    int ran = 0;
    for (int it = 0; it < k.max_iter; ++it) {
        if (c.rc[g] == 0 && s.Bl > 0) {
            const int rc = dfm_em_iterate_batch_dev(s.h, s.Bl, c.T, c.N, c.r, s.x_d, s.lam_d, s.R_d, s.A_d, s.Q_d, s.mu_d, s.P0_d, it,
                                                    k.max_iter, k.tol, s.ll_d, s.it_d, s.act_d, k.f_smooth ? s.f_d : nullptr,
                                                    k.P_smooth ? s.P_d : nullptr, k.flags);
            if (rc != 0) bail(rc, dfm_last_error(s.h));
        }
        ++ran;
        if (c.G == 1 && !(k.tol > 0.0)) continue;             // one GPU, no stopping rule: nothing to agree on
        if (c.rc[g] == 0 && s.Bl > 0)
            hipLaunchKernelGGL(pack_exchange_kernel, dim3((s.Bl + 255) / 256), dim3(256), 0, d.st, s.Bl, it, k.max_iter, s.ll_d,
                               s.act_d, s.send);
        else
            (void)hipMemsetAsync(s.send, 0, (size_t)2 * s.mx * sizeof(double), d.st);
        const double* src = s.send;
        if (c.G > 1) {   // the exchange: {loglik, active} of every replicate of the job, on this GPU's stream over xGMI
            const int e = Rc.all_gather(s.send, s.recv, (size_t)2 * s.mx, kNcclDouble, c.comm[g], d.st);
            if (e != 0) bail(DFM_E_COMM, std::string("ncclAllGather: ") + Rc.error_string(e));
            src = s.recv;
        }
        (void)hipMemcpyAsync(gathered.data(), src, (size_t)2 * s.mx * c.G * sizeof(double), hipMemcpyDeviceToHost, d.st);
        if (hipStreamSynchronize(d.st) != hipSuccess) { bail(DFM_E_COMM, "stream failed during the exchange"); break; }
        bool any = false;
        for (int q = 0; q < c.G; ++q) {
            const Shard t = shard_of(c.B, c.G, q);
            for (int b = 0; b < t.hi - t.lo; ++b) any = any || gathered[((size_t)q * s.mx + b) * 2 + 1] != 0.0;
        }
        if (k.tol > 0.0 && !any) break;                        // identical data on every thread -> identical decision
    }
    k.iters_run[g] = ran;
    if (c.rc[g] == 0 && s.Bl > 0) {
        const Shard sh = shard_of(c.B, c.G, g);
        const size_t Bl = (size_t)s.Bl, T = c.T, N = c.N, r = c.r, np = r * (r + 1) / 2, o = (size_t)sh.lo;
        d.down(k.Lam + o * N * r, s.lam_d, Bl * N * r * 8); d.down(k.R + o * N, s.R_d, Bl * N * 8);
        d.down(k.A + o * r * r, s.A_d, Bl * r * r * 8); d.down(k.Q + o * r * r, s.Q_d, Bl * r * r * 8);
        d.down(k.mu0 + o * r, s.mu_d, Bl * r * 8); d.down(k.P0 + o * r * r, s.P0_d, Bl * r * r * 8);
        d.down(k.loglik_path + o * mi, s.ll_d, Bl * mi * 8); d.down(k.iters + o, s.it_d, Bl * sizeof(int));
        if (k.f_smooth) d.down(k.f_smooth + o * T * r, s.f_d, Bl * T * r * 8);
        if (k.P_smooth) d.down(k.P_smooth + o * T * np, s.P_d, Bl * T * np * 8);
        if (hipStreamSynchronize(d.st) != hipSuccess) bail(DFM_E_NUMERIC, "hipStreamSynchronize failed after the EM loop");
    }
}

void em_teardown(Common& c, int g, GpuState& s) {
    (void)hipSetDevice(c.dev[g]);
    if (s.h) { (void)dfm_synchronize(s.h); (void)dfm_set_stream(s.h, nullptr); }
    if (s.d.base) { (void)hipFree(s.d.base); s.d.base = nullptr; }
    if (s.h) (void)dfm_destroy(s.h);
    if (s.d.st) (void)hipStreamDestroy(s.d.st);
}

int join_status(Common& c, char* err, int cap) {
    for (int g = 0; g < c.G; ++g)
        if (c.rc[g] != 0) {
            char buf[640];
            snprintf(buf, sizeof(buf), "GPU %d (device %d): %s", g, c.dev[g], c.msg[g].c_str());
            set_err(err, cap, buf);
            return c.rc[g];
        }
    return 0;
}

}  // namespace

extern "C" {

int dfm_em_batch_multi(int ngpu, const int* device_ids, int B, int T, int N, int r, const double* panel, double* Lam,
                       double* R, double* A, double* Q, double* mu0, double* P0, int max_iter, double tol,
                       double* loglik_path, int* iters, double* f_smooth, double* P_smooth, unsigned flags,
                       int* iterations_run, char* err, int err_cap) {
    set_err(err, err_cap, "");
    Common c;
    if (int rc = check_common(ngpu, device_ids, B, T, N, r, err, err_cap, c.dev)) return rc;
    if (!panel || !Lam || !R || !A || !Q || !mu0 || !P0 || !loglik_path || !iters) { set_err(err, err_cap, "required pointer is NULL"); return DFM_E_NULL; }
    if (max_iter < 1) { set_err(err, err_cap, "max_iter must be >= 1"); return DFM_E_DIMS; }
    c.G = ngpu; c.B = B; c.T = T; c.N = N; c.r = r;
    c.rc.assign(ngpu, 0); c.msg.assign(ngpu, "");
    if (ngpu > 1)
        if (int rc = init_comms(c, err, err_cap)) return rc;
    EmCall k{panel, Lam, R, A, Q, mu0, P0, max_iter, tol, loglik_path, iters, f_smooth, P_smooth, flags, std::vector<int>(ngpu, 0)};
    std::vector<GpuState> st(ngpu);
    auto on_all = [&](auto fn) {                               // GPU 0 on the calling thread, one more thread per further GPU
        std::vector<std::thread> th;
        for (int g = 1; g < ngpu; ++g) th.emplace_back([&, g] { fn(g); });
        fn(0);
        for (auto& t : th) t.join();
    };
    on_all([&](int g) { em_setup(c, k, g, st[g]); });
    if (!c.failed.load()) on_all([&](int g) { em_loop(c, k, g, st[g]); });
    on_all([&](int g) { em_teardown(c, g, st[g]); });
    destroy_comms(c);
    if (iterations_run) *iterations_run = k.iters_run[0];
    if (int rc = join_status(c, err, err_cap)) return rc;
    for (int b = 0; b < B; ++b)
        if (!isfinite(loglik_path[(size_t)b * max_iter])) { set_err(err, err_cap, "non-finite log-likelihood (Q or P0 not positive definite?)"); return DFM_E_NUMERIC; }
    return 0;
}

// The smoother pass has no exchange at all: every GPU's thread runs the host-pointer entry point on its block.
int dfm_ks_pass_batch_multi(int ngpu, const int* device_ids, int B, int T, int N, int r, const double* panel,
                            const double* Lam, const double* R, const double* A, const double* Q, const double* mu0,
                            const double* P0, double* f_smooth, double* P_smooth, double* loglik, unsigned flags,
                            char* err, int err_cap) {
    set_err(err, err_cap, "");
    Common c;
    if (int rc = check_common(ngpu, device_ids, B, T, N, r, err, err_cap, c.dev)) return rc;
    if (!panel || !Lam || !R || !A || !Q || !mu0 || !P0 || !f_smooth || !loglik) { set_err(err, err_cap, "required pointer is NULL"); return DFM_E_NULL; }
    c.G = ngpu; c.B = B; c.T = T; c.N = N; c.r = r;
    c.rc.assign(ngpu, 0); c.msg.assign(ngpu, "");
    auto work = [&](int g) {
        const Shard sh = shard_of(B, ngpu, g);
        const int Bl = sh.hi - sh.lo;
        if (Bl <= 0) return;
        dfm_handle* h = nullptr;
        int rc = dfm_create(&h, c.dev[g], nullptr);
        if (rc != 0) { c.rc[g] = rc; c.msg[g] = "dfm_create failed"; return; }
        const size_t o = (size_t)sh.lo, np = (size_t)r * (r + 1) / 2;
        rc = dfm_ks_pass_batch(h, Bl, T, N, r, panel + o * T * N, Lam + o * N * r, R + o * N, A + o * r * r, Q + o * r * r,
                               mu0 + o * r, P0 + o * r * r, f_smooth + o * T * r, P_smooth ? P_smooth + o * T * np : nullptr,
                               loglik + o, flags);
        if (rc != 0) { c.rc[g] = rc; c.msg[g] = dfm_last_error(h); }
        (void)dfm_destroy(h);
    };
    std::vector<std::thread> th;
    for (int g = 1; g < ngpu; ++g) th.emplace_back(work, g);
    work(0);
    for (auto& t : th) t.join();
    return join_status(c, err, err_cap);
}

}  // extern "C"
