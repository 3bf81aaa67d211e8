#!/usr/bin/env python
"""bench.py -- Kalman-smoother passes/sec on synthetic N=200 T=500 r=8 panels (BASELINE.json metric).

A "step" = one full Kalman-smoother pass (filter + RTS smoother + log-likelihood)
over ONE batch of replicates already resident in HBM.  N=1: BASELINE configs[1] (batch = 1024
replicates on 1 x MI355X).  N>1: one process per GPU (torch.distributed, backend nccl = RCCL), the
replicate batch is sharded with the same 1024 replicates per GPU (weak scaling: the N = 1 line equals the single-GPU
bench); the per-replicate log-likelihoods are all-gathered ONCE per timed block (north_star prescribes the collective per EM
iteration; `--gather step` / secondary.pass_gather_every_step = after every pass).  A default `--gpus N` run also times, as
`secondary`: BASELINE configs[2]'s per-GPU shard (`c3`: 8192 replicates per GPU = 65 536 at N = 8) and the EM iteration with
its all-gather of {loglik, active} (`em`).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats 9] [--batch-per-gpu 1024] [--missing 0.0]
                  [--mode pass|em|pca] [--driver torch|lib] [--no-secondary] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` WITHOUT a torchrun environment launches the N ranks itself (re-exec under torch.distributed.run on
127.0.0.1) and fails loudly when the box has fewer than N devices; n_gpus in the line is dist.get_world_size().

Timing: an untimed pre-heat of >= 50 ms of the same steps (clocks ramp on a fresh box), W untimed warm-up steps, then
`--repeats` blocks of EXACTLY K steps, each block bracketed by barrier + torch.cuda.synchronize() on both sides and
reduced with MAX over ranks; `value` / `ms_per_step` are the MEDIAN block, min / max / every block are printed beside
them.  Panels come from the product's own device generator (dfm_synth_panels_dev: counter-based Philox keyed by (seed,
global replicate index); checked cell by cell against its host restatement in the GPU tests).

--mode em : a step = ONE EM ITERATION (E-step pass + M-step, SURVEY §8(d) "EM iteration") of the replicate-sharded
            driver shard.em_batch_sharded -- dfm_em_iterate_batch_dev on the shard, then the all-gather of {loglik,
            active} every iteration (north_star's collective).  Reported as EM iterations/s (a different metric).
--mode pca: a step = the PCA initialisation (pca_score + OLS start) of the batch (dfm_pca_init_batch_dev).
--driver lib: the same steps through the LIBRARY's multi-GPU object (dfm_multi, csrc/multi.hip: what a Julia host binds):
            ONE process drives --gpus N devices, the job is generated where it lives (dfm_multi_synth), `--mode em` times
            dfm_multi_em (one ncclAllGather per iteration inside the library), `--mode pass` dfm_multi_ks_pass.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, HIP-event timed on the launch stream inside
this script; `ceiling_measured` = this box's own streaming ceilings from dfm_hbm_probe), "cpu_baseline" (oracle/
dfm_oracle.c, the C restatement, timed on the host cores of this box on a bounded sample of the same workload; rank 0,
N=1 only), "device" (clocks / power sampled while the bench batch runs) and -- default invocation only -- "secondary":
the other lines of the path (B = 8192 shard, 10 % missing, EM, config 4, config 4 with missing cells, PCA start), three
blocks each, same method.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the CPU baseline's OpenMP runtime reads these when it is loaded: one thread per core, no migration
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
FP64_MATRIX_PEAK_TFLOPS = 78.6  # AMD MI355X spec sheet, FP64 matrix (the guide's table has no fp64 row; measured
                                # v_mfma_f64_4x4x4 issue rate here: 68 TF/s, scripts/microbench/mfma64.hip)
PREHEAT_MS = 50.0


def algorithmic_bytes(N, T, r):
    """every input read once, every output written once, no scratch."""
    inputs = 8 * (N * T + N * r + N + 2 * r * r + r + r * r)
    outputs = 8 * (T * r + T * r * (r + 1) // 2 + 1)
    return inputs, outputs


def source_hash():
    """sha256 over the kernel sources: a committed PMC traffic file is quoted only for the code it was measured on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "dynamic_factor_models_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/dfm_hip.h"]:
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(kernel, workload_key):
    """HBM bytes per launch of `kernel` from the rocprofv3 --pmc passes (scripts/gpu_profile.sh -> profiles/rNN/
    pmc_traffic*.json), ONLY when that file was produced by exactly this source tree on this workload; else None."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        for name in ("pmc_traffic.json", "pmc_traffic_c4.json", "pmc_traffic_missing10.json", "pmc_traffic_missing10_b8192.json",
                     "pmc_traffic_c4_missing10.json"):                 # headline; BASELINE config 4; the missing-cell lines
            try:
                with open(os.path.join(ROOT, "profiles", rnd, name)) as fh:
                    d = json.load(fh)
                if d.get("_source_hash") == source_hash() and d.get("_workload") == workload_key and kernel in d:
                    return d[kernel].get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                pass
    return None


# ---------------------------------------------------------------------------------------------------------------------
# launcher decision (pure: tested on CPU)
def launch_plan(gpus: int, env: dict, n_devices: int, driver: str = "torch"):
    """What `bench.py --gpus N` does given the environment: ("inline", why) = run in this process; ("spawn", why) =
    re-exec N ranks under torch.distributed.run; ("error", why) = refuse."""
    if gpus < 1:
        return "error", f"--gpus {gpus}: need at least one GPU"
    world = env.get("WORLD_SIZE")
    if world is not None:
        if int(world) != gpus:
            return "error", f"--gpus {gpus} but WORLD_SIZE={world}: the launcher and the flag disagree"
        if driver == "lib" and int(world) > 1:
            return "error", "--driver lib is ONE process driving N GPUs: run it without torchrun"
        return "inline", f"rank {env.get('RANK', '0')} of a torchrun job of {world}"
    if n_devices < gpus:
        return "error", f"--gpus {gpus} but this box exposes {n_devices} HIP device(s)"
    if gpus == 1 or driver == "lib":
        return "inline", "single process"
    return "spawn", f"no torchrun environment: launching {gpus} ranks (torch.distributed.run, 127.0.0.1)"


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(gpus: int, argv):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------------
import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner with C stdio on STDOUT when its first communicator is created -- buffered, so it lands
    BEHIND the JSON line at exit and the driver (one JSON line on stdout) would read the banner.  While a communicator is
    being created, file descriptor 1 points at stderr; C stdio is flushed before it is pointed back."""
    import ctypes
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def host_cpu_info():
    """Cores this process may really use: affinity mask and the cgroup CPU quota (the box shows 256 cores, the container
    may own fewer: an "all cores" figure measured on more threads than that is an oversubscription artefact)."""
    info = dict(os_cpu_count=os.cpu_count())
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        info["affinity"] = None
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                 # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()
            quota = None if q == "max" else float(q) / float(per)
    except Exception:  # noqa: BLE001
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = float(fq.read()), float(fp.read())
                quota = None if q <= 0 else q / per
        except Exception:  # noqa: BLE001
            pass
    info["cgroup_cpu_quota"] = quota
    usable = info["affinity"] or info["os_cpu_count"] or 1
    if quota:
        usable = max(1, min(usable, int(quota)))
    info["usable"] = usable
    return info


def cpu_baseline(panel_host, params_host, target_seconds=20.0):
    """Time the C restatement (oracle/dfm_oracle.c, OpenMP over replicates) on this box's host cores on a bounded
    sample of the same workload.  Thread counts up to what the container may use (affinity, cgroup quota) are probed for
    >= 1 s each on >= 8 replicates per thread into persistent output buffers; the fastest is then timed for the rest of
    the budget, and one thread beside it.  OMP_PROC_BIND=close / OMP_PLACES=cores (set at the top of this file)."""
    import numpy as np
    from oracle import c_oracle as co
    info = host_cpu_info()
    cap = max(1, min(info["usable"], co.num_threads()))
    S, T, N = panel_host.shape
    r = params_host[0].shape[2]
    out = (np.zeros((S, T, r)), np.zeros((S, T, r * (r + 1) // 2)), np.zeros(S))
    co.ks_pass_batch(panel_host, *params_host, out=out)          # warm: thread pool up, every page touched

    def rate(n, seconds):
        Sn = min(S, max(8 * n, 16))
        args = (panel_host[:Sn],) + tuple(p[:Sn] for p in params_host)
        o = tuple(x[:Sn] for x in out)
        co.ks_pass_batch(*args, out=o, nthreads=n)               # this thread count's pool, warm
        done, t0 = 0, time.perf_counter()
        while True:
            co.ks_pass_batch(*args, out=o, nthreads=n)
            done += Sn
            el = time.perf_counter() - t0
            if el >= seconds:
                return done / el, done, el, Sn

    counts = sorted({c for c in (cap, cap // 2, cap // 4, 64, 32, 16, 8) if 1 <= c <= cap})
    probe = {n: rate(n, 1.0)[0] for n in counts}
    cores = max(probe, key=probe.get)
    left = max(3.0, target_seconds - 1.0 * len(counts) - 3.0)
    v, done, el, Sn = rate(cores, left)
    v1, d1, e1, _ = rate(1, 3.0)
    co.ks_pass_batch(panel_host[:16], *[p[:16] for p in params_host], out=tuple(x[:16] for x in out), nthreads=cap)
    return dict(value=v, unit="passes/s", cores=cores, kind="port", per_thread=v / cores,
                thread_probe={str(k): round(x, 1) for k, x in sorted(probe.items())}, host=info,
                single_thread=dict(value=v1, cores=1, sample=f"{d1} passes in {e1:.1f} s"),
                sample=f"{done} passes ({Sn} distinct replicates of the bench batch = {Sn / cores:.1f} per thread, repeated) "
                       f"in {el:.1f} s; oracle/dfm_oracle.c, gcc -O2 -fopenmp, {cores} threads pinned (OMP_PROC_BIND=close), "
                       f"outputs into persistent buffers; every probed thread count ran >= 1 s")


def device_telemetry():
    """sclk / mclk / power / temperature from rocm-smi, sampled by the caller WHILE the bench batch is running."""
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"],
                             capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out[out.index("{"):])
        card = d.get("card0") or next(iter(d.values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(s in kl for s in ("sclk", "mclk", "fclk", "socclk", "power", "temperature (sensor junction)", "temperature (sensor memory)",
                                     "performance level")):
                keep[k] = v
        keep.update(host_box())
        return keep
    except Exception as e:  # noqa: BLE001
        return dict(error=f"rocm-smi not readable: {e}", **host_box())


def host_box():
    """Host kernel of the box.  About one box in ten of this pool runs 6.18.50 (the others 6.18.51); on those the kernel's code
    is cold at EVERY launch and the headline measures 0.40-0.44 instead of 0.54-0.58 of the HBM peak with the same binary
    (clocks, HBM ceilings and idle latencies identical): profiles/r03/slow_boxes/README.md, scripts/microbench/icache.hip."""
    import platform
    rel = platform.release()
    out = dict(host_kernel=rel)
    if rel.startswith("6.18.50"):
        out["box_kind"] = ("slow kind: kernel code not cached across launches on this host kernel (first execution of every "
                           "role of the one-launch pass is 1.5-4x slower); same binary gives 0.54-0.58 on the 6.18.51 boxes "
                           "-- profiles/r03/slow_boxes/README.md")
    return out


# ---------------------------------------------------------------------------------------------------------------------
def kernel_bytes_table(B, N, T, r):
    """Algorithmic bytes per launch of the kernels a roofline line may name (DESIGN.md "Kernels"): compulsory inputs read
    once + outputs written once.  Keys are the names rocprofv3 prints."""
    b_in, b_out = algorithmic_bytes(N, T, r)
    panel_b = 8 * (N * T + N * r + N)                      # panel + loadings + idiosyncratic variances
    npack = r * (r + 1) // 2
    seq = B * (b_in - panel_b + b_out)
    return {"collapse_mfma_kernel": B * panel_b, "collapse_dma_kernel": B * panel_b, "collapse_wide_kernel": B * panel_b,
            "collapse_wide2_kernel": B * panel_b, "collapse_kernel": B * panel_b, "collapse_miss_kernel": B * panel_b,
            "recursion_chunk_kernel": seq,
            "pass_fused_kernel": B * (b_in + b_out),
            "recursion_kernel": seq, "recursion_wave_kernel": seq, "recursion_pair_kernel": seq, "recursion_tile_kernel": seq,
            "meanscan_kernel": B * (b_in - panel_b + 8 * (T * r + 1)), "meanscan_mfma_kernel": B * (b_in - panel_b + 8 * (T * r + 1)),
            "pfill_kernel": B * 8 * T * npack,
            "mstep_mfma_kernel": B * 8 * (N * T + T * r), "mstep_wide_kernel": B * 8 * (N * T + T * r),
            "mstep_lam_kernel": B * 8 * (N * T + T * (r + npack)),
            "gram_kernel": B * 8 * (N * r + N), "wide_prep_kernel": B * 8 * (N * r + N), "cov_kernel": B * 8 * (3 * r * r + r),
            "cov_grid_kernel": B * 8 * (3 * r * r + r), "cov_tile_kernel": B * 8 * (3 * r * r + r),
            "ct_miss_wide_kernel": B * 8 * T * npack, "ct_miss_wide2_kernel": B * 8 * T * npack,
            "ct_miss_slice_kernel": B * 8 * (N * T + T * npack)}      # (the panel once over its launches + the C_t rows)


def em_iteration_bytes(N, T, r, k, q=0):
    """Compulsory bytes of ONE EM iteration of one replicate with a k-wide state (k = r p: VAR(p) factor dynamics in companion form;
    q > 0: AR(q) idiosyncratic terms on top): the panel twice (E-step, loadings step), the parameters in and out, the log-likelihood."""
    par = N * r + N + N * q + r * k + r * r + k + k * k
    return 8 * (2 * N * T + 2 * par + 1)


def gram_flops(kernel, B, N, T):
    """(executed, useful) flops per launch of the X'X kernels of the PCA start.  The product is symmetric: the matrix-pipe kernels
    compute only the upper triangle of their tiling -- 16 x 16 tiles (gram_xx_mfma / gram_xx_dma_kernel, pca.hip) or pairs of
    128-series blocks (gram_xx_wide_kernel) -- so `executed` is what the MFMA units retire and the figure `frac` is priced on
    (counting the full 2 T N^2 B would let a symmetric-product kernel report frac > 1); `useful` = T N (N + 1) B,
    the flops of the distinct entries."""
    useful = float(T) * N * (N + 1) * B
    if kernel in ("gram_xx_mfma_kernel", "gram_xx_dma_kernel"):
        nt = (N + 15) // 16
        return 2.0 * T * 256 * (nt * (nt + 1) // 2) * B, useful
    if kernel == "gram_xx_wide_kernel":
        nb = (N + 127) // 128
        return 2.0 * T * 128 * 128 * (nb * (nb + 1) // 2) * B, useful
    return 2.0 * T * N * N * B, useful                       # gram_xx_kernel (VALU): the full product


class Workload:
    """One bench line: a batch resident in HBM, `step(k)` = k steps of the mode, timed as the contract prescribes."""

    def __init__(self, torch, dist, ctx, shard, world, rank, dev, B, N, T, r, missing, mode, seed=20160415, gather="block", cold=0):
        self.torch, self.dist, self.ctx, self.shard = torch, dist, ctx, shard
        # cold > 1 (pass mode, one rank): `cold` DISTINCT resident batches (panels, parameters AND output buffers), step i on batch
        # i % cold -- no output line of one pass survives in the 256-MB Infinity Cache to be overwritten by the next (VERDICT r4 #2:
        # the default loop re-runs ONE batch, and its 0.18 GB of outputs are write-combined in the cache between passes)
        self.cold = int(cold) if (mode == "pass" and world == 1) else 0
        # multi-rank pass mode: WHEN the replicates' log-likelihoods are all-gathered.  north_star prescribes the collective per EM
        # iteration; a pass is not an iteration, so the default gathers ONCE per timed block of K passes ("block": the result of the
        # job reaches every rank once); "step" = after every pass (on RCCL's stream beside the next pass) is the secondary line
        # `pass_gather_every_step`.  EM mode always gathers {loglik, active} every iteration (shard.em_batch_sharded).
        self.gather = gather
        self.world, self.rank, self.dev = world, rank, dev
        self.B, self.N, self.T, self.r, self.missing, self.mode = B, N, T, r, missing, mode
        # (DFM_BENCH_FORCE_DIST=1 under a 1-rank torchrun: the multi-rank step -- async all-gather on RCCL's stream, barrier fences --
        # on a one-GPU box; diagnostics)
        self.distributed = world > 1 or (dist is not None and dist.is_initialized() and os.environ.get("DFM_BENCH_FORCE_DIST") == "1")
        self.may_missing = missing > 0.0
        # this rank's replicates [rank B, (rank + 1) B) of the job's world * B (shard.replicate_range), generated where they live
        self.panel, self.params = ctx.synth_panels(seed, rank * B, B, T, N, r, missing_prob=missing)
        f64 = torch.float64
        self.f = torch.empty((B, T, r), dtype=f64, device=dev)
        self.P = torch.empty((B, T, r * (r + 1) // 2), dtype=f64, device=dev) if mode == "pass" else None
        self.ll = torch.empty((B,), dtype=f64, device=dev)
        # multi-rank pass mode: the step's all-gather runs on RCCL's own stream BESIDE the next step's pass (two loglik / gather
        # buffers in turn; a buffer is reused only after the gather that read it has completed)
        self.ll_pair = [self.ll, torch.empty((B,), dtype=f64, device=dev)] if self.distributed else [self.ll]
        self.ll_all = [torch.empty((world * B,), dtype=f64, device=dev) for _ in range(2)] if self.distributed else None
        self.stream = None
        if self.distributed and os.environ.get("DFM_BENCH_NO_PRIO") != "1":
            lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
            self.stream = torch.cuda.Stream(device=dev, priority=hi)
        self.cold_sets = []
        for c in range(1, self.cold):
            pan, par = ctx.synth_panels(seed + 7919 * c, rank * B, B, T, N, r, missing_prob=missing)
            self.cold_sets.append((pan, par, torch.empty_like(self.f), torch.empty_like(self.P), torch.empty_like(self.ll)))
        self.em_params = None
        if mode == "em":
            if self.may_missing:
                self.em_params = [p.clone() for p in self.params]       # DGP parameters as the start (PCA needs a balanced panel)
            else:
                self.em_params = list(ctx.pca_init_batch(self.panel, r, want_factors=False)[:6])

    def steps(self, k, profile=False):
        # multi-rank: the passes go on a HIGH-priority stream.  The step's collective and the next pass become eligible at the same
        # moment; at equal priority RCCL's workgroup sometimes takes a CU first and that CU's pass workgroup (one per CU, nothing
        # fits beside it) starts ~20 us late.  With the pass preferred, the collective runs in the TAIL of the next pass, where
        # CUs go idle one by one anyway.
        if self.distributed and self.stream is not None and not profile:
            with self.torch.cuda.stream(self.stream):
                return self._steps(k, profile)
        return self._steps(k, profile)

    def _steps(self, k, profile=False):
        ctx, dist = self.ctx, self.dist
        if self.mode == "pass":
            pending = [None, None]
            for it in range(k):
                u = it & 1 if self.distributed else 0
                if pending[u] is not None:
                    pending[u].wait()                  # (the gather of step it - 2: long done; orders the buffer's reuse)
                    pending[u] = None
                if self.cold > 1 and it % self.cold:
                    pan, par, f_, P_, ll_ = self.cold_sets[it % self.cold - 1]
                    ctx.ks_pass_batch(pan, *par, may_have_missing=self.may_missing, out=(f_, P_, ll_))
                    continue
                ctx.ks_pass_batch(self.panel, *self.params, may_have_missing=self.may_missing, out=(self.f, self.P, self.ll_pair[u]))
                if self.distributed and not profile and (self.gather == "step" or it == k - 1):
                    pending[u] = dist.all_gather_into_tensor(self.ll_all[u], self.ll_pair[u], async_op=True)
            for h in pending:
                if h is not None:
                    h.wait()
        elif self.mode == "em":
            # k EM iterations of the sharded driver: dfm_em_iterate_batch_dev + the all-gather of {loglik, active} EVERY
            # iteration (tol = 0: no early stop, so exactly k iterations are timed)
            self.shard.em_batch_sharded(ctx, self.panel, *self.em_params, B_global=self.world * self.B, max_iter=k, tol=0.0,
                                        want_smooth=False, may_have_missing=self.may_missing)
        else:
            for _ in range(k):
                ctx.pca_init_batch(self.panel, self.r, want_factors=False)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.distributed:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, steps, warmup, repeats, preheat_ms=PREHEAT_MS):
        torch, dist = self.torch, self.dist
        # pre-heat: the same steps, untimed, until >= preheat_ms of device time have passed (DVFS ramp of a fresh box)
        # (every rank decides by its OWN clock how long it pre-heats, so the pre-heat steps must not contain a collective:
        # the pass and the PCA start run without their all-gather here; the EM driver's iterations end in one, so a
        # multi-rank EM job pre-heats a fixed number of iterations instead)
        self.fence()
        t0 = time.perf_counter()
        heat = 0
        if self.distributed and self.mode == "em":
            self.steps(2 * steps); heat = 2 * steps
        else:
            while (time.perf_counter() - t0) * 1e3 < preheat_ms:
                self.steps(max(2, steps // 4), profile=True); heat += max(2, steps // 4)
                torch.cuda.synchronize()
        self.steps(max(warmup, 1) if self.mode == "em" else warmup)
        blocks = []
        for _ in range(max(repeats, 1)):
            self.fence()
            t0 = time.perf_counter()
            self.steps(steps)
            self.fence()
            el = time.perf_counter() - t0
            if self.distributed:
                tt = torch.tensor([el], dtype=torch.float64, device=self.dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            blocks.append(el)
        srt = sorted(blocks)
        elapsed = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
        if self.mode == "pass":
            assert bool(torch.isfinite(self.ll).all()), "non-finite log-likelihood in the bench batch"
            self.ctx.synchronize()              # + the status word of the last pass (an expired bounded wait would raise)
        # roofline leg: K more steps with a HIP-event pair around every kernel launch (on the launch stream)
        self.ctx.profile_enable(True)
        self.steps(steps, profile=True)
        prof = self.ctx.profile_read()
        self.ctx.profile_enable(False)
        return dict(ms_per_step=1e3 * elapsed / steps, value=self.world * self.B * steps / elapsed, blocks=blocks, sorted=srt,
                    avg={k: v[0] / v[1] for k, v in prof.items()}, launches={k: v[1] for k, v in prof.items()}, steps=steps,
                    preheat_steps=heat)

    def roofline(self, res):
        """The roofline object of one line (rank 0)."""
        B, N, T, r = self.B, self.N, self.T, self.r
        avg, ms_per_step = res["avg"], res["ms_per_step"]
        b_in, b_out = algorithmic_bytes(N, T, r)
        workload_key = f"{self.mode}:B{B}:N{N}:T{T}:r{r}:m{self.missing}"
        kern_bytes = kernel_bytes_table(B, N, T, r)
        kernels_ms = {k: round(v, 4) for k, v in avg.items()}
        if self.mode == "pca":
            dom = max(avg, key=avg.get)
            gx = next((k for k in ("gram_xx_kernel", "gram_xx_mfma_kernel", "gram_xx_dma_kernel", "gram_xx_wide_kernel") if k in avg), None)
            out = dict(bound="mfma", kernel=dom, achieved=None, peak=FP64_MATRIX_PEAK_TFLOPS, unit="TFLOP/s", frac=None, traffic=None,
                       avg_launch_ms=avg[dom], kernels_ms=kernels_ms,
                       note="pca_kernel (subspace iteration, Rayleigh-Ritz, scores, OLS / VAR start) is a chain of dependent small-matrix "
                            "stages on an L2-resident Gram matrix: latency-bound, frac = null (no roofline claim; stage times by the "
                            "diagnostics build's DFM_PCA_STOP, DESIGN.md section 10.9); `gram` = X'X of the batch on v_mfma_f64_16x16x4, "
                            "priced on the flops the kernel EXECUTES (upper triangle of its tiling), not on 2 T N^2")
            if gx:
                ex, useful = gram_flops(gx, B, N, T)
                ach = ex / (avg[gx] * 1e-3) / 1e12
                out["gram"] = dict(kernel=gx, achieved=ach, peak=FP64_MATRIX_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP64_MATRIX_PEAK_TFLOPS,
                                   avg_launch_ms=avg[gx], flops_per_launch=ex, useful_flops_per_launch=useful,
                                   useful_tflops=useful / (avg[gx] * 1e-3) / 1e12)
                if dom == gx:
                    out.update(achieved=ach, frac=ach / FP64_MATRIX_PEAK_TFLOPS)
            return out
        cands = [k for k in avg if k in kern_bytes]
        dom = max(cands, key=avg.get)
        # a kernel cannot take longer than the step that contains it: the event pair adds its own latency (measured +3..8 %),
        # so when the step IS one kernel the step's wall clock is the better estimate of the launch duration
        one_kernel = sum(res["launches"].values()) == res["launches"][dom]
        dur = min(avg[dom], ms_per_step) if one_kernel else avg[dom]
        achieved = kern_bytes[dom] / (dur * 1e-3) / 1e9
        unit_bytes = (b_in + b_out) if self.mode == "pass" else (b_in + b_out + 8 * N * T + 8 * (N * r + N + 2 * r * r))
        whole = B * unit_bytes / (ms_per_step * 1e-3) / 1e9
        out = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                   traffic=measured_traffic(dom, workload_key), avg_launch_ms=dur, avg_launch_ms_hip_events=avg[dom],
                   duration_source=("min(HIP-event pair, wall clock of the step): the step is this one kernel" if one_kernel
                                    else "HIP-event pairs on the launch stream"),
                   bytes_per_launch=kern_bytes[dom], kernels_ms=kernels_ms,
                   note=("pass_fused_kernel = the whole pass in one launch: every input read once, every output written once.  The timed loop "
                         "re-runs ONE resident batch: its 0.18 GB of outputs are still in the 256-MB Infinity Cache when the next pass "
                         "overwrites them (every pass computes and stores everything; ~18 % of the algorithmic bytes are write-combined "
                         "there instead of reaching HBM inside the timed region) -- secondary.pass_cold cycles over 5 distinct batches and "
                         "is the cache-cold figure of the same kernel"
                         if dom == "pass_fused_kernel" else
                         "panel with missing cells: collapse_miss_kernel streams the panel once and writes one table row per period; "
                         "recursion_chunk_kernel runs the T periods as 64 time chunks (one per lane, 16 + 16 steps each at T = 500)"
                         if dom in ("recursion_chunk_kernel", "collapse_miss_kernel") else
                         "wide state with missing cells: collapse_wide2_kernel streams the panel once, ct_miss_slice_kernel forms C_t of the periods "
                         "with missing cells from slices of lam / sqrt(R) resident in LDS, recursion_tile_kernel runs each replicate as time chunks "
                         "(one workgroup each, warmed up over 16 periods, boundaries checked) of dependent 32 x 32 inversions on the matrix pipe -- "
                         "latency-bound, not HBM-bound"
                         if dom == "recursion_tile_kernel" else
                         "sequential path (panel with missing cells): the collapse streams the panel once, the recursion kernel is "
                         "a chain of T dependent r x r inversions per replicate -- latency-bound, not HBM-bound"
                         if dom.startswith("recursion") else "dominant kernel of this mode by HIP-event time"),
                   whole_step=dict(bytes_per_unit=unit_bytes, achieved=whole, frac=whole / HBM_PEAK_GBS,
                                   note="SURVEY §8(d) algorithmic bytes per " + ("pass" if self.mode == "pass" else "EM iteration")
                                        + " x units/s of ONE GPU (median wall clock of the timed blocks) / HBM peak"))
        out["whole_pass"] = out["whole_step"]      # round-1 key kept for the driver's readers
        return out

    def free(self):
        for k in ("panel", "params", "f", "P", "ll", "ll_pair", "ll_all", "em_params", "cold_sets"):
            setattr(self, k, None)
        self.torch.cuda.empty_cache()


METRIC = {"pass": ("Kalman-smoother passes/sec", "passes/s", "one full Kalman-smoother pass per step"),
          "em": ("EM iterations/sec (E-step pass + M-step)", "EM iterations/s", "one EM iteration (pass + M-step) per step"),
          "pca": ("PCA initialisations/sec (pca_score + OLS start)", "initialisations/s", "one PCA initialisation per step")}

SECONDARY = [   # (key, dict(B, N, T, r, missing, mode), steps, warmup) -- three blocks each
    ("pass_cold", dict(B=1024, N=200, T=500, r=8, missing=0.0, mode="pass", cold=5), 20, 5),
    ("b8192", dict(B=8192, N=200, T=500, r=8, missing=0.0, mode="pass"), 5, 2),
    ("missing10", dict(B=1024, N=200, T=500, r=8, missing=0.1, mode="pass"), 10, 3),
    ("em", dict(B=1024, N=200, T=500, r=8, missing=0.0, mode="em"), 10, 3),
    ("em_missing10", dict(B=1024, N=200, T=500, r=8, missing=0.1, mode="em"), 10, 3),
    ("missing10_b8192", dict(B=8192, N=200, T=500, r=8, missing=0.1, mode="pass"), 3, 1),
    ("em_missing10_b8192", dict(B=8192, N=200, T=500, r=8, missing=0.1, mode="em"), 3, 1),
    ("pca", dict(B=1024, N=200, T=500, r=8, missing=0.0, mode="pca"), 3, 1),
    ("c4", dict(B=256, N=1000, T=2000, r=20, missing=0.0, mode="pass"), 5, 2),
    ("c4_em", dict(B=256, N=1000, T=2000, r=20, missing=0.0, mode="em"), 3, 1),
    ("c4_missing10", dict(B=256, N=1000, T=2000, r=20, missing=0.1, mode="pass"), 2, 1),
    ("c4_em_missing10", dict(B=256, N=1000, T=2000, r=20, missing=0.1, mode="em"), 2, 1),
    # a batch that leaves most CUs idle under one workgroup per replicate: recursion_tile_kernel cuts each replicate into 16 time chunks
    ("c4_missing10_b32", dict(B=32, N=1000, T=2000, r=20, missing=0.1, mode="pass"), 3, 1),
]


def config1_config5_lines(torch, ctx, dev, cpu_seconds=2.0):
    """BASELINE configs[0] and configs[4] on one GPU, driver-visible (the lines of scripts/bench_extra.py):
      c1_als  -- Stock-Watson real panel (tests/golden/sw_panel.npz, built from the reference's spreadsheet), r = 4, PCA start + 10
                 ALS sweeps = the reference's `estimate_factor!(m, 10)` (dfm_functions.ipynb:328-382), 4096 independent runs in ONE
                 dfm_als_batch_dev call; beside it oracle/als_oracle.py (NumPy, one thread) on the same problem, results compared;
      c5_boot -- 10 000 wild-bootstrap draws of the 4-factor VAR(4) -> re-estimation -> Cholesky -> IRFs to 12 horizons -> 5 bands
                 (dfm_functions.ipynb:444-492, 793-825), beside oracle/boot_oracle.py on a bounded sample of draws.
    Both are latency-bound small-matrix work on L2-resident inputs: no roofline claim."""
    import ctypes
    import numpy as np
    from dynamic_factor_models_amd import api
    from oracle import als_oracle as ao
    from oracle import boot_oracle as bo
    out = {}
    d = np.load(os.path.join(ROOT, "tests", "golden", "sw_panel.npz"))
    bp, inc, cat = d["bpdata"], d["inclcode"], d["bpcatcode"]
    real = np.isin(np.floor(cat), [1, 2, 3, 5])
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    K = 5
    # ---- config 1
    t_line = time.perf_counter()
    z, _ = api.standardize_data(bp[2:224][:, real][:, inc[real] == 1])
    z = np.ascontiguousarray(z)
    T, N = z.shape
    F0 = api.pca_start(ctx, z, 4)
    B = 4096
    zt = torch.from_numpy(z).to(dev)
    F_in = torch.from_numpy(np.repeat(F0[None], B, axis=0)).to(dev)
    F = F_in.clone()
    Lam = torch.empty((B, N, 4), dtype=torch.float64, device=dev)
    iters = torch.empty(B, dtype=torch.int32, device=dev)
    ssr = torch.empty(B, dtype=torch.float64, device=dev)

    def als_call():
        F.copy_(F_in)
        ctx._sync_stream()
        rc = ctx._lib.dfm_als_batch_dev(ctx._h, B, T, N, 4, p(zt), 0, None, p(F), p(Lam), 20, 10, 1e-8, None, 0, p(iters), p(ssr), None)
        assert rc == 0, rc
    als_call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        als_call()
    torch.cuda.synchronize()
    gpu_s = (time.perf_counter() - t0) / K
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < cpu_seconds:
        o = ao.estimate_factor(bp[:, real], inc[real], 3, 224, 4, max_iter=10, solver="normal", compute_r2_flag=False, f0=F0)
        n += 1
    cpu_s = (time.perf_counter() - t0) / n
    ok = abs(float(ssr[0]) - o["ssr"]) < 1e-8 * o["ssr"] and abs(float(ssr[-1]) - o["ssr"]) < 1e-8 * o["ssr"]
    als_bytes = 8 * T * N + B * 8 * (2 * T * 4 + N * 4 + 2)        # the shared panel once; per run: start factors in, factors + loadings + ssr + iters out
    out["c1_als"] = dict(workload="BASELINE configs[0]: Stock-Watson real panel (222 x 58, 12 700 observed cells), r=4, PCA start + 10 ALS sweeps "
                                  "(estimate_factor!(m, 10)), 4096 runs per dfm_als_batch_dev call",
                         value=B / gpu_s, unit="ALS runs/s", ms_per_step=1e3 * gpu_s, batch=B, sweeps_per_s=10 * B / gpu_s, dominant="als_kernel",
                         whole_step=als_bytes / gpu_s / 1e9 / HBM_PEAK_GBS, compulsory_bytes=als_bytes, matches_oracle_ssr=bool(ok),
                         note="latency-bound (222 + 58 small dependent solves per sweep and run); the 103-KB panel is L2-resident: whole_step "
                              "(compulsory bytes / time / HBM peak) says how far from any memory bound this small-matrix work is, not a roofline claim",
                         cpu_baseline=dict(value=1.0 / cpu_s, unit="ALS runs/s", cores=1, kind="port",
                                           sample=f"{n} runs of oracle/als_oracle.py (NumPy normal equations) in {cpu_seconds:.0f} s"),
                         seconds=round(time.perf_counter() - t_line, 2))
    # ---- config 5
    t_line = time.perf_counter()
    m = api.DFMModel(bp, inc, 20, 40, 3, 224, 0, 4, 1e-8, 4, 4)
    api.estimate(m, api.NonParametric(), ctx=ctx)
    v = m.factor_var_model
    rows = np.nonzero(~np.isnan(v.resid).any(axis=1))[0]
    y = v.y[rows[0] - 4: rows[-1] + 1]
    resid = np.zeros_like(y); resid[4:] = v.resid[rows]
    Bd, H = 10000, 12
    yt, bt, et = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (y, v.betahat, resid))
    irf = torch.empty((Bd, 4, H, 4), dtype=torch.float64, device=dev)
    q = torch.tensor([0.05, 0.16, 0.5, 0.84, 0.95], dtype=torch.float64, device=dev)
    bands = torch.empty((5, 4 * H * 4), dtype=torch.float64, device=dev)

    def boot_call():
        ctx._sync_stream()
        rc = ctx._lib.dfm_var_bootstrap_irf_dev(ctx._h, Bd, y.shape[0], 4, 4, H, p(yt), p(bt), p(et), None, ctypes.c_uint64(20160415),
                                                ctypes.c_int64(0), None, p(irf))
        assert rc == 0, rc
        rc = ctx._lib.dfm_quantile_bands_dev(ctx._h, Bd, 4 * H * 4, 5, p(irf), p(q), p(bands))
        assert rc == 0, rc
    boot_call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        boot_call()
    torch.cuda.synchronize()
    gpu_s = (time.perf_counter() - t0) / K
    g = np.random.default_rng(0)
    nd = 256
    signs = np.where(g.random((nd, y.shape[0])) < 0.5, -1.0, 1.0)
    t0 = time.perf_counter(); done = 0
    irf_o = None
    while time.perf_counter() - t0 < cpu_seconds:
        irf_o = bo.var_bootstrap_irf(y, 4, H, signs)[0]; done += nd
    cpu_s = (time.perf_counter() - t0) / done
    # the same draws (the oracle's signs) on the GPU: responses equal to the oracle's to 1e-9 of their scale
    irf_g = ctx.var_bootstrap_irf_host(y, v.betahat, resid, 4, H, nd, signs=signs)
    irf_g = irf_g[0] if isinstance(irf_g, tuple) else irf_g
    boot_ok = bool(np.abs(np.asarray(irf_g) - irf_o).max() <= 1e-9 * np.abs(irf_o).max())
    boot_bytes = 8 * (2 * y.size + v.betahat.size) + Bd * 8 * 4 * H * 4 + 5 * 8 * 4 * H * 4   # shared inputs once; per draw its IRFs out; the bands
    out["c5_boot"] = dict(workload="BASELINE configs[4] on one GPU: 10000 wild-bootstrap draws x VAR(4) of the 4 Stock-Watson factors (T=222) -> "
                                   "IRFs to 12 horizons -> 5/16/50/84/95 % bands",
                          value=Bd / gpu_s, unit="bootstrap draws/s", ms_per_step=1e3 * gpu_s, draws=Bd, dominant="var_boot_kernel",
                          whole_step=boot_bytes / gpu_s / 1e9 / HBM_PEAK_GBS, compulsory_bytes=boot_bytes,
                          bands_finite=bool(torch.isfinite(bands).all()), matches_oracle=boot_ok,
                          note="latency-bound (218 dependent periods per draw, 17 x 17 normal equations); draw + re-estimation + Cholesky + IRF + bands",
                          cpu_baseline=dict(value=1.0 / cpu_s, unit="bootstrap draws/s", cores=1, kind="port",
                                            sample=f"{done} draws of oracle/boot_oracle.py (NumPy) in {cpu_s * done:.1f} s"),
                          seconds=round(time.perf_counter() - t_line, 2))
    return out


def f3_lines(torch, ctx, dev, cpu_seconds=2.0):
    """SURVEY 8 f3, driver-visible: one EM iteration of the model with VAR(p) factor dynamics (companion state
    r p = 16; dfm_functions.ipynb:477-492) and of the model with AR(q) idiosyncratic terms on top (state 20), both at the
    Stock-Watson window's shape (139 series x 222 periods, r = 4, p = q = 4: what `estimate!` hands over, :405-412), 10 % of the
    cells missing, 1024 replicates per call -- beside the oracle (NumPy, one thread), first replicate's likelihood path compared.
    Companion states run the covariance-form sequential recursion (recursion_wave_kernel<16 | 32, COV>): latency-bound, no
    roofline claim."""
    import numpy as np
    from oracle import ar_oracle as aro
    from oracle import varp_oracle as vo
    out = {}
    Bv, Nv, Tv, rv, pv, qv, miss, nit = 1024, 139, 222, 4, 4, 4, 0.1, 5
    tile = lambda a: torch.from_numpy(np.ascontiguousarray(np.tile(a, (Bv // 16,) + (1,) * (a.ndim - 1)))).to(dev)
    # ---- VAR(p)
    t_line = time.perf_counter()
    KV = ("Lam", "R", "Avar", "Q", "mu0", "P0")
    xs, qs = [], []
    for b in range(16):
        x = vo.synth_varp(b, Nv, Tv, rv, pv, missing=miss)
        xs.append(x); qs.append(vo.varp_init(np.nan_to_num(x), rv, pv)[0])
    xv = tile(np.stack(xs))
    d0 = {k: tile(np.stack([q[k] for q in qs])) for k in KV}
    s_it = None
    for rep in range(4):                                          # (the first call warms the clocks up after the CPU legs before it: best of the other three)
        dd = {k: v.clone() for k, v in d0.items()}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        path, its, _, _ = ctx.em_varp_batch(xv, *[dd[k] for k in KV], max_iter=nit, tol=0.0, may_have_missing=True)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / nit
        if rep > 0: s_it = el if s_it is None else min(s_it, el)
    _, opath, _ = vo.em_varp(xs[0], dict(qs[0]), pv, max_iter=nit, tol=0.0)
    ok = bool(np.allclose(path[0].cpu().numpy(), opath, rtol=1e-8) and np.allclose(path[16].cpu().numpy(), opath, rtol=1e-8))
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < cpu_seconds:
        vo.em_step_varp(xs[0], p=pv, **dict(qs[0])); n += 1
    cpu_s = (time.perf_counter() - t0) / n
    out["varp_em"] = dict(workload=f"VAR({pv}) factor dynamics, r={rv} (companion state {rv * pv}), N={Nv} T={Tv} (Stock-Watson :All window shape), "
                                   f"{miss:.0%} missing, batch {Bv}: one EM iteration (dfm_em_varp_batch)",
                          value=Bv / s_it, unit="EM iterations/s", ms_per_step=1e3 * s_it, batch=Bv,
                          whole_step=Bv * em_iteration_bytes(Nv, Tv, rv, rv * pv) / s_it / 1e9 / HBM_PEAK_GBS,
                          compulsory_bytes=Bv * em_iteration_bytes(Nv, Tv, rv, rv * pv),
                          dominant="recursion_comp_kernel", matches_oracle=ok,
                          note="16-wide companion state, singular Q: one wave per replicate on v_mfma_f64_16x16x4 tiles, information form as a block "
                               "elimination with 4 x 4 pivots (no 16 x 16 inversion per period; round 5: recursion_wave_kernel<16, COV>, 4.2 ms) -- a "
                               "dependent chain per replicate, latency-bound",
                          cpu_baseline=dict(value=1.0 / cpu_s, unit="EM iterations/s", cores=1, kind="port",
                                            sample=f"{n} iterations of oracle/varp_oracle.py em_step_varp (NumPy) in {cpu_seconds:.0f} s"),
                          seconds=round(time.perf_counter() - t_line, 2))
    # ---- AR(q) idiosyncratic terms
    t_line = time.perf_counter()
    KA = ("Lam", "sig2", "rho", "Avar", "Q", "mu0", "P0")
    xs, sts = zip(*[aro.synth_ar(b, Nv, Tv, rv, pv, qv, missing=miss) for b in range(16)])
    xa = tile(np.stack(xs))
    a0 = {k: tile(np.stack([st[k] for st in sts])) for k in KA}
    s_it = None
    for rep in range(3):
        aa = {k: v.clone() for k, v in a0.items()}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        path, its, _, _ = ctx.em_ar_batch(xa, *[aa[k] for k in KA], max_iter=nit)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / nit
        if rep > 0: s_it = el if s_it is None else min(s_it, el)
    _, opath, _ = aro.em_ar(xs[0], {k: sts[0][k] for k in KA}, max_iter=nit)
    ok = bool(np.allclose(path[0].cpu().numpy(), opath, rtol=1e-7) and np.allclose(path[16].cpu().numpy(), opath, rtol=1e-7))
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < cpu_seconds:
        aro.em_step_ar(xs[0], **{k: sts[0][k] for k in KA}); n += 1
    cpu_s = (time.perf_counter() - t0) / n
    out["ar_em"] = dict(workload=f"AR({qv}) idiosyncratic terms + VAR({pv}) factors, r={rv} (state {rv * (qv + 1)}), N={Nv} T={Tv}, {miss:.0%} missing, "
                                 f"batch {Bv}: one ECM iteration (dfm_em_ar_batch)",
                        value=Bv / s_it, unit="EM iterations/s", ms_per_step=1e3 * s_it, batch=Bv,
                        whole_step=Bv * em_iteration_bytes(Nv, Tv, rv, rv * (qv + 1), qv) / s_it / 1e9 / HBM_PEAK_GBS,
                        compulsory_bytes=Bv * em_iteration_bytes(Nv, Tv, rv, rv * (qv + 1), qv),
                        dominant="mstep_ar_kernel", matches_oracle=ok,
                        note="quasi-differenced observation equation, 20-wide companion state padded to 32: recursion_comp_kernel<2> (one wave per replicate, 2 x 2 "
                             "matrix-pipe tiles, 4 x 4 pivots; round 5: recursion_wave_kernel<32, COV>, 33.9 ms of the 43.5) and collapse_kernel<32,2,1,20> (the real state width as a template parameter: "
                             "5.6 -> 1.1 ms) -- what is left is the series CM-steps (mstep_ar_kernel 4.3 ms of the 7.8)",
                        cpu_baseline=dict(value=1.0 / cpu_s, unit="EM iterations/s", cores=1, kind="port",
                                          sample=f"{n} iterations of oracle/ar_oracle.py em_step_ar (NumPy) in {cpu_seconds:.0f} s"),
                        seconds=round(time.perf_counter() - t_line, 2))
    return out


def c1_em_lines(torch, ctx, dev, cpu_seconds=2.0):
    """BASELINE configs[0] in its PARAMETRIC form, driver-visible: the Stock-Watson panel's :All window (222 periods x
    the included series with >= 20 observations, its own ragged missing pattern), r = 4, factor VAR(p) with p = 1 and with the model's
    own n_factorlag = 4 (dfm_functions.ipynb:120-146; companion state 16, :477-492).  The point estimate (PCA start + 10 EM
    iterations through api.estimate(m, Parametric())) is untimed set-up; from it 1024 parametric-bootstrap panels are drawn with the
    window's missing pattern (what `estimate!(m, Parametric(); nrep)` does) and the timed unit is the batch's 10 EM iterations from
    the point estimate, panels and parameters resident in HBM.  Beside it the oracle's EM (NumPy, one thread) on replicate 0, whose
    log-likelihood path is compared."""
    import numpy as np
    from dynamic_factor_models_amd import api
    from oracle import kalman_oracle as ko
    from oracle import varp_oracle as vo
    out = {}
    d = np.load(os.path.join(ROOT, "tests", "golden", "sw_panel.npz"))
    r, init, last, Bc, nit = 4, 3, 224, 1024, 10
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for lags in (1, 4):
        t_line = time.perf_counter()
        m = api.DFMModel(d["bpdata"], d["inclcode"], 20, 40, init, last, 0, r, 1e-8, 4, 4)
        api.estimate(m, api.Parametric(), max_em_iter=nit, tol_em=0.0, factor_lags=lags, ctx=ctx)
        q = m.em_params
        z, _ = api.standardize_data(d["bpdata"][init - 1:last][:, d["inclcode"] == 1])
        z = z[:, (~np.isnan(z)).sum(axis=0) >= 20]
        T, N = z.shape
        k = r * lags
        Avar = q["A"]                                               # ([A_1 .. A_p], r x r p; the r x r transition at p = 1)
        rng = np.random.default_rng(20160415 + lags)
        LQ, LS, sq = api._psd_sqrt(q["Q"]), api._psd_sqrt(q["P0"]), np.sqrt(q["R"])
        st = q["mu0"][None] + rng.standard_normal((Bc, k)) @ LS.T                   # companion state [f_t .. f_t-p+1]
        panels = np.empty((Bc, T, N))
        for t in range(T):
            f = st @ Avar.T + rng.standard_normal((Bc, r)) @ LQ.T
            st = np.concatenate([f, st[:, :k - r]], axis=1)
            panels[:, t] = f @ q["Lam"].T + sq * rng.standard_normal((Bc, N))
        panels[:, np.isnan(z)] = np.nan
        keys = ("Lam", "R", "Avar" if lags > 1 else "A", "Q", "mu0", "P0")
        host = {kk: np.repeat(q["A" if kk == "Avar" else kk][None], Bc, axis=0) for kk in keys}
        x = up(panels)
        d0 = {kk: up(v) for kk, v in host.items()}
        run = ctx.em_varp_batch if lags > 1 else ctx.em_batch
        s_job, path = None, None
        for rep_ in range(4):                                         # (first call: clocks after the CPU set-up; best of the other three)
            dd = {kk: v.clone() for kk, v in d0.items()}
            torch.cuda.synchronize(); t0 = time.perf_counter()
            path, its, _, _ = run(x, *[dd[kk] for kk in keys], max_iter=nit, tol=0.0, may_have_missing=True)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
            if rep_ > 0: s_job = el if s_job is None else min(s_job, el)
        start = {kk: host[kk][0] for kk in keys}
        t0 = time.perf_counter(); n = 0; opath = None
        while n == 0 or time.perf_counter() - t0 < cpu_seconds:
            opath = (vo.em_varp(panels[0], dict(start), lags, nit)[1] if lags > 1 else ko.em(panels[0], dict(start), max_iter=nit, tol=0.0)[1])
            n += 1
        cpu_s = (time.perf_counter() - t0) / n
        got = path[0].cpu().numpy()
        err = float(np.max(np.abs(got - opath) / np.abs(opath)))
        nb = Bc * em_iteration_bytes(N, T, r, k) * nit
        out["c1_em" if lags == 1 else "c1_em_p4"] = dict(
            workload=f"BASELINE configs[0], parametric form: Stock-Watson :All window ({T} x {N}, {np.isnan(z).mean():.1%} of the cells missing, ragged), "
                     f"r={r}, factor VAR({lags}), {Bc} parametric-bootstrap replicates x {nit} EM iterations from the point estimate",
            value=Bc / s_job, unit="replicate estimations/s", ms_per_step=1e3 * s_job, batch=Bc, em_iterations=nit,
            whole_step=nb / s_job / 1e9 / HBM_PEAK_GBS, compulsory_bytes=nb,
            dominant="recursion_wave_kernel" if lags == 1 else "recursion_comp_kernel",
            matches_oracle=bool(err <= 1e-8), loglik_path_max_rel_err=err,
            cpu_baseline=dict(value=1.0 / cpu_s, unit="replicate estimations/s", cores=1, kind="port",
                              sample=f"{n} x {nit} EM iterations of oracle/{'varp_oracle.py em_varp' if lags > 1 else 'kalman_oracle.py em'} (NumPy) on replicate 0"),
            seconds=round(time.perf_counter() - t_line, 2))
    return out


def secondary_plan(world: int, default_line: bool, no_secondary: bool):
    """Which secondary lines a run times (pure: tested on CPU).  One GPU, default invocation: the other lines of the path.
    N > 1 GPUs, default invocation (what the driver's SCALE run launches): BASELINE configs[2]'s per-GPU shard (8192 replicates
    per GPU = 65 536 at N = 8), the EM line with its all-gather per iteration, and the pass with the gather after EVERY step."""
    if not default_line or no_secondary:
        return []
    if world == 1:
        return [k for k, *_ in SECONDARY] + ["c1_als", "c5_boot", "varp_em", "ar_em", "c1_em", "c1_em_p4"]
    return ["c3", "em", "pass_gather_every_step"]


MULTI_SECONDARY = {   # key -> (cfg, steps, warmup, gather)
    "c3": (dict(B=8192, N=200, T=500, r=8, missing=0.0, mode="pass"), 5, 2, "block"),
    "em": (dict(B=1024, N=200, T=500, r=8, missing=0.0, mode="em"), 10, 3, "block"),
    "pass_gather_every_step": (dict(B=1024, N=200, T=500, r=8, missing=0.0, mode="pass"), 20, 3, "step"),
}


HBM_BYTES_PER_GPU = 288 * 10**9


def resident_bytes(cfg, lib=None):
    """HBM bytes ONE GPU holds for a line's batch (pure: tested on CPU for every line a default run times, incl. the N-GPU plan):
    panels + parameters + outputs, x the number of distinct resident batches of a cache-cold line, + the library's workspace
    (dfm_workspace_bytes: the larger of the plans the entry points may take for the shape)."""
    B, N, T, r = cfg["B"], cfg["N"], cfg["T"], cfg["r"]
    b_in, b_out = algorithmic_bytes(N, T, r)
    if lib is None:
        from dynamic_factor_models_amd import _lib
        lib = _lib.load()
    flags = 1 if cfg.get("missing", 0.0) > 0 else 0                # DFM_F_MAY_HAVE_MISSING
    ws = int(lib.dfm_workspace_bytes(B, T, N, r, flags))
    copies = max(1, int(cfg.get("cold", 0)))
    em = 2 if cfg.get("mode") == "em" else 1                      # (EM lines keep the start parameters beside the working copy)
    return copies * B * (b_in * em + b_out) + ws


def check_distinct_devices(rank_devices, world):
    """A multi-rank line must run on `world` DISTINCT GPUs: N ranks on one device would print an N-GPU line measured on one.  Returns the
    number of distinct devices; raises SystemExit when ranks share one (DFM_BENCH_ALLOW_SHARED_DEVICE=1: the forced 1-GPU groups of the
    development runs)."""
    ids = {(d or {}).get("pci_bus_id") or (d or {}).get("uuid") or f"rank{i}" for i, d in enumerate(rank_devices)}
    if len(ids) < world and os.environ.get("DFM_BENCH_ALLOW_SHARED_DEVICE") != "1":
        raise SystemExit(f"bench.py: {world} ranks on {len(ids)} distinct device(s) {sorted(ids)} -- every rank needs its own GPU "
                         "(LOCAL_RANK -> device); refusing to print a multi-GPU line")
    return len(ids)


def workload_name(N, T, r, B):
    return ("BASELINE configs[1]" if (N, T, r, B) == (200, 500, 8, 1024) else
            "BASELINE configs[2] per-GPU shard" if (N, T, r, B) == (200, 500, 8, 8192) else
            "BASELINE configs[3]" if (N, T, r) == (1000, 2000, 20) else "custom")


def run_lib_driver(args, torch):
    """--driver lib: ONE process, the library's dfm_multi object over --gpus N devices (what the Julia host binds)."""
    from dynamic_factor_models_amd import DfmMulti
    G, B, N, T, r = args.gpus, args.batch_per_gpu, args.N, args.T, args.r
    with stdout_to_stderr():                                      # (ncclCommInitAll prints RCCL's banner on stdout)
        m = DfmMulti(G, force_comm=args.force_comm)
    try:
        m.synth(20160415, 0, G * B, T, N, r, missing_prob=args.missing, pca_start=(args.mode == "em" and args.missing == 0.0))
        may = args.missing > 0.0

        def steps(k):
            if args.mode == "em":
                assert m.em(max_iter=k, tol=0.0, want_smooth=False, want_P=False, may_have_missing=may) == k
            else:
                for _ in range(k):
                    m.ks_pass(want_P=True, may_have_missing=may)
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < PREHEAT_MS:
            steps(max(2, args.steps // 4))
        steps(max(args.warmup, 1))
        blocks = []
        for _ in range(max(args.repeats, 1)):
            t0 = time.perf_counter()
            steps(args.steps)                    # (every call of the object synchronises all its GPUs before it returns)
            blocks.append(time.perf_counter() - t0)
        srt = sorted(blocks)
        el = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
        name, unit, what = METRIC["em" if args.mode == "em" else "pass"]
        b_in, b_out = algorithmic_bytes(N, T, r)
        unit_bytes = (b_in + b_out) if args.mode != "em" else (b_in + b_out + 8 * N * T + 8 * (N * r + N + 2 * r * r))
        ms = 1e3 * el / args.steps
        whole = B * unit_bytes / (ms * 1e-3) / 1e9
        out = dict(metric=f"{name}, N={N} T={T} r={r} panel", value=G * B * args.steps / el, unit=unit, n_gpus=G, steps=args.steps,
                   warmup=args.warmup, ms_per_step=ms, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64",
                   data="synthetic (generated on the owning GPU by dfm_multi_synth, Philox4x32-10 keyed by (seed, global replicate))",
                   config=dict(workload=workload_name(N, T, r, B) + f": synthetic panel N={N} T={T} r={r}, batch={B} replicates per GPU, {what}",
                               N=N, T=T, r=r, batch_per_gpu=B, global_batch=G * B, missing=args.missing, mode=args.mode,
                               driver="lib (dfm_multi: one process, one host thread per GPU, library-owned RCCL communicator"
                                      + (", ncclAllGather of {loglik, active} per EM iteration)" if m.has_comm else ", no communicator)"),
                               parallelism=f"replicate-sharded x{G} inside libdfmhip.so", has_comm=m.has_comm, multi_ngpu=m.ngpu),
                   timing=dict(repeats=len(blocks), ms_per_step_blocks=[round(1e3 * b / args.steps, 5) for b in blocks],
                               note="wall clock of the synchronising library calls (host-side thread fork/join and the per-iteration "
                                    "D2H of the gathered convergence state included)"),
                   roofline=dict(bound="hbm", kernel=None, achieved=whole, peak=HBM_PEAK_GBS, unit="GB/s", frac=whole / HBM_PEAK_GBS, traffic=None,
                                 whole_step=dict(bytes_per_unit=unit_bytes, achieved=whole, frac=whole / HBM_PEAK_GBS)),
                   cpu_baseline=None, source_hash=source_hash())
        print(json.dumps(out), flush=True)       # (before the object's teardown: RCCL's exit path must not eat the line)
    finally:
        m.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=9)
    ap.add_argument("--batch-per-gpu", type=int, default=1024)
    ap.add_argument("--N", type=int, default=200)
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--r", type=int, default=8)
    ap.add_argument("--missing", type=float, default=0.0)
    ap.add_argument("--mode", choices=("pass", "em", "pca"), default="pass")
    ap.add_argument("--driver", choices=("torch", "lib"), default="torch")
    ap.add_argument("--force-comm", action="store_true", help="--driver lib: build the RCCL communicator also for one GPU")
    ap.add_argument("--gather", choices=("block", "step"), default="block",
                    help="multi-rank pass mode: all-gather the log-likelihoods once per timed block (default) or after every pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the product path has no CPU fallback")
    what, why = launch_plan(args.gpus, os.environ, torch.cuda.device_count(), args.driver)
    if what == "error":
        raise SystemExit("bench.py: " + why)
    if what == "spawn":
        print("bench.py: " + why, file=sys.stderr)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.driver == "lib":
        if args.mode == "pca":
            raise SystemExit("--driver lib times --mode pass or --mode em")
        return run_lib_driver(args, torch)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("DFM_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        # The step's collective moves 8 KB per rank.  RCCL's default channel count puts its kernel on several CUs at the moment
        # the next pass (one persistent workgroup per CU, 146 KB of LDS: nothing fits beside it) wants all of them, and every
        # workgroup that has to wait delays the whole pass.  One channel = one contended CU: measured with a forced 1-rank group
        # 3.89 -> 4.14 M passes/s (4.59 M without the collective).  The user's own setting wins.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "1")
        with stdout_to_stderr():                                  # (RCCL's banner goes to stderr, not in front of / behind the JSON line)
            dist.init_process_group(backend="nccl", device_id=dev)
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)                                # creates the communicator now
            torch.cuda.synchronize()
        world = dist.get_world_size()                             # n_gpus in the line is what the job really has
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {world} ranks")
        # which physical device every rank computes on (a SCALE record must show N DISTINCT GPUs, not N ranks on one)
        pr_ = torch.cuda.get_device_properties(dev)
        mine = dict(rank=rank, local_device=local_rank, name=pr_.name, pci_bus_id=getattr(pr_, "pci_bus_id", None), uuid=str(getattr(pr_, "uuid", "")))
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)
        check_distinct_devices(rank_devices, world)

    from dynamic_factor_models_amd import DfmContext, shard
    ctx = DfmContext(local_rank)
    B, N, T, r = args.batch_per_gpu, args.N, args.T, args.r
    default_line = (args.mode, B, N, T, r, args.missing) == ("pass", 1024, 200, 500, 8, 0.0)

    wl = Workload(torch, dist, ctx, shard, world, rank, dev, B, N, T, r, args.missing, args.mode, gather=args.gather)
    res = wl.run(args.steps, args.warmup, args.repeats)
    # clocks / power while the bench batch runs: enqueue ~0.5 s of steps, read rocm-smi meanwhile (rank 0)
    telemetry = None
    if rank == 0:
        if args.mode != "em":                                     # (no collective in these steps: rank 0 may run them alone)
            wl.steps(int(min(4000, max(50, 500.0 / max(res["ms_per_step"], 1e-3)))), profile=True)
        telemetry = device_telemetry()                            # (EM mode: read right behind the timed blocks)
        torch.cuda.synchronize()
    ceiling = None
    if rank == 0 and world == 1:
        try:
            pr = ctx.hbm_probe(1 << 30, 10)
            ceiling = dict(read_dma_gbs=pr["read_dma"], copy_gbs=pr["copy"], write_gbs=pr["write"],
                           note="dfm_hbm_probe on this device right after the timed blocks: 1 GiB, 10 launches each -- read-only LDS-DMA "
                                "ring (the collapse's pattern), 16-byte copy (read + write counted), write-only")
        except Exception as e:  # noqa: BLE001
            ceiling = dict(error=str(e))

    out = None
    if rank == 0:
        roofline = wl.roofline(res)
        roofline["ceiling_measured"] = ceiling
        if ceiling and "copy_gbs" in ceiling and roofline.get("whole_step"):
            roofline["whole_step"]["frac_of_measured_copy_ceiling"] = roofline["whole_step"]["achieved"] / ceiling["copy_gbs"]
        cpu = None
        if world == 1 and not args.no_cpu_baseline and args.mode == "pass":
            from oracle import c_oracle as co
            S = min(B, max(256, 8 * co.num_threads()))
            ph = wl.panel[:S].cpu().numpy()
            pr = [p[:S].cpu().numpy() for p in wl.params]
            cpu = cpu_baseline(ph, pr, args.cpu_seconds)
        name, unit, what_step = METRIC[args.mode]
        coll = {"pass": " + ONE all_gather(loglik) per timed block of K passes (north_star prescribes the collective per EM iteration, not per "
                        "pass; secondary.pass_gather_every_step times the per-step form)",
                "em": " + all_gather({loglik, active}) per EM iteration", "pca": ""}[args.mode]
        srt = res["sorted"]
        out = dict(metric=f"{name}, N={N} T={T} r={r} panel", value=res["value"], unit=unit, n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=res["ms_per_step"], higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64",
                   data="synthetic (device-generated, Philox4x32-10 keyed by (seed, global replicate))",
                   config=dict(workload=workload_name(N, T, r, B) + f": synthetic panel N={N} T={T} r={r}, batch={B} replicates per GPU, {what_step}"
                                        + (f", {args.missing:.0%} cells missing" if args.missing > 0 else ", balanced"),
                               N=N, T=T, r=r, batch_per_gpu=B, global_batch=world * B, missing=args.missing, mode=args.mode,
                               parallelism=f"replicate-sharded x{world}" + (coll if world > 1 else ""),
                               process_group=(dict(backend=dist.get_backend(), world_size=dist.get_world_size(),
                                                   ranks_per_node=int(os.environ.get("LOCAL_WORLD_SIZE", world)),
                                                   devices=rank_devices,
                                                   distinct_devices=len({(d or {}).get("pci_bus_id", i) for i, d in enumerate(rank_devices)}),
                                                   nccl_max_nchannels=os.environ.get("NCCL_MAX_NCHANNELS"))
                                              if (world > 1 or force_dist) else None)),
                   timing=dict(repeats=len(res["blocks"]), statistic="median of the timed blocks (each: K steps between fences, MAX over ranks)",
                               preheat=f">= {PREHEAT_MS:.0f} ms of untimed steps before the W warm-up steps ({res['preheat_steps']} steps)",
                               ms_per_step_min=1e3 * srt[0] / args.steps, ms_per_step_max=1e3 * srt[-1] / args.steps,
                               value_min=world * B * args.steps / srt[-1], value_max=world * B * args.steps / srt[0],
                               ms_per_step_blocks=[round(1e3 * b / args.steps, 5) for b in res["blocks"]]),
                   roofline=roofline, cpu_baseline=cpu, device=telemetry, host_cores=os.cpu_count(), source_hash=source_hash())
    wl.free()

    # ---- the other lines of the path, driver-visible (default invocation only; about a minute together) ----
    plan = secondary_plan(world, default_line, args.no_secondary)
    sec = {}
    for key in plan:
        t0 = time.perf_counter()
        try:
            if key in ("c1_als", "c5_boot", "varp_em", "ar_em", "c1_em", "c1_em_p4"):
                continue                                          # (measured by the calls below)
            if world == 1:
                cfg, k, w = next((c, kk, ww) for kk_, c, kk, ww in SECONDARY if kk_ == key)
                gather = "block"
            else:
                cfg, k, w, gather = MULTI_SECONDARY[key]
            s = Workload(torch, dist, ctx, shard, world, rank, dev, cfg["B"], cfg["N"], cfg["T"], cfg["r"], cfg["missing"], cfg["mode"], gather=gather,
                         cold=cfg.get("cold", 0))
            rs = s.run(k, w, 3, preheat_ms=20.0)
            if rank == 0:
                rf = s.roofline(rs)
                sec[key] = dict(workload=workload_name(cfg["N"], cfg["T"], cfg["r"], cfg["B"]), **cfg, value=rs["value"], unit=METRIC[cfg["mode"]][1],
                                n_gpus=world, global_batch=world * cfg["B"], ms_per_step=rs["ms_per_step"],
                                ms_per_step_blocks=[round(1e3 * b / k, 5) for b in rs["blocks"]], steps=k,
                                whole_step=(rf.get("whole_step") or {}).get("frac"), dominant=rf["kernel"],
                                dominant_frac=rf.get("frac"), kernels_ms=rf["kernels_ms"], gram=rf.get("gram"), seconds=None)
                if cfg.get("cold"):
                    sec[key]["note"] = (f"the headline's pass cycling over {cfg['cold']} distinct resident batches ({cfg['cold']} x 1.0 GB: panels, parameters and "
                                        "output buffers all distinct): nothing a pass wrote is still in the 256-MB Infinity Cache when the next pass runs")
                if world > 1:
                    sec[key]["collective"] = ("all_gather({loglik, active}) per EM iteration" if cfg["mode"] == "em" else
                                              "all_gather(loglik) after every pass" if gather == "step" else "one all_gather(loglik) per timed block")
            s.free()
        except Exception as e:  # noqa: BLE001  (a secondary line must never take the headline down)
            sec[key] = dict(error=f"{type(e).__name__}: {e}")
        if key in sec:
            sec[key]["seconds"] = round(time.perf_counter() - t0, 2)
    if "c1_als" in plan and rank == 0:
        try:
            sec.update(config1_config5_lines(torch, ctx, dev))
        except Exception as e:  # noqa: BLE001
            sec["c1_als"] = dict(error=f"{type(e).__name__}: {e}")
    if "varp_em" in plan and rank == 0:
        try:
            sec.update(f3_lines(torch, ctx, dev))
        except Exception as e:  # noqa: BLE001
            sec["varp_em"] = dict(error=f"{type(e).__name__}: {e}")
    if "c1_em" in plan and rank == 0:
        try:
            sec.update(c1_em_lines(torch, ctx, dev))
        except Exception as e:  # noqa: BLE001
            sec["c1_em"] = dict(error=f"{type(e).__name__}: {e}")
    if plan and rank == 0:
        out["secondary"] = sec
        # driver-visible scalars: the cache-cold figure of the headline's kernel INSIDE the roofline object (the driver's record keeps
        # the scalars of `roofline`; `secondary` is cut off by its stored tail), and one compact [ms per step, fraction of the HBM peak]
        # pair per secondary line at the very END of the line
        cold = sec.get("pass_cold") or {}
        if cold.get("whole_step") is not None:
            out["roofline"]["cold_frac"] = cold["whole_step"]
            out["roofline"]["cold_ms_per_step"] = cold["ms_per_step"]
        for key_, tag in (("c4", "frac_c4"), ("c4_em", "frac_c4_em"), ("missing10", "frac_missing10"), ("missing10_b8192", "frac_missing10_b8192"),
                          ("em", "frac_em"), ("c4_missing10", "frac_c4_missing10")):
            if (sec.get(key_) or {}).get("whole_step") is not None:
                out["roofline"][tag] = sec[key_]["whole_step"]
        out["secondary_summary"] = {k_: [None if v.get("ms_per_step") is None else round(v["ms_per_step"], 4),
                                         None if v.get("whole_step") is None else round(v["whole_step"], 4)] if "error" not in v else "error"
                                    for k_, v in sec.items()}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist is not None and dist.is_initialized():      # (also the forced 1-rank group of DFM_BENCH_FORCE_DIST=1)
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
