#!/usr/bin/env python
"""bench.py -- Kalman-smoother passes/sec on synthetic N=200 T=500 r=8 panels (BASELINE.json metric).

A "step" = one full Kalman-smoother pass (filter + RTS smoother + log-likelihood)
over ONE batch of replicates already resident in HBM.  N=1: BASELINE configs[1] (batch = 1024
replicates on 1 x MI355X).  N>1: one process per GPU (torch.distributed, backend nccl = RCCL), the
replicate batch is sharded with the same 1024 replicates per GPU (weak scaling), and -- as
north_star prescribes -- one all_gather of the per-replicate log-likelihoods closes every step.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats 9] [--batch-per-gpu 1024] [--missing 0.0]
                  [--mode pass|em|pca]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Timing: W untimed warm-up steps, then `--repeats` blocks of EXACTLY K steps, each block bracketed by barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over ranks; `value` / `ms_per_step` are the MEDIAN block
(boxes of this pool differ by +-8 % and a 5 ms region is noisy), min / max / every block are printed beside them.
Panels come from the product's own device generator (dfm_synth_panels_dev: counter-based Philox keyed by (seed,
global replicate index); checked cell by cell against its host restatement in the GPU tests).

--mode em : a step = ONE EM ITERATION (E-step pass + M-step, SURVEY §8(d) "EM iteration") of the replicate-sharded
            driver shard.em_batch_sharded -- dfm_em_iterate_batch_dev on the shard, then the all-gather of {loglik,
            active} every iteration (north_star's collective).  Reported as EM iterations/s (a different metric).
--mode pca: a step = the PCA initialisation (pca_score + OLS start) of the batch (dfm_pca_init_batch_dev).

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, HIP-event timed on the
launch stream inside this script) and "cpu_baseline" (oracle/dfm_oracle.c, the C restatement, timed
on the host cores of this box on a bounded sample of the same workload; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
FP64_MATRIX_PEAK_TFLOPS = 78.6  # AMD MI355X spec sheet, FP64 matrix (the guide's table has no fp64 row; measured
                                # v_mfma_f64_4x4x4 issue rate here: 68 TF/s, scripts/microbench/mfma64.hip)


def algorithmic_bytes(N, T, r):
    """every input read once, every output written once, no scratch."""
    inputs = 8 * (N * T + N * r + N + 2 * r * r + r + r * r)
    outputs = 8 * (T * r + T * r * (r + 1) // 2 + 1)
    return inputs, outputs


def synth_on_device(torch, dev, B, N, T, r, seed, missing=0.0):
    """(scripts/): SURVEY §8(d) replicates from the product's device generator; returns (panel, params)."""
    from dynamic_factor_models_amd import DfmContext
    c = DfmContext(dev.index or 0)
    try:
        out = c.synth_panels(seed, 0, B, T, N, r, missing_prob=missing)
        torch.cuda.synchronize()
    finally:
        c.close()
    return out


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
    """HBM bytes per launch of `kernel` from the rocprofv3 --pmc passes (scripts/gpu_profile.sh -> profiles/r02/
    pmc_traffic.json), ONLY when that file was produced by exactly this source tree on this workload; else None."""
    for name in ("pmc_traffic.json", "pmc_traffic_c4.json"):      # headline workload; BASELINE config 4
        try:
            with open(os.path.join(ROOT, "profiles", "r02", name)) as fh:
                d = json.load(fh)
            if d.get("_source_hash") == source_hash() and d.get("_workload") == workload_key:
                for k in (kernel, kernel.replace("collapse_wide_kernel", "collapse_wide2_kernel")):   # (profile scope -> rocprof name)
                    if k in d:
                        return d[k].get("hbm_bytes_per_launch")
        except Exception:  # noqa: BLE001
            pass
    return None


def cpu_baseline(panel_host, params_host, target_seconds=12.0):
    """Time the C restatement (oracle/dfm_oracle.c, OpenMP over replicates) on this box's host cores on a bounded
    sample of the same workload: all threads (>= 8 replicates per thread, persistent output buffers touched before the
    clock starts) and one thread."""
    import numpy as np
    from oracle import c_oracle as co
    cores = co.num_threads()
    S, T, N = panel_host.shape
    r = params_host[0].shape[2]
    out = (np.zeros((S, T, r)), np.zeros((S, T, r * (r + 1) // 2)), np.zeros(S))
    co.ks_pass_batch(panel_host, *params_host, out=out)          # warm: thread pool up, every page touched
    # the box may give this container fewer CPUs than it shows (cgroup quota): take the thread count that is fastest
    # on a short probe, so that the "all cores" figure is not an oversubscription artefact
    probe = {}
    for n in sorted({cores, max(cores // 2, 1), max(cores // 4, 1), max(cores // 8, 1), min(cores, 32), min(cores, 16), min(cores, 8)}):
        Sn = min(S, max(8 * n, 16))
        tp = time.perf_counter()
        co.ks_pass_batch(panel_host[:Sn], *[p[:Sn] for p in params_host], out=tuple(o[:Sn] for o in out), nthreads=n)
        probe[n] = Sn / (time.perf_counter() - tp)
    cores = max(probe, key=probe.get)
    S = min(S, max(8 * cores, 16))
    panel_host = panel_host[:S]; params_host = [p[:S] for p in params_host]; out = tuple(o[:S] for o in out)
    co.ks_pass_batch(panel_host, *params_host, out=out, nthreads=cores)
    done, t0 = 0, time.perf_counter()
    while True:
        co.ks_pass_batch(panel_host, *params_host, out=out)
        done += S
        el = time.perf_counter() - t0
        if el >= 0.75 * target_seconds:
            break
    S1 = min(S, 16)
    sub = (panel_host[:S1],) + tuple(p[:S1] for p in params_host)
    out1 = tuple(o[:S1] for o in out)
    co.ks_pass_batch(*sub, out=out1, nthreads=1)
    d1, t1 = 0, time.perf_counter()
    while True:
        co.ks_pass_batch(*sub, out=out1, nthreads=1)
        d1 += S1
        e1 = time.perf_counter() - t1
        if e1 >= 0.25 * target_seconds:
            break
    co.ks_pass_batch(*sub, out=out1, nthreads=cores)             # leave the pool at its default size
    return dict(value=done / el, unit="passes/s", cores=cores, kind="port", per_thread=done / el / cores,
                thread_probe={str(k): round(v, 1) for k, v in sorted(probe.items())},
                single_thread=dict(value=d1 / e1, cores=1, sample=f"{d1} passes in {e1:.1f} s"),
                sample=f"{done} passes ({S} distinct replicates of the bench batch = {S / cores:.1f} per thread, repeated) "
                       f"in {el:.1f} s; oracle/dfm_oracle.c, gcc -O2 -fopenmp, {cores} threads, outputs into "
                       f"persistent buffers")


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        dist.init_process_group(backend="nccl", device_id=dev)

    from dynamic_factor_models_amd import DfmContext, shard
    ctx = DfmContext(local_rank)

    B, N, T, r = args.batch_per_gpu, args.N, args.T, args.r
    seed = 20160415
    # this rank's replicates [rank B, (rank + 1) B) of the job's world * B (shard.replicate_range), generated where they live
    panel, params = ctx.synth_panels(seed, rank * B, B, T, N, r, missing_prob=args.missing)
    may_missing = args.missing > 0.0
    f = torch.empty((B, T, r), dtype=torch.float64, device=dev)
    P = torch.empty((B, T, r * (r + 1) // 2), dtype=torch.float64, device=dev)
    ll = torch.empty((B,), dtype=torch.float64, device=dev)
    ll_all = torch.empty((world * B,), dtype=torch.float64, device=dev) if distributed else None
    em_params = None
    if args.mode == "em":
        if may_missing:
            em_params = [p.clone() for p in params]                 # DGP parameters as the start (PCA needs a balanced panel)
        else:
            em_params = list(ctx.pca_init_batch(panel, r, want_factors=False)[:6])

    def run_pass():
        ctx.ks_pass_batch(panel, *params, may_have_missing=may_missing, out=(f, P, ll))

    def steps(k, profile=False):
        if args.mode == "pass":
            for _ in range(k):
                run_pass()
                if distributed and not profile:   # north_star: a single RCCL all-gather of the replicates' log-likelihoods
                    dist.all_gather_into_tensor(ll_all, ll)
        elif args.mode == "em":
            # k EM iterations of the sharded driver: dfm_em_iterate_batch_dev + the all-gather of {loglik, active} EVERY
            # iteration (tol = 0: no early stop, so exactly k iterations are timed)
            shard.em_batch_sharded(ctx, panel, *em_params, B_global=world * B, max_iter=k, tol=0.0, want_smooth=False,
                                   may_have_missing=may_missing)
        else:
            for _ in range(k):
                ctx.pca_init_batch(panel, r, want_factors=False)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    steps(max(args.warmup, 1) if args.mode == "em" else args.warmup)
    blocks = []
    for _ in range(max(args.repeats, 1)):
        fence()
        t0 = time.perf_counter()
        steps(args.steps)
        fence()
        el = time.perf_counter() - t0
        if distributed:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        blocks.append(el)
    srt = sorted(blocks)
    elapsed = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    ms_per_step = 1e3 * elapsed / args.steps
    per_s = world * B * args.steps / elapsed
    if args.mode == "pass":
        assert bool(torch.isfinite(ll).all()), "non-finite log-likelihood in the bench batch"

    # ---- roofline leg: K more steps with a HIP-event pair around every kernel launch (on the launch stream)
    ctx.profile_enable(True)
    steps(args.steps, profile=True)
    prof = ctx.profile_read()
    ctx.profile_enable(False)

    if rank == 0:
        b_in, b_out = algorithmic_bytes(N, T, r)
        panel_b = 8 * (N * T + N * r + N)                      # panel + loadings + idiosyncratic variances
        npack = r * (r + 1) // 2
        # algorithmic bytes per launch (DESIGN.md "Kernels"): compulsory inputs read once + outputs written once
        kern_bytes = {"collapse_mfma_kernel": B * panel_b, "collapse_dma_kernel": B * panel_b,
                      "collapse_wide_kernel": B * panel_b, "collapse_kernel": B * panel_b,
                      "pass_fused_kernel": B * (b_in + b_out),
                      "recursion_kernel": B * (b_in - panel_b + b_out),
                      "meanscan_kernel": B * (b_in - panel_b + 8 * (T * r + 1)),
                      "pfill_kernel": B * 8 * T * npack,
                      "mstep_mfma_kernel": B * 8 * (N * T + T * r), "mstep_lam_kernel": B * 8 * (N * T + T * (r + npack)),
                      "gram_kernel": B * 8 * (N * r + N), "cov_kernel": B * 8 * (3 * r * r + r)}
        avg = {k: v[0] / v[1] for k, v in prof.items()}
        fused = "collapse_mfma_kernel" in avg and "cov_kernel" not in avg and "pfill_kernel" not in avg
        if fused:   # the covariance workgroups + P_smooth fill ride in the collapse launch: it also writes P_smooth
            kern_bytes["collapse_mfma_kernel"] += B * 8 * (T * npack + 3 * r * r + r)
        workload_key = f"{args.mode}:B{B}:N{N}:T{T}:r{r}:m{args.missing}"
        if args.mode == "pca" and "gram_xx_kernel" in avg:
            dom = max(avg, key=avg.get)
            flops = {"gram_xx_kernel": 2.0 * T * N * N * B}
            if dom in flops:
                ach = flops[dom] / (avg[dom] * 1e-3) / 1e12
                roofline = dict(bound="mfma", kernel=dom, achieved=ach, peak=FP64_MATRIX_PEAK_TFLOPS, unit="TFLOP/s",
                                frac=ach / FP64_MATRIX_PEAK_TFLOPS, traffic=measured_traffic(dom, workload_key),
                                avg_launch_ms=avg[dom], flops_per_launch=flops[dom],
                                kernels_ms={k: round(v, 4) for k, v in avg.items()},
                                note="X'X of the batch (T x N by N x N per replicate) on v_mfma_f64_16x16x4; the subspace "
                                     "iteration behind it (pca_kernel) is latency-bound small-matrix work")
            else:
                roofline = dict(bound="hbm", kernel=dom, achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None,
                                avg_launch_ms=avg[dom], kernels_ms={k: round(v, 4) for k, v in avg.items()},
                                note="pca_kernel (subspace iteration, Rayleigh-Ritz, OLS start) dominates: latency-bound "
                                     "small-matrix work on an L2-resident Gram matrix -- no roofline claim")
        else:
            cands = [k for k in avg if k in kern_bytes]
            dom = max(cands, key=avg.get)
            achieved = kern_bytes[dom] / (avg[dom] * 1e-3) / 1e9
            unit_bytes = (b_in + b_out) if args.mode == "pass" else (b_in + b_out + 8 * N * T + 8 * (N * r + N + 2 * r * r))
            roofline = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=achieved / HBM_PEAK_GBS, traffic=measured_traffic(dom, workload_key),
                            avg_launch_ms=avg[dom], bytes_per_launch=kern_bytes[dom],
                            kernels_ms={k: round(v, 4) for k, v in avg.items()},
                            note=("pass_fused_kernel = the whole pass in one launch: every input read once, every output written once"
                                  if dom == "pass_fused_kernel" else
                                  "collapse_mfma_kernel = streaming collapse + the covariance workgroups and the P_smooth fill at the "
                                  "front of the same grid: algorithmic bytes = panel + Lam + R read, P_smooth written" if fused else
                                  "sequential path (panel with missing cells): collapse_kernel streams the panel once, the recursion "
                                  "kernel is a chain of T dependent r x r inversions per replicate -- latency-bound, not HBM-bound"
                                  if "recursion_kernel" in avg else
                                  "dominant kernel of this mode by HIP-event time"),
                            whole_step=dict(bytes_per_unit=unit_bytes,
                                            achieved=B * unit_bytes / (ms_per_step * 1e-3) / 1e9,
                                            frac=B * unit_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            note="SURVEY §8(d) algorithmic bytes per " + ("pass" if args.mode == "pass" else "EM iteration")
                                                 + " x units/s of ONE GPU (median wall clock of the timed blocks) / HBM peak"))
            roofline["whole_pass"] = roofline["whole_step"]      # round-1 key kept for the driver's readers
        cpu = None
        if world == 1 and not args.no_cpu_baseline and args.mode == "pass":
            import numpy as np
            from oracle import c_oracle as co
            S = min(B, max(256, 8 * co.num_threads()))
            ph = panel[:S].cpu().numpy()
            pr = [p[:S].cpu().numpy() for p in params]
            cpu = cpu_baseline(ph, pr, args.cpu_seconds)
        metric = {"pass": f"Kalman-smoother passes/sec, N={N} T={T} r={r} panel",
                  "em": f"EM iterations/sec (E-step pass + M-step), N={N} T={T} r={r} panel",
                  "pca": f"PCA initialisations/sec (pca_score + OLS start), N={N} T={T} r={r} panel"}[args.mode]
        unit = {"pass": "passes/s", "em": "EM iterations/s", "pca": "initialisations/s"}[args.mode]
        what = {"pass": "one full Kalman-smoother pass per step", "em": "one EM iteration (pass + M-step) per step",
                "pca": "one PCA initialisation per step"}[args.mode]
        coll = {"pass": " + all_gather(loglik) per step", "em": " + all_gather({loglik, active}) per EM iteration", "pca": ""}[args.mode]
        out = dict(metric=metric, value=per_s, unit=unit, n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=ms_per_step, higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64",
                   data="synthetic (device-generated, Philox4x32-10 keyed by (seed, global replicate))",
                   config=dict(workload=("BASELINE configs[1]" if (N, T, r, B) == (200, 500, 8, 1024) else
                                         "BASELINE configs[2] per-GPU shard" if (N, T, r, B) == (200, 500, 8, 8192) else
                                         "BASELINE configs[3]" if (N, T, r) == (1000, 2000, 20) else "custom")
                                        + f": synthetic panel N={N} T={T} r={r}, "
                                        f"batch={B} replicates per GPU, {what}"
                                        + (f", {args.missing:.0%} cells missing" if may_missing else ", balanced"),
                               N=N, T=T, r=r, batch_per_gpu=B, global_batch=world * B, missing=args.missing, mode=args.mode,
                               parallelism=f"replicate-sharded x{world}" + (coll if distributed else "")),
                   timing=dict(repeats=len(blocks), statistic="median of the timed blocks (each: K steps between fences, MAX over ranks)",
                               ms_per_step_min=1e3 * srt[0] / args.steps, ms_per_step_max=1e3 * srt[-1] / args.steps,
                               value_min=world * B * args.steps / srt[-1], value_max=world * B * args.steps / srt[0],
                               ms_per_step_blocks=[round(1e3 * b / args.steps, 5) for b in blocks]),
                   roofline=roofline, cpu_baseline=cpu, host_cores=os.cpu_count(), source_hash=source_hash())
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
