#!/usr/bin/env python
"""bench.py -- Kalman-smoother passes/sec on synthetic N=200 T=500 r=8 panels (BASELINE.json metric).

A "step" = one full Kalman-smoother pass (filter + RTS smoother + log-likelihood)
over ONE batch of replicates already resident in HBM.  N=1: BASELINE configs[1] (batch = 1024
replicates on 1 x MI355X).  N>1: one process per GPU (torch.distributed, backend nccl = RCCL), the
replicate batch is sharded with the same 1024 replicates per GPU (weak scaling), and -- as
north_star prescribes -- one all_gather of the per-replicate log-likelihoods closes every step.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch-per-gpu 1024] [--missing 0.0]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel, HIP-event timed on the
launch stream inside this script) and "cpu_baseline" (oracle/dfm_oracle.c, the C restatement, timed
on the host cores of this box on a bounded sample of the same workload; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def algorithmic_bytes(N, T, r):
    """every input read once, every output written once, no scratch."""
    inputs = 8 * (N * T + N * r + N + 2 * r * r + r + r * r)
    outputs = 8 * (T * r + T * r * (r + 1) // 2 + 1)
    return inputs, outputs


def synth_on_device(torch, dev, B, N, T, r, seed, missing=0.0):
    """SURVEY.md §8(d) DGP drawn on the device with torch (input generation only -- plumbing):
    lam ~ N(0,1), A = diag(linspace(.5,.9,r)), Q = I - AA', R ~ U(.5,1.5), columns standardised.
    Returns the panel and the DGP parameters rescaled to the standardised panel."""
    g = torch.Generator(device=dev).manual_seed(seed)
    f64 = dict(dtype=torch.float64, device=dev)
    Lam = torch.randn((B, N, r), generator=g, **f64)
    a = torch.linspace(0.5, 0.9, r, **f64)
    R = 0.5 + torch.rand((B, N), generator=g, **f64)
    q = torch.sqrt(1.0 - a * a)
    f = torch.randn((B, r), generator=g, **f64)
    panel = torch.empty((B, T, N), **f64)
    sqR = torch.sqrt(R)
    for t in range(T):
        f = a * f + q * torch.randn((B, r), generator=g, **f64)
        panel[:, t, :] = torch.einsum("bnk,bk->bn", Lam, f) + sqR * torch.randn((B, N), generator=g, **f64)
    mu = panel.mean(dim=1, keepdim=True)
    sd = panel.std(dim=1, unbiased=False, keepdim=True)
    panel = ((panel - mu) / sd).contiguous()
    sdv = sd.squeeze(1)
    Lam = (Lam / sdv.unsqueeze(-1)).contiguous()
    R = (R / (sdv * sdv)).contiguous()
    A = torch.diag(a).expand(B, r, r).contiguous()
    Q = torch.diag(1.0 - a * a).expand(B, r, r).contiguous()
    mu0 = torch.zeros((B, r), **f64)
    P0 = torch.eye(r, **f64).expand(B, r, r).contiguous()
    if missing > 0.0:
        m = torch.rand((B, T, N), generator=g, **f64) < missing
        panel = torch.where(m, torch.full_like(panel, float("nan")), panel)
    return panel, (Lam, R, A, Q, mu0, P0)


def cpu_baseline(panel_host, params_host, target_seconds=12.0):
    """Time the C restatement (oracle/dfm_oracle.c, OpenMP over replicates) on this box's host
    cores on a bounded sample of the same workload."""
    from oracle import c_oracle as co
    cores = co.num_threads()
    S = panel_host.shape[0]
    co.ks_pass_batch(panel_host[:min(S, cores)], *[p[:min(S, cores)] for p in params_host])  # warm
    done, t0 = 0, time.perf_counter()
    while True:
        co.ks_pass_batch(panel_host, *params_host)
        done += S
        el = time.perf_counter() - t0
        if el >= target_seconds:
            break
    return dict(value=done / el, unit="passes/s", cores=cores, kind="port",
                sample=f"{done} passes ({S} distinct replicates of the bench batch, repeated) in {el:.1f} s; "
                       f"oracle/dfm_oracle.c, gcc -O2 -fopenmp, {cores} threads")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=1024)
    ap.add_argument("--N", type=int, default=200)
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--r", type=int, default=8)
    ap.add_argument("--missing", type=float, default=0.0)
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

    from dynamic_factor_models_amd import DfmContext
    ctx = DfmContext(local_rank)

    B, N, T, r = args.batch_per_gpu, args.N, args.T, args.r
    panel, params = synth_on_device(torch, dev, B, N, T, r, seed=20160415 + 7919 * rank, missing=args.missing)
    may_missing = args.missing > 0.0
    f = torch.empty((B, T, r), dtype=torch.float64, device=dev)
    P = torch.empty((B, T, r * (r + 1) // 2), dtype=torch.float64, device=dev)
    ll = torch.empty((B,), dtype=torch.float64, device=dev)
    ll_all = torch.empty((world * B,), dtype=torch.float64, device=dev) if distributed else None

    def step():
        ctx.ks_pass_batch(panel, *params, may_have_missing=may_missing, out=(f, P, ll))
        if distributed:   # north_star: a single RCCL all-gather of the replicates' log-likelihoods
            dist.all_gather_into_tensor(ll_all, ll)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps
    passes_per_s = world * B * args.steps / elapsed
    assert bool(torch.isfinite(ll).all()), "non-finite log-likelihood in the bench batch"

    # ---- roofline leg: same K steps again with a HIP-event pair around every kernel launch
    ctx.profile_enable(True)
    for _ in range(args.steps):
        ctx.ks_pass_batch(panel, *params, may_have_missing=may_missing, out=(f, P, ll))
    prof = ctx.profile_read()
    ctx.profile_enable(False)

    if rank == 0:
        b_in, b_out = algorithmic_bytes(N, T, r)
        panel_b = 8 * (N * T + N * r + N)                      # panel + loadings + idiosyncratic variances
        # algorithmic bytes per launch (DESIGN.md "Kernels"): compulsory inputs read once + outputs written once
        npack = r * (r + 1) // 2
        kern_bytes = {"collapse_mfma_kernel": B * panel_b, "collapse_dma_kernel": B * panel_b,
                      "collapse_wide_kernel": B * panel_b,
                      "collapse_kernel": B * panel_b,
                      "recursion_kernel": B * (b_in - panel_b + b_out),
                      "meanscan_kernel": B * (b_in - panel_b + 8 * (T * r + 1)),
                      "pfill_kernel": B * 8 * T * npack,
                      "gram_kernel": B * 8 * (N * r + N), "cov_kernel": B * 8 * (3 * r * r + r)}
        avg = {k: v[0] / v[1] for k, v in prof.items()}
        fused = "collapse_mfma_kernel" in avg and "cov_kernel" not in avg and "pfill_kernel" not in avg
        if fused:   # the covariance workgroups + P_smooth fill ride in the collapse launch: it also writes P_smooth
            kern_bytes["collapse_mfma_kernel"] += B * 8 * (T * npack + 3 * r * r + r)
        dom = max((k for k in avg if k in ("collapse_mfma_kernel", "collapse_dma_kernel", "collapse_wide_kernel", "collapse_kernel",
                                            "recursion_kernel", "meanscan_kernel")), key=avg.get)
        achieved = kern_bytes.get(dom, 0) / (avg[dom] * 1e-3) / 1e9
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")   # rocprofv3 --pmc passes (scripts/gpu_profile.sh)
        if os.path.exists(pmc_file) and (B, N, T, r, may_missing) == (1024, 200, 500, 8, False):
            try:
                with open(pmc_file) as fh:
                    traffic = json.load(fh).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        roofline = dict(bound="hbm", kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                        avg_launch_ms=avg[dom], bytes_per_launch=kern_bytes.get(dom, 0),
                        kernels_ms={k: round(v, 4) for k, v in avg.items()},
                        note=("collapse_mfma_kernel = streaming collapse + the covariance workgroups and the P_smooth fill at the "
                              "front of the same grid: algorithmic bytes = panel + Lam + R read, P_smooth written" if fused else
                              "sequential path (panel with missing cells): collapse_kernel streams the panel once, the recursion "
                              "kernel is a chain of T dependent r x r inversions per replicate -- latency-bound, not HBM-bound"
                              if "recursion_kernel" in avg else
                              "cov_kernel and pfill_kernel run beside the streaming collapse (forked stream); their "
                              "durations overlap it and each other's memory traffic"),
                        whole_pass=dict(bytes_per_pass=b_in + b_out,
                                        achieved=B * (b_in + b_out) / (ms_per_step * 1e-3) / 1e9,
                                        frac=B * (b_in + b_out) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        note="SURVEY §8(d) algorithmic bytes per pass x passes/s of ONE GPU "
                                             "(wall clock of the timed region) / HBM peak"))
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            S = min(B, 256)
            ph = panel[:S].cpu().numpy()
            pr = [p[:S].cpu().numpy() for p in params]
            cpu = cpu_baseline(ph, pr, args.cpu_seconds)
        out = dict(metric=f"Kalman-smoother passes/sec, N={N} T={T} r={r} panel", value=passes_per_s,
                   unit="passes/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step,
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=("BASELINE configs[1]" if (N, T, r) == (200, 500, 8) else
                                         "BASELINE configs[3]" if (N, T, r) == (1000, 2000, 20) else "custom")
                                        + f": synthetic panel N={N} T={T} r={r}, "
                                        f"batch={B} replicates per GPU, one full Kalman-smoother pass per step"
                                        + (f", {args.missing:.0%} cells missing" if may_missing else ", balanced"),
                               N=N, T=T, r=r, batch_per_gpu=B, global_batch=world * B, missing=args.missing,
                               parallelism=f"replicate-sharded x{world}" + (" + all_gather(loglik)" if distributed else "")),
                   roofline=roofline, cpu_baseline=cpu, host_cores=os.cpu_count())
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
