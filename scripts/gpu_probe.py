#!/usr/bin/env python
"""GPU bring-up probe (run through gpurun): parity of the balanced fast path against the C oracle and
against the general path, then per-kernel timings of the collapse variants at BASELINE config 2.
Writes gpurun_out/probe.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from dynamic_factor_models_amd import DfmContext  # noqa: E402
from oracle import c_oracle as co  # noqa: E402
from oracle import kalman_oracle as ko  # noqa: E402

dev = torch.device("cuda", 0)
out = {"parity": [], "timing": []}


def batch(B, N, T, r, seed=ko.SEED0, weak=False):
    reps = [ko.synth_replicate(b, N, T, r, seed=seed) for b in range(B)]
    panel = np.stack([x for x, _ in reps])
    st = {k: np.stack([p[k] for _, p in reps]) for k in reps[0][1]}
    if weak:   # weak signal + persistent factors: slow Riccati convergence (many distinct covariance steps)
        st["Lam"] = st["Lam"] * 0.05
        st["A"] = np.stack([np.diag(np.linspace(0.9, 0.99, r))] * B)
        st["Q"] = np.stack([np.eye(r) * 0.05] * B)
    return panel, st


def run(ctx, panel, st):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f, P, ll = ctx.ks_pass_batch(t(panel), t(st["Lam"]), t(st["R"]), t(st["A"]), t(st["Q"]), t(st["mu0"]),
                                 t(st["P0"]), may_have_missing=False)
    torch.cuda.synchronize()
    return f.cpu().numpy(), P.cpu().numpy(), ll.cpu().numpy()


def ctx_with(**env):
    for k, v in env.items():
        os.environ[k] = str(v)
    c = DfmContext(0)
    for k in env:
        os.environ.pop(k, None)
    return c


import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--skip-parity", action="store_true")
ap.add_argument("--variants", default="0,1,2,3,4")
ap.add_argument("--no-side", default="0")
ap.add_argument("--scan-abl", default="0")
ap.add_argument("--sub", default="0")
cli = ap.parse_args()

cases = [(16, 200, 500, 8, False), (5, 40, 50, 4, False), (3, 64, 48, 12, False), (2, 100, 40, 20, False),
         (4, 30, 41, 3, False), (3, 20, 25, 1, False), (3, 50, 7, 2, False), (2, 20, 1, 2, False),
         (2, 20, 2, 4, False), (3, 10, 300, 2, True), (2, 12, 600, 4, True), (9, 200, 222, 8, False),
         (2, 1000, 64, 8, False), (2, 400, 50, 16, False)]
fast = ctx_with()
gen = ctx_with(DFM_FORCE_GENERAL=1)
for (B, N, T, r, weak) in ([] if cli.skip_parity else cases):
    panel, st = batch(B, N, T, r, weak=weak)
    ref = co.ks_pass_batch(panel, st["Lam"], st["R"], st["A"], st["Q"], st["mu0"], st["P0"])
    rec = dict(B=B, N=N, T=T, r=r, weak=weak)
    for name, c in (("fast", fast), ("general", gen)):
        try:
            f, P, ll = run(c, panel, st)
            rec[name] = dict(ll=float(np.max(np.abs(ll - ref[2]) / np.abs(ref[2]))),
                             f=float(np.abs(f - ref[0]).max() / np.abs(ref[0]).max()),
                             P=float(np.abs(P - ref[1]).max() / np.abs(ref[1]).max()))
        except Exception as e:  # noqa: BLE001
            rec[name] = "ERR " + str(e)[:200]
    out["parity"].append(rec)
    print(rec, flush=True)

# ---------------------------------------------------------------- timing at config 2
sys.path.insert(0, ROOT)
import bench  # noqa: E402

B, N, T, r = 1024, 200, 500, 8
panel, params = bench.synth_on_device(torch, dev, B, N, T, r, seed=1)
f = torch.empty((B, T, r), dtype=torch.float64, device=dev)
P = torch.empty((B, T, r * (r + 1) // 2), dtype=torch.float64, device=dev)
ll = torch.empty((B,), dtype=torch.float64, device=dev)
runs = [("general", dict(DFM_FORCE_GENERAL=1))]
for ns in cli.no_side.split(","):
    for ab in cli.scan_abl.split(","):
        for sb in cli.sub.split(","):
            runs += [(f"fast_v{v}_noside{ns}_abl{ab}_sub{sb}", dict(DFM_COLLAPSE_VARIANT=v, DFM_NO_SIDE=ns, DFM_SCAN_ABL=ab, DFM_SUBBATCH=sb)) for v in cli.variants.split(",")]
for tag, env in runs:
    c = ctx_with(**env)
    for _ in range(5):
        c.ks_pass_batch(panel, *params, may_have_missing=False, out=(f, P, ll))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 30
    for _ in range(K):
        c.ks_pass_batch(panel, *params, may_have_missing=False, out=(f, P, ll))
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / K
    c.profile_enable(True)
    for _ in range(10):
        c.ks_pass_batch(panel, *params, may_have_missing=False, out=(f, P, ll))
    prof = {k: round(v[0] / v[1], 4) for k, v in c.profile_read().items()}
    c.profile_enable(False)
    rec = dict(tag=tag, ms_per_pass_batch=round(ms, 4), passes_per_s=round(B / ms * 1e3), kernels_ms=prof,
               ll0=float(ll[0].item()))
    out["timing"].append(rec)
    print(rec, flush=True)
    c.close()

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w") as fh:
    json.dump(out, fh, indent=1)
