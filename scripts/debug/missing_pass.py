"""Timing of the general (missing-cell) path per kernel on the config-2 shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from dynamic_factor_models_amd import DfmContext
ctx = DfmContext()
dev = torch.device("cuda", ctx.device)
B, N, T, r = 1024, 200, 500, 8
for miss in (0.1, 0.5, 0.01):
    panel, params = bench.synth_on_device(torch, dev, B, N, T, r, seed=1, missing=miss)
    for _ in range(2):
        ctx.ks_pass_batch(panel, *params, may_have_missing=True)
    ctx.profile_enable(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        ctx.ks_pass_batch(panel, *params, may_have_missing=True)
    torch.cuda.synchronize(); s = (time.perf_counter() - t0) / 10
    prof = {k: round(v[0] / max(v[1], 1), 4) for k, v in ctx.profile_read().items() if v[1]}
    ctx.profile_enable(False)
    print(f"missing {miss}: {1e3 * s:.3f} ms/pass-batch  {prof}", flush=True)
