import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from dynamic_factor_models_amd import DfmContext
ctx = DfmContext(); dev = torch.device("cuda", ctx.device)
for (B, N, T, r) in [(1024, 200, 500, 4), (1024, 139, 222, 4), (4096, 139, 222, 4), (1024, 200, 500, 6)]:
    panel, params = bench.synth_on_device(torch, dev, B, N, T, r, seed=1, missing=0.1)
    pp = [x.clone() for x in params]
    for _ in range(2): ctx.em_step_batch(panel, *pp, may_have_missing=True)
    ctx.profile_enable(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ctx.em_step_batch(panel, *pp, may_have_missing=True)
    torch.cuda.synchronize(); s = (time.perf_counter() - t0) / 10
    print((B, N, T, r), "EM iteration 10% missing:", round(1e3 * s, 3), "ms", {k: round(v[0] / max(v[1], 1), 4) for k, v in ctx.profile_read().items() if v[1]}, flush=True)
    ctx.profile_enable(False)
