import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from dynamic_factor_models_amd import DfmContext
ctx = DfmContext(); dev = torch.device("cuda", ctx.device)
N, T, r = 200, 500, int(os.environ.get("RR", "8"))
for B in (1024, 2048, 3072, 4096, 8192):
    panel, params = bench.synth_on_device(torch, dev, B, N, T, r, seed=1, missing=0.1)
    for _ in range(2): ctx.ks_pass_batch(panel, *params, may_have_missing=True)
    ctx.profile_enable(True)
    torch.cuda.synchronize()
    for _ in range(5): ctx.ks_pass_batch(panel, *params, may_have_missing=True)
    torch.cuda.synchronize()
    print("B", B, {k: round(v[0] / max(v[1], 1), 4) for k, v in ctx.profile_read().items() if v[1]}, flush=True)
    ctx.profile_enable(False)
    del panel, params
