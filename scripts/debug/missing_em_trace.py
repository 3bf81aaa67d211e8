"""Workload for rocprofv3 --kernel-trace --stats: general path (10 % missing) on the config-2 shape -- smoother passes,
EM iterations, and the VAR(4) companion EM on the Stock-Watson shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from dynamic_factor_models_amd import DfmContext
ctx = DfmContext()
dev = torch.device("cuda", ctx.device)
B, N, T, r = 1024, 200, 500, 8
panel, params = bench.synth_on_device(torch, dev, B, N, T, r, seed=1, missing=0.1)
for _ in range(8):
    ctx.ks_pass_batch(panel, *params, may_have_missing=True)
pp = [x.clone() for x in params]
for _ in range(8):
    ctx.em_step_batch(panel, *pp, may_have_missing=True)
torch.cuda.synchronize()
