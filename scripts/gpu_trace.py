#!/usr/bin/env python
"""Run K passes of the bench workload (for rocprofv3 --kernel-trace timelines)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dynamic_factor_models_amd import DfmContext
dev = torch.device("cuda", 0)
B, N, T, r = 1024, 200, 500, 8
panel, params = bench.synth_on_device(torch, dev, B, N, T, r, seed=1)
f = torch.empty((B, T, r), dtype=torch.float64, device=dev)
P = torch.empty((B, T, r * (r + 1) // 2), dtype=torch.float64, device=dev)
ll = torch.empty((B,), dtype=torch.float64, device=dev)
c = DfmContext(0)
for _ in range(int(os.environ.get("K", "12"))):
    c.ks_pass_batch(panel, *params, may_have_missing=False, out=(f, P, ll))
torch.cuda.synchronize()
c.close()
