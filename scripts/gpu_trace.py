#!/usr/bin/env python
"""Run K steps of the bench workload (for rocprofv3 --kernel-trace timelines and --pmc passes).
Env: K (steps, default 12), B / N / T / R (shape), MISSING, MODE = pass | em."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dynamic_factor_models_amd import DfmContext
dev = torch.device("cuda", 0)
E = lambda k, d: type(d)(os.environ.get(k, d))
B, N, T, r, miss, mode = E("B", 1024), E("N", 200), E("T", 500), E("R", 8), E("MISSING", 0.0), E("MODE", "pass")
c = DfmContext(0)
panel, params = c.synth_panels(20160415, 0, B, T, N, r, missing_prob=miss)
f = torch.empty((B, T, r), dtype=torch.float64, device=dev)
P = torch.empty((B, T, r * (r + 1) // 2), dtype=torch.float64, device=dev)
ll = torch.empty((B,), dtype=torch.float64, device=dev)
K = int(os.environ.get("K", "12"))
if mode == "em":
    start = list(c.pca_init_batch(panel, r, want_factors=False)[:6]) if miss == 0.0 else [p.clone() for p in params]
    c.em_batch(panel, *start, max_iter=K, tol=0.0, want_smooth=False, may_have_missing=miss > 0)
else:
    for _ in range(K):
        c.ks_pass_batch(panel, *params, may_have_missing=miss > 0, out=(f, P, ll))
torch.cuda.synchronize()
c.close()
