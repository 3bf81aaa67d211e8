#!/usr/bin/env python
"""Per-kernel milliseconds of one EM iteration (library event pairs): python scripts/dbg/em_prof.py B T N r missing"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dynamic_factor_models_amd import DfmContext
B, T, N, r = (int(x) for x in sys.argv[1:5]); miss = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
c = DfmContext(0)
panel, par = c.synth_panels(7, 0, B, T, N, r, missing_prob=miss)
par = [p.clone() for p in par]
for _ in range(2):
    c.em_step_batch(panel, *par, may_have_missing=miss > 0)
torch.cuda.synchronize()
c.profile_enable(True)
K = 3
for _ in range(K):
    c.em_step_batch(panel, *par, may_have_missing=miss > 0)
torch.cuda.synchronize()
tot = 0.0
for name, (ms, n) in c.profile_read().items():
    print(f"{name:34s} {ms / K:9.3f} ms per iteration  ({n // K} launches)")
    tot += ms / K
print(f"{'sum':34s} {tot:9.3f} ms")
