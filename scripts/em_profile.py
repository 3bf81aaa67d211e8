#!/usr/bin/env python
"""Per-kernel times of one EM iteration at the config-2 shape (balanced): run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from dynamic_factor_models_amd import DfmContext
ctx = DfmContext(); dev = torch.device("cuda", 0)
panel, params = bench.synth_on_device(torch, dev, 1024, 200, 500, 8, seed=1)
pp = [x.clone() for x in params]
for _ in range(3): ctx.em_step_batch(panel, *pp, may_have_missing=False)
ctx.profile_enable(True)
for _ in range(10): ctx.em_step_batch(panel, *pp, may_have_missing=False)
print({k: round(v[0] / v[1], 4) for k, v in ctx.profile_read().items()})
