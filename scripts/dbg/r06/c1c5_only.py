"""bench.py's config-1 (ALS) and config-5 (bootstrap bands) lines alone (development A/B)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
import bench
from dynamic_factor_models_amd import DfmContext
ctx = DfmContext(0)
out = bench.config1_config5_lines(torch, ctx, torch.device("cuda", 0), cpu_seconds=0.5)
for k, v in out.items():
    print(k, round(v["ms_per_step"], 4), {a: v.get(a) for a in ("matches_oracle_ssr", "matches_oracle", "error") if a in v})
ctx.close()
