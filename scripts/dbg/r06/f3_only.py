"""The SURVEY 8 f3 lines and the parametric config-1 lines of bench.py alone (development A/B)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
import bench
from dynamic_factor_models_amd import DfmContext
ctx = DfmContext(0)
dev = torch.device("cuda", 0)
out = {}
out.update(bench.f3_lines(torch, ctx, dev, cpu_seconds=0.5))
out.update(bench.c1_em_lines(torch, ctx, dev, cpu_seconds=0.5))
for k, v in out.items():
    print(k, round(v["ms_per_step"], 4), "whole_step", round(v["whole_step"], 4), "matches_oracle", v.get("matches_oracle"), v.get("loglik_path_max_rel_err"))
ctx.close()
