#!/usr/bin/env python
"""Phase timeline of the fused pass from the s_memrealtime stamps (DFM_SCAN_ABL=256, DFM_PF_PROF_FILE): per replicate
[0] iteration start, [1] wave 0 segment done, [5] last stream wave done, [6] cov start, [7] Gram done, [8] recursion done,
[9] fill issued, [2] past barrier A, [3] scan done, [4] past barrier B.  Ticks are 10 ns."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DFM_PASS_FUSED"] = "1"
os.environ["DFM_SCAN_ABL"] = "256"
os.environ["DFM_PF_PROF_FILE"] = "/tmp/pf_prof.txt"
from dynamic_factor_models_amd import DfmContext
B = int(os.environ.get("B", 1024))
c = DfmContext(0)
panel, par = c.synth_panels(1, 0, B, 500, 200, 8)
for _ in range(3):
    c.ks_pass_batch(panel, *par, may_have_missing=False)
torch.cuda.synchronize()
d = np.loadtxt("/tmp/pf_prof.txt")
b = d[:, 0].astype(int); s = d[:, 1:] * 0.01   # us
t0 = s[:, 0].min()
ncu = min(B, 256)
print("replicates", B, "span us", (s[:, 4].max() - t0))
for name, a, z in [("stream wave0", 0, 1), ("stream last wave", 0, 5), ("gram", 6, 7), ("cov recursion", 7, 8), ("fill", 8, 9),
                   ("wave0 wait at A", 1, 2), ("scan", 2, 3), ("wait at B", 3, 4), ("iteration", 0, 4)]:
    v = s[:, z] - s[:, a]
    print(f"{name:18s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")
print("scan phases (us; [cycles]):")
names = ["fwd transient", "fwd phase 1", "fwd carry scan", "fwd phase 3", "terminal + bwd phase 1", "bwd carry scan", "bwd phase 3", "bwd transient"]
for k, nm in enumerate(names):
    v = s[:, 11 + k] - s[:, 10 + k]; cy = (d[:, 1 + 21 + k] - d[:, 1 + 20 + k])
    print(f"  {nm:24s} {v.mean():6.2f} us  [{cy.mean():8.0f} cycles]  -> {cy.mean() / max(v.mean(), 1e-9):6.0f} MHz")
print("  E (transient steps) unknown here; iteration clock:", ((d[:, 1 + 28] - d[:, 1 + 20]) / np.maximum(s[:, 18] - s[:, 10], 1e-9)).mean(), "MHz")
for rnd in range((B + ncu - 1) // ncu):
    sel = (b // ncu) == rnd
    print(f"round {rnd}: start {s[sel, 0].mean() - t0:7.1f}  stream end {s[sel, 5].mean() - t0:7.1f}  cov end {s[sel, 9].mean() - t0:7.1f}  scan end {s[sel, 3].mean() - t0:7.1f}")
c.close()
