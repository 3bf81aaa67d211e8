#!/usr/bin/env python
"""Phase timeline of the one-launch pass (pass_fused.hip) from its s_memrealtime stamps (DFM_SCAN_ABL=256,
DFM_PF_PROF_FILE).  Per replicate (10 ns ticks):
  stream: [0] iteration start (wave 0), [1] b_t buffer free, [2] wave 0's segment done, [3] last stream wave done
  cov:    [6] start, [7] Gram done, [8] recursion done (tables published)
  mover:  [4] tables of the replicate published by its covariance wave, [5] table set in LDS (tab_ready)
  scan:   [10] start waiting, [11] tables ready, [12] b_t ready, [14..21] scan phases, [22] done (buffer released)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DFM_LIB", "diag")   # the stamps exist only in the diagnostics build: python -m dynamic_factor_models_amd.build --diag
os.environ["DFM_PASS_FUSED"] = "1"
os.environ["DFM_SCAN_ABL"] = str(256 | int(os.environ.get("PF_ABL", "0")))   # PF_ABL=512: the round-2 scan
os.environ["DFM_PF_PROF_FILE"] = "/tmp/pf_prof.txt"
from dynamic_factor_models_amd import DfmContext
B = int(os.environ.get("B", 1024))
c = DfmContext(0)
panel, par = c.synth_panels(1, 0, B, 500, 200, 8)
# PF_POLLUTE=1: a pass of another shape (another 105 KB instantiation of the kernel) before every profiled pass -- the
# instruction cache starts cold, as it does at EVERY launch on the boxes with host kernel 6.18.50 (scripts/microbench/icache.hip)
pollute = os.environ.get("PF_POLLUTE", "0") == "1"
if pollute:
    panel2, par2 = c.synth_panels(2, 0, B, 500, 208, 8)
for _ in range(3):
    if pollute:
        c.ks_pass_batch(panel2, *par2, may_have_missing=False)
    c.ks_pass_batch(panel, *par, may_have_missing=False)
torch.cuda.synchronize()
d = np.loadtxt("/tmp/pf_prof.txt")
b = d[:, 0].astype(int); s = d[:, 1:] * 0.01   # us
t0 = s[:, 0].min()
ncu = min(B, 256)
print("replicates", B, "span us", (s[:, 22].max() - t0))
for name, a, z in [("stream: wait buffer", 0, 1), ("stream: wave 0 segment", 1, 2), ("stream: last wave - w0", 2, 3), ("cov: gram", 6, 7),
                   ("cov: recursion", 7, 8), ("mover: load + wait set", 4, 5), ("scan: wait tables", 10, 11), ("scan: wait b_t", 11, 12),
                   ("scan: compute", 12, 22), ("replicate: start -> scan end", 0, 22)]:
    v = s[:, z] - s[:, a]
    print(f"{name:30s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")
# the covariance wave's Gram step: one batch of 54 global loads, then 4 log() and 50 MFMAs -- per round of the workgroup's
# covariance waves (round 0 = the chain everything behind it waits for; code cold at launch on some boxes)
for rnd in range(2):
    sel = (b // (2 * ncu)) == rnd
    if sel.any():
        print(f"cov round {rnd}: gram loads {(s[sel, 36] - s[sel, 6]).mean():6.2f}  gram compute {(s[sel, 7] - s[sel, 36]).mean():6.2f}  recursion {(s[sel, 8] - s[sel, 7]).mean():6.2f}")
names = ["sync + fwd transient", "fwd phase 1", "fwd carry scan", "fwd phase 3", "terminal + bwd phase 1", "bwd carry scan", "bwd phase 3",
         "bwd transient", "loglik + release"]
prev = 12
print("scan phases (us):")
for k, nm in enumerate(names):
    cur = 14 + k
    v = s[:, cur] - s[:, prev]
    print(f"  {nm:24s} {v.mean():6.2f}")
    prev = cur
nsw = int(os.environ.get("DFM_PASS_NSW", 5))
print("per stream wave: segment duration (us) / end relative to wave 0's end")
for w in range(nsw):
    dur = s[:, 24 + w] - s[:, 32 + w]; rel = s[:, 24 + w] - s[:, 24]
    print(f"  wave {w}: duration mean {dur.mean():6.2f} p10 {np.percentile(dur, 10):6.2f} p90 {np.percentile(dur, 90):6.2f}   end - end(w0) mean {rel.mean():6.2f}   start - start(w0) {(s[:, 32 + w] - s[:, 32]).mean():6.2f}")
# per workgroup (replicates b % ncu): when its last scan ended, when its first covariance chain ended, its stream's total
wg = b % ncu
fin = np.array([s[wg == g, 22].max() - t0 for g in range(ncu)])
cov0 = np.array([s[(wg == g) & (b // ncu == 0), 8].max() - t0 for g in range(ncu)])
str_end = np.array([s[wg == g, 3].max() - t0 for g in range(ncu)])
pc = lambda v: " ".join(f"{np.percentile(v, q):7.1f}" for q in (0, 10, 50, 90, 99, 100))
print("per workgroup (us; min p10 p50 p90 p99 max):")
print("  last scan end      ", pc(fin))
print("  last stream end    ", pc(str_end))
print("  first cov chain end", pc(cov0))
slow = np.argsort(fin)[-8:]
print("  slowest workgroups :", [(int(g), round(float(fin[g]), 1), round(float(cov0[g]), 1)) for g in slow], "(id, last scan end, first cov end)")
print("  XCD (id % 8) of the 32 slowest:", sorted(int(g) % 8 for g in np.argsort(fin)[-32:]))
# where the hardware put the 11 waves of a workgroup: SIMD of each role (stream x4, cov x2, mover, scan x4)
hw = np.rint(d[:ncu, 41:52]).astype(np.int64)
simd = (hw >> 4) & 3
from collections import Counter
pat = Counter("".join(str(int(v)) for v in row) for row in simd)
print("wave -> SIMD patterns (waves 0..10 = 4 stream, 2 cov, mover, 4 scan):", pat.most_common(6))
cu = (hw[:, 0] >> 8) & 15; se = (hw[:, 0] >> 13) & 7
print("distinct (SE, CU) of the workgroups:", len(set(zip(se.tolist(), cu.tolist()))), " same-CU for all waves of a WG:",
      bool(np.all(((hw >> 8) & 15) == cu[:, None])))
for rnd in range(min(8, (B + ncu - 1) // ncu)):
    sel = (b // ncu) == rnd
    print(f"round {rnd}: stream start {s[sel, 0].mean() - t0:7.1f}  buffer free {s[sel, 1].mean() - t0:7.1f}  stream end {s[sel, 3].mean() - t0:7.1f}"
          f"  cov start {s[sel, 6].mean() - t0:7.1f}  cov end {s[sel, 8].mean() - t0:7.1f}  scan start {s[sel, 12].mean() - t0:7.1f}  scan end {s[sel, 22].mean() - t0:7.1f}")
c.close()
