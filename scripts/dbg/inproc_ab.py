#!/usr/bin/env python
"""A/B of two DFM_SCAN_ABL settings of the one-launch pass INSIDE one process on the same buffers (processes of this pool
differ by +-3 % on identical code -- physical placement of the 819 MB panel -- so cross-process A/B needs many repeats).
Usage: python scripts/dbg/inproc_ab.py <A> <B> [<C> ...] [batch=N]   with A, B, ... = a DFM_SCAN_ABL value or "ENV=VAL,ENV=VAL";
the pseudo-variable LIB=<path of a libdfmhip build> loads ANOTHER library for that variant (e.g. the previous round's build
under gpurun_tmp/, or lib/libdfmhip_diag.so): both live in this process, each context keeps the one it was created with. """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dynamic_factor_models_amd import DfmContext
from dynamic_factor_models_amd import _lib as _L
DEFAULT_SO = _L.SO_PATH
VARS = [a for a in sys.argv[1:] if not a.startswith("batch=")]
Brep = ([int(a[6:]) for a in sys.argv[1:] if a.startswith("batch=")] or [1024])[0]
ctxs = []
KNOBS = ("DFM_SCAN_ABL", "DFM_PASS_NSW", "DFM_PASS_NCOV", "DFM_PASS_FUSED", "DFM_NUM_CU")
for v in VARS:
    for k in KNOBS:
        os.environ.pop(k, None)
    so = DEFAULT_SO
    if "=" in v:
        for kv in v.split(","):
            k, x = kv.split("=")
            if k == "LIB": so = x if os.path.isabs(x) else os.path.join(ROOT, x)
            else: os.environ[k] = x
    else:
        os.environ["DFM_SCAN_ABL"] = v
    if so != _L.SO_PATH:
        _L.SO_PATH = so; _L._lib = None
    ctxs.append(DfmContext(0))
panel, par = ctxs[0].synth_panels(20160415, 0, Brep, 500, 200, 8)
dev = panel.device
f = torch.empty((Brep, 500, 8), dtype=torch.float64, device=dev); P = torch.empty((Brep, 500, 36), dtype=torch.float64, device=dev)
ll = torch.empty((Brep,), dtype=torch.float64, device=dev)
K = 200 if Brep <= 1024 else 30
def run(c, k):
    for _ in range(k):
        c.ks_pass_batch(panel, *par, may_have_missing=False, out=(f, P, ll))
for c in ctxs:
    run(c, K); torch.cuda.synchronize()
res = {i: [] for i in range(len(VARS))}
for rnd in range(8):
    for i, c in enumerate(ctxs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(c, K); torch.cuda.synchronize()
        res[i].append((time.perf_counter() - t0) / K * 1e3)
for i, v in enumerate(VARS):
    r = sorted(res[i])
    print(f"{v:>22}: median {r[len(r)//2]:.5f} ms  min {r[0]:.5f}  max {r[-1]:.5f}  -> {Brep / r[len(r)//2] * 1e3 / 1e6:.3f} M passes/s   all {[round(x, 4) for x in res[i]]}")
