#!/usr/bin/env python
"""The one-launch pass with COLD kernel code at every launch, in one process, for several library builds: passes of two shapes
(N = 200 and N = 208: two different instantiations of pass_fused_kernel, together more code than the 64-KB instruction cache of a CU
pair) alternate, so every launch finds its code evicted -- the regime of the pool's slow boxes (host kernel 6.18.50: code cold at
every launch, profiles/r03/slow_boxes/README.md), reproduced on any box.  Prints ms per pass of shape A alone (warm) and inside the
alternation (cold), per build.  Usage: python scripts/dbg/cold_ab.py LIB=<path> [LIB=<path> ...]  ("0" = the default build)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dynamic_factor_models_amd import DfmContext
from dynamic_factor_models_amd import _lib as _L
DEFAULT_SO = _L.SO_PATH
ctxs = []
for v in sys.argv[1:]:
    so = DEFAULT_SO
    if v.startswith("LIB="):
        x = v[4:]
        so = x if os.path.isabs(x) else os.path.join(ROOT, x)
    if so != _L.SO_PATH:
        _L.SO_PATH = so; _L._lib = None
    ctxs.append((v, DfmContext(0)))
B, T, r = 1024, 500, 8
c0 = ctxs[0][1]
pa, para = c0.synth_panels(20160415, 0, B, T, 200, r)
pb, parb = c0.synth_panels(7, 0, B, T, 208, r)
dev = pa.device
f = torch.empty((B, T, r), dtype=torch.float64, device=dev); P = torch.empty((B, T, 36), dtype=torch.float64, device=dev)
ll = torch.empty((B,), dtype=torch.float64, device=dev)
K = 100
def warm(c, k):
    for _ in range(k):
        c.ks_pass_batch(pa, *para, may_have_missing=False, out=(f, P, ll))
def alt(c, k):
    for _ in range(k):
        c.ks_pass_batch(pa, *para, may_have_missing=False, out=(f, P, ll))
        c.ks_pass_batch(pb, *parb, may_have_missing=False, out=(f, P, ll))
def only_b(c, k):
    for _ in range(k):
        c.ks_pass_batch(pb, *parb, may_have_missing=False, out=(f, P, ll))
res = {v: dict(warm=[], alt=[], b=[]) for v, _ in ctxs}
for v, c in ctxs:
    warm(c, 50); alt(c, 20); torch.cuda.synchronize()
for rnd in range(6):
    for v, c in ctxs:
        for name, fn in (("warm", warm), ("b", only_b), ("alt", alt)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fn(c, K); torch.cuda.synchronize()
            res[v][name].append((time.perf_counter() - t0) / K * 1e3)
med = lambda a: sorted(a)[len(a) // 2]
for v, _ in ctxs:
    w, bb, al = med(res[v]["warm"]), med(res[v]["b"]), med(res[v]["alt"])
    print(f"{v:>34}: N=200 warm {w:.4f} ms | N=208 warm {bb:.4f} ms | alternating pair {al:.4f} ms -> cold penalty per pass {(al - w - bb) / 2 * 1e3:+.1f} us")
