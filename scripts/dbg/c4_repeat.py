"""Repeat the config-4 pass and report where repeated calls / the -2 x panel call differ (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from dynamic_factor_models_amd import DfmContext
from test_gpu_ks_pass import _batch
B, N, T, r = int(os.environ.get("B", 256)), 1000, 2000, 20
ctx = DfmContext(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(4)
panel = torch.randn((B, T, N), dtype=torch.float64, device=dev, generator=g)
_, st = _batch(1, N, T, r)
rep = lambda a: torch.from_numpy(a).to(dev).expand(B, *a.shape[1:]).contiguous()
args = [rep(st[k]) for k in ("Lam", "R", "A", "Q", "mu0", "P0")]
outs = []
for i in range(4):
    f, P, ll = ctx.ks_pass_batch(panel, *args, may_have_missing=False)
    torch.cuda.synchronize()
    outs.append((f.clone(), P.clone(), ll.clone()))
for i in range(1, 4):
    d = (outs[i][0] != outs[0][0])
    print("repeat", i, "f differs at", int(d.sum()), "P", int((outs[i][1] != outs[0][1]).sum()), "ll", int((outs[i][2] != outs[0][2]).sum()))
    if d.any():
        idx = d.nonzero()
        print("  replicates", torch.unique(idx[:, 0]).tolist()[:20], "t range", int(idx[:, 1].min()), int(idx[:, 1].max()), "max abs", float((outs[i][0] - outs[0][0]).abs().max()))
os.environ["DFM_WIDE_OLD"] = "1"
ctx2 = DfmContext(0)
fo, Po, llo = ctx2.ks_pass_batch(panel, *args, may_have_missing=False)
torch.cuda.synchronize()
for i in range(4):
    e = (outs[i][0] - fo).abs().amax(dim=(1, 2))
    print("call", i, "vs old kernel: max abs", float(e.max()), "replicates over 1e-9:", (e > 1e-9).nonzero().flatten().tolist())
    if (e > 1e-9).any():
        bb = int(e.argmax()); et = (outs[i][0][bb] - fo[bb]).abs().amax(dim=1)
        tt = (et > 1e-9).nonzero().flatten(); print("   worst replicate", bb, "t", int(tt.min()), int(tt.max()), "peak t", int(et.argmax()))
panel.mul_(-2.0)
f2, P2, _ = ctx.ks_pass_batch(panel, *args, may_have_missing=False)
torch.cuda.synchronize()
d = (f2 != -2.0 * outs[0][0])
print("-2x: differs at", int(d.sum()), "max abs", float((f2 + 2.0 * outs[0][0]).abs().max()))
if d.any():
    idx = d.nonzero()
    print("  replicates", torch.unique(idx[:, 0]).tolist()[:40], "t range", int(idx[:, 1].min()), int(idx[:, 1].max()))
    e = (f2 + 2.0 * outs[0][0]).abs()
    print("  per-replicate max", e.amax(dim=(1, 2))[:16].tolist())
