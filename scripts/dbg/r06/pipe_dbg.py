import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from dynamic_factor_models_amd import DfmContext
ctx = DfmContext()
B, N, T, r = 4200, 40, 64, 8
panel, par = ctx.synth_panels(11, 0, B, T, N, r, missing_prob=0.1)
f, P, ll = ctx.ks_pass_batch(panel, *par, may_have_missing=True)
torch.cuda.synchronize()
print("fallbacks", ctx.chunk_fallbacks())
for Bs in (2000, 1000, 104):
    for b0 in range(0, B, Bs):
        sl = slice(b0, min(B, b0 + Bs))
        f1, P1, ll1 = ctx.ks_pass_batch(panel[sl].contiguous(), *[p[sl].contiguous() for p in par], may_have_missing=True)
        torch.cuda.synchronize()
        d = (ll[sl] - ll1).abs()
        bad = torch.nonzero(d > 0).flatten()
        if len(bad):
            print(Bs, b0, "ll mismatches", len(bad), "first", (bad[:5] + b0).tolist(), "max", d.max().item(), "f max", (f[sl] - f1).abs().max().item(), ctx.chunk_fallbacks())
f2, P2, ll2 = ctx.ks_pass_batch(panel, *par, may_have_missing=True)
torch.cuda.synchronize()
print("run-to-run equal:", torch.equal(ll, ll2), torch.equal(f, f2))
