import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import kalman_oracle as ko
from dynamic_factor_models_amd import DfmContext
B, N, T, r, missing = [int(x) for x in sys.argv[1:5]] + [float(sys.argv[5])]
KEYS = ("Lam", "R", "A", "Q", "mu0", "P0")
panels, starts = [], []
for b in range(B):
    x, _ = ko.synth_replicate(100 + b, N, T, r, missing=missing)
    p0, _ = ko.pca_init(np.nan_to_num(x), r)
    panels.append(x); starts.append(p0)
panel = np.stack(panels); st = {k: np.stack([s[k] for s in starts]) for k in starts[0]}
c = DfmContext()
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
f, P, ll = c.ks_pass_batch(dev(panel), *[dev(st[k]) for k in KEYS], may_have_missing=True)
torch.cuda.synchronize()
print("GPU pass loglik", ll.cpu().numpy())
for b in range(B):
    out = ko.ks_pass(panel[b], *[st[k][b] for k in KEYS]) if hasattr(ko, "ks_pass") else None
    if out is not None: print("oracle", out["loglik"] if isinstance(out, dict) else out[-1])
print("min eig Q", [np.linalg.eigvalsh(st["Q"][b]).min() for b in range(B)], "min R", st["R"].min(), "min eig P0", [np.linalg.eigvalsh(st["P0"][b]).min() for b in range(B)])
