import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dynamic_factor_models_amd import DfmContext
from oracle import varp_oracle as vo
KEYS = ("Lam", "R", "Avar", "Q", "mu0", "P0")
N, T, r, p, miss = 40, 30, 4, 4, 0.0
x = vo.synth_varp(0, N, T, r, p, missing=miss)
q, _ = vo.varp_init(np.nan_to_num(x), r, p)
ctx = DfmContext(0)
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a[None])).to(dev)
f, P, ll = ctx.ks_pass_varp_batch(t(x), *[t(q[k]) for k in KEYS])
torch.cuda.synchronize()
o = vo.kfs_pass_varp(x, p=p, **{k: q[k] for k in KEYS})
tri = np.tril_indices(r)
Pw = o["P_smooth"][:, :r, :r][:, tri[0], tri[1]]
Pg = P[0].cpu().numpy()
print("ll", ll[0].item(), o["loglik"])
for tt in range(T):
    print(tt, "f err %.1e" % np.abs(f[0, tt].cpu().numpy() - o["f_smooth"][tt, :r]).max(), "P err %.1e" % np.abs(Pg[tt] - Pw[tt]).max(), Pg[tt][:4], Pw[tt][:4])
