import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import kalman_oracle as ko
from dynamic_factor_models_amd import DfmContext, api
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = np.load(os.path.join(ROOT, "tests", "golden", "sw_panel.npz"))
bp, inc = d["bpdata"], d["inclcode"]
ctx = DfmContext(); dev = torch.device("cuda", ctx.device)
zz, _ = api.standardize_data(bp[2:224][:, inc == 1])
zz = zz[:, (~np.isnan(zz)).sum(axis=0) >= 20]
print("missing fraction", np.isnan(zz).mean(), "rows with any missing", np.isnan(zz).any(axis=1).mean(), "max missing in a row", np.isnan(zz).sum(axis=1).max())
xb, balm = api.drop_missing_col(zz)
p0, F00 = ko.pca_init(xb, 4)
Lm = np.empty((zz.shape[1], 4)); Rm = np.empty(zz.shape[1])
Lm[balm] = p0["Lam"]; Rm[balm] = p0["R"]
for i in np.nonzero(~balm)[0]:
    ok = ~np.isnan(zz[:, i]); bb = np.linalg.lstsq(F00[ok], zz[ok, i], rcond=None)[0]; ee = zz[ok, i] - F00[ok] @ bb
    Lm[i] = bb; Rm[i] = ee @ ee / ok.sum()
start = dict(Lam=Lm, R=Rm, A=p0["A"], Q=p0["Q"], mu0=p0["mu0"], P0=p0["P0"])
Bs = 1024
rep = lambda a: torch.from_numpy(np.ascontiguousarray(np.broadcast_to(a, (Bs,) + a.shape))).to(dev)
zt_ = rep(zz); k6 = ("Lam", "R", "A", "Q", "mu0", "P0")
dd = {k: rep(start[k]) for k in k6}
ctx.em_batch(zt_, *[dd[k] for k in k6], max_iter=2, tol=0.0, may_have_missing=True)
ctx.profile_enable(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
ctx.em_batch(zt_, *[dd[k] for k in k6], max_iter=5, tol=0.0, may_have_missing=True)
torch.cuda.synchronize(); s = (time.perf_counter() - t0) / 5
print("SW EM iteration", round(1e3 * s, 3), "ms", {k: round(v[0] / max(v[1], 1), 4) for k, v in ctx.profile_read().items() if v[1]})
ctx.profile_enable(False)
d0 = {k: rep(start[k]) for k in k6}
for trial in range(3):
    dd = {k: v.clone() for k, v in d0.items()}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    path, its, fsw, Psw = ctx.em_batch(zt_, *[dd[k] for k in k6], max_iter=10, tol=0.0, may_have_missing=True)
    torch.cuda.synchronize(); print("trial", trial, "10 iterations from the start:", round(1e3 * (time.perf_counter() - t0), 3), "ms", flush=True)
