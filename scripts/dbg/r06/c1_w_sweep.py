"""c1_em (bench.py: Stock-Watson window, 1024 parametric-bootstrap replicates x 10 EM iterations, factor VAR(1)): time and chunk
fallbacks against the warm-up length DFM_CHUNK_W (development A/B; the switch is read at dfm_create)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dynamic_factor_models_amd import DfmContext, api
d = np.load(os.path.join(ROOT, "tests", "golden", "sw_panel.npz"))
r, init, last, Bc, nit = 4, 3, 224, 1024, 10
dev = torch.device("cuda", 0)
ctx0 = DfmContext(0)
m = api.DFMModel(d["bpdata"], d["inclcode"], 20, 40, init, last, 0, r, 1e-8, 4, 4)
api.estimate(m, api.Parametric(), max_em_iter=nit, tol_em=0.0, factor_lags=1, ctx=ctx0)
q = m.em_params
z, _ = api.standardize_data(d["bpdata"][init - 1:last][:, d["inclcode"] == 1])
z = z[:, (~np.isnan(z)).sum(axis=0) >= 20]
T, N = z.shape
rng = np.random.default_rng(20160416)
LQ, LS, sq = api._psd_sqrt(q["Q"]), api._psd_sqrt(q["P0"]), np.sqrt(q["R"])
st = q["mu0"][None] + rng.standard_normal((Bc, r)) @ LS.T
panels = np.empty((Bc, T, N))
for t in range(T):
    st = st @ q["A"].T + rng.standard_normal((Bc, r)) @ LQ.T
    panels[:, t] = st @ q["Lam"].T + sq * rng.standard_normal((Bc, N))
panels[:, np.isnan(z)] = np.nan
keys = ("Lam", "R", "A", "Q", "mu0", "P0")
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x = up(panels)
d0 = {k: up(np.repeat(q[k][None], Bc, axis=0)) for k in keys}
ctx0.close()
for W in sys.argv[1:]:
    os.environ["DFM_CHUNK_W"] = W
    ctx = DfmContext(0)
    best = None
    for rep in range(4):
        dd = {k: v.clone() for k, v in d0.items()}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.em_batch(x, *[dd[k] for k in keys], max_iter=nit, tol=0.0, may_have_missing=True)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        if rep: best = el if best is None else min(best, el)
    print("W", W, "ms per 10 iterations %.3f" % (1e3 * best), "fallbacks (last iteration)", ctx.chunk_fallbacks())
    ctx.close()
