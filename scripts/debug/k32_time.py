import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import varp_oracle as vo
from dynamic_factor_models_amd import DfmContext
ctx = DfmContext(); dev = torch.device("cuda", ctx.device)
KEYS = ("Lam", "R", "Avar", "Q", "mu0", "P0")
for (Bv, Nv, Tv, rv, pv, miss) in [(256, 139, 222, 8, 4, 0.1), (1024, 139, 222, 8, 4, 0.1), (256, 139, 222, 5, 4, 0.1)]:
    xs, qs = [], []
    for b in range(4):
        x = vo.synth_varp(b, Nv, Tv, rv, pv, missing=miss)
        xs.append(x); qs.append(vo.varp_init(np.nan_to_num(x), rv, pv)[0])
    reps = Bv // 4
    tile = lambda a: torch.from_numpy(np.ascontiguousarray(np.tile(a, (reps,) + (1,) * (a.ndim - 1)))).to(dev)
    xv = tile(np.stack(xs))
    dd = {k: tile(np.stack([q[k] for q in qs])) for k in KEYS}
    ctx.em_varp_batch(xv, *[dd[k] for k in KEYS], max_iter=1, tol=0.0, may_have_missing=True)
    ctx.profile_enable(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.em_varp_batch(xv, *[dd[k] for k in KEYS], max_iter=3, tol=0.0, may_have_missing=True)
    torch.cuda.synchronize(); s = (time.perf_counter() - t0) / 3
    print((Bv, Nv, Tv, rv, pv, miss), "varp EM iteration", round(1e3 * s, 3), "ms", {k: round(v[0] / max(v[1], 1), 4) for k, v in ctx.profile_read().items() if v[1]}, flush=True)
    ctx.profile_enable(False)
