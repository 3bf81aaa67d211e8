"""One EM iteration of the VAR(p) model on recursion_mbf16_kernel's shapes (r <= 3: blocks narrower than 4), 1024 replicates of the
Stock-Watson window's shape: ms per iteration, likelihood path against the oracle."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from dynamic_factor_models_amd import DfmContext
from oracle import varp_oracle as vo
ctx = DfmContext(0); dev = torch.device("cuda", 0)
Bv, Nv, Tv, miss, nit = 1024, 139, 222, 0.1, 5
KV = ("Lam", "R", "Avar", "Q", "mu0", "P0")
for rv, pv in ((3, 4), (2, 6), (3, 5)):
    tile = lambda a: torch.from_numpy(np.ascontiguousarray(np.tile(a, (Bv // 16,) + (1,) * (a.ndim - 1)))).to(dev)
    xs, qs = [], []
    for b in range(16):
        x = vo.synth_varp(b, Nv, Tv, rv, pv, missing=miss)
        xs.append(x); qs.append(vo.varp_init(np.nan_to_num(x), rv, pv)[0])
    xv = tile(np.stack(xs)); d0 = {k: tile(np.stack([q[k] for q in qs])) for k in KV}
    best = None
    for rep in range(4):
        dd = {k: v.clone() for k, v in d0.items()}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        path, its, _, _ = ctx.em_varp_batch(xv, *[dd[k] for k in KV], max_iter=nit, tol=0.0, may_have_missing=True)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / nit
        if rep > 0: best = el if best is None else min(best, el)
    _, opath, _ = vo.em_varp(xs[0], dict(qs[0]), pv, max_iter=nit, tol=0.0)
    ok = bool(np.allclose(path[0].cpu().numpy(), opath, rtol=1e-8))
    print(f"r={rv} p={pv}: {1e3 * best:.4f} ms per EM iteration, matches_oracle {ok}")
