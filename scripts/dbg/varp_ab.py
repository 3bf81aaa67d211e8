"""VAR(4) companion EM iteration (k = 16) and AR-idiosyncratic pass: timing for A/B between two source trees."""
import os, sys, time
ROOT = os.environ.get("DFM_TREE", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dynamic_factor_models_amd import DfmContext
import dynamic_factor_models_amd as pkg
print("tree", os.path.dirname(pkg.__file__))
c = DfmContext(0)
dev = torch.device("cuda", 0)
B, N, T, r, p = 1024, 139, 222, 4, 4
for miss in (0.0, 0.1):
    panel, par = c.synth_panels(7, 0, B, T, N, r, missing_prob=miss)
    Lam, R, A, Q, mu0, P0 = [x.clone() for x in par]
    k = r * p
    Avar = torch.zeros((B, r, k), dtype=torch.float64, device=dev); Avar[:, :, :r] = A
    mu0k = torch.zeros((B, k), dtype=torch.float64, device=dev)
    P0k = torch.eye(k, dtype=torch.float64, device=dev).expand(B, k, k).contiguous()
    fn = getattr(c, "em_varp_batch")
    def run(it):
        return fn(panel, Lam.clone(), R.clone(), Avar.clone(), Q.clone(), mu0k.clone(), P0k.clone(), max_iter=it, tol=0.0, want_smooth=False, may_have_missing=miss > 0)
    run(1); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(6); torch.cuda.synchronize(); t1 = time.perf_counter()
    run(1); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("missing", miss, "ms per EM iteration %.3f" % (1e3 * ((t1 - t0) - (t2 - t1)) / 5))
