"""VAR(4) companion EM (k = 16), balanced panel only, 3 iterations: for in-kernel phase prints (DFM_WAVE_PROF builds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from dynamic_factor_models_amd import DfmContext
c = DfmContext(0)
dev = torch.device("cuda", 0)
B, N, T, r, p = 1024, 139, 222, 4, 4
miss = float(os.environ.get("MISSING", "0"))
panel, par = c.synth_panels(7, 0, B, T, N, r, missing_prob=miss)
Lam, R, A, Q, mu0, P0 = [x.clone() for x in par]
k = r * p
Avar = torch.zeros((B, r, k), dtype=torch.float64, device=dev); Avar[:, :, :r] = A
mu0k = torch.zeros((B, k), dtype=torch.float64, device=dev)
P0k = torch.eye(k, dtype=torch.float64, device=dev).expand(B, k, k).contiguous()
c.em_varp_batch(panel, Lam, R, Avar, Q, mu0k, P0k, max_iter=3, tol=0.0, want_smooth=False, may_have_missing=miss > 0)
torch.cuda.synchronize()
