"""Debug: balanced pass on PCA start parameters for short panels with many factors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import kalman_oracle as ko, c_oracle as co
from dynamic_factor_models_amd import DfmContext
ctx = DfmContext()
dev = torch.device("cuda", ctx.device)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
keys = ("Lam", "R", "A", "Q", "mu0", "P0")
for (B, N, T, r) in [(1, 52, 18, 11), (1, 52, 25, 11), (1, 52, 40, 11), (1, 52, 18, 8), (1, 52, 18, 4), (1, 52, 14, 12), (2, 30, 13, 9)]:
    reps = [ko.synth_replicate(b, N, T, r, seed=ko.SEED0 + 17 * N + T) for b in range(B)]
    panel = np.stack([x for x, _ in reps])
    starts = [ko.pca_init(panel[b], r)[0] for b in range(B)]
    st = {k: np.stack([s[k] for s in starts]) for k in keys}
    f, P, ll = ctx.ks_pass_batch(t(panel), *[t(st[k]) for k in keys], may_have_missing=False)
    torch.cuda.synchronize()
    fo, Po, llo = co.ks_pass_batch(panel, *[st[k] for k in keys])
    print((B, N, T, r), "ll", ll.cpu().numpy(), llo, "f err", np.abs(f.cpu().numpy() - fo).max(), "P err", np.abs(P.cpu().numpy() - Po).max(),
          "minR", st["R"].min(), "eigA", np.abs(np.linalg.eigvals(st["A"][0])).max(), flush=True)
