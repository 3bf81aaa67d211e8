"""Do two passes over panels with missing cells, enqueued on two handles with their own streams, overlap on the device?
collapse_miss_kernel is bound by the memory pipe, recursion_chunk_kernel by fp64 VALU issue: run one after the other (one stream)
each leaves the other's resource idle.  Prints ms per pass of B replicates: one handle alone, two handles alternating."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from dynamic_factor_models_amd import DfmContext  # noqa: E402

B = int(os.environ.get("B", 1024)); N, T, r = 200, 500, 8
K = int(os.environ.get("K", 20))
MISSING = float(os.environ.get("MISSING", 0.1))
NH = int(os.environ.get("NH", 2))
EM = int(os.environ.get("EM", 0))

ctxs = [DfmContext(0, use_torch_stream=False) for _ in range(NH)]
gen = DfmContext(0)
data = []
for k in range(NH):
    panel, par = gen.synth_panels(1234 + k, 0, B, T, N, r, missing_prob=MISSING)
    out = (torch.empty((B, T, r), dtype=torch.float64, device="cuda"), torch.empty((B, T, r * (r + 1) // 2), dtype=torch.float64, device="cuda"),
           torch.empty((B,), dtype=torch.float64, device="cuda"))
    data.append((panel, par, out))
torch.cuda.synchronize()


def one(c, d):
    if EM:
        return c.em_step_batch(d[0], *d[1], may_have_missing=True)
    return c.ks_pass_batch(d[0], *d[1], want_P=True, may_have_missing=True, out=d[2])


def run(handles, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        for h in handles:
            one(ctxs[h], data[h])
    for h in handles:
        ctxs[h].synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / (k * len(handles))


for h in range(NH):
    run([h], 3)
run(list(range(NH)), 3)
a = [run([0], K) for _ in range(3)]
b = [run(list(range(NH)), K) for _ in range(3)]
print(f"B={B} missing={MISSING} em={EM}: one handle {min(a):.4f} ms per pass; {NH} handles alternating {min(b):.4f} ms per pass "
      f"(ratio {min(b) / min(a):.3f})")
