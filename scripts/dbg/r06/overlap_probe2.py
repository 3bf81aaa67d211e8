"""overlap_probe.py with the second handle's stream started half a pass late (a spin kernel in front of its first pass): does a
collapse beside the other handle's recursion run faster than collapse beside collapse?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from dynamic_factor_models_amd import DfmContext  # noqa: E402

B = int(os.environ.get("B", 1024)); N, T, r = 200, 500, 8
K = int(os.environ.get("K", 20))
MISSING = float(os.environ.get("MISSING", 0.1))
NH = 2
ctxs = [DfmContext(0) for _ in range(NH)]
streams = [torch.cuda.Stream() for _ in range(NH)]
gen = DfmContext(0)
data = []
for k in range(NH):
    panel, par = gen.synth_panels(1234 + k, 0, B, T, N, r, missing_prob=MISSING)
    out = (torch.empty((B, T, r), dtype=torch.float64, device="cuda"), torch.empty((B, T, r * (r + 1) // 2), dtype=torch.float64, device="cuda"),
           torch.empty((B,), dtype=torch.float64, device="cuda"))
    data.append((panel, par, out))
torch.cuda.synchronize()


def one(h):
    d = data[h]
    with torch.cuda.stream(streams[h]):
        ctxs[h].ks_pass_batch(d[0], *d[1], want_P=True, may_have_missing=True, out=d[2])


def run(handles, k, delay_cycles=0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if delay_cycles and len(handles) > 1:
        with torch.cuda.stream(streams[handles[1]]):
            torch.cuda._sleep(delay_cycles)
    for _ in range(k):
        for h in handles:
            one(h)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / (k * len(handles))


for h in range(NH):
    run([h], 3)
run([0, 1], 3)
a = min(run([0], K) for _ in range(3))
print(f"B={B}: one handle {a:.4f} ms per pass")
for us in (0, 100, 200, 300, 400, 600):
    cyc = int(us * 1e-6 * 100e6)          # torch.cuda._sleep counts a 100 MHz clock on ROCm builds (wall_clock64)
    b = min(run([0, 1], K, cyc) for _ in range(3))
    print(f"   two handles, second delayed by ~{us} us (x B/1024): {b:.4f} ms per pass (ratio {b / a:.3f})")
