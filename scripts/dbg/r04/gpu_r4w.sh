#!/bin/bash
# config 4 balanced: does the streaming collapse overlap the covariance kernel at all?  In-order run (DFM_NO_SIDE=1, diagnostics
# build) against the default three-stream schedule, kernel durations from the event pairs around each launch
TAG=${1:-r4w}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
for k in 1 2; do
for ns in 0 1; do
  DFM_LIB=diag DFM_NO_SIDE=$ns timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --mode pass --steps 10 --warmup 3 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/c4_ns$ns.json 2> $OUT/c4_ns$ns.err
  python - $OUT/c4_ns$ns.json $ns <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("c4 pass NO_SIDE", sys.argv[2], "ms %.4f whole %.3f"%(d["ms_per_step"], r["whole_step"]["frac"]), r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
done; done 2>&1 | tee $OUT/c4_lines.txt
