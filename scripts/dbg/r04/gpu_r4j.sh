#!/bin/bash
# PCA start: phase-stop timing (diagnostics build) at C2
TAG=${1:-r4j}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for x in 0 1 2 3 4 5; do
  DFM_LIB=diag DFM_PCA_STOP=$x timeout 300 python bench.py --mode pca --steps 5 --warmup 1 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/p$x.json 2> $OUT/p$x.err
  python - $OUT/p$x.json $x <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("pca stop", sys.argv[2], "ms %.4f"%d["ms_per_step"], r["kernels_ms"])
except Exception as e: print("stop", sys.argv[2], "unreadable", e)
PY
done
tail -3 $OUT/p0.err
