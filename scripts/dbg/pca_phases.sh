#!/bin/bash
# DFM_PCA_STOP=k phase-stop timing of the PCA start; ARGS = extra bench.py arguments (shape)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pcaph; mkdir -p $OUT
for x in 0 1 2 3 4 5; do
  DFM_PCA_STOP=$x timeout 300 python bench.py --mode pca $ARGS --steps ${STEPS:-5} --warmup 1 --repeats 3 > $OUT/p$x.json 2> $OUT/p$x.err
  python - $OUT/p$x.json $x <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print("pca stop", sys.argv[2], "ms %.4f"%d["ms_per_step"], r["kernels_ms"])
except Exception as e: print("stop", sys.argv[2], "unreadable", e)
PY
done
