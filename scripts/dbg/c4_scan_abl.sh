#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-c4scan}; mkdir -p $OUT
for x in 0 2 8 16 32 64 128; do
  DFM_SCAN_ABL=$x DFM_NO_SIDE=1 timeout 200 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline > $OUT/abl$x.json 2> $OUT/abl$x.err
  python - $OUT/abl$x.json $x <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("scan abl", sys.argv[2], "ms %.4f"%d["ms_per_step"], r["kernels_ms"])
PY
done
