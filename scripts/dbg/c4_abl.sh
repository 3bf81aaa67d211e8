#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-c4abl}; mkdir -p $OUT
for x in ${ABLS:-0 1 2 3 4 6 14}; do
  DFM_W2_ABL=$x timeout 200 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline > $OUT/abl$x.json 2> $OUT/abl$x.err
  python - $OUT/abl$x.json $x <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("abl", sys.argv[2], "ms %.4f"%d["ms_per_step"], "collapse %.4f"%r["kernels_ms"]["collapse_wide_kernel"])
PY
done
