#!/bin/bash
# config 4: edge tests, first-call check, bench under the two workgroup->tile maps (diagnostic)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-c4ab}; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_ks_pass.py -q -m gpu -k "wide or c4 or config4 or edge or general" 2>&1 | tail -4
timeout 200 python scripts/dbg/c4_repeat.py 2>&1 | grep "call 0\|-2x"
for x in ${XS:-1 0}; do
  DFM_WIDE_XCD=$x timeout 200 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 5 --warmup 2 --repeats 5 --no-cpu-baseline > $OUT/c4_xcd$x.json 2> $OUT/c4_xcd$x.err
  python - $OUT/c4_xcd$x.json $x <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("xcd", sys.argv[2], "ms %.4f"%d["ms_per_step"], r["kernels_ms"], "whole %.4f"%r["whole_step"]["frac"])
PY
done
