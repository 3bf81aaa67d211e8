#!/bin/bash
# config 4 balanced with the covariance recursion in the tile layout (cov_tile_kernel): GPU tests, then the bench lines with it
# (default) and without (DFM_NO_COV_TILE=1: cov_grid_kernel<32>)
TAG=${1:-r4x}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | grep "passed\|failed\|rror\|assert" | tail -8 > $OUT/pytest.log
cat $OUT/pytest.log
for k in 1 2; do
for off in 0 1; do
for mode in pass em; do
  DFM_NO_COV_TILE=$off timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --mode $mode --steps 10 --warmup 3 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/c4_${mode}_off$off.json 2> $OUT/c4_${mode}_off$off.err
  python - $OUT/c4_${mode}_off$off.json $mode $off <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("c4", sys.argv[2], "NO_COV_TILE", sys.argv[3], "ms %.4f whole %.3f"%(d["ms_per_step"], r["whole_step"]["frac"]), r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
done; done; done 2>&1 | tee $OUT/c4_lines.txt
