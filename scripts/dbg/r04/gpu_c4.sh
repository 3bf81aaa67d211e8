#!/bin/bash
# config 4 balanced: the pass and EM lines, K times each (usage: gpu_c4.sh TAG [K] [pytest-args...])
TAG=${1:-c4}; K=${2:-2}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
if [ -n "$3" ]; then
  timeout 1200 python -m pytest ${@:3} -q -m gpu -x 2>&1 | grep "passed\|failed\|rror\|assert" | tail -8 > $OUT/pytest.log
  cat $OUT/pytest.log
fi
for k in $(seq 1 $K); do
for mode in pass em; do
  timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --mode $mode --steps 10 --warmup 3 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/c4_${mode}.json 2> $OUT/c4_${mode}.err
  python - $OUT/c4_${mode}.json $mode <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("c4", sys.argv[2], "ms %.4f whole %.3f"%(d["ms_per_step"], r["whole_step"]["frac"]), r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
done; done 2>&1 | tee $OUT/c4_lines.txt
