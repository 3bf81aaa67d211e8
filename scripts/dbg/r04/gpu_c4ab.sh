#!/bin/bash
# config 4 balanced: A/B of a route switch (usage: gpu_c4ab.sh TAG ENVVAR [K] [pytest-args...]): lines with ENVVAR unset / =1
TAG=${1:-c4ab}; VAR=$2; K=${3:-2}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
if [ -n "$4" ]; then
  timeout 1500 python -m pytest ${@:4} -q -m gpu -x 2>&1 | grep "passed\|failed\|rror\|assert" | tail -8 > $OUT/pytest.log
  cat $OUT/pytest.log
fi
for k in $(seq 1 $K); do
for off in 0 1; do
for mode in pass em; do
  env $VAR=$off timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --mode $mode --steps 10 --warmup 3 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/c4_${mode}_$off.json 2> $OUT/c4_${mode}_$off.err
  python - $OUT/c4_${mode}_$off.json $mode $VAR $off <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("c4", sys.argv[2], sys.argv[3], sys.argv[4], "ms %.4f whole %.3f"%(d["ms_per_step"], r["whole_step"]["frac"]), r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
done; done; done 2>&1 | tee $OUT/c4_lines.txt
