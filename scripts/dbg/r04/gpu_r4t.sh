#!/bin/bash
TAG=${1:-r4t}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
timeout 900 python -m pytest tests/test_gpu_pass_fused.py tests/test_gpu_em.py tests/test_gpu_round4.py tests/test_gpu_round3.py -q -m gpu --maxfail=12 2>&1 | grep "passed\|failed" > $OUT/pytest.log
cat $OUT/pytest.log
for mo in 0 1 0 1; do
  DFM_MALL_ORDER=$mo timeout 300 python bench.py --mode em --steps 30 --warmup 5 --repeats 7 --no-cpu-baseline --no-secondary > $OUT/em_mo$mo.json 2> $OUT/em_mo$mo.err
  python - $OUT/em_mo$mo.json $mo <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("em MALL_ORDER", sys.argv[2], "ms %.4f whole %.3f"%(d["ms_per_step"], r["whole_step"]["frac"]), r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
done
