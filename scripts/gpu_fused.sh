#!/bin/bash
# Fused-pass round: parity of the one-launch pass, then A/B bench lines (two-launch default vs DFM_PASS_FUSED=1, stream-wave sweep).
TAG=${1:-fused}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pass_fused.py -q --maxfail=30 2>&1 | tail -70 > $OUT/pytest_fused.log
tail -40 $OUT/pytest_fused.log
if grep -q "passed" $OUT/pytest_fused.log; then
  DFM_PASS_FUSED=0 timeout 200 python bench.py --no-cpu-baseline --repeats 5 --steps 30 > $OUT/bench_two.json 2> $OUT/bench_two.err
  for nsw in 5 4 3; do
    DFM_PASS_FUSED=1 DFM_PASS_NSW=$nsw timeout 200 python bench.py --no-cpu-baseline --repeats 5 --steps 30 > $OUT/bench_fused_$nsw.json 2> $OUT/bench_fused_$nsw.err
  done
  DFM_PASS_FUSED=1 timeout 200 python bench.py --no-cpu-baseline --repeats 5 --steps 10 --batch-per-gpu 8192 > $OUT/bench_fused_b8192.json 2> $OUT/bench_fused_b8192.err
  DFM_PASS_FUSED=1 timeout 200 python bench.py --no-cpu-baseline --repeats 5 --steps 20 --mode em > $OUT/bench_fused_em.json 2> $OUT/bench_fused_em.err
  for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1].split("/")[-1], "value=%.4g ms=%.4f [%.4f..%.4f]" % (d["value"], d["ms_per_step"], d["timing"]["ms_per_step_min"], d["timing"]["ms_per_step_max"]), "dom=%s frac=%.3f whole=%.3f" % (r["kernel"], r["frac"], r["whole_step"]["frac"]), r["kernels_ms"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
  done
fi
tail -3 $OUT/*.err 2>/dev/null | head -40
