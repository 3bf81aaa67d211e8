#!/bin/bash
TAG=${1:-r4n}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
for mm in 1 2; do for bb in 1024 8192; do
  DFM_MSTEP_MISS=$mm timeout 300 python bench.py --mode em --missing 0.1 --batch-per-gpu $bb --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/em_m${mm}_b$bb.json 2> $OUT/em_m${mm}_b$bb.err
  python - $OUT/em_m${mm}_b$bb.json $mm $bb <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("MSTEP_MISS", sys.argv[2], "B", sys.argv[3], "ms %.4f"%d["ms_per_step"], r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
done; done
