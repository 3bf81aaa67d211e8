#!/bin/bash
# nt on the PANEL rows only of the wide kernels: config 4 pass and EM per library
TAG=${1:-r4s}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
for n in default w2nt mwnt bothnt; do
  L=""; [ $n != default ] && L=$R/gpurun_tmp/libdfmhip_$n.so
  for mode in pass em; do
    DFM_LIB=$L timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --mode $mode --steps 5 --warmup 2 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/c4_${mode}_$n.json 2> $OUT/c4_${mode}_$n.err
    python - $OUT/c4_${mode}_$n.json $n $mode <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("c4", sys.argv[3], sys.argv[2], "ms %.4f whole %.3f"%(d["ms_per_step"], r["whole_step"]["frac"]), r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
  done
done
