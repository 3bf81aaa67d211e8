#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-c4style}; mkdir -p $OUT
run() { # name, env...
  env "${@:2}" timeout 200 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline > $OUT/$1.json 2> $OUT/$1.err
  python - $OUT/$1.json $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2], "ms %.4f"%d["ms_per_step"], "collapse %.4f"%r["kernels_ms"]["collapse_wide_kernel"], "whole %.4f"%r["whole_step"]["frac"], {k:v for k,v in r["kernels_ms"].items() if k!="collapse_wide_kernel"})
PY
}
timeout 400 python -m pytest tests/test_gpu_ks_pass.py -q -m gpu -k "wide or c4 or config4 or edge or general" 2>&1 | tail -3
timeout 200 python scripts/dbg/c4_repeat.py 2>&1 | grep "call 0\|-2x"
run s0 A=1
run serial DFM_NO_SIDE=1
run stream_only DFM_NO_SIDE=1 DFM_W2_ABL=4
run compute_only DFM_NO_SIDE=1 DFM_W2_ABL=32
run s0_again A=1
DFM_NO_SIDE=1 DFM_W2_ABL=256 K=2 B=256 N=1000 T=2000 R=20 timeout 100 python scripts/gpu_trace.py 2>&1 | grep W2STAMP | head -140 > $OUT/stamps.txt
wc -l $OUT/stamps.txt
