#!/bin/bash
# collapse_ks_kernel at config 4: ablations of the diagnostics build (WRONG results on purpose): stream only, compute only, no 4x4x4 part
TAG=${1:-ksabl}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in "DFM_KS_ABL=0" "DFM_KS_ABL=1" "DFM_KS_ABL=2" "DFM_KS_ABL=4" "DFM_KS_ABL=0 DFM_DMA_NT=0" "DFM_KS_ABL=1 DFM_DMA_NT=0" "DFM_KS_ABL=0 DFM_KS_IR=512" "DFM_KS_ABL=0 DFM_KS_IR=2000" "DFM_NO_COLLAPSE_KS=1"; do
  env DFM_LIB=diag $v timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --mode em --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/x.json 2> $OUT/x.err
  python - $OUT/x.json "$v" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print(sys.argv[2], "| em ms %.4f"%d["ms_per_step"], {k:v for k,v in r["kernels_ms"].items() if "collapse" in k or "prep" in k})
except Exception as e: print(sys.argv[2], "unreadable", e)
PY
done 2>&1 | tee $OUT/lines.txt
