#!/bin/bash
# config 4 balanced after the mean scan's rework: GPU tests, bench lines (P_smooth fill as a burst beside the scan / as a trickle of
# n workgroups under the collapse), phase stamps of workgroup 0
TAG=${1:-r4v}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | grep "passed\|failed\|rror" | tail -5 > $OUT/pytest.log
cat $OUT/pytest.log
for k in 1 2; do
for tr in 0 16 32 64 128; do
  DFM_PFILL_TRICKLE=$tr timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --mode pass --steps 10 --warmup 3 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/c4_pass_tr$tr.json 2> $OUT/c4_pass_tr$tr.err
  python - $OUT/c4_pass_tr$tr.json $tr <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("c4 pass trickle", sys.argv[2], "ms %.4f whole %.3f"%(d["ms_per_step"], r["whole_step"]["frac"]), r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
done; done 2>&1 | tee $OUT/c4_lines.txt
timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --mode em --steps 10 --warmup 3 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/c4_em.json 2> $OUT/c4_em.err
python - $OUT/c4_em.json <<'PY' | tee -a $OUT/c4_lines.txt
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
print("c4 em ms %.4f whole %.3f"%(d["ms_per_step"], r["whole_step"]["frac"]), r["kernels_ms"])
PY
DFM_LIB=diag DFM_SCAN_ABL=256 timeout 200 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary > $OUT/c4_stamps.json 2> $OUT/c4_stamps.err
grep -h S3STAMP $OUT/c4_stamps.json $OUT/c4_stamps.err | tail -4 | tee $OUT/s3stamp.txt
