#!/bin/bash
# EM at C2: one workgroup per replicate in the loadings step (transition step, stream, finish in one launch; default) against the
# round-4 layout (front waves + segment waves + mstep_finish_kernel: DFM_MSTEP_FINISH=1)
TAG=${1:-emfin}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_em.py tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_ks_pass.py -q -m gpu -x 2>&1 | grep "passed\|failed\|rror" | tail -4 | tee $OUT/pytest.log
for k in 1 2 3; do
for sep in 0 1; do
  DFM_MSTEP_FINISH=$sep timeout 300 python bench.py --mode em --steps 30 --warmup 5 --repeats 7 --no-cpu-baseline --no-secondary > $OUT/em_$sep.json 2> $OUT/em_$sep.err
  python - $OUT/em_$sep.json $sep <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("em MSTEP_FINISH", sys.argv[2], "ms %.4f whole %.3f"%(d["ms_per_step"], r["whole_step"]["frac"]), r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
done; done 2>&1 | tee $OUT/lines.txt
for sep in 0 1; do
  DFM_MSTEP_FINISH=$sep timeout 300 python bench.py --mode em --batch-per-gpu 8192 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/em8_$sep.json 2> $OUT/em8_$sep.err
  python - $OUT/em8_$sep.json $sep <<'PY' | tee -a $OUT/lines.txt
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("em B=8192 MSTEP_FINISH", sys.argv[2], "ms %.4f whole %.3f"%(d["ms_per_step"], r["whole_step"]["frac"]), r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
done
