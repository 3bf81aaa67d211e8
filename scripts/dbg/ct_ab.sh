#!/bin/bash
# config 4 with 10 % missing cells: parity of the Rp = 32 missing-cell path, then ct_miss_wide2 vs the round-2 kernel (DFM_CT_OLD=1)
TAG=${1:-ct}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_ks_pass.py tests/test_gpu_em.py tests/test_gpu_fuzz.py -q -x 2>&1 | grep -v "^$" | tail -6
B="--no-cpu-baseline --no-secondary --repeats 3 --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --steps 2 --warmup 1"
for rep in 1 2; do
  timeout 300 python bench.py $B > $OUT/new_$rep.json 2> $OUT/new_$rep.err
  DFM_CT_OLD=1 timeout 300 python bench.py $B > $OUT/old_$rep.json 2> $OUT/old_$rep.err
done
for f in $OUT/new_1 $OUT/old_1 $OUT/new_2 $OUT/old_2; do
  python - $f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "value=%.4g ms=%.4f" % (d["value"], d["ms_per_step"]), d["roofline"]["kernels_ms"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -q -n 3 $OUT/*.err | grep -v amdgpu.ids | head
