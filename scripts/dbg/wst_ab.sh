#!/bin/bash
# weights staging of the one-launch pass (DFM_SCAN_ABL bit 16 = off): parity, then alternating bench lines on one box
TAG=${1:-wst}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pass_fused.py tests/test_gpu_round3.py tests/test_gpu_em.py tests/test_gpu_fuzz.py tests/test_gpu_ks_pass.py -q -x 2>&1 | grep -v "^$" | tail -4
B="--no-cpu-baseline --no-secondary --repeats 5"
for rep in 1 2 3; do
for ABL in 0 65536; do
  DFM_SCAN_ABL=$ABL timeout 200 python bench.py $B > $OUT/b1024_${ABL}_$rep.json 2> $OUT/b1024_${ABL}_$rep.err
done
done
for rep in 1 2; do
for ABL in 0 65536; do
  DFM_SCAN_ABL=$ABL timeout 200 python bench.py $B --batch-per-gpu 8192 --steps 10 --warmup 2 --repeats 3 > $OUT/b8192_${ABL}_$rep.json 2> $OUT/b8192_${ABL}_$rep.err
done
done
for ABL in 0 65536; do
  DFM_SCAN_ABL=$ABL timeout 200 python bench.py $B --mode em --steps 20 > $OUT/em_$ABL.json 2> $OUT/em_$ABL.err
done
B=1024 DFM_PASS_NSW=4 timeout 120 python scripts/pf_prof.py > $OUT/timeline_new.txt 2>&1
for f in $OUT/b1024_*.json $OUT/b8192_*.json $OUT/em_*.json; do
  python - $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "value=%.4g ms=%.4f" % (d["value"], d["ms_per_step"]), d["timing"]["ms_per_step_blocks"], "whole=%.4f" % d["roofline"]["whole_step"]["frac"], d["roofline"]["kernels_ms"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -q -n 3 $OUT/*.err | grep -v amdgpu.ids | head -20
tail -6 $OUT/timeline_new.txt
