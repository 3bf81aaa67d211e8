#!/bin/bash
# One box: parity of the default tree, then the variants of the one-launch pass selected by DFM_SCAN_ABL bits
#   512 = round-2 scan; 1024 = scan priority always 3; 2048 = always 0 (default: dynamic); bits 12-14 = deferred fills + 1
TAG=${1:-ab2}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pass_fused.py tests/test_gpu_round3.py tests/test_gpu_em.py tests/test_gpu_fuzz.py tests/test_gpu_ks_pass.py -q -x 2>&1 | tail -30 > $OUT/pytest_sel.log
tail -5 $OUT/pytest_sel.log
for ABL in 8192 4096 12288 ; do
  DFM_SCAN_ABL=$ABL timeout 300 python -m pytest tests/test_gpu_pass_fused.py -q -x -k "matches_oracle or full_size" 2>&1 | tail -2
done
B="--no-cpu-baseline --no-secondary --repeats 5"
for rep in 1 2; do
for ABL in 0 4096 12288 5120 6144 4608 512; do
  DFM_SCAN_ABL=$ABL timeout 200 python bench.py $B > $OUT/b1024_${ABL}_$rep.json 2> $OUT/b1024_${ABL}_$rep.err
done
done
for ABL in 0 4096 5120 6144 4608; do
  DFM_SCAN_ABL=$ABL timeout 200 python bench.py $B --batch-per-gpu 8192 --steps 10 --warmup 2 --repeats 3 > $OUT/b8192_$ABL.json 2> $OUT/b8192_$ABL.err
done
for ABL in 0 4608; do
  DFM_SCAN_ABL=$ABL timeout 200 python bench.py $B --mode em --steps 20 > $OUT/em_$ABL.json 2> $OUT/em_$ABL.err
done
B=1024 DFM_PASS_NSW=4 timeout 120 python scripts/pf_prof.py > $OUT/timeline_new.txt 2>&1
for f in $OUT/b1024_*.json $OUT/b8192_*.json $OUT/em_*.json; do
  python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "value=%.4g ms=%.4f" % (d["value"], d["ms_per_step"]), d["timing"]["ms_per_step_blocks"], "whole=%.4f" % d["roofline"]["whole_step"]["frac"], d["roofline"]["kernels_ms"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -q -n 3 $OUT/*.err | grep -v amdgpu.ids | head -40
tail -26 $OUT/timeline_new.txt
