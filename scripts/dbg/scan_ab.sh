#!/bin/bash
# A/B of the one-launch pass's scan (scan_reg vs the round-2 scan_lds, DFM_SCAN_ABL=512) on ONE box: parity first, then
# alternating bench lines, B = 8192, EM, and both in-kernel timelines.   Usage: scripts/dbg/scan_ab.sh <tag>
TAG=${1:-ab}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pass_fused.py tests/test_gpu_round3.py tests/test_gpu_em.py tests/test_gpu_fuzz.py tests/test_gpu_ks_pass.py -q -x 2>&1 | tail -30 > $OUT/pytest_sel.log
tail -5 $OUT/pytest_sel.log
B="--no-cpu-baseline --no-secondary --repeats 5"
for i in 1 2; do
  timeout 200 python bench.py $B > $OUT/new_$i.json 2> $OUT/new_$i.err
  DFM_SCAN_ABL=512 timeout 200 python bench.py $B > $OUT/old_$i.json 2> $OUT/old_$i.err
done
timeout 200 python bench.py $B --batch-per-gpu 8192 --steps 10 --warmup 2 > $OUT/new_b8192.json 2> $OUT/new_b8192.err
DFM_SCAN_ABL=512 timeout 200 python bench.py $B --batch-per-gpu 8192 --steps 10 --warmup 2 > $OUT/old_b8192.json 2> $OUT/old_b8192.err
timeout 200 python bench.py $B --mode em --steps 20 > $OUT/new_em.json 2> $OUT/new_em.err
DFM_SCAN_ABL=512 timeout 200 python bench.py $B --mode em --steps 20 > $OUT/old_em.json 2> $OUT/old_em.err
B=1024 DFM_PASS_NSW=4 timeout 120 python scripts/pf_prof.py > $OUT/timeline_new.txt 2>&1
B=1024 DFM_PASS_NSW=4 PF_ABL=512 timeout 120 python scripts/pf_prof.py > $OUT/timeline_old.txt 2>&1
for f in new_1 old_1 new_2 old_2 new_b8192 old_b8192 new_em old_em; do
  python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "value=%.4g ms=%.4f" % (d["value"], d["ms_per_step"]), d["timing"]["ms_per_step_blocks"], "whole=%.4f" % d["roofline"]["whole_step"]["frac"], d["roofline"]["kernels_ms"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -3 $OUT/*.err | grep -v amdgpu.ids | head -40
echo NEW; tail -26 $OUT/timeline_new.txt; echo OLD; tail -26 $OUT/timeline_old.txt | head -12
# the forced-communicator run of the library driver that died silently in round-3 call a
timeout 120 python -X faulthandler -u bench.py --driver lib --mode em --force-comm --steps 5 --warmup 1 --repeats 2 > $OUT/lib_comm.json 2> $OUT/lib_comm.err; echo "lib_comm rc=$?"
tail -30 $OUT/lib_comm.err; cat $OUT/lib_comm.json | head -c 600
