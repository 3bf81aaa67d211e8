#!/bin/bash
TAG=${1:-r4q}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
timeout 900 python -m pytest tests/test_gpu_pass_fused.py tests/test_gpu_em.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_ks_pass.py -q -m gpu --maxfail=12 2>&1 | tail -10 > $OUT/pytest.log
tail -4 $OUT/pytest.log
for nt in 0 1; do for mode in pass em; do
  DFM_DMA_NT=$nt timeout 300 python bench.py --mode $mode --steps 30 --warmup 5 --repeats 7 --no-cpu-baseline --no-secondary > $OUT/${mode}_nt$nt.json 2> $OUT/${mode}_nt$nt.err
  python - $OUT/${mode}_nt$nt.json $mode $nt <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print(sys.argv[2], "nt", sys.argv[3], "ms %.4f whole %.3f"%(d["ms_per_step"], r["whole_step"]["frac"]), r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
done; done
