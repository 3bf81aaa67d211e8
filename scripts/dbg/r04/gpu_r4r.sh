#!/bin/bash
TAG=${1:-r4r}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
timeout 900 python -m pytest tests/test_gpu_ks_pass.py tests/test_gpu_em.py tests/test_gpu_round3.py -q -m gpu --maxfail=12 2>&1 | tail -5 > $OUT/pytest.log
tail -2 $OUT/pytest.log
for nt in 0 1; do for mode in pass em; do
  DFM_DMA_NT=$nt timeout 300 python bench.py --mode $mode --missing 0.1 --steps 10 --warmup 3 --repeats 7 --no-cpu-baseline --no-secondary > $OUT/${mode}_m_nt$nt.json 2> $OUT/${mode}_m_nt$nt.err
  python - $OUT/${mode}_m_nt$nt.json $mode $nt <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print(sys.argv[2], "missing10 nt", sys.argv[3], "ms %.4f whole %.3f"%(d["ms_per_step"], r["whole_step"]["frac"]), r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
done; done
