#!/bin/bash
# Round 4: C_t kernel with five stage buffers + fallbacks.
TAG=${1:-r4g}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_fuzz.py tests/test_gpu_em.py tests/test_gpu_round4.py -q -m gpu --maxfail=12 2>&1 | tail -30 > $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --steps 2 --warmup 1 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_c4m.json 2> $OUT/bench_c4m.err
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "value=%.4g ms=%.4f whole=%.3f" % (d["value"], d["ms_per_step"], d["roofline"]["whole_step"]["frac"]), d["roofline"].get("kernels_ms"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
tail -3 $OUT/bench_c4m.err
