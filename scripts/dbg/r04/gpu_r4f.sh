#!/bin/bash
# Round 4: C_t kernel v3 (W-only stages of 64 series, compact rows), transition M-step inside the loadings step's launch.
TAG=${1:-r4f}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
timeout 1200 python -m pytest tests -q -m gpu --maxfail=12 2>&1 | tail -40 > $OUT/pytest.log
tail -8 $OUT/pytest.log
timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --steps 2 --warmup 1 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_c4m.json 2> $OUT/bench_c4m.err
timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --mode em --steps 2 --warmup 1 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_c4m_em.json 2> $OUT/bench_c4m_em.err
timeout 300 python bench.py --mode em --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/bench_em.json 2> $OUT/bench_em.err
DFM_NO_DEFER_EM=1 timeout 300 python bench.py --mode em --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/bench_em_nodefer.json 2> $OUT/bench_em_nodefer.err
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "value=%.4g ms=%.4f whole=%.3f" % (d["value"], d["ms_per_step"], d["roofline"]["whole_step"]["frac"]), d["roofline"].get("kernels_ms"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
tail -3 $OUT/bench_em.err
