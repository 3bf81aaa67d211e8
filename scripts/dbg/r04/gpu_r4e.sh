#!/bin/bash
# Round 4: ct_miss_wide2 with two periods per wave, recursion_tile with direct operand reloads; then the full default bench line.
TAG=${1:-r4e}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_fuzz.py tests/test_gpu_em.py tests/test_gpu_round4.py tests/test_gpu_mstep_miss.py -q -m gpu --maxfail=12 2>&1 | tail -40 > $OUT/pytest.log
tail -8 $OUT/pytest.log
DFM_LIB=gpurun_tmp/libdfmhip_tprof.so B=256 N=1000 T=2000 R=20 MISSING=0.1 K=2 timeout 300 python scripts/gpu_trace.py 2>&1 | grep TILEPROF > $OUT/tileprof.txt
cat $OUT/tileprof.txt
timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --steps 2 --warmup 1 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_c4m.json 2> $OUT/bench_c4m.err
timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --mode em --steps 2 --warmup 1 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_c4m_em.json 2> $OUT/bench_c4m_em.err
for f in $OUT/bench_c4m*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "value=%.4g ms=%.3f" % (d["value"], d["ms_per_step"]), d["roofline"].get("kernels_ms"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    print("headline value=%.4g ms=%.4f frac=%s" % (d["value"], d["ms_per_step"], r.get("frac")))
    for k, v in (d.get("secondary") or {}).items():
        print("   ", k, {a: v.get(a) for a in ("value", "ms_per_step", "whole_step", "dominant", "kernels_ms", "seconds", "error", "gram", "cpu_baseline") if v.get(a) is not None})
except Exception as e:
    print("unreadable:", e)
PY
tail -5 $OUT/bench.err
