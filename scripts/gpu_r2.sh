#!/bin/bash
# Round-2 GPU round: parity suite + smoke + the bench lines of BASELINE configs 2, 3 (per-GPU shard), 4, the EM and
# PCA modes, the 10 %-missing row, and the rocprofv3 kernel stats of the headline.
# Usage: scripts/gpu_r2.sh <tag>   (outputs under gpurun_out/<tag>/);  SKIP_TESTS=1 / SKIP_EXTRA=1 / PROFILE=1
TAG=${1:-r2}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt; nproc >> $OUT/device.txt
(rocm-smi --showperflevel --showclocks --showpower --showmemorypartition --showcomputepartition 2>/dev/null | grep -v '^=' | grep -v '^$') >> $OUT/device.txt
[ -x $R/scripts/microbench/chainlat ] && timeout 60 $R/scripts/microbench/chainlat > $OUT/chainlat.txt 2>&1
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --maxfail=25 2>&1 | tail -60 > $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 > $OUT/smoke.log
fi
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
if [ -z "$SKIP_EXTRA" ]; then
  DFM_PASS_FUSED=0 timeout 300 python bench.py --no-cpu-baseline --repeats 5 > $OUT/bench_two_launch.json 2> $OUT/bench_two_launch.err
  timeout 300 python bench.py --batch-per-gpu 8192 --steps 10 --warmup 2 --repeats 5 --no-cpu-baseline > $OUT/bench_b8192.json 2> $OUT/bench_b8192.err
  timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 5 --warmup 2 --repeats 5 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
  timeout 300 python bench.py --mode em --steps 20 --warmup 3 --repeats 5 > $OUT/bench_em.json 2> $OUT/bench_em.err
  timeout 300 python bench.py --mode pca --steps 5 --warmup 1 --repeats 5 > $OUT/bench_pca.json 2> $OUT/bench_pca.err
  timeout 300 python bench.py --missing 0.1 --steps 20 --warmup 3 --repeats 5 --no-cpu-baseline > $OUT/bench_missing10.json 2> $OUT/bench_missing10.err
  timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --steps 3 --warmup 1 --repeats 3 --no-cpu-baseline > $OUT/bench_c4_missing10.json 2> $OUT/bench_c4_missing10.err
  timeout 300 python bench.py --mode em --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 3 --warmup 1 --repeats 3 --no-cpu-baseline > $OUT/bench_c4_em.json 2> $OUT/bench_c4_em.err
  bash $R/scripts/dbg/extra_shapes.sh $TAG > /dev/null 2>&1
fi
if [ -n "$PROFILE" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err)
  cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
fi
(echo '--- after the runs'; rocm-smi --showclocks --showpower 2>/dev/null | grep -v '^=' | grep -v '^$') >> $OUT/device.txt
tail -25 $OUT/pytest_gpu.log 2>/dev/null; cat $OUT/smoke.log 2>/dev/null
for f in bench bench_two_launch bench_b8192 bench_c4 bench_em bench_pca bench_missing10 bench_c4_missing10 bench_c4_em; do
  [ -f $OUT/$f.json ] && python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], "value=%.4g %s ms=%.4f [%.4f..%.4f]" % (d["value"], d["unit"], d["ms_per_step"], d["timing"]["ms_per_step_min"], d["timing"]["ms_per_step_max"]),
          "dom=%s frac=%s whole=%s" % (r["kernel"], r.get("frac"), (r.get("whole_step") or {}).get("frac")), r["kernels_ms"], "cpu=", (d.get("cpu_baseline") or {}).get("value"), ((d.get("cpu_baseline") or {}).get("single_thread") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -3 $OUT/*.err 2>/dev/null | head -60
