#!/bin/bash
# Round-3 GPU round: parity suite + smoke + the default bench line (headline + secondary lines), the one-launch pass's in-kernel
# timeline on THIS box, the library driver, kernel stats and a few PMC passes.
# Usage: scripts/gpu_r3.sh <tag>   (outputs under gpurun_out/<tag>/);  SKIP_TESTS=1 / SKIP_PMC=1 / SKIP_LIB=1
TAG=${1:-r3}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt; nproc >> $OUT/device.txt
(rocm-smi --showperflevel --showclocks --showpower --showmemorypartition --showcomputepartition 2>/dev/null | grep -v '^=' | grep -v '^$') >> $OUT/device.txt
[ -x $R/scripts/microbench/chainlat ] && timeout 60 $R/scripts/microbench/chainlat > $OUT/chainlat.txt 2>&1
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --maxfail=25 2>&1 | tail -80 > $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 > $OUT/smoke.log
fi
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
B=1024 DFM_PASS_NSW=4 timeout 120 python scripts/pf_prof.py > $OUT/fused_timeline_B1024.txt 2>&1
if [ -z "$SKIP_LIB" ]; then
  timeout 300 python bench.py --driver lib --mode em --steps 20 --warmup 3 --repeats 5 > $OUT/bench_lib_em.json 2> $OUT/bench_lib_em.err
  timeout 300 python bench.py --driver lib --mode em --force-comm --steps 20 --warmup 3 --repeats 5 > $OUT/bench_lib_em_comm.json 2> $OUT/bench_lib_em_comm.err
  timeout 300 python bench.py --driver lib --mode em --batch-per-gpu 8192 --steps 10 --warmup 2 --repeats 3 > $OUT/bench_lib_em_b8192.json 2> $OUT/bench_lib_em_b8192.err
  timeout 300 python bench.py --mode em --batch-per-gpu 8192 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline > $OUT/bench_em_b8192.json 2> $OUT/bench_em_b8192.err
  timeout 300 python bench.py --driver lib --mode pass --steps 20 --warmup 3 --repeats 5 > $OUT/bench_lib_pass.json 2> $OUT/bench_lib_pass.err
fi
if [ -z "$SKIP_PMC" ]; then
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err)
  cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
  rm -rf $OUT/stats
  (cd /tmp && timeout 60 rocprofv3 -L > $OUT/counters_avail.txt 2>&1)
  for C in SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU; do
    (cd /tmp && K=6 timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/pmc_$C.err)
    f=$(find $OUT/pmc_$C -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python - $f $C <<'PY' >> $OUT/pmc_counters.txt
import csv, sys, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    k = row.get("Kernel_Name", "?")[:60]
    tot[k][0] += float(row.get("Counter_Value", 0)); tot[k][1] += 1
for k, (v, n) in tot.items():
    print(sys.argv[2], k, "sum", v, "rows", n)
PY
    rm -rf $OUT/pmc_$C
  done
fi
(echo '--- after the runs'; rocm-smi --showclocks --showpower 2>/dev/null | grep -v '^=' | grep -v '^$') >> $OUT/device.txt
tail -25 $OUT/pytest_gpu.log 2>/dev/null; cat $OUT/smoke.log 2>/dev/null
for f in bench bench_lib_em bench_lib_em_comm bench_lib_em_b8192 bench_em_b8192 bench_lib_pass bench_under_rocprof; do
  [ -f $OUT/$f.json ] && python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], "value=%.4g %s ms=%.4f" % (d["value"], d["unit"], d["ms_per_step"]), d["timing"].get("ms_per_step_blocks"),
          "dom=%s frac=%s whole=%s" % (r.get("kernel"), r.get("frac"), (r.get("whole_step") or {}).get("frac")), r.get("kernels_ms"), "ceiling=", r.get("ceiling_measured"),
          "cpu=", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("thread_probe"), "dev=", d.get("device"))
    for k, v in (d.get("secondary") or {}).items():
        print("   ", k, {a: v.get(a) for a in ("value", "ms_per_step", "whole_step", "dominant", "kernels_ms", "seconds", "error") if v.get(a) is not None})
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -4 $OUT/*.err 2>/dev/null | head -80
cat $OUT/fused_timeline_B1024.txt | tail -22
cat $OUT/pmc_counters.txt 2>/dev/null | head -40
