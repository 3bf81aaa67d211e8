#!/bin/bash
# Round-6 GPU round on the final tree: parity suite + smoke + the default bench line (headline + secondary lines), rocprofv3 kernel
# stats of the same command and of config 4 (balanced / 10 % missing / EM), PMC traffic passes (FETCH_SIZE, WRITE_SIZE: separate runs).
# Usage: scripts/gpu_r6.sh <tag>   (outputs under gpurun_out/<tag>/);  SKIP_TESTS=1 / SKIP_PMC=1
TAG=${1:-r6}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(uname -r; rocm-smi --showproductname 2>/dev/null | head -8; nproc) > $OUT/device.txt
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --maxfail=25 2>&1 | tail -40 > $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 > $OUT/smoke.log
fi
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/stats
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --missing 0.1 --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_missing10_under_rocprof.json 2> $OUT/bench_missing10_under_rocprof.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_missing10.csv 2>/dev/null; rm -rf $OUT/stats
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --missing 0.1 --batch-per-gpu 8192 --steps 3 --warmup 1 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_missing10_b8192_under_rocprof.json 2> $OUT/bench_missing10_b8192_under_rocprof.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_missing10_b8192.csv 2>/dev/null; rm -rf $OUT/stats
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --missing 0.1 --mode em --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_em_missing10_under_rocprof.json 2> $OUT/bench_em_missing10_under_rocprof.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_em_missing10.csv 2>/dev/null; rm -rf $OUT/stats
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --steps 2 --warmup 1 --repeats 2 --no-cpu-baseline --no-secondary > $OUT/bench_c4m_under_rocprof.json 2> $OUT/bench_c4m_under_rocprof.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_c4_missing10.csv 2>/dev/null; rm -rf $OUT/stats
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --mode em --steps 2 --warmup 1 --repeats 2 --no-cpu-baseline --no-secondary > $OUT/bench_c4m_em_under_rocprof.json 2> $OUT/bench_c4m_em_under_rocprof.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_c4_em_missing10.csv 2>/dev/null; rm -rf $OUT/stats
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_c4_under_rocprof.json 2> $OUT/bench_c4_under_rocprof.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_c4.csv 2>/dev/null; rm -rf $OUT/stats
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --mode em --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_em_under_rocprof.json 2> $OUT/bench_em_under_rocprof.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_em.csv 2>/dev/null; rm -rf $OUT/stats
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/scripts/dbg/r06/f3_only.py > $OUT/f3_under_rocprof.txt 2> $OUT/f3_under_rocprof.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_f3.csv 2>/dev/null; rm -rf $OUT/stats
if [ -z "$SKIP_PMC" ]; then
  (cd /tmp && K=6 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/pmc_fetch.err)
  (cd /tmp && K=6 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/pmc_write.err)
  if [ -x $R/scripts/microbench/readbw ]; then
    (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_calib -o p -- $R/scripts/microbench/readbw > $OUT/readbw.txt 2> $OUT/pmc_calib.err)
  fi
  python scripts/pmc_summary.py $OUT > $OUT/pmc_traffic.json 2> $OUT/pmc_summary.err
  for cfg in "missing10 1024 6" "missing10_b8192 8192 2"; do
    set -- $cfg
    mkdir -p $OUT/$1
    (cd /tmp && K=$3 B=$2 MISSING=0.1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$1/pmc_fetch -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/$1/pmc_fetch.err)
    (cd /tmp && K=$3 B=$2 MISSING=0.1 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$1/pmc_write -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/$1/pmc_write.err)
    [ -d $OUT/pmc_calib ] && cp -r $OUT/pmc_calib $OUT/$1/pmc_calib
    DFM_PMC_WORKLOAD=pass:B$2:N200:T500:r8:m0.1 python scripts/pmc_summary.py $OUT/$1 > $OUT/pmc_traffic_$1.json 2>> $OUT/pmc_summary.err
    rm -rf $OUT/$1
  done
  mkdir -p $OUT/c4m
  (cd /tmp && K=2 B=256 N=1000 T=2000 R=20 MISSING=0.1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c4m/pmc_fetch -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/c4m/pmc_fetch.err)
  (cd /tmp && K=2 B=256 N=1000 T=2000 R=20 MISSING=0.1 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c4m/pmc_write -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/c4m/pmc_write.err)
  [ -d $OUT/pmc_calib ] && cp -r $OUT/pmc_calib $OUT/c4m/pmc_calib
  DFM_PMC_WORKLOAD=pass:B256:N1000:T2000:r20:m0.1 python scripts/pmc_summary.py $OUT/c4m > $OUT/pmc_traffic_c4_missing10.json 2>> $OUT/pmc_summary.err
  mkdir -p $OUT/c4
  (cd /tmp && K=4 B=256 N=1000 T=2000 R=20 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c4/pmc_fetch -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/c4/pmc_fetch.err)
  (cd /tmp && K=4 B=256 N=1000 T=2000 R=20 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c4/pmc_write -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/c4/pmc_write.err)
  [ -d $OUT/pmc_calib ] && cp -r $OUT/pmc_calib $OUT/c4/pmc_calib
  DFM_PMC_WORKLOAD=pass:B256:N1000:T2000:r20:m0.0 python scripts/pmc_summary.py $OUT/c4 > $OUT/pmc_traffic_c4.json 2>> $OUT/pmc_summary.err
  rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_calib $OUT/c4m $OUT/c4
fi
tail -6 $OUT/pytest_gpu.log 2>/dev/null; cat $OUT/smoke.log 2>/dev/null
python - $OUT/bench.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    print("headline value=%.4g ms=%.4f frac=%s traffic=%s" % (d["value"], d["ms_per_step"], r.get("frac"), r.get("traffic")), d["timing"].get("ms_per_step_blocks"))
    for k, v in (d.get("secondary") or {}).items():
        print("   ", k, {a: v.get(a) for a in ("value", "ms_per_step", "whole_step", "kernels_ms", "seconds", "error") if v.get(a) is not None})
except Exception as e:
    print("unreadable:", e)
PY
head -12 $OUT/kernel_stats.csv; head -8 $OUT/kernel_stats_missing10.csv; head -6 $OUT/kernel_stats_f3.csv; cat $OUT/pmc_traffic.json | head -30; tail -3 $OUT/*.err | head -60
