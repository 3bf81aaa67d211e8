#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of bench.py, then PMC passes
# (FETCH_SIZE and WRITE_SIZE in separate runs, never combined with other trace domains) on the same
# workload and on the read-bandwidth microbenchmark (known byte count = calibration of FETCH_SIZE).
# Usage: scripts/gpu_profile.sh [tag]   -> gpurun_out/<tag>/
TAG=${1:-prof}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
K=6 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/pmc_fetch.err
K=6 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/pmc_write.err
if [ -x $R/scripts/microbench/readbw ]; then
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_calib -o p -- $R/scripts/microbench/readbw > $OUT/readbw.txt 2> $OUT/pmc_calib.err
fi
# BASELINE config 4 (N = 1000, T = 2000, r = 20, 256 replicates): kernel stats + the same two PMC passes
mkdir -p $OUT/c4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4/stats -o bench -- python $R/bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline > $OUT/c4/bench_under_rocprof.json 2> $OUT/c4/bench_under_rocprof.err
cp $(find $OUT/c4/stats -name '*kernel_stats.csv' | head -1) $OUT/c4_kernel_stats.csv 2>/dev/null
K=3 B=256 N=1000 T=2000 R=20 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c4/pmc_fetch -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/c4/pmc_fetch.err
K=3 B=256 N=1000 T=2000 R=20 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c4/pmc_write -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/c4/pmc_write.err
[ -d $OUT/pmc_calib ] && cp -r $OUT/pmc_calib $OUT/c4/pmc_calib
cd $R
python scripts/pmc_summary.py $OUT > $OUT/pmc_traffic.json 2> $OUT/pmc_summary.err
DFM_PMC_WORKLOAD=pass:B256:N1000:T2000:r20:m0.0 python scripts/pmc_summary.py $OUT/c4 > $OUT/pmc_traffic_c4.json 2>> $OUT/pmc_summary.err
DFM_SCAN_ABL=256 K=3 B=256 N=1000 T=2000 R=20 timeout 100 python scripts/gpu_trace.py 2>&1 | grep S3STAMP > $OUT/c4_scan_phases.txt
find $OUT -name "*.csv" | head -20
cat $OUT/pmc_traffic.json | head -40
