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
cd $R
python scripts/pmc_summary.py $OUT > $OUT/pmc_traffic.json 2> $OUT/pmc_summary.err
find $OUT -name "*.csv" | head -20
cat $OUT/pmc_traffic.json | head -40
