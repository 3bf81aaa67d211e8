#!/bin/bash
# rocprofv3 kernel stats of one bench.py command line.  Usage: scripts/gpu_prof_one.sh <tag> <bench args...>   -> gpurun_out/<tag>/kernel_stats.csv
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/stats
head -12 $OUT/kernel_stats.csv | cut -c1-200
