#!/bin/bash
# One gpurun call: parity suite + smoke + bench + probe timings + kernel-trace timeline.
# Usage: scripts/gpu_round.sh <tag> [probe args...]
TAG=${1:-round}; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 > $OUT/smoke.log
fi
timeout 400 python bench.py --steps 30 --warmup 5 ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err
if [ -n "$PROBE" ]; then
  timeout 600 python scripts/gpu_probe.py --skip-parity "$@" > $OUT/probe.log 2>&1
fi
if [ -n "$TRACE" ]; then
  (cd /tmp && K=6 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/trace.err)
  python scripts/trace_summary.py $(find $OUT/trace -name '*kernel_trace.csv' | head -1) 24 > $OUT/timeline.txt 2>&1
fi
tail -3 $OUT/pytest_gpu.log 2>/dev/null; cat $OUT/smoke.log 2>/dev/null; cat $OUT/bench.json; tail -2 $OUT/bench.err
