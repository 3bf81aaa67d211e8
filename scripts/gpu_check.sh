#!/bin/bash
# Run on the GPU box through gpurun: parity tests, smoke, bench, rocprofv3 kernel trace.
# Usage: scripts/gpu_check.sh [tag]   (outputs under gpurun_out/<tag>/)
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.txt
nproc >> $OUT/device.txt
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -40 > $OUT/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_prof.json 2> $GRAFT_REPO_ROOT/$OUT/bench_prof.err)
find $OUT/prof -name '*stats*' | head
tail -5 $OUT/pytest_gpu.log; cat $OUT/smoke.log | tail -3; cat $OUT/bench.json; tail -3 $OUT/bench.err
