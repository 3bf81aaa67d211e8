#!/bin/bash
# Round-4 first GPU call: the whole GPU suite (with the round-4 tests) + the default bench line on this box.
TAG=${1:-r4a}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt; rocm-smi --showproductname 2>/dev/null | head -8 >> $OUT/device.txt
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_fuzz.py -q -m gpu --maxfail=25 2>&1 | tail -60 > $OUT/pytest_r4.log
timeout 1200 python -m pytest tests -q -m gpu --maxfail=25 --deselect tests/test_gpu_round4.py --deselect tests/test_gpu_fuzz.py 2>&1 | tail -40 > $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -30 $OUT/pytest_r4.log; tail -8 $OUT/pytest_gpu.log; tail -3 $OUT/bench.err; head -c 3000 $OUT/bench.json
