#!/bin/bash
TAG=${1:-r4i}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_pca_synth.py tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_ks_pass.py -q -m gpu --maxfail=12 2>&1 | tail -30 > $OUT/pytest.log
tail -8 $OUT/pytest.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 2 --warmup 1 --repeats 2 --no-cpu-baseline --no-secondary > $OUT/bench_c4.json 2> $OUT/bench_c4.err)
grep -i "synth" $(find $OUT/stats -name '*kernel_stats.csv' | head -1); rm -rf $OUT/stats
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/bench.py --steps 5 --warmup 1 --repeats 2 --no-cpu-baseline --no-secondary > $OUT/bench_c2.json 2> $OUT/bench_c2.err)
grep -i "synth" $(find $OUT/stats -name '*kernel_stats.csv' | head -1); rm -rf $OUT/stats
