#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4i; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_chunk.py -q -m gpu --maxfail=10 2>&1 | tail -40 > $OUT/pt.log
for m in pass em; do
  timeout 300 python bench.py --missing 0.1 --batch-per-gpu 8192 --mode $m --steps 3 --warmup 1 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/b8192_$m.json 2> $OUT/b8192_$m.err
  echo $m $(grep -o '"ms_per_step": [0-9.]*' $OUT/b8192_$m.json | head -1)
done
cat $OUT/pt.log
