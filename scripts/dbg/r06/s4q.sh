#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ks_pass.py tests/test_gpu_em.py tests/test_gpu_round4.py -q -m gpu --maxfail=10 2>&1 | tail -4 > $OUT/pt.log
for i in 1 2; do
timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 10 --warmup 2 --repeats 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 >> $OUT/c4.txt
DFM_LIB=$R/gpurun_tmp/libdfmhip_before.so timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 10 --warmup 2 --repeats 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed 's/^/before /' >> $OUT/c4.txt
done
cat $OUT/pt.log $OUT/c4.txt
