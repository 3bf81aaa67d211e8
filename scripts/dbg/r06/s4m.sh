#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4m; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_varp.py tests/test_gpu_ar.py tests/test_gpu_ar_em.py tests/test_gpu_api.py -q -m gpu --maxfail=10 2>&1 | tail -8 > $OUT/pt.log
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/scripts/dbg/r06/f3_only.py > $OUT/f3.txt 2> $OUT/f3_rp.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_f3.csv 2>/dev/null; rm -rf $OUT/stats
cat $OUT/pt.log; tail -4 $OUT/f3.txt; grep "recursion_comp\|recursion_mbf16" $OUT/kernel_stats_f3.csv | cut -d, -f1-4 | cut -c1-110
