#!/bin/bash
# session 4, call c: r <= 4 on the 8-wide state through collapse_miss_kernel's table mode (lam_w): tests + the config-1 lines, A/B in the diagnostics build
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_chunk.py tests/test_gpu_api.py tests/test_gpu_em.py tests/test_gpu_ks_pass.py tests/test_gpu_round4.py tests/test_gpu_fuzz.py -q -m gpu --maxfail=10 2>&1 | tail -45 > $OUT/pt.log
timeout 300 python scripts/dbg/r06/f3_only.py > $OUT/f3.txt 2>&1
DFM_LIB=diag DFM_NARROW_TAB=0 DFM_ODD_PAD8=0 timeout 300 python scripts/dbg/r06/f3_only.py > $OUT/f3_old.txt 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/scripts/dbg/r06/f3_only.py > /dev/null 2> $OUT/f3_rp.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_f3.csv 2>/dev/null; rm -rf $OUT/stats
cat $OUT/pt.log; grep c1_em $OUT/f3.txt $OUT/f3_old.txt
