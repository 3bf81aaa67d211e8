#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4n; mkdir -p $OUT; export TMPDIR=/tmp
echo "after (cpick / pick4 prvalues)" > $OUT/mbf.txt
timeout 300 python scripts/dbg/r06/mbf_time.py 2>&1 | grep -v amdgpu >> $OUT/mbf.txt
echo "before" >> $OUT/mbf.txt
DFM_LIB=$R/gpurun_tmp/libdfmhip_before.so timeout 300 python scripts/dbg/r06/mbf_time.py 2>&1 | grep -v amdgpu >> $OUT/mbf.txt
timeout 1200 python -m pytest tests/test_gpu_varp.py tests/test_gpu_ar.py tests/test_gpu_ar_em.py -q -m gpu --maxfail=10 2>&1 | tail -4 >> $OUT/mbf.txt
cat $OUT/mbf.txt
