#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4o; mkdir -p $OUT; export TMPDIR=/tmp
echo after > $OUT/als.txt; timeout 300 python scripts/dbg/r06/c1c5_only.py 2>&1 | grep -v amdgpu >> $OUT/als.txt
echo before >> $OUT/als.txt; DFM_LIB=$R/gpurun_tmp/libdfmhip_before.so timeout 300 python scripts/dbg/r06/c1c5_only.py 2>&1 | grep -v amdgpu >> $OUT/als.txt
timeout 900 python -m pytest tests/test_gpu_als.py -q -m gpu 2>&1 | tail -3 >> $OUT/als.txt
cat $OUT/als.txt
