#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4k; mkdir -p $OUT; export TMPDIR=/tmp
DFM_LIB=diag timeout 1200 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_ar_em.py tests/test_gpu_fuzz.py tests/test_gpu_ks_pass.py tests/test_gpu_mstep_miss.py tests/test_gpu_pass_fused.py tests/test_gpu_chunk.py -q -m gpu --maxfail=10 2>&1 | tail -15 > $OUT/pytest_diag.log
cat $OUT/pytest_diag.log
