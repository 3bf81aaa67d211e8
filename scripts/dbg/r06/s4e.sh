#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4e; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_varp.py tests/test_gpu_api.py tests/test_gpu_ar.py tests/test_gpu_ar_em.py -q -m gpu --maxfail=10 2>&1 | tail -30 > $OUT/pt.log
timeout 300 python scripts/dbg/r06/f3_only.py > $OUT/f3.txt 2>&1
cat $OUT/pt.log; cat $OUT/f3.txt | tail -5
