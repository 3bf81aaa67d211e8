#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4j; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu --maxfail=10 2>&1 | tail -30 > $OUT/pt.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 > $OUT/smoke.log
cat $OUT/pt.log $OUT/smoke.log
