#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4b; mkdir -p $OUT; export TMPDIR=/tmp
B=1024 K=20 timeout 300 python scripts/dbg/r06/overlap_probe2.py > $OUT/overlap2.txt 2>&1
B=4096 K=6 timeout 300 python scripts/dbg/r06/overlap_probe2.py >> $OUT/overlap2.txt 2>&1
cat $OUT/overlap2.txt
