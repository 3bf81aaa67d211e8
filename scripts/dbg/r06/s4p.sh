#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4p; mkdir -p $OUT; export TMPDIR=/tmp
DFM_LIB=diag DFM_SCAN_ABL=256 timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary 2>&1 | grep S3STAMP | tail -6 > $OUT/stamps.txt
DFM_LIB=diag DFM_SCAN_ABL=256 timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --mode em --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary 2>&1 | grep S3STAMP | tail -3 >> $OUT/stamps.txt
cat $OUT/stamps.txt
