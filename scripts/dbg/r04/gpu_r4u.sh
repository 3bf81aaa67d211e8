#!/bin/bash
# config 4, balanced: where the 0.36 ms of meanscan_mfma_kernel go (phase stamps of workgroup 0, diagnostics build), and what the
# length of the contiguous row piece costs an LDS-DMA stream (scripts/microbench/segbw.hip)
TAG=${1:-r4u}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
timeout 120 scripts/microbench/segbw > $OUT/segbw.txt 2>&1; cat $OUT/segbw.txt
DFM_LIB=diag DFM_SCAN_ABL=256 timeout 200 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary > $OUT/c4_stamps.json 2> $OUT/c4_stamps.err
grep S3STAMP $OUT/c4_stamps.json $OUT/c4_stamps.err | tail -4 | tee $OUT/s3stamp.txt
