#!/bin/bash
TAG=${1:-r4l}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
timeout 600 python scripts/dbg/cold_ab.py 0 LIB=gpurun_tmp/libdfmhip_r3.so 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" > $OUT/cold_ab.txt
cat $OUT/cold_ab.txt
