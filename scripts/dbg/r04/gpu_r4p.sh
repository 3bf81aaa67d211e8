#!/bin/bash
TAG=${1:-r4p}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
for bb in 512 1024 1536 2048 3072 4096; do
  timeout 300 python scripts/dbg/inproc_ab.py 0 LIB=gpurun_tmp/libdfmhip_nt.so LIB=gpurun_tmp/libdfmhip_ntsc1.so batch=$bb 2>&1 | grep "median" | sed "s/^/B=$bb /" >> $OUT/ab_nt_sweep.txt
done
cat $OUT/ab_nt_sweep.txt | cut -c1-140
