#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4l; mkdir -p $OUT; export TMPDIR=/tmp; rm -f $OUT/abl.txt
for abl in 0 1 4; do
  (cd /tmp && DFM_LIB=diag DFM_COMP_ABL=$abl timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st$abl -o b -- python $R/scripts/dbg/r06/f3_only.py > /dev/null 2>&1)
  echo "DFM_COMP_ABL=$abl" >> $OUT/abl.txt
  grep "recursion_comp_kernel" $(find $OUT/st$abl -name '*kernel_stats.csv' | head -1) | cut -d, -f1-4 | cut -c1-120 >> $OUT/abl.txt
  rm -rf $OUT/st$abl
done
cat $OUT/abl.txt
