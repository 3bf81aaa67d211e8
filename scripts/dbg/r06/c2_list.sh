#!/bin/bash
for mm in 1 2; do
  export DFM_MSTEP_MISS=$mm
  bash scripts/gpu_prof_one.sh c2l_$mm --missing 0.1 --mode em --steps 10 --warmup 3 > /dev/null 2>&1
  echo "MSTEP_MISS=$mm"; grep -E "mstep|mmw" gpurun_out/c2l_$mm/kernel_stats.csv | sed 's/(dfm::MstepArgs[^"]*"/"/' | cut -d, -f1-4
done
