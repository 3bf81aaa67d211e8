#!/bin/bash
# timing ablations of mstep_miss_list_kernel (DFM_ML_DBG bits; results are WRONG on purpose).  Usage: ml_ablate.sh "0 1 2 4 8"
for d in ${1:-0 1 2 4 8 9 15}; do
  export DFM_ML_DBG=$d
  bash scripts/gpu_prof_one.sh abl_$d --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --mode em --missing 0.1 --steps 2 --warmup 1 > /dev/null 2>&1
  echo "dbg=$d $(grep list_kernel gpurun_out/abl_$d/kernel_stats.csv | sed 's/.*unsigned int)",//' | cut -d, -f1-3)"
done
