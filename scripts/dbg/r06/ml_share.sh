#!/bin/bash
# list form against the dense product of the loadings step over the share of missing cells (config 4)
for miss in 0.02 0.05 0.2 0.3; do
  for mode in 0 2; do
    export DFM_MSTEP_LIST=$mode
    bash scripts/gpu_prof_one.sh sh_${miss}_$mode --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --mode em --missing $miss --steps 2 --warmup 1 > /dev/null 2>&1
    echo "missing=$miss list_mode=$mode $(grep 'mstep_miss_.*kernel<' gpurun_out/sh_${miss}_$mode/kernel_stats.csv | sed 's/(dfm::MstepArgs.*)",/ /' | cut -d, -f1-3)"
  done
done
