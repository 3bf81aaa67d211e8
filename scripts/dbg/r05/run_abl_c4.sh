#!/bin/bash
# development: time config 4 with missing cells with each ablation library.  usage: run_abl_c4.sh <base> v1 v2 ...
base=$1; shift
for v in "$@"; do
  echo -n "ABL $base $v: "
  lib=dynamic_factor_models_amd/lib/abl/libdfm_${base}_$v.so
  [ "$v" = "0" ] && lib=dynamic_factor_models_amd/lib/libdfmhip.so
  DFM_LIB=$lib timeout 300 python bench.py --batch-per-gpu 256 --N 1000 --T 2000 --r 20 --missing 0.1 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --repeats 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernels_ms'])"
done
