#!/bin/bash
# development: time the missing-cell pass with each ablation library.  usage: run_abl_lib.sh <base> v1 v2 ...
base=$1; shift
for v in "$@"; do
  echo -n "ABL $base $v: "
  DFM_LIB=dynamic_factor_models_amd/lib/abl/libdfm_${base}_$v.so timeout 200 python bench.py --missing 0.1 --no-cpu-baseline --no-secondary --steps 10 --warmup 3 --repeats 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernels_ms'])"
done
