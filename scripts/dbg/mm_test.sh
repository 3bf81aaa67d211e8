#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "kernel: $(uname -r)"
timeout 900 python -m pytest tests/test_gpu_em.py tests/test_gpu_round3.py tests/test_gpu_fuzz.py tests/test_gpu_ar_em.py -x -q -m gpu 2>&1 | tail -8
for M in 0 1; do
  echo "--- DFM_MSTEP_MISS=$M: EM iteration at config 4 with 10 % missing (B = 256)"
  DFM_MSTEP_MISS=$M timeout 600 python bench.py --mode em --missing 0.1 --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 --repeats 2 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('value=%.4g ms=%.3f'%(d['value'],d['ms_per_step']), d.get('kernels_ms'))"
done
