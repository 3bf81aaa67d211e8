#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "kernel: $(uname -r)"
timeout 900 python -m pytest tests -q -m gpu -k "config4 or c4 or wide or Rp16 or r20 or r16" 2>&1 | grep -E "passed|failed" | tail -2
for S in 1 2 4; do
  for MODE in pass em; do
  echo "--- DFM_WIDE_SUB=$S  config 4 $MODE"
  DFM_WIDE_SUB=$S timeout 300 python bench.py --mode $MODE --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --no-cpu-baseline --no-secondary --steps 10 --warmup 3 --repeats 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('value=%.5g ms=%.4f whole=%.3f'%(d['value'],d['ms_per_step'],d['roofline']['whole_step']['frac']), d['roofline'].get('kernels_ms'))"
  done
done
