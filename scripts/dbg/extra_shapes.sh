#!/bin/bash
# pass / EM iteration at shapes outside the BASELINE configs (Rp = 16; narrow states with wide cross-sections)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-extra}; mkdir -p $OUT
for c in "--r 12" "--r 16 --N 400" "--r 8 --N 1000 --batch-per-gpu 512"; do
  for m in pass em; do
    timeout 300 python bench.py --mode $m $c --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$m', '$c', '| value %.4g %s | ms %.4f | whole %s |' % (d['value'], d['unit'], d['ms_per_step'], (r.get('whole_step') or {}).get('frac')), r['kernels_ms'])"
  done
done | tee $OUT/extra_shapes.txt
for c in "--N 1000 --T 2000 --r 20 --batch-per-gpu 256" "--r 12" "--r 16 --N 400" "--r 8 --N 1000 --batch-per-gpu 512"; do
  timeout 500 python bench.py --mode pca $c --steps 2 --warmup 1 --repeats 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('pca', '$c', '| value %.4g %s | ms %.4f |' % (d['value'], d['unit'], d['ms_per_step']), r['kernels_ms'])"
done | tee $OUT/extra_pca.txt
