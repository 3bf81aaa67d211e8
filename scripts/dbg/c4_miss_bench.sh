#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-c4miss}; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_em.py -q -m gpu -x -k "path_and_params" 2>&1 | tail -2
timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --steps 3 --warmup 1 --repeats 3 --no-cpu-baseline > $OUT/c4_missing10.json 2> $OUT/c4_missing10.err
tail -2 $OUT/c4_missing10.err
python - $OUT/c4_missing10.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("c4 missing", "value %.4g"%d["value"], "ms %.4f"%d["ms_per_step"], r["kernels_ms"])
PY
