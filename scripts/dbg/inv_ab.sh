#!/bin/bash
# Rp = 32 paths after a change to Grid<32>: parity, then config 4 (balanced, EM, 10 % missing) bench lines
TAG=${1:-inv}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_ks_pass.py tests/test_gpu_em.py tests/test_gpu_fuzz.py tests/test_gpu_varp.py tests/test_gpu_ar.py tests/test_gpu_ar_em.py -q 2>&1 | grep -v "^$" | tail -8
B="--no-cpu-baseline --no-secondary --repeats 3 --N 1000 --T 2000 --r 20 --batch-per-gpu 256"
timeout 300 python bench.py $B --missing 0.1 --steps 2 --warmup 1 > $OUT/c4m.json 2> $OUT/c4m.err
timeout 300 python bench.py $B --steps 5 --warmup 2 > $OUT/c4.json 2> $OUT/c4.err
timeout 300 python bench.py $B --mode em --steps 3 --warmup 1 > $OUT/c4em.json 2> $OUT/c4em.err
timeout 300 python bench.py $B --mode em --missing 0.1 --steps 2 --warmup 1 > $OUT/c4emm.json 2> $OUT/c4emm.err
timeout 300 python scripts/debug/varp_time.py > $OUT/varp.txt 2>&1
for f in c4m c4 c4em c4emm; do
  python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "value=%.4g ms=%.4f" % (d["value"], d["ms_per_step"]), d["roofline"]["kernels_ms"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
tail -q -n 3 $OUT/*.err | grep -v amdgpu.ids | head
tail -12 $OUT/varp.txt
