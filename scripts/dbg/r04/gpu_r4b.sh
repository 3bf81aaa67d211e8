#!/bin/bash
# Round 4: recursion_tile_kernel -- parity of every Rp = 32 path with missing cells, then A/B timing against recursion_wave_kernel<32>.
TAG=${1:-r4b}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_round4.py tests/test_gpu_em.py tests/test_gpu_mstep_miss.py tests/test_gpu_ks_pass.py -q -m gpu --maxfail=12 -x 2>&1 | tail -60 > $OUT/pytest_tile.log
tail -30 $OUT/pytest_tile.log
timeout 600 python -m pytest tests/test_gpu_round3.py -q -m gpu -k config4 2>&1 | tail -20 > $OUT/pytest_c4.log
tail -12 $OUT/pytest_c4.log
for nt in 0 1; do
  DFM_NO_TILE=$nt timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --steps 2 --warmup 1 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_c4m_notile$nt.json 2> $OUT/bench_c4m_notile$nt.err
  DFM_NO_TILE=$nt timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --mode em --steps 2 --warmup 1 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_c4m_em_notile$nt.json 2> $OUT/bench_c4m_em_notile$nt.err
done
for f in $OUT/bench_c4m*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "value=%.4g ms=%.3f" % (d["value"], d["ms_per_step"]), d["roofline"].get("kernels_ms"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
tail -3 $OUT/*.err | head -40
