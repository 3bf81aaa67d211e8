#!/bin/bash
# Round 4: (1) pass_fused variants in ONE process: round-3 library | kPfEcap 6 + scan_seq | 8 | 10 | 6 + scan_lds fallback,
# (2) recursion_tile with the LDL' pivot solve and VGPR-form MFMAs: parity, timing, spans.
TAG=${1:-r4d}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
V="LIB=gpurun_tmp/libdfmhip_r3.so LIB=gpurun_tmp/libdfmhip_e6.so LIB=gpurun_tmp/libdfmhip_e8.so LIB=gpurun_tmp/libdfmhip_e10.so LIB=gpurun_tmp/libdfmhip_e6lds.so"
timeout 300 python scripts/dbg/inproc_ab.py $V > $OUT/ab_headline.txt 2>&1
timeout 300 python scripts/dbg/inproc_ab.py $V batch=8192 > $OUT/ab_b8192.txt 2>&1
cat $OUT/ab_headline.txt $OUT/ab_b8192.txt | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_fuzz.py tests/test_gpu_em.py tests/test_gpu_round4.py tests/test_gpu_pass_fused.py tests/test_gpu_ks_pass.py -q -m gpu --maxfail=12 2>&1 | tail -40 > $OUT/pytest.log
tail -8 $OUT/pytest.log
DFM_LIB=gpurun_tmp/libdfmhip_tprof.so B=256 N=1000 T=2000 R=20 MISSING=0.1 K=2 timeout 300 python scripts/gpu_trace.py 2>&1 | grep TILEPROF > $OUT/tileprof.txt
cat $OUT/tileprof.txt
timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --steps 2 --warmup 1 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_c4m.json 2> $OUT/bench_c4m.err
timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --mode em --steps 2 --warmup 1 --repeats 3 --no-cpu-baseline --no-secondary > $OUT/bench_c4m_em.json 2> $OUT/bench_c4m_em.err
for f in $OUT/bench_c4m*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "value=%.4g ms=%.3f" % (d["value"], d["ms_per_step"]), d["roofline"].get("kernels_ms"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
