#!/bin/bash
TAG=${1:-r4k}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pca_synth.py tests/test_gpu_round4.py tests/test_gpu_em.py tests/test_gpu_api.py -q -m gpu --maxfail=12 2>&1 | tail -30 > $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 300 python bench.py --mode pca --steps 5 --warmup 1 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/bench_pca.json 2> $OUT/bench_pca.err
python - $OUT/bench_pca.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
print("pca ms %.4f value %.4g"%(d["ms_per_step"], d["value"]), r["kernels_ms"], r.get("gram",{}).get("frac"))
PY
