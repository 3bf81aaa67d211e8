#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "kernel: $(uname -r)"
timeout 900 python -m pytest tests/test_gpu_em.py tests/test_gpu_round3.py tests/test_gpu_fuzz.py tests/test_gpu_ar_em.py -x -q -m gpu 2>&1 | tail -4
DFM_MSTEP_MISS=2 timeout 900 python -m pytest tests/test_gpu_em.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o e -- python $GRAFT_REPO_ROOT/scripts/dbg/em_prof.py 256 2000 1000 20 0.1 > /dev/null 2>&1; f=$(find /tmp/st -name "*kernel_stats.csv" | head -1); python - $f <<PY
import csv,sys
for row in csv.DictReader(open(sys.argv[1])):
    if float(row["AverageNs"]) > 5e4: print(row["Name"][:60].ljust(60), row["Calls"], "avg us", round(float(row["AverageNs"])/1e3, 1))
PY
