#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "kernel: $(uname -r)"
timeout 900 python -m pytest tests/test_gpu_em.py tests/test_gpu_round3.py tests/test_gpu_fuzz.py tests/test_gpu_ar_em.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
DFM_MSTEP_MISS=2 timeout 900 python -m pytest tests/test_gpu_em.py tests/test_gpu_fuzz.py tests/test_gpu_round3.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
DFM_MSTEP_MISS=2 DFM_MM_KP=8 timeout 900 python -m pytest tests/test_gpu_em.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
echo "--- C4 missing EM: default (deep stages)"; python scripts/dbg/em_prof.py 256 2000 1000 20 0.1 2>&1 | grep "mstep\|sum"
echo "--- C4 missing EM: DFM_MM_KP=8"; DFM_MM_KP=8 python scripts/dbg/em_prof.py 256 2000 1000 20 0.1 2>&1 | grep "mstep\|sum"
echo "--- C2 missing EM: mstep_lam"; python scripts/dbg/em_prof.py 1024 500 200 8 0.1 2>&1 | grep "mstep\|sum"
echo "--- C2 missing EM: mstep_miss KP=32"; DFM_MSTEP_MISS=2 python scripts/dbg/em_prof.py 1024 500 200 8 0.1 2>&1 | grep "mstep\|sum"
echo "--- C2 missing EM: mstep_miss KP=8"; DFM_MSTEP_MISS=2 DFM_MM_KP=8 python scripts/dbg/em_prof.py 1024 500 200 8 0.1 2>&1 | grep "mstep\|sum"
