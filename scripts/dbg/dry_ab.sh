#!/bin/bash
# the scan rehearsal (DFM_SCAN_ABL bit 15 = off) A/B in one process, + parity of the one-launch pass with it on
cd $GRAFT_REPO_ROOT
echo "kernel: $(uname -r)"
if [ -n "$ONLYSLOW" ] && ! uname -r | grep -q "6.18.50"; then echo "fast box: nothing to do"; exit 0; fi
timeout 300 python scripts/dbg/inproc_ab.py 32768 0 2>&1 | grep -v amdgpu | tail -2 | cut -c1-110
timeout 300 python scripts/dbg/inproc_ab.py 32768 0 batch=8192 2>&1 | grep -v amdgpu | tail -2 | cut -c1-110
[ -z "$NOTEST" ] && timeout 600 python -m pytest tests/test_gpu_pass_fused.py tests/test_gpu_round3.py tests/test_gpu_ks_pass.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
B=1024 DFM_PASS_NSW=4 timeout 120 python scripts/pf_prof.py 2>&1 | grep -v amdgpu | grep '^round\|first cov\|last scan end\|cov round'
