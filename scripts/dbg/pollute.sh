#!/bin/bash
# does a cold instruction cache at launch reproduce the slow boxes' timeline on a fast one?
cd $GRAFT_REPO_ROOT
echo "kernel: $(uname -r)"
B=1024 DFM_PASS_NSW=4 timeout 120 python scripts/pf_prof.py 2>&1 | grep -v amdgpu | grep "^round\|first cov\|last scan end\|span\|cov: " 
echo "--- polluted"
B=1024 DFM_PASS_NSW=4 PF_POLLUTE=1 timeout 120 python scripts/pf_prof.py 2>&1 | grep -v amdgpu | grep "^round\|first cov\|last scan end\|span\|cov: "
