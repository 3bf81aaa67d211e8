#!/bin/bash
# on a slow box only: in-process sweep of the one-launch pass's switches at the headline shape
cd $GRAFT_REPO_ROOT
echo "kernel: $(uname -r)"
if ! uname -r | grep -q "6.18.50"; then echo "fast box: nothing to do"; exit 0; fi
timeout 400 python scripts/dbg/inproc_ab.py 0 1024 2048 12288 16384 "DFM_PASS_NCOV=1" "DFM_PASS_NSW=5" "DFM_PASS_NSW=3" 2>&1 | grep -v amdgpu | tail -8 | cut -c1-100
