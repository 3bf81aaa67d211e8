#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "kernel: $(uname -r)"
if ! uname -r | grep -q "6.18.50"; then echo "fast box: nothing to do"; exit 0; fi
uptime; cat /proc/uptime
timeout 60 scripts/microbench/icache | sed -n '1,6p'
timeout 300 python scripts/dbg/inproc_ab.py 32768 0 2>&1 | grep -v amdgpu | tail -2 | cut -c1-110
