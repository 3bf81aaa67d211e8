#!/bin/bash
# what kind of box is this?  chain latencies, streaming ceilings, the headline line and the one-launch pass's in-kernel timeline
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -v "^=\|^$" | head -12
[ -x scripts/microbench/chainlat ] && timeout 60 scripts/microbench/chainlat 2>&1 | head -14
timeout 300 python bench.py --no-cpu-baseline --no-secondary --repeats 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('bench value=%.4g ms=%.4f whole=%.4f'%(d['value'],d['ms_per_step'],r['whole_step']['frac']), r['ceiling_measured'], d['device'])"
B=1024 DFM_PASS_NSW=4 timeout 120 python scripts/pf_prof.py 2>&1 | grep -v amdgpu | tail -42
B=1024 DFM_PASS_NSW=4 PF_ABL=512 timeout 120 python scripts/pf_prof.py 2>&1 | grep -v amdgpu | tail -5
