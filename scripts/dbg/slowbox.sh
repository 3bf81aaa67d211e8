#!/bin/bash
# What kind of box is this?  About one box in ten of this pool runs host kernel 6.18.50 (the others 6.18.51) and gives the
# headline 0.43 instead of 0.55-0.58 of the HBM peak with identical clocks, HBM ceilings and idle chain latencies.  On those
# boxes the kernel's code is not in L2 (nor the instruction caches) at the start of a launch and a cold line costs ~0.2 us
# beside the stream: the first Gram step of every workgroup takes 20-26 us instead of 5, the first covariance chain 76-92
# instead of 50, the first scan 55 instead of 25 (profiles/r03/slow_boxes/).
cd $GRAFT_REPO_ROOT
echo "kernel: $(uname -r)"
timeout 60 scripts/microbench/icache
timeout 120 python bench.py --no-cpu-baseline --no-secondary --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('bench value=%.4g frac=%.3f'%(d['value'], d['roofline']['whole_step']['frac']))"
B=1024 DFM_PASS_NSW=4 timeout 120 python scripts/pf_prof.py 2>&1 | grep -v amdgpu | grep '^round\|first cov\|last scan end\|span\|cov round\|scan: compute'
