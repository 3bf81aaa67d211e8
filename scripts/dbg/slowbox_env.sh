#!/bin/bash
# on a box of the slow kind: do the runtime's cache-scope switches change the cold-code behaviour?
cd $GRAFT_REPO_ROOT
echo "kernel: $(uname -r)"
if [ -z "$ANYBOX" ] && ! uname -r | grep -q "6.18.50"; then echo "fast box: nothing to do"; exit 0; fi
run() {
  echo "=== $*"
  env "$@" timeout 60 scripts/microbench/icache | sed -n '2,5p'
  env "$@" timeout 120 python bench.py --no-cpu-baseline --no-secondary --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('bench value=%.4g ms=%.4f'%(d['value'], d['ms_per_step']))"
}
run X=1
run AMD_OPT_FLUSH=0
run ROC_SYSTEM_SCOPE_SIGNAL=0
run AMD_OPT_FLUSH=0 ROC_SYSTEM_SCOPE_SIGNAL=0
run GPU_FLUSH_ON_EXECUTION=1
run HSA_ENABLE_SDMA=0
