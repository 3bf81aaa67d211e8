#!/bin/bash
# config-4 collapse: HBM read traffic of collapse_wide2 under the two workgroup->tile maps (diagnostic)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-c4pmc}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for x in 1 0; do
  mkdir -p $OUT/x$x
  DFM_WIDE_XCD=$x K=3 B=256 N=1000 T=2000 R=20 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/x$x/pmc_fetch -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/x$x/err
  python $R/scripts/pmc_summary.py $OUT/x$x 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if isinstance(v,dict) and 'fetch_bytes' in v: print($x, k, 'fetch MB/launch %.1f'%(v['fetch_bytes']/1e6), 'launches', v.get('launches'))
"
done
for i in 1 2 3; do timeout 200 python $R/scripts/dbg/c4_repeat.py 2>&1 | grep "call 0\|-2x"; done
