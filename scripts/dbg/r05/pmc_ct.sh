#!/bin/bash
# development: cache counters of the C_t kernel at config 4 with missing cells (two passes: memory-side bytes, L2 hits / misses)
R=$(pwd); OUT=$R/gpurun_out/pmc_ct; mkdir -p $OUT; export TMPDIR=/tmp
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
  tag=$(echo $set | tr ' ' '_')
  (cd /tmp && K=1 B=256 N=1000 T=2000 R=20 MISSING=0.1 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o p -- python $R/scripts/gpu_trace.py > /dev/null 2> $OUT/$tag.err)
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][:40]
    if "ct_miss" in k or "recursion_tile" in k or "collapse_wide2" in k:
        acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
for k, v in sorted(acc.items()): print(k, v)
PY
done
