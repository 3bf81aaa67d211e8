#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4h; mkdir -p $OUT; export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --missing 0.1 --batch-per-gpu 8192 --mode $MODE --steps 3 --warmup 1 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/$tag.json 2> $OUT/$tag.err; echo $tag $(grep -o '"ms_per_step": [0-9.]*' $OUT/$tag.json | head -1); }
for MODE in pass em; do
run ${MODE}_nopipe DFM_PIPE=0
run ${MODE}_sub512 DFM_PIPE_SUB=512
run ${MODE}_sub1024 DFM_PIPE_SUB=1024
run ${MODE}_sub2048 DFM_PIPE_SUB=2048
run ${MODE}_sub4096 DFM_PIPE_SUB=4096
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $R/bench.py --missing 0.1 --batch-per-gpu 8192 --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/trace.err)
python - $OUT <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'collapse_miss' in r['Kernel_Name'] or 'recursion_chunk_kernel' in r['Kernel_Name']]
rows = rows[-32:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    print(r['Kernel_Name'][:40].ljust(40), r.get('Queue_Id'), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3)
PY
rm -rf $OUT/trace
