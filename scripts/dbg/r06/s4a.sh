#!/bin/bash
# session 4, call a: tile_gsums_ahead under rocprof (config 4, 10 % missing, EM), its tests, the two-stream overlap probe
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/s4a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_tile_chunk.py tests/test_gpu_ar_em.py -q -m gpu -x 2>&1 | tail -5 > $OUT/pt.log
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --missing 0.1 --mode em --steps 2 --warmup 1 --repeats 2 --no-cpu-baseline --no-secondary > $OUT/c4m_em.json 2> $OUT/c4m_em.err)
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_c4_em_missing10.csv 2>/dev/null; rm -rf $OUT/stats
for b in 1024 4096; do
  B=$b K=10 timeout 300 python scripts/dbg/r06/overlap_probe.py >> $OUT/overlap.txt 2>&1
  B=$b K=10 EM=1 timeout 300 python scripts/dbg/r06/overlap_probe.py >> $OUT/overlap.txt 2>&1
done
B=1024 K=10 NH=4 timeout 300 python scripts/dbg/r06/overlap_probe.py >> $OUT/overlap.txt 2>&1
cat $OUT/pt.log; head -8 $OUT/kernel_stats_c4_em_missing10.csv | cut -c1-150; cat $OUT/overlap.txt
