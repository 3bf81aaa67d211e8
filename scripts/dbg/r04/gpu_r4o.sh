#!/bin/bash
# cache-policy modifiers on the streaming LDS-DMA loads: headline in one process, config 4 per library
TAG=${1:-r4o}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
uname -r > $OUT/device.txt
V="0 LIB=gpurun_tmp/libdfmhip_nt.so LIB=gpurun_tmp/libdfmhip_sc1.so LIB=gpurun_tmp/libdfmhip_ntsc1.so LIB=gpurun_tmp/libdfmhip_sc0.so"
timeout 300 python scripts/dbg/inproc_ab.py $V 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" > $OUT/ab_dma_mod_B1024.txt
timeout 300 python scripts/dbg/inproc_ab.py $V batch=8192 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" > $OUT/ab_dma_mod_B8192.txt
cat $OUT/ab_dma_mod_B1024.txt $OUT/ab_dma_mod_B8192.txt
for n in default nt sc1 ntsc1 sc0; do
  L=""; [ $n != default ] && L=$R/gpurun_tmp/libdfmhip_$n.so
  DFM_LIB=$L timeout 300 python bench.py --N 1000 --T 2000 --r 20 --batch-per-gpu 256 --steps 5 --warmup 2 --repeats 5 --no-cpu-baseline --no-secondary > $OUT/c4_$n.json 2> $OUT/c4_$n.err
  python - $OUT/c4_$n.json $n <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1]); r=d["roofline"]
    print("c4", sys.argv[2], "ms %.4f"%d["ms_per_step"], r["kernels_ms"])
except Exception as e: print("unreadable", e)
PY
done
