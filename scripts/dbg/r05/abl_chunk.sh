#!/bin/bash
# development: one library per ablation of recursion_chunk_kernel (csrc/recursion_chunk.hip DFM_CK_ABL); run the missing-cell bench with each
set -e
cd "$(dirname "$0")/../../.."
L=dynamic_factor_models_amd/lib
mkdir -p $L/abl
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -DDFM_CK_ABL=$v -c dynamic_factor_models_amd/csrc/recursion_chunk.hip -o $L/abl/rc_$v.o
  objs=$(ls $L/*.o | grep -v recursion_chunk.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/libdfm_abl$v.so $objs $L/abl/rc_$v.o -ldl -lpthread
done
