#!/bin/bash
# development: libraries with one source file compiled under an ablation macro.  usage: abl_lib.sh <file.hip> <MACRO> v1 v2 ...
set -e
cd "$(dirname "$0")/../../.."
L=dynamic_factor_models_amd/lib
f=$1; macro=$2; shift 2
mkdir -p $L/abl
base=$(basename $f .hip)
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $ABL_FLAGS -D$macro=$v -c dynamic_factor_models_amd/csrc/$f -o $L/abl/${base}_$v.o
  objs=$(ls $L/*.o | grep -vF "/$base.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/libdfm_${base}_$v.so $objs $L/abl/${base}_$v.o -ldl -lpthread
done
