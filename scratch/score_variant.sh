#!/bin/bash
# builds gpurun_variants/libsw_<name>.so: the product objects with sw_score.hip recompiled with extra flags
#   scratch/score_variant.sh name -DX=1 ...
set -e
name=$1; shift
cd "$(dirname "$0")/.."
V=gpurun_variants
mkdir -p $V
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -x hip dlsa_amd/csrc/sw_score.hip -mllvm -structurizecfg-skip-uniform-regions=true "$@" -o $V/sw_score_$name.o
objs=$(ls dlsa_amd/lib/obj/*.o | grep -v "sw_score.hip.o")
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $V/libsw_$name.so $objs $V/sw_score_$name.o
rm -f $V/sw_score_$name.o
echo built $V/libsw_$name.so
