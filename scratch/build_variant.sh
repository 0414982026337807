#!/bin/bash
# builds a library variant with extra -D flags into scratch/variants/<name>.so :  build_variant.sh name -DX=1 ...
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p scratch/variants/obj_$name
objs=""
for s in sw_exact.hip sw_score.hip sw_topk.hip sw_lib.cpp sw_worker.cpp; do
  extra=""
  [ $s = sw_score.hip ] && extra="-mllvm -structurizecfg-skip-uniform-regions=true"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -x hip dlsa_amd/csrc/$s -o scratch/variants/obj_$name/$s.o $extra "$@" &
  objs="$objs scratch/variants/obj_$name/$s.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o scratch/variants/lib_$name.so $objs
rm -rf scratch/variants/obj_$name
echo built scratch/variants/lib_$name.so
