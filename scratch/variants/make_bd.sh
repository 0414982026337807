#!/bin/bash
# usage: make_bd.sh B D   -- the whole library with B query rows per hand-over block and a ring of D blocks per wave
set -e
B=$1; D=$2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=/tmp/variant_b${B}d${D}
rm -rf $W && mkdir -p $W
cp $ROOT/dlsa_amd/csrc/*.hip $ROOT/dlsa_amd/csrc/*.cpp $ROOT/dlsa_amd/csrc/*.h $ROOT/dlsa_amd/csrc/gen_score_asm.py $W/
sed -i "s/^def render(Dslots=3, B=12):/def render(Dslots=$D, B=$B):/" $W/gen_score_asm.py
sed -i 's#"../../include/sw_worker.h"#"'$ROOT'/include/sw_worker.h"#' $W/sw_host.h
(cd $W && python3 gen_score_asm.py > /dev/null)
objs=""
for src in sw_exact.hip sw_score.hip sw_topk.hip sw_lib.cpp sw_db.cpp sw_node.cpp sw_package.cpp sw_worker.cpp; do
  extra=""
  [ $src = sw_score.hip ] && extra="-mllvm -structurizecfg-skip-uniform-regions=true"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSW_SCORE_B=$B -I$ROOT/include -c -x hip $W/$src -o $W/$src.o $extra &
  objs="$objs $W/$src.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $ROOT/scratch/variants/libsw_b${B}d${D}.so $objs
echo built libsw_b${B}d${D}.so
