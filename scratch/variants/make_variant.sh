#!/bin/bash
# usage: make_variant.sh NAME  (env SW_GEN_* flags are read by a patched copy of the generator)
# Builds scratch/variants/libsw_NAME.so: the in-tree library with sw_score.hip compiled against a variant row loop.
set -e
NAME=$1
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=/tmp/variant_$NAME
rm -rf $W && mkdir -p $W
cp $ROOT/dlsa_amd/csrc/*.hip $ROOT/dlsa_amd/csrc/*.h $ROOT/dlsa_amd/csrc/*.inc $W/
cp ${GEN:-$ROOT/dlsa_amd/csrc/gen_score_asm.py} $W/gen_score_asm.py
(cd $W && python3 gen_score_asm.py >/dev/null)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -x hip $W/sw_score.hip -o $W/sw_score.o \
  -I$ROOT/dlsa_amd/csrc -I$ROOT/include $EXTRA -mllvm -structurizecfg-skip-uniform-regions=true -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "SGPRs Spill|VGPRs Spill|VGPRs:|ScratchSize" | sort | uniq -c | head -8
OBJ=$ROOT/dlsa_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $ROOT/scratch/variants/libsw_$NAME.so \
  $OBJ/sw_exact.hip.o $W/sw_score.o $OBJ/sw_topk.hip.o $OBJ/sw_lib.cpp.o $OBJ/sw_db.cpp.o $OBJ/sw_node.cpp.o $OBJ/sw_package.cpp.o $OBJ/sw_worker.cpp.o
echo built libsw_$NAME.so
