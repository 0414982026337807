#!/bin/bash
# usage: make_exact_variant.sh NAME [-DFLAG=..]...   builds scratch/variants/libsw_NAME.so with sw_exact.hip recompiled
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=/tmp/xvariant_$NAME
rm -rf $W && mkdir -p $W
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -x hip $ROOT/dlsa_amd/csrc/sw_exact.hip -o $W/sw_exact.o \
  -I$ROOT/dlsa_amd/csrc -I$ROOT/include "$@" -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|LDS Size" | paste - - - - | grep walk | sort | uniq | head -8
OBJ=$ROOT/dlsa_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $ROOT/scratch/variants/libsw_$NAME.so \
  $W/sw_exact.o $OBJ/sw_score.hip.o $OBJ/sw_topk.hip.o $OBJ/sw_lib.cpp.o $OBJ/sw_db.cpp.o $OBJ/sw_node.cpp.o $OBJ/sw_package.cpp.o $OBJ/sw_worker.cpp.o
echo built libsw_$NAME.so
