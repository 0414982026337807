#!/bin/bash
# Builds scratch/asan/libsw_hip.so (host code instrumented with AddressSanitizer, or ThreadSanitizer with SAN=thread;
# device code untouched) and the
# stress program against it.  Run scratch/asan/stress on a GPU box.
set -e
SAN=${SAN:-address}   # address (thread builds, but the HIP runtime finds no device under ThreadSanitizer's memory layout)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/scratch/asan
mkdir -p $OUT/obj
CS=$ROOT/dlsa_amd/csrc
for src in sw_exact.hip sw_score.hip sw_topk.hip sw_lib.cpp sw_db.cpp sw_node.cpp sw_package.cpp sw_worker.cpp; do
  extra=""
  [ $src = sw_score.hip ] && extra="-mllvm -structurizecfg-skip-uniform-regions=true"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -g -std=c++17 -fPIC -Xarch_host -fsanitize=$SAN -Xarch_host -fno-omit-frame-pointer \
    -c -x hip $CS/$src -o $OUT/obj/$src.o $extra
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -fno-gpu-sanitize -fsanitize=$SAN -shared-libsan -o $OUT/libsw_hip.so $OUT/obj/*.o
/opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=$SAN -shared-libsan -fno-omit-frame-pointer -I$ROOT/include \
  -o $OUT/stress $OUT/stress.cpp -L$OUT -lsw_hip -Wl,-rpath,$OUT -Wl,-rpath,/opt/rocm/lib/llvm/lib/clang/22/lib/linux -pthread
echo built
