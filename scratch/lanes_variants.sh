#!/bin/bash
# Timing-only variants of the lane fill kernel (results may be wrong): where does its time go?
# Build here (CPU container): scratch/lanes_variants.sh build ; run on the GPU box: scratch/lanes_variants.sh run
set -e
cd "$(dirname "$0")/.."
V=gpurun_variants
if [ "$1" = build ]; then
  mkdir -p $V
  for v in NO_WAIT "X -DSW_LANES_SLEEP=4" "X -DSW_LANES_SLEEP=8" "X -DSW_LANES_SLEEP=32" "X -DSW_LANES_SLEEP=64" "X -DSW_LANES_RB=8"; do
    name=$(echo $v | tr -d ' ' | tr -c 'A-Z0-9_\n' '_')
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -x hip dlsa_amd/csrc/sw_lanes.hip -DSW_LANES_TIMING_$v -o $V/sw_lanes_$name.o
    objs=$(ls dlsa_amd/lib/obj/*.o | grep -v sw_lanes.hip.o)
    hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $V/libsw_$name.so $objs $V/sw_lanes_$name.o
  done
  ls -la $V
else
  for lib in "" $V/libsw_*.so; do
    echo "== ${lib:-product library}"
    SW_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 120 python3 scratch/lanes_ab.py 1000 1000 8000 2>&1 | grep '"lanes": "1"' | cut -c1-170
    SW_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 120 python3 scratch/lanes_ab.py 150 150 100000 100 2>&1 | grep '"lanes": "1"' | cut -c1-170
  done
fi
