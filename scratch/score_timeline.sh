#!/bin/bash
# Development build of the score kernel that accumulates, per wave, the time spent waiting for a ticket, setting a
# strip up, waiting for the left neighbour's block / for room in the right neighbour's ring / for another workgroup's
# boundary column (SW_SCORE_TIMELINE).  Build here: scratch/score_timeline.sh build ; GPU box: scratch/score_timeline.sh run <case>
set -e
cd "$(dirname "$0")/.."
V=gpurun_variants
if [ "$1" = build ]; then
  mkdir -p $V
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -x hip dlsa_amd/csrc/sw_score.hip -DSW_SCORE_TIMELINE -mllvm -structurizecfg-skip-uniform-regions=true $EXTRA -o $V/sw_score_tl.o
  objs=$(ls dlsa_amd/lib/obj/*.o | grep -v "sw_score.hip.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $V/libsw_score_tl.so $objs $V/sw_score_tl.o
  ls -la $V/libsw_score_tl.so
else
  shift
  SW_HIP_LIB=$PWD/$V/libsw_score_tl.so timeout -k 10 200 python3 scratch/score_timeline.py "$@"
fi
