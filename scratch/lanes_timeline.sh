#!/bin/bash
# Development build of the lane fill kernel that records when every strip ran and how long it waited for its left
# neighbour (SW_LANES_TIMELINE).  Build here: scratch/lanes_timeline.sh build ; on the GPU box: scratch/lanes_timeline.sh run
set -e
cd "$(dirname "$0")/.."
V=gpurun_variants
if [ "$1" = build ]; then
  mkdir -p $V
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -x hip dlsa_amd/csrc/sw_lanes.hip -DSW_LANES_TIMELINE $EXTRA -o $V/sw_lanes_tl.o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -x hip dlsa_amd/csrc/sw_lanes.cpp -DSW_LANES_TIMELINE -o $V/sw_lanes_cpp_tl.o
  objs=$(ls dlsa_amd/lib/obj/*.o | grep -v "sw_lanes.hip.o\|sw_lanes.cpp.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $V/libsw_timeline.so $objs $V/sw_lanes_tl.o $V/sw_lanes_cpp_tl.o
  ls -la $V/libsw_timeline.so
else
  rm -f gpurun_out/lanes_timeline.bin
  SW_LANES_TIMELINE_FILE=$PWD/gpurun_out/lanes_timeline.bin SW_HIP_LIB=$PWD/$V/libsw_timeline.so timeout -k 10 120 python3 scratch/lanes_ab.py 1000 1000 8000 2>&1 | grep '"lanes": "1"' | cut -c1-170
  python3 scratch/lanes_timeline.py gpurun_out/lanes_timeline.bin
fi
