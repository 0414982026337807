#!/bin/bash
# builds gpurun_variants/libsw_<name>.so from a COPY of dlsa_amd/csrc with the row loop regenerated for other block
# sizes:  scratch/full_variant.sh name <B> <Bsmall> [-DX=1 ...]     (the flags go to every source)
set -e
name=$1; B=$2; BS=$3; shift 3
cd "$(dirname "$0")/.."
V=$PWD/gpurun_variants; S=$V/src_$name
rm -rf $S; mkdir -p $S/d/csrc $S/include
cp dlsa_amd/csrc/*.hip dlsa_amd/csrc/*.cpp dlsa_amd/csrc/*.h dlsa_amd/csrc/gen_score_asm.py $S/d/csrc/
cp include/*.h $S/include/
python3 - <<PY
import sys; sys.path.insert(0, "$S/d/csrc")
import gen_score_asm as g
text, n = g.render(B=$B, Bsmall=$BS)
open("$S/d/csrc/sw_score_asm.inc", "w").write(text)
PY
objs=""
for s in sw_exact.hip sw_lanes.hip sw_score.hip sw_topk.hip sw_lib.cpp sw_lanes.cpp sw_db.cpp sw_node.cpp sw_package.cpp sw_worker.cpp; do
  extra=""
  [ $s = sw_score.hip ] && extra="-mllvm -structurizecfg-skip-uniform-regions=true"
  (cd $S/d/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -x hip $s -o $S/$s.o $extra -DSW_SCORE_B=$B -DSW_SCORE_B_SMALL=$BS "$@") &
  objs="$objs $S/$s.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $V/libsw_$name.so $objs
rm -rf $S
echo built $V/libsw_$name.so
