#!/bin/bash
# Builds the library of a previous commit next to the current one for same-box A/B runs:
#   tools/build_prev.sh [rev]   ->  lightmotif_amd/csrc/liblightmotif_hip_prev.so
# (select it at run time with LM_HIP_LIBRARY=lightmotif_amd/csrc/liblightmotif_hip_prev.so)
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$ROOT/gpurun_out/prev_src
rm -rf $W; mkdir -p $W/include $W/csrc
git -C $ROOT archive $REV include lightmotif_amd/csrc | tar -x -C $W
SRC=$W/lightmotif_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fno-fast-math -I$W/include -I$SRC"
OBJS=""
for u in score reduce hits discrete layout api; do
  hipcc $FLAGS -c $SRC/$u.hip -o $W/$u.o & OBJS="$OBJS $W/$u.o"
done
for i in 0 1 2 3 4 5 6 7 8; do
  hipcc $FLAGS -DLM_M_LO=$((4*i+1)) -DLM_M_HI=$((4*i+4)) -DLM_INST_ID=$i -c $SRC/score_inst.hip -o $W/inst_$i.o & OBJS="$OBJS $W/inst_$i.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/lightmotif_amd/csrc/liblightmotif_hip_prev.so $OBJS
echo built $ROOT/lightmotif_amd/csrc/liblightmotif_hip_prev.so from $REV
