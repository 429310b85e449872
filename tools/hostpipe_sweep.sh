#!/bin/bash
# Shape of the host-pointer tile pipeline (csrc/hostptr.hip), swept on one box with the -DLM_HIP_DEV_SWITCHES build:
#   python tools/build_variant.py dev -DLM_HIP_DEV_SWITCHES && bash tools/hostpipe_sweep.sh > profiles/r04_hostpipe_sweep.txt
# Each line: tile MB / pinned ring slots / copier threads -> 1 Gbp x M = 20 through lm_hip_score_f32, with the
# pipeline's own account of where its threads waited (LM_HIP_PIPE_TRACE).
cd "$(dirname "$0")/.."
export LM_HIP_LIBRARY=$PWD/lightmotif_amd/csrc/liblightmotif_hip_dev.so LM_HIP_PIPE_TRACE=1
for cfg in "32 4 4" "32 4 2" "32 4 8" "32 6 4" "32 8 8" "16 4 4" "16 8 4" "8 8 4" "64 4 4" "64 4 8"; do
  set -- $cfg
  echo "== tile $1 MB, $2 ring slots, $3 copiers"
  LM_HIP_PIPE_TILE_MB=$1 LM_HIP_PIPE_OUT_SLOTS=$2 LM_HIP_PIPE_COPIERS=$3 \
    python tools/host_pointer_bench.py --skip c1,block,threads --big ${BIG:-1000000000} 2>&1 | grep -v "^$" | tail -6
done
