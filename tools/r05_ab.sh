#!/bin/bash
# A/B of library variants (tools/build_variant.py) on the single-motif fused scans: bash tools/r05_ab.sh tag1 tag2 ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/ab
for rep in 1 2; do
  for tag in base "$@"; do
    if [ $tag = base ]; then unset LM_HIP_LIBRARY; else export LM_HIP_LIBRARY=$ROOT/lightmotif_amd/csrc/liblightmotif_hip_$tag.so; fi
    echo "== $tag rep$rep"
    timeout 150 python tools/fused_tsweep.py ${AB_MS:-20} 0 2> gpurun_out/ab/$tag.err || echo "failed/timeout"
  done
done
