#!/bin/bash
# interleaved same-box A/B of tools/api_overhead.py (fused calls at 1 Gbp x M = 20) between library variants
for rep in 1 2; do for tag in "$@"; do
  if [ $tag = base ]; then unset LM_HIP_LIBRARY; else export LM_HIP_LIBRARY=$PWD/lightmotif_amd/csrc/liblightmotif_hip_$tag.so; fi
  python tools/api_overhead.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag rep$rep', {k: d[k] for k in ('fused_argmax_ms', 'fused_threshold_p1e-05_prefilter1_ms', 'fused_threshold_p0.001_prefilter1_ms')})"
done; done
