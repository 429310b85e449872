#!/bin/bash
# usage: tools/fused_ab.sh "<M list>" tag1 tag2 ...  -- fused threshold (exact) / fused argmax call times per library variant
MS=$1; shift
for rep in 1 2; do
for tag in "$@"; do
  if [ $tag = base ]; then unset LM_HIP_LIBRARY; else export LM_HIP_LIBRARY=$PWD/lightmotif_amd/csrc/liblightmotif_hip_$tag.so; fi
  python tools/msweep.py 1000000000 $MS 2>&1 >/dev/null | python -c "
import sys, json
out = []
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line); out.append('M=%d store %.3f thr %.3f argmax %.3f' % (r['M'], r['store']['ms'], r['fused_threshold_exact']['call_ms'], r['fused_argmax']['call_ms']))
print('$tag rep$rep ms:', ' | '.join(out))"
done; done
