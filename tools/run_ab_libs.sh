#!/bin/bash
# usage: tools/run_ab_libs.sh "<M list>" tag1 tag2 ...   (tags of tools/build_variant.py; "base" = the shipped library)
MS=$1; shift
for rep in 1 2; do
for tag in "$@"; do
  if [ $tag = base ]; then unset LM_HIP_LIBRARY; else export LM_HIP_LIBRARY=$PWD/lightmotif_amd/csrc/liblightmotif_hip_$tag.so; fi
  python tools/msweep.py 1000000000 $MS 2>&1 >/dev/null | python -c "
import sys, json
out = []
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line); out.append('M=%d %.4f' % (r['M'], r['store']['ms']))
print('$tag rep$rep store ms:', '  '.join(out))"
done; done
