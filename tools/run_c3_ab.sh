#!/bin/bash
# interleaved same-box A/B of bench.py --config c3 between the shipped library and variants (tools/build_variant.py)
for rep in ${REPS:-1 2 3}; do
for tag in "$@"; do
  if [ $tag = base ]; then unset LM_HIP_LIBRARY; else export LM_HIP_LIBRARY=$PWD/lightmotif_amd/csrc/liblightmotif_hip_$tag.so; fi
  python bench.py --config c3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag rep$rep', d['ms_per_step'], 'ms per threshold batch')"
done; done
