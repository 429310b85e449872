#!/bin/bash
# A/B of library variants on the prefilter-bound calls: the JASPAR batch (bench.py --config c3) and single-motif fused scans
mkdir -p gpurun_out/pf2
L=$PWD/lightmotif_amd/csrc
O=gpurun_out/pf2/ab.txt
for rep in 1 2; do
  for tag in base "$@"; do
    if [ $tag = base ]; then unset LM_HIP_LIBRARY; else export LM_HIP_LIBRARY=$L/liblightmotif_hip_$tag.so; fi
    echo "== $tag rep$rep" >> $O
    timeout 300 python bench.py --config c3 --steps 10 --warmup 3 --no-cpu-baseline 2>> gpurun_out/pf2/ab.err | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line); e = r.get('extras', {})
        print('c3 ms_per_step', r.get('ms_per_step'), {k: e[k] for k in e if 'ms' in k})" >> $O
    timeout 300 python tools/msweep.py 1000000000 8,10,12,15,20,28 2>> gpurun_out/pf2/ab.err > gpurun_out/pf2/ms.json
    python -c "
import json
d = json.load(open('gpurun_out/pf2/ms.json'))
for x in d['sweep']: print('M=%d thr %.4f argmax %.4f' % (x['M'], x['fused_threshold_prefilter']['call_ms'], x['fused_argmax']['call_ms']))" >> $O
  done
done
cat $O; tail -3 gpurun_out/pf2/ab.err
