#!/bin/bash
# One-pass store kernels of 65 ... 128 rows (score_xlong_inst.hip) against the slices of <= 64 rows, interleaved on one box:
#   gpurun --timeout 900 -- 'bash tools/xlong_ab.sh'   ->  gpurun_out/xlong_ab.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
OUT=gpurun_out/xlong_ab.txt; : > $OUT
for rep in 1 2; do
  for x in 1 0; do
    LM_HIP_XLONG=$x timeout 300 python tools/msweep.py 1000000000 ${MS:-65,72,80,88,96,100,104,112,120,128} 2>/dev/null < /dev/null | python -c "
import sys, json
d = json.load(sys.stdin)
for e in d['sweep']:
    s, t = e['store'], e['fused_threshold_exact']
    print('rep $rep xlong=$x M=%3d store %-18s %.3f ms (min %.3f) lds_frac %.3f | chunked exact threshold %.3f ms' % (e['M'], s['kernel'], s['ms'], s['ms_min'], s['lds_frac'], t['call_ms']))
" >> $OUT
  done
done
cat $OUT
