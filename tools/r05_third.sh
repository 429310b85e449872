#!/bin/bash
# round 5, third GPU call: the hand-pipelined pair scans -- GPU suite, bench line, M-sweep, counters of the fused scans
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05c
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/gputest.log" 2>&1
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
timeout 600 python tools/msweep.py 1000000000 8,10,12,15,20,24,28,33,36 > "$OUT/msweep.json" 2> "$OUT/msweep.err"
GRAFT_REPO_ROOT=$ROOT timeout 900 bash tools/collect_stalls.sh r05c_fused "python $ROOT/tools/fused_cmd.py --reps 6" > "$OUT/stalls_fused.log" 2>&1
GRAFT_REPO_ROOT=$ROOT timeout 900 bash tools/collect_stalls.sh r05c_c3 "python $ROOT/tools/fused_cmd.py --c3 --reps 2" > "$OUT/stalls_c3.log" 2>&1
tail -5 "$OUT/gputest.log"; cat "$OUT/bench_default.json" | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); e = r['extras']
print('value', r['value'], 'fused thr', e['fused_score_threshold']['ms'], e['fused_score_threshold']['roofline']['frac'], 'argmax', e['fused_score_argmax']['ms'], e['fused_score_argmax']['roofline']['frac'])
c3 = e['configs']['c3']; print('c3 thr ms', c3['fused_threshold_ms'], c3['roofline']['frac'], 'argmax ms', c3['fused_argmax_ms'])"
python - <<P
import json
d = json.load(open('$OUT/msweep.json'))
for x in d['sweep']: print('M=%d thr %.4f argmax %.4f' % (x['M'], x['fused_threshold_prefilter']['call_ms'], x['fused_argmax']['call_ms']))
P
