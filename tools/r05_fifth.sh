#!/bin/bash
# round 5, fifth GPU call: whole-round stream lengths, realistic inputs, u8 store by stream length
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05e
mkdir -p "$OUT"
cd "$ROOT"
make -C tests/cpp > "$OUT/make.log" 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/gputest.log" 2>&1
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
timeout 900 python tools/realistic_inputs.py --json "$OUT/realistic_inputs.json" > "$OUT/realistic.log" 2>&1
timeout 300 python tools/u8_tsweep.py 20 0,50,74,98,146,242,482,962 > "$OUT/u8_tsweep.json" 2> "$OUT/u8_tsweep.err"
timeout 400 python tools/msweep.py 1000000000 8,12,15,20,24,28,36 > "$OUT/msweep.json" 2> "$OUT/msweep.err"
tail -3 "$OUT/gputest.log"; tail -3 "$OUT/bench_default.err"; python - <<P
import json
r = json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); e = r['extras']
print('value', r['value'], {k: r['roofline'].get(k) for k in ('frac', 'sclk_mhz_sustained', 'lds_frac_at_sustained_clock', 'valu_frac_at_sustained_clock')})
for k in ('fused_score_threshold', 'fused_score_argmax'):
    f = e[k]; print(k, f['ms'], {x: f['roofline'].get(x) for x in ('frac', 'sclk_mhz_sustained', 'lds_frac_at_sustained_clock', 'valu_frac_at_sustained_clock')})
c3 = e['configs']['c3']; print('c3 thr ms', c3['fused_threshold_ms'], c3['roofline']['frac'], 'argmax ms', c3['fused_argmax_ms'])
P
tail -60 "$OUT/realistic.log"; cat "$OUT/u8_tsweep.json"; tail -2 "$OUT/u8_tsweep.err"
python - <<P
import json
d = json.load(open('$OUT/msweep.json'))
for x in d['sweep']: print('M=%d thr %.4f argmax %.4f' % (x['M'], x['fused_threshold_prefilter']['call_ms'], x['fused_argmax']['call_ms']))
P
