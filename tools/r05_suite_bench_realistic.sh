#!/bin/bash
# round 5 GPU call: suite with the realistic-input / 8-rank tests, bench line with the per-XCD clock marks, realistic inputs
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05m
mkdir -p "$OUT"
cd "$ROOT"
make -C tests/cpp > "$OUT/make.log" 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q > "$OUT/gputest.log" 2>&1
timeout 700 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
timeout 900 python tools/realistic_inputs.py --json "$OUT/realistic_inputs.json" > "$OUT/realistic.log" 2>&1
grep -E "passed|failed|Error" "$OUT/gputest.log" | tail -5; tail -3 "$OUT/bench_default.err"; python - <<P
import json
r = json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); e = r['extras']
print('value', r['value'], {k: r['roofline'].get(k) for k in ('frac', 'sclk_mhz_sustained', 'lds_frac_at_sustained_clock', 'valu_frac_at_sustained_clock')})
for k in ('fused_score_threshold', 'fused_score_argmax'):
    f = e[k]; print(k, f['ms'], {x: f['roofline'].get(x) for x in ('frac', 'sclk_mhz_sustained', 'lds_frac_at_sustained_clock', 'valu_frac_at_sustained_clock')})
c3 = e['configs']['c3']; print('c3 thr ms', c3['fused_threshold_ms'], c3['roofline']['frac'], 'argmax ms', c3['fused_argmax_ms'], c3.get('realistic'))
print(e['configs']['c1']['C1_generic_bench_geometry'])
d = json.load(open('$OUT/realistic_inputs.json')); print(d['realistic_over_uniform']); print({k: v for k, v in d['inputs']['realistic'].items() if not isinstance(v, (list, dict))}); print({k: v for k, v in d['inputs']['uniform'].items() if not isinstance(v, (list, dict))})
P
