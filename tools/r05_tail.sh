#!/bin/bash
# round 5, launch-bound tail: suite, the short-order A/B, the kernel timeline of one fused threshold / argmax call, the fused
# calls under rocprofv3, the bench line
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05t
mkdir -p "$OUT"; cd "$ROOT"
make -C tests/cpp > "$OUT/make.log" 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q > "$OUT/gputest.log" 2>&1
grep -E "passed|failed|Error" "$OUT/gputest.log" | tail -5
timeout 600 python tools/short_order_ab.py --json "$OUT/short_order_ab.json" > "$OUT/short_order_ab.log" 2>&1; tail -10 "$OUT/short_order_ab.log"
bash tools/timeline_threshold.sh 1e-5 > "$OUT/timeline_fused.txt" 2>&1; grep -v rocprofv3 "$OUT/timeline_fused.txt" | tail -24
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fprof" -o fused -- \
    python "$ROOT/tools/fused_cmd.py" --reps 40 > "$OUT/fused_prof.log" 2>&1 )
cp $(find "$OUT/fprof" -name '*kernel_stats.csv' | head -1) "$OUT/fused_kernel_stats.csv" 2>/dev/null; rm -rf "$OUT/fprof"
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -c 200 "$OUT/bench_default.json"
