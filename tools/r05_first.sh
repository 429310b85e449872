#!/bin/bash
# round 5, first GPU call: clocks, crossover of the host-pointer path, counters of the fused scans and the headline kernel
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05a
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python tools/clock_probe.py --json "$OUT/clock_probe.json" > "$OUT/clock_probe.log" 2>&1
timeout 400 python tools/crossover.py --json "$OUT/crossover.json" > "$OUT/crossover.log" 2>&1
GRAFT_REPO_ROOT=$ROOT bash tools/collect_stalls.sh r05_fused "python $ROOT/tools/fused_cmd.py --reps 6" > "$OUT/stalls_fused.log" 2>&1
GRAFT_REPO_ROOT=$ROOT bash tools/collect_stalls.sh r05_c3 "python $ROOT/tools/fused_cmd.py --c3 --reps 2" > "$OUT/stalls_c3.log" 2>&1
GRAFT_REPO_ROOT=$ROOT bash tools/collect_stalls.sh r05_store "python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras --preheat-ms 0" > "$OUT/stalls_store.log" 2>&1
tail -40 "$OUT/clock_probe.log"; tail -60 "$OUT/crossover.log" | head -80
