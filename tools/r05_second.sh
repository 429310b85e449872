#!/bin/bash
# round 5, second GPU call: the Dispatch::Hip twin, the full GPU suite on the LUT-decode binary, crossover, bench, clocks
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05b
mkdir -p "$OUT"
cd "$ROOT"
make -C tests/cpp > "$OUT/make.log" 2>&1
timeout 600 tests/cpp/test_dispatch > "$OUT/test_dispatch.log" 2>&1; echo "test_dispatch rc=$?" >> "$OUT/test_dispatch.log"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/gputest.log" 2>&1
timeout 400 python tools/crossover.py --json "$OUT/crossover.json" > "$OUT/crossover.log" 2>&1
timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
timeout 300 python tools/clock_probe.py --json "$OUT/clock_probe.json" > "$OUT/clock_probe.log" 2>&1
tail -5 "$OUT/test_dispatch.log"; tail -5 "$OUT/gputest.log"; tail -30 "$OUT/crossover.log"; cat "$OUT/bench_default.json"
