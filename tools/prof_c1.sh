#!/bin/bash
# kernel durations behind configs[0]'s latency: tools/c1_latency.py (few repetitions) under rocprofv3 --kernel-trace
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r03d}; mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && LM_C1_REPS=${2:-100} timeout 150 rocprofv3 --kernel-trace --stats --output-format csv \
    -d "$OUT/prof_c1" -o c1 -- python "$ROOT/tools/c1_latency.py" > "$OUT/c1_under_rocprof.json" 2> "$OUT/c1_prof.err" < /dev/null )
f=$(find "$OUT/prof_c1" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/c1_kernel_stats.csv"; head -14 "$f" | cut -c1-200; else echo "no kernel_stats.csv"; ls -R "$OUT/prof_c1" | head; fi
