#!/bin/bash
# round 5, final collection on the shipped binary: profiles (tools/collect_profiles.sh), the fused calls under rocprofv3,
# the long fuzz soak
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
bash tools/collect_profiles.sh r05 > "$ROOT/gpurun_out/r05_collect.log" 2>&1
OUT=$ROOT/gpurun_out/r05
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fprof" -o fused -- \
    python "$ROOT/tools/fused_cmd.py" --reps 40 > "$OUT/fused_prof.log" 2>&1 )
cp $(find "$OUT/fprof" -name '*kernel_stats.csv' | head -1) "$OUT/fused_kernel_stats.csv" 2>/dev/null; rm -rf "$OUT/fprof"
LM_FUZZ_FIRST=240 LM_FUZZ_LAST=4240 LM_FUZZ_BATCH_FIRST=40 LM_FUZZ_BATCH_LAST=440 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > "$OUT/fuzz_soak.log" 2>&1
tail -2 "$OUT/fuzz_soak.log"; tail -c 600 "$OUT/bench_default.json"; echo; head -3 "$OUT/fused_kernel_stats.csv" | cut -c1-160; cat "$OUT/pmc_summary.json" | head -30
