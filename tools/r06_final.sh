#!/bin/bash
# round 6, final collection on the shipped binary: profiles (tools/collect_profiles.sh), counters of the protein block scan and the
# store kernel, the fused calls by phase, the protein three-way A/B, Scanner::max, the minimal erratum reproducer, a fuzz soak
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
bash tools/collect_profiles.sh r06 > "$ROOT/gpurun_out/r06_collect.log" 2>&1
OUT=$ROOT/gpurun_out/r06
timeout 120 python tools/fused_phases.py > "$OUT/fused_phases.txt" 2>&1
timeout 300 python tools/protein_pair_ab.py 4 7 8 9 12 16 20 24 28 32 36 > "$OUT/protein_ab.txt" 2>&1
timeout 120 python tools/scanmax_time.py > "$OUT/scanmax.json" 2>/dev/null
timeout 120 python tools/scanmax_time.py 100000000 >> "$OUT/scanmax.json" 2>/dev/null
( cd tools/kbench && timeout 120 ./shift64_repro > "$OUT/shift64_repro.txt" 2>&1 )
( cd tests/cpp && timeout 300 ./test_dispatch > "$OUT/test_dispatch.txt" 2>&1 )
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/fprof" -o fused -- \
    python "$ROOT/tools/fused_cmd.py" --reps 40 > "$OUT/fused_prof.log" 2>&1 )
cp $(find "$OUT/fprof" -name '*kernel_stats.csv' | head -1) "$OUT/fused_kernel_stats.csv" 2>/dev/null; rm -rf "$OUT/fprof"
LM_FUZZ_FIRST=8240 LM_FUZZ_LAST=10240 LM_FUZZ_BATCH_FIRST=840 LM_FUZZ_BATCH_LAST=1040 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > "$OUT/fuzz_soak.log" 2>&1
tail -2 "$OUT/fuzz_soak.log"; tail -c 400 "$OUT/bench_default.json"; echo; head -3 "$OUT/fused_kernel_stats.csv" | cut -c1-160; cat "$OUT/pmc_summary.json" | head -5
