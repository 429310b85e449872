#!/bin/bash
# Collects the measurements committed under profiles/ (run on a GPU box via gpurun):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
# Everything lands in gpurun_out/<tag>/; copy what should be judged into profiles/.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
timeout 400 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
timeout 300 python tools/bench_configs.py > "$OUT/bench_configs.json" 2> "$OUT/bench_configs.err"
timeout 200 python tools/api_overhead.py > "$OUT/api_overhead.json" 2> "$OUT/api_overhead.err"
timeout 100 tools/kbench/stripe_bench > "$OUT/stripe_bench.txt" 2>&1
# the same box's ceiling for the store kernel: trivial kernels moving its 1 B read : 4 B written mix
timeout 100 tools/kbench/mix_bench > "$OUT/mix_bench.txt" 2>&1
timeout 200 python tools/handle_flow.py > "$OUT/handle_flow.json" 2> "$OUT/handle_flow.err"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo --single-device \
    --length 400000000 --no-cpu-baseline 2> "$OUT/bench_2rank.err" | grep '^{' > "$OUT/bench_2rank_gloo_single_device.json"  # (gloo prints a banner on stdout)
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv \
    -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --no-cpu-baseline > "$OUT/prof_bench.json" 2> "$OUT/prof.err" )
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for p in glob.glob(out + "/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(p)))
    with open(out + "/bench_kernel_stats.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
for p in glob.glob(out + "/prof/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(p)) if "lm::" in r["Kernel_Name"]]
    with open(out + "/bench_kernel_trace_lm.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
PY
rm -rf "$OUT/prof"
for pv in 1e-5 1e-3; do
  GRAFT_REPO_ROOT=$ROOT bash "$ROOT/tools/timeline_threshold.sh" $pv 2>/dev/null | grep -v simple_timer > "$OUT/timeline_fused_p$pv.txt"
done
cd "$ROOT"
tail -c 600 "$OUT/bench_default.json"; echo; head -5 "$OUT/bench_kernel_stats.csv" | cut -c1-200
