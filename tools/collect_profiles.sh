#!/bin/bash
# Collects the measurements committed under profiles/ (run on a GPU box via gpurun):
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r02'
# Everything lands in gpurun_out/<tag>/; copy what should be judged into profiles/.
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
timeout 400 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
timeout 300 python bench.py --config c3 > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
timeout 400 python tools/bench_configs.py > "$OUT/bench_configs.json" 2> "$OUT/bench_configs.err"
timeout 200 python tools/api_overhead.py > "$OUT/api_overhead.json" 2> "$OUT/api_overhead.err"
timeout 400 python tools/msweep.py > "$OUT/msweep.json" 2> "$OUT/msweep.err"
timeout 300 python tools/msweep.py 1000000000 40,48,64,100,150 > "$OUT/msweep_long.json" 2> "$OUT/msweep_long.err"
timeout 100 python tools/c1_latency.py > "$OUT/c1_latency.json" 2> "$OUT/c1_latency.err"
timeout 200 python tools/ingest_bench.py > "$OUT/ingest.json" 2> "$OUT/ingest.err"
timeout 100 tools/kbench/mix_bench > "$OUT/mix_bench.txt" 2>&1
timeout 200 python tools/track_ab.py 20 > "$OUT/track_ab.txt" 2>/dev/null
timeout 200 python tools/track_ab.py 12 >> "$OUT/track_ab.txt" 2>/dev/null
timeout 200 python tools/handle_flow.py > "$OUT/handle_flow.json" 2> "$OUT/handle_flow.err"
# the literal drop-in path: host-pointer entry points on pageable host matrices (dna.rs loop, Scanner block, threads, 1 Gbp)
timeout 300 python tools/host_pointer_bench.py --json "$OUT/host_pointer.json" > "$OUT/host_pointer.log" 2>&1
# the N > 1 control flow of bench.py on the box's one GPU: 2 ranks over gloo (torch merge), and the C-ABI merge path
# through a communicator of one rank -- functional checks, NOT scaling numbers
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo --single-device \
    --length 400000000 --no-cpu-baseline 2> "$OUT/bench_2rank.err" | grep '^{' > "$OUT/bench_2rank_gloo_single_device.json"
timeout 200 python bench.py --merge cabi --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/bench_cabi_merge_one_rank.json" 2> "$OUT/bench_cabi.err"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29518 bench.py --config c3 --gpus 2 --steps 4 --warmup 1 --dist-backend gloo --single-device \
    --no-cpu-baseline 2> "$OUT/bench_c3_2rank.err" | grep '^{' > "$OUT/bench_c3_2rank_gloo_single_device.json"
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv \
    -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --no-cpu-baseline --no-extras > "$OUT/prof_bench.json" 2> "$OUT/prof.err" )
# (--no-extras: the end-to-end extra launches the SAME store kernel on 262 144-row tiles through lm_hip_score_f32, which
#  would be averaged into the headline kernel's row of the statistics)
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
GRAFT_REPO_ROOT=$ROOT bash "$ROOT/tools/collect_pmc.sh" > "$OUT/pmc.log" 2>&1
cp "$ROOT/gpurun_out/pmc/summary.json" "$OUT/pmc_summary.json" 2>/dev/null
cp "$ROOT/gpurun_out/pmc/pmc_traffic.json" "$OUT/pmc_traffic.json" 2>/dev/null
cd "$ROOT"
tail -c 700 "$OUT/bench_default.json"; echo; head -4 "$OUT/bench_kernel_stats.csv" | cut -c1-200; cat "$OUT/pmc_summary.json"
