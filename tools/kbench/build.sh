#!/bin/bash
# Cross-compiles the kernel sweep tool for gfx950 (no GPU needed).
#   kbench_nt <L> <K> <reps> <tag>            full sweep (prefetch x T), HBM calibration kernels
#   kbench_nt <L> <K> <rounds> <tag> ab       interleaved A/B of block size / XCD remap / register cap
set -e
cd "$(dirname "$0")"
INC="-I../../include -I../../lightmotif_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize"
hipcc $FLAGS -DLM_SCORE_NT_STORE=1 $INC kbench.hip -o kbench_nt &
if [ "$1" = "all" ]; then
  hipcc $FLAGS -DLM_SCORE_NT_STORE=0 $INC kbench.hip -o kbench_plain &
fi
for t in mix_bench stripe_bench valu_bench graph_gap_bench shift64_repro; do   # stand-alone micro-benchmarks (HBM mix, stripe kernel, VALU rates, graph replay, the shift erratum)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $INC $t.hip -o $t &
done
wait
ls kbench_* mix_bench stripe_bench valu_bench
