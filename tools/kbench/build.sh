#!/bin/bash
# Cross-compiles the kernel sweep tool for gfx950 (no GPU needed).
set -e
cd "$(dirname "$0")"
INC="-I../../include -I../../lightmotif_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize"
hipcc $FLAGS -DLM_SCORE_NT_STORE=1 $INC kbench.hip -o kbench_nt &
hipcc $FLAGS -DLM_SCORE_NT_STORE=0 $INC kbench.hip -o kbench_plain &
wait
ls -la kbench_nt kbench_plain
