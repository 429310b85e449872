#!/bin/bash
# Cross-compiles the kernel sweep tool for gfx950 (no GPU needed).
set -e
cd "$(dirname "$0")"
INC="-I../../include -I../../lightmotif_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off"
hipcc $FLAGS $INC kbench.hip -o kbench_slp &
hipcc $FLAGS -fno-slp-vectorize $INC kbench.hip -o kbench_noslp &
wait
ls -la kbench_slp kbench_noslp
