cd tools/kbench
./shift64_repro
./shift64_repro 16777216 5
