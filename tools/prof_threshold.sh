#!/bin/bash
# rocprofv3 kernel trace of the fused threshold at several hit rates (GPU box only)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_thr
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o thr -- python $GRAFT_REPO_ROOT/tools/api_overhead.py 1000000000 20 > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
for p in f:
    rows = list(csv.DictReader(open(p)))
    for r in rows[:25]:
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), r["AverageNs"].rjust(12), r["Percentage"].rjust(8))
PY
