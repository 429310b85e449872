#!/bin/bash
# rocprofv3 kernel statistics of one bench_configs.py configuration (GPU box only):
#   bash tools/prof_config.sh c3
CFG=${1:-c3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$CFG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $CFG -- python $GRAFT_REPO_ROOT/tools/bench_configs.py $CFG > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-400
python - <<PY
import csv, glob
p = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(p)))[:22]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(7), ("%.1f" % (float(r["TotalDurationNs"])/1e6)).rjust(9), "ms", ("%.1f" % (float(r["AverageNs"])/1e3)).rjust(9), "us", r["Percentage"].rjust(7))
PY
