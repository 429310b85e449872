cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sm -o sm -- python $GRAFT_REPO_ROOT/tools/scanmax_time.py > $GRAFT_REPO_ROOT/gpurun_out/prof_sm.json 2>$GRAFT_REPO_ROOT/gpurun_out/prof_sm.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
for p in glob.glob("gpurun_out/prof_sm/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(p)))[:14]:
        print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
rm -rf gpurun_out/prof_sm
