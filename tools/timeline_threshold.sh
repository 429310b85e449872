#!/bin/bash
# rocprofv3 kernel trace of the last fused threshold / fused argmax call of
# tools/timeline_threshold.py: per-kernel start offset and duration (GPU box only)
PV=${1:-1e-5}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python $GRAFT_REPO_ROOT/tools/timeline_threshold.py $PV > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python - <<PY
import csv, glob
p = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
def dump(last_name, title, before, after):
    hits = [i for i, r in enumerate(rows) if last_name in r["Kernel_Name"]]
    if not hits:
        print("no", last_name); return
    i1 = hits[-1]
    sel = rows[max(0, i1 - before): i1 + after + 1]
    t0 = int(sel[0]["Start_Timestamp"])
    print(title)
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  {(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:80]}")
dump("hits_rank_emit", "fused threshold (last call)", 12, 2)
dump("hits_best_single", "fused argmax (last call)", 9, 1)
PY
