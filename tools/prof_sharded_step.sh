#!/bin/bash
# rocprofv3 kernel trace of the sharded step (score_into with the tracked maximum + merge) on one rank:
#   gpurun -- 'bash tools/prof_sharded_step.sh <tag> [bench flags]'   -> gpurun_out/<tag>_kernel_stats.csv, <tag>_gaps.txt
TAG=${1:-sharded}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o t -- \
    python "$ROOT/bench.py" --merge cabi --no-cpu-baseline --steps 100 --warmup 20 "$@" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}.err"
python - /tmp/prof_$TAG "$OUT/$TAG" <<'PY'
import csv, glob, sys, collections
src, out = sys.argv[1], sys.argv[2]
for p in glob.glob(src + "/**/*kernel_stats.csv", recursive=True):
    open(out + "_kernel_stats.csv", "w").write(open(p).read())
for p in glob.glob(src + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
    merges = [i for i, r in enumerate(rows) if "globalize_record" in r["Kernel_Name"]]
    rows = rows[merges[len(merges) // 4]:merges[-2]]   # steps that carry a merge, past the warm-up
    # one step = from a score_c32 start to the next score_c32 start
    starts = [i for i, r in enumerate(rows) if "score_c32" in r["Kernel_Name"]]
    with open(out + "_window.txt", "w") as f:      # raw sequence of three steps
        t0 = int(rows[starts[20]]["Start_Timestamp"])
        for r in rows[starts[20]:starts[23]]:
            f.write("%9.1f %9.1f q=%s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                              r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
    with open(out + "_gaps.txt", "w") as f:
        per = collections.defaultdict(list)
        for a, b in zip(starts[10:60], starts[11:61]):
            t0 = int(rows[a]["Start_Timestamp"])
            for r in rows[a:b]:
                per[r["Kernel_Name"][:60]].append((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0))
            per["(next step starts)"].append((int(rows[b]["Start_Timestamp"]) - t0, 0))
        for k, v in per.items():
            f.write("%-62s start %8.1f us  end %8.1f us   (n=%d)\n" % (k, sum(x[0] for x in v) / len(v) / 1e3,
                                                                        sum(x[1] for x in v) / len(v) / 1e3, len(v)))
PY
cat "$OUT/${TAG}_gaps.txt" "$OUT/${TAG}_window.txt"
