#!/bin/bash
# HBM traffic of the store kernel from PMC counters: one counter per pass, --kernel-trace only
# (gpurun refuses --pmc together with the hip/hsa trace domains).  Run on a GPU box:
#   gpurun --timeout 900 -- 'bash tools/collect_pmc.sh'
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -o p -- \
      python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/$C.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for p in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "score_c32<20, 0" in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    res[c] = {"launches": len(vals), "mean": sum(vals) / max(len(vals), 1)}
print(json.dumps(res))
json.dump(res, open(out + "/summary.json", "w"))
PY
