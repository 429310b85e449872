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
      python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/$C.log" 2>&1
done
python - "$OUT" "$ROOT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
sys.path.insert(0, sys.argv[2])
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
# the file bench.py quotes `roofline.traffic` from (profiles/pmc_traffic.json): unit and gfx950 corrections as
# MI355X_MICROARCH.md's HBM section prescribes (KiB; FETCH_SIZE under-reports reads by 2x on gfx950)
if res["FETCH_SIZE"]["launches"] and res["WRITE_SIZE"]["launches"]:
    rd = int(round(2 * res["FETCH_SIZE"]["mean"] * 1024)); wr = int(round(res["WRITE_SIZE"]["mean"] * 1024))
    json.dump({"_comment": "HBM traffic of ONE launch of the store kernel on the bench.py workload (1 Gbp, M=20), from separate "
                           "rocprofv3 --pmc passes (tools/collect_pmc.sh), mean over %d / %d launches of `python bench.py --steps 5 "
                           "--warmup 2` incl. its preheat" % (res["FETCH_SIZE"]["launches"], res["WRITE_SIZE"]["launches"]),
               "kernel": "lm::score_c32<20, 0, ...> (quad-gathered symbol loads)",
               "FETCH_SIZE_KiB": round(res["FETCH_SIZE"]["mean"], 1), "WRITE_SIZE_KiB": round(res["WRITE_SIZE"]["mean"], 1),
               "calibration": "WRITE_SIZE in KiB (factor 1.00); FETCH_SIZE under-reports by exactly 2x on gfx950 (MI355X_MICROARCH.md HBM "
                              "section; calibrated in round 2 on kernels of known traffic): read bytes = 2 * FETCH_SIZE * 1024",
               "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
               "algorithmic_bytes_per_launch": 5000000000,
               # the sources these counters were taken on: bench.py says `traffic_current: false` when they have changed since
               "store_kernel_digest": __import__("bench").store_kernel_digest()}, open(out + "/pmc_traffic.json", "w"), indent=1)
PY
