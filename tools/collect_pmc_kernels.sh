#!/bin/bash
# LDS bank conflicts and instruction mix of the main kernels (one counter group per pass).
#   gpurun --timeout 900 -- 'bash tools/collect_pmc_kernels.sh'
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_kernels
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/g$i" -o p -- \
      python "$ROOT/tools/api_overhead.py" 400000000 20 > "$OUT/g$i.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "lm::" not in n:
            continue
        k = n[n.index("lm::"):].split("(")[0][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as f:
    for k in sorted(acc):
        line = k.ljust(42) + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items()))
        print(line); f.write(line + "\n")
PY
