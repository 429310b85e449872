#!/bin/bash
# Issue / stall breakdown of the library's kernels from the SQ counters (one counter group per
# pass, --kernel-trace only: gpurun refuses --pmc together with the hip/hsa trace domains).
#   gpurun --timeout 1200 -- 'bash tools/collect_stalls.sh r02 "python bench.py --steps 6 --warmup 3 --no-cpu-baseline --preheat-ms 0"'
# Units (MI355X_MICROARCH.md, "rocprofv3 PMC slots"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
# count quad-cycles summed over waves; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CMD=${2:-"python $ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --preheat-ms 0"}
OUT=$ROOT/gpurun_out/stalls_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
GROUPS_=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"
 "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
 "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_WAVE32_LDS SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_FLAT"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LEVEL_WAVES"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr"
 "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
if [ "$3" = tlb ]; then   # third argument "tlb": address-translation counters only
GROUPS_=(
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
 "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum"
 "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
)
fi
i=0
for G in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/g$i" -o p -- $CMD \
      > "$OUT/g$i.log" 2>&1 || echo "group $i failed: $G" >> "$OUT/failed.txt"
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "lm::" not in n and "kb::" not in n and "mix_" not in n and not n.startswith("fill"):
            continue
        k = (n[n.index("::") - 2:] if "::" in n.split("(")[0] else n.replace("void ", "")).split("(")[0][:48]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
# kernel durations of the same passes (ns), to turn GRBM_GUI_ACTIVE into a clock
for p in glob.glob(out + "/g*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"]
        if "lm::" not in n and "kb::" not in n and "mix_" not in n and not n.startswith("fill"):
            continue
        k = (n[n.index("::") - 2:] if "::" in n.split("(")[0] else n.replace("void ", "")).split("(")[0][:48]
        acc[k]["duration_ns(all passes)"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
with open(out + "/summary.txt", "w") as f:
    for k in sorted(acc):
        f.write(k + "\n")
        for c, v in sorted(acc[k].items()):
            f.write(f"    {c:34s} {sum(v)/len(v):14.5g}   (n={len(v)})\n")
print(open(out + "/summary.txt").read()[:6000])
PY
rm -rf "$OUT"/g*/  # the raw per-dispatch CSVs are large
