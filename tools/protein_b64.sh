#!/bin/bash
# Protein (K = 21) LDS experiment of round 3: 16-byte reads (4 * odd row stride, shipped) against single 8-byte reads
# (2 * odd stride, `tools/build_variant.py b64 -DLM_LDS_B64=1`): parity of the variant, interleaved timings of
# configs[4], and the LDS counters of both.   gpurun --timeout 1500 -- 'bash tools/protein_b64.sh r03f'
TAG=${1:-r03f}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd "$ROOT"
B64=$ROOT/lightmotif_amd/csrc/liblightmotif_hip_b64.so
LM_HIP_LIBRARY=$B64 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_discrete.py -m gpu -x -q < /dev/null 2>&1 | grep -E "passed|failed|rror" | tail -3 > "$OUT/b64_parity.txt"
cat "$OUT/b64_parity.txt"
for rep in 1 2 3; do
  for tag in base b64; do
    if [ $tag = base ]; then unset LM_HIP_LIBRARY; else export LM_HIP_LIBRARY=$B64; fi
    timeout 120 python tools/bench_configs.py c5 2>/dev/null < /dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line)
        print('$tag rep$rep: store %.4f ms (hbm %.3f)  fused thr prefilter %.4f ms  exact %.4f ms  fused argmax %.4f ms' % (r['ms'], r['hbm_frac'], r['fused_threshold_prefilter_ms'], r['fused_threshold_exact_ms'], r['fused_argmax_ms']))"
  done
done | tee "$OUT/c5_ab.txt"
unset LM_HIP_LIBRARY
cd /tmp && export TMPDIR=/tmp
for tag in base b64; do
  if [ $tag = base ]; then unset LM_HIP_LIBRARY; else export LM_HIP_LIBRARY=$B64; fi
  i=0
  for G in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d "$OUT/pmc_${tag}_g$i" -o p -- python "$ROOT/tools/bench_configs.py" c5 > "$OUT/pmc_${tag}_g$i.log" 2>&1 < /dev/null || echo "group $i failed ($tag)"
  done
done
unset LM_HIP_LIBRARY
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
with open(out + "/lds_counters.txt", "w") as f:
    for tag in ("base", "b64"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for p in glob.glob(f"{out}/pmc_{tag}_g*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(p)):
                n = r["Kernel_Name"]
                if "lm::score_c32" not in n:
                    continue
                k = n[n.index("lm::"):].split("(")[0][:60]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(acc):
            c = {name: sum(v) / len(v) for name, v in acc[k].items()}
            line = f"{tag:5s} {k:58s} " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items()))
            if c.get("SQ_LDS_IDX_ACTIVE"):
                line += f"  | conflict/active={c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.3f}"
                if c.get("SQ_INSTS_LDS"):
                    line += f"  LDS cycles/instr={c['SQ_LDS_IDX_ACTIVE'] / c['SQ_INSTS_LDS']:.2f}"
            f.write(line + "\n")
print(open(out + "/lds_counters.txt").read()[:5000])
PY
rm -rf "$OUT"/pmc_*_g*/
