#!/bin/bash
# A/B of an upper bound on the wavefronts per SIMD (tools/build_variant.py tags mwl = long family, mws = short family)
mkdir -p gpurun_out/maxw
L=$PWD/lightmotif_amd/csrc
O=gpurun_out/maxw/ab.txt
for rep in 1 2; do
  for tag in base mwl; do
    if [ $tag = base ]; then unset LM_HIP_LIBRARY; else export LM_HIP_LIBRARY=$L/liblightmotif_hip_$tag.so; fi
    for tr in plain track; do
      echo "== $tag rep$rep long $tr" >> $O
      timeout 300 python tools/store_ab.py 1000000000 40,44,48,52,56,60,64 $tr >> $O 2>> gpurun_out/maxw/ab.err
    done
  done
  for tag in base mws; do
    if [ $tag = base ]; then unset LM_HIP_LIBRARY; else export LM_HIP_LIBRARY=$L/liblightmotif_hip_$tag.so; fi
    for tr in plain track; do
      echo "== $tag rep$rep short $tr" >> $O
      timeout 300 python tools/store_ab.py 1000000000 8,12,16,20,24,28,32,33,36 $tr >> $O 2>> gpurun_out/maxw/ab.err
    done
  done
done
python - <<'PY'
import json, collections
rows = collections.defaultdict(list)
tag = None
for line in open("gpurun_out/maxw/ab.txt"):
    if line.startswith("=="):
        _, tag, rep, fam, tr = line.split()
    elif line.startswith("{"):
        r = json.loads(line)
        rows[(fam, tr, r["M"], tag)].append((r["ms"], r["sha"]))
for (fam, tr, m, tag), v in sorted(rows.items()):
    print(fam, tr, "M=%d" % m, tag, " ".join("%.4f" % x[0] for x in v), v[0][1][:8])
PY
