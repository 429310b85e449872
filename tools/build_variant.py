#!/usr/bin/env python3
"""Builds a variant of the library with extra compiler flags next to the shipped one, for A/B runs on
one box:   python tools/build_variant.py e0 -DLM_LDS_WHOLE_ROWS=0
-> lightmotif_amd/csrc/liblightmotif_hip_e0.so (git-ignored); select it with LM_HIP_LIBRARY=<path>.
`--short` anywhere among the flags: only the short family's units (score_inst.hip) are rebuilt with them."""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from lightmotif_amd import build as B  # noqa: E402

tag, extra = sys.argv[1], sys.argv[2:]
long_extra = []
if "--long" in extra:       # flags after --long go to the long-family units (score_long_inst.hip) only
    i = extra.index("--long")
    extra, long_extra = extra[:i], extra[i + 1:]
short_only = "--short" in extra   # the flags go to the short family (score_inst.hip, M = 1 ... 36) only: other objects are the shipped ones
if short_only:
    extra = [e for e in extra if e != "--short"]
obj = B.CSRC / f"_obj_{tag}"
obj.mkdir(exist_ok=True)
hipcc = B._hipcc()
cmds, objs = [], []
for unit in B.UNITS:
    o = obj / (unit + ".o")
    objs.append(o)
    cmds.append([hipcc, *B.FLAGS, *extra, "-c", str(B.CSRC / unit), "-o", str(o)])
for inst, lo, hi in B.INST:
    o = obj / f"score_inst_{inst}.o"
    objs.append(o)
    cmds.append([hipcc, *B.FLAGS, *extra, f"-DLM_M_LO={lo}", f"-DLM_M_HI={hi}", f"-DLM_INST_ID={inst}", "-c",
                 str(B.CSRC / "score_inst.hip"), "-o", str(o)])
for m in B.LONG:
    o = obj / f"score_long_inst_{m}.o"
    objs.append(o)
    cmds.append([hipcc, *B.FLAGS, *B.LONG_FLAGS, *extra, *long_extra, f"-DLM_LONG_M={m}", "-c",
                 str(B.CSRC / "score_long_inst.hip"), "-o", str(o)])
for m in B.XLONG:
    o = obj / f"score_xlong_inst_{m}.o"
    objs.append(o)
    cmds.append([hipcc, *B.FLAGS, *B.LONG_FLAGS, *extra, f"-DLM_XLONG_M={m}", "-c",
                 str(B.CSRC / "score_xlong_inst.hip"), "-o", str(o)])
for lo, hi in B.PAIR:
    o = obj / f"score_pair_inst_{lo}.o"
    objs.append(o)
    cmds.append([hipcc, *B.FLAGS, *B.LONG_FLAGS, *extra, f"-DLM_PAIR_LO={lo}", f"-DLM_PAIR_HI={hi}", "-c",
                 str(B.CSRC / "score_pair_inst.hip"), "-o", str(o)])
if long_extra and not extra:   # only the long units differ: reuse the shipped objects for the rest
    keep = [c for c in cmds if "score_long_inst.hip" in " ".join(c)]
    objs = [B.OBJ / o.name if "score_long_inst" not in o.name else o for o in objs]
    cmds = keep
if short_only:
    cmds = [c for c in cmds if "score_inst.hip" in " ".join(c)]
    objs = [o if o.name.startswith("score_inst_") else B.OBJ / o.name for o in objs]
with ThreadPoolExecutor(max_workers=8) as ex:
    list(ex.map(B._run, cmds))
lib = B.CSRC / f"liblightmotif_hip_{tag}.so"
B._run([hipcc, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", str(lib), *map(str, objs), "-ldl"])
print(lib)
