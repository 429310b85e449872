"""Builds ``liblightmotif_hip.so`` (HIP kernels + C ABI) for gfx950 with hipcc.

Cross-compiles without a GPU.  Run as ``python -m lightmotif_amd.build`` or
through ``__graft_entry__.build()``.  The library is written next to the
sources (``lightmotif_amd/csrc/``) so it ships with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
ROOT = PKG.parent
OBJ = CSRC / "_obj"
LIB = CSRC / "liblightmotif_hip.so"

ARCH = "gfx950"
# -fno-slp-vectorize: packed v_pk_add_f32 measured slower than scalar v_add_f32 on
# MI355X for this kernel (profiles/r01_kbench_*.txt).  -ffp-contract=off and no
# fast-math: the score is M sequential IEEE adds (pli/mod.rs:98-102).
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-slp-vectorize", "-fno-fast-math", "-Wall", "-Wno-unused-function",
         f"-I{ROOT / 'include'}", f"-I{CSRC}"]
# score_inst.hip is compiled 9 times: motif lengths 4*i+1 .. 4*i+4 (M = 1 .. 36)
INST = [(i, 4 * i + 1, 4 * i + 4) for i in range(9)]
# score_long_inst.hip is compiled once per padded long motif length M' = 40, 44 ... 64 (exact f32 kernels only);
# the raised unroll budget is what keeps their accumulators in registers (see the file's header)
LONG = list(range(40, 65, 4))
LONG_FLAGS = ["-mllvm", "-pragma-unroll-threshold=10000000"]
# score_xlong_inst.hip: the plain store kernel alone for padded lengths 72, 80, 88 (one pass for motifs of 65 ... 88 rows)
XLONG = list(range(72, 89, 8))
# score_pair_inst.hip: the pair-symbol prefilter scan alone for the lengths beyond the exact kernels (65 ... 128)
PAIR = [(65, 80), (81, 96), (97, 112), (113, 128)]
UNITS = ["score_plan.hip", "score_store.hip", "score_argmax.hip", "score_threshold.hip", "reduce.hip", "hits.hip", "discrete.hip", "layout.hip", "scanmax.hip", "context.hip", "pssm.hip", "score_api.hip", "handles.hip",
         "hostptr.hip", "comm.hip"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP back-end cannot be built")
    return exe


def _newer(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: " + " ".join(cmd) + "\n" + r.stdout)
    if r.stdout.strip() and os.environ.get("LM_BUILD_VERBOSE"):
        print(r.stdout)


def build(force: bool = False, jobs: int | None = None) -> Path:
    hipcc = _hipcc()
    OBJ.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.hpp")) + [ROOT / "include" / "lightmotif_hip.h", Path(__file__)]
    cmds, objs = [], []
    for unit in UNITS:
        obj = OBJ / (unit + ".o")
        objs.append(obj)
        if force or _newer(obj, [CSRC / unit] + headers):
            cmds.append([hipcc, *FLAGS, "-c", str(CSRC / unit), "-o", str(obj)])
    for inst, lo, hi in INST:
        obj = OBJ / f"score_inst_{inst}.o"
        objs.append(obj)
        if force or _newer(obj, [CSRC / "score_inst.hip"] + headers):
            cmds.append([hipcc, *FLAGS, f"-DLM_M_LO={lo}", f"-DLM_M_HI={hi}", f"-DLM_INST_ID={inst}",
                         "-c", str(CSRC / "score_inst.hip"), "-o", str(obj)])
    for m in LONG:
        obj = OBJ / f"score_long_inst_{m}.o"
        objs.append(obj)
        if force or _newer(obj, [CSRC / "score_long_inst.hip"] + headers):
            cmds.append([hipcc, *FLAGS, *LONG_FLAGS, f"-DLM_LONG_M={m}", "-c", str(CSRC / "score_long_inst.hip"),
                         "-o", str(obj)])
    for m in XLONG:
        obj = OBJ / f"score_xlong_inst_{m}.o"
        objs.append(obj)
        if force or _newer(obj, [CSRC / "score_xlong_inst.hip"] + headers):
            cmds.append([hipcc, *FLAGS, *LONG_FLAGS, f"-DLM_XLONG_M={m}", "-c", str(CSRC / "score_xlong_inst.hip"),
                         "-o", str(obj)])
    for lo, hi in PAIR:
        obj = OBJ / f"score_pair_inst_{lo}.o"
        objs.append(obj)
        if force or _newer(obj, [CSRC / "score_pair_inst.hip"] + headers):
            cmds.append([hipcc, *FLAGS, *LONG_FLAGS, f"-DLM_PAIR_LO={lo}", f"-DLM_PAIR_HI={hi}", "-c",
                         str(CSRC / "score_pair_inst.hip"), "-o", str(obj)])
    if cmds:
        # longest units first (the long family and the high motif lengths unroll into the largest kernels), so that
        # the pool does not end on one straggler
        def weight(cmd):
            text = " ".join(cmd)
            for m in LONG:
                if f"-DLM_LONG_M={m}" in text:
                    return 1000 + m
            for m in XLONG:
                if f"-DLM_XLONG_M={m}" in text:
                    return 900 + m
            for inst, lo, hi in INST:
                if f"-DLM_INST_ID={inst}" in text:
                    return 100 + hi
            if "score_pair_inst.hip" in text:
                return 90
            return 50 if "score_" in text else 0
        cmds.sort(key=weight, reverse=True)
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(_run, cmds))
    if cmds or force or not LIB.exists():
        # -ldl: comm.hip opens librccl.so.1 on first use (no link-time dependency on RCCL)
        _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB),
              *map(str, objs), "-ldl"])
        _audit(LIB)
    return LIB


def _audit(lib: Path) -> None:
    """A library with a 64-bit shift by the last allocated VGPR in any kernel is not shipped: that instruction returns
    garbage on gfx950 depending on what else runs on the chip (tools/isa_audit.py).  hipcc does not avoid the
    allocation; when it happens, perturb the kernel named in the message (one more live register is enough)."""
    tools = str(ROOT / "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import isa_audit
    hits, _, _ = isa_audit.audit(lib)
    if hits:
        lib.unlink()
        raise RuntimeError("isa_audit: 64-bit shift by the last allocated VGPR in: " +
                           "; ".join(f"{k} `{ins}`" for k, ins, _, _ in hits))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
