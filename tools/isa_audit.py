#!/usr/bin/env python3
"""Audits every gfx950 kernel of the built library for the hardware fault behind the round-5 parity loss:

    a 64-bit shift (v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64) whose SHIFT AMOUNT sits in the last VGPR of the
    wavefront's allocation -- register N with N % 8 == 7 and N + 1 not allocated -- returns a wrong result on MI355X,
    depending on what the neighbouring allocation holds.

LLVM knows this erratum for gfx11 (`FeatureShift64HighRegBug`, GCNHazardRecognizer::fixShift64HighRegBug) and moves
the amount to another register there; hipcc 7.0 does not apply the workaround for gfx950, where
tools/kbench/shift64_repro reproduces it in isolation.  The kernels are disassembled from the code objects embedded
in liblightmotif_hip.so (clang offload bundles), their register allocation comes from the metadata notes.

    python tools/isa_audit.py [--lib PATH] [-v]          exit status 1 on a hit
"""
from __future__ import annotations

import argparse
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
from kernel_regs import READELF, code_objects  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
SHIFTS = ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")
GRANULE = 8  # VGPR allocation granule of gfx90a and later (unified register file)


def kernel_vgprs(elf: Path) -> dict[str, tuple[int, int]]:
    """mangled kernel name -> (vgpr_count, agpr_count) from the code object's metadata"""
    notes = subprocess.run([READELF, "--notes", str(elf)], capture_output=True, text=True, check=True).stdout
    out = {}
    for block in re.finditer(r"- \.agpr_count:.*?(?=\n\s+- \.agpr_count:|\namdhsa\.target|\Z)", notes, re.S):
        b = block.group(0)
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b)
        name = g("name").group(1)
        out[name] = (int(g("vgpr_count").group(1)), int(g("agpr_count").group(1)))
    return out


def shift_hazards(disasm: str, vgprs: dict[str, tuple[int, int]]):
    """(kernel, instruction, amount register, vgprs allocated) for every 64-bit shift by the last allocated VGPR"""
    found, checked = [], 0
    kernel = None
    for line in disasm.split("\n"):
        m = re.match(r"[0-9a-f]+ <([^>]+)>:", line)
        if m:
            kernel = m.group(1)
            continue
        m = re.match(r"\s*(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)\S*\s+v\[\d+:\d+\],\s*(\S+?),", line)
        if not m or kernel not in vgprs:
            continue
        checked += 1
        amount = m.group(2)
        r = re.fullmatch(r"v(\d+)", amount)
        if not r:
            continue  # an SGPR or a constant
        n = int(r.group(1))
        count, agprs = vgprs[kernel]
        allocated = (count + GRANULE - 1) // GRANULE * GRANULE  # (with AGPRs the register behind the last VGPR is a0: allocated)
        if n % GRANULE == GRANULE - 1 and n + 1 >= allocated and agprs == 0:
            found.append((kernel, line.split("//")[0].strip(), n, allocated))
    return found, checked


def audit(lib: Path):
    hits, nk, nshift = [], 0, 0
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(lib.read_bytes())):
            f = Path(td) / f"co{i}.elf"
            f.write_bytes(co)
            vg = kernel_vgprs(f)
            nk += len(vg)
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", str(f)], capture_output=True, text=True, check=True).stdout
            h, c = shift_hazards(dis, vg)
            hits += h
            nshift += c
    return hits, nk, nshift


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=str(ROOT / "lightmotif_amd" / "csrc" / "liblightmotif_hip.so"))
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    hits, nk, nshift = audit(Path(a.lib))
    names = subprocess.run(["c++filt"], input="\n".join(h[0] for h in hits), capture_output=True, text=True).stdout.split("\n")
    for (k, ins, n, alloc), dn in zip(hits, names):
        short = dn.split("(")[0].replace("void ", "")
        print(f"HIT  {short}: `{ins}` -- amount in v{n}, the last of {alloc} allocated VGPRs")
    print(f"{nk} kernels, {nshift} 64-bit shifts by a register or constant, {len(hits)} with the amount in the last allocated VGPR")
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
