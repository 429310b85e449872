#!/usr/bin/env python3
"""VGPRs, scratch (spill) bytes, LDS and wavefronts per SIMD of every gfx950 kernel in the built library: walks the
clang offload bundles embedded in liblightmotif_hip.so, extracts the code objects and reads their metadata notes
(llvm-readelf).  `python tools/kernel_regs.py [--lib PATH] [--grep prefilter2] [--spills]`."""
import argparse
import re
import struct
import subprocess
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"


def code_objects(blob: bytes):
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def kernels(lib: Path):
    out = []
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(lib.read_bytes())):
            f = Path(td) / f"co{i}.elf"
            f.write_bytes(co)
            notes = subprocess.run([READELF, "--notes", str(f)], capture_output=True, text=True).stdout
            for blockm in re.finditer(r"- \.agpr_count:.*?(?=\n\s+- \.agpr_count:|\namdhsa\.target|\Z)", notes, re.S):
                b = blockm.group(0)
                g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", b) or [None, None])[1]
                out.append({"name": g("name"), "vgpr": int(g("vgpr_count") or 0), "sgpr": int(g("sgpr_count") or 0),
                            "scratch": int(g("private_segment_fixed_size") or 0), "lds": int(g("group_segment_fixed_size") or 0),
                            "spill_vgpr": int(g("vgpr_spill_count") or 0)})
    names = subprocess.run([CXXFILT], input="\n".join(k["name"] for k in out), capture_output=True, text=True).stdout.split("\n")
    for k, n in zip(out, names):
        k["demangled"] = re.sub(r"\(.*", "", n).replace("void ", "")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=str(ROOT / "lightmotif_amd" / "csrc" / "liblightmotif_hip.so"))
    ap.add_argument("--grep", default="")
    ap.add_argument("--spills", action="store_true", help="only kernels that use scratch")
    a = ap.parse_args()
    ks = [k for k in kernels(Path(a.lib)) if a.grep in k["demangled"] and (k["scratch"] or not a.spills)]
    for k in sorted(ks, key=lambda k: k["demangled"]):
        waves = min(8, 512 // max(k["vgpr"], 1)) if k["vgpr"] else 8
        print(f"{k['demangled']:60s} vgpr {k['vgpr']:3d} (<= {waves} waves/SIMD)  scratch {k['scratch']:4d} B  spilled vgprs {k['spill_vgpr']:3d}  static lds {k['lds']}")
    print(f"{len(ks)} kernels")


if __name__ == "__main__":
    main()
