"""CPU checks of the C-ABI library: it loads, exports every symbol declared in
include/lightmotif_hip.h, and refuses to run without a gfx950 device (no CPU
fallback).  No compute is attempted here."""
import ctypes as C
import os
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "lightmotif_hip.h").read_text()


def declared_symbols():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(lm_hip_[a-z0-9_]+)\s*\(", body)))


def test_header_declares_what_the_binding_binds():
    from lightmotif_amd import _ffi
    assert declared_symbols() == sorted(_ffi.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from lightmotif_amd import _ffi
    L = _ffi.lib()  # raises if the .so is missing or a symbol is absent
    out = subprocess.run(["nm", "-D", "--defined-only", str(_ffi.LIB_PATH)], capture_output=True,
                         text=True, check=True).stdout
    exported = set(re.findall(r" T (lm_hip_[a-z0-9_]+)", out))
    assert set(declared_symbols()) <= exported
    assert L.lm_hip_abi_version() == 1


def test_every_entry_point_cites_the_reference():
    """Each declaration block in the header names the reference lines it stands in for."""
    cites = re.findall(r"[a-z0-9_/]+\.rs:\d+", HEADER)
    assert len(cites) >= 30
    for name in ("pli/mod.rs:72-106", "pli/mod.rs:135-155", "pli/mod.rs:210-221", "seq.rs:369-381",
                 "avx2.rs:889-904", "dense.rs:126-128", "scores.rs:155-157"):
        assert name in HEADER


def test_stride_matches_reference_table():
    from lightmotif_amd import _ffi
    L = _ffi.lib()
    table = [(32, 1, 32), (16, 1, 32), (32, 4, 32), (8, 4, 8), (16, 4, 16), (33, 1, 64),
             (5, 4, 8), (21, 4, 24), (1, 1, 32)]  # dense.rs:367-391 + SURVEY A4
    for cols, elem, want in table:
        assert L.lm_hip_stride(cols, elem) == want


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_device_is_unsupported_backend_not_a_fallback():
    import lightmotif_amd as lm
    assert lm.Pipeline.device_count() == 0
    with pytest.raises(lm.UnsupportedBackend):
        lm.Pipeline.hip(0)
    # the host-pointer entry points must fail too, not compute on the CPU
    from lightmotif_amd import _ffi
    L = _ffi.lib()
    found = C.c_int(0)
    buf = (C.c_float * 32)()
    st = L.lm_hip_argmax_f32(buf, 1, 32, 32, C.byref(found), None, None)
    assert st == _ffi.ERR_NO_DEVICE


def test_product_never_touches_the_oracle():
    """The shipped package may not import, link or call anything under oracle/."""
    for path in (ROOT / "lightmotif_amd").rglob("*"):
        if path.suffix in {".py", ".hip", ".hpp", ".cpp", ".h"}:
            text = path.read_text()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), path
            assert "lm_oracle" not in text and "lmo_" not in text and "lma_" not in text, path
    out = subprocess.run(["ldd", str(ROOT / "lightmotif_amd" / "csrc" / "liblightmotif_hip.so")],
                         capture_output=True, text=True).stdout
    assert "lm_oracle" not in out and "lm_avx2" not in out


NULL_CALLS = r"""
import ctypes as C, json, sys
sys.path.insert(0, sys.argv[1])
from lightmotif_amd import _ffi
L = _ffi.lib()
out = {}
for name, (res, args) in _ffi.SIGNATURES.items():
    if res is not C.c_int or not args:
        continue
    vals = [0.0 if a in (C.c_float, C.c_double) else b"D" if a is C.c_char else 0 if a in (C.c_int, C.c_size_t, C.c_uint, C.c_uint8)
            else None for a in args]
    print(name, file=sys.stderr, flush=True)          # the last name printed is the one that crashed, if any does
    st = getattr(L, name)(*vals)
    out[name] = [st, _ffi.last_error() if st else ""]
print(json.dumps(out))
"""


def test_null_arguments_are_a_status_not_a_crash():
    """SURVEY 8(b) error conventions: misuse comes back as a status with a message, never as a fault or an exception
    across the ABI.  Every int-returning entry point is called with null handles / pointers and zero sizes (before
    any device work, so this runs without a GPU); `*_destroy(NULL)` is a no-op like free(NULL)."""
    import json
    import sys
    r = subprocess.run([sys.executable, "-c", NULL_CALLS, str(ROOT)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, "crashed in " + (r.stderr.strip().splitlines() or ["?"])[-1]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(got) >= 65
    for name, (st, msg) in got.items():
        if name.endswith("_destroy") or name in ("lm_hip_host_spread_lanes", "lm_hip_host_reuse_scores", "lm_hip_host_set_crossover"):   # (plain switches: nothing to misuse)
            assert st == 0, (name, st, msg)
        else:
            assert st != 0 and msg, (name, st, msg)


def test_header_is_plain_c99_and_links_from_c(tmp_path):
    """The boundary is a C ABI (bindgen / cgo / ctypes read the header as C, not C++): it must parse as strict C99,
    and a C program linked against the library must resolve its symbols (no compute: only the version and the
    no-device status are asked for)."""
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include "lightmotif_hip.h"\n'
                   "int main(void) {\n"
                   "    lm_hip_ctx *ctx = NULL;\n"
                   "    int n = -1;\n"
                   "    if (lm_hip_abi_version() != 1) return 2;\n"
                   "    if (lm_hip_stride(32, 1) != 32 || lm_hip_stride(5, 4) != 8 || lm_hip_stride(21, 4) != 24) return 3;\n"
                   "    (void)lm_hip_device_count(&n);\n"
                   "    if (n <= 0 && lm_hip_ctx_create(0, &ctx) == LM_HIP_OK) return 4;  /* no device: never a fallback */\n"
                   '    printf("devices %d\\n", n);\n'
                   "    return 0;\n}\n")
    from lightmotif_amd import _ffi
    exe = tmp_path / "probe"
    libdir = _ffi.LIB_PATH.parent
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe),
                    f"-L{libdir}", "-llightmotif_hip", f"-Wl,-rpath,{libdir}"], check=True, capture_output=True, text=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
