"""Misuse of the C ABI on a live context: a status and a message, never a fault (error conventions:
"C functions return int status ... No C++ exceptions across the ABI"; the Rust shim turns non-zero into panic!).

The calls run in a child process so that a fault would be reported as the NAME of the entry point, not as a dead
test session."""
import json
import re
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

PROBE = r"""
import ctypes as C, json, re, sys
sys.path.insert(0, sys.argv[1])
from lightmotif_amd import _ffi
L = _ffi.lib()
header = open(sys.argv[1] + "/include/lightmotif_hip.h").read()
body = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", header, flags=re.S))
ctx_first = set(re.findall(r"\b(lm_hip_[a-z0-9_]+)\s*\(\s*(?:const\s+)?lm_hip_ctx\s*\*", body))
ctx = C.c_void_p()
assert L.lm_hip_ctx_create(0, C.byref(ctx)) == 0
out = {}
for name, (res, args) in _ffi.SIGNATURES.items():
    if res is not C.c_int or name not in ctx_first or name == "lm_hip_ctx_destroy":
        continue
    vals = [0.0 if a in (C.c_float, C.c_double) else b"D" if a is C.c_char else 0 if a in (C.c_int, C.c_size_t, C.c_uint, C.c_uint8)
            else None for a in args]
    vals[0] = ctx
    print(name, file=sys.stderr, flush=True)
    st = getattr(L, name)(*vals)
    msg = _ffi.last_error() if st else ""
    out[name] = [st, msg, L.lm_hip_ctx_sync(ctx)]
# the context still works afterwards
import numpy as np
import lightmotif_amd as lm
pli = lm.Pipeline.hip(0)
seq = pli.stripe(lm.EncodedSequence(np.arange(100, dtype=np.uint8) % 4), 32)
print(json.dumps(out))
"""

# entry points whose all-zero argument list is a valid (empty) request
EMPTY_IS_VALID = {"lm_hip_ctx_sync", "lm_hip_ctx_set_rows_per_stream", "lm_hip_ctx_set_prefilter",
                  "lm_hip_ctx_set_track_argmax", "lm_hip_encode_dptr", "lm_hip_ctx_set_xcd_remap",
                  # diagnostics whose outputs are optional: a bracket with nothing in it, counts nobody asked for
                  "lm_hip_ctx_last_scan_counts"}


def test_null_arguments_on_a_live_context():
    r = subprocess.run([sys.executable, "-c", PROBE, str(ROOT)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "crashed in " + (r.stderr.strip().splitlines() or ["?"])[-1] + "\n" + r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(got) >= 48
    for name, (st, msg, sync) in got.items():
        assert sync == 0, (name, "left the context's stream in an error state")
        if name in EMPTY_IS_VALID:
            assert st == 0, (name, st, msg)
        else:
            assert st != 0 and msg, (name, st, msg)


MUTATE = r"""
import ctypes as C, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from lightmotif_amd import _ffi
L = _ffi.lib()
vp = C.c_void_p
ctx = vp(); assert L.lm_hip_ctx_create(0, C.byref(ctx)) == 0
rng = np.random.default_rng(1)
enc = rng.integers(0, 4, 5000, dtype=np.uint8)
w = np.zeros((9, 8), np.float32); w[:, :5] = rng.normal(0, 2, (9, 5)).astype(np.float32)
pssm = vp(); assert L.lm_hip_pssm_create(ctx, w.ctypes.data, 9, 8, 5, C.byref(pssm)) == 0
seq = vp(); assert L.lm_hip_seq_from_encoded(ctx, enc.ctypes.data, enc.size, 32, 5, C.byref(seq)) == 0
assert L.lm_hip_seq_configure_wrap(ctx, seq, 8) == 0
scores = vp(); assert L.lm_hip_scores_create(ctx, 32, C.byref(scores)) == 0
assert L.lm_hip_score_into(ctx, pssm, seq, scores) == 0
found, coords, val, n = C.c_int(), _ffi.Coords(), C.c_float(), C.c_size_t()
cptr, fptr, hptr = C.POINTER(_ffi.Coords)(), C.POINTER(C.c_float)(), C.POINTER(_ffi.Hit)()
hit = _ffi.Hit()
dw = rng.integers(0, 20, (9, 5), dtype=np.uint8)
text = np.frombuffer(b"ACTG", np.uint8)[enc].copy()
packed = np.zeros((enc.size + 3) // 4, np.uint8)
host = np.zeros((200, 32), np.float32); hostseq = np.zeros((200, 32), np.uint8)
out_h = vp()
pl = (vp * 1)(pssm)
thr = (C.c_float * 1)(1.0)
off = (C.c_size_t * 2)()
calls = {
    "lm_hip_pssm_create": [ctx, w.ctypes.data, 9, 8, 5, C.byref(out_h)],
    "lm_hip_pssm_reverse_complement": [ctx, pssm, C.byref(out_h)],
    "lm_hip_seq_from_encoded": [ctx, enc.ctypes.data, enc.size, 32, 5, C.byref(out_h)],
    "lm_hip_seq_from_ascii": [ctx, b"D", text.ctypes.data, text.size, 32, 0, C.byref(out_h), C.byref(n)],
    "lm_hip_seq_from_2bit": [ctx, packed.ctypes.data, None, None, 0, enc.size, 32, C.byref(out_h)],
    "lm_hip_seq_configure_wrap": [ctx, seq, 8],
    "lm_hip_seq_download": [ctx, seq, hostseq.ctypes.data],
    "lm_hip_scores_create": [ctx, 32, C.byref(out_h)],
    "lm_hip_scores_download": [ctx, scores, host.ctypes.data],
    "lm_hip_scores_download_rows": [ctx, scores, 0, 10, host.ctypes.data],
    "lm_hip_score_rows_into": [ctx, pssm, seq, 0, 100, scores],
    "lm_hip_score_into": [ctx, pssm, seq, scores],
    "lm_hip_argmax": [ctx, scores, C.byref(found), C.byref(coords), C.byref(val)],
    "lm_hip_max": [ctx, scores, C.byref(found), C.byref(val)],
    "lm_hip_threshold": [ctx, scores, 1.0, C.byref(cptr), C.byref(n)],
    "lm_hip_scan_f32": [ctx, pssm, seq, 1.0, C.byref(hptr), C.byref(n)],
    "lm_hip_scan_max_f32": [ctx, pssm, seq, dw.ctypes.data, 5, 1, 10, 0, 0, 0.0, 0, C.byref(found), C.byref(hit)],
    "lm_hip_scan_argmax_batch": [ctx, pl, 1, seq, C.byref(found), C.byref(coords), C.byref(val)],
    "lm_hip_scan_threshold_batch": [ctx, pl, thr, 1, seq, off, C.byref(cptr), C.byref(fptr)],
    "lm_hip_score_u8": [ctx, dw.ctypes.data, 9, 5, 5, seq, 0, 100, 1, hostseq.ctypes.data, 32, C.byref(n), C.byref(n)],
}
import json
res = {}
for name, good in calls.items():
    st = getattr(L, name)(*good)
    res[name + ":valid"] = [st, _ffi.last_error() if st else "", 0]
    args = _ffi.SIGNATURES[name][1]
    for i, a in enumerate(args):
        if i == 0 or a in (C.c_float, C.c_double, C.c_char, C.c_int, C.c_size_t, C.c_uint, C.c_uint8) or good[i] is None:
            continue
        bad = list(good); bad[i] = None
        print(name, "arg", i, file=sys.stderr, flush=True)
        st = getattr(L, name)(*bad)
        res[f"{name}:{i}"] = [st, _ffi.last_error() if st else "", L.lm_hip_ctx_sync(ctx)]
print(json.dumps(res))
"""

# pointer arguments the header documents as optional outputs
OPTIONAL = {"lm_hip_seq_from_ascii:7", "lm_hip_argmax:3", "lm_hip_argmax:4", "lm_hip_max:3", "lm_hip_scan_argmax_batch:5",
            "lm_hip_scan_argmax_batch:6", "lm_hip_scan_threshold_batch:7", "lm_hip_score_u8:11", "lm_hip_score_u8:12"}


def test_each_pointer_nulled_in_turn_on_valid_calls():
    """Twenty handle-based entry points, called correctly once and then with every pointer argument nulled in turn:
    the valid call succeeds, a missing required pointer is LM_HIP_ERR_BAD_ARGS with a message, a missing optional
    output is accepted, nothing faults and the context's stream stays usable."""
    r = subprocess.run([sys.executable, "-c", MUTATE, str(ROOT)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "crashed in " + (r.stderr.strip().splitlines() or ["?"])[-1] + "\n" + r.stderr[-2000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(got) >= 75
    for key, (st, msg, sync) in got.items():
        assert sync == 0, key
        if key.endswith(":valid") or key in OPTIONAL:
            assert st == 0, (key, st, msg)
        else:
            assert st == 1 and msg, (key, st, msg)
