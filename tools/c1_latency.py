#!/usr/bin/env python3
"""configs[0] (lightmotif-bench dna.rs:81-109) latency: `score_into` + `argmax` per iteration on a 464 165 bp
sequence, one re-used StripedScores, at the dispatch geometry (C = 32) and the Generic bench geometry (C = 1).
Prints the wall time per iteration through the Python mirror and through bare ctypes calls of the C ABI
(no Python object work in between), and of the pieces.  GPU box only:  python tools/c1_latency.py"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402
from lightmotif_amd import _ffi  # noqa: E402


import os
REPS = int(os.environ.get("LM_C1_REPS", "2000"))


def wall(fn, reps=None, warm=None):
    reps = reps or REPS
    warm = warm if warm is not None else max(reps // 10, 5)
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    pli = lm.Pipeline.hip(0)
    L = pli._L
    length = 464_165
    rng = np.random.default_rng(0xEC011)
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    enc[391_677:391_677 + 15] = lm.EncodedSequence("GTTGACCTTATCAAC").data
    pssm = lm.create(["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"]).counts.normalize(0.1).log_odds()
    out = {}
    for cols in (32, 1):
        seq = pli.stripe(lm.EncodedSequence(enc), cols)
        seq.configure(pssm)
        scores = lm.StripedScores.empty(pli, cols)
        hp, hs, hq, hc = pssm._device(pli), seq._h, scores._h, pli._h
        found, best, val = C.c_int(0), _ffi.Coords(), C.c_float(0)

        def py_iter():
            pli.score_into(pssm, seq, scores)
            return pli.argmax(scores)

        def c_iter():
            L.lm_hip_score_into(hc, hp, hs, hq)
            L.lm_hip_argmax(hc, hq, C.byref(found), C.byref(best), C.byref(val))

        def c_score_sync():
            L.lm_hip_score_into(hc, hp, hs, hq)
            L.lm_hip_ctx_sync(hc)

        slen, swrap, srows, sstride, scols, sptr = seq._info()

        def c_fused():
            L.lm_hip_score_argmax_f32_dptr(hc, hp, sptr, srows + swrap, sstride, scols, swrap, slen, 0, srows,
                                           C.byref(found), C.byref(best), C.byref(val))

        r = {"python_mirror_us": round(wall(py_iter), 2), "kernel": pli.last_kernel}
        r["c_abi_us"] = round(wall(c_iter), 2)
        r["best_position"] = int(scores.offset(best.row, best.col))
        r["score_into_plus_sync_us"] = round(wall(c_score_sync), 2)
        r["sync_only_us"] = round(wall(lambda: L.lm_hip_ctx_sync(hc)), 2)
        r["fused_score_argmax_us"] = round(wall(c_fused), 2)
        r["fused_python_mirror_us"] = round(wall(lambda: pli.score_argmax(pssm, seq)), 2)
        r["fused_kernel"] = pli.last_kernel
        pli.set_track_argmax(False)
        r["untracked_two_launch_us"] = round(wall(c_iter), 2)
        pli.set_track_argmax(True)
        out[f"C{cols}"] = r
    print(json.dumps(out))


if __name__ == "__main__":
    main()
