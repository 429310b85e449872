#!/usr/bin/env python3
"""Wall time per C-ABI call (median of N, inputs resident) next to the kernel time HIP events
see, to expose host-side overhead of the reductions / fused forms.  GPU box only:

    python tools/api_overhead.py [length] [motif_len]
"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402
from lightmotif_amd._ffi import Coords  # noqa: E402

COLS = 32


def main():
    length = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    reps = 60
    dev = torch.device("cuda", 0)
    pli = lm.Pipeline.hip()
    L = pli._L
    rng = np.random.default_rng(3)
    sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]
    pssm = lm.create(sites).counts.normalize(0.1).log_odds()
    rows = -(-length // COLS)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    seq = torch.randint(0, 4, (rows + m - 1, COLS), dtype=torch.uint8, device=dev, generator=gen)
    pli.configure_wrap_dptr(seq.data_ptr(), rows, COLS, COLS, m - 1, 4)
    scores = torch.empty((rows, COLS), dtype=torch.float32, device=dev)
    pli.score_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows,
                   scores.data_ptr(), COLS)
    torch.cuda.synchronize()
    sample = scores[: min(rows, 1 << 18)].flatten()
    out = {"length": length, "motif_len": m}
    h, p = pli._h, pssm._device(pli)
    sp, op = C.c_void_p(seq.data_ptr()), C.c_void_p(scores.data_ptr())
    found, best, value = C.c_int(0), Coords(), C.c_float(0)
    orow, mi = C.c_size_t(0), C.c_size_t(0)

    def med(fn):
        for _ in range(5):
            fn()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return round(float(np.median(ts)) * 1e3, 4)

    def score():
        L.lm_hip_score_f32_dptr(h, p, sp, rows + m - 1, COLS, COLS, m - 1, length, 0, rows, op, COLS,
                                C.byref(orow), C.byref(mi))
        torch.cuda.synchronize()

    out["score_ms"] = med(score)
    out["argmax_ms"] = med(lambda: L.lm_hip_argmax_f32_dptr(h, op, rows, COLS, COLS, C.byref(found),
                                                           C.byref(best), C.byref(value)))
    out["fused_argmax_ms"] = med(lambda: L.lm_hip_score_argmax_f32_dptr(
        h, p, sp, rows + m - 1, COLS, COLS, m - 1, length, 0, rows, C.byref(found), C.byref(best),
        C.byref(value)))
    for pv in (1e-5, 1e-4, 1e-3):
        t = float(torch.quantile(sample[torch.isfinite(sample)].float(), 1 - pv))
        n = C.c_size_t(0)

        def thr():
            ptr = C.POINTER(Coords)()
            L.lm_hip_threshold_f32_dptr(h, op, rows, COLS, COLS, C.c_float(t), C.byref(ptr), C.byref(n))
            L.lm_hip_free(ptr)

        def fthr():
            ptr, vals = C.POINTER(Coords)(), C.POINTER(C.c_float)()
            L.lm_hip_score_threshold_f32_dptr(h, p, sp, rows + m - 1, COLS, COLS, m - 1, length, 0, rows,
                                              C.c_float(t), C.byref(ptr), C.byref(vals), C.byref(n))
            L.lm_hip_free(ptr)
            L.lm_hip_free(vals)

        counts = []
        for flag in (1, 0):
            L.lm_hip_ctx_set_prefilter(h, flag)
            out[f"fused_threshold_p{pv:g}_prefilter{flag}_ms"] = med(fthr)
            counts.append(int(n.value))
        L.lm_hip_ctx_set_prefilter(h, 1)
        out[f"threshold_p{pv:g}_ms"] = med(thr)
        counts.append(int(n.value))
        out[f"hits_p{pv:g}"] = counts
        out[f"t_p{pv:g}"] = t
    print(json.dumps(out))


if __name__ == "__main__":
    main()
