#!/usr/bin/env python3
"""Kernel timeline of ONE fused threshold call (and one fused argmax call) at 1 Gbp x M = 20:
start offset and duration of every kernel, to see where the call's wall time goes beyond
the scan.  GPU box only; run under rocprofv3 by tools/timeline_threshold.sh.

    python tools/timeline_threshold.py [pvalue]
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402
from lightmotif_amd._ffi import Coords  # noqa: E402

COLS = 32


def main():
    pv = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
    length, m = 1_000_000_000, 20
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
    n_s = 1 << 18
    scores = torch.empty((n_s, COLS), dtype=torch.float32, device=dev)
    pli.score_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, n_s,
                   scores.data_ptr(), COLS)
    torch.cuda.synchronize()
    t = float(torch.quantile(scores.flatten()[: 1 << 23], 1 - pv))
    h, p, sp = pli._h, pssm._device(pli), C.c_void_p(seq.data_ptr())
    n = C.c_size_t(0)
    for _ in range(6):
        ptr, vals = C.POINTER(Coords)(), C.POINTER(C.c_float)()
        L.lm_hip_score_threshold_f32_dptr(h, p, sp, rows + m - 1, COLS, COLS, m - 1, length, 0, rows,
                                          C.c_float(t), C.byref(ptr), C.byref(vals), C.byref(n))
        L.lm_hip_free(ptr)
        L.lm_hip_free(vals)
    found, best, value = C.c_int(0), Coords(), C.c_float(0)
    for _ in range(6):
        L.lm_hip_score_argmax_f32_dptr(h, p, sp, rows + m - 1, COLS, COLS, m - 1, length, 0, rows,
                                       C.byref(found), C.byref(best), C.byref(value))
    print("hits", n.value)


if __name__ == "__main__":
    main()
