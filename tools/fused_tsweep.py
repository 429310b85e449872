#!/usr/bin/env python3
"""Fused threshold / argmax scans (pair prefilter): call time by stream length (lm_hip_ctx_set_rows_per_stream; 0 = the
planner's choice), 1 Gbp, interleaved rounds:  python tools/fused_tsweep.py 20 0,98,146,242,482,962,1922"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import lightmotif_amd as lm  # noqa: E402
from bench_configs import motif, resident_sequence  # noqa: E402

ms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20").split(",")]
targets = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,98,146,242,482,962,1922").split(",")]
length = 1_000_000_000
torch.cuda.set_device(0)
stream = torch.cuda.current_stream()
pli = lm.Pipeline.hip(0, stream=stream.cuda_stream)
mmax = max(ms)
seq, rows = resident_sequence(pli, length, 5, mmax - 1, 11)
out = []
for m in ms:
    pssm = motif(np.random.default_rng(m), m)
    thr = pssm.score_for_pvalue(1e-5)
    thr_ms = {t: [] for t in targets}
    am_ms = {t: [] for t in targets}
    for r in range(-2, 7):
        for t in targets:
            pli.set_rows_per_stream(t)
            for store, fn in ((thr_ms, lambda: pli.score_threshold_dptr(pssm, seq.data_ptr(), rows + mmax - 1, 32, 32, mmax - 1, length, 0, rows, thr)),
                              (am_ms, lambda: pli.score_argmax_dptr(pssm, seq.data_ptr(), rows + mmax - 1, 32, 32, mmax - 1, length, 0, rows))):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                if r >= 0:
                    store[t].append((time.perf_counter() - t0) / 5 * 1e3)
    out.append({"m": m, "kernel": pli.last_kernel,
                "fused_threshold_ms_by_rows_per_stream": {str(t): round(float(np.median(v)), 4) for t, v in thr_ms.items()},
                "fused_argmax_ms_by_rows_per_stream": {str(t): round(float(np.median(v)), 4) for t, v in am_ms.items()}})
    print(json.dumps(out[-1]), flush=True)
