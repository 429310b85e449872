#!/usr/bin/env python3
"""Score<u8> store (score_c32_u8_pairs): kernel time by stream length (lm_hip_ctx_set_rows_per_stream; 0 = the planner's
default), 1 Gbp, interleaved rounds:  python tools/u8_tsweep.py 20 0,50,74,98,146,242,482,962"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import lightmotif_amd as lm  # noqa: E402
from bench_configs import motif, resident_sequence  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
targets = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,50,74,98,146,242,482,962").split(",")]
length = 1_000_000_000
torch.cuda.set_device(0)
stream = torch.cuda.current_stream()
pli = lm.Pipeline.hip(0, stream=stream.cuda_stream)
seq, rows = resident_sequence(pli, length, 5, m - 1, 11)
dm = motif(np.random.default_rng(2), m).to_discrete()
out = torch.empty((rows, 32), dtype=torch.uint8, device=seq.device)
times = {t: [] for t in targets}
for r in range(-2, 8):
    for t in targets:
        pli.set_rows_per_stream(t)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(3):
            pli.score_u8_dptr(dm, seq.data_ptr(), rows + m - 1, 32, 32, m - 1, length, 0, rows, out.data_ptr(), 32)
        b.record(stream)
        torch.cuda.synchronize()
        if r >= 0:
            times[t].append(a.elapsed_time(b) / 3)
res = {"m": m, "kernel": pli.last_kernel,
       "ms_by_rows_per_stream": {str(t): round(float(np.median(v)), 4) for t, v in times.items()},
       "hbm_frac_by_rows_per_stream": {str(t): round(2e9 / (float(np.median(v)) * 1e-3) / 8e12, 4) for t, v in times.items()}}
print(json.dumps(res))
