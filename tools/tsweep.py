#!/usr/bin/env python3
"""Same-process sweep of the store kernel's stream length (lm_hip_ctx_set_rows_per_stream) per motif
length, interleaved over rounds:  python tools/tsweep.py 9,15,21,25 24,32,48,64,96,128,192"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402

ms = [int(x) for x in sys.argv[1].split(",")]
targets = [int(x) for x in sys.argv[2].split(",")]
length = 1_000_000_000
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
pli = lm.Pipeline.hip(0, stream=stream.cuda_stream)
rows = -(-length // 32)
mmax = max(ms)
seq = torch.empty((rows + mmax - 1, 32), dtype=torch.uint8, device=dev)
seq[:rows] = torch.randint(0, 4, (rows, 32), dtype=torch.uint8, device=dev)
pli.configure_wrap_dptr(seq.data_ptr(), rows, 32, 32, mmax - 1, 4)
out = torch.empty((rows, 32), dtype=torch.float32, device=dev)
for m in ms:
    rng = np.random.default_rng(m)
    pssm = lm.create(["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]).counts.normalize(0.1).log_odds()
    cfgs = [0] + targets
    times = {t: [] for t in cfgs}
    for r in range(-3, 15):
        for t in cfgs:
            pli.set_rows_per_stream(t)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                a.record(stream)
                pli.score_dptr(pssm, seq.data_ptr(), rows + mmax - 1, 32, 32, mmax - 1, length, 0, rows, out.data_ptr(), 32)
                b.record(stream)
            torch.cuda.synchronize()
            if r >= 0:
                times[t].append(a.elapsed_time(b))
    print(f"M={m}: " + "  ".join(f"{'default' if t == 0 else t}:{np.median(v):.4f}" for t, v in times.items()), flush=True)
