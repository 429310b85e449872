#!/usr/bin/env python3
"""The reference's own flow on resident handles: pli.score_into(pssm, seq, scores) then scores.argmax()
(lightmotif-bench dna.rs:104-107), at the harness size and at 1 Gbp.  GPU box only."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402


def run(length, m, reps):
    pli = lm.Pipeline.hip()
    rng = np.random.default_rng(1)
    sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]
    pssm = lm.create(sites).counts.normalize(0.1).log_odds()
    enc = torch.randint(0, 4, (length,), dtype=torch.uint8, device="cuda")
    rows = -(-length // 32)
    data = torch.empty((rows + m - 1, 32), dtype=torch.uint8, device="cuda")
    pli.stripe_dptr(enc.data_ptr(), length, 32, 4, m - 1, data.data_ptr(), 32)
    torch.cuda.synchronize()
    seq = pli.upload(data.cpu().numpy(), length, m - 1, 32)
    scores = lm.StripedScores.empty(pli, 32)
    for _ in range(5):
        pli.score_into(pssm, seq, scores)
        best = pli.argmax(scores)
    ts, ta, tb = [], [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pli.score_into(pssm, seq, scores)
        pli.sync()
        t1 = time.perf_counter()
        best = pli.argmax(scores)
        t2 = time.perf_counter()
        ts.append(t1 - t0); ta.append(t2 - t1); tb.append(t2 - t0)
    raw = pli.argmax_dptr(scores.data_ptr, scores.rows, 32, 32)
    assert raw[0] == best, (raw, best)
    return {"length": length, "motif_len": m, "score_into_ms": round(float(np.median(ts)) * 1e3, 4),
            "argmax_ms": round(float(np.median(ta)) * 1e3, 4), "both_ms": round(float(np.median(tb)) * 1e3, 4)}


if __name__ == "__main__":
    print(json.dumps([run(464_165, 15, 200), run(1_000_000_000, 20, 30)]))
