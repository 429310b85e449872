#!/usr/bin/env python3
"""Stress of the polled small-input paths (per-wavefront records in pinned memory, folded by the host): many
iterations alternating between inputs with DIFFERENT answers on ONE scores handle, every result checked -- a stale or
torn record would show as a wrong cell.  GPU box only:  python tools/stress_small_argmax.py [seconds]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
pli = lm.Pipeline.hip(0)
rng = np.random.default_rng(99)
cases = []
for cols, length, m in ((32, 464_165, 15), (32, 50_000, 20), (32, 1_000_003, 8), (1, 60_000, 15), (16, 200_000, 12), (32, 3_000_000, 37)):
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    p = np.zeros((m, 8), np.float32)
    p[:, :4] = rng.normal(0, 2, (m, 4))
    p[:, 4] = -np.inf
    pssm = lm.ScoringMatrix(p)
    seq = pli.stripe(lm.EncodedSequence(enc), cols)
    seq.configure(pssm)
    scores = lm.StripedScores.empty(pli, cols)
    pli.set_track_argmax(False)
    pli.score_into(pssm, seq, scores)
    want = (pli.argmax(scores), pli.max(scores))              # two-launch reference on the stored matrix
    want_fused = pli.score_argmax(pssm, seq)
    pli.set_track_argmax(True)
    assert want_fused == want, (want_fused, want)
    cases.append((cols, pssm, seq, want))
handles = {32: lm.StripedScores.empty(pli, 32), 1: lm.StripedScores.empty(pli, 1), 16: lm.StripedScores.empty(pli, 16)}
n = bad = 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < seconds:
    for cols, pssm, seq, want in cases:
        s = handles[cols]
        pli.score_into(pssm, seq, s)
        got = (pli.argmax(s), pli.max(s))
        fused = pli.score_argmax(pssm, seq)
        n += 1
        if got != want or fused != want:
            bad += 1
            if bad < 5:
                print("MISMATCH", cols, got, fused, want)
print(f"{n} iterations x (tracked + fused), {bad} mismatches")
sys.exit(1 if bad else 0)
