#!/usr/bin/env python3
"""Same-process A/B of score_into with and without the tracked maximum (1 Gbp, M = 20): blocks of launches
alternate between the two, HIP events on the launch stream around every score_into (so the tracked form
includes argmax_fold + argmax_finalize_locate).  GPU box only:  python tools/track_ab.py [M]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402
import bench  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
length = 1_000_000_000
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
pli = lm.Pipeline.hip(0, stream=stream.cuda_stream)
rows = -(-length // 32)
shard = bench.synth_shard(rows, 0, rows, length, m - 1, dev)
pli.configure_wrap_dptr(shard.data_ptr(), rows, 32, 32, m - 1, 4)
seq = pli.adopt_sequence(shard.data_ptr(), rows, m - 1, 32, 32, length, keepalive=shard)
rng = np.random.default_rng(2)
pssm = lm.create(["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]).counts.normalize(0.1).log_odds()
scores = lm.StripedScores.empty(pli, 32)
for _ in range(300):
    pli.score_into(pssm, seq, scores)
torch.cuda.synchronize()
times = {False: [], True: []}
for rnd in range(12):
    for track in (False, True):
        pli.set_track_argmax(track)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(25)]
        for a, b in ev:
            a.record(stream)
            pli.score_into(pssm, seq, scores)
            b.record(stream)
        torch.cuda.synchronize()
        if rnd >= 2:
            times[track] += [a.elapsed_time(b) for a, b in ev[5:]]
for track, v in times.items():
    print(f"M={m} track_argmax={int(track)}: median {np.median(v):.4f} ms  mean {np.mean(v):.4f}  min {min(v):.4f}  (n={len(v)})")
