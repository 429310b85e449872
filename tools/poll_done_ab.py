#!/usr/bin/env python3
"""Context option "poll_done" off and on, interleaved on one box: single fused threshold calls (p = 1e-5) at 1 Gbp x M = 20 DNA,
200 Mres x M = 12 protein and 1 Mbp x M = 20 -- wall time per call (minimum and median of 40 after 60 warm-up calls per side).
GPU box only.    python tools/poll_done_ab.py"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench as B  # noqa: E402
import lightmotif_amd as lm  # noqa: E402

COLS = 32
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
plis = {}
for on in (0, 1):
    p = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
    p.set_option("poll_done", on)
    plis[on] = p


def case(tag, pssm, seq, rows, m, length, k):
    t = pssm.score_for_pvalue(1e-5)
    calls = {on: (lambda p=plis[on]: p.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, t)) for on in plis}
    want = None
    for on in plis:
        for _ in range(60):
            got = calls[on]()
        sig = (len(got[0]), int(np.asarray(got[0], dtype=np.int64).sum()))
        assert want is None or sig == want, (sig, want)
        want = sig
    ts = {0: [], 1: []}
    for r in range(40):
        for on in ((0, 1) if r % 2 == 0 else (1, 0)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            calls[on]()
            ts[on].append((time.perf_counter() - t0) * 1e6)
    print(json.dumps({"case": tag, "hits": want[0], **{f"poll_done_{on}": {"min_us": round(min(ts[on]), 1), "median_us": round(float(np.median(ts[on])), 1)} for on in ts}}), flush=True)


for length in (1_000_000_000, 1_000_000):
    m = 20
    rows = -(-length // COLS)
    shard = B.synth_shard(rows, 0, rows, length, m - 1, dev)
    case(f"dna {length} x M = 20", B.synth_pssm(m), shard, rows, m, length, 5)
    del shard
length, m = 200_000_000, 12
rows = -(-length // COLS)
prng = np.random.default_rng(5)
sym = lm.lib.PROTEIN_SYMBOLS[:-1]
ppssm = lm.create(["".join(sym[i] for i in prng.integers(0, len(sym), m)) for _ in range(6)], protein=True).counts.normalize(0.1).log_odds()
gen = torch.Generator(device=dev)
gen.manual_seed(55)
pseq = torch.empty((rows + m - 1, COLS), dtype=torch.uint8, device=dev)
pseq[:rows] = torch.randint(0, 20, (rows, COLS), dtype=torch.uint8, device=dev, generator=gen)
plis[0].configure_wrap_dptr(pseq.data_ptr(), rows, COLS, COLS, m - 1, 20)
case("protein 200 Mres x M = 12", ppssm, pseq, rows, m, length, 21)
