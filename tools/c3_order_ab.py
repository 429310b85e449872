#!/usr/bin/env python3
"""configs[2] (the JASPAR batch, prepared once) with context option "order_groups" off and on, interleaved in one process:
wall time of the call and the library's own phases.  GPU box only.
    python tools/c3_order_ab.py [rounds]"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import bench_configs as bc  # noqa: E402
import lightmotif_amd as lm  # noqa: E402
from lightmotif_amd import io as lmio  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
torch.cuda.set_device(0)
pssms = [r.matrix.normalize(0.1).log_odds() for r in lmio.read(ROOT / "tests" / "golden" / "JASPAR2024.pwm.gz")]
ts = [p.score_for_pvalue(1e-5) for p in pssms]
length = 100_000_000
wrap = max(len(p) for p in pssms) - 1
pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
pli.set_option("time_scan", 1)
enc_seq, rows = bc.resident_sequence(pli, length, 5, wrap, 33)
seq = pli.upload(enc_seq.cpu().numpy(), length, wrap, 32)
batch = pli.prepare_batch(pssms, ts)
want = None
for on in (0, 1, 0, 1):   # warm both orders, same hits
    pli.set_option("order_groups", on)
    res = pli.scan_threshold_batch(batch, None, seq)
    sig = (sum(len(c) for c, _ in res), int(sum(int(np.asarray(c, dtype=np.uint64).sum() & 0xFFFFFFFF) for c, _ in res)))
    assert want is None or sig == want, (sig, want)
    want = sig
times = {0: [], 1: []}
phases = {0: [], 1: []}
for r in range(rounds):
    for on in (0, 1) if r % 2 == 0 else (1, 0):
        pli.set_option("order_groups", on)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pli.scan_threshold_batch(batch, None, seq)
        times[on].append((time.perf_counter() - t0) * 1e3)
        phases[on].append(pli.last_phases_ms)
for on in (0, 1):
    ph = np.median(np.asarray(phases[on]), axis=0)
    print(json.dumps({"order_groups": on, "call_ms_median": round(float(np.median(times[on])), 3), "call_ms_min": round(min(times[on]), 3),
                      "scan_ms": round(float(ph[0]), 3), "rescore_ms": round(float(ph[1]), 3), "order_ms": round(float(ph[2]), 3),
                      "host_ms": round(float(ph[3]), 3), "hits": want[0]}), flush=True)
