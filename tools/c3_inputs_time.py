#!/usr/bin/env python3
"""The JASPAR batch (configs[2], prepared motif list) on the uniform and on the non-i.i.d. 100 Mbp (tools/realistic_inputs.py):
call time and the library's phases, one JSON line.  For A/B runs of library builds on one box
(LM_HIP_LIBRARY=<variant> python tools/c3_inputs_time.py).  GPU box only."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import bench  # noqa: E402
import lightmotif_amd as lm  # noqa: E402
import realistic_inputs as ri  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
st = bench.c3_setup(pli, dev, 1, 0, 100_000_000, 0)
batch = pli.prepare_batch(st["pssms"], st["ts"])
rseq = pli.stripe(lm.EncodedSequence(ri.realistic_dna(100_000_000)))
rseq.configure_wrap(st["max_m"] - 1)
out = {"library": os.environ.get("LM_HIP_LIBRARY", "shipped")}
for name, seq in (("uniform", st["seq"]), ("realistic", rseq)):
    for _ in range(3):
        res = pli.scan_threshold_batch(batch, None, seq)
    ts = []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = pli.scan_threshold_batch(batch, None, seq)
        ts.append((time.perf_counter() - t0) * 1e3)
    pli.set_option("time_scan", 1)
    ph = []
    for _ in range(3):
        pli.scan_threshold_batch(batch, None, seq)
        ph.append(pli.last_phases_ms)
    pli.set_option("time_scan", 0)
    am = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pli.scan_argmax_batch(st["pssms"], seq)
        am.append((time.perf_counter() - t0) * 1e3)
    out[name] = {"threshold_ms": round(float(np.median(ts)), 3), "scan_ms": round(float(np.median([p[0] for p in ph])), 3),
                 "argmax_ms": round(float(np.median(am[1:])), 3), "hits": int(res.counts.sum())}
print(json.dumps(out), flush=True)
