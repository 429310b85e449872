#!/usr/bin/env python3
"""A/B in one process: the JASPAR threshold batch with long hit lists ordered by radix sort (default) or by the bucket
passes (option "sort_hits" = 0), interleaved; uniform and non-i.i.d. 100 Mbp.  python tools/sort_ab.py"""
import json
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
c3 = bench.c3_setup(pli, dev, 1, 0, 100_000_000)
pssms, ts = c3["pssms"], c3["ts"]
rseq = pli.stripe(lm.EncodedSequence(ri.realistic_dna(100_000_000)))
rseq.configure_wrap(c3["max_m"] - 1)
out = {}
for name, seq in (("uniform", c3["seq"]), ("realistic", rseq)):
    times = {0: [], 1: []}
    for rep in range(-1, 5):
        for flag in (1, 0):
            pli.set_option("sort_hits", flag)
            pli.scan_threshold_batch(pssms, ts, seq)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pli.scan_threshold_batch(pssms, ts, seq)
            torch.cuda.synchronize()
            if rep >= 0:
                times[flag].append((time.perf_counter() - t0) * 1e3)
    out[name] = {"radix_sort_ms": round(float(np.median(times[1])), 3), "bucket_passes_ms": round(float(np.median(times[0])), 3)}
print(json.dumps(out))
