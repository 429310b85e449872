#!/usr/bin/env python3
"""`Scanner::max` with the reference's own walk (scan.rs:200-249) on the device (csrc/scanmax.hip): wall time per call
on a resident 1 Gbp sequence, M = 20, at the CLI's p = 1e-5 threshold and at a threshold every cell passes, next to
`max_valid()` (fused argmax + one scan).  GPU box only."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import lightmotif_amd as lm  # noqa: E402

length = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
m = 20
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
pli = lm.Pipeline.hip(0)
rows = -(-length // 32)
shard = bench.synth_shard(rows, 0, rows, length, m - 1, dev)
pli.configure_wrap_dptr(shard.data_ptr(), rows, 32, 32, m - 1, 4)
torch.cuda.synchronize()
seq = pli.adopt_sequence(shard.data_ptr(), rows, m - 1, 32, 32, length, keepalive=shard)
pssm = bench.synth_pssm(m)
out = {"length": length, "M": m}
for name, t in (("p1e-5", pssm.score_for_pvalue(1e-5)), ("every_cell", -1e30)):
    rec = {}
    for label, call in (("max_reference_walk", lambda: lm.Scanner(pssm, seq, threshold=t).max()),
                        ("max_valid", lambda: lm.Scanner(pssm, seq, threshold=t).max_valid())):
        call()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            hit = call()
            ts.append(time.perf_counter() - t0)
        rec[label + "_ms"] = round(min(ts) * 1e3, 3)
        rec[label] = None if hit is None else [hit.position, round(hit.score, 4)]
    out[name] = rec
print(json.dumps(out))
