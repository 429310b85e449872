#!/usr/bin/env python3
"""Where a single fused threshold call spends its time (lm_hip_ctx_last_phases_ms): scan kernel | exact re-scoring |
ordering kernels (HIP events on the library's stream) | the host's share, next to the whole call on the wall clock --
configs[1] (1 Gbp x M = 20 DNA) and configs[4] (200 Mres x M = 12 protein), p = 1e-5.  GPU box only.
    python tools/fused_phases.py"""
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
pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)


def report(tag, pssm, seq, rows, m, length):
    t = pssm.score_for_pvalue(1e-5)
    call = lambda: pli.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, t)  # noqa: E731
    for _ in range(60):
        out = call()
    ts = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = call()
        ts.append((time.perf_counter() - t0) * 1e6)
    ph = {}
    k = B.scan_kernel_ms(pli, call, phases=ph)
    print(json.dumps({"case": tag, "kernel": pli.last_kernel, "hits": len(out[0]), "scan_info": pli.last_scan_info, "call_us_min": round(min(ts), 1),
                      "call_us_median": round(float(np.median(ts)), 1), "scan_kernel_us": round(k * 1e3, 1), **ph}), flush=True)


length, m = 1_000_000_000, 20
rows = -(-length // COLS)
shard = B.synth_shard(rows, 0, rows, length, m - 1, dev)
report("c2 dna 1 Gbp x M = 20", B.synth_pssm(m), shard, rows, m, length)
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
pli.configure_wrap_dptr(pseq.data_ptr(), rows, COLS, COLS, m - 1, 20)
report("c5 protein 200 Mres x M = 12", ppssm, pseq, rows, m, length)
