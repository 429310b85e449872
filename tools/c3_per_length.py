#!/usr/bin/env python3
"""configs[2] by motif length: wall time of the fused threshold batch (p = 1e-5) over the JASPAR motifs of ONE length."""
import sys, time, collections
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import lightmotif_amd as lm  # noqa: E402
import bench_configs as bc  # noqa: E402
from lightmotif_amd import io as lmio  # noqa: E402
torch.cuda.set_device(0)
pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
pssms = [r.matrix.normalize(0.1).log_odds() for r in lmio.read(ROOT / "tests" / "golden" / "JASPAR2024.pwm.gz")]
length = 100_000_000
enc_seq, rows = bc.resident_sequence(pli, length, 5, max(len(p) for p in pssms) - 1, 33)
seq = pli.upload(enc_seq.cpu().numpy(), length, max(len(p) for p in pssms) - 1, 32)
by = collections.defaultdict(list)
for p in pssms:
    p._device(pli)
    by[len(p)].append(p)
for m in sorted(by):
    ps = by[m]
    ts = [p.score_for_pvalue(1e-5) for p in ps]
    res = pli.scan_threshold_batch(ps, ts, seq)
    tt = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); res = pli.scan_threshold_batch(ps, ts, seq); tt.append(time.perf_counter() - t0)
    t = float(np.median(tt))
    hits = sum(len(c) for c, _ in res)
    print(f"M={m:2d}: {len(ps):4d} motifs  {t*1e3:8.3f} ms  {t*1e6/len(ps):7.2f} us per motif  {hits:8d} hits  ({pli.last_kernel})")
