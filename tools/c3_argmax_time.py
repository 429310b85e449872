#!/usr/bin/env python3
"""Wall time of the fused argmax batch of configs[2] (median of N calls)."""
import sys, time
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
for p in pssms:
    p._device(pli)
ts = []
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter(); res = pli.scan_argmax_batch(pssms, seq); ts.append((time.perf_counter() - t0) * 1e3)
import hashlib
h = hashlib.sha1(repr(res).encode()).hexdigest()[:12]
print(f"argmax batch: median {np.median(ts[2:]):.3f} ms  min {min(ts):.3f} ms  result digest {h}")
