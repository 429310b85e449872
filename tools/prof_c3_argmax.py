#!/usr/bin/env python3
"""Only the fused argmax batch of configs[2] (2 346 JASPAR matrices x 100 Mbp), N calls -- for rocprofv3:
   cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d out -o t -- python $ROOT/tools/prof_c3_argmax.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n):
    res = pli.scan_argmax_batch(pssms, seq)
print(len(res), "motifs", res[:2])
