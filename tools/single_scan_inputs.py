#!/usr/bin/env python3
"""Single fused threshold scans (p = 1e-5) and the Score<u8> pair store by motif length on a uniform ACGT sequence and on the
non-i.i.d. one with N (tools/realistic_inputs.py), 200 Mbp each: call time and scan-kernel time.  For A/B runs of library
builds (LM_HIP_LIBRARY=<variant> python tools/single_scan_inputs.py 12,20,28,36).  GPU box only."""
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
import bench as B  # noqa: E402
import lightmotif_amd as lm  # noqa: E402
import realistic_inputs as ri  # noqa: E402

lengths = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "12,20,28,36").split(",")]
n = 200_000_000
torch.cuda.set_device(0)
pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(11)
seqs = {"uniform": rng.integers(0, 4, n, dtype=np.uint8), "with_N": ri.realistic_dna(n)}
out = {"library": os.environ.get("LM_HIP_LIBRARY", "shipped"), "length": n}
for name, enc in seqs.items():
    seq = pli.stripe(lm.EncodedSequence(enc))
    seq.configure_wrap(max(lengths) - 1)
    torch.cuda.synchronize()
    for m in lengths:
        pssm = B.synth_pssm(m)
        t = pssm.score_for_pvalue(1e-5)
        call = lambda: pli.score_threshold(pssm, seq, t)  # noqa: E731
        for _ in range(20):
            got = call()
        ts = []
        for _ in range(15):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = call()
            ts.append((time.perf_counter() - t0) * 1e3)
        k = B.scan_kernel_ms(pli, call)
        out.setdefault(name, {})[str(m)] = {"call_ms": round(float(np.median(ts)), 4), "scan_kernel_ms": round(k, 4), "hits": len(got[0]),
                                             "kernel": pli.last_kernel}
    del seq
print(json.dumps(out), flush=True)
