#!/usr/bin/env python3
"""The round-5 protein fault in the REAL library: fused threshold of protein motifs through the one-symbol prefilter against
the exact kernel, repeated; prints hits and candidate pieces per call.  Select a library variant with LM_HIP_LIBRARY
(tools/build_variant.py ringraw --short -DLM_RING_LOOKAHEAD_RAW = the round-4 look-ahead of MP - 1).
    python tools/ring_real.py [residues = 50e6] [M ...]"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402

COLS = 32
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
length = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
ms = [int(x) for x in sys.argv[2:]] or [7, 8, 9, 12]
mmax = 36
rows = -(-length // COLS)
gen = torch.Generator(device=dev)
gen.manual_seed(76)
seq = torch.empty((rows + mmax - 1, COLS), dtype=torch.uint8, device=dev)
seq[:rows] = torch.randint(0, 20, (rows, COLS), dtype=torch.uint8, device=dev, generator=gen)


def pipeline(**options):
    p = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
    for k, v in options.items():
        p.set_option(k, v)
    return p


single = pipeline(pair_prefilter=0, pair_prefilter_protein=0)
exact = pipeline(prefilter=0)
exact.configure_wrap_dptr(seq.data_ptr(), rows, COLS, COLS, mmax - 1, 20)
sym = lm.lib.PROTEIN_SYMBOLS[:-1]
for m in ms:
    prng = np.random.default_rng(21000 + m)
    sites = ["".join(sym[i] for i in prng.integers(0, len(sym), m)) for _ in range(6)]
    pssm = lm.create(sites, protein=True).counts.normalize(0.1).log_odds()
    t = pssm.score_for_pvalue(1e-4)
    args = (pssm, seq.data_ptr(), rows + mmax - 1, COLS, COLS, mmax - 1, length, 0, rows, t)
    want = exact.score_threshold_dptr(*args)
    runs = []
    for _ in range(6):
        got = single.score_threshold_dptr(*args)
        runs.append({"hits": len(got[0]), "same": bool(np.array_equal(got[0], want[0])), "counts": list(single.last_scan_counts)})
    print(json.dumps({"m": m, "library": lm._ffi.lib_path() if hasattr(lm._ffi, "lib_path") else "", "kernel": single.last_kernel,
                      "exact_hits": len(want[0]), "runs": runs}), flush=True)
