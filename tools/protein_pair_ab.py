#!/usr/bin/env python3
"""Protein (K = 21) fused scans: the one-symbol prefilter (default) against the 441-row pair scan (option
"pair_prefilter_protein" = 1), interleaved in one process; 200 Mres.  python tools/protein_pair_ab.py [M ...]"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402

COLS = 32
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
length = 200_000_000
rows = -(-length // COLS)
ms_list = [int(x) for x in sys.argv[1:]] or [8, 12, 20]
mmax = max(ms_list)
gen = torch.Generator(device=dev)
gen.manual_seed(55)
pseq = torch.empty((rows + mmax - 1, COLS), dtype=torch.uint8, device=dev)
pseq[:rows] = torch.randint(0, 20, (rows, COLS), dtype=torch.uint8, device=dev, generator=gen)
pli.configure_wrap_dptr(pseq.data_ptr(), rows, COLS, COLS, mmax - 1, 20)
sym = lm.lib.PROTEIN_SYMBOLS[:-1]
for m in ms_list:
    prng = np.random.default_rng(m)
    sites = ["".join(sym[i] for i in prng.integers(0, len(sym), m)) for _ in range(6)]
    pssm = lm.create(sites, protein=True).counts.normalize(0.1).log_odds()
    thr = pssm.score_for_pvalue(1e-5)
    res = {}
    for rep in range(-1, 5):
        for flag in (0, 1):
            pli.set_option("pair_prefilter_protein", flag)
            for name, fn in (("threshold", lambda: pli.score_threshold_dptr(pssm, pseq.data_ptr(), rows + mmax - 1, COLS, COLS, mmax - 1, length, 0, rows, thr)),
                             ("argmax", lambda: pli.score_argmax_dptr(pssm, pseq.data_ptr(), rows + mmax - 1, COLS, COLS, mmax - 1, length, 0, rows))):
                out = fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    out = fn()
                torch.cuda.synchronize()
                if rep >= 0:
                    res.setdefault((name, flag), []).append((time.perf_counter() - t0) / 5 * 1e3)
                res[(name, flag, "kernel")] = pli.last_kernel
                res[(name, flag, "n")] = len(out[0]) if name == "threshold" else out[0]
    print(json.dumps({"m": m, **{f"{n}_{'pair' if f else 'single'}_ms": round(float(np.median(v)), 4) for (n, f, *r), v in res.items() if not r},
                      "kernels": {f"{n}_{'pair' if f else 'single'}": res[(n, f, "kernel")] for n in ("threshold", "argmax") for f in (0, 1)},
                      "same_results": res[("threshold", 0, "n")] == res[("threshold", 1, "n")] and res[("argmax", 0, "n")] == res[("argmax", 1, "n")]}), flush=True)
