#!/usr/bin/env python3
"""Protein (K = 21) fused scans, interleaved in one process over 200 Mres: the one-symbol prefilter on 4-row symbol blocks
(default, score_prefilter_blk.hpp), the same with a byte load per lane and row (option "block_prefilter" = 0: the round-5
kernel) and the 441-row pair scan (option "pair_prefilter_protein" = 1).  Call times (median of 5 x 5) and the scan kernel
alone (option "time_scan").  python tools/protein_pair_ab.py [M ...]"""
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
        for flag in ("blocks", "bytes", "pairs"):
            pli.set_option("pair_prefilter_protein", int(flag == "pairs"))
            pli.set_option("block_prefilter", int(flag != "bytes"))
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
                if rep == 4:   # the scan kernel alone, events around it on the library's stream
                    pli.set_option("time_scan", 1)
                    ks = []
                    for _ in range(5):
                        fn()
                        ks.append(pli.last_scan_kernel_ms or 0.0)
                    pli.set_option("time_scan", 0)
                    res[(name, flag, "kernel_ms")] = round(float(np.median(ks)), 4)
    forms = ("blocks", "bytes", "pairs")
    print(json.dumps({"m": m, **{f"{n}_{f}_ms": round(float(np.median(v)), 4) for (n, f, *r), v in res.items() if not r},
                      **{f"{n}_{f}_scan_kernel_ms": res[(n, f, "kernel_ms")] for n in ("threshold", "argmax") for f in forms},
                      "kernels": {f"{n}_{f}": res[(n, f, "kernel")] for n in ("threshold", "argmax") for f in forms},
                      "hits": res[("threshold", "blocks", "n")],
                      "same_results": all(res[("threshold", f, "n")] == res[("threshold", "blocks", "n")] and
                                          res[("argmax", f, "n")] == res[("argmax", "blocks", "n")] for f in forms)}), flush=True)
