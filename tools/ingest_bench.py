#!/usr/bin/env python3
"""Host sequence -> resident StripedSequence at genome scale: 1 Gbp of ASCII text / symbol bytes /
2-bit packed bases from pageable host memory, against the bare H2D copy of the same buffer (torch: hipMemcpy from
pageable memory) -- the floor of the step.  GPU box only:  python tools/ingest_bench.py [length]"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402


def best(fn, reps=4):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        del out
    return min(ts) * 1e3, float(np.median(ts)) * 1e3


def main():
    length = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
    torch.cuda.set_device(0)
    pli = lm.Pipeline.hip(0)
    rng = np.random.default_rng(1)
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    for a in range(0, length, length // 7 + 1):          # a genome's N: a few long runs (1 % here)
        enc[a:a + length // 700] = 4
    text = np.frombuffer(b"ACTGN", np.uint8)[enc]
    packed, mask = lm.pack_2bit(enc)
    _, runs = lm.pack_2bit(enc, runs=True)
    dev = torch.empty(length, dtype=torch.uint8, device="cuda")
    h2d = best(lambda: dev.copy_(torch.from_numpy(text)))
    dev4 = torch.empty(packed.size, dtype=torch.uint8, device="cuda")
    h2d4 = best(lambda: dev4.copy_(torch.from_numpy(packed)))
    out = {"length": length,
           "h2d_copy_ms": {"min": round(h2d[0], 2), "median": round(h2d[1], 2), "GBps": round(length / h2d[0] / 1e6, 1)},
           "h2d_copy_packed_ms": {"min": round(h2d4[0], 2), "median": round(h2d4[1], 2)}}
    for name, fn in (("ascii", lambda: pli.stripe_ascii(text)),
                     ("encoded", lambda: pli.stripe(lm.EncodedSequence(enc))),
                     ("two_bit_n_runs", lambda: pli.stripe_2bit(packed, length, None, n_runs=runs)),
                     ("two_bit_n_mask", lambda: pli.stripe_2bit(packed, length, mask))):
        t = best(fn)
        out[name + "_to_striped_ms"] = {"min": round(t[0], 2), "median": round(t[1], 2),
                                        "x_h2d_copy": round(t[0] / h2d[0], 3), "Gbp_per_s": round(length / t[0] / 1e6, 1)}
    # parity of the three forms on the whole genome
    a = pli.stripe_ascii(text)
    b = pli.stripe_2bit(packed, length, mask)
    c = pli.stripe_2bit(packed, length, None, n_runs=runs)
    ma = a.matrix()
    out["n_runs"] = int(len(runs))
    out["forms_agree"] = bool(np.array_equal(ma, b.matrix()) and np.array_equal(ma, c.matrix()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
