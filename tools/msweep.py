#!/usr/bin/env python3
"""M-sweep of the materialising store kernel, the exact fused kernel and the prefilter scan
(VERDICT r1 item 3): DNA, 1 Gbp, M = 8, 15, 20, 28, 33, 36 (+ 12 / 24), kernel time by HIP events
on the launch stream, each with its HBM fraction (5 B/pos store, 1 B/pos fused; 8 TB/s spec) and
its LDS-gather fraction (4*M B/pos -- 2*M' for the u16 prefilter, M' for the pair table -- against
256 B/clk/CU x 256 CUs x 2.4 GHz).  GPU box only:

    python tools/msweep.py [length] > profiles/r02_msweep.json
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402

COLS = 32
HBM = 8.0e12
LDS = 256 * 256 * 2.4e9


def events_ms(fn, stream, reps=30, warm=10):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(stream)
        fn()
        b.record(stream)
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2], t[0]


def main():
    length = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
    ms_list = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 12, 15, 20, 24, 28, 33, 36]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    pli = lm.Pipeline.hip(0, stream=stream.cuda_stream)
    rows = -(-length // COLS)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    mmax = max(ms_list)
    seq = torch.empty((rows + mmax - 1, COLS), dtype=torch.uint8, device=dev)
    seq[:rows] = torch.randint(0, 4, (rows, COLS), dtype=torch.uint8, device=dev, generator=gen)
    pli.configure_wrap_dptr(seq.data_ptr(), rows, COLS, COLS, mmax - 1, 4)
    out = torch.empty((rows, COLS), dtype=torch.float32, device=dev)
    for _ in range(200):   # clocks
        pli.score_dptr(lm.ScoringMatrix(np.zeros((8, 8), np.float32)), seq.data_ptr(), rows + mmax - 1, COLS, COLS,
                       mmax - 1, length, 0, rows, out.data_ptr(), COLS)
    torch.cuda.synchronize()
    res = {"length": length, "rows": rows, "hbm_peak": HBM, "lds_peak_bytes_per_s": LDS, "sweep": []}
    for m in ms_list:
        rng = np.random.default_rng(m)
        sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]
        pssm = lm.create(sites).counts.normalize(0.1).log_odds()
        args = (pssm, seq.data_ptr(), rows + mmax - 1, COLS, COLS, mmax - 1, length, 0, rows)
        pos = rows * COLS
        rec = {"M": m}
        t, tmin = events_ms(lambda: pli.score_dptr(*args, out.data_ptr(), COLS), stream)
        rec["store"] = {"kernel": pli.last_kernel, "ms": round(t, 4), "ms_min": round(tmin, 4),
                        "Gpos_s": round(pos / t / 1e6, 1), "hbm_frac": round(5 * pos / (t * 1e-3) / HBM, 4),
                        "lds_frac": round(4 * m * pos / (t * 1e-3) / LDS, 4)}
        if len(sys.argv) > 3:      # stream-length sweep of the store kernel: rows per stream (0 = the planner's default)
            rec["store_rows_per_stream"] = {}
            for rps in [int(x) for x in sys.argv[3].split(",")]:
                pli.set_rows_per_stream(rps)
                t2, _ = events_ms(lambda: pli.score_dptr(*args, out.data_ptr(), COLS), stream, reps=15, warm=5)
                rec["store_rows_per_stream"][str(rps)] = round(t2, 4)
            pli.set_rows_per_stream(0)
        sample = out[: 1 << 18].flatten()
        thr = float(torch.quantile(sample[torch.isfinite(sample)].float(), 1 - 1e-5))
        for name, pre in (("fused_threshold_prefilter", True), ("fused_threshold_exact", False)):
            pli.set_prefilter(pre)
            fn = lambda: pli.score_threshold_dptr(*args, thr)   # noqa: E731  (call wall incl. read-back)
            import time
            for _ in range(5):
                fn()
            ts = []
            for _ in range(15):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                got = fn()
                ts.append((time.perf_counter() - t0) * 1e3)
            tc = float(np.median(ts))
            kern = pli.last_kernel
            nb = 1
            mp = m
            if pre and "prefilter2" in kern:
                mp = m + (3 - m % 4) % 4          # padded to 3 mod 4; one (M'+1) x u16 row per TWO input rows
                ldsb = (mp + 1) * 2 / 2
            elif pre and "prefilter" in kern:
                mp = m + m % 2
                ldsb = 2 * mp
            else:
                ldsb = 4 * m
            # (the threshold is a sample quantile of the scores: short motifs have few distinct scores, ties make their lists long)
            rec[name] = {"kernel": kern, "call_ms": round(tc, 4), "hits": int(len(got[0])), "Gpos_s": round(pos / tc / 1e6, 1),
                         "hbm_frac_1B": round(nb * pos / (tc * 1e-3) / HBM, 4),
                         "lds_bytes_per_pos": ldsb, "lds_frac": round(ldsb * pos / (tc * 1e-3) / LDS, 4)}
        pli.set_prefilter(True)
        ts = []
        for it in range(12):     # fused argmax, call wall incl. the read-back of the record
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pli.score_argmax_dptr(*args)
            if it >= 4:
                ts.append((time.perf_counter() - t0) * 1e3)
        tc = float(np.median(ts))
        rec["fused_argmax"] = {"kernel": pli.last_kernel, "call_ms": round(tc, 4), "Gpos_s": round(pos / tc / 1e6, 1)}
        res["sweep"].append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
