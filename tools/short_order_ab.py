#!/usr/bin/env python3
"""A/B of the short hit-list ordering (csrc/hits.hip, option "short_order") on one box: fused threshold calls of one
length-20 motif at p = 1e-5 over 1 Gbp ... 1 Mbp, interleaved, median of the per-call wall times.

    python tools/short_order_ab.py [--json profiles/r05_short_order_ab.json]"""
import argparse, json, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import lightmotif_amd as lm
from bench_configs import motif, resident_sequence

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--json", default=""); a = ap.parse_args()
    torch.cuda.set_device(0)
    m = 20
    plis = {}
    for name, v in (("five_launches", 0), ("short_form", 1)):
        p = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream); p.set_option("short_order", v); plis[name] = p
    pssm = motif(np.random.default_rng(m), m)
    out = {"what": __doc__.split("\n\n")[0], "lengths": {}}
    for length in (1_000_000_000, 100_000_000, 10_000_000, 4_600_000, 1_000_000):
        seq, rows = resident_sequence(plis["short_form"], length, 5, m - 1, 11)
        for pv in (1e-5, 1e-4):
            thr = pssm.score_for_pvalue(pv)
            call = lambda p: p.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, 32, 32, m - 1, length, 0, rows, thr)
            ref = None; ts = {k: [] for k in plis}
            for rep in range(60):
                for k, p in plis.items():
                    t0 = time.perf_counter(); h = call(p); dt = time.perf_counter() - t0
                    if rep >= 10: ts[k].append(dt)
                    if ref is None: ref = h
                    assert np.array_equal(np.asarray(h[0]), np.asarray(ref[0])) and np.array_equal(np.asarray(h[1], np.float32).view(np.uint32), np.asarray(ref[1], np.float32).view(np.uint32))
            r = {k: round(float(np.median(v)) * 1e6, 1) for k, v in ts.items()}
            r["hits"] = int(len(ref[0])); r["ratio"] = round(r["short_form"] / r["five_launches"], 3)
            am = {}
            for k in ("short_form",):
                p = plis[k]; tt = []
                for rep in range(40):
                    t0 = time.perf_counter(); p.score_argmax_dptr(pssm, seq.data_ptr(), rows + m - 1, 32, 32, m - 1, length, 0, rows); tt.append(time.perf_counter() - t0)
                am[k] = round(float(np.median(tt[8:])) * 1e6, 1)
            r["fused_argmax_us"] = am
            out["lengths"][f"{length}@{pv:g}"] = r
            print(length, pv, r, flush=True)
        del seq
    if a.json: Path(a.json).write_text(json.dumps(out, indent=1) + "\n")
main()
