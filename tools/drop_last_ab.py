#!/usr/bin/env python3
"""A/B of the drop-last form of a single pair scan (context option "drop_last"; csrc/score_threshold.hip) on one box: fused
threshold (p = 1e-5) and fused argmax calls over 1 Gbp for M = 8 ... 36, interleaved; per call the wall time, the scan
kernel's own time (option "time_scan") and the candidate pieces the scan flagged.

    python tools/drop_last_ab.py [--json profiles/r05_drop_last_ab.json]"""
import argparse, json, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import lightmotif_amd as lm
from bench_configs import motif, resident_sequence


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--json", default=""); ap.add_argument("--length", type=int, default=1_000_000_000)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    plis = {}
    for name, v in (("all_rows", 0), ("drop_last", 1)):
        p = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream); p.set_option("drop_last", v); p.set_option("time_scan", 1)
        plis[name] = p
    length = a.length
    seq, rows = resident_sequence(plis["all_rows"], length, 5, 35, 11)
    out = {"what": __doc__.split("\n\n")[0], "length": length, "by_m": {}}
    for m in (8, 12, 16, 20, 24, 28, 32, 36):
        pssm = motif(np.random.default_rng(m), m); thr = pssm.score_for_pvalue(1e-5)
        r = {}
        for kind in ("threshold", "argmax"):
            call = (lambda p: p.score_threshold_dptr(pssm, seq.data_ptr(), rows + 35, 32, 32, 35, length, 0, rows, thr)) if kind == "threshold" \
                else (lambda p: p.score_argmax_dptr(pssm, seq.data_ptr(), rows + 35, 32, 32, 35, length, 0, rows))
            ref = None; w = {k: [] for k in plis}; kk = {k: [] for k in plis}; cands = {}
            for rep in range(70):
                for k, p in plis.items():
                    t0 = time.perf_counter(); h = call(p); dt = time.perf_counter() - t0
                    if rep >= 30:
                        w[k].append(dt); kk[k].append(p.last_scan_kernel_ms or 0.0)
                    if kind == "threshold":
                        cands[k] = p.last_scan_counts[1]
                        if ref is None: ref = h
                        assert np.array_equal(np.asarray(h[0]), np.asarray(ref[0]))
                    else:
                        if ref is None: ref = h
                        assert h == ref
            r[kind] = {k: {"call_us": round(float(np.median(w[k])) * 1e6, 1), "scan_kernel_us": round(float(np.median(kk[k])) * 1e3, 1),
                           **({"candidate_pieces": int(cands[k]), "hits": int(len(ref[0]))} if kind == "threshold" else {})} for k in plis}
            r[kind]["call_ratio"] = round(r[kind]["drop_last"]["call_us"] / r[kind]["all_rows"]["call_us"], 3)
        out["by_m"][m] = r
        print(m, json.dumps(r), flush=True)
    if a.json: Path(a.json).write_text(json.dumps(out, indent=1) + "\n")
main()
