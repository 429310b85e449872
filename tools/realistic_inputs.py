#!/usr/bin/env python3
"""Non-i.i.d. inputs: how slow can a real genome make the fused scans?  

Every throughput number of bench.py is taken on i.i.d. uniform ACGT without N.  The fused paths' cost depends on the
candidate density: the discrete prefilter over-estimates, N scores as the row minimum, low-complexity tracts repeat
whatever k-mer they hold thousands of times.  This tool generates a seeded 100 Mbp sequence with the features of real
genomes that matter for that --

  * 5 % N, in runs (assembly gaps: log-normal lengths around 50 kb);
  * microsatellites and homopolymers: a tract every ~20 kb, unit of 1 ... 6 bases, 20 ... 600 bp long;
  * isochores: blocks of 100 kb at 35 % or 65 % GC

-- and runs, on it and on a uniform sequence of the same length: the JASPAR 2024 threshold batch (p = 1e-5 per motif) and
argmax batch (BASELINE configs[2]), one fused threshold scan of a length-20 motif at p = 1e-5, and Scanner::max.  Parity:
a sample of motifs is checked over the WHOLE realistic sequence against the AVX2 port of the oracle (hits, values, best
cell).  The CPU side of the comparison -- the reference's Scanner degrades on the same inputs -- is its block loop on the
AVX2 port over the first 2 Mbp of both sequences (u8 scores, candidates >= the discrete threshold, exact re-scoring).

    python tools/realistic_inputs.py [--length 100000000] [--json profiles/r05_realistic_inputs.json]
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

COLS = 32
NTHREADS = max(os.cpu_count() or 1, 1)
# symbol order of the Dna alphabet (abc.rs:138-160): A C T G N
A_, C_, T_, G_, N_ = 0, 1, 2, 3, 4


def realistic_dna(length: int, seed: int = 0x5EED0005, n_frac: float = 0.05, tract_every: int = 20_000,
                  block: int = 100_000) -> np.ndarray:
    """Encoded symbols (uint8, 0 ... 4) of a seeded sequence with isochores, low-complexity tracts and N runs."""
    rng = np.random.default_rng(seed)
    nblocks = -(-length // block)
    gc = np.where(rng.random(nblocks) < 0.5, 0.35, 0.65).astype(np.float32)
    out = np.empty(length, np.uint8)
    step = 1 << 24
    for lo in range(0, length, step):   # in pieces: 100 Mbp of float64 randoms at once would be 1.6 GB
        hi = min(lo + step, length)
        u = rng.random(hi - lo, dtype=np.float32)
        v = rng.integers(0, 2, hi - lo, dtype=np.uint8)
        is_gc = u < gc[np.arange(lo, hi) // block]
        out[lo:hi] = np.where(is_gc, np.where(v == 1, G_, C_), np.where(v == 1, T_, A_)).astype(np.uint8)
    # microsatellites / homopolymers
    ntr = max(length // tract_every, 1)
    starts = rng.integers(0, max(length - 700, 1), ntr)
    units = rng.integers(1, 7, ntr)
    totals = rng.integers(20, 601, ntr)
    for s, ul, tot in zip(starts, units, totals):
        unit = rng.integers(0, 4, ul, dtype=np.uint8)
        e = min(int(s + tot), length)
        out[s:e] = np.resize(unit, e - int(s))
    # N runs until the fraction is reached
    target, done = int(n_frac * length), 0
    while done < target:
        ln = int(min(max(rng.lognormal(np.log(50_000), 1.0), 100), target - done + 100, length))
        s = int(rng.integers(0, max(length - ln, 1)))
        out[s:s + ln] = N_
        done += ln
    return out


def uniform_dna(length: int, seed: int = 0x5EED0006) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 4, length, dtype=np.uint8)


def describe(enc: np.ndarray) -> dict:
    n = int((enc == N_).sum())
    gcn = int(((enc == C_) | (enc == G_)).sum())
    return {"length": int(enc.size), "n_fraction": round(n / enc.size, 4), "gc_fraction_of_acgt": round(gcn / max(enc.size - n, 1), 4)}


def timed(fn, reps, warm=1):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def cpu_scanner_block_loop(enc: np.ndarray, pssm, threshold: float) -> dict:
    """scan.rs:169-198 on the AVX2 port, one thread: per 256-row block u8 scores, max, candidates >= the discrete
    threshold, exact re-scoring of each candidate."""
    from oracle import c_oracle as co
    m = pssm.data.shape[0]
    dm = pssm.to_discrete()
    s = co.stripe(enc, COLS, 5)
    co.configure_wrap(s, m - 1)
    rows = s.rows
    w = co.aligned_empty((m, 32), np.uint8)
    w[:] = 0
    w[:, :dm.data.shape[1]] = dm.data
    t = dm.scale(threshold)
    out = co.aligned_empty((256, COLS), np.uint8)
    p = np.ascontiguousarray(pssm.data, np.float32)
    cands = hits = 0
    t0 = time.perf_counter()
    for r0 in range(0, rows, 256):
        r1 = min(r0 + 256, rows)
        co.avx2_score_rows_u8(s, w, out=out, row_begin=r0, row_end=r1)
        blk = out[: r1 - r0, :COLS]
        if blk.max(initial=0) >= t:
            rr, cc = np.nonzero(blk >= t)
            for r, c in zip(rr.tolist(), cc.tolist()):
                idx = c * rows + r0 + r
                if idx + m <= enc.size:
                    cands += 1
                    if co.score_position(s, p, idx) >= np.float32(threshold):
                        hits += 1
    return {"ms": round((time.perf_counter() - t0) * 1e3, 2), "candidates": cands, "hits": hits}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--length", type=int, default=100_000_000)
    ap.add_argument("--json", default="")
    ap.add_argument("--motifs", type=int, default=0, help="first N matrices only (0 = all 2 346)")
    ap.add_argument("--parity-motifs", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch

    import lightmotif_amd as lm
    from lightmotif_amd import io as lmio
    from oracle import c_oracle as co

    pli = lm.Pipeline.hip(0)   # fails loudly without a device
    pssms = [r.matrix.normalize(0.1).log_odds() for r in lmio.read(ROOT / "tests" / "golden" / "JASPAR2024.pwm.gz")]
    if a.motifs:
        pssms = pssms[:a.motifs]
    ts = [p.score_for_pvalue(1e-5) for p in pssms]
    rng = np.random.default_rng(20)
    m20 = lm.create(["".join("ACTG"[i] for i in rng.integers(0, 4, 20)) for _ in range(10)]).counts.normalize(0.1).log_odds()
    t20 = m20.score_for_pvalue(1e-5)
    report = {"what": __doc__.split("\n\n")[0], "motifs": len(pssms), "inputs": {}}
    for name, enc in (("uniform", uniform_dna(a.length)), ("realistic", realistic_dna(a.length))):
        seq = pli.stripe(lm.EncodedSequence(enc))
        seq.configure_wrap(max(len(p) for p in pssms + [m20]) - 1)   # wrap rows for the longest motif serve every shorter one
        r = {"sequence": describe(enc)}
        res = [None]

        def thr_batch():
            res[0] = pli.scan_threshold_batch(pssms, ts, seq)
        r["jaspar_threshold_batch_ms"] = round(timed(thr_batch, a.reps, warm=2), 3)
        hits, cands = pli.last_scan_counts
        r["jaspar_threshold_hits"], r["jaspar_threshold_candidate_pieces"] = hits, cands
        r["candidate_pieces_per_hit"] = round(cands / max(hits, 1), 2)
        am = [None]

        def am_batch():
            am[0] = pli.scan_argmax_batch(pssms, seq)
        r["jaspar_argmax_batch_ms"] = round(timed(am_batch, a.reps, warm=2), 3)
        one = [None]

        def thr_one():
            one[0] = pli.score_threshold(m20, seq, t20)
        r["fused_threshold_m20_ms"] = round(timed(thr_one, 10, warm=3), 4)
        h1, c1 = pli.last_scan_counts
        r["fused_threshold_m20_hits"], r["fused_threshold_m20_candidate_pieces"] = h1, c1
        sc = lm.Scanner(m20, seq, threshold=t20)
        best = [None]

        def smax():
            best[0] = lm.Scanner(m20, seq, threshold=t20).max()
        r["scanner_max_m20_ms"] = round(timed(smax, 5, warm=2), 4)
        del sc
        # parity over the whole sequence for a sample of motifs, against the AVX2 port of the oracle
        if name == "realistic":
            s = co.stripe(enc, COLS, 5)
            checked = []
            pick = np.linspace(0, len(pssms) - 1, a.parity_motifs).astype(int).tolist()
            for i in pick:
                p = pssms[i]
                m = p.data.shape[0]
                co.configure_wrap(s, m - 1)
                pa = co.aligned_empty(p.data.shape, np.float32)
                pa[:] = p.data
                want = co.avx2_score_rows(s, pa, threads=NTHREADS)
                rc = co.threshold(want, COLS, ts[i])
                g_rc, g_val = res[0][i]
                ok = np.array_equal(np.asarray(g_rc), rc) and np.array_equal(
                    np.asarray(g_val, np.float32).view(np.uint32), want[rc[:, 0], rc[:, 1]].view(np.uint32))
                ok_am = am[0][i] is not None and am[0][i][0] == co.argmax(want, COLS)
                checked.append({"motif": i, "m": m, "hits": int(len(rc)), "threshold_matches_oracle": bool(ok),
                                "argmax_matches_oracle": bool(ok_am)})
            r["parity_vs_avx2_port_whole_sequence"] = checked
            co.configure_wrap(s, 19)
            pa = co.aligned_empty(m20.data.shape, np.float32)
            pa[:] = m20.data
            want = co.avx2_score_rows(s, pa, threads=NTHREADS)
            rc = co.threshold(want, COLS, t20)
            r["fused_threshold_m20_matches_oracle"] = bool(np.array_equal(np.asarray(one[0][0]), [tuple(x) for x in rc.tolist()])
                                                           if len(rc) else len(one[0][0]) == 0)
        r["cpu_scanner_first_2mbp_avx2_port_1_thread"] = cpu_scanner_block_loop(enc[:2_000_000], m20, t20)
        report["inputs"][name] = r
        del seq
        torch.cuda.empty_cache()
    u, q = report["inputs"]["uniform"], report["inputs"]["realistic"]
    report["realistic_over_uniform"] = {k: round(q[k] / u[k], 3) for k in
                                        ("jaspar_threshold_batch_ms", "jaspar_argmax_batch_ms", "fused_threshold_m20_ms", "scanner_max_m20_ms")}
    report["realistic_over_uniform"]["cpu_scanner_block_loop"] = round(
        q["cpu_scanner_first_2mbp_avx2_port_1_thread"]["ms"] / u["cpu_scanner_first_2mbp_avx2_port_1_thread"]["ms"], 3)
    text = json.dumps(report, indent=1)
    if a.json:
        Path(a.json).write_text(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
