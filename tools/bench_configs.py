#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json (not the bench.py headline line):

  c1  MX000001-style len-15 PSSM over a 464 165 bp stand-in for the first tenth of E. coli
      (lightmotif-bench/dna.rs:81-109 harness: score_into + argmax per iteration; the real
      ecoli.txt is absent from the reference mount, so `best == 391677` cannot be checked)
  c3  2 346 DNA PSSMs with JASPAR 2024 CORE's length histogram (SURVEY 8d [probe]) over a
      100 Mbp resident sequence: fused argmax and fused threshold (p ~ 1e-5) per motif
  c5  protein (K = 21) len-12 PSSM over 200 Mres: score() materialised

Prints one JSON object per configuration.  Run on a GPU box:  python tools/bench_configs.py
"""
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
JASPAR_HIST = {4: 21, 5: 48, 6: 284, 7: 347, 8: 454, 9: 280, 10: 281, 11: 157, 12: 97, 13: 99, 14: 85,
               15: 72, 16: 40, 17: 21, 18: 14, 19: 19, 20: 8, 21: 11, 22: 1, 24: 2, 29: 2, 30: 1,
               31: 1, 33: 1}


def resident_sequence(pli, length, k, wrap, seed):
    dev = torch.device("cuda", 0)
    rows = -(-length // COLS)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    seq = torch.empty((rows + wrap, COLS), dtype=torch.uint8, device=dev)
    seq[:rows] = torch.randint(0, k - 1, (rows, COLS), dtype=torch.uint8, device=dev, generator=gen)
    if length < rows * COLS:
        idx = torch.arange(length, rows * COLS, device=dev)
        seq[idx % rows, idx // rows] = k - 1
    pli.configure_wrap_dptr(seq.data_ptr(), rows, COLS, COLS, wrap, k - 1)
    torch.cuda.synchronize()
    return seq, rows


def motif(rng, m, protein=False, nsites=10):
    sym = lm.lib.PROTEIN_SYMBOLS[:-1] if protein else "ACTG"
    sites = ["".join(sym[i] for i in rng.integers(0, len(sym), m)) for _ in range(nsites)]
    return lm.create(sites, protein=protein).counts.normalize(0.1).log_odds()


def timeit(fn, reps):
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def config1(pli):
    length, m = 464_165, 15
    rng = np.random.default_rng(1)
    pssm = lm.create(["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"]).counts.normalize(0.1).log_odds()
    seq, rows = resident_sequence(pli, length, 5, m - 1, 11)
    out = torch.empty((rows, COLS), dtype=torch.float32, device=seq.device)

    def it():  # dna.rs:104-107: score_into + argmax
        pli.score_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, out.data_ptr(), COLS)
        return pli.argmax_dptr(out.data_ptr(), rows, COLS, COLS)

    def fused():
        return pli.score_argmax_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows)

    assert it() == fused()
    t, tf = timeit(it, 200), timeit(fused, 200)
    return {"config": "c1: len-15 PSSM x 464165 bp (E. coli/10 stand-in), score_into + argmax per iteration",
            "us_per_iter": round(t * 1e6, 2), "Mpos_per_s": round(length / t / 1e6, 1),
            "fused_us_per_iter": round(tf * 1e6, 2), "fused_Mpos_per_s": round(length / tf / 1e6, 1),
            "note": "latency-bound: two launches + one 16-byte read-back per iteration"}


def config3(pli):
    length = 100_000_000
    rng = np.random.default_rng(3)
    fixture = ROOT / "tests" / "golden" / "JASPAR2024.pwm.gz"
    if fixture.exists():
        # the reference's own bench fixture (lightmotif-io/benches/JASPAR2024.pwm), converted like
        # the CLI does: pseudocount 0.1, uniform background (lightmotif-cli main.rs:473-478)
        from lightmotif_amd import io as lmio
        pssms = [r.matrix.normalize(0.1).log_odds() for r in lmio.read(fixture)]
        lengths = [len(p) for p in pssms]
        source = "JASPAR 2024 CORE matrices"
    else:
        lengths = [m for m, c in sorted(JASPAR_HIST.items()) for _ in range(c)]
        rng.shuffle(lengths)
        pssms = [motif(rng, m) for m in lengths]
        source = "synthetic motifs with JASPAR 2024 CORE's length histogram"
    enc_seq, rows = resident_sequence(pli, length, 5, max(lengths) - 1, 33)
    seq = pli.upload_from_device(enc_seq, length, max(lengths) - 1) if hasattr(pli, "upload_from_device") else None
    if seq is None:
        host = enc_seq.cpu().numpy()
        seq = pli.upload(host, length, max(lengths) - 1, COLS)
    for p in pssms:  # device tables
        p._device(pli)
    t_am = timeit(lambda: pli.scan_argmax_batch(pssms, seq), 3)
    # per-motif score threshold for p = 1e-5, the CLI default (lightmotif-cli main.rs:487-498),
    # from the MEME-style score distribution (lightmotif_amd/dist.py <- pwm/dist.rs); motifs too
    # short to reach 1e-5 get a threshold above their maximum, i.e. no hits
    ts = [p.score_for_pvalue(1e-5) for p in pssms]
    res = pli.scan_threshold_batch(pssms, ts, seq)
    t_th = timeit(lambda: pli.scan_threshold_batch(pssms, ts, seq), 3)
    cells = len(pssms) * rows * COLS
    return {"config": f"c3: {len(pssms)} DNA PSSMs ({source}, sum M = {sum(lengths)}) x 100 Mbp resident, "
                      "thresholds at p = 1e-5 per motif",
            "fused_argmax_s": round(t_am, 4), "fused_argmax_Gcell_per_s": round(cells / t_am / 1e9, 1),
            "fused_threshold_s": round(t_th, 4), "fused_threshold_Gcell_per_s": round(cells / t_th / 1e9, 1),
            "threshold_hits_total": int(sum(len(r[0]) for r in res)),
            "note": "wall time per batched call from Python; cells = sum over motifs of the positions each is "
                    "scored at, so Gcell/s is an EQUIVALENT rate: the sequence (100 MB) stays in L2 / Infinity "
                    "Cache, motifs up to length 9 are settled by the argmax from the last rows of the range, the "
                    "others take the packed-u16 pair scan, 2-4 motifs per pass, with exact re-scoring of the "
                    "candidates (DESIGN 4.1b, 4.1c, 4.6); the scans are instruction-issue bound, no HBM "
                    "fraction applies"}


def config5(pli):
    length, m = 200_000_000, 12
    rng = np.random.default_rng(5)
    pssm = motif(rng, m, protein=True, nsites=6)
    seq, rows = resident_sequence(pli, length, 21, m - 1, 55)
    out = torch.empty((rows, COLS), dtype=torch.float32, device=seq.device)

    def it():
        pli.score_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, out.data_ptr(), COLS)

    for _ in range(20):
        it()
    t = timeit(it, 50)
    store_kernel = pli.last_kernel

    def fused():
        return pli.score_argmax_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows)

    assert fused() == pli.argmax_dptr(out.data_ptr(), rows, COLS, COLS)
    tf = timeit(fused, 30)
    sample = out[: 1 << 18].flatten()
    thr = float(torch.quantile(sample[torch.isfinite(sample)].float(), 1 - 1e-5))
    th = {}
    for on in (True, False):   # u16 prefilter (pair scan over 441 rows, or LM_HIP_PAIR_PREFILTER=0: 21 rows) vs exact f32
        pli.set_prefilter(on)
        call = lambda: pli.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, thr)  # noqa: E731
        call()
        th[on] = (timeit(call, 20), pli.last_kernel, len(call()[0]))
    pli.set_prefilter(True)
    assert th[True][2] == th[False][2]
    return {"config": "c5: protein (K=21) len-12 PSSM x 200 Mres, score() materialised", "kernel": store_kernel,
            "fused_threshold_prefilter_ms": round(th[True][0] * 1e3, 4), "fused_threshold_prefilter_kernel": th[True][1],
            "fused_threshold_exact_ms": round(th[False][0] * 1e3, 4), "fused_threshold_hits": th[True][2],
            "ms": round(t * 1e3, 4), "Gpos_per_s": round(rows * COLS / t / 1e9, 1),
            "GBps": round(5 * rows * COLS / t / 1e9, 1), "hbm_frac": round(5 * rows * COLS / t / 8e12, 4),
            "fused_argmax_ms": round(tf * 1e3, 4), "fused_argmax_Gpos_per_s": round(rows * COLS / tf / 1e9, 1),
            "fused_argmax_Tlookup_per_s": round(m * rows * COLS / tf / 1e12, 2)}


def config_layout(pli):
    """encode + stripe + configure_wrap on the device (byte work, HBM-bound:
    1 B read + 1 B written per symbol for each of encode and stripe)."""
    dev = torch.device("cuda", 0)
    length = 1_000_000_000
    rows = -(-length // COLS)
    ascii_ = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[
        torch.randint(0, 4, (length,), device=dev, dtype=torch.int64)]
    enc = torch.empty(length, dtype=torch.uint8, device=dev)
    data = torch.empty((rows + 19, COLS), dtype=torch.uint8, device=dev)
    t_enc = timeit(lambda: pli.encode_dptr(ascii_.data_ptr(), length, enc.data_ptr()), 5)
    t_str = timeit(lambda: pli.stripe_dptr(enc.data_ptr(), length, COLS, 4, 19, data.data_ptr(), COLS), 5)
    want = enc.view(COLS, rows).t()
    assert torch.equal(data[:rows], want)
    return {"config": "layout: encode + stripe(+wrap 19) of 1 Gbp on the device",
            "encode_ms": round(t_enc * 1e3, 3), "encode_GBps": round(2 * length / t_enc / 1e9, 1),
            "stripe_ms": round(t_str * 1e3, 3), "stripe_GBps": round(2 * length / t_str / 1e9, 1),
            "note": "algorithmic traffic 2 B per symbol each; HBM-bound byte kernels"}


def config_u8(pli):
    """SURVEY 8a `score_u8_*` / 8(f) rank 2: `Score<u8, ..>` with a DiscreteMatrix plus the u8
    reductions the Scanner runs on it (1 B read + 1 B written per cell; 1 B read per cell)."""
    length, m = 1_000_000_000, 20
    seq, rows = resident_sequence(pli, length, 5, m - 1, 11)
    dm = motif(np.random.default_rng(2), m).to_discrete()
    out = torch.empty((rows, COLS), dtype=torch.uint8, device=seq.device)

    def score():
        pli.score_u8_dptr(dm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows,
                          out.data_ptr(), COLS)

    for _ in range(3):
        score()
    t_sc = timeit(score, 20)
    kernel = pli.last_kernel
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:  # the pipeline runs on torch's current stream here
        a.record()
        score()
        b.record()
    torch.cuda.synchronize()
    t_ev = float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e-3
    t_am = timeit(lambda: pli.argmax_u8_dptr(out.data_ptr(), rows, COLS, COLS), 10)
    t = int(out[: 1 << 20].max()) - 2
    t_th = timeit(lambda: pli.threshold_u8_dptr(out.data_ptr(), rows, COLS, COLS, t), 10)
    nh = pli.threshold_u8_dptr(out.data_ptr(), rows, COLS, COLS, t).shape[0]
    return {"config": "u8: DiscreteMatrix (M = 20) x 1 Gbp, Score<u8> materialised + Maximum<u8> + Threshold<u8>",
            "kernel": kernel, "score_call_ms": round(t_sc * 1e3, 4), "score_kernel_ms": round(t_ev * 1e3, 4),
            "score_Gpos_per_s": round(length / t_ev / 1e9, 1),
            "score_GBps": round(2 * length / t_ev / 1e9, 1), "score_hbm_frac": round(2 * length / t_ev / 8e12, 3),
            "argmax_ms": round(t_am * 1e3, 4), "argmax_GBps": round(length / t_am / 1e9, 1),
            "threshold_ms": round(t_th * 1e3, 4), "threshold_hits": int(nh),
            "note": "score_kernel_ms from HIP events on the launch stream, the other times are wall time per call incl. launch + synchronisation; algorithmic traffic 2 B / 1 B per cell"}


def config_shapes(pli):
    """Beyond the unrolled C = 32 kernels: long motifs (sliced in place) and other column counts
    (score_tiled).  100 Mbp each; C = 1 is the Generic bench geometry of dna.rs:113-116, bound by 32-byte
    rows (64 B of traffic per position)."""
    out = {"config": "shapes: score() beyond C = 32 / M <= 36, 100 Mbp each", "rows": []}
    rng = np.random.default_rng(9)
    for cols, m, length in ((32, 40, 100_000_000), (32, 64, 100_000_000), (32, 100, 100_000_000), (16, 20, 100_000_000),
                            (16, 33, 100_000_000), (1, 15, 20_000_000)):
        enc = rng.integers(0, 4, length, dtype=np.uint8)
        seq = pli.stripe(lm.EncodedSequence(enc), cols)
        pssm = motif(rng, m)
        seq.configure(pssm)
        scores = lm.StripedScores.empty(pli, cols)
        for _ in range(3):
            pli.score_into(pssm, seq, scores)
        t = timeit(lambda: pli.score_into(pssm, seq, scores), 10)
        row = {"C": cols, "M": m, "kernel": pli.last_kernel, "ms": round(t * 1e3, 3),
               "Gpos_per_s": round(length / t / 1e9, 1)}
        pli.score_argmax(pssm, seq)
        tf = timeit(lambda: pli.score_argmax(pssm, seq), 5)      # fused argmax, call wall incl. read-back
        row.update({"fused_argmax_kernel": pli.last_kernel, "fused_argmax_ms": round(tf * 1e3, 3),
                    "fused_argmax_Gpos_per_s": round(length / tf / 1e9, 1)})
        out["rows"].append(row)
        del seq, scores
    return out


if __name__ == "__main__":
    torch.cuda.set_device(0)
    pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
    which = sys.argv[1:] or ["c1", "c5", "layout", "u8", "shapes", "c3"]
    for name in which:
        print(json.dumps({"c1": config1, "c3": config3, "c5": config5, "layout": config_layout, "u8": config_u8,
                          "shapes": config_shapes}[name](pli)), flush=True)
