"""Non-i.i.d. sequences: 5 % N in runs, microsatellites / homopolymers, 35 % / 65 % GC isochores
(tools/realistic_inputs.realistic_dna).  The fused paths' COST depends on candidate density; their RESULTS must not:
every route -- batched fused threshold / argmax, single fused scans, Scanner, Score<u8> -- against the oracle over the
whole sequence, bit for bit.  Plus the round's two diagnostics (clock bracket, scan counts)."""
import sys
from pathlib import Path

import numpy as np
import pytest

import lightmotif_amd as lm
from lightmotif_amd import io as lmio
from oracle import c_oracle as co

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import realistic_inputs as ri  # noqa: E402

gpu = pytest.mark.gpu
COLS = 32
FIXTURE = Path(__file__).parent / "golden" / "JASPAR2024.pwm.gz"


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def oracle_scores(ref, pssm):
    co.configure_wrap(ref, pssm.data.shape[0] - 1)
    p = co.aligned_empty(pssm.data.shape, np.float32)
    p[:] = pssm.data
    return co.avx2_score_rows(ref, p, threads=8)


@pytest.fixture(scope="module")
def realistic():
    enc = ri.realistic_dna(6_000_011, seed=0xBEEF, tract_every=5_000, block=50_000)
    d = ri.describe(enc)
    assert 0.04 < d["n_fraction"] < 0.07
    return enc


def test_generator_is_seeded_and_has_the_features_it_names():
    a, b = ri.realistic_dna(300_000, seed=5), ri.realistic_dna(300_000, seed=5)
    assert np.array_equal(a, b) and not np.array_equal(a, ri.realistic_dna(300_000, seed=6))
    assert a.max() == 4 and (a == 4).mean() > 0.02                     # N, in runs
    edges = np.diff(np.r_[0, (a == 4).astype(np.int8), 0])
    runs = np.flatnonzero(edges == -1) - np.flatnonzero(edges == 1)
    assert runs.max() >= 100
    c = ri.realistic_dna(1_000_000, seed=7, block=20_000).reshape(-1, 20_000)
    acgt = (c != 4).sum(axis=1)
    gc = (((c == 1) | (c == 3)).sum(axis=1) / np.maximum(acgt, 1))[acgt > 10_000]
    assert gc.min() < 0.40 and gc.max() > 0.60                        # isochores at 35 % and 65 % GC


@gpu
def test_fused_batch_on_a_realistic_sequence_matches_the_oracle(pli, realistic):
    enc = realistic
    pssms = [r.matrix.normalize(0.1).log_odds() for r in lmio.read(FIXTURE)]
    pick = [pssms[i] for i in np.linspace(0, len(pssms) - 1, 40).astype(int)]
    ts = [p.score_for_pvalue(1e-4) for p in pick]
    seq = pli.stripe(lm.EncodedSequence(enc))
    seq.configure_wrap(max(len(p) for p in pick) - 1)
    thr = pli.scan_threshold_batch(pick, ts, seq)
    hits, cands = pli.last_scan_counts
    assert hits == sum(len(c) for c, _ in thr) and cands >= 0
    am = pli.scan_argmax_batch(pick, seq)
    ref = co.stripe(enc, COLS, 5)
    nonempty = 0
    for p, t, (g_rc, g_val), g_am in zip(pick, ts, thr, am):
        want = oracle_scores(ref, p)
        rc = co.threshold(want, COLS, t)
        assert np.array_equal(np.asarray(g_rc).reshape(-1, 2), rc.reshape(-1, 2)), len(p)
        assert np.array_equal(bits(g_val), bits(want[rc[:, 0], rc[:, 1]]))
        assert g_am is not None and g_am[0] == co.argmax(want, COLS) and bits(g_am[1]) == bits(want[g_am[0]])
        nonempty += len(rc) > 0
    assert nonempty >= 5


@gpu
@pytest.mark.parametrize("m", [8, 15, 20, 33])
def test_single_fused_scans_scanner_and_u8_on_a_realistic_sequence(pli, realistic, m):
    enc = realistic
    rng = np.random.default_rng(m)
    pssm = lm.create(["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(8)]).counts.normalize(0.1).log_odds()
    seq = pli.stripe(lm.EncodedSequence(enc))
    seq.configure(pssm)
    ref = co.stripe(enc, COLS, 5)
    want = oracle_scores(ref, pssm)
    flat = want[:, :COLS]
    t = float(np.partition(flat.ravel(), flat.size - 400)[flat.size - 400])
    rc = [tuple(x) for x in co.threshold(want, COLS, t).tolist()]
    g_rc, g_val = pli.score_threshold(pssm, seq, t)
    assert pli.last_kernel.startswith("score_c32_prefilter2") and g_rc == rc
    assert np.array_equal(bits(g_val), bits([want[r, c] for r, c in rc]))
    assert pli.score_argmax(pssm, seq)[0] == co.argmax(want, COLS)
    # Scanner: every hit by position, and max()
    by_pos = flat.T.reshape(-1)[: len(enc) - m + 1]
    want_pos = np.nonzero(by_pos >= np.float32(t))[0]
    sc = lm.Scanner(pssm, seq, threshold=t)
    assert sc.positions.tolist() == want_pos.tolist()
    best = lm.Scanner(pssm, seq, threshold=t).max()
    assert best is not None and np.float32(best.score) == by_pos.max()
    # Score<u8> with the DiscreteMatrix, through the pair kernel
    dm = pssm.to_discrete()
    u8, _ = pli.score_discrete(dm, seq, saturate=False)       # Generic: wrapping `+=` (pli/mod.rs:98-102)
    want_u8, _ = co.score_rows_u8(ref, dm.data)
    assert np.array_equal(u8[:, :COLS], want_u8[:, :COLS])
    sat, _ = pli.score_discrete(dm, seq)                      # the SIMD tiers: saturating adds (avx2.rs:336)
    w = co.aligned_empty((m, 32), np.uint8)
    w[:] = 0
    w[:, :dm.data.shape[1]] = dm.data
    assert np.array_equal(sat[:, :COLS], co.avx2_score_rows_u8(ref, w)[:, :COLS])


@gpu
def test_clock_probe_and_scan_counts(pli):
    rng = np.random.default_rng(3)
    enc = rng.integers(0, 4, 8_000_000, dtype=np.uint8)
    pssm = lm.create(["".join("ACTG"[i] for i in rng.integers(0, 4, 20)) for _ in range(8)]).counts.normalize(0.1).log_odds()
    seq = pli.stripe(lm.EncodedSequence(enc))
    seq.configure(pssm)
    scores = lm.StripedScores.empty(pli, COLS)
    mhz, beside_ms, alone_ms = pli.sustained_clock_mhz(lambda: pli.score_into(pssm, seq, scores), 0.2)
    assert mhz is not None and 300 < mhz < 2600 and beside_ms > 0 and alone_ms > 0      # a shader clock
    t = pssm.score_for_pvalue(1e-4)
    rc, _ = pli.score_threshold(pssm, seq, t)
    hits, cands = pli.last_scan_counts
    assert hits == len(rc) and cands * 32 >= hits
    # the scan kernel's own duration: only when asked for, and then for both fused calls
    assert pli.last_scan_kernel_ms is None
    pli.set_option("time_scan", 1)
    try:
        import time
        t0 = time.perf_counter()
        rc2, _ = pli.score_threshold(pssm, seq, t)
        wall_ms = (time.perf_counter() - t0) * 1e3
        k_thr = pli.last_scan_kernel_ms
        # ... the call by phase (scan | re-scoring | ordering on the stream, the rest on the host clock), and what the scan
        # kernel that ran looked up: a length-20 motif as 19 or 20 rows of a pair table of ((rows | 3) + 1) bytes per position
        ph = pli.last_phases_ms
        assert ph is not None and ph[0] == k_thr and ph[1] >= 0 and ph[2] >= 0 and 0 <= ph[3] < wall_ms
        assert sum(ph) <= wall_ms + 0.05   # (the first timed call also creates the events: its wall time is far above the phases)
        srows, sbytes = pli.last_scan_info
        assert srows in (19, 20) and sbytes == (srows | 3) + 1
        best = pli.score_argmax(pssm, seq)
        k_am = pli.last_scan_kernel_ms if pli.last_kernel.startswith("score_c32_prefilter") else 0.001
    finally:
        pli.set_option("time_scan", 0)
    assert rc2 == rc and best is not None and 0 < k_thr < wall_ms and k_am > 0
    pli.score_threshold(pssm, seq, t)
    assert pli.last_scan_kernel_ms is None and pli.last_phases_ms is None
