"""Seeded differential test: random shapes, motif lengths, row ranges, stream lengths and
thresholds through every entry point of the path, GPU vs the CPU oracle, bit for bit."""
import numpy as np
import pytest

import lightmotif_amd as lm
from host_walk import scanner_max_strict_host
from oracle import c_oracle as co
from oracle import np_oracle as no

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


import os

SEEDS = range(int(os.environ.get("LM_FUZZ_FIRST", "0")), int(os.environ.get("LM_FUZZ_LAST", "240")))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_configuration(pli, seed):
    rng = np.random.default_rng(90_000 + seed)
    protein = seed % 7 == 3
    k = 21 if protein else 5
    m = int(rng.integers(1, 41))
    if seed % 11 == 5:                      # sliced motifs: store, continuation and chunked reductions
        m = int(rng.integers(37, 130))
    length = int(rng.choice([m, m + 1, 33, 1000, 4097, 20_000, 131_072, 250_001]))
    length = max(length, 1)
    cols = 32 if seed % 5 else int(rng.choice([1, 3, 16, 33]))
    enc = rng.integers(0, k - (1 if seed % 3 else 0), length, dtype=np.uint8)  # sometimes with N / X
    kind = rng.choice(["normal", "ties", "finite_default", "neg_inf_cells"])
    p = np.zeros((m, co.stride(k, 4)), np.float32)
    p[:, :k] = rng.integers(-3, 4, (m, k)) if kind == "ties" else rng.normal(0, 2, (m, k))
    if kind != "finite_default":
        p[:, k - 1] = -np.inf
    if kind == "neg_inf_cells":
        p[rng.random((m, co.stride(k, 4))) < 0.05] = -np.inf
    extra_wrap = int(rng.integers(0, 3))

    ref = co.stripe(enc, cols, k)
    co.configure_wrap(ref, max(m - 1, 0) + extra_wrap)
    rows_total = ref.rows
    a = int(rng.integers(0, max(rows_total, 1)))
    b = int(rng.integers(a, rows_total + 1))
    if seed % 2 == 0:
        a, b = 0, rows_total
    want, want_mi = co.score_rows(ref, p, a, b)

    seq = pli.stripe(lm.EncodedSequence(enc, protein=protein), cols)
    seq.configure_wrap(max(m - 1, 0) + extra_wrap)
    pssm = lm.ScoringMatrix(p, protein=protein)
    rps = int(rng.choice([0, 0, 7, 33, 64, 1000, 50_000]))
    pli.set_rows_per_stream(rps)
    pli.set_prefilter(bool(seed % 4))
    try:
        scores = lm.StripedScores.empty(pli, cols)
        pli.score_rows_into(pssm, seq, range(a, b), scores)
        got = scores.matrix()
        assert got.shape == want.shape and scores.max_index == want_mi
        assert np.array_equal(bits(got[:, :cols]), bits(want[:, :cols])), pli.last_kernel
        assert pli.argmax(scores) == co.argmax(want, cols)
        fused = pli.score_argmax(pssm, seq, range(a, b))
        # Score<u8, ..> with random discrete weights on the same geometry: both overflow flavours
        top = min(int(rng.choice([3, 255 // m + 1, 255])), 255)
        dm = lm.DiscreteMatrix(rng.integers(0, top + 1, (m, k), dtype=np.uint8), 1.0, np.zeros(m, np.float32), 0.0,
                               protein=protein)
        want_wrap, wmi = co.score_rows_u8(ref, dm.data, a, b)
        got_wrap, gmi = pli.score_discrete(dm, seq, rows=range(a, b), saturate=False)
        assert gmi == wmi and np.array_equal(got_wrap[:, :cols], want_wrap[:, :cols]), pli.last_kernel
        if want_wrap.shape[0]:
            want_sat = no.score_rows_u8_saturating(ref.data, cols, length, dm.data, a, b)
            got_sat, _ = pli.score_discrete(dm, seq, rows=range(a, b), saturate=True)
            assert np.array_equal(got_sat[:, :cols], want_sat), pli.last_kernel
        if want.shape[0] == 0:
            assert fused is None
            return
        assert fused[0] == co.argmax(want, cols), pli.last_kernel
        finite = np.sort(want[:, :cols][np.isfinite(want[:, :cols])])
        ts = [0.0]
        if finite.size:
            qs = rng.choice([0.0, 0.5, 0.9, 0.999, 1.0], 2)
            ts += [float(finite[min(int(q * (finite.size - 1)), finite.size - 1)]) for q in qs]
            ts.append(float(finite[-1]) + 1.0)
        for t in ts:
            wrc = [tuple(map(int, rc)) for rc in co.threshold(want, cols, float(t))]
            assert pli.threshold(scores, float(t)) == wrc
            frc, fval = pli.score_threshold(pssm, seq, float(t), range(a, b))
            assert frc == wrc, (t, pli.last_kernel)
            assert np.array_equal(bits(fval), bits([want[r, c] for r, c in wrc]))
        if not protein and a == 0 and b == rows_total and cols == 32 and length >= m:
            t = ts[min(1, len(ts) - 1)]
            by_pos = want[:, :32].T.reshape(-1)[: length - m + 1]
            scanner = lm.Scanner(pssm, seq, threshold=t)
            wpos = np.nonzero(by_pos >= np.float32(t))[0]
            assert scanner.positions.tolist() == wpos.tolist()
            assert np.array_equal(bits(scanner.scores), bits(by_pos[wpos]))
        if a == 0 and b == rows_total and length >= m and kind in ("normal", "ties", "finite_default"):
            # Scanner::max: the walk on the device (scanmax.hip) against the host walk of the downloaded matrices,
            # both overflow flavours of the u8 scores; a candidate window that leaves the matrix is an IndexError in both
            def walk(fn):
                try:
                    hit = fn()
                    return None if hit is None else (hit.position, bits([hit.score])[0])
                except IndexError:
                    return "IndexError"
            for sat in (True, False):
                for t in ts[1:3]:
                    got = walk(lambda: lm.Scanner(pssm, seq, threshold=t).max(sat))
                    host = walk(lambda: scanner_max_strict_host(lm.Scanner(pssm, seq, threshold=t), sat))
                    assert got == host, (t, sat, got, host)
    finally:
        pli.set_rows_per_stream(0)
        pli.set_prefilter(True)


BATCH_SEEDS = range(int(os.environ.get("LM_FUZZ_BATCH_FIRST", "0")), int(os.environ.get("LM_FUZZ_BATCH_LAST", "40")))


@pytest.mark.parametrize("seed", BATCH_SEEDS)
def test_random_batch(pli, seed):
    """Random many-motif batches (equal lengths share a launch, odd shapes go to the generic kernel,
    some thresholds select nothing or everything): every job must equal its own oracle result."""
    rng = np.random.default_rng(70_000 + seed)
    length = int(rng.choice([5_000, 40_000, 150_001]))
    enc = rng.integers(0, 5 if seed % 4 == 0 else 4, length, dtype=np.uint8)
    n = int(rng.integers(1, 13))
    pool = [int(x) for x in rng.integers(1, 41, 4)]          # few distinct lengths -> grouped launches
    lengths = [int(rng.choice(pool)) for _ in range(n)]
    pssms_np = []
    for i, m in enumerate(lengths):
        p = np.zeros((m, 8), np.float32)
        p[:, :5] = rng.integers(-2, 3, (m, 5)) if (seed + i) % 3 == 0 else rng.normal(0, 2, (m, 5))
        if (seed + i) % 5:
            p[:, 4] = -np.inf
        pssms_np.append(p)
    wrap = max(lengths) - 1
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, wrap)
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure_wrap(wrap)
    pssms = [lm.ScoringMatrix(p) for p in pssms_np]
    wants = [co.score_rows(ref, p)[0] for p in pssms_np]
    ts = []
    for w in wants:
        finite = np.sort(w[:, :32][np.isfinite(w[:, :32])])
        q = float(rng.choice([0.0, 0.5, 0.99, 0.9999, 1.0]))
        ts.append(float(finite[min(int(q * (finite.size - 1)), finite.size - 1)]) if finite.size else 0.0)
        if rng.random() < 0.15:
            ts[-1] = float(finite[-1]) + 1.0 if finite.size else 1.0
    pli.set_prefilter(bool(seed % 3))
    try:
        got_am = pli.scan_argmax_batch(pssms, seq)
        got_th = pli.scan_threshold_batch(pssms, ts, seq)
    finally:
        pli.set_prefilter(True)
    for i, w in enumerate(wants):
        if w.shape[0] == 0:
            assert got_am[i] is None and len(got_th[i][0]) == 0
            continue
        assert got_am[i][0] == co.argmax(w, 32), (i, lengths[i])
        assert bits(np.float32(got_am[i][1])) == bits(co.max_(w, 32))
        wrc = co.threshold(w, 32, ts[i]).astype(np.int64).reshape(-1, 2)
        assert np.array_equal(got_th[i][0], wrc), (i, lengths[i], ts[i])
        assert np.array_equal(bits(got_th[i][1]), bits(w[wrc[:, 0], wrc[:, 1]]))


@pytest.mark.parametrize("seed", range(24))
def test_random_large(pli, seed):
    """Sizes where the big-input routes switch on (tracked argmax of score_into from 8 Mi cells,
    candidate-route fused argmax from ~100 M cells): random motif lengths and matrix kinds
    against the C oracle."""
    rng = np.random.default_rng(50_000 + seed)
    m = int(rng.integers(6, 37))
    length = int(rng.choice([9_000_000, 40_000_000, 101_000_000])) + int(rng.integers(0, 97))
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    if seed % 3 == 0:
        enc[rng.random(length) < 0.01] = 4
    p = np.zeros((m, 8), np.float32)
    p[:, :4] = rng.integers(-2, 3, (m, 4)) if seed % 4 == 1 else rng.normal(0, 2, (m, 4))
    p[:, 4] = -np.inf if seed % 5 else rng.normal(0, 2, m)
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, m - 1)
    want, _ = co.score_rows(ref, p)
    want_am = co.argmax(want, 32)
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure_wrap(m - 1)
    pssm = lm.ScoringMatrix(p)
    fused = pli.score_argmax(pssm, seq)
    assert fused[0] == want_am, (m, length, pli.last_kernel)
    assert bits(np.float32(fused[1])) == bits(co.max_(want, 32))
    scores = pli.score(pssm, seq)
    assert pli.argmax(scores) == want_am                       # tracked by the store kernel
    sample = scores.rows_matrix(0, min(scores.rows, 4096))
    assert np.array_equal(bits(sample[:, :32]), bits(want[:sample.shape[0], :32]))
    finite = want[:, :32][np.isfinite(want[:, :32])]
    t = float(np.partition(finite, -2000)[-2000]) if finite.size > 2000 else 0.0
    wrc = co.threshold(want, 32, t).astype(np.int64).reshape(-1, 2)
    rc, vals = pli.score_threshold_dptr(pssm, seq.data_ptr, seq.rows + seq.wrap, 32, 32, seq.wrap,
                                        length, 0, seq.rows, t)
    assert np.array_equal(rc, wrc), (m, length, t)
    assert np.array_equal(bits(vals), bits(want[wrc[:, 0], wrc[:, 1]]))
