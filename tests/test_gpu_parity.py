"""Parity of the HIP path (through the C ABI) with the CPU oracle.

Bar (BASELINE.json north_star): argmax / threshold indices bit-exact vs the
reference Generic pipeline, scores within 1 ulp f32.  The kernels keep the
reference's add order, so scores are compared BIT-EXACTLY (0 ulp) here."""
import json
from pathlib import Path

import numpy as np
import pytest

import lightmotif_amd as lm
from host_walk import scanner_max_strict_host
from oracle import c_oracle as co
from oracle import np_oracle as no

pytestmark = pytest.mark.gpu

GOLD = json.loads((Path(__file__).parent / "golden" / "reference_vectors.json").read_text())
CASES = np.load(Path(__file__).parent / "golden" / "generated_cases.npz")
CASE_NAMES = sorted({k.split("/")[0] for k in CASES.files})


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def random_pssm(rng, m, k, kind="normal"):
    p = np.zeros((m, co.stride(k, 4)), np.float32)
    if kind == "ties":
        p[:, :k] = rng.integers(-2, 3, (m, k))
    else:
        p[:, :k] = rng.normal(0, 2, (m, k))
    p[:, k - 1] = -np.inf
    return p


def check_against_oracle(pli, enc, pssm_np, cols, k, rows=None, extra_wrap=0, thresholds=()):
    """Runs score / argmax / max / threshold / fused forms on the GPU and on the oracle."""
    protein = k == 21
    m = pssm_np.shape[0]
    s = co.stripe(enc, cols, k)
    co.configure_wrap(s, max(m - 1, 0) + extra_wrap)
    a, b = (0, s.rows) if rows is None else rows
    want, want_mi = co.score_rows(s, pssm_np, a, b)

    seq = pli.stripe(lm.EncodedSequence(enc, protein=protein), cols)
    seq.configure_wrap(max(m - 1, 0) + extra_wrap)
    assert np.array_equal(seq.matrix(), s.data), "striped matrix differs"
    pssm = lm.ScoringMatrix(pssm_np, protein=protein)
    scores = lm.StripedScores.empty(pli, cols)
    pli.score_rows_into(pssm, seq, range(a, b), scores)
    got = scores.matrix()
    assert got.shape == want.shape
    assert scores.max_index == want_mi
    assert np.array_equal(bits(got[:, :cols]), bits(want[:, :cols])), \
        f"scores differ ({pli.last_kernel})"
    # Maximum
    assert pli.argmax(scores) == co.argmax(want, cols)
    wmax = co.max_(want, cols)
    gmax = pli.max(scores)
    assert (gmax is None) == (wmax is None)
    if wmax is not None:
        assert bits(np.float32(gmax)) == bits(wmax)
    # fused argmax
    fused = pli.score_argmax(pssm, seq, range(a, b))
    if want.shape[0] == 0:
        assert fused is None
    else:
        assert fused[0] == co.argmax(want, cols)
        assert bits(np.float32(fused[1])) == bits(wmax)
    # Threshold (row-major order is part of the contract here)
    for t in thresholds:
        wrc = [tuple(map(int, rc)) for rc in co.threshold(want, cols, float(t))]
        assert pli.threshold(scores, float(t)) == wrc, f"threshold({t})"
        frc, fval = pli.score_threshold(pssm, seq, float(t), range(a, b))
        assert frc == wrc, f"fused threshold({t})"
        assert np.array_equal(bits(fval), bits([want[r, c] for r, c in wrc]))
    return scores, want


# ---- the reference's own known answers -------------------------------------------------


def golden_objects(pli, cols=32):
    g = GOLD["G1_scores"]
    motif = lm.create(g["patterns"])
    pssm = motif.counts.normalize(g["pseudocount"]).log_odds()
    seq = pli.stripe(lm.EncodedSequence(g["sequence"]), cols)
    return g, pssm, seq


@pytest.mark.parametrize("cols", [32, 1, 16])
def test_g1_scores_like_tests_dna_rs(pli, cols):
    g, pssm, seq = golden_objects(pli, cols)
    seq.configure(pssm)
    result = pli.score(pssm, seq)
    scores = result.unstripe()
    assert len(scores) == len(g["expected"])                            # tests/dna.rs:81
    assert np.abs(scores - np.float32(g["expected"])).max() < 1e-5       # tests/dna.rs:83-90
    # test_score_rows (tests/dna.rs:40-63): exact equality
    sc = lm.StripedScores.empty(pli, cols)
    pli.score_rows_into(pssm, seq, range(0, 2), sc)
    m = sc.matrix()
    assert m.shape[0] == 2 and m[0, 0] == np.float32(g["exact"]["0"])
    if seq.rows > 1:
        assert m[1, 0] == np.float32(g["exact"]["1"])
        pli.score_rows_into(pssm, seq, range(1, 2), sc)
        m = sc.matrix()
        assert m.shape[0] == 1 and m[0, 0] == np.float32(g["exact"]["1"])


@pytest.mark.parametrize("cols", [32, 1, 16])
def test_g2_g3_argmax_threshold_like_tests_dna_rs(pli, cols):
    _, pssm, seq = golden_objects(pli, cols)
    seq.configure(pssm)
    result = pli.score(pssm, seq)
    mc = pli.argmax(result)
    assert result.offset(*mc) == GOLD["G2_argmax"]["offset"]             # tests/dna.rs:138
    for case in GOLD["G3_threshold"]["cases"]:
        idx = sorted(result.offset(r, c) for r, c in pli.threshold(result, case["t"]))
        assert idx == case["sorted_offsets"]                             # tests/dna.rs:158-172


def test_readme_example(pli):
    """README.md:77-90 through the user-facing objects (default 32 columns)."""
    g = GOLD["G1_scores"]
    pssm = lm.create(g["patterns"]).counts.normalize(0.1).log_odds()
    striped = lm.stripe(g["sequence"])
    scores = pssm.calculate(striped)                                     # lib.rs:855-874
    assert len(scores) == 50
    assert scores[0] == np.float32(-23.07094)
    assert scores.argmax() == 18
    assert scores.threshold(10.0) == []
    assert scores.threshold(-10.0) == [18, 32, 27]                       # SURVEY A3 [probe], C = 32
    assert abs(scores.max() - (-5.50167)) < 1e-5
    for x, y in zip(scores, g["expected"]):                              # test_pipeline.py:72-73
        assert round(x - y, 5) == 0


def test_g5_scanner_like_test_scanner_py(pli):
    g = GOLD["G1_scores"]
    pssm = lm.create(g["patterns"]).counts.normalize(0.1).log_odds()
    seq = lm.stripe(g["sequence"])
    seq.configure(pssm)
    assert len(list(lm.scan(pssm, seq))) == 0                            # test_scanner.py:71-72
    hits = list(lm.scan(pssm, seq, threshold=-10.0))
    assert len(hits) == 3
    hits.sort(key=lambda h: h.position)
    for h, w in zip(hits, GOLD["G5_scanner"]["threshold_m10"]):
        assert h.position == w["position"] and abs(h.score - w["score"]) < 1e-5


@pytest.mark.parametrize("length,frac", [(100_003, 0.5), (250_000, 0.01), (64, 1.0), (999_999, 1e-4)])
def test_scanner_dense_hit_lists(pli, length, frac):
    """Scanner hits = every position p with score(p) >= t and p + M <= L (scan.rs:185-190),
    each exactly once; `positions` ascending (the device-side ordering of the hit list,
    hits.hip), iteration in the reference's block-wise pop order -- also when a large share of
    the cells qualifies."""
    rng = np.random.default_rng(length)
    m = 9
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    p = random_pssm(rng, m, 5)
    p[:, 4] = rng.normal(0, 2, m)                      # finite N weights: the padded tail scores too
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, m - 1)
    want, _ = co.score_rows(ref, p)
    rows = want.shape[0]
    pos_scores = want[:, :32].T.reshape(-1)[: length - m + 1]     # position = col * rows + row
    t = float(np.sort(pos_scores)[max(0, int(len(pos_scores) * (1 - frac)) - 1)]) if frac < 1 else -1e30
    wpos = np.nonzero(pos_scores >= np.float32(t))[0]
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure_wrap(m - 1)
    scanner = lm.Scanner(lm.ScoringMatrix(p), seq, threshold=t)
    assert scanner.positions.tolist() == wpos.tolist()
    assert np.array_equal(bits(scanner.scores), bits(pos_scores[wpos]))
    assert rows * 32 >= length
    # iteration = the reference's yield order (blocks ascending, last pushed first; scan.rs:184-198)
    for block_size in (256, 1, 7):
        if block_size != 256 and length > 300_000:
            continue
        want_seq = no.scanner_collect(want, 32, length, m, t, block_size)
        got_seq = list(lm.Scanner(lm.ScoringMatrix(p), seq, threshold=t, block_size=block_size))
        assert [h.position for h in got_seq] == [i for i, _ in want_seq]
        assert np.array_equal(bits([h.score for h in got_seq]), bits([x for _, x in want_seq]))
    best = no.scanner_max(want, 32, length, m, t)                    # scan.rs:200-249
    got = lm.Scanner(lm.ScoringMatrix(p), seq, threshold=t).max_valid()
    assert (got.position, np.float32(got.score)) == (best[0], best[1])
    it = lm.Scanner(lm.ScoringMatrix(p), seq, threshold=t)
    first = next(it)                                                  # max() sees what is left
    rest = [h for h in no.scanner_collect(want, 32, length, m, t, 256)][1:]
    if rest:
        bi, bs = max(rest, key=lambda h: (h[1], h[0]))
        after = it.max_valid()
        assert (after.position, np.float32(after.score)) == (bi, bs) and len(it) == 0
    else:
        assert it.max_valid() is None
    assert first.position == no.scanner_collect(want, 32, length, m, t, 256)[0][0]


def test_scanner_max_valid_does_not_build_the_hit_list(pli):
    """`Scanner.max_valid` on a fresh scanner with a threshold that selects every
    position: the best hit comes from the fused argmax + one scan at its score, the full hit
    list is never materialised; ties go to the greater POSITION, not the later (row, col) cell."""
    rng = np.random.default_rng(12)
    length, m = 300_000, 6
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    p = np.zeros((m, 8), np.float32)
    p[:, :4] = rng.integers(0, 2, (m, 4))            # many exact ties at the maximum
    p[:, 4] = -np.inf
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure_wrap(m - 1)
    sc = lm.Scanner(lm.ScoringMatrix(p), seq, threshold=-1e30)
    best = sc.max_valid()
    assert sc._positions is None or sc._positions.size == 0      # nothing was collected
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, m - 1)
    want, _ = co.score_rows(ref, p)
    by_pos = want[:, :32].T.reshape(-1)[: length - m + 1]
    top = np.nonzero(by_pos == by_pos.max())[0]
    assert top.size > 1 and (best.position, best.score) == (int(top[-1]), float(by_pos.max()))
    assert lm.Scanner(lm.ScoringMatrix(p), seq, threshold=float(by_pos.max()) + 0.5).max_valid() is None
    assert len(sc) == 0 and sc.max_valid() is None                # consumed


def test_reverse_complement_on_the_device(pli):
    """pwm/mod.rs:566-577: the library's resident reverse complement is the matrix the host mirror
    builds; scoring the reverse-complemented sequence with it mirrors the forward scores."""
    rng = np.random.default_rng(21)
    length, m = 50_000, 13
    text = "".join("ACGT"[i] for i in rng.integers(0, 4, length))
    sites = ["".join("ACGT"[i] for i in rng.integers(0, 4, m)) for _ in range(7)]
    pssm = lm.create(sites).counts.normalize(0.1).log_odds()
    seq = lm.stripe(text)
    fwd = pssm.calculate(seq).unstripe()                      # uploads pssm to this pipeline
    assert pssm._dev
    rc = pssm.reverse_complement()
    assert set(rc._dev) == set(pssm._dev)                    # made by lm_hip_pssm_reverse_complement
    host_only = lm.ScoringMatrix(rc.data, rc.background)     # same matrix, uploaded from the host
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rseq = lm.stripe("".join(comp[c] for c in reversed(text)))
    a = rc.calculate(rseq)
    b = host_only.calculate(rseq)
    assert np.array_equal(bits(a.unstripe()), bits(b.unstripe())) and a.argmax() == b.argmax()
    assert np.allclose(a.unstripe()[::-1], fwd, atol=1e-4)   # same terms, opposite add order
    with pytest.raises(ValueError):
        lm.create(["ACDEF"], protein=True).pssm.reverse_complement()


@pytest.mark.parametrize("m,count", [(4, 5), (9, 7), (11, 2), (13, 3), (22, 5), (30, 3)])
def test_several_motifs_of_one_length_per_pass(pli, m, count):
    """Batches of equal-length DNA motifs run several motifs per pass of the pair scan
    (score_c32_prefilter2_multi; job table padded when the count is not a multiple): every
    job must equal its own oracle result, also next to a motif of another length."""
    rng = np.random.default_rng(100 * m + count)
    length = 150_001
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, 40)
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure_wrap(40)
    mats = [random_pssm(rng, m, 5) for _ in range(count)] + [random_pssm(rng, m + 1, 5)]
    pssms = [lm.ScoringMatrix(p) for p in mats]
    wants = [co.score_rows(ref, p)[0] for p in mats]
    ts = [float(np.sort(w[:, :32][np.isfinite(w[:, :32])])[-(50 + 40 * i)]) for i, w in enumerate(wants)]
    got = pli.scan_threshold_batch(pssms, ts, seq)
    if m <= 23:  # (the one-symbol scan when the pair scan is switched off for a whole run)
        assert pli.last_kernel in ("score_c32_prefilter2_multi", "score_c32_prefilter2", "score_c32_prefilter")
    for (coords, vals), w, t in zip(got, wants, ts):
        assert np.array_equal(coords, co.threshold(w, 32, t))
        assert np.array_equal(bits(vals), bits(w[coords[:, 0], coords[:, 1]]))
    best = pli.scan_argmax_batch(pssms, seq)
    for b, w in zip(best, wants):
        assert b[0] == co.argmax(w, 32)


def test_indexing_reads_single_rows(pli):
    """``scores[i]`` = cell (i % rows, i / rows) (scores.rs:246-254), fetched without downloading
    the matrix; windows of rows come back like the corresponding slice of the full copy."""
    g, pssm, seq = golden_objects(pli)
    scores = pssm.calculate(seq)
    full = scores.matrix()
    for i, want in enumerate(g["expected"]):
        assert abs(scores[i] - want) < 1e-5
    assert scores[-1] == scores[len(scores) - 1]
    view = np.asarray(scores)                                            # lib.rs:1051-1085, 1129-1139
    assert view.shape == (scores.columns, scores.rows) and view.dtype == np.float32
    assert not view.flags.writeable
    assert np.array_equal(bits(view.ravel()[: len(scores)]), bits(scores.unstripe()))
    with pytest.raises(IndexError):
        scores[len(scores)]
    assert np.array_equal(bits(scores.rows_matrix(1, 2)), bits(full[1:2]))
    assert scores.rows_matrix(2, 2).shape == (0, full.shape[1])
    with pytest.raises(lm.LightmotifHipError):
        scores.rows_matrix(0, full.shape[0] + 1)


def test_g7_empty_row_range_does_not_fail(pli):
    g = GOLD["G7_empty_range"]
    seq = pli.stripe(lm.EncodedSequence(g["sequence"]), g["columns"])
    pssm = lm.create(g["patterns"]).counts.normalize(g["pseudocount"]).log_odds()
    seq.configure(pssm)
    sc = lm.StripedScores.empty(pli, g["columns"])
    pli.score_rows_into(pssm, seq, range(1, 1), sc)                      # pli/mod.rs:603-623
    assert sc.rows == 0 and sc.is_empty() and sc.max_index == 0
    assert pli.argmax(sc) is None and pli.max(sc) is None and pli.threshold(sc, 0.0) == []


# ---- committed fixtures + seeded random cases vs the oracle ---------------------------------


@pytest.mark.parametrize("name", CASE_NAMES)
def test_generated_fixtures(pli, name):
    c = {k.split("/", 1)[1]: CASES[k] for k in CASES.files if k.startswith(name + "/")}
    cols, k = int(c["cols"]), int(c["k"])
    m = c["pssm"].shape[0]
    extra = int(c["wrap"]) - max(m - 1, 0)
    scores, _ = check_against_oracle(pli, c["encoded"], c["pssm"], cols, k,
                                     rows=tuple(map(int, c["row_range"])), extra_wrap=extra,
                                     thresholds=list(c["thresholds"]))
    assert np.array_equal(bits(scores.matrix()), bits(c["scores"]))      # the committed bits
    am = pli.argmax(scores)
    assert (am if am else (-1, -1)) == tuple(c["argmax"])
    for i, t in enumerate(c["thresholds"]):
        got = np.array(pli.threshold(scores, float(t)), dtype=np.int64).reshape(-1, 2)
        assert np.array_equal(got, c[f"threshold_{i}"])


@pytest.mark.parametrize("m", list(range(1, 35)) + [40, 64])
def test_every_motif_length_dna(pli, m):
    """Unrolled kernels exist for M = 1..36; longer motifs take the generic kernel."""
    rng = np.random.default_rng(1000 + m)
    length = int(rng.integers(32 * (m + 2), 9000))
    enc = rng.integers(0, 5, length, dtype=np.uint8)   # includes N -> -inf scores
    p = random_pssm(rng, m, 5)
    check_against_oracle(pli, enc, p, 32, 5, thresholds=[0.0, -np.inf])
    if m <= 36:
        assert pli.last_kernel.startswith("score_c32") or length // 32 < m + 1


@pytest.mark.parametrize("length", [0, 1, 14, 15, 16, 31, 32, 33, 63, 64, 65, 479, 480, 481, 2047, 4099])
def test_ragged_lengths(pli, length):
    rng = np.random.default_rng(length)
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    check_against_oracle(pli, enc, random_pssm(rng, 15, 5), 32, 5, thresholds=[1.0])


@pytest.mark.parametrize("rows", [(0, 1), (3, 4), (0, 16), (5, 37), (100, 301), (299, 300), (7, 7), (9, 2)])
def test_row_ranges(pli, rows):
    rng = np.random.default_rng(42)
    enc = rng.integers(0, 4, 9601, dtype=np.uint8)       # 301 rows
    check_against_oracle(pli, enc, random_pssm(rng, 20, 5), 32, 5, rows=rows, thresholds=[2.0])


@pytest.mark.parametrize("rps", [21, 61, 500, 100000])
def test_rows_per_stream_knob_does_not_change_results(pli, rps):
    rng = np.random.default_rng(5)
    enc = rng.integers(0, 4, 200_003, dtype=np.uint8)
    pli.set_rows_per_stream(rps)
    try:
        check_against_oracle(pli, enc, random_pssm(rng, 20, 5), 32, 5, thresholds=[8.0])
    finally:
        pli.set_rows_per_stream(0)


@pytest.mark.parametrize("cols", [1, 2, 4, 16, 32, 33])
def test_other_column_counts(pli, cols):
    rng = np.random.default_rng(cols)
    enc = rng.integers(0, 5, 777, dtype=np.uint8)
    check_against_oracle(pli, enc, random_pssm(rng, 9, 5, "ties"), cols, 5, thresholds=[0.0, 3.0])


@pytest.mark.parametrize("m", [1, 5, 12, 20, 33])
def test_protein(pli, m):
    rng = np.random.default_rng(m)
    enc = rng.integers(0, 21, 20_011, dtype=np.uint8)
    check_against_oracle(pli, enc, random_pssm(rng, m, 21), 32, 21, thresholds=[4.0])


@pytest.mark.parametrize("m", [5, 12, 20, 31])
def test_protein_pair_prefilter_opt_in(m):
    """The 441-row pair scan (score_c32_prefilter2<M, 21>) is off by default (measured 4 % slower)
    but stays bit-exact: a pipeline with the option "pair_prefilter_protein" set must use it and agree."""
    pair = lm.Pipeline.hip(0)
    pair.set_option("pair_prefilter_protein", 1)
    rng = np.random.default_rng(100 + m)
    enc = rng.integers(0, 21, 40_009, dtype=np.uint8)
    p = random_pssm(rng, m, 21)
    _, want = check_against_oracle(pair, enc, p, 32, 21, thresholds=[4.0])
    seq = pair.stripe(lm.EncodedSequence(enc, protein=True), 32)
    seq.configure_wrap(m - 1)
    t = float(np.sort(want[:, :32][np.isfinite(want[:, :32])])[-20])
    frc, _ = pair.score_threshold(lm.ScoringMatrix(p, protein=True), seq, t)
    assert pair.last_kernel == "score_c32_prefilter2"
    assert frc == [tuple(map(int, rc)) for rc in co.threshold(want, 32, t)]


def test_thresholds_above_the_best_kmer_are_not_scanned(pli):
    """A threshold above B = the sequential f32 sum of the PSSM's row maxima selects nothing whatever the
    sequence (rounding is monotone), so the fused threshold skips the scan; at t == B the planted best k-mer
    must still be found, and the batch form must keep its per-motif layout when some motifs are skipped."""
    rng = np.random.default_rng(77)
    length = 300_007
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    motifs = []
    for m in (6, 9, 14):
        p = np.zeros((m, 8), np.float32)
        p[:, :4] = rng.normal(0, 2, (m, 4))
        p[:, 4] = -np.inf
        motifs.append(p)
    p = motifs[1]
    cons = np.argmax(p[:, :4], axis=1).astype(np.uint8)
    enc[200_000:200_000 + len(cons)] = cons
    b = np.float32(0)
    for j in range(len(cons)):
        b = np.float32(b + p[j, cons[j]])
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, 13)
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure_wrap(13)
    pssm = lm.ScoringMatrix(p)
    want, _ = co.score_rows(ref, p)
    above = float(np.nextafter(b, np.float32(np.inf)))
    pli.score_argmax(pssm, seq)                               # (sets last_kernel to something else)
    before = pli.last_kernel
    assert pli.score_threshold(pssm, seq, above) == ([], [])
    assert pli.last_kernel == before                          # nothing was launched
    sc = lm.Scanner(pssm, seq, threshold=above)               # the Scanner's hit list through the same shortcut
    assert len(sc.positions) == 0 and list(sc) == []
    sc = lm.Scanner(pssm, seq, threshold=float(b))
    assert 200_000 in sc.positions.tolist() and all(h.score == float(b) for h in sc)
    frc, fval = pli.score_threshold(pssm, seq, float(b))
    wrc = [tuple(map(int, rc)) for rc in co.threshold(want, 32, float(b))]
    assert frc == wrc and len(frc) >= 1 and all(v == float(b) for v in fval)
    # batch: the middle motif's threshold is unreachable, the others' are not
    pssms = [lm.ScoringMatrix(q) for q in motifs]
    wants = [co.score_rows(ref, q)[0] for q in motifs]
    ts = [float(np.sort(wants[0][:, :32][np.isfinite(wants[0][:, :32])])[-40]), above,
          float(np.sort(wants[2][:, :32][np.isfinite(wants[2][:, :32])])[-40])]
    res = pli.scan_threshold_batch(pssms, ts, seq)
    for (coords, vals), w, t in zip(res, wants, ts):
        wrc = [tuple(map(int, rc)) for rc in co.threshold(w, 32, t)]
        assert [tuple(map(int, rc)) for rc in coords] == wrc
    assert len(res[1][0]) == 0 and len(res[0][0]) >= 40
    full = lm.Pipeline.hip(0)
    full.set_option("skip_unreachable", 0)
    seq2 = full.stripe(lm.EncodedSequence(enc), 32)
    seq2.configure_wrap(13)
    assert full.score_threshold(lm.ScoringMatrix(p), seq2, above) == ([], [])
    assert full.last_kernel.startswith("score_c32")          # ... which the switch turns back into a scan


def test_million_positions_bitwise(pli):
    rng = np.random.default_rng(99)
    enc = rng.integers(0, 4, 1_000_003, dtype=np.uint8)
    p = random_pssm(rng, 20, 5)
    scores, want = check_against_oracle(pli, enc, p, 32, 5, thresholds=[12.0])
    assert scores.rows == 31251


def test_ties_take_the_last_cell_in_row_major_order(pli):
    """pli/mod.rs:146 `>=`: differs from the AVX2 and SSE2 rules."""
    enc = np.zeros(32 * 50, np.uint8)                     # all 'A' -> every score equal
    p = np.zeros((4, 8), np.float32)
    p[:, 0] = 1.5
    p[:, 4] = -np.inf
    scores, want = check_against_oracle(pli, enc, p, 32, 5, thresholds=[6.0, 6.5])
    # 50 rows x 32 cols; the last 3 positions (tail of column 31) touch the N wrap -> -inf
    assert pli.argmax(scores) == co.argmax(want, 32) == (49, 30)


def test_nan_and_all_neg_inf(pli):
    rng = np.random.default_rng(3)
    enc = rng.integers(0, 4, 3200, dtype=np.uint8)
    p = random_pssm(rng, 6, 5)
    p[2, 1] = np.nan
    check_against_oracle(pli, enc, p, 32, 5, thresholds=[0.0])
    # scores[0][0] NaN -> argmax (0, 0) (pli/mod.rs:142-146)
    enc2 = enc.copy()
    enc2[2] = 1
    scores, want = check_against_oracle(pli, enc2, p, 32, 5)
    assert np.isnan(want[0, 0]) and pli.argmax(scores) == (0, 0)
    # every score -inf: answer is the very last cell, even past max_index
    p2 = np.full((6, 8), -np.inf, np.float32)
    scores, want = check_against_oracle(pli, enc, p2, 32, 5, thresholds=[-np.inf])
    assert pli.argmax(scores) == (99, 31)


def test_finite_default_column_exposes_padded_tail(pli):
    """A5: Threshold/Maximum ignore max_index; reproduce, do not fix."""
    rng = np.random.default_rng(8)
    enc = rng.integers(0, 4, 1000, dtype=np.uint8)        # 32 rows, 24 padded cells
    p = random_pssm(rng, 5, 5)
    p[:, 4] = 50.0                                        # N scores high
    scores, want = check_against_oracle(pli, enc, p, 32, 5, thresholds=[100.0])
    r, c = pli.argmax(scores)
    assert scores.offset(r, c) >= len(scores)             # offset beyond len(scores)


def test_not_enough_wrap_rows_is_an_error(pli):
    seq = pli.stripe(lm.EncodedSequence("ACGT" * 100), 32)
    pssm = lm.ScoringMatrix(np.zeros((10, 8), np.float32))
    with pytest.raises(lm.LightmotifHipError, match="not enough wrapping rows for motif of length 10"):
        pli.score(pssm, seq)                              # avx2.rs:832-837 panics
    seq.configure_wrap(8)
    with pytest.raises(lm.LightmotifHipError, match="not enough wrapping rows"):
        pli.score(pssm, seq)
    seq.configure_wrap(9)
    assert pli.score(pssm, seq).rows == seq.rows
    sc = lm.StripedScores.empty(pli, 32)
    with pytest.raises(lm.LightmotifHipError):
        pli.score_rows_into(pssm, seq, range(0, seq.rows + 1), sc)


def test_host_pointer_entry_points(pli):
    """The forms a Rust shim would call with pointers into its own Vecs."""
    import ctypes as C
    from lightmotif_amd import _ffi
    L = _ffi.lib()
    rng = np.random.default_rng(17)
    enc = rng.integers(0, 4, 50_000, dtype=np.uint8)
    p = random_pssm(rng, 20, 5)
    s = co.stripe(enc, 32, 5)
    co.configure_wrap(s, 19)
    want, mi = co.score_rows(s, p, 10, 900)
    out = np.zeros((890, 32), np.float32)
    orow, omi = C.c_size_t(0), C.c_size_t(0)
    st = L.lm_hip_score_f32(s.data.ctypes.data, s.data.shape[0], 32, 32, s.wrap, s.length,
                            p.ctypes.data, 20, 8, 5, 10, 900, out.ctypes.data, 32,
                            C.byref(orow), C.byref(omi))
    assert st == 0, L.lm_hip_last_error()
    assert (orow.value, omi.value) == (890, mi)
    assert np.array_equal(bits(out), bits(want))
    found, best, val = C.c_int(0), _ffi.Coords(), C.c_float(0)
    assert L.lm_hip_argmax_f32(out.ctypes.data, 890, 32, 32, C.byref(found), C.byref(best), C.byref(val)) == 0
    assert found.value == 1 and (best.row, best.col) == co.argmax(want, 32)
    # Maximum::max (pli/mod.rs:158-160) = the value at the Generic argmax, through its own exports
    mfound, mval = C.c_int(0), C.c_float(0)
    assert L.lm_hip_max_f32(out.ctypes.data, 890, 32, 32, C.byref(mfound), C.byref(mval)) == 0
    assert mfound.value == 1 and np.float32(mval.value) == want[co.argmax(want, 32)] == np.float32(val.value)
    neg = np.full((7, 32), -np.inf, np.float32)            # all -inf -> -inf (Avx2::max_f32 would say 0.0)
    assert L.lm_hip_max_f32(neg.ctypes.data, 7, 32, 32, C.byref(mfound), C.byref(mval)) == 0
    assert mfound.value == 1 and mval.value == -np.inf
    neg[0, 0] = np.nan                                      # NaN first cell -> argmax (0, 0) -> NaN
    assert L.lm_hip_max_f32(neg.ctypes.data, 7, 32, 32, C.byref(mfound), C.byref(mval)) == 0
    assert mfound.value == 1 and np.isnan(mval.value)
    assert L.lm_hip_max_f32(neg.ctypes.data, 0, 32, 32, C.byref(mfound), C.byref(mval)) == 0 and mfound.value == 0
    ptr, n = C.POINTER(_ffi.Coords)(), C.c_size_t(0)
    assert L.lm_hip_threshold_f32(out.ctypes.data, 890, 32, 32, 9.0, C.byref(ptr), C.byref(n)) == 0
    got = [(ptr[i].row, ptr[i].col) for i in range(n.value)]
    L.lm_hip_free(ptr)
    assert got == [tuple(map(int, rc)) for rc in co.threshold(want, 32, 9.0)]
    # degenerate: L < M is not an error (pli/mod.rs:85-88)
    st = L.lm_hip_score_f32(s.data.ctypes.data, s.data.shape[0], 32, 32, s.wrap, 5,
                            p.ctypes.data, 20, 8, 5, 0, 10, out.ctypes.data, 32,
                            C.byref(orow), C.byref(omi))
    assert st == 0 and (orow.value, omi.value) == (0, 0)


# ---- many motifs x one resident sequence (BASELINE.json configs[2]) ----------------------------


def test_batch_scan_matches_per_motif_oracle(pli):
    rng = np.random.default_rng(2024)
    length = 300_017
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    enc[rng.random(length) < 0.002] = 4                        # a few N -> -inf scores
    lengths = [4, 5, 6, 8, 8, 11, 15, 20, 24, 29, 33, 40]      # JASPAR spans 4..33; 40 -> generic kernel
    pssms_np = [random_pssm(rng, m, 5, "ties" if i % 3 == 0 else "normal") for i, m in enumerate(lengths)]
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure_wrap(max(lengths) - 1)                        # main.rs:540-546 configure_wrap(max_m)
    pssms = [lm.ScoringMatrix(p) for p in pssms_np]
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, max(lengths) - 1)
    wants = [co.score_rows(ref, p)[0] for p in pssms_np]
    ts = [float(np.sort(w[:, :32][np.isfinite(w[:, :32])])[-200:][0]) for w in wants]

    got_am = pli.scan_argmax_batch(pssms, seq)
    got_th = pli.scan_threshold_batch(pssms, ts, seq)
    for i, w in enumerate(wants):
        assert got_am[i][0] == co.argmax(w, 32), lengths[i]
        assert bits(np.float32(got_am[i][1])) == bits(co.max_(w, 32))
        wrc = co.threshold(w, 32, ts[i]).astype(np.int64)
        assert np.array_equal(got_th[i][0], wrc), lengths[i]
        assert np.array_equal(bits(got_th[i][1]), bits(w[wrc[:, 0], wrc[:, 1]]))
    # a motif longer than the sequence is a degenerate job, not an error (pli/mod.rs:85-88)
    short = pli.stripe(lm.EncodedSequence(enc[:10]), 32)
    short.configure_wrap(39)
    res = pli.scan_argmax_batch(pssms, short)
    assert [r is None for r in res] == [m > 10 for m in lengths]
    # not enough wrap rows for the longest motif -> the reference panics
    bare = pli.stripe(lm.EncodedSequence(enc), 32)
    bare.configure_wrap(10)
    with pytest.raises(lm.LightmotifHipError, match="not enough wrapping rows"):
        pli.scan_argmax_batch(pssms, bare)


# ---- discrete prefilter of the fused threshold scan (score_prefilter.hpp) -----------------------


@pytest.mark.parametrize("m", [1, 2, 3, 7, 8, 15, 20, 21, 33, 36])
@pytest.mark.parametrize("kind", ["normal", "ties"])
def test_prefilter_and_exact_fused_threshold_agree(pli, m, kind):
    """The packed 16-bit prefilter may only over-select; after exact re-scoring the hit
    list must equal the f32 kernel's and the oracle's, for even and odd motif lengths."""
    rng = np.random.default_rng(7000 + m)
    length = 250_003
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    enc[rng.random(length) < 0.01] = 4                       # N -> -inf weights in the window
    p = random_pssm(rng, m, 5, kind)
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, max(m - 1, 0))
    want, _ = co.score_rows(ref, p)
    finite = np.sort(want[:, :32][np.isfinite(want[:, :32])])
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure_wrap(max(m - 1, 0))
    pssm = lm.ScoringMatrix(p)
    # thresholds: deep tail, an exact score value (>= must include ties), mid-range, below everything
    for t in (float(finite[-50]), float(finite[-1]), float(finite[len(finite) // 2]), float(finite[0]) - 1.0,
              float(finite[-1]) + 1.0):
        wrc = [tuple(map(int, rc)) for rc in co.threshold(want, 32, t)]
        got = {}
        for on in (True, False):
            pli.set_prefilter(on)
            try:
                rc, val = pli.score_threshold(pssm, seq, t)
            finally:
                pli.set_prefilter(True)
            got[on] = (rc, val, pli.last_kernel)
            assert rc == wrc, (m, kind, t, on, pli.last_kernel)
            assert np.array_equal(bits(val), bits([want[r, c] for r, c in wrc]))
        assert got[False][2].startswith("score_c32<")
    # with a meaningful threshold the prefilter kernel is the one that runs
    pli.score_threshold(pssm, seq, float(finite[-50]))
    if m >= 2:
        assert pli.last_kernel .startswith("score_c32_prefilter")


def test_prefilter_is_skipped_when_it_cannot_be_sound(pli):
    rng = np.random.default_rng(99)
    enc = rng.integers(0, 4, 100_000, dtype=np.uint8)
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure_wrap(9)
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, 9)
    for poison in (np.nan, np.inf):
        p = random_pssm(rng, 10, 5)
        p[3, 2] = poison                                      # NaN / +inf weights: no discrete bound
        want, _ = co.score_rows(ref, p)
        rc, _ = pli.score_threshold(lm.ScoringMatrix(p), seq, 5.0)
        assert pli.last_kernel.startswith("score_c32<10,2>")
        assert rc == [tuple(map(int, x)) for x in co.threshold(want, 32, 5.0)]
    p = random_pssm(rng, 10, 5)
    want, _ = co.score_rows(ref, p)
    rc, _ = pli.score_threshold(lm.ScoringMatrix(p), seq, float("-inf"))   # selects everything non-NaN
    assert pli.last_kernel.startswith("score_c32<10,2>")
    assert len(rc) == want.shape[0] * 32


def test_entry_points_are_thread_safe(pli):
    """Callers use the reference concurrently (CLI worker threads main.rs:270, Python with the GIL
    released lib.rs:865): a context serialises its own work, separate contexts run side by side."""
    import threading
    rng = np.random.default_rng(5150)
    enc = rng.integers(0, 4, 200_003, dtype=np.uint8)
    motifs = [random_pssm(rng, m, 5) for m in (6, 11, 20, 27)]
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, 26)
    wants = [co.score_rows(ref, p)[0] for p in motifs]
    want_am = [co.argmax(w, 32) for w in wants]
    ts = [float(np.sort(w[:, :32][np.isfinite(w[:, :32])])[-300]) for w in wants]
    want_th = [[tuple(map(int, rc)) for rc in co.threshold(w, 32, t)] for w, t in zip(wants, ts)]
    other = lm.Pipeline.hip()
    errors = []
    text = np.frombuffer(b"ACTG", np.uint8)[enc]
    packed, _ = lm.pack_2bit(enc)
    # Scanner::max walks (scanmax.hip), expected from the host walk of a scanner run before the threads start
    want_walk = []
    seq0 = pli.stripe(lm.EncodedSequence(enc), 32)
    seq0.configure_wrap(26)
    for p_, t_ in zip(motifs, ts):
        h = scanner_max_strict_host(lm.Scanner(lm.ScoringMatrix(p_), seq0, threshold=t_))
        want_walk.append(None if h is None else (h.position, h.score))

    def worker(p, k):
        try:
            # the three ingest forms (symbol bytes, text tile by tile, 2 bits per base) give the same resident matrix
            seq = (p.stripe(lm.EncodedSequence(enc), 32) if k % 3 == 0 else p.stripe_ascii(text) if k % 3 == 1
                   else p.stripe_2bit(packed, enc.size))
            seq.configure_wrap(26)
            assert np.array_equal(seq.matrix(), ref.data[:, :32])
            for it in range(6):
                i = (k + it) % len(motifs)
                pssm = lm.ScoringMatrix(motifs[i])
                scores = p.score(pssm, seq)
                assert np.array_equal(bits(scores.matrix()[:, :32]), bits(wants[i][:, :32]))
                assert p.argmax(scores) == want_am[i]
                assert p.score_argmax(pssm, seq)[0] == want_am[i]
                assert p.score_threshold(pssm, seq, ts[i])[0] == want_th[i]
                h = lm.Scanner(pssm, seq, threshold=ts[i]).max()
                assert (None if h is None else (h.position, h.score)) == want_walk[i]
        except Exception as e:  # noqa: BLE001 - reported to the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(pli if k % 2 == 0 else other, k)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("kind", ["random", "below_threshold_first_candidate", "finite_n_tail", "overestimate_skip",
                                  "partially_consumed"])
def test_scanner_max_strict_reference_mode(pli, kind):
    """`Scanner.max()` (the default call) walks scan.rs:200-249 as written -- u8 DiscreteMatrix
    scores steer the candidates, no `position + M <= L` test, first candidate accepted without
    the f32 threshold test, level = u8 score of the current best -- and must equal the oracle's
    literal restatement (np_oracle.scanner_max_strict) on inputs built to hit each quirk;
    `max_valid()` is the opt-in best VALID hit."""
    rng = np.random.default_rng(sum(map(ord, kind)))
    length, m = 30_011, 9
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    p = np.zeros((m, 8), np.float32)
    p[:, :4] = rng.normal(0, 2, (m, 4))
    p[:, 4] = -np.inf
    if kind == "finite_n_tail":
        # N scores high in the motif's last rows only: windows that START inside the sequence and run
        # into the padded tail hold the maximum (position + M > L), the all-N windows score low
        p[:6, 4] = -20.0
        p[6:, 4] = 4.0
        length = 30_000 - 7                             # tail cells exist (rows * 32 > L)
        enc = enc[:length].copy()
        enc[length - 6:] = p[:6, :4].argmax(axis=1)     # the best start of such a window: 6 real symbols + 3 N
    if kind == "overestimate_skip":
        p[:, :4] = rng.integers(-3, 4, (m, 4)) + rng.random((m, 4)).astype(np.float32) * 0.3
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, m - 1)
    scores, _ = co.score_rows(ref, p)
    w, factor, offsets, offset = no.to_discrete(p, 5)
    d = no.score_rows_u8_saturating(ref.data, 32, length, w, 0, ref.rows)
    scale = lambda x: no.discrete_scale(x, factor, offset)   # noqa: E731
    finite = scores[:, :32][np.isfinite(scores[:, :32])]
    t = float(np.quantile(finite, 0.999))
    if kind == "below_threshold_first_candidate":
        t = float(finite.max()) + 0.05                  # no f32 score reaches t; u8 over-estimates may
    pssm = lm.ScoringMatrix(p)
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure(pssm)
    assert np.array_equal(pli.score_discrete(pssm.to_discrete(), seq)[0][:, :32], d[:, :32])
    for bs in (256, 7):
        pending = ()
        sc = lm.Scanner(pssm, seq, threshold=t, block_size=bs)
        if kind == "partially_consumed":
            order = no.scanner_collect(scores, 32, length, m, t, bs)
            took = [next(sc) for _ in range(3)]
            assert [(h.position, np.float32(h.score)) for h in took] == [(i, s) for i, s in order[:3]]
            blk = (order[2][0] % ref.rows) // bs
            rest_same_block = [(i, s) for i, s in order[3:] if (i % ref.rows) // bs == blk]
            pending = rest_same_block[::-1]             # the reference's vector order (popped from the end)
            first_row = (blk + 1) * bs
            want = no.scanner_max_strict(scores[first_row:], d[first_row:], 32, t, scale, bs, pending=pending)
            if want is not None and want not in pending:   # block-relative -> global position
                col, r = divmod(want[0], scores[first_row:].shape[0])
                want = (col * ref.rows + first_row + r, want[1])
        else:
            want = no.scanner_max_strict(scores, d, 32, t, scale, bs)
        got = sc.max()                                  # the walk on the device (lm_hip_scan_max_f32)
        if want is None:
            assert got is None, (kind, bs)
        else:
            assert got is not None and got.position == want[0] and np.float32(got.score) == want[1], (kind, bs, got, want)
        if kind != "partially_consumed":                # ... and the same walk on the host from the downloaded matrices
            host = scanner_max_strict_host(lm.Scanner(pssm, seq, threshold=t, block_size=bs))
            assert (host is None) == (got is None) and (host is None or (host.position, host.score) == (got.position, got.score))
    # the opt-in variant: the best valid hit, whatever the u8 scores say
    best = no.scanner_max(scores, 32, length, m, t)
    got = lm.Scanner(pssm, seq, threshold=t).max_valid()
    assert (got is None) == (best is None)
    if best is not None:
        assert got.position == best[0] and np.float32(got.score) == best[1]
    if kind == "below_threshold_first_candidate":
        strict = lm.Scanner(pssm, seq, threshold=t).max()
        assert best is None and (strict is None or strict.score < t)
    if kind == "finite_n_tail":
        strict = lm.Scanner(pssm, seq, threshold=t).max()
        assert strict.position + m > length             # the reference reports a position in the padded tail
        # ... and where a candidate's window leaves the striped matrix, `seq[pos + j]` panics in the
        # reference (seq.rs:433-442 indexes column C): the strict mode raises likewise
        p2 = p.copy()
        p2[:, 4] = 3.0
        hot = lm.ScoringMatrix(p2)
        with pytest.raises(IndexError):
            lm.Scanner(hot, seq, threshold=t).max()


def test_scanner_on_protein(pli):
    """The Rust Scanner is alphabet-generic (scan.rs:96-136; `to_discrete` pwm/mod.rs:665-696 for any
    K).  Protein sequences go through the one-symbol u16 prefilter (21 table rows) + exact
    re-scoring: the hits, their yield order and max() against the oracle's restatement."""
    rng = np.random.default_rng(21)
    length, m = 500_011, 12
    enc = rng.integers(0, 21, length, dtype=np.uint8)
    enc[rng.random(length) < 0.99] %= 20                      # a sprinkle of X
    sites = ["".join(lm.lib.PROTEIN_SYMBOLS[i] for i in rng.integers(0, 20, m)) for _ in range(6)]
    pssm = lm.create(sites, protein=True).counts.normalize(0.1).log_odds()
    ref = co.stripe(enc, 32, 21)
    co.configure_wrap(ref, m - 1)
    want, _ = co.score_rows(ref, pssm.data)
    seq = pli.stripe(lm.EncodedSequence(enc, protein=True), 32)
    seq.configure(pssm)
    finite = want[:, :32][np.isfinite(want[:, :32])]
    for q, bs in ((0.9999, 256), (0.999, 64)):
        t = float(np.quantile(finite, q))
        order = no.scanner_collect(want, 32, length, m, t, bs)
        sc = lm.Scanner(pssm, seq, threshold=t, block_size=bs)
        got = [(h.position, np.float32(h.score)) for h in sc]
        assert len(got) > 20 and got == [(i, s) for i, s in order]
        best = no.scanner_max(want, 32, length, m, t)
        top = lm.Scanner(pssm, seq, threshold=t, block_size=bs).max_valid()
        assert (top.position, np.float32(top.score)) == best
    for on in (True, False):                                  # prefilter on / off: same hits
        pli.set_prefilter(on)
        try:
            pos = lm.Scanner(pssm, seq, threshold=t).positions
        finally:
            pli.set_prefilter(True)
        assert pos.tolist() == sorted(i for i, _ in order)
    with pytest.raises(ValueError):
        lm.scan(pssm, seq)                                    # the Python binding's helper stays DNA-only (lib.rs)


ABYB1_SITES = ["SFKELGFDSLTAVELRNRLAAAT", "AFKELGFDSLAAIQLRNRLLADV", "PSRRLGFDSLTAVELRNQLAAST", "AFREIGFDSLTAVELRNRLGAAA",
               "SLMEEGLDSLAAVELGGTLQRDT", "GFFDLGMDSLMAVELRRRIEQGV"]


def test_real_protein_abyB1_text_to_hits(pli):
    """The reference's own protein bench fixture (lightmotif/benches/abyB1.txt, 5 781 residues, with the six 23-residue
    sites of lightmotif/benches/score.rs:179-189): real letters through the text ingest (strict, and lossy with foreign
    bytes mixed in), then score / argmax / max / threshold / Scanner on the device against the oracle -- every other
    protein test draws uniform random symbols."""
    text = (Path(__file__).parent / "golden" / "abyB1.txt").read_bytes()
    assert len(text) == 5781
    enc = co.encode(text, "P")
    m = len(ABYB1_SITES[0])
    want_pssm = co.pssm_from_sites([co.encode(x, "P") for x in ABYB1_SITES], 21, 0.1)
    pssm = lm.create(ABYB1_SITES, protein=True).counts.normalize(0.1).log_odds()
    assert np.array_equal(bits(pssm.data[:, :21]), bits(want_pssm[:, :21]))
    ref = co.stripe(enc, 32, 21)
    co.configure_wrap(ref, m - 1)
    want, mi = co.score_rows(ref, want_pssm)
    seq = pli.stripe_ascii(text, protein=True)                         # Encode + Stripe on the device (strict)
    assert np.array_equal(seq.matrix(), ref.data[:ref.rows])
    seq.configure(pssm)
    assert np.array_equal(seq.matrix(), ref.data)
    scores = pli.score(pssm, seq)
    assert scores.max_index == mi == 5781 - m + 1
    assert np.array_equal(bits(scores.matrix()[:, :32]), bits(want[:, :32]))
    assert pli.last_kernel.startswith("score_c32<24") or pli.last_kernel.startswith("score_c32<23")
    assert pli.argmax(scores) == co.argmax(want, 32)
    assert bits(np.float32(pli.max(scores))) == bits(co.max_(want, 32))
    assert pli.score_argmax(pssm, seq)[0] == co.argmax(want, 32)
    finite = want[:, :32][np.isfinite(want[:, :32])]
    for t in (float(np.quantile(finite, 0.99)), 0.0, float(finite.max())):
        wrc = [tuple(map(int, rc)) for rc in co.threshold(want, 32, t)]
        assert [tuple(map(int, rc)) for rc in pli.threshold(scores, t)] == wrc
        frc, fv = pli.score_threshold(pssm, seq, t)
        assert [tuple(map(int, rc)) for rc in frc] == wrc
        assert np.array_equal(bits(fv), bits(np.array([want[r, c] for r, c in wrc], np.float32)))
    # the sites themselves are in the sequence's neighbourhood: the best window scores well above the background
    best = co.argmax(want, 32)
    assert want[best] > 20.0
    # Scanner (scan.rs:150-250): hits in the reference's yield order, and max()
    t = float(np.quantile(finite, 0.995))
    order = no.scanner_collect(want, 32, len(enc), m, t, 256)
    got = [(h.position, np.float32(h.score)) for h in lm.Scanner(pssm, seq, threshold=t)]
    assert len(got) >= 10 and got == [(i, s) for i, s in order]
    top = lm.Scanner(pssm, seq, threshold=t).max_valid()
    assert (top.position, np.float32(top.score)) == no.scanner_max(want, 32, len(enc), m, t)
    # lossy ingest: bytes outside the alphabet (lower case, digits, a newline) become X; strict ingest names the first one
    dirty = bytearray(text)
    for i in (0, 17, 1000, 5780):
        dirty[i] = b"a1\n*"[i % 4]
    with pytest.raises(lm.InvalidSymbol):
        pli.stripe_ascii(bytes(dirty), protein=True)
    lossy = pli.stripe_ascii(bytes(dirty), protein=True, lossy=True)
    enc_l = co.encode(bytes(dirty), "P", lossy=True)
    assert [int(enc_l[i]) for i in (0, 17, 1000, 5780)] == [20] * 4
    ref_l = co.stripe(enc_l, 32, 21)
    co.configure_wrap(ref_l, m - 1)
    lossy.configure(pssm)
    assert np.array_equal(lossy.matrix(), ref_l.data)
    want_l, _ = co.score_rows(ref_l, want_pssm)
    assert np.array_equal(bits(pli.score(pssm, lossy).matrix()[:, :32]), bits(want_l[:, :32]))


@pytest.mark.parametrize("m", [1, 3, 4, 15, 16, 20, 31, 33, 36, 37, 50, 64])
def test_small_score_into_tracks_its_argmax_in_the_same_launch(pli, m):
    """`score_into` + `argmax` on handles below 8 Mi cells (lightmotif-bench dna.rs:104-107 at its own
    size): the store kernel tracks (value, cell) and its last workgroup folds the records
    (MODE_STORE_TRACK), `argmax` / `max` read the pinned record.  Same scores bit for bit, the Generic
    argmax rule on ties / all -inf / NaN first cell / shards without the first-cell rule, over and over
    on one handle (the ticket counter must come back to zero every time)."""
    rng = np.random.default_rng(5000 + m)
    scores = lm.StripedScores.empty(pli, 32)
    for trial, (length, kind) in enumerate([(100_003, "normal"), (464_165, "ties"), (5_000, "normal"), (40 * 32 + m, "ties"),
                                            (200_000, "neg_inf"), (150_001, "nan_first"), (150_001, "nan_elsewhere")]):
        length = max(length, 2 * m + 64)
        enc = rng.integers(0, 4, length, dtype=np.uint8)
        p = random_pssm(rng, m, 5, "ties" if kind == "ties" else "normal")
        if kind == "neg_inf":
            p[:, :5] = -np.inf
        if kind == "nan_first":
            p[0, int(enc[0])] = np.nan
        if kind == "nan_elsewhere":
            enc[0], enc[1:50] = 0, 1
            enc[1000] = 2
            p[0, 2] = np.nan                       # NaN cells exist, but not the first cell (symbol 0)
            enc[enc == 2] = 3
            enc[1000] = 2
        ref = co.stripe(enc, 32, 5)
        co.configure_wrap(ref, m - 1)
        seq = pli.stripe(lm.EncodedSequence(enc), 32)
        seq.configure_wrap(m - 1)
        pssm = lm.ScoringMatrix(p)
        for a, b in ((0, ref.rows), (7, ref.rows - 3)):
            if b - a < 1:
                continue
            want, _ = co.score_rows(ref, p, a, b)
            for rule in (True, False):
                scores.set_first_cell_rule(rule)
                pli.score_rows_into(pssm, seq, range(a, b), scores)
                got_am = pli.argmax(scores)
                assert np.array_equal(bits(scores.matrix()[:, :32]), bits(want[:, :32])), (m, kind, a, b)
                if rule or not np.isnan(want[0, 0]):
                    assert got_am == co.argmax(want, 32), (m, kind, a, b, rule, pli.last_kernel)
                    v = pli.max(scores)
                    assert bits(np.float32(v)) == bits(co.max_(want, 32))
                else:   # a shard that does not hold the matrix's first cell: NaN never wins
                    flat = np.where(np.isnan(want[:, :32]), -np.inf, want[:, :32])
                    r, c = np.argwhere(flat == flat.max())[-1]
                    assert got_am == (int(r), int(c))
                # ... and the separate pass over the stored matrix agrees with the tracked record
                assert pli.argmax_dptr(scores.data_ptr, b - a, 32, 32, first_cell_rule=rule)[0] == got_am
            scores.set_first_cell_rule(True)
        # the fused form of one job folds its records in the kernel as well
        want, _ = co.score_rows(ref, p)
        got = pli.score_argmax(pssm, seq)
        assert got[0] == co.argmax(want, 32)
        assert bits(np.float32(got[1])) == bits(co.max_(want, 32))


@pytest.mark.parametrize("kind", ["normal", "ties", "low_threshold", "protein"])
def test_scanner_max_walk_on_the_device_at_size(pli, kind):
    """`Scanner.max()` = scan.rs:200-249 walked on the device window by window (csrc/scanmax.hip), against the
    oracle's line-by-line restatement on 3 Mbp: many windows, updates inside and across windows, exact ties at
    the maximum (equal scores go to the greater POSITION, not the later cell), a threshold every cell passes."""
    rng = np.random.default_rng(sum(map(ord, kind)))
    protein = kind == "protein"
    k = 21 if protein else 5
    length, m = 3_000_017, 11
    enc = rng.integers(0, k - 1, length, dtype=np.uint8)
    p = np.zeros((m, co.stride(k, 4)), np.float32)
    p[:, :k] = rng.integers(-2, 3, (m, k)) if kind == "ties" else rng.normal(0, 2, (m, k))
    p[:, k - 1] = -np.inf
    ref = co.stripe(enc, 32, k)
    co.configure_wrap(ref, m - 1)
    scores, _ = co.score_rows(ref, p)
    w, factor, offsets, offset = no.to_discrete(p, k)
    d = no.score_rows_u8_saturating(ref.data, 32, length, w, 0, ref.rows)
    scale = lambda x: no.discrete_scale(x, factor, offset)   # noqa: E731
    finite = scores[:, :32][np.isfinite(scores[:, :32])]
    t = -1e30 if kind == "low_threshold" else float(np.quantile(finite, 0.9999))
    pssm = lm.ScoringMatrix(p, protein=protein)
    seq = pli.stripe(lm.EncodedSequence(enc, protein=protein), 32)
    seq.configure(pssm)
    want = no.scanner_max_strict(scores, d, 32, t, scale, 256)
    # both routes of csrc/scanmax.hip: the walk over a candidate LIST (one flag scan at the starting level + records in
    # row-major order walked by the host; DNA, rare candidates) and the walk over windows of materialised u8 scores
    for by_list in (1, 0):
        pli.set_option("list_scan_max", by_list)
        try:
            got = lm.Scanner(pssm, seq, threshold=t).max()
        finally:
            pli.set_option("list_scan_max", 1)
        listed = by_list and not protein and kind != "low_threshold"
        assert pli.last_kernel.startswith("score_c32_prefilter2+scanmax_gate" if listed else "scanmax_find"), (kind, by_list, pli.last_kernel)
        assert got is not None and (got.position, np.float32(got.score)) == (want[0], want[1]), (kind, by_list, got, want)


@pytest.mark.parametrize("kind", ["poly_a", "repeat", "random_then_poly_a", "poly_a_2mbp"])
def test_scanner_max_on_degenerate_sequences(pli, kind):
    """Homopolymers and low-complexity repeats: "equal score at a greater position replaces the best" (scan.rs:237) then
    fires once per ROW, not ~ln n times.  The device walk consumes such runs inside its update kernel (a wavefront walking
    on behind every found cell) instead of one search + update launch pair per row; same answer as the reference's
    single pass, in bounded time."""
    import time
    rng = np.random.default_rng(sum(map(ord, kind)))
    k, m = 5, 12
    length = 2_000_003 if kind == "poly_a_2mbp" else 600_011
    if kind.startswith("poly_a"):
        enc = np.zeros(length, np.uint8)
    elif kind == "repeat":
        enc = np.tile(np.array([0, 1, 3], np.uint8), length // 3 + 1)[:length]
    else:
        enc = rng.integers(0, 4, length, dtype=np.uint8)
        enc[length // 3:] = 0
    p = np.zeros((m, co.stride(k, 4)), np.float32)
    p[:, :k] = rng.normal(0, 2, (m, k))
    p[:, 0] += 1.0                                             # the repeated letter scores well: every cell is a candidate
    p[:, k - 1] = -np.inf
    ref = co.stripe(enc, 32, k)
    co.configure_wrap(ref, m - 1)
    scores, _ = co.score_rows(ref, p)
    w, factor, offsets, offset = no.to_discrete(p, k)
    d = no.score_rows_u8_saturating(ref.data, 32, length, w, 0, ref.rows)
    scale = lambda x: no.discrete_scale(x, factor, offset)   # noqa: E731
    t = float(np.quantile(scores[:, :32][np.isfinite(scores[:, :32])], 0.5))
    pssm = lm.ScoringMatrix(p)
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure(pssm)
    lm.Scanner(pssm, seq, threshold=t).max()                   # warm (tables, buffers)
    t0 = time.perf_counter()
    got = lm.Scanner(pssm, seq, threshold=t).max()
    dt = time.perf_counter() - t0
    if kind != "poly_a_2mbp":                                  # (the oracle's walk is a Python loop over every candidate)
        want = no.scanner_max_strict(scores, d, 32, t, scale, 256)
        assert got is not None and (got.position, np.float32(got.score)) == (want[0], want[1]), (got, want)
    else:                                                      # constant scores: the last valid-or-not cell in position order wins
        flat_best = scores[:, :32].max()
        pos = np.argwhere(scores[:, :32] == flat_best)
        want_pos = int((pos[:, 1] * ref.rows + pos[:, 0]).max())
        assert got.position == want_pos and np.float32(got.score) == flat_best
    assert dt < 3.0, f"{dt:.2f} s for {length} positions"


@pytest.mark.parametrize("threshold", ["p1e-4", "every_cell"])
def test_scanner_max_walk_batched_windows_stall_and_resume(pli, threshold):
    """The walk beyond its first windows (csrc/scanmax.hip): windows of 65 536 rows and more are enqueued in batches
    with three search / update rounds each and no wait in between; a window that needs more rounds stalls the batch
    and is finished under the host's eyes.  8 Mbp with sites of rising strength planted inside ONE late window
    (five records in a row: more than three rounds) and again in the next one, against the oracle's line-by-line
    restatement of scan.rs:200-249."""
    rng = np.random.default_rng(20_260_929)
    length, m, k = 8_000_003, 20, 5
    rows = -(-length // 32)
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    p = np.zeros((m, co.stride(k, 4)), np.float32)
    p[:, :4] = rng.normal(-1, 1, (m, 4))
    p[np.arange(m), rng.integers(0, 4, m)] = 3.0       # a consensus far above anything random: every planted site is a record
    p[:, k - 1] = -np.inf
    order = np.argsort(p[:, :4], axis=1)
    best_sym, second_sym, worst_sym = order[:, 3], order[:, 2], order[:, 0]

    def plant(row, col, mismatches, sub):  # cell (row, col) <-> position col * rows + row (pli/mod.rs:191-193)
        site = best_sym.copy()
        site[:mismatches] = sub[:mismatches]
        enc[col * rows + row: col * rows + row + m] = site
    for i, mm in enumerate((6, 5, 4, 3)):              # window of rows 61 440 ... 126 975 (the first batched one): four records
        plant(70_000 + 40 * i, 0, mm, worst_sym)
    for i, mm in enumerate((3, 2, 1, 0)):              # the window behind it: four more, each above the last
        plant(200_000 + 40 * i, 3, mm, second_sym)
    ref = co.stripe(enc, 32, k)
    co.configure_wrap(ref, m - 1)
    scores, _ = co.score_rows(ref, p)
    w, factor, offsets, offset = no.to_discrete(p, k)
    d = no.score_rows_u8_saturating(ref.data, 32, length, w, 0, ref.rows)
    scale = lambda x: no.discrete_scale(x, factor, offset)   # noqa: E731
    finite = scores[:, :32][np.isfinite(scores[:, :32])]
    t = -1e30 if threshold == "every_cell" else float(np.quantile(finite, 0.9999))
    pssm = lm.ScoringMatrix(p)
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure(pssm)
    want = no.scanner_max_strict(scores, d, 32, t, scale, 256)
    pli.set_option("list_scan_max", 0)                 # the window walk, whatever the threshold
    try:
        got = lm.Scanner(pssm, seq, threshold=t).max()
    finally:
        pli.set_option("list_scan_max", 1)
    assert pli.last_kernel == "scanmax_find (a batched window resumed)"     # the stall path was taken
    assert got is not None and (got.position, np.float32(got.score)) == (want[0], want[1]), (got, want)
    got = lm.Scanner(pssm, seq, threshold=t).max()     # the default route: the candidate list where candidates are rare
    assert pli.last_kernel.startswith("score_c32_prefilter2+scanmax_gate" if threshold == "p1e-4" else "scanmax_find")
    assert got is not None and (got.position, np.float32(got.score)) == (want[0], want[1]), (got, want)
    # a scanner that has already yielded part of its hits walks on from where it stands (scan.rs:200-215)
    sc = lm.Scanner(pssm, seq, threshold=float(np.quantile(finite, 0.99999)), block_size=4096)
    it = iter(sc)
    for _ in range(3):
        next(it)
    host = lm.Scanner(pssm, seq, threshold=float(np.quantile(finite, 0.99999)), block_size=4096)
    ith = iter(host)
    for _ in range(3):
        next(ith)
    a, b = sc.max(), scanner_max_strict_host(host)
    assert (a is None) == (b is None) and (a is None or (a.position, a.score) == (b.position, b.score))


@pytest.mark.parametrize("cols", [1, 16, 33])
@pytest.mark.parametrize("m", [5, 15, 20, 33])
def test_small_score_into_tracks_its_argmax_off_the_c32_kernels(pli, cols, m):
    """The same one-launch flow where `score_tiled` writes the scores -- C = 1 is the Generic geometry of the
    reference's bench (dna.rs:113-116), C = 16 / 33 odd layouts: per-wavefront records into pinned memory, folded
    by the host; ties, all -inf, NaN first cell, row ranges, the handle re-used."""
    rng = np.random.default_rng(7000 + 100 * cols + m)
    scores = lm.StripedScores.empty(pli, cols)
    for length, kind in ((60_011, "normal"), (120_000, "ties"), (30_000, "neg_inf"), (45_001, "nan_first")):
        enc = rng.integers(0, 4, length, dtype=np.uint8)
        p = random_pssm(rng, m, 5, "ties" if kind == "ties" else "normal")
        if kind == "neg_inf":
            p[:, :5] = -np.inf
        if kind == "nan_first":
            p[0, int(enc[0])] = np.nan
        ref = co.stripe(enc, cols, 5)
        co.configure_wrap(ref, m - 1)
        seq = pli.stripe(lm.EncodedSequence(enc), cols)
        seq.configure_wrap(m - 1)
        pssm = lm.ScoringMatrix(p)
        for a, b in ((0, ref.rows), (3, ref.rows - 2)):
            want, _ = co.score_rows(ref, p, a, b)
            pli.score_rows_into(pssm, seq, range(a, b), scores)
            got = pli.argmax(scores)
            assert np.array_equal(bits(scores.matrix()[:, :cols]), bits(want[:, :cols])), (cols, m, kind)
            assert got == co.argmax(want, cols), (cols, m, kind, a, b, pli.last_kernel)
            assert bits(np.float32(pli.max(scores))) == bits(co.max_(want, cols))


def test_polled_small_argmax_under_stress():
    """`tools/stress_small_argmax.py` for a few seconds: the polled per-wavefront records (tracked `score_into` +
    `argmax`, fused `score_argmax`) alternate between inputs with different answers on shared handles; a stale or
    torn record would show as a wrong cell (2 M iterations clean on the builder's run)."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "stress_small_argmax.py"), "4"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("cols", [16, 1, 33])
def test_scanner_max_walk_on_other_column_counts(pli, cols):
    """The device walk of `Scanner::max` on striped matrices of 16 / 1 / 33 columns (generic u8 and f32 store
    kernels, dense window buffers) against the host walk of the downloaded matrices."""
    rng = np.random.default_rng(300 + cols)
    length, m = 40_003, 9
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    p = random_pssm(rng, m, 5)
    pssm = lm.ScoringMatrix(p)
    seq = pli.stripe(lm.EncodedSequence(enc), cols)
    seq.configure(pssm)
    mat = pli.score(pssm, seq).matrix()[:, :cols]
    for q in (0.999, 0.5):
        t = float(np.quantile(mat[np.isfinite(mat)], q))
        got = lm.Scanner(pssm, seq, threshold=t).max()
        host = scanner_max_strict_host(lm.Scanner(pssm, seq, threshold=t))
        assert (got is None) == (host is None)
        assert got is None or (got.position, got.score) == (host.position, host.score), (cols, q, got, host)
