"""The CPU oracle against every literal vector the reference's own tests hold for
the scoring path (tests/golden/reference_vectors.json), the independent numpy
restatement, the committed generated fixtures and the AVX2 baseline port."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import np_oracle as no

GOLD = json.loads((Path(__file__).parent / "golden" / "reference_vectors.json").read_text())
CASES = np.load(Path(__file__).parent / "golden" / "generated_cases.npz")
CASE_NAMES = sorted({k.split("/")[0] for k in CASES.files})
DNA = "ACTGN"


def golden_setup(cols):
    g = GOLD["G1_scores"]
    s = co.stripe(co.encode(g["sequence"]), cols)
    pssm = co.pssm_from_sites([co.encode(p) for p in g["patterns"]], pseudocount=g["pseudocount"])
    co.configure_wrap(s, pssm.shape[0] - 1)
    return g, s, pssm


@pytest.mark.parametrize("cols", [32, 1, 16])
def test_g1_scores(cols):
    g, s, pssm = golden_setup(cols)
    scores, mi = co.score_rows(s, pssm)
    u = co.unstripe(scores, cols, mi)
    assert len(u) == g["unstripe_len"]                               # tests/dna.rs:81
    assert np.abs(u - np.float32(g["expected"])).max() < g["tolerance"]  # tests/dna.rs:83-90
    for i, v in g["exact"].items():                                  # tests/dna.rs:57-58
        assert u[int(i)] == np.float32(v)
    # score_rows_into(0..2) and (1..2): tests/dna.rs:55-62
    sc, _ = co.score_rows(s, pssm, 0, 2)
    assert sc.shape[0] == 2 and sc[0, 0] == np.float32(g["exact"]["0"])
    if s.rows > 1:
        assert sc[1, 0] == np.float32(g["exact"]["1"])
        sc, _ = co.score_rows(s, pssm, 1, 2)
        assert sc.shape[0] == 1 and sc[0, 0] == np.float32(g["exact"]["1"])
    # score_position: tests/dna.rs:175-199
    for i in range(50):
        assert abs(co.score_position(s, pssm, i) - g["expected"][i]) < 1e-5


@pytest.mark.parametrize("cols", [32, 1, 16])
def test_g2_g3_argmax_threshold(cols):
    _, s, pssm = golden_setup(cols)
    scores, _ = co.score_rows(s, pssm)
    am = co.argmax(scores, cols)
    assert co.offset(scores.shape[0], *am) == GOLD["G2_argmax"]["offset"]
    assert co.max_(scores, cols) == scores[am]
    for case in GOLD["G3_threshold"]["cases"]:
        rc = co.threshold(scores, cols, case["t"])
        offs = sorted(co.offset(scores.shape[0], int(r), int(c)) for r, c in rc)
        assert offs == case["sorted_offsets"]


def test_g4_stripe_literals():
    g = GOLD["G4_stripe"]
    enc = co.encode(g["sequence"])
    s4 = co.stripe(enc, 4)
    assert ["".join(DNA[x] for x in row[:4]) for row in s4.data] == g["c4_rows"]
    s2 = co.stripe(enc, 2)
    assert ["".join(DNA[x] for x in row[:2]) for row in s2.data] == g["c2_rows"]
    co.configure_wrap(s4, 2)
    assert ["".join(DNA[x] for x in row[:4]) for row in s4.data] == g["c4_wrap2_rows"]


def test_g5_scanner_hits_by_brute_force():
    """scan.rs:185-190: hits = positions with score >= t and pos + M <= L."""
    _, s, pssm = golden_setup(32)
    scores, mi = co.score_rows(s, pssm)
    u = co.unstripe(scores, 32, mi)
    assert (u >= 0.0).sum() == GOLD["G5_scanner"]["threshold_0_hits"]
    hits = [(i, float(u[i])) for i in np.flatnonzero(u >= -10.0)]
    want = GOLD["G5_scanner"]["threshold_m10"]
    assert [h[0] for h in hits] == [w["position"] for w in want]
    for h, w in zip(hits, want):
        assert abs(h[1] - w["score"]) < 1e-5


def test_g5_scanner_restatement_yield_order_and_max():
    """`Scanner::next` / `Scanner::max` restated (scan.rs:166-249): the golden hit set
    (scan.rs:290-297, 311-314) whatever the block size, yielded block by block with the last
    pushed hit first, and the best hit of scan.rs:327-333."""
    _, s, pssm = golden_setup(32)
    scores, _ = co.score_rows(s, pssm)
    want = GOLD["G5_scanner"]["threshold_m10"]
    length, m = len(GOLD["G1_scores"]["sequence"]), pssm.shape[0]
    for block_size in (256, 1, 2):
        seq = no.scanner_collect(scores, 32, length, m, -10.0, block_size)
        assert sorted(i for i, _ in seq) == [w["position"] for w in want]
    # 64 bp at C = 32: 2 rows; the hits are cells (0, 9), (1, 13), (0, 16)
    assert [i for i, _ in no.scanner_collect(scores, 32, length, m, -10.0, 256)] == [27, 32, 18]
    assert [i for i, _ in no.scanner_collect(scores, 32, length, m, -10.0, 1)] == [32, 18, 27]
    assert no.scanner_collect(scores, 32, length, m, 0.0) == []
    best = no.scanner_max(scores, 32, length, m, -10.0)
    assert best[0] == 18 and abs(float(best[1]) - (-5.50167)) < 1e-5
    assert no.scanner_max(scores, 32, length, m, 0.0) is None


def test_discrete_matrix_restatement_on_the_golden_pssm():
    """pwm/mod.rs:665-696 + tests/dna.rs:93-120 (`test_score_discrete`): the u8 scores of the
    golden PSSM's DiscreteMatrix, unscaled, are >= the 50 expected f32 scores; the Generic
    (wrapping) and SIMD (saturating) bodies agree here; scale() under-estimates."""
    g = GOLD["G1_scores"]
    _, s, pssm = golden_setup(32)
    weights, factor, offsets, offset = no.to_discrete(pssm, 5)
    assert weights.shape == (pssm.shape[0], 5) and weights.max() <= 255
    assert (weights[:, 4] == 0).all()                      # N scores -inf -> `as u8` = 0
    u8, mi = co.score_rows_u8(s, weights)
    assert mi == len(g["expected"])
    rows = u8.shape[0]
    for i, want in enumerate(g["expected"]):
        assert no.discrete_unscale(int(u8[i % rows, i // rows]), factor, offset) >= want
    sat = no.score_rows_u8_saturating(s.data, 32, len(g["sequence"]), weights, 0, s.rows)
    assert np.array_equal(sat, u8[:, :32])
    f32, _ = co.score_rows(s, pssm)
    for t in (-10.0, -15.0, 0.0):                          # scan.rs:169-190: no false negatives
        assert (u8[:, :32][f32[:, :32] >= np.float32(t)] >= no.discrete_scale(t, factor, offset)).all()


def test_g6_stride_table():
    for c in GOLD["G6_stride"]["cases"]:
        assert co.stride(c["cols"], c["elem"]) == c["stride"]
        assert no.stride(c["cols"], c["elem"]) == c["stride"]
    assert co.stride(5, 4) == 8 and co.stride(21, 4) == 24


def test_g7_empty_row_range():
    g = GOLD["G7_empty_range"]
    s = co.stripe(co.encode(g["sequence"]), g["columns"])
    pssm = co.pssm_from_sites([co.encode(p) for p in g["patterns"]], pseudocount=g["pseudocount"])
    co.configure_wrap(s, pssm.shape[0] - 1)
    sc, mi = co.score_rows(s, pssm, *g["rows"])
    assert sc.shape[0] == g["expected_rows"] and mi == 0


def test_g9_encode():
    g = GOLD["G9_encode"]
    assert "".join(DNA[x] for x in co.encode(g["sequence"])) == g["sequence"]
    with pytest.raises(ValueError, match=r"'\.'"):
        co.encode(g["unknowns"])
    lossy = co.encode(g["unknowns"], lossy=True)
    assert "".join(DNA[x] for x in lossy) == g["unknowns"].replace(".", "N")


def test_short_sequence_is_empty():
    s = co.stripe(co.encode("ACGT"), 32)
    pssm = co.pssm_from_sites([co.encode("ACGTAC")])
    co.configure_wrap(s, 5)
    sc, mi = co.score_rows(s, pssm)
    assert sc.shape[0] == 0 and mi == 0
    assert co.argmax(sc, 32) is None


def test_argmax_rules_ties_nan():
    """pli/mod.rs:135-155: last maximal cell in (row, col) order; NaN never wins."""
    sc = np.zeros((3, 32), np.float32)
    assert co.argmax(sc, 32) == (2, 31) == no.argmax(sc, 32)
    sc[1, 5] = np.nan
    assert co.argmax(sc, 32) == (2, 31)
    sc[0, 0] = np.nan           # scores[0] is NaN: nothing ever compares >= NaN
    assert co.argmax(sc, 32) == (0, 0) == no.argmax(sc, 32)
    sc[:] = -np.inf
    assert co.argmax(sc, 32) == (2, 31)
    sc[1, 7] = 3.0
    sc[1, 9] = 3.0
    sc[0, 30] = 3.0
    assert co.argmax(sc, 32) == (1, 9)


@pytest.mark.parametrize("name", CASE_NAMES)
def test_generated_fixtures_match_both_restatements(name):
    c = {k.split("/", 1)[1]: CASES[k] for k in CASES.files if k.startswith(name + "/")}
    cols, k = int(c["cols"]), int(c["k"])
    enc, pssm = c["encoded"], c["pssm"]
    m = pssm.shape[0]
    s = co.stripe(enc, cols, k)
    co.configure_wrap(s, int(c["wrap"]))
    assert np.array_equal(s.data, c["striped"])
    a, b = map(int, c["row_range"])
    sc, mi = co.score_rows(s, pssm, a, b)
    assert mi == int(c["max_index"])
    assert np.array_equal(sc.view(np.uint32), c["scores"].view(np.uint32))
    sc2, _ = no.score_rows(c["striped"], cols, len(enc), pssm, a, b)
    assert np.array_equal(sc2.view(np.uint32), c["scores"].view(np.uint32))
    am = co.argmax(sc, cols)
    assert (am if am else (-1, -1)) == tuple(c["argmax"])
    for i, t in enumerate(c["thresholds"]):
        assert np.array_equal(co.threshold(sc, cols, float(t)).astype(np.int64), c[f"threshold_{i}"])
    assert m == pssm.shape[0]


def test_avx2_port_matches_generic_scores_bitwise():
    """avx2.rs:104-199 vs pli/mod.rs:96-105: identical f32 add order."""
    rng = np.random.default_rng(7)
    for k, m in ((5, 20), (5, 7), (21, 12)):
        enc = rng.integers(0, k, size=70001, dtype=np.uint8)
        s = co.stripe(enc, 32, k)
        pssm = np.zeros((m, co.stride(k, 4)), np.float32)
        pssm[:, :k] = rng.normal(0, 3, (m, k))
        pssm[:, k - 1] = -np.inf
        al = co.aligned_empty(pssm.shape, np.float32)
        al[:] = pssm
        with pytest.raises(RuntimeError, match="not enough wrapping rows"):  # avx2.rs:832-837
            co.avx2_score_rows(s, al)
        co.configure_wrap(s, m - 1)
        want, _ = co.score_rows(s, al)
        for threads in (1, 3):
            got = co.avx2_score_rows(s, al, threads=threads)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        got = co.avx2_score_rows(s, al, row_begin=5, row_end=90)
        assert np.array_equal(got.view(np.uint32), want[5:90].view(np.uint32))


def test_avx2_argmax_agrees_when_maximum_is_unique():
    """avx2.rs:351-426 has a different tie rule but must agree with
    Generic whenever the maximum is unique (tests/argmax.rs:41-52)."""
    rng = np.random.default_rng(11)
    sc = co.aligned_empty((500, 32), np.float32)
    sc[:] = rng.normal(0, 1, sc.shape)
    sc[123, 17] = 99.0
    assert co.avx2_argmax(sc, 500 * 32) == co.argmax(sc, 32) == (123, 17)


def test_g5_scanner_max_strict_restatement():
    """`Scanner::max` restated literally (scan.rs:200-249, u8 DiscreteMatrix steering included):
    on the reference's own test (scan.rs:327-333) it finds the same best hit (18, -5.50167) as the
    quirk-free restatement; and the two documented quirks show up where they must."""
    _, s, pssm = golden_setup(32)
    scores, _ = co.score_rows(s, pssm)
    w, factor, offsets, offset = no.to_discrete(pssm, 5)
    d = no.score_rows_u8_saturating(s.data, 32, len(GOLD["G1_scores"]["sequence"]), w, 0, s.rows)
    scale = lambda x: no.discrete_scale(x, factor, offset)   # noqa: E731
    best = no.scanner_max_strict(scores, d, 32, -10.0, scale)
    assert best[0] == 18 and abs(float(best[1]) - (-5.50167)) < 1e-5
    for bs in (1, 2):
        assert no.scanner_max_strict(scores, d, 32, -10.0, scale, block_size=bs)[0] == 18
    # quirk (b): nothing reaches t = 0.0 in f32 (test_scanner.py:71-72), but the first cell whose u8
    # score reaches scale(0.0) is accepted without the f32 test (scan.rs:240-242)
    t = 0.0
    assert no.scanner_max(scores, 32, len(GOLD["G1_scores"]["sequence"]), pssm.shape[0], t) is None
    strict = no.scanner_max_strict(scores, d, 32, t, scale)
    reach = np.argwhere(d[:, :32] >= scale(np.float32(t)))
    if reach.size:
        assert strict is not None and float(strict[1]) < t
    else:
        assert strict is None
