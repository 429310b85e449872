"""`Score<u8, ..>` / `Maximum<u8, ..>` / `Threshold<u8, ..>` with a DiscreteMatrix (the pass
the reference's Scanner runs first, scan.rs:174-184) against the oracle: bit-exact u8 scores for
both overflow behaviours (Generic's wrapping `+=`, the SIMD back-ends' saturating adds), the
reference's own property test (tests/dna.rs:93-120), odd geometries through the generic kernel,
and BASELINE's full size through window samples."""
import os

import numpy as np
import pytest
import torch

import lightmotif_amd as lm
from oracle import c_oracle as co
from oracle import np_oracle as no

pytestmark = pytest.mark.gpu
PAIRS = True   # the DNA pair-symbol scan is the shipped route (option "pair_prefilter" = 0: one symbol per lookup)
GOLD_SEQ = "ATGTCCCAACAACGATACCCCGAGCCCATCGCCGTCATCGGCTCGGCATGCAGATTCCCAGGCG"
PATTERNS = ["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"]


def last_max(scores: np.ndarray, cols: int):
    """pli/mod.rs:135-155 for u8: the maximal cell that is last in (row, col) order."""
    flat = scores[:, :cols]
    r, c = np.argwhere(flat == flat.max())[-1]
    return (int(r), int(c)), int(flat.max())


def test_golden_scores_over_estimate_like_the_reference_test(pli):
    """tests/dna.rs:93-120 `test_score_discrete`: unscale(u8 score) >= f32 score everywhere."""
    import json
    from pathlib import Path
    gold = json.loads((Path(__file__).parent / "golden" / "reference_vectors.json").read_text())["G1_scores"]
    pssm = lm.create(gold["patterns"]).counts.normalize(0.1).log_odds()
    dm = pssm.to_discrete()
    seq = lm.stripe(gold["sequence"])
    seq.configure(pssm)
    scores, max_index = pli.score_discrete(dm, seq)
    assert max_index == len(gold["expected"]) and scores.shape == (seq.rows, 32)
    rows = scores.shape[0]
    for i, want in enumerate(gold["expected"]):
        assert dm.unscale(int(scores[i % rows, i // rows])) >= want
    # and bit-for-bit what the oracle computes, for both overflow behaviours
    enc = co.encode(gold["sequence"])
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, len(pssm))
    want_wrap, mi = co.score_rows_u8(ref, dm.data)
    assert mi == max_index
    assert np.array_equal(pli.score_discrete(dm, seq, saturate=False)[0][:, :32], want_wrap[:, :32])
    want_sat = no.score_rows_u8_saturating(ref.data, 32, len(enc), dm.data[:, :5], 0, ref.rows)
    assert np.array_equal(scores[:, :32], want_sat)


CASES = [  # (length, m, protein, weight ceiling, columns)
    (1000, 1, False, 255, 32), (5000, 2, False, 255, 32), (40_000, 7, False, 40, 32),
    (40_000, 8, False, 255, 32), (100_003, 15, False, 17, 32), (100_003, 20, False, 255, 32),
    (65_000, 33, False, 7, 32), (65_000, 36, False, 255, 32), (30_000, 12, True, 21, 32),
    (30_000, 12, True, 255, 32), (3000, 9, False, 28, 16), (3000, 9, False, 255, 1),
    (20_000, 40, False, 255, 32), (64, 15, False, 17, 32), (40, 36, False, 255, 32),
    (300_007, 37, False, 6, 32), (300_007, 73, False, 255, 32), (100_003, 100, False, 3, 32), (50_000, 64, True, 4, 32),
]


@pytest.mark.parametrize("length,m,protein,top,cols", CASES)
def test_u8_scores_reductions_and_row_ranges(pli, length, m, protein, top, cols):
    rng = np.random.default_rng(length * 131 + m * 7 + cols)
    k = 21 if protein else 5
    enc = rng.integers(0, k if m % 2 else k - 1, length, dtype=np.uint8)
    weights = rng.integers(0, top + 1, (m, k), dtype=np.uint8)
    dm = lm.DiscreteMatrix(weights, 1.0, np.zeros(m, np.float32), 0.0, protein=protein)
    ref = co.stripe(enc, cols, k)
    co.configure_wrap(ref, m)
    seq = pli.stripe(lm.EncodedSequence(enc, protein=protein), cols)
    seq.configure_wrap(m - 1)
    rows = ref.rows
    want_wrap, mi = co.score_rows_u8(ref, weights)
    want_sat = no.score_rows_u8_saturating(ref.data, cols, length, weights, 0, rows) if length >= m else want_wrap
    if top == 255 and m > 1 and length >= m:
        assert not np.array_equal(want_wrap[:, :cols], want_sat)   # the two behaviours really differ
    for saturate, want in ((True, want_sat), (False, want_wrap)):
        got, gmi = pli.score_discrete(dm, seq, saturate=saturate)
        assert gmi == mi and got.shape[0] == want.shape[0]
        assert np.array_equal(got[:, :cols], want[:, :cols]), (saturate, pli.last_kernel)
        for a, b in ((1, rows), (rows // 3, rows // 3 + 1), (rows // 2, rows - 1), (2, 2)):
            if 0 <= a <= b <= rows and got.shape[0]:
                part, _ = pli.score_discrete(dm, seq, rows=range(a, b), saturate=saturate)
                assert np.array_equal(part[:, :cols], want[a:b, :cols])
    if cols == 32 and 1 <= m <= 36 and rows >= (m | 3) + 3:
        assert pli.last_kernel == ("score_c32_u8_pairs" if PAIRS and not protein and m >= 2 else "score_c32_u8")
    elif cols == 32 and m > 36 and rows // 2 > 40:
        assert pli.last_kernel == "score_c32_u8_sliced"     # slices of <= 36 rows through the fast kernels, added bytewise
    elif cols != 32:
        assert pli.last_kernel == "score_generic_u8"
    if want_sat.shape[0] == 0:
        return
    # Maximum<u8> / Threshold<u8> on the device copy of the saturating scores
    dev = torch.from_numpy(np.ascontiguousarray(want_sat[:, :cols])).cuda()
    assert pli.argmax_u8_dptr(dev.data_ptr(), dev.shape[0], cols, cols) == last_max(want_sat, cols)
    for t in sorted({0, 1, int(want_sat.max()), int(np.median(want_sat)), 255}):
        want_hits = np.argwhere(want_sat[:, :cols] >= t)
        got_hits = pli.threshold_u8_dptr(dev.data_ptr(), dev.shape[0], cols, cols, t)
        assert np.array_equal(got_hits, want_hits), t
    # padded rows (stride > columns) take the strided paths
    padded = torch.zeros((dev.shape[0], cols + 5), dtype=torch.uint8, device="cuda")
    padded[:, :cols] = dev
    padded[:, cols:] = 255
    torch.cuda.synchronize()   # torch's stream filled it; the pipeline runs on its own stream
    assert pli.argmax_u8_dptr(padded.data_ptr(), dev.shape[0], cols + 5, cols) == last_max(want_sat, cols)
    t = int(want_sat.max())
    assert np.array_equal(pli.threshold_u8_dptr(padded.data_ptr(), dev.shape[0], cols + 5, cols, t),
                          np.argwhere(want_sat[:, :cols] >= t))


@pytest.mark.parametrize("m", [2, 3, 4, 9, 20, 23, 36])
def test_one_symbol_scan_and_unaligned_buffers(m):
    """The DNA pair-symbol scan can be switched off (option "pair_prefilter" = 0 -> one symbol per
    lookup) and needs 4-byte aligned matrices; every route gives the same bytes."""
    single = lm.Pipeline.hip(0)
    single.set_option("pair_prefilter", 0)
    paired = lm.Pipeline.hip(0)
    rng = np.random.default_rng(m)
    length = 70_001
    rows = -(-length // 32)
    enc = rng.integers(0, 5, length, dtype=np.uint8)
    weights = rng.integers(0, 256, (m, 5), dtype=np.uint8)
    dm = lm.DiscreteMatrix(weights, 1.0, np.zeros(m, np.float32), 0.0)
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, m)
    want = no.score_rows_u8_saturating(ref.data, 32, length, weights, 0, rows)
    base = torch.zeros((rows + m - 1) * 32 + 64, dtype=torch.uint8, device="cuda")
    outb = torch.zeros(rows * 32 + 64, dtype=torch.uint8, device="cuda")
    host = torch.from_numpy(np.ascontiguousarray(ref.data[: rows + m - 1, :32]).reshape(-1))
    for pli, name in ((paired, "score_c32_u8_pairs"), (single, "score_c32_u8")):
        for so, oo in ((0, 0), (1, 0), (0, 2), (3, 1)):
            base[so: so + host.numel()] = host.cuda()
            outb.zero_()
            torch.cuda.synchronize()
            pli.score_u8_dptr(dm, base.data_ptr() + so, rows + m - 1, 32, 32, m - 1, length, 0, rows,
                              outb.data_ptr() + oo, 32)
            torch.cuda.synchronize()
            got = outb[oo: oo + rows * 32].cpu().numpy().reshape(rows, 32)
            assert np.array_equal(got, want), (name, so, oo, pli.last_kernel)
            if oo % 4:
                assert pli.last_kernel == "score_generic_u8"
            elif so % 4 == 0:
                assert pli.last_kernel == name
            else:
                assert pli.last_kernel == "score_c32_u8"   # byte symbol loads need no alignment


@pytest.mark.parametrize("m,protein", [(1, False), (2, False), (3, False), (1, True), (2, True), (3, True)])
def test_short_motif_u8_stores_agree_run_after_run(m, protein):
    """The shortest motifs through the u8 store kernels at a size that fills the chip (4 M cells),
    pair and one-symbol routes, each run twice: the same bytes every time, and the oracle's.  (The
    look-ahead of those kernels at M = 1, 2 was once capped for a fault whose cause turned out to be
    the 64-bit shift erratum of DESIGN 4.9; this is the run-to-run check that stands behind
    lifting the cap.)"""
    k = 21 if protein else 5
    rng = np.random.default_rng(900 + m + 10 * k)
    length = (1 << 22) + 77
    rows = -(-length // 32)
    enc = rng.integers(0, k, length, dtype=np.uint8)
    weights = rng.integers(0, 256, (m, k), dtype=np.uint8)
    dm = lm.DiscreteMatrix(weights, 1.0, np.zeros(m, np.float32), 0.0, protein=protein)
    ref = co.stripe(enc, 32, k)
    co.configure_wrap(ref, m)
    want = no.score_rows_u8_saturating(ref.data, 32, length, weights, 0, rows)
    dev = torch.from_numpy(np.ascontiguousarray(ref.data[: rows + m - 1, :32])).cuda()
    out = torch.empty((rows, 32), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    routes = []
    for pairs in (1, 0):
        pli = lm.Pipeline.hip(0)
        pli.set_option("pair_prefilter", pairs)
        for run in range(2):
            out.fill_(0x5A)
            torch.cuda.synchronize()
            pli.score_u8_dptr(dm, dev.data_ptr(), rows + m - 1, 32, 32, m - 1, length, 0, rows, out.data_ptr(), 32)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want), (pairs, run, pli.last_kernel)
        routes.append(pli.last_kernel)
    assert "score_generic_u8" not in routes, routes


def test_scanner_prefilter_property_on_discrete_scores(pli):
    """scan.rs:169-190: every position with f32 score >= t has a u8 score >= scale(t)."""
    rng = np.random.default_rng(5)
    length, m = 200_000, 11
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(8)]
    pssm = lm.create(sites).counts.normalize(0.1).log_odds()
    dm = pssm.to_discrete()
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure(pssm)
    f32 = pssm.score(seq).matrix()[:, :32]
    u8, _ = pli.score_discrete(dm, seq)
    for t in (-5.0, 0.0, 4.0, float(f32[np.isfinite(f32)].max())):
        assert (u8[:, :32][f32 >= np.float32(t)] >= dm.scale(t)).all()
    finite = np.isfinite(f32)
    unscaled = u8[:, :32].astype(np.float32) * np.float32(dm.factor) + np.float32(dm.offset)
    assert (unscaled[finite] >= f32[finite]).all()              # pwm/mod.rs:745-751 doc test


def test_errors_and_degenerate_inputs(pli):
    dm = lm.DiscreteMatrix(np.ones((4, 8), np.uint8), 1.0, np.zeros(4, np.float32), 0.0)
    seq = pli.stripe(lm.EncodedSequence("ACGTACGTACGT"), 32)
    with pytest.raises(lm.LightmotifHipError, match="not enough wrapping rows"):
        pli.score_discrete(dm, seq)                               # avx2.rs:832-837
    seq.configure_wrap(3)
    got, mi = pli.score_discrete(dm, seq, rows=range(1, 1))       # pli/mod.rs:85-88
    assert got.shape[0] == 0 and mi == 0
    short = pli.stripe(lm.EncodedSequence("ACG"), 32)
    short.configure_wrap(3)
    got, mi = pli.score_discrete(dm, short)
    assert got.shape[0] == 0 and mi == 0
    assert pli.argmax_u8_dptr(0, 0, 32, 32) is None               # pli/mod.rs:136-138
    assert pli.threshold_u8_dptr(0, 0, 32, 32, 3).shape == (0, 2)


def test_full_size_u8_scores(pli):
    """1 Gbp x len-20 (BASELINE configs[1]) through windows against the oracle, the
    saturate / wrap relation and the reductions against torch."""
    length, m = 1_000_000_000, 20
    dev = torch.device("cuda", 0)
    rows = -(-length // 32)
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    seq = torch.empty((rows + m - 1, 32), dtype=torch.uint8, device=dev)
    seq[:rows] = torch.randint(0, 4, (rows, 32), dtype=torch.uint8, device=dev, generator=gen)
    torch.cuda.synchronize()
    pli.configure_wrap_dptr(seq.data_ptr(), rows, 32, 32, m - 1, 4)
    rng = np.random.default_rng(3)
    sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]
    dm = lm.create(sites).counts.normalize(0.1).log_odds().to_discrete()
    out = torch.empty((rows, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()   # the pipeline has its own stream: torch's fills must have landed
    assert pli.score_u8_dptr(dm, seq.data_ptr(), rows + m - 1, 32, 32, m - 1, length, 0, rows,
                             out.data_ptr(), 32) == (rows, length + 1 - m)
    torch.cuda.synchronize()
    assert pli.last_kernel == ("score_c32_u8_pairs" if PAIRS else "score_c32_u8")
    for a in (0, rows // 2 - 333, rows - 2048):
        win = seq[a: a + 2048 + m - 1].cpu().numpy()
        want = no.score_rows_u8_saturating(win, 32, 1 << 40, dm.data[:, :5], 0, 2048)
        assert np.array_equal(out[a: a + 2048].cpu().numpy(), want)
    best = pli.argmax_u8_dptr(out.data_ptr(), rows, 32, 32)
    mx = int(out.max())
    flat_idx = int(torch.nonzero(out.flatten() == mx)[-1])
    assert best == ((flat_idx // 32, flat_idx % 32), mx)
    t = mx - 3
    want_hits = torch.nonzero(out >= t).cpu().numpy()
    assert np.array_equal(pli.threshold_u8_dptr(out.data_ptr(), rows, 32, 32, t), want_hits)
    # row ranges are consistent with the full matrix
    part = torch.empty((5000, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    pli.score_u8_dptr(dm, seq.data_ptr(), rows + m - 1, 32, 32, m - 1, length, 777, 5777, part.data_ptr(), 32)
    torch.cuda.synchronize()
    assert torch.equal(part, out[777:5777])


def test_no_device_memory_growth_over_repeated_calls():
    """Contexts, PSSMs (incl. the library-made reverse complement), sequences, u8 scoring with
    changing matrices, fused calls: device memory returns to where it started."""
    rng = np.random.default_rng(77)
    enc = rng.integers(0, 4, 300_000, dtype=np.uint8)

    def cycle():
        pli = lm.Pipeline.hip(0)
        seq = pli.stripe(lm.EncodedSequence(enc), 32)
        seq.configure_wrap(19)
        for i in range(6):
            m = int(rng.integers(2, 20))
            sites = ["".join("ACTG"[j] for j in rng.integers(0, 4, m)) for _ in range(5)]
            pssm = lm.create(sites).counts.normalize(0.1).log_odds()
            pssm.calculate(seq).argmax()
            rc = pssm.reverse_complement()
            pli.score_argmax(rc, seq)
            pli.score_threshold(pssm, seq, 3.0)
            pli.score_discrete(pssm.to_discrete(), seq, saturate=bool(i % 2))
            list(lm.Scanner(pssm, seq, threshold=5.0))
        del seq, pli

    cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(25):
        cycle()
    import gc
    gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (64 << 20), f"device memory shrank by {(free0 - free1) >> 20} MiB"
