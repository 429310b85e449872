"""Encode / Stripe / configure_wrap on the device vs the oracle."""
import json
from pathlib import Path

import numpy as np
import pytest

import lightmotif_amd as lm
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
GOLD = json.loads((Path(__file__).parent / "golden" / "reference_vectors.json").read_text())
DNA = "ACTGN"


def test_g4_literals(pli):
    g = GOLD["G4_stripe"]
    enc = lm.EncodedSequence(g["sequence"])
    s4 = pli.stripe(enc, 4)
    assert ["".join(DNA[x] for x in row[:4]) for row in s4.matrix()] == g["c4_rows"]
    s2 = pli.stripe(enc, 2)
    assert ["".join(DNA[x] for x in row[:2]) for row in s2.matrix()] == g["c2_rows"]
    s4.configure_wrap(2)
    assert ["".join(DNA[x] for x in row[:4]) for row in s4.matrix()] == g["c4_wrap2_rows"]
    assert s4.wrap == 2 and s4.rows == 2 and len(s4) == 5


@pytest.mark.parametrize("length", [0, 1, 31, 32, 33, 64, 1000, 8191, 8192, 8193, 100_001])
@pytest.mark.parametrize("cols", [32, 16, 1])
def test_stripe_matches_oracle(pli, length, cols):
    """tests/stripe.rs:17-45 property + exact padding bytes."""
    rng = np.random.default_rng(length * 7 + cols)
    enc = rng.integers(0, 5, length, dtype=np.uint8)
    want = co.stripe(enc, cols, 5)
    got = pli.stripe(lm.EncodedSequence(enc), cols)
    assert got.rows == want.rows and len(got) == length and got.stride == want.stride
    m = got.matrix()
    assert np.array_equal(m, want.data)
    for i in range(0, length, max(1, length // 50)):
        assert m[i % got.rows, i // got.rows] == enc[i]


@pytest.mark.parametrize("wraps", [(3,), (14, 19), (5, 2, 40), (100,)])
def test_configure_wrap_matches_oracle_including_wrap_longer_than_rows(pli, wraps):
    rng = np.random.default_rng(1)
    for length in (64, 97, 3000):                        # 64 bp -> 2 rows: wrap >> rows
        enc = rng.integers(0, 5, length, dtype=np.uint8)
        want = co.stripe(enc, 32, 5)
        got = pli.stripe(lm.EncodedSequence(enc), 32)
        for w in wraps:
            co.configure_wrap(want, w)
            got.configure_wrap(w)
            assert got.wrap == want.wrap
            assert np.array_equal(got.matrix(), want.data)


def test_encode_on_device(pli):
    g = GOLD["G9_encode"]
    s = pli.stripe_ascii(g["sequence"])
    assert np.array_equal(s.matrix(), co.stripe(co.encode(g["sequence"]), 32).data)
    with pytest.raises(lm.InvalidSymbol, match=r"'\.'"):
        pli.stripe_ascii(g["unknowns"])                   # tests/encode.rs:22-26
    lossy = pli.stripe_ascii(g["unknowns"], lossy=True)
    assert np.array_equal(lossy.matrix(), co.stripe(co.encode(g["unknowns"], lossy=True), 32).data)
    rng = np.random.default_rng(2)
    text = bytes(rng.choice(list(b"ACDEFGHIKLMNPQRSTVWYX"), 10_007).tolist())
    p = pli.stripe_ascii(text, protein=True)
    assert np.array_equal(p.matrix(), co.stripe(co.encode(text, "P"), 32, 21).data)
    bad = bytearray(text)
    bad[5000] = ord("B")
    bad[7000] = ord("Z")
    with pytest.raises(lm.InvalidSymbol, match="'B'"):    # first invalid symbol wins
        pli.stripe_ascii(bytes(bad), protein=True)


def test_striped_sequence_buffer_like_test_sequence_py(pli):
    """lightmotif-py tests/test_sequence.py:60-75 (TestStripedSequence.test_memoryview): the striped matrix
    as the reference's 2-D byte buffer, shape (columns, rows), strides (1, stride): ``[column, row]``
    (lib.rs:303-317)."""
    A, C, T, G, N = range(5)
    s1 = pli.stripe(lm.EncodedSequence("ATGC"))
    mem = np.asarray(s1)
    assert (mem[0, 0], mem[1, 0], mem[2, 0], mem[3, 0]) == (A, T, G, C)
    s2 = pli.stripe(lm.EncodedSequence("ATGTCCCAACAACGATACCCCGAGCCCATCGCCGTCATCGGCTCGGCATGCAGATTCCCAGGCG"))
    mem = np.asarray(s2)
    assert mem.shape == (32, 2) and (mem[0, 0], mem[0, 1], mem[1, 0]) == (A, T, G)   # position i at [i // 2, i % 2]
    # StripedSequence.copy (lib.rs:367): independent of the original, wrap rows included
    s2.configure_wrap(3)
    c = s2.copy()
    assert (len(c), c.wrap, c.rows, c.columns) == (len(s2), 3, 2, 32) and np.array_equal(c.matrix(), s2.matrix())
    c.configure_wrap(7)
    assert s2.wrap == 3 and c.wrap == 7


@pytest.mark.parametrize("cols", [32, 16, 1, 33])
@pytest.mark.parametrize("length", [0, 1, 5, 31, 33, 1000, 16_385, 1_048_577, 3_000_001])
def test_host_ingest_goes_tile_by_tile(pli, length, cols, monkeypatch):
    """`lm_hip_seq_from_encoded` / `_from_ascii` upload the caller's buffer tile by tile (two staging tiles, one
    strided copy per tile, conversion fused into the stripe kernel): byte-exact against the oracle's stripe for
    lengths that leave partial columns, partial tiles and a partial last 16-row piece."""
    if cols == 1 and length > 1_100_000:
        pytest.skip("C = 1 rows are 32 bytes each")
    rng = np.random.default_rng(length * 131 + cols)
    enc = rng.integers(0, 5, length, dtype=np.uint8)
    want = co.stripe(enc, cols, 5)
    got = pli.stripe(lm.EncodedSequence(enc), cols)
    assert (got.rows, len(got), got.stride) == (want.rows, length, want.stride)
    assert np.array_equal(got.matrix(), want.data)
    text = np.frombuffer(b"ACTGN", np.uint8)[enc]
    got = pli.stripe_ascii(text, columns=cols)
    assert np.array_equal(got.matrix(), want.data)
    if length:
        # strict: the FIRST offending position is reported, wherever its tile / column is
        bad = text.copy()
        spots = sorted({length - 1, length // 2, (length * 7) // 8})
        for at in spots[::-1]:
            bad[at] = ord("x") if at != spots[0] else ord(".")
            with pytest.raises(lm.InvalidSymbol, match=r"'\.'" if at == spots[0] else "'x'"):
                pli.stripe_ascii(bad, columns=cols)
        lossy = pli.stripe_ascii(bad, lossy=True, columns=cols)       # seq.rs:122-129: unknown -> N
        e2 = enc.copy()
        e2[spots] = 4
        assert np.array_equal(lossy.matrix(), co.stripe(e2, cols, 5).data)
        e3 = enc.copy()
        e3[spots[-1]] = 5                                               # not a Nucleotide
        with pytest.raises(lm.InvalidSymbol):
            pli.stripe(lm.EncodedSequence(e3), cols)


def test_tiled_ingest_with_several_tiles(pli):
    """70 Mbp at C = 32: three staging tiles (the double buffer is re-used), protein text as well."""
    rng = np.random.default_rng(70)
    length = 70_000_013
    enc = rng.integers(0, 5, length, dtype=np.uint8)
    got = pli.stripe(lm.EncodedSequence(enc), 32)
    want = co.stripe(enc, 32, 5)
    assert np.array_equal(got.matrix(), want.data)
    del got
    text = np.frombuffer(b"ACTGN", np.uint8)[enc]
    text[length - 5] = ord("?")
    text[40_000_000] = ord("!")
    with pytest.raises(lm.InvalidSymbol, match="'!'"):
        pli.stripe_ascii(text)
    penc = rng.integers(0, 21, 40_000_003, dtype=np.uint8)
    ptext = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYX", np.uint8)[penc]
    got = pli.stripe_ascii(ptext, protein=True)
    assert np.array_equal(got.matrix(), co.stripe(penc, 32, 21).data)


@pytest.mark.parametrize("cols", [32, 16, 1])
@pytest.mark.parametrize("length", [0, 1, 3, 4, 5, 63, 64, 65, 1000, 16_387, 2_000_003])
@pytest.mark.parametrize("with_n", [False, True])
def test_two_bit_ingest(pli, length, cols, with_n):
    """`lm_hip_seq_from_2bit`: 4 bases per byte (+ optional N mask) -> the same striped matrix as the
    symbol-byte path, for every bit offset a column can start at (rows % 4 != 0)."""
    if cols == 1 and length > 100_000:
        pytest.skip("C = 1 rows are 32 bytes each")
    rng = np.random.default_rng(length + cols)
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    if with_n and length:
        enc[rng.random(length) < 0.03] = 4
        enc[length - 1] = 4
    packed, mask = lm.pack_2bit(enc)
    assert (mask is not None) == bool(with_n and length)
    got = pli.stripe_2bit(packed, length, mask, cols)
    want = co.stripe(enc, cols, 5)
    assert (got.rows, len(got)) == (want.rows, length)
    assert np.array_equal(got.matrix(), want.data)
    # N as runs (the .2bit container's nBlockStarts / nBlockSizes) instead of a mask
    if with_n and length > 100:
        enc[length // 3: length // 3 + 57] = 4
        enc[:9] = 4
    packed, runs = lm.pack_2bit(enc, runs=True)
    got = pli.stripe_2bit(packed, length, None, cols, n_runs=runs)
    assert np.array_equal(got.matrix(), co.stripe(enc, cols, 5).data)
    if runs is not None:
        with pytest.raises(lm.LightmotifHipError, match="leaves the sequence"):
            pli.stripe_2bit(packed, length, None, cols, n_runs=np.array([[length - 1, 2]], np.uint64))
