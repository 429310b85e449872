"""The BASELINE.json configurations that had no `-m gpu` run in round 1, at config scale:

* C1  the reference's own bench harness (lightmotif-bench/dna.rs:81-116): the MX000001-style
      two-15-mer PSSM over a 464 165 bp sequence (the first tenth of U00096; `ecoli.txt` is
      absent from the reference mount, so a seeded random stand-in of that length), striped
      at C = 32 (dispatch / avx2 geometry) AND C = 1 (the Generic bench geometry, dna.rs:
      113-116), `score_into` + `argmax` against the oracle bit for bit;
* C3  the 2 346 matrices of the reference's `lightmotif-io/benches/JASPAR2024.pwm` (committed
      as a data fixture), converted like the CLI (main.rs:469-498), over a 100 Mbp resident
      sequence through `lm_hip_scan_argmax_batch` / `lm_hip_scan_threshold_batch`;
* the multi-GPU shard entry points (`*_shard_*`, first_cell_rule = rank == 0) on ONE GPU:
  a matrix cut into 2-3 row shards, merged with the rules of lightmotif_amd.distributed,
  against the whole-matrix oracle (NaN first cell, cross-shard ties, all -inf).
"""
import os
from pathlib import Path

import numpy as np
import pytest
import torch

import lightmotif_amd as lm
from lightmotif_amd import distributed as D
from lightmotif_amd import io as lmio
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
COLS = 32
FIXTURE = Path(__file__).parent / "golden" / "JASPAR2024.pwm.gz"


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def gpu_pli():
    torch.cuda.set_device(0)
    return lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)


# ---- C1 ---------------------------------------------------------------------------------------


@pytest.mark.parametrize("cols", [32, 1], ids=["C32_dispatch_geometry", "C1_generic_bench_geometry"])
def test_c1_reference_bench_harness(pli, cols):
    """dna.rs:81-109: stripe, configure(pssm), then `score_into` + `argmax` per iteration."""
    length = 464_165                                   # U00096 is 4 641 652 bp; dna.rs:88 takes len / 10
    rng = np.random.default_rng(0xEC011)
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    # plant the two sites so the maximum is a real occurrence somewhere late in the sequence
    site = lm.EncodedSequence("GTTGACCTTATCAAC").data
    enc[391_677:391_677 + 15] = site                   # the position the reference asserts on E. coli (dna.rs:247)
    motif = lm.create(["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"])
    pssm = motif.counts.normalize(0.1).log_odds()      # to_freq(0.1).to_scoring(None) (dna.rs:95-99)
    ref = co.stripe(enc, cols, 5)
    co.configure_wrap(ref, len(pssm) - 1)
    want, want_mi = co.score_rows(ref, pssm.data)
    assert np.array_equal(co.pssm_from_sites([lm.EncodedSequence(s).data for s in
                                              ("GTTGACCTTATCAAC", "GTTGATCCAGTCAAC")])[:, :5], pssm.data[:, :5])

    seq = pli.stripe(lm.EncodedSequence(enc), cols)
    seq.configure(pssm)
    assert (seq.rows, seq.stride, seq.wrap) == (ref.rows, 32, 14)
    assert np.array_equal(seq.matrix()[:, :cols], ref.data[:, :cols])
    scores = lm.StripedScores.empty(pli, cols)
    for _ in range(3):                                 # the bench re-uses one StripedScores
        pli.score_into(pssm, seq, scores)
        best = pli.argmax(scores)
    got = scores.matrix()
    assert scores.max_index == want_mi == length - 14 and got.shape == want.shape
    assert np.array_equal(bits(got[:, :cols]), bits(want[:, :cols]))
    assert best == co.argmax(want, cols)
    assert scores.offset(*best) == 391_677             # the planted best site, at either geometry
    assert np.float32(pli.max(scores)) == want[best]
    assert pli.score_argmax(pssm, seq) == (best, float(want[best]))
    t = float(np.sort(want[:, :cols].ravel())[-40])
    assert pli.threshold(scores, t) == [tuple(map(int, rc)) for rc in co.threshold(want, cols, t)]
    assert sorted(scores.threshold(t)) == sorted(int(c) * ref.rows + int(r) for r, c in co.threshold(want, cols, t))


# ---- C3 ---------------------------------------------------------------------------------------


def test_c3_full_jaspar_batch_over_100_mbp(gpu_pli):
    pli = gpu_pli
    dev = torch.device("cuda", 0)
    records = list(lmio.read(FIXTURE))
    assert len(records) == 2346
    pssms = [r.matrix.normalize(0.1).log_odds() for r in records]      # main.rs:473-478
    lengths = np.array([len(p) for p in pssms])
    assert lengths.min() == 4 and lengths.max() == 33 and int(lengths.sum()) == 22090   # SURVEY 8 [probe]
    max_m = int(lengths.max())
    length = 100_000_000
    rows = -(-length // COLS)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xC3)
    mat = torch.empty((rows + max_m - 1, COLS), dtype=torch.uint8, device=dev)
    mat[:rows] = torch.randint(0, 4, (rows, COLS), dtype=torch.uint8, device=dev, generator=gen)
    pli.configure_wrap_dptr(mat.data_ptr(), rows, COLS, COLS, max_m - 1, 4)     # main.rs:540-546
    torch.cuda.synchronize()
    host = mat.cpu().numpy()
    seq = pli.upload(host, length, max_m - 1, COLS)
    ts = [p.score_for_pvalue(1e-5) for p in pssms]                              # main.rs:487-498

    batch_am = pli.scan_argmax_batch(pssms, seq)
    batch_th = pli.scan_threshold_batch(pssms, ts, seq)
    assert len(batch_am) == len(batch_th) == 2346
    total_hits = sum(len(c) for c, _ in batch_th)
    assert 10_000 < total_hits < 50_000_000

    # (1) every motif: batch == the fused single-motif calls
    for i, p in enumerate(pssms):
        m = len(p)
        single = pli.score_argmax_dptr(p, seq.data_ptr, rows + max_m - 1, COLS, COLS, max_m - 1, length, 0, rows)
        assert batch_am[i] == single, (i, m)
        hits, vals = pli.score_threshold_dptr(p, seq.data_ptr, rows + max_m - 1, COLS, COLS, max_m - 1,
                                              length, 0, rows, ts[i])
        assert len(hits) == len(batch_th[i][0]), (i, m)
        if i % 16 == 0:
            assert np.array_equal(hits, batch_th[i][0]) and np.array_equal(bits(vals), bits(batch_th[i][1]))

    # (2) >= 50 motifs covering every length 4..33 present in the fixture, WHOLE sequence: the oracle (AVX2 port on all host
    # threads, pinned bit-equal to the Generic restatement by tests/test_oracle_golden.py) scores all 100 Mbp; the materialised
    # matrix must equal it bit for bit, and the batch's argmax / hit lists must equal the Generic reductions of the ORACLE's
    # matrix (pli/mod.rs:135-155, 210-221) -- nothing is derived from GPU output (lightmotif/tests/argmax.rs:41-52, scan.rs:25-43)
    chosen = []
    for m in sorted(set(lengths.tolist())):
        idx = np.flatnonzero(lengths == m)
        chosen += idx[:: max(1, len(idx) // 3)][:3].tolist()
    assert len(chosen) >= 50 and {int(lengths[i]) for i in chosen} == set(lengths.tolist())
    scores = torch.empty((rows, COLS), dtype=torch.float32, device=dev)
    ahost = co.aligned_empty(host.shape, np.uint8)
    ahost[:] = host
    want = co.aligned_empty((rows, COLS), np.float32)
    threads = os.cpu_count() or 1
    for i in chosen:
        p, m = pssms[i], int(lengths[i])
        pli.score_dptr(p, seq.data_ptr, rows + max_m - 1, COLS, COLS, max_m - 1, length, 0, rows,
                       scores.data_ptr(), COLS)
        torch.cuda.synchronize()
        ref = co.Striped(ahost, length, max_m - 1, COLS, 5)
        w = co.aligned_empty(p.data.shape, np.float32)
        w[:] = p.data
        co.avx2_score_rows(ref, w, out=want, threads=threads)
        assert np.array_equal(bits(scores.cpu().numpy()), bits(want)), (i, m)
        flat = want.reshape(-1)
        vmax = flat.max()
        last = flat.size - 1 - int(np.argmax(flat[::-1] == vmax))           # the LAST maximal cell in row-major order
        assert batch_am[i] == ((last // COLS, last % COLS), float(vmax)), (i, m)
        nz = np.flatnonzero(flat >= np.float32(ts[i]))
        assert np.array_equal(np.asarray(batch_th[i][0], np.int64).reshape(-1, 2), np.stack([nz // COLS, nz % COLS], axis=1)), (i, m)
        assert np.array_equal(bits(batch_th[i][1]), bits(flat[nz])), (i, m)


# ---- the shard entry points + merge rules on one GPU -----------------------------------------


def _planted(kind, rng, length, m):
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    p = np.zeros((m, 8), np.float32)
    p[:, :4] = rng.integers(-2, 3, (m, 4)) if kind in ("ties", "nan_first") else rng.normal(0, 2, (m, 4))
    p[:, 4] = -np.inf
    if kind == "nan_first":
        p[0, int(enc[0])] = np.nan                     # scores[0][0] (and many other cells) are NaN
    if kind == "all_neg_inf":
        p[:, :4] = -np.inf
    return enc, p


@pytest.mark.parametrize("shards", [2, 3])
@pytest.mark.parametrize("kind", ["random", "ties", "nan_first", "all_neg_inf", "nan_elsewhere"])
def test_shard_entry_points_merge_to_the_whole_matrix_answer(gpu_pli, kind, shards):
    """What rank g of a row-sharded job calls: lm_hip_argmax_shard_f32_dptr and
    lm_hip_score_argmax_shard_f32_dptr with first_cell_rule = (g == 0) on rows [a_g, b_g) +
    halo, then the merge (distributed.combine_*).  Must equal the Generic argmax / threshold of
    the whole matrix (pli/mod.rs:135-155, 210-221)."""
    pli = gpu_pli
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng([len(kind), ord(kind[0]), ord(kind[-1]), shards])
    length, m = 3_000_017, 9
    enc, p = _planted("random" if kind == "nan_elsewhere" else kind, rng, length, m)
    rows = -(-length // COLS)
    spans = [D.shard_rows(rows, shards, g) for g in range(shards)]
    if kind == "nan_elsewhere":
        # NaN cells only where the window starts with the first symbol of shard 1 (position a1 = cell
        # (a1, 0)): that shard's first cell is NaN, which must NOT trigger the first-cell rule there;
        # the matrix's own first cell starts with another symbol
        a1 = spans[1][0]
        enc[0] = (enc[a1] + 1) % 4
        p[0, int(enc[a1])] = np.nan
    ref = co.stripe(enc, COLS, 5)
    co.configure_wrap(ref, m - 1)
    assert ref.rows == rows
    pssm = lm.ScoringMatrix(p)
    want, _ = co.score_rows(ref, p)
    want_am = co.argmax(want, COLS)
    finite = want[:, :COLS][np.isfinite(want[:, :COLS])]
    t = float(np.quantile(finite, 0.9995)) if finite.size else 0.0
    want_thr = np.asarray(co.threshold(want, COLS, t), np.int64).reshape(-1, 2)

    mats, fused, thr = [], [], []
    for g, (a, b) in enumerate(spans):
        shard = torch.from_numpy(ref.data[a:b + m - 1].copy()).to(dev)          # rows + halo
        out = torch.empty((b - a, COLS), dtype=torch.float32, device=dev)
        pli.score_dptr(pssm, shard.data_ptr(), b - a + m - 1, COLS, COLS, m - 1, length, 0, b - a,
                       out.data_ptr(), COLS)
        torch.cuda.synchronize()
        assert np.array_equal(bits(out.cpu().numpy()), bits(want[a:b, :COLS]))
        loc = pli.argmax_dptr(out.data_ptr(), b - a, COLS, COLS, first_cell_rule=g == 0)
        fus = pli.score_argmax_dptr(pssm, shard.data_ptr(), b - a + m - 1, COLS, COLS, m - 1, length,
                                    0, b - a, first_cell_rule=g == 0)
        for name, rec in (("materialised", loc), ("fused", fus)):
            if rec is not None and g > 0:
                assert rec[1] == rec[1], f"{name}: a NaN left a shard that does not hold row 0"
        mats.append(None if loc is None else ((loc[0][0] + a, loc[0][1]), loc[1]))
        fused.append(None if fus is None else ((fus[0][0] + a, fus[0][1]), fus[1]))
        thr.append(pli.threshold_dptr(out.data_ptr(), b - a, COLS, COLS, t))
        f_hits, _ = pli.score_threshold_dptr(pssm, shard.data_ptr(), b - a + m - 1, COLS, COLS, m - 1,
                                             length, 0, b - a, t)
        assert np.array_equal(f_hits, thr[-1])
    for name, recs in (("materialised", mats), ("fused", fused)):
        got = D.combine_argmax(recs)
        assert got is not None and got[0] == want_am, (name, got, want_am)
        if want[want_am] == want[want_am]:
            assert np.float32(got[1]) == want[want_am]
        else:
            assert got[1] != got[1]
    got_thr = D.combine_threshold(thr, [a for a, _ in spans])
    assert np.array_equal(got_thr, want_thr)
    # with the rule applied on every shard (the single-GPU entry point misused) the NaN cases differ
    if kind == "nan_elsewhere":
        a, b = spans[1]
        shard = torch.from_numpy(ref.data[a:b + m - 1].copy()).to(dev)
        wrong = pli.score_argmax_dptr(pssm, shard.data_ptr(), b - a + m - 1, COLS, COLS, m - 1, length, 0, b - a,
                                      first_cell_rule=True)
        assert wrong[0] == (0, 0) and wrong[1] != wrong[1]


# ---- ADVICE round 1 ---------------------------------------------------------------------------


@pytest.mark.parametrize("wrap", [40, 63, 200])
def test_upload_adopts_any_wrap(pli, wrap):
    """A StripedSequence already configured for a long motif (configure_wrap(max_m) of the CLI,
    M = 41 / 64 shapes) must be adoptable: round 1 refused wrap > 32."""
    rng = np.random.default_rng(wrap)
    enc = rng.integers(0, 4, 5_003, dtype=np.uint8)
    ref = co.stripe(enc, COLS, 5, extra_rows=wrap + 8)
    co.configure_wrap(ref, wrap)
    seq = pli.upload(ref.data, len(enc), wrap, COLS)
    assert (seq.rows, seq.wrap) == (ref.rows, wrap)
    assert np.array_equal(seq.matrix(), ref.data)
    m = min(wrap + 1, 64)
    p = np.zeros((m, 8), np.float32)
    p[:, :4] = rng.normal(0, 2, (m, 4))
    p[:, 4] = -np.inf
    want, _ = co.score_rows(ref, p)
    got = pli.score(lm.ScoringMatrix(p), seq).matrix()
    assert np.array_equal(bits(got[:, :COLS]), bits(want[:, :COLS]))


def test_symbol_bytes_are_validated_at_the_handle_entry_points(pli):
    enc = np.zeros(1000, np.uint8)
    enc[777] = 5                                                     # not a Nucleotide
    with pytest.raises(lm.InvalidSymbol):
        pli.stripe(lm.EncodedSequence(enc), COLS)
    enc[777] = 4
    ref = co.stripe(enc, COLS, 5)
    data = ref.data.copy()
    data[3, 7] = 9
    with pytest.raises(lm.InvalidSymbol):
        pli.upload(data, len(enc), 0, COLS)
    enc21 = np.full(100, 20, np.uint8)                               # X is a valid AminoAcid
    assert len(pli.stripe(lm.EncodedSequence(enc21, protein=True), COLS)) == 100
    # padding bytes past `cols` are not symbols: garbage there (a Rust Row's struct padding) is fine
    ref16 = co.stripe(enc, 16, 5)
    d16 = ref16.data.copy()
    d16[:, 16:] = 0xAB
    seq = pli.upload(d16, len(enc), 0, 16)
    assert seq.columns == 16


def test_padding_past_cols_is_the_default_symbol(pli):
    """dense.rs:144-147 fills fresh rows with T::default() = N / X; stride > cols layouts keep
    every byte a valid symbol."""
    enc = np.arange(100, dtype=np.uint8) % 4
    for cols in (1, 16, 33):
        seq = pli.stripe(lm.EncodedSequence(enc), cols)
        seq.configure_wrap(3)
        ref = co.stripe(enc, cols, 5)
        co.configure_wrap(ref, 3)
        got = seq.matrix()
        assert np.array_equal(got, ref.data)
        assert (got[:, cols:] == 4).all()


def test_device_ordinals_lists_usable_devices():
    ords = lm.Pipeline.device_ordinals()
    assert len(ords) == lm.Pipeline.device_count() >= 1
    assert lm.Pipeline.hip(ords[0]) is not None


def test_c_abi_communicator_single_rank(gpu_pli):
    """lm_hip_comm_* with nranks = 1 on the box's one GPU: librccl is opened by the library
    itself, the communicator initialises, and halo / argmax / max / threshold merges run through
    real RCCL collectives (all_gather, broadcast) -- the degenerate world every rank-count
    shares.  (Two ranks cannot share one GPU under RCCL; world sizes 2 and 3 are covered by the
    gloo tests of the merge rules and by the driver's multi-GPU bench.)"""
    pli = gpu_pli
    dev = torch.device("cuda", 0)
    comm = D.CabiComm.from_torch(pli)
    assert (comm.rank, comm.nranks) == (0, 1)
    rng = np.random.default_rng(5)
    length, m = 2_000_003, 11
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    p = np.zeros((m, 8), np.float32)
    p[:, :4] = rng.integers(-2, 3, (m, 4))
    p[:, 4] = -np.inf
    ref = co.stripe(enc, COLS, 5)
    rows = ref.rows
    shard = torch.empty((rows + m - 1, COLS), dtype=torch.uint8, device=dev)
    shard[:rows] = torch.from_numpy(ref.data[:rows].copy()).to(dev)
    shard[rows:] = 77
    comm.exchange_halo(shard, m - 1, COLS, 4)           # world of one: the wrap rows of seq.rs:373-378
    co.configure_wrap(ref, m - 1)
    assert np.array_equal(shard.cpu().numpy(), ref.data)
    want, _ = co.score_rows(ref, p)
    pssm = lm.ScoringMatrix(p)
    seq = pli.adopt_sequence(shard.data_ptr(), rows, m - 1, COLS, COLS, length, keepalive=shard)
    scores = lm.StripedScores.empty(pli, COLS)
    pli.score_into(pssm, seq, scores)
    want_am = co.argmax(want, COLS)
    got = comm.argmax_sharded(scores, 0)
    assert got == (want_am, float(want[want_am]))
    assert comm.merge_argmax(got, 0) == got and comm.merge_argmax(None, 0) is None
    assert comm.merge_argmax(((5, 6), 1.25), 1000) == ((1005, 6), 1.25)
    assert comm.merge_max(2.5) == 2.5 and comm.merge_max(None) is None
    t = float(np.sort(want[:, :COLS].ravel())[-5000])
    hits = pli.threshold_dptr(scores.data_ptr, rows, COLS, COLS, t)
    merged = comm.merge_threshold(hits, 123)
    assert np.array_equal(merged[:, 0], hits[:, 0] + 123) and np.array_equal(merged[:, 1], hits[:, 1])
    assert comm.merge_threshold(np.zeros((0, 2), np.int64), 0).shape == (0, 2)
    # the shard flag: a NaN first cell is reported only when the rule is on
    p2 = p.copy()
    p2[0, int(enc[0])] = np.nan
    pssm2 = lm.ScoringMatrix(p2)
    for rule in (True, False):
        scores.set_first_cell_rule(rule)
        pli.score_into(pssm2, seq, scores)
        rec = pli.argmax(scores), pli.max(scores)
        mat = pli.argmax_dptr(scores.data_ptr, rows, COLS, COLS, first_cell_rule=rule)
        assert rec[0] == mat[0]
        assert (rec[0] == (0, 0) and rec[1] != rec[1]) if rule else (rec[1] == rec[1])
    comm.close()


def test_pipelined_sharded_argmax_merge(gpu_pli):
    """lm_hip_argmax_sharded_begin / _end: the shard is scored over while the previous merge is in
    flight (its record, all_gather and read-back live in the communicator's own slots and stream);
    every ticket must return what the synchronous call returns for ITS step's matrix -- tracked
    records and materialised ones, first-cell NaN, two in flight at most."""
    pli = gpu_pli
    comm = D.CabiComm.from_torch(pli)
    rng = np.random.default_rng(31)
    length = 3_000_017
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    ref = co.stripe(enc, COLS, 5)
    co.configure_wrap(ref, 24)
    seq = pli.stripe(lm.EncodedSequence(enc), COLS)
    seq.configure_wrap(24)
    scores = lm.StripedScores.empty(pli, COLS)
    motifs = []
    for i, m in enumerate((20, 7, 25, 12, 20, 16)):
        p = np.zeros((m, 8), np.float32)
        p[:, :4] = rng.integers(-2, 3, (m, 4)) if i % 2 else rng.normal(0, 2, (m, 4))
        p[:, 4] = -np.inf
        if i == 3:
            p[0, int(enc[0])] = np.nan                   # first-cell rule through the pipelined form
        want, _ = co.score_rows(ref, p)
        am = co.argmax(want, COLS)
        motifs.append((lm.ScoringMatrix(p), (am, float(want[am]))))
    for track in (True, False):
        pli.set_track_argmax(track)
        pending, got = None, []
        for pssm, _ in motifs:
            pli.score_into(pssm, seq, scores)
            ticket = comm.argmax_sharded_begin(scores, 0)
            if pending is not None:
                got.append(comm.argmax_sharded_end(pending))
            pending = ticket
        got.append(comm.argmax_sharded_end(pending))
        for g, (_, w) in zip(got, motifs):
            assert g[0] == w[0] and (g[1] == w[1] or (g[1] != g[1] and w[1] != w[1])), (track, g, w)
    pli.set_track_argmax(True)
    # misuse: a third merge while two are in flight; a ticket that is not in flight
    t0 = comm.argmax_sharded_begin(scores, 0)
    t1 = comm.argmax_sharded_begin(scores, 0)
    with pytest.raises(lm.LightmotifHipError):
        comm.argmax_sharded_begin(scores, 0)
    a, b = comm.argmax_sharded_end(t0), comm.argmax_sharded_end(t1)
    assert a == b == comm.argmax_sharded(scores, 0)
    with pytest.raises(lm.LightmotifHipError):
        comm.argmax_sharded_end(t0)
    # a shard that does not start at row 0 never applies the first-cell rule; rows come back global
    t = comm.argmax_sharded_begin(scores, 1000)
    r = comm.argmax_sharded_end(t)
    assert r[0][0] >= 1000
    comm.close()


def test_adopted_sequence_borrows_the_callers_matrix(gpu_pli):
    pli = gpu_pli
    dev = torch.device("cuda", 0)
    enc = np.arange(5000, dtype=np.uint8) % 4
    ref = co.stripe(enc, COLS, 5)
    co.configure_wrap(ref, 9)
    t = torch.zeros((ref.rows + 12, COLS), dtype=torch.uint8, device=dev)
    t[:ref.rows + 9] = torch.from_numpy(ref.data.copy()).to(dev)
    seq = pli.adopt_sequence(t.data_ptr(), ref.rows, 9, COLS, COLS, len(enc), keepalive=t, capacity_rows=t.shape[0])
    assert seq.data_ptr == t.data_ptr() and (seq.rows, seq.wrap) == (ref.rows, 9)
    seq.configure_wrap(12)                               # fits the rows handed over
    co.configure_wrap(ref, 12)
    assert np.array_equal(t.cpu().numpy(), ref.data)
    with pytest.raises(lm.LightmotifHipError):
        seq.configure_wrap(13)                           # would need a reallocation of memory it does not own
    del seq
    torch.cuda.synchronize()
    assert int(t[0, 0]) == int(ref.data[0, 0])           # still the caller's, not freed


# ---- beyond the unrolled C = 32 kernels: long motifs and other column counts ------------------


def long_kernel(m, mode):
    """36 < M <= 64: ONE pass of the long kernel family at the padded length M' = 4 * ceil(M / 4)
    (score_long_inst.hip); 65 ... 88: the plain STORE alone still in one pass at M' = 8 * ceil(M / 8)
    (score_xlong_inst.hip); beyond: slices of <= 64 rows (store + in-place continuation)."""
    if m <= 64:
        return f"score_c32<{-(-m // 4) * 4},{mode}>"
    if m <= 88 and mode == 0:
        return f"score_c32<{-(-m // 8) * 8},0>"
    return "score_c32_sliced" if mode == 0 else "score_c32_sliced+reduce"


@pytest.mark.parametrize("m", [37, 38, 40, 41, 44, 47, 48, 52, 55, 56, 60, 63, 64, 65, 71, 72, 73, 80, 81, 88, 95, 96, 97, 100, 104, 105,
                               128, 129, 150])
def test_long_motifs_are_scored_in_slices(pli, m):
    """M > 36 at C = 32: up to 64 rows in ONE pass (score_c32<M', 0> of the long family, leading zero
    rows up to a multiple of 4), up to 88 rows the plain store still in one pass (leading zero rows up to a multiple
    of 8: up to seven of them, two symbol blocks that may lie before the matrix); longer motifs in slices of <= 64 motif rows, the first through the
    store kernel, the others continuing in place from the partial sums (score_c32<M', MODE_CONTINUE>)
    -- the same sequential adds, so bit-exact against the oracle on ragged lengths, row ranges,
    -inf / NaN weights, N runs.  The reference's AVX2 loop takes any M (avx2.rs:146-193)."""
    rng = np.random.default_rng(m)
    length = 3_000_000 + 17 * m
    enc = rng.integers(0, 5, length, dtype=np.uint8)
    enc[rng.random(length) < 0.97] %= 4
    p = np.zeros((m, 8), np.float32)
    p[:, :4] = rng.normal(0, 2, (m, 4))
    p[:, 4] = -np.inf
    if m % 2:
        p[rng.integers(0, m), 4] = 1.5
        p[rng.integers(0, m), rng.integers(0, 4)] = np.nan if m == 47 else -np.inf
    ref = co.stripe(enc, COLS, 5)
    co.configure_wrap(ref, m - 1)
    seq = pli.stripe(lm.EncodedSequence(enc), COLS)
    seq.configure_wrap(m - 1)
    pssm = lm.ScoringMatrix(p)
    scores = lm.StripedScores.empty(pli, COLS)
    for a, b in ((0, ref.rows), (1234, ref.rows - 5), (ref.rows - 3 * m, ref.rows)):
        want, _ = co.score_rows(ref, p, a, b)
        pli.score_rows_into(pssm, seq, range(a, b), scores)
        assert pli.last_kernel == long_kernel(m, 0), pli.last_kernel
        assert np.array_equal(bits(scores.matrix()[:, :COLS]), bits(want[:, :COLS])), (m, a, b)
    want, _ = co.score_rows(ref, p)
    pli.score_into(pssm, seq, scores)
    assert pli.argmax(scores) == co.argmax(want, COLS)
    assert pli.score_argmax(pssm, seq)[0] == co.argmax(want, COLS)
    # a range shorter than one group falls back, same values
    want, _ = co.score_rows(ref, p, 10, 10 + m // 2)
    pli.score_rows_into(pssm, seq, range(10, 10 + m // 2), scores)
    assert np.array_equal(bits(scores.matrix()[:, :COLS]), bits(want[:, :COLS]))


@pytest.mark.parametrize("m,chunk_rows", [(37, 5000), (40, 4096), (45, 4096), (53, 4096), (59, 4096), (64, 1 << 20),
                                          (65, 5000), (73, 7777), (100, 20_000), (127, 4096), (128, 4096), (129, 4096),
                                          (150, 128)])
def test_long_motifs_fused_reductions_go_chunk_by_chunk(m, chunk_rows):
    """score_argmax / score_threshold / Scanner-style hits of M > 36.  Up to 64 rows: the fused kernels of
    the long family (score_c32<M', 1 | 2>), no score matrix.  Beyond: the sliced store path into a
    reusable chunk buffer + a reduction per chunk (score_launch.hpp, KIND_CHUNKED) instead of one thread per
    cell.  Same cells, same values: bit-exact against the oracle's materialised matrix, row-major hit
    order, last-maximal-cell ties across chunk borders, first-cell NaN rule, row sub-ranges."""
    pli = lm.Pipeline.hip(0)
    pli.set_option("chunk_rows", chunk_rows)
    rng = np.random.default_rng(7000 + m)
    length = 1_500_000 + 13 * m
    enc = rng.integers(0, 5, length, dtype=np.uint8)
    enc[rng.random(length) < 0.98] %= 4
    p = np.zeros((m, 8), np.float32)
    if m in (40, 150):      # few distinct values -> many tied maxima, in several chunks
        p[:, :4] = rng.integers(0, 2, (m, 4))
    else:
        p[:, :4] = rng.normal(0, 2, (m, 4))
    p[:, 4] = -np.inf if m != 73 else -3.0
    ref = co.stripe(enc, COLS, 5)
    co.configure_wrap(ref, m - 1)
    seq = pli.stripe(lm.EncodedSequence(enc), COLS)
    seq.configure_wrap(m - 1)
    pssm = lm.ScoringMatrix(p)
    for a, b in ((0, ref.rows), (123, ref.rows - 77)):
        want, _ = co.score_rows(ref, p, a, b)
        got = pli.score_argmax(pssm, seq, range(a, b))
        assert pli.last_kernel == long_kernel(m, 1), pli.last_kernel
        assert got[0] == co.argmax(want, COLS)
        assert bits(np.float32(got[1])) == bits(co.max_(want, COLS))
        finite = np.sort(want[:, :COLS][np.isfinite(want[:, :COLS])])
        for t in (float(finite[-50]), float(finite[-5000])):
            wrc = [tuple(map(int, rc)) for rc in co.threshold(want, COLS, t)]
            frc, fval = pli.score_threshold(pssm, seq, t, range(a, b))
            # DNA up to 128 rows: the pair-symbol prefilter scan flags the candidates (exact re-scoring decides)
            assert pli.last_kernel == ("score_c32_prefilter2" if m <= 128 else long_kernel(m, 2)), pli.last_kernel
            pli.set_prefilter(False)                 # ... and the exact fused kernel of the long family gives the same list
            frc2, fval2 = pli.score_threshold(pssm, seq, t, range(a, b))
            pli.set_prefilter(True)
            assert pli.last_kernel == long_kernel(m, 2) and frc2 == wrc and np.array_equal(bits(fval2), bits(fval))
            assert frc == wrc
            assert np.array_equal(bits(fval), bits([want[r, c] for r, c in wrc]))
    # first-cell NaN rule (pli/mod.rs:142-146) through the chunked path
    q = p.copy()
    q[0, int(enc[0])] = np.nan
    want, _ = co.score_rows(ref, q)
    assert np.isnan(want[0, 0])
    got = pli.score_argmax(lm.ScoringMatrix(q), seq)
    assert got[0] == co.argmax(want, COLS) == (0, 0) and np.isnan(got[1])


@pytest.mark.parametrize("m,k", [(4, 5), (7, 5), (8, 5), (15, 5), (20, 5), (21, 5), (28, 5), (32, 5), (36, 5), (12, 21),
                                 (5, 21)])
def test_sixteen_columns_take_the_unrolled_store_kernel(pli, m, k):
    """C = 16 is the column count of the reference's 16-lane back-ends (sse2.rs / neon.rs: U16): the
    rotating-accumulator store kernel with four 16-column streams per wavefront (score_c32<M', 0, ..., 16>,
    M padded to a multiple of 4 with leading zero rows like at C = 32), bit-exact against the oracle on
    ragged lengths and row ranges; the sequence rows keep their 32-byte stride (dense.rs:43-48)."""
    rng = np.random.default_rng(1600 + m)
    length = 2_000_003
    enc = rng.integers(0, k, length, dtype=np.uint8)
    enc[rng.random(length) < 0.95] %= (k - 1)
    p = np.zeros((m, co.stride(k, 4)), np.float32)
    p[:, :k] = rng.normal(0, 2, (m, k))
    p[:, k - 1] = -np.inf
    ref = co.stripe(enc, 16, k)
    co.configure_wrap(ref, m - 1)
    seq = pli.stripe(lm.EncodedSequence(enc, protein=k == 21), 16)
    seq.configure_wrap(m - 1)
    pssm = lm.ScoringMatrix(p, protein=k == 21)
    scores = lm.StripedScores.empty(pli, 16)
    for a, b in ((0, ref.rows), (7, ref.rows - 11), (ref.rows - 3 * m - 2, ref.rows), (1000, 1000 + m + 1)):
        want, _ = co.score_rows(ref, p, a, b)
        pli.score_rows_into(pssm, seq, range(a, b), scores)
        if b - a > m + 4:                       # (a range shorter than one padded group falls back, same values)
            assert pli.last_kernel.startswith("score_c32<"), pli.last_kernel
        assert np.array_equal(bits(scores.matrix()[:, :16]), bits(want[:, :16])), (m, a, b)
    want, _ = co.score_rows(ref, p)
    pli.score_into(pssm, seq, scores)
    assert pli.argmax(scores) == co.argmax(want, 16)
    # the fused forms: the same store kernel into a chunk buffer + a reduction per chunk
    got = pli.score_argmax(pssm, seq)
    assert pli.last_kernel == "score_store+reduce", pli.last_kernel
    assert got[0] == co.argmax(want, 16) and bits(np.float32(got[1])) == bits(co.max_(want, 16))
    finite = np.sort(want[:, :16][np.isfinite(want[:, :16])])
    for t in (float(finite[-30]), float(finite[-3000])):
        wrc = [tuple(map(int, rc)) for rc in co.threshold(want, 16, t)]
        frc, fval = pli.score_threshold(pssm, seq, t)
        assert pli.last_kernel == "score_store+reduce", pli.last_kernel
        assert frc == wrc and np.array_equal(bits(fval), bits([want[r, c] for r, c in wrc]))


@pytest.mark.parametrize("cols,m,k", [(16, 33, 5), (1, 15, 5), (2, 7, 5), (33, 12, 5), (16, 70, 5),
                                      (4, 100, 5), (8, 40, 21)])
def test_other_geometries_take_the_tiled_kernel(pli, cols, m, k):
    """Column counts other than 32 (the 16-lane back-ends' geometry, the Generic bench's C = 1,
    dna.rs:113-116) and motifs beyond 64: score_tiled, bit-exact against the oracle."""
    rng = np.random.default_rng(cols * 1000 + m)
    length = 400_003 if cols > 1 else 100_003
    enc = rng.integers(0, k - 1, length, dtype=np.uint8)
    p = np.zeros((m, co.stride(k, 4)), np.float32)
    p[:, :k] = rng.normal(0, 2, (m, k))
    p[:, k - 1] = -np.inf
    ref = co.stripe(enc, cols, k)
    co.configure_wrap(ref, m - 1)
    seq = pli.stripe(lm.EncodedSequence(enc, protein=k == 21), cols)
    seq.configure_wrap(m - 1)
    pssm = lm.ScoringMatrix(p, protein=k == 21)
    scores = lm.StripedScores.empty(pli, cols)
    for a, b in ((0, ref.rows), (7, ref.rows - 11), (ref.rows - 3, ref.rows)):
        want, _ = co.score_rows(ref, p, a, b)
        pli.score_rows_into(pssm, seq, range(a, b), scores)
        assert pli.last_kernel == "score_tiled", pli.last_kernel
        assert np.array_equal(bits(scores.matrix()[:, :cols]), bits(want[:, :cols])), (cols, m, a, b)
    assert pli.argmax(scores) == co.argmax(want, cols)
    # fused forms: store into dense rows of a chunk buffer + reduction per chunk
    a, b = ref.rows - 3, ref.rows
    big = pli.score_argmax(pssm, seq)
    # (short motifs may be settled from the last rows -- a small sub-range, scanned cell by cell -- before that)
    # (one launch when a single chunk holds the matrix: the tiled kernel's workgroup records, folded by the host)
    assert pli.last_kernel in ("score_store+reduce", "score_tiled+host_fold") or (m <= 9 and pli.last_kernel.startswith("score_generic<")), \
        pli.last_kernel
    full, _ = co.score_rows(ref, p)
    assert big[0] == co.argmax(full, cols)
    finite = np.sort(full[:, :cols][np.isfinite(full[:, :cols])])
    t = float(finite[-200])
    wrc = [tuple(map(int, rc)) for rc in co.threshold(full, cols, t)]
    frc, fval = pli.score_threshold(pssm, seq, t)
    assert frc == wrc and np.array_equal(bits(fval), bits([full[r, c] for r, c in wrc]))


def test_symbol_validation_covers_sequences_of_2_32_symbols_and_more(gpu_pli):
    """An encoded sequence of >= 2^32 symbols is validated to its last byte (the column count
    used to be narrowed to 32 bits, so only len mod 2^32 bytes were looked at)."""
    length = 2 ** 32 + 4096 + 3
    enc = np.zeros(length, np.uint8)
    enc[1::7] = 3
    seq = gpu_pli.stripe(lm.EncodedSequence(enc), COLS)              # all valid
    assert len(seq) == length
    del seq
    for bad in (2 ** 32 + 1000, length - 1, 2 ** 32 - 1):
        enc[bad] = 5                                                 # not a Nucleotide
        with pytest.raises(lm.InvalidSymbol):
            gpu_pli.stripe(lm.EncodedSequence(enc), COLS)
        enc[bad] = 0


@pytest.mark.parametrize("m", [37, 44, 52, 64, 70, 88, 90])
def test_long_protein_motifs_take_the_wide_long_kernels(pli, m):
    """Protein (K = 21) motifs of 37..64 rows: the long family's WIDE instantiations (8-byte LDS reads over rows of
    2 * odd dwords), store and fused; 65..88 the one-pass store, beyond that the sliced path.  Bit-exact against the oracle incl. X runs."""
    rng = np.random.default_rng(2100 + m)
    length = 400_000 + 7 * m
    enc = rng.integers(0, 21, length, dtype=np.uint8)
    enc[rng.random(length) < 0.98] %= 20
    p = np.zeros((m, 24), np.float32)
    p[:, :21] = rng.normal(0, 2, (m, 21))
    p[:, 20] = -np.inf
    ref = co.stripe(enc, COLS, 21)
    co.configure_wrap(ref, m - 1)
    seq = pli.stripe(lm.EncodedSequence(enc, protein=True), COLS)
    seq.configure_wrap(m - 1)
    pssm = lm.ScoringMatrix(p, protein=True)
    scores = lm.StripedScores.empty(pli, COLS)
    for a, b in ((0, ref.rows), (17, ref.rows - 9)):
        want, _ = co.score_rows(ref, p, a, b)
        pli.score_rows_into(pssm, seq, range(a, b), scores)
        assert pli.last_kernel == long_kernel(m, 0), pli.last_kernel
        assert np.array_equal(bits(scores.matrix()[:, :COLS]), bits(want[:, :COLS])), (m, a, b)
        assert pli.argmax(scores) == co.argmax(want, COLS)
        got = pli.score_argmax(pssm, seq, range(a, b))
        assert got[0] == co.argmax(want, COLS) and pli.last_kernel == long_kernel(m, 1)
        finite = np.sort(want[:, :COLS][np.isfinite(want[:, :COLS])])
        t = float(finite[-300])
        frc, fval = pli.score_threshold(pssm, seq, t, range(a, b))
        assert frc == [tuple(map(int, rc)) for rc in co.threshold(want, COLS, t)]
        assert np.array_equal(bits(fval), bits([want[r, c] for r, c in frc]))
