"""BASELINE.json's full size (1 Gbp x len-20) through size-independent properties:
window samples against the oracle, row-range consistency, exact scaling by powers of
two, argmax / threshold against an independent torch formulation, fused == materialised.
Also the protein configuration (K = 21, M = 12, 200 Mres)."""
import os

import numpy as np
import pytest
import torch

import lightmotif_amd as lm
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
COLS = 32


def make_workload(pli, length, m, k, seed):
    dev = torch.device("cuda", 0)
    rows = -(-length // COLS)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    if k == 5:
        # DNA: the SplitMix64 stream SURVEY 8(d) prescribes (2 bits per base, position i at [i % R][i / R], N past the end) --
        # the generator of bench.py, seeded per case
        import bench
        seq = bench.synth_shard(rows, 0, rows, length, m - 1, dev, seed=0x5EED0001 + seed)
    else:
        seq = torch.empty((rows + m - 1, COLS), dtype=torch.uint8, device=dev)
        seq[:rows] = torch.randint(0, k - 1, (rows, COLS), dtype=torch.uint8, device=dev, generator=gen)
        if length < rows * COLS:  # padded tail = default symbol (pli/mod.rs:194-196)
            idx = torch.arange(length, rows * COLS, device=dev)
            seq[idx % rows, idx // rows] = k - 1
    pli.configure_wrap_dptr(seq.data_ptr(), rows, COLS, COLS, m - 1, k - 1)
    rng = np.random.default_rng(seed)
    sym = lm.lib.PROTEIN_SYMBOLS if k == 21 else lm.lib.DNA_SYMBOLS
    sites = ["".join(sym[i] for i in rng.integers(0, k - 1, m)) for _ in range(10)]
    pssm = lm.create(sites, protein=k == 21).counts.normalize(0.1).log_odds()
    return seq, rows, pssm


def score_all(pli, pssm, seq, rows, m, length, out=None, a=0, b=None):
    b = rows if b is None else b
    if out is None:
        out = torch.empty((b - a, COLS), dtype=torch.float32, device=seq.device)
    got = pli.score_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, a, b,
                         out.data_ptr(), COLS)
    assert got == (b - a, length + 1 - m)
    torch.cuda.synchronize()
    return out


@pytest.fixture(scope="module")
def gpu_pli():
    torch.cuda.set_device(0)
    return lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("length,m,k", [(1_000_000_000, 20, 5), (999_999_937, 15, 5), (200_000_000, 12, 21),
                                        (200_000_000, 50, 5), (150_000_000, 100, 5), (150_000_000, 75, 5)],
                         ids=["dna_1Gbp_m20", "dna_ragged_m15", "protein_200M_m12", "dna_200M_m50_long_family",
                              "dna_150M_m100_sliced_store_pair_scan", "dna_150M_m75_one_pass_store_pair_scan"])
def test_full_size_properties(gpu_pli, length, m, k):
    pli = gpu_pli
    seq, rows, pssm = make_workload(pli, length, m, k, seed=1234 + m)
    scores = score_all(pli, pssm, seq, rows, m, length)
    if m > 88:
        assert pli.last_kernel == "score_c32_sliced"
    elif m > 64:
        assert pli.last_kernel == f"score_c32<{(m + 7) // 8 * 8},0>"   # one pass, padded to 8 | M (score_xlong_inst.hip)
    else:
        # padded with leading zero rows to 4 | M (33 ... 35 run unpadded; 37 ... 64: the long family)
        mp = m if m % 4 == 0 or 32 < (m + 3) // 4 * 4 <= 36 else (m + 3) // 4 * 4
        assert pli.last_kernel == f"score_c32<{mp},0>"

    # (1) windows against the oracle, bit for bit (start, middle, the wrap-touching end)
    for a in (0, rows // 2 - 777, rows - 4096):
        b = min(a + 4096, rows)
        host = seq[a:b + m - 1].cpu().numpy()
        win = co.Striped(host, length, m - 1, COLS, k)
        want, _ = co.score_rows(win, pssm.data, 0, b - a)
        got = scores[a:b].cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"window at row {a}"
    host_wrap = seq[rows:].cpu().numpy()
    host_head = seq[:m - 1].cpu().numpy()
    assert np.array_equal(host_wrap[:, :COLS - 1], host_head[:, 1:COLS]) and (host_wrap[:, COLS - 1] == k - 1).all()

    # (2) score_rows_into over sub-ranges == the same rows of the full result (pli/mod.rs:72-78)
    for a, b in ((0, 1), (5, 6 + m), (rows // 3, rows // 3 + 100_003), (rows - 2 * m - 1, rows)):
        part = score_all(pli, pssm, seq, rows, m, length, a=a, b=b)
        assert torch.equal(part.view(torch.int32), scores[a:b].view(torch.int32)), (a, b)

    # (3) scaling the PSSM by a power of two scales every score exactly (adds commute with 2^k)
    scaled = lm.ScoringMatrix(pssm.data * np.float32(4.0), protein=pssm.protein)
    s4 = score_all(pli, scaled, seq, rows, m, length)
    assert torch.equal(s4, scores * 4.0)
    del s4

    # (4) argmax: Generic rule = last maximal cell in row-major order; fused == materialised
    flat = scores.view(-1)
    vmax = flat.max()
    last = int(torch.nonzero(flat == vmax)[-1])
    want_am = ((last // COLS, last % COLS), float(vmax))
    assert pli.argmax_dptr(scores.data_ptr(), rows, COLS, COLS) == want_am
    assert pli.score_argmax_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows) == want_am

    # (5) threshold at a ~1e-5 tail: row-major (row, col) list == torch.nonzero order
    t = float(torch.quantile(flat[:8_000_000][torch.isfinite(flat[:8_000_000])], 1 - 1e-5))
    want_hits = torch.nonzero(scores >= t).cpu().numpy()
    got_hits = pli.threshold_dptr(scores.data_ptr(), rows, COLS, COLS, t)
    assert got_hits.shape[0] > 100 and np.array_equal(got_hits, want_hits)
    f_hits, f_vals = pli.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1,
                                              length, 0, rows, t)
    assert np.array_equal(f_hits, want_hits)
    assert np.array_equal(f_vals, scores[want_hits[:, 0], want_hits[:, 1]].cpu().numpy())

    # (5b) a ~1e-3 tail: the candidate list, the re-scoring kernel and the device-side
    # ordering of the hit list at ~10^6 hits per Gbp, with and without the prefilter
    t3 = float(torch.quantile(flat[:8_000_000][torch.isfinite(flat[:8_000_000])], 1 - 1e-3))
    want_hits = torch.nonzero(scores >= t3).cpu().numpy()
    want_vals = scores[want_hits[:, 0], want_hits[:, 1]].cpu().numpy()
    for on in (True, False):
        pli.set_prefilter(on)
        try:
            f_hits, f_vals = pli.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1,
                                                      length, 0, rows, t3)
        finally:
            pli.set_prefilter(True)
        assert np.array_equal(f_hits, want_hits), f"prefilter={on}"
        assert np.array_equal(f_vals, want_vals)
    del want_hits, want_vals, f_hits, f_vals

    # (6) the padded tail really scores -inf (N/X column is -inf, pwm/mod.rs:422-423)
    n_pad = rows * COLS - (length + 1 - m)
    tail = torch.arange(length + 1 - m, rows * COLS, device=scores.device)
    assert n_pad == tail.numel()
    assert torch.isneginf(scores[tail % rows, tail // rows]).all()


def _generic_reductions_by_chunk(want, a, t, state):
    """Folds one chunk (rows a ... of the WHOLE-matrix oracle scores) into the Generic argmax (last maximal cell in row-major
    order, NaN never: pli/mod.rs:135-155) and the row-major hit list of `>= t` (pli/mod.rs:210-221)."""
    flat = want.reshape(-1)
    vmax = flat.max()                                     # (no NaN in these workloads: -inf and finite sums only)
    if state["best"] is None or vmax >= state["best"][1]:
        last = flat.size - 1 - int(np.argmax(flat[::-1] == vmax))
        state["best"] = ((a + last // COLS, last % COLS), vmax)
    nz = np.flatnonzero(flat >= np.float32(t))
    state["hits"].append(np.stack([a + nz // COLS, nz % COLS], axis=1))
    state["vals"].append(flat[nz])


@pytest.mark.parametrize("length,m,k", [(1_000_000_000, 20, 5), (200_000_000, 12, 21), (999_999_937, 15, 5)],
                         ids=["c2_dna_1Gbp_m20", "c5_protein_200M_m12", "dna_ragged_m15"])
def test_whole_matrix_against_the_oracle_at_baseline_sizes(gpu_pli, length, m, k):
    """EVERY cell of the BASELINE configurations against the CPU oracle -- the reference's own large tests are whole-genome
    relative oracles (lightmotif/tests/argmax.rs:41-52, tests/scan.rs:25-43: every position against Generic).  The oracle
    side is the AVX2 port on all host threads (oracle/lm_avx2.c, itself pinned bit-equal to the Generic restatement by
    tests/test_oracle_golden.py::test_avx2_port_matches_generic_scores_bitwise), chunk by chunk; the expectations of argmax,
    threshold and of every fused route are derived from THAT matrix with the Generic rules, never from the GPU's output."""
    pli = gpu_pli
    seq, rows, pssm = make_workload(pli, length, m, k, seed=4321 + m)
    scores = score_all(pli, pssm, seq, rows, m, length)
    host = co.aligned_empty((rows + m - 1, COLS), np.uint8)
    host[:] = seq.cpu().numpy()
    ref = co.Striped(host, length, m - 1, COLS, k)
    weights = co.aligned_empty(pssm.data.shape, np.float32)
    weights[:] = pssm.data
    threads = os.cpu_count() or 1
    # the generic restatement itself on the first rows (one thread: a few hundred thousand cells), then the port everywhere
    gen, _ = co.score_rows(ref, pssm.data, 0, 8192)
    chunk = 1 << 22
    buf = co.aligned_empty((chunk, COLS), np.float32)
    pinned = torch.empty((chunk, COLS), dtype=torch.float32).pin_memory()
    # a p ~ 1e-5 tail, chosen on the ORACLE's first chunk
    first = co.avx2_score_rows(ref, weights, out=buf[:min(chunk, rows)], row_begin=0, row_end=min(chunk, rows), threads=threads)
    assert np.array_equal(first[:8192].view(np.uint32), gen.view(np.uint32))
    finite = first[np.isfinite(first)]
    t = float(np.partition(finite, finite.size - finite.size // 100_000)[finite.size - finite.size // 100_000])
    state = {"best": None, "hits": [], "vals": []}
    compared = 0
    for a in range(0, rows, chunk):
        b = min(a + chunk, rows)
        want = co.avx2_score_rows(ref, weights, out=buf[:b - a], row_begin=a, row_end=b, threads=threads)
        pinned[:b - a].copy_(scores[a:b])
        torch.cuda.synchronize()
        got = pinned[:b - a].numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"rows {a} ... {b}"
        compared += got.size
        _generic_reductions_by_chunk(want, a, t, state)
    assert compared == rows * COLS >= length                          # every cell of the matrix, padded tail included
    want_am = (state["best"][0], float(state["best"][1]))
    want_hits = np.concatenate(state["hits"]).astype(np.int64)
    want_vals = np.concatenate(state["vals"])
    assert want_hits.shape[0] > 1000

    # Maximum / Threshold on the stored matrix, and every fused route, against the oracle's answers
    assert pli.argmax_dptr(scores.data_ptr(), rows, COLS, COLS) == want_am
    assert np.array_equal(np.asarray(pli.threshold_dptr(scores.data_ptr(), rows, COLS, COLS, t), np.int64), want_hits)
    del scores
    args = (pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows)
    stream = torch.cuda.current_stream().cuda_stream
    routes = {"default": {}, "exact": {"prefilter": 0}, "one_symbol": {"pair_prefilter": 0, "pair_prefilter_protein": 0},
              "pairs": {"pair_prefilter_protein": 1}, "pairs_all_rows": {"pair_prefilter_protein": 1, "drop_last": 0},
              "counted_first": {"speculate_order": 0, "short_order": 0}}
    for name, options in routes.items():
        p = lm.Pipeline.hip(0, stream=stream)
        for key, value in options.items():
            p.set_option(key, value)
        for attempt in range(2):                          # twice: a timing-dependent loss must show
            assert p.score_argmax_dptr(*args) == want_am, (name, attempt, p.last_kernel)
            f_hits, f_vals = p.score_threshold_dptr(*args, t)
            assert np.array_equal(np.asarray(f_hits, np.int64), want_hits), (name, attempt, p.last_kernel, len(f_hits), len(want_hits))
            assert np.array_equal(np.asarray(f_vals, np.float32).view(np.uint32), want_vals.view(np.uint32)), (name, attempt)

    # the stores again (two runs of one launch must agree bit for bit with the first, which the oracle checked above)
    again = score_all(pli, pssm, seq, rows, m, length)
    for a in (0, rows // 2, rows - min(chunk, rows)):
        b = min(a + chunk, rows)
        want = co.avx2_score_rows(ref, weights, out=buf[:b - a], row_begin=a, row_end=b, threads=threads)
        assert np.array_equal(again[a:b].cpu().numpy().view(np.uint32), want.view(np.uint32)), f"second run, rows {a} ... {b}"
    del again
    if k == 5:
        # Score<u8> with the DiscreteMatrix (avx2.rs:292-347: saturating sums), every cell, twice
        dm = pssm.to_discrete()
        w8 = co.aligned_empty((m, 32), np.uint8)
        w8[:] = dm.data
        want_u8 = co.avx2_score_rows_u8(ref, w8)
        out = torch.empty((rows, COLS), dtype=torch.uint8, device=seq.device)
        for attempt in range(2):
            out.zero_()
            assert pli.score_u8_dptr(dm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, out.data_ptr(), COLS)[0] == rows
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want_u8[:, :COLS]), (attempt, pli.last_kernel)


def test_device_stripe_of_100M_positions_round_trips(gpu_pli):
    """Stripe on the device at scale: position i must land at [i % R][i / R]."""
    pli = gpu_pli
    dev = torch.device("cuda", 0)
    length = 100_000_037
    enc = torch.randint(0, 4, (length,), dtype=torch.uint8, device=dev)
    rows = -(-length // COLS)
    data = torch.empty((rows + 19, COLS), dtype=torch.uint8, device=dev)
    pli.stripe_dptr(enc.data_ptr(), length, COLS, 4, 19, data.data_ptr(), COLS)
    torch.cuda.synchronize()
    want = torch.full((rows * COLS,), 4, dtype=torch.uint8, device=dev)
    want[:length] = enc
    assert torch.equal(data[:rows], want.view(COLS, rows).t())
    assert torch.equal(data[rows:, :COLS - 1], data[:19, 1:]) and (data[rows:, COLS - 1] == 4).all()


def test_more_positions_than_u32_max(gpu_pli):
    """4.4 Gbp on one GPU: flat indices and offsets exceed 2^32.  The reference's AVX2
    argmax refuses such inputs (avx2.rs:354-358); Generic (usize) handles them, so must we."""
    pli = gpu_pli
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info()
    length, m, k = 4_400_000_000, 20, 5
    rows = -(-length // COLS)
    if free < 30 * (1 << 30):
        pytest.skip("needs ~26 GB of free HBM")
    assert rows * COLS > 2 ** 32
    seq, rows, pssm = make_workload(pli, length, m, k, seed=77)
    # plant the consensus of the motif at the very end of the sequence (last column, last valid rows)
    best_syms = torch.from_numpy(np.argmax(pssm.data[:, :4], axis=1).astype(np.uint8)).to(dev)
    pos0 = length - m - 5                       # position -> (pos % rows, pos // rows)
    idx = torch.arange(pos0, pos0 + m, device=dev)
    seq[idx % rows, idx // rows] = best_syms
    scores = score_all(pli, pssm, seq, rows, m, length)
    want_cell = (pos0 % rows, pos0 // rows)
    assert want_cell[0] * COLS + want_cell[1] > 2 ** 32
    max_score = float(np.float32(sum(np.float32(pssm.data[j, int(best_syms[j])]) for j in range(m))))
    got = pli.argmax_dptr(scores.data_ptr(), rows, COLS, COLS)
    fused = pli.score_argmax_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows)
    assert got[0] == want_cell == fused[0]
    assert abs(got[1] - max_score) < 1e-3 and got[1] == fused[1]
    # window at the end against the oracle
    a = rows - 2048
    host = seq[a:rows + m - 1].cpu().numpy()
    win = co.Striped(host, length, m - 1, COLS, k)
    want, _ = co.score_rows(win, pssm.data, 0, rows - a)
    assert np.array_equal(scores[a:].cpu().numpy().view(np.uint32), want.view(np.uint32))
    # threshold just below the planted maximum: the hit list must contain the planted cell
    hits = pli.threshold_dptr(scores.data_ptr(), rows, COLS, COLS, got[1] - 1e-3)
    assert [want_cell[0], want_cell[1]] in hits.tolist()
    f_hits, _ = pli.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length,
                                         0, rows, got[1] - 1e-3)
    assert np.array_equal(f_hits, hits)


@pytest.mark.parametrize("kind", ["normal", "ties", "late_maximum", "mostly_n"])
def test_fused_argmax_candidate_route_against_oracle(gpu_pli, kind):
    """From ~100 M cells per call the fused argmax takes the sample -> prefilter scan -> exact re-scoring
    route (score_argmax.hip: argmax_by_prefilter).  It must return the Generic answer -- the LAST
    maximal cell -- also when many cells tie (lists overflow -> exact kernel), when the sample
    misses the region holding the maximum, and when most of the sample is -inf."""
    import lightmotif_amd as lm
    from oracle import c_oracle as co
    pli = gpu_pli
    rng = np.random.default_rng({"normal": 1, "ties": 2, "late_maximum": 3, "mostly_n": 4}[kind])
    length, m = 104_000_123, 16
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    p = np.zeros((m, 8), np.float32)
    p[:, :4] = rng.integers(-2, 3, (m, 4)) if kind == "ties" else rng.normal(0, 2, (m, 4))
    p[:, 4] = -np.inf
    if kind == "late_maximum":
        # the consensus occurs exactly once, in the last striped rows of the last column
        best = "".join("ACTG"[i] for i in p[:, :4].argmax(axis=1))
        enc[length - 40:length - 40 + m] = [("ACTG").index(ch) for ch in best]
    if kind == "mostly_n":
        enc[rng.random(length) < 0.97] = 4
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, m - 1)
    want, _ = co.score_rows(ref, p)
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure_wrap(m - 1)
    pssm = lm.ScoringMatrix(p)
    got = pli.score_argmax(pssm, seq)
    assert got[0] == co.argmax(want, 32)
    assert np.float32(got[1]).view(np.uint32) == np.float32(co.max_(want, 32)).view(np.uint32)
    if kind == "normal":
        assert pli.last_kernel .startswith("score_c32_prefilter")
        pli.set_prefilter(False)
        try:
            assert pli.score_argmax(pssm, seq) == got and pli.last_kernel.startswith("score_c32<16,1>")
        finally:
            pli.set_prefilter(True)
    # a batch mixing a qualifying motif, a short one (too many ties -> exact kernel) and a long one
    others = [lm.ScoringMatrix(np.ascontiguousarray(p[:5])), pssm,
              lm.ScoringMatrix(np.concatenate([p, p[:4]]))]
    seq.configure_wrap(19)
    ref20 = co.stripe(enc, 32, 5)
    co.configure_wrap(ref20, 19)
    res = pli.scan_argmax_batch(others, seq)
    for q, r in zip(others, res):
        w, _ = co.score_rows(ref20, q.data)
        assert r[0] == co.argmax(w, 32), len(q)


@pytest.mark.parametrize("kind", ["normal", "ties", "all_neg_inf", "nan_weights", "first_cell_nan"])
def test_tracked_argmax_of_score_into(gpu_pli, kind):
    """score_into on >= 8 Mi cells tracks the maximum in the store kernel; argmax on the same
    handle must still be the Generic answer (last maximal cell; NaN never wins; a NaN in cell
    (0, 0) wins outright), equal to the second-pass reduction, with the knob on and off."""
    import lightmotif_amd as lm
    from oracle import c_oracle as co
    pli = gpu_pli
    rng = np.random.default_rng({"normal": 11, "ties": 12, "all_neg_inf": 13, "nan_weights": 14,
                                 "first_cell_nan": 15}[kind])
    length, m = 9_000_017, 12
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    p = np.zeros((m, 8), np.float32)
    p[:, :4] = rng.integers(-1, 2, (m, 4)) if kind == "ties" else rng.normal(0, 2, (m, 4))
    p[:, 4] = -np.inf
    if kind == "all_neg_inf":
        enc[:] = 4
    if kind == "nan_weights":
        p[3, 1] = np.nan                       # every window with a C at offset 3 scores NaN
    if kind == "first_cell_nan":
        p[0, int(enc[0])] = np.nan             # cell (0, 0) is NaN (and many others)
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, m - 1)
    want, _ = co.score_rows(ref, p)
    want_am = co.argmax(want, 32)
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure_wrap(m - 1)
    pssm = lm.ScoringMatrix(p)
    for on in (True, False):
        pli.set_track_argmax(on)
        try:
            scores = pli.score(pssm, seq)
            got = pli.argmax(scores)
            gmax = pli.max(scores)
        finally:
            pli.set_track_argmax(True)
        assert got == want_am, (kind, on)
        wmax = co.max_(want, 32)
        assert np.float32(gmax).view(np.uint32) == np.float32(wmax).view(np.uint32), (kind, on)
        assert pli.argmax_dptr(scores.data_ptr, scores.rows, 32, 32)[0] == want_am
    # scoring another motif into the same handle replaces the cached result
    p2 = p.copy()
    p2[:, :4] = rng.normal(0, 2, (m, 4))
    p2[np.isnan(p2)] = 0.5
    want2, _ = co.score_rows(ref, p2)
    pli.score_into(lm.ScoringMatrix(p2), seq, scores)
    assert pli.argmax(scores) == co.argmax(want2, 32)


@pytest.mark.parametrize("offset", [1, 2, 3])
def test_unaligned_sequence_pointer(gpu_pli, offset):
    """Device pointers handed over by a caller need not be 4-byte aligned: the kernels that fetch
    symbols with dword loads must step aside (byte-load variants), results unchanged."""
    pli = gpu_pli
    dev = torch.device("cuda", 0)
    length, m = 3_000_017, 20
    rows = -(-length // COLS)
    rng = np.random.default_rng(offset)
    sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]
    pssm = lm.create(sites).counts.normalize(0.1).log_odds()
    flat = torch.zeros((rows + m - 1) * COLS + 8, dtype=torch.uint8, device=dev)
    aligned = flat[: (rows + m - 1) * COLS].view(rows + m - 1, COLS)
    aligned[:rows] = torch.randint(0, 4, (rows, COLS), dtype=torch.uint8, device=dev)
    pli.configure_wrap_dptr(aligned.data_ptr(), rows, COLS, COLS, m - 1, 4)
    shifted = torch.zeros_like(flat)
    shifted[offset: offset + aligned.numel()] = aligned.reshape(-1)
    ptr_a, ptr_u = aligned.data_ptr(), shifted.data_ptr() + offset
    assert ptr_u % 4 == offset % 4
    out_a = torch.empty((rows, COLS), dtype=torch.float32, device=dev)
    out_u = torch.empty_like(out_a)
    pli.score_dptr(pssm, ptr_a, rows + m - 1, COLS, COLS, m - 1, length, 0, rows, out_a.data_ptr(), COLS)
    pli.score_dptr(pssm, ptr_u, rows + m - 1, COLS, COLS, m - 1, length, 0, rows, out_u.data_ptr(), COLS)
    torch.cuda.synchronize()
    assert torch.equal(out_a.view(torch.int32), out_u.view(torch.int32))
    am_a = pli.score_argmax_dptr(pssm, ptr_a, rows + m - 1, COLS, COLS, m - 1, length, 0, rows)
    am_u = pli.score_argmax_dptr(pssm, ptr_u, rows + m - 1, COLS, COLS, m - 1, length, 0, rows)
    assert am_a == am_u
    t = float(torch.quantile(out_a.view(-1)[:4_000_000], 1 - 1e-4))
    th_a = pli.score_threshold_dptr(pssm, ptr_a, rows + m - 1, COLS, COLS, m - 1, length, 0, rows, t)
    assert pli.last_kernel == "score_c32_prefilter2"
    th_u = pli.score_threshold_dptr(pssm, ptr_u, rows + m - 1, COLS, COLS, m - 1, length, 0, rows, t)
    assert pli.last_kernel == "score_c32_prefilter"
    assert np.array_equal(th_a[0], th_u[0]) and np.array_equal(th_a[1], th_u[1]) and len(th_a[0]) > 50


def test_hit_ordering_with_and_without_known_count():
    """The fused threshold orders its hit list either after reading the count
    (option "speculate_order" = 0) or before (default; sized from the previous call and redone
    when that guess is off): both give the materialised result whatever the call history."""
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream().cuda_stream
    plis = {}
    for flag in ("0", "1", "0-buckets", "1-buckets", "1-five-launches"):
        plis[flag] = lm.Pipeline.hip(0, stream=stream)
        plis[flag].set_option("speculate_order", int(flag[0]))
        if flag.endswith("buckets"):
            plis[flag].set_option("sort_hits", 0)     # long lists: radix sort (default) or the bucket passes
        if flag.endswith("five-launches"):
            plis[flag].set_option("short_order", 0)   # short lists: counted by the re-scoring kernel (default) or by the bucket passes
    length, m = 40_000_003, 10
    seq, rows, pssm = make_workload(plis["1"], length, m, 5, seed=77)
    scores = score_all(plis["1"], pssm, seq, rows, m, length)
    flat = scores.flatten()
    sample = flat[torch.randint(0, flat.numel(), (1 << 20,), device=flat.device)]
    qs = torch.tensor([1.0, 0.99999, 0.9, 0.999999, 0.999, 1.0, 0.97, 0.9999], device=flat.device)
    ts = torch.quantile(sample, qs).tolist()
    ts[0] = float(flat.max()) + 1.0  # no hit at all
    ts[5] = float(flat.max())        # the best cell(s) only
    for t in ts:  # the hit count jumps by orders of magnitude between calls
        want = torch.nonzero(scores >= t).cpu().numpy()  # the padded tail scores -inf
        want_vals = scores[want[:, 0], want[:, 1]].cpu().numpy()
        for flag, pli in plis.items():
            hits, vals = pli.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1,
                                                  length, 0, rows, t)
            assert np.array_equal(hits, want), (flag, t)
            assert np.array_equal(vals, want_vals), (flag, t)


@pytest.mark.parametrize("m,kind", [(1, "random"), (3, "random"), (5, "ties"), (6, "random"), (6, "absent_in_suffix"),
                                    (4, "only_at_the_end"), (7, "finite_n"), (9, "random"), (5, "range")])
def test_fused_argmax_of_short_motifs_from_the_last_rows(m, kind):
    """Short motifs are settled from the last rows of the range when those hold a best k-mer
    (score == sum of the row maxima); otherwise the usual routes run.  Both must give the
    Generic argmax (pli/mod.rs:135-155) -- also when the best k-mer is missing from the suffix,
    sits in the very last valid window only, ties abound, N has finite weights, or a row
    range is scored."""
    plain = lm.Pipeline.hip(0)
    plain.set_option("suffix_argmax", 0)
    suffix = lm.Pipeline.hip(0)
    rng = np.random.default_rng(1000 + m)
    length = 6_000_011
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    p = np.zeros((m, 8), np.float32)
    p[:, :4] = rng.integers(-2, 3, (m, 4)) if kind == "ties" else rng.normal(0, 2, (m, 4))
    p[:, 4] = rng.normal(0, 1, m) if kind == "finite_n" else -np.inf
    best = p[:, :4].argmax(axis=1).astype(np.uint8)
    rows = -(-length // 32)
    if kind in ("absent_in_suffix", "only_at_the_end"):
        worst = p[:, :4].argmin(axis=1)
        # the last 60 000 rows of every column avoid the best k-mer's first symbol
        ref = co.stripe(enc, 32, 5)
        view = ref.data[:rows, :32]
        tail = view[rows - 60_000:]
        tail[tail == best[0]] = np.uint8(worst[0] if worst[0] != best[0] else (best[0] + 1) % 4)
        enc = view.T.reshape(-1)[:length].copy()
        if kind == "only_at_the_end":
            enc[length - m:] = best                      # the very last valid window
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, m - 1)
    a, b = (rows // 5, rows - 777) if kind == "range" else (0, rows)
    want, _ = co.score_rows(ref, p, a, b)
    want_cell = co.argmax(want, 32)
    want_val = want[want_cell]
    pssm = lm.ScoringMatrix(p)
    for pli in (suffix, plain):
        seq = pli.stripe(lm.EncodedSequence(enc), 32)
        seq.configure_wrap(m - 1)
        got = pli.score_argmax(pssm, seq, range(a, b))
        assert got[0] == want_cell and np.float32(got[1]) == want_val, (kind, pli is suffix)
        many = pli.scan_argmax_batch([pssm, pssm.reverse_complement(), pssm], seq)
        if kind != "range":
            assert many[0] == got and many[2] == got


def test_fused_argmax_suffix_route_random_motifs(pli):
    """Many random short motifs (lengths 1..8, small-integer weights = many ties, some with a
    scoring N) over one 5 Mbp sequence: batch and single calls equal the oracle's Generic argmax."""
    rng = np.random.default_rng(4242)
    length = 5_000_003
    enc = rng.integers(0, 5, length, dtype=np.uint8)
    enc[rng.random(length) < 0.98] %= 4                      # a sprinkle of N
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, 8)
    seq = pli.stripe(lm.EncodedSequence(enc), 32)
    seq.configure_wrap(8)
    pssms, wants = [], []
    for i in range(24):
        m = int(rng.integers(1, 9))
        p = np.zeros((m, 8), np.float32)
        p[:, :4] = rng.integers(-3, 4, (m, 4)) if i % 3 == 0 else rng.normal(0, 2, (m, 4))
        p[:, 4] = rng.normal(-1, 1, m) if i % 4 == 1 else -np.inf
        want, _ = co.score_rows(ref, p)
        cell = co.argmax(want, 32)
        pssms.append(lm.ScoringMatrix(p))
        wants.append((cell, float(want[cell])))
    got = pli.scan_argmax_batch(pssms, seq)
    assert got == wants
    for k in (0, 7, 13):
        assert pli.score_argmax(pssms[k], seq) == wants[k]


def test_candidate_route_argmax_batches_share_passes(gpu_pli):
    """Three and five motifs of one length over 40 Mbp each (>= 100 M cells per call: the candidate
    route; several motifs per pass, padded job table): the batch equals the single-motif calls."""
    pli = gpu_pli
    length = 40_000_000
    for m, count in ((12, 3), (10, 5)):
        seq, rows, _ = make_workload(pli, length, m, 5, seed=31 + m)
        rng = np.random.default_rng(m)
        pssms = []
        for _ in range(count):
            sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(8)]
            pssms.append(lm.create(sites).counts.normalize(0.1).log_odds())
        singles = [pli.score_argmax_dptr(p, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows)
                   for p in pssms]
        handle = lm.StripedSequence.from_device(pli, seq, length, m - 1) if hasattr(lm.StripedSequence, "from_device") else None
        if handle is None:
            host = seq[:rows].cpu().numpy()
            enc = host.T.reshape(-1)[:length].copy()
            handle = pli.stripe(lm.EncodedSequence(enc), 32)
            handle.configure_wrap(m - 1)
        batch = pli.scan_argmax_batch(pssms, handle)
        assert pli.last_kernel in ("score_c32_prefilter2_multi", "argmax_collect", "score_c32_prefilter2",
                                   "score_c32_prefilter")
        assert batch == singles


@pytest.mark.parametrize("m", [20, 40], ids=["m20", "m40_long"])
def test_more_than_2_32_cells_on_one_gpu(gpu_pli, m):
    """4.5 Gbp on one GPU (4.5 GB of symbols, 18 GB of scores): row-major cell indices and sequence
    positions pass 2^32, which `configs[3]`'s per-GPU shards never do.  The consensus k-mer of the PSSM
    is planted at a low cell and at cells / positions beyond 2^32, so the maximum value occurs there
    (and wherever chance put an equally good k-mer): argmax (materialised, tracked, fused) must return the LAST planted cell, threshold
    (materialised, fused, Scanner positions) exactly the planted ones in order, and sampled windows
    must equal the oracle bit for bit."""
    pli = gpu_pli
    dev = torch.device("cuda", 0)
    length, k = 4_500_000_000, 5
    rows = -(-length // COLS)
    assert rows * COLS > 2 ** 32
    gen = torch.Generator(device=dev)
    gen.manual_seed(4242)
    seq = torch.empty((rows + m - 1, COLS), dtype=torch.uint8, device=dev)
    step = 1 << 24
    for a in range(0, rows, step):
        b = min(a + step, rows)
        seq[a:b] = torch.randint(0, 4, (b - a, COLS), dtype=torch.uint8, device=dev, generator=gen)
    rng = np.random.default_rng(4242)
    sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]
    pssm = lm.create(sites).counts.normalize(0.1).log_odds()
    consensus = torch.from_numpy(np.argmax(pssm.data[:, :4], axis=1).astype(np.uint8)).to(dev)
    planted = [(1000, 3), (139_000_000, 17), (139_000_007, 31)]          # (row, col), ascending row-major order
    for r, c in planted:
        seq[r:r + m, c] = consensus
    pli.configure_wrap_dptr(seq.data_ptr(), rows, COLS, COLS, m - 1, 4)
    assert planted[1][0] * COLS + planted[1][1] > 2 ** 32
    assert planted[2][1] * rows + planted[2][0] > 2 ** 32              # sequence position of the last one
    best = float(np.float32(0.0))
    acc = np.float32(0.0)
    for j in range(m):
        acc = np.float32(acc + pssm.data[j, int(consensus[j])])
    best = float(acc)

    scores = score_all(pli, pssm, seq, rows, m, length)
    for a in (0, planted[1][0] - 100, rows - 4096):
        b = min(a + 4096, rows)
        win = co.Striped(seq[a:b + m - 1].cpu().numpy(), length, m - 1, COLS, k)
        want, _ = co.score_rows(win, pssm.data, 0, b - a)
        assert np.array_equal(scores[a:b].cpu().numpy().view(np.uint32), want.view(np.uint32)), f"window at row {a}"
    for r, c in planted:
        assert float(scores[r, c]) == best
    # argmax: the last maximal cell in row-major order (pli/mod.rs:146 `>=`)
    want_am = (planted[-1], best)
    assert pli.argmax_dptr(scores.data_ptr(), rows, COLS, COLS) == want_am
    assert pli.score_argmax_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows) == want_am
    # threshold at the maximum: the planted cells (+ any natural occurrence of an equally good k-mer), row-major
    parts = []                                   # (torch.nonzero itself fails beyond 2^32 elements: go by pieces)
    for a in range(0, rows, 1 << 25):
        nz = torch.nonzero(scores[a:a + (1 << 25)] >= best)
        nz[:, 0] += a
        parts.append(nz.cpu().numpy())
    want_hits = np.concatenate(parts)
    assert set(planted) <= set(map(tuple, want_hits.tolist())) and len(want_hits) < 100
    assert np.array_equal(pli.threshold_dptr(scores.data_ptr(), rows, COLS, COLS, best), want_hits)
    f_hits, f_vals = pli.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows,
                                              best)
    assert np.array_equal(f_hits, want_hits) and (f_vals == np.float32(best)).all()
    # a p ~ 1e-5 tail over 4.5e9 cells: ~45 000 hits whose keys pass 2^32; fused == materialised
    sample = scores[:1 << 18].flatten()
    t = float(torch.quantile(sample[torch.isfinite(sample)], 1 - 1e-5))
    mat_hits = pli.threshold_dptr(scores.data_ptr(), rows, COLS, COLS, t)
    f_hits, f_vals = pli.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, t)
    assert mat_hits.shape[0] > 10_000 and np.array_equal(f_hits, mat_hits)
    flat = mat_hits[:, 0] * COLS + mat_hits[:, 1]
    assert (np.diff(flat) > 0).all() and flat[-1] > 2 ** 32
    idx = torch.from_numpy(mat_hits).to(dev)
    assert np.array_equal(f_vals, scores[idx[:, 0], idx[:, 1]].cpu().numpy())
    del scores, idx
    # the reference's own flow on handles: score_into (tracked maximum) + argmax; Scanner positions
    sseq = pli.adopt_sequence(seq.data_ptr(), rows, m - 1, COLS, COLS, length, keepalive=seq)
    sc = lm.StripedScores.empty(pli, COLS)
    pli.score_into(pssm, sseq, sc)
    assert pli.argmax(sc) == planted[-1] and sc.max_index == length - m + 1
    del sc
    scanner = lm.Scanner(pssm, sseq, threshold=best)
    want_pos = sorted(int(c) * rows + int(r) for r, c in want_hits.tolist())
    assert scanner.positions.tolist() == want_pos and want_pos[-1] > 2 ** 32
    assert (scanner.scores == np.float32(best)).all()


# cells per trip of one argmax_flat workgroup (reduce.hip kArgmaxSpan) and per threshold_count chunk
_SPAN, _CHUNK = 8192, 4096


@pytest.mark.parametrize("ncells", [1, 3, 4, 5, 1023, 1024, _CHUNK - 1, _CHUNK, _CHUNK + 1, _SPAN - 1, _SPAN, _SPAN + 1,
                                    3 * _SPAN + 5, 17 * _SPAN + 1027, 4097 * _SPAN + 7, 65536 * _SPAN + 8 * 4096 + 3])
def test_materialised_reductions_at_span_and_chunk_boundaries(gpu_pli, ncells):
    """Maximum / Threshold over a stored matrix (pli/mod.rs:135-160, 210-221) on sizes around the
    kernels' work units: argmax_flat gives every workgroup one contiguous span (one trip of 8 192 cells
    up to 65 536 workgroups, records beyond 4 096 folded), threshold_count reads whole 4 096-cell chunks
    lane-contiguously and the ragged last chunk cell by cell.  One column, so the matrix is flat for every
    size; ties planted at the first and last cell and on both sides of a span boundary."""
    pli, dev = gpu_pli, torch.device("cuda", 0)
    rng = np.random.default_rng(ncells % 9973)
    host = rng.standard_normal(ncells).astype(np.float32)
    top = np.float32(host.max() + 1)
    for at in {0, ncells - 1, min(_SPAN - 1, ncells - 1), min(_SPAN, ncells - 1), ncells // 2}:
        host[at] = top
    if ncells > 7:
        host[5] = np.nan
        host[ncells - 3] = -np.inf
    scores = torch.from_numpy(host).to(dev)
    want = co.argmax(host.reshape(-1, 1), 1)
    got = pli.argmax_dptr(scores.data_ptr(), ncells, 1, 1)
    assert got is not None and got[0] == want == (ncells - 1, 0) and got[1] == top
    for t in (float(top), 2.5, -np.inf):
        if int((host >= np.float32(t)).sum()) > 2_000_000:
            continue
        w = co.threshold(host.reshape(-1, 1), 1, t)
        g = pli.threshold_dptr(scores.data_ptr(), ncells, 1, 1, t)
        assert np.array_equal(g.astype(np.uint64), w.astype(np.uint64)), (ncells, t)
    # the last maximal cell somewhere inside: every later cell is smaller
    if ncells > 2 * _SPAN:
        host2 = host.copy()
        host2[host2 == top] = 0
        for at in (_SPAN - 1, _SPAN, ncells - _SPAN - 1):
            host2[at] = top
        scores.copy_(torch.from_numpy(host2))
        assert pli.argmax_dptr(scores.data_ptr(), ncells, 1, 1)[0] == co.argmax(host2.reshape(-1, 1), 1) \
            == (ncells - _SPAN - 1, 0)
    # a NaN first cell answers (0, 0) under the first-cell rule, and is ignored without it
    host[0] = np.nan
    scores.copy_(torch.from_numpy(host))
    assert pli.argmax_dptr(scores.data_ptr(), ncells, 1, 1)[0] == co.argmax(host.reshape(-1, 1), 1) == (0, 0)
    if ncells > 1:
        assert pli.argmax_dptr(scores.data_ptr(), ncells, 1, 1, first_cell_rule=False)[0] == (ncells - 1, 0)


def test_dense_results_at_full_size(gpu_pli):
    """A threshold EVERY cell passes, at BASELINE's full size: 10^9 (row, col) pairs = 16 GB of coordinates (pli/mod.rs:210-221
    pushes them all).  A finite N weight makes the padded tail qualify too.  Each entry point either returns
    the complete row-major list or a clean LM_HIP_ERR_OOM; afterwards the context works, holds no tens of gigabytes of
    scratch, and the page-locked result pool is back under its cap."""
    import ctypes as C
    from lightmotif_amd import _ffi
    L = _ffi.lib()
    pli = gpu_pli
    length, m = 1_000_000_000 - 5, 20
    seq, rows, pssm = make_workload(pli, length, m, 5, seed=4242)
    w = pssm.data.copy()
    w[:, 4] = -1.0                                                      # N scores finitely: no -inf anywhere
    pssm = lm.ScoringMatrix(w)
    scores = score_all(pli, pssm, seq, rows, m, length)
    ncells = rows * COLS
    free0, _ = torch.cuda.mem_get_info()

    def pool_ok():
        idle, used, budget = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        assert L.lm_hip_result_pool_info(C.byref(idle), C.byref(used), C.byref(budget)) == 0
        return idle.value <= budget.value and used.value == 0

    def small_call_works():
        t = float(scores[:4096].max())
        got = pli.threshold_dptr(scores.data_ptr(), 4096, COLS, COLS, t)
        want = torch.nonzero(scores[:4096] >= t).cpu().numpy()
        return np.array_equal(got, want)

    # 1. Threshold on the materialised matrix
    ptr, n = C.POINTER(_ffi.Coords)(), C.c_size_t(0)
    st = L.lm_hip_threshold_f32_dptr(pli._h, C.c_void_p(scores.data_ptr()), rows, COLS, COLS, C.c_float(float("-inf")),
                                     C.byref(ptr), C.byref(n))
    assert st in (_ffi.OK, _ffi.ERR_OOM), _ffi.last_error()
    if st == _ffi.OK:
        assert n.value == ncells
        for i in (0, 1, 31, 32, 33, ncells // 2 + 7, ncells - 2, ncells - 1):
            assert (ptr[i].row, ptr[i].col) == (i // COLS, i % COLS), i
        L.lm_hip_free(ptr)
    assert small_call_works() and pool_ok()

    # 2. fused score + threshold (coordinates and values)
    ptr, vals, n = C.POINTER(_ffi.Coords)(), C.POINTER(C.c_float)(), C.c_size_t(0)
    st = L.lm_hip_score_threshold_f32_dptr(pli._h, pssm._device(pli), C.c_void_p(seq.data_ptr()), rows + m - 1, COLS, COLS,
                                           m - 1, length, 0, rows, C.c_float(float("-inf")), C.byref(ptr), C.byref(vals),
                                           C.byref(n))
    assert st in (_ffi.OK, _ffi.ERR_OOM), _ffi.last_error()
    if st == _ffi.OK:
        assert n.value == ncells
        idx = [0, 1, 33, ncells // 3, ncells - 1]
        want = scores.flatten()[torch.tensor(idx, device=scores.device)].cpu().numpy()
        for i, wv in zip(idx, want):
            assert (ptr[i].row, ptr[i].col) == (i // COLS, i % COLS) and np.float32(vals[i]) == wv, i
        L.lm_hip_free(ptr)
        L.lm_hip_free(vals)
    assert small_call_works() and pool_ok()

    # 3. Scanner semantics: ascending position, only windows that fit (scan.rs:185-190)
    h = pli.adopt_sequence(seq.data_ptr(), rows, m - 1, COLS, COLS, length, keepalive=seq)
    hits, n = C.POINTER(_ffi.Hit)(), C.c_size_t(0)
    st = L.lm_hip_scan_f32(pli._h, pssm._device(pli), h._h, C.c_float(float("-inf")), C.byref(hits), C.byref(n))
    assert st in (_ffi.OK, _ffi.ERR_OOM), _ffi.last_error()
    if st == _ffi.OK:
        assert n.value == length - m + 1
        for i in (0, 1, 12345, rows, rows + 1, n.value - 1):
            assert hits[i].position == i and np.float32(hits[i].score) == np.float32(scores[i % rows, i // rows].item()), i
        L.lm_hip_free(hits)
    assert small_call_works() and pool_ok()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (6 << 30), f"the context kept {(free0 - free1) >> 30} GB of scratch after dense results"
