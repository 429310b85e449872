"""The literal drop-in path: the host-pointer entry points a reference-side shim binds --
`lm_hip_score_f32`, `lm_hip_argmax_f32`, `lm_hip_max_f32`, `lm_hip_threshold_f32`, `lm_hip_score_u8` -- against
the oracle, bit for bit, over every route inside `csrc/hostptr.hip`: zero-copy (tiny), copy (small), the tile
pipeline (large, several tiles, ragged last tile), the piecewise fallback when the pipeline is taken, row ranges,
`out_stride != cols`, column counts other than 32, protein, the per-thread PSSM cache and its eviction, and host
threads calling in concurrently (pwm/mod.rs:640-648, scores.rs:181-213, avx2.rs:889-904, scan.rs:174-178)."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

import lightmotif_amd as lm
from lightmotif_amd import _ffi
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def random_pssm(rng, m, k, kind="normal"):
    p = np.zeros((m, co.stride(k, 4)), np.float32)
    p[:, :k] = rng.integers(-2, 3, (m, k)) if kind == "ties" else rng.normal(0, 2, (m, k))
    p[:, k - 1] = -np.inf
    return p


def aligned(p):
    q = co.aligned_empty(p.shape, np.float32)
    q[:] = p
    return q


def host_score(s, p, k, a, b, out_stride=None, sentinel=None):
    """lm_hip_score_f32 on the oracle's own striped matrix; returns (out incl. padding columns, out_rows, max_index)"""
    L = _ffi.lib()
    cols = s.cols
    ost = co.stride(cols, 4) if out_stride is None else out_stride
    out = np.full((max(b - a, 0), ost), np.float32(777.0) if sentinel is None else sentinel, np.float32)
    orow, omi = C.c_size_t(123), C.c_size_t(456)
    st = L.lm_hip_score_f32(s.data.ctypes.data, s.data.shape[0], s.stride, cols, s.wrap, s.length, p.ctypes.data,
                            p.shape[0], p.shape[1], k, a, b, out.ctypes.data, ost, C.byref(orow), C.byref(omi))
    assert st == 0, _ffi.last_error()
    return out, orow.value, omi.value


def host_argmax(scores, rows, stride, cols):
    L = _ffi.lib()
    found, best, val = C.c_int(0), _ffi.Coords(), C.c_float(0)
    assert L.lm_hip_argmax_f32(scores.ctypes.data, rows, stride, cols, C.byref(found), C.byref(best), C.byref(val)) == 0, \
        _ffi.last_error()
    return ((best.row, best.col), np.float32(val.value)) if found.value else None


def host_threshold(scores, rows, stride, cols, t):
    L = _ffi.lib()
    ptr, n = C.POINTER(_ffi.Coords)(), C.c_size_t(0)
    assert L.lm_hip_threshold_f32(scores.ctypes.data, rows, stride, cols, t, C.byref(ptr), C.byref(n)) == 0, _ffi.last_error()
    got = np.array([(ptr[i].row, ptr[i].col) for i in range(n.value)], dtype=np.uintp).reshape(-1, 2)
    L.lm_hip_free(ptr)
    return got


def striped(rng, length, cols, k, m):
    enc = rng.integers(0, k - 1, length, dtype=np.uint8)
    enc[rng.random(length) < 0.001] = k - 1
    s = co.stripe(enc, cols, k)
    co.configure_wrap(s, m - 1)
    return s


@pytest.mark.parametrize("length,cols,k,m,rows", [
    (6_000, 32, 5, 15, None),            # tiny: 188 rows -- symbols and scores through pinned memory, no copy command
    (32_768 - 40, 32, 5, 20, None),      # the largest zero-copy shape (1023 rows)
    (32_768 + 64, 32, 5, 20, None),      # just beyond it: the copy path
    (464_165, 32, 5, 15, None),          # the reference's own bench size (dna.rs:81-109)
    (464_165, 32, 5, 15, (1000, 9000)),  # a row range (Score::score_rows_into)
    (200_000, 32, 21, 12, None),         # protein, wide-alphabet kernels
    (150_000, 20, 5, 9, (7, 4000)),      # 20 columns: stride 32 in, 24 out; generic / tiled kernels
    (90_000, 1, 5, 15, None),            # C = 1 (lightmotif-bench's Generic geometry)
    (500_000, 32, 5, 40, None),          # the long kernel family
])
def test_score_f32_matches_oracle(length, cols, k, m, rows):
    rng = np.random.default_rng(length + cols + m)
    s = striped(rng, length, cols, k, m)
    p = random_pssm(rng, m, k)
    a, b = (0, s.rows) if rows is None else rows
    want, mi = co.score_rows(s, p, a, b)
    got, orow, omi = host_score(s, p, k, a, b)
    assert (orow, omi) == (b - a, mi)
    assert np.array_equal(bits(got[:, :cols]), bits(want[:, :cols]))
    assert np.all(got[:, cols:] == 777.0), "alignment padding of the caller's rows was written (pli/mod.rs:103 leaves it)"
    am = host_argmax(got, b - a, got.shape[1], cols)
    assert am[0] == co.argmax(want, cols) and bits(am[1]) == bits(co.max_(want, cols))
    t = float(np.sort(want[:, :cols][np.isfinite(want[:, :cols])])[-50])
    assert np.array_equal(host_threshold(got, b - a, got.shape[1], cols, t), co.threshold(want, cols, t))


@pytest.mark.parametrize("out_stride", [32, 40])
def test_pipeline_tiles_row_range_and_out_stride(out_stride):
    """Large enough for the tile pipeline (101 MB of scores: a dozen tiles and a ragged one), a row range that starts inside
    the matrix, the caller's rows wider than the scored columns: every tile boundary must be seamless."""
    rng = np.random.default_rng(out_stride)
    m, rows_total = 20, 262_144 * 3 + 5_000
    length = rows_total * 32 - 11
    s = striped(rng, length, 32, 5, m)
    p = random_pssm(rng, m, 5)
    a, b = 1_234, rows_total - 321
    want = co.aligned_empty((b - a, 32), np.float32)
    co.avx2_score_rows(s, aligned(p), out=want, row_begin=a, row_end=b, threads=8)  # == Generic bitwise (test_oracle_golden)
    chk, _ = co.score_rows(s, p, a, a + 2_000)
    assert np.array_equal(bits(chk[:, :32]), bits(want[:2_000]))
    got, orow, omi = host_score(s, p, 5, a, b, out_stride=out_stride)
    assert (orow, omi) == (b - a, length + 1 - m)
    assert np.array_equal(bits(got[:, :32]), bits(want))
    if out_stride > 32:
        assert np.all(got[:, 32:] == 777.0)
    # the reductions on the same host matrix (4 B per cell up): dense or padded rows
    am = host_argmax(got, b - a, out_stride, 32)
    assert am[0] == co.argmax(want, 32)
    t = float(np.partition(want[np.isfinite(want)], -300)[-300])
    assert np.array_equal(host_threshold(got, b - a, out_stride, 32, t), co.threshold(want, 32, t))


def test_pssm_cache_alternating_motifs_and_eviction():
    """One thread, many matrices: the lane keeps the device tables of the last 16 -- alternating between matrices of the
    same shape, more than 16 distinct ones, and a matrix that differs from a cached one in a single weight."""
    rng = np.random.default_rng(4)
    s = striped(rng, 70_000, 32, 5, 33)
    motifs = [random_pssm(rng, m, 5) for m in (8, 8, 12, 15, 20, 20, 24, 33)] + [random_pssm(rng, 11, 5) for _ in range(14)]
    twin = motifs[4].copy()
    twin[7, 2] = np.nextafter(twin[7, 2], np.float32(np.inf))
    motifs.append(twin)
    wants = [co.score_rows(s, p)[0] for p in motifs]
    order = list(range(len(motifs))) * 2 + [4, len(motifs) - 1, 4, 0, 1, 0, 1]
    for i in order:
        got, _, _ = host_score(s, motifs[i], 5, 0, s.rows)
        assert np.array_equal(bits(got[:, :32]), bits(wants[i][:, :32])), i


def test_threads_are_bit_exact_and_overlap():
    """Eight host threads (the CLI's `-j`, main.rs:270; Python threads with the GIL released, lib.rs:865) call the
    host-pointer functions at once: each gets a lane of its own, results stay bit-identical, and the wall time is well
    under the same calls made one after the other."""
    L = _ffi.lib()
    rng = np.random.default_rng(8)
    nthreads, m, iters = 8, 15, 60
    jobs = []
    for t in range(nthreads):
        s = striped(rng, 464_165 + 1000 * t, 32, 5, m)
        p = random_pssm(rng, m, 5)
        want, _ = co.score_rows(s, p)
        jobs.append((s, p, want, np.zeros((s.rows, 32), np.float32)))

    def work(job, n):
        s, p, want, out = job
        orow, omi = C.c_size_t(0), C.c_size_t(0)
        found, best, val = C.c_int(0), _ffi.Coords(), C.c_float(0)
        for _ in range(n):
            st = L.lm_hip_score_f32(s.data.ctypes.data, s.data.shape[0], 32, 32, s.wrap, s.length, p.ctypes.data, m, p.shape[1],
                                    5, 0, s.rows, out.ctypes.data, 32, C.byref(orow), C.byref(omi))
            assert st == 0
            assert L.lm_hip_argmax_f32(out.ctypes.data, s.rows, 32, 32, C.byref(found), C.byref(best), C.byref(val)) == 0
        job_best.append(((best.row, best.col), co.argmax(want, 32)))

    job_best = []
    for j in jobs:                                   # warm: lanes of the main thread, tables, clocks
        work(j, 3)
    th = [threading.Thread(target=work, args=(j, 3)) for j in jobs]   # ... and the threads' own lanes
    [x.start() for x in th]
    [x.join() for x in th]
    ratios = []
    for attempt in range(3):                         # wall-clock ratios on a shared box: best of three
        job_best.clear()
        t0 = time.perf_counter()
        for j in jobs:
            work(j, iters)
        serial = time.perf_counter() - t0
        for *_, out in jobs:
            out[:] = 0
        job_best.clear()
        th = [threading.Thread(target=work, args=(j, iters)) for j in jobs]
        t0 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        parallel = time.perf_counter() - t0
        assert len(job_best) == nthreads and all(g == w for g, w in job_best)
        for s, p, want, out in jobs:
            assert np.array_equal(bits(out), bits(want[:, :32]))
        ratios.append(parallel / serial)
        print(f"8 threads: serial {serial * 1e3:.1f} ms, parallel {parallel * 1e3:.1f} ms, ratio {parallel / serial:.2f}")
        if ratios[-1] < 0.7:
            break
    # measured 0.49-0.62 (profiles/r04_host_pointer.json): at this size eight threads are bound by the link (floor 0.48).
    # A wall-clock ratio is not a correctness property (a shared or loaded box, a slower link): the suite asserts the
    # bits above and only that the threads were not SERIALISED outright; tools/host_pointer_bench.py records the ratio.
    if min(ratios) >= 0.8:
        import warnings
        warnings.warn(f"host-pointer calls of 8 threads overlapped less than expected: parallel / serial = {ratios}")
    assert min(ratios) < 1.5, ratios


def test_two_large_calls_at_once():
    """Two threads enter with pipeline-sized matrices at the same moment: they take turns on the pinned ring (the link is
    the bound) -- both bit-exact."""
    rng = np.random.default_rng(21)
    m = 20
    jobs = []
    for t in range(2):
        rows = 262_144 * 3 + 999 + t          # 100 MB of scores each: both take the tile pipeline
        s = striped(rng, rows * 32 - 5, 32, 5, m)
        p = random_pssm(rng, m, 5)
        want = co.aligned_empty((rows, 32), np.float32)
        co.avx2_score_rows(s, aligned(p), out=want, row_end=rows, threads=8)
        jobs.append((s, p, want))
    res = [None, None]

    def work(i):
        s, p, _ = jobs[i]
        res[i] = host_score(s, p, 5, 0, s.rows)[0]
    for _ in range(2):
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [x.start() for x in th]
        [x.join() for x in th]
        for i in range(2):
            assert np.array_equal(bits(res[i]), bits(jobs[i][2]))


def test_errors_leave_the_lane_usable():
    L = _ffi.lib()
    rng = np.random.default_rng(2)
    s = striped(rng, 40_000, 32, 5, 10)            # 9 wrap rows
    p20, p10 = random_pssm(rng, 20, 5), random_pssm(rng, 10, 5)
    out = np.zeros((s.rows, 32), np.float32)
    orow, omi = C.c_size_t(0), C.c_size_t(0)

    def call(p, a, b, seq_ptr=None, out_ptr=None, out_stride=32):
        return L.lm_hip_score_f32(s.data.ctypes.data if seq_ptr is None else seq_ptr, s.data.shape[0], 32, 32, s.wrap, s.length,
                                  p.ctypes.data, p.shape[0], p.shape[1], 5, a, b, out.ctypes.data if out_ptr is None else out_ptr,
                                  out_stride, C.byref(orow), C.byref(omi))
    assert call(p20, 0, s.rows) == _ffi.ERR_WRAP                       # avx2.rs:832-837
    assert b"wrapping rows" in L.lm_hip_last_error()
    assert call(p10, 0, s.rows + 1) == _ffi.ERR_BAD_ARGS               # rows past the sequence
    assert call(p10, 0, s.rows, out_stride=16) == _ffi.ERR_BAD_ARGS    # rows narrower than the columns
    assert call(p10, 0, s.rows, out_ptr=0) == _ffi.ERR_BAD_ARGS
    assert call(p10, 5, 5) == 0 and (orow.value, omi.value) == (0, 0)  # empty range: Ok, resized to nothing (pli/mod.rs:85-88)
    assert call(p10, 0, s.rows) == 0
    want, mi = co.score_rows(s, p10)
    assert (orow.value, omi.value) == (s.rows, mi) and np.array_equal(bits(out), bits(want[:, :32]))


def test_reductions_on_special_values_and_padded_rows():
    rng = np.random.default_rng(6)
    sc = rng.normal(0, 3, (5_000, 40)).astype(np.float32)
    sc[rng.random(sc.shape) < 0.01] = -np.inf
    sc[17, 3] = sc[4_000, 31] = sc[:, :32].max() + 1                    # tie: the LAST maximal cell wins (pli/mod.rs:146)
    sc[:, 32:] = np.inf                                                 # padding must never be looked at
    dense = np.ascontiguousarray(sc[:, :32])
    assert host_argmax(sc, 5_000, 40, 32)[0] == co.argmax(dense, 32) == (4_000, 31)
    assert np.array_equal(host_threshold(sc, 5_000, 40, 32, 6.0), co.threshold(dense, 32, 6.0))
    sc[0, 0] = np.nan                                                   # NaN first cell -> (0, 0) (pli/mod.rs:142-146)
    got = host_argmax(sc, 5_000, 40, 32)
    assert got[0] == (0, 0) and np.isnan(got[1])
    sc[0, 0], sc[9, 9] = 0.0, np.nan                                    # NaN elsewhere never wins
    assert host_argmax(sc, 5_000, 40, 32)[0] == (4_000, 31)
    assert len(host_threshold(sc, 5_000, 40, 32, float("-inf"))) == 5_000 * 32 - 1   # all but the NaN cell


def test_large_reduction_inputs_release_their_staging():
    """A 320 MB host matrix (above the 256 MB a lane keeps): uploaded, reduced, staging released -- and again."""
    rng = np.random.default_rng(12)
    rows = 2_500_000
    sc = rng.normal(0, 1, (rows, 32)).astype(np.float32)
    sc[1_999_999, 7] = 50.0
    sc[2_400_000, 1] = 50.0
    for _ in range(2):
        assert host_argmax(sc, rows, 32, 32)[0] == (2_400_000, 1)
        got = host_threshold(sc, rows, 32, 32, 49.0)
        assert [tuple(map(int, x)) for x in got] == [(1_999_999, 7), (2_400_000, 1)]


@pytest.mark.parametrize("out_stride", [32, 48])
def test_score_u8_scanner_blocks(pli, out_stride):
    """Scanner::next's inner call (scan.rs:174-178): u8 scores of 256-row blocks into the caller's host matrix -- the
    kernel writes pinned memory directly; also a last, shorter block and rows wider than the columns."""
    L = _ffi.lib()
    rng = np.random.default_rng(31)
    m = 15
    enc = rng.integers(0, 4, 100_000, dtype=np.uint8)
    s = co.stripe(enc, 32, 5)
    co.configure_wrap(s, m - 1)
    seq = pli.stripe(lm.EncodedSequence(enc))
    seq.configure_wrap(m - 1)
    w = np.zeros((m, 32), np.uint8)
    w[:, :5] = rng.integers(0, 14, (m, 5))
    want, mi = co.score_rows_u8(s, w)
    orow, omi = C.c_size_t(0), C.c_size_t(0)
    for a in list(range(0, s.rows, 256)):
        b = min(a + 256, s.rows)
        out = np.full((b - a, out_stride), 99, np.uint8)
        st = L.lm_hip_score_u8(pli._h, w.ctypes.data, m, 32, 5, seq._h, a, b, 0, out.ctypes.data, out_stride,
                               C.byref(orow), C.byref(omi))
        assert st == 0, _ffi.last_error()
        assert (orow.value, omi.value) == (b - a, mi)
        assert np.array_equal(out[:, :32], want[a:b, :32]) and np.all(out[:, 32:] == 99)


def host_score_u8(s, w, k, a, b, saturate, out_stride=None):
    L = _ffi.lib()
    ost = co.stride(s.cols, 1) if out_stride is None else out_stride
    out = np.full((max(b - a, 0), ost), 99, np.uint8)
    orow, omi = C.c_size_t(0), C.c_size_t(0)
    st = L.lm_hip_score_u8_host(s.data.ctypes.data, s.data.shape[0], s.stride, s.cols, s.wrap, s.length, w.ctypes.data, w.shape[0],
                                w.shape[1], k, a, b, int(saturate), out.ctypes.data, ost, C.byref(orow), C.byref(omi))
    assert st == 0, _ffi.last_error()
    return out, orow.value, omi.value


@pytest.mark.parametrize("length,cols,k,m,rows,out_stride", [
    (10_000, 32, 5, 15, (0, 256), None),          # one Scanner block (scan.rs:174-178): zero-copy both ways
    (10_000, 32, 5, 15, (256, 313), 48),          # the last, short block; the caller's rows wider than the columns
    (700_000, 32, 5, 20, None, None),             # copy path
    (300_000, 32, 21, 12, (5, 9000), None),       # protein
    (120_000, 20, 5, 9, None, None),              # 20 columns
    (400_000, 32, 5, 40, None, None),             # beyond 36 rows: slices added bytewise
])
def test_score_u8_host_matches_oracle(length, cols, k, m, rows, out_stride):
    """`Score<u8, A, C>` with a DiscreteMatrix on HOST matrices (`lm_hip_score_u8_host`, what a shim's `impl Score<u8, ..> for
    Pipeline<A, Hip>` binds): both overflow flavours -- Generic's wrapping `+=` against the C oracle, the SIMD back-ends'
    saturating adds (avx2.rs:336) against the numpy restatement -- with weights large enough to overflow."""
    from oracle import np_oracle as no
    rng = np.random.default_rng(length + m + cols)
    s = striped(rng, length, cols, k, m)
    w = np.zeros((m, 32), np.uint8)
    w[:, :k] = rng.integers(0, 40, (m, k))
    a, b = (0, s.rows) if rows is None else rows
    want, mi = co.score_rows_u8(s, w, a, b)
    got, orow, omi = host_score_u8(s, w, k, a, b, saturate=False, out_stride=out_stride)
    assert (orow, omi) == (b - a, mi)
    assert np.array_equal(got[:, :cols], want[:, :cols])
    assert np.all(got[:, cols:] == 99), "alignment padding of the caller's rows was written"
    if b - a <= 30_000:
        sat = no.score_rows_u8_saturating(s.data, cols, length, w[:, :k], a, b)
        got_s, _, _ = host_score_u8(s, w, k, a, b, saturate=True, out_stride=out_stride)
        assert np.array_equal(got_s[:, :cols], sat) and sat.max() == 255


def test_score_u8_host_through_the_tile_pipeline():
    """102 Mbp: 102 MB of u8 scores through the tile pipeline (twelve tiles and a ragged one); wrapping adds against the C oracle."""
    rng = np.random.default_rng(64)
    m, rows_total = 15, 3_200_000 + 77
    s = striped(rng, rows_total * 32 - 3, 32, 5, m)
    w = np.zeros((m, 32), np.uint8)
    w[:, :5] = rng.integers(0, 17, (m, 5))
    a, b = 11, rows_total - 5
    want, mi = co.score_rows_u8(s, w, a, b)
    got, orow, omi = host_score_u8(s, w, 5, a, b, saturate=False)
    assert (orow, omi) == (b - a, mi) and np.array_equal(got[:, :32], want[:, :32])


def test_host_pointer_fuzz():
    """160 random shapes through the host-pointer entry points -- column counts 1 ... 40, motif lengths 1 ... 44, both alphabets,
    row ranges, padded caller rows, -inf / tied weights -- f32 scores, argmax, max and threshold against the oracle, bit for bit."""
    rng = np.random.default_rng(0xF0221)
    for case in range(160):
        k = 21 if rng.random() < 0.25 else 5
        cols = int(rng.choice([32, 32, 32, 1, 2, 7, 16, 20, 33, 40]))
        m = int(rng.integers(1, 45))
        length = int(rng.integers(m, 60_000)) if rng.random() < 0.9 else int(rng.integers(1, m + 1))
        s = striped(rng, length, cols, k, m)
        p = random_pssm(rng, m, k, "ties" if rng.random() < 0.3 else "normal")
        if rng.random() < 0.2:
            p[rng.integers(0, m), rng.integers(0, k - 1)] = -np.inf
        a = int(rng.integers(0, s.rows)) if rng.random() < 0.4 else 0
        b = int(rng.integers(a, s.rows + 1)) if rng.random() < 0.4 else s.rows
        ost = co.stride(cols, 4) + (8 if rng.random() < 0.3 else 0)
        want, mi = co.score_rows(s, p, a, b)
        got, orow, omi = host_score(s, p, k, a, b, out_stride=ost)
        tag = (case, k, cols, m, length, a, b, ost)
        assert (orow, omi) == (want.shape[0], mi), tag
        if want.shape[0] == 0:
            continue
        assert np.array_equal(bits(got[:, :cols]), bits(want[:, :cols])), tag
        assert np.all(got[:, cols:] == 777.0), tag
        am = host_argmax(got, b - a, ost, cols)
        wam = co.argmax(want, cols)
        assert (am[0] if am else None) == wam, tag
        if wam is not None:
            assert bits(am[1]) == bits(co.max_(want, cols)), tag
        finite = want[:, :cols][np.isfinite(want[:, :cols])]
        t = float(np.quantile(finite, 0.98)) if finite.size else 0.0
        assert np.array_equal(host_threshold(got, b - a, ost, cols, t), co.threshold(want, cols, t)), tag


def test_host_trim_gives_the_ring_back_and_the_path_still_works():
    """`lm_hip_host_trim`: after a pipeline-sized call the process holds 128 MB of page-locked ring + device tiles; trimming
    returns them (device memory visibly), and the next calls -- small and large -- set up again and stay bit-exact."""
    L = _ffi.lib()

    def device_free_bytes():
        # hipMemGetInfo of the HIP runtime the library itself is linked to (torch must not be imported after the library
        # in one process on this image -- README -- so the runtime is asked directly)
        path = next(line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line)
        free, total = C.c_size_t(0), C.c_size_t(0)
        assert C.CDLL(path).hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value
    rng = np.random.default_rng(77)
    m = 12
    rows = 262_144 * 3 + 100
    s = striped(rng, rows * 32 - 9, 32, 5, m)
    p = random_pssm(rng, m, 5)
    want = co.aligned_empty((rows, 32), np.float32)
    co.avx2_score_rows(s, aligned(p), out=want, row_end=rows, threads=8)
    assert np.array_equal(bits(host_score(s, p, 5, 0, rows)[0]), bits(want))
    free_before = device_free_bytes()
    assert L.lm_hip_host_trim() == 0, _ffi.last_error()
    free_after = device_free_bytes()
    assert free_after >= free_before + (32 << 20), (free_before, free_after)     # four 8.4 MB score tiles + three symbol tiles
    assert L.lm_hip_host_trim() == 0                       # idempotent
    small = striped(rng, 50_000, 32, 5, m)
    w_small, _ = co.score_rows(small, p)
    assert np.array_equal(bits(host_score(small, p, 5, 0, small.rows)[0][:, :32]), bits(w_small[:, :32]))
    assert np.array_equal(bits(host_score(s, p, 5, 0, rows)[0]), bits(want))


@pytest.mark.parametrize("length,cols,padded", [(464_165, 32, False), (2_000, 32, False), (90_001, 32, True), (50_000, 16, False)])
def test_kept_scores_serve_the_next_reduction_and_nothing_else(length, cols, padded):
    """`lm_hip_host_reuse_scores(1)` (the reference's bench loop, dna.rs:81-116: `score_into` then `argmax` on the same
    matrix): argmax / threshold of the matrix the thread's last score call wrote reduce the copy left on the device --
    same results as the upload path; anything else (another matrix, another shape, a sampled cell rewritten, a score call
    in between, the switch off) takes the upload, and its results follow the HOST matrix."""
    L = _ffi.lib()
    rng = np.random.default_rng(length + cols)
    k, m = 5, 15
    s = striped(rng, length, cols, k, m)
    p = aligned(random_pssm(rng, m, k, "ties"))
    rows = s.rows
    want, _ = co.score_rows(s, p)

    def count():
        n = C.c_size_t(0)
        assert L.lm_hip_host_reuse_count(C.byref(n)) == 0
        return n.value

    ost = co.stride(cols, 4) + (8 if padded else 0)
    assert L.lm_hip_host_reuse_scores(1) == 0
    try:
        out, orow, _ = host_score(s, p, k, 0, rows, out_stride=ost)
        assert orow == rows and np.array_equal(bits(out[:, :cols]), bits(want[:, :cols]))
        c0 = count()
        assert host_argmax(out, rows, ost, cols)[0] == co.argmax(want, cols)
        assert count() == c0 + 1                       # the kept copy
        t = float(np.sort(want[:, :cols][np.isfinite(want[:, :cols])])[-20])
        assert np.array_equal(host_threshold(out, rows, ost, cols, t), np.asarray(co.threshold(want, cols, t), np.uintp).reshape(-1, 2))
        assert count() == c0 + 2                       # threshold as well, and it does not consume the copy
        # another matrix of the same shape: uploaded
        other = out.copy()
        other[rows // 2, 0] = np.float32(1e9)
        assert host_argmax(other, rows, ost, cols) == ((rows // 2, 0), np.float32(1e9))
        assert count() == c0 + 2
        # a sampled cell (the first) rewritten in place: the digest differs, the upload path answers from the host matrix
        out[0, 0] = np.float32(2e9)
        assert host_argmax(out, rows, ost, cols) == ((0, 0), np.float32(2e9))
        assert count() == c0 + 2
        # fewer rows of the kept matrix: another shape
        out2, _, _ = host_score(s, p, k, 0, rows, out_stride=ost)
        if rows > 4:
            assert host_argmax(out2, rows - 3, ost, cols)[0] == co.argmax(want[: rows - 3], cols)
            assert count() == c0 + 2
        # a score call in between drops what was kept (its own result is kept instead)
        out3, _, _ = host_score(s, p, k, 0, rows, out_stride=ost)
        assert host_argmax(out2, rows, ost, cols)[0] == co.argmax(want, cols)
        assert count() == c0 + 2
        assert host_argmax(out3, rows, ost, cols)[0] == co.argmax(want, cols)
        assert count() == c0 + 3
        # switched off: nothing is kept, nothing reused
        assert L.lm_hip_host_reuse_scores(0) == 0
        out4, _, _ = host_score(s, p, k, 0, rows, out_stride=ost)
        assert host_argmax(out4, rows, ost, cols)[0] == co.argmax(want, cols)
        assert count() == c0 + 3
    finally:
        L.lm_hip_host_reuse_scores(0)


def test_kept_scores_with_host_threads():
    """`lm_hip_host_reuse_scores(1)` with six host threads in their `score_into` + `argmax` + `threshold` loops at once: what is
    kept is per lane (per thread), so every thread's reductions answer for ITS matrix, bit for bit, and take the kept copy."""
    L = _ffi.lib()
    rng = np.random.default_rng(81)
    nthreads, m, iters = 6, 15, 25
    jobs = []
    for t in range(nthreads):
        s = striped(rng, 120_000 + 37_000 * t, 32, 5, m)
        p = aligned(random_pssm(rng, m, 5, "ties" if t % 2 else "normal"))
        want, _ = co.score_rows(s, p)
        cut = float(np.sort(want[:, :32][np.isfinite(want[:, :32])])[-10])
        jobs.append((s, p, want, cut, np.zeros((s.rows, 32), np.float32)))
    errors, reused = [], []

    def work(job):
        try:
            s, p, want, cut, out = job
            orow, omi, n0, n1 = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
            assert L.lm_hip_host_reuse_count(C.byref(n0)) == 0
            want_best = co.argmax(want, 32)
            want_hits = np.asarray(co.threshold(want, 32, cut), np.uintp).reshape(-1, 2)
            for _ in range(iters):
                out[:] = 0
                st = L.lm_hip_score_f32(s.data.ctypes.data, s.data.shape[0], 32, 32, s.wrap, s.length, p.ctypes.data, m, p.shape[1],
                                        5, 0, s.rows, out.ctypes.data, 32, C.byref(orow), C.byref(omi))
                assert st == 0 and np.array_equal(bits(out), bits(want[:, :32]))
                assert host_argmax(out, s.rows, 32, 32)[0] == want_best
                assert np.array_equal(host_threshold(out, s.rows, 32, 32, cut), want_hits)
            assert L.lm_hip_host_reuse_count(C.byref(n1)) == 0
            reused.append(n1.value - n0.value)
        except Exception as exc:   # noqa: BLE001  (reported from the main thread)
            errors.append(repr(exc))

    assert L.lm_hip_host_reuse_scores(1) == 0
    try:
        th = [threading.Thread(target=work, args=(j,)) for j in jobs]
        [x.start() for x in th]
        [x.join() for x in th]
    finally:
        L.lm_hip_host_reuse_scores(0)
    assert not errors, errors
    # (a lane taken over from an exited thread may owe an lm_hip_host_trim its staging: its first score call then keeps nothing)
    assert all(2 * (iters - 1) <= r <= 2 * iters for r in reused) and len(reused) == nthreads, reused
