"""The literal drop-in path, timed: the host-pointer entry points the Rust shim binds on
pageable host buffers -- `lm_hip_score_f32`, `lm_hip_argmax_f32`, `lm_hip_threshold_f32`, `lm_hip_score_u8`.

  c1        lightmotif-bench/dna.rs:104-107 as is: score + argmax per iteration on host matrices (464 165 bp, M = 15),
            next to the 1-thread AVX2 port of the same loop
  block     Scanner's 256-row block (scan.rs:174-178): lm_hip_score_u8 into a host StripedScores<u8>
  big       1 Gbp x M = 20 through lm_hip_score_f32: ms, Gpos/s, link GB/s
  threads   8 host threads, one 50 Mbp call each, against the same calls one after the other

    python tools/host_pointer_bench.py [--json out.json] [--big 1000000000]
"""
import argparse
import ctypes as C
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lightmotif_amd as lm  # noqa: E402
from lightmotif_amd import _ffi  # noqa: E402

L = _ffi.lib()
COLS = 32


def striped(enc: np.ndarray, m: int, k: int = 5):
    """host StripedSequence matrix (rows + m - 1, 32) of encoded symbols, wrap rows configured (seq.rs:369-381)"""
    length = len(enc)
    rows = -(-length // COLS)
    pad = np.full(rows * COLS, k - 1, np.uint8)
    pad[:length] = enc
    mat = np.empty((rows + m - 1, COLS), np.uint8)
    mat[:rows] = pad.reshape(COLS, rows).T
    mat[rows:, :COLS - 1] = mat[:m - 1, 1:]
    mat[rows:, COLS - 1] = k - 1
    return mat, rows, length


def score_f32(mat, rows, length, pssm, out, row_begin=0, row_end=None):
    orow, mi = C.c_size_t(0), C.c_size_t(0)
    m = pssm.shape[0]
    st = L.lm_hip_score_f32(mat.ctypes.data, mat.shape[0], COLS, COLS, mat.shape[0] - rows, length, pssm.ctypes.data, m,
                            pssm.shape[1], 5, row_begin, rows if row_end is None else row_end, out.ctypes.data, out.shape[1],
                            C.byref(orow), C.byref(mi))
    assert st == 0, _ffi.last_error()
    return orow.value, mi.value


def argmax_f32(scores, rows):
    found, best, value = C.c_int(0), _ffi.Coords(), C.c_float(0)
    st = L.lm_hip_argmax_f32(scores.ctypes.data, rows, scores.shape[1], COLS, C.byref(found), C.byref(best), C.byref(value))
    assert st == 0, _ffi.last_error()
    return (best.row, best.col) if found.value else None


def loop_us(fn, reps, warm):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6, float(np.min(ts)) * 1e6


def bench_c1():
    from oracle import c_oracle as co
    length = 464_165
    rng = np.random.default_rng(0xEC011)
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    enc[391_677:391_677 + 15] = lm.EncodedSequence("GTTGACCTTATCAAC").data
    pssm = lm.create(["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"]).counts.normalize(0.1).log_odds().data
    m = pssm.shape[0]
    mat, rows, _ = striped(enc, m)
    out = co.aligned_empty((rows, COLS), np.float32)
    best = [None]

    def it():
        score_f32(mat, rows, length, pssm, out)
        best[0] = argmax_f32(out, rows)
    med, mn = loop_us(it, 300, 30)
    s_med, _ = loop_us(lambda: score_f32(mat, rows, length, pssm, out), 300, 30)
    a_med, _ = loop_us(lambda: argmax_f32(out, rows), 300, 30)
    pos = best[0][1] * rows + best[0][0]
    # the same loop with lm_hip_host_reuse_scores(1): argmax reduces the device copy the score call left (opt-in: the
    # caller promises not to write to `out` in between, as the reference's loop does not)
    assert L.lm_hip_host_reuse_scores(1) == 0
    try:
        before = C.c_size_t(0)
        L.lm_hip_host_reuse_count(C.byref(before))
        r_med, r_min = loop_us(it, 300, 30)
        after = C.c_size_t(0)
        L.lm_hip_host_reuse_count(C.byref(after))
        reuse_pos = best[0][1] * rows + best[0][0]
    finally:
        L.lm_hip_host_reuse_scores(0)
    # the AVX2 port, one thread, same loop (avx2.rs:104-199 + 351-426)
    data = co.aligned_empty(mat.shape, np.uint8)
    data[:] = mat
    s = co.Striped(data, length, m - 1, COLS, 5)
    p = co.aligned_empty(pssm.shape, np.float32)
    p[:] = pssm
    cout = co.aligned_empty((rows, COLS), np.float32)

    def cpu_it():
        co.avx2_score_rows(s, p, out=cout, row_end=rows, threads=1)
        co.avx2_argmax(cout, length + 1 - m)
    c_med, c_min = loop_us(cpu_it, 100, 10)
    same = bool(np.array_equal(cout.view(np.uint32), out.view(np.uint32)))
    return {"host_pointer_us_per_iter": round(med, 1), "host_pointer_us_min": round(mn, 1),
            "host_pointer_reuse_us_per_iter": round(r_med, 1), "host_pointer_reuse_us_min": round(r_min, 1),
            "reuse_calls_that_took_the_kept_copy": int(after.value - before.value), "reuse_same_position": bool(reuse_pos == pos),
            "score_f32_us": round(s_med, 1),
            "argmax_f32_us": round(a_med, 1), "host_pointer_best_position": int(pos), "avx2_port_1_thread_us_per_iter": round(c_med, 1),
            "avx2_port_1_thread_us_min": round(c_min, 1), "scores_match_avx2_port_bitwise": same,
            "x_avx2_port": round(c_med / med, 2)}


def bench_readme_10kb():
    """The reference's second published benchmark (README.md:111-118): the highest-scoring
    position of a 10 kb sequence, MX000001 (M = 15) -- 12.797 us with AVX2 on an i7-10710U.  Through host pointers
    (lm_hip_score_f32 + lm_hip_argmax_f32) and with the one-thread AVX2 port on this box (Avx2::score_f32_rows_into +
    Avx2::argmax_f32, called as the shim would: no thread start, no conversions).  A `Dispatch::Hip { cpu }` variant keeps
    this size on its CPU tier; `crossover_positions` = the shipped policy at this motif length."""
    from oracle import c_oracle as co
    length = 10_000
    enc = np.random.default_rng(0x10CB).integers(0, 4, length, dtype=np.uint8)
    pssm = lm.create(["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"]).counts.normalize(0.1).log_odds().data
    m = pssm.shape[0]
    mat, rows, _ = striped(enc, m)
    out = co.aligned_empty((rows, COLS), np.float32)
    best = [None]

    def it():
        score_f32(mat, rows, length, pssm, out)
        best[0] = argmax_f32(out, rows)
    med, mn = loop_us(it, 500, 50)
    data = co.aligned_empty(mat.shape, np.uint8)
    data[:] = mat
    p = co.aligned_empty(pssm.shape, np.float32)
    p[:] = pssm
    cout = co.aligned_empty((rows, COLS), np.float32)
    A = co.avx2()
    u8p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    seq_p, p_p, cout_p = data.ctypes.data_as(u8p), p.ctypes.data_as(f32p), cout.ctypes.data_as(f32p)
    r_, c_ = C.c_size_t(0), C.c_size_t(0)

    def cpu_it():
        A.lma_score_rows_f32(seq_p, COLS, m - 1, length, p_p, m, p.shape[1], 5, 0, rows, cout_p, COLS)
        A.lma_argmax_f32(cout_p, rows, COLS, length + 1 - m, C.byref(r_), C.byref(c_))
    c_med, c_min = loop_us(cpu_it, 500, 50)
    same = bool(np.array_equal(cout.view(np.uint32), out.view(np.uint32)))
    return {"workload": "README.md:111-118: score + argmax of a 10 kb sequence, MX000001 (M = 15), per iteration",
            "host_pointer_us": round(med, 2), "host_pointer_us_min": round(mn, 2), "avx2_port_us": round(c_med, 2),
            "avx2_port_us_min": round(c_min, 2), "published_avx2_us": 12.797, "scores_match_avx2_port_bitwise": same,
            "dispatch_hip_route": "cpu tier (10 000 cells < every crossover)"}


def crossover_positions(m: int = 15, k: int = 5):
    """lm_hip_host_crossover per `dispatch.rs` site: cells (= positions) from which `Dispatch::Hip { cpu }` sends a call on
    host matrices to the GPU; None = the site stays on the CPU tier at every size."""
    names = ["encode", "score_f32", "score_u8", "stripe", "maximum_f32", "maximum_u8", "threshold_f32", "threshold_u8", "scan"]
    out = {}
    for op, name in enumerate(names):
        cells = C.c_size_t(0)
        st = L.lm_hip_host_crossover(op, m, k, C.byref(cells))
        assert st == 0, _ffi.last_error()
        out[name] = None if cells.value == C.c_size_t(-1).value else int(cells.value)
    return {"motif_len": m, "cells": out,
            "source": "cost model of csrc/hostptr.hip (kCost*), constants measured by tools/crossover.py + tests/cpp/test_dispatch --bench "
                      "on MI355X + EPYC 9575F, one CPU thread (profiles/r05_crossover.json)"}


def bench_block():
    """Scanner::next's inner step (scan.rs:174-178): pli.score_rows_into(&dm, seq, row..row+256, &mut dscores) -- u8 scores of
    one 256-row block land in the caller's host matrix"""
    length, m = 10_000_000, 15
    rng = np.random.default_rng(3)
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    pli = lm.Pipeline.hip(0)
    seq = pli.stripe(lm.EncodedSequence(enc))
    pssm = lm.create(["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"]).counts.normalize(0.1).log_odds()
    seq.configure(pssm)
    w = rng.integers(0, 12, (m, 32), dtype=np.uint8)     # a DenseMatrix<u8, K> with its 32-byte rows
    out = np.zeros((256, COLS), np.uint8)
    orow, mi = C.c_size_t(0), C.c_size_t(0)
    row = [0]

    def it():
        st = L.lm_hip_score_u8(pli._h, w.ctypes.data, m, 32, 5, seq._h, row[0], row[0] + 256, 1, out.ctypes.data, COLS,
                               C.byref(orow), C.byref(mi))
        assert st == 0, _ffi.last_error()
        row[0] = (row[0] + 256) % (seq.rows - 256)
    med, mn = loop_us(it, 1000, 100)
    # ... and with the sequence on the HOST as well (lm_hip_score_u8_host: 8.6 KB up, 8 KB down per block, zero-copy)
    host_mat = seq.matrix()
    hrow = [0]

    def host_it():
        st = L.lm_hip_score_u8_host(host_mat.ctypes.data, host_mat.shape[0], 32, COLS, m - 1, length, w.ctypes.data, m, 32, 5,
                                    hrow[0], hrow[0] + 256, 1, out.ctypes.data, COLS, C.byref(orow), C.byref(mi))
        assert st == 0, _ffi.last_error()
        hrow[0] = (hrow[0] + 256) % (seq.rows - 256)
    h_med, h_min = loop_us(host_it, 1000, 100)
    # the same block on one CPU thread: the oracle's Generic u8 restatement (pli/mod.rs:72-106 with u8 weights; the
    # reference's AVX2 u8 kernel, avx2.rs:294-347, has no port here -- this is the slower of the two CPU forms)
    from oracle import c_oracle as co
    ref = co.stripe(enc, 32, 5)
    co.configure_wrap(ref, m - 1)
    crow = [0]

    def cpu_it():
        co.score_rows_u8(ref, w, crow[0], crow[0] + 256)
        crow[0] = (crow[0] + 256) % (ref.rows - 256)
    c_med, _ = loop_us(cpu_it, 300, 30)
    return {"scanner_block_us": round(med, 2), "scanner_block_us_min": round(mn, 2), "scanner_block_rows": 256,
            "scanner_block_host_sequence_us": round(h_med, 2), "scanner_block_generic_cpu_1_thread_us": round(c_med, 2)}


def bench_big(length: int, m: int = 20, mat=None, pssm=None, reps: int = 4, check=None):
    """1 B per position up, 4 B per position down: `lm_hip_score_f32` on a whole striped matrix in pageable memory.
    `mat` / `pssm`: the caller's own data (bench.py: the resident shard copied to the host); `check(out) -> bool`: its
    own verification of the result (default: head and tail against the oracle)."""
    rng = np.random.default_rng(0)
    rows = -(-length // COLS)
    if mat is None:
        mat = rng.integers(0, 4, (rows + m - 1, COLS), dtype=np.uint8)
        mat[rows:, :31] = mat[:m - 1, 1:]
        mat[rows:, 31] = 4
    if pssm is None:
        sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]
        pssm = lm.create(sites).counts.normalize(0.1).log_odds().data
    out = np.zeros((rows, COLS), np.float32)
    score_f32(mat, rows, length, pssm, out)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        score_f32(mat, rows, length, pssm, out)
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    if check is not None:
        ok = bool(check(out))
    else:
        from oracle import c_oracle as co
        chk = min(rows, 1 << 16)
        s = co.Striped(np.ascontiguousarray(mat[:chk + m - 1]), length, m - 1, COLS, 5)
        want, _ = co.score_rows(s, pssm, 0, chk)
        ok = bool(np.array_equal(want.view(np.uint32), out[:chk].view(np.uint32)))
        tail = co.Striped(np.ascontiguousarray(mat[rows - chk:]), length, m - 1, COLS, 5)
        want_t, _ = co.score_rows(tail, pssm, 0, chk)
        ok = ok and bool(np.array_equal(want_t.view(np.uint32), out[rows - chk:].view(np.uint32)))
    return {"length": length, "host_pointer_ms": round(t * 1e3, 2), "host_pointer_median_ms": round(float(np.median(ts)) * 1e3, 2),
            "gpos": round(rows * COLS / t / 1e9, 2), "link_gbs": round(5 * rows * COLS / t / 1e9, 1),
            "d2h_gbs": round(4 * rows * COLS / t / 1e9, 1), "verified": ok, "all_ms": [round(x * 1e3, 2) for x in ts]}


def bench_threads(nthreads: int = 8, length: int = 50_000_000, m: int = 20):
    rng = np.random.default_rng(9)
    rows = -(-length // COLS)
    sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]
    pssm = lm.create(sites).counts.normalize(0.1).log_odds().data
    jobs = []
    for t in range(nthreads):
        mat = rng.integers(0, 4, (rows + m - 1, COLS), dtype=np.uint8)
        mat[rows:, :31] = mat[:m - 1, 1:]
        mat[rows:, 31] = 4
        jobs.append((mat, np.zeros((rows, COLS), np.float32)))
    for mat, out in jobs:
        score_f32(mat, rows, length, pssm, out)
    ref = [out.copy() for _, out in jobs[:2]]
    t0 = time.perf_counter()
    for mat, out in jobs:
        score_f32(mat, rows, length, pssm, out)
    serial = time.perf_counter() - t0
    for _, out in jobs:
        out[:] = 0
    th = [threading.Thread(target=score_f32, args=(mat, rows, length, pssm, out)) for mat, out in jobs]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    par = time.perf_counter() - t0
    same = all(np.array_equal(r.view(np.uint32), jobs[i][1].view(np.uint32)) for i, r in enumerate(ref))
    return {"threads": nthreads, "length_each": length, "serial_ms": round(serial * 1e3, 1), "parallel_ms": round(par * 1e3, 1),
            "parallel_over_serial": round(par / serial, 3), "bit_identical": bool(same)}


def bench_threads_small(nthreads: int = 8, iters: int = 100):
    """The CLI's `-j` workers (main.rs:270) / GIL-released Python threads (lib.rs:865) at the reference's own bench size:
    every thread loops score + argmax on its own host matrices; each gets a lane (context, stream, staging) of its own."""
    length, m = 464_165, 15
    rng = np.random.default_rng(11)
    pssm = lm.create(["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"]).counts.normalize(0.1).log_odds().data
    jobs = []
    for _ in range(nthreads):
        mat, rows, _ = striped(rng.integers(0, 4, length, dtype=np.uint8), m)
        jobs.append((mat, rows, np.zeros((rows, COLS), np.float32)))

    def work(job, n):
        mat, rows, out = job
        for _ in range(n):
            score_f32(mat, rows, length, pssm, out)
            argmax_f32(out, rows)
    for j in jobs:
        work(j, 3)
    t0 = time.perf_counter()
    for j in jobs:
        work(j, iters)
    serial = time.perf_counter() - t0

    def run(n):
        th = [threading.Thread(target=work, args=(j, n)) for j in jobs]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        return time.perf_counter() - t0
    run(3)
    par = run(iters)
    return {"threads": nthreads, "iterations_each": iters, "serial_ms": round(serial * 1e3, 1), "parallel_ms": round(par * 1e3, 1),
            "parallel_over_serial": round(par / serial, 3),
            "us_per_iteration_aggregate": round(par / (nthreads * iters) * 1e6, 1)}


def bench_sizes(sizes=(1_000_000, 4_641_652, 10_000_000, 20_000_000, 50_000_000, 100_000_000, 200_000_000), m: int = 20):
    """lm_hip_score_f32 by sequence length (4 641 652 = E. coli K12, the reference's README benchmark): where the copy path
    ends and the tile pipeline begins (96 MB of scores = 25 Mbp)"""
    rng = np.random.default_rng(5)
    sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]
    pssm = lm.create(sites).counts.normalize(0.1).log_odds().data
    out = {}
    for length in sizes:
        rows = -(-length // COLS)
        mat = rng.integers(0, 4, (rows + m - 1, COLS), dtype=np.uint8)
        mat[rows:, :31] = mat[:m - 1, 1:]
        mat[rows:, 31] = 4
        res = np.zeros((rows, COLS), np.float32)
        med, mn = loop_us(lambda: score_f32(mat, rows, length, pssm, res), 12 if length > 20_000_000 else 40, 3)
        out[str(length)] = {"ms": round(med / 1e3, 3), "ms_min": round(mn / 1e3, 3), "gpos": round(rows * COLS / med / 1e3, 2)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--big", type=int, default=1_000_000_000)
    ap.add_argument("--skip", default="")
    a = ap.parse_args()
    res = {}
    if "c1" not in a.skip:
        res["c1"] = bench_c1()
        print(json.dumps({"c1": res["c1"]}), flush=True)
    if "block" not in a.skip:
        res["scanner_block"] = bench_block()
        print(json.dumps({"scanner_block": res["scanner_block"]}), flush=True)
    if "threads" not in a.skip:
        res["threads_small_calls"] = bench_threads_small()
        print(json.dumps({"threads_small_calls": res["threads_small_calls"]}), flush=True)
        res["threads"] = bench_threads()
        print(json.dumps({"threads": res["threads"]}), flush=True)
    if "sizes" not in a.skip:
        res["by_length"] = bench_sizes()
        print(json.dumps({"by_length": res["by_length"]}), flush=True)
    if "big" not in a.skip:
        res["end_to_end"] = bench_big(a.big)
        print(json.dumps({"end_to_end": res["end_to_end"]}), flush=True)
    if a.json:
        Path(a.json).write_text(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
