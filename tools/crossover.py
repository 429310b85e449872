#!/usr/bin/env python3
"""Where `Dispatch::Hip` should hand a call on HOST matrices to the GPU: each host-pointer entry point against ONE thread of
the reference's best CPU tier (the AVX2 port of oracle/lm_avx2.c; the reference core crate is single-threaded per call) at
1 k ... 4.6 M positions, MX000001 (M = 15) -- incl. the reference's second published benchmark, `score` + `argmax` of a
10 kb sequence (README.md:111-118: AVX2 12.797 us on an i7-10710U).

    python tools/crossover.py [--json out.json]

Per operation the crossover is the smallest measured length from which the GPU call is faster at EVERY larger length
(log-interpolated between the two neighbouring lengths); `None` = the CPU tier wins everywhere measured.
"""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402
from lightmotif_amd import _ffi  # noqa: E402
from oracle import c_oracle as co  # noqa: E402  (the comparator: a tool, not the product)

L = _ffi.lib()
COLS = 32
LENGTHS = [1_000, 10_000, 50_000, 100_000, 464_165, 1_000_000, 4_641_652, 20_000_000]


def med_us(fn, reps, warm):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6


def crossover(lengths, gpu, cpu):
    """smallest length from which gpu < cpu at every larger measured length, log-interpolated"""
    n = len(lengths)
    first = n
    for i in range(n - 1, -1, -1):
        if gpu[i] < cpu[i]:
            first = i
        else:
            break
    if first == n:
        return None
    if first == 0:
        return lengths[0]
    # ratio r = gpu / cpu crosses 1 between lengths[first - 1] and lengths[first]
    r0, r1 = np.log(gpu[first - 1] / cpu[first - 1]), np.log(gpu[first] / cpu[first])
    x0, x1 = np.log(lengths[first - 1]), np.log(lengths[first])
    return int(round(float(np.exp(x0 + (x1 - x0) * r0 / (r0 - r1)))))


def measure(length, m=15):
    rng = np.random.default_rng(length)
    enc = rng.integers(0, 4, length, dtype=np.uint8)
    sites = ["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"]
    if m != 15:  # other motif lengths: the CPU's Score cost is proportional to M, the GPU call's is not
        srng = np.random.default_rng(m)
        sites = ["".join("ACTG"[i] for i in srng.integers(0, 4, m)) for _ in range(8)]
    pssm = lm.create(sites).counts.normalize(0.1).log_odds()
    dm = pssm.to_discrete()
    s = co.stripe(enc, COLS, 5)
    co.configure_wrap(s, m - 1)
    rows = s.rows
    mat = s.data
    p = co.aligned_empty(pssm.data.shape, np.float32)
    p[:] = pssm.data
    w = co.aligned_empty((m, 32), np.uint8)
    w[:] = 0
    w[:, :dm.data.shape[1]] = dm.data
    out = co.aligned_empty((rows, COLS), np.float32)
    cout = co.aligned_empty((rows, COLS), np.float32)
    out8 = co.aligned_empty((rows, COLS), np.uint8)
    cout8 = co.aligned_empty((rows, COLS), np.uint8)
    orow, mi = C.c_size_t(0), C.c_size_t(0)
    found, best, value = C.c_int(0), _ffi.Coords(), C.c_float(0)
    total_rows = mat.shape[0]
    reps, warm = (400, 40) if length <= 1_000_000 else (40, 5) if length <= 5_000_000 else (12, 3)

    def g_score():
        st = L.lm_hip_score_f32(mat.ctypes.data, total_rows, s.stride, COLS, s.wrap, length, p.ctypes.data, m, p.shape[1], 5, 0, rows,
                                out.ctypes.data, COLS, C.byref(orow), C.byref(mi))
        assert st == 0, _ffi.last_error()

    A = co.avx2()
    Lo = co.lib()
    u8p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    seq_p = mat.ctypes.data_as(u8p)
    p_p, cout_p = p.ctypes.data_as(f32p), cout.ctypes.data_as(f32p)

    def c_score():  # Avx2::score_f32_rows_into, one thread, called as the shim would (no thread start, no conversions)
        A.lma_score_rows_f32(seq_p, s.stride, s.wrap, length, p_p, m, p.shape[1], 5, 0, rows, cout_p, COLS)

    def g_argmax():
        st = L.lm_hip_argmax_f32(out.ctypes.data, rows, COLS, COLS, C.byref(found), C.byref(best), C.byref(value))
        assert st == 0, _ffi.last_error()

    r_, c_ = C.c_size_t(0), C.c_size_t(0)

    def c_argmax():  # the GENERIC body (pli/mod.rs:135-155): the rule a `Hip` variant must keep at every size
        Lo.lmo_argmax_f32(cout_p, rows, COLS, COLS, C.byref(r_), C.byref(c_))

    def c_argmax_avx2():  # Avx2::argmax_f32 (different tie rule): for reference
        A.lma_argmax_f32(cout_p, rows, COLS, length + 1 - m, C.byref(r_), C.byref(c_))

    g_score()
    c_score()
    assert np.array_equal(out.view(np.uint32), cout.view(np.uint32))
    flat = cout[:, :COLS].ravel()
    t = float(np.partition(flat, flat.size - max(1, length // 100_000))[flat.size - max(1, length // 100_000)])
    n = C.c_size_t(0)

    def g_thr():
        ptr = C.POINTER(_ffi.Coords)()
        st = L.lm_hip_threshold_f32(out.ctypes.data, rows, COLS, COLS, C.c_float(t), C.byref(ptr), C.byref(n))
        assert st == 0, _ffi.last_error()
        L.lm_hip_free(ptr)

    rc_buf = np.empty(2 * (rows * COLS // 50 + 1024), np.uint64)
    rc_p = rc_buf.ctypes.data_as(C.POINTER(C.c_size_t))

    def c_thr():  # the default body of Threshold (pli/mod.rs:210-221): the reference has no SIMD form
        Lo.lmo_threshold_f32(cout_p, rows, COLS, COLS, C.c_float(t), rc_p, rc_buf.size // 2)

    def g_u8():
        st = L.lm_hip_score_u8_host(mat.ctypes.data, total_rows, s.stride, COLS, s.wrap, length, w.ctypes.data, m, 32, 5, 0, rows, 1,
                                    out8.ctypes.data, COLS, C.byref(orow), C.byref(mi))
        assert st == 0, _ffi.last_error()

    w_p, cout8_p = w.ctypes.data_as(u8p), cout8.ctypes.data_as(u8p)

    def c_u8():
        A.lma_score_rows_u8(seq_p, s.stride, s.wrap, length, w_p, m, 32, 0, rows, cout8_p, COLS)

    g_u8()
    c_u8()
    assert np.array_equal(out8, cout8)
    r = {"length": length, "rows": rows, "m": m}
    for name, g, c in (("score_f32", g_score, c_score), ("argmax_f32", g_argmax, c_argmax), ("threshold_f32", g_thr, c_thr),
                       ("score_u8", g_u8, c_u8)):
        r[name] = {"host_pointer_us": round(med_us(g, reps, warm), 2), "cpu_1_thread_us": round(med_us(c, max(reps // 4, 5), max(warm // 4, 2)), 2)}
    r["argmax_f32"]["avx2_argmax_us"] = round(med_us(c_argmax_avx2, max(reps // 4, 5), max(warm // 4, 2)), 2)
    r["score_plus_argmax"] = {"host_pointer_us": round(med_us(lambda: (g_score(), g_argmax()), reps, warm), 2),
                              "cpu_1_thread_us": round(med_us(lambda: (c_score(), c_argmax()), max(reps // 4, 5), max(warm // 4, 2)), 2)}
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--lengths", default=",".join(map(str, LENGTHS)))
    args = ap.parse_args()
    lengths = [int(x) for x in args.lengths.split(",")]
    lm.Pipeline.hip(0)  # fails loudly without a device
    rows = [measure(n) for n in lengths]
    out = {"motif": "MX000001 (M = 15)",
           "cpu_tier": "one thread: Score = the AVX2 port (oracle/lm_avx2.c), Maximum / Threshold = the Generic default bodies "
                       "(oracle/lm_oracle.c; Avx2::argmax_f32 has another tie rule and is listed for reference)",
           "published_readme_10kb_avx2_us": 12.797, "by_length": rows, "crossover_positions": {}}
    for op in ("score_f32", "argmax_f32", "threshold_f32", "score_u8", "score_plus_argmax"):
        out["crossover_positions"][op] = crossover(lengths, [r[op]["host_pointer_us"] for r in rows], [r[op]["cpu_1_thread_us"] for r in rows])
    # Score at other motif lengths (the fit behind lm_hip_host_crossover)
    out["score_by_motif_length"] = {}
    for mm in (4, 8, 30):
        sub = [n for n in lengths if n <= 5_000_000]
        rr = [measure(n, mm) for n in sub]
        out["score_by_motif_length"][str(mm)] = {
            "by_length": [{"length": r["length"], "score_f32": r["score_f32"], "score_u8": r["score_u8"]} for r in rr],
            "crossover_score_f32": crossover(sub, [r["score_f32"]["host_pointer_us"] for r in rr], [r["score_f32"]["cpu_1_thread_us"] for r in rr]),
            "crossover_score_u8": crossover(sub, [r["score_u8"]["host_pointer_us"] for r in rr], [r["score_u8"]["cpu_1_thread_us"] for r in rr])}
    ten = next((r for r in rows if r["length"] == 10_000), None)
    if ten:
        out["readme_10kb"] = {"host_pointer_us": ten["score_plus_argmax"]["host_pointer_us"],
                              "avx2_port_us": ten["score_plus_argmax"]["cpu_1_thread_us"], "published_avx2_us": 12.797}
    print(json.dumps(out, indent=1))
    if args.json:
        Path(args.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
