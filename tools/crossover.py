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
    pssm = lm.create(["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"]).counts.normalize(0.1).log_odds()
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

    def c_score():
        co.avx2_score_rows(s, p, out=cout, row_end=rows, threads=1)

    def g_argmax():
        st = L.lm_hip_argmax_f32(out.ctypes.data, rows, COLS, COLS, C.byref(found), C.byref(best), C.byref(value))
        assert st == 0, _ffi.last_error()

    def c_argmax():
        co.avx2_argmax(cout, length + 1 - m)

    g_score()
    c_score()
    assert np.array_equal(out.view(np.uint32), cout.view(np.uint32))
    t = float(np.sort(cout[:, :COLS].ravel())[-max(1, length // 100_000)])
    n = C.c_size_t(0)

    def g_thr():
        ptr = C.POINTER(_ffi.Coords)()
        st = L.lm_hip_threshold_f32(out.ctypes.data, rows, COLS, COLS, C.c_float(t), C.byref(ptr), C.byref(n))
        assert st == 0, _ffi.last_error()
        L.lm_hip_free(ptr)

    def c_thr():
        co.threshold(cout, COLS, t)   # the default body of Threshold (pli/mod.rs:210-221): the reference has no SIMD form

    def g_u8():
        st = L.lm_hip_score_u8_host(mat.ctypes.data, total_rows, s.stride, COLS, s.wrap, length, w.ctypes.data, m, 32, 5, 0, rows, 1,
                                    out8.ctypes.data, COLS, C.byref(orow), C.byref(mi))
        assert st == 0, _ffi.last_error()

    def c_u8():
        co.avx2_score_rows_u8(s, w, out=cout8, row_end=rows)

    g_u8()
    c_u8()
    assert np.array_equal(out8, cout8)
    r = {"length": length, "rows": rows}
    for name, g, c in (("score_f32", g_score, c_score), ("argmax_f32", g_argmax, c_argmax), ("threshold_f32", g_thr, c_thr),
                       ("score_u8", g_u8, c_u8)):
        r[name] = {"host_pointer_us": round(med_us(g, reps, warm), 2), "cpu_1_thread_us": round(med_us(c, max(reps // 4, 5), max(warm // 4, 2)), 2)}
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
    out = {"motif": "MX000001 (M = 15)", "cpu_tier": "AVX2 port, 1 thread (oracle/lm_avx2.c); threshold: the Generic default body",
           "published_readme_10kb_avx2_us": 12.797, "by_length": rows, "crossover_positions": {}}
    for op in ("score_f32", "argmax_f32", "threshold_f32", "score_u8", "score_plus_argmax"):
        out["crossover_positions"][op] = crossover(lengths, [r[op]["host_pointer_us"] for r in rows], [r[op]["cpu_1_thread_us"] for r in rows])
    ten = next((r for r in rows if r["length"] == 10_000), None)
    if ten:
        out["readme_10kb"] = {"host_pointer_us": ten["score_plus_argmax"]["host_pointer_us"],
                              "avx2_port_us": ten["score_plus_argmax"]["cpu_1_thread_us"], "published_avx2_us": 12.797}
    print(json.dumps(out, indent=1))
    if args.json:
        Path(args.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
