#!/usr/bin/env python3
"""The shader clock the part sustains under each of the library's loads (`lm_hip_device_clock_mhz`: one sleeping
wavefront counting s_memtime ticks per s_memrealtime tick on a stream of its own, from a second host thread while the
main thread keeps the load's kernels queued).  Also prints what sysfs offers for the same question.  GPU box only:

    python tools/clock_probe.py [--json out.json] [--length 1000000000]
"""
import argparse
import ctypes as C
import glob
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402
from lightmotif_amd._ffi import Coords  # noqa: E402

COLS = 32


def sysfs_clocks():
    out = {}
    for p in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")) + sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")):
        try:
            out[p] = open(p).read().strip()[:200]
        except OSError as e:
            out[p] = f"unreadable: {e}"
    return out


def clock_under(L, load, seconds=0.6, window_us=5_000):
    """load(): enqueues ~a few ms of work and returns without waiting; the main thread keeps it queued for `seconds`
    while a second thread takes windows of the shader clock."""
    samples, stop = [], threading.Event()

    def sampler():
        mhz = C.c_double(0)
        while not stop.is_set():
            if L.lm_hip_device_clock_mhz(0, window_us, C.byref(mhz)) == 0:
                samples.append(mhz.value)

    # preheat: the clock settles over the first tens of milliseconds of a load
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        load()
        torch.cuda.synchronize()
    # the same loop without the probe: the probe must not change what it measures
    t0 = time.perf_counter()
    n0 = 0
    while time.perf_counter() - t0 < 0.3:
        load()
        n0 += 1
        if n0 % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    alone_ms = (time.perf_counter() - t0) / n0 * 1e3
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        load()
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    stop.set()
    th.join()
    s = np.asarray(samples[1:-1] if len(samples) > 4 else samples)
    return {"mhz_median": round(float(np.median(s)), 1) if len(s) else None,
            "mhz_min": round(float(s.min()), 1) if len(s) else None, "mhz_max": round(float(s.max()), 1) if len(s) else None,
            "windows": int(len(s)), "calls": n, "ms_per_call": round(wall / n * 1e3, 4), "ms_per_call_without_probe": round(alone_ms, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--length", type=int, default=1_000_000_000)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    pli = lm.Pipeline.hip()
    L = pli._L
    out = {"sysfs": sysfs_clocks()}
    mhz = C.c_double(0)
    time.sleep(0.5)
    assert L.lm_hip_device_clock_mhz(0, 50_000, C.byref(mhz)) == 0
    out["idle_mhz"] = round(mhz.value, 1)

    length, m = args.length, 20
    rng = np.random.default_rng(3)
    pssm = lm.create(["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]).counts.normalize(0.1).log_odds()
    rows = -(-length // COLS)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    seq = torch.randint(0, 4, (rows + m - 1, COLS), dtype=torch.uint8, device=dev, generator=gen)
    pli.configure_wrap_dptr(seq.data_ptr(), rows, COLS, COLS, m - 1, 4)
    scores = torch.empty((rows, COLS), dtype=torch.float32, device=dev)
    h, p = pli._h, pssm._device(pli)
    sp, op = C.c_void_p(seq.data_ptr()), C.c_void_p(scores.data_ptr())
    orow, mi = C.c_size_t(0), C.c_size_t(0)
    found, best, value = C.c_int(0), Coords(), C.c_float(0)

    def store():
        L.lm_hip_score_f32_dptr(h, p, sp, rows + m - 1, COLS, COLS, m - 1, length, 0, rows, op, COLS, C.byref(orow), C.byref(mi))
    store()
    torch.cuda.synchronize()
    sample = scores[: min(rows, 1 << 18)].flatten()
    t = float(torch.quantile(sample[torch.isfinite(sample)].float(), 1 - 1e-5))
    n = C.c_size_t(0)

    def fused_threshold():
        ptr, vals = C.POINTER(Coords)(), C.POINTER(C.c_float)()
        L.lm_hip_score_threshold_f32_dptr(h, p, sp, rows + m - 1, COLS, COLS, m - 1, length, 0, rows, C.c_float(t), C.byref(ptr),
                                          C.byref(vals), C.byref(n))
        L.lm_hip_free(ptr)
        L.lm_hip_free(vals)

    def fused_argmax():
        L.lm_hip_score_argmax_f32_dptr(h, p, sp, rows + m - 1, COLS, COLS, m - 1, length, 0, rows, C.byref(found), C.byref(best), C.byref(value))

    def exact_threshold():
        L.lm_hip_ctx_set_prefilter(h, 0)
        fused_threshold()
        L.lm_hip_ctx_set_prefilter(h, 1)

    def argmax_stored():
        L.lm_hip_argmax_f32_dptr(h, op, rows, COLS, COLS, C.byref(found), C.byref(best), C.byref(value))

    out["store_m20"] = clock_under(L, store)
    out["fused_threshold_m20"] = clock_under(L, fused_threshold)
    out["fused_argmax_m20"] = clock_under(L, fused_argmax)
    out["exact_fused_threshold_m20"] = clock_under(L, exact_threshold)
    out["argmax_stored"] = clock_under(L, argmax_stored)
    del seq, scores
    torch.cuda.empty_cache()
    # the JASPAR batch (configs[2])
    try:
        sys.path.insert(0, str(ROOT))
        import bench  # noqa: E402
        from lightmotif_amd import distributed as D
        st = bench.c3_setup(pli, dev, 1, 0, 100_000_000, 0)

        def c3_batch():
            D.scan_threshold_batch_sharded(pli, st["pssms"], st["ts"], st["seq"], device=dev, parts=st["parts"])
        out["c3_threshold_batch"] = clock_under(L, c3_batch, seconds=1.0)
    except Exception as e:  # the probe is a tool: report, do not fail the rest
        out["c3_threshold_batch"] = {"error": repr(e)[:300]}
    print(json.dumps(out, indent=1))
    if args.json:
        Path(args.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
