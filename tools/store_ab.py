#!/usr/bin/env python3
"""Store kernel only, for A/B runs of library variants (tools/build_variant.py; LM_HIP_LIBRARY selects one):
median / minimum kernel time of `score_into` on a resident DNA sequence per motif length, and a digest of the
score matrix so that two libraries can be compared bit for bit.  GPU box only:

    python tools/store_ab.py 1000000000 24,28,33,36 [track]     (track: the handle path's default, the store kernel
                                                                  that tracks the best value on the way)
"""
import hashlib
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm  # noqa: E402
import bench as B  # noqa: E402


def device_digest(ptr: int, n: int, dev) -> str:
    """Two wrapping 64-bit sums over the matrix's bit patterns (plain, and weighted by position), on the device."""
    class Foreign:                                 # torch adopts the library's buffer through the CUDA array interface
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}
    t = torch.as_tensor(Foreign(), device=dev)
    s0 = s1 = 0
    step = 1 << 26
    for a in range(0, n, step):
        v = t[a:a + step].to(torch.int64)
        w = (torch.arange(a, a + v.numel(), device=dev, dtype=torch.int64) % 65521) + 1
        s0 = (s0 + int(v.sum().item())) & 0xFFFFFFFFFFFFFFFF
        s1 = (s1 + int((v * w).sum().item())) & 0xFFFFFFFFFFFFFFFF
    return "%016x%016x" % (s0, s1)


def main():
    length = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
    ms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "24,28,33,36").split(",")]
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream()
    pli = lm.Pipeline.hip(0, stream=stream.cuda_stream)
    track = len(sys.argv) > 3 and sys.argv[3] == "track"
    pli.set_track_argmax(track)                    # off: the plain store kernel (bench.py's N = 1 step)
    rows = -(-length // B.COLS)
    for m in ms:
        pssm = B.synth_pssm(m)
        shard = B.synth_shard(rows, 0, rows, length, m - 1, dev)
        if m > 1:                                  # wrap rows (seq.rs:373-378): the next column's first rows, N in the last
            shard[rows:] = 4
            shard[rows:, :B.COLS - 1] = shard[:m - 1, 1:]
        seq = pli.adopt_sequence(shard.data_ptr(), rows, m - 1, B.COLS, B.COLS, length)
        scores = lm.StripedScores.empty(pli, B.COLS)
        for _ in range(10):
            pli.score_into(pssm, seq, scores)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record(stream)
            pli.score_into(pssm, seq, scores)
            b.record(stream)
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in ev)
        digest = device_digest(scores.data_ptr, scores.rows * scores.stride, dev)
        print(json.dumps({"M": m, "tracked": track, "kernel": pli.last_kernel, "ms": round(t[len(t) // 2], 4), "ms_min": round(t[0], 4),
                          "sha": digest}), flush=True)
        del scores, seq, shard
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
