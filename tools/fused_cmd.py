#!/usr/bin/env python3
"""A command for counter collection (tools/collect_stalls.sh): the fused scans only -- `reps` fused threshold calls
(p = 1e-5) and fused argmax calls over 1 Gbp x M, or (--c3) the JASPAR threshold batch over 100 Mbp.

    bash tools/collect_stalls.sh r05f "python tools/fused_cmd.py --reps 6"
    bash tools/collect_stalls.sh r05c3 "python tools/fused_cmd.py --c3 --reps 2"
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import lightmotif_amd as lm  # noqa: E402
from lightmotif_amd._ffi import Coords  # noqa: E402

COLS = 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--length", type=int, default=1_000_000_000)
    ap.add_argument("--motif-len", type=int, default=20)
    ap.add_argument("--c3", action="store_true")
    ap.add_argument("--protein", action="store_true", help="configs[4]: 200 Mres x M = 12 protein, the block scan (score_prefilter_blk.hpp)")
    ap.add_argument("--motifs", type=int, default=0)
    ap.add_argument("--realistic", action="store_true", help="with --c3: the non-i.i.d. 100 Mbp of tools/realistic_inputs.py instead of the uniform one")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    pli = lm.Pipeline.hip()
    L = pli._L
    if args.c3:
        import bench
        from lightmotif_amd import distributed as D
        st = bench.c3_setup(pli, dev, 1, 0, 100_000_000, args.motifs)
        seq = st["seq"]
        if args.realistic:
            import realistic_inputs as ri
            seq = pli.stripe(lm.EncodedSequence(ri.realistic_dna(100_000_000)))
            seq.configure_wrap(st["max_m"] - 1)
        prepared = D.prepare_sharded_batch(pli, st["pssms"], st["ts"], parts=st["parts"])
        for _ in range(args.reps):
            D.scan_threshold_batch_sharded(pli, st["pssms"], st["ts"], seq, device=dev, parts=st["parts"], prepared=prepared)
        torch.cuda.synchronize()
        return
    length, m = args.length, args.motif_len
    rng = np.random.default_rng(3)
    k = 5
    if args.protein:
        k = 21
        length = args.length if args.length != 1_000_000_000 else 200_000_000
        m = args.motif_len if args.motif_len != 20 else 12
        sym = lm.lib.PROTEIN_SYMBOLS[:-1]
        pssm = lm.create(["".join(sym[i] for i in rng.integers(0, len(sym), m)) for _ in range(6)], protein=True).counts.normalize(0.1).log_odds()
    else:
        pssm = lm.create(["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]).counts.normalize(0.1).log_odds()
    rows = -(-length // COLS)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    seq = torch.randint(0, k - 1, (rows + m - 1, COLS), dtype=torch.uint8, device=dev, generator=gen)
    pli.configure_wrap_dptr(seq.data_ptr(), rows, COLS, COLS, m - 1, k - 1)
    h, p = pli._h, pssm._device(pli)
    sp = C.c_void_p(seq.data_ptr())
    t = float(pssm.score_for_pvalue(1e-5))
    n = C.c_size_t(0)
    found, best, value = C.c_int(0), Coords(), C.c_float(0)
    for _ in range(args.reps):
        ptr, vals = C.POINTER(Coords)(), C.POINTER(C.c_float)()
        L.lm_hip_score_threshold_f32_dptr(h, p, sp, rows + m - 1, COLS, COLS, m - 1, length, 0, rows, C.c_float(t), C.byref(ptr),
                                          C.byref(vals), C.byref(n))
        L.lm_hip_free(ptr)
        L.lm_hip_free(vals)
        L.lm_hip_score_argmax_f32_dptr(h, p, sp, rows + m - 1, COLS, COLS, m - 1, length, 0, rows, C.byref(found), C.byref(best),
                                       C.byref(value))
    print("hits", n.value, "argmax", best.row, best.col, value.value)


if __name__ == "__main__":
    main()
