"""world_size-2 (and 3) runs of the multi-GPU path on CPU with the gloo backend:
row sharding + halo hand-over + argmax / threshold / max merge must reproduce the
single-process oracle result on the whole sequence."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lightmotif_amd import distributed as D
        from oracle import c_oracle as co
        rng = np.random.default_rng(case["seed"])
        k, m, cols = 5, case["m"], 32
        enc = rng.integers(0, 4, case["length"], dtype=np.uint8)
        pssm = np.zeros((m, 8), np.float32)
        pssm[:, :5] = rng.integers(-2, 3, (m, 5)) if case["ties"] else rng.normal(0, 2, (m, 5))
        pssm[:, 4] = -np.inf
        if case.get("nan_first"):
            pssm[0, int(enc[0])] = np.nan
        full = co.stripe(enc, cols, k)
        co.configure_wrap(full, m - 1)
        R = full.rows
        a, b = D.shard_rows(R, world, rank)
        # this rank's shard: its own rows, halo filled by the exchange
        shard = torch.zeros((b - a + m - 1, 32), dtype=torch.uint8)
        shard[:b - a] = torch.from_numpy(full.data[a:b].copy())
        D.exchange_halo(shard, m - 1, cols, k - 1)
        assert np.array_equal(shard.numpy(), full.data[a:b + m - 1]), "halo differs"
        local = co.Striped(shard.numpy(), case["length"], m - 1, cols, k)
        sc, _ = co.score_rows(local, pssm, 0, b - a)
        # local argmax WITHOUT the first-cell rule on ranks > 0 (the shard API's contract)
        am = co.argmax(sc, cols)
        if rank > 0 and am is not None and np.isnan(sc[0, 0]):
            finite = np.where(np.isnan(sc[:, :cols]), -np.inf, sc[:, :cols])
            cand = np.argwhere(finite >= finite.max())
            am = tuple(map(int, cand[-1]))
        loc = None if am is None else (am, float(sc[am]))
        got_am = D.merge_argmax(loc, a)
        got_thr = D.merge_threshold([tuple(map(int, x)) for x in co.threshold(sc, cols, case["t"])], a)
        got_max = D.merge_max(None if loc is None else loc[1])
        want_sc, _ = co.score_rows(full, pssm)
        want_am = co.argmax(want_sc, cols)
        want_thr = [tuple(map(int, x)) for x in co.threshold(want_sc, cols, case["t"])]
        ok = (got_am[0] == want_am and got_thr == want_thr)
        if not case.get("nan_first"):
            ok = ok and np.float32(got_am[1]) == want_sc[want_am] and np.float32(got_max) == want_sc[want_am]
        q.put((rank, ok, got_am, want_am, len(got_thr), len(want_thr)))
    finally:
        dist.destroy_process_group()


CASES = [
    dict(seed=1, length=20_000, m=20, ties=False, t=6.0),
    dict(seed=2, length=9_999, m=7, ties=True, t=3.0),     # many equal maxima across shards
    dict(seed=3, length=5_000, m=12, ties=True, t=-np.inf, nan_first=True),
]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", CASES, ids=["plain", "ties", "nan_first"])
def test_sharded_pipeline_equals_single_process(world, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, got_am, want_am, ng, nw in results:
        assert ok, (rank, got_am, want_am, ng, nw)


def test_shard_rows_and_motif_partition():
    from lightmotif_amd import distributed as D
    for total, world in ((31_250_000, 8), (7, 3), (2, 4)):
        spans = [D.shard_rows(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    lengths = [4] * 21 + [8] * 454 + [33, 31, 30, 29, 29] + [12] * 97
    parts = D.shard_motifs(lengths, 8)
    assert sorted(i for p in parts for i in p) == list(range(len(lengths)))
    loads = [sum(lengths[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(lengths)


def test_single_process_merges_are_identity():
    from lightmotif_amd import distributed as D
    assert D.merge_argmax(((3, 4), 1.5), 10) == ((13, 4), 1.5)
    assert D.merge_argmax(None, 0) is None
    assert D.merge_threshold([(0, 1), (2, 3)], 5) == [(5, 1), (7, 3)]
    assert D.merge_max(2.5) == 2.5 and D.merge_max(None) is None
