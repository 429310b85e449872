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
        got_thr = D.merge_threshold(co.threshold(sc, cols, case["t"]), a)
        got_max = D.merge_max(None if loc is None else loc[1])
        want_sc, _ = co.score_rows(full, pssm)
        want_am = co.argmax(want_sc, cols)
        want_thr = np.asarray(co.threshold(want_sc, cols, case["t"]), np.int64).reshape(-1, 2)
        ok = (got_am[0] == want_am and got_thr.dtype == np.int64 and np.array_equal(got_thr, want_thr))
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
    assert D.merge_threshold([(0, 1), (2, 3)], 5).tolist() == [[5, 1], [7, 3]]
    assert D.merge_threshold([], 5).shape == (0, 2)
    # the pure rules: ties -> the later (row, col); NaN only through the first-cell rule of shard 0
    nan = float("nan")
    assert D.combine_argmax([((1, 2), 3.0), None, ((9, 0), 3.0)]) == ((9, 0), 3.0)
    assert D.combine_argmax([((1, 31), 3.0), ((1, 2), 3.0)]) == ((1, 31), 3.0)
    assert D.combine_argmax([((0, 0), nan), ((9, 0), 3.0)])[0] == (0, 0)
    assert D.combine_argmax([None, None]) is None
    assert D.combine_threshold([np.array([[0, 1]]), np.zeros((0, 2)), np.array([[0, 0], [3, 4]])],
                               [0, 10, 20]).tolist() == [[0, 1], [20, 0], [23, 4]]
    assert D.merge_max(2.5) == 2.5 and D.merge_max(None) is None


def test_c_abi_combine_rule_equals_the_python_rule():
    """lm_hip_combine_argmax (host-side C, no device needed) against distributed.combine_argmax on
    random record lists with ties, empty shards, NaN values and the first-cell (0,0) NaN."""
    from lightmotif_amd import distributed as D
    rng = np.random.default_rng(7)
    nan = float("nan")
    for trial in range(300):
        n = int(rng.integers(1, 9))
        recs, row = [], 0
        for g in range(n):
            span = int(rng.integers(1, 50))
            kind = rng.integers(0, 6)
            if kind == 0:
                recs.append(None)
            elif kind == 1 and g == 0:
                recs.append(((0, 0), nan))                      # first-cell rule of the shard holding row 0
            elif kind == 1:
                recs.append(((row + int(rng.integers(0, span)), int(rng.integers(0, 32))), nan))  # must be ignored
            else:
                recs.append(((row + int(rng.integers(0, span)), int(rng.integers(0, 32))),
                             float(np.float32(rng.integers(-2, 3)))))
            row += span
        want = D.combine_argmax(recs)
        got = D.combine_argmax_cabi(recs)
        if want is None:
            assert got is None
        else:
            assert got[0] == want[0] and (got[1] == want[1] or (got[1] != got[1] and want[1] != want[1])), (recs, got, want)


class _OraclePipeline:
    """Stand-in for Pipeline in the CPU run of the motif-sharded batch: scores with the oracle."""

    def __init__(self, striped):
        self.s = striped

    def _scores(self, p):
        from oracle import c_oracle as co
        return co.score_rows(self.s, p.data)[0]

    def scan_argmax_batch(self, pssms, seq):
        from oracle import c_oracle as co
        out = []
        for p in pssms:
            sc = self._scores(p)
            cell = co.argmax(sc, 32)
            out.append(None if cell is None else (cell, float(sc[cell])))
        return out

    def prepare_batch(self, pssms, ts):
        return _Prepared(pssms, ts)

    def scan_threshold_batch(self, pssms, ts, seq):
        from oracle import c_oracle as co
        if isinstance(pssms, _Prepared):   # the form Pipeline.prepare_batch hands back: thresholds travel inside
            assert ts is None
            pssms, ts = pssms.pssms, pssms.ts
        out = []
        for p, t in zip(pssms, ts):
            sc = self._scores(p)
            rc = np.asarray(co.threshold(sc, 32, t), np.int64).reshape(-1, 2)
            out.append((rc, sc[rc[:, 0], rc[:, 1]].astype(np.float32)))
        return out


class _Prepared:
    def __init__(self, pssms, ts):
        self.pssms, self.ts = list(pssms), list(ts)

    def __len__(self):
        return len(self.pssms)


class _P:
    def __init__(self, data):
        self.data = data

    def __len__(self):
        return self.data.shape[0]


def _motif_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lightmotif_amd import distributed as D
        from oracle import c_oracle as co
        rng = np.random.default_rng(99)
        enc = rng.integers(0, 4, 6_000, dtype=np.uint8)
        s = co.stripe(enc, 32, 5)
        co.configure_wrap(s, 32)
        pssms = []
        for m in [4, 33, 8, 8, 12, 5, 20, 9, 6, 15, 7]:
            p = np.zeros((m, 8), np.float32)
            p[:, :4] = rng.integers(-2, 3, (m, 4)) if m % 2 else rng.normal(0, 2, (m, 4))
            p[:, 4] = -np.inf
            pssms.append(_P(p))
        ts = [1.5 + 0.1 * i for i in range(len(pssms))]
        pli = _OraclePipeline(s)
        want_am = pli.scan_argmax_batch(pssms, None)
        want_th = pli.scan_threshold_batch(pssms, ts, None)
        got_am = D.scan_argmax_batch_sharded(pli, pssms, None)
        got_th = D.scan_threshold_batch_sharded(pli, pssms, ts, None)
        prepared = D.prepare_sharded_batch(pli, pssms, ts)   # the job-loop form: this rank's share built once
        got_prepared = D.scan_threshold_batch_sharded(pli, pssms, ts, None, prepared=prepared)
        ok = got_am == want_am and len(got_th) == len(want_th) == len(got_prepared)
        for (gc, gv), (pc, pv), (wc, wv) in zip(got_th, got_prepared, want_th):
            ok = ok and np.array_equal(gc, wc) and np.array_equal(gv.view(np.uint32), wv.view(np.uint32))
            ok = ok and np.array_equal(pc, wc) and np.array_equal(pv.view(np.uint32), wv.view(np.uint32))
        parts = D.shard_motifs([len(p) for p in pssms], world)
        q.put((rank, bool(ok), len(parts[rank])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_motif_sharded_batch_equals_single_process(world):
    """configs[2] across GPUs: every rank scans its share of the motif list over the whole
    sequence; the gathered per-motif results equal the single-process batch (main.rs:502-561)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_motif_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results), results
    assert all(n > 0 for _, _, n in results)
