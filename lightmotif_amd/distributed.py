"""Sharding of the scoring path across the GPUs of one node.

The path shards embarrassingly, in two ways:

* **rows of one long sequence** -- output row ``r`` depends only on input rows
  ``r .. r+M-1`` (pli/mod.rs:99-101), which is why ``score_rows_into`` takes a row
  range (pli/mod.rs:72-78).  One process per GPU owns a contiguous range of the
  ``R`` striped rows plus an ``M-1``-row halo; scoring needs NO collective;
* **motifs of a many-motif batch** -- the reference CLI sends every motif against
  every sequence as independent jobs (lightmotif-cli main.rs:502-561).  Every rank
  holds the whole (100 MB-scale) sequence and scans its share of the motif list
  (``shard_motifs``, balanced on sum(M)); results are gathered in motif order
  (``scan_argmax_batch_sharded`` / ``scan_threshold_batch_sharded``).

RCCL is used only for

* the one-off halo hand-over at set-up (each rank receives the first ``M-1`` rows
  of its successor; the last rank receives rank 0's rows to build the
  reference's wrap rows, seq.rs:373-378), and
* the final merge: ``all_gather`` of one 32-byte ``(found, score, row, col)`` record
  per rank for argmax; hit counts + exact-length hit lists for threshold.

Two transports carry the merge:

* the C ABI's own communicator (``lm_hip_comm_*`` / ``lm_hip_merge_*`` in
  include/lightmotif_hip.h, RCCL bound directly by the library) -- what a Rust or
  C++ host uses, and what ``bench.py`` times on GPUs;
* ``torch.distributed`` (backend "nccl" = RCCL, or "gloo" on CPU tensors, which is
  how the tests cover world_size > 1 without GPUs).

The merge *rules* live in the pure functions ``combine_argmax`` /
``combine_threshold`` below (and, identically, in csrc/comm.hip); both transports
only move the records.
"""
from __future__ import annotations

import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_rows(total_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row range ``[a, b)`` of rank ``rank``."""
    a = total_rows * rank // world_size
    b = total_rows * (rank + 1) // world_size
    return a, b


def shard_motifs(lengths: Sequence[float], world_size: int) -> List[List[int]]:
    """Partition a motif list over ranks balancing the sum of per-motif weights -- the motif
    lengths, or any better cost estimate (``bench.py --config c3`` passes the expected scan cost:
    table reads per position, ~0 for a motif whose threshold no cell can reach):
    longest-processing-time greedy.  Returns motif indices per rank."""
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    load = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda j: load[j])
        out[r].append(i)
        load[r] += lengths[i]
    for lst in out:
        lst.sort()
    return out


def _f32_bits(x: float) -> int:
    return struct.unpack("<i", struct.pack("<f", x))[0]


def _bits_f32(b: int) -> float:
    return struct.unpack("<f", struct.pack("<i", b))[0]


def _world(group=None) -> Tuple[int, int]:
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _coll_device(device, group=None):
    """gloo moves host tensors only; nccl (= RCCL) device tensors only."""
    if dist.is_initialized() and dist.get_backend(group) == "gloo":
        return torch.device("cpu")
    return torch.device(device)


# ---- the merge rules (pure) -------------------------------------------------------


def combine_argmax(records: Sequence[Optional[Tuple[Tuple[int, int], float]]]):
    """Global ``Maximum::argmax`` from per-shard results given in ascending row order.

    ``records[g]`` is ``((row, col), value)`` in GLOBAL row coordinates, computed
    WITHOUT the first-cell rule on every shard but the one holding row 0
    (``lm_hip_argmax_shard_f32_dptr(first_cell_rule = g == 0)``), or ``None`` for an
    empty shard.

    Rule (pli/mod.rs:135-155): maximal score; ties go to the LAST cell in
    (row, col) order; NaN never wins, except that a NaN in the matrix's very first
    cell -- reported by shard 0 through the first-cell rule -- wins outright.
    """
    best = None
    for rec in records:
        if rec is None:
            continue
        (r, c), v = rec
        if v != v:          # NaN can only be the first-cell rule of the shard holding row 0
            if r == 0 and c == 0:
                return (0, 0), v
            continue
        if best is None or v > best[1] or (v == best[1] and (r, c) > best[0]):
            best = ((r, c), v)
    return best


def combine_threshold(lists: Sequence[np.ndarray], row_offsets: Sequence[int]) -> np.ndarray:
    """Global ``Threshold::threshold`` list as an ``(n, 2)`` int64 array: shards hold
    ascending contiguous row ranges, so concatenating the per-shard row-major lists in
    shard order IS the reference's row-major order (pli/mod.rs:212-218)."""
    parts = []
    for coords, off in zip(lists, row_offsets):
        a = np.asarray(coords, dtype=np.int64).reshape(-1, 2).copy()
        a[:, 0] += off
        parts.append(a)
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, 2), np.int64)


# ---- torch.distributed transport ----------------------------------------------------


def merge_argmax(local: Optional[Tuple[Tuple[int, int], float]], row_offset: int,
                 device: torch.device | str = "cpu", group=None):
    """``combine_argmax`` over the ranks of ``group``: ``local`` is this rank's shard
    result with rows relative to the shard, ``row_offset`` the shard's first global row.
    Every rank returns the same ``((row, col), value)`` (or ``None``)."""
    rank, world = _world(group)
    rec = torch.zeros(4, dtype=torch.int64)
    if local is not None:
        (r, c), v = local
        rec[0], rec[1], rec[2], rec[3] = 1, _f32_bits(v), r + row_offset, c
    if world == 1:
        recs = [rec]
    else:
        rec = rec.to(_coll_device(device, group))
        bufs = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(bufs, rec, group=group)
        recs = [b.cpu() for b in bufs]
    return combine_argmax([None if int(b[0]) == 0 else ((int(b[2]), int(b[3])), _bits_f32(int(b[1])))
                           for b in recs])


def _gather_exact(mine: torch.Tensor, device, group=None) -> List[torch.Tensor]:
    """Variable-length gather of 2-D int64/float tensors along dim 0: counts by
    ``all_gather``, then one broadcast per rank with that rank's exact length -- no
    padding to the longest list (at a 1e-3 hit rate a rank holds ~1e6 records)."""
    rank, world = _world(group)
    dev = _coll_device(device, group)
    n = torch.tensor([mine.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(x.item()) for x in counts]
    out = []
    for src, cnt in enumerate(counts):
        buf = mine.to(dev) if src == rank else torch.empty((cnt,) + tuple(mine.shape[1:]),
                                                           dtype=mine.dtype, device=dev)
        if cnt:
            src_global = src if group is None else dist.get_global_rank(group, src)
            dist.broadcast(buf, src=src_global, group=group)
        out.append(buf)
    return out


def merge_threshold(local_coords, row_offset: int, device: torch.device | str = "cpu",
                    group=None) -> np.ndarray:
    """``combine_threshold`` over the ranks of ``group``.  ``local_coords`` is this rank's
    ``(n, 2)`` array (or list) of (row, col) in row-major order with rows relative to the
    shard.  Returns the global ``(N, 2)`` int64 array on every rank."""
    rank, world = _world(group)
    mine = np.asarray(local_coords, dtype=np.int64).reshape(-1, 2).copy()
    mine[:, 0] += row_offset
    if world == 1:
        return mine
    bufs = _gather_exact(torch.from_numpy(mine), device, group)
    return np.concatenate([b.cpu().numpy() for b in bufs], axis=0)


def merge_max(local: Optional[float], device: torch.device | str = "cpu", group=None) -> Optional[float]:
    """``Maximum::max`` across shards (value of the merged argmax when no NaN is involved)."""
    rank, world = _world(group)
    t = torch.tensor([float("-inf") if local is None else local, 0.0 if local is None else 1.0],
                     dtype=torch.float32, device=_coll_device(device, group) if world > 1 else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t[0]) if float(t[1]) > 0 else None


def exchange_halo(shard: torch.Tensor, halo_rows: int, columns: int, default_symbol: int,
                  group=None) -> torch.Tensor:
    """Fills the last ``halo_rows`` rows of ``shard`` (shape ``(rows + halo, stride)``
    uint8, first ``rows`` rows already hold this rank's part of the striped matrix).

    Rank g < G-1 receives the first ``halo_rows`` rows of rank g+1.  The last rank
    receives rank 0's first rows and turns them into the reference's wrap rows:
    ``wrap[i][j] = data[i][j+1]``, last column = default symbol (seq.rs:373-378).
    Requires every shard to have at least ``halo_rows`` rows.
    """
    rank, world = _world(group)
    rows = shard.shape[0] - halo_rows
    if halo_rows == 0:
        return shard
    head = shard[:halo_rows].contiguous()
    if world > 1 and head.is_cuda and dist.get_backend(group) == "gloo":
        head = head.cpu()  # gloo collectives run on host tensors: stage through the host
    if world == 1:
        recv = head
    else:
        # every rank needs the head rows of its successor: (M-1) x 32 bytes each.  An
        # all_gather of all heads (a few KB in total) is the sturdiest way to move them --
        # no point-to-point pairing to get wrong, one collective every rank enters alike.
        heads = [torch.empty_like(head) for _ in range(world)]
        dist.all_gather(heads, head, group=group)
        recv = heads[(rank + 1) % world]
    if rank == world - 1:
        wrapped = torch.full_like(recv, default_symbol)   # padding past `columns` too (dense.rs fill)
        wrapped[:, :columns - 1] = recv[:, 1:columns]
        recv = wrapped
    shard[rows:] = recv.to(shard.device)
    return shard


# ---- many-motif batches sharded by motif (configs[2] across GPUs) ------------------


def scan_argmax_batch_sharded(pli, pssms, seq, device: torch.device | str = "cpu", group=None,
                              parts: Optional[List[List[int]]] = None):
    """``Pipeline.scan_argmax_batch`` with the motif list split over the ranks.

    Every rank holds the whole sequence (``seq``) and the whole motif list; rank g scans
    ``shard_motifs(...)[g]`` with ONE batched call and the per-motif results are gathered
    back into motif order.  Returns, on every rank, the list a single-process
    ``pli.scan_argmax_batch(pssms, seq)`` returns."""
    rank, world = _world(group)
    if parts is None:
        parts = shard_motifs([len(p) for p in pssms], world)
    mine = parts[rank]
    local = pli.scan_argmax_batch([pssms[i] for i in mine], seq) if mine else []
    rec = np.zeros((len(mine), 5), np.int64)
    for j, (i, res) in enumerate(zip(mine, local)):
        rec[j, 0] = i
        if res is not None:
            rec[j, 1:] = (1, _f32_bits(res[1]), res[0][0], res[0][1])
    if world == 1:
        allrec = rec
    else:
        allrec = np.concatenate([b.cpu().numpy() for b in _gather_exact(torch.from_numpy(rec), device, group)])
    out: List[Optional[Tuple[Tuple[int, int], float]]] = [None] * len(pssms)
    for i, found, bits, r, c in allrec.tolist():
        out[i] = ((r, c), _bits_f32(bits)) if found else None
    return out


class _InMotifOrder(Sequence):
    """One rank's results (in the order of its share) read in motif order; elements are cut on access like
    ``BatchHits`` cuts them."""

    def __init__(self, local, mine: List[int], n: int):
        self._local, self._n = local, n
        self._where = {i: j for j, i in enumerate(mine)}

    def __len__(self) -> int:
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        j = self._where.get(i)
        return None if j is None else self._local[j]


def prepare_sharded_batch(pli, pssms, thresholds, group=None, parts: Optional[List[List[int]]] = None):
    """This rank's share of the motif list as a ``MotifBatch`` (``Pipeline.prepare_batch``), built once for a loop of
    ``scan_threshold_batch_sharded(..., prepared=...)`` calls over many sequences; None when the share is empty."""
    rank, world = _world(group)
    if parts is None:
        parts = shard_motifs([len(p) for p in pssms], world)
    mine = parts[rank]
    return pli.prepare_batch([pssms[i] for i in mine], [thresholds[i] for i in mine]) if mine else None


def scan_threshold_batch_sharded(pli, pssms, thresholds, seq, device: torch.device | str = "cpu",
                                 group=None, parts: Optional[List[List[int]]] = None, prepared=None):
    """``Pipeline.scan_threshold_batch`` with the motif list split over the ranks: per
    motif ``(coords (n_i, 2) int64 in row-major order, values (n_i,) f32)``, in motif
    order, identical on every rank to the single-process call.  ``prepared``: what
    ``prepare_sharded_batch`` returned for the same motifs, thresholds and ``parts``."""
    rank, world = _world(group)
    if parts is None:
        parts = shard_motifs([len(p) for p in pssms], world)
    mine = parts[rank]
    if prepared is not None and len(prepared) != len(mine):
        raise ValueError("prepared batch does not match this rank's share of the motif list")
    if prepared is not None:
        local = pli.scan_threshold_batch(prepared, None, seq)
    else:
        local = pli.scan_threshold_batch([pssms[i] for i in mine], [thresholds[i] for i in mine], seq) if mine else []
    if world == 1:
        return _InMotifOrder(local, mine, len(pssms))
    head = np.zeros((len(mine), 2), np.int64)           # (motif, hit count)
    for j, (i, (coords, _)) in enumerate(zip(mine, local)):
        head[j] = (i, len(coords))
    n_local = int(head[:, 1].sum())
    body = np.zeros((n_local, 3), np.int64)             # (row, col, f32 bits) per hit
    pos = 0
    for coords, vals in local:
        n = len(coords)
        body[pos:pos + n, :2] = coords
        body[pos:pos + n, 2] = np.asarray(vals, np.float32).view(np.int32)
        pos += n
    heads = [b.cpu().numpy() for b in _gather_exact(torch.from_numpy(head), device, group)]
    bodies = [b.cpu().numpy() for b in _gather_exact(torch.from_numpy(body), device, group)]
    out = [None] * len(pssms)
    for h, b in zip(heads, bodies):
        pos = 0
        for i, n in h.tolist():
            seg = b[pos:pos + n]
            out[i] = (np.ascontiguousarray(seg[:, :2]),
                      np.ascontiguousarray(seg[:, 2]).astype(np.int32).view(np.float32))
            pos += n
    return out


# ---- the C ABI's own RCCL communicator (what a Rust / C++ host uses) ------------------


class CabiComm:
    """``lm_hip_comm_*`` / ``lm_hip_merge_*`` of include/lightmotif_hip.h: RCCL bound directly
    by the library, no torch in the data path.  ``torch.distributed`` (or any other channel)
    is needed once, to hand rank 0's 128-byte unique id to the other ranks."""

    def __init__(self, pli, unique_id: bytes, nranks: int, rank: int):
        import ctypes as C
        from . import _ffi
        self._pli, self._L, self._C = pli, pli._L, C
        self.rank, self.nranks = rank, nranks
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _ffi.check(self._L.lm_hip_comm_create(pli._h, buf, nranks, rank, C.byref(h)))
        self._h = h

    @staticmethod
    def unique_id(pli) -> bytes:
        import ctypes as C
        from . import _ffi
        buf = (C.c_uint8 * 128)()
        _ffi.check(pli._L.lm_hip_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_torch(cls, pli, device: torch.device | str = "cpu", group=None,
                   deadline_s: Optional[float] = None) -> "CabiComm":
        """Rank 0 draws the unique id, ``torch.distributed`` broadcasts its 128 bytes, every rank joins.

        ``deadline_s``: ``ncclCommInitRank`` blocks until every rank has arrived and has no timeout of its own; with
        a deadline it runs on a helper thread and a rank that waits longer raises ``TimeoutError`` (the thread
        is left behind -- the caller is expected to fall back to another transport and to leave the process with
        ``os._exit`` at the end)."""
        rank, world = _world(group)
        dev = _coll_device(device, group) if world > 1 else torch.device("cpu")
        t = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            t = torch.frombuffer(bytearray(cls.unique_id(pli)), dtype=torch.uint8).clone()
        if world > 1:
            t = t.to(dev)
            dist.broadcast(t, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
        uid = bytes(t.cpu().numpy().tobytes())
        if deadline_s is None:
            return cls(pli, uid, world, rank)
        import threading
        box: dict = {}

        def join():
            try:
                box["comm"] = cls(pli, uid, world, rank)
            except BaseException as e:     # noqa: BLE001 -- handed to the caller's thread
                box["error"] = e
        th = threading.Thread(target=join, daemon=True, name="lm-hip-comm-init")
        th.start()
        th.join(deadline_s)
        if th.is_alive():
            raise TimeoutError(f"ncclCommInitRank did not return within {deadline_s:.0f} s on rank {rank}")
        if "error" in box:
            raise box["error"]
        return box["comm"]

    def info(self) -> Tuple[int, int]:
        """``(rank, nranks)`` as the library's communicator holds them (``lm_hip_comm_info``): what RCCL was
        initialised with, not what the launcher's environment says."""
        from . import _ffi
        r, n = self._C.c_int(-1), self._C.c_int(-1)
        _ffi.check(self._L.lm_hip_comm_info(self._h, self._C.byref(r), self._C.byref(n)))
        return r.value, n.value

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.lm_hip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def exchange_halo(self, shard: torch.Tensor, halo_rows: int, columns: int, default_symbol: int) -> None:
        from . import _ffi
        rows = shard.shape[0] - halo_rows
        _ffi.check(self._L.lm_hip_exchange_halo_dptr(self._pli._h, self._h, self._C.c_void_p(shard.data_ptr()),
                                                     rows, shard.shape[1], columns, halo_rows, default_symbol))

    def merge_argmax(self, local, row_offset: int):
        from . import _ffi
        C = self._C
        found, best, value = C.c_int(0), _ffi.Coords(), C.c_float(0)
        loc = _ffi.Coords()
        v = 0.0
        if local is not None:
            (loc.row, loc.col), v = local
        _ffi.check(self._L.lm_hip_merge_argmax(self._pli._h, self._h, int(local is not None), C.byref(loc), v,
                                               row_offset, C.byref(found), C.byref(best), C.byref(value)))
        return ((best.row, best.col), float(value.value)) if found.value else None

    def argmax_sharded(self, scores, row_offset: int):
        """``StripedScores::argmax`` of the whole matrix from the resident shard ``scores``."""
        from . import _ffi
        C = self._C
        found, best, value = C.c_int(0), _ffi.Coords(), C.c_float(0)
        _ffi.check(self._L.lm_hip_argmax_sharded(self._pli._h, self._h, scores._h, row_offset, C.byref(found),
                                                 C.byref(best), C.byref(value)))
        return ((best.row, best.col), float(value.value)) if found.value else None

    def argmax_sharded_begin(self, scores, row_offset: int) -> int:
        """Enqueues the merge of ``scores``' argmax and returns a ticket at once; the shard may be
        overwritten by the next ``score_into`` (the merge overlaps it).  At most two in flight."""
        from . import _ffi
        ticket = self._C.c_int(-1)
        _ffi.check(self._L.lm_hip_argmax_sharded_begin(self._pli._h, self._h, scores._h, row_offset,
                                                       self._C.byref(ticket)))
        return ticket.value

    def argmax_sharded_end(self, ticket: int):
        """Result of the merge enqueued under ``ticket``: same value as ``argmax_sharded``."""
        from . import _ffi
        C = self._C
        found, best, value = C.c_int(0), _ffi.Coords(), C.c_float(0)
        _ffi.check(self._L.lm_hip_argmax_sharded_end(self._pli._h, self._h, ticket, C.byref(found), C.byref(best),
                                                     C.byref(value)))
        return ((best.row, best.col), float(value.value)) if found.value else None

    def merge_max(self, local: Optional[float]) -> Optional[float]:
        from . import _ffi
        C = self._C
        found, value = C.c_int(0), C.c_float(0)
        _ffi.check(self._L.lm_hip_merge_max(self._pli._h, self._h, int(local is not None),
                                            0.0 if local is None else local, C.byref(found), C.byref(value)))
        return float(value.value) if found.value else None

    def merge_threshold(self, local_coords, row_offset: int) -> np.ndarray:
        from . import _ffi
        C = self._C
        mine = np.ascontiguousarray(np.asarray(local_coords, dtype=np.uint64).reshape(-1, 2))
        ptr, n = C.POINTER(_ffi.Coords)(), C.c_size_t(0)
        _ffi.check(self._L.lm_hip_merge_threshold(self._pli._h, self._h,
                                                  mine.ctypes.data_as(C.POINTER(_ffi.Coords)), mine.shape[0],
                                                  row_offset, C.byref(ptr), C.byref(n)))
        return self._pli._take_coords_array(ptr, n.value)


def combine_argmax_cabi(records):
    """``combine_argmax`` through the C ABI's host-side rule (``lm_hip_combine_argmax``): the
    restatement a non-Python host links against; the tests hold the two against each other."""
    import ctypes as C
    from . import _ffi
    L = _ffi.lib()
    n = len(records)
    found = (C.c_int * n)()
    best = (_ffi.Coords * n)()
    value = (C.c_float * n)()
    for i, rec in enumerate(records):
        if rec is not None:
            found[i] = 1
            (best[i].row, best[i].col), value[i] = rec
    fo, bo, vo = C.c_int(0), _ffi.Coords(), C.c_float(0)
    _ffi.check(L.lm_hip_combine_argmax(found, best, value, n, C.byref(fo), C.byref(bo), C.byref(vo)))
    return ((bo.row, bo.col), float(vo.value)) if fo.value else None
