"""Sharding of the scoring path across the GPUs of one node.

The path shards embarrassingly: output row ``r`` depends only on input rows
``r .. r+M-1`` (pli/mod.rs:99-101), which is why ``score_rows_into`` takes a row
range (pli/mod.rs:72-78).  One process per GPU owns a contiguous range of the
``R`` striped rows plus an ``M-1``-row halo; scoring needs NO collective.  RCCL
(``torch.distributed`` backend "nccl") is used only for

* the one-off halo hand-over at set-up (each rank receives the first ``M-1`` rows
  of its successor; the last rank receives rank 0's rows to build the
  reference's wrap rows, seq.rs:373-378), and
* the final merge: ``all_gather`` of one 24-byte ``(score, row, col)`` record per
  rank for argmax, ``all_gather`` of hit counts + padded hit lists for threshold.

Payloads are bytes to kilobytes, so xGMI bandwidth is irrelevant; latency is all
that matters.  The same code runs on CPU tensors with the ``gloo`` backend, which
is how the tests cover world_size > 1 without GPUs.
"""
from __future__ import annotations

import struct
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_rows(total_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row range ``[a, b)`` of rank ``rank``."""
    a = total_rows * rank // world_size
    b = total_rows * (rank + 1) // world_size
    return a, b


def shard_motifs(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Partition a motif list over ranks balancing sum(M) (cost is proportional to
    the motif length): longest-processing-time greedy.  Returns motif indices per rank."""
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


def merge_argmax(local: Optional[Tuple[Tuple[int, int], float]], row_offset: int,
                 device: torch.device | str = "cpu", group=None):
    """Global ``Maximum::argmax`` from per-shard results.

    ``local`` is ``((row, col), value)`` of this rank's shard (rows relative to the
    shard) computed WITHOUT the first-cell rule on ranks > 0, or ``None`` if the
    shard is empty; ``row_offset`` is the shard's first global row.  Every rank
    returns the same ``((row, col), value)`` in global coordinates (or ``None``).

    Rule (pli/mod.rs:135-155): maximal score; ties go to the LAST cell in
    (row, col) order; NaN never wins, except that a NaN in the matrix's very first
    cell -- reported by rank 0 through the first-cell rule -- wins outright.
    """
    rank, world = _world(group)
    rec = torch.zeros(4, dtype=torch.int64)
    if local is not None:
        (r, c), v = local
        rec[0], rec[1], rec[2], rec[3] = 1, _f32_bits(v), r + row_offset, c
    if world == 1:
        recs = [rec]
    else:
        rec = rec.to(device)
        bufs = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(bufs, rec, group=group)
        recs = [b.cpu() for b in bufs]
    best = None
    for i, b in enumerate(recs):
        if int(b[0]) == 0:
            continue
        v, r, c = _bits_f32(int(b[1])), int(b[2]), int(b[3])
        if v != v:          # NaN can only be the first-cell rule of the shard holding row 0
            if r == 0 and c == 0:
                return (0, 0), v
            continue
        if best is None or v > best[1] or (v == best[1] and (r, c) > best[0]):
            best = ((r, c), v)
    return best


def merge_threshold(local_coords: Sequence[Tuple[int, int]], row_offset: int,
                    device: torch.device | str = "cpu", group=None) -> List[Tuple[int, int]]:
    """Global ``Threshold::threshold`` list: shards hold ascending contiguous row
    ranges, so concatenating the per-shard row-major lists in rank order IS the
    reference's row-major order (pli/mod.rs:212-218)."""
    rank, world = _world(group)
    import numpy as np
    mine = torch.from_numpy(np.asarray(local_coords, dtype=np.int64).reshape(-1, 2).copy())
    mine[:, 0] += row_offset
    if world == 1:
        return [(int(r), int(c)) for r, c in mine.tolist()]
    n = torch.tensor([mine.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(x.item()) for x in counts]
    cap = max(max(counts), 1)
    padded = torch.zeros((cap, 2), dtype=torch.int64, device=device)
    padded[:mine.shape[0]] = mine.to(device)
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    out: List[Tuple[int, int]] = []
    for cnt, buf in zip(counts, bufs):
        out.extend((int(r), int(c)) for r, c in buf[:cnt].cpu().tolist())
    return out


def merge_max(local: Optional[float], device: torch.device | str = "cpu", group=None) -> Optional[float]:
    """``Maximum::max`` across shards (value of the merged argmax when no NaN is involved)."""
    rank, world = _world(group)
    t = torch.tensor([float("-inf") if local is None else local, 0.0 if local is None else 1.0],
                     dtype=torch.float32, device=device)
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
        wrapped = torch.zeros_like(recv)
        wrapped[:, :columns - 1] = recv[:, 1:columns]
        wrapped[:, columns - 1] = default_symbol
        recv = wrapped
    shard[rows:] = recv.to(shard.device)
    return shard
