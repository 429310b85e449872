"""Host-side mirror of lightmotif's scoring interface on top of the HIP C ABI.

Two layers, both thin:

* :class:`Pipeline` -- the reference's trait surface (``Encode`` / ``Stripe`` /
  ``Score`` / ``Maximum`` / ``Threshold``, lightmotif/src/pli/mod.rs:34-222) for the
  ``Hip`` back-end: same method names, argument meaning and error behaviour
  (misuse raises where the reference panics; degenerate input gives empty results).
* ``EncodedSequence`` / ``StripedSequence`` / ``CountMatrix`` / ``WeightMatrix`` /
  ``ScoringMatrix`` / ``StripedScores`` / ``create`` / ``stripe`` / ``scan`` -- the
  user-facing objects of the reference's Python module
  (lightmotif-py/lightmotif/lib.rs, lib.pyi) so its tests read the same here.

All sequence and score data live in device memory; every scoring operation runs
in the hand-written gfx950 kernels.  Nothing here falls back to the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _ffi
from ._ffi import Coords, InvalidSymbol, LightmotifHipError, UnsupportedBackend, check

__all__ = [
    "pack_2bit",
    "Pipeline", "EncodedSequence", "StripedSequence", "CountMatrix", "WeightMatrix",
    "ScoringMatrix", "DiscreteMatrix", "StripedScores", "Scanner", "Hit", "Motif", "create", "stripe", "scan",
    "UnsupportedBackend", "InvalidSymbol", "LightmotifHipError", "DEFAULT_COLUMNS",
]

DNA_SYMBOLS = "ACTGN"                     # abc.rs:106-108
PROTEIN_SYMBOLS = "ACDEFGHIKLMNPQRSTVWYX"  # abc.rs:193-256
DEFAULT_COLUMNS = 32                      # dispatch.rs:45 / dense.rs:17 on x86-64


def _symbols(protein: bool) -> str:
    return PROTEIN_SYMBOLS if protein else DNA_SYMBOLS


def _k(protein: bool) -> int:
    return len(_symbols(protein))


def stride(cols: int, elem_size: int) -> int:
    """DenseMatrix::stride (dense.rs:126-128)."""
    return int(_ffi.lib().lm_hip_stride(cols, elem_size))


# --- Pipeline -----------------------------------------------------------------


class _HostBlock:
    """A malloc'ed block returned by the library, exposed to numpy through the array
    interface and released with ``lm_hip_free`` when the last array on it dies."""

    def __init__(self, L, addr: int, nbytes: int):
        self._L, self._addr = L, addr
        self.__array_interface__ = {"data": (addr, False), "shape": (nbytes,), "typestr": "|u1",
                                    "version": 3}

    def __del__(self):
        try:
            self._L.lm_hip_free(C.c_void_p(self._addr))
        except Exception:  # interpreter shutdown
            pass


class Pipeline:
    """``Pipeline<A, Hip>``: owns a device context (stream + scratch)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        L = _ffi.lib()
        h = C.c_void_p()
        if stream is None:
            check(L.lm_hip_ctx_create(device, C.byref(h)))
        else:
            check(L.lm_hip_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h)))
        self._L = L
        self._h = h
        self.device = device

    # Pipeline::avx2()/neon() return Result<_, UnsupportedBackend> (pli/mod.rs:401-407)
    @classmethod
    def hip(cls, device: int = 0, stream: Optional[int] = None) -> "Pipeline":
        return cls(device, stream)

    @staticmethod
    def device_count() -> int:
        n = C.c_int(0)
        check(_ffi.lib().lm_hip_device_count(C.byref(n)))
        return n.value

    @staticmethod
    def device_ordinals() -> List[int]:
        """HIP ordinals of the usable (gfx950) devices -- what ``Pipeline.hip(device)`` takes;
        on a node with other GPUs in between this is not ``range(device_count())``."""
        out, o = [], C.c_int(0)
        for i in range(Pipeline.device_count()):
            check(_ffi.lib().lm_hip_device_ordinal(i, C.byref(o)))
            out.append(o.value)
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.lm_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self) -> None:
        check(self._L.lm_hip_ctx_sync(self._h))

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        check(self._L.lm_hip_ctx_stream(self._h, C.byref(s)))
        return s.value or 0

    def set_rows_per_stream(self, rows: int) -> None:
        check(self._L.lm_hip_ctx_set_rows_per_stream(self._h, rows))

    def set_prefilter(self, enabled: bool) -> None:
        check(self._L.lm_hip_ctx_set_prefilter(self._h, int(enabled)))

    def set_track_argmax(self, enabled: bool) -> None:
        """``score_into`` on large matrices also tracks the maximum so that ``argmax`` on the
        same scores is free (default on)."""
        check(self._L.lm_hip_ctx_set_track_argmax(self._h, int(enabled)))

    def set_option(self, name: str, value: float) -> None:
        """An alternative path inside the library with identical results (``lm_hip_ctx_set_option``): for tests and A/B runs."""
        check(self._L.lm_hip_ctx_set_option(self._h, name.encode(), float(value)))

    @property
    def last_kernel(self) -> str:
        return self._L.lm_hip_ctx_last_kernel(self._h).decode()

    @property
    def last_scan_counts(self) -> Tuple[int, int]:
        """(hits, candidate pieces) of the last fused threshold scan on this pipeline (lm_hip_ctx_last_scan_counts)."""
        h, c = C.c_ulonglong(0), C.c_ulonglong(0)
        check(self._L.lm_hip_ctx_last_scan_counts(self._h, C.byref(h), C.byref(c)))
        return int(h.value), int(c.value)

    @property
    def last_phases_ms(self) -> Optional[Tuple[float, float, float, float]]:
        """(scan, re-scoring, ordering, host share) of the last fused threshold call in ms, with ``set_option("time_scan", 1)``;
        None otherwise (lm_hip_ctx_last_phases_ms)."""
        ph = (C.c_float * 4)()
        check(self._L.lm_hip_ctx_last_phases_ms(self._h, ph))
        return tuple(float(x) for x in ph) if ph[0] >= 0 else None

    @property
    def last_scan_info(self) -> Tuple[int, int]:
        """(motif rows scanned, LDS table bytes per position) of the last single-job fused scan on this pipeline; zeros
        when no prefilter / exact scan kernel ran (lm_hip_ctx_last_scan_info)."""
        r, b = C.c_size_t(0), C.c_size_t(0)
        check(self._L.lm_hip_ctx_last_scan_info(self._h, C.byref(r), C.byref(b)))
        return int(r.value), int(b.value)

    @property
    def last_scan_kernel_ms(self) -> Optional[float]:
        """Duration of the scan kernel(s) of the last fused call, with ``set_option("time_scan", 1)``; None otherwise
        (lm_hip_ctx_last_scan_kernel_ms)."""
        ms = C.c_float(-1.0)
        check(self._L.lm_hip_ctx_last_scan_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value) if ms.value >= 0 else None

    def sustained_clock_mhz(self, load, seconds: float = 0.25, window_us: int = 3000):
        """Shader clock (MHz, median over windows) the device sustains while `load()` is called back to back for
        `seconds` from this thread; a second thread runs lm_hip_device_clock_mhz windows beside it.  Returns
        (mhz or None, ms per call with the probe running, ms per call without)."""
        import threading
        import time
        dev = C.c_int(self.device)

        def run(duration):
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < duration:
                load()
                n += 1
            self.sync()
            return (time.perf_counter() - t0) / max(n, 1) * 1e3
        run(0.05)
        alone = run(seconds / 2)
        samples, stop = [], threading.Event()

        def sampler():
            mhz = C.c_double(0.0)
            while not stop.is_set():
                if self._L.lm_hip_device_clock_mhz(dev.value, window_us, C.byref(mhz)) == 0:
                    samples.append(mhz.value)
        th = threading.Thread(target=sampler)
        th.start()
        beside = run(seconds)
        stop.set()
        th.join()
        s = sorted(samples[1:-1] if len(samples) > 4 else samples)
        return (s[len(s) // 2] if s else None), beside, alone

    # -- Encode / Stripe (pli/mod.rs:34-67, 164-201) ---------------------------

    def encode(self, sequence: Union[str, bytes], protein: bool = False) -> "EncodedSequence":
        return EncodedSequence(sequence, protein=protein)

    def stripe(self, encoded: "EncodedSequence", columns: int = DEFAULT_COLUMNS) -> "StripedSequence":
        h = C.c_void_p()
        data = np.ascontiguousarray(encoded.data, dtype=np.uint8)
        check(self._L.lm_hip_seq_from_encoded(self._h, data.ctypes.data, data.size, columns,
                                              _k(encoded.protein), C.byref(h)))
        return StripedSequence(self, h, encoded.protein)

    def stripe_2bit(self, packed: np.ndarray, length: int, n_mask: Optional[np.ndarray] = None,
                    columns: int = DEFAULT_COLUMNS, n_runs: Optional[np.ndarray] = None) -> "StripedSequence":
        """A DNA sequence held 4 bases per byte (``pack_2bit``) -> resident StripedSequence: a quarter of the
        bytes over PCIe, unpacked straight into the striped matrix (``lm_hip_seq_from_2bit``).  N positions as an
        ``(n, 2)`` array of runs ``(start, size)`` -- the .2bit container's form -- and / or as a bit mask."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        if packed.size < (length + 3) // 4:
            raise ValueError("packed buffer shorter than length / 4")
        mptr = rptr = None
        nruns = 0
        if n_mask is not None:
            n_mask = np.ascontiguousarray(n_mask, dtype=np.uint8)
            if n_mask.size < (length + 7) // 8:
                raise ValueError("N mask shorter than length / 8")
            mptr = n_mask.ctypes.data
        if n_runs is not None and len(n_runs):
            n_runs = np.ascontiguousarray(np.asarray(n_runs, dtype=np.uint64).reshape(-1, 2))
            rptr, nruns = n_runs.ctypes.data, n_runs.shape[0]
        h = C.c_void_p()
        check(self._L.lm_hip_seq_from_2bit(self._h, packed.ctypes.data, mptr, rptr, nruns, length, columns, C.byref(h)))
        return StripedSequence(self, h, False)

    def stripe_ascii(self, sequence: Union[str, bytes, np.ndarray], protein: bool = False, lossy: bool = False,
                     columns: int = DEFAULT_COLUMNS) -> "StripedSequence":
        """encode (+ encode_lossy) and stripe entirely on the device."""
        if isinstance(sequence, np.ndarray):     # a genome already in memory as bytes: no copy
            raw = buf = np.ascontiguousarray(sequence, dtype=np.uint8)
        else:
            raw = sequence.encode("ascii", "replace") if isinstance(sequence, str) else bytes(sequence)
            buf = np.frombuffer(raw, dtype=np.uint8)
        h = C.c_void_p()
        bad = C.c_size_t(0)
        st = self._L.lm_hip_seq_from_ascii(self._h, b"P" if protein else b"D", buf.ctypes.data,
                                           buf.size, columns, int(lossy), C.byref(h), C.byref(bad))
        if st == _ffi.ERR_INVALID_SYMBOL:
            raise InvalidSymbol(f"Invalid symbol in sequence: {chr(int(raw[bad.value]))!r}")
        check(st)
        return StripedSequence(self, h, protein)

    def upload(self, data: np.ndarray, length: int, wrap: int, columns: int,
               protein: bool = False) -> "StripedSequence":
        """Adopt an already striped host matrix ((rows+wrap) x stride u8)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        h = C.c_void_p()
        check(self._L.lm_hip_seq_upload(self._h, data.ctypes.data, data.shape[0], data.shape[1],
                                        columns, wrap, length, _k(protein), C.byref(h)))
        return StripedSequence(self, h, protein)

    def adopt_sequence(self, data_ptr: int, rows: int, wrap: int, columns: int, stride_: int,
                       length: int, protein: bool = False, keepalive=None,
                       capacity_rows: Optional[int] = None) -> "StripedSequence":
        """A StripedSequence over a striped matrix that already lives on the device and stays
        the caller's (``(rows + wrap) x stride`` bytes at ``data_ptr``, e.g. a torch tensor or
        one row shard of a multi-GPU job): nothing is copied.  ``capacity_rows`` = rows the
        buffer has room for (configure_wrap may grow into them).  ``keepalive`` is held by the
        returned object so the buffer outlives it."""
        h = C.c_void_p()
        check(self._L.lm_hip_seq_adopt_dptr(self._h, C.c_void_p(data_ptr), rows + wrap,
                                            max(capacity_rows or 0, rows + wrap), stride_, columns,
                                            wrap, length, _k(protein), C.byref(h)))
        seq = StripedSequence(self, h, protein)
        seq._keepalive = keepalive
        return seq

    # -- Score (pli/mod.rs:69-130) ----------------------------------------------

    def score_rows_into(self, pssm: "ScoringMatrix", seq: "StripedSequence", rows: range,
                        scores: "StripedScores") -> None:
        check(self._L.lm_hip_score_rows_into(self._h, pssm._device(self), seq._h,
                                             rows.start, max(rows.stop, rows.start), scores._h))

    def score_into(self, pssm: "ScoringMatrix", seq: "StripedSequence",
                   scores: "StripedScores") -> None:
        check(self._L.lm_hip_score_into(self._h, pssm._device(self), seq._h, scores._h))

    def score(self, pssm: "ScoringMatrix", seq: "StripedSequence") -> "StripedScores":
        scores = StripedScores.empty(self, seq.columns)
        self.score_into(pssm, seq, scores)
        return scores

    # -- Maximum / Threshold (pli/mod.rs:132-161, 203-222) ------------------------

    def argmax(self, scores: "StripedScores") -> Optional[Tuple[int, int]]:
        found, best, _ = self._argmax(scores)
        return (best.row, best.col) if found else None

    def max(self, scores: "StripedScores") -> Optional[float]:
        """``Maximum::max`` (pli/mod.rs:158-160) through its own export, ``lm_hip_max``."""
        found, value = C.c_int(0), C.c_float(0)
        check(self._L.lm_hip_max(self._h, scores._h, C.byref(found), C.byref(value)))
        return float(value.value) if found.value else None

    def _argmax(self, scores: "StripedScores"):
        found, best, value = C.c_int(0), Coords(), C.c_float(0)
        check(self._L.lm_hip_argmax(self._h, scores._h, C.byref(found), C.byref(best), C.byref(value)))
        return bool(found.value), best, float(value.value)

    def argmax_handle_shard(self, scores: "StripedScores", first_cell_rule: bool):
        """``((row, col), value)`` of one row shard held in a handle (rows relative to the
        shard), with the first-cell rule applied only when the shard holds row 0."""
        scores.set_first_cell_rule(first_cell_rule)
        found, best, value = self._argmax(scores)
        return ((best.row, best.col), value) if found else None

    def threshold(self, scores: "StripedScores", threshold: float) -> List[Tuple[int, int]]:
        ptr, n = C.POINTER(Coords)(), C.c_size_t(0)
        check(self._L.lm_hip_threshold(self._h, scores._h, threshold, C.byref(ptr), C.byref(n)))
        return self._take_coords(ptr, n.value)

    def _take_coords(self, ptr, n: int) -> List[Tuple[int, int]]:
        try:
            if n == 0:
                return []
            arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_size_t)), shape=(n, 2))
            return [(int(r), int(c)) for r, c in arr]
        finally:
            if ptr:
                self._L.lm_hip_free(ptr)

    # -- fused forms ----------------------------------------------------------------

    def score_argmax(self, pssm: "ScoringMatrix", seq: "StripedSequence",
                     rows: Optional[range] = None):
        """score_rows_into + argmax without writing the scores: ((row, col), value) or None."""
        length, wrap, nrows, stride_, columns, data_ptr = seq._info()    # one call: a small scan is latency-bound
        rows = range(0, nrows) if rows is None else rows
        found, best, value = C.c_int(0), Coords(), C.c_float(0)
        check(self._L.lm_hip_score_argmax_f32_dptr(
            self._h, pssm._device(self), data_ptr, nrows + wrap, stride_, columns,
            wrap, length, rows.start, max(rows.stop, rows.start), C.byref(found),
            C.byref(best), C.byref(value)))
        return ((best.row, best.col), float(value.value)) if found.value else None

    def score_threshold(self, pssm: "ScoringMatrix", seq: "StripedSequence", threshold: float,
                        rows: Optional[range] = None):
        """score_rows_into + threshold without writing the scores: ([(row, col)], [value])."""
        length, wrap, nrows, stride_, columns, data_ptr = seq._info()
        rows = range(0, nrows) if rows is None else rows
        ptr, vals, n = C.POINTER(Coords)(), C.POINTER(C.c_float)(), C.c_size_t(0)
        check(self._L.lm_hip_score_threshold_f32_dptr(
            self._h, pssm._device(self), data_ptr, nrows + wrap, stride_, columns,
            wrap, length, rows.start, max(rows.stop, rows.start), threshold,
            C.byref(ptr), C.byref(vals), C.byref(n)))
        try:
            values = [float(vals[i]) for i in range(n.value)]
        finally:
            if vals:
                self._L.lm_hip_free(vals)
        return self._take_coords(ptr, n.value), values

    # -- many motifs x one resident sequence (the CLI's job fan-out, main.rs:554-561) ----

    def scan_argmax_batch(self, pssms: Sequence["ScoringMatrix"], seq: "StripedSequence"):
        """Per motif: ``((row, col), value)`` of the best cell, or ``None`` (L < M)."""
        n = len(pssms)
        handles = (C.c_void_p * n)(*[p._device(self) for p in pssms])
        found = (C.c_int * n)()
        best = (Coords * n)()
        value = (C.c_float * n)()
        check(self._L.lm_hip_scan_argmax_batch(self._h, handles, n, seq._h, found, best, value))
        return [((best[i].row, best[i].col), float(value[i])) if found[i] else None for i in range(n)]

    def prepare_batch(self, pssms: Sequence["ScoringMatrix"], thresholds: Optional[Sequence[float]] = None) -> "MotifBatch":
        """The motif list of a batch in the form the C ABI takes it (device handles, thresholds), built once: a job loop that
        scans sequence after sequence with the same motifs (the CLI, main.rs:502-561) does not rebuild it per call."""
        return MotifBatch(self, pssms, thresholds)

    def scan_threshold_batch(self, pssms: Union[Sequence["ScoringMatrix"], "MotifBatch"],
                             thresholds: Optional[Sequence[float]], seq: "StripedSequence") -> "BatchHits":
        """Per motif: ``(coords (n_i, 2) int64 in row-major order, values (n_i,) f32)`` -- a sequence of such pairs (views
        into ONE pair of arrays the library returned, cut on access)."""
        batch = pssms if isinstance(pssms, MotifBatch) else MotifBatch(self, pssms, thresholds)
        n = len(batch)
        counts = np.zeros(n, dtype=np.uintp)
        ptr, vals = C.POINTER(Coords)(), C.POINTER(C.c_float)()
        check(self._L.lm_hip_scan_threshold_batch(self._h, batch.handles, batch.thresholds, n, seq._h,
                                                  counts.ctypes.data_as(C.POINTER(C.c_size_t)), C.byref(ptr), C.byref(vals)))
        total = int(counts.sum())
        values = self._take_array(vals, total, np.float32)
        coords = self._take_coords_array(ptr, total)
        return BatchHits(coords, values, counts)

    # -- raw device-pointer forms (used with torch tensors by bench.py / tests) -------

    def score_dptr(self, pssm: "ScoringMatrix", seq_ptr: int, seq_rows_total: int, seq_stride: int,
                   columns: int, wrap: int, length: int, row_begin: int, row_end: int,
                   out_ptr: int, out_stride: int) -> Tuple[int, int]:
        orow, mi = C.c_size_t(0), C.c_size_t(0)
        check(self._L.lm_hip_score_f32_dptr(self._h, pssm._device(self), C.c_void_p(seq_ptr),
                                            seq_rows_total, seq_stride, columns, wrap, length,
                                            row_begin, row_end, C.c_void_p(out_ptr), out_stride,
                                            C.byref(orow), C.byref(mi)))
        return orow.value, mi.value

    def argmax_dptr(self, scores_ptr: int, rows: int, stride_: int, columns: int,
                    first_cell_rule: bool = True):
        found, best, value = C.c_int(0), Coords(), C.c_float(0)
        check(self._L.lm_hip_argmax_shard_f32_dptr(self._h, C.c_void_p(scores_ptr), rows, stride_,
                                                   columns, int(first_cell_rule), C.byref(found),
                                                   C.byref(best), C.byref(value)))
        return ((best.row, best.col), float(value.value)) if found.value else None

    def _take_array(self, ptr, count: int, dtype) -> np.ndarray:
        """Wraps a malloc'ed result of the library WITHOUT copying it; the block is
        handed back to ``lm_hip_free`` when the last view of the array dies."""
        addr = C.cast(ptr, C.c_void_p).value
        if not addr or count == 0:
            if addr:
                self._L.lm_hip_free(C.c_void_p(addr))
            return np.zeros(0, dtype=dtype)
        block = _HostBlock(self._L, addr, count * np.dtype(dtype).itemsize)
        return np.asarray(block).view(dtype)

    def _take_coords_array(self, ptr, n: int) -> np.ndarray:
        # lm_hip_coords = {size_t row, col}; indices stay far below 2^63
        return self._take_array(ptr, 2 * n, np.int64).reshape(n, 2)

    def threshold_dptr(self, scores_ptr: int, rows: int, stride_: int, columns: int,
                       threshold: float) -> np.ndarray:
        """(n, 2) int64 array of (row, col) in the reference's row-major order."""
        ptr, n = C.POINTER(Coords)(), C.c_size_t(0)
        check(self._L.lm_hip_threshold_f32_dptr(self._h, C.c_void_p(scores_ptr), rows, stride_,
                                                columns, threshold, C.byref(ptr), C.byref(n)))
        return self._take_coords_array(ptr, n.value)

    # -- Score / Maximum / Threshold on u8: a DiscreteMatrix's scores ----------------------------

    def score_discrete(self, dm: "DiscreteMatrix", seq: "StripedSequence", rows: Optional[range] = None,
                       saturate: bool = True) -> Tuple[np.ndarray, int]:
        """``Score<u8, ..>::score_rows_into(&dm, &seq, rows, &mut scores)`` (pli/mod.rs:72-106):
        the u8 score matrix ``(rows, stride(C, 1))`` on the host and ``max_index``.
        ``saturate``: the SIMD back-ends' saturating adds (avx2.rs:336) or Generic's wrapping."""
        rows = range(0, seq.rows) if rows is None else rows
        n = max(rows.stop - rows.start, 0)
        st = stride(seq.columns, 1)
        out = np.zeros((n, st), dtype=np.uint8)
        orow, mi = C.c_size_t(0), C.c_size_t(0)
        check(self._L.lm_hip_score_u8(self._h, dm.data.ctypes.data, len(dm), dm.data.shape[1], dm.k,
                                      seq._h, rows.start, max(rows.stop, rows.start), int(saturate),
                                      out.ctypes.data, st, C.byref(orow), C.byref(mi)))
        return out[:orow.value], int(mi.value)

    def score_u8_dptr(self, dm: "DiscreteMatrix", seq_ptr: int, seq_rows_total: int, seq_stride: int,
                      columns: int, wrap: int, length: int, row_begin: int, row_end: int, out_ptr: int,
                      out_stride: int, saturate: bool = True) -> Tuple[int, int]:
        orow, mi = C.c_size_t(0), C.c_size_t(0)
        check(self._L.lm_hip_score_u8_dptr(self._h, dm.data.ctypes.data, len(dm), dm.data.shape[1], dm.k,
                                           C.c_void_p(seq_ptr), seq_rows_total, seq_stride, columns, wrap,
                                           length, row_begin, row_end, C.c_void_p(out_ptr), out_stride,
                                           int(saturate), C.byref(orow), C.byref(mi)))
        return int(orow.value), int(mi.value)

    def argmax_u8_dptr(self, scores_ptr: int, rows: int, stride_: int, columns: int):
        """``Maximum<u8, C>`` (pli/mod.rs:135-160): ``((row, col), value)`` or ``None``."""
        found, best, value = C.c_int(0), Coords(), C.c_uint8(0)
        check(self._L.lm_hip_argmax_u8_dptr(self._h, C.c_void_p(scores_ptr), rows, stride_, columns,
                                            C.byref(found), C.byref(best), C.byref(value)))
        return ((best.row, best.col), int(value.value)) if found.value else None

    def threshold_u8_dptr(self, scores_ptr: int, rows: int, stride_: int, columns: int, t: int) -> np.ndarray:
        """``Threshold<u8, C>`` (pli/mod.rs:210-221): (n, 2) int64 (row, col) in row-major order."""
        ptr, n = C.POINTER(Coords)(), C.c_size_t(0)
        check(self._L.lm_hip_threshold_u8_dptr(self._h, C.c_void_p(scores_ptr), rows, stride_, columns,
                                               C.c_uint8(t), C.byref(ptr), C.byref(n)))
        return self._take_coords_array(ptr, n.value)

    def score_threshold_dptr(self, pssm: "ScoringMatrix", seq_ptr: int, seq_rows_total: int,
                             seq_stride: int, columns: int, wrap: int, length: int,
                             row_begin: int, row_end: int, threshold: float):
        """Fused form: ((n, 2) int64 coords, (n,) f32 values), rows relative to row_begin."""
        ptr, vals, n = C.POINTER(Coords)(), C.POINTER(C.c_float)(), C.c_size_t(0)
        check(self._L.lm_hip_score_threshold_f32_dptr(
            self._h, pssm._device(self), C.c_void_p(seq_ptr), seq_rows_total, seq_stride, columns,
            wrap, length, row_begin, row_end, threshold, C.byref(ptr), C.byref(vals), C.byref(n)))
        return self._take_coords_array(ptr, n.value), self._take_array(vals, n.value, np.float32)

    def score_argmax_dptr(self, pssm: "ScoringMatrix", seq_ptr: int, seq_rows_total: int,
                          seq_stride: int, columns: int, wrap: int, length: int, row_begin: int,
                          row_end: int, first_cell_rule: bool = True):
        found, best, value = C.c_int(0), Coords(), C.c_float(0)
        check(self._L.lm_hip_score_argmax_shard_f32_dptr(
            self._h, pssm._device(self), C.c_void_p(seq_ptr), seq_rows_total, seq_stride, columns,
            wrap, length, row_begin, row_end, int(first_cell_rule), C.byref(found), C.byref(best),
            C.byref(value)))
        return ((best.row, best.col), float(value.value)) if found.value else None

    def stripe_dptr(self, encoded_ptr: int, length: int, columns: int, default_symbol: int,
                    wrap: int, data_ptr: int, stride_: int) -> None:
        check(self._L.lm_hip_stripe_dptr(self._h, C.c_void_p(encoded_ptr), length, columns,
                                         default_symbol, wrap, C.c_void_p(data_ptr), stride_))

    def configure_wrap_dptr(self, data_ptr: int, rows: int, stride_: int, columns: int,
                            new_wrap: int, default_symbol: int) -> None:
        check(self._L.lm_hip_configure_wrap_dptr(self._h, C.c_void_p(data_ptr), rows, stride_,
                                                 columns, new_wrap, default_symbol))

    def encode_dptr(self, ascii_ptr: int, length: int, dst_ptr: int, protein: bool = False,
                    lossy: bool = False) -> None:
        bad = C.c_size_t(0)
        st = self._L.lm_hip_encode_dptr(self._h, b"P" if protein else b"D", C.c_void_p(ascii_ptr),
                                        length, int(lossy), C.c_void_p(dst_ptr), C.byref(bad))
        if st == _ffi.ERR_INVALID_SYMBOL:
            raise InvalidSymbol(f"Invalid symbol in sequence at position {bad.value}")
        check(st)


_default: Optional[Pipeline] = None


def default_pipeline() -> Pipeline:
    """The analogue of ``Pipeline::dispatch()`` (pli/mod.rs:269-308): the reference
    re-creates a zero-sized pipeline per call (pwm/mod.rs:646, scores.rs:182); the
    GPU context it needs is cached here instead."""
    global _default
    if _default is None:
        _default = Pipeline.hip()
    return _default


# --- sequences ----------------------------------------------------------------------


class EncodedSequence:
    """seq.rs:83-98: a vector of symbol indices (host side; 1 byte per symbol)."""

    def __init__(self, sequence: Union[str, bytes, np.ndarray], *, protein: bool = False,
                 lossy: bool = False):
        self.protein = protein
        if isinstance(sequence, np.ndarray):
            self.data = np.ascontiguousarray(sequence, dtype=np.uint8)
            return
        raw = sequence.encode("utf-8") if isinstance(sequence, str) else bytes(sequence)
        lut = np.full(256, 255, dtype=np.uint8)
        for i, ch in enumerate(_symbols(protein)):
            lut[ord(ch)] = i
        data = lut[np.frombuffer(raw, dtype=np.uint8)]
        bad = np.flatnonzero(data == 255)
        if bad.size:
            if not lossy:  # pli/mod.rs:63 -> Err(InvalidSymbol)
                raise InvalidSymbol(f"Invalid symbol in sequence: {chr(raw[int(bad[0])])!r}")
            data[bad] = _k(protein) - 1  # seq.rs:126 unwrap_or_default
        self.data = data

    @classmethod
    def encode_lossy(cls, sequence: Union[str, bytes], *, protein: bool = False) -> "EncodedSequence":
        return cls(sequence, protein=protein, lossy=True)

    def __len__(self) -> int:
        return int(self.data.size)

    def __getitem__(self, index: int) -> int:
        """Symbol index at ``index`` (lightmotif-py lib.rs ``EncodedSequence.__getitem__``: negative
        indices count from the end, out of range raises IndexError)."""
        n = int(self.data.size)
        i = index + n if index < 0 else index
        if not 0 <= i < n:
            raise IndexError("sequence index out of range")
        return int(self.data[i])

    def __iter__(self):
        return (int(x) for x in self.data)

    def __array__(self, dtype=None, copy=None):
        """The buffer the reference exposes through ``memoryview`` (1-D, one byte per symbol)."""
        return self.data if dtype is None else self.data.astype(dtype)

    def __str__(self) -> str:
        sym = _symbols(self.protein)
        return "".join(sym[i] for i in self.data)

    def copy(self) -> "EncodedSequence":
        return EncodedSequence(self.data.copy(), protein=self.protein)

    __copy__ = copy

    def stripe(self, columns: int = DEFAULT_COLUMNS) -> "StripedSequence":
        return default_pipeline().stripe(self, columns)


class StripedSequence:
    """seq.rs:288-294 ``{length, wrap, data}``, resident on the device."""

    def __init__(self, pli: Pipeline, handle: C.c_void_p, protein: bool):
        self._pli, self._h, self.protein = pli, handle, protein

    def __del__(self):
        try:
            if self._h:
                self._pli._L.lm_hip_seq_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _info(self):
        v = [C.c_size_t(0) for _ in range(5)]
        p = C.c_void_p()
        check(self._pli._L.lm_hip_seq_info(self._h, *[C.byref(x) for x in v], C.byref(p)))
        return [x.value for x in v] + [p.value or 0]

    def __len__(self) -> int:
        return self._info()[0]

    @property
    def wrap(self) -> int:
        return self._info()[1]

    @property
    def rows(self) -> int:
        """matrix().rows() - wrap()"""
        return self._info()[2]

    @property
    def stride(self) -> int:
        return self._info()[3]

    @property
    def columns(self) -> int:
        return self._info()[4]

    @property
    def data_ptr(self) -> int:
        return self._info()[5]

    def configure(self, motif: "ScoringMatrix") -> None:
        """seq.rs:362-366"""
        if len(motif) > 0:
            self.configure_wrap(len(motif) - 1)

    def configure_wrap(self, m: int) -> None:
        """seq.rs:369-381"""
        check(self._pli._L.lm_hip_seq_configure_wrap(self._pli._h, self._h, m))

    def matrix(self) -> np.ndarray:
        """Host copy of the (rows + wrap) x stride byte matrix."""
        _, wrap, rows, st, _, _ = self._info()
        out = np.empty((rows + wrap, st), dtype=np.uint8)
        check(self._pli._L.lm_hip_seq_download(self._pli._h, self._h, out.ctypes.data))
        return out

    def copy(self) -> "StripedSequence":
        """lib.rs ``StripedSequence.copy``: an independent sequence with the same rows and wrap rows."""
        length, wrap, _, _, cols, _ = self._info()
        return self._pli.upload(self.matrix(), length, wrap, cols, protein=self.protein)

    __copy__ = copy

    def __array__(self, dtype=None, copy=None):
        """What the reference exposes through ``memoryview(striped)``: shape ``(columns, rows)`` with
        strides ``(1, stride)`` (lib.rs:303-317), i.e. indexed ``[column, row]``; a host copy here, the
        matrix lives on the device."""
        _, _, rows, _, cols, _ = self._info()
        m = self.matrix()[:rows, :cols].T
        return m if dtype is None else m.astype(dtype)


# --- matrices ------------------------------------------------------------------------


def _dict_to_rows(values: Dict[str, Iterable[float]], protein: bool, dtype) -> np.ndarray:
    sym = _symbols(protein)
    k = len(sym)
    length = None
    for key, col in values.items():
        if key not in sym:
            raise ValueError(f"Invalid symbol: {key!r}")
        col = list(col)
        if length is None:
            length = len(col)
        elif length != len(col):
            raise ValueError("Invalid number of rows")
    out = np.zeros((length or 0, k), dtype=dtype)
    for key, col in values.items():
        out[:, sym.index(key)] = np.asarray(list(col), dtype=dtype)
    return out


class CountMatrix:
    """pwm/mod.rs:178-258 (counts are M x K u32)."""

    def __init__(self, values: Union[Dict[str, Iterable[int]], np.ndarray], *, protein: bool = False):
        self.protein = protein
        self.data = (_dict_to_rows(values, protein, np.uint32) if isinstance(values, dict)
                     else np.ascontiguousarray(values, dtype=np.uint32))

    @classmethod
    def from_sequences(cls, sequences: Sequence[EncodedSequence], protein: bool = False) -> "CountMatrix":
        """pwm/mod.rs:209-237"""
        seqs = list(sequences)
        m = len(seqs[0]) if seqs else 0
        data = np.zeros((m, _k(protein)), dtype=np.uint32)
        for s in seqs:
            if len(s) != m:
                raise ValueError("Inconsistent sequence length")
            np.add.at(data, (np.arange(m), s.data), 1)
        return cls(data, protein=protein)

    def __len__(self) -> int:
        return self.data.shape[0]

    def __getitem__(self, i: int) -> List[int]:
        return [int(x) for x in self.data[i]]

    def __eq__(self, other) -> bool:
        return isinstance(other, CountMatrix) and np.array_equal(self.data, other.data)

    def normalize(self, pseudocount: Union[None, float, Dict[str, float]] = None) -> "WeightMatrix":
        """lib.rs:501-526: ``to_freq(pseudo).to_weight(None)`` in f32."""
        k = self.data.shape[1]
        f32 = np.float32
        if pseudocount is None:
            pseudo = np.zeros(k, dtype=f32)
        elif isinstance(pseudocount, dict):
            pseudo = _dict_to_rows({a: [b] for a, b in pseudocount.items()}, self.protein, f32)[0]
        else:  # abc.rs:558-573: every symbol but the default one
            pseudo = np.full(k, f32(pseudocount), dtype=f32)
            pseudo[k - 1] = 0
        bg = uniform_background(k)
        out = np.zeros((len(self), k), dtype=f32)
        for i in range(len(self)):
            row = (self.data[i].astype(f32) + pseudo).astype(f32)  # pwm/mod.rs:249-251
            total = f32(0)
            for x in row:                                            # :252 sequential f32 sum
                total = f32(total + x)
            with np.errstate(divide="ignore", invalid="ignore"):
                row = (row / total).astype(f32)                      # :253-255
                out[i] = np.where(bg == 0, f32(0), row / np.where(bg == 0, f32(1), bg))  # :384-390
        return WeightMatrix(out, bg, protein=self.protein)


def uniform_background(k: int) -> np.ndarray:
    """abc.rs:473-487: 1/(K-1) everywhere, 0 for the default symbol."""
    bg = np.full(k, np.float32(1) / np.float32(k - 1), dtype=np.float32)
    bg[k - 1] = 0
    return bg


class WeightMatrix:
    """pwm/mod.rs:450-456: odds ratios."""

    def __init__(self, data: np.ndarray, background: np.ndarray, *, protein: bool = False):
        self.data = np.ascontiguousarray(data, dtype=np.float32)
        self.background = np.asarray(background, dtype=np.float32)
        self.protein = protein

    def __len__(self) -> int:
        return self.data.shape[0]

    def __getitem__(self, i: int) -> List[float]:
        return [float(x) for x in self.data[i]]

    def __eq__(self, other) -> bool:
        return (isinstance(other, WeightMatrix) and self.protein == other.protein
                and np.array_equal(self.data, other.data, equal_nan=True))

    def log_odds(self, background: Optional[Dict[str, float]] = None, base: float = 2.0) -> "ScoringMatrix":
        """lib.rs WeightMatrix.log_odds -> rescale + to_scoring_with_base (pwm/mod.rs:505-526)."""
        data, bg = self.data, self.background
        if background is not None:
            nb = _dict_to_rows({a: [b] for a, b in background.items()}, self.protein, np.float32)[0]
            with np.errstate(divide="ignore", invalid="ignore"):
                data = (data * (bg / nb)).astype(np.float32)  # pwm/mod.rs:477-481
            bg = nb
        with np.errstate(divide="ignore"):
            if base == 2.0:
                out = np.log2(data, dtype=np.float32)
            elif base == 10.0:
                out = np.log10(data, dtype=np.float32)
            else:
                out = (np.log(data, dtype=np.float32) / np.log(np.float32(base))).astype(np.float32)
        return ScoringMatrix(out, bg, protein=self.protein)


class ScoringMatrix:
    """pwm/mod.rs:561-564 ``{background, data: DenseMatrix<f32, K>}``.

    ``data`` is kept in the reference's padded layout (M x stride(K) f32: stride 8
    for DNA, 24 for protein) so the pointer handed to the C ABI is what a Rust
    caller would hand over."""

    def __init__(self, values: Union[Dict[str, Iterable[float]], np.ndarray],
                 background: Union[None, Dict[str, float], np.ndarray] = None, *,
                 protein: bool = False):
        self.protein = protein
        k = _k(protein)
        dense = (_dict_to_rows(values, protein, np.float32) if isinstance(values, dict)
                 else np.asarray(values, dtype=np.float32))
        if dense.ndim != 2 or dense.shape[1] < k:
            raise ValueError("scoring matrix must be M x K")
        st = stride(k, 4)
        self.data = np.zeros((dense.shape[0], st), dtype=np.float32)  # dense.rs:144-147
        self.data[:, :k] = dense[:, :k]
        if isinstance(background, dict):
            background = _dict_to_rows({a: [b] for a, b in background.items()}, protein, np.float32)[0]
        self.background = uniform_background(k) if background is None else np.asarray(background, np.float32)
        self._dev: Dict[int, C.c_void_p] = {}
        self._plis: Dict[int, Pipeline] = {}

    @property
    def k(self) -> int:
        return _k(self.protein)

    def __len__(self) -> int:
        return self.data.shape[0]

    def __getitem__(self, i: int) -> List[float]:
        """Row ``i``: the K scores of motif position ``i`` (lib.rs ``ScoringMatrix.__getitem__``)."""
        n = self.data.shape[0]
        j = i + n if i < 0 else i
        if not 0 <= j < n:
            raise IndexError("list index out of range")
        return [float(x) for x in self.data[j, :self.k]]

    def __eq__(self, other) -> bool:
        return (isinstance(other, ScoringMatrix) and self.protein == other.protein
                and np.array_equal(self.data, other.data, equal_nan=True))

    def _device(self, pli: Pipeline) -> C.c_void_p:
        key = id(pli)
        if key not in self._dev:
            h = C.c_void_p()
            check(pli._L.lm_hip_pssm_create(pli._h, self.data.ctypes.data, self.data.shape[0],
                                            self.data.shape[1], self.k, C.byref(h)))
            self._dev[key] = h
            self._plis[key] = pli
        return self._dev[key]

    def __del__(self):
        try:
            for key, h in self._dev.items():
                self._plis[key]._L.lm_hip_pssm_destroy(h)
        except Exception:
            pass

    def calculate(self, sequence: StripedSequence) -> "StripedScores":
        """lib.rs:855-874: ``configure(pssm)`` then full-range ``pli.score``."""
        if sequence.protein != self.protein:
            raise ValueError("alphabet mismatch")
        sequence.configure(self)
        return sequence._pli.score(self, sequence)

    def score(self, sequence: StripedSequence) -> "StripedScores":
        """pwm/mod.rs:640-648 (the caller configures the wrap rows, like in Rust)."""
        return sequence._pli.score(self, sequence)

    def _extreme_score(self, pick) -> float:
        total = np.float32(0.0)
        for row in self.data[:, :self.k - 1]:      # the default symbol (N / X) is left out
            total = np.float32(total + pick(row))
        return float(total)

    def min_score(self) -> float:
        """pwm/mod.rs:592-602: sum over positions of the lowest weight (f32, in row order)."""
        return self._extreme_score(np.min)

    def max_score(self) -> float:
        """pwm/mod.rs:605-615: sum over positions of the highest weight."""
        return self._extreme_score(np.max)

    def to_discrete(self) -> "DiscreteMatrix":
        """pwm/mod.rs:665-696: u8 weights that over-estimate the scores, in f32 arithmetic like
        the reference (row offsets = row minima over the K-1 real symbols with -inf counted
        as -max_score, one global factor = (max_score - offset) / 255, weights rounded UP and
        converted with Rust's saturating ``as u8``)."""
        k = self.k
        p = self.data[:, :k]
        max_score = np.float32(self.max_score())
        offsets = np.empty(len(self), np.float32)
        for j in range(len(self)):
            row = np.where(np.isinf(p[j, :k - 1]), np.float32(-max_score), p[j, :k - 1])
            offsets[j] = row.min()
        offset = np.float32(0.0)
        for x in offsets:
            offset = np.float32(offset + x)
        factor = np.float32(np.float32(max_score - offset) / np.float32(255))
        with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
            scaled = np.ceil((p - offsets[:, None]).astype(np.float32) / factor)
            scaled = np.where(np.isnan(scaled), np.float32(0), np.clip(scaled, 0, 255))
        data = np.zeros((len(self), stride(k, 1)), dtype=np.uint8)  # DenseMatrix<u8, K>
        data[:, :k] = scaled.astype(np.uint8)
        return DiscreteMatrix(data, float(factor), offsets, float(offset), protein=self.protein)

    @property
    def score_distribution(self):
        """pwm/mod.rs:698-705 ``to_score_distribution`` (MEME-style, pwm/dist.rs)."""
        from .dist import ScoreDistribution
        if getattr(self, "_dist", None) is None:
            self._dist = ScoreDistribution(self)
        return self._dist

    def pvalue(self, score: float, method: str = "meme") -> float:
        """lib.pyi ``ScoringMatrix.pvalue``; only the MEME method is on this path
        (TFM-PVALUE is a separate GPL crate, out of scope)."""
        if method != "meme":
            raise ValueError(f"unsupported method: {method!r}")
        return self.score_distribution.pvalue(score)

    def score_for_pvalue(self, pvalue: float, method: str = "meme") -> float:
        """lib.pyi ``ScoringMatrix.score`` (renamed here: ``score`` is the scoring call)."""
        if method != "meme":
            raise ValueError(f"unsupported method: {method!r}")
        return self.score_distribution.score(pvalue)

    def reverse_complement(self) -> "ScoringMatrix":
        """pwm/mod.rs:566-577 (DNA: A<->T, C<->G, N->N).  Device copies of ``self`` get their
        complement made by the library (``lm_hip_pssm_reverse_complement``), without an upload."""
        if self.protein:
            raise ValueError("cannot complement a protein matrix")
        comp = [2, 3, 0, 1, 4]
        rc = ScoringMatrix(self.data[::-1, :5][:, comp], self.background, protein=False)
        for key, h in self._dev.items():
            pli = self._plis[key]
            out = C.c_void_p()
            check(pli._L.lm_hip_pssm_reverse_complement(pli._h, h, C.byref(out)))
            rc._dev[key], rc._plis[key] = out, pli
        return rc


class DiscreteMatrix:
    """pwm/mod.rs:754-791 ``{data: DenseMatrix<u8, K>, factor, offsets, offset}``."""

    def __init__(self, data: np.ndarray, factor: float, offsets: np.ndarray, offset: float, *,
                 protein: bool = False):
        self.data = np.ascontiguousarray(data, dtype=np.uint8)
        self.factor, self.offsets, self.offset, self.protein = factor, offsets, offset, protein

    @property
    def k(self) -> int:
        return _k(self.protein)

    def __len__(self) -> int:
        return self.data.shape[0]

    def scale(self, score: float) -> int:
        """pwm/mod.rs:777-779: rounds DOWN (f32 -> u8 threshold), saturating ``as u8``."""
        with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
            v = np.floor(np.float32(np.float32(score) - np.float32(self.offset)) / np.float32(self.factor))
        return 0 if v != v else int(min(max(v, 0), 255))

    def unscale(self, score: int) -> float:
        """pwm/mod.rs:783-785"""
        return float(np.float32(np.float32(score) * np.float32(self.factor) + np.float32(self.offset)))


class MotifBatch:
    """Device handles (+ thresholds) of a motif list, as arrays the C ABI reads (Pipeline.prepare_batch)."""

    def __init__(self, pli: "Pipeline", pssms: Sequence["ScoringMatrix"], thresholds: Optional[Sequence[float]] = None):
        self.pssms = list(pssms)          # (keeps the matrices, hence their device tables, alive)
        n = len(self.pssms)
        self.handles = (C.c_void_p * n)(*[p._device(pli) for p in self.pssms])
        self.thresholds = None
        if thresholds is not None:
            ts = np.ascontiguousarray(thresholds, dtype=np.float32)
            if ts.shape != (n,):
                raise ValueError("one threshold per motif")
            self._ts = ts
            self.thresholds = ts.ctypes.data_as(C.POINTER(C.c_float))

    def __len__(self) -> int:
        return len(self.pssms)


class BatchHits(Sequence):
    """The result of a batched fused threshold scan: per motif ``(coords (n_i, 2) int64, values (n_i,) f32)``, cut out of
    the two arrays the library returned when an element is asked for (2 346 slices cost a millisecond the scan does not)."""

    def __init__(self, coords: np.ndarray, values: np.ndarray, counts: np.ndarray):
        self.coords, self.values, self.counts = coords, values, counts
        self._starts = np.concatenate(([0], np.cumsum(counts, dtype=np.int64)))

    def __len__(self) -> int:
        return len(self.counts)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        a, b = int(self._starts[i]), int(self._starts[i + 1])
        return self.coords[a:b], self.values[a:b]

    @property
    def total(self) -> int:
        return int(self._starts[-1])


# --- scores -----------------------------------------------------------------------------


class StripedScores:
    """scores.rs:102-107 ``{data, max_index}``, resident on the device."""

    def __init__(self, pli: Pipeline, handle: C.c_void_p):
        self._pli, self._h = pli, handle

    @classmethod
    def empty(cls, pli: Optional[Pipeline] = None, columns: int = DEFAULT_COLUMNS) -> "StripedScores":
        pli = pli or default_pipeline()
        h = C.c_void_p()
        check(pli._L.lm_hip_scores_create(pli._h, columns, C.byref(h)))
        return cls(pli, h)

    def __del__(self):
        try:
            if self._h:
                self._pli._L.lm_hip_scores_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _info(self):
        v = [C.c_size_t(0) for _ in range(4)]
        p = C.c_void_p()
        check(self._pli._L.lm_hip_scores_info(self._h, *[C.byref(x) for x in v], C.byref(p)))
        return [x.value for x in v] + [p.value or 0]

    @property
    def rows(self) -> int:
        return self._info()[0]

    @property
    def stride(self) -> int:
        return self._info()[1]

    @property
    def columns(self) -> int:
        return self._info()[2]

    @property
    def max_index(self) -> int:
        return self._info()[3]

    @property
    def data_ptr(self) -> int:
        return self._info()[4]

    def set_first_cell_rule(self, enabled: bool) -> None:
        """This StripedScores is a row shard of a larger matrix: ``enabled=False`` when it does
        not hold the matrix's first row (Maximum::argmax's NaN first-cell rule is skipped)."""
        check(self._pli._L.lm_hip_scores_set_first_cell_rule(self._h, int(bool(enabled))))

    def is_empty(self) -> bool:
        return self.rows == 0

    def matrix(self) -> np.ndarray:
        """Host copy of the rows x stride f32 matrix."""
        rows, st = self.rows, self.stride
        out = np.empty((rows, st), dtype=np.float32)
        check(self._pli._L.lm_hip_scores_download(self._pli._h, self._h, out.ctypes.data))
        return out

    def __array__(self, dtype=None, copy=None) -> np.ndarray:
        """``numpy.asarray(scores)``: the reference exports the matrix through the buffer
        protocol as a read-only (columns, rows) view with strides (4, stride * 4)
        (lib.rs:1051-1085, 1129-1139), i.e. ``a[col, row]`` -- flattened in C order that is
        the score of every position ``col * rows + row`` in sequence order."""
        a = self.matrix()[:, :self.columns].T
        a.flags.writeable = False
        return a if dtype is None else a.astype(dtype)

    def offset(self, row: int, col: int) -> int:
        """scores.rs:155-157"""
        return col * self.rows + row

    def __len__(self) -> int:
        """scores.rs:274-279: min(max_index, rows * C)"""
        rows, _, cols, mi, _ = self._info()
        return min(mi, rows * cols)

    def unstripe(self) -> np.ndarray:
        """scores.rs:167-170"""
        m = self.matrix()
        n, rows = len(self), self.rows
        i = np.arange(n)
        return m[i % rows, i // rows] if rows else np.zeros(0, np.float32)

    def __iter__(self) -> Iterator[float]:
        return iter(float(x) for x in self.unstripe())

    def __getitem__(self, index: int) -> float:
        """scores.rs:246-254 (lib.rs:1038-1050 raises IndexError past the end)."""
        n = len(self)
        if index < 0:
            index += n
        if not 0 <= index < n:
            raise IndexError(index)
        rows = self.rows
        return float(self.rows_matrix(index % rows, index % rows + 1)[0, index // rows])

    def rows_matrix(self, row_begin: int, row_end: int) -> np.ndarray:
        """Host copy of rows ``[row_begin, row_end)`` only (``(n, stride)`` f32)."""
        out = np.empty((max(row_end - row_begin, 0), self.stride), dtype=np.float32)
        check(self._pli._L.lm_hip_scores_download_rows(self._pli._h, self._h, row_begin, row_end,
                                                       out.ctypes.data))
        return out

    def argmax(self) -> Optional[int]:
        """scores.rs:190-192"""
        mc = self._pli.argmax(self)
        return None if mc is None else self.offset(*mc)

    def max(self) -> Optional[float]:
        """scores.rs:181-183"""
        return self._pli.max(self)

    def threshold(self, threshold: float) -> List[int]:
        """scores.rs:207-213 (unsorted: row-major order of the striped matrix)."""
        rows = self.rows
        return [c * rows + r for r, c in self._pli.threshold(self, threshold)]


# --- Scanner (scan.rs:96-250 semantics on the fused kernel) ----------------------------------


class Hit:
    """scan.rs:52-75"""

    __slots__ = ("position", "score")

    def __init__(self, position: int, score: float):
        self.position, self.score = position, score

    def __repr__(self) -> str:
        return f"Hit(position={self.position}, score={self.score})"


# lm_hip_hit {size_t position; float score;} as the C compiler lays it out
_HIT_DTYPE = np.dtype({"names": ["position", "score"], "formats": ["<u8", "<f4"],
                       "offsets": [_ffi.Hit.position.offset, _ffi.Hit.score.offset],
                       "itemsize": C.sizeof(_ffi.Hit)})


class Scanner:
    """All positions with ``score >= threshold`` and ``position + M <= L``
    (scan.rs:185-190), found by the fused score+threshold kernel: a packed u16
    discrete prefilter (the device form of scan.rs:169-178's DiscreteMatrix) with exact
    f32 re-scoring of the candidates.

    Iteration yields the hits in the reference's order: blocks of ``block_size`` rows in
    ascending order (scan.rs:174-176, 196), and inside a block the LAST cell in row-major
    (row, col) order first -- the reference pushes a block's hits in ``Threshold`` order and
    pops them from the end of the vector (scan.rs:184-198).  ``positions`` / ``scores`` hold
    the same hits in ascending position, as the device returns them."""

    def __init__(self, pssm: ScoringMatrix, sequence: StripedSequence, threshold: float = 0.0,
                 block_size: int = 256):
        if pssm.protein != sequence.protein:
            raise ValueError("motif and sequence alphabets differ")
        if block_size < 1:
            raise ValueError("block_size must be positive")
        self.block_size = block_size
        self.threshold = threshold
        if sequence.wrap < len(pssm) - 1:  # scan.rs:127-131 panics
            raise ValueError(f"not enough wrapping rows for motif of length {len(pssm)}")
        self._pssm, self._seq = pssm, sequence
        self._positions: Optional[np.ndarray] = None  # filled by the first use (one fused scan)
        self._scores: Optional[np.ndarray] = None
        self._order: Optional[np.ndarray] = None
        self._next = 0

    def _scan(self, threshold: float) -> Tuple[np.ndarray, np.ndarray]:
        pli = self._seq._pli
        ptr, n = C.POINTER(_ffi.Hit)(), C.c_size_t(0)
        check(pli._L.lm_hip_scan_f32(pli._h, self._pssm._device(pli), self._seq._h, threshold,
                                     C.byref(ptr), C.byref(n)))
        try:
            raw = np.frombuffer(C.string_at(ptr, n.value * C.sizeof(_ffi.Hit)) if n.value else b"",
                                dtype=_HIT_DTYPE)
        finally:
            if ptr:
                pli._L.lm_hip_free(ptr)
        return raw["position"].astype(np.int64), raw["score"].astype(np.float32)

    def _collect(self) -> None:
        if self._order is not None:
            return
        self._positions, self._scores = self._scan(self.threshold)
        rows = max(self._seq.rows, 1)
        row, col = self._positions % rows, self._positions // rows
        # yield order: block ascending, then (row, col) descending
        self._order = np.lexsort((-col, -row, row // self.block_size))

    @property
    def positions(self) -> np.ndarray:
        """The hits' positions in ascending order (int64)."""
        self._collect()
        return self._positions

    @property
    def scores(self) -> np.ndarray:
        """The f32 scores of ``positions``."""
        self._collect()
        return self._scores

    def __iter__(self) -> "Scanner":
        return self

    def __next__(self) -> Hit:
        self._collect()
        if self._next >= self._order.size:
            raise StopIteration
        i = self._order[self._next]
        self._next += 1
        return Hit(int(self._positions[i]), float(self._scores[i]))

    def __len__(self) -> int:
        """Hits not yet yielded."""
        self._collect()
        return int(self._order.size - self._next)

    @staticmethod
    def _best(pos: np.ndarray, sc: np.ndarray) -> Optional[Hit]:
        if sc.size == 0:
            return None
        top = np.nonzero(sc == sc.max())[0]
        i = top[np.argmax(pos[top])]
        return Hit(int(pos[i]), float(sc[i]))

    def max(self, saturate: bool = True, strict_reference: Optional[bool] = None) -> Optional[Hit]:
        """``Scanner::max`` exactly as the reference computes it (scan.rs:200-249); consumes the scanner.

        ``strict_reference`` is accepted for callers written against the first rounds of this package (where the
        reference's walk was opt-in) and ignored with a DeprecationWarning: the walk is the only behaviour of
        ``max()`` now; ``strict_reference=False`` callers want :meth:`max_valid`.

        The u8 DiscreteMatrix scores steer which cells are looked at: starting from the best pending hit
        (or none) and the level ``dm.scale(threshold)``, cells are visited block by block in row-major
        order; a cell whose u8 score reaches the current level is re-scored in f32 and replaces the best
        hit when its score is greater, or equal at a greater position (scan.rs:237) -- and the level
        becomes ITS u8 score.  Consequences the reference has and this keeps: (a) positions are not
        tested against ``position + M <= L``; (b) while no hit is held the first candidate is accepted
        even if its f32 score is below the threshold; (c) a better cell whose u8 score lies under the
        level (the u8 score of the current best, an over-estimate) is skipped (scan.rs:227-243).
        ``saturate``: the u8 adds of the x86-64 ``dispatch`` pipeline (avx2.rs:336); ``False`` =
        Generic's wrapping adds.  :meth:`max_valid` is the variant without those corner cases."""
        if strict_reference is not None:
            import warnings
            warnings.warn("Scanner.max(strict_reference=...) is deprecated: max() always walks like the reference "
                          "(scan.rs:200-249); use max_valid() for the greatest valid hit", DeprecationWarning, stacklevel=2)
        return self._max_device(saturate)

    def _pending_state(self):
        """The reference scanner's state after the ``next()`` calls made so far (scan.rs:169-198): ``row`` = the
        block after the one the last yielded hit came from, ``hits`` = the not yet yielded hits of that block --
        derived from the complete hit list: (best pending hit or None, first row of the walk)."""
        thr = np.float32(self.threshold)
        rows, bs = self._seq.rows, self.block_size
        best: Optional[Tuple[int, np.float32]] = None
        first_block = 0
        if self._order is not None and self._next > 0:
            last = self._order[self._next - 1]
            blk = (int(self._positions[last]) % rows) // bs
            first_block = blk + 1
            rest = self._order[self._next:]
            rest = rest[(self._positions[rest] % rows) // bs == blk]
            for i in rest[::-1]:                       # the vector's order = reverse of the yield order
                sc = np.float32(self._scores[i])       # scan.rs:207-210; max_by keeps the LAST of equals
                if sc >= thr and (best is None or not (sc < best[1])):
                    best = (int(self._positions[i]), sc)
        self._order = np.zeros(0, np.int64)            # consumed (scan.rs:200 takes `self`)
        self._positions, self._scores, self._next = np.zeros(0, np.int64), np.zeros(0, np.float32), 0
        return best, first_block * bs

    def _max_device(self, saturate: bool) -> Optional[Hit]:
        """scan.rs:200-249 through ``lm_hip_scan_max_f32``: the walk itself runs on the device (csrc/scanmax.hip)."""
        pli, seq, pssm = self._seq._pli, self._seq, self._pssm
        best, first_row = self._pending_state()
        if seq.rows == 0 or len(seq) < len(pssm):
            return None if best is None else Hit(best[0], float(best[1]))
        dm = pssm.to_discrete()
        level = dm.scale(float(best[1])) if best is not None else dm.scale(float(np.float32(self.threshold)))
        w = np.ascontiguousarray(dm.data, dtype=np.uint8)
        found, hit = C.c_int(0), _ffi.Hit()
        st = pli._L.lm_hip_scan_max_f32(pli._h, pssm._device(pli), seq._h, w.ctypes.data, w.shape[1], int(bool(saturate)),
                                        int(level), int(best is not None), 0 if best is None else best[0],
                                        0.0 if best is None else float(best[1]), first_row, C.byref(found), C.byref(hit))
        if st == _ffi.ERR_BAD_ARGS and "leaves the striped matrix" in _ffi.last_error():
            raise IndexError("Scanner.max: " + _ffi.last_error())   # seq[pos + j] past the matrix: the reference panics
        check(st)
        return Hit(int(hit.position), float(hit.score)) if found.value else None

    def max_valid(self) -> Optional[Hit]:
        """NOT the reference's ``max()``: the best VALID hit not yet yielded -- ``score >= threshold``
        and ``position + M <= L`` like every hit ``__next__`` yields (scan.rs:186-189); greater score
        wins, equal scores go to the greater position.  Consumes the scanner.  On a fresh scanner the
        hit list is never built (a low threshold would select most of the sequence): the fused argmax
        gives the greatest score S of the matrix, and one scan at max(threshold, S) returns the few
        valid positions that reach it.  Differs from :meth:`max` exactly in that method's (a)-(c)."""
        if self._order is None:
            pli = self._seq._pli
            top = pli.score_argmax(self._pssm, self._seq)
            if top is None:
                self._order = np.zeros(0, np.int64)
                return None
            s_max = top[1]
            if s_max == s_max and s_max >= self.threshold:
                pos, sc = self._scan(s_max)
                if sc.size:  # valid positions (position + M <= L) that reach the matrix maximum
                    self._order = np.zeros(0, np.int64)
                    self._positions, self._scores = pos[:0], sc[:0]
                    return self._best(pos, sc)
                # the maximum sits in the padded tail only (finite weights for N): full list
            elif s_max == s_max:
                self._order = np.zeros(0, np.int64)  # nothing reaches the threshold
                self._positions, self._scores = np.zeros(0, np.int64), np.zeros(0, np.float32)
                return None
            self._collect()
        rest = self._order[self._next:]
        self._next = self._order.size
        return self._best(self._positions[rest], self._scores[rest])


# --- module-level helpers (lib.rs:1335-1451) ---------------------------------------------------


class Motif:
    def __init__(self, counts: Optional[CountMatrix], pwm: WeightMatrix, pssm: ScoringMatrix,
                 name: Optional[str] = None):
        self.counts, self.pwm, self.pssm, self.name = counts, pwm, pssm, name

    @property
    def protein(self) -> bool:
        return self.pssm.protein


def create(sequences: Iterable[str], *, protein: bool = False, name: Optional[str] = None) -> Motif:
    """lib.rs:1352-1386: counts -> to_freq(0.0).to_weight(None) -> to_scoring()."""
    encoded = [EncodedSequence(s, protein=protein) for s in sequences]
    counts = CountMatrix.from_sequences(encoded, protein=protein)
    pwm = counts.normalize(0.0)
    return Motif(counts, pwm, pwm.log_odds(), name)


def pack_2bit(encoded: np.ndarray, runs: bool = False):
    """Symbol bytes of a DNA sequence (A0 C1 T2 G3 N4, abc.rs:115-135) -> ``(packed, n)`` as
    ``Pipeline.stripe_2bit`` takes them: base i in bits 2*(i%4).. of byte i//4; N positions are packed as A and
    described by ``n`` -- a bit mask (bit i%8 of byte i//8; default) or, with ``runs=True``, an ``(k, 2)`` uint64
    array of ``(start, size)`` runs like a .2bit file's N blocks; ``None`` when the sequence has no N."""
    enc = np.ascontiguousarray(encoded, dtype=np.uint8)
    n = enc.size
    is_n = enc > 3
    two = np.where(is_n, 0, enc).astype(np.uint8)
    pad = (-n) % 4
    if pad:
        two = np.concatenate([two, np.zeros(pad, np.uint8)])
    q = two.reshape(-1, 4)
    packed = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)
    if not is_n.any():
        return packed, None
    if not runs:
        return packed, np.packbits(is_n, bitorder="little")
    edges = np.flatnonzero(np.diff(np.concatenate([[0], is_n.view(np.int8), [0]])))
    return packed, np.stack([edges[0::2], edges[1::2] - edges[0::2]], axis=1).astype(np.uint64)


def stripe(sequence: str, *, protein: bool = False) -> StripedSequence:
    """lib.rs:1402-1409"""
    return EncodedSequence(sequence, protein=protein).stripe()


def scan(pssm: ScoringMatrix, sequence: StripedSequence, *, threshold: float = 0.0,
         block_size: int = 256) -> Scanner:
    """lib.rs:1437-1451.  The reference's Python ``scan()`` is DNA-only (its PyO3 class is
    instantiated for ``Dna``); the Rust ``Scanner`` is alphabet-generic (scan.rs:96-136), and so
    is the ``Scanner`` class here -- this module-level helper keeps the binding's restriction."""
    if pssm.protein or sequence.protein:
        raise ValueError("scanner only supports DNA")  # lib.rs scan()
    return Scanner(pssm, sequence, threshold, block_size)
