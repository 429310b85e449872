"""ctypes front-end of the C oracle (``lm_oracle.c``) and AVX2 port (``lm_avx2.c``).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Shapes follow the
reference's memory layout: striped sequence ``(rows+wrap, stride)``
u8, PSSM ``(M, stride(K))`` f32, scores ``(rows, stride(C))`` f32.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_BUILD = _HERE / "_build"

_sz = C.c_size_t
_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_szp = C.POINTER(C.c_size_t)


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (``make -C oracle``)."""
    if force:
        subprocess.run(["make", "-C", str(_HERE), "clean"], check=True,
                       stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", str(_HERE)], check=True,
                   stdout=subprocess.DEVNULL)


def _load(name: str) -> C.CDLL:
    path = _BUILD / name
    src_newer = (not path.exists()) or any(
        p.stat().st_mtime > path.stat().st_mtime
        for p in _HERE.glob("*.[ch]"))
    if src_newer and os.access(_HERE, os.W_OK):
        build()
    return C.CDLL(str(path))


_lib = None
_avx = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = _load("liblm_oracle.so")
        L.lmo_stride.restype = _sz
        L.lmo_stride.argtypes = [_sz, _sz]
        L.lmo_encode.restype = _sz
        L.lmo_encode.argtypes = [C.c_char, _u8p, _sz, C.c_int, _u8p]
        L.lmo_stripe.restype = _sz
        L.lmo_stripe.argtypes = [_u8p, _sz, _sz, C.c_uint8, _u8p, _sz]
        L.lmo_configure_wrap.restype = _sz
        L.lmo_configure_wrap.argtypes = [_u8p, _sz, _sz, _sz, _sz, _sz, C.c_uint8]
        L.lmo_score_rows_f32.restype = None
        L.lmo_score_rows_f32.argtypes = [_u8p, _sz, _sz, _sz, _f32p, _sz, _sz,
                                         _sz, _sz, _f32p, _sz, _szp, _szp]
        L.lmo_score_rows_u8.restype = None
        L.lmo_score_rows_u8.argtypes = [_u8p, _sz, _sz, _sz, _u8p, _sz, _sz,
                                        _sz, _sz, _u8p, _sz, _szp, _szp]
        L.lmo_argmax_f32.restype = C.c_int
        L.lmo_argmax_f32.argtypes = [_f32p, _sz, _sz, _sz, _szp, _szp]
        L.lmo_max_f32.restype = C.c_int
        L.lmo_max_f32.argtypes = [_f32p, _sz, _sz, _sz, _f32p]
        L.lmo_threshold_f32.restype = _sz
        L.lmo_threshold_f32.argtypes = [_f32p, _sz, _sz, _sz, C.c_float, _szp, _sz]
        L.lmo_offset.restype = _sz
        L.lmo_offset.argtypes = [_sz, _sz, _sz]
        L.lmo_unstripe_f32.restype = _sz
        L.lmo_unstripe_f32.argtypes = [_f32p, _sz, _sz, _sz, _sz, _f32p]
        L.lmo_pssm_from_sites.restype = None
        L.lmo_pssm_from_sites.argtypes = [_u8p, _sz, _sz, _sz, C.c_float, _f32p,
                                          _f32p, _sz]
        L.lmo_score_position.restype = C.c_float
        L.lmo_score_position.argtypes = [_u8p, _sz, _sz, _f32p, _sz, _sz, _sz]
        _lib = L
    return _lib


def avx2() -> C.CDLL:
    global _avx
    if _avx is None:
        L = _load("liblm_avx2.so")
        L.lma_score_rows_f32.restype = C.c_int
        L.lma_score_rows_f32.argtypes = [_u8p, _sz, _sz, _sz, _f32p, _sz, _sz,
                                         _sz, _sz, _sz, _f32p, _sz]
        L.lma_score_rows_f32_mt.restype = C.c_int
        L.lma_score_rows_f32_mt.argtypes = [_u8p, _sz, _sz, _sz, _f32p, _sz, _sz,
                                            _sz, _sz, _sz, _f32p, _sz, C.c_int]
        L.lma_score_rows_u8.restype = C.c_int
        L.lma_score_rows_u8.argtypes = [_u8p, _sz, _sz, _sz, _u8p, _sz, _sz, _sz, _sz, _u8p, _sz]
        L.lma_argmax_f32.restype = C.c_int
        L.lma_argmax_f32.argtypes = [_f32p, _sz, _sz, _sz, _szp, _szp]
        _avx = L
    return _avx


def _p8(a: np.ndarray):
    return a.ctypes.data_as(_u8p)


def _pf(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def aligned_empty(shape, dtype, align: int = 32) -> np.ndarray:
    """numpy array whose base pointer is `align`-byte aligned (dense.rs:43)."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    raw = np.empty(n + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].view(dtype).reshape(shape)


DNA_K, PROTEIN_K = 5, 21


def alphabet_k(alphabet: str) -> int:
    return PROTEIN_K if alphabet == "P" else DNA_K


def stride(cols: int, elem_size: int) -> int:
    return int(lib().lmo_stride(cols, elem_size))


def encode(text: bytes | str, alphabet: str = "D", lossy: bool = False) -> np.ndarray:
    if isinstance(text, str):
        text = text.encode("ascii")
    src = np.frombuffer(text, dtype=np.uint8)
    dst = np.empty(len(src), dtype=np.uint8)
    bad = lib().lmo_encode(alphabet.encode(), _p8(src), len(src), int(lossy), _p8(dst))
    if bad:
        raise ValueError(f"invalid symbol {chr(src[bad - 1])!r} at {bad - 1}")
    return dst


class Striped:
    """(data, length, wrap, cols) -- the fields of seq.rs:288-294."""

    def __init__(self, data: np.ndarray, length: int, wrap: int, cols: int, k: int):
        self.data, self.length, self.wrap, self.cols, self.k = data, length, wrap, cols, k

    @property
    def rows(self) -> int:
        return self.data.shape[0] - self.wrap

    @property
    def stride(self) -> int:
        return self.data.shape[1]


def stripe(encoded: np.ndarray, cols: int = 32, k: int = DNA_K, extra_rows: int = 64) -> Striped:
    encoded = np.ascontiguousarray(encoded, dtype=np.uint8)
    st = stride(cols, 1)
    rows = -(-len(encoded) // cols)
    buf = aligned_empty((rows + extra_rows, st), np.uint8)
    buf[:] = k - 1   # spare rows like fresh DenseMatrix rows: T::default() (dense.rs:144-147)
    got = lib().lmo_stripe(_p8(encoded), len(encoded), cols, k - 1, _p8(buf), st)
    assert got == rows
    s = Striped(buf[:rows], len(encoded), 0, cols, k)
    s._buf = buf
    return s


def configure_wrap(s: Striped, m: int) -> Striped:
    """seq.rs:369-381; `m` is the number of wrap rows wanted (motif_len - 1)."""
    if m <= s.wrap:
        return s
    rows = s.rows
    need = rows + m
    if need > s._buf.shape[0]:
        nb = aligned_empty((need + 32, s.stride), np.uint8)
        nb[:] = s.k - 1
        nb[:s.data.shape[0]] = s.data
        s._buf = nb
    wrap = lib().lmo_configure_wrap(_p8(s._buf), rows, s.stride, s.cols, s.wrap, m, s.k - 1)
    s.wrap = int(wrap)
    s.data = s._buf[:rows + s.wrap]
    return s


def pssm_from_sites(sites: list[np.ndarray], k: int = DNA_K, pseudocount: float = 0.1,
                    background: np.ndarray | None = None) -> np.ndarray:
    m = len(sites[0])
    flat = np.ascontiguousarray(np.concatenate(sites), dtype=np.uint8)
    st = stride(k, 4)
    out = aligned_empty((m, st), np.float32)
    bg = None if background is None else np.ascontiguousarray(background, dtype=np.float32)
    lib().lmo_pssm_from_sites(_p8(flat), len(sites), m, k, pseudocount,
                              None if bg is None else _pf(bg), _pf(out), st)
    return out


def score_rows(s: Striped, pssm: np.ndarray, row_begin: int | None = None,
               row_end: int | None = None):
    """Returns (scores[(rows, stride(cols)) f32], max_index)."""
    a = 0 if row_begin is None else row_begin
    b = s.rows if row_end is None else row_end
    m = pssm.shape[0]
    ost = stride(s.cols, 4)
    n = max(b - a, 0)
    out = aligned_empty((n, ost), np.float32)
    out[:] = 0
    orow, mi = C.c_size_t(0), C.c_size_t(0)
    pssm = np.ascontiguousarray(pssm, dtype=np.float32)
    lib().lmo_score_rows_f32(_p8(s.data), s.stride, s.cols, s.length, _pf(pssm), m,
                             pssm.shape[1], a, b, _pf(out), ost, C.byref(orow), C.byref(mi))
    return out[:orow.value], int(mi.value)


def score_rows_u8(s: Striped, weights: np.ndarray, row_begin: int | None = None,
                  row_end: int | None = None):
    """Generic `Score<u8, ..>` with a DiscreteMatrix's weights ``(M, stride) u8`` (wrapping
    `+=`, pli/mod.rs:98-102).  Returns (scores[(rows, stride(cols, 1)) u8], max_index)."""
    a = 0 if row_begin is None else row_begin
    b = s.rows if row_end is None else row_end
    ost = stride(s.cols, 1)
    n = max(b - a, 0)
    out = aligned_empty((n, ost), np.uint8)
    out[:] = 0
    orow, mi = C.c_size_t(0), C.c_size_t(0)
    weights = np.ascontiguousarray(weights, dtype=np.uint8)
    lib().lmo_score_rows_u8(_p8(s.data), s.stride, s.cols, s.length, _p8(weights), weights.shape[0],
                            weights.shape[1], a, b, _p8(out), ost, C.byref(orow), C.byref(mi))
    return out[:orow.value], int(mi.value)


def argmax(scores: np.ndarray, cols: int):
    r, c = C.c_size_t(0), C.c_size_t(0)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    ok = lib().lmo_argmax_f32(_pf(scores), scores.shape[0], scores.shape[1] if scores.ndim == 2 else cols,
                              cols, C.byref(r), C.byref(c))
    return (int(r.value), int(c.value)) if ok else None


def max_(scores: np.ndarray, cols: int):
    v = C.c_float(0)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    ok = lib().lmo_max_f32(_pf(scores), scores.shape[0], scores.shape[1], cols, C.byref(v))
    return np.float32(v.value) if ok else None


def threshold(scores: np.ndarray, cols: int, t: float) -> np.ndarray:
    """(n, 2) array of (row, col) in the reference's row-major order."""
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    rows = scores.shape[0]
    st = scores.shape[1] if rows else cols
    n = lib().lmo_threshold_f32(_pf(scores), rows, st, cols, t, None, 0)
    rc = np.empty((n, 2), dtype=np.uintp)
    if n:
        lib().lmo_threshold_f32(_pf(scores), rows, st, cols, t,
                                rc.ctypes.data_as(_szp), n)
    return rc


def offset(rows: int, row: int, col: int) -> int:
    return int(lib().lmo_offset(rows, row, col))


def unstripe(scores: np.ndarray, cols: int, max_index: int) -> np.ndarray:
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    rows = scores.shape[0]
    dst = np.empty(min(max_index, rows * cols), dtype=np.float32)
    if rows:
        lib().lmo_unstripe_f32(_pf(scores), rows, scores.shape[1], cols, max_index, _pf(dst))
    return dst


def score_position(s: Striped, pssm: np.ndarray, pos: int) -> np.float32:
    pssm = np.ascontiguousarray(pssm, dtype=np.float32)
    return np.float32(lib().lmo_score_position(_p8(s.data), s.stride, s.rows, _pf(pssm),
                                               pssm.shape[0], pssm.shape[1], pos))


# --- AVX2 port (cpu_baseline) -------------------------------------------------

def avx2_score_rows(s: Striped, pssm: np.ndarray, out: np.ndarray | None = None,
                    row_begin: int | None = None, row_end: int | None = None,
                    threads: int = 1) -> np.ndarray:
    assert s.cols == 32
    a = 0 if row_begin is None else row_begin
    b = s.rows if row_end is None else row_end
    pssm = np.ascontiguousarray(pssm, dtype=np.float32)
    assert pssm.ctypes.data % 32 == 0 and s.data.ctypes.data % 32 == 0
    if out is None:
        out = aligned_empty((max(b - a, 0), 32), np.float32)
    k = s.k
    rc = avx2().lma_score_rows_f32_mt(_p8(s.data), s.stride, s.wrap, s.length, _pf(pssm),
                                      pssm.shape[0], pssm.shape[1], k, a, b, _pf(out), 32,
                                      threads)
    if rc == 2:
        raise RuntimeError(f"not enough wrapping rows for motif of length {pssm.shape[0]}")
    if s.length < pssm.shape[0] or a >= b:
        return out[:0]
    return out


def avx2_argmax(scores: np.ndarray, max_index: int):
    r, c = C.c_size_t(0), C.c_size_t(0)
    ok = avx2().lma_argmax_f32(_pf(scores), scores.shape[0], scores.shape[1], max_index,
                               C.byref(r), C.byref(c))
    if ok < 0:
        raise OverflowError("more than u32::MAX positions")
    return (int(r.value), int(c.value)) if ok else None


def avx2_score_rows_u8(s: Striped, weights: np.ndarray, out: np.ndarray | None = None,
                       row_begin: int | None = None, row_end: int | None = None) -> np.ndarray:
    """avx2.rs:292-347 (what Dispatch::Avx2 runs for Score<u8, Dna>): saturating sums.  `weights`: (M, stride >= 16) u8,
    16-byte aligned rows (a DenseMatrix<u8, K> has 32-byte rows)."""
    assert s.cols == 32 and weights.dtype == np.uint8 and weights.shape[1] >= 16
    a = 0 if row_begin is None else row_begin
    b = s.rows if row_end is None else row_end
    assert weights.ctypes.data % 16 == 0 and weights.strides[0] % 16 == 0 and s.data.ctypes.data % 32 == 0
    if out is None:
        out = aligned_empty((max(b - a, 0), 32), np.uint8)
    rc = avx2().lma_score_rows_u8(_p8(s.data), s.stride, s.wrap, s.length, _p8(weights), weights.shape[0],
                                  weights.strides[0], a, b, _p8(out), out.strides[0])
    if rc == 2:
        raise RuntimeError(f"not enough wrapping rows for motif of length {weights.shape[0]}")
    if s.length < weights.shape[0] or a >= b:
        return out[:0]
    return out
