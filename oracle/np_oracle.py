"""numpy-float32 restatement of the reference's Generic pipeline.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Written independently
of ``lm_oracle.c`` so the two can cross-check each other; every function cites
the reference lines it follows (crate ``lightmotif/``).
"""
from __future__ import annotations

import numpy as np

DNA = "ACTGN"                    # abc.rs:106-108
PROTEIN = "ACDEFGHIKLMNPQRSTVWYX"  # abc.rs:193-256


def stride(cols: int, elem_size: int) -> int:
    """dense.rs:43-48,126-128 (x86-64: rows aligned to 32 bytes)."""
    return -(-cols * elem_size // 32) * 32 // elem_size


def encode(text: str, alphabet: str = DNA) -> np.ndarray:
    """pli/mod.rs:56-66 with Symbol::from_ascii (abc.rs:166-171)."""
    lut = {c: i for i, c in enumerate(alphabet)}
    try:
        return np.array([lut[c] for c in text], dtype=np.uint8)
    except KeyError as e:  # err.rs InvalidSymbol
        raise ValueError(f"invalid symbol {e.args[0]!r}") from None


def stripe(encoded: np.ndarray, cols: int, default: int) -> np.ndarray:
    """pli/mod.rs:178-200: position i -> data[i % rows][i / rows]."""
    n = len(encoded)
    rows = -(-n // cols)
    data = np.full((rows, stride(cols, 1)), default, dtype=np.uint8)   # dense.rs:144-147: T::default()
    if rows == 0:
        return data
    i = np.arange(rows * cols)
    vals = np.full(rows * cols, default, dtype=np.uint8)
    vals[:n] = encoded
    data[i % rows, i // rows] = vals
    return data


def configure_wrap(data: np.ndarray, rows: int, cols: int, wrap: int, m: int, default: int):
    """seq.rs:369-381.  Returns (data, wrap)."""
    if m <= wrap:
        return data, wrap
    out = np.full((rows + m, data.shape[1]), default, dtype=np.uint8)  # resize: T::default()
    out[:rows + wrap] = data[:rows + wrap]
    for i in range(m):
        out[rows + i, :cols - 1] = out[i, 1:cols]
        out[rows + i, cols - 1] = default
    return out, m


def pssm_from_sites(sites: list[np.ndarray], k: int, pseudocount: float = 0.1) -> np.ndarray:
    """pwm/mod.rs:209-237, 240-258, 415-430 with the uniform background of
    abc.rs:473-487; all arithmetic in float32 like the reference."""
    m = len(sites[0])
    f32 = np.float32
    out = np.zeros((m, stride(k, 4)), dtype=f32)
    bg = np.array([f32(1) / f32(k - 1)] * (k - 1) + [f32(0)], dtype=f32)
    for i in range(m):
        row = np.zeros(k, dtype=f32)
        for j in range(k):
            cnt = sum(int(s[i] == j) for s in sites)
            row[j] = f32(cnt) + (f32(pseudocount) if j != k - 1 else f32(0))
        tot = f32(0)
        for j in range(k):
            tot = f32(tot + row[j])
        row = (row / tot).astype(f32)
        with np.errstate(divide="ignore"):
            for j in range(k):
                out[i, j] = -np.inf if bg[j] == 0 else np.log2(f32(row[j] / bg[j]), dtype=f32)
    return out


def score_rows(data: np.ndarray, cols: int, length: int, pssm: np.ndarray,
               row_begin: int, row_end: int):
    """pli/mod.rs:72-106.  Returns (scores (rows, stride(cols,4)) f32, max_index)."""
    m = pssm.shape[0]
    if length < m or row_begin >= row_end:
        return np.zeros((0, stride(cols, 4)), dtype=np.float32), 0
    n = row_end - row_begin
    out = np.zeros((n, stride(cols, 4)), dtype=np.float32)
    acc = np.zeros((n, cols), dtype=np.float32)          # :98 T::default()
    with np.errstate(invalid="ignore"):
        for j in range(m):                                   # :99, sequential in j
            sym = data[row_begin + j:row_end + j, :cols]     # :100
            acc = (acc + pssm[j][sym]).astype(np.float32)    # :101 one f32 add per row
    out[:, :cols] = acc
    return out, max(length + 1 - m, 0)


def argmax(scores: np.ndarray, cols: int):
    """pli/mod.rs:135-155: last maximal cell in (row, col) order; NaN never wins."""
    if scores.shape[0] == 0:
        return None
    best = scores[0, 0]
    br = bc = 0
    flat = scores[:, :cols]
    if np.isnan(best):
        return (0, 0)
    with np.errstate(invalid="ignore"):
        mx = np.nanmax(flat)
        cand = np.argwhere(flat >= mx)
    br, bc = cand[-1]
    return (int(br), int(bc))


def threshold(scores: np.ndarray, cols: int, t: float) -> np.ndarray:
    """pli/mod.rs:210-221: row-major (row, col) list of cells with x >= t."""
    with np.errstate(invalid="ignore"):
        return np.argwhere(scores[:, :cols] >= np.float32(t))


def unstripe(scores: np.ndarray, cols: int, max_index: int) -> np.ndarray:
    """scores.rs:270-288."""
    rows = scores.shape[0]
    end = min(max_index, rows * cols)
    i = np.arange(end)
    return scores[i % rows, i // rows] if rows else np.zeros(0, np.float32)


def scanner_collect(scores: np.ndarray, cols: int, length: int, m: int, t: float,
                    block_size: int = 256) -> list[tuple[int, np.float32]]:
    """scan.rs:166-198 (`Scanner::next` until exhaustion), with an exact prefilter: the
    u8 DiscreteMatrix pass only selects candidates, every hit is decided by the f32
    `score_position` (scan.rs:187-190), so the yielded (position, score) sequence is fixed
    by the f32 scores.  Blocks of `block_size` rows in ascending order (scan.rs:174-176,
    196); inside a block the candidates are pushed in `Threshold`'s row-major order
    (scan.rs:184, pli/mod.rs:212-218) and yielded by `Vec::pop` (scan.rs:198): LAST pushed
    first.  `scores` = full score matrix (rows x >= cols) of the configured sequence.
    """
    rows = scores.shape[0]
    out: list[tuple[int, np.float32]] = []
    tt = np.float32(t)
    for row0 in range(0, rows, max(block_size, 1)):
        block: list[tuple[int, np.float32]] = []
        for r in range(row0, min(row0 + block_size, rows)):
            for c in range(cols):
                index = c * rows + r                       # scan.rs:185
                if index + m <= length:                    # scan.rs:186
                    s = scores[r, c]
                    if s >= tt:                            # scan.rs:188
                        block.append((index, s))
        while block:
            out.append(block.pop())                        # scan.rs:198
    return out


def scanner_max(scores: np.ndarray, cols: int, length: int, m: int, t: float):
    """scan.rs:200-249 (`Scanner::max` on a fresh scanner) for inputs where the discrete
    under-estimate does not matter: the best hit with `score >= t`; greater score wins,
    equal scores go to the greater position (scan.rs:237).  (The reference also accepts, as
    its FIRST candidate only, a cell whose u8 score reaches the scaled threshold while its
    f32 score is below `t`, and does not test `index + M <= L` here; neither can be the best
    hit when some valid position scores >= t and padded windows score -inf.)"""
    best = None
    for index, s in scanner_collect(scores, cols, length, m, t, block_size=1 << 30):
        if best is None or s > best[1] or (s == best[1] and index > best[0]):
            best = (index, s)
    return best


def scanner_max_strict(scores: np.ndarray, dscores: np.ndarray, cols: int, t: float, scale,
                       block_size: int = 256, pending=()):
    """scan.rs:200-249 `Scanner::max`, line by line, quirks included.  `scores` = the f32
    score matrix (what `score_position` returns for cell (row, col), pwm/mod.rs:651-662),
    `dscores` = the u8 scores of the DiscreteMatrix the reference's pipeline produces (on
    x86-64 `dispatch` = AVX2: saturating adds, avx2.rs:336), `scale` = DiscreteMatrix::scale
    (pwm/mod.rs:782-784), `pending` = hits already collected but not yet yielded (scan.rs:207).

    Quirks reproduced on purpose: (a) no `index + M <= L` test (scan.rs:230-232); (b) while
    no best hit exists the FIRST candidate is accepted without `score >= threshold`
    (scan.rs:240-242) and `best_discrete` keeps the scaled threshold; (c) once a hit is held,
    cells are filtered by the u8 score of the CURRENT best (scan.rs:229, 238), an over-estimate,
    so a cell with a greater f32 score but a smaller u8 score is skipped."""
    rows = scores.shape[0]
    tt = np.float32(t)
    best = None
    for pos, sc in pending:                                   # scan.rs:207-210 (max_by: last of equals)
        if sc >= tt and (best is None or not (sc < best[1])):
            best = (pos, np.float32(sc))
    best_discrete = scale(best[1]) if best is not None else scale(tt)   # scan.rs:211-214
    for row0 in range(0, rows, block_size):                   # scan.rs:220-247 (blocks past the
        end = min(row0 + block_size, rows)                    #   sequence rows are empty)
        d = dscores[row0:end, :cols]
        if d.size == 0 or int(d.max()) < best_discrete:       # scan.rs:227
            continue
        cand = np.argwhere(d >= best_discrete)                # Threshold: row-major (pli/mod.rs:212-218)
        for r, c in cand:
            dscore = int(d[r, c])
            if dscore >= best_discrete:                       # scan.rs:229 (best_discrete moves)
                index = int(c) * rows + row0 + int(r)
                score = scores[row0 + r, c]
                if best is not None:
                    if score > best[1] or (score == best[1] and index > best[0]):   # scan.rs:236-239
                        best = (index, score)
                        best_discrete = dscore
                else:
                    best = (index, score)                     # scan.rs:241
    return best


# ---- DiscreteMatrix (pwm/mod.rs:665-696, 754-791) ------------------------------------------


def _rust_f32_to_u8(x: np.ndarray) -> np.ndarray:
    """Rust `f32 as u8`: saturating, NaN -> 0."""
    x = np.asarray(x, dtype=np.float32)
    with np.errstate(invalid="ignore"):
        y = np.where(np.isnan(x), np.float32(0), np.clip(x, np.float32(0), np.float32(255)))
    return np.trunc(y).astype(np.uint8)


def to_discrete(pssm: np.ndarray, k: int):
    """pwm/mod.rs:665-696 `ScoringMatrix::to_discrete` in f32 arithmetic.  `pssm` is
    ``(M, >= k) f32``; returns (weights ``(M, k) u8``, factor, offsets ``(M,) f32``, offset).
    max_score (pwm/mod.rs:604-615) = sum over rows of the row maximum over the first
    k-1 symbols; a -inf weight counts as -max_score for the row minimum."""
    p = np.asarray(pssm, dtype=np.float32)[:, :k]
    m = p.shape[0]
    max_score = np.float32(0)
    for j in range(m):                              # sequential f32 sum, as `.sum::<f32>()`
        max_score = np.float32(max_score + p[j, :k - 1].max())
    offsets = np.empty(m, np.float32)
    for j in range(m):
        row = np.where(np.isinf(p[j, :k - 1]), np.float32(-max_score), p[j, :k - 1])
        offsets[j] = row.min()
    offset = np.float32(0)
    for j in range(m):
        offset = np.float32(offset + offsets[j])
    factor = np.float32(np.float32(max_score - offset) / np.float32(255))
    with np.errstate(invalid="ignore", over="ignore"):
        scaled = np.ceil((p - offsets[:, None]).astype(np.float32) / factor)
    return _rust_f32_to_u8(scaled), factor, offsets, offset


def discrete_scale(score: float, factor, offset) -> int:
    """pwm/mod.rs:777-779: floor((score - offset) / factor) as u8."""
    with np.errstate(invalid="ignore", over="ignore"):
        v = np.floor(np.float32(np.float32(score) - offset) / factor)
    return int(_rust_f32_to_u8(np.array([v]))[0])


def discrete_unscale(score: int, factor, offset) -> np.float32:
    """pwm/mod.rs:783-785"""
    return np.float32(np.float32(score) * factor + offset)


def score_rows_u8_saturating(data: np.ndarray, cols: int, length: int, weights: np.ndarray,
                             row_begin: int, row_end: int) -> np.ndarray:
    """avx2.rs:294-347: per row, M saturating byte adds (`_mm256_adds_epu8`, :336) of the
    shuffled weights, from zero.  ``data`` = striped matrix incl. wrap rows."""
    m = weights.shape[0]
    if length < m or row_begin >= row_end:
        return np.zeros((0, cols), np.uint8)
    out = np.zeros((row_end - row_begin, cols), np.uint16)
    for j in range(m):
        y = weights[j][data[row_begin + j: row_end + j, :cols]].astype(np.uint16)
        out = np.minimum(out + y, 255)             # adds_epu8
    return out.astype(np.uint8)
