"""Test helper: ``Scanner::max`` of the reference walked on the host from downloaded score matrices (moved out of the
product package in round 5: it is a cross-check of the device walk, not a path users run)."""
import numpy as np

from lightmotif_amd import Hit


def scanner_max_strict_host(scanner, saturate: bool = True):
    """``Scanner::max`` of the reference as written (scan.rs:200-249), walked on the HOST from the downloaded u8
    and f32 score matrices: the cross-check of the device walk (``Scanner.max`` -> ``lm_hip_scan_max_f32``).

    State of the reference scanner after some ``next()`` calls: ``row`` = the block after
    the one the last yielded hit came from, ``hits`` = the not yet yielded hits of that
    block (scan.rs:169-198).  The same state is derived here from the complete hit list."""
    pli, seq, pssm = scanner._seq._pli, scanner._seq, scanner._pssm
    thr = np.float32(scanner.threshold)
    rows, cols, m, bs = seq.rows, seq.columns, len(pssm), scanner.block_size
    best, first_row = scanner._pending_state()
    first_block = first_row // bs
    if rows == 0 or len(seq) < m:
        return None if best is None else Hit(best[0], float(best[1]))
    dm = pssm.to_discrete()
    level = dm.scale(float(best[1])) if best is not None else dm.scale(float(thr))   # scan.rs:211-214
    d_all, _ = pli.score_discrete(dm, seq, saturate=saturate)
    f_all = pli.score(pssm, seq)
    starts = np.arange(0, rows, bs)
    block_max = np.maximum.reduceat(d_all[:, :cols].max(axis=1), starts)
    for b in range(first_block, starts.size):
        if int(block_max[b]) < level:              # scan.rs:227
            continue
        r0 = b * bs
        d = d_all[r0:r0 + bs, :cols]
        rr, cc = np.nonzero(d >= level)            # Threshold: row-major (row, col) order
        fs = f_all.rows_matrix(r0, min(r0 + bs, rows))
        for r, c in zip(rr.tolist(), cc.tolist()):
            ds = int(d[r, c])
            if ds < level:                         # scan.rs:229: the level moves inside the loop
                continue
            index = c * rows + r0 + r
            if index + m > rows * cols:            # seq[pos + j] past the matrix: the reference panics
                raise IndexError(f"Scanner.max: position {index} + {m} leaves the striped matrix")
            score = np.float32(fs[r, c])           # = score_position (pwm/mod.rs:651-662)
            if best is None:
                best = (index, score)              # scan.rs:241: no threshold test, level unchanged
            elif score > best[1] or (score == best[1] and index > best[0]):
                best = (index, score)
                level = ds
    return None if best is None else Hit(best[0], float(best[1]))
