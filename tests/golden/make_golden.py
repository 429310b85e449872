#!/usr/bin/env python3
"""Generates tests/golden/generated_cases.npz from the CPU oracle.

The reference cannot be built or imported here (Rust), so cases
that its own tests pin with no literal (ties, NaN, the padded tail, protein,
odd geometries) are fixed by running the C restatement of the Generic pipeline
(oracle/lm_oracle.c, itself pinned by tests/golden/reference_vectors.json and
cross-checked against the independent numpy restatement).  Re-run with
    python tests/golden/make_golden.py
Inputs and expected outputs are stored as arrays; nothing of the reference's
source text is included.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import c_oracle as co, np_oracle as no  # noqa: E402


def make_case(rng, name, length, m, cols, k, *, special=None, rows=None, extra_wrap=0):
    enc = rng.integers(0, k - 1, size=length, dtype=np.uint8)
    if special == "with_default":  # sprinkle N / X symbols
        enc[rng.random(length) < 0.05] = k - 1
    pssm = np.zeros((m, co.stride(k, 4)), dtype=np.float32)
    pssm[:, :k] = rng.normal(0, 2, size=(m, k)).astype(np.float32)
    pssm[:, k - 1] = -np.inf
    if special == "ties":  # few distinct weights -> many equal scores
        pssm[:, :k - 1] = rng.integers(-1, 2, size=(m, k - 1)).astype(np.float32)
    if special == "nan":
        pssm[m // 2, 1] = np.nan
    if special == "finite_default":  # A5: a finite N column exposes the padded tail
        pssm[:, k - 1] = rng.normal(0, 1, size=m).astype(np.float32)
    if special == "denormal":
        pssm[:, :k - 1] = (rng.integers(1, 1000, size=(m, k - 1)) * 1e-42).astype(np.float32)
    if special == "signed_zero":
        pssm[:, :k - 1] = -0.0
    s = co.stripe(enc, cols, k)
    co.configure_wrap(s, max(m - 1, 0) + extra_wrap)
    a, b = (0, s.rows) if rows is None else rows
    scores, mi = co.score_rows(s, pssm, a, b)
    # independent cross-check
    d = no.stripe(enc, cols, k - 1)
    d, w = no.configure_wrap(d, d.shape[0], cols, 0, max(m - 1, 0) + extra_wrap, k - 1)
    assert np.array_equal(d, s.data), name
    sc2, mi2 = no.score_rows(d, cols, length, pssm, a, b)
    assert mi2 == mi and np.array_equal(sc2.view(np.uint32), scores.view(np.uint32)), name
    am = co.argmax(scores, cols)
    assert am == no.argmax(scores, cols), name
    out = {
        "encoded": enc, "pssm": pssm, "cols": np.int64(cols), "k": np.int64(k),
        "row_range": np.array([a, b], dtype=np.int64), "wrap": np.int64(s.wrap),
        "striped": s.data.copy(), "scores": scores.copy(), "max_index": np.int64(mi),
        "argmax": np.array(am if am is not None else (-1, -1), dtype=np.int64),
    }
    finite = scores[:, :cols][np.isfinite(scores[:, :cols])]
    ts = [np.float32(-np.inf), np.float32(0.0)]
    if finite.size:
        ts += [np.float32(np.quantile(finite, 0.9)), np.float32(finite.max())]
    out["thresholds"] = np.array(ts, dtype=np.float32)
    for i, t in enumerate(ts):
        rc = co.threshold(scores, cols, float(t))
        assert np.array_equal(rc.astype(np.int64), no.threshold(scores, cols, t).astype(np.int64)), name
        out[f"threshold_{i}"] = rc.astype(np.int64)
    return {f"{name}/{key}": val for key, val in out.items()}


def main():
    rng = np.random.default_rng(0x5EED0003)
    cases = {}
    spec = [
        ("dna_c32_m20", 5000, 20, 32, 5, {}),
        ("dna_c32_m15_ragged", 4097, 15, 32, 5, {"special": "with_default"}),
        ("dna_c32_ties", 3000, 6, 32, 5, {"special": "ties"}),
        ("dna_c32_nan", 2000, 8, 32, 5, {"special": "nan"}),
        ("dna_c32_finite_default", 1001, 10, 32, 5, {"special": "finite_default"}),
        ("dna_c32_denormal", 1500, 12, 32, 5, {"special": "denormal"}),
        ("dna_c32_signed_zero", 700, 5, 32, 5, {"special": "signed_zero"}),
        ("dna_c32_rows", 6000, 20, 32, 5, {"rows": (17, 140)}),
        ("dna_c32_extra_wrap", 2500, 9, 32, 5, {"extra_wrap": 7}),
        ("dna_c32_short", 40, 15, 32, 5, {}),
        ("dna_c32_m1", 999, 1, 32, 5, {}),
        ("dna_c32_m33", 4000, 33, 32, 5, {}),
        ("dna_c1", 300, 7, 1, 5, {}),
        ("dna_c16", 1000, 11, 16, 5, {"special": "ties"}),
        ("protein_c32_m12", 5000, 12, 32, 21, {"special": "with_default"}),
        ("protein_c32_ties", 2000, 5, 32, 21, {"special": "ties"}),
    ]
    for name, length, m, cols, k, kw in spec:
        cases.update(make_case(rng, name, length, m, cols, k, **kw))
    out = ROOT / "tests" / "golden" / "generated_cases.npz"
    np.savez_compressed(out, **cases)
    print(out, out.stat().st_size, "bytes,", len(spec), "cases")


if __name__ == "__main__":
    main()
