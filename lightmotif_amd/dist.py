"""p-value <-> score conversion for a ScoringMatrix: the MEME-style
discretised score distribution of lightmotif/src/pwm/dist.rs, restated in numpy float64.
Pure host code; it turns the CLI's ``--pvalue 1e-5`` (lightmotif-cli main.rs:487-498)
into the f32 threshold handed to the fused GPU scan."""
from __future__ import annotations

import numpy as np

CDF_RANGE = 1000  # dist.rs:47


class ScoreDistribution:
    """dist.rs:51-226"""

    def __init__(self, pssm):
        k = pssm.k
        data = pssm.data[:, :k].astype(np.float64)
        finite = data[np.isfinite(data)]
        small, large = float(finite.min()), float(finite.max())          # dist.rs:133-148
        if small == large:
            small = large - 1.0
        offset = np.floor(small)                                        # :154
        scale = np.floor(CDF_RANGE / (large - offset))                  # :155
        with np.errstate(invalid="ignore"):
            x = (data - offset) * scale                                 # :158-163
            # f64::round rounds half away from zero (numpy's round is half-to-even)
            disc = np.where(np.isfinite(x), np.sign(x) * np.floor(np.abs(x) + 0.5), -np.inf)
        m = data.shape[0]
        size = m * CDF_RANGE + 1
        bg = np.asarray(pssm.background, dtype=np.float64)
        pdf_new = np.zeros(size)
        pdf_new[0] = 1.0
        for i in range(m):                                               # :173-191
            mx = i * CDF_RANGE
            pdf_old, pdf_new = pdf_new, np.zeros(size)
            for a in range(k):
                s = disc[i, a]
                if np.isfinite(s):                                       # `s != i32::MIN`
                    s = int(s)
                    pdf_new[s:s + mx + 1] += pdf_old[:mx + 1] * bg[a]
        # dist.rs:197-215: sf[i] = min(pdf[i] + sf[i+1], 1.0) from the top down.  The terms are >= 0, so the
        # clamp only ever holds a saturated tail at 1.0: a sequential f64 running sum (np.cumsum adds in
        # order) clipped afterwards is the same number at every index.  max_score = the highest index >= 1
        # with mass, min_score = the lowest index <= size - 2 with mass (0 when there is none).
        nz = np.nonzero(pdf_new > 0.0)[0]
        hi, lo = nz[nz >= 1], nz[nz <= size - 2]
        max_score = int(hi[-1]) if hi.size else 0
        min_score = int(lo[0]) if lo.size else 0
        sf = np.minimum(np.cumsum(pdf_new[::-1])[::-1], 1.0)
        self._scale, self._offset, self._rows = float(scale), int(offset), m
        self.sf, self.min_score, self.max_score = sf, min_score, max_score

    def scale(self, score: float) -> int:
        """dist.rs:77-81"""
        x = (float(np.float32(score)) - self._rows * self._offset) * self._scale
        return int(np.sign(x) * np.floor(abs(x) + 0.5))

    def unscale(self, score: int) -> float:
        """dist.rs:84-88 (f32 arithmetic)"""
        return float(np.float32(score) / np.float32(self._scale) + np.float32(self._rows * self._offset))

    def pvalue(self, score: float) -> float:
        """dist.rs:91-101"""
        scaled = self.scale(score)
        if scaled < self.min_score:
            return 1.0
        if scaled >= len(self.sf):
            return 0.0
        return float(self.sf[scaled])

    def score(self, pvalue: float) -> float:
        """dist.rs:104-116: binary search of the (descending) survival function."""
        if pvalue >= 1.0:
            return self.unscale(self.min_score)
        if pvalue <= 0.0:
            return self.unscale(self.max_score)
        lo, hi = 0, len(self.sf)                  # slice::binary_search_by with cmp = pvalue.partial_cmp(x)
        while lo < hi:
            mid = lo + (hi - lo) // 2
            x = self.sf[mid]
            if pvalue == x:
                return self.unscale(mid)
            if pvalue < x:                        # Ordering::Less -> search the right half
                lo = mid + 1
            else:
                hi = mid
        return self.unscale(lo)

    def min_pvalue(self) -> float:
        """dist.rs:125-127"""
        return float(self.sf[self.max_score])
