"""Host-side "next" rows: JASPAR-2016 reader (lightmotif-io/src/jaspar16) and the MEME-style
score distribution (lightmotif/src/pwm/dist.rs), against the literals of the reference's tests."""
import io

import numpy as np
import pytest

import lightmotif_amd as lm
from lightmotif_amd import dist as lmdist
from lightmotif_amd import io as lmio

RUNX1 = (">MA0001.1 RUNX1\n"
         "A [10 12  4  1  2  2  0  0  0  8 13 ]\n"
         "C [ 2  2  7  1  0  8  0  0  1  2  2 ]\n"
         "G [ 3  1  1  0 23  0 26 26  0  0  4 ]\n"
         "T [11 11 14 24  1 16  0  0 25 16  7 ]\n")


def test_jaspar16_single_and_multi():
    """jaspar16/mod.rs:146-174"""
    recs = list(lmio.read(io.StringIO(RUNX1)))
    assert len(recs) == 1 and recs[0].id == "MA0001.1" and recs[0].description == "RUNX1"
    assert len(recs[0].matrix) == 11
    assert recs[0].matrix[0] == [10, 2, 11, 3, 0]            # A C T G N order (abc.rs:106-108)
    recs = list(lmio.read(io.StringIO(RUNX1 + RUNX1.replace("MA0001.1", "MA0002.1"))))
    assert [r.id for r in recs] == ["MA0001.1", "MA0002.1"]


def test_jaspar16_tab_header_like_the_fixture():
    """lightmotif-io/tests/jaspar16.rs:5-14 (MA0001.3.pfm uses a TAB and padded brackets)"""
    text = (">MA0001.3\tAGL3\n"
            "A  [     0      0     82     40     56     35     65     25     64      0 ]\n"
            "C  [    92     79      1      4      0      0      1      4      0      0 ]\n"
            "G  [     0      0      2      3      1      0      4      3     28     92 ]\n"
            "T  [     3     16     10     48     38     60     25     63      3      3 ]\n")
    (rec,) = list(lmio.read(io.StringIO(text)))
    assert rec.id == "MA0001.3" and rec.description == "AGL3" and len(rec.matrix) == 10


def test_jaspar16_errors():
    with pytest.raises(ValueError, match="duplicate"):       # parse.rs:62-64
        list(lmio.read(io.StringIO(">X\nA [1 2]\nA [1 2]\n")))
    with pytest.raises(ValueError, match="inconsistent"):    # parse.rs:66-68
        list(lmio.read(io.StringIO(">X\nA [1 2]\nC [1 2 3]\n")))
    with pytest.raises(ValueError, match="invalid symbol"):
        list(lmio.read(io.StringIO(">X\nB [1 2]\n")))


def ma0045():
    """dist.rs:246-270 / test_pvalue.py:8-16"""
    return lm.CountMatrix({
        "A": [3, 7, 9, 3, 11, 11, 11, 3, 4, 3, 8, 8, 9, 9, 11, 2],
        "C": [5, 0, 1, 6, 0, 0, 0, 3, 1, 4, 5, 1, 0, 5, 0, 7],
        "T": [2, 4, 3, 1, 0, 1, 1, 6, 1, 1, 0, 1, 3, 0, 0, 5],
        "G": [4, 3, 1, 4, 3, 2, 2, 2, 8, 6, 1, 4, 2, 0, 3, 0],
    }).normalize(pseudocount=0.25).log_odds()


def almost(x, y, places):
    return round(x * 10 ** places) == round(y * 10 ** places)


def test_score_distribution_pvalue_literals():
    """dist.rs:272-280 and lightmotif-py test_pvalue.py:18-20 (method="meme")"""
    cdf = lmdist.ScoreDistribution(ma0045())
    assert almost(cdf.pvalue(8.89385), 0.0003, 5)
    assert almost(cdf.pvalue(12.66480), 0.00001, 5)
    assert almost(cdf.pvalue(17.71508), 1e-9, 9)
    assert abs(cdf.pvalue(8.7708) - 0.00032910) < 5e-6


def test_score_distribution_score_literals():
    """dist.rs:282-292 and test_pvalue.py:22-24"""
    cdf = lmdist.ScoreDistribution(ma0045())
    assert almost(cdf.score(0.00001), 12.66480, 5)
    assert almost(cdf.score(0.0003), 8.89385, 5)
    assert almost(cdf.score(1e-9), 17.71508, 4)
    assert abs(cdf.score(0.00033) - 8.765) < 5e-4
    assert cdf.pvalue(-1e9) == 1.0 and cdf.pvalue(1e9) == 0.0
    assert cdf.score(1.0) == cdf.unscale(cdf.min_score) and cdf.score(0.0) == cdf.unscale(cdf.max_score)


def test_jaspar2024_core_fixture():
    """The reference's bench fixture (lightmotif-io/benches/JASPAR2024.pwm, committed gzipped as
    data): 2 346 DNA count matrices whose length histogram is the one SURVEY.md 8(d) quotes for
    configs[2]; every record must parse and convert to a PSSM like the CLI does (main.rs:473-478)."""
    from pathlib import Path
    records = list(lmio.read(Path(__file__).parent / "golden" / "JASPAR2024.pwm.gz"))
    assert len(records) == 2346
    hist = {}
    for r in records:
        hist[len(r.matrix)] = hist.get(len(r.matrix), 0) + 1
    assert hist == {4: 21, 5: 48, 6: 284, 7: 347, 8: 454, 9: 280, 10: 281, 11: 157, 12: 97, 13: 99,
                    14: 85, 15: 72, 16: 40, 17: 21, 18: 14, 19: 19, 20: 8, 21: 11, 22: 1, 24: 2,
                    29: 2, 30: 1, 31: 1, 33: 1}
    assert sum(k * v for k, v in hist.items()) == 22090
    assert records[0].id == "MA0004.1" and records[0].description == "Arnt"
    first = records[0].matrix
    assert first.data[:, :4].tolist()[0] == [4, 16, 0, 0]          # A C T G columns of position 0
    pssm = first.normalize(0.1).log_odds()
    assert np.isfinite(pssm.data[:, :4]).all() and np.isneginf(pssm.data[:, 4]).all()
