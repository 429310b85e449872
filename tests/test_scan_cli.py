"""The CLI-shaped driver (lightmotif_amd/scan_cli.py <- lightmotif-cli/src/main.rs): host pieces on
the CPU, the whole pipeline against the oracle on the GPU."""
import gzip
import io

import numpy as np
import pytest

import lightmotif_amd as lm
from lightmotif_amd import scan_cli

MATRICES = (">MA0001.1\tFIRST\n"
            "A  [ 10 12  4  1  2  2  0  0 ]\n"
            "C  [  2  2  7  1  0  8  0  0 ]\n"
            "G  [  3  1  1  0 23  0 26 26 ]\n"
            "T  [ 11 11 14 24  1 16  0  0 ]\n"
            ">MA0002.1\tSECOND\n"
            "A  [ 20  0  0  5  9 ]\n"
            "C  [  0 20  0  5  1 ]\n"
            "G  [  0  0 20  5  1 ]\n"
            "T  [  0  0  0  5  9 ]\n")


def test_fasta_reader_and_number_formats():
    text = ">chr1 some description\nACGT\nNNAC\n\n>chr2\nGGG\n>empty\n"
    assert list(scan_cli.read_fasta(io.StringIO(text))) == [("chr1", "ACGTNNAC"), ("chr2", "GGG"), ("empty", "")]
    # Rust `{}` / `{:e}` of f32 values (main.rs:587-600)
    assert scan_cli._fmt_score(np.float32(-5.50167)) == "-5.50167"
    assert scan_cli._fmt_score(12.5) == "12.5" and scan_cli._fmt_score(1.0) == "1"
    assert scan_cli._fmt_exp(0.00032910) == "3.291e-4"
    assert scan_cli._fmt_exp(1.0) == "1e0" and scan_cli._fmt_exp(12.5) == "1.25e1"


def test_threshold_selection_follows_the_cli():
    pssms = [r.matrix.normalize(0.1).log_odds() for r in lm.io.read(io.StringIO(MATRICES))]
    by_p = scan_cli.thresholds_for(pssms, 1e-3, None, None)
    assert by_p == [p.score_for_pvalue(1e-3) for p in pssms]
    assert scan_cli.thresholds_for(pssms, None, None, None) == [p.score_for_pvalue(1e-5) for p in pssms]
    rel = scan_cli.thresholds_for(pssms, None, 0.5, None)
    assert rel == [float(np.float32(p.max_score()) * np.float32(0.5)) for p in pssms]
    assert scan_cli.thresholds_for(pssms, None, None, 3.25) == [3.25, 3.25]
    # max / min score: per-position extremes over A, C, T, G (pwm/mod.rs:592-615)
    for p in pssms:
        assert np.isclose(p.max_score(), p.data[:, :4].max(axis=1).sum(), rtol=1e-6)
        assert np.isclose(p.min_score(), p.data[:, :4].min(axis=1).sum(), rtol=1e-6)
        assert p.min_score() < 0 < p.max_score()


@pytest.mark.gpu
@pytest.mark.parametrize("reverse", [False, True])
def test_cli_end_to_end_against_the_oracle(tmp_path, reverse):
    from oracle import c_oracle as co
    rng = np.random.default_rng(11)
    seqs = {"chrA": "".join(rng.choice(list("ACGT"), 20_011)), "chrB": "".join(rng.choice(list("ACGTN"), 7_777))}
    fasta = tmp_path / "genome.fa.gz"
    with gzip.open(fasta, "wt") as fh:
        for name, s in seqs.items():
            fh.write(f">{name} test record\n")
            for i in range(0, len(s), 70):
                fh.write(s[i:i + 70] + "\n")
    mats = tmp_path / "motifs.pwm"
    mats.write_text(MATRICES)
    out = tmp_path / "hits.tsv"
    argv = ["-m", str(mats), "-s", str(fasta), "-o", str(out), "-P", "1e-3"] + (["--reverse"] if reverse else [])
    assert scan_cli.main(argv) == 0

    records = list(lm.io.read(io.StringIO(MATRICES)))
    direct = [r.matrix.normalize(0.1).log_odds() for r in records]
    want = []
    for si, (name, s) in enumerate(seqs.items()):
        enc = lm.EncodedSequence(s, lossy=True).data
        for strand in ("+", "-") if reverse else ("+",):
            for mi, p in enumerate(direct):
                q = p if strand == "+" else p.reverse_complement()
                t = np.float32(p.score_for_pvalue(1e-3))
                st = co.stripe(enc, 32, 5)
                co.configure_wrap(st, 8)
                scores, _ = co.score_rows(st, q.data)
                rows = scores.shape[0]
                by_pos = scores[:, :32].T.reshape(-1)[: len(s) - len(p) + 1]
                for pos in np.nonzero(by_pos >= t)[0]:
                    want.append((si + 1, name, mi + 1, records[mi].id, int(pos), strand,
                                 scan_cli._fmt_score(by_pos[pos]),
                                 scan_cli._fmt_exp(p.score_distribution.pvalue(float(by_pos[pos])))))
    lines = out.read_text().splitlines()
    assert lines[0].split("\t") == ["seq_index", "seq_name", "motif_index", "motif_name", "pos", "strand",
                                    "score", "pvalue"]
    got = [tuple(int(x) if i in (0, 2, 4) else x for i, x in enumerate(l.split("\t"))) for l in lines[1:]]
    assert len(want) > 20
    assert got == want
