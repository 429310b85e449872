"""Minimal command-line driver with the shape of ``lightmotif-cli``:

    python -m lightmotif_amd.scan_cli -m motifs.pwm[.gz] -s genome.fa[.gz] -o hits.tsv -P 1e-5

Behaviour follows lightmotif-cli/src/main.rs:
  * matrices: JASPAR-2016 count matrices -> ``to_freq(0.1).to_scoring(None)`` (main.rs:473-478);
  * threshold per motif: ``--pvalue`` through the MEME-style score distribution, or
    ``--rel-threshold`` (x max score) or ``--abs-threshold``; default p = 1e-5 (main.rs:479-489);
  * every FASTA record is encoded lossily, striped and given ``max_m`` wrap rows
    (main.rs:540-546) -- here on the device, from the raw text;
  * the reference fans (motif, sequence) pairs out to worker threads (main.rs:554-561); here
    all motifs of one record go to the GPU as ONE batched fused scan;
  * output: TSV ``seq_index seq_name motif_index motif_name pos strand score pvalue`` with
    1-based indices and the p-value in exponent notation (main.rs:527-531, 587-600).
    The reference writes hits in worker-completion order; this driver writes them grouped by
    sequence, then motif, then position.
``--reverse`` also scans the reverse-complement matrix and reports strand ``-``
(main.rs:343-362).  There is no CPU path: without a gfx950 device the scan fails.
"""
from __future__ import annotations

import argparse
import gzip
import sys
from typing import Iterator, List, Optional, Sequence, TextIO, Tuple

import numpy as np

from . import io as lmio
from .lib import Pipeline, ScoringMatrix, StripedSequence


def _open_text(path: str) -> TextIO:
    with open(path, "rb") as fh:                     # main.rs:424-436: sniff the gzip magic
        magic = fh.read(2)
    return gzip.open(path, "rt") if magic == b"\x1f\x8b" else open(path, "r")


def read_fasta(handle: TextIO) -> Iterator[Tuple[str, str]]:
    """(name, sequence) per record; name = the header up to the first whitespace."""
    name, chunks = None, []
    for line in handle:
        if line.startswith(">"):
            if name is not None:
                yield name, "".join(chunks)
            head = line[1:].strip()
            name, chunks = (head.split()[0] if head else ""), []
        elif name is not None:
            chunks.append(line.strip())
    if name is not None:
        yield name, "".join(chunks)


def _fmt_score(x) -> str:                            # Rust `{}` of an f32: shortest round-trip digits
    return np.format_float_positional(np.float32(x), unique=True, trim="-")


def _fmt_exp(x) -> str:                              # Rust `{:e}` of an f32
    return np.format_float_scientific(np.float32(x), unique=True, trim="-", exp_digits=1).replace("e+", "e")


def thresholds_for(pssms: Sequence[ScoringMatrix], pvalue: Optional[float], rel: Optional[float],
                   absolute: Optional[float]) -> List[float]:
    out = []
    for p in pssms:
        if pvalue is not None:
            out.append(p.score_for_pvalue(pvalue))
        elif rel is not None:
            out.append(float(np.float32(p.max_score()) * np.float32(rel)))
        elif absolute is not None:
            out.append(float(absolute))
        else:
            out.append(p.score_for_pvalue(1e-5))
    return out


def scan_record(pli: Pipeline, seq: StripedSequence, pssms: Sequence[ScoringMatrix],
                thresholds: Sequence[float]):
    """Per motif: (positions ascending, scores) with ``score >= t`` and ``pos + M <= L`` (scan.rs:185-190)."""
    rows, length = seq.rows, len(seq)
    out = []
    for (coords, values), p in zip(pli.scan_threshold_batch(pssms, thresholds, seq), pssms):
        pos = coords[:, 1] * rows + coords[:, 0]
        keep = pos + len(p) <= length
        pos, values = pos[keep], values[keep]
        order = np.argsort(pos, kind="stable")
        out.append((pos[order], values[order]))
    return out


def main(argv: Optional[Sequence[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="lightmotif_amd.scan_cli", description=__doc__.split("\n\n")[0])
    ap.add_argument("-m", "--matrices", required=True, help="JASPAR-2016 count matrices (optionally gzipped)")
    ap.add_argument("-s", "--sequences", required=True, help="FASTA file (optionally gzipped)")
    ap.add_argument("-o", "--output", required=True, help="TSV file to write")
    group = ap.add_mutually_exclusive_group()
    group.add_argument("-P", "--pvalue", type=float)
    group.add_argument("--abs-threshold", type=float)
    group.add_argument("--rel-threshold", type=float)
    ap.add_argument("--reverse", action="store_true", help="also scan the reverse-complement matrices")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)

    print("Loading matrices")
    with _open_text(args.matrices) as fh:
        records = list(lmio.read(fh))
    lengths = [len(r.matrix) for r in records]
    print(f"Loaded {len(records)} matrices (M={min(lengths, default=0)}..{max(lengths, default=0)})")
    print("Preparing motifs")
    direct = [r.matrix.normalize(0.1).log_odds() for r in records]
    thresholds = thresholds_for(direct, args.pvalue, args.rel_threshold, args.abs_threshold)
    strands = [("+", direct)]
    if args.reverse:
        strands.append(("-", [p.reverse_complement() for p in direct]))
    max_m = max(lengths, default=0)

    pli = Pipeline.hip(args.device)
    n_hits = 0
    with open(args.output, "w") as out, _open_text(args.sequences) as fasta:
        out.write("seq_index\tseq_name\tmotif_index\tmotif_name\tpos\tstrand\tscore\tpvalue\n")
        for si, (name, text) in enumerate(read_fasta(fasta)):
            seq = pli.stripe_ascii(text, lossy=True)
            seq.configure_wrap(max_m)                                      # main.rs:543
            for strand, pssms in strands:
                for mi, (pos, scores) in enumerate(scan_record(pli, seq, pssms, thresholds)):
                    if len(pos) == 0:
                        continue
                    dist = direct[mi].score_distribution                  # main.rs:335: motif.dist
                    ident = records[mi].id
                    for p, s in zip(pos.tolist(), scores.tolist()):
                        out.write(f"{si + 1}\t{name}\t{mi + 1}\t{ident}\t{p}\t{strand}\t{_fmt_score(s)}\t"
                                  f"{_fmt_exp(dist.pvalue(s))}\n")
                    n_hits += len(pos)
    print(f"Wrote {n_hits} hits to {args.output}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
