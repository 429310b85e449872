"""JASPAR 2016 ("jaspar16") count-matrix reader -- host-side text parsing that feeds
the GPU path.  Behaviour follows the reference's
lightmotif-io/src/jaspar16/{mod,parse}.rs:

    >MA0001.3\tAGL3                     header: id, optional description (parse.rs:84-105)
    A  [ 0  0 82 40 ... ]               one line per symbol, counts in brackets (parse.rs:40-51)

A symbol may not appear twice and all rows must have the same length
(parse.rs:53-78, ``InvalidData``).  Missing symbols (``N``) count zero.
"""
from __future__ import annotations

import gzip
import os
import re
from dataclasses import dataclass
from typing import Iterator, Optional, TextIO, Union

import numpy as np

from .lib import CountMatrix, _symbols

_ROW = re.compile(r"^\s*(\S)\s+\[\s*([0-9\s]*?)\s*\]\s*$")


@dataclass
class Record:
    """jaspar16/mod.rs:33-65"""
    id: str
    description: Optional[str]
    matrix: CountMatrix


def read(file: Union[str, TextIO], *, protein: bool = False) -> Iterator[Record]:
    """Iterate over the records of a JASPAR-2016 formatted file (jaspar16/mod.rs:139)."""
    close = False
    if isinstance(file, (str, os.PathLike)):
        path = os.fspath(file)
        file = gzip.open(path, "rt") if path.endswith(".gz") else open(path, "r")
        close = True
    try:
        header, rows = None, []
        for line in file:
            if line.startswith(">"):
                if header is not None:
                    yield _build(header, rows, protein)
                header, rows = line, []
            elif line.strip():
                if header is None:
                    raise ValueError("matrix row before the first header")
                rows.append(line)
        if header is not None:
            yield _build(header, rows, protein)
    finally:
        if close:
            file.close()


def _build(header: str, rows: list, protein: bool) -> Record:
    parts = header[1:].strip().split(None, 1)   # id = up to the first whitespace (parse.rs:86-90)
    if not parts:
        raise ValueError("empty record header")
    ident = parts[0]
    desc = parts[1].strip() if len(parts) > 1 and parts[1].strip() else None
    sym = _symbols(protein)
    if not rows:
        raise ValueError(f"record {ident} has no matrix")
    counts, done, length = None, set(), None
    for line in rows:
        m = _ROW.match(line)
        if not m:
            raise ValueError(f"record {ident}: cannot parse matrix row {line!r}")
        letter, body = m.group(1), m.group(2)
        if letter not in sym:
            raise ValueError(f"record {ident}: invalid symbol {letter!r}")
        vals = [int(x) for x in body.split()]
        if letter in done:                       # parse.rs:62-64
            raise ValueError(f"record {ident}: duplicate symbol {letter!r}")
        if length is None:
            length = len(vals)
            counts = np.zeros((length, len(sym)), dtype=np.uint32)
        elif len(vals) != length:                # parse.rs:66-68
            raise ValueError(f"record {ident}: inconsistent row length")
        counts[:, sym.index(letter)] = vals
        done.add(letter)
    return Record(ident, desc, CountMatrix(counts, protein=protein))
