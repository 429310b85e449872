"""Host-side logic of lightmotif_amd.lib that needs no device: encoding, count ->
weight -> scoring matrix construction (pwm/mod.rs:209-258, 376-431, 505-526), layout."""
import json
from pathlib import Path

import numpy as np
import pytest

import lightmotif_amd as lm
from lightmotif_amd import lib
from oracle import c_oracle as co

GOLD = json.loads((Path(__file__).parent / "golden" / "reference_vectors.json").read_text())


def test_encode_matches_reference_alphabets():
    g = GOLD["G9_encode"]
    e = lm.EncodedSequence(g["sequence"])
    assert str(e) == g["sequence"] and len(e) == 64
    assert np.array_equal(e.data, co.encode(g["sequence"]))
    with pytest.raises(lm.InvalidSymbol, match=r"'\.'"):
        lm.EncodedSequence(g["unknowns"])
    lossy = lm.EncodedSequence.encode_lossy(g["unknowns"])
    assert np.array_equal(lossy.data, co.encode(g["unknowns"], lossy=True))
    p = lm.EncodedSequence("ACDEFGHIKLMNPQRSTVWYX", protein=True)
    assert list(p.data) == list(range(21))
    with pytest.raises(lm.InvalidSymbol):
        lm.EncodedSequence("PILFFRLK")  # lib.rs module doc example: protein text as DNA


def test_encoded_sequence_protocol_like_test_sequence_py():
    """lightmotif-py tests/test_sequence.py:10-57 (TestEncodedSequence): len, index, IndexError, iteration,
    and the 1-D byte buffer (``memoryview`` there, ``np.asarray`` here -- a pure-Python class on 3.10)."""
    A, C, T, G, N = range(5)
    s1, s2 = lm.EncodedSequence("ATGC"), lm.EncodedSequence("ATGCTTAGATAC")
    assert (len(s1), len(s2)) == (4, 12)                                        # test_len
    assert [s1[i] for i in range(4)] == [A, T, G, C]                             # test_index
    assert [s2[i] for i in range(7)] == [A, T, G, C, T, T, A]
    assert s1[-1] == C
    with pytest.raises(IndexError):                                             # test_index_error
        _ = s1[10]
    mem = np.asarray(s1)                                                        # test_memoryview
    assert mem.shape == (4,) and mem.dtype == np.uint8 and list(mem) == [A, T, G, C]
    assert list(s1) == [A, T, G, C]                                             # test_iter
    assert str(s1.copy()) == "ATGC"


def test_matrix_protocols_of_the_python_module():
    """lightmotif-py lib.rs: CountMatrix / WeightMatrix / ScoringMatrix are sized, indexable by motif
    position (a row of K values), comparable, and carry `protein` (lib.rs:450-480, 568-598, 764-841)."""
    motif = lm.create(["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"])
    w = motif.counts.normalize(0.1)
    assert w == motif.counts.normalize(0.1) and w != motif.counts.normalize(0.2)
    p = w.log_odds()
    assert len(p) == 15 and len(p[0]) == 5 and p[-1] == p[14] and p[0][4] == float("-inf")
    assert p[0][3] == float(np.float32(p.data[0, 3])) and not p.protein and not w.protein
    with pytest.raises(IndexError):
        _ = p[15]
    assert p == lm.ScoringMatrix({a: [row[i] for row in (p[j] for j in range(15))] for i, a in enumerate("ACTGN")})
    import copy
    e = lm.EncodedSequence("ATGC")
    assert str(copy.copy(e)) == "ATGC" and copy.copy(e).data is not e.data


def test_create_normalize_log_odds_equals_oracle_pssm():
    g = GOLD["G1_scores"]
    motif = lm.create(g["patterns"])
    assert motif.counts[0] == [0, 0, 0, 2, 0] and len(motif.counts) == 15
    pssm = motif.counts.normalize(g["pseudocount"]).log_odds()
    want = co.pssm_from_sites([co.encode(p) for p in g["patterns"]], pseudocount=g["pseudocount"])
    assert pssm.data.shape == want.shape == (15, 8)          # DenseMatrix<f32, U5> stride 8
    assert np.array_equal(pssm.data.view(np.uint32), want.view(np.uint32))
    assert np.all(np.isneginf(pssm.data[:, 4]))              # pwm/mod.rs:422-423
    # pseudocount 0 -> -inf for unseen symbols (tests/scan.rs:47-60 uses it)
    z = motif.pssm
    assert np.isneginf(z.data[0, 0]) and z.data[0, 3] == np.float32(2.0)


def test_protein_pssm_and_layout():
    sites = ["PILFFRLK", "KDMLKEYL", "PFRLTHKL"]
    motif = lm.create(sites, protein=True)
    pssm = motif.counts.normalize(0.1).log_odds()
    want = co.pssm_from_sites([co.encode(s, "P") for s in sites], k=21, pseudocount=0.1)
    assert pssm.data.shape == (8, 24)
    assert np.array_equal(pssm.data.view(np.uint32), want.view(np.uint32))


def test_scoring_matrix_from_dict_and_reverse_complement():
    sm = lm.ScoringMatrix({"A": [1, 2], "C": [3, 4], "T": [5, 6], "G": [7, 8]})
    assert sm.data.shape == (2, 8) and sm.data[0, 4] == 0.0   # lib.rs:729-745 missing -> 0.0
    rc = sm.reverse_complement()
    assert list(rc.data[0, :4]) == [6, 8, 2, 4]               # row 1 with A<->T, C<->G
    with pytest.raises(ValueError):
        lm.ScoringMatrix({"A": [1, 2], "C": [3]})


def test_stride_helper():
    for c in GOLD["G6_stride"]["cases"]:
        assert lib.stride(c["cols"], c["elem"]) == c["stride"]


def test_to_discrete_equals_the_oracle_restatement():
    """pwm/mod.rs:665-696, 777-785 (host side of `Score<u8, ..>`): weights, factor, offsets,
    scale / unscale, for the golden PSSM, a pseudocount-0 PSSM (-inf weights) and protein."""
    from oracle import np_oracle as no
    g = GOLD["G1_scores"]
    rng = np.random.default_rng(8)
    prot_sites = ["".join(lib.PROTEIN_SYMBOLS[i] for i in rng.integers(0, 20, 9)) for _ in range(6)]
    for pssm in (lm.create(g["patterns"]).counts.normalize(0.1).log_odds(),
                 lm.create(g["patterns"]).pssm,
                 lm.create(prot_sites, protein=True).counts.normalize(0.1).log_odds()):
        dm = pssm.to_discrete()
        w, factor, offsets, offset = no.to_discrete(pssm.data, pssm.k)
        assert dm.data.shape == (len(pssm), lib.stride(pssm.k, 1))
        assert np.array_equal(dm.data[:, :pssm.k], w) and not dm.data[:, pssm.k:].any()
        assert np.float32(dm.factor) == factor and np.float32(dm.offset) == offset
        assert np.array_equal(np.asarray(dm.offsets, np.float32), offsets)
        for t in (-30.0, -10.0, 0.0, 3.5, pssm.max_score(), 1e9, float("-inf")):
            assert dm.scale(t) == no.discrete_scale(t, factor, offset)
        for q in (0, 1, 100, 255):
            assert np.float32(dm.unscale(q)) == no.discrete_unscale(q, factor, offset)
        assert dm.scale(pssm.max_score()) <= 255


def test_bench_helpers_without_a_gpu(monkeypatch):
    """bench.py's launch plumbing: the host topology the cpu_baseline object states, and the command line a
    launcher-less `--gpus N` run re-executes itself with."""
    import importlib.util
    import subprocess
    import sys
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("lm_bench", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    topo = bench.cpu_topology()
    assert topo["sockets"] >= 1 and 1 <= topo["cores"] <= topo["threads"]
    # the committed PMC traffic figure names the store-kernel sources it was counted on (roofline.traffic_current)
    import json
    pmc = json.loads((Path(__file__).resolve().parent.parent / "profiles" / "pmc_traffic.json").read_text())
    digest = bench.store_kernel_digest()
    assert len(digest) == 16 and len(pmc["store_kernel_digest"]) == 16
    if pmc["store_kernel_digest"] != digest:
        import warnings
        warnings.warn("profiles/pmc_traffic.json was counted on other store-kernel sources: bench.py reports "
                      "roofline.traffic_current = false until tools/collect_pmc.sh is re-run")
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 0
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    with pytest.raises(SystemExit) as e:
        bench.self_launch(4)
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_roofline_blocks_without_a_gpu():
    """The roofline arithmetic bench.py prints for the extras: the fused forms and configs[2] are priced against
    the LDS-gather ceiling with (M | 3) + 1 bytes of pair table per scanned (motif, position) cell, the 1 B per position of
    HBM reads beside it; the best-k-mer bound that decides which motifs cannot reach a threshold."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("lm_bench2", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows, m = 31_250_000, 20
    # what the scan looked up comes from the library (lm_hip_ctx_last_scan_info): a single scan of M = 20, 24, ... 36 may look
    # M - 1 rows up (the drop-last form): 19 rows -> 20 B per position; the call's tail = call - scan kernel
    fr = bench.fused_roofline(0.25, rows, m, "score_c32_prefilter2", kernel_ms=0.19, scan_info=(19, 20))
    rf = fr["roofline"]
    assert rf["bound"] == "lds" and rf["motif_rows_scanned"] == 19 and rf["lds_bytes_per_position"] == 20 and rf["peak"] == 157.3
    assert fr["tail_us"] == 60.0 and "timing" in fr
    # nothing reported (no scan kernel of the prefilter families ran): the motif's own pair table
    assert bench.fused_roofline(0.25, rows, m, "x")["roofline"]["lds_bytes_per_position"] == 24
    assert abs(rf["achieved"] - 20 * 1e9 / 0.25e-3 / 1e12) < 0.01 and abs(rf["frac"] - rf["achieved"] / 157.2864) < 1e-3
    r15 = bench.fused_roofline(0.25, rows, 15, "score_c32_prefilter2", kernel_ms=0.2, scan_info=(15, 16))["roofline"]
    assert r15["motif_rows_scanned"] == 15 and r15["lds_bytes_per_position"] == 16 and r15["kernel_ms"] == 0.2
    assert abs(r15["kernel_frac"] - 16 * 1e9 / 0.2e-3 / 157.2864e12) < 1e-3
    assert abs(rf["hbm_read_frac"] - (1e9 / 0.25e-3 / 1e9) / 8000.0) < 1e-4 and fr["Gpos_s"] == 4000.0
    lr = bench.lds_roofline(157.2864e12 * 0.5, 1.0, "model")
    assert lr["bound"] == "lds" and lr["frac"] == 0.5 and lr["unit"] == "TB/s"
    # (M | 3) + 1: the pair scan pads the motif to 3 (mod 4) rows and reads one u16 per padded row + 1, per PAIR of input rows
    assert [(mm | 3) + 1 for mm in (4, 7, 8, 11, 12, 20, 33)] == [8, 8, 12, 12, 16, 24, 36]
    pssm = lm.create(["ACGT", "ACGA", "ACCT"]).counts.normalize(0.1).log_odds()
    want = np.float32(0)
    for row in pssm.data[:, :4]:                     # the sequential f32 sum of the row maxima, in motif order
        want = np.float32(want + row.max())
    assert bench.best_kmer_score(pssm) == want


def test_no_built_artefact_is_tracked():
    """History stays source-only: nothing git tracks is an ELF object / executable / shared library (built `.so` files and
    the kbench tools travel to the GPU box untracked)."""
    import subprocess
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run(["git", "ls-files", "-z"], cwd=root, capture_output=True)
    if r.returncode != 0 or not r.stdout:
        pytest.skip("not a git checkout")
    bad = []
    for name in r.stdout.decode().split("\0"):
        p = root / name
        if name and p.is_file():
            with open(p, "rb") as f:
                if f.read(4) == b"\x7fELF":
                    bad.append(name)
    assert not bad, bad


def test_batch_hits_cuts_views_on_access():
    """`Pipeline.scan_threshold_batch` returns ONE pair of arrays and the per-motif counts; `BatchHits` is the sequence of
    per-motif (coords, values) pairs the callers iterate over (lightmotif-cli main.rs:554-561 fans the same results out per
    motif), cut on access -- 2 346 slices per call cost the Python wrapper a millisecond the scan does not."""
    from lightmotif_amd.lib import BatchHits
    counts = np.array([2, 0, 3, 1], dtype=np.uintp)
    coords = np.arange(12, dtype=np.int64).reshape(6, 2)
    values = np.arange(6, dtype=np.float32)
    b = BatchHits(coords, values, counts)
    assert len(b) == 4 and b.total == 6
    assert [len(c) for c, _ in b] == [2, 0, 3, 1]
    c2, v2 = b[2]
    assert c2.tolist() == [[4, 5], [6, 7], [8, 9]] and v2.tolist() == [2.0, 3.0, 4.0]
    assert b[-1][1].tolist() == [5.0] and b[1][0].shape == (0, 2)
    assert [v.tolist() for _, v in b[1:3]] == [[], [2.0, 3.0, 4.0]]
    assert c2.base is not None                       # a view, not a copy
    with pytest.raises(IndexError):
        b[4]
    assert sum(len(c) for c, _ in b) == b.total      # what bench.py's hit count does
