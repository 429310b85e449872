"""The short form of the hit-list ordering (csrc/hits.hip, ShortOrder): one job, a short list expected -- the re-scoring
kernel counts the records per bucket, two launches order them.  What can go wrong there is state: the bucket counts and
cursors live in the context and must be zero between calls, the geometry is guessed from the PREVIOUS call's count, and
a guess that is far off (or hits that cluster) aborts the ranking and sends the call to the exact form.  So: one context,
calls whose hit counts jump across the form's limit in both directions, clustered hits, both key orders -- every result
against the materialised route (store kernel + Threshold on the stored matrix, which the oracle suites pin bit for bit),
and one case against the oracle itself."""
import numpy as np
import pytest
import torch

import lightmotif_amd as lm

pytestmark = pytest.mark.gpu
COLS = 32


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def motif(m, seed):
    rng = np.random.default_rng(seed)
    return lm.create(["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(8)]).counts.normalize(0.1).log_odds()


@pytest.mark.parametrize("poll_done", [1, 0])
def test_hit_counts_jumping_across_the_short_form_in_one_context(poll_done):
    """(`poll_done`: the end of a short-form call is read from the word the ranking kernel's last workgroup raises in pinned
    memory -- the shipped form -- or waited for on the stream; both walk the same sequence of list sizes.)"""
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    length, m = 60_000_000, 20
    rows = -(-length // COLS)
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    seq = torch.empty((rows + m - 1, COLS), dtype=torch.uint8, device=dev)
    seq[:rows] = torch.randint(0, 4, (rows, COLS), dtype=torch.uint8, device=dev, generator=gen)
    # clustered hits: a stretch of one column repeats the consensus k-mer back to back (every m-th row scores the maximum,
    # thousands of hits in a handful of buckets)
    pssm = motif(m, 5)
    consensus = torch.tensor(np.argmax(pssm.data[:, :4], axis=1).astype(np.uint8), device=dev)
    seq[100_000:100_000 + 2_000 * m, 7] = consensus.repeat(2_000)
    pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
    pli.set_option("poll_done", poll_done)
    ref = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
    pli.configure_wrap_dptr(seq.data_ptr(), rows, COLS, COLS, m - 1, 4)
    out = torch.empty((rows, COLS), dtype=torch.float32, device=dev)
    ref.score_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, out.data_ptr(), COLS)
    torch.cuda.synchronize()
    flat = out.flatten()
    k_of = lambda n: float(torch.topk(flat[: 40_000_000], n).values[-1])  # noqa: E731  (threshold leaving ~n * 1.5 hits)
    seen = set()
    # expected hits ~: 3 k, 30 k (long form: the guess was short), 30 k again (short form, 25 k..40 k), 300 k (abort ->
    # exact form), 3 k (guess far too long), 3 k, 0 hits, 3 k
    for n in (2_000, 20_000, 20_000, 200_000, 2_000, 2_000, 0, 2_000, 26_000, 27_000):
        t = k_of(n) if n else 1e9
        want_rc = ref.threshold_dptr(out.data_ptr(), rows, COLS, COLS, t)
        got_rc, got_v = pli.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, t)
        assert np.array_equal(np.asarray(got_rc).reshape(-1, 2), np.asarray(want_rc).reshape(-1, 2)), (n, len(got_rc), len(want_rc))
        if len(got_rc):
            rc = torch.as_tensor(np.asarray(got_rc).reshape(-1, 2).astype(np.int64), device=dev)
            assert np.array_equal(bits(got_v), bits(out[rc[:, 0], rc[:, 1]].cpu().numpy())), n
        seen.add(pli.last_kernel)
    assert any(k.startswith("score_c32_prefilter2") for k in seen)


def test_scanner_positions_through_the_short_form_match_the_oracle(pli, oracle):
    co = oracle
    rng = np.random.default_rng(31)
    enc = rng.integers(0, 4, 3_000_017, dtype=np.uint8)
    enc[500_000:500_400] = 0          # a homopolymer: hits in consecutive positions
    pssm = motif(12, 9)
    seq = pli.stripe(lm.EncodedSequence(enc))
    seq.configure(pssm)
    ref = co.stripe(enc, COLS, 5)
    co.configure_wrap(ref, 11)
    p = co.aligned_empty(pssm.data.shape, np.float32)
    p[:] = pssm.data
    want = co.avx2_score_rows(ref, p, threads=8)
    by_pos = want[:, :COLS].T.reshape(-1)[: len(enc) - 12 + 1]
    for frac in (2e-4, 3e-3, 2e-4, 1e-5):       # ~600, 9 000, 600, 30 hits: the guess is wrong in both directions
        t = float(np.partition(by_pos, int(len(by_pos) * (1 - frac)))[int(len(by_pos) * (1 - frac))])
        want_pos = np.nonzero(by_pos >= np.float32(t))[0]
        sc = lm.Scanner(pssm, seq, threshold=t)
        assert sc.positions.tolist() == want_pos.tolist(), frac
        rc, vals = pli.score_threshold(pssm, seq, t)
        wrc = co.threshold(want, COLS, t)
        assert rc == [tuple(x) for x in wrc.tolist()]
        assert np.array_equal(bits(vals), bits(want[wrc[:, 0], wrc[:, 1]]))
