"""The drop-last form of a single pair scan (csrc/score_threshold.hip; lm_hip_pssm::d_image2_drop): motifs of M = 20, 24, ... 36
rows are looked up over their first M - 1 rows and the last row is credited with its best weight.  The prefilter then flags
MORE -- never less -- and every candidate is re-scored over all M rows, so hits, values and the argmax must not change:
both settings of the context option "drop_last" against the materialised route (store kernel + Threshold / argmax on the
stored matrix, which the oracle suites pin bit for bit), on a sequence with N runs and a tract of the consensus, for every
such length and its neighbours, at thresholds from a handful of hits to tens of thousands; one case against the oracle."""
import numpy as np
import pytest
import torch

import lightmotif_amd as lm

pytestmark = pytest.mark.gpu
COLS = 32


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def motif(m, seed, flat_last=False):
    rng = np.random.default_rng(seed)
    sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(8)]
    if flat_last:   # a last column without information: the form's best case
        sites = [s[:-1] + "ACTG"[i % 4] for i, s in enumerate(sites)]
    return lm.create(sites).counts.normalize(0.1).log_odds()


def pipeline(**options):
    p = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
    for k, v in options.items():
        p.set_option(k, v)
    return p


@pytest.mark.parametrize("m", [7, 8, 9, 12, 16, 19, 20, 21, 24, 28, 32, 36])
def test_both_forms_give_the_materialised_hits_and_argmax(m):
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    length = 24_000_000 if m <= 12 else 120_000_000      # (the candidate route of the fused argmax starts at 100 M cells)
    rows = -(-length // COLS)
    gen = torch.Generator(device=dev)
    gen.manual_seed(900 + m)
    seq = torch.empty((rows + m - 1, COLS), dtype=torch.uint8, device=dev)
    seq[:rows] = torch.randint(0, 4, (rows, COLS), dtype=torch.uint8, device=dev, generator=gen)
    seq[5_000:9_000, 3] = 4                                                    # a run of N
    for flat_last in (False, True):
        pssm = motif(m, 31 * m + flat_last, flat_last)
        consensus = torch.tensor(np.argmax(pssm.data[:, :4], axis=1).astype(np.uint8), device=dev)
        seq[200_000:200_000 + 50 * m, 11] = consensus.repeat(50)              # clustered hits, the best score among them
        plis = {"drop": pipeline(drop_last=1), "full": pipeline(drop_last=0)}
        ref = pipeline()
        ref.configure_wrap_dptr(seq.data_ptr(), rows, COLS, COLS, m - 1, 4)
        out = torch.empty((rows, COLS), dtype=torch.float32, device=dev)
        ref.score_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, out.data_ptr(), COLS)
        torch.cuda.synchronize()
        flat = out.flatten()
        want_am = ref.argmax_dptr(out.data_ptr(), rows, COLS, COLS)
        top = torch.topk(flat[: 20_000_000], 30_000).values
        for n in (20, 3_000, 30_000):
            t = float(top[n - 1])
            want = ref.threshold_dptr(out.data_ptr(), rows, COLS, COLS, t)
            for name, p in plis.items():
                got_rc, got_v = p.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows, t)
                assert np.array_equal(np.asarray(got_rc).reshape(-1, 2), np.asarray(want).reshape(-1, 2)), (m, flat_last, n, name, p.last_kernel)
                rc = torch.as_tensor(np.asarray(got_rc).reshape(-1, 2).astype(np.int64), device=dev)
                assert np.array_equal(bits(got_v), bits(out[rc[:, 0], rc[:, 1]].cpu().numpy())), (m, n, name)
        for name, p in plis.items():
            got = p.score_argmax_dptr(pssm, seq.data_ptr(), rows + m - 1, COLS, COLS, m - 1, length, 0, rows)
            assert got is not None and (got[0], np.float32(got[1])) == (want_am[0], np.float32(want_am[1])), (m, flat_last, name, p.last_kernel)


def test_drop_last_form_against_the_oracle_with_positions(pli, oracle):
    co = oracle
    rng = np.random.default_rng(77)
    enc = rng.integers(0, 5, 2_500_003, dtype=np.uint8)          # N anywhere
    for m in (8, 20):
        pssm = motif(m, 5 + m)
        seq = pli.stripe(lm.EncodedSequence(enc))
        seq.configure(pssm)
        ref = co.stripe(enc, COLS, 5)
        co.configure_wrap(ref, m - 1)
        p = co.aligned_empty(pssm.data.shape, np.float32)
        p[:] = pssm.data
        want = co.avx2_score_rows(ref, p, threads=8)
        by_pos = want[:, :COLS].T.reshape(-1)[: len(enc) - m + 1]
        finite = by_pos[np.isfinite(by_pos)]
        for frac in (1e-3, 1e-5):
            t = float(np.partition(finite, int(len(finite) * (1 - frac)))[int(len(finite) * (1 - frac))])
            sc = lm.Scanner(pssm, seq, threshold=t)
            assert sc.positions.tolist() == np.nonzero(by_pos >= np.float32(t))[0].tolist(), (m, frac)
            rc, vals = pli.score_threshold(pssm, seq, t)
            wrc = co.threshold(want, COLS, t)
            assert rc == [tuple(x) for x in wrc.tolist()] and np.array_equal(bits(vals), bits(want[wrc[:, 0], wrc[:, 1]]))
        assert pli.score_argmax(pssm, seq)[0] == co.argmax(want, COLS)
