"""Out-of-range writes: the device-pointer entry points write into the CALLER's memory (a Rust
shim hands over `scores.matrix_mut()[0].as_mut_ptr()`, avx2.rs:115-119), so every kernel family
is run with its output embedded in a larger buffer of canary words -- before, after, and in the
alignment padding between `cols` and the row stride (the reference leaves that padding alone,
pli/mod.rs:103) -- and the canaries must survive while the payload equals the oracle's."""
import numpy as np
import pytest
import torch

import lightmotif_amd as lm
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu

CANARY_F32 = 0x7FC0DEAD      # a NaN payload no score can produce
CANARY_U8 = 0xA5
PAD_ROWS = 64


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def make_case(pli, rng, cols, m, k, length):
    enc = rng.integers(0, k, length, dtype=np.uint8)
    enc[rng.random(length) < 0.9] %= (k - 1)
    p = np.zeros((m, co.stride(k, 4)), np.float32)
    p[:, :k] = rng.normal(0, 2, (m, k))
    p[:, k - 1] = -np.inf
    ref = co.stripe(enc, cols, k)
    co.configure_wrap(ref, m - 1)
    dev = torch.device("cuda", 0)
    seq = torch.from_numpy(ref.data.copy()).to(dev)
    return enc, p, ref, seq


# (cols, M, K, length, out_stride or None = the reference's stride): every store kernel family --
# unrolled C = 32 (plain, padded-to-4, quad loads), sliced (in-place continuation + cell tail), tiled, generic
STORE_CASES = [
    (32, 20, 5, 300_007, None), (32, 15, 5, 300_007, None), (32, 7, 5, 50_001, None), (32, 33, 5, 200_003, None),
    (32, 36, 5, 150_001, None), (32, 40, 5, 400_009, None), (32, 100, 5, 400_009, None), (32, 12, 21, 200_003, None),
    (32, 1, 5, 10_007, None), (32, 20, 5, 300_007, 40), (16, 20, 5, 200_003, None), (16, 9, 5, 100_003, 24),
    (1, 15, 5, 20_011, None), (33, 12, 5, 100_003, None), (4, 64, 5, 50_021, None), (8, 40, 21, 50_021, 16),
    (32, 20, 5, 700, None), (32, 70, 5, 3_000, None),
]


@pytest.mark.parametrize("cols,m,k,length,ostride", STORE_CASES)
def test_score_rows_into_writes_only_its_cells(pli, cols, m, k, length, ostride):
    rng = np.random.default_rng(cols * 7919 + m * 31 + k)
    enc, p, ref, seq = make_case(pli, rng, cols, m, k, length)
    pssm = lm.ScoringMatrix(p, protein=k == 21)
    ostride = co.stride(cols, 4) if ostride is None else ostride
    dev = seq.device
    for a, b in ((0, ref.rows), (3, ref.rows - 5), (ref.rows // 2, ref.rows // 2 + 1)):
        n = b - a
        buf = torch.full(((n + 2 * PAD_ROWS) * ostride + 8,), CANARY_F32, dtype=torch.int32, device=dev)
        out = buf[PAD_ROWS * ostride: (PAD_ROWS + n) * ostride]
        torch.cuda.synchronize()        # the canary fill runs on torch's stream, the kernels on the context's own
        orow, _ = pli.score_dptr(pssm, seq.data_ptr(), seq.shape[0], seq.shape[1], cols, m - 1, length, a, b,
                                 out.data_ptr(), ostride)
        torch.cuda.synchronize()
        assert orow == n
        host = buf.cpu().numpy().view(np.uint32)
        assert (host[: PAD_ROWS * ostride] == CANARY_F32).all(), (pli.last_kernel, "wrote before the matrix")
        assert (host[(PAD_ROWS + n) * ostride:] == CANARY_F32).all(), (pli.last_kernel, "wrote past the matrix")
        mat = host[PAD_ROWS * ostride: (PAD_ROWS + n) * ostride].reshape(n, ostride)
        assert (mat[:, cols:] == CANARY_F32).all(), (pli.last_kernel, "wrote into the row padding")
        want, _ = co.score_rows(ref, p, a, b)
        assert np.array_equal(mat[:, :cols], bits(want[:, :cols])), pli.last_kernel


@pytest.mark.parametrize("cols,m,k,length", [(32, 20, 5, 300_007), (32, 13, 5, 100_003), (32, 5, 21, 100_003),
                                             (16, 11, 5, 100_003), (32, 40, 5, 100_003), (1, 8, 5, 10_007)])
@pytest.mark.parametrize("saturate", [False, True])
def test_score_u8_writes_only_its_cells(pli, cols, m, k, length, saturate):
    rng = np.random.default_rng(cols * 131 + m)
    enc, _, ref, seq = make_case(pli, rng, cols, m, k, length)
    dm = lm.DiscreteMatrix(rng.integers(0, 255 // m + 1, (m, k), dtype=np.uint8), 1.0, np.zeros(m, np.float32), 0.0,
                           protein=k == 21)
    ostride = co.stride(cols, 1)
    a, b = 2, ref.rows - 3
    n = b - a
    dev = seq.device
    buf = torch.full(((n + 2 * PAD_ROWS) * ostride + 32,), CANARY_U8, dtype=torch.uint8, device=dev)
    out = buf[PAD_ROWS * ostride: (PAD_ROWS + n) * ostride]
    torch.cuda.synchronize()
    orow, _ = pli.score_u8_dptr(dm, seq.data_ptr(), seq.shape[0], seq.shape[1], cols, m - 1, length, a, b,
                                out.data_ptr(), ostride, saturate=saturate)
    torch.cuda.synchronize()
    assert orow == n
    host = buf.cpu().numpy()
    assert (host[: PAD_ROWS * ostride] == CANARY_U8).all(), pli.last_kernel
    assert (host[(PAD_ROWS + n) * ostride:] == CANARY_U8).all(), pli.last_kernel
    mat = host[PAD_ROWS * ostride: (PAD_ROWS + n) * ostride].reshape(n, ostride)
    assert (mat[:, cols:] == CANARY_U8).all(), (pli.last_kernel, "wrote into the row padding")
    want, _ = co.score_rows_u8(ref, dm.data, a, b)     # the sums stay below 256: both flavours agree
    assert np.array_equal(mat[:, :cols], want[:, :cols]), pli.last_kernel


@pytest.mark.parametrize("cols,length,wrap", [(32, 100_001, 19), (32, 31, 5), (16, 50_003, 40), (1, 1_000, 3),
                                              (33, 10_007, 12), (32, 64, 200)])
def test_stripe_and_wrap_write_only_their_rows(pli, cols, length, wrap):
    rng = np.random.default_rng(length + cols)
    enc = rng.integers(0, 5, length, dtype=np.uint8)
    ref = co.stripe(enc, cols, 5)
    co.configure_wrap(ref, wrap)
    stride_ = co.stride(cols, 1)
    rows_total = ref.data.shape[0]
    dev = torch.device("cuda", 0)
    buf = torch.full(((rows_total + 2 * PAD_ROWS) * stride_,), CANARY_U8, dtype=torch.uint8, device=dev)
    data = buf[PAD_ROWS * stride_: (PAD_ROWS + rows_total) * stride_]
    d_enc = torch.from_numpy(enc).to(dev)
    torch.cuda.synchronize()
    pli.stripe_dptr(d_enc.data_ptr(), length, cols, 4, wrap, data.data_ptr(), stride_)
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    assert (host[: PAD_ROWS * stride_] == CANARY_U8).all()
    assert (host[(PAD_ROWS + rows_total) * stride_:] == CANARY_U8).all()
    assert np.array_equal(host[PAD_ROWS * stride_: (PAD_ROWS + rows_total) * stride_].reshape(rows_total, stride_),
                          ref.data)


# ---- reads: nothing past the last wrap row ------------------------------------------------------------------

_READ_BOUNDS_CHILD = r'''
import ctypes as C, sys
import numpy as np, torch
import lightmotif_amd as lm
from oracle import c_oracle as co
hip = C.CDLL("libamdhip64.so")
torch.cuda.set_device(0)
pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
PAGE = 2 << 20
rng = np.random.default_rng(7)
checked = 0
for k, m, length in ((5, 20, 3_000_017), (5, 12, 2_000_003), (5, 31, 2_000_003), (5, 3, 1_000_003), (21, 12, 2_000_003), (21, 7, 1_000_003)):
    enc = rng.integers(0, k - 1, length, dtype=np.uint8)
    ref = co.stripe(enc, 32, k)
    co.configure_wrap(ref, m - 1)                         # exactly M - 1 wrap rows: the least the reference asks for (avx2.rs:832-837)
    data = np.ascontiguousarray(ref.data[:, :32])
    nbytes = data.size
    alloc = (nbytes + PAGE - 1) // PAGE * PAGE
    base = C.c_void_p()
    assert hip.hipMalloc(C.byref(base), C.c_size_t(alloc)) == 0
    ptr = base.value + alloc - nbytes                      # the matrix ENDS where the allocation ends
    ptr -= ptr % 4                                         # (dword-aligned, as every DenseMatrix row is)
    assert hip.hipMemcpy(C.c_void_p(ptr), data.ctypes.data_as(C.c_void_p), C.c_size_t(nbytes), 1) == 0
    p = np.zeros((m, co.stride(k, 4)), np.float32)
    p[:, :k - 1] = rng.normal(0, 2, (m, k - 1))
    p[:, k - 1] = -np.inf
    pssm = lm.ScoringMatrix(p, protein=k == 21)
    rows = ref.rows
    want, _ = co.score_rows(ref, p)
    t = float(np.sort(want[:, :32].ravel())[-200])
    rc = [tuple(map(int, x)) for x in co.threshold(want, 32, t)]
    args = (pssm, ptr, rows + m - 1, 32, 32, m - 1, length, 0, rows)
    hits, _ = pli.score_threshold_dptr(*args, t)           # pair scan / block scan: symbol blocks requested ahead of use
    assert [tuple(map(int, x)) for x in np.asarray(hits).reshape(-1, 2)] == rc, (k, m, pli.last_kernel)
    assert pli.score_argmax_dptr(*args)[0] == co.argmax(want, 32), (k, m)
    if k == 5:
        batch = pli.scan_threshold_batch([pssm, pssm.reverse_complement(), pssm], [t, t, t],
                                         pli.adopt_sequence(ptr, rows, m - 1, 32, 32, length))   # several motifs per pass
        assert [tuple(map(int, x)) for x in np.asarray(batch[0][0]).reshape(-1, 2)] == rc
        dm = pssm.to_discrete()
        out = torch.empty((rows, 32), dtype=torch.uint8, device="cuda")
        pli.score_u8_dptr(dm, ptr, rows + m - 1, 32, 32, m - 1, length, 0, rows, out.data_ptr(), 32)   # the u8 store on the pair pipeline
        torch.cuda.synchronize()
        w8 = co.aligned_empty((m, 32), np.uint8)
        w8[:] = dm.data
        assert np.array_equal(out.cpu().numpy(), co.avx2_score_rows_u8(ref, w8)[:, :32]), (k, m, pli.last_kernel)
    torch.cuda.synchronize()
    checked += 1
    hip.hipFree(base)
print("READ_BOUNDS_OK", checked)
'''


def test_scans_read_nothing_past_the_last_wrap_row():
    """The scans request symbol blocks ahead of their use; a request of the group before a stream's last one used to reach
    128 bytes past the stream's input (round-5 ADVICE: results unaffected, but a matrix ending on an unmapped page boundary
    could fault).  A matrix with exactly M - 1 wrap rows (avx2.rs:832-837) is placed so that it ends where a raw hipMalloc
    allocation of whole 2 MB pages ends, and every scan family runs over it -- in a child process, so that a memory fault
    shows as a failed test and not as a dead test run."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, PYTHONPATH=str(root) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", _READ_BOUNDS_CHILD], capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert r.returncode == 0 and "READ_BOUNDS_OK 6" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
