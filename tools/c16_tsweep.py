import sys, numpy as np, torch
sys.path.insert(0, ".")
import lightmotif_amd as lm
cols = 16; length = 500_000_000
pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
rows = -(-length // cols)
for m in (8, 20, 32):
    seq = torch.randint(0, 4, (rows + m - 1, 32), dtype=torch.uint8, device="cuda")
    out = torch.empty((rows, cols), dtype=torch.float32, device="cuda")
    rng = np.random.default_rng(m)
    pssm = lm.create(["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]).counts.normalize(0.1).log_odds()
    args = (pssm, seq.data_ptr(), rows + m - 1, 32, cols, m - 1, length, 0, rows, out.data_ptr(), cols)
    res = []
    for t in (0, 16, 32, 64, 128, 256, 512):
        pli.set_rows_per_stream(t)
        for _ in range(30): pli.score_dptr(*args)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record(); pli.score_dptr(*args); b.record()
        torch.cuda.synchronize()
        res.append("%s:%.4f" % (t or "default", sorted(a.elapsed_time(b) for a, b in ev)[15]))
    print(f"C=16 M={m} {pli.last_kernel} ms per 500 Mbp by rows_per_stream:", "  ".join(res))
    del seq, out
