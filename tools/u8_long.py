import sys, numpy as np, torch
sys.path.insert(0, ".")
import lightmotif_amd as lm
length = 1_000_000_000
pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
rows = -(-length // 32)
for m in (20, 40, 100):
    seq = torch.randint(0, 4, (rows + m - 1, 32), dtype=torch.uint8, device="cuda")
    out = torch.empty((rows, 32), dtype=torch.uint8, device="cuda")
    rng = np.random.default_rng(m)
    dm = lm.DiscreteMatrix(rng.integers(0, 255 // m + 1, (m, 5), dtype=np.uint8), 1.0, np.zeros(m, np.float32), 0.0)
    args = (dm, seq.data_ptr(), rows + m - 1, 32, 32, m - 1, length, 0, rows, out.data_ptr(), 32)
    for _ in range(5): pli.score_u8_dptr(*args)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); pli.score_u8_dptr(*args); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)[5]
    print(f"u8 scores M={m}: {t:.3f} ms per Gbp = {length/t/1e6:.0f} Gpos/s ({pli.last_kernel})")
    del seq, out
