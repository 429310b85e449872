import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import lightmotif_amd as lm
cols = 16; length = 500_000_000; m = 20
pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
rows = -(-length // cols)
seq = torch.randint(0, 4, (rows + m - 1, 32), dtype=torch.uint8, device="cuda")
rng = np.random.default_rng(m)
pssm = lm.create(["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]).counts.normalize(0.1).log_odds()
args = (pssm, seq.data_ptr(), rows + m - 1, 32, cols, m - 1, length, 0, rows)
for name, fn in (("fused argmax", lambda: pli.score_argmax_dptr(*args)), ("fused threshold t=12", lambda: pli.score_threshold_dptr(*args, 12.0))):
    for _ in range(3): fn()
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"C=16 M={m} {name}: {np.median(ts):.3f} ms per {length/1e6:.0f} Mbp = {length/np.median(ts)/1e6:.0f} Gpos/s  ({pli.last_kernel})")
