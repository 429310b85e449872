import sys, time, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import lightmotif_amd as lm
sys.path.insert(0, str(ROOT / "tools"))
import bench_configs as bc
torch.cuda.set_device(0)
pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
length, m = 1_000_000_000, 20
seq, rows = bc.resident_sequence(pli, length, 5, m - 1, 11)
dm = bc.motif(np.random.default_rng(2), m).to_discrete()
out = torch.empty((rows, 32), dtype=torch.uint8, device=seq.device)
for T in (0, 64, 128, 256, 512, 1024, 4096):
    pli.set_rows_per_stream(T)
    def score():
        pli.score_u8_dptr(dm, seq.data_ptr(), rows + m - 1, 32, 32, m - 1, length, 0, rows, out.data_ptr(), 32)
    for _ in range(3): score()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); score(); b.record()
    torch.cuda.synchronize()
    print(T, pli.last_kernel, "event ms", round(float(np.median([a.elapsed_time(b) for a, b in ev])), 4), "wall", round(bc.timeit(score, 20) * 1e3, 4))
