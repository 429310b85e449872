"""Fused threshold calls of one process, one after the other: wall time and the scan kernel's own duration (context option
"time_scan") per call -- shows the warm-up of a fresh process (the first ~60 calls run their scan kernel at 0.25 ms, later
ones at 0.22) and what a call costs beside its scan.  python tools/scan_kernel_timer_check.py"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import numpy as np, torch
import lightmotif_amd as lm
from bench_configs import motif, resident_sequence
torch.cuda.set_device(0)
pli = lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
m = 20; length = 1_000_000_000
seq, rows = resident_sequence(pli, length, 5, m - 1, 11)
pssm = motif(np.random.default_rng(m), m); thr = pssm.score_for_pvalue(1e-5)
call = lambda: pli.score_threshold_dptr(pssm, seq.data_ptr(), rows + m - 1, 32, 32, m - 1, length, 0, rows, thr)
pli.set_option("time_scan", 1)
w = []; k = []
for i in range(160):
    if i == 120:
        time.sleep(0.5)   # a pause: does the warm-up start over?
    t0 = time.perf_counter(); call(); w.append((time.perf_counter() - t0) * 1e3); k.append(pli.last_scan_kernel_ms)
for lo in range(0, 160, 10):
    print(f"calls {lo:3d}-{lo + 9:3d}: wall ms {np.median(w[lo:lo + 10]):.4f}  scan kernel ms {np.median(k[lo:lo + 10]):.4f}")
