import os, sys, time
os.environ["LM_HIP_TRACE"]="1"
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np, torch
import lightmotif_amd as lm
from bench_configs import motif, resident_sequence
torch.cuda.set_device(0)
pli=lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
m=20; length=1_000_000_000
seq, rows = resident_sequence(pli, length, 5, m-1, 11)
pssm=motif(np.random.default_rng(m), m); thr=pssm.score_for_pvalue(1e-5)
for i in range(6):
    t0=time.perf_counter()
    pli.score_threshold_dptr(pssm, seq.data_ptr(), rows+m-1, 32, 32, m-1, length, 0, rows, thr)
    print("python call ms", (time.perf_counter()-t0)*1e3, file=sys.stderr)
