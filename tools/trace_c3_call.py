import os, sys, time
os.environ["LM_HIP_TRACE"]="1"
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np, torch, bench
import lightmotif_amd as lm
torch.cuda.set_device(0); dev=torch.device("cuda",0)
pli=lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
c3=bench.c3_setup(pli, dev, 1, 0, 100_000_000)
for flag in (1,1,1,0,0,0,1,0):
    pli.set_option("sort_hits", flag)
    t0=time.perf_counter(); r=pli.scan_threshold_batch(c3["pssms"], c3["ts"], c3["seq"]); torch.cuda.synchronize()
    print("sort_hits", flag, "python call ms", round((time.perf_counter()-t0)*1e3,2), file=sys.stderr)
