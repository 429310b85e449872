"""Every prefilter route (one symbol / pairs) against the exact fused kernel, protein and DNA, M = 4 ... 36, two p-values:
prints MISMATCH lines (round 5 found the protein one-symbol kernels of M = 7, 8 this way).  python tools/prefilter_route_scan.py [length]"""
import sys
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import lightmotif_amd as lm
COLS=32
torch.cuda.set_device(0); dev=torch.device("cuda",0)
length=int(sys.argv[1]) if len(sys.argv)>1 else 50_000_000; rows=-(-length//COLS); mmax=36
def mk(opts):
    p=lm.Pipeline.hip(0, stream=torch.cuda.current_stream().cuda_stream)
    for k,v in opts.items(): p.set_option(k,v)
    return p
for K,protein in ((21,True),(5,False)):
    gen=torch.Generator(device=dev); gen.manual_seed(55)
    seq=torch.empty((rows+mmax-1,COLS),dtype=torch.uint8,device=dev)
    seq[:rows]=torch.randint(0,K-1,(rows,COLS),dtype=torch.uint8,device=dev,generator=gen)
    plis={"single":mk({"pair_prefilter":0,"pair_prefilter_protein":0}),"pair":mk({"pair_prefilter_protein":1}),"exact":mk({"prefilter":0})}
    plis["exact"].configure_wrap_dptr(seq.data_ptr(), rows, COLS, COLS, mmax-1, K-1)
    sym=lm.lib.PROTEIN_SYMBOLS[:-1] if protein else "ACTG"
    for m in list(range(1,37)):
        prng=np.random.default_rng(m)
        sites=["".join(sym[i] for i in prng.integers(0,len(sym),m)) for _ in range(6)]
        pssm=lm.create(sites, protein=protein).counts.normalize(0.1).log_odds()
        for pv in (1e-4,1e-6):
            thr=pssm.score_for_pvalue(pv)
            r={}
            if pv == 1e-4:   # the materialised route (store kernel + Threshold on the stored matrix) as a fourth, independent answer
                if "out" not in globals():
                    globals()["out"] = torch.empty((rows, COLS), dtype=torch.float32, device=dev)
                st = plis["exact"]
                st.score_dptr(pssm, seq.data_ptr(), rows+mmax-1, COLS, COLS, mmax-1, length, 0, rows, out.data_ptr(), COLS)
                r["stored"]=(st.threshold_dptr(out.data_ptr(), rows, COLS, COLS, thr), st.last_kernel)
            for name,p in plis.items():
                h=p.score_threshold_dptr(pssm, seq.data_ptr(), rows+mmax-1, COLS, COLS, mmax-1, length, 0, rows, thr)
                r[name]=(h[0], p.last_kernel)
            ok_s=np.array_equal(r["single"][0], r["exact"][0]); ok_p=np.array_equal(r["pair"][0], r["exact"][0])
            if "stored" in r and not np.array_equal(np.asarray(r["stored"][0]), np.asarray(r["exact"][0])):
                ok_s = False
            if not (ok_s and ok_p):
                print("MISMATCH K",K,"m",m,"p",pv,"thr",round(thr,3),{k:(len(v[0]),v[1]) for k,v in r.items()})
    print("done K",K)
