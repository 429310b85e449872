import sys, time, ctypes as C, numpy as np
sys.path.insert(0, ".")
import lightmotif_amd as lm
from lightmotif_amd import _ffi
pli = lm.Pipeline.hip(0); L = pli._L
rng = np.random.default_rng(0xEC011)
enc = rng.integers(0, 4, 464_165, dtype=np.uint8)
pssm = lm.create(["GTTGACCTTATCAAC", "GTTGATCCAGTCAAC"]).counts.normalize(0.1).log_odds()
seq = pli.stripe(lm.EncodedSequence(enc), 32); seq.configure(pssm)
scores = lm.StripedScores.empty(pli, 32)
hp, hs, hq, hc = pssm._device(pli), seq._h, scores._h, pli._h
found, best, val = C.c_int(0), _ffi.Coords(), C.c_float(0)
for _ in range(200):
    L.lm_hip_score_into(hc, hp, hs, hq); L.lm_hip_argmax(hc, hq, C.byref(found), C.byref(best), C.byref(val))
tc = ta = 0.0
N = 3000
pc = time.perf_counter
for _ in range(N):
    t0 = pc(); L.lm_hip_score_into(hc, hp, hs, hq); t1 = pc()
    L.lm_hip_argmax(hc, hq, C.byref(found), C.byref(best), C.byref(val)); t2 = pc()
    tc += t1 - t0; ta += t2 - t1
print(f"score_into call {tc/N*1e6:.2f} us, argmax (poll) {ta/N*1e6:.2f} us, total {(tc+ta)/N*1e6:.2f}")
# empty ctypes call cost
t0 = pc()
for _ in range(N): L.lm_hip_abi_version()
print(f"ctypes trivial call {(pc()-t0)/N*1e6:.2f} us")
