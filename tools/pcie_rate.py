import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import ctypes as C
import lightmotif_amd as lm
from lightmotif_amd import _ffi
L = _ffi.lib()
length, m = 200_000_000, 20
rows = -(-length // 32)
rng = np.random.default_rng(0)
seq = rng.integers(0, 4, (rows + m - 1, 32), dtype=np.uint8)
seq[rows:, :31] = seq[:m - 1, 1:]; seq[rows:, 31] = 4
sites = ["".join("ACTG"[i] for i in rng.integers(0, 4, m)) for _ in range(10)]
pssm = lm.create(sites).counts.normalize(0.1).log_odds()
out = np.empty((rows, 32), np.float32)
orow, mi = C.c_size_t(0), C.c_size_t(0)
def call():
    st = L.lm_hip_score_f32(seq.ctypes.data, rows + m - 1, 32, 32, m - 1, length, pssm.data.ctypes.data, m,
                            pssm.data.shape[1], 5, 0, rows, out.ctypes.data, 32, C.byref(orow), C.byref(mi))
    assert st == 0, L.lm_hip_last_error()
call()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
t = min(ts)
print(f"host-pointer lm_hip_score_f32, {length/1e6:.0f} Mbp: {t*1e3:.1f} ms = {length/t/1e9:.2f} Gpos/s "
      f"({5*length/t/1e9:.1f} GB/s over PCIe incl. staging)")
