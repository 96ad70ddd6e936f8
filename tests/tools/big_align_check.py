import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import tsta_amd
from tests import util
eng = tsta_amd.Engine(0)
rng = np.random.default_rng(5)
pairs = []
for i in range(4000):
    n = int(rng.integers(300, 1500))
    a = util.random_dna(10_000 + i, n)
    pairs.append((a, util.mutate(a, 20_000 + i, 0.08)))
eng.psa_align_batch(pairs[:10])
t = time.perf_counter(); res = eng.psa_align_batch(pairs); dt = time.perf_counter() - t
k, nl = eng.last_kernel_ms()
cells = sum(len(a) * len(b) for a, b in pairs)
print("4000 short pairs: wall %.1f ms kernels %.1f ms launches %d  %.1f GCUPS" % (dt * 1e3, k, nl, cells / dt / 1e9))
for i in range(0, 4000, 199):
    o = util.oracle_align(*pairs[i])
    assert res[i] == (o[0], o[2], o[3]), i
sc = eng.psa_score_batch(pairs)
assert all(int(sc[i]) == res[i][0] for i in range(4000))
print("ok")
