import sys
sys.path.insert(0, "/root/repo")
import numpy as np
import tsta_amd
from tests import util
eng = tsta_amd.Engine(0)
rng = np.random.default_rng(12)
M = int(rng.integers(0, 7)); X = int(rng.integers(-9, min(M, 1))); E = int(rng.integers(-5, 1)); O = int(rng.integers(-12, 1))
print(M, X, E, O)
prm = tsta_amd.Params(M, X, E, O)
pairs = []
for k in range(24):
    n = int(rng.integers(6000, 26000)); m = int(rng.integers(6000, 26000))
    a = util.random_dna(int(rng.integers(1 << 30)), n)
    kind = k % 6
    if kind == 0: b = util.mutate(a, int(rng.integers(1 << 30)), 0.02 + 0.2 * rng.random())[:m]
    elif kind == 1: b = util.random_dna(int(rng.integers(1 << 30)), m)
    elif kind == 2: b = util.low_complexity(int(rng.integers(1 << 30)), m, int(rng.integers(1, 4)))
    elif kind == 3: b = np.concatenate([a[: n // 3], a[2 * n // 3:]])
    elif kind == 4: b = np.concatenate([util.random_dna(5, m // 2), a[n // 2:]])[:m]
    else: b = a[::-1].copy()[:m]
    if len(b) == 0: b = a[:1]
    pairs.append((a, b))
print([(len(a), len(b)) for a, b in pairs])
eng.set_option("psa_packed", 0); eng.set_option("psa_bands", 0)
ref = eng.psa_score_batch(pairs, prm)
eng.set_option("psa_packed", -1); eng.set_option("psa_bands", 16)
got = eng.psa_score_batch(pairs, prm)
print("ref", ref[8], "got", got[8], "oracle", util.oracle_maxscore(pairs[8][0], pairs[8][1], M=M, X=X, E=E, O=O))

for sub in ([8, 18], [18, 8]):
    pp = [pairs[i] for i in sub]
    eng.set_option("psa_packed", 0); eng.set_option("psa_bands", 0)
    r = eng.psa_score_batch(pp, prm)
    for bands in (16, 8):
        for elide, perm in ((1, 1), (0, 1), (1, 0), (0, 0)):
            for tile in (64, 128):
                eng.set_option("psa_packed", -1); eng.set_option("psa_bands", bands); eng.set_option("trace_tile", tile)
                eng.set_option("psa_elide", elide); eng.set_option("psa_perm", perm)
                g = eng.psa_score_batch(pp, prm)
                print(sub, "bands", bands, "elide", elide, "perm", perm, "tile", tile, r.tolist(), g.tolist(), "OK" if np.array_equal(r, g) else "DIFF")
    for packed in (32, 16, 8):
        eng.set_option("psa_packed", packed); eng.set_option("psa_bands", -1); eng.set_option("psa_elide", 1); eng.set_option("psa_perm", 1)
        g = eng.psa_score_batch(pp, prm)
        print(sub, "packed", packed, r.tolist(), g.tolist(), "OK" if np.array_equal(r, g) else "DIFF")
