import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, tsta_amd
from tests import util
e = tsta_amd.Engine(0)
for N in (1100, 2100, 9000):
    pairs = []
    for i in range(N):
        l = 20 + (i * 7) % 90
        a = util.random_dna(10_000 + i, l)
        pairs.append((a, util.mutate(a, 20_000 + i, 0.1)))
    got = e.psa_score_batch(pairs); want = util.oracle_maxscore_batch(pairs)
    bad = np.nonzero(got != want)[0]
    print(N, "mismatches", len(bad), [(int(k), len(pairs[k][0]), len(pairs[k][1]), int(got[k]), int(want[k])) for k in bad[:12]], flush=True)
