"""One-off (round 4): the single-pair trace fill in its split form (solver bands + emitter tiles, forced at any size; the emitters keep
the running maximum) against the form without it and against the oracle: maxsorce and both rows, 40 pairs of odd sizes (partial last
bands, the maximum far from the end), pure-ACGT and generic step.  usage: split_form_fuzz.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, tsta_amd
from tests import util
eng = tsta_amd.Engine(0)
rng = np.random.default_rng(77)
bad = 0
for t in range(40):
    n = int(rng.integers(300, 6000)); m = int(rng.integers(300, 6000))
    a = util.random_dna(int(rng.integers(1 << 30)), n)
    kind = t % 4
    if kind == 0: b = util.mutate(a, int(rng.integers(1 << 30)), 0.1)[:m]
    elif kind == 1: b = util.random_dna(int(rng.integers(1 << 30)), m)
    elif kind == 2: b = np.concatenate([util.random_dna(5, m // 2), a[: m // 2]])     # best score far from the end
    else: b = np.concatenate([a[n // 2:], util.random_dna(6, m // 3)])
    sc = [(2, -5, -2, -4), (1, -1, -1, 0), (3, -4, -3, -4), (10, -10, -3, -12)][int(rng.integers(0, 4))]
    p = tsta_amd.Params(M=sc[0], X=sc[1], E=sc[2], O=sc[3])
    for perm in (1, 0):
        eng.set_option("psa_perm", perm)
        eng.set_option("trace_split", 0); w = eng.psa_align(a, b, p)
        eng.set_option("trace_split", 2); g = eng.psa_align(a, b, p)
        form = eng.stat("last_trace_form")
        o = util.oracle_align(a, b, *[sc[0], sc[1], sc[2], sc[3]])
        if g != w or (g[0], g[1], g[2]) != (o[0], o[2], o[3]):
            bad += 1; print("DIFF", t, n, m, sc, perm, g[0], w[0], o[0], form, flush=True)
eng.set_option("trace_split", 1); eng.set_option("psa_perm", 1)
print("split-form fuzz: 40 pairs x 2 alphabets paths, mismatches:", bad)
