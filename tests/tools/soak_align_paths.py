#!/usr/bin/env python3
"""One-off soak (not part of the test suite): ragged batches x random scoring sets, the packed trace fill (K2p) against
the int32 trace fill (K2): scores and both gapped rows of every pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tsta_amd
from tests import util

eng = tsta_amd.Engine(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
t0 = time.time(); bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    if len(sys.argv) > 3 and sys.argv[3] == 'wide':      # scoring sets that push some paths out of the 16-bit domain
        M = int(rng.integers(0, 41)); X = int(rng.integers(-60, min(M, 1))); E = int(rng.integers(-20, 1)); O = int(rng.integers(-40, 1))
    else:
        M = int(rng.integers(0, 7)); X = int(rng.integers(-9, min(M, 1))); E = int(rng.integers(-5, 1)); O = int(rng.integers(-12, 1))
    prm = tsta_amd.Params(M, X, E, O)
    pairs = []
    for k in range(30):
        lo, hi = (1, 400) if os.environ.get("SOAK_SMALL") else (1500, 9000)
        n = int(rng.integers(lo, hi)); m = int(rng.integers(lo, hi))
        a = util.random_dna(int(rng.integers(1 << 30)), n)
        kind = k % 6
        if kind == 0: b = util.mutate(a, int(rng.integers(1 << 30)), 0.02 + 0.2 * rng.random())[:m]
        elif kind == 1: b = util.random_dna(int(rng.integers(1 << 30)), m)
        elif kind == 2: b = util.low_complexity(int(rng.integers(1 << 30)), m, int(rng.integers(1, 4)))
        elif kind == 3: b = np.concatenate([a[: n // 3], a[2 * n // 3:]])
        elif kind == 4: b = np.concatenate([util.random_dna(5, m // 2), a[n // 2:]])[:m]
        else: b = a[::-1].copy()[:m]
        if len(b) == 0: b = a[:1]
        pairs.append((a, b) if k % 2 else (b, a))
    eng.set_option("trace_rows", 4); eng.set_option("trace_packed", 0)
    ref = eng.psa_align_batch(pairs, prm)
    for elide, perm, tile in ((1, 1, 0), (0, 0, 64), (1, 0, 256)):
        eng.set_option("trace_packed", 1); eng.set_option("psa_elide", elide); eng.set_option("psa_perm", perm); eng.set_option("trace_tile", tile)
        got = eng.psa_align_batch(pairs, prm)
        if got != ref:
            bad += 1
            print("MISMATCH", trial, (M, X, E, O), (elide, perm, tile), [i for i in range(len(pairs)) if got[i] != ref[i]], flush=True)
    eng.set_option("trace_tile", 0)
    print("trial", trial, (M, X, E, O), "%.0f s" % (time.time() - t0), flush=True)
print("mismatches:", bad)
