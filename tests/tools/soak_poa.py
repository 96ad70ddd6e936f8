#!/usr/bin/env python3
"""One-off soak (not part of the test suite): random DAGs x random scoring sets, tsta_poa_fill / tsta_poa_align against
the oracle fill and the Python restatement of the walk."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tsta_amd
from tests import util
from tests.test_poa_fill_gpu import random_dag

eng = tsta_amd.Engine(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0; t0 = time.time()
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    M = int(rng.integers(0, 7)); X = int(rng.integers(-9, min(M, 1))); E = int(rng.integers(-5, 1)); O = int(rng.integers(-12, 1))
    V = int(rng.integers(1, 1800)); mp = int(rng.integers(1, 9)); n = int(rng.integers(1, 1500))
    pred_off, pred, base, is_sink, seq = random_dag(int(rng.integers(1 << 30)), V, mp, n)
    prm = tsta_amd.Params(M, X, E, O)
    if trial % 3 == 0:                                     # literal N in the sequence (K4's segment-by-segment path)
        seq = seq.copy(); seq[rng.integers(0, len(seq), size=max(1, len(seq) // 40))] = ord("N")
    # the plain kernels against the plain DP: routing to the reference's arithmetic off (random DAGs spread their predecessor rows far
    # beyond what it holds exactly, some scorings lie outside its rule: round 4 sends those through K4e, checked in the second leg)
    eng.set_option("poa_ref_arith", 0)
    for cols, dataflow in ((2, 1), (2, 0), (4, 0)):        # K4d (solver + emitters), K4 with 2 / 4 columns per lane
        eng.set_option("poa_cols", cols); eng.set_option("poa_dataflow", dataflow)
        got = eng.poa_fill(pred_off, pred, base, is_sink, seq, prm)
        want = util.oracle_poa_fill(pred_off, pred, base, is_sink, seq, M=M, X=X, E=E, O=O)
        ok = got[0] == want[0] and np.array_equal(got[1], want[1]) and all(np.array_equal(g, w) for g, w in zip(got[2:], want[2:]))
        steps, end_col, last = eng.poa_align(pred_off, pred, base, is_sink, seq, prm)
        wsteps, wend = util.poa_walk(pred_off, pred, is_sink, want[1], want[2], want[3], want[4])
        ok = ok and end_col == wend and [(int(s["node"]), int(s["col"]), int(s["code"]), int(s["peek"])) for s in steps] == wsteps
        if not ok:
            bad += 1
            print("MISMATCH", trial, (M, X, E, O), (V, mp, n), (cols, dataflow), flush=True)
    eng.set_option("poa_cols", 0); eng.set_option("poa_dataflow", 1)
    # K4e (every fill in the reference's own arithmetic) against the int8 restatement, at -W 10 and another -W
    if M >= 1 and X < M and E <= -1 and int(np.diff(pred_off).max()) <= 8:
        eng.set_option("poa_ref_arith", 2)
        for w in (10, int(rng.choice([3, 5, 16]))):
            eng.set_option("poa_ref_block", 64 * w)
            got = eng.poa_fill(pred_off, pred, base, is_sink, seq, prm)
            if eng.last_status != 0:
                continue                                    # (the path could not run: LDS)
            want = util.oracle_poa_fill_i8(pred_off, pred, base, is_sink, seq, w, M=M, X=X, E=E, O=O)
            if not (got[0] == want[0] and all(np.array_equal(g, x) for g, x in zip(got[1:], want[1:]))):
                bad += 1
                print("MISMATCH (reference arithmetic)", trial, (M, X, E, O), (V, mp, n), "W", w, flush=True)
        eng.set_option("poa_ref_block", 640)
    eng.set_option("poa_ref_arith", 1)
print("mismatches:", bad, "%.0f s" % (time.time() - t0))
