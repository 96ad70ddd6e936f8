#!/usr/bin/env python3
"""Generate tests/golden/msa_domain_golden.json: for which SCORINGS is TSTA_msa's output the plain partial-order DP?

The reference's POA fill keeps row differences in saturating int8 with a cutoff at 60 and speculative lane starts
(c-t-simd.c:196-261,357-404,438-447): beyond a scoring domain its output is a property of that arithmetic and of -W.  The
engine's routing rule (poa_fill.hip: poa_scoring_needs_reference_arithmetic) sends such scorings through the kernel that works in
the reference's own arithmetic (K4e); inside the rule's domain the fast plain-DP kernels serve the call.  The rule is empirical:
    fast path  iff  M >= 1, X < M, E <= -1, O <= 0, all four in int8, and
                    W max(|E|, M) + |O| + |X| < 64                                (W = -W of the reference build, L = 64 W)
(over 2 168 scanned (scoring, -W) points — plain-DP restatement against the int8 restatement, which equalled the compiled
reference on every point where both were run — the smallest value of the left-hand side with a difference was 71)
This script runs the UNMODIFIED reference (oracle/_ref/TSTA_msa) at the case's -W on seeded sequence families — half of them
unpadded (lengths = multiples of 64 W), half ragged — for scorings on both sides of the rule, and records the md5 of its output.
Data only.  The tests (tests/test_msa_host.py::test_msa_domain_*) check on the CPU that the plain-DP restatement equals the
reference wherever the rule says "fast path" and that the int8 restatement equals it everywhere; on the GPU that the TSTA_msa
drop-in does.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util  # noqa: E402
from tests.golden.make_poa_golden import family, write_multi_fasta  # noqa: E402


def fast_path(M, X, E, O, W):
    """Python mirror of poa_fill.hip: !poa_scoring_needs_reference_arithmetic (used to place grid points; the GPU test asks the engine)."""
    if not (M >= 1 and X < M and E <= -1 and O <= 0 and all(-128 <= v <= 127 for v in (M, X, E, O))):
        return False
    return W * max(-E, M) - O + abs(X) < 64


def domain_case_seqs(c):
    L = 64 * c["W"]
    length = L if not c["ragged"] else L + L // 3
    return [np.array(s) for s in family(c["seed"], c["n_seq"], length, c["rate"], jitter=1 if c["ragged"] else 0)]


def specs():
    rng = np.random.RandomState(4242)
    out = []
    Ws = [3, 5, 10, 16, 24, 32]
    Ms = [1, 1, 2, 2, 3, 4, 5, 6, 8, 10, 15]
    Xs = [0, -1, -2, -3, -4, -5, -6, -8, -10, -15, -20, -30, -45, -60, -100, -128]
    Es = [-1, -1, -2, -2, -3, -4, -5, -6, -8, -10]
    Os = [0, 0, -1, -2, -4, -6, -8, -10, -12, -16, -20, -30, -40, -50]
    n_in = n_out = 0
    while n_in < 90 or n_out < 70:
        W = int(rng.choice(Ws)); M = int(rng.choice(Ms)); X = int(rng.choice(Xs)); E = int(rng.choice(Es)); O = int(rng.choice(Os))
        if X >= M:
            continue
        inside = fast_path(M, X, E, O, W)
        q = W * max(-E, M) - O + abs(X)
        if inside and (n_in >= 90 or (q < 44 and rng.rand() < 0.7)):          # inside: mostly near the edge
            continue
        if not inside and (n_out >= 70 or (q > 150 and rng.rand() < 0.8)):    # outside: mostly just beyond it
            continue
        k = len(out)
        out.append(dict(W=W, scoring=dict(M=M, X=X, E=E, O=O), seed=300 + 7 * k, n_seq=3 + k % 3, rate=[0.06, 0.12, 0.2][k % 3],
                        ragged=bool(k % 2)))
        n_in += inside; n_out += not inside
    # fixed points: the documented parameter sets at the default -W, scorings of the other fixtures, sign / tie oddities the
    # reference survives (it crashes or hangs on some: E >= 0 with M = 0, positive gap scores)
    for W, (M, X, E, O) in ((10, (2, -5, -2, -4)), (10, (2, -3, -2, -4)), (10, (1, -1, -1, 0)), (10, (5, -4, -1, -10)), (10, (10, -10, -3, -12)),
                            (10, (3, -60, -3, -29)), (16, (2, -30, -2, -27)), (32, (1, -30, -1, -27)), (3, (2, -15, -1, -50)),
                            (3, (4, -128, -3, -40)), (10, (30, -30, -2, -4)), (10, (2, -5, -29, -4)), (10, (2, -5, -2, -62)),
                            (10, (3, 3, -2, -4)), (10, (2, 3, -2, -4)), (10, (0, -3, -2, -4)), (10, (2, -5, 0, -4)), (10, (0, 0, -1, 0))):
        k = len(out)
        out.append(dict(W=W, scoring=dict(M=M, X=X, E=E, O=O), seed=300 + 7 * k, n_seq=4, rate=0.1, ragged=bool(k % 2)))
    return out


def run_ref(fa, sc, W, out):
    cmd = [os.path.join(util.REF_DIR, "TSTA_msa"), "-M", str(sc["M"]), "-X", str(sc["X"]), "-E", str(sc["E"]), "-O", str(sc["O"]),
           "-T", "2", "-W", str(W), "-i", fa, "-o", out]
    for _ in range(4):                       # the reference's thread pool can hang (pthreadpool.c:96-117): time limit + retry
        if os.path.exists(out):
            os.remove(out)
        try:
            r = subprocess.run(cmd, capture_output=True, timeout=120)
        except subprocess.TimeoutExpired:
            continue
        if r.returncode != 0 or not os.path.exists(out):
            return None
        data = open(out, "rb").read()
        return hashlib.md5(data).hexdigest(), len(data)
    return None


def one(c):
    with tempfile.TemporaryDirectory() as td:
        fa, out = os.path.join(td, "in.fa"), os.path.join(td, "o.txt")
        write_multi_fasta(fa, domain_case_seqs(c))
        r = run_ref(fa, c["scoring"], c["W"], out)
    if r is None:
        return None
    g = dict(c); g.update(md5=r[0], size=r[1])
    return g


def main():
    with ThreadPoolExecutor(4) as ex:
        res = [g for g in ex.map(one, specs()) if g is not None]
    n_in = sum(fast_path(W=g["W"], **g["scoring"]) for g in res)
    with open(os.path.join(util.GOLDEN_DIR, "msa_domain_golden.json"), "w") as f:
        json.dump(res, f, indent=0)
    print("wrote", len(res), "cases,", n_in, "on the fast path by the rule")


if __name__ == "__main__":
    main()
