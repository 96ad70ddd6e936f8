#!/usr/bin/env python3
"""Generate tests/golden/msa_n_golden.json: what the UNMODIFIED reference TSTA_msa (oracle/_ref/TSTA_msa) writes for
multi-FASTA inputs that contain literal 'N' characters, at -W 10 (its default), 5, 30, 3 and 16.

Why several -W: the reference forces the column delta to 0 in 'N' columns in its SECOND pass only (c-t-simd.c:461,481);
its speculative first pass (:357-404) does not, so the carries it hands from one SIMD lane segment to the next are
computed without the rule and the output depends on the blocking (-W, SIMD width).  A case whose five outputs agree is
recorded as `w_invariant`: there the reference's behaviour IS a function of the input, and the engine (which applies the
rule consistently) must reproduce it byte for byte.  The other cases are recorded with all their digests as evidence.
Data only; inputs are regenerated from the spec by n_case_seqs()."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util  # noqa: E402
from tests.golden.make_poa_golden import family, write_multi_fasta  # noqa: E402

WS = (10, 5, 30, 3, 16)


def n_case_specs():
    out = []
    for seed in range(1, 41):
        out.append(dict(seed=seed, n_seq=2 + seed % 5, length=[150, 333, 640, 700, 1300][seed % 5], rate=[0.02, 0.1, 0.2][seed % 3],
                        n_count=[1, 1, 2, 3, 5, 10][seed % 6], first_too=bool(seed % 2)))
    return out


def n_case_seqs(c):
    seqs = [np.array(s) for s in family(c["seed"] * 17, c["n_seq"], c["length"], c["rate"], jitter=0)]
    rng = np.random.default_rng(c["seed"])
    for _ in range(c["n_count"]):
        k = int(rng.integers(0 if c["first_too"] else 1, c["n_seq"]))
        seqs[k][int(rng.integers(0, len(seqs[k])))] = ord("N")
    return seqs


def run_ref(fasta, w, out):
    for _ in range(3):                               # the reference's thread pool can deadlock: time limit + retry
        try:
            r = subprocess.run([os.path.join(util.REF_DIR, "TSTA_msa"), "-T", "4", "-W", str(w), "-i", fasta, "-o", out],
                               capture_output=True, timeout=120)
            if r.returncode == 0:
                data = open(out, "rb").read()
                return hashlib.md5(data).hexdigest(), len(data)
        except subprocess.TimeoutExpired:
            pass
    raise RuntimeError("reference failed on %s -W %d" % (fasta, w))


def main():
    golden = []
    with tempfile.TemporaryDirectory() as td:
        fa, out = os.path.join(td, "in.fa"), os.path.join(td, "out.txt")
        for c in n_case_specs():
            write_multi_fasta(fa, n_case_seqs(c))
            res = [run_ref(fa, w, out) for w in WS]
            g = dict(c)
            g.update(md5=res[0][0], size=res[0][1], md5_by_w={str(w): r[0] for w, r in zip(WS, res)},
                     w_invariant=len({r[0] for r in res}) == 1)
            golden.append(g)
            print(c["seed"], "invariant" if g["w_invariant"] else "W-dependent: %d distinct" % len({r[0] for r in res}), flush=True)
    with open(os.path.join(util.GOLDEN_DIR, "msa_n_golden.json"), "w") as f:
        json.dump(golden, f, indent=0)
    print("wrote", len(golden), "cases,", sum(g["w_invariant"] for g in golden), "W-invariant")


if __name__ == "__main__":
    main()
