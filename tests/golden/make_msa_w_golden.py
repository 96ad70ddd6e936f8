#!/usr/bin/env python3
"""Generate tests/golden/msa_w_golden.json: what the UNMODIFIED reference TSTA_msa (oracle/_ref/TSTA_msa) writes for 'N'-FREE
inputs at values of -W / shapes where its output is a property of its int8 arithmetic rather than of the alignment problem:
  * -W 30..32: the 'N' padding of the last block is W columns per byte lane; the first pass (c-t-simd.c:357-404) scores them as
    mismatches, the second (:461) as 0, and from about W |E| + |O| >= 60 that shift exceeds the cutoff of the offset trackers;
  * a first sequence much shorter than the next: the sequence then spans more blocks of L = 64 W than the graph, the case
    the first-column branch (:194) does not guard.
A case records the md5 per -W and whether all of them agree (`w_invariant`).  Data only; inputs are regenerated from the spec by
w_case_seqs()."""
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util  # noqa: E402
from tests.golden.make_poa_golden import family, write_multi_fasta  # noqa: E402
from tests.golden.make_msa_n_golden import run_ref  # noqa: E402


def w_case_specs():
    return [dict(name="pad_medium", seed=53, n_seq=4, length=1500, rate=0.08, first_len=0, ws=[10, 30, 31, 32]),
            dict(name="pad_1000", seed=61, n_seq=3, length=1000, rate=0.1, first_len=0, ws=[10, 32]),
            dict(name="pad_300", seed=7, n_seq=5, length=300, rate=0.15, first_len=0, ws=[10, 28, 32]),
            dict(name="short_first_a", seed=71, n_seq=3, length=1200, rate=0.1, first_len=300, ws=[10, 3, 5]),
            dict(name="short_first_b", seed=73, n_seq=4, length=2000, rate=0.05, first_len=500, ws=[10, 4, 2])]


def w_case_seqs(c):
    seqs = [np.array(s) for s in family(c["seed"], c["n_seq"], c["length"], c["rate"], jitter=0)]
    if c["first_len"]:
        seqs[0] = seqs[0][:c["first_len"]]
    return seqs


def main():
    golden = []
    with tempfile.TemporaryDirectory() as td:
        fa, out = os.path.join(td, "in.fa"), os.path.join(td, "out.txt")
        for c in w_case_specs():
            write_multi_fasta(fa, w_case_seqs(c))
            res = [run_ref(fa, w, out) for w in c["ws"]]
            g = dict(c)
            g.update(md5_by_w={str(w): r[0] for w, r in zip(c["ws"], res)}, size_by_w={str(w): r[1] for w, r in zip(c["ws"], res)},
                     w_invariant=len({r[0] for r in res}) == 1)
            golden.append(g)
            print(c["name"], "invariant" if g["w_invariant"] else "W-dependent: %d distinct" % len({r[0] for r in res}), flush=True)
    with open(os.path.join(util.GOLDEN_DIR, "msa_w_golden.json"), "w") as f:
        json.dump(golden, f, indent=0)


if __name__ == "__main__":
    main()
