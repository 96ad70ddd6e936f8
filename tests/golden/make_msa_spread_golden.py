#!/usr/bin/env python3
"""Generate tests/golden/msa_spread_golden.json: inputs on which the reference's POA output differs from the plain DP because two
predecessor rows of one node lie further apart than its int8 offsets hold exactly (the clipped offset of a predecessor that is cut
off from the maximum still feeds the F / E flags of the cell, c-t-simd.c:203-222,521-522,564-565), at the DEFAULT scoring and -W.
Found in round 4 by a campaign of structurally varied random families (`campaign_family`, also used by
tests/tools/soak_msa_campaign.py): seed 3066 — seven 200-300 bp sequences at 30 % divergence whose graph gets a 90-node jump edge
(predecessor spread 179) — prints 19 fsource bytes that are not the DP's; the engine's run-time limit on that spread (100 minus what
one row can add; it was 354 - 2m until then) now sends such fills through the reference's own arithmetic.  Plus the family on which a
random-scoring soak found the same effect at M=1 X=-9 E=-1 O=-16.  Data only: seeds, scoring, -W, md5 of the reference's output."""
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


def campaign_family(seed):
    """2-7 sequences of 80-2600 bp at 1-30 % divergence, with long deletions / low-complexity insertions / ragged prefixes."""
    rng = np.random.default_rng(seed)
    n_seq = int(rng.integers(2, 8)); length = int(rng.integers(80, 2600)); rate = float(rng.choice([0.01, 0.03, 0.06, 0.1, 0.15, 0.2, 0.3]))
    seqs = [np.array(s) for s in family(int(rng.integers(1 << 30)), n_seq, length, rate, jitter=int(rng.integers(0, 2)))]
    kind = int(rng.integers(0, 5))
    if kind == 1:                                   # long deletions in some sequences
        for k in range(n_seq):
            if rng.random() < 0.5 and len(seqs[k]) > 200:
                a = int(rng.integers(0, len(seqs[k]) - 100)); g = int(rng.integers(20, min(300, len(seqs[k]) - a - 10)))
                seqs[k] = np.concatenate([seqs[k][:a], seqs[k][a + g:]])
    elif kind == 2:                                 # low-complexity insertions
        for k in range(n_seq):
            if rng.random() < 0.5:
                a = int(rng.integers(0, len(seqs[k]))); rep = util.low_complexity(int(rng.integers(1 << 30)), int(rng.integers(10, 120)), int(rng.integers(1, 4)))
                seqs[k] = np.concatenate([seqs[k][:a], rep, seqs[k][a:]])
    elif kind == 3:                                 # ragged prefixes
        for k in range(1, n_seq):
            seqs[k] = seqs[k][:max(30, int(len(seqs[k]) * float(rng.uniform(0.5, 1.0))))]
    return seqs


def soak_scoring_family(trial_wanted=164, seed=41):
    """The family trial 164 of `soak_msa_outside_domain.py 41 400 scoring` drew (the generator's draws replayed)."""
    rng = np.random.default_rng(seed)
    seqs = None
    for trial in range(trial_wanted + 1):
        n_seq = int(rng.integers(2, 8)); length = int(rng.integers(100, 3000)); rate = float(rng.choice([0.02, 0.05, 0.1, 0.2]))
        rng.integers(0, 6)
        w = int(rng.choice([10, 10, 10, 1, 2, 3, 5, 7, 16, 24, 30, 32]))
        family(int(rng.integers(1 << 30)), n_seq, length, rate)
        m = int(rng.integers(1, 13)); rng.integers(-10, 0); rng.integers(-40, 1); rng.integers(-40, m)
        length = 64 * w * int(rng.integers(1, 3)) if trial % 2 else length
        seqs = [np.array(s) for s in family(int(rng.integers(1 << 30)), n_seq, min(length, 3000), rate, jitter=trial % 2 == 0)]
    return seqs


def spread_case_seqs(c):
    return campaign_family(c["seed"]) if c["kind"] == "campaign" else soak_scoring_family()


CASES = [dict(kind="campaign", seed=3066, scoring=dict(M=2, X=-5, E=-2, O=-4), W=10),
         dict(kind="campaign", seed=3066, scoring=dict(M=2, X=-5, E=-2, O=-4), W=16),
         dict(kind="campaign", seed=3066, scoring=dict(M=2, X=-3, E=-2, O=-4), W=16),
         dict(kind="soak164", seed=41, scoring=dict(M=1, X=-9, E=-1, O=-16), W=24),
         dict(kind="soak164", seed=41, scoring=dict(M=1, X=-9, E=-1, O=-16), W=10),
         dict(kind="soak164", seed=41, scoring=dict(M=2, X=-5, E=-2, O=-4), W=10)]


def main():
    out = []
    with tempfile.TemporaryDirectory() as td:
        fa, fo = os.path.join(td, "in.fa"), os.path.join(td, "o.txt")
        for c in CASES:
            write_multi_fasta(fa, spread_case_seqs(c))
            sc = c["scoring"]
            r = subprocess.run([os.path.join(util.REF_DIR, "TSTA_msa"), "-M", str(sc["M"]), "-X", str(sc["X"]), "-E", str(sc["E"]), "-O", str(sc["O"]),
                                "-T", "2", "-W", str(c["W"]), "-i", fa, "-o", fo], capture_output=True, timeout=120)
            assert r.returncode == 0
            data = open(fo, "rb").read()
            out.append(dict(c, md5=hashlib.md5(data).hexdigest(), size=len(data), lengths=[len(s) for s in spread_case_seqs(c)]))
    with open(os.path.join(util.GOLDEN_DIR, "msa_spread_golden.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
