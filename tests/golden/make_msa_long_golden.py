#!/usr/bin/env python3
"""Generate tests/golden/msa_long_golden.json: md5 + size + stdout/stderr of the UNMODIFIED reference TSTA_msa
(oracle/_ref/TSTA_msa, built by `make -C oracle ref`) on LONG seeded families — the lengths the reference's README
recommends the tool for (README.md:42-43) — where the single-launch POA fill runs hundreds of 128-column chunks and the
linked walker crosses hundreds of windows.  Data only; the tests regenerate the inputs from the spec."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util  # noqa: E402
from tests.golden.make_msa_golden import run_ref  # noqa: E402
from tests.golden.make_poa_golden import family, write_multi_fasta  # noqa: E402

SPECS = [dict(seed=9001, n_seq=3, length=30_000, rate=0.10, jitter=0, scoring=dict(M=2, X=-5, E=-2, O=-4)),
         dict(seed=9002, n_seq=4, length=12_000, rate=0.15, jitter=0, scoring=dict(M=2, X=-3, E=-2, O=-4))]


def long_case_seqs(c):
    return family(c["seed"], c["n_seq"], c["length"], c["rate"], jitter=c["jitter"])


def main():
    golden = []
    with tempfile.TemporaryDirectory() as td:
        fa, out = os.path.join(td, "in.fa"), os.path.join(td, "out.txt")
        for c in SPECS:
            write_multi_fasta(fa, long_case_seqs(c))
            t0 = time.time()
            md5, size, so, se = run_ref(fa, c["scoring"], out)
            g = dict(c); g.update(kind="long", md5=md5, size=size, stdout=so, stderr=se, reference_seconds=round(time.time() - t0, 2))
            golden.append(g)
            print(g["seed"], md5, size, g["reference_seconds"], "s")
    with open(os.path.join(util.GOLDEN_DIR, "msa_long_golden.json"), "w") as f:
        json.dump(golden, f, indent=0)
    print("wrote", len(golden), "cases")


if __name__ == "__main__":
    main()
