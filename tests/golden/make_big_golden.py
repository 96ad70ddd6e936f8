#!/usr/bin/env python3
"""Generate tests/golden/psa_big_golden.json: maxsorce + md5 of the output file of the UNMODIFIED reference TSTA_psa
(oracle/_ref/TSTA_psa) on the config #3 pair of bench_configs.py: x = random_dna(777, 100000), y = mutate(x, 778, 0.05)
(100 000 x ~100 134 with traceback; the reference needs ~30 GB of host memory and about a minute for it).  Data only."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util  # noqa: E402


def main():
    x = util.random_dna(777, 100_000); y = util.mutate(x, 778, 0.05)
    with tempfile.TemporaryDirectory() as td:
        fa, fb, out = os.path.join(td, "a.fa"), os.path.join(td, "b.fa"), os.path.join(td, "out.txt")
        util.write_fasta(fa, "x", x); util.write_fasta(fb, "y", y)
        for attempt in range(3):                     # the reference's thread pool can deadlock: time limit + retry
            try:
                r = subprocess.run([os.path.join(util.REF_DIR, "TSTA_psa"), "-T", "8", "-1", fa, "-2", fb, "-o", out],
                                   capture_output=True, timeout=900)
                if r.returncode == 0 and b"maxsorce=" in r.stdout:
                    break
            except subprocess.TimeoutExpired:
                pass
        else:
            raise RuntimeError("reference failed")
        data = open(out, "rb").read()
        g = dict(seed_x=777, seed_y=778, rate=0.05, len_x=len(x), len_y=len(y), scoring=dict(M=2, X=-5, E=-2, O=-4),
                 maxsorce=int(r.stdout.decode().split("maxsorce=")[1].split()[0]), md5=hashlib.md5(data).hexdigest(), size=len(data),
                 sha256=hashlib.sha256(data).hexdigest())
    with open(os.path.join(util.GOLDEN_DIR, "psa_big_golden.json"), "w") as f:
        json.dump(g, f, indent=1)
    print(g)


if __name__ == "__main__":
    main()
