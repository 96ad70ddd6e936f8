#!/usr/bin/env python3
"""Generate tests/golden/msa_golden.json: md5 + size of the output file of the UNMODIFIED reference TSTA_msa
(oracle/_ref/TSTA_msa, built by `make -C oracle ref`) on seeded multi-FASTA inputs + the README example.
Data only; inputs are regenerated from the spec by tests (family() in make_poa_golden.py)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util  # noqa: E402
from tests.golden.make_poa_golden import family, write_multi_fasta  # noqa: E402

PARAMS = [dict(M=2, X=-5, E=-2, O=-4), dict(M=2, X=-3, E=-2, O=-4), dict(M=1, X=-1, E=-1, O=0), dict(M=5, X=-4, E=-1, O=-10)]


def case_specs():
    out = []
    for seed in range(1, 41):
        n_seq = 2 + seed % 7                       # the reference crashes on a single-sequence input
        length = [60, 150, 333, 640, 700, 1300][seed % 6]
        rate = [0.02, 0.1, 0.2, 0.3][seed % 4]
        out.append(dict(seed=seed * 13, n_seq=n_seq, length=length, rate=rate, jitter=0 if seed % 3 else 1,
                        scoring=PARAMS[seed % 4]))
    return out


def case_seqs(c):
    seqs = family(c["seed"], c["n_seq"], c["length"], c["rate"], jitter=c["jitter"])
    if c["jitter"]:
        seqs = sorted(seqs, key=len, reverse=True)   # longest first: keeps the sequence within the graph's L-block count
    return seqs


def run_ref(fasta, sc, out):
    r = subprocess.run([os.path.join(util.REF_DIR, "TSTA_msa"), "-M", str(sc["M"]), "-X", str(sc["X"]), "-E", str(sc["E"]),
                        "-O", str(sc["O"]), "-T", "4", "-i", fasta, "-o", out], capture_output=True, timeout=600)
    assert r.returncode == 0, r
    data = open(out, "rb").read()
    return hashlib.md5(data).hexdigest(), len(data), r.stdout.decode(), r.stderr.decode()


def main():
    golden = []
    with tempfile.TemporaryDirectory() as td:
        fa, out = os.path.join(td, "in.fa"), os.path.join(td, "out.txt")
        for c in case_specs():
            write_multi_fasta(fa, case_seqs(c))
            md5, size, so, se = run_ref(fa, c["scoring"], out)
            g = dict(c); g.update(kind="family", md5=md5, size=size, stdout=so, stderr=se)
            golden.append(g)
        for sc in PARAMS[:2]:                        # README example (README.md:26-40) and example_msa.sh:4
            md5, size, so, se = run_ref("/root/reference/example/msa/seq/seq1.fa", sc, out)
            golden.append(dict(kind="example", scoring=sc, md5=md5, size=size, stdout=so, stderr=se))
    with open(os.path.join(util.GOLDEN_DIR, "msa_golden.json"), "w") as f:
        json.dump(golden, f, indent=0)
    print("wrote", len(golden), "cases")


if __name__ == "__main__":
    main()
