#!/usr/bin/env python3
"""One-off soak on the GPU box (not part of the test suite): random sequence families through the drop-in TSTA_msa
(GPU) and the compiled reference (oracle/_ref/TSTA_msa), byte comparison of the output files."""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tests import util
from tests.golden.make_poa_golden import family, write_multi_fasta
ref = os.path.join(util.REF_DIR, "TSTA_msa")
assert os.path.exists(ref), "needs oracle/_ref/TSTA_msa"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    n_seq = int(rng.integers(2, 14)); length = int(rng.integers(300, 7000)); rate = float(rng.choice([0.02, 0.05, 0.1, 0.2]))
    sc = [(2, -5, -2, -4), (1, -1, -1, 0), (3, -4, -3, -4), (2, -3, -2, -4)][int(rng.integers(0, 4))]
    fa = "/tmp/soak_msa.fa"
    write_multi_fasta(fa, family(int(rng.integers(1 << 30)), n_seq, length, rate))
    args = ["-M", str(sc[0]), "-X", str(sc[1]), "-E", str(sc[2]), "-O", str(sc[3]), "-i", fa]
    a = subprocess.run([os.path.join(ROOT, "tsta_amd/bin/TSTA_msa")] + args + ["-o", "/tmp/ours.txt"], capture_output=True)
    b = subprocess.run([ref, "-T", "16"] + args + ["-o", "/tmp/ref.txt"], capture_output=True)
    if b.returncode != 0:
        print("trial", trial, "reference failed rc", b.returncode, (n_seq, length, rate, sc), flush=True)
        continue
    m1 = hashlib.md5(open("/tmp/ours.txt", "rb").read()).hexdigest(); m2 = hashlib.md5(open("/tmp/ref.txt", "rb").read()).hexdigest()
    same = a.returncode == 0 and m1 == m2 and a.stdout == b.stdout
    if not same:
        bad += 1
    print("trial", trial, (n_seq, length, rate, sc), "MATCH" if same else "DIFFER rc=%d" % a.returncode, flush=True)
print("mismatches:", bad)
