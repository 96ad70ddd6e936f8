#!/usr/bin/env python3
"""One-off soak on the GPU box (not part of the test suite): random pairs through the drop-in TSTA_psa (GPU) and the
compiled reference (oracle/_ref/TSTA_psa), byte comparison of stdout and the output files."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tests import util
ref = os.path.join(util.REF_DIR, "TSTA_psa")
assert os.path.exists(ref), "needs oracle/_ref/TSTA_psa"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
def fasta(path, name, seq):
    with open(path, "wb") as f:
        f.write(b">" + name + b"\n" + bytes(seq) + b"\n")
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    n = int(rng.integers(200, 20000)); m = int(rng.integers(200, 20000))
    a = util.random_dna(int(rng.integers(1 << 30)), n)
    kind = trial % 4
    if kind == 0: b = util.mutate(a, int(rng.integers(1 << 30)), 0.02 + 0.2 * rng.random())[:m]
    elif kind == 1: b = util.random_dna(int(rng.integers(1 << 30)), m)
    elif kind == 2: b = util.low_complexity(int(rng.integers(1 << 30)), m, 2)
    else: b = np.concatenate([a[: n // 3], a[2 * n // 3:]])
    sc = [(2, -5, -2, -4), (1, -1, -1, 0), (3, -4, -3, -4), (2, -3, -2, -4), (5, -4, -1, -10)][int(rng.integers(0, 5))]
    if len(sys.argv) > 3 and sys.argv[3] == "wide":
        # round 4: random scorings anywhere INSIDE the reference-exactness domain (tsta_psa_reference_exact), shorter and structurally
        # more varied pairs: identical, homopolymers, a few bases only
        import tsta_amd
        while True:
            sc = (int(rng.integers(0, 100)), int(rng.integers(-128, 1)), int(rng.integers(-28, 1)), int(rng.integers(-62, 1)))
            if tsta_amd.psa_reference_exact(M=sc[0], X=sc[1], E=sc[2], O=sc[3])[0]:
                break
        n = int(rng.integers(1, 4000)); m = int(rng.integers(1, 4000))
        if trial % 7 == 0: n = int(rng.integers(1, 6)); m = int(rng.integers(1, 6))
        a = util.random_dna(int(rng.integers(1 << 30)), n)
        k7 = trial % 6
        if k7 == 0: b = util.mutate(a, int(rng.integers(1 << 30)), 0.02 + 0.3 * rng.random())[:m]
        elif k7 == 1: b = util.random_dna(int(rng.integers(1 << 30)), m)
        elif k7 == 2: b = util.low_complexity(int(rng.integers(1 << 30)), m, int(rng.integers(1, 4)))
        elif k7 == 3: b = np.concatenate([a[: n // 3], a[2 * n // 3:]])
        elif k7 == 4: b = a.copy()
        else: a = np.full(n, ord("A"), np.uint8); b = np.full(m, ord("A"), np.uint8)
        if len(b) == 0: b = a[:1]
        if len(sys.argv) > 4 and sys.argv[4] == "alphabet":      # bytes that are not ACGT: PSA compares them as they are (psa.c:248-259)
            def sprinkle(x):
                x = x.copy()
                hit = rng.random(len(x)) < 0.03
                x[hit] = rng.choice(np.frombuffer(b"NnacgtRYKMSW*-", dtype=np.uint8), int(hit.sum()))
                return x
            a = sprinkle(a); b = sprinkle(b)
    fasta("/tmp/sa.fa", b"a", a); fasta("/tmp/sb.fa", b"b", b)
    args = ["-M", str(sc[0]), "-X", str(sc[1]), "-E", str(sc[2]), "-O", str(sc[3]), "-1", "/tmp/sa.fa", "-2", "/tmp/sb.fa"]
    x = subprocess.run([os.path.join(ROOT, "tsta_amd/bin/TSTA_psa")] + args + ["-o", "/tmp/ours.txt"], capture_output=True)
    y = subprocess.run([ref, "-T", "16"] + args + ["-o", "/tmp/ref.txt"], capture_output=True)
    same = x.returncode == 0 and y.returncode == 0 and x.stdout == y.stdout and open("/tmp/ours.txt", "rb").read() == open("/tmp/ref.txt", "rb").read()
    bad += 0 if same else 1
    print("trial", trial, (n, len(b), sc), "MATCH" if same else "DIFFER", x.stdout[:30], y.stdout[:30], flush=True)
print("mismatches:", bad)
