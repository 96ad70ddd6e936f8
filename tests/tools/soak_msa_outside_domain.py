#!/usr/bin/env python3
"""One-off soak on the GPU box (not part of the test suite): inputs OUTSIDE the reference's domain — literal 'N's, sequences of
very different lengths (a short first one: the sequence then spans more blocks than the graph), random -W (1..32) and scoring —
through the drop-in TSTA_msa (GPU) and the compiled reference (oracle/_ref/TSTA_msa): byte comparison of the output files.
What the reference prints there is a property of its int8 arithmetic; the drop-in runs such fills in that arithmetic (K4e)."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tests import util
from tests.golden.make_poa_golden import family, write_multi_fasta
ref = os.path.join(util.REF_DIR, "TSTA_msa")
assert os.path.exists(ref), "needs oracle/_ref/TSTA_msa"
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0; done = 0; warned = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    n_seq = int(rng.integers(2, 8)); length = int(rng.integers(100, 3000)); rate = float(rng.choice([0.02, 0.05, 0.1, 0.2]))
    sc = [(2, -5, -2, -4), (2, -5, -2, -4), (1, -1, -1, 0), (3, -4, -3, -4), (2, -3, -2, -4), (4, -6, -3, -8)][int(rng.integers(0, 6))]
    w = int(rng.choice([10, 10, 10, 1, 2, 3, 5, 7, 16, 24, 30, 32]))
    seqs = [np.array(s) for s in family(int(rng.integers(1 << 30)), n_seq, length, rate)]
    kind = trial % 3
    if len(sys.argv) > 3 and sys.argv[3] == "scoring":
        # round 4: random SCORINGS on both sides of the routing rule (tsta_poa_reference_exact), 'N'-free equal-length or ragged
        # families: the drop-in must print the reference's bytes wherever the reference survives its own arithmetic
        sc = (int(rng.integers(1, 13)), 0, int(rng.integers(-10, 0)), int(rng.integers(-40, 1)))
        sc = (sc[0], int(rng.integers(-40, sc[0])), sc[2], sc[3])
        length = 64 * w * int(rng.integers(1, 3)) if trial % 2 else length
        seqs = [np.array(s) for s in family(int(rng.integers(1 << 30)), n_seq, min(length, 3000), rate, jitter=trial % 2 == 0)]
        kind = 3
    if kind in (0, 2):                                            # literal N's, anywhere
        for _ in range(int(rng.integers(1, 8))):
            k = int(rng.integers(0, n_seq)); seqs[k][int(rng.integers(0, len(seqs[k])))] = ord("N")
    if kind in (1, 2):                                            # ragged: every sequence keeps a random prefix
        for k in range(n_seq):
            seqs[k] = seqs[k][:max(20, int(len(seqs[k]) * float(rng.uniform(0.15, 1.0))))]
    fa = "/tmp/soak_out.fa"
    write_multi_fasta(fa, seqs)
    args = ["-M", str(sc[0]), "-X", str(sc[1]), "-E", str(sc[2]), "-O", str(sc[3]), "-W", str(w), "-i", fa]
    try:
        b = subprocess.run([ref, "-T", "8"] + args + ["-o", "/tmp/ref.txt"], capture_output=True, timeout=60)
    except subprocess.TimeoutExpired:
        print("trial", trial, "reference hung", flush=True); continue
    if b.returncode != 0:
        print("trial", trial, "reference failed rc", b.returncode, (n_seq, length, rate, sc, w), flush=True)
        continue
    a = subprocess.run([os.path.join(ROOT, "tsta_amd/bin/TSTA_msa")] + args + ["-o", "/tmp/ours.txt"], capture_output=True, timeout=600)
    m1 = hashlib.md5(open("/tmp/ours.txt", "rb").read()).hexdigest() if a.returncode == 0 else "-"
    m2 = hashlib.md5(open("/tmp/ref.txt", "rb").read()).hexdigest()
    same = a.returncode == 0 and m1 == m2 and a.stdout == b.stdout
    done += 1; bad += (not same); warned += (b"warning:" in a.stderr)
    print("trial", trial, ["N", "ragged", "N+ragged", "scoring"][kind], (n_seq, [len(s) for s in seqs], rate, sc, "W=%d" % w),
          "MATCH" if same else "DIFFER rc=%d %s" % (a.returncode, a.stderr.decode()[-200:]), "(warned)" if b"warning:" in a.stderr else "", flush=True)
print("compared:", done, "mismatches:", bad, "runs with a warning (fell back to the plain DP):", warned)
