#!/usr/bin/env python3
"""One-off soak on the GPU box (round 4): a few LONG families (3-5 sequences of 4-20 kbp, 2-15 % divergence, some with long
deletions) through the drop-in TSTA_msa and the compiled reference, byte comparison.  usage: soak_msa_long.py seed n_cases"""
import hashlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import util
from tests.golden.make_poa_golden import family, write_multi_fasta
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
ref = os.path.join(util.REF_DIR, "TSTA_msa"); ours = os.path.join(ROOT, "tsta_amd/bin/TSTA_msa")
bad = done = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    n_seq = int(rng.integers(3, 6)); length = int(rng.integers(4000, 20000)); rate = float(rng.choice([0.02, 0.05, 0.1, 0.15]))
    seqs = [np.array(s) for s in family(int(rng.integers(1 << 30)), n_seq, length, rate, jitter=int(rng.integers(0, 2)))]
    if trial % 2:
        for k in range(1, n_seq):
            a = int(rng.integers(0, len(seqs[k]) - 1500)); g = int(rng.integers(100, 1200))
            seqs[k] = np.concatenate([seqs[k][:a], seqs[k][a + g:]])
    seqs.sort(key=lambda s: -len(s))                       # (first longest: the reference's block geometry stays inside its domain)
    fa = "/tmp/long.fa"; write_multi_fasta(fa, seqs)
    t0 = time.time()
    b = subprocess.run([ref, "-T", "32", "-i", fa, "-o", "/tmp/long_ref.txt"], capture_output=True, timeout=600)
    t1 = time.time()
    env = dict(os.environ, TSTA_MSA_SPREAD="1")
    a = subprocess.run([ours, "-i", fa, "-o", "/tmp/long_ours.txt"], capture_output=True, timeout=600, env=env)
    t2 = time.time()
    same = a.returncode == 0 and b.returncode == 0 and hashlib.md5(open("/tmp/long_ours.txt", "rb").read()).digest() == hashlib.md5(open("/tmp/long_ref.txt", "rb").read()).digest()
    done += 1; bad += not same
    k4e = sum(1 for l in a.stderr.decode().splitlines() if "K4e" in l)
    print("trial", trial, [len(s) for s in seqs], rate, "MATCH" if same else "DIFFER", "reference %.1f s, drop-in %.1f s, steps in K4e: %d" % (t1 - t0, t2 - t1, k4e), flush=True)
print("compared:", done, "mismatches:", bad)
