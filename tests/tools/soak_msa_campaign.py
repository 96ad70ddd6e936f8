#!/usr/bin/env python3
"""One-off soak on the GPU box (round 4): structurally varied random families — 2-7 sequences of 80-2600 bp at 1-30 % divergence, with
long deletions, low-complexity insertions, ragged prefixes — through the drop-in TSTA_msa (GPU; whatever path its routing picks) and
the compiled reference, byte comparison of the output files.  usage: soak_msa_campaign.py "M,X,E,O" W n_cases [first_seed]"""
import hashlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import util
from tests.golden.make_poa_golden import write_multi_fasta


from tests.golden.make_msa_spread_golden import campaign_family as make   # the generator the spread fixture was found with


if __name__ == "__main__":
    sc = [int(x) for x in sys.argv[1].split(",")]; W = int(sys.argv[2]); N = int(sys.argv[3]); base = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    ref = os.path.join(util.REF_DIR, "TSTA_msa"); ours = os.path.join(ROOT, "tsta_amd/bin/TSTA_msa")
    bad = done = refbad = 0
    for seed in range(base, base + N):
        fa = "/tmp/camp.fa"
        write_multi_fasta(fa, make(seed))
        args = ["-M", str(sc[0]), "-X", str(sc[1]), "-E", str(sc[2]), "-O", str(sc[3]), "-W", str(W), "-i", fa]
        try:
            b = subprocess.run([ref, "-T", "8"] + args + ["-o", "/tmp/camp_ref.txt"], capture_output=True, timeout=60)
        except subprocess.TimeoutExpired:
            refbad += 1; continue
        if b.returncode != 0:
            refbad += 1; continue
        a = subprocess.run([ours] + args + ["-o", "/tmp/camp_ours.txt"], capture_output=True, timeout=600)
        same = a.returncode == 0 and hashlib.md5(open("/tmp/camp_ours.txt", "rb").read()).digest() == hashlib.md5(open("/tmp/camp_ref.txt", "rb").read()).digest()
        done += 1; bad += (not same)
        if not same:
            print("seed", seed, "DIFFER rc=%d" % a.returncode, a.stderr.decode()[-160:], flush=True)
        if done % 100 == 0:
            print("...", done, "compared,", bad, "mismatches", flush=True)
    print("scoring", sc, "W", W, "compared:", done, "mismatches:", bad, "reference failed / hung:", refbad, flush=True)
