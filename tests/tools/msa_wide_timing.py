#!/usr/bin/env python3
"""One-off: TSTA_msa on a generated family, K4d (single launch) against K4 (launch per tile anti-diagonal), same output
expected.  usage: msa_wide_timing.py N_SEQ LENGTH RATE"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden.make_poa_golden import family, write_multi_fasta
n_seq, length, rate = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
fa = "/tmp/msa_wide.fa"
write_multi_fasta(fa, family(777, n_seq, length, rate))
for df in ("1", "0"):
    env = dict(os.environ, TSTA_POA_DATAFLOW=df, TSTA_MSA_TIMING="1")
    t = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "tsta_amd/bin/TSTA_msa"), "-i", fa, "-o", "/tmp/wide_%s.txt" % df], capture_output=True, env=env, timeout=600)
    dt = time.perf_counter() - t
    print("dataflow=%s rc %d %.2f s md5 %s %s" % (df, r.returncode, dt, hashlib.md5(open("/tmp/wide_%s.txt" % df, "rb").read()).hexdigest(),
          r.stderr.decode().strip().split("\n")[-1]), flush=True)
