#!/usr/bin/env python3
"""Large TSTA_msa check: runs the drop-in CLI (GPU) and, when present, the reference binary on the same generated
family and compares the output files.  usage: msa_large_check.py N_SEQ LENGTH RATE"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import util
from tests.golden.make_poa_golden import family, write_multi_fasta
n_seq, length, rate = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
fa = "/tmp/msa_large.fa"
write_multi_fasta(fa, family(4242, n_seq, length, rate))
def run(cmd):
    t = time.perf_counter(); r = subprocess.run(cmd, capture_output=True); dt = time.perf_counter() - t
    return r, dt
r, dt = run([os.path.join(ROOT, "tsta_amd/bin/TSTA_msa"), "-i", fa, "-o", "/tmp/ours.txt"])
print("ours rc", r.returncode, "%.2f s" % dt, r.stderr.decode()[-200:])
m1 = hashlib.md5(open("/tmp/ours.txt", "rb").read()).hexdigest()
ref = os.path.join(util.REF_DIR, "TSTA_msa")
if os.path.exists(ref) and util.cpu_has("avx512bw"):
    r, dt = run([ref, "-T", str(len(os.sched_getaffinity(0))), "-i", fa, "-o", "/tmp/ref.txt"])
    print("ref  rc", r.returncode, "%.2f s" % dt)
    m2 = hashlib.md5(open("/tmp/ref.txt", "rb").read()).hexdigest()
    print("md5", m1, m2, "MATCH" if m1 == m2 else "DIFFER")
else:
    print("md5", m1)
