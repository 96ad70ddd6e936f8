#!/usr/bin/env python3
"""Large TSTA_psa check: drop-in CLI (GPU) vs the reference binary on the same generated pair: stdout + output file.
usage: psa_large_check.py LENGTH RATE"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import util
length, rate = int(sys.argv[1]), float(sys.argv[2])
x = util.random_dna(99, length); y = util.mutate(x, 98, rate)
util.write_fasta("/tmp/pa.fa", "a", x, width=80); util.write_fasta("/tmp/pb.fa", "b", y)
def run(cmd):
    t = time.perf_counter(); r = subprocess.run(cmd, capture_output=True); return r, time.perf_counter() - t
for name in ("TSTA_psa", "TSTA_psa_notrace"):
    r, dt = run([os.path.join(ROOT, "tsta_amd/bin", name), "-1", "/tmp/pa.fa", "-2", "/tmp/pb.fa", "-o", "/tmp/ours.txt"])
    print(name, "ours rc", r.returncode, "%.2f s" % dt, r.stdout.decode().strip(), r.stderr.decode()[-200:])
    m1 = hashlib.md5(open("/tmp/ours.txt", "rb").read()).hexdigest() if name == "TSTA_psa" else ""
    ref = os.path.join(util.REF_DIR, name)
    if os.path.exists(ref) and util.cpu_has("avx512bw"):
        r2, dt2 = run([ref, "-T", str(len(os.sched_getaffinity(0))), "-1", "/tmp/pa.fa", "-2", "/tmp/pb.fa", "-o", "/tmp/ref.txt"])
        m2 = hashlib.md5(open("/tmp/ref.txt", "rb").read()).hexdigest() if name == "TSTA_psa" else ""
        print(name, "ref  rc", r2.returncode, "%.2f s" % dt2, r2.stdout.decode().strip(),
              "MATCH" if (r.stdout == r2.stdout and m1 == m2) else "DIFFER")
