"""One-off (GPU box): the README example with a literal 'N' put into three of its sequences — the drop-in (blocked fill K4n for
those steps) against the reference binary: same bytes? how long?"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import util
src = open(os.path.join(util.GOLDEN_DIR, "msa_seq1.fa")).read().split(">")[1:]
recs = [(r.split("\n", 1)[0], "".join(r.split("\n")[1:])) for r in src]
for k, pos in ((1, 1234), (2, 17), (3, 4999)):
    name, s = recs[k]; recs[k] = (name, s[:pos] + "N" + s[pos + 1:])
fa = "/tmp/readme_n.fa"
open(fa, "w").write("".join(">%s\n%s\n" % r for r in recs))
def run(cmd):
    t0 = time.perf_counter(); r = subprocess.run(cmd, capture_output=True, timeout=600); dt = time.perf_counter() - t0
    return dt, r
for w in (10, 5, 16):
    dt, r = run([os.path.join(ROOT, "tsta_amd/bin/TSTA_msa"), "-T", "4", "-W", str(w), "-i", fa, "-o", "/tmp/o1.txt"])
    dt2, r2 = run([os.path.join(ROOT, "oracle/_ref/TSTA_msa"), "-T", "64", "-W", str(w), "-i", fa, "-o", "/tmp/o2.txt"])
    same = hashlib.md5(open("/tmp/o1.txt", "rb").read()).hexdigest() == hashlib.md5(open("/tmp/o2.txt", "rb").read()).hexdigest()
    print("-W %d: drop-in %.2f s, reference %.2f s, identical output: %s | %s" % (w, dt, dt2, same, r.stderr.decode().strip().split("\n")[-1][:200]), flush=True)
