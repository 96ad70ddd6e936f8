"""One-off: the README MSA example (BASELINE config #4) through the TSTA_msa drop-in and through the reference binary on the
same box: wall time of the whole process and, for the drop-in, where it goes (TSTA_MSA_TIMING=1)."""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fa = os.path.join(ROOT, "tests", "golden", "msa_seq1.fa")
def run(exe, env=None):
    best = None
    for _ in range(5):
        t = time.perf_counter()
        p = subprocess.run([exe, "-i", fa, "-o", "/tmp/o.txt"], capture_output=True, env=env, timeout=120)
        w = time.perf_counter() - t
        if best is None or w < best[0]:
            best = (w, p.stderr.decode().strip().split("\n")[-1], hashlib.md5(open("/tmp/o.txt", "rb").read()).hexdigest())
    return best
ours = run(os.path.join(ROOT, "tsta_amd/bin/TSTA_msa"), dict(os.environ, TSTA_MSA_TIMING="1"))
print("drop-in   best of 5: %.3f s  md5 %s  %s" % (ours[0], ours[2], ours[1]))
ref = os.path.join(ROOT, "oracle/_ref/TSTA_msa")
if os.path.exists(ref):
    for T in (10, 64):
        try:
            b = None
            for _ in range(5):
                t = time.perf_counter()
                subprocess.run([ref, "-T", str(T), "-i", fa, "-o", "/tmp/r.txt"], capture_output=True, timeout=60)
                w = time.perf_counter() - t
                b = w if b is None else min(b, w)
            print("reference -T %d best of 5: %.3f s  md5 %s" % (T, b, hashlib.md5(open("/tmp/r.txt", "rb").read()).hexdigest()))
        except subprocess.TimeoutExpired:
            print("reference -T %d: timed out (its thread pool can deadlock)" % T)
