"""One-off: config #3 (one 100 kbp pair with traceback) under the launch-per-tile pipeline and the persistent single-launch
fill (sc1 hand-off), rows per lane 1 / 2 / 4; all outputs must agree (and equal the reference's md5 when the fixture exists)."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tsta_amd
from tests import util

eng = tsta_amd.Engine(0)
x = util.random_dna(777, 100_000); y = util.mutate(x, 778, 0.05)
gold = None
gp = os.path.join(util.GOLDEN_DIR, "psa_big_golden.json")
if os.path.exists(gp):
    gold = json.load(open(gp))
ref = None
reps = int(os.environ.get("REPS", 3))
for persistent in (0, 1):
    for R in (2, 1, 4):
        eng.set_option("trace_persistent", persistent); eng.set_option("trace_rows", R)
        best = None
        for rep in range(reps):
            t0 = time.perf_counter()
            ms, ra, rb = eng.psa_align(x, y)
            wall = time.perf_counter() - t0
            k, nl = eng.last_kernel_ms()
            best = k if best is None else min(best, k)
            md5 = hashlib.md5(b">1\n" + ra + b"\n>2\n" + rb).hexdigest()
            if ref is None:
                ref = (ms, md5)
            ok = (ms, md5) == ref and (gold is None or (ms == gold["maxsorce"] and md5 == gold["md5"]))
            print(json.dumps({"persistent": persistent, "R": R, "rep": rep, "kernel_ms": k, "launches": nl, "wall_ms": wall * 1e3, "maxsorce": ms,
                              "md5": md5, "ok": ok, "vs_reference_md5": None if gold is None else md5 == gold["md5"]}), flush=True)
            assert ok, "output differs"
eng.close()
