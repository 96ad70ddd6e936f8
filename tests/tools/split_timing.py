"""One-off: config #3 pair through the single-pair trace fills (split = solver bands + emitter tiles / quad), kernel ms + md5 of the
reference's output file, then small sizes in the split form against the oracle."""
import sys, time, json, hashlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tsta_amd
from tests import util
g = json.load(open(os.path.join(util.GOLDEN_DIR, 'psa_big_golden.json')))
x = util.random_dna(g["seed_x"], g["len_x"]); y = util.mutate(x, g["seed_y"], g["rate"])
eng = tsta_amd.Engine(0)
for name, opts in (("split", dict(trace_split=1)), ("quad", dict(trace_split=0)), ("split", dict(trace_split=1))):
    for k, v in opts.items():
        eng.set_option(k, v)
    ks = []
    for i in range(3):
        ms, ra, rb = eng.psa_align(x, y)
        ks.append(round(eng.last_kernel_ms()[0], 3))
    data = b">1\n" + ra + b"\n>2\n" + rb
    print(name, "kernel_ms", ks, "md5ok", hashlib.md5(data).hexdigest() == g["md5"], "reruns", eng.stat("single_launch_reruns"), flush=True)
# small sizes vs oracle
eng.set_option("trace_split", 2)
for (l1, l2) in [(70, 64), (129, 128), (1000, 257), (3001, 2999), (9000, 8200), (20000, 19000)]:
    a = util.random_dna(l1 * 3 + l2, l1); b = util.mutate(a, l1, 0.08)[:l2]
    got = eng.psa_align(a, b); o = util.oracle_align(a, b)
    print(l1, l2, got == (o[0], o[2], o[3]), "reruns", eng.stat("single_launch_reruns"), flush=True)
