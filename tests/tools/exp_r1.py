import sys, time, json, hashlib
sys.path.insert(0, '/root/repo')
import numpy as np
import tsta_amd
from tests import util
g = json.load(open('/root/repo/tests/golden/psa_big_golden.json'))
x = util.random_dna(g["seed_x"], g["len_x"]); y = util.mutate(x, g["seed_y"], g["rate"])
eng = tsta_amd.Engine(0)
for R, lds in ((2, 1), (1, 2), (1, 0), (2, 0)):
    eng.set_option("trace_rows", R); eng.set_option("trace_lds", lds)
    ks = []
    for i in range(3):
        ms, ra, rb = eng.psa_align(x, y)
        ks.append(eng.last_kernel_ms()[0])
    data = b">1\n" + ra + b"\n>2\n" + rb
    print("R", R, "lds", lds, "kernel_ms", ks, "md5ok", hashlib.md5(data).hexdigest() == g["md5"], flush=True)
