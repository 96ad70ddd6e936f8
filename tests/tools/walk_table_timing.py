"""One-off: config #3's pair (100 kbp) with the walk's band table off / at several corridor widths: kernel ms of the whole alignment,
bands crossed by look-up.  usage: walk_table_timing.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tsta_amd
from tests import util
g = json.load(open(os.path.join(util.GOLDEN_DIR, 'psa_big_golden.json')))
x = util.random_dna(g["seed_x"], g["len_x"]); y = util.mutate(x, g["seed_y"], g["rate"])
eng = tsta_amd.Engine(0)
ref = None
for table, corridor in ((0, 1024), (1, 256), (1, 512), (1, 1024), (1, 2048), (1, 4096)):
    eng.set_option("walk_table", table); eng.set_option("walk_corridor", corridor)
    ks = []
    for i in range(4):
        r = eng.psa_align(x, y)
        ks.append(round(eng.last_kernel_ms()[0], 3))
    if ref is None: ref = r
    print("walk_table", table, "corridor", corridor, "kernel_ms", ks, "jumps", eng.stat("last_walk_jumps"), "of", (len(y) - 1) // 128, "same output:", r == ref, flush=True)
