"""One-off: kernel ms of six config #3 alignments; with TSTA_SPLIT_CLOCK=1 the library prints when every solver band started and
how long its sweep took (s_memrealtime), with TSTA_SPLIT_NOEMIT=1 the emitters do nothing (solver chain alone; no trace!)."""
import sys, time, json, hashlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tsta_amd
from tests import util
g = json.load(open(os.path.join(util.GOLDEN_DIR, 'psa_big_golden.json')))
x = util.random_dna(g["seed_x"], g["len_x"]); y = util.mutate(x, g["seed_y"], g["rate"])
eng = tsta_amd.Engine(0)
ks = []
for i in range(6):
    ms, ra, rb = eng.psa_align(x, y)
    ks.append(round(eng.last_kernel_ms()[0], 3))
print(os.environ.get("TSTA_SPLIT_EMIT_WG"), os.environ.get("TSTA_SPLIT_NOEMIT"), "kernel_ms", ks, flush=True)
