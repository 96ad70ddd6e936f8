"""One-off soak of the in-launch hand-offs (handoff.h: sc1 stores / polls, no fences) under repetition and uneven load:
  (1) config #3 pair, single-launch K2, against the reference's md5, many times, with a second engine hammering the chip;
  (2) 64 x 20 kbp pairs with traceback, dataflow K2p, against the launch pipeline's result, many times;
  (3) the README POA steps, K4d, against the reference dumps, many times.
Prints one line per phase; exits non-zero on the first difference."""
import hashlib, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tsta_amd
from tests import util
from tests.test_poa_oracle_golden import FIXTURES, check_against_fixture, load

reps = int(os.environ.get("SOAK_REPS", 20))
eng = tsta_amd.Engine(0)
noise = tsta_amd.Engine(0)
stop = False
def hammer():
    pairs = [(util.random_dna(10 + i, 3000 + 17 * i), util.random_dna(99 + i, 2500)) for i in range(600)]
    while not stop:
        noise.psa_score_batch(pairs)
th = threading.Thread(target=hammer); th.start()
try:
    g = json.load(open(os.path.join(util.GOLDEN_DIR, "psa_big_golden.json")))
    x = util.random_dna(g["seed_x"], g["len_x"]); y = util.mutate(x, g["seed_y"], g["rate"])
    t0 = time.time()
    for r in range(reps):
        ms, ra, rb = eng.psa_align(x, y)
        assert ms == g["maxsorce"] and hashlib.md5(b">1\n" + ra + b"\n>2\n" + rb).hexdigest() == g["md5"], ("config 3", r)
    print("config #3 single-launch fill (solver bands + emitter tiles): %d runs identical to the reference md5 (%.1f s, chip shared with a second engine; "
          "%d of them run again through the launch pipeline after an abandoned wait)" % (reps, time.time() - t0, eng.stat("single_launch_reruns")), flush=True)

    pairs = []
    for i in range(64):
        a = util.random_dna(5000 + i, 20000 + 311 * (i % 5)); pairs.append((a, util.mutate(a, 6000 + i, 0.05)))
    eng.set_option("trace_persistent", 0); want = eng.psa_align_batch(pairs); eng.set_option("trace_persistent", 1)
    t0 = time.time()
    for r in range(reps):
        assert eng.psa_align_batch(pairs) == want, ("batch", r)
    print("64 ragged pairs with traceback, dataflow K2p: %d runs identical to the launch pipeline (%.1f s)" % (reps, time.time() - t0), flush=True)

    t0 = time.time()
    fx = [f for f in FIXTURES if "readme" in f or "medium" in f]
    for r in range(reps):
        for path in fx:
            z, sc = load(path)
            best, last, src, es, fs = eng.poa_fill(z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"], tsta_amd.Params(**sc))
            check_against_fixture(z, best, last, src, es, fs)
    print("K4d on %d reference dumps: %d rounds identical (%.1f s)" % (len(fx), reps, time.time() - t0), flush=True)
finally:
    stop = True; th.join()
eng.close(); noise.close()
