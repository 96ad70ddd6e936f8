#!/usr/bin/env python3
"""One-off diagnostics: where and when the couples of the headline shard ran (TSTA_PSA_CLOCK=2 -> per-couple log of
psa_score_packed_kernel).  Prints the spread of couple durations per round and how the waves sat on the SIMDs."""
import collections
import os
import sys

import numpy as np
import torch

os.environ["TSTA_PSA_CLOCK"] = "2"
os.environ["TSTA_PSA_CLOCK_FILE"] = "/tmp/psa_wave_log.txt"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tsta_amd  # noqa: E402
from bench import make_shard  # noqa: E402

P, L = int(os.environ.get("P", 8192)), int(os.environ.get("L", 50000))
dev = torch.device("cuda", 0)
eng = tsta_amd.Engine(0)
seqs, off1, len1, off2, len2 = make_shard(P, L, 1000, dev)
scores = torch.empty(P, dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
with torch.cuda.stream(st):
    for i in range(2):
        eng.psa_score_batch_dev(tsta_amd.Params(), P, seqs.data_ptr(), off1.data_ptr(), len1.data_ptr(), off2.data_ptr(), len2.data_ptr(), scores.data_ptr(), L, st.cuda_stream)
        torch.cuda.synchronize()
print("kernel ms", eng.last_kernel_ms()[0])
rows = np.loadtxt("/tmp/psa_wave_log.txt")
t0, t1 = rows[:, 1] / 1e3, rows[:, 2] / 1e3                      # ms
dur = t1 - t0
print("couples %d  start min/max %.1f/%.1f ms  end max %.1f ms" % (len(rows), t0.min(), t0.max(), t1.max()))
first = t0 < 5.0
print("round 1: n %d  duration ms min/median/max %.1f %.1f %.1f" % (first.sum(), dur[first].min(), np.median(dur[first]), dur[first].max()))
if (~first).any():
    print("later  : n %d  duration ms min/median/max %.1f %.1f %.1f" % ((~first).sum(), dur[~first].min(), np.median(dur[~first]), dur[~first].max()))
simd = collections.Counter((int(r[3]), int(r[4]), int(r[5]), int(r[6])) for r in rows[first])
print("round 1: SIMDs used %d; waves per SIMD histogram %s" % (len(simd), sorted(collections.Counter(simd.values()).items())))
cu = collections.Counter((int(r[3]), int(r[4]), int(r[5])) for r in rows[first])
print("round 1: CUs used %d; waves per CU histogram %s" % (len(cu), sorted(collections.Counter(cu.values()).items())))
xcc = collections.Counter(int(r[3]) for r in rows[first])
print("round 1: waves per XCC %s" % sorted(xcc.items()))
# duration of round-1 couples by how many waves shared their SIMD
by = collections.defaultdict(list)
for r, d in zip(rows[first], dur[first]):
    by[simd[(int(r[3]), int(r[4]), int(r[5]), int(r[6]))]].append(d)
for k in sorted(by):
    print("  SIMD shared by %d wave(s): n %d median %.1f ms" % (k, len(by[k]), np.median(by[k])))
hist, edges = np.histogram(dur, bins=12)
print("duration histogram (all):", list(zip(np.round(edges[:-1]).astype(int).tolist(), hist.tolist())))
