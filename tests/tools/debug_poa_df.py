import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, tsta_amd
from tests import util
from tests.test_poa_fill_gpu import random_dag
import faulthandler; faulthandler.dump_traceback_later(40, exit=True)
e = tsta_amd.Engine(0)
print('engine ok', flush=True)
for df in (0, 1):
  e.set_option('poa_dataflow', df); print('dataflow', df, flush=True)
  for seed, V, mp, n in [(1, 20, 1, 40), (1, 50, 3, 40), (2, 100, 2, 300), (3, 64, 41, 700), (5, 700, 12, 2000)]:
      pred_off, pred, base, is_sink, seq = random_dag(seed, V, mp, n)
      t0 = time.time()
      try:
          got = e.poa_fill(pred_off, pred, base, is_sink, seq)
      except Exception as ex:
          print(seed, V, mp, n, "ERROR", ex, "%.2f s" % (time.time() - t0), flush=True); continue
      want = util.oracle_poa_fill(pred_off, pred, base, is_sink, seq)
      ok = got[0] == want[0] and np.array_equal(got[1], want[1]) and all(np.array_equal(g, w) for g, w in zip(got[2:], want[2:]))
      print(seed, V, mp, n, "ok" if ok else "MISMATCH", "%.3f s" % (time.time() - t0), "kernel_ms", e.last_kernel_ms(), flush=True)
      if not ok:
          for name, g, w in zip(("source", "esource", "fsource"), got[2:], want[2:]):
              bad = np.argwhere(g != w)
              print("  ", name, len(bad), bad[:6].tolist())
          print("   last", np.argwhere(got[1] != want[1])[:6].tolist(), got[0], want[0])
