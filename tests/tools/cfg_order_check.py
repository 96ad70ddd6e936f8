#!/usr/bin/env python3
"""One-off: POA fill / align on an engine whose workspaces were grown by the big PSA configs first."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tsta_amd, bench_configs
from tests import util
eng = tsta_amd.Engine(0)
for w in sys.argv[1:]:
    t = time.time()
    if w == "3": bench_configs.config3(eng)
    elif w == "5t": bench_configs.config5t(eng, 64)
    elif w == "fill":
        z = np.load(os.path.join(util.GOLDEN_DIR, "poa_readme_step2.npz"))
        eng.poa_fill(z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"])
    elif w == "align":
        z = np.load(os.path.join(util.GOLDEN_DIR, "poa_readme_step2.npz"))
        eng.poa_align(z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"])
    print(w, "ok %.2f s" % (time.time() - t), flush=True)
