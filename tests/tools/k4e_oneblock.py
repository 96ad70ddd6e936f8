"""One-off: K4e on the README step-1 chain cut to 1 / 2 / 8 blocks of columns (640 each): us per node with and without a left block."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tsta_amd
from tests import util
eng = tsta_amd.Engine(0)
eng.set_option("poa_ref_arith", 2)
z = np.load(os.path.join(util.GOLDEN_DIR, "poa_readme_step1.npz"))
V = len(z["base"])
for n in (640, 1280, 2560, 5000):
    args = (z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"][:n])
    ms = []
    for rep in range(3):
        eng.poa_fill(*args)
        ms.append(eng.last_kernel_ms()[0])
    print("n %d (%d blocks): %.2f ms = %.2f us per node" % (n, (n + 639) // 640, min(ms), min(ms) * 1e3 / V), flush=True)
