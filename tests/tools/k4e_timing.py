"""One-off (GPU box): what a fill costs in the reference's own arithmetic (K4e, poa_ref8.hip) next to the plain kernels, on the
README example's graphs (tests/golden/poa_readme_step*.npz) at -W 10: kernel ms per fill, us per node."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tsta_amd
from tests import util
eng = tsta_amd.Engine(0)
for step in (1, 2, 4):
    z = np.load(os.path.join(util.GOLDEN_DIR, "poa_readme_step%d.npz" % step))
    args = (z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"])
    V = len(z["base"])
    res = {}
    for mode in (1, 2):
        eng.set_option("poa_ref_arith", mode)
        ms = []
        for rep in range(3):
            got = eng.poa_fill(*args)
            ms.append(eng.last_kernel_ms()[0])
        res[mode] = (min(ms), got)
    same = all(np.array_equal(a, b) for a, b in zip(res[1][1][1:], res[2][1][1:]))
    print("step %d: V %d n %d | plain %.2f ms | K4e %.2f ms = %.2f us per node | same arrays: %s" %
          (step, V, len(z["seq"]), res[1][0], res[2][0], res[2][0] * 1e3 / V, same), flush=True)
