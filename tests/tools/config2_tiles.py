"""One-off: config #2 (1024 x 5 kbp, K1b) at tile lengths 64 / 128 / 192 / 256 and rows per lane 16 / 8."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tsta_amd
from bench import make_shard
P, L = int(os.environ.get("P", 1024)), int(os.environ.get("L", 5000))
dev = torch.device("cuda", 0)
eng = tsta_amd.Engine(0)
seqs, off1, len1, off2, len2 = make_shard(P, L, 77, dev)
scores = torch.empty(P, dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)
ref = None
for bands in (16, 8):
    for T in ((64, 128, 192, 256, 320) if L < 20000 else (128, 256, 512)):
        eng.set_option("psa_bands", bands); eng.set_option("trace_tile", T)
        ms = []
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            for i in range(6):
                eng.psa_score_batch_dev(tsta_amd.Params(), P, seqs.data_ptr(), off1.data_ptr(), len1.data_ptr(), off2.data_ptr(), len2.data_ptr(), scores.data_ptr(), L, st.cuda_stream)
                eng.psa_score_batch_dev_status(st.cuda_stream)
                if i: ms.append(eng.last_kernel_ms()[0])
        sc = scores.cpu().numpy()
        if ref is None: ref = sc.copy()
        print("R %2d T %3d: %.3f ms  %.1f GCUPS  %s" % (bands, T, np.median(ms), P * L * L / np.median(ms) / 1e6, "same" if np.array_equal(sc, ref) else "DIFFERENT"), flush=True)
