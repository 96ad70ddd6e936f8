#!/usr/bin/env python3
"""One-off: the headline shard (8192 pairs of 50 kbp) HBM-resident vs through the host-pointer form with the upload overlapped
(psa_overlap = 1, the default) and not (0).  Prints ms per call: host clock and HIP events."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tsta_amd  # noqa: E402
from bench import make_shard  # noqa: E402

P, L = int(os.environ.get("P", 8192)), int(os.environ.get("L", 50000))
dev = torch.device("cuda", 0)
eng = tsta_amd.Engine(0)
prm = tsta_amd.Params()
seqs, off1, len1, off2, len2 = make_shard(P, L, 1000, dev)
scores = torch.empty(P, dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
with torch.cuda.stream(st):
    for i in range(3):
        t0 = time.perf_counter()
        eng.psa_score_batch_dev(prm, P, seqs.data_ptr(), off1.data_ptr(), len1.data_ptr(), off2.data_ptr(), len2.data_ptr(), scores.data_ptr(), L, st.cuda_stream)
        torch.cuda.synchronize()
        print("resident       wall %8.2f ms  kernels %8.2f ms" % ((time.perf_counter() - t0) * 1e3, eng.last_kernel_ms()[0]), flush=True)
res = scores.cpu().numpy()
h = [x.cpu().numpy() for x in (seqs, off1, len1, off2, len2)]
del seqs
torch.cuda.empty_cache()
for ov in (1, 0, 1):
    eng.set_option("psa_overlap", ov)
    for i in range(3):
        t0 = time.perf_counter()
        got = eng.psa_score_batch_arrays(*h, prm)
        print("host overlap=%d wall %8.2f ms  kernels %8.2f ms  sub-batches %d  %s" % (ov, (time.perf_counter() - t0) * 1e3, eng.last_kernel_ms()[0],
              eng.stat("last_sub_batches"), "ok" if np.array_equal(got, res) else "MISMATCH"), flush=True)
