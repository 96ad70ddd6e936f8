"""One-off: the bench's related-pairs shard against the oracle at several sizes (diagnosing zeros in the related line)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tsta_amd, bench
from tests import util

dev = torch.device("cuda", 0)
eng = tsta_amd.Engine(0)
params = tsta_amd.Params()
for P, L in ((1200, 1500), (1200, 5000), (2048, 20000), (8192, 50000)):
    seqs, off1, len1, off2, len2 = bench.make_shard(P, L, 5000, dev, related=5.0)
    sc = torch.full((P,), -7777, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.psa_score_batch_dev(params, P, seqs.data_ptr(), off1.data_ptr(), len1.data_ptr(), off2.data_ptr(), len2.data_ptr(), sc.data_ptr(), L, 0)
    torch.cuda.synchronize()
    ms, nl = eng.last_kernel_ms()
    h = sc.cpu().numpy()
    picks = [0, 1, P // 2, P - 1]
    pairs = [(seqs[int(off1[k]):int(off1[k]) + L].cpu().numpy(), seqs[int(off2[k]):int(off2[k]) + L].cpu().numpy()) for k in picks]
    want = util.oracle_maxscore_batch(pairs)
    print(P, L, "ms", ms, "launches", nl, "got", h[picks].tolist(), "want", want.tolist(), "unwritten", int((h == -7777).sum()), "zeros", int((h == 0).sum()),
          "min/max", int(h.min()), int(h.max()), flush=True)
    del seqs
