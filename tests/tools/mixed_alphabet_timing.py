#!/usr/bin/env python3
"""One-off: the headline shard with 1 % 'N' and soft-masked (lowercase) runs — any byte that is not ACGT sends a couple down the
generic compare of the packed kernel — against the pure-ACGT shard.  Scores of sampled pairs are checked against the oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tsta_amd  # noqa: E402
from bench import make_shard, make_mixed_alphabet  # noqa: E402
from tests import util  # noqa: E402

P, L = int(os.environ.get("P", 8192)), int(os.environ.get("L", 50000))
dev = torch.device("cuda", 0)
eng = tsta_amd.Engine(0)
prm = tsta_amd.Params()
seqs, off1, len1, off2, len2 = make_shard(P, L, 1000, dev)
scores = torch.empty(P, dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)


def run(tag):
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for i in range(2):
            eng.psa_score_batch_dev(prm, P, seqs.data_ptr(), off1.data_ptr(), len1.data_ptr(), off2.data_ptr(), len2.data_ptr(), scores.data_ptr(), L, st.cuda_stream)
            torch.cuda.synchronize()
    ms = eng.last_kernel_ms()[0]
    print("%-34s %8.1f ms  %7.1f GCUPS" % (tag, ms, float(P) * L * L / ms / 1e6), flush=True)


run("pure ACGT")
make_mixed_alphabet(seqs, 4242)
run("1 % N + 10 % lowercase runs")
sc = scores.cpu().numpy()
for k in (0, P // 2, P - 1):
    a = seqs[int(off1[k]):int(off1[k]) + L].cpu().numpy(); b = seqs[int(off2[k]):int(off2[k]) + L].cpu().numpy()
    assert util.oracle_maxscore(a, b) == int(sc[k]), k
print("sampled scores = oracle")
for key in ("psa_elide", "psa_perm"):
    eng.set_option(key, 0)
run("same, psa_elide=0 psa_perm=0")
