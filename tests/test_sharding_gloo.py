"""N>1 path on CPU: world_size-2 (and 3, ragged) gloo runs of the shard + score-gather logic bench.py uses.
The per-shard compute here is the oracle (no GPU in this container); on GPUs the same code runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import util
from tsta_amd.sharding import gather_scores, shard_bounds


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(n_pairs, world, rank)
        pairs = [(util.random_dna(100 + k, 60 + k), util.mutate(util.random_dna(100 + k, 60 + k), 7 + k, 0.1))
                 for k in range(lo, hi)]
        local = torch.from_numpy(util.oracle_maxscore_batch(pairs, threads=1)) if pairs else torch.zeros(0, dtype=torch.int32)
        full = gather_scores(local, n_pairs)
        dist.barrier()
        if rank == 0:
            q.put(full.numpy().tolist())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_pairs", [(2, 16), (2, 7), (3, 10)])
def test_sharded_scores_match_single_process(world, n_pairs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pairs = [(util.random_dna(100 + k, 60 + k), util.mutate(util.random_dna(100 + k, 60 + k), 7 + k, 0.1))
             for k in range(n_pairs)]
    want = util.oracle_maxscore_batch(pairs, threads=2).tolist()
    assert got == want


def test_shard_bounds_cover_batch():
    for n in (0, 1, 7, 8192, 65536):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
