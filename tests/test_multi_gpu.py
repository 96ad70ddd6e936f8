"""Several GPUs of one node: the C-ABI shard + gather (tsta_multi_*, multi.hip), the -G flag of TSTA_psa_batch, the
engine under torch.distributed (RCCL) and bench.py's --gpus handling, all against the oracle.  On a one-GPU box the
device lists have one entry (and the RCCL all-gather one rank); with more devices the same tests shard for real."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import tsta_amd
from tests import util

pytestmark = pytest.mark.gpu
BIN = os.path.join(util.ROOT, "tsta_amd", "bin")


def ragged_pairs(n, seed=300):
    pairs = []
    for i in range(n):
        a = util.random_dna(seed + i, 150 + 61 * (i % 13))
        pairs.append((a, util.mutate(a, seed + 500 + i, 0.08) if i % 3 else util.random_dna(seed + 900 + i, 100 + 7 * i % 400)))
    pairs[3] = (pairs[3][0], np.zeros(0, dtype=np.uint8))          # an empty sequence
    return pairs


@pytest.mark.parametrize("gather", [0, 1, 2])
def test_score_batch_multi_matches_oracle(gather):
    pairs = ragged_pairs(90)
    want = util.oracle_maxscore_batch(pairs)
    ndev = tsta_amd.lib().tsta_device_count()
    for devices in ([0], None if ndev > 1 else [0]):
        m = tsta_amd.Multi(devices=devices)                       # None: every device of the node
        try:
            assert m.n_dev == (ndev if devices is None else 1)
            m.set_option("gather", gather)                        # 0: per-device D2H, 1: RCCL all-gather, 2: + replicas compared
            got = m.psa_score_batch(pairs)
            assert np.array_equal(got, want), (devices, gather)
            got = m.psa_score_batch(pairs[:1])                    # fewer pairs than devices: empty shards
            assert np.array_equal(got, want[:1])
        finally:
            m.close()


def test_multi_large_shards_use_the_packed_kernel():
    # enough equal pairs that every device's shard goes down K1p (couples) — the path bench.py times
    ndev = tsta_amd.lib().tsta_device_count()
    pairs = [(util.random_dna(7000 + 2 * i, 1500), util.random_dna(7001 + 2 * i, 1500)) for i in range(1100 * min(ndev, 2))]
    for k in range(0, len(pairs), 97):
        pairs[k] = (pairs[k][0], util.mutate(pairs[k][0], 9000 + k, 0.05))
    want = util.oracle_maxscore_batch(pairs)
    m = tsta_amd.Multi(n_dev=min(ndev, 2))
    try:
        m.set_option("gather", 2)
        assert np.array_equal(m.psa_score_batch(pairs), want)
    finally:
        m.close()


def test_align_batch_multi_matches_single_engine():
    pairs = ragged_pairs(24, seed=4000)
    m = tsta_amd.Multi(n_dev=0)
    e = tsta_amd.Engine(0)
    try:
        got = m.psa_align_batch(pairs)
        assert got == e.psa_align_batch(pairs)
        for (ms, ra, rb), (a, b) in list(zip(got, pairs))[:3]:
            assert (ms, ra, rb) == tuple(util.oracle_align(a, b)[i] for i in (0, 2, 3))
    finally:
        m.close(); e.close()


def test_multi_rejects_bad_arguments():
    L = tsta_amd.lib()
    ndev = L.tsta_device_count()
    with pytest.raises(tsta_amd.TstaError):
        tsta_amd.Multi(devices=[0, 0])                            # duplicate device
    with pytest.raises(tsta_amd.TstaError):
        tsta_amd.Multi(devices=[ndev])                            # no such device
    with pytest.raises(tsta_amd.TstaError):
        tsta_amd.Multi(n_dev=ndev + 1)
    m = tsta_amd.Multi(n_dev=1)
    try:
        with pytest.raises(tsta_amd.TstaError):
            m.set_option("gather", 3)
        assert len(tsta_amd.device_uuid(0)) == 32
    finally:
        m.close()


def test_batch_cli_G(tmp_path):
    tsta_amd.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(util.ROOT, "tsta_amd", "host")])
    pairs = ragged_pairs(30, seed=8000)
    fa, fb = tmp_path / "a.fa", tmp_path / "b.fa"
    fa.write_bytes(b"".join(b">a%d\n" % i + bytes(p[0]) + b"\n" for i, p in enumerate(pairs)))
    fb.write_bytes(b"".join(b">b%d\n" % i + bytes(p[1]) + b"\n" for i, p in enumerate(pairs)))
    want = b"".join(b"maxsorce=%d\n" % w for w in util.oracle_maxscore_batch(pairs))
    exe = os.path.join(BIN, "TSTA_psa_batch")
    for extra in (["-G", "1"], ["-G", "0"], ["-G", "0", "-R"]):
        r = subprocess.run([exe, "-1", str(fa), "-2", str(fb)] + extra, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert r.stdout == want, extra
    out0, out1 = tmp_path / "o0.txt", tmp_path / "o1.txt"
    r0 = subprocess.run([exe, "-1", str(fa), "-2", str(fb), "-o", str(out0)], capture_output=True, timeout=300)
    r1 = subprocess.run([exe, "-1", str(fa), "-2", str(fb), "-G", "0", "-o", str(out1)], capture_output=True, timeout=300)
    assert r0.returncode == 0 and r1.returncode == 0, (r0.stderr, r1.stderr)
    assert out0.read_bytes() == out1.read_bytes() and r0.stdout == r1.stdout == want


_RANK_SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    import numpy as np, torch, torch.distributed as dist
    import tsta_amd
    from tests import util
    from tsta_amd.sharding import gather_scores, shard_bounds
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    lr = int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(lr)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lr))
    n = 37
    pairs = [(util.random_dna(100 + k, 300 + 11 * k), util.mutate(util.random_dna(100 + k, 300 + 11 * k), 7 + k, 0.1)) for k in range(n)]
    lo, hi = shard_bounds(n, world, rank)
    eng = tsta_amd.Engine(lr)
    local = torch.from_numpy(eng.psa_score_batch(pairs[lo:hi])).cuda()
    full = gather_scores(local, n)                      # RCCL all-gather (backend "nccl" is RCCL on ROCm)
    want = util.oracle_maxscore_batch(pairs)
    assert np.array_equal(full.cpu().numpy(), want), (rank, full.cpu().numpy().tolist(), want.tolist())
    dist.barrier(); eng.close(); dist.destroy_process_group()
    print("rank %%d ok" %% rank)
""")


def test_engine_plus_rccl_gather_under_torch_distributed(tmp_path):
    # the product engine (not the oracle) per rank + the score gather bench.py uses, over RCCL: world 1 always, world 2
    # when the box has a second GPU
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % {"root": util.ROOT})
    ndev = tsta_amd.lib().tsta_device_count()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for world in ([1, 2] if ndev > 1 else [1]):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                            "--master-addr", "127.0.0.1", "--master-port", str(29500 + 7 * world), str(script)],
                           capture_output=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        assert r.stdout.decode().count("ok") == world


def test_bench_honours_gpus(tmp_path):
    # `python bench.py --gpus 2` with no launcher: it starts torch.distributed.run as a child itself.  On a one-GPU box the
    # two ranks rehearse on device 0 (gloo for the collectives): n_gpus must read 2, ranks_seen the number of distinct devices.
    ndev = tsta_amd.lib().tsta_device_count()
    env = dict(os.environ)
    if ndev < 2:
        env["TSTA_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--pairs-per-gpu", "1200", "--len", "1500", "--no-cpu-baseline"], capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == (2 if ndev >= 2 else 1)
    assert j["verified_pairs"] >= 3 and j["unit"] == "GCUPS" and j["value"] > 0
    # a mismatch between --gpus and the launcher's world size is refused
    bad = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "1", "--steps", "1"], capture_output=True,
                         timeout=300, env=dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"))
    assert bad.returncode != 0 and b"WORLD_SIZE=2" in bad.stderr
