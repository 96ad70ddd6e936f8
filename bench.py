#!/usr/bin/env python3
"""bench.py — GCUPS of batched 50 kbp x 50 kbp global affine-gap alignment (score only) on N MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by torch.distributed.run, one rank
per GPU — or by itself: run as plain `python bench.py --gpus N` (no RANK in the environment) it starts that launcher as
a CHILD process before touching the GPU and relays rank 0's JSON line.  Every rank asserts WORLD_SIZE == --gpus, and the
line carries `ranks_seen` (distinct device UUIDs, all-gathered).  A "step" = one pass of the hot path (tsta_psa_score_batch_dev through the C-ABI) over this rank's shard
of the batch, inputs already resident in HBM, followed (N>1) by the RCCL all-gather of the int32 scores — the only
collective on the path.  Workload = BASELINE.json config #5 per-GPU shard (independent pairs, weak scaling):
PAIRS_PER_GPU pairs of LEN x LEN random DNA per GPU, default scoring (M=2 X=-5 E=-2 O=-4).

Every rank re-scores --verify-pairs pairs of ITS shard (drawn from a per-run seed, printed) with the reference binary and the
mismatch count is all-reduced; the line carries per-rank ms min/max.  At N = 1 the line also carries `configs`: BASELINE
configs #2, #3, #4 and the 64 x 50 kbp traceback batch, each with kernel ms (HIP events), wall ms, GCUPS, its SURVEY 8(d)
roofline, a `verified` flag (reference data) and the reference's CPU path beside it (bench_configs.py).

Rank 0 prints ONE JSON line with `roofline` (HBM, algorithmic bytes; plus the VALU roofline that
actually bounds this integer recurrence) and `cpu_baseline` (the reference's own AVX512+pthreadpool binary from
oracle/_ref timed on this host's cores on a bounded sample; falls back to the scalar oracle port).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import tsta_amd  # noqa: E402
from tsta_amd.sharding import gather_scores  # noqa: E402

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD-32 x 2.4 GHz = 7.86e13 int32 lane-ops/s
# K1p steady-state loop (R=32, pure-ACGT couple, block proven unable to raise the maximum): 532 VALU instructions per
# 2 steps of 64 cells = 8 per cell pair (f-sub, v_perm lookup, diag-sub, 4 v_pk_max, g-sub) + 10 per step for the lane
# hand-off.  The generic loop (any bytes, running max kept) is 11.3 per cell pair = 5.66 per cell.
VALU_OPS_PER_CELL = 4.16              # loop-body count (disassembly); superseded by the SQ_INSTS_VALU measurement in profiles/valu_psa_score.json
VALU_OPS_PER_CELL_GENERIC = 5.66
# what a SIMD sustains on this kernel's instruction mix (v_pk_max_u16 / v_perm_b32 / DPP at 4 cycles, v_sub_u32 at ~2.3,
# alternating: 3.88): cycles per wave64 VALU instruction per SIMD at >= 2 waves per SIMD, tools/cell_microbench.hip,
# profiles/r02a_cell_microbench.txt ("alternating sub / pk_max", "4 rows, clustered")
MIX_CYCLES_PER_INSTR = 3.88


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs-per-gpu", type=int, default=int(os.environ.get("TSTA_BENCH_PAIRS", 8192)))
    ap.add_argument("--len", type=int, default=50_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-generic", action="store_true", help="skip the extra pass down the generic kernel path")
    ap.add_argument("--no-related", action="store_true", help="skip the extra pass over related (5 %% divergent) pairs")
    ap.add_argument("--no-verify", action="store_true", help="skip the check of sampled scores against the reference / oracle")
    ap.add_argument("--cpu-sample-pairs", type=int, default=0, help="0 = auto (about 10-30 s of CPU work)")
    ap.add_argument("--verify-pairs", type=int, default=32, help="pairs of EVERY rank's shard re-scored by the reference (drawn from a per-run seed)")
    ap.add_argument("--verify-seed", type=int, default=0, help="0 = from the clock (printed in the JSON line)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (host-resident shard through tsta_psa_score_batch)")
    ap.add_argument("--e2e-steps", type=int, default=2)
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs #2 / #3 / #4 / 5t (N = 1 only; bench_configs.py)")
    return ap.parse_args()


def make_shard(n_pairs, length, seed, device, related=None):
    """Synthetic iid uniform ACGT, generated on the GPU (inputs are HBM-resident before the timed region).
    TSTA_BENCH_RELATED=<percent> (experiments only, not the BASELINE workload): the second sequence of every pair is the
    first with that percentage of substituted positions instead of an independent draw."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    idx = torch.randint(0, 4, (2 * n_pairs * length,), generator=g, device=device, dtype=torch.uint8)
    if related is None:
        related = float(os.environ.get("TSTA_BENCH_RELATED", 0))
    if related > 0:
        v = idx.view(n_pairs, 2, length)
        for lo in range(0, n_pairs, 256):
            a = v[lo:lo + 256, 0]
            hit = torch.rand(a.shape, generator=g, device=device) < related / 100.0
            sub = torch.randint(1, 4, a.shape, generator=g, device=device, dtype=torch.uint8)
            v[lo:lo + 256, 1] = torch.where(hit, (a + sub) % 4, a)
    alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    seqs = alpha[idx.long()] if idx.numel() < (1 << 28) else torch.cat(
        [alpha[c.long()] for c in idx.split(1 << 28)])
    del idx
    k = torch.arange(n_pairs, device=device, dtype=torch.int64)
    off1 = 2 * k * length
    off2 = off1 + length
    ln = torch.full((n_pairs,), length, dtype=torch.int32, device=device)
    return seqs, off1, ln, off2, ln.clone()


def make_mixed_alphabet(seqs, seed):
    """In place: 1 % of the bases become 'N' and ~10 % are soft-masked (lowercase) in runs of 1 000 — what assemblies and
    repeat-masked references look like.  PSA compares bytes (psa.c:248-259): 'N' and lowercase are ordinary characters."""
    g = torch.Generator(device=seqs.device)
    g.manual_seed(seed)
    n = seqs.numel()
    for lo in range(0, n, 1 << 28):
        v = seqs[lo:lo + (1 << 28)]
        runs = (torch.rand((v.numel() + 999) // 1000, generator=g, device=seqs.device) < 0.10).repeat_interleave(1000)[:v.numel()]
        v[runs] |= 0x20                                             # ASCII lowercase
        v[torch.rand(v.numel(), generator=g, device=seqs.device) < 0.01] = ord("N")
        del runs


T_START = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries only the JSON line)"""
    sys.stderr.write("[bench %6.1f s] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


def self_launch(args):
    """`python bench.py --gpus N` (N>1) without a launcher: start torch.distributed.run as a CHILD process — before this
    process has made any GPU call; a process that has initialised the GPU must never exec another — let it print rank 0's
    JSON line on our stdout, and exit with its code."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def run_reference(binary, threads, fa, fb, limit=30.0, tries=4):
    """One TSTA_psa_notrace process -> its stdout.  The reference's thread pool can deadlock (no join, unlocked shutdown
    flag: pthreadpool.c:96-117; seen here under oversubscription), so every run has a time limit and is retried."""
    last = None
    for _ in range(tries):
        try:
            r = subprocess.run([binary, "-T", str(threads), "-1", fa, "-2", fb], capture_output=True, check=True, timeout=limit)
            if b"maxsorce=" in r.stdout:
                return r.stdout
            last = RuntimeError("no maxsorce in output")
        except (subprocess.TimeoutExpired, subprocess.CalledProcessError) as ex:
            last = ex
            log("reference run failed or hung (%r); retrying" % (ex,))
    raise last


def reference_binary():
    """oracle/_ref/TSTA_psa_notrace (the unmodified reference compiled by oracle/Makefile) if this host can run it."""
    from tests import util
    for name, flag in (("TSTA_psa_notrace", "avx512bw"), ("TSTA_psa_notrace_avx2", "avx2")):
        b = os.path.join(util.REF_DIR, name)
        if os.path.exists(b) and util.cpu_has(flag):
            return b
    return None


def reference_scores(pairs, cores):
    """maxsorce (psa.c:666) of every (a, b) pair of uint8 arrays as the REFERENCE computes it: its own binary, one process
    per pair (several at a time: one pair's block anti-diagonals cannot feed more than ~32 threads); where that binary
    cannot run, the scalar oracle port.  -> (scores, "reference" | "port")"""
    from concurrent.futures import ThreadPoolExecutor
    from tests import util
    binary = reference_binary()
    if binary:
        try:
            longest = max([max(len(a), len(b)) for a, b in pairs] + [1])
            t_each = max(1, min(cores, 32 if longest >= 20000 else 4))
            workers = max(1, cores // t_each)
            with tempfile.TemporaryDirectory() as td:
                def one(i):
                    a, b = pairs[i]
                    fa, fb = os.path.join(td, "a%d.fa" % i), os.path.join(td, "b%d.fa" % i)
                    util.write_fasta(fa, "a", a); util.write_fasta(fb, "b", b)
                    so = run_reference(binary, t_each, fa, fb, limit=120.0)
                    return int(so.decode().split("maxsorce=")[1].split()[0])
                with ThreadPoolExecutor(max_workers=workers) as ex:
                    got = list(ex.map(one, range(len(pairs))))
            return got, "reference"
        except Exception as ex:
            sys.stderr.write("reference_scores: reference binary failed (%r); using the oracle port\n" % (ex,))
    return [int(x) for x in util.oracle_maxscore_batch(pairs, threads=cores)], "port"


def verify_against_reference(seqs, off1, len1, off2, len2, scores, picks, cores):
    """D2H the sequences of the picked pairs of the shard that was just timed and require the reference's maxsorce for
    each of them to equal the kernel's."""
    pairs = []
    for k in picks:
        a = seqs[int(off1[k]):int(off1[k]) + int(len1[k])].cpu().numpy()
        b = seqs[int(off2[k]):int(off2[k]) + int(len2[k])].cpu().numpy()
        pairs.append((a, b))
    log("verifying pairs %s against the reference" % (picks,))
    want, kind = reference_scores(pairs, cores)
    got = [int(scores[k]) for k in picks]
    assert got == want, "kernel scores differ from the %s on pairs %s: %s vs %s" % (kind, picks, got, want)
    return {"verified_pairs": len(picks), "verified_by": kind, "pairs": [int(k) for k in picks], "scores": got}


def cpu_throughput(length, cores, seed=777):
    """The reference used for THROUGHPUT rather than latency: k concurrent TSTA_psa_notrace processes with -T cores/k
    each (one 79-block anti-diagonal cannot feed hundreds of threads).  Best of k in {cores/32, cores/8} (>=1)."""
    from tests import util
    binary = reference_binary()
    if not binary:
        return None
    best = None
    with tempfile.TemporaryDirectory() as td:
        n_files = 8
        files = []
        for i in range(n_files):
            fa, fb = os.path.join(td, "a%d.fa" % i), os.path.join(td, "b%d.fa" % i)
            util.write_fasta(fa, "a", util.random_dna(seed + 2 * i, length)); util.write_fasta(fb, "b", util.random_dna(seed + 2 * i + 1, length))
            files.append((fa, fb))
        for k in sorted({max(1, cores // 32), max(1, cores // 8), max(1, cores // 2)}):
            t_each = max(1, cores // k)
            log("cpu_throughput: %d processes x -T %d" % (k, t_each))
            # calibrate on one lone run, then every worker (a shell loop: no Python between its runs) aligns `reps` pairs
            # back to back, about 8 s of work each
            c0 = time.perf_counter()
            try:
                run_reference(binary, t_each, files[0][0], files[0][1], limit=60.0)
            except Exception:
                continue
            one = max(1e-3, time.perf_counter() - c0)
            limit = max(5.0, 20.0 * one)            # a run that takes this long has deadlocked: it is killed and not counted
            span = 8                                # every worker (a shell loop: no Python between its runs) aligns pairs back to back for ~8 s
            procs = []
            t0 = time.perf_counter()
            for w in range(k):
                fa, fb = files[w % n_files]
                sh = ("end=$(( $(date +%%s) + %d )); n=0; t=0; while [ $(date +%%s) -lt $end ]; do t=$((t+1)); timeout -k 2 %.0f %s -T %d -1 %s -2 %s > /dev/null && n=$((n+1)); done; echo $n $t"
                      % (span, limit, binary, t_each, fa, fb))
                procs.append(subprocess.Popen(["sh", "-c", sh], stdout=subprocess.PIPE))
            done = tried = 0
            for pr in procs:
                so, _ = pr.communicate()
                try:
                    a, b = so.decode().split()
                    done += int(a); tried += int(b)
                except ValueError:
                    pass
            dt = time.perf_counter() - t0
            if done == 0:
                continue
            v = float(done) * length * length / dt / 1e9
            if best is None or v > best["value"]:
                best = {"value": v, "unit": "GCUPS", "cores": cores, "kind": "reference",
                        "sample": "%d concurrent oracle/_ref/%s -T %d workers aligning pairs of %dx%d back to back for %.1f s; %d of %d runs "
                                  "completed (the reference's thread pool can deadlock: such runs are killed at %.0f s)"
                                  % (k, os.path.basename(binary), t_each, length, length, dt, done, tried, limit)}
    return best


def cpu_baseline(length, sample_pairs, seed=12345):
    """TSTA's own CPU path on this host: oracle/_ref/TSTA_psa_notrace (AVX512 + pthreadpool, -T = all cores),
    one process per pair as the CLI is (psa.c:539-668).  Bounded sample of the same workload."""
    from tests import util
    try:
        cores = len(os.sched_getaffinity(0))          # the cores this process may actually use
    except AttributeError:
        cores = os.cpu_count() or 1
    binary = os.path.join(util.REF_DIR, "TSTA_psa_notrace")
    use_ref = os.path.exists(binary) and util.cpu_has("avx512bw")
    if not use_ref:
        alt = os.path.join(util.REF_DIR, "TSTA_psa_notrace_avx2")
        if os.path.exists(alt) and util.cpu_has("avx2"):
            binary, use_ref = alt, True
    pairs = [(util.random_dna(seed + 2 * i, length), util.random_dna(seed + 2 * i + 1, length))
             for i in range(max(1, sample_pairs))]
    cells = float(len(pairs)) * length * length
    if use_ref:
        try:
            with tempfile.TemporaryDirectory() as td:
                files = []
                for i, (a, b) in enumerate(pairs):
                    fa, fb = os.path.join(td, "a%d.fa" % i), os.path.join(td, "b%d.fa" % i)
                    util.write_fasta(fa, "a", a); util.write_fasta(fb, "b", b)
                    files.append((fa, fb))
                run_reference(binary, cores, files[0][0], files[0][1])     # warm-up (page in the binary)
                # repeat the sample until about TARGET_S seconds of CPU work have been timed (bounded)
                t0 = time.perf_counter()
                done = 0
                while True:
                    for fa, fb in files:
                        run_reference(binary, cores, fa, fb)
                    done += len(files)
                    dt = time.perf_counter() - t0
                    if dt >= 12.0 or done >= 4096:
                        break
                cells = float(done) * length * length
                pairs = pairs * (done // len(files))
            return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "reference",
                    "sample": "%d pairs of %dx%d random DNA, oracle/_ref/%s -T %d, one process per pair, %.1f s"
                              % (len(pairs), length, length, os.path.basename(binary), cores, dt)}
        except Exception as ex:      # binary cannot run on this host: fall through to the port
            sys.stderr.write("cpu_baseline: reference binary failed (%r); using the oracle port\n" % (ex,))
    pairs = pairs[:max(1, min(len(pairs), cores))]
    cells = float(len(pairs)) * length * length
    t0 = time.perf_counter()
    util.oracle_maxscore_batch(pairs, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "port",
            "sample": "%d pairs of %dx%d random DNA, scalar oracle, OpenMP over pairs, %.1f s" % (len(pairs), length, length, dt)}


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or run `python bench.py --gpus N` directly)" % (args.gpus, world)
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    # rehearsal of the N>1 control flow on a box with ONE GPU (never used by the driver): every rank computes on device 0
    # and the collectives run over gloo on host copies.  Numbers from such a run mean nothing.
    rehearsal = os.environ.get("TSTA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    P, L = args.pairs_per_gpu, args.len
    log("rank %d/%d: generating %d pairs of %d x %d" % (rank, world, P, L, L))
    eng = tsta_amd.Engine(local_rank)
    params = tsta_amd.Params()
    seqs, off1, len1, off2, len2 = make_shard(P, L, 1000 + rank, dev)
    scores = torch.empty(P, dtype=torch.int32, device=dev)
    gathered = [None]
    tstream = torch.cuda.Stream(device=dev)          # explicit stream: kernel, events and RCCL all ride on it
    stream = tstream.cuda_stream
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        eng.psa_score_batch_dev(params, P, seqs.data_ptr(), off1.data_ptr(), len1.data_ptr(), off2.data_ptr(),
                                len2.data_ptr(), scores.data_ptr(), L, stream)
        if i is not None:
            ev[i][1].record()
        if world > 1:
            # the path's only collective: final score gather (RCCL over xGMI; 4 B per pair)
            gathered[0] = gather_scores(scores.cpu() if rehearsal else scores, P * world)

    torch.cuda.synchronize()
    log("warmup + %d timed steps" % args.steps)
    with torch.cuda.stream(tstream):
        for _ in range(args.warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    my_kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps > 0 else float("nan")
    rank_ms = [[dt / max(1, args.steps) * 1e3, my_kern_ms]]
    if world > 1:
        t = torch.tensor(rank_ms[0], dtype=torch.float64, device="cpu" if rehearsal else dev)
        allt = torch.empty(2 * world, dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(allt, t)
        rank_ms = allt.cpu().view(world, 2).tolist()
        dt = max(r[0] for r in rank_ms) * 1e-3 * max(1, args.steps)          # MAX over ranks
    # which devices took part: all-gather of the 16-byte device UUIDs (a rehearsal on one GPU shows ONE distinct UUID)
    uuid = tsta_amd.device_uuid(local_rank)
    uuids = [uuid]
    if world > 1:
        u = torch.tensor(list(bytes.fromhex(uuid)), dtype=torch.uint8, device="cpu" if rehearsal else dev)
        allu = torch.empty(16 * world, dtype=torch.uint8, device=u.device)
        dist.all_gather_into_tensor(allu, u)
        uuids = [bytes(allu[16 * r:16 * r + 16].cpu().tolist()).hex() for r in range(world)]
        assert gathered[0] is not None and gathered[0].numel() == P * world
        assert torch.equal(gathered[0][rank * P:(rank + 1) * P].to(scores.device), scores), "gathered scores differ from this rank's shard"

    # ---- the metric as SURVEY 8(d) defines it: wall = first H2D enqueue -> last result visible on the host.  The SAME shard, now
    # HOST-resident (pageable memory, as a caller of the C-ABI would hold it), through the host-pointer form tsta_psa_score_batch:
    # the engine cuts it into sub-batches of whole rounds of the chip and uploads sub-batch j + 1 on a second stream while j
    # computes (psa_api.hip), scores come back to the host.  Both clocks: the host's around the call, HIP events from the first
    # kernel to the last.  Every rank does it (same staging at any N); the value is whole-job cells over the slowest rank's wall.
    # Not the headline `value` (the contract asks for HBM-resident inputs there).
    e2e = None
    if not args.no_e2e:
        log("end to end: host-resident shard -> scores on the host")
        h_seqs = seqs.cpu().numpy()
        h_off1, h_len1, h_off2, h_len2 = (x.cpu().numpy() for x in (off1, len1, off2, len2))
        resident = scores.cpu().numpy().copy()
        walls, kms = [], []
        for i in range(1 + max(1, args.e2e_steps)):                  # the first call allocates the engine's staging buffers
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            got = eng.psa_score_batch_arrays(h_seqs, h_off1, h_len1, h_off2, h_len2, params)
            w = (time.perf_counter() - t1) * 1e3
            if i:
                walls.append(w); kms.append(eng.last_kernel_ms()[0])
        assert np.array_equal(got, resident), "end-to-end scores differ from the HBM-resident run's"
        n_sub = eng.stat("last_sub_batches")
        my = [float(np.mean(walls)), float(np.mean(kms))]
        allw = [my]
        if world > 1:
            t = torch.tensor(my, dtype=torch.float64, device="cpu" if rehearsal else dev)
            allt = torch.empty(2 * world, dtype=torch.float64, device=t.device)
            dist.all_gather_into_tensor(allt, t)
            allw = allt.cpu().view(world, 2).tolist()
        # twice the shard in one call (two balanced sub-batches): the second one's upload rides behind the first one's kernels
        h2 = np.concatenate([h_seqs, h_seqs]); o1 = np.concatenate([h_off1, h_off1 + h_seqs.nbytes]); o2 = np.concatenate([h_off2, h_off2 + h_seqs.nbytes])
        l1, l2 = np.concatenate([h_len1, h_len1]), np.concatenate([h_len2, h_len2])
        t1 = time.perf_counter()
        got2 = eng.psa_score_batch_arrays(h2, o1, l1, o2, l2, params)
        wall2 = (time.perf_counter() - t1) * 1e3
        assert np.array_equal(got2[:P], resident) and np.array_equal(got2[P:], resident)
        two = {"pairs_per_rank": 2 * P, "wall_ms": wall2, "kernel_ms": eng.last_kernel_ms()[0], "sub_batches": eng.stat("last_sub_batches"),
               "value_this_rank": 2.0 * P * L * L / (wall2 * 1e-3) / 1e9}
        del h2
        wall_ms = max(r[0] for r in allw)
        e2e = {"value": float(P) * world * L * L / (wall_ms * 1e-3) / 1e9, "unit": "GCUPS", "wall_ms": wall_ms,
               "kernel_ms": max(r[1] for r in allw), "steps": len(walls), "sub_batches": n_sub,
               "host_bytes_per_rank": int(h_seqs.nbytes), "scores_equal_resident_run": True, "two_shards_in_one_call": two,
               "note": "SURVEY 8(d) metric: host clock around tsta_psa_score_batch (pageable host shard in, scores on the host out; max "
                       "over ranks), kernel_ms = HIP events from the first sub-batch's first kernel to the last one's end; upload of "
                       "sub-batch j + 1 overlaps the kernels of j on a second stream, only the first sub-batch's upload is exposed"}
        del h_seqs

    # the same shard once more down the generic path (any byte alphabet, running max kept in every block): what the
    # kernel does when neither of its two data-dependent shortcuts applies.  Results must be identical.
    generic_ms = None
    log("timed region done: %.1f ms per step" % (dt / max(1, args.steps) * 1e3))
    if world == 1 and not args.no_generic:
        first = scores.clone()
        eng.set_option("psa_elide", 0); eng.set_option("psa_perm", 0)
        with torch.cuda.stream(tstream):
            step()
            torch.cuda.synchronize()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record(); step(); g1.record()
            torch.cuda.synchronize()
        generic_ms = g0.elapsed_time(g1)
        eng.set_option("psa_elide", 1); eng.set_option("psa_perm", 1)
        assert torch.equal(first, scores), "generic and shortcut paths disagree"

    # sanity on the result of the last step: random DNA gives a degenerate maxsorce in a tiny range
    sc = scores.cpu().numpy()
    assert sc.min() >= -100 and sc.max() < 2 * L, (sc.min(), sc.max())
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1

    # the scores of the shard that was just timed, tied to the reference (psa.c:666): EVERY rank sends --verify-pairs pairs of
    # ITS shard — drawn from a per-run seed that goes into the JSON line — back to the host and through
    # oracle/_ref/TSTA_psa_notrace (the scalar oracle port where that binary cannot run), the host cores split across the
    # ranks; the mismatch counts are all-reduced and rank 0 requires the sum to be 0: an N-rank run proves N shards.
    verified = None
    vseed = args.verify_seed or (int(time.time()) & 0x7fffffff)
    if world > 1:
        vs = torch.tensor([vseed], dtype=torch.int64, device="cpu" if rehearsal else dev)
        dist.broadcast(vs, src=0)
        vseed = int(vs.item())
    if not args.no_verify:
        picks = sorted(int(x) for x in np.random.default_rng([vseed, rank]).choice(P, min(P, max(1, args.verify_pairs)), replace=False))
        my_cores = max(1, cores // world)
        bad, info = 0, None
        try:
            info = verify_against_reference(seqs, off1.cpu().numpy(), len1.cpu().numpy(), off2.cpu().numpy(), len2.cpu().numpy(), sc, picks, my_cores)
        except AssertionError as ex:
            bad = 1
            log("rank %d: VERIFICATION FAILED: %s" % (rank, ex))
        checked = len(picks)
        if world > 1:
            t = torch.tensor([bad, checked], dtype=torch.int64, device="cpu" if rehearsal else dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            bad, checked = int(t[0].item()), int(t[1].item())
        assert bad == 0, "%d rank(s) computed scores that differ from the reference" % bad
        verified = {"verified_pairs": checked, "verified_ranks": world, "verify_seed": vseed,
                    "verified_by": info["verified_by"] if info else None, "pairs": info["pairs"] if info else None,
                    "scores": info["scores"] if info else None,
                    "verify_note": "every rank re-scored %d pairs of its own shard (numpy default_rng([verify_seed, rank]).choice) with the "
                                   "reference binary on %d of the host's %d cores; mismatching ranks all-reduced: 0" % (len(picks), my_cores, cores)}

    # the shard with a realistic non-ACGT content: 1 % 'N', 10 % soft-masked (lowercase) runs.  Any such byte sends a couple down
    # the generic byte compare (xor + min instead of the byte-permute lookup); the running-max elision is data-, not alphabet-
    # dependent and still applies.  In place (the pure shard is not needed any more), sampled pairs against the reference.
    mixed = None
    if world == 1 and not args.no_generic:
        log("mixed-alphabet shard (1 % N, 10 % lowercase)")
        make_mixed_alphabet(seqs, 4242 + rank)
        torch.cuda.synchronize()
        with torch.cuda.stream(tstream):
            step()
            torch.cuda.synchronize()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record(); step(); g1.record()
            torch.cuda.synchronize()
        mixed = {"value": float(P) * L * L / (g0.elapsed_time(g1) * 1e-3) / 1e9, "unit": "GCUPS", "ms": g0.elapsed_time(g1),
                 "note": "same kernel and options as the headline on the shard with 1 % 'N' and 10 % lowercase runs (bytes compared "
                         "case-sensitively, psa.c:248-259): generic compare, elision still applies"}
        if not args.no_verify:
            msc = scores.cpu().numpy()
            mixed.update(verify_against_reference(seqs, off1.cpu().numpy(), len1.cpu().numpy(), off2.cpu().numpy(), len2.cpu().numpy(), msc,
                                                  sorted(int(x) for x in np.random.default_rng([vseed, 98]).choice(P, min(P, 8), replace=False)), cores))

    # a shard of RELATED pairs (second sequence = first with 5 % substitutions): real workloads are not iid DNA; the
    # running-max elision then applies to every band except those the high-scoring diagonal crosses
    related = None
    if world == 1 and not args.no_related:
        log("related-pairs shard")
        del seqs
        torch.cuda.empty_cache()
        rseqs, roff1, rlen1, roff2, rlen2 = make_shard(P, L, 5000 + rank, dev, related=5.0)
        rscores = torch.full((P,), -2 ** 31, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()                    # the shard is generated on torch's default stream, the kernel runs on tstream
        with torch.cuda.stream(tstream):
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(2):
                if i == 1:
                    r0.record()
                eng.psa_score_batch_dev(params, P, rseqs.data_ptr(), roff1.data_ptr(), rlen1.data_ptr(), roff2.data_ptr(),
                                        rlen2.data_ptr(), rscores.data_ptr(), L, stream)
            r1.record()
            torch.cuda.synchronize()
        related = {"value": float(P) * L * L / (r0.elapsed_time(r1) * 1e-3) / 1e9, "unit": "GCUPS", "ms": r0.elapsed_time(r1),
                   "note": "same kernel, %d pairs of %d x %d where the second sequence is the first with 5 %% substitutions" % (P, L, L)}
        if not args.no_verify:
            rsc = rscores.cpu().numpy()
            related.update(verify_against_reference(rseqs, roff1.cpu().numpy(), rlen1.cpu().numpy(), roff2.cpu().numpy(),
                                                    rlen2.cpu().numpy(), rsc,
                                                    sorted(int(x) for x in np.random.default_rng([vseed, 99]).choice(P, min(P, 8), replace=False)), cores))
        del rseqs

    if rank == 0:
        cells_per_step = float(P) * world * L * L
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps > 0 else float("nan")
        alg_bytes = float(P) * (2 * L + 4)                      # (len_a+len_b) B in + 4 B out per pair
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic_psa_score.json")
        if os.path.exists(tp):
            with open(tp) as f:
                tj = json.load(f)
            if tj.get("pairs_per_gpu") == P and tj.get("len") == L:
                traffic = tj.get("hbm_bytes_per_launch")
        # VALU roofline: instructions per cell MEASURED with SQ_INSTS_VALU on this workload (profiles/valu_psa_score.json, made
        # by tools/collect_profiles.py from a rocprofv3 --pmc run of this script) when it matches, else the loop-body count
        ops, ops_src, clk = VALU_OPS_PER_CELL, "loop body (disassembly)", 2.38e9
        vp = os.path.join(ROOT, "profiles", "valu_psa_score.json")
        if os.path.exists(vp):
            with open(vp) as f:
                vj = json.load(f)
            if vj.get("pairs_per_gpu") == P and vj.get("len") == L:
                ops, ops_src = vj["ops_per_cell"], "SQ_INSTS_VALU x 64 / cells (profiles/valu_psa_score.json)"
                clk = (vj.get("shader_clock_GHz_from_GRBM_GUI_ACTIVE") or 2.38) * 1e9
        lane_ops = float(P) * L * L * ops / (kern_ms * 1e-3)
        n_simd = 256 * 4
        cyc_per_instr = kern_ms * 1e-3 * clk / (float(P) * L * L * ops / 64.0 / n_simd)
        valu = {"ops_per_cell": ops, "ops_per_cell_source": ops_src, "achieved": lane_ops / 1e12, "peak": VALU_PEAK_LANEOPS / 1e12,
                "unit": "T lane-ops/s", "frac": lane_ops / VALU_PEAK_LANEOPS,
                "cycles_per_valu_instr_per_simd": cyc_per_instr, "mix_floor_cycles_per_instr": MIX_CYCLES_PER_INSTR,
                "frac_of_measured_ceiling": MIX_CYCLES_PER_INSTR / cyc_per_instr, "shader_clock_hz": clk,
                "note": "peak = data-sheet rate (one wave64 instruction per 2 cycles per SIMD), reached only by pure add/sub/logic "
                        "streams; this kernel's mix sustains one per %.2f cycles in isolation (profiles/r02a_cell_microbench.txt), "
                        "and SQ counters show every VALU instruction of it accounted one quad-cycle "
                        "(SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU, profiles/valu_psa_score.json)" % MIX_CYCLES_PER_INSTR}
        out = {
            "metric": "GCUPS (DP cells/s) for batched 50kbp global affine-gap alignment",
            "value": cells_per_step * args.steps / dt / 1e9,
            "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / max(1, args.steps) * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16 packed x2 (biased, int32 per-pair base)", "data": "synthetic",
            "config": {"workload": "BASELINE config #5 per-GPU shard: %d pairs/GPU of %d x %d random DNA, affine gap "
                                   "M=2 X=-5 E=-2 O=-4, score only (maxsorce), inputs HBM-resident" % (P, L, L),
                       "kernel_path": "psa_score_packed_kernel<32,2>: pure-ACGT couples use the v_perm mismatch lookup; blocks whose "
                                      "entering scores bound them below the pair's current maximum skip the running max "
                                      "(every cell is still computed; exact for any input)",
                       "pairs_per_gpu": P, "len": L, "parallelism": "pairs sharded x%d, RCCL all-gather of scores" % world},
            "roofline": {"bound": "hbm", "achieved": alg_bytes / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "psa_score_packed_kernel", "kernel_ms": kern_ms,
                         "note": "integer max-plus recurrence: VALU-issue-bound, HBM fraction is ~0 by construction "
                                 "(SURVEY 8d); see valu_roofline.  traffic (PMC, profiles/traffic_psa_score.json) is not "
                                 "re-reading of inputs: it is the boundary row every 2048-row band hands to the next (per column "
                                 "and couple 8 B: the packed 16-bit H' / Ebar' halves of both pairs; written once, read once: "
                                 "4096 couples x 24 hand-offs x 50000 columns x 8 B x 2) plus the column sequence re-read per "
                                 "band, ~1 % of HBM bandwidth"},
            "valu_roofline": valu,
        }
        out["rank_ms_per_step"] = {"min": min(r[0] for r in rank_ms), "max": max(r[0] for r in rank_ms),
                                   "kernel_min": min(r[1] for r in rank_ms), "kernel_max": max(r[1] for r in rank_ms),
                                   "per_rank": [round(r[0], 3) for r in rank_ms]}
        out["ranks_seen"] = len(set(uuids))
        out["device_uuids"] = uuids
        out["build_info"] = tsta_amd.build_info()
        if verified is not None:
            out.update(verified)
        if e2e is not None:
            e2e["ratio_to_resident"] = e2e["value"] / out["value"]
            out["e2e"] = e2e
        if related is not None:
            out["related_path"] = related
        if mixed is not None:
            out["mixed_alphabet_path"] = mixed
        if generic_ms is not None:
            out["generic_path"] = {"value": float(P) * L * L / (generic_ms * 1e-3) / 1e9, "unit": "GCUPS", "ms": generic_ms,
                                   "ops_per_cell": VALU_OPS_PER_CELL_GENERIC,
                                   "note": "same shard, psa_elide=0 psa_perm=0: no data-dependent shortcut; identical scores"}
        if world == 1 and not args.no_configs:
            # BASELINE configs #2 / #3 / #4 and the traceback batch, each with its SURVEY 8(d) roofline, verified against the
            # reference's data and with the reference's CPU path beside it (bench_configs.py); the device is theirs now
            try:
                del seqs
            except NameError:
                pass
            torch.cuda.empty_cache()
            from bench_configs import measure_configs
            out["configs"] = measure_configs(eng, cores=cores, verify=not args.no_verify, cpu=not args.no_cpu_baseline, log=log)
        if not args.no_cpu_baseline and world == 1:
            n_s = args.cpu_sample_pairs or min(256, max(4, int(round(16 * (50_000.0 / L) ** 2))))
            log("cpu_baseline")
            out["cpu_baseline"] = cpu_baseline(L, n_s)
            thr = cpu_throughput(L, out["cpu_baseline"]["cores"])
            if thr is not None:
                out["cpu_baseline_throughput"] = thr
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
