#!/usr/bin/env python3
"""bench_configs.py — secondary measurements for BASELINE.json configs #2, #3, #4 (bench.py carries the headline
metric / config #5 shard).  One JSON line per config; kernel time from HIP events inside the library."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import tsta_amd  # noqa: E402
from bench import make_shard  # noqa: E402
from tests import util  # noqa: E402


def config2(eng, reps=5):
    P, L = 1024, 5000
    dev = torch.device("cuda", 0)
    seqs, off1, len1, off2, len2 = make_shard(P, L, 77, dev)
    scores = torch.empty(P, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    prm = tsta_amd.Params()
    ms = []
    with torch.cuda.stream(st):
        for i in range(reps + 1):
            eng.psa_score_batch_dev(prm, P, seqs.data_ptr(), off1.data_ptr(), len1.data_ptr(), off2.data_ptr(),
                                    len2.data_ptr(), scores.data_ptr(), L, st.cuda_stream)
            k, _ = eng.last_kernel_ms()
            if i:
                ms.append(k)
    t = float(np.median(ms))
    return {"config": "#2: 1024 pairs 5 kbp x 5 kbp, score only", "kernel_ms": t, "gcups": P * L * L / t / 1e6,
            "valu_frac_at_10ops": P * L * L * 10 / (t * 1e-3) / 7.86432e13}


def config5m(eng, P=128, L=50_000, reps=2):
    """Not a BASELINE config: the config #5 pair shape in a batch far too small for one wave per couple (K1b)."""
    dev = torch.device("cuda", 0)
    seqs, off1, len1, off2, len2 = make_shard(P, L, 78, dev)
    scores = torch.empty(P, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    prm = tsta_amd.Params()
    ms = []
    with torch.cuda.stream(st):
        for i in range(reps + 1):
            eng.psa_score_batch_dev(prm, P, seqs.data_ptr(), off1.data_ptr(), len1.data_ptr(), off2.data_ptr(),
                                    len2.data_ptr(), scores.data_ptr(), L, st.cuda_stream)
            k, nl = eng.last_kernel_ms()
            if i:
                ms.append(k)
    t = float(np.median(ms))
    return {"config": "#5 shape, only %d pairs of %d x %d, score only (band pipeline K1b)" % (P, L, L), "kernel_ms": t,
            "launches": nl, "gcups": P * L * L / t / 1e6}


def config5h(eng, P=8192, L=50_000):
    """The bench.py shard through the HOST-pointer entry point: H2D of the sequences + kernels + D2H of the scores."""
    import ctypes as C
    dev = torch.device("cuda", 0)
    seqs, off1, len1, off2, len2 = make_shard(P, L, 1000, dev)
    h_seqs = seqs.cpu().numpy(); h_o1 = off1.cpu().numpy(); h_l1 = len1.cpu().numpy(); h_o2 = off2.cpu().numpy(); h_l2 = len2.cpu().numpy()
    del seqs
    torch.cuda.empty_cache()
    out = np.zeros(P, dtype=np.int32)
    prm = tsta_amd.Params()
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    wall = []
    for _ in range(2):
        t0 = time.perf_counter()
        rc = tsta_amd.lib().tsta_psa_score_batch(eng._h, C.byref(prm), P, ptr(h_seqs), h_seqs.size, ptr(h_o1), ptr(h_l1), ptr(h_o2), ptr(h_l2), ptr(out))
        wall.append(time.perf_counter() - t0)
        assert rc == 0
    k, _ = eng.last_kernel_ms()
    return {"config": "#5 shard (8192 x 50 kbp^2) through tsta_psa_score_batch (host pointers, pageable memory): H2D + kernels + D2H",
            "wall_ms_first": wall[0] * 1e3, "wall_ms": wall[1] * 1e3, "kernel_ms": k, "gcups_wall": P * float(L) * L / wall[1] / 1e9,
            "gcups_kernel": P * float(L) * L / k / 1e6, "h2d_bytes": int(h_seqs.size)}


def config5s(eng, P=4096, L=50_000):
    """Not a BASELINE config: the config #5 pair shape with a scoring whose mismatch costs at most two gap extensions
    (M=2 X=-4 E=-2 O=-4, minimap2's): pure-ACGT couples then run their elided blocks with BOTH gap extensions free."""
    dev = torch.device("cuda", 0)
    seqs, off1, len1, off2, len2 = make_shard(P, L, 79, dev)
    scores = torch.empty(P, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    prm = tsta_amd.Params(2, -4, -2, -4)
    out = {}
    with torch.cuda.stream(st):
        for ff in (1, 0):
            eng.set_option("psa_ffree", ff)
            ms = []
            for i in range(2):
                eng.psa_score_batch_dev(prm, P, seqs.data_ptr(), off1.data_ptr(), len1.data_ptr(), off2.data_ptr(),
                                        len2.data_ptr(), scores.data_ptr(), L, st.cuda_stream)
                k, _ = eng.last_kernel_ms()
                ms.append(k)
            out[ff] = (ms[-1], scores.clone())
    eng.set_option("psa_ffree", 1)
    assert torch.equal(out[0][1], out[1][1])
    return {"config": "#5 shape, %d pairs of %d x %d, score only, scoring M=2 X=-4 E=-2 O=-4" % (P, L, L),
            "kernel_ms_free_free_frame": out[1][0], "gcups_free_free_frame": P * float(L) * L / out[1][0] / 1e6,
            "kernel_ms_without": out[0][0], "gcups_without": P * float(L) * L / out[0][0] / 1e6}


def config3(eng):
    x = util.random_dna(777, 100_000); y = util.mutate(x, 778, 0.05)
    eng.psa_align(x[:3000], y[:3000])                       # warm-up (allocations are per-size, this only pages code in)
    t0 = time.perf_counter()
    ms, ra, rb = eng.psa_align(x, y)
    wall = time.perf_counter() - t0
    k, nl = eng.last_kernel_ms()
    cells = float(len(x)) * len(y)
    t0 = time.perf_counter()
    ms2, ra2, rb2 = eng.psa_align(x, y)                     # second call: workspaces already allocated
    wall2 = time.perf_counter() - t0
    k2, _ = eng.last_kernel_ms()
    assert (ms, ra, rb) == (ms2, ra2, rb2)
    return {"config": "#3: one 100 kbp x ~100 kbp pair with traceback (1 B/cell packed trace in HBM)",
            "maxsorce": ms, "alignment_len": len(ra), "kernel_ms_first": k, "kernel_ms": k2, "launches": nl,
            "wall_ms_first": wall * 1e3, "wall_ms": wall2 * 1e3, "gcups_kernel": cells / k2 / 1e6,
            "gcups_wall": cells / wall2 / 1e9, "trace_bytes": cells,
            "trace_write_GBs": cells / (k2 * 1e-3) / 1e9}


def config5t(eng, n_pairs=64, L=50_000):
    """Not a BASELINE config: the config #5 pair shape WITH traceback, as many pairs as fit the device in one group."""
    pairs = []
    for i in range(n_pairs):
        a = util.random_dna(5000 + i, L)
        pairs.append((a, util.mutate(a, 6000 + i, 0.05)[:L]))
    # a one-shot caller (TSTA_psa_batch -o): fresh engine, first call.  Device memory costs ~35 ms per GB to allocate, so the
    # engine works in groups of <= 16 GB of trace on its first call and grows the workspace over the next calls
    fresh = tsta_amd.Engine(0)
    t0 = time.perf_counter()
    res = fresh.psa_align_batch(pairs)
    wall = time.perf_counter() - t0
    fresh.close()
    eng.psa_align_batch(pairs); eng.psa_align_batch(pairs); eng.psa_align_batch(pairs)     # grow the workspace to everything the batch wants
    k, nl = eng.last_kernel_ms()
    cells = float(sum(len(a) * len(b) for a, b in pairs))
    t0 = time.perf_counter()
    res2 = eng.psa_align_batch(pairs)
    wall2 = time.perf_counter() - t0
    k2, _ = eng.last_kernel_ms()
    assert res == res2
    one = eng.psa_align(*pairs[0])
    assert one == res[0]
    return {"config": "#5 shape with traceback: %d pairs of %d x ~%d in one group (1 B/cell packed trace in HBM)" % (n_pairs, L, L),
            "kernel_ms_first": k, "kernel_ms": k2, "launches": nl, "wall_ms_first": wall * 1e3, "wall_ms": wall2 * 1e3,
            "gcups_kernel": cells / k2 / 1e6, "gcups_wall": cells / wall2 / 1e9, "trace_bytes": cells,
            "trace_write_GBs": cells / (k2 * 1e-3) / 1e9, "hbm_write_frac_of_8TBs": cells / (k2 * 1e-3) / 8e12}


def config4(eng):
    out = []
    tot_cells, tot_k, tot_wall = 0.0, 0.0, 0.0
    for step in range(1, 5):
        z = np.load(os.path.join(util.GOLDEN_DIR, "poa_readme_step%d.npz" % step))
        args = (z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"])
        eng.poa_fill(*args)                                 # warm-up / allocations
        t0 = time.perf_counter()
        best, last, src, es, fs = eng.poa_fill(*args)
        wall = time.perf_counter() - t0
        k, _ = eng.last_kernel_ms()
        eng.poa_align(*args)
        t0 = time.perf_counter()
        steps, end_col, _ = eng.poa_align(*args)                # fill + on-device walk: what TSTA_msa uses
        wall2 = time.perf_counter() - t0
        k2, _ = eng.last_kernel_ms()
        cells = float(len(z["base"])) * len(z["seq"])
        tot_cells += cells; tot_k += k; tot_wall += wall2
        out.append({"step": step, "nodes": int(len(z["base"])), "fill_kernel_ms": k, "fill_wall_ms_with_trace_d2h": wall * 1e3,
                    "fill_plus_walk_kernel_ms": k2, "fill_plus_walk_wall_ms": wall2 * 1e3, "walk_steps": int(len(steps)),
                    "best_lastsorce": int(last[best])})
    return {"config": "#4: TSTA_msa README example, 4 POA fills (5 x 5 kbp)", "steps": out,
            "fill_kernel_ms_total": tot_k, "gcups_fill_kernel": tot_cells / tot_k / 1e6,
            "fill_plus_walk_wall_ms_total": tot_wall * 1e3}


if __name__ == "__main__":
    eng = tsta_amd.Engine(0)
    which = sys.argv[1:] or ["2", "3", "4"]
    # (Round 1 measured the launch-per-tile pipelines before anything initialised PyTorch's HIP context: the four overlapping
    # launch sequences of tsta_psa_align_batch stopped overlapping once the process had other streams.  Cause: hardware queues;
    # the single-launch forms of round 2 do not depend on them.  The order is kept.)
    if "3" in which:
        print(json.dumps(config3(eng)), flush=True)
    if "5t" in which:
        if os.environ.get("TSTA_BENCH_RELEASE"):
            eng.release_workspaces()
        if os.environ.get("TSTA_BENCH_TORCH_TOUCH"):          # experiment: does any torch GPU activity change what follows?
            n_el = int(os.environ["TSTA_BENCH_TORCH_TOUCH"])
            x = torch.zeros(n_el, device="cuda", dtype=torch.uint8); torch.cuda.synchronize(); del x
        print(json.dumps(config5t(eng, int(os.environ.get("TSTA_BENCH_ALIGN_PAIRS", 64)))), flush=True)
    if "4" in which:
        print(json.dumps(config4(eng)), flush=True)
    if "2" in which:
        print(json.dumps(config2(eng)), flush=True)
    if "5s" in which:
        print(json.dumps(config5s(eng)), flush=True)
    if "5m" in which:
        print(json.dumps(config5m(eng, int(os.environ.get("TSTA_BENCH_MID_PAIRS", 128)))), flush=True)
    if "5h" in which:
        print(json.dumps(config5h(eng)), flush=True)
    eng.close()
