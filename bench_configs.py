#!/usr/bin/env python3
"""bench_configs.py — BASELINE.json configs #2, #3, #4 and the config #5 pair shape with traceback ("5t"), measured the way
SURVEY 8(d) defines their rooflines, each VERIFIED against the reference's own data in the same run and with the reference's
CPU path timed beside it where that is cheap.  bench.py imports `measure_configs()` and puts the result into its JSON line
(`"configs": {...}`); run directly (`python bench_configs.py 2 3 4 5t [5m 5s 5h]`) it prints one JSON line per config.

Per config: `kernel_ms` (HIP events on the stream the kernels run on, inside the library), `wall_ms` (host clock around the
C-ABI call: uploads, kernels, downloads, string assembly), `gcups`, `roofline` (trace modes: cells x 1 B / kernel time against
the 8 TB/s HBM peak; POA: (3 B trace + 8 B rows) per cell and the time per graph row; score mode: algorithmic bytes, as the
headline), `verified` + what it was verified against, `cpu_baseline` (oracle/_ref = the unmodified reference compiled by
oracle/Makefile, kind "reference"; on a host that cannot run it, the scalar oracle port)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import tsta_amd  # noqa: E402
from tests import util  # noqa: E402

HBM_PEAK_GBS = 8000.0
BIN = os.path.join(ROOT, "tsta_amd", "bin")


def _log(msg):
    sys.stderr.write("[configs] %s\n" % msg)
    sys.stderr.flush()


def _cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _ref(name):
    """oracle/_ref/<name> if it exists and this host can run it (the default builds are AVX512)."""
    b = os.path.join(util.REF_DIR, name)
    return b if os.path.exists(b) and util.cpu_has("avx512bw") else None


def _run(cmd, limit=120.0, tries=3):
    """A reference process with a time limit and retries (its thread pool can deadlock: pthreadpool.c:96-117)."""
    last = None
    for _ in range(tries):
        try:
            return subprocess.run(cmd, capture_output=True, check=True, timeout=limit)
        except (subprocess.TimeoutExpired, subprocess.CalledProcessError) as ex:
            last = ex
    raise last


def make_shard(n_pairs, length, seed, device):
    from bench import make_shard as ms
    return ms(n_pairs, length, seed, device, related=0.0)


TRACE_FORMS = {0: "psa_trace_tile_kernel<2,true,false> (one launch per tile diagonal)", 1: "psa_trace_tile_kernel<2,true,true> (single launch)",
               2: "psa_trace_quad_kernel<2>", 3: "psa_trace_split_kernel<2,false> (solver bands + emitter tiles)",
               4: "psa_trace_split_kernel<2,true> (solver bands + emitter tiles)", 5: "psa_trace_packed dataflow"}


def _build_hash():
    """The 16 hex digits of the library this process runs (tsta_build_info: SHA-256 over the kernel sources)."""
    try:
        return tsta_amd.lib().tsta_build_info().decode().split()[1]
    except Exception:
        return None


def _traffic(key):
    """-> (HBM bytes per launch of a config's dominant kernel, provenance).  The bytes come from COMMITTED PMC passes
    (profiles/traffic_configs.json, made by tools/collect_profiles.py from separate FETCH_SIZE / WRITE_SIZE runs of this file; gfx950
    correction applied there), not from this run: the provenance names the file and the build it was measured on, and the value is
    dropped (None) when that build is not the one running now."""
    src = {"file": "profiles/traffic_configs.json", "measured_on_build": None, "this_build": _build_hash()}
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic_configs.json")))
        src["measured_on_build"] = t.get("_build")
        v = float(t[key]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None, src
    if src["measured_on_build"] != src["this_build"]:
        src["dropped"] = "PMC passes were made on another build of the kernels"
        return None, src
    return v, src


def _roofline_traffic(d, key):
    d["traffic"], d["traffic_source"] = _traffic(key)
    return d


def _trace_roofline(cells, kernel_ms, kernel, traffic_key=None):
    ach = cells / (kernel_ms * 1e-3) / 1e9
    tr, tsrc = _traffic(traffic_key) if traffic_key else (None, None)
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": tr, "traffic_source": tsrc,
            "kernel": kernel, "note": "SURVEY 8(d) trace mode: cells x 1 B of packed trace / kernel time (fill + on-device walk) against "
                                      "the 8 TB/s HBM-write roof; the binding resource is the chain of dependent steps (one pair) or "
                                      "VALU issue (batches), DESIGN.md section 4"}


# ------------------------------------------------------------------------------------------------ config #2
def config2(eng, cores=None, reps=5, verify=True, cpu=True, seed=20252):
    """1024 pairs of 5 kbp x 5 kbp random DNA, score only (K1b band pipeline, one launch)."""
    P, L = 1024, 5000
    cores = cores or _cores()
    dev = torch.device("cuda", eng.device)
    seqs, off1, len1, off2, len2 = make_shard(P, L, 77, dev)
    scores = torch.empty(P, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    prm = tsta_amd.Params()
    ms, wall = [], []
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        for i in range(reps + 1):
            t0 = time.perf_counter()
            eng.psa_score_batch_dev(prm, P, seqs.data_ptr(), off1.data_ptr(), len1.data_ptr(), off2.data_ptr(),
                                    len2.data_ptr(), scores.data_ptr(), L, st.cuda_stream)
            eng.psa_score_batch_dev_status(st.cuda_stream)             # synchronises; raises on an abandoned single launch
            w = time.perf_counter() - t0
            k, nl = eng.last_kernel_ms()
            if i:
                ms.append(k); wall.append(w * 1e3)
    t = float(np.median(ms))
    cells = float(P) * L * L
    alg = float(P) * (2 * L + 4)
    out = {"config": "#2: 1024 pairs 5 kbp x 5 kbp random DNA, score only, inputs HBM-resident", "kernel_ms": t,
           "wall_ms": float(np.median(wall)), "launches": nl, "gcups": cells / t / 1e6,
           "roofline": {"bound": "hbm", "achieved": alg / (t * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": _traffic("psa_score_bands_kernel")[0],
                        "traffic_source": _traffic("psa_score_bands_kernel")[1], "kernel": "psa_score_bands_kernel<16,true>",
                        "valu_frac_at_4p4_ops_per_cell": cells * 4.4 / (t * 1e-3) / 7.86432e13,
                        "note": "score mode: algorithmic bytes (len_a + len_b + 4 per pair), ~0 by construction (SURVEY 8d); the kernel "
                                "is VALU-issue-bound and, at this size, by the band pipeline's ramp (5 bands of 1 024 rows x 79 tiles per couple)"}}
    sc = scores.cpu().numpy()
    if verify:
        picks = sorted(int(x) for x in np.random.default_rng(seed).choice(P, 16, replace=False))
        h_seqs = seqs.cpu().numpy(); o1 = off1.cpu().numpy(); o2 = off2.cpu().numpy()
        pairs = [(h_seqs[o1[k]:o1[k] + L], h_seqs[o2[k]:o2[k] + L]) for k in picks]
        from bench import reference_scores
        want, kind = reference_scores(pairs, cores)
        got = [int(sc[k]) for k in picks]
        assert got == want, ("config #2: kernel scores differ from the %s" % kind, picks, got, want)
        out.update(verified=True, verified_by="%s maxsorce of %d sampled pairs (seed %d)" % (kind, len(picks), seed), verified_pairs=picks)
    if cpu:
        out["cpu_baseline"] = _cpu_psa_notrace_small(L, cores)
    return out


def _cpu_psa_notrace_small(L, cores, seed=4242, span=6.0):
    """TSTA_psa_notrace on 5 kbp pairs: a pair is 8 blocks of 640 wide, so one process cannot use a machine; run as many
    concurrent processes as cores/4 with -T 4 each, back to back for `span` seconds."""
    binary = _ref("TSTA_psa_notrace")
    if not binary:
        pairs = [(util.random_dna(seed + 2 * i, L), util.random_dna(seed + 2 * i + 1, L)) for i in range(4 * cores)]
        t0 = time.perf_counter(); util.oracle_maxscore_batch(pairs, threads=cores); dt = time.perf_counter() - t0
        return {"value": len(pairs) * float(L) * L / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "port",
                "sample": "%d pairs of %dx%d, scalar oracle, OpenMP over pairs, %.1f s" % (len(pairs), L, L, dt)}
    k, t_each = max(1, cores // 4), min(4, cores)
    with tempfile.TemporaryDirectory() as td:
        files = []
        for i in range(8):
            fa, fb = os.path.join(td, "a%d.fa" % i), os.path.join(td, "b%d.fa" % i)
            util.write_fasta(fa, "a", util.random_dna(seed + 2 * i, L)); util.write_fasta(fb, "b", util.random_dna(seed + 2 * i + 1, L))
            files.append((fa, fb))
        _run([binary, "-T", str(t_each), "-1", files[0][0], "-2", files[0][1]])
        procs = []
        t0 = time.perf_counter()
        for w in range(k):
            fa, fb = files[w % 8]
            sh = ("end=$(( $(date +%%s) + %d )); n=0; while [ $(date +%%s) -lt $end ]; do timeout -k 1 3 %s -T %d -1 %s -2 %s > /dev/null && n=$((n+1)); done; echo $n"
                  % (int(span), binary, t_each, fa, fb))
            procs.append(subprocess.Popen(["sh", "-c", sh], stdout=subprocess.PIPE))
        done = 0
        for pr in procs:
            so, _ = pr.communicate()
            try:
                done += int(so.decode().split()[0])
            except (ValueError, IndexError):
                pass
        dt = time.perf_counter() - t0
    return {"value": done * float(L) * L / dt / 1e9, "unit": "GCUPS", "cores": cores, "kind": "reference",
            "sample": "%d concurrent oracle/_ref/TSTA_psa_notrace -T %d processes aligning %dx%d pairs back to back for %.1f s (%d runs, "
                      "process start-up included: the reference is one process per pair; a run whose thread pool deadlocks is killed at 3 s)" % (k, t_each, L, L, dt, done)}


# ------------------------------------------------------------------------------------------------ config #3
def config3(eng, cores=None, verify=True, cpu=True):
    """One 100 kbp x ~100 kbp pair with traceback: the pair of tests/golden/psa_big_golden.json, whose output file the
    reference wrote in 62 s with 30 GB of host memory (make_big_golden.py)."""
    cores = cores or _cores()
    with open(os.path.join(util.GOLDEN_DIR, "psa_big_golden.json")) as f:
        g = json.load(f)
    x = util.random_dna(g["seed_x"], g["len_x"]); y = util.mutate(x, g["seed_y"], g["rate"])
    eng.psa_align(x[:3000], y[:3000])                       # pages the code in (allocations are per size)
    t0 = time.perf_counter()
    ms, ra, rb = eng.psa_align(x, y)
    wall1 = time.perf_counter() - t0
    k1, nl = eng.last_kernel_ms()
    ks, walls = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        ms2, ra2, rb2 = eng.psa_align(x, y)                 # workspaces allocated
        walls.append((time.perf_counter() - t0) * 1e3)
        ks.append(eng.last_kernel_ms()[0])
    assert (ms, ra, rb) == (ms2, ra2, rb2)
    k2, wall2 = float(np.median(ks)), float(np.median(walls))
    form = eng.stat("last_trace_form")                     # what actually ran (the label below is derived from it)
    jumps = eng.stat("last_walk_jumps")                    # bands the walk crossed through its table (K3t)
    walk = ("psa_walk_origin_kernel + psa_walk_kernel<2,32,true> + psa_walk_emit_kernel (%d of %d bands by table look-up)" % (jumps, (len(y) - 1) // 128)
            if jumps else "psa_walk_kernel<2,32,false>")
    cells = float(len(x)) * len(y)
    out = {"config": "#3: one %d x %d pair with traceback (1 B/cell packed trace in HBM: %.1f GB)" % (len(x), len(y), cells / 1e9),
           "maxsorce": ms, "alignment_len": len(ra), "kernel_ms_first": k1, "kernel_ms": k2, "launches": nl,
           "wall_ms_first": wall1 * 1e3, "wall_ms": wall2, "gcups": cells / k2 / 1e6, "gcups_wall": cells / wall2 / 1e6,
           "reruns": eng.stat("single_launch_reruns"), "trace_form": form, "walk_table_jumps": jumps,
           "roofline": _trace_roofline(cells, k2, TRACE_FORMS.get(form, "form %d" % form) + " + " + walk,
                                       "psa_trace_split_kernel" if form in (3, 4) else None)}
    if out["reruns"]:
        out["flag"] = "a single launch was abandoned and run again through the launch pipeline: kernel_ms is not the single-launch form's"
    if verify:
        data = b">1\n" + ra + b"\n>2\n" + rb                 # what TSTA_psa writes (psa.c:512-516)
        ok = (ms == g["maxsorce"] and len(data) == g["size"] and hashlib.md5(data).hexdigest() == g["md5"] and
              hashlib.sha256(data).hexdigest() == g["sha256"])
        assert ok, "config #3: output differs from the reference's file (tests/golden/psa_big_golden.json)"
        out.update(verified=True, verified_by="md5 + sha256 + size + maxsorce of the reference's own output file for this pair "
                                               "(tests/golden/psa_big_golden.json, oracle/_ref/TSTA_psa)", md5=g["md5"])
    if cpu:
        out["cpu_baseline"] = _cpu_psa_trace(20_000, cores)
        out["cpu_baseline"]["note"] = ("the reference at THIS size took 62.7 s and 30 GB of host memory in the build container (BASELINE.md, "
                                       "make_big_golden.py); timed here at 20 kbp, where it takes a second")
    return out


def _cpu_psa_trace(L, cores, seed=5151):
    """TSTA_psa (with traceback) on one L x L pair, -T = all cores; best of 3."""
    binary = _ref("TSTA_psa")
    a = util.random_dna(seed, L); b = util.mutate(a, seed + 1, 0.05)[:L]
    cells = float(len(a)) * len(b)
    if not binary:
        t0 = time.perf_counter(); util.oracle_align(a, b); dt = time.perf_counter() - t0
        return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": 1, "kind": "port",
                "sample": "one %dx%d pair with traceback, scalar oracle, %.2f s" % (len(a), len(b), dt)}
    with tempfile.TemporaryDirectory() as td:
        fa, fb, fo = os.path.join(td, "a.fa"), os.path.join(td, "b.fa"), os.path.join(td, "o.txt")
        util.write_fasta(fa, "a", a); util.write_fasta(fb, "b", b)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            _run([binary, "-T", str(min(cores, 64)), "-1", fa, "-2", fb, "-o", fo])
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    return {"value": cells / best / 1e9, "unit": "GCUPS", "cores": cores, "kind": "reference",
            "sample": "one %dx%d pair with traceback, oracle/_ref/TSTA_psa -T %d, whole process, best of 3: %.2f s" % (len(a), len(b), min(cores, 64), best)}


# ------------------------------------------------------------------------------------------------ config #4
README_LASTSORCE = [(5000, -5451), (6885, -3101), (8599, -1776), (10130, -870)]       # README.md:28-39 (poa_len, best lastsorce)
README_MD5 = "e7b253a265351c1e4f97dd3b663da7ee"


def config4(eng, cores=None, verify=True, cpu=True):
    """TSTA_msa README example (5 x 5 kbp): the four POA fills + on-device walks, then the drop-in CLI end to end."""
    cores = cores or _cores()
    steps = []
    tot_cells = tot_fill = tot_fw = tot_wall = 0.0
    rows = 0
    ok = True
    for step in range(1, 5):
        z = np.load(os.path.join(util.GOLDEN_DIR, "poa_readme_step%d.npz" % step))
        args = (z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"])
        eng.poa_align(*args)                                 # allocations (the graph grows from step to step)
        fk, fw, ww = [], [], []
        for _ in range(3):
            best, last, _, _, _ = eng.poa_fill(*args)       # (with the three trace arrays, as the walk needs them)
            fk.append(eng.last_kernel_ms()[0])
            t0 = time.perf_counter()
            st, end_col, last2 = eng.poa_align(*args)        # fill + on-device walk: what TSTA_msa calls per sequence
            ww.append((time.perf_counter() - t0) * 1e3)
            fw.append(eng.last_kernel_ms()[0])
        V, n = len(z["base"]), len(z["seq"])
        cells = float(V) * n
        k_fill, k_fw, wall = float(np.median(fk)), float(np.median(fw)), float(np.median(ww))
        tot_cells += cells; tot_fill += k_fill; tot_fw += k_fw; tot_wall += wall; rows += V
        ok = ok and (V, int(last[best])) == README_LASTSORCE[step - 1]
        steps.append({"step": step, "nodes": V, "fill_kernel_ms": k_fill, "fill_plus_walk_kernel_ms": k_fw, "fill_plus_walk_wall_ms": wall,
                      "walk_table_jumps": eng.stat("last_walk_jumps"),
                      "walk_steps": int(len(st)), "best_lastsorce": int(last[best]), "us_per_row": k_fill * 1e3 / V})
    byts = 11.0 * tot_cells                                  # 3 B trace codes + 8 B H / Eagg row per cell
    out = {"config": "#4: TSTA_msa README example (5 x 5 kbp): 4 POA fills + on-device walks", "steps": steps,
           "kernel_ms": tot_fw, "fill_kernel_ms": tot_fill, "walk_kernel_ms": tot_fw - tot_fill, "wall_ms": tot_wall,
           "gcups": tot_cells / tot_fw / 1e6, "gcups_fill": tot_cells / tot_fill / 1e6, "reruns": eng.stat("single_launch_reruns"),
           "roofline": {"bound": "hbm", "achieved": byts / (tot_fill * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": byts / (tot_fill * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": _traffic("poa_dataflow_kernel")[0], "traffic_source": _traffic("poa_dataflow_kernel")[1], "kernel": "poa_dataflow_kernel",
                        "us_per_row": tot_fill * 1e3 / rows,
                        "note": "SURVEY 8(d) POA: 3 B/cell trace + 8 B/cell rows against 8 TB/s, and the time per graph row: the fill "
                                "is the chain of its rows (5 000 columns are 40 workgroups wide), i.e. latency-bound, DESIGN.md section 5"}}
    # the drop-in end to end, next to the reference binary on the same host
    fa = os.path.join(util.GOLDEN_DIR, "msa_seq1.fa")
    ours = os.path.join(BIN, "TSTA_msa")
    cli = None
    if os.path.exists(ours):
        with tempfile.TemporaryDirectory() as td:
            fo = os.path.join(td, "o.txt")
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                r = subprocess.run([ours, "-T", "4", "-i", fa, "-o", fo], capture_output=True, timeout=120)
                ts.append(time.perf_counter() - t0)
                assert r.returncode == 0, r.stderr[-300:]
            md5 = hashlib.md5(open(fo, "rb").read()).hexdigest()
        ok = ok and md5 == README_MD5
        cli = {"tsta_msa_wall_s": min(ts), "md5": md5}
    assert ok or not verify, ("config #4: README known answers / md5 not reproduced", steps, cli)
    if verify:
        out.update(verified=True, verified_by="README.md:28-39 poa_len + lastsorce of all four steps" +
                                              ("; TSTA_msa drop-in output md5 %s = the reference's (README / msa_golden.json)" % README_MD5 if cli else ""))
    if cli:
        out["cli"] = cli
    if cpu:
        ref = _ref("TSTA_msa")
        if ref:
            with tempfile.TemporaryDirectory() as td:
                fo = os.path.join(td, "o.txt")
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    _run([ref, "-T", str(min(cores, 64)), "-i", fa, "-o", fo])
                    ts.append(time.perf_counter() - t0)
            out["cpu_baseline"] = {"value": tot_cells / min(ts) / 1e9, "unit": "GCUPS", "cores": cores, "kind": "reference", "wall_s": min(ts),
                                   "sample": "oracle/_ref/TSTA_msa -T %d on the README example, whole process (fills + node_fuse + output), "
                                             "best of 3: %.3f s" % (min(cores, 64), min(ts))}
            if cli:
                cli["reference_wall_s"] = min(ts)
    return out


# ------------------------------------------------------------------------------------------------ config #5 shape with traceback
def config5t(eng, cores=None, n_pairs=64, L=50_000, verify=True, cpu=True):
    """Not a BASELINE config: the config #5 pair shape WITH traceback, all pairs in one group (1 B/cell: 160 GB of trace)."""
    cores = cores or _cores()
    pairs = []
    for i in range(n_pairs):
        a = util.random_dna(5000 + i, L)
        pairs.append((a, util.mutate(a, 6000 + i, 0.05)[:L]))
    cells = float(sum(len(a) * len(b) for a, b in pairs))
    # the whole batch as ONE group: lift the first-call workspace cap (a one-shot caller would rather work in groups of
    # 16 GB than pay ~35 ms per GB of hipMalloc); the allocation is timed separately
    eng.set_option("align_ws_gb", 4096)
    t0 = time.perf_counter()
    res = eng.psa_align_batch(pairs)
    wall1 = time.perf_counter() - t0
    k1, nl = eng.last_kernel_ms()
    ks, walls = [], []
    for _ in range(2):
        t0 = time.perf_counter()
        res2 = eng.psa_align_batch(pairs)
        walls.append((time.perf_counter() - t0) * 1e3)
        ks.append(eng.last_kernel_ms()[0])
    eng.set_option("align_ws_gb", 0)
    assert res == res2
    k2, wall2 = float(min(ks)), float(min(walls))
    out = {"config": "#5 shape with traceback: %d pairs of %d x ~%d in one group (%.0f GB of packed trace in HBM)" % (n_pairs, L, L, cells / 1e9),
           "kernel_ms_first": k1, "kernel_ms": k2, "launches": nl, "wall_ms_first_with_allocation": wall1 * 1e3, "wall_ms": wall2,
           "gcups": cells / k2 / 1e6, "gcups_wall": cells / wall2 / 1e6, "reruns": eng.stat("single_launch_reruns"),
           "roofline": _trace_roofline(cells, k2, "psa_trace_packed_tile_kernel<4,true> + psa_walk_kernel<4,32,false>", "psa_trace_packed_tile_kernel")}
    if verify:
        for (ms, ra, rb), (a, b) in zip(res, pairs):
            la, lb = (a, b) if len(a) >= len(b) else (b, a)
            assert ra.replace(b"-", b"") == bytes(la) and rb.replace(b"-", b"") == bytes(lb)
        by = "every alignment reproduces its inputs when gaps are removed"
        binary = _ref("TSTA_psa")
        picks = [0, n_pairs - 1] if n_pairs > 1 else [0]
        if binary:
            with tempfile.TemporaryDirectory() as td:
                for k in picks:
                    fa, fb, fo = os.path.join(td, "a.fa"), os.path.join(td, "b.fa"), os.path.join(td, "o.txt")
                    util.write_fasta(fa, "a", pairs[k][0]); util.write_fasta(fb, "b", pairs[k][1])
                    r = _run([binary, "-T", str(min(cores, 64)), "-1", fa, "-2", fb, "-o", fo], limit=300.0)
                    want_ms = int(r.stdout.decode().split("maxsorce=")[1].split()[0])
                    ms, ra, rb = res[k]
                    assert ms == want_ms and open(fo, "rb").read() == b">1\n" + ra + b"\n>2\n" + rb, "5t: pair %d differs from the reference's output file" % k
            by += "; pairs %s byte-identical to oracle/_ref/TSTA_psa's output file and maxsorce" % picks
        else:
            for k in picks:
                o = util.oracle_scores(*pairs[k])
                assert res[k][0] == o[0] and util.alignment_score(res[k][1], res[k][2]) == o[1]
            by += "; pairs %s: maxsorce and path score equal the oracle's" % picks
        out.update(verified=True, verified_by=by)
    if cpu:
        out["cpu_baseline"] = _cpu_psa_trace(20_000, cores)
    eng.release_workspaces()
    return out


# ------------------------------------------------------------------------------------------------ experiments (not BASELINE configs)
def config5m(eng, P=128, L=50_000, reps=2):
    """The config #5 pair shape in a batch far too small for one wave per couple (K1b)."""
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
    """The config #5 pair shape with a scoring whose mismatch costs at most two gap extensions (M=2 X=-4 E=-2 O=-4,
    minimap2's): pure-ACGT couples then run their elided blocks with BOTH gap extensions free."""
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


CONFIGS = {"2": config2, "3": config3, "4": config4, "5t": config5t}


def measure_configs(eng, which=("3", "4", "2", "5t"), cores=None, verify=True, cpu=True, log=_log):
    """-> {"2": {...}, "3": {...}, "4": {...}, "5t": {...}} for bench.py's JSON line (N = 1 only)."""
    out = {}
    for w in which:
        t0 = time.perf_counter()
        log("config %s" % w)
        out[w] = CONFIGS[w](eng, cores=cores, verify=verify, cpu=cpu)
        out[w]["bench_seconds"] = time.perf_counter() - t0
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    eng = tsta_amd.Engine(0)
    which = sys.argv[1:] or ["2", "3", "4"]
    no_cpu = bool(os.environ.get("TSTA_BENCH_NO_CPU"))
    for w in which:
        if w in CONFIGS:
            r = CONFIGS[w](eng, verify=not os.environ.get("TSTA_BENCH_NO_VERIFY"), cpu=not no_cpu)
        else:
            r = {"5m": lambda: config5m(eng, int(os.environ.get("TSTA_BENCH_MID_PAIRS", 128))), "5s": lambda: config5s(eng),
                 "5h": lambda: config5h(eng)}[w]()
        r["build_info"] = tsta_amd.build_info()
        print(json.dumps(r), flush=True)
    eng.close()
