"""GPU parity: K4 (poa_fill_kernel) through tsta_poa_fill vs dumps of the compiled reference (every trace byte /
sha256, sinks' lastsorce, best sink) and vs the oracle on synthetic DAGs with many predecessors."""
import glob
import os

import numpy as np
import pytest

import tsta_amd
from tests import util
from tests.test_poa_oracle_golden import FIXTURES, check_against_fixture, load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def _engine():
    e = tsta_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture()
def eng(_engine):
    # the plain kernels (K4 / K4d) against the plain oracle on arbitrary DAGs: "poa_ref_arith" = 0 keeps calls the reference's own
    # arithmetic would answer differently (far predecessors, sequences spanning more blocks than the graph) on those kernels;
    # the tests of the reference-arithmetic path switch it on themselves
    _engine.set_option("poa_ref_arith", 0)
    yield _engine
    _engine.set_option("poa_ref_arith", 1)


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[4:-4] for p in FIXTURES])
def test_reference_dumps(eng, path):
    z, sc = load(path)
    best, last, src, es, fs = eng.poa_fill(z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"],
                                           tsta_amd.Params(**sc))
    check_against_fixture(z, best, last, src, es, fs)


def random_dag(seed, V, max_pred, n):
    """A random DAG in topological order (predecessors anywhere earlier, several sources/sinks) + a sequence."""
    r = util.splitmix64(seed, 4 * V + 8)
    base = np.frombuffer(b"ACGT", np.uint8)[(r[:V] >> np.uint64(62)).astype(np.int64)]
    pred_off = [0]; pred = []
    has_succ = np.zeros(V, bool)
    for v in range(V):
        k = int(r[V + v] % np.uint64(max_pred + 1)) if v > 0 else 0
        if v > 0 and int(r[2 * V + v] % np.uint64(10)) != 0:
            k = max(k, 1)                      # mostly connected; ~10% extra sources
        k = min(k, v)
        ps = set()
        if k >= 1:
            ps.add(v - 1)                      # chain edge first, like a POA backbone
        q = util.splitmix64(seed * 7919 + v, max(1, k))
        for x in range(k):
            if len(ps) >= k:
                break
            ps.add(int(q[x] % np.uint64(v)))
        ps = sorted(ps, reverse=True)          # arbitrary but fixed order of pre[]
        for pidx in ps:
            has_succ[pidx] = True
        pred.extend(ps); pred_off.append(len(pred))
    is_sink = (~has_succ).astype(np.uint8)
    seq = util.random_dna(seed + 1, n)
    return np.array(pred_off, np.int32), np.array(pred, np.int32), base, is_sink, seq


@pytest.mark.parametrize("seed,V,max_pred,n", [(1, 50, 3, 40), (2, 300, 8, 1), (3, 64, 41, 700), (4, 1000, 5, 9),
                                               (5, 700, 12, 8193), (6, 1, 0, 100), (7, 400, 2, 1025),
                                               (8, 2500, 3, 16384), (9, 300, 2, 16385),
                                               (10, 200, 3, 32640), (11, 150, 2, 32641), (12, 120, 2, 40000)])    # 255 / 256 / 313 chunks: more chunks than CUs (roles by ticket)
def test_random_dags_vs_oracle(eng, seed, V, max_pred, n):
    pred_off, pred, base, is_sink, seq = random_dag(seed, V, max_pred, n)
    for sc in (dict(M=2, X=-5, E=-2, O=-4), dict(M=1, X=-1, E=-1, O=0)):
        got = eng.poa_fill(pred_off, pred, base, is_sink, seq, tsta_amd.Params(**sc))
        want = util.oracle_poa_fill(pred_off, pred, base, is_sink, seq, **sc)
        assert got[0] == want[0]
        assert np.array_equal(got[1], want[1])
        for g, w in zip(got[2:], want[2:]):
            assert np.array_equal(g, w)


def test_four_columns_per_lane_variant(eng):
    # the default is 2 columns per lane; exercise the 256-column-tile instantiation too
    try:
        eng.set_option("poa_cols", 4)
        for path in FIXTURES[:6] + FIXTURES[-3:]:
            z, sc = load(path)
            best, last, src, es, fs = eng.poa_fill(z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"],
                                                   tsta_amd.Params(**sc))
            check_against_fixture(z, best, last, src, es, fs)
        pred_off, pred, base, is_sink, seq = random_dag(5, 700, 12, 8193)
        got = eng.poa_fill(pred_off, pred, base, is_sink, seq)
        want = util.oracle_poa_fill(pred_off, pred, base, is_sink, seq)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]) and all(np.array_equal(g, w) for g, w in zip(got[2:], want[2:]))
    finally:
        eng.set_option("poa_cols", 0)


def test_abandoned_single_launch_is_run_again(eng):
    # the single-launch fill needs all its workgroups resident; a launch that abandons its waits is repeated through the
    # launch-per-diagonal form by the entry points themselves ("poa_dataflow" = 2 reports every launch as abandoned once)
    z, sc = load(FIXTURES[0])
    args = (z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"])
    want = eng.poa_fill(*args, tsta_amd.Params(**sc))
    want_steps, want_end, _ = eng.poa_align(*args, tsta_amd.Params(**sc))
    try:
        eng.set_option("poa_dataflow", 2)
        got = eng.poa_fill(*args, tsta_amd.Params(**sc))
        steps, end_col, _ = eng.poa_align(*args, tsta_amd.Params(**sc))
    finally:
        eng.set_option("poa_dataflow", 1)
    assert got[0] == want[0] and all(np.array_equal(g, w) for g, w in zip(got[1:], want[1:]))
    assert end_col == want_end and np.array_equal(steps, want_steps)


def test_launch_per_diagonal_variant(eng):
    # the default for these sizes is the single-launch dataflow fill (poa_dataflow.hip: solver + emitter waves); the
    # launch-per-anti-diagonal K4 (which also serves sequences with 'N') must give the same bytes
    try:
        eng.set_option("poa_dataflow", 0)
        for path in FIXTURES[:4] + FIXTURES[-4:]:
            z, sc = load(path)
            best, last, src, es, fs = eng.poa_fill(z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"],
                                                   tsta_amd.Params(**sc))
            check_against_fixture(z, best, last, src, es, fs)
        for seed, V, mp, n in [(3, 64, 41, 700), (5, 700, 12, 8193), (7, 400, 2, 1025)]:
            pred_off, pred, base, is_sink, seq = random_dag(seed, V, mp, n)
            got = eng.poa_fill(pred_off, pred, base, is_sink, seq)
            want = util.oracle_poa_fill(pred_off, pred, base, is_sink, seq)
            assert got[0] == want[0] and np.array_equal(got[1], want[1]) and all(np.array_equal(g, w) for g, w in zip(got[2:], want[2:]))
    finally:
        eng.set_option("poa_dataflow", 1)


def test_rejects_bad_dags(eng):
    base = np.frombuffer(b"ACG", np.uint8); sink = np.array([0, 0, 1], np.uint8); seq = util.random_dna(1, 10)
    with pytest.raises(tsta_amd.TstaError):       # predecessor not earlier in topological order
        eng.poa_fill([0, 0, 1, 2], [2, 1], base, sink, seq)
    V = 50
    pred_off = [0] + [0] * 43 + [43] * (V - 43)   # node 43 has 43 predecessors (>= 42: outside the trace code space)
    with pytest.raises(tsta_amd.TstaError):
        eng.poa_fill(np.array(pred_off, np.int32), np.arange(43, dtype=np.int32), util.random_dna(2, V),
                     np.ones(V, np.uint8), seq)


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[4:-4] for p in FIXTURES])
def test_on_device_walk_matches_host_walk(eng, path):
    # tsta_poa_align (fill + walk on the GPU) vs the same walk done in Python over the arrays tsta_poa_fill returns
    z, sc = load(path)
    args = (z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"])
    best, last, src, es, fs = eng.poa_fill(*args, tsta_amd.Params(**sc))
    want, want_end = util.poa_walk(z["pred_off"], z["pred"], z["is_sink"], last, src, es, fs)
    steps, end_col, last2 = eng.poa_align(*args, tsta_amd.Params(**sc))
    assert np.array_equal(last, last2)
    assert end_col == want_end
    got = [(int(s["node"]), int(s["col"]), int(s["code"]), int(s["peek"])) for s in steps]
    assert got == want


def test_on_device_walk_random_dags(eng):
    for seed, V, mp, n in [(11, 300, 6, 257), (12, 2000, 3, 100), (13, 90, 20, 1500)]:
        pred_off, pred, base, is_sink, seq = random_dag(seed, V, mp, n)
        best, last, src, es, fs = eng.poa_fill(pred_off, pred, base, is_sink, seq)
        want, want_end = util.poa_walk(pred_off, pred, is_sink, last, src, es, fs)
        steps, end_col, _ = eng.poa_align(pred_off, pred, base, is_sink, seq)
        got = [(int(s["node"]), int(s["col"]), int(s["code"]), int(s["peek"])) for s in steps]
        assert (got, end_col) == (want, want_end)


def sprinkle_n(seq, seed, positions=(), count=0):
    seq = seq.copy()
    for c in positions:
        if 0 <= c < len(seq):
            seq[c] = ord("N")
    if count and len(seq):
        idx = (util.splitmix64(seed, count) % np.uint64(len(seq))).astype(np.int64)
        seq[idx] = ord("N")
    return seq


@pytest.mark.parametrize("cols", [2, 4])
def test_literal_n_columns_vs_oracle(eng, cols):
    # a literal 'N' in the sequence forces H(r,c) := H(r,c-1) for every node, trace codes by equality (c-t-simd.c:461,481,
    # 485-506): first / last column, runs of N, several N per tile and per lane, tiles without N next to tiles with N,
    # nodes whose base is 'N'; every trace byte, lastscore and the best sink against the oracle; the call reports
    # TSTA_WARN_REFERENCE_UNDEFINED (the reference's own output on such input depends on -W)
    # (this is the CONSISTENT rule of the tile kernel K4, "poa_ref_arith" = 0; by default such sequences are filled in the
    # reference's own arithmetic and blocking: test_literal_n_in_the_reference_arithmetic)
    try:
        eng.set_option("poa_cols", cols); eng.set_option("poa_ref_arith", 0)
        cases = [(21, 120, 3, 300, (0, 1, 2, 299, 150, 151, 152, 153), 0), (22, 400, 6, 1000, (127, 128, 129, 255, 256, 511, 512, 999), 12),
                 (23, 70, 20, 129, (), 40), (24, 900, 4, 2000, (5,), 0), (25, 33, 2, 64, tuple(range(0, 64, 3)), 0), (26, 10, 2, 7, (6,), 0),
                 (27, 200, 5, 513, tuple(range(100, 140)), 5)]
        for seed, V, mp, n, pos, cnt in cases:
            pred_off, pred, base, is_sink, seq = random_dag(seed, V, mp, n)
            seq = sprinkle_n(seq, seed, pos, cnt)
            base = base.copy(); base[::17] = ord("N")
            for sc in (dict(M=2, X=-5, E=-2, O=-4), dict(M=1, X=-1, E=-1, O=0)):
                got = eng.poa_fill(pred_off, pred, base, is_sink, seq, tsta_amd.Params(**sc))
                assert eng.last_status == 1 and "literal 'N'" in eng.last_warning()
                want = util.oracle_poa_fill(pred_off, pred, base, is_sink, seq, **sc)
                assert got[0] == want[0], (seed, sc)
                assert np.array_equal(got[1], want[1]), (seed, sc)
                for name, g, w in zip(("source", "esource", "fsource"), got[2:], want[2:]):
                    bad = np.argwhere(g != w)
                    assert len(bad) == 0, (seed, sc, name, bad[:5].tolist(), g[tuple(bad[0])], w[tuple(bad[0])])
            # the on-device walk over those arrays
            best, last, src, es, fs = eng.poa_fill(pred_off, pred, base, is_sink, seq)
            want_steps, want_end = util.poa_walk(pred_off, pred, is_sink, last, src, es, fs)
            steps, end_col, _ = eng.poa_align(pred_off, pred, base, is_sink, seq)
            assert eng.last_status == 1
            assert ([(int(t["node"]), int(t["col"]), int(t["code"]), int(t["peek"])) for t in steps], end_col) == (want_steps, want_end)
    finally:
        eng.set_option("poa_cols", 0)


@pytest.mark.parametrize("W", [10, 3, 5, 16, 30, 1, 32])
def test_literal_n_in_the_reference_arithmetic(eng, W):
    # K4e (poa_ref8.hip): what the reference's AVX512 build computes at -W W on a sequence with a literal 'N' — its int8 differences,
    # saturating offset trackers, both passes, the block schedule — every trace byte, lastscore and the best sink against
    # oracle_poa_fill_i8, which reproduces the compiled reference on all 200 outputs of msa_n_golden.json (tests/test_msa_host.py).
    # Random DAGs with far predecessors drive the trackers into their reservoirs and beyond the cutoff: same bytes there too.
    try:
        eng.set_option("poa_ref_block", 64 * W); eng.set_option("poa_ref_arith", 1)
        cases = [(21, 120, 3, 300, (0, 1, 2, 299, 150, 151, 152, 153), 0), (22, 400, 6, 1000, (127, 128, 129, 255, 256, 511, 512, 999), 12),
                 (23, 70, 20, 129, (), 40), (24, 900, 4, 2000, (5,), 0), (25, 33, 2, 64, tuple(range(0, 64, 3)), 0), (26, 10, 2, 7, (6,), 0),
                 (27, 200, 5, 513, tuple(range(100, 140)), 5), (28, 1500, 3, 1400, (639, 640, 641, 1279), 3)]
        for seed, V, mp, n, pos, cnt in cases:
            pred_off, pred, base, is_sink, seq = random_dag(seed, V, mp, n)
            seq = sprinkle_n(seq, seed, pos, cnt)
            base = base.copy(); base[::17] = ord("N")
            for sc in (dict(M=2, X=-5, E=-2, O=-4), dict(M=1, X=-1, E=-1, O=0)):
                got = eng.poa_fill(pred_off, pred, base, is_sink, seq, tsta_amd.Params(**sc))
                assert eng.last_status == 0 and eng.last_warning() == "", (seed, eng.last_warning())
                want = util.oracle_poa_fill_i8(pred_off, pred, base, is_sink, seq, W, **sc)
                assert got[0] == want[0], (seed, sc)
                assert np.array_equal(got[1], want[1]), (seed, sc, np.argwhere(got[1] != want[1])[:5].tolist())
                for name, g, w in zip(("source", "esource", "fsource"), got[2:], want[2:]):
                    bad = np.argwhere(g != w)
                    assert len(bad) == 0, (seed, sc, name, len(bad), bad[:5].tolist(), g[tuple(bad[0])], w[tuple(bad[0])])
            best, last, src, es, fs = eng.poa_fill(pred_off, pred, base, is_sink, seq)
            want_steps, want_end = util.poa_walk(pred_off, pred, is_sink, last, src, es, fs)
            steps, end_col, _ = eng.poa_align(pred_off, pred, base, is_sink, seq)
            assert ([(int(t["node"]), int(t["col"]), int(t["code"]), int(t["peek"])) for t in steps], end_col) == (want_steps, want_end)
        # a sequence without 'N': status 0 always means "the reference's bytes" — the plain kernels inside the reference's domain, this
        # kernel from -W 28 on (padding rule, poa_needs_reference_arithmetic) or after the plain fill saw predecessor rows too far apart
        pred_off, pred, base, is_sink, seq = random_dag(29, 900, 2, 700)
        got = eng.poa_fill(pred_off, pred, base, is_sink, seq)
        assert eng.last_status == 0 and eng.last_warning() == ""
        want = util.oracle_poa_fill_i8(pred_off, pred, base, is_sink, seq, W)
        assert got[0] == want[0] and all(np.array_equal(g, w) for g, w in zip(got[1:], want[1:]))
    finally:
        eng.set_option("poa_ref_block", 640)


@pytest.mark.parametrize("W", [10, 4, 32])
def test_every_fill_in_the_reference_arithmetic(eng, W):
    # "poa_ref_arith" = 2: K4e for every fill.  On the dumps of the reference itself (made at -W 10) it must give what the plain
    # kernels give; at other -W it must follow the int8 restatement wherever that goes (at -W 32 the reference leaves the DP on the
    # padded "medium" inputs: tests/test_msa_host.py holds the reference's own bytes for that)
    try:
        eng.set_option("poa_ref_block", 64 * W); eng.set_option("poa_ref_arith", 2)
        for f in FIXTURES:
            if "readme" in f and W != 10:
                continue
            z, sc = load(f)
            args = (z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"])
            got = eng.poa_fill(*args, tsta_amd.Params(**sc))
            assert eng.last_status == 0
            want = util.oracle_poa_fill_i8(*args, W, **sc)
            assert got[0] == want[0] and all(np.array_equal(g, w) for g, w in zip(got[1:], want[1:])), f
            if W == 10:
                check_against_fixture(z, *got)
    finally:
        eng.set_option("poa_ref_block", 640); eng.set_option("poa_ref_arith", 1)


def test_reference_arithmetic_fuzz(eng):
    # K4e against the int8 restatement on random (DAG, sequence, scoring, -W): every shape of control()'s schedule (more column
    # blocks than row blocks and the other way round, single blocks), in-degrees up to 12, literal 'N's or none, scorings that drive
    # the offsets into their reservoirs
    rng = np.random.default_rng(20260403)
    try:
        eng.set_option("poa_ref_arith", 2)
        for it in range(48):
            V = int(rng.integers(1, 420)); n = int(rng.integers(1, 520)); mp = int(rng.integers(1, 13)); W = int(rng.integers(1, 33))
            sc = dict(M=int(rng.integers(1, 7)), X=-int(rng.integers(1, 9)), E=-int(rng.integers(1, 7)), O=-int(rng.integers(0, 13)))
            pred_off, pred, base, is_sink, seq = random_dag(1000 + it, V, mp, n)
            if it % 3 == 0:
                seq = sprinkle_n(seq, it, (), int(rng.integers(1, 6)))
            eng.set_option("poa_ref_block", 64 * W)
            got = eng.poa_fill(pred_off, pred, base, is_sink, seq, tsta_amd.Params(**sc))
            if eng.last_status != 0:                             # (in-degree x W beyond the LDS: the plain DP with its warning)
                assert mp * ((W + 3) // 4 * 2 + 3) * 256 > 152 * 1024
                continue
            want = util.oracle_poa_fill_i8(pred_off, pred, base, is_sink, seq, W, **sc)
            assert got[0] == want[0] and all(np.array_equal(g, w) for g, w in zip(got[1:], want[1:])), (it, V, n, mp, W, sc)
            if it % 5 == 0:                                      # scores only (no trace arrays on the device)
                best, last = eng.poa_fill(pred_off, pred, base, is_sink, seq, tsta_amd.Params(**sc), want_trace=False)[:2]
                assert best == want[0] and np.array_equal(last, want[1]), (it, "no trace")
        # the largest lane state the kernel takes: in-degree 32 at -W 32 = 152 KB of dynamic LDS (+ 7 KB static of the CU's 160 KB)
        pred_off, pred, base, is_sink, seq = random_dag(77, 300, 32, 700)
        assert int(np.max(np.diff(pred_off))) == 32
        seq = sprinkle_n(seq, 77, (), 3)
        eng.set_option("poa_ref_block", 64 * 32)
        got = eng.poa_fill(pred_off, pred, base, is_sink, seq)
        assert eng.last_status == 0
        want = util.oracle_poa_fill_i8(pred_off, pred, base, is_sink, seq, 32)
        assert got[0] == want[0] and all(np.array_equal(g, w) for g, w in zip(got[1:], want[1:]))
        # ... and one predecessor more does not fit: the plain DP with its warning
        pred_off, pred, base, is_sink, seq = random_dag(78, 300, 33, 700)
        if int(np.max(np.diff(pred_off))) == 33:
            eng.poa_fill(pred_off, pred, base, is_sink, sprinkle_n(seq, 78, (), 3))
            assert eng.last_status == 1 and "literal 'N'" in eng.last_warning()
    finally:
        eng.set_option("poa_ref_block", 640)


def test_reference_undefined_warnings(eng):
    # inside the reference's domain (a dump of the reference itself): status 0, no warning
    z, sc = load(FIXTURES[0])
    eng.poa_fill(z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"], tsta_amd.Params(**sc), want_trace=False)
    assert eng.last_status == 0 and eng.last_warning() == ""
    # (2) the sequence spans more L-blocks than the graph (c-t-simd.c:194 vs :263): L = option poa_ref_block
    V = 600
    chain_off = np.arange(V + 1, dtype=np.int32); chain_off[1:] -= 1; chain_off[0] = 0
    chain_pred = np.arange(V - 1, dtype=np.int32)
    cbase = util.random_dna(32, V); csink = np.zeros(V, np.uint8); csink[-1] = 1
    seq = util.random_dna(34, 700)
    eng.set_option("poa_ref_arith", 0)                          # (the plain kernels only: the call says what it cannot vouch for)
    eng.poa_fill(chain_off, chain_pred, cbase, csink, seq, want_trace=False)
    assert eng.last_status == 1 and "spans more blocks" in eng.last_warning()
    eng.set_option("poa_ref_arith", 1)                          # default: such a fill runs in the reference's own arithmetic instead
    got = eng.poa_fill(chain_off, chain_pred, cbase, csink, seq)
    assert eng.last_status == 0 and eng.last_warning() == ""
    want = util.oracle_poa_fill_i8(chain_off, chain_pred, cbase, csink, seq, 10)
    assert got[0] == want[0] and all(np.array_equal(g, w) for g, w in zip(got[1:], want[1:]))
    eng.set_option("poa_ref_block", 1024)
    eng.poa_fill(chain_off, chain_pred, cbase, csink, seq, want_trace=False)
    assert eng.last_status == 0
    eng.set_option("poa_ref_block", 640)
    # (3) two predecessor rows of one node far apart: a chain that spells the sequence and a chain of a base the sequence
    # never has, joined in a final node
    n = 200
    seq = util.random_dna(33, n); seq[seq == ord("T")] = ord("A")
    V = 2 * n + 1
    base = np.concatenate([seq, np.full(n, ord("T"), np.uint8), [ord("A")]]).astype(np.uint8)
    pred_off, pred = [0], []
    for v in range(V):
        if v == 2 * n:
            pred += [n - 1, 2 * n - 1]
        elif v not in (0, n):
            pred += [v - 1]
        pred_off.append(len(pred))
    is_sink = np.zeros(V, np.uint8); is_sink[-1] = 1
    args = (np.array(pred_off, np.int32), np.array(pred, np.int32), base, is_sink, seq)
    eng.set_option("poa_ref_arith", 0)
    got = eng.poa_fill(*args)
    assert eng.last_status == 1 and "apart" in eng.last_warning()
    want = util.oracle_poa_fill(*args)
    assert got[0] == want[0] and np.array_equal(got[1], want[1]) and all(np.array_equal(g, w) for g, w in zip(got[2:], want[2:]))
    # default: the spread is noticed after the plain fill, the call is run again in the reference's arithmetic
    eng.set_option("poa_ref_arith", 1)
    got = eng.poa_fill(*args)
    assert eng.last_status == 0 and eng.last_warning() == ""
    want = util.oracle_poa_fill_i8(*args, 10)
    assert got[0] == want[0] and all(np.array_equal(g, w) for g, w in zip(got[1:], want[1:]))
    # ... and so is the walk (tsta_poa_align: the first run's step list is dropped, the second run's returned)
    want_steps, want_end = util.poa_walk(args[0], args[1], is_sink, want[1], want[2], want[3], want[4])
    steps, end_col, _ = eng.poa_align(*args)
    assert eng.last_status == 0
    assert ([(int(t["node"]), int(t["col"]), int(t["code"]), int(t["peek"])) for t in steps], end_col) == (want_steps, want_end)


def _walk_tuple(steps):
    return [(int(s["node"]), int(s["col"]), int(s["code"]), int(s["peek"])) for s in steps]


@pytest.mark.parametrize("step", [1, 2, 3, 4])
def test_walk_through_the_block_table_readme_graphs(eng, step):
    """K4t (poa_fill.hip): the POA walk crossing its column blocks by table look-up returns the steps of the step-by-step walk, on the
    four graphs of the README example (5 000 columns, 5 000 - 10 130 nodes), and the table is really in use."""
    z = np.load(os.path.join(util.GOLDEN_DIR, "poa_readme_step%d.npz" % step))
    args = (z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"])
    try:
        eng.set_option("walk_table", 0)
        want, want_end, last0 = eng.poa_align(*args)
        assert eng.stat("last_walk_jumps") == 0
        eng.set_option("walk_table", 1)
        got, got_end, last1 = eng.poa_align(*args)
        jumps = eng.stat("last_walk_jumps")
    finally:
        eng.set_option("walk_table", 1)
    assert got_end == want_end and np.array_equal(last0, last1)
    assert _walk_tuple(got) == _walk_tuple(want)
    assert jumps >= 0.8 * (len(z["seq"]) // 32 - 1), jumps
    if step == 4:                                            # ... and both equal the Python restatement of node_fuse's walk over the arrays
        best, last, src, es, fs = eng.poa_fill(*args)
        ref, ref_end = util.poa_walk(z["pred_off"], z["pred"], z["is_sink"], last, src, es, fs)
        assert (_walk_tuple(got), got_end) == (ref, ref_end)


def test_walk_through_the_block_table_random_dags(eng):
    """graphs whose walks leave the corridor, end inside a block, start away from the last node: every block the table cannot serve is
    walked step by step, the result is the same"""
    try:
        for seed, V, mp, n in [(21, 1200, 4, 700), (22, 3000, 6, 1500), (23, 5000, 3, 300), (24, 1500, 12, 2600), (25, 2048, 2, 96)]:
            pred_off, pred, base, is_sink, seq = random_dag(seed, V, mp, n)
            eng.set_option("walk_table", 0)
            want, want_end, _ = eng.poa_align(pred_off, pred, base, is_sink, seq)
            eng.set_option("walk_table", 2)
            got, got_end, _ = eng.poa_align(pred_off, pred, base, is_sink, seq)
            assert (_walk_tuple(got), got_end) == (_walk_tuple(want), want_end), (seed, V, mp, n)
    finally:
        eng.set_option("walk_table", 1)
