"""GPU parity: K2 (trace fill) + K3 (walker) through tsta_psa_align vs the reference-built golden vectors
(maxsorce + md5 of the exact output file bytes) and vs the oracle's trace().  Bit-exact."""
import hashlib
import os

import numpy as np
import pytest

import tsta_amd
from tests import util
from tests.test_oracle_golden import GOLDEN, case_inputs, output_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = tsta_amd.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("idx", range(len(GOLDEN)))
def test_golden_output_bytes(eng, idx):
    c = GOLDEN[idx]
    s1, s2 = case_inputs(c)
    ms, ra, rb = eng.psa_align(s1, s2, tsta_amd.Params(**c["scoring"]))
    assert ms == c["maxsorce"]
    out = output_bytes(ra, rb)
    assert len(out) == c["out_size"]
    assert hashlib.md5(out).hexdigest() == c["out_md5"]


@pytest.mark.parametrize("scoring", [dict(M=2, X=-5, E=-2, O=-4), dict(M=1, X=-1, E=-1, O=0),
                                     dict(M=10, X=-10, E=-3, O=-12), dict(M=2, X=-3, E=-2, O=-4)])
def test_random_and_mutated_vs_oracle(eng, scoring):
    cases = []
    for i, l in enumerate([1, 2, 5, 31, 64, 65, 127, 128, 129, 255, 257, 500, 777, 1300, 2100, 4000]):
        a = util.random_dna(3000 + i, l)
        cases.append((a, util.mutate(a, 3100 + i, 0.12)))
        cases.append((util.random_dna(3200 + i, l), util.random_dna(3300 + i, max(1, (l * 2) // 3))))
        cases.append((util.low_complexity(3400 + i, max(1, l // 2), 2), util.low_complexity(3400 + i, l, 2)))
    for s1, s2 in cases:
        ms, ra, rb = eng.psa_align(s1, s2, tsta_amd.Params(**scoring))
        oms, oend, oa, ob = util.oracle_align(s1, s2, **scoring)
        assert ms == oms
        assert ra == oa and rb == ob, (len(s1), len(s2), scoring)


def test_empty_and_single(eng):
    e = np.zeros(0, dtype=np.uint8)
    a = util.random_dna(5, 33)
    for s1, s2 in [(a, e), (e, a), (e, e), (a[:1], a[:1]), (a[:1], a[1:2])]:
        ms, ra, rb = eng.psa_align(s1, s2)
        oms, oend, oa, ob = util.oracle_align(s1, s2)
        assert (ms, ra, rb) == (oms, oa, ob)


def test_band_crossing_sizes(eng):
    # row counts around the band sizes (R=2: 128 rows, R=4: 256 rows from 16384 rows) and tile lengths
    for (l1, l2) in [(128, 128), (129, 128), (256, 255), (1024, 130), (3000, 257), (17000, 16500)]:
        a = util.random_dna(l1 * 7 + l2, l1)
        b = util.mutate(a, l1 + 13, 0.1)[:l2]
        ms, ra, rb = eng.psa_align(a, b)
        oms, oend, oa, ob = util.oracle_align(a, b)
        assert ms == oms and ra == oa and rb == ob, (l1, l2)


def test_full_size_100k_properties(eng):
    # BASELINE config #3: one 100 kbp x 100 kbp pair with traceback (10 GB packed trace in HBM).  The oracle's
    # 3 B/cell trace does not fit at this size; check size-independent properties instead:
    #  (1) removing gaps reproduces the inputs, (2) the path's affine score equals H(m,n) from the O(n) oracle,
    #  (3) maxsorce equals the oracle's.
    x = util.random_dna(777, 100_000)
    y = util.mutate(x, 778, 0.05)
    ms, ra, rb = eng.psa_align(x, y)
    la, lb = (x, y) if len(x) >= len(y) else (y, x)
    assert ra.replace(b"-", b"") == bytes(la) and rb.replace(b"-", b"") == bytes(lb)
    oms, oend = util.oracle_scores(x, y)
    assert ms == oms
    assert util.alignment_score(ra, rb) == oend


def test_full_size_100k_md5_matches_reference(eng):
    # BASELINE config #3 bit-exact at full size: the reference's own output file for this pair (tests/golden/psa_big_golden.json,
    # made by make_big_golden.py from oracle/_ref/TSTA_psa: 62 s and 30 GB of host memory there), with the single-launch
    # fill (default) and with the launch-per-tile pipeline
    import hashlib
    import json
    with open(os.path.join(util.GOLDEN_DIR, "psa_big_golden.json")) as f:
        g = json.load(f)
    x = util.random_dna(g["seed_x"], g["len_x"]); y = util.mutate(x, g["seed_y"], g["rate"])
    assert len(y) == g["len_y"]
    try:
        for persistent, lds, split in ((1, 1, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1)):   # split (default) / quad / memory-linked / launches
            eng.set_option("trace_persistent", persistent); eng.set_option("trace_lds", lds); eng.set_option("trace_split", split)
            ms, ra, rb = eng.psa_align(x, y)
            data = b">1\n" + ra + b"\n>2\n" + rb                      # what TSTA_psa writes (psa.c:512-516)
            assert ms == g["maxsorce"] and len(data) == g["size"]
            assert hashlib.md5(data).hexdigest() == g["md5"] and hashlib.sha256(data).hexdigest() == g["sha256"], persistent
    finally:
        eng.set_option("trace_persistent", 1); eng.set_option("trace_lds", 1); eng.set_option("trace_split", 1)


def test_single_launch_fill_vs_tile_pipeline(eng):
    # the persistent fill (sc1 hand-off between bands, tickets recycled) against the launch-per-tile pipeline and the oracle:
    # rows per lane 1 / 2 / 4, sizes around band and block boundaries; then MANY unpaired jobs in one group (more bands
    # than resident waves, uneven lengths: the hand-off under load)
    try:
        for R in (1, 2, 4):
            eng.set_option("trace_rows", R)
            for (l1, l2) in [(70, 64), (129, 128), (1000, 257), (3001, 2999), (9000, 8200)]:
                a = util.random_dna(l1 * 3 + l2 + R, l1); b = util.mutate(a, l1 + R, 0.08)[:l2]
                eng.set_option("trace_persistent", 1); got1 = eng.psa_align(a, b)      # R=2: four bands per workgroup linked through LDS
                eng.set_option("trace_split", 2); got3 = eng.psa_align(a, b)           # R=2: ... as solvers, the trace from emitter tiles (forced at any size)
                eng.set_option("trace_split", 0); got4 = eng.psa_align(a, b)
                eng.set_option("trace_split", 1)
                eng.set_option("trace_lds", 0); got2 = eng.psa_align(a, b)             # every hand-off through memory
                eng.set_option("trace_lds", 1)
                eng.set_option("trace_persistent", 0); got0 = eng.psa_align(a, b)
                o = util.oracle_align(a, b)
                assert got1 == got2 == got3 == got4 == got0 == (o[0], o[2], o[3]), (R, l1, l2)
        # the split form has two code paths for the substitution term: a byte table for pure-ACGT pairs, compare + select for any
        # other byte (lower case, N, ...): the same pair through both, and a pair that only the second can take
        eng.set_option("trace_rows", 0); eng.set_option("trace_split", 2)
        a = util.random_dna(4711, 5000); b = util.mutate(a, 4712, 0.07)
        o = util.oracle_align(a, b)
        assert eng.psa_align(a, b) == (o[0], o[2], o[3])
        eng.set_option("psa_perm", 0)
        assert eng.psa_align(a, b) == (o[0], o[2], o[3])
        eng.set_option("psa_perm", 1)
        a2 = a.copy(); a2[::97] = ord("N"); a2[5::211] = ord("a"); b2 = b.copy(); b2[3::89] = ord("N"); b2[7::301] = ord("t")
        o = util.oracle_align(a2, b2)
        assert eng.psa_align(a2, b2) == (o[0], o[2], o[3])
        eng.set_option("trace_split", 1)
        eng.set_option("trace_rows", 0); eng.set_option("trace_persistent", 1); eng.set_option("trace_packed", 0)
        pairs = []
        for i in range(48):
            a = util.random_dna(9100 + i, 2000 + 173 * (i % 11))
            pairs.append((a, util.mutate(a, 9200 + i, 0.1)))
        got = eng.psa_align_batch(pairs)
        eng.set_option("trace_persistent", 0)
        assert got == eng.psa_align_batch(pairs)
        for (ms, ra, rb), (a, b) in list(zip(got, pairs))[::7]:
            o = util.oracle_align(a, b)
            assert (ms, ra, rb) == (o[0], o[2], o[3])
    finally:
        eng.set_option("trace_rows", 0); eng.set_option("trace_persistent", 1); eng.set_option("trace_packed", 1); eng.set_option("trace_lds", 1)
        eng.set_option("trace_split", 1); eng.set_option("psa_perm", 1)


def test_rows_per_lane_4_variant(eng):
    # the R=4 instantiation of the trace kernels and a longer tile (defaults: R=2, T=64)
    try:
        eng.set_option("trace_rows", 4); eng.set_option("trace_tile", 128)
        for (l1, l2) in [(300, 257), (1500, 1400), (5000, 4900)]:
            a = util.random_dna(l1 + l2, l1); b = util.mutate(a, l1, 0.1)[:l2]
            got = eng.psa_align(a, b); o = util.oracle_align(a, b)
            assert got == (o[0], o[2], o[3]), (l1, l2)
    finally:
        eng.set_option("trace_rows", 0); eng.set_option("trace_tile", 0)


def test_fuzz_alignment_vs_oracle(eng):
    # random scoring sets (including zero / positive-ish corner values) x random pairs: full output strings vs trace()
    rng = np.random.default_rng(77)
    for trial in range(60):
        M = int(rng.integers(0, 9)); X = int(rng.integers(-12, max(M, 1))); E = int(rng.integers(-6, 1)); O = int(rng.integers(-14, 1))
        n = int(rng.integers(1, 1400)); m = int(rng.integers(1, 1400))
        a = util.random_dna(int(rng.integers(1 << 30)), n)
        kind = trial % 4
        if kind == 0:
            b = util.mutate(a, int(rng.integers(1 << 30)), 0.02 + 0.3 * rng.random())[:m]
        elif kind == 1:
            b = util.random_dna(int(rng.integers(1 << 30)), m)
        elif kind == 2:
            b = util.low_complexity(int(rng.integers(1 << 30)), m, int(rng.integers(1, 4)))
        else:
            b = np.concatenate([a[: n // 3], a[2 * n // 3:]])              # one long deletion
        if len(b) == 0:
            b = a[:1]
        sc = dict(M=M, X=X, E=E, O=O)
        got = eng.psa_align(a, b, tsta_amd.Params(**sc))
        o = util.oracle_align(a, b, **sc)
        assert got == (o[0], o[2], o[3]), (trial, sc, n, len(b))


# ---- tsta_psa_align_batch: many pairs in flight together (what a loop of TSTA_psa processes would produce) ----------

def _ragged_cases(seed, count, lo, hi):
    rng = np.random.default_rng(seed)
    cases = []
    for t in range(count):
        n = int(rng.integers(lo, hi)); m = int(rng.integers(lo, hi))
        a = util.random_dna(int(rng.integers(1 << 30)), n)
        kind = t % 4
        if kind == 0:
            b = util.mutate(a, int(rng.integers(1 << 30)), 0.1)[:m]
        elif kind == 1:
            b = util.random_dna(int(rng.integers(1 << 30)), m)
        elif kind == 2:
            b = util.low_complexity(int(rng.integers(1 << 30)), m, 2)
        else:
            b = np.concatenate([a[: n // 3], a[2 * n // 3:]])
        if len(b) == 0:
            b = a[:1]
        cases.append((a, b) if t % 3 else (b, a))
    return cases


def test_batch_golden_output_bytes(eng):
    by_params = {}
    for c in GOLDEN:
        by_params.setdefault(tuple(sorted(c["scoring"].items())), []).append(c)
    for key, cs in by_params.items():
        res = eng.psa_align_batch([case_inputs(c) for c in cs], tsta_amd.Params(**dict(key)))
        for c, (ms, ra, rb) in zip(cs, res):
            assert ms == c["maxsorce"]
            out = output_bytes(ra, rb)
            assert len(out) == c["out_size"] and hashlib.md5(out).hexdigest() == c["out_md5"], c["name"] if "name" in c else c


@pytest.mark.parametrize("scoring", [dict(M=2, X=-5, E=-2, O=-4), dict(M=10, X=-10, E=-3, O=-12), dict(M=1, X=-1, E=-1, O=0)])
def test_batch_ragged_vs_oracle(eng, scoring):
    e = np.zeros(0, dtype=np.uint8)
    cases = _ragged_cases(41, 40, 1, 2500) + [(e, util.random_dna(1, 50)), (util.random_dna(2, 7), e), (e, e)]
    res = eng.psa_align_batch(cases, tsta_amd.Params(**scoring))
    assert len(res) == len(cases)
    for (s1, s2), got in zip(cases, res):
        o = util.oracle_align(s1, s2, **scoring)
        assert got == (o[0], o[2], o[3]), (len(s1), len(s2), scoring)
    # identical to the one-pair entry point
    for (s1, s2), got in list(zip(cases, res))[:6]:
        assert got == eng.psa_align(s1, s2, tsta_amd.Params(**scoring))


def test_batch_many_bands_uses_tall_bands_and_long_tiles(eng):
    # enough bands in flight (>= 16 per CU at 256 rows each) switch the group to R=4 / T=128; results do not change,
    # also with the other instantiation forced and with groups split by a small workspace budget
    cases = _ragged_cases(43, 24, 3000, 9000) + _ragged_cases(44, 200, 1500, 2500)
    want = [util.oracle_align(s1, s2) for s1, s2 in cases[:24]]
    res = eng.psa_align_batch(cases)
    for w, got in zip(want, res):
        assert got == (w[0], w[2], w[3])
    try:
        eng.set_option("trace_rows", 2); eng.set_option("trace_tile", 128)
        res2 = eng.psa_align_batch(cases)
        # R=4: ~100 couples in the packed fill, split over 4 overlapping launch sequences / kept on one stream
        eng.set_option("trace_rows", 4); eng.set_option("trace_tile", 0)
        res4 = eng.psa_align_batch(cases)
        eng.set_option("trace_streams", 1)
        res4s = eng.psa_align_batch(cases)
    finally:
        eng.set_option("trace_rows", 0); eng.set_option("trace_tile", 0); eng.set_option("trace_streams", 0)
    assert res2 == res and res4 == res and res4s == res


def test_batch_split_into_groups_by_budget(eng, monkeypatch):
    cases = _ragged_cases(45, 30, 800, 3000)
    res = eng.psa_align_batch(cases)
    monkeypatch.setenv("TSTA_ALIGN_BUDGET_MB", "40")          # a few pairs per group
    res2 = eng.psa_align_batch(cases)
    assert res2 == res
    monkeypatch.setenv("TSTA_ALIGN_BUDGET_MB", "1")           # not even one pair fits: clean error, nothing leaked
    with pytest.raises(tsta_amd.TstaError):
        eng.psa_align_batch(cases)
    monkeypatch.delenv("TSTA_ALIGN_BUDGET_MB")
    assert eng.psa_align_batch(cases[:3]) == res[:3]


def test_batch_full_size_50k_properties(eng):
    # BASELINE-size pairs with traceback, several in flight: row lengths, gap-free projections and the score of the
    # alignment equal to H(m,n)'s defining recurrence value computed from the rows themselves
    L = 50_000
    cases = []
    for i in range(6):
        a = util.random_dna(900 + i, L)
        cases.append((a, util.mutate(a, 950 + i, 0.05)[:L] if i % 2 == 0 else util.random_dna(970 + i, L)))
    res = eng.psa_align_batch(cases)
    for (s1, s2), (ms, ra, rb) in zip(cases, res):
        assert len(ra) == len(rb)
        la, lb = (s1, s2) if len(s1) >= len(s2) else (s2, s1)
        assert ra.replace(b"-", b"") == bytes(la) and rb.replace(b"-", b"") == bytes(lb)
        oms, oend = util.oracle_scores(s1, s2)
        assert ms == oms and util.alignment_score(ra, rb) == oend
    one = eng.psa_align(*cases[0])
    assert one == res[0]
    # the ~15 GB of trace this batch left in the engine can be handed back; the next call simply allocates again
    eng.release_workspaces()
    assert eng.psa_align(*cases[1]) == res[1]


@pytest.mark.parametrize("scoring", [dict(M=2, X=-5, E=-2, O=-4), dict(M=1, X=-1, E=-1, O=0), dict(M=4, X=-6, E=-3, O=-5),
                                     dict(M=10, X=-10, E=-3, O=-12),       # too wide for 16-bit lanes at R=4: stays on K2
                                     dict(M=2, X=3, E=-2, O=-4)])          # X > M: stays on K2
def test_batch_packed_trace_fill(eng, scoring):
    # K2p (psa_trace_packed.hip): jobs of similar geometry are filled two per wave in packed 16-bit lanes; trace bytes,
    # walk and output must be identical to K2's.  Equal sizes, unequal couples, an odd job out, a non-ACGT byte,
    # identical / unrelated / low-complexity content, sizes around the band (256 rows) and block (64 columns) edges.
    rng = np.random.default_rng(91)
    cases = []
    for (n, m) in [(256, 256), (257, 255), (64, 64), (1, 1), (700, 300), (700, 290), (1300, 1290), (1290, 1300),
                   (2049, 2047), (2049, 2040), (3000, 512), (3000, 500), (5000, 4100), (4900, 4000), (777, 5)]:
        a = util.random_dna(int(rng.integers(1 << 30)), n)
        kind = len(cases) % 4
        if kind == 0:
            b = util.mutate(a, int(rng.integers(1 << 30)), 0.08)[:m]
        elif kind == 1:
            b = util.random_dna(int(rng.integers(1 << 30)), m)
        elif kind == 2:
            b = a[:m].copy()
        else:
            b = util.low_complexity(int(rng.integers(1 << 30)), m, 2)
        if len(b) < m:
            b = np.concatenate([b, util.random_dna(5, m - len(b))])
        cases.append((a, b) if len(cases) % 3 else (b, a))
    xn = cases[4][0].copy(); xn[100] = ord("N")
    cases.append((xn, cases[4][1]))
    want = [util.oracle_align(s1, s2, **scoring) for s1, s2 in cases]
    try:
        eng.set_option("trace_rows", 4)
        for packed, elide, perm in ((1, 1, 1), (1, 0, 1), (1, 1, 0), (1, 0, 0), (0, 1, 1)):
            eng.set_option("trace_packed", packed); eng.set_option("psa_elide", elide); eng.set_option("psa_perm", perm)
            for tile in (64, 256):
                eng.set_option("trace_tile", tile)
                res = eng.psa_align_batch(cases, tsta_amd.Params(**scoring))
                for w, got, (s1, s2) in zip(want, res, cases):
                    assert got == (w[0], w[2], w[3]), (packed, elide, perm, tile, len(s1), len(s2), scoring)
    finally:
        eng.set_option("trace_rows", 0); eng.set_option("trace_packed", 1); eng.set_option("trace_tile", 0)
        eng.set_option("psa_elide", 1); eng.set_option("psa_perm", 1)


def _structured_pair(seed, n, rate, big_del=0, big_ins=0):
    """a pair whose path drifts off the end-anchored diagonal: point mutations + one long deletion / insertion in the middle"""
    a = util.random_dna(seed, n)
    b = util.mutate(a, seed + 1, rate)
    mid = len(b) // 2
    if big_del:
        b = np.concatenate([b[:mid], b[mid + big_del:]])
    if big_ins:
        b = np.concatenate([b[:mid], util.random_dna(seed + 2, big_ins), b[mid:]])
    return a, b


@pytest.mark.parametrize("scoring", [dict(M=2, X=-5, E=-2, O=-4), dict(M=1, X=-1, E=-1, O=0), dict(M=10, X=-10, E=-3, O=-12)])
def test_walk_through_the_band_table_equals_the_step_walk(eng, scoring):
    """K3t (psa_trace.hip): the single-pair walk crossing its bands by table look-up emits the ops of the step-by-step walk, whatever part
    of the path the corridor covers: narrow corridors, long indels that carry the path out of it and back, both orientations."""
    p = tsta_amd.Params(**scoring)
    cases = [_structured_pair(9100, 3000, 0.08), _structured_pair(9110, 6000, 0.15), _structured_pair(9120, 5000, 0.05, big_del=700),
             _structured_pair(9130, 5000, 0.05, big_ins=900), _structured_pair(9140, 9000, 0.10, big_del=300, big_ins=300),
             (util.random_dna(9150, 2500), util.random_dna(9151, 4100)), (util.low_complexity(9160, 4000), util.low_complexity(9161, 3800))]
    try:
        for a, b in cases:
            for x, y in ((a, b), (b, a)):
                eng.set_option("walk_table", 0)
                want = eng.psa_align(x, y, p)
                assert eng.stat("last_walk_jumps") == 0
                for corridor in (256, 1024):
                    eng.set_option("walk_table", 2); eng.set_option("walk_corridor", corridor)
                    got = eng.psa_align(x, y, p)
                    assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
                    if min(len(x), len(y)) >= 2048 and corridor == 1024:
                        assert eng.stat("last_walk_jumps") > 0            # (the table is really in use)
        a, b = cases[1]
        eng.set_option("walk_table", 2); eng.set_option("walk_corridor", 1024)
        ms, ra, rb = eng.psa_align(a, b, p)
        o_ms, _, o_ra, o_rb = util.oracle_align(a, b, **scoring)
        assert (ms, ra, rb) == (o_ms, o_ra, o_rb)
    finally:
        eng.set_option("walk_table", 1); eng.set_option("walk_corridor", 1024)


def test_walk_table_is_the_default_on_the_100k_pair(eng):
    """config #3: nearly every band of the 100 kbp pair is crossed by look-up; the output is the reference's file either way
    (test_full_size_100k_md5_matches_reference runs with the default options)"""
    import json
    g = json.load(open(os.path.join(util.GOLDEN_DIR, "psa_big_golden.json")))
    x = util.random_dna(g["seed_x"], g["len_x"]); y = util.mutate(x, g["seed_y"], g["rate"])
    ms, ra, rb = eng.psa_align(x, y)
    jumps = eng.stat("last_walk_jumps")
    nb = (len(y) - 1) // 128
    assert jumps >= 0.9 * nb, (jumps, nb)          # (all but the bands entered in the middle of a vertical gap)
    try:
        eng.set_option("walk_table", 0)
        ms0, ra0, rb0 = eng.psa_align(x, y)
    finally:
        eng.set_option("walk_table", 1)
    assert (ms, ra, rb) == (ms0, ra0, rb0)
