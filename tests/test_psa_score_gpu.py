"""GPU parity: K1 (psa_score_batch_kernel) through the C-ABI vs the oracle and the reference-built golden vectors.
Bit-exact (integer scores)."""
import json
import os

import numpy as np
import pytest

import tsta_amd
from tests import util
from tests.test_oracle_golden import GOLDEN, case_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = tsta_amd.Engine(0)
    yield e
    e.close()


def test_golden_vectors(eng):
    # every golden case of the compiled reference, grouped by scoring set into one batch each
    by_params = {}
    for c in GOLDEN:
        by_params.setdefault(json.dumps(c["scoring"], sort_keys=True), []).append(c)
    for key, cs in by_params.items():
        sc = json.loads(key)
        pairs = [case_inputs(c) for c in cs]
        got = eng.psa_score_batch(pairs, tsta_amd.Params(**sc))
        want = np.array([c["maxsorce"] for c in cs], dtype=np.int32)
        assert np.array_equal(got, want), (sc, got.tolist(), want.tolist())


@pytest.mark.parametrize("scoring", [dict(M=2, X=-5, E=-2, O=-4), dict(M=10, X=-10, E=-3, O=-12),
                                     dict(M=1, X=-1, E=-1, O=0), dict(M=3, X=2, E=-1, O=-1)])
def test_ragged_batch_vs_oracle(eng, scoring):
    pairs = []
    lens = [1, 2, 3, 7, 63, 64, 65, 127, 128, 129, 511, 512, 513, 1000, 1025, 2047, 2049, 3000]
    for i, l in enumerate(lens):
        a = util.random_dna(500 + i, l)
        pairs.append((a, util.mutate(a, 900 + i, 0.08)))                  # related: non-degenerate maxsorce
        pairs.append((util.random_dna(700 + i, l), util.random_dna(800 + i, max(1, l // 3))))   # ragged, unrelated
        pairs.append((util.random_dna(300 + i, max(1, l // 5)), util.random_dna(400 + i, l)))   # shorter first: swap
    got = eng.psa_score_batch(pairs, tsta_amd.Params(**scoring))
    want = util.oracle_maxscore_batch(pairs, **scoring)
    assert np.array_equal(got, want)


def test_empty_and_degenerate(eng):
    e = np.zeros(0, dtype=np.uint8)
    a = util.random_dna(1, 100)
    pairs = [(a, e), (e, a), (e, e), (a[:1], a[:1]), (a, a)]
    got = eng.psa_score_batch(pairs)
    want = util.oracle_maxscore_batch(pairs)
    assert np.array_equal(got, want)
    assert got[0] == -100 and got[2] == -100 and got[4] == 200
    assert len(eng.psa_score_batch([])) == 0


def test_non_acgt_bytes_case_sensitive(eng):
    # psa.c:253 compares raw bytes: 'a' != 'A', 'N' is an ordinary character
    a = np.frombuffer(b"ACGTNNacgtACGT" * 20, dtype=np.uint8)
    b = np.frombuffer(b"ACGTNNACGTACGT" * 20, dtype=np.uint8)
    got = eng.psa_score_batch([(a, b)])
    assert got[0] == util.oracle_maxscore(a, b)


def test_more_pairs_than_resident_waves(eng):
    # > 8192 pairs: every persistent wave takes several tickets and reuses its boundary row
    pairs = []
    for i in range(9000):
        l = 20 + (i * 7) % 90
        a = util.random_dna(10_000 + i, l)
        pairs.append((a, util.mutate(a, 20_000 + i, 0.1)))
    got = eng.psa_score_batch(pairs)
    want = util.oracle_maxscore_batch(pairs)
    assert np.array_equal(got, want)


def test_config2_shape_subset_vs_oracle(eng):
    # BASELINE config #2 shape (5 kbp x 5 kbp) on a subset the oracle finishes in seconds, plus the mutated
    # companion set SURVEY 8(d) asks for (random DNA alone gives a degenerate maxsorce)
    pairs = [(util.random_dna(2 * p, 5000), util.random_dna(2 * p + 1, 5000)) for p in range(24)]
    pairs += [(util.random_dna(5000 + p, 5000), util.mutate(util.random_dna(5000 + p, 5000), 77 + p, 0.05))
              for p in range(24)]
    got = eng.psa_score_batch(pairs)
    want = util.oracle_maxscore_batch(pairs)
    assert np.array_equal(got, want)
    assert want[24:].min() > 1000            # the companion set really is non-degenerate


def test_full_size_property_identical_pairs(eng):
    # full 50 kbp size (BASELINE metric shape), size-independent property: aligning x with itself scores M*len
    # (every prefix diagonal is optimal and the maximum cell is H(n,n)); and score is symmetric under swap.
    x = util.random_dna(42, 50_000)
    y = util.mutate(x, 43, 0.05)
    got = eng.psa_score_batch([(x, x), (x, y), (y, x)])
    assert got[0] == 2 * 50_000
    assert got[1] == util.oracle_maxscore(x, y)
    assert got[2] == util.oracle_maxscore(y, x)


def _fixed_len(seq, n, seed):
    seq = np.asarray(seq, dtype=np.uint8)
    if len(seq) >= n:
        return seq[:n]
    return np.concatenate([seq, util.random_dna(seed, n - len(seq))])


@pytest.mark.parametrize("scoring", [dict(M=2, X=-5, E=-2, O=-4),          # packed R=16
                                     dict(M=10, X=-10, E=-3, O=-12),        # packed R=8 (range bound)
                                     dict(M=1, X=-1, E=-1, O=0),
                                     dict(M=5, X=-250, E=-1, O=-3),         # M-X = 255: edge of the packed select
                                     dict(M=100, X=-100, E=-30, O=-60),     # too wide for 16-bit lanes: int32 path
                                     dict(M=2, X=3, E=-2, O=-4)])           # X > M: int32 path
def test_packed_couples_equal_geometry(eng, scoring):
    # adjacent pairs with identical (n, m) are scored two-per-wave in packed 16-bit lanes (psa_score_packed.hip);
    # results must be identical to the int32 path / oracle, including related sequences with large positive scores
    pairs = []
    for i, (n, m) in enumerate([(64, 64), (65, 1), (1, 1), (700, 70), (1024, 1024), (1025, 1023), (2048, 2047),
                                (3000, 3000), (5000, 4096), (4097, 129)]):
        for j in range(2):                                           # the two members of the couple differ in content
            a = util.random_dna(9000 + 10 * i + j, n)
            kind = (i + j) % 3
            if kind == 0:
                b = _fixed_len(util.mutate(a, 9500 + 10 * i + j, 0.06), m, 77 + i)      # related
            elif kind == 1:
                b = util.random_dna(9700 + 10 * i + j, m)                                # unrelated
            else:
                b = _fixed_len(a[::-1].copy(), m, 5)                                      # reversed: poor alignment
            pairs.append((a, b) if j == 0 else (b, a))                                    # second member swapped
    want = util.oracle_maxscore_batch(pairs, **scoring)
    try:
        for mode in (32, 16, 8, 0):      # forced packed R=32 / 16 / 8 (falls back when the scoring does not fit) / int32 only
            eng.set_option("psa_packed", mode)
            got = eng.psa_score_batch(pairs, tsta_amd.Params(**scoring))
            assert np.array_equal(got, want), (mode, scoring, got.tolist(), want.tolist())
    finally:
        eng.set_option("psa_packed", -1)


@pytest.mark.parametrize("scoring", [dict(M=2, X=-5, E=-2, O=-4), dict(M=1, X=-1, E=-1, O=0), dict(M=3, X=-1, E=0, O=-7),
                                     dict(M=4, X=-6, E=-3, O=-5), dict(M=2, X=-5, E=-9, O=-1),
                                     dict(M=3, X=-4, E=-3, O=-4)])          # R=32 with the elided frame 600 below the 16-bit bound
def test_packed_max_elision(eng, scoring):
    # Blocks whose entering scores prove that maxsorce cannot rise there run without the running max, in a row-shifted
    # frame (psa_score_packed.hip: pstep_elided).  The decision is data dependent: cover maxima reached early, late,
    # in the middle, never (floor -100), and pairs hovering around the bound; with and without the elision.
    n = 7000
    x = util.random_dna(61, n); y = util.random_dna(62, n); z = util.random_dna(63, n)
    A = np.full(n, ord("A"), np.uint8)
    pairs = [(x, y), (x, x.copy()), (y, util.mutate(y, 64, 0.03)[:n]),
             (np.concatenate([x[:1500], y[1500:]]), np.concatenate([x[:1500], z[1500:]])),       # maximum early, then decay
             (np.concatenate([y[:4500], x[4500:]]), np.concatenate([z[:4500], x[4500:]])),       # climbs back late
             (np.concatenate([y[:2500], x[2500:4500], y[4500:]]), np.concatenate([z[:2500], x[2500:4500], z[4500:]])),
             (np.concatenate([x[:3000], A[:3000]]), np.concatenate([A[:3000], x[:3000]])),       # long forced gaps
             (x, np.concatenate([x[3500:], x[:3500]])),                                           # off-diagonal identity
             (x[:6000], util.mutate(x, 65, 0.25)[:4200]), (z, util.mutate(z, 66, 0.12)[:n]),
             (util.low_complexity(67, n, 2), util.low_complexity(68, n, 2)), (A, A.copy())]
    # pure-ACGT couples take the byte-permute mismatch lookup; a single other byte (here: n / lower case) sends the
    # couple down the generic compare, and the reference compares bytes case-sensitively (psa.c:253)
    xn = x.copy(); xn[5000] = ord("N"); yl = y.copy(); yl[10:20] += 32
    pairs += [(xn, x.copy()), (x, yl), (xn, xn.copy()), (yl, util.mutate(y, 69, 0.05)[:n])]
    want = util.oracle_maxscore_batch(pairs, **scoring)
    try:
        for mode in (32, 16, 8):
            # (ffree: scorings with X >= -2|E| run the elided blocks of pure-ACGT couples with both gap extensions free)
            for elide, perm, ffree in ((1, 1, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1), (0, 0, 0)):
                eng.set_option("psa_packed", mode)
                eng.set_option("psa_elide", elide)
                eng.set_option("psa_perm", perm)
                eng.set_option("psa_ffree", ffree)
                got = eng.psa_score_batch(pairs, tsta_amd.Params(**scoring))
                assert np.array_equal(got, want), (mode, elide, perm, ffree, scoring, got.tolist(), want.tolist())
    finally:
        eng.set_option("psa_packed", -1)
        eng.set_option("psa_elide", 1)
        eng.set_option("psa_perm", 1)
        eng.set_option("psa_ffree", 1)


def test_packed_adversarial_content(eng):
    n = 6000
    A = np.full(n, ord("A"), np.uint8); C = np.full(n, ord("C"), np.uint8)
    x = util.random_dna(31, n)
    lc = util.low_complexity(32, n, 2)
    pairs = [(A, A.copy()), (A, C),                 # all-match diagonal / all-mismatch
             (x, x.copy()), (x, x[::-1].copy()),
             (lc, lc.copy()), (lc, np.roll(lc, 1)),
             (np.concatenate([x[:3000], A[:3000]]), np.concatenate([A[:3000], x[:3000]])),   # long forced gaps
             (x, np.concatenate([x[3000:], x[:3000]]))]
    try:
        eng.set_option("psa_packed", 32)
        # the last two sit just inside the 16-bit range bound of the packed kernel for R=16 and for R=8
        for sc in (dict(M=2, X=-5, E=-2, O=-4), dict(M=3, X=-1, E=0, O=-7), dict(M=5, X=-4, E=-6, O=-8),
                   dict(M=10, X=-9, E=-10, O=-12)):
            got = eng.psa_score_batch(pairs, tsta_amd.Params(**sc))
            want = util.oracle_maxscore_batch(pairs, **sc)
            assert np.array_equal(got, want), (sc, got.tolist(), want.tolist())
    finally:
        eng.set_option("psa_packed", -1)


def test_packed_and_int32_paths_agree_at_50k(eng, monkeypatch):
    # full BASELINE length: related 50 kbp couples, packed path vs oracle (O(n) memory)
    x = util.random_dna(4242, 50_000)
    y = _fixed_len(util.mutate(x, 4243, 0.05), 50_000, 1)
    z = _fixed_len(util.mutate(x, 4244, 0.15), 50_000, 2)
    want = [util.oracle_maxscore(x, y), util.oracle_maxscore(x, z)]
    try:
        for mode in (32, 16, 0):
            eng.set_option("psa_packed", mode)
            got = eng.psa_score_batch([(x, y), (x, z)])
            assert got.tolist() == want, mode
    finally:
        eng.set_option("psa_packed", -1)


def test_auto_mode_large_batch_uses_both_kernels(eng):
    # > 1024 pairs: auto mode packs the equal-geometry couples and leaves the rest to the int32 kernel in the same call
    pairs = []
    for i in range(1300):
        n = 200 + (i // 2) % 7 * 31                     # couples (2c, 2c+1) share n; every 5th couple is broken on purpose
        a = util.random_dna(50_000 + i, n)
        b = _fixed_len(util.mutate(a, 60_000 + i, 0.1), n - (3 if i % 10 == 9 else 0), i)
        pairs.append((a, b))
    got = eng.psa_score_batch(pairs)
    want = util.oracle_maxscore_batch(pairs)
    assert np.array_equal(got, want)


def test_host_api_groups_equal_geometry(eng):
    # the host-pointer form orders pairs by geometry so interleaved equal-geometry pairs are packed; output order is the caller's
    pairs = []
    for i in range(1500):
        n = [300, 450, 300, 610][i % 4]
        a = util.random_dna(80_000 + i, n)
        pairs.append((a, _fixed_len(util.mutate(a, 81_000 + i, 0.08), n - (i % 4), i)) if i % 3 else (_fixed_len(a[::-1].copy(), n - (i % 4), 3), a))
    got = eng.psa_score_batch(pairs)
    want = util.oracle_maxscore_batch(pairs)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("scoring", [dict(M=2, X=-5, E=-2, O=-4), dict(M=10, X=-10, E=-3, O=-12)])
def test_packed_couples_of_unequal_size(eng, scoring):
    # pairs of different size may share a wave (the smaller is padded to the larger and masked): every combination of
    # longer/shorter columns and rows, around the band / block boundaries, related and unrelated content
    dims = [((1024, 1024), (1000, 900)), ((2048, 1030), (2047, 1024)), ((700, 64), (690, 65)), ((3000, 2900), (2800, 2950 - 200)),
            ((129, 128), (128, 127)), ((5000, 4096), (4990, 4097 - 97)), ((64, 1), (63, 2)), ((2000, 2000), (1999, 1)),
            ((4097, 1100), (4000, 1024))]
    pairs = []
    for i, ((n1, m1), (n2, m2)) in enumerate(dims):
        for j, (n, m) in enumerate(((n1, m1), (n2, m2))):
            a = util.random_dna(91_000 + 10 * i + j, n)
            b = _fixed_len(util.mutate(a, 92_000 + 10 * i + j, 0.07), m, i) if (i + j) % 2 == 0 else util.random_dna(93_000 + 10 * i + j, m)
            pairs.append((a, b) if (i % 3) else (b, a))
    want = util.oracle_maxscore_batch(pairs, **scoring)
    try:
        for mode in (32, 16, 8):
            eng.set_option("psa_packed", mode)
            got = eng.psa_score_batch(pairs, tsta_amd.Params(**scoring))
            assert np.array_equal(got, want), (mode, scoring, got.tolist(), want.tolist())
    finally:
        eng.set_option("psa_packed", -1)


def test_set_option_rejects_unknown(eng):
    with pytest.raises(tsta_amd.TstaError):
        eng.set_option("no_such_knob", 1)
    with pytest.raises(tsta_amd.TstaError):
        eng.set_option("psa_packed", 5)


def test_config2_full_size_all_pairs(eng):
    # BASELINE config #2 at full size: 1024 pairs of 5 kbp x 5 kbp random DNA, every score checked against the oracle
    pairs = [(util.random_dna(2 * p, 5000), util.random_dna(2 * p + 1, 5000)) for p in range(1024)]
    got = eng.psa_score_batch(pairs)
    want = util.oracle_maxscore_batch(pairs, threads=os.cpu_count() or 8)
    assert np.array_equal(got, want)


def test_config5_shape_sample(eng):
    # BASELINE config #5 shape (50 kbp x 50 kbp, packed kernel in auto mode since > 1024 pairs): 1100 pairs on the GPU,
    # a sample of them against the O(n)-memory oracle, plus an in-batch duplicate as a size-independent consistency check
    base = [(util.random_dna(7000 + 2 * p, 50_000), util.random_dna(7001 + 2 * p, 50_000)) for p in range(8)]
    x = util.random_dna(4242, 50_000)
    rel = (x, _fixed_len(util.mutate(x, 4243, 0.05), 50_000, 1))
    pairs = [base[p % 8] for p in range(1098)] + [rel, rel]
    got = eng.psa_score_batch(pairs)
    for p in range(8):
        assert np.all(got[p:1098:8] == got[p])                     # identical inputs, identical outputs in every slot
    for p in (0, 5):
        assert got[p] == util.oracle_maxscore(*base[p])
    assert got[1098] == got[1099] == util.oracle_maxscore(*rel)


def test_error_paths(eng):
    import ctypes as C
    L = tsta_amd.lib()
    prm = tsta_amd.Params()
    one = np.zeros(1, dtype=np.int64); l = np.array([5], dtype=np.int32); out = np.zeros(1, dtype=np.int32)
    seq = util.random_dna(1, 8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    # sequence range outside the buffer
    off_bad = np.array([6], dtype=np.int64)
    assert L.tsta_psa_score_batch(eng._h, C.byref(prm), 1, p(seq), 8, p(off_bad), p(l), p(one), p(l), p(out)) == -1
    # null pointers / negative counts
    assert L.tsta_psa_score_batch(eng._h, C.byref(prm), 1, None, 8, p(one), p(l), p(one), p(l), p(out)) == -1
    assert L.tsta_psa_score_batch(eng._h, C.byref(prm), -1, p(seq), 8, p(one), p(l), p(one), p(l), p(out)) == -1
    assert L.tsta_psa_score_batch(None, C.byref(prm), 1, p(seq), 8, p(one), p(l), p(one), p(l), p(out)) == -1
    ms = C.c_int32(); ra = C.c_void_p(); rb = C.c_void_p(); ln = C.c_int64()
    # longer than the supported maximum (2^24)
    assert L.tsta_psa_align(eng._h, C.byref(prm), p(seq), (1 << 24) + 1, p(seq), 4, C.byref(ms), C.byref(ra), C.byref(rb),
                            C.byref(ln)) == -5
    assert L.tsta_psa_align(eng._h, C.byref(prm), p(seq), -3, p(seq), 4, C.byref(ms), C.byref(ra), C.byref(rb), C.byref(ln)) == -1
    assert b"invalid" in L.tsta_strerror(-1)


def test_few_long_pairs_use_the_pipelined_single_pair_fill(eng):
    # <= 64 pairs, each >= 2e8 cells: scored by the multi-wave single-pair fill (psa_trace.hip, TRACE=false)
    x = util.random_dna(555, 21_000)
    pairs = [(x, _fixed_len(util.mutate(x, 556, 0.06), 15_000, 3)),
             (util.random_dna(557, 14_500), util.random_dna(558, 20_011)),        # shorter first: swapped
             (x, x.copy())]
    got = eng.psa_score_batch(pairs)
    want = util.oracle_maxscore_batch(pairs, threads=3)
    assert np.array_equal(got, want)
    for sc in (dict(M=10, X=-10, E=-3, O=-12), dict(M=1, X=-1, E=-1, O=0)):
        got = eng.psa_score_batch(pairs[:2], tsta_amd.Params(**sc))
        assert np.array_equal(got, util.oracle_maxscore_batch(pairs[:2], threads=2, **sc))


def test_fuzz_packed_vs_int32_vs_oracle(eng):
    # random scoring sets inside the packed kernel's domain x random ragged batches: packed R=16, packed R=8 and the int32
    # kernel must agree on every pair (GPU vs GPU, so the batches can be big); a slice is also checked against the oracle
    rng = np.random.default_rng(20241003)
    try:
        for trial in range(40):
            M = int(rng.integers(0, 9)); X = int(rng.integers(-12, min(M, 1))); E = int(rng.integers(-6, 1)); O = int(rng.integers(-14, 1))
            prm = tsta_amd.Params(M, X, E, O)
            pairs = []
            for k in range(48):
                n = int(rng.integers(1, 2600)); m = int(rng.integers(1, 2600))
                a = util.random_dna(int(rng.integers(1 << 30)), n)
                kind = k % 4
                if kind == 0:
                    b = _fixed_len(util.mutate(a, int(rng.integers(1 << 30)), 0.03 + 0.3 * rng.random()), m, k)
                elif kind == 1:
                    b = util.random_dna(int(rng.integers(1 << 30)), m)
                elif kind == 2:
                    b = util.low_complexity(int(rng.integers(1 << 30)), m, int(rng.integers(1, 5)))
                else:
                    b = _fixed_len(a, m, k)
                pairs.append((a, b))
            res = {}
            eng.set_option("psa_f16max", 2 if trial % 2 else 1)     # every other trial: the binary16-order kernel whatever the host's estimate says
            for mode in (0, 32, 16, 8):
                eng.set_option("psa_packed", mode)
                res[mode] = eng.psa_score_batch(pairs, prm)
            assert np.array_equal(res[0], res[32]), (trial, M, X, E, O)
            assert np.array_equal(res[0], res[16]), (trial, M, X, E, O)
            assert np.array_equal(res[0], res[8]), (trial, M, X, E, O)
            eng.set_option("psa_packed", -1)
            for bands in (8, 16):                      # K1b, the band pipeline, forced
                eng.set_option("psa_bands", bands)
                assert np.array_equal(res[0], eng.psa_score_batch(pairs, prm)), (trial, bands, M, X, E, O)
            eng.set_option("psa_bands", -1)
            if trial % 8 == 0:
                assert np.array_equal(res[0], util.oracle_maxscore_batch(pairs, M=M, X=X, E=E, O=O)), (trial, M, X, E, O)
    finally:
        eng.set_option("psa_packed", -1); eng.set_option("psa_bands", -1); eng.set_option("psa_f16max", 1)


@pytest.mark.parametrize("scoring", [dict(M=2, X=-5, E=-2, O=-4), dict(M=1, X=-1, E=-1, O=0), dict(M=4, X=-6, E=-3, O=-5),
                                     dict(M=10, X=-10, E=-3, O=-12)])
def test_band_pipeline_for_small_batches(eng, scoring):
    # K1b (psa_score_bands_kernel): a batch too small to fill the chip runs the bands of every couple concurrently.
    # Equal and unequal couples, an odd pair out, pairs shorter than one band, sizes around band (512 / 1024 rows) and
    # block (64 columns) edges, related / unrelated / identical content; forced R=8 / R=16 / off, two tile lengths.
    rng = np.random.default_rng(123)
    pairs = []
    for (n, m) in [(1024, 1024), (1025, 1023), (2048, 2047), (513, 511), (5000, 5000), (5000, 4990), (7000, 3000),
                   (6900, 3100), (3000, 64), (3000, 60), (9000, 8900), (8800, 9000), (100, 90), (1500, 1500), (4097, 4095),
                   (8000, 100), (7900, 7000)]:     # the last two are neighbours in geometry order but too unlike to share a wave: K1 takes them
        a = util.random_dna(int(rng.integers(1 << 30)), n)
        kind = len(pairs) % 3
        if kind == 0:
            b = _fixed_len(util.mutate(a, int(rng.integers(1 << 30)), 0.07), m, 3)
        elif kind == 1:
            b = util.random_dna(int(rng.integers(1 << 30)), m)
        else:
            b = _fixed_len(a.copy(), m, 4)
        pairs.append((a, b) if len(pairs) % 2 else (b, a))
    want = util.oracle_maxscore_batch(pairs, **scoring)
    try:
        for mode in (-1, 8, 16, 0):
            eng.set_option("psa_bands", mode)
            for tile, elide, perm in ((0, 1, 1), (128, 1, 1), (0, 0, 1), (0, 1, 0), (128, 0, 0)):
                eng.set_option("trace_tile", tile); eng.set_option("psa_elide", elide); eng.set_option("psa_perm", perm)
                got = eng.psa_score_batch(pairs, tsta_amd.Params(**scoring))
                assert np.array_equal(got, want), (mode, tile, elide, perm, scoring, got.tolist(), want.tolist())
    finally:
        eng.set_option("psa_bands", -1); eng.set_option("trace_tile", 0); eng.set_option("psa_elide", 1); eng.set_option("psa_perm", 1)


def test_band_pipeline_few_long_pairs(eng):
    # a handful of 30 kbp pairs: previously one wave each (or one pair at a time); now every band of every pair in flight
    pairs = []
    for i in range(7):
        a = util.random_dna(4000 + i, 30_000 - 17 * i)
        pairs.append((a, util.mutate(a, 4100 + i, 0.04)[:29_000] if i % 2 else util.random_dna(4200 + i, 29_500)))
    got = eng.psa_score_batch(pairs)
    want = util.oracle_maxscore_batch(pairs)
    assert np.array_equal(got, want)


def test_2bit_packed_input(eng):
    # tsta_psa_score_batch_2bit: same scores as the byte form on the same (pure ACGT) sequences; lengths that are not
    # multiples of 4, empty sequences, a batch large enough for the packed kernel and one small enough for the band pipeline
    e = np.zeros(0, dtype=np.uint8)
    pairs = [(util.random_dna(70 + i, 1 + 37 * i), util.mutate(util.random_dna(70 + i, 1 + 37 * i), 170 + i, 0.1)) for i in range(40)]
    pairs += [(e, util.random_dna(1, 9)), (util.random_dna(2, 5), e), (util.random_dna(3, 3001), util.random_dna(4, 2999))]
    assert np.array_equal(eng.psa_score_batch_2bit(pairs), util.oracle_maxscore_batch(pairs))
    big = [(util.random_dna(500 + i, 300 + i % 7), util.random_dna(900 + i, 290 + i % 5)) for i in range(1200)]
    assert np.array_equal(eng.psa_score_batch_2bit(big), eng.psa_score_batch(big))
    with pytest.raises(tsta_amd.TstaError):
        eng.psa_score_batch_2bit([(np.frombuffer(b"ACGN", dtype=np.uint8), util.random_dna(1, 4))])


def test_unequal_couples_garbage_stays_in_range(eng):
    # Regression (found by tests/tools/soak_score_paths.py): in a couple whose LOW half has fewer columns, that half is
    # garbage beyond its end; if the garbage leaves the 16-bit window, the 32-bit adds of the packed step carry into
    # the other pair's half (here: +1 on pair 8 with free gaps, M=4 X=-7 E=0 O=0).  Every packed path vs the int32
    # kernel on the batch that exposed it, and the affected pair against the oracle.
    rng = np.random.default_rng(12)
    M = int(rng.integers(0, 7)); X = int(rng.integers(-9, min(M, 1))); E = int(rng.integers(-5, 1)); O = int(rng.integers(-12, 1))
    assert (M, X, E, O) == (4, -7, 0, 0)
    prm = tsta_amd.Params(M, X, E, O)
    pairs = []
    for k in range(24):
        n = int(rng.integers(6000, 26000)); m = int(rng.integers(6000, 26000))
        a = util.random_dna(int(rng.integers(1 << 30)), n)
        kind = k % 6
        if kind == 0: b = util.mutate(a, int(rng.integers(1 << 30)), 0.02 + 0.2 * rng.random())[:m]
        elif kind == 1: b = util.random_dna(int(rng.integers(1 << 30)), m)
        elif kind == 2: b = util.low_complexity(int(rng.integers(1 << 30)), m, int(rng.integers(1, 4)))
        elif kind == 3: b = np.concatenate([a[: n // 3], a[2 * n // 3:]])
        elif kind == 4: b = np.concatenate([util.random_dna(5, m // 2), a[n // 2:]])[:m]
        else: b = a[::-1].copy()[:m]
        pairs.append((a, b))
    try:
        eng.set_option("psa_packed", 0); eng.set_option("psa_bands", 0)
        ref = eng.psa_score_batch(pairs, prm)
        assert ref[8] == util.oracle_maxscore(pairs[8][0], pairs[8][1], M=M, X=X, E=E, O=O)
        for packed, bands in ((-1, 16), (-1, 8), (32, -1), (16, -1), (8, -1), (-1, -1)):
            eng.set_option("psa_packed", packed); eng.set_option("psa_bands", bands)
            assert np.array_equal(eng.psa_score_batch(pairs, prm), ref), (packed, bands)
        # the same couples with traceback (K2p shares the packed tile)
        eng.set_option("psa_packed", -1); eng.set_option("psa_bands", -1); eng.set_option("trace_rows", 4)
        sub = [pairs[23], pairs[8], pairs[18], pairs[16]]
        res = eng.psa_align_batch(sub, prm)
        eng.set_option("trace_packed", 0)
        assert res == eng.psa_align_batch(sub, prm)
    finally:
        eng.set_option("psa_packed", -1); eng.set_option("psa_bands", -1); eng.set_option("trace_rows", 0); eng.set_option("trace_packed", 1)


def test_overlapped_upload_sub_batches(eng):
    # host-pointer form, batch worth >= 2 balanced launches (4 couples per SIMD each): cut into sub-batches whose uploads overlap the previous one's
    # kernels (psa_api.hip).  Ragged pairs, some shared scoring-domain edge; scores = oracle, and the same with the overlap off.
    n = 2 * 256 * 4 * 4 * 2 + 77
    pairs = []
    for i in range(n):
        l = 30 + (i * 37) % 150
        a = util.random_dna(70000 + i, l)
        pairs.append((a, util.mutate(a, 90000 + i, 0.1)) if i % 3 else (a, util.random_dna(80000 + i, max(1, l // 2))))
    want = util.oracle_maxscore_batch(pairs)
    got = eng.psa_score_batch(pairs)
    assert eng.stat("last_sub_batches") >= 2
    assert np.array_equal(got, want)
    eng.set_option("psa_overlap", 0)
    try:
        got0 = eng.psa_score_batch(pairs)
        assert eng.stat("last_sub_batches") == 1 and np.array_equal(got0, want)
    finally:
        eng.set_option("psa_overlap", 1)
    # sequences shared between pairs (every record against every record): the sub-batches' byte ranges overlap -> one upload
    seqs = np.concatenate([p[0] for p in pairs[:100]])
    lens = np.array([len(p[0]) for p in pairs[:100]], dtype=np.int32)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    ii, jj = np.meshgrid(np.arange(100), np.arange(100), indexing="ij")
    ii = np.tile(ii.ravel(), 1); jj = jj.ravel()
    got2 = eng.psa_score_batch_arrays(seqs, offs[ii], lens[ii], offs[jj], lens[jj])
    assert eng.stat("last_sub_batches") == 1
    sel = [0, 57, 4242, 9999]
    assert [int(got2[k]) for k in sel] == [util.oracle_maxscore(pairs[ii[k]][0], pairs[jj[k]][0]) for k in sel]


@pytest.mark.parametrize("scoring", [dict(M=2, X=-5, E=-2, O=-4), dict(M=2, X=-4, E=-4, O=-2), dict(M=3, X=-9, E=-7, O=-5),
                                     dict(M=6, X=-20, E=-3, O=-12)])
def test_f16_order_maximum_every_mode(eng, scoring):
    # the packed kernel with its scores kept as positive-normal binary16 bit patterns (option "psa_f16max"): steady blocks whose
    # operands provably stay in range take max3 as one v_pk_maximum3_f16 (2: always that kernel; 1: where the host's estimate allows),
    # the same kernel without the fast step (3), every couple handed to the int32 kernel through the redo flags (4), and the
    # round-3 kernel (0).  Content that spreads a band's scores:
    # identical sequences (rows far below the diagonal lie 64 R |E| under the band's top row), unrelated ones, long indels, couples of
    # unequal size, lengths around the band / block boundaries.
    pairs = []
    for i, n in enumerate([64, 65, 300, 1023, 1024, 1025, 2047, 2048, 2049, 2500, 4096, 5000, 5000, 6200]):
        a = util.random_dna(310_000 + i, n)
        pairs.append((a, a.copy()))                                                        # identical
        pairs.append((a, util.mutate(a, 320_000 + i, 0.1)))                                # related
        pairs.append((a, util.random_dna(330_000 + i, max(1, n - (i % 5) * 7))))           # unrelated, slightly ragged
        pairs.append((a, np.concatenate([a[:n // 3], a[n // 3 + min(n // 4, 400):]])))     # one long gap
        pairs.append((np.full(n, ord("A"), np.uint8), np.full(max(1, n - 3 * (i % 4)), ord("A"), np.uint8)))   # homopolymer: flat H'
    want = util.oracle_maxscore_batch(pairs, **scoring)
    prm = tsta_amd.Params(**scoring)
    try:
        for packed in (32, 16, 8):
            eng.set_option("psa_packed", packed)
            for mode in (2, 1, 3, 4, 0):
                eng.set_option("psa_f16max", mode)
                for elide in (1, 0):
                    eng.set_option("psa_elide", elide)
                    got = eng.psa_score_batch(pairs, prm)
                    assert np.array_equal(got, want), (scoring, packed, mode, elide, np.nonzero(got != want)[0][:8].tolist())
        eng.set_option("psa_packed", -1)
        for bands in (16, 8):                                   # the band pipeline (K1b) shares the step and the range check
            eng.set_option("psa_bands", bands)
            for mode in (2, 1, 3, 4, 0):
                eng.set_option("psa_f16max", mode)
                got = eng.psa_score_batch(pairs, prm)
                assert np.array_equal(got, want), (scoring, "bands", bands, mode, np.nonzero(got != want)[0][:8].tolist())
    finally:
        eng.set_option("psa_packed", -1); eng.set_option("psa_bands", -1); eng.set_option("psa_f16max", 1); eng.set_option("psa_elide", 1)


def test_f16_order_maximum_keeps_ordinary_batches_on_the_packed_kernels(eng):
    # no couple of ordinary inputs may be handed to the int32 kernel (that would be exact, and three times slower): ragged related /
    # unrelated pairs through K1p (forced) and through the band pipeline, default scoring
    pairs = []
    for i in range(300):
        n = 500 + (i * 131) % 4000
        a = util.random_dna(370_000 + i, n)
        pairs.append((a, util.mutate(a, 380_000 + i, 0.08)) if i % 2 else (a, util.random_dna(390_000 + i, max(1, n - i % 50))))
    want = util.oracle_maxscore_batch(pairs)
    try:
        for packed, bands in ((32, -1), (16, -1), (-1, 16), (-1, 8)):
            eng.set_option("psa_packed", packed); eng.set_option("psa_bands", bands)
            assert np.array_equal(eng.psa_score_batch(pairs), want)
            assert eng.stat("psa_redo_pairs") == 0, (packed, bands)
        eng.set_option("psa_f16max", 4); eng.set_option("psa_packed", 32); eng.set_option("psa_bands", -1)
        assert np.array_equal(eng.psa_score_batch(pairs), want)
        assert eng.stat("psa_redo_pairs") > 0                      # (the counter does see a redo)
    finally:
        eng.set_option("psa_packed", -1); eng.set_option("psa_bands", -1); eng.set_option("psa_f16max", 1)


def test_f16_order_maximum_worst_case_spread_falls_back(eng):
    # scores that DO leave the binary16-pattern range inside a band (|E| = 7: 2 048 rows below the diagonal of identical sequences lie
    # 14 336 under the band's top row) and ones that leave the 16-bit window around 0x4000 altogether (|E| = 12 at 32 rows per lane):
    # forced into the F3 instantiation (mode 2), results must still be the oracle's
    pairs = []
    for i, n in enumerate([2100, 4200, 5000, 9000]):
        a = util.random_dna(340_000 + i, n)
        pairs += [(a, a.copy()), (a, util.mutate(a, 350_000 + i, 0.03)), (a, util.random_dna(360_000 + i, n))]
    try:
        eng.set_option("psa_packed", 32)
        # (the last two: the exact column-0 values of a 2 048-row band do not fit 16 bits around 0x4000 — 2 048 x 8 and x 9 below the
        #  band's top row — while the host's symmetric window bound still admits 32 rows per lane)
        for scoring in (dict(M=2, X=-5, E=-7, O=-4), dict(M=1, X=-3, E=-12, O=0), dict(M=5, X=-5, E=-1, O=-30),
                        dict(M=0, X=-3, E=-8, O=0), dict(M=1, X=-2, E=-9, O=0)):
            want = util.oracle_maxscore_batch(pairs, **scoring)
            for mode in (2, 0):
                eng.set_option("psa_f16max", mode)
                for packed, bands in ((32, -1), (-1, 16)):
                    eng.set_option("psa_packed", packed); eng.set_option("psa_bands", bands)
                    got = eng.psa_score_batch(pairs, tsta_amd.Params(**scoring))
                    assert np.array_equal(got, want), (scoring, mode, packed, bands, np.nonzero(got != want)[0][:8].tolist())
    finally:
        eng.set_option("psa_packed", -1); eng.set_option("psa_bands", -1); eng.set_option("psa_f16max", 1)
