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
