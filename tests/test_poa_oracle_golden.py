"""Pins oracle/poa_oracle.c against dumps of the compiled reference's POA fill (tests/golden/make_poa_golden.py):
every trace byte of source/esource/fsource (small cases) or their sha256 (README example, medium), the sinks'
lastsorce, and the README known answers (README.md:26-40)."""
import glob
import hashlib
import os

import numpy as np
import pytest

from tests import util

FIXTURES = sorted(glob.glob(os.path.join(util.GOLDEN_DIR, "poa_*.npz")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(path):
    z = np.load(path)
    M, X, E, O = [int(v) for v in z["scoring"]]
    return z, dict(M=M, X=X, E=E, O=O)


def check_against_fixture(z, best, last, src, es, fs):
    sinks = np.nonzero(z["is_sink"])[0]
    assert np.array_equal(last[sinks], z["lastsorce"][sinks])
    # best sink: max lastsorce, ties -> lowest index (c-t-simd.c:737-750)
    assert best == sinks[np.argmax(z["lastsorce"][sinks])]
    if "source" in z.files:
        assert np.array_equal(src, z["source"])
        assert np.array_equal(es, z["esource"])
        assert np.array_equal(fs, z["fsource"])
    assert [sha(src), sha(es), sha(fs)] == list(z["sha"])


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[4:-4] for p in FIXTURES])
def test_poa_oracle_matches_reference_dump(path):
    z, sc = load(path)
    best, last, src, es, fs = util.oracle_poa_fill(z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"], **sc)
    check_against_fixture(z, best, last, src, es, fs)


def test_readme_known_answers():
    # README.md:28-39: poa_len and best-sink lastsorce per step
    want = [(5000, -5451), (6885, -3101), (8599, -1776), (10130, -870)]
    for k, (V, ls) in enumerate(want):
        z = np.load(os.path.join(util.GOLDEN_DIR, "poa_readme_step%d.npz" % (k + 1)))
        assert len(z["base"]) == V
        sinks = np.nonzero(z["is_sink"])[0]
        assert z["lastsorce"][sinks].max() == ls


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[4:-4] for p in FIXTURES])
def test_poa_oracle_i8_matches_reference_dump(path):
    # the restatement in the reference's own int8 representation (oracle/poa_oracle_i8.c) against the same dumps; the dumps were
    # made at the default -W 10 (make_poa_golden.py)
    z, sc = load(path)
    best, last, src, es, fs = util.oracle_poa_fill_i8(z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"], 10, **sc)
    check_against_fixture(z, best, last, src, es, fs)


def test_poa_oracle_i8_equals_plain_oracle_at_other_w():
    z, sc = load([f for f in FIXTURES if f.endswith("poa_medium_step2.npz")][0])
    args = (z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"])
    want = util.oracle_poa_fill(*args, **sc)
    for w in (1, 3, 16):                                         # (at -W 32 the reference's int8 lane sums wrap on this input: test_msa_host.py)
        got = util.oracle_poa_fill_i8(*args, w, **sc)
        assert got[0] == want[0] and all(np.array_equal(g, x) for g, x in zip(got[1:], want[1:])), w
