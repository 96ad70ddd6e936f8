"""Row P3d: where is TSTA_psa's output the plain affine-gap DP the engine computes, and does the engine say so where it is not?

The reference keeps row()'s quantities as wrapping int8 differences with MIN = -100 as minus infinity (psa.c:110,219-232,
244-246,285-327): outside a scoring domain its output depends on the SIMD width of the build and on -W.  The closed form of that
domain is `tsta_psa_reference_exact` (include/tsta_engine.h); tests/golden/psa_domain_golden.json holds what four builds / -W
settings of the compiled reference print on a scoring grid across every edge of it (make_psa_domain_golden.py).

CPU tests: the closed form against the fixture (inside: all builds = the oracle's DP; outside: the reference differs), so the
oracle is pinned on the edge scorings too.  GPU tests: inside the domain the engine returns the fixture's values with TSTA_OK;
outside it returns the plain DP with TSTA_WARN_REFERENCE_UNDEFINED and a text, through every PSA entry point and the CLIs.
"""
import hashlib
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import tsta_amd
from tests import util
from tests.golden.make_psa_domain_golden import pair_inputs

with open(os.path.join(util.GOLDEN_DIR, "psa_domain_golden.json")) as f:
    DOMAIN = json.load(f)
PAIRS = {c["name"]: pair_inputs(c) for c in DOMAIN["pairs"]}
CASES = DOMAIN["cases"]
WARN = 1                                                     # TSTA_WARN_REFERENCE_UNDEFINED


def inside(sc):
    return tsta_amd.psa_reference_exact(**sc)[0]


def out_md5(row_a, row_b):
    return hashlib.md5(b">1\n" + row_a + b"\n>2\n" + row_b).hexdigest()


# ------------------------------------------------------------------------------------------------ CPU

def test_domain_function_known_edges():
    ok = lambda **kw: tsta_amd.psa_reference_exact(**kw)
    assert ok() == (True, "")                                             # defaults 2 -5 -2 -4
    assert ok(M=2, X=-3)[0] and ok(M=10, X=-10, E=-3, O=-12)[0] and ok(M=1, X=-1, E=-1, O=0)[0] and ok(M=5, X=-4, E=-1, O=-10)[0]
    assert ok(E=-28)[0] and not ok(E=-29)[0]                              # MIN + E
    assert ok(O=-60)[0] and not ok(O=-62)[0]                              # 2(O+E)+E
    assert ok(M=121, X=-121)[0] and not ok(M=122, X=-122)[0]              # M-(O+E)
    assert ok(X=-128)[0] and not ok(X=-129)[0]
    assert ok(X=0)[0] and not ok(X=1)[0] and not ok(E=1)[0] and not ok(O=1)[0]
    inside_, why = ok(E=-30)
    assert not inside_ and "E < -28" in why and "psa.c" in why
    assert tsta_amd.lib().tsta_psa_reference_exact(None, None, 0) == -1   # TSTA_ERR_INVALID_ARG


def test_fixture_covers_both_sides():
    n_in = sum(inside(c["scoring"]) for c in CASES)
    assert len(CASES) >= 200 and n_in >= 100 and len(CASES) - n_in >= 60


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_psa_domain_fixture(idx):
    """Inside the closed form: the four reference runs agree with each other and with the oracle's plain DP, on every pair.
    Outside: the reference differs from the DP somewhere (except 0 < X <= M, which only tiny pairs expose: next test)."""
    c = CASES[idx]; sc = c["scoring"]
    all_equal = True
    for name, runs in c["ref"].items():
        a, b = PAIRS[name]
        ms, _, ra, rb = util.oracle_align(a, b, **sc)
        want = [ms, out_md5(ra, rb)]
        all_equal = all_equal and all(r == want for r in runs)
        if inside(sc):
            assert all(r == want for r in runs), (sc, name, want, runs)
    if not inside(sc) and not 0 < sc["X"] <= sc["M"]:
        assert not all_equal, ("closed form needlessly conservative here", sc)


def test_positive_mismatch_differs_on_tiny_pairs():
    for c in DOMAIN["positive_x"]:
        a = np.frombuffer(c["a"].encode(), np.uint8); b = np.frombuffer(c["b"].encode(), np.uint8)
        ms = util.oracle_maxscore(a, b, **c["scoring"])
        assert not inside(c["scoring"]) and all(r[0] != ms for r in c["ref"]), c


# ------------------------------------------------------------------------------------------------ GPU

@pytest.fixture(scope="module")
def eng():
    e = tsta_amd.Engine(0)
    yield e
    e.close()


def _edge_cases(want_inside, limit):
    sel = [c for c in CASES if inside(c["scoring"]) == want_inside]
    return sel[:: max(1, len(sel) // limit)][:limit]


@pytest.mark.gpu
def test_inside_domain_engine_equals_reference(eng):
    # every in-domain scoring of the fixture (incl. the ones ON the edges), all seven pairs in one batch each: scores = what the
    # four reference runs printed, status TSTA_OK
    names = list(PAIRS)
    for c in CASES:
        sc = c["scoring"]
        if not inside(sc):
            continue
        got = eng.psa_score_batch([PAIRS[n] for n in names], tsta_amd.Params(**sc))
        assert eng.last_status == 0, sc
        want = [c["ref"][n][0][0] for n in names]
        assert got.tolist() == want, (sc, got.tolist(), want)


@pytest.mark.gpu
def test_inside_domain_alignment_equals_reference(eng):
    for c in _edge_cases(True, 40):
        sc = c["scoring"]
        for name in ("mut300", "del120", "mut1500"):
            a, b = PAIRS[name]
            ms, ra, rb = eng.psa_align(a, b, tsta_amd.Params(**sc))
            assert eng.last_status == 0
            assert [ms, out_md5(ra, rb)] == c["ref"][name][0], (sc, name)


@pytest.mark.gpu
def test_outside_domain_warns_and_returns_the_plain_dp(eng):
    names = list(PAIRS)
    for c in CASES:
        sc = c["scoring"]
        if inside(sc):
            continue
        pairs = [PAIRS[n] for n in names]
        got = eng.psa_score_batch(pairs, tsta_amd.Params(**sc))
        assert eng.last_status == WARN, sc
        w = eng.last_warning()
        assert "int8" in w and "M=%d X=%d E=%d O=%d" % (sc["M"], sc["X"], sc["E"], sc["O"]) in w
        assert got.tolist() == util.oracle_maxscore_batch(pairs, **sc).tolist(), sc
    for c in _edge_cases(False, 12):
        sc = c["scoring"]
        a, b = PAIRS["mut300"]
        ms, ra, rb = eng.psa_align(a, b, tsta_amd.Params(**sc))
        assert eng.last_status == WARN and eng.last_warning()
        o = util.oracle_align(a, b, **sc)
        assert (ms, ra, rb) == (o[0], o[2], o[3])
        res = eng.psa_align_batch([(a, b), PAIRS["mut64"]], tsta_amd.Params(**sc))
        assert eng.last_status == WARN and res[0] == (o[0], o[2], o[3])
    # an in-domain call afterwards is clean again
    eng.psa_score_batch([PAIRS["mut64"]])
    assert eng.last_status == 0 and eng.last_warning() == ""


@pytest.mark.gpu
def test_outside_domain_other_entry_points(eng):
    sc = dict(M=2, X=-5, E=-30, O=-4)
    prm = tsta_amd.Params(**sc)
    pairs = [PAIRS["mut64"], PAIRS["mut300"]]
    want = util.oracle_maxscore_batch(pairs, **sc).tolist()
    assert eng.psa_score_batch_2bit(pairs, prm).tolist() == want and eng.last_status == WARN
    m = tsta_amd.Multi()
    try:
        assert m.psa_score_batch(pairs, prm).tolist() == want and m.last_status == WARN and "E < -28" in m.last_warning()
        res = m.psa_align_batch(pairs, prm)
        assert m.last_status == WARN and [r[0] for r in res] == want
        assert m.psa_score_batch(pairs).tolist() == util.oracle_maxscore_batch(pairs).tolist() and m.last_status == 0
    finally:
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tool", ["TSTA_psa", "TSTA_psa_notrace"])
def test_cli_notes_reference_undefined_scoring_on_stderr(tool):
    exe = os.path.join(util.ROOT, "tsta_amd", "bin", tool)
    a, b = PAIRS["mut300"]
    with tempfile.TemporaryDirectory() as td:
        f1, f2, fo = (os.path.join(td, n) for n in ("a.fa", "b.fa", "o.txt"))
        util.write_fasta(f1, "a", a); util.write_fasta(f2, "b", b)
        for sc, warn in ((dict(M=2, X=-5, E=-28, O=-4), False), (dict(M=2, X=-5, E=-29, O=-4), True)):
            cmd = [exe, "-M", str(sc["M"]), "-X", str(sc["X"]), "-E", str(sc["E"]), "-O", str(sc["O"]), "-1", f1, "-2", f2, "-o", fo]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=td)
            o = util.oracle_align(a, b, **sc)
            assert r.returncode == 0 and r.stdout == "maxsorce=%d\n" % o[0]          # stdout keeps the reference's format
            assert ("warning:" in r.stderr and "E < -28" in r.stderr) == warn, r.stderr
            if tool == "TSTA_psa":
                with open(fo, "rb") as f:
                    assert f.read() == b">1\n" + o[2] + b"\n>2\n" + o[3]
