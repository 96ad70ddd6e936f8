"""Pins the oracle (oracle/psa_oracle.c) against golden vectors produced by the unmodified reference
(tests/golden/make_golden.py) and against the README known answers (README.md:16-24)."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import util
from tests.golden.make_golden import make_case_inputs

with open(os.path.join(util.GOLDEN_DIR, "psa_golden.json")) as f:
    GOLDEN = json.load(f)


def read_fasta_first(path):
    seq = []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if seq:
                    break
                continue
            seq.append(line.strip())
    return np.frombuffer(b"".join(seq), dtype=np.uint8)


def case_inputs(c):
    if c["kind"] == "example":
        return (read_fasta_first(os.path.join(util.GOLDEN_DIR, "seqa1.fa")),
                read_fasta_first(os.path.join(util.GOLDEN_DIR, "seqb1.fa")))
    return make_case_inputs(c)


def output_bytes(row_a, row_b):
    return b">1\n" + row_a + b"\n>2\n" + row_b          # psa.c:512-516 (no trailing newline)


@pytest.mark.parametrize("idx", range(len(GOLDEN)))
def test_oracle_matches_reference_golden(idx):
    c = GOLDEN[idx]
    s1, s2 = case_inputs(c)
    sc = c["scoring"]
    assert util.oracle_maxscore(s1, s2, **sc) == c["maxsorce"]
    ms, end, ra, rb = util.oracle_align(s1, s2, **sc)
    assert ms == c["maxsorce"]
    out = output_bytes(ra, rb)
    assert len(out) == c["out_size"]
    assert hashlib.md5(out).hexdigest() == c["out_md5"]


def test_readme_known_answer():
    # README.md:16-24: both CLIs print maxsorce=-5 on the example pair with defaults
    c = [g for g in GOLDEN if g["kind"] == "example" and g["params"] == "default"][0]
    assert c["maxsorce"] == -5 and c["out_md5"] == "c6eef3a801d4bb89a6c8d7658e301cbe"
    c = [g for g in GOLDEN if g["kind"] == "example" and g["params"] == "script"][0]
    assert c["maxsorce"] == -3 and c["out_md5"] == "175ca544f12b3fcb8c1522dc87eb85b3"


def test_oracle_batch_matches_single():
    pairs = [(util.random_dna(10 + i, 50 + 13 * i), util.mutate(util.random_dna(10 + i, 50 + 13 * i), 99 + i, 0.1))
             for i in range(12)]
    got = util.oracle_maxscore_batch(pairs, threads=4)
    for (a, b), g in zip(pairs, got):
        assert util.oracle_maxscore(a, b) == g


def test_oracle_empty_inputs():
    a = util.random_dna(3, 10)
    e = np.zeros(0, dtype=np.uint8)
    assert util.oracle_maxscore(a, e) == -100        # ms floor, psa.c:110,590
    ms, end, ra, rb = util.oracle_align(a, e)
    assert ms == -100 and ra == bytes(a) and rb == b"-" * 10
