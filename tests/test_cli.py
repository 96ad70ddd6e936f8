"""Drop-in CLIs (tsta_amd/host/cli_psa.c): flag handling / usage text vs the reference binary (CPU), FASTA reader
behaviour (CPU), and end-to-end byte parity of stdout + output file on the GPU."""
import gzip
import hashlib
import os
import subprocess

import numpy as np
import pytest

from tests import util

BIN = os.path.join(util.ROOT, "tsta_amd", "bin")


@pytest.fixture(scope="module")
def built():
    import tsta_amd
    tsta_amd.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(util.ROOT, "tsta_amd", "host")])
    return BIN


def run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, timeout=300, **kw)


@pytest.mark.parametrize("name", ["TSTA_psa", "TSTA_psa_notrace"])
def test_usage_matches_reference(built, name):
    ours = run([os.path.join(built, name)])
    assert ours.returncode == 0 and ours.stdout.startswith(b"input file is not specified\n")   # psa.c:583-587
    if util.ref_available(name) and util.cpu_has("avx512bw"):
        ref = run([os.path.join(util.REF_DIR, name)])
        assert ours.stdout == ref.stdout and ours.returncode == ref.returncode
        ours = run([os.path.join(built, name), "-Z"]); ref = run([os.path.join(util.REF_DIR, name), "-Z"])
        assert ours.stdout == ref.stdout and ours.returncode == ref.returncode                # unknown flag -> usage


def test_fasta_reader(built, tmp_path):
    dump = os.path.join(built, "fasta_dump")
    p = tmp_path / "a.fa"
    p.write_bytes(b"junk before\n>s1 comment here\nACGT\r\nacgtNN\n\nTT\n>s2\nGG\n>empty\n>s4\nA")
    out = run([dump, str(p)]).stdout.decode().splitlines()
    assert out == ["4", "12 ACGTacgtNNTT", "2 GG", "0 ", "1 A"]
    assert run([dump, str(p), "1"]).stdout.decode().splitlines() == ["1", "12 ACGTacgtNNTT"]   # first record only
    g = tmp_path / "a.fa.gz"
    g.write_bytes(gzip.compress(p.read_bytes()))                                             # gzip by magic
    assert run([dump, str(g)]).stdout.decode().splitlines() == out
    q = tmp_path / "r.fq"
    q.write_bytes(b"@r1\nACGT\n+\nIIII\n")
    r = run([dump, str(q)])
    assert r.returncode == 1 and b"Cannot read fasta record from a fastq file." in r.stderr   # seqio.c:416-423
    r = run([dump, str(tmp_path / "missing.fa")])
    assert r.returncode == 1 and b"does not exist" in r.stderr                                # seqio.c:175-183


@pytest.mark.gpu
def test_cli_end_to_end_example(built, tmp_path):
    a = os.path.join(util.GOLDEN_DIR, "seqa1.fa"); b = os.path.join(util.GOLDEN_DIR, "seqb1.fa")
    r = run([os.path.join(built, "TSTA_psa_notrace"), "-1", a, "-2", b])
    assert r.returncode == 0 and r.stdout == b"maxsorce=-5\n"                                 # README.md:21-24
    out = tmp_path / "o.txt"
    r = run([os.path.join(built, "TSTA_psa"), "-1", a, "-2", b, "-o", str(out)])
    assert r.returncode == 0 and r.stdout == b"maxsorce=-5\n"                                 # README.md:16-19
    assert hashlib.md5(out.read_bytes()).hexdigest() == "c6eef3a801d4bb89a6c8d7658e301cbe"
    r = run([os.path.join(built, "TSTA_psa"), "-M", "2", "-X", "-3", "-E", "-2", "-O", "-4", "-T", "30", "-W", "30",
             "-1", a, "-2", b, "-o", str(out)])                                              # example_psa.sh:4
    assert r.stdout == b"maxsorce=-3\n"
    assert hashlib.md5(out.read_bytes()).hexdigest() == "175ca544f12b3fcb8c1522dc87eb85b3"


@pytest.mark.gpu
def test_cli_gz_crlf_swapped_inputs(built, tmp_path):
    x = util.random_dna(5, 1500); y = util.mutate(x, 6, 0.1)[:1200]
    fa = tmp_path / "x.fa.gz"; fb = tmp_path / "y.fa"
    fa.write_bytes(gzip.compress(b">x\n" + bytes(x) + b"\n"))
    fb.write_bytes(b">y\r\n" + b"\r\n".join(bytes(y)[i:i + 60] for i in range(0, len(y), 60)) + b"\r\n>second\nAAAA\n")
    out = tmp_path / "o.txt"
    r = run([os.path.join(built, "TSTA_psa"), "-1", str(fb), "-2", str(fa), "-o", str(out)])   # shorter first: swap
    ms, end, ra, rb = util.oracle_align(y, x)
    assert r.stdout == b"maxsorce=%d\n" % ms
    assert out.read_bytes() == b">1\n" + ra + b"\n>2\n" + rb


@pytest.mark.gpu
def test_batch_cli(built, tmp_path):
    # TSTA_psa_batch: record k of -1 against record k of -2, ragged lengths, one `maxsorce=` line per pair
    pairs = []
    for i in range(40):
        a = util.random_dna(700 + i, 200 + 37 * (i % 9))
        pairs.append((a, util.mutate(a, 800 + i, 0.1)))
    fa, fb = tmp_path / "a.fa", tmp_path / "b.fa"
    fa.write_bytes(b"".join(b">a%d\n" % i + bytes(p[0]) + b"\n" for i, p in enumerate(pairs)))
    fb.write_bytes(b"".join(b">b%d\n" % i + bytes(p[1]) + b"\n" for i, p in enumerate(pairs)))
    r = run([os.path.join(built, "TSTA_psa_batch"), "-1", str(fa), "-2", str(fb)])
    assert r.returncode == 0, r.stderr
    want = util.oracle_maxscore_batch(pairs)
    assert r.stdout == b"".join(b"maxsorce=%d\n" % w for w in want)
    # with -o: every pair aligned with traceback in one tsta_psa_align_batch call; per pair the bytes TSTA_psa writes
    out = tmp_path / "alignments.txt"
    r = run([os.path.join(built, "TSTA_psa_batch"), "-1", str(fa), "-2", str(fb), "-o", str(out)])
    assert r.returncode == 0, r.stderr
    assert r.stdout == b"".join(b"maxsorce=%d\n" % w for w in want)
    expect = b""
    for a, b in pairs:
        ms, end, ra, rb = util.oracle_align(a, b)
        expect += b">1\n" + ra + b"\n>2\n" + rb + b"\n"
    assert out.read_bytes() == expect
    # with -A: all pairs (every record of -1 against every record of -2), sequences stored once
    fa3, fb3 = tmp_path / "a3.fa", tmp_path / "b3.fa"
    fa3.write_bytes(b"".join(b">a%d\n" % i + bytes(p[0]) + b"\n" for i, p in enumerate(pairs[:5])))
    fb3.write_bytes(b"".join(b">b%d\n" % i + bytes(p[1]) + b"\n" for i, p in enumerate(pairs[:7])))
    r = run([os.path.join(built, "TSTA_psa_batch"), "-1", str(fa3), "-2", str(fb3), "-A"])
    assert r.returncode == 0, r.stderr
    cross = [(pairs[i][0], pairs[j][1]) for i in range(5) for j in range(7)]
    wc = util.oracle_maxscore_batch(cross)
    assert r.stdout == b"".join(b"%d\t%d\tmaxsorce=%d\n" % (k // 7 + 1, k % 7 + 1, w) for k, w in enumerate(wc))
