#!/usr/bin/env python3
"""Generate tests/golden/psa_golden.json by running the UNMODIFIED reference (oracle/_ref/TSTA_psa*, built by
`make -C oracle ref` straight from /root/reference) on seeded inputs.

Runs only where oracle/_ref exists (the build container).  The fixture holds DATA only: the case spec
(seeds / lengths / kind / scoring), the reference's printed maxsorce and the md5 + length of its output
file.  Inputs are regenerated from the spec by tests/util.py, so no sequence data needs committing.

Every case is run with the AVX512 (L=640), AVX2 (L=320) and SSE4.2 (L=160) builds and two -W values; a case
is recorded only if all runs agree (the reference's uninitialised pad bytes, psa.c:206-210, are the one
known source of disagreement).
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util  # noqa: E402

PARAM_SETS = {
    "default": dict(M=2, X=-5, E=-2, O=-4),          # psa.c:113-119
    "script": dict(M=2, X=-3, E=-2, O=-4),           # example/psa/example_psa.sh:4
    "open0": dict(M=2, X=-5, E=-2, O=0),             # linear gap
    "big": dict(M=10, X=-10, E=-3, O=-12),
    "unit": dict(M=1, X=-1, E=-1, O=0),
    "wide": dict(M=5, X=-4, E=-1, O=-10),
}


def make_case_inputs(c):
    """(s1, s2) uint8 arrays for a case spec (shared with the tests)."""
    k = c["kind"]
    if k == "random":
        return util.random_dna(c["seed"], c["len1"]), util.random_dna(c["seed"] + 1, c["len2"])
    if k == "mutated":
        a = util.random_dna(c["seed"], c["len1"])
        return a, util.mutate(a, c["seed"] + 7, c["rate"])
    if k == "mutated_swapped":          # shorter given first: exercises psa.c:194-199
        a = util.random_dna(c["seed"], c["len1"])
        return util.mutate(a, c["seed"] + 7, c["rate"]), a
    if k == "lowcx":
        return util.low_complexity(c["seed"], c["len1"], 3), util.low_complexity(c["seed"], c["len2"], 3)
    if k == "homopolymer":
        import numpy as np
        return np.full(c["len1"], ord("A"), np.uint8), np.full(c["len2"], ord("A"), np.uint8)
    raise ValueError(k)


def cases():
    out = []
    seed = 1000
    for ps in ("default", "script"):
        for (l1, l2) in [(1, 1), (2, 1), (1, 7), (5, 5), (63, 64), (64, 64), (65, 63), (100, 37), (640, 640),
                         (641, 639), (1000, 1000), (1279, 1281), (700, 50), (50, 1900), (2000, 2000)]:
            seed += 2
            out.append(dict(kind="random", seed=seed, len1=l1, len2=l2, params=ps))
    for ps in PARAM_SETS:
        for (l, rate) in [(64, 0.05), (300, 0.05), (641, 0.2), (1000, 0.05), (1500, 0.1), (2560, 0.02)]:
            seed += 2
            out.append(dict(kind="mutated", seed=seed, len1=l, rate=rate, params=ps))
        seed += 2
        out.append(dict(kind="mutated_swapped", seed=seed, len1=900, rate=0.08, params=ps))
        seed += 2
        out.append(dict(kind="lowcx", seed=seed, len1=500, len2=480, params=ps))
        seed += 2
        out.append(dict(kind="homopolymer", seed=seed, len1=200, len2=170, params=ps))
    seed += 2
    out.append(dict(kind="mutated", seed=seed, len1=5000, rate=0.05, params="default"))
    seed += 2
    out.append(dict(kind="mutated", seed=seed, len1=6400, rate=0.03, params="script"))
    return out


def run_ref(binary, f1, f2, p, W, out):
    cmd = [os.path.join(util.REF_DIR, binary), "-M", str(p["M"]), "-X", str(p["X"]), "-E", str(p["E"]),
           "-O", str(p["O"]), "-T", "4", "-W", str(W), "-1", f1, "-2", f2, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("maxsorce=")]
    if r.returncode != 0 or not line:
        return None
    with open(out, "rb") as f:
        data = f.read()
    return int(line[0].split("=")[1]), hashlib.md5(data).hexdigest(), len(data)


def main():
    golden = []
    with tempfile.TemporaryDirectory() as td:
        f1, f2, fo = (os.path.join(td, n) for n in ("a.fa", "b.fa", "o.txt"))
        for c in cases():
            s1, s2 = make_case_inputs(c)
            util.write_fasta(f1, "s1", s1)
            util.write_fasta(f2, "s2", s2, width=70)      # multi-line FASTA for the second input
            p = PARAM_SETS[c["params"]]
            res = set()
            for binary, W in (("TSTA_psa", 10), ("TSTA_psa", 3), ("TSTA_psa_avx2", 10), ("TSTA_psa_sse", 10)):
                res.add(run_ref(binary, f1, f2, p, W, fo))
            if len(res) != 1 or None in res:
                print("SKIP (reference runs disagree):", c, res, file=sys.stderr)
                continue
            ms, md5, size = res.pop()
            g = dict(c); g.update(scoring=p, maxsorce=ms, out_md5=md5, out_size=size)
            golden.append(g)
    # README example (README.md:16-24; example/psa/example_psa.sh:4), recorded from the same binaries
    ex = "/root/reference/example/psa/seq/"
    with tempfile.TemporaryDirectory() as td:
        fo = os.path.join(td, "o.txt")
        for name in ("default", "script"):
            r = run_ref("TSTA_psa", ex + "seqa1.fa", ex + "seqb1.fa", PARAM_SETS[name], 10, fo)
            golden.append(dict(kind="example", params=name, scoring=PARAM_SETS[name], maxsorce=r[0],
                               out_md5=r[1], out_size=r[2]))
    path = os.path.join(util.GOLDEN_DIR, "psa_golden.json")
    with open(path, "w") as f:
        json.dump(golden, f, indent=0)
    print("wrote", path, len(golden), "cases")


if __name__ == "__main__":
    main()
