#!/usr/bin/env python3
"""Generate tests/golden/poa_*.npz: DAGs + expected POA-fill outputs dumped from the compiled reference.

Runs only in the build container (needs /root/reference).  Recipe: a SCRATCH copy of
msa/c-t-simd.c is made in a temp dir OUTSIDE the repo, one dump block is inserted in control() right before the
call of node_fuse() (c-t-simd.c:1046-1047, i.e. after the fill wavefront and before node_fuse mutates the trace
arrays), and it is compiled with the unmodified msa.c / topo.c / result.c / pthreadpool.c / seqio.c.  Nothing of
the reference's source is written into the repo; the fixtures hold DATA only:
  per alignment step: the sequence, the DAG in p->sort[] order (base, in/out degree, predecessor indices in
  pre[] order), lastsorce of the sinks, and the three trace arrays source/esource/fsource de-striped to plain
  [node][column] order for the real columns (small cases) or their sha256 (README example, 150 MB per step).
The output file md5 of the patched binary is checked against the unpatched one (the dump must not perturb it).
"""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util  # noqa: E402

REF = "/root/reference"

DUMP_BLOCK = r'''
	{ const char* dp_ = getenv("TSTA_DUMP");
	  if (dp_) { FILE* df_ = fopen(dp_, "ab"); int n_real_ = (int)strlen(A); int v_ = p->len;
	    fwrite("STEP", 1, 4, df_); fwrite(&v_, 4, 1, df_); fwrite(&n_real_, 4, 1, df_); fwrite(A, 1, n_real_, df_);
	    for (int r_ = 0; r_ < p->len; r_++) { poa* nd_ = p->sort[r_]; char b_ = nd_->base;
	      fwrite(&b_, 1, 1, df_); fwrite(&nd_->sub, 4, 1, df_); fwrite(&nd_->in, 4, 1, df_); fwrite(&nd_->out, 4, 1, df_);
	      fwrite(&nd_->lastsorce, 4, 1, df_);
	      for (int k_ = 0; k_ < nd_->in; k_++) { int s_ = nd_->pre[k_]->sub; fwrite(&s_, 4, 1, df_); }
	      for (int c_ = 0; c_ < n_real_; c_++) fwrite(&nd_->source[NUM2(c_)], 1, 1, df_);
	      for (int c_ = 0; c_ < n_real_; c_++) fwrite(&nd_->esource[NUM2(c_)], 1, 1, df_);
	      for (int c_ = 0; c_ < n_real_; c_++) fwrite(&nd_->fsource[NUM2(c_)], 1, 1, df_);
	    }
	    fclose(df_); } }
'''


def build_patched(td):
    with open(os.path.join(REF, "msa", "c-t-simd.c")) as f:
        src = f.read()
    marker = "\tint last = maxtag + 1;"
    assert src.count(marker) == 1
    src = src.replace(marker, DUMP_BLOCK + marker)
    patched = os.path.join(td, "c-t-simd.c")
    with open(patched, "w") as f:
        f.write(src)
    exe = os.path.join(td, "TSTA_msa_dump")
    simd = ["-mavx512f", "-mavx512bw", "-mavx512dq"] if util.cpu_has("avx512bw") else ["-mavx2"]
    cmd = ["gcc", "-O3", "-w"] + simd + ["-I" + REF, "-I" + REF + "/msa", "-I" + REF + "/psa", "-o", exe,
           REF + "/msa/msa.c", patched, REF + "/msa/topo.c", REF + "/msa/result.c",
           REF + "/pthreadpool/pthreadpool.c", REF + "/seqio.c", "-lz", "-lm", "-lpthread"]
    subprocess.check_call(cmd)
    return exe


def parse_dump(path):
    with open(path, "rb") as f:
        buf = f.read()
    pos, steps = 0, []
    while pos < len(buf):
        assert buf[pos:pos + 4] == b"STEP"
        V, n = struct.unpack_from("<ii", buf, pos + 4); pos += 12
        seq = np.frombuffer(buf, np.uint8, n, pos).copy(); pos += n
        base = np.zeros(V, np.uint8); is_sink = np.zeros(V, np.uint8); last = np.zeros(V, np.int32)
        pred_off = np.zeros(V + 1, np.int32); pred = []
        src = np.zeros((V, n), np.int8); es = np.zeros((V, n), np.int8); fs = np.zeros((V, n), np.int8)
        for r in range(V):
            b, = struct.unpack_from("<B", buf, pos); pos += 1
            sub, nin, nout, ls = struct.unpack_from("<iiii", buf, pos); pos += 16
            assert sub == r, (sub, r)
            base[r] = b; is_sink[r] = 1 if nout == 0 else 0; last[r] = ls
            ps = struct.unpack_from("<%di" % nin, buf, pos); pos += 4 * nin
            assert all(0 <= x < r for x in ps), (r, ps)
            pred.extend(ps); pred_off[r + 1] = len(pred)
            src[r] = np.frombuffer(buf, np.int8, n, pos); pos += n
            es[r] = np.frombuffer(buf, np.int8, n, pos); pos += n
            fs[r] = np.frombuffer(buf, np.int8, n, pos); pos += n
        steps.append(dict(seq=seq, base=base, is_sink=is_sink, lastsorce=last, pred_off=pred_off,
                          pred=np.array(pred, np.int32), source=src, esource=es, fsource=fs))
    return steps


def family(seed, n_seq, length, rate, jitter=0):
    root = util.random_dna(seed, length)
    out = []
    for k in range(n_seq):
        s = util.mutate(root, seed + 100 + k, rate)
        if jitter == 0:
            s = s[:length] if len(s) >= length else np.concatenate([s, util.random_dna(seed + 500 + k, length - len(s))])
        out.append(s)
    return out


def write_multi_fasta(path, seqs):
    with open(path, "wb") as f:
        for k, s in enumerate(seqs):
            f.write(b">s%d\n" % k + bytes(s) + b"\n")


CASES = [  # name, generator args, scoring, keep full trace arrays?
    ("small_a", dict(seed=11, n_seq=5, length=300, rate=0.10), dict(M=2, X=-5, E=-2, O=-4), True),
    ("small_b", dict(seed=23, n_seq=6, length=150, rate=0.20), dict(M=2, X=-3, E=-2, O=-4), True),
    ("small_c", dict(seed=37, n_seq=4, length=500, rate=0.05, jitter=1), dict(M=2, X=-5, E=-2, O=-4), True),
    ("small_d", dict(seed=41, n_seq=5, length=200, rate=0.15), dict(M=1, X=-1, E=-1, O=0), True),
    ("medium", dict(seed=53, n_seq=4, length=1500, rate=0.08), dict(M=2, X=-5, E=-2, O=-4), False),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save_steps(name, steps, scoring, full):
    for k, st in enumerate(steps):
        d = dict(seq=st["seq"], base=st["base"], is_sink=st["is_sink"], lastsorce=st["lastsorce"],
                 pred_off=st["pred_off"], pred=st["pred"],
                 scoring=np.array([scoring["M"], scoring["X"], scoring["E"], scoring["O"]], np.int32),
                 sha=np.array([sha(st["source"]), sha(st["esource"]), sha(st["fsource"])]))
        if full:
            d.update(source=st["source"], esource=st["esource"], fsource=st["fsource"])
        np.savez_compressed(os.path.join(util.GOLDEN_DIR, "poa_%s_step%d.npz" % (name, k + 1)), **d)
        print(name, "step", k + 1, "V=%d n=%d maxin=%d" % (len(st["base"]), len(st["seq"]),
                                                          int(np.diff(st["pred_off"]).max())))


def run(exe, fasta, scoring, td):
    dump = os.path.join(td, "dump.bin")
    if os.path.exists(dump):
        os.remove(dump)
    out = os.path.join(td, "out.txt")
    env = dict(os.environ, TSTA_DUMP=dump)
    subprocess.run([exe, "-M", str(scoring["M"]), "-X", str(scoring["X"]), "-E", str(scoring["E"]), "-O",
                    str(scoring["O"]), "-T", "4", "-i", fasta, "-o", out], env=env, check=True,
                   capture_output=True, timeout=900)
    with open(out, "rb") as f:
        md5 = hashlib.md5(f.read()).hexdigest()
    return parse_dump(dump), md5


def main():
    with tempfile.TemporaryDirectory() as td:
        exe = build_patched(td)
        # README example (README.md:26-40): known answers for graph size and best-sink lastsorce per step
        default = dict(M=2, X=-5, E=-2, O=-4)
        steps, md5 = run(exe, os.path.join(REF, "example/msa/seq/seq1.fa"), default, td)
        assert md5 == "e7b253a265351c1e4f97dd3b663da7ee", md5       # dump did not perturb the run
        assert [len(s["base"]) for s in steps] == [5000, 6885, 8599, 10130]
        save_steps("readme", steps, default, full=False)
        for name, gen, scoring, full in CASES:
            fa = os.path.join(td, name + ".fa")
            write_multi_fasta(fa, family(**gen))
            steps, _ = run(exe, fa, scoring, td)
            ref_out = os.path.join(td, "ref_out.txt")
            subprocess.run([os.path.join(util.REF_DIR, "TSTA_msa"), "-M", str(scoring["M"]), "-X", str(scoring["X"]),
                            "-E", str(scoring["E"]), "-O", str(scoring["O"]), "-T", "4", "-i", fa, "-o", ref_out],
                           check=True, capture_output=True, timeout=900)
            save_steps(name, steps, scoring, full)


if __name__ == "__main__":
    main()
