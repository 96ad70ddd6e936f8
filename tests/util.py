"""Shared test helpers: deterministic DNA generator, oracle loader (ctypes), FASTA writer.

The oracle (oracle/liboracle.so) is TEST INFRASTRUCTURE: imported only from tests/, bench.py's cpu_baseline
leg and __graft_entry__.smoke().
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

_ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)
_MASK = (1 << 64) - 1


def splitmix64(seed, n):
    """n 64-bit values of the counter-based splitmix64 stream `seed` (vectorised, platform independent)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed & _MASK) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def random_dna(seed, n):
    """iid uniform ACGT of length n (uint8 array)."""
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    return _ALPHA[(splitmix64(seed, n) >> np.uint64(62)).astype(np.int64)]


def mutate(seq, seed, rate=0.05):
    """Substitutions / insertions / deletions, each at rate/3 per base (deterministic in seed)."""
    n = len(seq)
    if n == 0:
        return seq.copy()
    r = splitmix64(seed ^ 0xA5A5A5A5, 2 * n)
    u = (r[:n] >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    nb = _ALPHA[(r[n:] >> np.uint64(62)).astype(np.int64)]
    out = []
    t = rate / 3.0
    for i in range(n):
        if u[i] < t:                       # substitution
            out.append(nb[i])
        elif u[i] < 2 * t:                 # insertion before the base
            out.append(nb[i]); out.append(seq[i])
        elif u[i] < 3 * t:                 # deletion
            continue
        else:
            out.append(seq[i])
    return np.array(out, dtype=np.uint8)


def low_complexity(seed, n, period=3):
    unit = random_dna(seed, period)
    return np.resize(unit, n).astype(np.uint8)


def write_fasta(path, name, seq, width=0):
    s = bytes(seq)
    with open(path, "wb") as f:
        f.write(b">" + name.encode() + b"\n")
        if width:
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + b"\n")
        else:
            f.write(s + b"\n")


# ----------------------------------------------------------------------------------- oracle (ctypes)

class OracleParams(C.Structure):
    _fields_ = [("M", C.c_int32), ("X", C.c_int32), ("E", C.c_int32), ("O", C.c_int32)]


class OracleDag(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("pred_off", C.c_void_p), ("pred", C.c_void_p),
                ("base", C.c_void_p), ("is_sink", C.c_void_p)]


_oracle = None


def build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("psa_oracle.c", "poa_oracle.c", "oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
    return so


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(build_oracle())
        lib.oracle_psa_maxscore.restype = C.c_int32
        lib.oracle_psa_maxscore.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(OracleParams)]
        lib.oracle_psa_scores.restype = None
        lib.oracle_psa_scores.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(OracleParams),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.oracle_psa_maxscore_batch.restype = None
        lib.oracle_psa_maxscore_batch.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.POINTER(OracleParams), C.c_void_p, C.c_int]
        lib.oracle_psa_align.restype = C.c_int32
        lib.oracle_psa_align.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(OracleParams),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int32)]
        lib.oracle_psa_fill_trace.restype = C.c_int32
        lib.oracle_psa_fill_trace.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(OracleParams),
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        lib.oracle_poa_fill.restype = C.c_int32
        lib.oracle_poa_fill.argtypes = [C.POINTER(OracleDag), C.c_void_p, C.c_int32, C.POINTER(OracleParams),
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_free.argtypes = [C.c_void_p]
        _oracle = lib
    return _oracle


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def oracle_maxscore(s1, s2, M=2, X=-5, E=-2, O=-4):
    s1 = np.ascontiguousarray(s1, dtype=np.uint8); s2 = np.ascontiguousarray(s2, dtype=np.uint8)
    p = OracleParams(M, X, E, O)
    return int(oracle().oracle_psa_maxscore(_ptr(s1), len(s1), _ptr(s2), len(s2), C.byref(p)))


def oracle_scores(s1, s2, M=2, X=-5, E=-2, O=-4):
    """-> (maxsorce, H(m,n)) in O(n) memory."""
    s1 = np.ascontiguousarray(s1, dtype=np.uint8); s2 = np.ascontiguousarray(s2, dtype=np.uint8)
    p = OracleParams(M, X, E, O); ms, es = C.c_int32(), C.c_int32()
    oracle().oracle_psa_scores(_ptr(s1), len(s1), _ptr(s2), len(s2), C.byref(p), C.byref(ms), C.byref(es))
    return int(ms.value), int(es.value)


def alignment_score(row_a, row_b, M=2, X=-5, E=-2, O=-4):
    """Affine score of a gapped alignment (two equal-length byte strings with '-')."""
    a = np.frombuffer(row_a, dtype=np.uint8); b = np.frombuffer(row_b, dtype=np.uint8)
    ga = a == ord("-"); gb = b == ord("-")
    assert not np.any(ga & gb)
    both = ~ga & ~gb
    score = int(np.sum(np.where(a[both] == b[both], M, X)))
    for g in (ga, gb):
        ext = int(np.sum(g))
        opens = int(np.sum(g[1:] & ~g[:-1])) + int(g[0]) if len(g) else 0
        score += ext * E + opens * O
    return score


def oracle_maxscore_batch(pairs, M=2, X=-5, E=-2, O=-4, threads=8):
    """pairs: list of (s1, s2) uint8 arrays -> int32 array of maxsorce."""
    flat, o1, l1, o2, l2 = [], [], [], [], []
    pos = 0
    for a, b in pairs:
        o1.append(pos); l1.append(len(a)); flat.append(np.asarray(a, dtype=np.uint8)); pos += len(a)
        o2.append(pos); l2.append(len(b)); flat.append(np.asarray(b, dtype=np.uint8)); pos += len(b)
    seqs = np.concatenate(flat) if pos else np.zeros(1, dtype=np.uint8)
    o1 = np.array(o1, dtype=np.int64); o2 = np.array(o2, dtype=np.int64)
    l1 = np.array(l1, dtype=np.int32); l2 = np.array(l2, dtype=np.int32)
    out = np.zeros(len(pairs), dtype=np.int32)
    p = OracleParams(M, X, E, O)
    oracle().oracle_psa_maxscore_batch(len(pairs), _ptr(seqs), _ptr(o1), _ptr(l1), _ptr(o2), _ptr(l2),
                                       C.byref(p), _ptr(out), threads)
    return out


def oracle_align(s1, s2, M=2, X=-5, E=-2, O=-4):
    """-> (maxsorce, end_score, row_a bytes, row_b bytes); row_a belongs to the longer input."""
    s1 = np.ascontiguousarray(s1, dtype=np.uint8); s2 = np.ascontiguousarray(s2, dtype=np.uint8)
    p = OracleParams(M, X, E, O)
    ra, rb, ln, es = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int32()
    ms = oracle().oracle_psa_align(_ptr(s1), len(s1), _ptr(s2), len(s2), C.byref(p),
                                   C.byref(ra), C.byref(rb), C.byref(ln), C.byref(es))
    a = C.string_at(ra, ln.value); b = C.string_at(rb, ln.value)
    oracle().oracle_free(ra); oracle().oracle_free(rb)
    return int(ms), int(es.value), a, b


def oracle_fill_trace(a, b, M=2, X=-5, E=-2, O=-4):
    """a = columns, b = rows (no swap). -> (maxsorce, end, back, fback, eback) int8 [m][n]."""
    a = np.ascontiguousarray(a, dtype=np.uint8); b = np.ascontiguousarray(b, dtype=np.uint8)
    n, m = len(a), len(b)
    back = np.zeros((m, n), dtype=np.int8); fb = np.zeros((m, n), dtype=np.int8); eb = np.zeros((m, n), dtype=np.int8)
    p = OracleParams(M, X, E, O); es = C.c_int32()
    ms = oracle().oracle_psa_fill_trace(_ptr(a), n, _ptr(b), m, C.byref(p), _ptr(back), _ptr(fb), _ptr(eb),
                                        C.byref(es))
    return int(ms), int(es.value), back, fb, eb


def oracle_poa_fill(pred_off, pred, base, is_sink, seq, M=2, X=-5, E=-2, O=-4, want_trace=True):
    pred_off = np.ascontiguousarray(pred_off, dtype=np.int32)
    pred = np.ascontiguousarray(pred if len(pred) else np.zeros(1), dtype=np.int32)
    base = np.ascontiguousarray(base, dtype=np.uint8); is_sink = np.ascontiguousarray(is_sink, dtype=np.uint8)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    V, n = len(base), len(seq)
    g = OracleDag(V, _ptr(pred_off).value, _ptr(pred).value, _ptr(base).value, _ptr(is_sink).value)
    p = OracleParams(M, X, E, O)
    last = np.zeros(V, dtype=np.int32)
    if want_trace:
        src = np.zeros((V, n), dtype=np.int8); es = np.zeros((V, n), dtype=np.int8); fs = np.zeros((V, n), dtype=np.int8)
        best = oracle().oracle_poa_fill(C.byref(g), _ptr(seq), n, C.byref(p), _ptr(src), _ptr(es), _ptr(fs), _ptr(last))
        return int(best), last, src, es, fs
    best = oracle().oracle_poa_fill(C.byref(g), _ptr(seq), n, C.byref(p), None, None, None, _ptr(last))
    return int(best), last, None, None, None


def oracle_poa_fill_blocked(pred_off, pred, base, is_sink, seq, W, M=2, X=-5, E=-2, O=-4):
    """The fill in the blocking of the reference's AVX512 build at -W W (oracle/poa_oracle.c: oracle_poa_fill_blocked)."""
    pred_off = np.ascontiguousarray(pred_off, dtype=np.int32)
    pred = np.ascontiguousarray(pred if len(pred) else np.zeros(1), dtype=np.int32)
    base = np.ascontiguousarray(base, dtype=np.uint8); is_sink = np.ascontiguousarray(is_sink, dtype=np.uint8)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    V, n = len(base), len(seq)
    g = OracleDag(V, _ptr(pred_off).value, _ptr(pred).value, _ptr(base).value, _ptr(is_sink).value)
    p = OracleParams(M, X, E, O)
    last = np.zeros(V, dtype=np.int32)
    src = np.zeros((V, n), dtype=np.int8); es = np.zeros((V, n), dtype=np.int8); fs = np.zeros((V, n), dtype=np.int8)
    lib = oracle()
    lib.oracle_poa_fill_blocked.restype = C.c_int32
    lib.oracle_poa_fill_blocked.argtypes = [C.POINTER(OracleDag), C.c_void_p, C.c_int32, C.POINTER(OracleParams), C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    best = lib.oracle_poa_fill_blocked(C.byref(g), _ptr(seq), n, C.byref(p), int(W), _ptr(src), _ptr(es), _ptr(fs), _ptr(last))
    return int(best), last, src, es, fs


def oracle_poa_fill_i8(pred_off, pred, base, is_sink, seq, W, M=2, X=-5, E=-2, O=-4):
    """The fill in the reference's own int8 number representation and block schedule at -W W (oracle/poa_oracle_i8.c)."""
    pred_off = np.ascontiguousarray(pred_off, dtype=np.int32)
    pred = np.ascontiguousarray(pred if len(pred) else np.zeros(1), dtype=np.int32)
    base = np.ascontiguousarray(base, dtype=np.uint8); is_sink = np.ascontiguousarray(is_sink, dtype=np.uint8)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    V, n = len(base), len(seq)
    g = OracleDag(V, _ptr(pred_off).value, _ptr(pred).value, _ptr(base).value, _ptr(is_sink).value)
    p = OracleParams(M, X, E, O)
    last = np.zeros(V, dtype=np.int32)
    src = np.zeros((V, n), dtype=np.int8); es = np.zeros((V, n), dtype=np.int8); fs = np.zeros((V, n), dtype=np.int8)
    lib = oracle()
    lib.oracle_poa_fill_i8.restype = C.c_int32
    lib.oracle_poa_fill_i8.argtypes = [C.POINTER(OracleDag), C.c_void_p, C.c_int32, C.POINTER(OracleParams), C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    best = lib.oracle_poa_fill_i8(C.byref(g), _ptr(seq), n, C.byref(p), int(W), _ptr(src), _ptr(es), _ptr(fs), _ptr(last))
    return int(best), last, src, es, fs


# ------------------------------------------------------------------------------ reference binaries

def ref_available(name="TSTA_psa"):
    return os.path.exists(os.path.join(REF_DIR, name))


def cpu_has(flag):
    try:
        with open("/proc/cpuinfo") as f:
            return flag in f.read()
    except OSError:
        return False


def poa_walk(pred_off, pred, is_sink, lastscore, src, es, fs):
    """Python restatement of the trace walk of node_fuse (c-t-simd.c:737-957) over full [V][n] arrays.
    -> (list of (node, col, code, peek), end_col).  Test infrastructure (mirrors walk_arrays in poa_graph.c)."""
    V, n = src.shape
    sinks = [i for i in range(1, V) if is_sink[i]]
    num1 = V - 1
    if sinks:
        best = max(lastscore[i] for i in sinks)
        num1 = min(i for i in sinks if lastscore[i] == best)
    num2, forced, steps = n - 1, -1, []
    while num1 != -1 and num2 != -1:
        code = forced if forced >= 0 else int(src[num1, num2])
        forced = -1
        kind, pidx = code // 42, code % 42
        npred = pred_off[num1 + 1] - pred_off[num1]
        psub = int(pred[pred_off[num1] + pidx]) if pidx < npred else -1
        peek = 0
        if kind == 3:
            f = int(fs[num1, num2])
            if num2 - 1 > 0 and (abs(f) == 1 or (abs(f) == 2 and fs[num1, num2 - 1] < 0)):
                forced = 126
            steps.append((num1, num2, code, 0)); num2 -= 1
        elif kind == 0:
            e = int(es[num1, num2])
            if psub > 0 and (-42 <= e <= 42 or ((e > 42 or e < -42) and es[psub, num2] < 0)):
                v = int(es[psub, num2]); r = abs(v) % 42          # C remainder: sign follows the dividend
                forced = r - 1
            steps.append((num1, num2, code, 0)); num1 = psub
        else:
            if psub >= 0 and num2 - 1 >= 0:
                peek = 1 if int(src[psub, num2 - 1]) // 42 == 1 else 0
            steps.append((num1, num2, code, peek)); num1 = psub; num2 -= 1
    return steps, num2
