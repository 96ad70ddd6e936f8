"""tsta_amd — MI355X-native drop-in for TSTA's alignment hot path.

Thin ctypes binding over the C-ABI shared library ``tsta_amd/lib/libtsta_hip.so`` (include/tsta_engine.h).
The product is the HIP library; this module only marshals numpy arrays / device pointers into it.  There is NO
CPU fallback: if the library is missing or no gfx950 device is usable, calls raise ``TstaError``.
"""
import ctypes as C
import os

import numpy as np

__all__ = ["TstaError", "Params", "Engine", "Multi", "lib", "lib_path", "build", "build_info", "source_hash", "device_uuid",
           "shard_bounds", "EXPORTED_SYMBOLS"]

_HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(_HERE, "lib", "libtsta_hip.so")

# every symbol include/tsta_engine.h declares
EXPORTED_SYMBOLS = [
    "tsta_strerror", "tsta_last_error", "tsta_device_count", "tsta_engine_create", "tsta_engine_destroy",
    "tsta_free", "tsta_psa_score_batch", "tsta_psa_score_batch_dev", "tsta_psa_align", "tsta_poa_fill",
    "tsta_last_kernel_ms", "tsta_engine_set_option", "tsta_poa_align", "tsta_psa_align_batch", "tsta_alignments_free",
    "tsta_pack_2bit", "tsta_psa_score_batch_2bit", "tsta_psa_score_batch_dev_2bit", "tsta_engine_release_workspaces",
    "tsta_multi_create", "tsta_multi_destroy", "tsta_multi_device_count", "tsta_multi_engine", "tsta_multi_last_error",
    "tsta_multi_set_option", "tsta_shard_bounds", "tsta_psa_score_batch_multi", "tsta_psa_align_batch_multi",
    "tsta_build_info", "tsta_device_uuid", "tsta_last_warning", "tsta_psa_score_batch_dev_status", "tsta_engine_stat",
    "tsta_psa_reference_exact", "tsta_poa_reference_exact",
]


class TstaError(RuntimeError):
    pass


class Alignment(C.Structure):
    """tsta_alignment (include/tsta_engine.h)."""
    _fields_ = [("maxscore", C.c_int32), ("reserved", C.c_int32), ("len", C.c_int64), ("row_a", C.c_void_p),
                ("row_b", C.c_void_p)]


class Params(C.Structure):
    """Scoring parameters: the reference's globals M, X, E, O (psa.c:114-117)."""
    _fields_ = [("M", C.c_int32), ("X", C.c_int32), ("E", C.c_int32), ("O", C.c_int32)]

    def __init__(self, M=2, X=-5, E=-2, O=-4):
        super().__init__(M, X, E, O)


class DagCsr(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("pred_off", C.c_void_p), ("pred", C.c_void_p),
                ("base", C.c_void_p), ("is_sink", C.c_void_p)]


def build(verbose=False):
    """Compile libtsta_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    import subprocess
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return lib_path


_lib = None


def lib():
    """The loaded C-ABI library (raises TstaError if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(lib_path):
            raise TstaError("HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C tsta_amd/csrc`); there is no CPU fallback" % lib_path)
        L = C.CDLL(lib_path)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        L.tsta_strerror.restype = C.c_char_p; L.tsta_strerror.argtypes = [C.c_int]
        L.tsta_last_error.restype = C.c_char_p; L.tsta_last_error.argtypes = [vp]
        L.tsta_last_warning.restype = C.c_char_p; L.tsta_last_warning.argtypes = [vp]
        L.tsta_device_count.restype = C.c_int; L.tsta_device_count.argtypes = []
        L.tsta_engine_create.restype = C.c_int; L.tsta_engine_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.tsta_engine_destroy.restype = None; L.tsta_engine_destroy.argtypes = [vp]
        L.tsta_free.restype = None; L.tsta_free.argtypes = [vp]
        L.tsta_engine_set_option.restype = C.c_int; L.tsta_engine_set_option.argtypes = [vp, C.c_char_p, i64]
        L.tsta_engine_release_workspaces.restype = C.c_int; L.tsta_engine_release_workspaces.argtypes = [vp]
        L.tsta_psa_score_batch.restype = C.c_int
        L.tsta_psa_score_batch.argtypes = [vp, C.POINTER(Params), i64, vp, i64, vp, vp, vp, vp, vp]
        L.tsta_psa_score_batch_dev.restype = C.c_int
        L.tsta_psa_score_batch_dev.argtypes = [vp, C.POINTER(Params), i64, vp, vp, vp, vp, vp, vp, i32, vp]
        L.tsta_pack_2bit.restype = C.c_int
        L.tsta_pack_2bit.argtypes = [vp, i64, vp]
        L.tsta_psa_score_batch_2bit.restype = C.c_int
        L.tsta_psa_score_batch_2bit.argtypes = [vp, C.POINTER(Params), i64, vp, i64, vp, vp, vp, vp, vp]
        L.tsta_psa_align_batch.restype = C.c_int
        L.tsta_psa_align_batch.argtypes = [vp, C.POINTER(Params), i64, vp, i64, vp, vp, vp, vp, C.POINTER(Alignment)]
        L.tsta_alignments_free.restype = None
        L.tsta_alignments_free.argtypes = [C.POINTER(Alignment), i64]
        L.tsta_psa_align.restype = C.c_int
        L.tsta_psa_align.argtypes = [vp, C.POINTER(Params), vp, i64, vp, i64, C.POINTER(i32), C.POINTER(vp),
                                     C.POINTER(vp), C.POINTER(i64)]
        L.tsta_poa_fill.restype = C.c_int
        L.tsta_poa_fill.argtypes = [vp, C.POINTER(Params), C.POINTER(DagCsr), vp, i32, vp, vp, vp, vp, C.POINTER(i32)]
        L.tsta_poa_align.restype = C.c_int
        L.tsta_poa_align.argtypes = [vp, C.POINTER(Params), C.POINTER(DagCsr), vp, i32, C.POINTER(vp), C.POINTER(i64),
                                     C.POINTER(i32), vp]
        L.tsta_psa_score_batch_dev_status.restype = C.c_int; L.tsta_psa_score_batch_dev_status.argtypes = [vp, vp]
        L.tsta_engine_stat.restype = C.c_int; L.tsta_engine_stat.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
        L.tsta_last_kernel_ms.restype = C.c_int
        L.tsta_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(i32)]
        L.tsta_build_info.restype = C.c_char_p; L.tsta_build_info.argtypes = []
        L.tsta_device_uuid.restype = C.c_int; L.tsta_device_uuid.argtypes = [C.c_int, C.c_char_p]
        L.tsta_shard_bounds.restype = C.c_int; L.tsta_shard_bounds.argtypes = [i64, vp, vp, C.c_int, vp]
        L.tsta_psa_reference_exact.restype = C.c_int; L.tsta_psa_reference_exact.argtypes = [C.POINTER(Params), C.c_char_p, C.c_size_t]
        L.tsta_poa_reference_exact.restype = C.c_int; L.tsta_poa_reference_exact.argtypes = [C.POINTER(Params), i32, C.c_char_p, C.c_size_t]
        L.tsta_multi_create.restype = C.c_int; L.tsta_multi_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.tsta_multi_destroy.restype = None; L.tsta_multi_destroy.argtypes = [vp]
        L.tsta_multi_device_count.restype = C.c_int; L.tsta_multi_device_count.argtypes = [vp]
        L.tsta_multi_engine.restype = vp; L.tsta_multi_engine.argtypes = [vp, C.c_int]
        L.tsta_multi_last_error.restype = C.c_char_p; L.tsta_multi_last_error.argtypes = [vp]
        L.tsta_multi_set_option.restype = C.c_int; L.tsta_multi_set_option.argtypes = [vp, C.c_char_p, i64]
        L.tsta_psa_score_batch_multi.restype = C.c_int
        L.tsta_psa_score_batch_multi.argtypes = [vp, C.POINTER(Params), i64, vp, i64, vp, vp, vp, vp, vp]
        L.tsta_psa_align_batch_multi.restype = C.c_int
        L.tsta_psa_align_batch_multi.argtypes = [vp, C.POINTER(Params), i64, vp, i64, vp, vp, vp, vp, C.POINTER(Alignment)]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def pack_pairs(pairs):
    """[(s1, s2), ...] of uint8 arrays / bytes -> (seqs, off1, len1, off2, len2) numpy arrays for the batch ABI."""
    flat, o1, l1, o2, l2 = [], [], [], [], []
    pos = 0
    for a, b in pairs:
        a = np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else np.asarray(a, dtype=np.uint8)
        b = np.frombuffer(b, dtype=np.uint8) if isinstance(b, (bytes, bytearray)) else np.asarray(b, dtype=np.uint8)
        o1.append(pos); l1.append(len(a)); flat.append(a); pos += len(a)
        o2.append(pos); l2.append(len(b)); flat.append(b); pos += len(b)
    seqs = np.concatenate(flat) if pos else np.zeros(0, dtype=np.uint8)
    return (np.ascontiguousarray(seqs), np.array(o1, dtype=np.int64), np.array(l1, dtype=np.int32),
            np.array(o2, dtype=np.int64), np.array(l2, dtype=np.int32))


def build_info():
    """'libtsta_hip <hash> gfx950' of the loaded library (tsta_build_info)."""
    return lib().tsta_build_info().decode()


def source_hash():
    """The hash tsta_amd/csrc/Makefile computes over the sources as they are NOW in the tree (a library built from
    other sources reports a different one through build_info())."""
    import glob
    import hashlib
    src = os.path.join(_HERE, "csrc")
    with open(os.path.join(src, "Makefile")) as f:
        srcs = next(l for l in f if l.startswith("SRCS")).split(":=")[1].split()
    names = sorted(set(srcs) | {os.path.basename(h) for h in glob.glob(os.path.join(src, "*.h"))})
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(src, n), "rb") as f:
            h.update(f.read())
    with open(os.path.join(_HERE, "..", "include", "tsta_engine.h"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:16]


def device_uuid(device=0):
    buf = C.create_string_buffer(33)
    rc = lib().tsta_device_uuid(int(device), buf)
    if rc != 0:
        raise TstaError("tsta_device_uuid(%d): %s" % (device, lib().tsta_strerror(rc).decode()))
    return buf.value.decode()


def shard_bounds(len1, len2, n_shards):
    """tsta_shard_bounds: contiguous shards balanced by DP cells -> int64 array of n_shards + 1 bounds."""
    len1 = np.ascontiguousarray(len1, dtype=np.int32); len2 = np.ascontiguousarray(len2, dtype=np.int32)
    b = np.zeros(n_shards + 1, dtype=np.int64)
    z = np.zeros(1, dtype=np.int32)
    rc = lib().tsta_shard_bounds(len(len1), _ptr(len1 if len(len1) else z), _ptr(len2 if len(len2) else z), int(n_shards), _ptr(b))
    if rc != 0:
        raise TstaError("tsta_shard_bounds: %s" % lib().tsta_strerror(rc).decode())
    return b


def psa_reference_exact(params=None, **kw):
    """tsta_psa_reference_exact: (inside, why) — is the reference's PSA output the plain DP for this scoring?  No device needed."""
    params = params or Params(**kw)
    buf = C.create_string_buffer(512)
    rc = lib().tsta_psa_reference_exact(C.byref(params), buf, len(buf))
    if rc < 0:
        raise TstaError("tsta_psa_reference_exact: %s" % lib().tsta_strerror(rc).decode())
    return bool(rc), buf.value.decode()


def poa_reference_exact(params=None, ref_block=640, **kw):
    """tsta_poa_reference_exact: (plain_dp, why) — do the fast POA kernels compute what the reference prints for this scoring?"""
    params = params or Params(**kw)
    buf = C.create_string_buffer(512)
    rc = lib().tsta_poa_reference_exact(C.byref(params), int(ref_block), buf, len(buf))
    if rc < 0:
        raise TstaError("tsta_poa_reference_exact: %s" % lib().tsta_strerror(rc).decode())
    return bool(rc), buf.value.decode()


class Multi:
    """Several GPUs of one node (mirrors ``tsta_multi``): a batch is sharded over the devices, one engine each."""

    def __init__(self, devices=None, n_dev=0):
        self._h = C.c_void_p()
        arr = None
        if devices is not None:
            arr = np.ascontiguousarray(devices, dtype=np.int32); n_dev = len(arr)
        rc = lib().tsta_multi_create(_ptr(arr) if arr is not None else None, int(n_dev), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise TstaError("tsta_multi_create: %s" % lib().tsta_strerror(rc).decode())
        self.n_dev = lib().tsta_multi_device_count(self._h)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().tsta_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        # rc > 0 = TSTA_WARN_* (success; text: tsta_last_warning of engine 0); rc < 0 = error
        self.last_status = int(rc)
        if rc < 0:
            raise TstaError("%s: %s (%s)" % (what, lib().tsta_strerror(rc).decode(), lib().tsta_multi_last_error(self._h).decode()))

    def last_warning(self):
        return lib().tsta_last_warning(lib().tsta_multi_engine(self._h, 0)).decode()

    def set_option(self, key, value):
        self._check(lib().tsta_multi_set_option(self._h, key.encode(), int(value)), "tsta_multi_set_option(%s)" % key)

    def set_engine_option(self, key, value):
        for i in range(self.n_dev):
            rc = lib().tsta_engine_set_option(lib().tsta_multi_engine(self._h, i), key.encode(), int(value))
            self._check(rc, "tsta_engine_set_option(%s)" % key)

    def psa_score_batch(self, pairs, params=None):
        params = params or Params()
        out = np.zeros(len(pairs), dtype=np.int32)
        if len(pairs) == 0:
            return out
        seqs, o1, l1, o2, l2 = pack_pairs(pairs)
        sq = seqs if len(seqs) else np.zeros(1, dtype=np.uint8)
        rc = lib().tsta_psa_score_batch_multi(self._h, C.byref(params), len(pairs), _ptr(sq), len(seqs), _ptr(o1), _ptr(l1),
                                              _ptr(o2), _ptr(l2), _ptr(out))
        self._check(rc, "tsta_psa_score_batch_multi")
        return out

    def psa_align_batch(self, pairs, params=None):
        params = params or Params()
        if len(pairs) == 0:
            return []
        seqs, o1, l1, o2, l2 = pack_pairs(pairs)
        sq = seqs if len(seqs) else np.zeros(1, dtype=np.uint8)
        out = (Alignment * len(pairs))()
        rc = lib().tsta_psa_align_batch_multi(self._h, C.byref(params), len(pairs), _ptr(sq), len(seqs), _ptr(o1), _ptr(l1),
                                              _ptr(o2), _ptr(l2), out)
        self._check(rc, "tsta_psa_align_batch_multi")
        res = [(int(o.maxscore), C.string_at(o.row_a, o.len), C.string_at(o.row_b, o.len)) for o in out]
        lib().tsta_alignments_free(out, len(pairs))
        return res


class Engine:
    """One engine per GPU / host thread (mirrors ``tsta_engine`` of include/tsta_engine.h)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().tsta_engine_create(int(device), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise TstaError("tsta_engine_create(device=%d): %s" % (device, lib().tsta_strerror(rc).decode()))
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().tsta_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def release_workspaces(self):
        """Hand the engine's device workspaces back to the device (they are re-allocated on demand)."""
        self._check(lib().tsta_engine_release_workspaces(self._h), "tsta_engine_release_workspaces")

    def set_option(self, key, value):
        """Tuning knob (see tsta_engine_set_option in include/tsta_engine.h); results never depend on it."""
        self._check(lib().tsta_engine_set_option(self._h, key.encode(), int(value)), "tsta_engine_set_option(%s)" % key)

    def _check(self, rc, what):
        # rc > 0 = TSTA_WARN_*: the call succeeded (see last_status / last_warning()); rc < 0 = error
        self.last_status = int(rc)
        if rc < 0:
            raise TstaError("%s: %s (%s)" % (what, lib().tsta_strerror(rc).decode(),
                                             lib().tsta_last_error(self._h).decode()))

    def last_warning(self):
        """Text behind the last positive (TSTA_WARN_*) status of this engine."""
        return lib().tsta_last_warning(self._h).decode()

    # ---- PSA score-only batch (TSTA_psa_notrace fill for many pairs) ------------------------------
    def psa_score_batch(self, pairs, params=None):
        """pairs: list of (s1, s2) in CLI order (-1, -2).  Returns int32 array of `maxsorce` (psa.c:666)."""
        params = params or Params()
        seqs, o1, l1, o2, l2 = pack_pairs(pairs)
        out = np.zeros(len(pairs), dtype=np.int32)
        if len(pairs) == 0:
            return out
        sq = seqs if len(seqs) else np.zeros(1, dtype=np.uint8)
        rc = lib().tsta_psa_score_batch(self._h, C.byref(params), len(pairs), _ptr(sq), len(seqs), _ptr(o1), _ptr(l1),
                                        _ptr(o2), _ptr(l2), _ptr(out))
        self._check(rc, "tsta_psa_score_batch")
        return out

    def psa_score_batch_arrays(self, seqs, off1, len1, off2, len2, params=None):
        """tsta_psa_score_batch on arrays already in the batch ABI's layout (host numpy arrays: uint8 bytes, int64 offsets, int32
        lengths) -> int32 scores.  What bench.py's end-to-end leg times: host-resident shard in, scores on the host out."""
        params = params or Params()
        n = len(off1)
        out = np.zeros(n, dtype=np.int32)
        if n == 0:
            return out
        rc = lib().tsta_psa_score_batch(self._h, C.byref(params), n, _ptr(seqs), len(seqs), _ptr(off1), _ptr(len1), _ptr(off2), _ptr(len2), _ptr(out))
        self._check(rc, "tsta_psa_score_batch")
        return out

    def psa_score_batch_2bit(self, pairs, params=None):
        """Like psa_score_batch for pure-ACGT pairs, handing the engine 2-bit packed sequences (4 bases per byte)."""
        params = params or Params()
        out = np.zeros(len(pairs), dtype=np.int32)
        if len(pairs) == 0:
            return out
        chunks, b1, l1, b2, l2 = [], [], [], [], []
        pos = 0
        for a, b in pairs:
            for seq, bo, ln in ((a, b1, l1), (b, b2, l2)):
                seq = np.ascontiguousarray(np.frombuffer(seq, dtype=np.uint8) if isinstance(seq, (bytes, bytearray)) else seq, dtype=np.uint8)
                dst = np.zeros((len(seq) + 3) // 4, dtype=np.uint8)
                rc = lib().tsta_pack_2bit(_ptr(seq if len(seq) else np.zeros(1, np.uint8)), len(seq), _ptr(dst if len(dst) else np.zeros(1, np.uint8)))
                self._check(rc, "tsta_pack_2bit")
                bo.append(pos); ln.append(len(seq)); chunks.append(dst); pos += len(dst)
        packed = np.concatenate(chunks) if pos else np.zeros(1, dtype=np.uint8)
        b1 = np.array(b1, dtype=np.int64); l1 = np.array(l1, dtype=np.int32)
        b2 = np.array(b2, dtype=np.int64); l2 = np.array(l2, dtype=np.int32)
        rc = lib().tsta_psa_score_batch_2bit(self._h, C.byref(params), len(pairs), _ptr(packed), pos, _ptr(b1), _ptr(l1),
                                             _ptr(b2), _ptr(l2), _ptr(out))
        self._check(rc, "tsta_psa_score_batch_2bit")
        return out

    def psa_score_batch_dev(self, params, n_pairs, d_seqs, d_off1, d_len1, d_off2, d_len2, d_out, max_len, stream=0):
        """Device-pointer form; arguments are integer device addresses (e.g. torch ``tensor.data_ptr()``),
        `stream` a hipStream_t handle (torch: ``torch.cuda.current_stream().cuda_stream``).  Asynchronous."""
        rc = lib().tsta_psa_score_batch_dev(self._h, C.byref(params), int(n_pairs), d_seqs, d_off1, d_len1, d_off2,
                                            d_len2, d_out, int(max_len), C.c_void_p(stream or None))
        self._check(rc, "tsta_psa_score_batch_dev")

    def psa_score_batch_dev_status(self, stream=0):
        """Synchronises `stream`; raises if the last psa_score_batch_dev's single-launch band pipeline abandoned a wait
        (its scores are then INT32_MIN).  See tsta_psa_score_batch_dev_status in include/tsta_engine.h."""
        self._check(lib().tsta_psa_score_batch_dev_status(self._h, C.c_void_p(stream or None)), "tsta_psa_score_batch_dev_status")

    def stat(self, key):
        v = C.c_int64()
        self._check(lib().tsta_engine_stat(self._h, key.encode(), C.byref(v)), "tsta_engine_stat(%s)" % key)
        return int(v.value)

    # ---- PSA with traceback (TSTA_psa) --------------------------------------------------------------
    def psa_align(self, s1, s2, params=None):
        """-> (maxsorce, row_a, row_b): the two gapped rows as TSTA_psa writes them (row_a = longer input)."""
        params = params or Params()
        s1 = np.ascontiguousarray(np.frombuffer(s1, dtype=np.uint8) if isinstance(s1, (bytes, bytearray)) else s1,
                                  dtype=np.uint8)
        s2 = np.ascontiguousarray(np.frombuffer(s2, dtype=np.uint8) if isinstance(s2, (bytes, bytearray)) else s2,
                                  dtype=np.uint8)
        ms, ra, rb, ln = C.c_int32(), C.c_void_p(), C.c_void_p(), C.c_int64()
        p1 = s1 if len(s1) else np.zeros(1, dtype=np.uint8)
        p2 = s2 if len(s2) else np.zeros(1, dtype=np.uint8)
        rc = lib().tsta_psa_align(self._h, C.byref(params), _ptr(p1), len(s1), _ptr(p2), len(s2), C.byref(ms),
                                  C.byref(ra), C.byref(rb), C.byref(ln))
        self._check(rc, "tsta_psa_align")
        a = C.string_at(ra, ln.value); b = C.string_at(rb, ln.value)
        lib().tsta_free(ra); lib().tsta_free(rb)
        return int(ms.value), a, b

    def psa_align_batch(self, pairs, params=None):
        """pairs: list of (s1, s2) in CLI order.  -> list of (maxsorce, row_a, row_b), one per pair, as a loop of
        TSTA_psa runs would write them; the pairs' fills are in flight together on the device."""
        params = params or Params()
        if len(pairs) == 0:
            return []
        seqs, o1, l1, o2, l2 = pack_pairs(pairs)
        sq = seqs if len(seqs) else np.zeros(1, dtype=np.uint8)
        out = (Alignment * len(pairs))()
        rc = lib().tsta_psa_align_batch(self._h, C.byref(params), len(pairs), _ptr(sq), len(seqs), _ptr(o1), _ptr(l1),
                                        _ptr(o2), _ptr(l2), out)
        self._check(rc, "tsta_psa_align_batch")
        res = [(int(o.maxscore), C.string_at(o.row_a, o.len), C.string_at(o.row_b, o.len)) for o in out]
        lib().tsta_alignments_free(out, len(pairs))
        return res

    # ---- POA fill (one control() wavefront of TSTA_msa) ---------------------------------------------
    def poa_fill(self, pred_off, pred, base, is_sink, seq, params=None, want_trace=True):
        """-> (best_sink, lastscore[V], source, esource, fsource) with int8 [V][n] trace arrays (or None)."""
        params = params or Params()
        pred_off = np.ascontiguousarray(pred_off, dtype=np.int32)
        pred = np.ascontiguousarray(pred if len(pred) else np.zeros(1), dtype=np.int32)
        base = np.ascontiguousarray(base, dtype=np.uint8)
        is_sink = np.ascontiguousarray(is_sink, dtype=np.uint8)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        V, n = len(base), len(seq)
        dag = DagCsr(V, _ptr(pred_off).value, _ptr(pred).value, _ptr(base).value, _ptr(is_sink).value)
        last = np.zeros(V, dtype=np.int32)
        best = C.c_int32(-1)
        src = es = fs = None
        if want_trace:
            src = np.zeros((V, n), dtype=np.int8); es = np.zeros((V, n), dtype=np.int8); fs = np.zeros((V, n), dtype=np.int8)
        rc = lib().tsta_poa_fill(self._h, C.byref(params), C.byref(dag), _ptr(seq), n,
                                 _ptr(src) if want_trace else None, _ptr(es) if want_trace else None,
                                 _ptr(fs) if want_trace else None, _ptr(last), C.byref(best))
        self._check(rc, "tsta_poa_fill")
        return int(best.value), last, src, es, fs

    def poa_align(self, pred_off, pred, base, is_sink, seq, params=None):
        """Fill + on-device traceback walk.  -> (steps, end_col, lastscore); steps is a structured array with fields
        node, col, code, peek (tsta_poa_step)."""
        params = params or Params()
        pred_off = np.ascontiguousarray(pred_off, dtype=np.int32)
        pred = np.ascontiguousarray(pred if len(pred) else np.zeros(1), dtype=np.int32)
        base = np.ascontiguousarray(base, dtype=np.uint8)
        is_sink = np.ascontiguousarray(is_sink, dtype=np.uint8)
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        V, n = len(base), len(seq)
        dag = DagCsr(V, _ptr(pred_off).value, _ptr(pred).value, _ptr(base).value, _ptr(is_sink).value)
        last = np.zeros(V, dtype=np.int32)
        st, ns, ec = C.c_void_p(), C.c_int64(), C.c_int32()
        rc = lib().tsta_poa_align(self._h, C.byref(params), C.byref(dag), _ptr(seq), n, C.byref(st), C.byref(ns),
                                  C.byref(ec), _ptr(last))
        self._check(rc, "tsta_poa_align")
        dt = np.dtype([("node", np.int32), ("col", np.int32), ("code", np.int8), ("peek", np.int8), ("pad", np.int16)])
        steps = np.frombuffer(C.string_at(st, ns.value * dt.itemsize), dtype=dt).copy() if ns.value else np.zeros(0, dt)
        lib().tsta_free(st)
        return steps, int(ec.value), last

    def last_kernel_ms(self):
        ms, nl = C.c_float(), C.c_int32()
        self._check(lib().tsta_last_kernel_ms(self._h, C.byref(ms), C.byref(nl)), "tsta_last_kernel_ms")
        return float(ms.value), int(nl.value)
