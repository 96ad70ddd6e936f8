"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every symbol
include/tsta_engine.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

import tsta_amd
from tests import util


@pytest.fixture(scope="module")
def built_lib():
    tsta_amd.build()
    return ctypes.CDLL(tsta_amd.lib_path)


def header_symbols():
    with open(os.path.join(util.ROOT, "include", "tsta_engine.h")) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tsta_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_listed():
    assert header_symbols() == sorted(tsta_amd.EXPORTED_SYMBOLS)


def test_library_exports_every_header_symbol(built_lib):
    for s in header_symbols():
        assert hasattr(built_lib, s), s


def test_strerror_and_no_cpu_fallback(built_lib):
    L = tsta_amd.lib()
    assert L.tsta_strerror(0) == b"ok"
    assert b"no CPU fallback" in L.tsta_strerror(-2)
    import torch
    if not torch.cuda.is_available():
        # without a GPU the product path must fail loudly, not fall back to any CPU code
        with pytest.raises(tsta_amd.TstaError):
            tsta_amd.Engine(0)


def test_product_does_not_import_oracle():
    # the oracle is test infrastructure: nothing under tsta_amd/ or include/ may reference it
    for root in ("tsta_amd", "include"):
        for dp, _, files in os.walk(os.path.join(util.ROOT, root)):
            for fn in files:
                if fn.endswith((".py", ".h", ".hip", ".c", ".cpp")) or fn == "Makefile":
                    with open(os.path.join(dp, fn), errors="ignore") as f:
                        body = f.read()
                    assert "liboracle" not in body and "oracle/" not in body and "oracle_psa" not in body, (dp, fn)


def test_pack_2bit_is_a_host_function():
    # tsta_pack_2bit needs no GPU: 4 bases per byte, base i in bits 2(i%4), codes A 0, C 1, T 2, G 3
    import ctypes as C
    import numpy as np
    lib = tsta_amd.lib()
    src = np.frombuffer(b"ACGTTGCAAC", dtype=np.uint8)
    dst = np.zeros(3, dtype=np.uint8)
    assert lib.tsta_pack_2bit(src.ctypes.data_as(C.c_void_p), len(src), dst.ctypes.data_as(C.c_void_p)) == 0
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    want = [sum(code[ch] << (2 * j) for j, ch in enumerate("ACGTTGCAAC"[i:i + 4])) for i in range(0, 10, 4)]
    assert dst.tolist() == want
    bad = np.frombuffer(b"ACNT", dtype=np.uint8)
    assert lib.tsta_pack_2bit(bad.ctypes.data_as(C.c_void_p), 4, dst.ctypes.data_as(C.c_void_p)) == -5      # TSTA_ERR_UNSUPPORTED


def test_build_info_matches_the_sources_in_the_tree(built_lib):
    # a stale libtsta_hip.so (built from other sources than the ones next to it) shows here and on the GPU box
    info = tsta_amd.build_info()
    assert info == "libtsta_hip %s gfx950" % tsta_amd.source_hash(), info
    with open(os.path.join(util.ROOT, "tsta_amd", "lib", "BUILD_INFO")) as f:
        assert f.read().strip() == info


def test_shard_bounds_is_a_host_function():
    # tsta_shard_bounds (multi-GPU partition of a batch): contiguous, covers the batch, balanced by DP cells
    import numpy as np
    for n in (0, 1, 7, 8192, 65536):
        for w in (1, 2, 3, 8):
            ln = np.full(n, 50_000, dtype=np.int32)
            b = tsta_amd.shard_bounds(ln, ln, w)
            assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0)
            assert np.diff(b).max() - np.diff(b).min() <= 1, (n, w, b)
    rng = np.random.default_rng(5)
    l1 = rng.integers(0, 60_000, 1000).astype(np.int32); l2 = rng.integers(0, 60_000, 1000).astype(np.int32)
    cells = l1.astype(np.int64) * l2 + 1
    for w in (2, 5, 8):
        b = tsta_amd.shard_bounds(l1, l2, w)
        per = np.array([cells[b[s]:b[s + 1]].sum() for s in range(w)])
        assert b[0] == 0 and b[-1] == 1000
        assert per.max() - per.min() <= 2 * cells.max(), (w, per)
    assert tsta_amd.lib().tsta_shard_bounds(1, None, None, 2, None) == -1        # TSTA_ERR_INVALID_ARG
