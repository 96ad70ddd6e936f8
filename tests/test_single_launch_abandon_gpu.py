"""GPU: every single-launch family (K1b band pipeline, the three trace fills, the POA dataflow fill) behaves the same way
when a wave abandons a bounded wait: no entry point returns success with a wrong number.  Engine option "test_abandon" = t
makes the holder of ticket t-1 compute its work but never publish it, so that its dependants really run into the bound
(handoff.h); the host-pointer forms must then run the call again through the launch-per-diagonal form and return what the
oracle says, the device-pointer form must poison its scores and report through tsta_psa_score_batch_dev_status.
Reference semantics of the values compared: psa/psa.c:416-424,590 (maxsorce), :435-518 (trace), msa/c-t-simd.c:737-957."""
import numpy as np
import pytest

import tsta_amd
from tests import util
from tests.test_poa_fill_gpu import FIXTURES, load

pytestmark = pytest.mark.gpu

INT32_MIN = -(1 << 31)


@pytest.fixture()
def eng():
    e = tsta_amd.Engine(0)
    yield e
    e.set_option("test_abandon", 0)
    e.close()


def _small_batch():
    # K1b's domain: fewer couples than 2 x SIMDs, sequences of several bands
    pairs = []
    for i in range(24):
        a = util.random_dna(7000 + i, 5000 - 13 * i)
        pairs.append((a, util.mutate(a, 7100 + i, 0.06) if i % 2 else util.random_dna(7200 + i, 4800)))
    return pairs


def test_band_pipeline_host_form_runs_again(eng):
    pairs = _small_batch()
    want = util.oracle_maxscore_batch(pairs)
    assert np.array_equal(eng.psa_score_batch(pairs), want) and eng.stat("single_launch_reruns") == 0
    for ticket in (1, 7, 40):
        before = eng.stat("single_launch_reruns")
        eng.set_option("test_abandon", ticket)
        got = eng.psa_score_batch(pairs)
        assert np.array_equal(got, want), (ticket, got.tolist(), want.tolist())
        assert eng.stat("single_launch_reruns") == before + 1, ticket
    eng.set_option("test_abandon", 0)
    assert np.array_equal(eng.psa_score_batch_2bit(pairs), want)
    eng.set_option("test_abandon", 5)
    before = eng.stat("single_launch_reruns")
    assert np.array_equal(eng.psa_score_batch_2bit(pairs), want)
    assert eng.stat("single_launch_reruns") == before + 1


class _Hip:
    """Device buffers through the HIP runtime libtsta_hip.so itself is linked against (no second runtime in the process)."""
    def __init__(self):
        import ctypes as C
        self.C = C
        self.rt = C.CDLL("libamdhip64.so")
        self.bufs = []

    def to_device(self, a):
        C = self.C
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), C.c_size_t(max(a.nbytes, 16))) == 0
        assert self.rt.hipMemcpy(p, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0
        self.bufs.append(p)
        return p.value

    def to_host(self, ptr, like):
        out = np.empty_like(like)
        assert self.rt.hipMemcpy(self.C.c_void_p(out.ctypes.data), self.C.c_void_p(ptr), self.C.c_size_t(out.nbytes), 2) == 0
        return out

    def free(self):
        for p in self.bufs:
            self.rt.hipFree(p)


def test_band_pipeline_device_form_poisons(eng):
    pairs = _small_batch()
    seqs, o1, l1, o2, l2 = tsta_amd.pack_pairs(pairs)
    hip = _Hip()
    try:
        d_seqs, d_o1, d_l1, d_o2, d_l2 = [hip.to_device(a) for a in (seqs, o1, l1, o2, l2)]
        host_out = np.zeros(len(pairs), dtype=np.int32)
        d_out = hip.to_device(host_out)
        max_len = int(max(l1.max(), l2.max()))
        args = (tsta_amd.Params(), len(pairs), d_seqs, d_o1, d_l1, d_o2, d_l2, d_out, max_len)
        eng.psa_score_batch_dev(*args)
        eng.psa_score_batch_dev_status()
        want = util.oracle_maxscore_batch(pairs)
        assert np.array_equal(hip.to_host(d_out, host_out), want)
        eng.set_option("test_abandon", 3)
        eng.psa_score_batch_dev(*args)
        with pytest.raises(tsta_amd.TstaError):
            eng.psa_score_batch_dev_status()
        got = hip.to_host(d_out, host_out)
        # every couple of the band pipeline is poisoned; a pair the int32 kernel scored (odd one out) keeps its true score
        assert np.all((got == INT32_MIN) | (got == want)) and np.count_nonzero(got == INT32_MIN) >= len(pairs) - 2, got.tolist()
        eng.set_option("test_abandon", 0)
        eng.psa_score_batch_dev(*args)
        eng.psa_score_batch_dev_status()
        assert np.array_equal(hip.to_host(d_out, host_out), want)
    finally:
        hip.free()


@pytest.mark.parametrize("variant", ["split", "quad", "memory", "rows4"])
def test_single_pair_trace_fill_runs_again(eng, variant):
    # tsta_psa_align: psa_trace_split_kernel (solver bands + emitter tiles) / psa_trace_quad_kernel (four bands per
    # workgroup, LDS-linked) / psa_trace_tile_kernel<.., PERSIST>
    a = util.random_dna(31, 9000); b = util.mutate(a, 32, 0.08)[:8200]
    o = util.oracle_align(a, b)
    if variant == "quad":
        eng.set_option("trace_split", 0)
    if variant == "memory":
        eng.set_option("trace_lds", 0)
    if variant == "rows4":
        eng.set_option("trace_rows", 4)
    assert eng.psa_align(a, b) == (o[0], o[2], o[3]) and eng.stat("single_launch_reruns") == 0
    eng.set_option("test_abandon", 2)
    assert eng.psa_align(a, b) == (o[0], o[2], o[3])
    assert eng.stat("single_launch_reruns") == 1


def test_coupled_trace_fill_runs_again(eng):
    # tsta_psa_align_batch: K2p in dataflow form (psa_trace_packed_tile_kernel<.., true>) + the odd job out beside it
    pairs = []
    for i in range(17):
        a = util.random_dna(9300 + i, 2500 + 97 * (i % 5))
        pairs.append((a, util.mutate(a, 9400 + i, 0.1)))
    eng.set_option("trace_rows", 4)                              # (couples exist at 4 rows per lane only; auto picks 2 for so few bands)
    want = eng.psa_align_batch(pairs)
    assert eng.stat("single_launch_reruns") == 0
    for (ms, ra, rb), (a, b) in list(zip(want, pairs))[::4]:
        o = util.oracle_align(a, b)
        assert (ms, ra, rb) == (o[0], o[2], o[3])
    eng.set_option("test_abandon", 4)
    assert eng.psa_align_batch(pairs) == want
    assert eng.stat("single_launch_reruns") >= 1


def test_poa_dataflow_chunk_that_never_publishes(eng):
    # K4d: chunk 0 (ticket 1) never publishes a block; chunk 1's fetcher runs into its bound, the launch drains, the walker
    # refuses to start on the arrays, and the entry points run the call again through K4 (poa_fill.hip: poa_with_fallback)
    z, sc = load([f for f in FIXTURES if f.endswith("poa_medium_step2.npz")][0])
    args = (z["pred_off"], z["pred"], z["base"], z["is_sink"], z["seq"])
    assert len(z["seq"]) > 256                                   # at least three chunks
    want = eng.poa_fill(*args, tsta_amd.Params(**sc))
    want_steps, want_end, want_last = eng.poa_align(*args, tsta_amd.Params(**sc))
    eng.set_option("test_abandon", 2)
    got = eng.poa_fill(*args, tsta_amd.Params(**sc))
    steps, end_col, last = eng.poa_align(*args, tsta_amd.Params(**sc))
    assert got[0] == want[0] and all(np.array_equal(g, w) for g, w in zip(got[1:], want[1:]))
    assert end_col == want_end and np.array_equal(steps, want_steps) and np.array_equal(last, want_last)
    assert eng.stat("single_launch_reruns") == 2


def test_blocked_poa_fill_block_that_never_publishes(eng):
    # K4e (poa_ref8.hip, sequences with a literal 'N'): block 0 never publishes a row; block 1 runs into its bound, the launch
    # drains, and the entry point runs the call again with one launch per block
    from tests.test_poa_fill_gpu import random_dag, sprinkle_n
    pred_off, pred, base, is_sink, seq = random_dag(31, 500, 4, 1500)
    seq = sprinkle_n(seq, 31, (700, 701), 4)
    eng.set_option("poa_ref_block", 64 * 5)                      # 320 columns per block: five blocks
    want = util.oracle_poa_fill_i8(pred_off, pred, base, is_sink, seq, 5)
    got = eng.poa_fill(pred_off, pred, base, is_sink, seq)
    assert got[0] == want[0] and all(np.array_equal(g, w) for g, w in zip(got[1:], want[1:])) and eng.stat("single_launch_reruns") == 0
    eng.set_option("test_abandon", 1)
    got = eng.poa_fill(pred_off, pred, base, is_sink, seq)
    assert got[0] == want[0] and all(np.array_equal(g, w) for g, w in zip(got[1:], want[1:]))
    assert eng.stat("single_launch_reruns") == 1
