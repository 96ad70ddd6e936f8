// engine.hip — C-ABI (include/tsta_engine.h) over the HIP kernels.  Host side of libtsta_hip.so.
// No CPU fallback anywhere: without a usable gfx950 device every entry point fails with TSTA_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/tsta_engine.h"
#include "engine.h"
#include "kernels.h"

namespace tsta {

void Engine::set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, sizeof(err), fmt, ap);
    va_end(ap);
}

int Engine::copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, const char* what)
{
    int rc = check(hipMemcpyAsync(dst, src, bytes, kind, stream), what);
    return rc ? rc : check(hipStreamSynchronize(stream), what);
}

int Engine::check(hipError_t e, const char* what)
{
    if (e == hipSuccess) return TSTA_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? TSTA_ERR_OOM : TSTA_ERR_HIP;
}

// the status words of the last single-launch fills live inside workspaces: a buffer that is freed (or regrown) takes them along,
// and a later tsta_psa_score_batch_dev_status() / walker must not read through the stale pointer
void Engine::forget_status_in(const DevBuf& b)
{
    const char* lo = static_cast<const char*>(b.p); const char* hi = lo + b.bytes;
    auto inside = [&](const void* q) { return q && static_cast<const char*>(q) >= lo && static_cast<const char*>(q) < hi; };
    if (inside(psa_status)) psa_status = nullptr;
    if (inside(poa_status)) poa_status = nullptr;
}

int Engine::reserve(DevBuf& b, size_t bytes, const char* what, bool growing)
{
    if (b.bytes >= bytes && b.p) return TSTA_OK;
    const bool regrow = b.p != nullptr;
    if (b.p) { forget_status_in(b); (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
    // `growing`: the caller's need rises from call to call (the POA graph gains nodes with every sequence of an MSA), and a
    // device allocation costs ~35 ms per GB: a buffer that had to grow once gets 50 % headroom instead of 12.5 %
    size_t want = bytes + ((growing && regrow) ? bytes / 2 : bytes / 8) + 256;
    int rc = check(hipMalloc(&b.p, want), what);
    if (rc != TSTA_OK) { b.p = nullptr; return rc; }
    b.bytes = want;
    return TSTA_OK;
}

// Where the reference's PSA output IS the plain DP the kernels compute (tsta_engine.h: tsta_psa_reference_exact).  The reference
// keeps every quantity of row() as an int8 DIFFERENCE against H(i-1,j-1) and adds / subtracts with wrap-around
// (mm_add_epi8 / mm_sub_epi8, psa.c:244-246,285-327); its "minus infinity" is MIN = -100 (psa.c:110).  With g = O + E,
// S = max(M, 0), s_min = min(M, X), the true ranges of those quantities are
//   e, f (gap states)            in [2g, S]          psa.c:225-231 (init 2E+O, 2(E+O)), :324, :310
//   h1 = s + ogap                in [max(s_min, 2g) + g, S + g]                                    :301
//   t, v (horizontal / vertical) in [g, S - g]                                                       :291, :329
// and each condition below says that one statement of row() stays inside int8 (or that MIN really acts as minus infinity);
// the first one that fails is reported.  Conservative by construction, and checked against the compiled reference (3 ISAs, two
// -W) on the grid of tests/golden/psa_domain_golden.json: no in-domain case differs, and every scanned out-of-domain case does.
bool psa_reference_exact(const tsta_params* p, char* why, size_t cap)
{
    const int M = p->M, X = p->X, E = p->E, O = p->O;
    const long long g = (long long)O + E, S = std::max(M, 0), smin = std::min(M, X);
    const char* w = nullptr;
    char buf[200];
    if (E > 0 || O > 0)
        w = "a positive gap score: the reference's lazy-F pass (psa.c:266-274) assumes E, O <= 0";
    else if (X > 0)
        w = "a positive mismatch score: the reference's maximum runs over its padded columns too (psa.c:333-357), which then outscore the real cells";
    else if (E < -28)
        w = "E < -28: MIN + E = -100 + E wraps around in int8 in the first pass (psa.c:110,260)";
    else if (2 * g + E < -128)
        w = "2(O+E)+E < -128: a gap state at its minimum 2(O+E) wraps when extended (psa.c:260,303,319)";
    else if (S - g > 127)
        w = "max(M,0)-O-E > 127: the horizontal / vertical difference M-(O+E) does not fit int8 (psa.c:291,329)";
    else if (smin < -128)
        w = "min(M,X) < -128: the substitution score itself does not fit int8 (psa.c:242-243)";
    else if (std::max(smin, 2 * g) + g < -128)
        w = "max(min(M,X),2(O+E))+O+E < -128: H+O+E against the diagonal wraps (psa.c:259,301)";
    else if (2 * g < -100 && smin + O < -100)
        w = "2(O+E) < -100 and min(M,X)+O < -100: real gap states lie below the reference's MIN = -100 (psa.c:110,247)";
    if (w && why && cap) {
        snprintf(buf, sizeof(buf), "M=%d X=%d E=%d O=%d: %s", M, X, E, O, w);
        snprintf(why, cap, "scoring outside the domain where the reference's int8 difference encoding is exact (%s): the reference's own "
                 "output there depends on its SIMD width and -W; this result is the plain affine-gap DP", buf);
    } else if (why && cap) why[0] = 0;
    return w == nullptr;
}

int psa_reference_warning(Engine* e, const tsta_params* p)
{
    return psa_reference_exact(p, e->warn, sizeof(e->warn)) ? TSTA_OK : TSTA_WARN_REFERENCE_UNDEFINED;
}

void Engine::timing_begin(hipStream_t s) { timing_stream = s; n_launches = 0; if (!timing_hold) (void)hipEventRecord(ev0, s); }
void Engine::timing_end(hipStream_t s) { (void)hipEventRecord(ev1, s); timing_valid = true; }

}  // namespace tsta

using tsta::Engine;

extern "C" {

const char* tsta_strerror(int status)
{
    switch (status) {
        case TSTA_OK: return "ok";
        case TSTA_WARN_REFERENCE_UNDEFINED: return "ok, but the reference's own output on this input depends on -W / SIMD width";
        case TSTA_ERR_INVALID_ARG: return "invalid argument";
        case TSTA_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
        case TSTA_ERR_HIP: return "HIP runtime error";
        case TSTA_ERR_OOM: return "out of device memory";
        case TSTA_ERR_UNSUPPORTED: return "input outside the supported domain";
        case TSTA_ERR_INTERNAL: return "internal error";
        default: return "unknown status";
    }
}

const char* tsta_last_error(const tsta_engine* eng)
{
    return eng ? reinterpret_cast<const Engine*>(eng)->err : "";
}

#ifndef TSTA_SRC_HASH
#error "build through tsta_amd/csrc/Makefile: it passes -DTSTA_SRC_HASH (SHA-256 of the sources)"
#endif
const char* tsta_build_info(void) { return "libtsta_hip " TSTA_SRC_HASH " gfx950"; }

int tsta_device_uuid(int device, char* out33)
{
    if (!out33) return TSTA_ERR_INVALID_ARG;
    out33[0] = 0;
    const int n = tsta_device_count();
    if (n <= 0) return TSTA_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return TSTA_ERR_INVALID_ARG;
    hipUUID u;
    if (hipDeviceGetUuid(&u, device) != hipSuccess) { (void)hipGetLastError(); return TSTA_ERR_HIP; }
    for (int i = 0; i < 16; ++i) snprintf(out33 + 2 * i, 3, "%02x", (unsigned)(unsigned char)u.bytes[i]);
    return TSTA_OK;
}

int tsta_psa_reference_exact(const tsta_params* p, char* why, size_t why_bytes)
{
    if (!p) return TSTA_ERR_INVALID_ARG;
    return tsta::psa_reference_exact(p, why, why_bytes) ? 1 : 0;
}

const char* tsta_last_warning(const tsta_engine* eng)
{
    return eng ? reinterpret_cast<const Engine*>(eng)->warn : "";
}

int tsta_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); return TSTA_ERR_NO_DEVICE; }
    return n;
}

int tsta_engine_create(int device, tsta_engine** out)
{
    if (!out) return TSTA_ERR_INVALID_ARG;
    *out = nullptr;
    int n = tsta_device_count();
    if (n <= 0) return TSTA_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return TSTA_ERR_INVALID_ARG;
    Engine* e = new (std::nothrow) Engine();
    if (!e) return TSTA_ERR_OOM;
    e->device = device;
    if (const char* v = getenv("TSTA_PSA_PACKED")) e->opt_psa_packed = atoi(v);
    if (const char* v = getenv("TSTA_PSA_ELIDE")) e->opt_psa_elide = atoi(v) != 0;
    if (const char* v = getenv("TSTA_TRACE_PERSISTENT")) e->opt_trace_persistent = atoi(v) != 0;
    if (const char* v = getenv("TSTA_TRACE_STREAMS")) e->opt_trace_streams = atoi(v);
    if (const char* v = getenv("TSTA_PSA_BANDS")) e->opt_psa_bands = atoi(v);
    if (const char* v = getenv("TSTA_PSA_PERM")) e->opt_psa_perm = atoi(v) != 0;
    if (const char* v = getenv("TSTA_PSA_F16MAX")) e->opt_psa_f16max = std::max(0, std::min(4, atoi(v)));
    if (const char* v = getenv("TSTA_WALK_TABLE")) e->opt_walk_table = std::max(0, std::min(2, atoi(v)));
    if (const char* v = getenv("TSTA_PSA_R")) e->opt_psa_rows = atoi(v);
    if (const char* v = getenv("TSTA_TRACE_R")) e->opt_trace_rows = atoi(v);
    if (const char* v = getenv("TSTA_TRACE_T")) e->opt_trace_tile = atoi(v);
    if (const char* v = getenv("TSTA_POA_DATAFLOW")) e->opt_poa_dataflow = atoi(v) != 0;
    int rc;
    hipDeviceProp_t prop;
    if ((rc = e->check(hipSetDevice(device), "hipSetDevice")) != TSTA_OK ||
        (rc = e->check(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties")) != TSTA_OK) {
        delete e; return rc;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {       // code objects are gfx950-only
        delete e; return TSTA_ERR_NO_DEVICE;
    }
    e->n_cu = prop.multiProcessorCount;
    if ((rc = e->check(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking), "hipStreamCreate")) != TSTA_OK ||
        (rc = e->check(hipEventCreate(&e->ev0), "hipEventCreate")) != TSTA_OK ||
        (rc = e->check(hipEventCreate(&e->ev1), "hipEventCreate")) != TSTA_OK ||
        (rc = e->reserve(e->counter, 4096, "hipMalloc(counter)")) != TSTA_OK) {
        tsta_engine_destroy(reinterpret_cast<tsta_engine*>(e)); return rc;
    }
    // hardware contract check: DPP whole-wave shifts behave as wave_ops.h assumes
    int* d = reinterpret_cast<int*>(e->counter.p);
    tsta::launch_dpp_selftest(d, e->stream);
    int h[704];
    if ((rc = e->check(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, e->stream), "selftest copy")) != TSTA_OK ||
        (rc = e->check(hipStreamSynchronize(e->stream), "selftest sync")) != TSTA_OK) {
        tsta_engine_destroy(reinterpret_cast<tsta_engine*>(e)); return rc;
    }
    for (int l = 0; l < 64; ++l) {
        const int ea = l == 0 ? 1000 : l - 1, eb = l == 63 ? 2063 : l + 1;
        // lane_exchange: cch/hu/eu shift down with injection at lane 0, sh/se shift up with injection at lane 63,
        // ach/th/te rotate up (lane 63 keeps its own value)
        const int x_cch = l == 0 ? 400 : 100 + l - 1, x_hu = l == 0 ? 500 : 700 + l - 1, x_eu = l == 0 ? 600 : 800 + l - 1;
        const int x_sh = l == 63 ? 763 : 200 + l + 1, x_se = l == 63 ? 863 : 300 + l + 1;
        const int x_ach = l == 63 ? 463 : 400 + l + 1, x_th = l == 63 ? 563 : 500 + l + 1, x_te = l == 63 ? 663 : 600 + l + 1;
        if (h[l] != ea || h[64 + l] != eb || h[128 + l] != x_cch || h[192 + l] != x_hu || h[256 + l] != x_eu ||
            h[640 + l] != (int)(0x00330000u | ((0x88776655u + (unsigned)l) >> 8 & 0xffu)) || h[320 + l] != x_sh || h[384 + l] != x_se || h[448 + l] != x_ach || h[512 + l] != x_th || h[576 + l] != x_te) {
            e->set_error("DPP wave shift self-test failed at lane %d", l);
            tsta_engine_destroy(reinterpret_cast<tsta_engine*>(e));
            return TSTA_ERR_INTERNAL;
        }
    }
    *out = reinterpret_cast<tsta_engine*>(e);
    return TSTA_OK;
}

void tsta_engine_destroy(tsta_engine* eng)
{
    if (!eng) return;
    Engine* e = reinterpret_cast<Engine*>(eng);
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (tsta::DevBuf* b : {&e->counter, &e->rows, &e->in_seqs, &e->in_meta, &e->out_scores, &e->trace, &e->scratch,
                            &e->scratch2, &e->pair_flags, &e->unpacked, &e->unpacked2, &e->ckpt, &e->walk_tab}) {
        if (b->p) (void)hipFree(b->p);
    }
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    for (int g = 0; g < 3; ++g) { if (e->side[g]) (void)hipStreamDestroy(e->side[g]); if (e->ev_join[g]) (void)hipEventDestroy(e->ev_join[g]); }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    for (hipEvent_t ev : e->ev_up) if (ev) (void)hipEventDestroy(ev);
    if (e->ev_tab) (void)hipEventDestroy(e->ev_tab);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

void tsta_free(void* p) { free(p); }

int tsta_engine_release_workspaces(tsta_engine* eng)
{
    if (!eng) return TSTA_ERR_INVALID_ARG;
    Engine* e = reinterpret_cast<Engine*>(eng);
    int rc = e->check(hipSetDevice(e->device), "hipSetDevice");
    if (rc != TSTA_OK) return rc;
    if ((rc = e->check(hipStreamSynchronize(e->stream), "hipStreamSynchronize")) != TSTA_OK) return rc;
    // everything but the small counter block (ticket words, self-test / result scalars)
    for (tsta::DevBuf* b : {&e->rows, &e->in_seqs, &e->in_meta, &e->out_scores, &e->trace, &e->scratch, &e->scratch2,
                            &e->pair_flags, &e->unpacked, &e->unpacked2, &e->ckpt, &e->walk_tab}) {
        if (b->p) { e->forget_status_in(*b); (void)hipFree(b->p); }
        if (b == &e->pair_flags) { e->last_redo = nullptr; e->last_redo_n = 0; }
        b->p = nullptr; b->bytes = 0;
    }
    return TSTA_OK;
}

int tsta_engine_set_option(tsta_engine* eng, const char* key, int64_t value)
{
    if (!eng || !key) return TSTA_ERR_INVALID_ARG;
    Engine* e = reinterpret_cast<Engine*>(eng);
    const int v = (int)value;
    if (!strcmp(key, "psa_packed") && (v == -1 || v == 0 || v == 8 || v == 16 || v == 32)) { e->opt_psa_packed = v; return TSTA_OK; }
    if (!strcmp(key, "psa_rows") && (v == 0 || v == 4 || v == 8 || v == 16)) { e->opt_psa_rows = v; return TSTA_OK; }
    if (!strcmp(key, "trace_rows") && (v == 0 || v == 1 || v == 2 || v == 4)) { e->opt_trace_rows = v; return TSTA_OK; }
    if (!strcmp(key, "psa_elide") && (v == 0 || v == 1)) { e->opt_psa_elide = v; return TSTA_OK; }
    if (!strcmp(key, "trace_persistent") && (v == 0 || v == 1)) { e->opt_trace_persistent = v; return TSTA_OK; }
    if (!strcmp(key, "trace_streams") && (v == 0 || v == 1)) { e->opt_trace_streams = v; return TSTA_OK; }
    if (!strcmp(key, "trace_packed") && (v == 0 || v == 1)) { e->opt_trace_packed = v; return TSTA_OK; }
    if (!strcmp(key, "psa_bands") && (v == -1 || v == 0 || v == 8 || v == 16)) { e->opt_psa_bands = (int)v; return TSTA_OK; }
    if (!strcmp(key, "psa_ffree") && (v == 0 || v == 1)) { e->opt_psa_ffree = v; return TSTA_OK; }
    if (!strcmp(key, "psa_perm") && (v == 0 || v == 1)) { e->opt_psa_perm = v; return TSTA_OK; }
    if (!strcmp(key, "psa_overlap") && (v == 0 || v == 1)) { e->opt_psa_overlap = v; return TSTA_OK; }
    if (!strcmp(key, "psa_f16max") && v >= 0 && v <= 4) { e->opt_psa_f16max = v; return TSTA_OK; }
    if (!strcmp(key, "poa_cols") && (v == 0 || v == 2 || v == 4)) { e->opt_poa_cols = v; return TSTA_OK; }
    if (!strcmp(key, "align_ws_gb") && v >= 0 && v <= 4096) { e->opt_align_ws_gb = v; return TSTA_OK; }
    if (!strcmp(key, "trace_lds") && (v == 0 || v == 1 || v == 2)) { e->opt_trace_lds = v; return TSTA_OK; }
    if (!strcmp(key, "poa_dataflow") && (v == 0 || v == 1 || v == 2)) { e->opt_poa_dataflow = v; return TSTA_OK; }
    if ((!strcmp(key, "poa_ref_arith") || !strcmp(key, "poa_n_blocked")) && (v == 0 || v == 1 || v == 2)) { e->opt_poa_n_blocked = v; return TSTA_OK; }
    if (!strcmp(key, "trace_split") && (v == 0 || v == 1 || v == 2)) { e->opt_trace_split = v; return TSTA_OK; }
    if (!strcmp(key, "walk_table") && v >= 0 && v <= 2) { e->opt_walk_table = v; return TSTA_OK; }
    if (!strcmp(key, "walk_corridor") && v >= 0 && v <= 65536 && v % 256 == 0) { e->opt_walk_corridor = v; return TSTA_OK; }
    if (!strcmp(key, "test_abandon") && v >= 0) { e->opt_test_abandon = v; return TSTA_OK; }
    if (!strcmp(key, "poa_ref_block") && v >= 16 && v <= (1 << 20) && v % 16 == 0) { e->opt_poa_ref_block = v; return TSTA_OK; }
    if (!strcmp(key, "trace_tile") && v >= 0 && v % 64 == 0 && v <= 4096) { e->opt_trace_tile = v; return TSTA_OK; }
    return TSTA_ERR_INVALID_ARG;
}

int tsta_engine_stat(const tsta_engine* eng, const char* key, int64_t* value)
{
    if (!eng || !key || !value) return TSTA_ERR_INVALID_ARG;
    const Engine* e = reinterpret_cast<const Engine*>(eng);
    if (!strcmp(key, "single_launch_reruns")) { *value = e->single_launch_reruns; return TSTA_OK; }
    if (!strcmp(key, "last_trace_form")) { *value = e->last_trace_form; return TSTA_OK; }
    if (!strcmp(key, "last_walk_jumps")) { *value = e->last_walk_jumps; return TSTA_OK; }
    if (!strcmp(key, "last_sub_batches")) { *value = e->last_sub_batches; return TSTA_OK; }
    if (!strcmp(key, "poa_last_spread")) { *value = e->last_poa_spread; return TSTA_OK; }
    if (!strcmp(key, "poa_last_ref_w")) { *value = e->poa_blocked_w; return TSTA_OK; }
    if (!strcmp(key, "psa_redo_pairs")) {               // pairs of the last score call the packed kernels handed to the int32 kernel (synchronises)
        *value = 0;
        if (!e->last_redo || e->last_redo_n <= 0 || !e->pair_flags.p) return TSTA_OK;
        std::vector<uint8_t> h((size_t)e->last_redo_n);
        Engine* me = const_cast<Engine*>(e);
        if (me->check(hipSetDevice(e->device), "hipSetDevice") != TSTA_OK || me->check(hipDeviceSynchronize(), "hipDeviceSynchronize") != TSTA_OK ||
            me->check(hipMemcpy(h.data(), e->last_redo, h.size(), hipMemcpyDeviceToHost), "D2H redo flags") != TSTA_OK) return TSTA_ERR_HIP;
        for (uint8_t x : h) *value += x != 0;
        return TSTA_OK;
    }
    return TSTA_ERR_INVALID_ARG;
}

int tsta_psa_score_batch_dev_status(tsta_engine* eng, void* stream)
{
    if (!eng) return TSTA_ERR_INVALID_ARG;
    Engine* e = reinterpret_cast<Engine*>(eng);
    int rc = e->check(hipSetDevice(e->device), "hipSetDevice");
    if (rc != TSTA_OK) return rc;
    hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : e->stream;
    unsigned st = 0;
    if (e->psa_status && (rc = e->check(hipMemcpyAsync(&st, e->psa_status, sizeof(st), hipMemcpyDeviceToHost, s), "D2H band pipeline status"))) return rc;
    if ((rc = e->check(hipStreamSynchronize(s), "hipStreamSynchronize"))) return rc;
    if (st) { e->set_error("single-launch band pipeline abandoned a wait (status %u): the scores of that call are INT32_MIN", st); return TSTA_ERR_INTERNAL; }
    return TSTA_OK;
}

int tsta_last_kernel_ms(tsta_engine* eng, float* ms, int32_t* n_launches)
{
    if (!eng || !ms) return TSTA_ERR_INVALID_ARG;
    Engine* e = reinterpret_cast<Engine*>(eng);
    if (!e->timing_valid) return TSTA_ERR_INVALID_ARG;
    int rc = e->check(hipEventSynchronize(e->ev1), "hipEventSynchronize");
    if (rc != TSTA_OK) return rc;
    rc = e->check(hipEventElapsedTime(ms, e->ev0, e->ev1), "hipEventElapsedTime");
    if (n_launches) *n_launches = e->n_launches;
    return rc;
}

}  // extern "C"
