// multi.hip — several GPUs of one node behind the C-ABI (include/tsta_engine.h: tsta_multi_*, tsta_shard_bounds,
// tsta_psa_score_batch_multi, tsta_psa_align_batch_multi).
//
// The reference has ONE driver loop over the blocks of ONE pair (psa/psa.c:619-639) and no notion of a batch, a second
// process or a second device.  Here a batch of independent pairs is cut into contiguous shards balanced by DP cells,
// one host thread drives one engine per device, and nothing crosses devices on the data path.  The only exchange is the
// final gather of the int32 scores: either each device copies its shard into the caller's array, or (option "gather")
// one grouped RCCL all-gather over xGMI leaves all scores on every device.  librccl.so (570 MB) is dlopen'ed on first
// use so that single-GPU callers and the CLIs never pay for loading it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/tsta_engine.h"
#include "engine.h"
#include "kernels.h"

namespace tsta {

// the five RCCL entry points used, with the types of <rccl/rccl.h> spelled out (ncclInt32 = 2, ncclSuccess = 0)
struct Rccl {
    void* handle = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

struct Multi {
    std::vector<int>          devices;
    std::vector<tsta_engine*> eng;
    int   gather = 0;
    Rccl  rccl;
    std::vector<void*> comms;           // one ncclComm_t per device once "gather" >= 1 has been used
    std::vector<DevBuf> recv;           // per device: n_dev x max_shard int32
    char  err[640] = {0};

    void set_error(const char* fmt, ...)
    {
        va_list ap; va_start(ap, fmt); vsnprintf(err, sizeof(err), fmt, ap); va_end(ap);
    }
    int n() const { return (int)devices.size(); }
};

static int rccl_open(Multi* m)
{
    Rccl& r = m->rccl;
    if (r.handle) return TSTA_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) { r.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (r.handle) break; }
    if (!r.handle) { m->set_error("cannot load librccl.so: %s", dlerror()); return TSTA_ERR_UNSUPPORTED; }
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.handle, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.handle, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.handle, "ncclGroupEnd"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.AllGather || !r.GetErrorString) {
        m->set_error("librccl.so lacks an expected symbol");
        dlclose(r.handle); r = Rccl();
        return TSTA_ERR_UNSUPPORTED;
    }
    return TSTA_OK;
}

static int rccl_check(Multi* m, int res, const char* what)
{
    if (res == 0) return TSTA_OK;
    m->set_error("%s: %s", what, m->rccl.GetErrorString(res));
    return TSTA_ERR_HIP;
}

static int rccl_comms(Multi* m)
{
    if (!m->comms.empty()) return TSTA_OK;
    int rc = rccl_open(m);
    if (rc != TSTA_OK) return rc;
    m->comms.assign((size_t)m->n(), nullptr);
    // RCCL announces its version on stdout when the first communicator is made; a drop-in CLI's stdout is compared byte
    // for byte with the reference's, so that line goes to stderr instead
    fflush(stdout);
    const int saved = dup(STDOUT_FILENO);
    if (saved >= 0) (void)dup2(STDERR_FILENO, STDOUT_FILENO);
    const int res = m->rccl.CommInitAll(m->comms.data(), m->n(), m->devices.data());
    if (saved >= 0) { fflush(stdout); (void)dup2(saved, STDOUT_FILENO); close(saved); }
    if ((rc = rccl_check(m, res, "ncclCommInitAll")) != TSTA_OK) m->comms.clear();
    return rc;
}

// what one device thread needs of the batch: its pairs, the byte range of seqs they touch, offsets rebased to it
struct Shard {
    int64_t lo = 0, hi = 0, base = 0, bytes = 0;
    std::vector<int64_t> off1, off2;
    int rc = TSTA_OK;
};

static int make_shards(Multi* m, int64_t n_pairs, int64_t seqs_bytes, const int64_t* off1, const int32_t* len1,
                       const int64_t* off2, const int32_t* len2, std::vector<Shard>& sh)
{
    const int nd = m->n();
    std::vector<int64_t> bounds((size_t)nd + 1);
    int rc = tsta_shard_bounds(n_pairs, len1, len2, nd, bounds.data());
    if (rc != TSTA_OK) return rc;
    sh.assign((size_t)nd, Shard());
    for (int d = 0; d < nd; ++d) {
        Shard& s = sh[d];
        s.lo = bounds[d]; s.hi = bounds[d + 1];
        int64_t b0 = seqs_bytes, b1 = 0;
        for (int64_t k = s.lo; k < s.hi; ++k) {
            if (len1[k] < 0 || len2[k] < 0 || off1[k] < 0 || off2[k] < 0 || off1[k] + len1[k] > seqs_bytes || off2[k] + len2[k] > seqs_bytes) {
                m->set_error("pair %lld: sequence range outside seqs[0..%lld)", (long long)k, (long long)seqs_bytes);
                return TSTA_ERR_INVALID_ARG;
            }
            b0 = std::min(b0, std::min(off1[k], off2[k]));
            b1 = std::max(b1, std::max(off1[k] + len1[k], off2[k] + len2[k]));
        }
        if (b1 < b0) { b0 = 0; b1 = 0; }
        s.base = b0; s.bytes = b1 - b0;
        s.off1.resize((size_t)(s.hi - s.lo)); s.off2.resize((size_t)(s.hi - s.lo));
        for (int64_t k = s.lo; k < s.hi; ++k) { s.off1[k - s.lo] = off1[k] - b0; s.off2[k - s.lo] = off2[k] - b0; }
    }
    return TSTA_OK;
}

static int first_failure(Multi* m, const std::vector<Shard>& sh, const char* what)
{
    for (int d = 0; d < m->n(); ++d)
        if (TSTA_FAILED(sh[d].rc)) {          // a TSTA_WARN_* of a shard is a success (the callers add the scoring warning themselves)
            m->set_error("%s on device %d: %s (%s)", what, m->devices[d], tsta_strerror(sh[d].rc), tsta_last_error(m->eng[d]));
            return sh[d].rc;
        }
    return TSTA_OK;
}

}  // namespace tsta

using tsta::Multi;
using tsta::Shard;

extern "C" {

int tsta_shard_bounds(int64_t n_pairs, const int32_t* len1, const int32_t* len2, int n_shards, int64_t* bounds)
{
    if (n_pairs < 0 || n_shards < 1 || !bounds || (n_pairs && (!len1 || !len2))) return TSTA_ERR_INVALID_ARG;
    // weight of a pair = its DP cells + 1; shard s ends at the first k whose running weight reaches (s+1)/n of the total
    unsigned __int128 total = 0;
    for (int64_t k = 0; k < n_pairs; ++k) {
        if (len1[k] < 0 || len2[k] < 0) return TSTA_ERR_INVALID_ARG;
        total += (unsigned __int128)((int64_t)len1[k] * (int64_t)len2[k] + 1);
    }
    bounds[0] = 0;
    unsigned __int128 run = 0;
    int64_t k = 0;
    for (int s = 1; s <= n_shards; ++s) {
        const unsigned __int128 target = total * (unsigned)s;             // compare run * n_shards >= total * s
        while (k < n_pairs && run * (unsigned)n_shards < target) { run += (unsigned __int128)((int64_t)len1[k] * (int64_t)len2[k] + 1); ++k; }
        bounds[s] = k;
    }
    bounds[n_shards] = n_pairs;
    return TSTA_OK;
}

int tsta_multi_create(const int* devices, int n_dev, tsta_multi** out)
{
    if (!out || n_dev < 0) return TSTA_ERR_INVALID_ARG;
    *out = nullptr;
    const int have = tsta_device_count();
    if (have <= 0) return TSTA_ERR_NO_DEVICE;
    if (n_dev == 0) { if (devices) return TSTA_ERR_INVALID_ARG; n_dev = have; }
    if (n_dev > have) return TSTA_ERR_INVALID_ARG;
    Multi* m = new (std::nothrow) Multi();
    if (!m) return TSTA_ERR_OOM;
    for (int i = 0; i < n_dev; ++i) {
        const int dev = devices ? devices[i] : i;
        if (dev < 0 || dev >= have || std::find(m->devices.begin(), m->devices.end(), dev) != m->devices.end()) { delete m; return TSTA_ERR_INVALID_ARG; }
        m->devices.push_back(dev);
    }
    for (int i = 0; i < n_dev; ++i) {
        tsta_engine* e = nullptr;
        const int rc = tsta_engine_create(m->devices[i], &e);
        if (rc != TSTA_OK) { tsta_multi_destroy(reinterpret_cast<tsta_multi*>(m)); return rc; }
        m->eng.push_back(e);
    }
    m->recv.assign((size_t)n_dev, tsta::DevBuf());
    *out = reinterpret_cast<tsta_multi*>(m);
    return TSTA_OK;
}

void tsta_multi_destroy(tsta_multi* mm)
{
    if (!mm) return;
    Multi* m = reinterpret_cast<Multi*>(mm);
    for (size_t i = 0; i < m->comms.size(); ++i) if (m->comms[i]) (void)m->rccl.CommDestroy(m->comms[i]);
    for (size_t i = 0; i < m->recv.size(); ++i)
        if (m->recv[i].p) { (void)hipSetDevice(m->devices[i]); (void)hipFree(m->recv[i].p); }
    for (tsta_engine* e : m->eng) tsta_engine_destroy(e);
    // librccl stays mapped: unloading it while its proxy threads wind down is not worth the risk
    delete m;
}

int tsta_multi_device_count(const tsta_multi* m) { return m ? reinterpret_cast<const Multi*>(m)->n() : TSTA_ERR_INVALID_ARG; }

tsta_engine* tsta_multi_engine(tsta_multi* mm, int i)
{
    Multi* m = reinterpret_cast<Multi*>(mm);
    return (m && i >= 0 && i < m->n()) ? m->eng[i] : nullptr;
}

const char* tsta_multi_last_error(const tsta_multi* m) { return m ? reinterpret_cast<const Multi*>(m)->err : ""; }

int tsta_multi_set_option(tsta_multi* mm, const char* key, int64_t value)
{
    if (!mm || !key) return TSTA_ERR_INVALID_ARG;
    Multi* m = reinterpret_cast<Multi*>(mm);
    if (!strcmp(key, "gather") && value >= 0 && value <= 2) { m->gather = (int)value; return TSTA_OK; }
    return TSTA_ERR_INVALID_ARG;
}

int tsta_psa_score_batch_multi(tsta_multi* mm, const tsta_params* p, int64_t n_pairs, const uint8_t* seqs,
                               int64_t seqs_bytes, const int64_t* off1, const int32_t* len1,
                               const int64_t* off2, const int32_t* len2, int32_t* out)
{
    if (!mm || !p || n_pairs < 0 || seqs_bytes < 0) return TSTA_ERR_INVALID_ARG;
    if (n_pairs == 0) return TSTA_OK;
    if (!seqs || !off1 || !len1 || !off2 || !len2 || !out) return TSTA_ERR_INVALID_ARG;
    Multi* m = reinterpret_cast<Multi*>(mm);
    const int nd = m->n();
    std::vector<Shard> sh;
    int rc = tsta::make_shards(m, n_pairs, seqs_bytes, off1, len1, off2, len2, sh);
    if (rc != TSTA_OK) return rc;
    int64_t max_shard = 1;
    for (const Shard& s : sh) max_shard = std::max(max_shard, s.hi - s.lo);
    const bool use_rccl = m->gather >= 1;
    if (use_rccl && (rc = tsta::rccl_comms(m)) != TSTA_OK) return rc;

    std::vector<int32_t*> d_scores((size_t)nd, nullptr);
    auto work = [&](int d) {
        Shard& s = sh[d];
        tsta::Engine* e = reinterpret_cast<tsta::Engine*>(m->eng[d]);
        s.rc = tsta::psa_score_batch_host(e, p, s.hi - s.lo, seqs + s.base, s.bytes, s.off1.data(), len1 + s.lo, s.off2.data(),
                                          len2 + s.lo, use_rccl ? nullptr : out + s.lo, max_shard, use_rccl ? &d_scores[d] : nullptr);
    };
    std::vector<std::thread> th;
    for (int d = 1; d < nd; ++d) th.emplace_back(work, d);
    work(0);
    for (std::thread& t : th) t.join();
    if ((rc = tsta::first_failure(m, sh, "tsta_psa_score_batch")) != TSTA_OK) return rc;
    // the scoring-domain warning (tsta_psa_reference_exact) is a property of `p` alone: its text goes to engine 0
    const int warn = tsta::psa_reference_warning(reinterpret_cast<tsta::Engine*>(m->eng[0]), p);
    if (!use_rccl) return warn;

    // ---- the path's only collective: all-gather of the padded int32 shards (RCCL over xGMI), one grouped call
    for (int d = 0; d < nd; ++d) {
        tsta::Engine* e = reinterpret_cast<tsta::Engine*>(m->eng[d]);
        if ((rc = e->check(hipSetDevice(e->device), "hipSetDevice")) != TSTA_OK) { m->set_error("%s", e->err); return rc; }
        const size_t want = (size_t)nd * (size_t)max_shard * sizeof(int32_t);
        if (m->recv[d].bytes < want) {
            if (m->recv[d].p) (void)hipFree(m->recv[d].p);
            m->recv[d] = tsta::DevBuf();
            if ((rc = e->check(hipMalloc(&m->recv[d].p, want), "hipMalloc(gathered scores)")) != TSTA_OK) { m->set_error("%s", e->err); return rc; }
            m->recv[d].bytes = want;
        }
    }
    if ((rc = tsta::rccl_check(m, m->rccl.GroupStart(), "ncclGroupStart")) != TSTA_OK) return rc;
    for (int d = 0; d < nd; ++d) {
        tsta::Engine* e = reinterpret_cast<tsta::Engine*>(m->eng[d]);
        const int res = m->rccl.AllGather(d_scores[d], m->recv[d].p, (size_t)max_shard, /*ncclInt32*/ 2, m->comms[d], e->stream);
        if (res != 0) { (void)m->rccl.GroupEnd(); return tsta::rccl_check(m, res, "ncclAllGather"); }
    }
    if ((rc = tsta::rccl_check(m, m->rccl.GroupEnd(), "ncclGroupEnd")) != TSTA_OK) return rc;
    std::vector<int32_t> host((size_t)nd * (size_t)max_shard), other;
    for (int d = 0; d < (m->gather == 2 ? nd : 1); ++d) {
        tsta::Engine* e = reinterpret_cast<tsta::Engine*>(m->eng[d]);
        std::vector<int32_t>& dst = d == 0 ? host : other;
        dst.resize(host.size());
        if ((rc = e->check(hipSetDevice(e->device), "hipSetDevice")) != TSTA_OK ||
            (rc = e->check(hipMemcpyAsync(dst.data(), m->recv[d].p, dst.size() * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream), "D2H gathered scores")) != TSTA_OK ||
            (rc = e->check(hipStreamSynchronize(e->stream), "gather sync")) != TSTA_OK) { m->set_error("%s", e->err); return rc; }
        if (d > 0)
            for (int q = 0; q < nd; ++q)
                if (memcmp(&host[(size_t)q * max_shard], &other[(size_t)q * max_shard], (size_t)(sh[q].hi - sh[q].lo) * sizeof(int32_t)) != 0) {
                    m->set_error("gathered scores on device %d differ from device %d (shard %d)", m->devices[d], m->devices[0], q);
                    return TSTA_ERR_INTERNAL;
                }
    }
    // the other devices' streams still hold their part of the collective: drain them before the buffers can be reused
    for (int d = 1; d < nd; ++d) {
        tsta::Engine* e = reinterpret_cast<tsta::Engine*>(m->eng[d]);
        if ((rc = e->check(hipSetDevice(e->device), "hipSetDevice")) != TSTA_OK ||
            (rc = e->check(hipStreamSynchronize(e->stream), "gather sync")) != TSTA_OK) { m->set_error("%s", e->err); return rc; }
    }
    for (int q = 0; q < nd; ++q)
        memcpy(out + sh[q].lo, &host[(size_t)q * max_shard], (size_t)(sh[q].hi - sh[q].lo) * sizeof(int32_t));
    return warn;
}

int tsta_psa_align_batch_multi(tsta_multi* mm, const tsta_params* p, int64_t n_pairs, const uint8_t* seqs,
                               int64_t seqs_bytes, const int64_t* off1, const int32_t* len1,
                               const int64_t* off2, const int32_t* len2, tsta_alignment* out)
{
    if (!mm || !p || n_pairs < 0 || seqs_bytes < 0) return TSTA_ERR_INVALID_ARG;
    if (n_pairs == 0) return TSTA_OK;
    if (!off1 || !len1 || !off2 || !len2 || !out || (seqs_bytes && !seqs)) return TSTA_ERR_INVALID_ARG;
    Multi* m = reinterpret_cast<Multi*>(mm);
    const int nd = m->n();
    for (int64_t k = 0; k < n_pairs; ++k) { out[k].maxscore = -100; out[k].reserved = 0; out[k].len = 0; out[k].row_a = out[k].row_b = nullptr; }
    std::vector<Shard> sh;
    int rc = tsta::make_shards(m, n_pairs, seqs_bytes, off1, len1, off2, len2, sh);
    if (rc != TSTA_OK) return rc;
    static const uint8_t none = 0;
    auto work = [&](int d) {
        Shard& s = sh[d];
        s.rc = tsta_psa_align_batch(m->eng[d], p, s.hi - s.lo, seqs ? seqs + s.base : &none, s.bytes, s.off1.data(), len1 + s.lo,
                                    s.off2.data(), len2 + s.lo, out + s.lo);
    };
    std::vector<std::thread> th;
    for (int d = 1; d < nd; ++d) th.emplace_back(work, d);
    work(0);
    for (std::thread& t : th) t.join();
    if ((rc = tsta::first_failure(m, sh, "tsta_psa_align_batch")) != TSTA_OK) { tsta_alignments_free(out, n_pairs); return rc; }
    return tsta::psa_reference_warning(reinterpret_cast<tsta::Engine*>(m->eng[0]), p);
}

}  // extern "C"
