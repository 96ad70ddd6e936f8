// psa_api.hip — C-ABI entry points for the PSA kernels (score-only batch).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#include "../../include/tsta_engine.h"
#include "engine.h"
#include "kernels.h"

using tsta::Engine;

namespace tsta {

// Rows per lane (band height = 64*R).  Larger R amortises the per-step lane hand-off over more cells but
// leaves lanes idle when the row sequence is shorter than a band.
static int psa_rows_per_lane(int32_t max_len)
{
    static int forced = -1;
    if (forced < 0) {
        const char* s = getenv("TSTA_PSA_R");        // tuning knob for experiments: 4, 8 or 16
        int v = s ? atoi(s) : 0;
        forced = (v == 4 || v == 8 || v == 16) ? v : 0;
    }
    if (forced) return forced;
    return max_len >= 4096 ? 16 : (max_len >= 1024 ? 8 : 4);
}

}  // namespace tsta

extern "C" {

int tsta_psa_score_batch_dev(tsta_engine* eng, const tsta_params* p, int64_t n_pairs, const uint8_t* d_seqs,
                             const int64_t* d_off1, const int32_t* d_len1, const int64_t* d_off2,
                             const int32_t* d_len2, int32_t* d_out, int32_t max_len, void* stream)
{
    if (!eng || !p || n_pairs < 0 || max_len < 0) return TSTA_ERR_INVALID_ARG;
    if (n_pairs == 0) return TSTA_OK;
    if (!d_seqs || !d_off1 || !d_len1 || !d_off2 || !d_len2 || !d_out) return TSTA_ERR_INVALID_ARG;
    if (n_pairs > 0x7fffffffLL) return TSTA_ERR_UNSUPPORTED;
    Engine* e = reinterpret_cast<Engine*>(eng);
    int rc = e->check(hipSetDevice(e->device), "hipSetDevice");
    if (rc != TSTA_OK) return rc;
    hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : e->stream;

    const int R = tsta::psa_rows_per_lane(max_len);
    const int per_cu = tsta::psa_score_blocks_per_cu(R);
    int64_t blocks = std::min<int64_t>((n_pairs + 3) / 4, (int64_t)e->n_cu * per_cu);
    if (blocks < 1) blocks = 1;
    const int64_t row_stride = ((int64_t)max_len + 63) / 64 * 64 + 64;
    const size_t need = (size_t)blocks * 4 * (size_t)row_stride * sizeof(int2);
    if ((rc = e->reserve(e->rows, need, "hipMalloc(boundary rows)")) != TSTA_OK) return rc;

    tsta::PsaBatchArgs a;
    a.seqs = d_seqs; a.off1 = d_off1; a.len1 = d_len1; a.off2 = d_off2; a.len2 = d_len2; a.out = d_out;
    a.rows = reinterpret_cast<int2*>(e->rows.p); a.row_stride = row_stride;
    a.counter = reinterpret_cast<unsigned int*>(e->counter.p);
    a.n_pairs = n_pairs; a.M = p->M; a.X = p->X; a.E = p->E; a.O = p->O;

    if ((rc = e->check(hipMemsetAsync(a.counter, 0, sizeof(unsigned int), s), "hipMemsetAsync")) != TSTA_OK) return rc;
    e->timing_begin(s);
    tsta::launch_psa_score_batch(R, a, (int)blocks, s);
    e->n_launches = 1;
    e->timing_end(s);
    return e->check(hipGetLastError(), "psa_score_batch launch");
}

int tsta_psa_score_batch(tsta_engine* eng, const tsta_params* p, int64_t n_pairs, const uint8_t* seqs,
                         int64_t seqs_bytes, const int64_t* off1, const int32_t* len1,
                         const int64_t* off2, const int32_t* len2, int32_t* out)
{
    if (!eng || !p || n_pairs < 0 || seqs_bytes < 0) return TSTA_ERR_INVALID_ARG;
    if (n_pairs == 0) return TSTA_OK;
    if (!seqs || !off1 || !len1 || !off2 || !len2 || !out) return TSTA_ERR_INVALID_ARG;
    Engine* e = reinterpret_cast<Engine*>(eng);
    int32_t max_len = 0;
    for (int64_t k = 0; k < n_pairs; ++k) {
        if (len1[k] < 0 || len2[k] < 0 || off1[k] < 0 || off2[k] < 0 ||
            off1[k] + len1[k] > seqs_bytes || off2[k] + len2[k] > seqs_bytes) {
            e->set_error("pair %lld: sequence range outside seqs[0..%lld)", (long long)k, (long long)seqs_bytes);
            return TSTA_ERR_INVALID_ARG;
        }
        max_len = std::max(max_len, std::max(len1[k], len2[k]));
    }
    int rc = e->check(hipSetDevice(e->device), "hipSetDevice");
    if (rc != TSTA_OK) return rc;
    const size_t nb = (size_t)n_pairs;
    const size_t meta_bytes = nb * (2 * sizeof(int64_t) + 2 * sizeof(int32_t));
    if ((rc = e->reserve(e->in_seqs, (size_t)seqs_bytes + 16, "hipMalloc(seqs)")) != TSTA_OK) return rc;
    if ((rc = e->reserve(e->in_meta, meta_bytes, "hipMalloc(meta)")) != TSTA_OK) return rc;
    if ((rc = e->reserve(e->out_scores, nb * sizeof(int32_t), "hipMalloc(scores)")) != TSTA_OK) return rc;
    char* m = reinterpret_cast<char*>(e->in_meta.p);
    int64_t* d_off1 = reinterpret_cast<int64_t*>(m);
    int64_t* d_off2 = d_off1 + nb;
    int32_t* d_len1 = reinterpret_cast<int32_t*>(d_off2 + nb);
    int32_t* d_len2 = d_len1 + nb;
    hipStream_t s = e->stream;
    if ((rc = e->check(hipMemcpyAsync(e->in_seqs.p, seqs, (size_t)seqs_bytes, hipMemcpyHostToDevice, s), "H2D seqs")) ||
        (rc = e->check(hipMemcpyAsync(d_off1, off1, nb * 8, hipMemcpyHostToDevice, s), "H2D off1")) ||
        (rc = e->check(hipMemcpyAsync(d_off2, off2, nb * 8, hipMemcpyHostToDevice, s), "H2D off2")) ||
        (rc = e->check(hipMemcpyAsync(d_len1, len1, nb * 4, hipMemcpyHostToDevice, s), "H2D len1")) ||
        (rc = e->check(hipMemcpyAsync(d_len2, len2, nb * 4, hipMemcpyHostToDevice, s), "H2D len2")))
        return rc;
    rc = tsta_psa_score_batch_dev(eng, p, n_pairs, reinterpret_cast<const uint8_t*>(e->in_seqs.p), d_off1, d_len1,
                                  d_off2, d_len2, reinterpret_cast<int32_t*>(e->out_scores.p), max_len, s);
    if (rc != TSTA_OK) return rc;
    if ((rc = e->check(hipMemcpyAsync(out, e->out_scores.p, nb * 4, hipMemcpyDeviceToHost, s), "D2H scores"))) return rc;
    return e->check(hipStreamSynchronize(s), "psa_score_batch sync");
}

}  // extern "C"
