// engine.h — the object behind the opaque `tsta_engine*` of include/tsta_engine.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace tsta {

struct DevBuf {
    void*  p = nullptr;
    size_t bytes = 0;
};

struct Engine {
    int         device = 0;
    int         n_cu = 0;
    hipStream_t stream = nullptr;
    hipEvent_t  ev0 = nullptr, ev1 = nullptr;
    hipStream_t timing_stream = nullptr;
    hipStream_t side[3] = {nullptr, nullptr, nullptr};   // extra streams: independent launch sequences of one call overlap
    hipEvent_t  ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    bool        timing_valid = false;
    bool        timing_hold = false;    // timing_begin keeps the start already recorded (a call made of several enqueued parts)
    hipEvent_t  ev_up[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // "sub-batch j is on the device" (overlapped upload, psa_api.hip)
    int         n_launches = 0;
    char        err[512] = {0};
    char        warn[512] = {0};   // text behind the last TSTA_WARN_* status

    // tuning knobs (tsta_engine_set_option); defaults may be preset by the TSTA_* environment variables
    int opt_psa_packed = -1, opt_psa_rows = 0, opt_trace_rows = 0, opt_trace_tile = 0, opt_poa_cols = 0, opt_psa_elide = 1, opt_psa_perm = 1, opt_trace_packed = 1, opt_trace_streams = 0, opt_psa_bands = -1, opt_trace_persistent = 1, opt_psa_ffree = 1, opt_poa_ref_block = 640, opt_align_ws_gb = 0, opt_poa_dataflow = 1, opt_trace_lds = 1, opt_test_abandon = 0, opt_trace_split = 1, opt_poa_n_blocked = 1, opt_psa_overlap = 1, opt_psa_f16max = 1, opt_walk_corridor = 512, opt_walk_table = 1;
    unsigned* psa_status = nullptr;   // device word the last tsta_psa_score_batch_dev's single-launch band pipeline (K1b) reports into, or null
    bool      no_single_launch = false;   // set around such a second run
    int       single_launch_reruns = 0;   // calls that were run again through the launch pipeline after an abandoned single launch
    int       poa_blocked_w = 0;      // > 0: the last POA fill ran in the reference's own arithmetic and blocking at that -W (poa_ref8.hip)
    bool      ref8_feasible = false;  // the last POA fill could have run through poa_ref8.hip (-W in range, LDS, memory)
    bool      force_ref8 = false;     // transient: run the next POA fill through poa_ref8.hip if it can be (second run after a spread beyond the trackers)
    unsigned* poa_status = nullptr;   // device word the last POA fill's waits report into (dataflow form), or null
    int       last_trace_form = 0;    // how the last trace fill ran: 0 launch per tile diagonal, 1 single launch (tiles through memory), 2 LDS-linked quads,
                                      // 3 / 4 solver bands + emitter tiles (generic / pure-ACGT table), 5 coupled pairs in dataflow form
    const uint8_t* last_redo = nullptr; int64_t last_redo_n = 0;   // redo flags of the last score call (psa_score_packed.hip: F3), for tsta_engine_stat
    int       last_poa_spread = 0;    // largest distance between two predecessor rows of one node the last plain POA fill saw
    int       last_sub_batches = 1;   // sub-batches the last host-pointer score call was cut into (overlapped upload)
    int align_calls = 0;   // tsta_psa_align_batch calls so far (its workspace cap grows with them)

    DevBuf counter;      // ticket counters / flags (zeroed per launch)
    DevBuf rows;         // PSA per-wave boundary rows
    DevBuf in_seqs;      // staging for host-pointer entry points
    DevBuf in_meta;
    DevBuf out_scores;
    DevBuf trace;        // packed trace matrix (tsta_psa_align) / POA trace arrays
    DevBuf scratch;
    DevBuf scratch2;
    DevBuf unpacked;     // ASCII copy of 2-bit packed input (tsta_psa_score_batch*_2bit)
    DevBuf unpacked2;    // K1b single-launch form: ticket / progress words and the tile table
    std::vector<int> host_tab;   // host copy of that table: must outlive the asynchronous upload ...
    hipEvent_t ev_tab = nullptr; // ... which this event marks the end of (waited for before the table is rewritten)
    DevBuf pair_flags;   // per-pair alphabet flags of the packed score kernel
    DevBuf ckpt;         // single-pair trace fill, split form: lane-state checkpoints the emitter tiles start from
    DevBuf walk_tab;     // single-pair traceback: band-crossing table of the walk + the segments it jumped (psa_trace.hip, K3t)
    int    last_walk_jumps = 0;   // bands the last single-pair walk crossed through the table

    void set_error(const char* fmt, ...);
    int  check(hipError_t e, const char* what);
    // small blocking copies ON THE ENGINE'S STREAM (a plain hipMemcpy would bring up the process's null stream: ~15 ms the first time)
    int  copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, const char* what);
    int  reserve(DevBuf& b, size_t bytes, const char* what, bool growing = false);
    void forget_status_in(const DevBuf& b);   // nulls psa_status / poa_status when they point into a buffer about to be freed
    void timing_begin(hipStream_t s);
    void timing_end(hipStream_t s);
};

}  // namespace tsta
