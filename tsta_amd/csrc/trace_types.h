// trace_types.h — device-side descriptors shared by the trace-fill kernels (psa_trace.hip: int32, one pair per wave;
// psa_trace_packed.hip: packed 16-bit, two pairs per wave) and the walker.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsta {

// one oriented pair (a = columns = longer sequence, b = rows) of a group, device-resident
struct TraceJob {
    const uint8_t* a; const uint8_t* b;
    int n, m;
    int2*     rows;                 // [W+1][row_stride]: rows[w] = top row of band w ({H, Ebar} per column)
    int64_t   row_stride;
    int*      state;                // [W][NS][64] per-lane DP state between tiles
    uint8_t*  trace;                // [W][band_bytes]
    int64_t   band_bytes;
    int       W, S_total;           // bands, total steps per band (multiple of 64)
    int       band0;                // index of this job's first band among the bands of the group
    uint8_t*  ops;                  // walker output (see WalkArgs)
    int*      out;                  // [8]: maxsorce, H(m,n), #ops, final i, final j
};

struct PsaTraceArgs {
    const TraceJob* jobs; int n_jobs;
    int g_lo, g_hi;                 // range of group-wide band indices this launch covers
    int T, d;                       // tile length (steps), launch index
    int M, X, E, O;
    // persistent form (one launch, flag hand-off between bands): ticket word, steps completed per band, error bits
    unsigned* ticket; int* progress; unsigned* status;
    int test_abandon;               // test hook (handoff.h): the holder of ticket test_abandon-1 never publishes; 0 = off
    // split form (psa_trace_split_kernel): solvers + emitters
    int* eprog;                     // [bands]: steps up to which a band's bottom row and checkpoints are published
    int* ckpt;                      // [bands][NC][NS][64]: lane state of a band at step c*C (c >= 1)
    int  C, NC, K;                  // emitter tile length (steps, multiple of 64), tiles per band, tile order u = w + c*K
    int  n_solver_wg;               // tickets below this are solver workgroups (four bands each)
    unsigned long long* dbg;        // diagnostics (TSTA_SPLIT_CLOCK=1): [band] {s_memrealtime at the band's first step, at its last}, or null
    unsigned* tile_ticket; const int* diag_first; int n_diag, n_tiles;   // emitter tiles: diagonal u = tickets diag_first[u] .. diag_first[u+1]
};

template <int R> struct TraceCfg {
    static constexpr int SPB = 16 / R;          // steps per 16-byte trace group
    static constexpr int NS = 2 * R + 5;        // state ints per lane
};

// two jobs of similar geometry sharing the waves of the packed trace fill: the couple sweeps max(n) columns and
// max(m) rows; both jobs' rows / trace buffers are laid out for THAT geometry (cells outside a job are garbage nobody reads)
struct TraceCouple {
    int  ja, jb;                    // indices into the job table
    int  band0;                     // first band among the bands of all couples of the group
    int  W, S_total;                // bands / steps per band of the couple
    int  perm;                      // 1: all four sequences are pure ACGT: byte-permute mismatch lookup (packed_step.h)
    int* state;                     // [W][packed_state_ints(R)][64] packed per-lane state between tiles
};

struct PsaTracePackedArgs {
    const TraceJob* jobs; const TraceCouple* couples; int n_couples;
    int g_lo, g_hi;
    int T, d;
    int M, X, E, O;
    unsigned pe, po, pm, pmx;       // |E|, |O|, M, M-X replicated in both 16-bit halves
    int elide;                      // 1: blocks that provably cannot raise the maximum drop the running max (packed_step.h)
    // single-launch form (handoff.h): ticket word, steps completed per band (indexed by band number), error bits
    unsigned* ticket; int* progress; unsigned* status;
    // ... its tiles in the order the launch pipeline runs them: "launch" d = tiles tile_first[d] .. tile_first[d+1] = bands tile_glo[d] ...
    const int* tile_first; const int* tile_glo; int n_d, n_tiles;
    int test_abandon;               // test hook (handoff.h): the holder of ticket test_abandon-1 never publishes; 0 = off
};

// ints of per-lane state a packed band keeps between tiles: hl/fb[R], hd, hbot, ebot, cch, mxA, mxB, baseA, baseB, in_star, cch2
constexpr int packed_state_ints(int R) { return 2 * R + 10; }

constexpr int kPackedTraceR = 4;    // rows per lane of the packed trace fill (trace layout = TraceCfg<4>)
void launch_psa_trace_packed_tile(const PsaTracePackedArgs& args, int blocks, hipStream_t stream);
void launch_psa_trace_packed_persistent(const PsaTracePackedArgs& args, int n_cu, hipStream_t stream);
void launch_psa_trace_packed_finish(const TraceJob* jobs, const TraceCouple* couples, int n_couples, hipStream_t stream);

}  // namespace tsta
