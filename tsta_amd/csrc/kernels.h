// kernels.h — host<->kernel launch interface inside libtsta_hip.so (not part of the public C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsta {

struct PsaBatchArgs {
    const uint8_t* seqs;
    const int64_t* off1;
    const int32_t* len1;
    const int64_t* off2;
    const int32_t* len2;
    int32_t*       out;
    int2*          rows;        // per-wave boundary row workspace: [slots][row_stride] of {H, Ebar}
    int64_t        row_stride;  // >= longest column sequence in the batch
    unsigned int*  counter;     // ticket counter, zeroed before every launch
    int64_t        n_pairs;
    int32_t        M, X, E, O;
    unsigned long long* clock_out;   // diagnostics only (TSTA_PSA_CLOCK=1): {memtime, memrealtime} x {begin, end} of wave 0
    int64_t        packed_couples;  // couples of positions (2c, 2c+1), c < packed_couples, are tried by the packed kernel first
    const uint32_t* order;          // position -> pair index (pairs ordered by geometry so neighbours are similar); NULL = identity
    uint8_t*       redo;            // per pair (zeroed before the launches, may be null): 1 = the packed kernel gave the pair up (its scores
                                    // left the 16-bit window, psa_score_packed.hip: F3) and the int32 kernel scores it after all
};

__host__ __device__ inline int64_t psa_pair_at(const uint32_t* order, int64_t pos) { return order ? (int64_t)order[pos] : pos; }

// The couple at positions (2c, 2c+1) of the geometry order goes to the packed-int16 kernel iff both are non-empty and padding the
// smaller one to the larger one's geometry (columns n = longer sequence, rows m = shorter) still beats scoring them
// separately with the int32 kernel: the packed kernel needs ~0.63x the instructions per (padded) cell pair, so the
// padded area must stay below ~0.8x the two areas together.  Evaluated identically by both kernels.
__host__ __device__ inline bool psa_couple_is_packed(const int32_t* len1, const int32_t* len2, const uint32_t* order, int64_t c)
{
    const int64_t pa = psa_pair_at(order, 2 * c), pb = psa_pair_at(order, 2 * c + 1);
    const int32_t a1 = len1[pa], a2 = len2[pa], b1 = len1[pb], b2 = len2[pb];
    const int64_t na = a1 > a2 ? a1 : a2, ma = a1 > a2 ? a2 : a1;
    const int64_t nb = b1 > b2 ? b1 : b2, mb = b1 > b2 ? b2 : b1;
    if (ma < 1 || mb < 1) return false;
    const int64_t pn = na > nb ? na : nb, pm = ma > mb ? ma : mb;
    return 5 * pn * pm <= 4 * (na * ma + nb * mb);
}

// how far (in score units) any value in flight can stray from a pair's anchor in the packed 16-bit kernels with
// rows-per-lane R: rows 64R + columns <= 192 (drain block), U per step, +M per column of the shifted frame.  Must stay
// below 28000 (u16 window around the bias).
inline int64_t psa_packed_window(int64_t M, int64_t X, int64_t E, int64_t O, int R)
{
    const int64_t mx = M > X ? M : X;
    const int64_t U = (mx > 0 ? mx : 0) - O - E;
    return (64LL * R + 256) * U - 2LL * O - E + 200LL * M;
}

// may pure-ACGT couples run their elided blocks in the free-free frame (packed_step.h: pstep_elided<FF>)?  Needs
// X >= -2|E| (the mismatch addend 2|E|+X is a non-negative byte), 2|E|+M <= 255 and room in the 16-bit window.
inline bool psa_ffree_ok(int64_t M, int64_t X, int64_t E, int64_t O, int R)
{
    const int64_t ae = -E;
    if (2 * ae + X < 0 || 2 * ae + M > 255) return false;
    const int64_t col = ae + M, lc = ae * R - col;
    return psa_packed_window(M, X, E, O, R) + 32 * (lc < 0 ? -lc : lc) + ae * R + 200 * col < 28000;
}

struct PsaPackedArgs {
    PsaBatchArgs b;                 // same batch description; b.counter -> this kernel's own ticket word
    uint32_t pe, po, pm, pmx;       // |E|, |O|, M, M-X replicated in both 16-bit halves
    const uint8_t* acgt;            // per pair: 1 iff both sequences consist of the bytes A, C, G, T only (may be null)
    int32_t  elide;                 // 1: blocks that provably cannot raise the maximum run without the running max
    int32_t  ffree;                 // 1: pure-ACGT couples run those blocks in the frame with both gap extensions free (X >= -2|E|)
    int32_t  f3;                    // 0: bias 0x8000, u16 maxima (window guaranteed by the host's bound) | 1: bias 0x4000, steady blocks whose
                                    // operands provably stay positive normal binary16 patterns take max3 as one v_pk_maximum3_f16 |
                                    // 2, 3 (tests): as 1 but never the fast step / every couple given up to the int32 kernel
    unsigned long long* wave_log;   // diagnostics only (TSTA_PSA_CLOCK=2): per couple {memrealtime begin, end, HW_ID | XCC_ID << 32, slot}
};

// K1b (psa_trace_packed.hip: psa_score_bands_kernel): band pipeline over the couples of a small batch
struct PsaBandsArgs {
    PsaBatchArgs b;                 // batch; b.rows = [pair in couple order][W_max+1][b.row_stride]; b.packed_couples = couples
    int*     state;                 // [couple][W_max][packed_state_ints(R)][64]
    const uint8_t* acgt;            // per pair: 1 iff pure ACGT (may be null), as in PsaPackedArgs
    int32_t  elide;                 // as in PsaPackedArgs
    int32_t  ffree;                 // as in PsaPackedArgs
    int32_t  f3;                    // as in PsaPackedArgs (b.redo must then be set)
    int64_t  rows_per_pair;         // (W_max+1) * row_stride
    int      W_max, T, d;           // bands per pair at max_len, tile length (steps), launch index
    uint32_t pe, po, pm, pmx;       // |E|, |O|, M, M-X replicated in both 16-bit halves
    // single-launch (dataflow) form: the same tiles as tickets, d-major; "launch" d = tiles tile_first[d] .. = bands
    // tile_wlo[d] .. tile_wlo[d] + tile_nw[d] - 1 of every couple; progress[couple * W_max + band] = steps published
    unsigned* ticket; int* progress; unsigned* status;
    const int* tile_first; const int* tile_wlo; const int* tile_nw; int n_d, n_tiles;
    int test_abandon;               // test hook (handoff.h): the wave that draws ticket test_abandon-1 never publishes its tile; 0 = off
};
void launch_psa_score_bands(int R, const PsaBandsArgs& args, int blocks, hipStream_t stream);
void launch_psa_score_bands_persistent(int R, const PsaBandsArgs& args, int n_cu, hipStream_t stream);
void launch_psa_score_bands_finish(int R, const PsaBandsArgs& args, hipStream_t stream);

void launch_psa_score_batch(int R, const PsaBatchArgs& args, int blocks, hipStream_t stream);
int  psa_score_blocks_per_cu(int R);
void launch_dpp_selftest(int* d_out, hipStream_t stream);

struct Engine;
}  // namespace tsta
struct tsta_params;
namespace tsta {
// psa_trace.hip: score-only fill of ONE oriented pair pipelined over many waves (host pointers; synchronous)
int psa_score_single_pair(Engine* e, const tsta_params* p, const uint8_t* a, int n, const uint8_t* b, int m, int32_t* maxscore);
void launch_psa_score_packed(int R, const PsaPackedArgs& args, int blocks, hipStream_t stream);
// psa_api.hip: tsta_psa_score_batch proper; with d_scores the scores (also) stay on the device (multi.hip gathers them)
int psa_score_batch_host(Engine* e, const tsta_params* p, int64_t n_pairs, const uint8_t* seqs, int64_t seqs_bytes,
                         const int64_t* off1, const int32_t* len1, const int64_t* off2, const int32_t* len2,
                         int32_t* out, int64_t dev_count, int32_t** d_scores);
int  psa_packed_blocks_per_cu(int R);
// engine.hip: is the reference's PSA output the plain DP for this scoring (tsta_engine.h: tsta_psa_reference_exact)?  The second
// form is what every PSA entry point ends a successful call with: sets e->warn, returns TSTA_OK or TSTA_WARN_REFERENCE_UNDEFINED.
bool psa_reference_exact(const tsta_params* p, char* why, size_t cap);
int  psa_reference_warning(Engine* e, const tsta_params* p);
// psa_api.hip: the bodies of tsta_psa_score_batch_dev[_2bit] (TSTA_OK or an error; the public forms add the warning)
int psa_score_batch_dev_impl(Engine* e, const tsta_params* p, int64_t n_pairs, const uint8_t* d_seqs, const int64_t* d_off1,
                             const int32_t* d_len1, const int64_t* d_off2, const int32_t* d_len2, int32_t* d_out, int32_t max_len, void* stream);

}  // namespace tsta
