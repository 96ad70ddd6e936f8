// psa_trace.hip — K2 (fill with packed trace) + K3 (traceback walker): tsta_psa_align (one pair) and
// tsta_psa_align_batch (many independent pairs in flight at once; each pair is a "job", the bands of all jobs of a
// group share the launches).
//
// Replaces the TRACE build's fill (row() with the back/fback/eback stores, psa/psa.c:294-327,358-362, driven
// by the block wavefront :593-639) and trace() (psa.c:435-518).
//
// K2.  Same lane-skewed band sweep as K1 (psa_score.hip), but the bands of the single pair run CONCURRENTLY:
// band w trails band w-1 by T+64 steps, so a launch `d` lets every band advance by one T-step tile
// (steps [S_w(d-1), S_w(d)), S_w(d) = (d+1)T - w(T+64)); everything band w reads of band w-1's bottom row in
// launch d was written in launches < d.  The kernel boundary is the only inter-band synchronisation, so the
// schedule cannot deadlock (no flags, no spins).  Per-lane DP state lives in HBM between tiles.
//
// Trace record: 1 byte per cell, written in the wave's own (skewed) order so that each 16/R steps of a wave
// are ONE contiguous 1 KiB store (16 B per lane):
//     trace[band][step / SPB][lane][step % SPB][row-in-lane],   SPB = 16 / R
//   bits 1:0  back   : bit0 = (H != D), bit1 = (H != F)   -> diag if !bit0, else left if !bit1, else up
//   bits 3:2  fge + 1, fge = sign((H+O) - F)  -> fge>=0 <=> Fopen(i,j+1) (psa.c:308), fge==0 <=> mask1 (psa.c:304)
//   bits 5:4  ege + 1, ege = sign((H+O) - E)  -> ege>=0 <=> Eopen(i+1,j) (psa.c:322), ege==0 <=> mask1 (psa.c:317)
// The reference's fback(i,j) in {1,2,-2} is (Fopen(i,j), Ftie(i,j)); Fopen(i,j) is stored one cell to the
// left (it is known when F(i,j-1) is formed), the walker looks there.  Same for eback one cell up.
//
// K3.  One wave walks the path: it stages a window of the trace (one band, NQ 1-KiB step groups) in LDS; runs of
// diagonal moves are taken 64 cells at a time (lane l probes cell (i-l, j-l), ballot + count-trailing-zeros);
// gap moves follow psa.c:441-463 incl. the forced-continuation overwrite; one op per alignment column is
// emitted; cells outside the window (band crossings) are read straight from HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/tsta_engine.h"
#include "engine.h"
#include "wave_ops.h"
#include "handoff.h"
#include "trace_types.h"
#include "kernels.h"

namespace tsta {

constexpr int kNegInfT = -(1 << 30);

// sign(x) in {-1, 0, 1}: one v_med3_i32 (the compiler expands min(max(x,-1),1) into compares and selects)
__device__ __forceinline__ int sign3(int x) { int r; asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(x)); return r; }

__device__ __forceinline__ int hcol0t(int i, int E, int O) { return i == 0 ? 0 : O + i * E; }

template <int R>
struct TLane {
    int hl[R], fb[R], bch[R];
    int hd, hbot, ebot, cch, mx;
    int th, te, ach, sh, se;
};

// one DP step of the lane's R rows; ADDS the R trace bytes, little-endian and each short of its +0x14 (see below),
// into `tb` (R<=4: one dword, R==8: two dwords).
// Trace byte = nd | nf<<1 | (fge+1)<<2 | (ege+1)<<4 with fge, ege in {-1,0,1} = (nd + 2nf + 4fge + 16ege) + 20: the
// bracket is built with three v_lshl_add_u32 (it may be negative), shifted into its byte with a fourth, and the +20
// of all four bytes of a dword is added once (0x14141414) by the caller — integer arithmetic is linear, borrows
// between bytes cancel.
// ACGT (split form, pure-ACGT pairs): the match / mismatch term is ONE v_bfe_i32 instead of compare + select — s.bch[k] holds the
// row's four substitution scores as bytes (score against A, C, T, G at byte (ASCII >> 1) & 3), the column character travels as
// 8 x that code, i.e. as the bit offset of its byte.  One instruction less per cell on the chain of dependent steps.
// LEAN (solvers of the split form, steady blocks): no running maximum — the emitter tiles, which compute every cell again off the
// chain, keep it — and no copy of the last row's H into hl[R-1] (nothing reads it inside a block: the caller sets it from hbot at
// the block's end).  Two instructions less on the chain of dependent steps.
template <int R, bool PRED, bool TRACE, bool FULL = false, bool ACGT = false, bool LEAN = false>   // FULL: every row of the band is a real row (no mask on the running max)
__device__ __forceinline__ void trace_step(TLane<R>& s, const int t, const int lane, const int n, const int nvalid,
                                           const int M, const int X, const int E, const int O, int (&tb)[(R + 3) / 4])
{
    unsigned hu_u, eu_u;
    lane_exchange_chain(reinterpret_cast<unsigned&>(s.cch), hu_u, eu_u, reinterpret_cast<unsigned&>(s.sh), reinterpret_cast<unsigned&>(s.se),
                  reinterpret_cast<unsigned&>(s.ach), reinterpret_cast<unsigned&>(s.th), reinterpret_cast<unsigned&>(s.te),
                  (unsigned)s.hbot, (unsigned)s.ebot);
    const int hu = (int)hu_u;
    int eu = (int)eu_u;

    const bool act = !PRED || ((unsigned)(t - lane) < (unsigned)n);
    int d = s.hd;
    int h = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int e = eu + E;
        const int f = s.fb[k] + E;
        const int dg = d + (ACGT ? __builtin_amdgcn_sbfe(s.bch[k], s.cch, 8) : (s.cch == s.bch[k] ? M : X));
        h = max3i(dg, e, f);
        const int g = h + O;
        if (TRACE) {
            const int nd = min(h - dg, 1);
            const int nf = min(h - f, 1);
            const int fge = sign3(g - f);       // sign(H+O-F)
            const int ege = sign3(g - e);       // sign(H+O-E)
            const int code = (ege << 4) + ((fge << 2) + ((nf << 1) + nd));
            tb[k / 4] = (code << (8 * (k % 4))) + tb[k / 4];
        }
        const int eb = max(e, g);
        const int fbn = max(f, g);
        d = s.hl[k];
        int hv = (FULL || k < nvalid) ? h : kNegInfT;
        if (PRED) {
            s.hl[k] = act ? h : s.hl[k];
            s.fb[k] = act ? fbn : s.fb[k];
            hv = act ? hv : kNegInfT;
        } else {
            if (!(LEAN && k == R - 1)) s.hl[k] = h;
            s.fb[k] = fbn;
        }
        if (!LEAN) s.mx = max(s.mx, hv);
        eu = eb;
    }
    s.hbot = h;
    s.ebot = eu;
    if (PRED) s.hd = act ? hu : s.hd; else s.hd = hu;
}

template <int R, bool PRED, bool TRACE, bool WT = false, bool FULL = false, bool ACGT = false>   // (WT unused: kept so call sites read the same)
__device__ __forceinline__ void trace_block(TLane<R>& s, const int t0, const int lane, const int n, const int nvalid,
                                            const int M, const int X, const int E, const int O, uint4* __restrict__ tq)
{
    constexpr int SPB = TraceCfg<R>::SPB;
    // tq -> trace group of step t0 for this lane (stride 64 uint4 per group)
#pragma unroll 1
    for (int g = 0; g < 64 / SPB; ++g) {
        int acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < SPB; ++q) {
            int tb[(R + 3) / 4];
#pragma unroll
            for (int x = 0; x < (R + 3) / 4; ++x) tb[x] = 0;
            trace_step<R, PRED, TRACE, FULL, ACGT>(s, t0 + g * SPB + q, lane, n, nvalid, M, X, E, O, tb);
            if constexpr (!TRACE) { (void)acc; }
            else if constexpr (R == 8) { acc[2 * q] = tb[0]; acc[2 * q + 1] = tb[1]; }
            else if constexpr (R == 4) { acc[q] = tb[0]; }
            else { acc[(q * R) / 4] = (tb[0] << (8 * ((q * R) % 4))) + acc[(q * R) / 4]; }
        }
        if (TRACE) {
            const uint4 v = make_uint4((unsigned)acc[0] + 0x14141414u, (unsigned)acc[1] + 0x14141414u,
                                       (unsigned)acc[2] + 0x14141414u, (unsigned)acc[3] + 0x14141414u);
            tq[(size_t)g * 64] = v;
        }
    }
}

// ---- the four bands of one workgroup handed to each other through LDS (psa_trace_quad_kernel below) -----------------------
// A band's bottom row reaches the band below as one {H, Ebar} entry per column in an LDS ring, G = 16 columns at a time and
// through the registers the memory hand-off uses: after every G steps the top G lanes of the capture registers (sh, se)
// hold the G columns lane 63 finished last (t-63-G .. t-64 after step t): one LDS store; the band below refills lanes
// 0..G-1 of its top-row registers (th, te) with the G columns its next G steps consume: one LDS load.  Nothing is added to
// a step.  Flow control at the same moments: the reader waits until the writer's `done` word covers its G columns (writer
// 64+G steps ahead), the writer until the reader is less than a ring behind.  The distance between two such bands is
// 80-95 steps instead of the 128 + ~25 (latency) of the hand-off through memory (measured on 100 kbp: G = 8: 28.4 ms, 16: 27.6, 32: 27.8).
constexpr int kRing = 4096;                               // entries per ring (power of two); + 64 dummy entries for the lanes that store nothing
                                                          // (3 rings = 98 KB of LDS: at most ONE such workgroup per compute unit, on purpose — two
                                                          // workgroups of bands on one CU would put two chain waves on every SIMD)
struct LdsLink {
    int2* ring_top;  int* done_up;                        // ring the band reads its top row from / steps the band above has finished (null: memory)
    int2* ring_bot;  int* done_down;                      // ring its bottom row goes to / steps the band below has finished (null: memory)
    int*  done_mine;
};

__device__ __forceinline__ void lds_wait_ge(int* word, int need, unsigned* status, int lane)
{
    int spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { if (lane == 0) atomicOr(status, 4u); break; }
        // once any wave of the launch has given up, nobody waits any more
        if ((spins & 4095) == 0 && __hip_atomic_load((const GlobalU32*)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler ordering only: the LDS performs a wave's operations in issue order
}

constexpr int kLinkG = 16;                                // columns per LDS hand-off (16 / R steps make one trace group: G is a multiple of it)

template <int R, bool PRED, bool FULL, bool LTOP, bool LBOT, bool TRACE = true, bool ACGT = false, bool LEAN = false>
__device__ __forceinline__ void trace_block_linked(TLane<R>& s, const int t0, const int lane, const int n, const int nvalid,
                                                   const int M, const int X, const int E, const int O, uint4* __restrict__ tq,
                                                   const LdsLink& L, const int S_total, unsigned* status,
                                                   const int* poll_word = nullptr, int* poll_val = nullptr)
{
    constexpr int SPB = TraceCfg<R>::SPB;
    constexpr int G = kLinkG > SPB ? kLinkG : SPB;
    static_assert(G % SPB == 0 && 64 % G == 0, "hand-off every G steps");
    __builtin_assume((t0 & 63) == 0);
    auto group = [&](const int g) {
        const int ts = t0 + g * SPB;
        // a band that takes its top row from memory: the progress word of the band above is looked at HALF A BLOCK before the
        // next block needs it (one sc1 load in flight behind the steps; the block's start then finds the answer in a register
        // instead of paying a round trip to L2 on the chain)
        if (!LTOP && poll_word != nullptr && g == 32 / SPB) *poll_val = load_sc1_int(poll_word);
        if ((ts & (G - 1)) == 0) {
            if (LTOP) {
                // the band above has finished step ts+G-1+64: columns ts .. ts+G-1 are in the ring; they go to lanes 0..G-1, from
                // where the per-step rotation of (th, te) brings one to lane 0 in each of the next G steps
                lds_wait_ge(L.done_up, min(ts + G + 64, S_total), status, lane);
                const int2 v = L.ring_top[(ts + (lane & (G - 1))) & (kRing - 1)];
                s.th = lane < G ? v.x : s.th; s.te = lane < G ? v.y : s.te;
            }
            if (LBOT) lds_wait_ge(L.done_down, ts + 32 - kRing, status, lane);   // the band below has read what is about to be overwritten
        }
        int acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < SPB; ++q) {
            int tb[(R + 3) / 4];
#pragma unroll
            for (int x = 0; x < (R + 3) / 4; ++x) tb[x] = 0;
            trace_step<R, PRED, TRACE, FULL, ACGT, LEAN>(s, ts + q, lane, n, nvalid, M, X, E, O, tb);
            if constexpr (!TRACE) { (void)acc; }
            else if constexpr (R == 8) { acc[2 * q] = tb[0]; acc[2 * q + 1] = tb[1]; }
            else if constexpr (R == 4) { acc[q] = tb[0]; }
            else { acc[(q * R) / 4] = (tb[0] << (8 * ((q * R) % 4))) + acc[(q * R) / 4]; }
        }
        if constexpr (LEAN) s.hl[R - 1] = s.hbot;              // (what the steps above did not keep; once per group, off the steps)
        if constexpr (TRACE)
            tq[(size_t)g * 64] = make_uint4((unsigned)acc[0] + 0x14141414u, (unsigned)acc[1] + 0x14141414u,
                                            (unsigned)acc[2] + 0x14141414u, (unsigned)acc[3] + 0x14141414u);
        if (((ts + SPB) & (G - 1)) == 0) {
            const int te = ts + SPB;                         // steps finished; lanes 64-G..63 of (sh, se) hold columns te-64-G .. te-65
            if (LBOT) {
                const int col = te - 64 - G + (lane - (64 - G));
                const int slot = (lane >= 64 - G && col >= 0) ? (col & (kRing - 1)) : kRing + lane;      // the others: dummy entries
                L.ring_bot[slot] = make_int2(s.sh, s.se);
            }
            // (the LDS performs one wave's operations in issue order: the entries are in before the word says so)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) __hip_atomic_store(L.done_mine, te, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    if constexpr (LEAN && G == 2 * SPB) {
        // the solvers' steady blocks: a hand-off period (two groups) per iteration (four: no faster) — the tests of "is this a hand-off step" fold away
        // and the loop's own instructions, which a lone wave pays like any other, are halved
#pragma unroll 1
        for (int g = 0; g < 64 / SPB; g += 2) { group(g); group(g + 1); }
    } else {
#pragma unroll 1
        for (int g = 0; g < 64 / SPB; ++g) group(g);
    }
}

// One pair, one launch, four consecutive bands per workgroup: band 4b+j runs in wave j.  Inside the workgroup the bands are
// linked through LDS rings (above); the first band of a workgroup takes its top row from memory and the last one leaves its
// bottom row there, exactly as psa_trace_tile_kernel<.., PERSIST> does (sc1 stores, progress word, sc1 polls and loads).
// A workgroup's four bands are a ticket drawn when it starts (HIP promises no dispatch order): the band a workgroup's first wave
// waits for belongs to a workgroup that started earlier, so the waits end whatever the residency.  (The host still launches this
// form only when there are no more workgroups than compute units: queued workgroups would only lengthen the chain.)
//
// SPLIT (psa_trace_split_kernel below): the bands are SOLVERS — the same sweep without the trace bytes, i.e. only what the
// next step and the next band need (~31 instead of ~56 VALU instructions per R = 2 step on the chain of dependent steps).
// Every band additionally leaves in memory what an EMITTER wave needs to compute the trace bytes of a C-step tile of the band
// on its own, off the chain: its bottom row (every band, not only a workgroup's last) and its lane state at every multiple of
// C steps (`ckpt`); `eprog[band]` says how far both have been published.  That publication is deferred by one 64-step block:
// the wave drains its stores (s_waitcnt vmcnt(0)) before the stores of the current block are issued, when those of the
// previous block have long landed, so the drain costs nothing on the chain.
// the four substitution scores of a row character as bytes, at byte (ASCII >> 1) & 3 the score against that character (ACGT variants)
__device__ __forceinline__ int acgt_score_table(int row_char, int M, int X)
{
    const unsigned xs = (unsigned)X & 0xffu, ms = (unsigned)M & 0xffu;
    unsigned t = xs * 0x01010101u;
    if (row_char < 0x100) { const unsigned sh = 8u * (((unsigned)row_char >> 1) & 3u); t = (t & ~(0xffu << sh)) | (ms << sh); }
    return (int)t;
}
__device__ __forceinline__ int acgt_shift(int col_char) { return (int)(8u * (((unsigned)col_char >> 1) & 3u)); }

template <int R, bool SPLIT, bool ACGT = false>
__device__ __forceinline__ void quad_band(const PsaTraceArgs& A, const int wg, const int wave, const int lane, int2 (*ring)[kRing + 64], int* done)
{
    constexpr int NS = TraceCfg<R>::NS;
    constexpr int SPB = TraceCfg<R>::SPB;
    constexpr bool TRACE = !SPLIT;
    const TraceJob& J = A.jobs[0];
    const int w = wg * 4 + wave, g = J.band0 + w;
    if (w >= J.W) return;
    const bool mute = A.test_abandon != 0 && wg == A.test_abandon - 1;   // test hook: never publishes
    const int n = J.n, m = J.m, M = A.M, X = A.X, E = A.E, O = A.O, S_total = J.S_total;
    const uint8_t* __restrict__ seq_a = J.a;
    const uint8_t* __restrict__ seq_b = J.b;
    const int i0 = w * 64 * R + lane * R;
    const int nvalid = m - i0;
    int* st = J.state + (size_t)w * NS * 64 + lane;
    TLane<R> s;
#pragma unroll
    for (int k = 0; k < R; ++k) s.bch[k] = (i0 + k < m) ? (int)seq_b[i0 + k] : 0x100;
    if (ACGT) {
#pragma unroll
        for (int k = 0; k < R; ++k) s.bch[k] = acgt_score_table(s.bch[k], M, X);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int hk = hcol0t(i0 + 1 + k, E, O);
        s.hl[k] = hk;
        s.fb[k] = hk + O;
    }
    s.hd = hcol0t(i0, E, O);
    s.hbot = 0; s.ebot = 0; s.cch = 0; s.mx = kNegInfT;
    s.sh = 0; s.se = 0; s.th = 0; s.te = 0;

    const bool top_lds = wave > 0, bot_lds = wave < 3 && w + 1 < J.W;
    LdsLink L;
    L.ring_top = top_lds ? ring[wave - 1] : nullptr;  L.done_up = top_lds ? &done[wave - 1] : nullptr;
    L.ring_bot = bot_lds ? ring[wave] : nullptr;      L.done_down = bot_lds ? &done[wave + 1] : nullptr;
    L.done_mine = &done[wave];
    const int2* __restrict__ top = J.rows + (size_t)w * J.row_stride;
    int2* __restrict__ bot = J.rows + (size_t)(w + 1) * J.row_stride;
    uint8_t* tband = J.trace + (size_t)w * J.band_bytes;
    int pending = 0;                                          // SPLIT: steps whose stores have been issued but not yet published
    int seen_above = 0;                                       // progress of the band above (another workgroup's) as last polled
    unsigned long long clk0 = 0;
    if (SPLIT && A.dbg) clk0 = __builtin_amdgcn_s_memrealtime();

    for (int t0 = 0; t0 < S_total; t0 += 64) {
        const int c = t0 + lane;
        int na = 0;
        if (!top_lds) {
            // (seen_above: what the look-ahead poll of the previous block returned; the wait is entered only when that is not enough)
            if (w > 0 && seen_above < min(t0 + 128, S_total)) wait_progress(&A.progress[g - 1], min(t0 + 128, S_total), A.status, lane);
            int2 nt = make_int2(0, 0);
            if (c < n) nt = (w > 0) ? load_sc1_int2(&top[c]) : top[c];
            s.th = nt.x; s.te = nt.y;
        }
        if (c < n) na = ACGT ? acgt_shift(seq_a[c]) : (int)seq_a[c];
        s.ach = na;
        uint4* tq = reinterpret_cast<uint4*>(tband) + ((size_t)(t0 / SPB) * 64 + lane);
        const bool steady = t0 >= 64 && t0 + 63 < n;
        const bool full = (w + 1) * 64 * R <= m;
        const int* poll = (!top_lds && w > 0) ? &A.progress[g - 1] : nullptr;
#define TSTA_LINKED(PRED_, FULL_)                                                                                                        \
        do {                                                                                                                             \
            constexpr bool LEAN_ = SPLIT && !(PRED_);          /* steady solver blocks: see trace_step */                                \
            if (top_lds && bot_lds)  trace_block_linked<R, PRED_, FULL_, true, true, TRACE, ACGT, LEAN_>(s, t0, lane, n, nvalid, M, X, E, O, tq, L, S_total, A.status);   \
            else if (top_lds)        trace_block_linked<R, PRED_, FULL_, true, false, TRACE, ACGT, LEAN_>(s, t0, lane, n, nvalid, M, X, E, O, tq, L, S_total, A.status);  \
            else if (bot_lds)        trace_block_linked<R, PRED_, FULL_, false, true, TRACE, ACGT, LEAN_>(s, t0, lane, n, nvalid, M, X, E, O, tq, L, S_total, A.status, poll, &seen_above);  \
            else                     trace_block_linked<R, PRED_, FULL_, false, false, TRACE, ACGT, LEAN_>(s, t0, lane, n, nvalid, M, X, E, O, tq, L, S_total, A.status, poll, &seen_above); \
        } while (0)
        if (steady && full) TSTA_LINKED(false, true);
        else if (steady)    TSTA_LINKED(false, false);
        else                TSTA_LINKED(true, false);
#undef TSTA_LINKED
        if (SPLIT && A.dbg && t0 == 0) clk0 = __builtin_amdgcn_s_memrealtime();
        const int cs = t0 - 64 + lane;
        if (SPLIT) {
            // what the previous block left in memory has had this block's time to land: publish it to the emitters
            if (pending > 0 && !mute) publish_progress(&A.eprog[g], pending, lane);
            if (cs >= 0 && cs < n) store_sc1_int2(&bot[cs], s.sh, s.se);
            const int te = t0 + 64;
            if (te % A.C == 0 && te < S_total) {                // lane state at step te: where an emitter tile of this band starts
                int* ck = A.ckpt + ((size_t)g * A.NC + te / A.C) * NS * 64 + lane;
#pragma unroll
                for (int k = 0; k < R; ++k) { store_sc1_int(&ck[(2 * k) * 64], s.hl[k]); store_sc1_int(&ck[(2 * k + 1) * 64], s.fb[k]); }
                store_sc1_int(&ck[(2 * R) * 64], s.hd); store_sc1_int(&ck[(2 * R + 1) * 64], s.hbot); store_sc1_int(&ck[(2 * R + 2) * 64], s.ebot);
                store_sc1_int(&ck[(2 * R + 3) * 64], s.cch);
            }
            pending = te;
            if (!bot_lds && !mute) publish_progress(&A.progress[g], te, lane);      // the next workgroup's first band waits for this one
        } else if (!bot_lds) {
            if (cs >= 0 && cs < n) store_sc1_int2(&bot[cs], s.sh, s.se);
            if (!mute) publish_progress(&A.progress[g], t0 + 64, lane);
        }
    }
    if (SPLIT && !mute) publish_progress(&A.eprog[g], S_total, lane);
    if (SPLIT && A.dbg && lane == 0) { A.dbg[2 * g] = clk0; A.dbg[2 * g + 1] = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll
    for (int k = 0; k < R; ++k) { st[(2 * k) * 64] = s.hl[k]; st[(2 * k + 1) * 64] = s.fb[k]; }
    st[(2 * R) * 64] = s.hd; st[(2 * R + 1) * 64] = s.hbot; st[(2 * R + 2) * 64] = s.ebot;
    st[(2 * R + 3) * 64] = s.cch;
    if (SPLIT) atomicMax(&st[(2 * R + 4) * 64], s.mx);        // (the maxima of the steady blocks come from the emitter tiles; the slot was set to -inf before the launch)
    else st[(2 * R + 4) * 64] = s.mx;
}

template <int R>
__global__ __launch_bounds__(256) void psa_trace_quad_kernel(PsaTraceArgs A)
{
    __shared__ int2 ring[3][kRing + 64];
    __shared__ int done[4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __shared__ int wg_s;
    if (threadIdx.x < 4) done[threadIdx.x] = 0;
    if (threadIdx.x == 0) wg_s = (int)atomicAdd(A.ticket, 1u);
    __syncthreads();
    quad_band<R, false>(A, __builtin_amdgcn_readfirstlane(wg_s), wave, lane, ring, done);
}

// emitter tiles of the split form are numbered along the diagonals u = w + c K (tile (band w, chunk c) becomes available ~when the
// tiles of diagonal u do): diag_first[u] = number of tiles on the diagonals before u, diag_first[n_diag] = all of them
__global__ __launch_bounds__(256) void psa_split_diag_kernel(int* diag_first, int n_diag, int W, int NC, int K, int* state, int NS, int mx_slot)
{
    // (and the bands' running-maximum slots start at -inf: solvers and emitter tiles fold their maxima in with atomicMax)
    for (int x = threadIdx.x; x < W * 64; x += 256) state[(size_t)(x / 64) * NS * 64 + mx_slot * 64 + x % 64] = kNegInfT;
    __shared__ int part[257];
    auto count = [&](int u) { const int c_lo = max(0, (u - (W - 1) + K - 1) / K), c_hi = min(NC - 1, u / K); return max(0, c_hi - c_lo + 1); };
    const int per = (n_diag + 255) / 256, u0 = (int)threadIdx.x * per, u1 = min(n_diag, u0 + per);
    int sum = 0;
    for (int u = u0; u < u1; ++u) sum += count(u);
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { int acc = 0; for (int t = 0; t < 256; ++t) { const int v = part[t]; part[t] = acc; acc += v; } part[256] = acc; }
    __syncthreads();
    int acc = part[threadIdx.x];
    for (int u = u0; u < u1; ++u) { diag_first[u] = acc; acc += count(u); }
    if (threadIdx.x == 0) diag_first[n_diag] = part[256];
}

// ---- split form: solvers on the chain, emitters beside it ----------------------------------------------------------------
// The chain of dependent steps of ONE pair is what its fill costs (n + bands x ~100 steps, one wave per band, a lone wave
// issuing one VALU instruction every ~5.2 cycles): the fewer instructions a step has, the shorter the fill.  Of the ~56
// instructions of an R = 2 step only ~31 feed the next step or the next band; the rest build and pack the trace byte.  Here
// the workgroups that start first (tickets below n_solver_wg) are the solvers of quad_band<R, true>; every other wave of the
// launch is an EMITTER: it draws a tile (band w, steps [cC, (c+1)C)) — tiles are numbered along the diagonals u = w + cK, the
// order in which the solvers make them available — waits until the band's checkpoint at step cC and the upper band's bottom
// row up to (c+1)C + 64 have been published (eprog), restores the lane state, and runs the same steps again WITH the trace
// bytes (the unchanged trace_block of the launch pipeline), storing nothing but the trace.  Emitter tiles are independent of
// each other: they fill the SIMDs the ramp of the bands leaves idle, the compute units the solvers do not occupy and the
// issue slots a lone solver wave cannot use (solvers run at the highest wave priority, emitters at the lowest).  Total work
// grows (every cell is computed twice), the chain shrinks; the trace bytes are the same bytes.
// Workgroups of EIGHT waves, one per compute unit (the rings' LDS sees to that): waves 0-3 of a solver workgroup are its four
// bands, one per SIMD; waves 4-7 — and all eight waves of the workgroups beyond the solvers — are emitters.
constexpr int kSplitThreads = 512;
template <int R, bool ACGT>
__global__ __launch_bounds__(kSplitThreads) void psa_trace_split_kernel(PsaTraceArgs A)
{
    constexpr int NS = TraceCfg<R>::NS;
    constexpr int SPB = TraceCfg<R>::SPB;
    __shared__ int2 ring[3][kRing + 64];
    __shared__ int done[4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __shared__ int wg_s;
    if (threadIdx.x < 4) done[threadIdx.x] = 0;
    if (threadIdx.x == 0) wg_s = (int)atomicAdd(A.ticket, 1u);
    __syncthreads();
    const int wg = __builtin_amdgcn_readfirstlane(wg_s);
    if (wg < A.n_solver_wg && wave < 4) {
        __builtin_amdgcn_s_setprio(3);
        quad_band<R, true, ACGT>(A, wg, wave, lane, ring, done);
        return;
    }
    __builtin_amdgcn_s_setprio(0);
    const TraceJob& J = A.jobs[0];
    const int n = J.n, M = A.M, X = A.X, E = A.E, O = A.O, S_total = J.S_total;
    const uint8_t* __restrict__ seq_a = J.a;
    const uint8_t* __restrict__ seq_b = J.b;
    for (;;) {
        unsigned ticket = 0;
        if (lane == 0) ticket = atomicAdd(A.tile_ticket, 1u);
        const int t = (int)__builtin_amdgcn_readfirstlane(ticket);
        if (t >= A.n_tiles) return;
        int lo = 0, hi = A.n_diag - 1;          // diagonal u = last one whose first tile is <= t
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (A.diag_first[mid] <= t) lo = mid; else hi = mid - 1; }
        const int u = lo;
        const int c_lo = max(0, (u - (J.W - 1) + A.K - 1) / A.K);
        const int c = c_lo + (t - A.diag_first[u]);
        const int w = u - c * A.K, g = J.band0 + w;
        const int s0 = c * A.C, s1 = min(S_total, s0 + A.C);
        if (s0 > 0) wait_progress(&A.eprog[g], s0, A.status, lane);                            // the band's checkpoint at s0
        if (w > 0)  wait_progress(&A.eprog[g - 1], min(s1 + 64, S_total), A.status, lane);     // the upper band's bottom row, 64 steps further
        const int i0 = w * 64 * R + lane * R;
        const bool fullband = (w + 1) * 64 * R <= J.m;
        TLane<R> s;
#pragma unroll
        for (int k = 0; k < R; ++k) s.bch[k] = (i0 + k < J.m) ? (int)seq_b[i0 + k] : 0x100;
        if (ACGT) {
#pragma unroll
            for (int k = 0; k < R; ++k) s.bch[k] = acgt_score_table(s.bch[k], M, X);
        }
        if (s0 == 0) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int hk = hcol0t(i0 + 1 + k, E, O);
                s.hl[k] = hk;
                s.fb[k] = hk + O;
            }
            s.hd = hcol0t(i0, E, O);
            s.hbot = 0; s.ebot = 0; s.cch = 0;
        } else {
            const int* ck = A.ckpt + ((size_t)g * A.NC + c) * NS * 64 + lane;
#pragma unroll
            for (int k = 0; k < R; ++k) { s.hl[k] = load_sc1_int(&ck[(2 * k) * 64]); s.fb[k] = load_sc1_int(&ck[(2 * k + 1) * 64]); }
            s.hd = load_sc1_int(&ck[(2 * R) * 64]); s.hbot = load_sc1_int(&ck[(2 * R + 1) * 64]); s.ebot = load_sc1_int(&ck[(2 * R + 2) * 64]);
            s.cch = load_sc1_int(&ck[(2 * R + 3) * 64]);
        }
        s.mx = kNegInfT; s.sh = 0; s.se = 0;
        const int2* __restrict__ top = J.rows + (size_t)w * J.row_stride;
        uint8_t* tband = J.trace + (size_t)w * J.band_bytes;
        for (int t0 = s0; t0 < s1; t0 += 64) {
            const int cc = t0 + lane;
            int2 nt = make_int2(0, 0); int na = 0;
            if (cc < n) {
                nt = (w > 0) ? load_sc1_int2(&top[cc]) : top[cc];
                na = ACGT ? acgt_shift(seq_a[cc]) : (int)seq_a[cc];
            }
            s.th = nt.x; s.te = nt.y; s.ach = na;
            uint4* tq = reinterpret_cast<uint4*>(tband) + ((size_t)(t0 / SPB) * 64 + lane);
            // (the running maximum of every cell is kept HERE, off the solvers' chain: rows beyond m masked in the last band)
            const bool steady = t0 >= 64 && t0 + 63 < n;
            if (steady && fullband) trace_block<R, false, true, false, true, ACGT>(s, t0, lane, n, R, M, X, E, O, tq);
            else if (steady)        trace_block<R, false, true, false, false, ACGT>(s, t0, lane, n, J.m - i0, M, X, E, O, tq);
            else if (fullband)      trace_block<R, true, true, false, true, ACGT>(s, t0, lane, n, R, M, X, E, O, tq);
            else                    trace_block<R, true, true, false, false, ACGT>(s, t0, lane, n, J.m - i0, M, X, E, O, tq);
        }
        atomicMax(&J.state[(size_t)w * NS * 64 + (2 * R + 4) * 64 + lane], s.mx);
    }
}

// PERSIST = false: one tile of the launch-per-diagonal pipeline (see file header).
// PERSIST = true: ONE launch for the whole fill.  A wave draws a ticket = its band (bands of a job are consecutive), sweeps
// all its steps, then draws the next ticket (the grid holds one wave per SIMD; there may be more bands than that).  Before
// each 64-step block it waits until the band above has published the bottom-row entries the block reads
// (progress[g-1] >= t0 + 128: the upper band's lane 63 trails its lane 0 by 63 columns and stores a block's captures at the
// block's end).  Tickets are drawn in band order by waves that are already running, so the band a wave waits for is held by
// a wave that started earlier and needs nothing from later ones: no deadlock, whatever the number of resident waves.
// Hand-off without fences (MI355X_MICROARCH.md, "Valid forms", first row of the sc1 table): the boundary-row entries are
// written with sc1 (write-through) stores, the storing wave drains them (s_waitcnt vmcnt(0)) and then publishes its progress
// word with an sc1 store; the reader polls that word with sc1 loads and, once it matches, reads the entries with sc1 loads
// (both bypass the reader's L1; no L2 write-back, no invalidate).  Every wave signals only for its own stores.  The trace
// bytes are plain stores: nothing reads them before the kernel ends.  A poll that does not end within ~2^22 sleeps sets
// status[0] and goes on (garbage, reported by the host) so that every wave always terminates.
template <int R, bool TRACE, bool PERSIST>
__global__ __launch_bounds__(256) void psa_trace_tile_kernel(PsaTraceArgs A)
{
    constexpr int NS = TraceCfg<R>::NS;
    constexpr int SPB = TraceCfg<R>::SPB;
    const int lane = threadIdx.x & 63;
    for (;;) {
    int g;
    if (PERSIST) {
        unsigned ticket = 0;
        if (lane == 0) ticket = atomicAdd(A.ticket, 1u);
        g = (int)__builtin_amdgcn_readfirstlane(ticket);
    } else {
        g = A.g_lo + blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    }
    if (g >= A.g_hi) return;
    const bool mute = PERSIST && A.test_abandon != 0 && g == A.test_abandon - 1;     // test hook: this band never publishes
    int lo = 0, hi = A.n_jobs - 1;              // job = last one whose first band is <= g
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (A.jobs[mid].band0 <= g) lo = mid; else hi = mid - 1; }
    const TraceJob& J = A.jobs[lo];
    const int w = g - J.band0;
    if (w >= J.W) { if (PERSIST) continue; return; }
    const int shift = w * (A.T + 64);
    int s0 = A.d * A.T - shift, s1 = (A.d + 1) * A.T - shift;
    s0 = max(0, min(s0, J.S_total));
    s1 = max(0, min(s1, J.S_total));
    if (PERSIST) { s0 = 0; s1 = J.S_total; }
    if (s0 >= s1) { if (PERSIST) continue; return; }

    const int n = J.n, m = J.m, M = A.M, X = A.X, E = A.E, O = A.O;
    const uint8_t* __restrict__ seq_a = J.a;
    const uint8_t* __restrict__ seq_b = J.b;
    const int i0 = w * 64 * R + lane * R;
    const int nvalid = m - i0;
    int* st = J.state + (size_t)w * NS * 64 + lane;
    TLane<R> s;
#pragma unroll
    for (int k = 0; k < R; ++k) s.bch[k] = (i0 + k < m) ? (int)seq_b[i0 + k] : 0x100;
    if (s0 == 0) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int hk = hcol0t(i0 + 1 + k, E, O);
            s.hl[k] = hk;
            s.fb[k] = hk + O;
        }
        s.hd = hcol0t(i0, E, O);
        s.hbot = 0; s.ebot = 0; s.cch = 0; s.mx = kNegInfT;
    } else {
#pragma unroll
        for (int k = 0; k < R; ++k) { s.hl[k] = st[(2 * k) * 64]; s.fb[k] = st[(2 * k + 1) * 64]; }
        s.hd = st[(2 * R) * 64]; s.hbot = st[(2 * R + 1) * 64]; s.ebot = st[(2 * R + 2) * 64];
        s.cch = st[(2 * R + 3) * 64]; s.mx = st[(2 * R + 4) * 64];
    }
    s.sh = 0; s.se = 0;

    const int2* __restrict__ top = J.rows + (size_t)w * J.row_stride;
    int2* __restrict__ bot = J.rows + (size_t)(w + 1) * J.row_stride;
    uint8_t* tband = J.trace + (size_t)w * J.band_bytes;

    for (int t0 = s0; t0 < s1; t0 += 64) {
        if (PERSIST && w > 0) wait_progress(&A.progress[g - 1], min(t0 + 128, J.S_total), A.status, lane);
        const int c = t0 + lane;
        int2 nt = make_int2(0, 0); int na = 0;
        if (c < n) {
            nt = (PERSIST && w > 0) ? load_sc1_int2(&top[c]) : top[c];      // handed off inside this launch: never a plain load
            na = seq_a[c];
        }
        s.th = nt.x; s.te = nt.y; s.ach = na;
        uint4* tq = reinterpret_cast<uint4*>(tband) + ((size_t)(t0 / SPB) * 64 + lane);
        const bool steady = t0 >= 64 && t0 + 63 < n;
        if (steady && (w + 1) * 64 * R <= m) trace_block<R, false, TRACE, false, true>(s, t0, lane, n, nvalid, M, X, E, O, tq);
        else if (steady)                     trace_block<R, false, TRACE, false>(s, t0, lane, n, nvalid, M, X, E, O, tq);
        else                                 trace_block<R, true, TRACE, false>(s, t0, lane, n, nvalid, M, X, E, O, tq);
        const int cs = t0 - 64 + lane;
        if (PERSIST) {
            if (cs >= 0 && cs < n) store_sc1_int2(&bot[cs], s.sh, s.se);
            if (!mute) publish_progress(&A.progress[g], t0 + 64, lane);
        } else {
            if (cs >= 0 && cs < n) bot[cs] = make_int2(s.sh, s.se);
        }
    }
#pragma unroll
    for (int k = 0; k < R; ++k) { st[(2 * k) * 64] = s.hl[k]; st[(2 * k + 1) * 64] = s.fb[k]; }
    st[(2 * R) * 64] = s.hd; st[(2 * R + 1) * 64] = s.hbot; st[(2 * R + 2) * 64] = s.ebot;
    st[(2 * R + 3) * 64] = s.cch; st[(2 * R + 4) * 64] = s.mx;
    if (!PERSIST) return;
    }
}

// boundary row of the matrix (psa.c:221-226) into rows[0] of every job (blockIdx.y = job)
__global__ void psa_trace_init_kernel(const TraceJob* jobs, int E, int O)
{
    const TraceJob& J = jobs[blockIdx.y];
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < J.n; c += gridDim.x * blockDim.x) {
        const int h = O + (c + 1) * E;
        J.rows[c] = make_int2(h, h + O);
    }
}

// per job (blockIdx.x): out[0] = max over all lanes' running maxima (floored at -100, psa.c:590), out[1] = H(m,n)
template <int R>
__global__ void psa_trace_finish_kernel(const TraceJob* jobs)
{
    constexpr int NS = TraceCfg<R>::NS;
    const TraceJob& J = jobs[blockIdx.x];
    const int* state = J.state;
    const int W = J.W, m = J.m;
    int mx = kNegInfT;
    // (1 024 threads, four independent loads in flight per thread: a long pair has ~50 000 lane maxima, and a loop of dependent
    // round trips over them was 77 us of a 100 kbp alignment)
    const int total = W * 64, stride = (int)blockDim.x;
    int idx = threadIdx.x;
    for (; idx + 3 * stride < total; idx += 4 * stride) {
        int v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int x = idx + q * stride; v[q] = state[(size_t)(x / 64) * NS * 64 + (2 * R + 4) * 64 + x % 64]; }
        mx = max(max(mx, max(v[0], v[1])), max(v[2], v[3]));
    }
    for (; idx < total; idx += stride) mx = max(mx, state[(size_t)(idx / 64) * NS * 64 + (2 * R + 4) * 64 + idx % 64]);
    __shared__ int red[1024];
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = (int)blockDim.x / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) {
        J.out[0] = max(red[0], -100);
        const int rr = (m - 1) % (64 * R), w = (m - 1) / (64 * R);
        J.out[1] = state[(size_t)w * NS * 64 + (2 * (rr % R)) * 64 + rr / R];
    }
}

// ------------------------------------------------------------------------------------------ K3 walker
struct WalkArgs {
    const uint8_t* trace; int64_t band_bytes;
    int n, m;
    uint8_t* ops;        // out: one op per alignment column, emitted from the END of the alignment backwards:
                         //      1 = diagonal, 0 = left (gap in b), 2 = up (gap in a)
    int* out;            // out[2] = number of ops emitted, out[3] = final i, out[4] = final j
};

// (global address space spelled out: the pointers come out of a struct in memory, the compiler would emit flat loads, and a flat
// load in flight makes every LDS wait of the walk wait for it too)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const u32x4_t GlobalConstU4;
__device__ __forceinline__ uint4 ld_global_u4(GlobalConstU4* p) { const u32x4_t v = *p; return make_uint4(v.x, v.y, v.z, v.w); }
typedef __attribute__((address_space(1))) const uint8_t GlobalConstU8;
typedef __attribute__((address_space(1))) uint8_t GlobalU8;

// ---- K3t (round 4): the walk of ONE long pair crosses its bands through a table ----------------------------------------------
// The walk is one wave following a chain of dependent steps (~30 ns per step, ~1.2 us per band crossed: 3.1 ms on a 100 kbp pair).
// But the part of it inside one band depends only on where, and in which state, it enters the band: from (i = b BAND, j, forced)
// to the first moment i = (b-1) BAND.  psa_walk_origin_kernel computes that for EVERY entry column of a corridor around the
// straight line between the matrix's corners (the walk is pinned at both), every band boundary and the free entry state
// (forced = -1; a band entered in the middle of a vertical gap, forced = 2, is rare — about one in a hundred — and walked step by
// step), one thread per entry, all of them at once: table entry = {columns moved, ops emitted, state at the exit}.  The walker then
// crosses a band with one table look-up (slices of 32 boundaries staged in LDS at a time, around the diagonal through the current
// cell), noting where it entered; psa_walk_emit_kernel walks every noted band again, one thread per band, and writes its ops at the
// offset the walker counted.  Exactness does not depend on the table: an entry that is missing (outside the corridor, a walk
// that ends inside the band or wanders for more than 512 steps) is simply a band the walker crosses step by step as before.
struct WalkTabArgs {
    unsigned* tab;       // [nb + 1][Wc]: boundary b = 1 .. nb (row i = b BAND), entry column lo(b) + idx, entered free (forced = -1); null: no table
    int Wc, nb;          // corridor width in columns (multiple of 256), boundaries
    int* lo;             // [nb + 1]: first column of boundary b's corridor (written by psa_walk_origin_kernel)
    int* seg;            // [2 (nb + 1)]: seg[2b] = entry column of a band crossed by look-up (0: not), seg[2b+1] = ops before it
};
// entry: bits 0-9 ops emitted, 10-19 columns moved, 20 forced == 2 at the exit, 31 valid

__device__ __forceinline__ int corridor_lo(int b, int band, int n, int m, int Wc)
{
    const int c = (int)((long long)n * (b * band) / m);      // the straight line from (m, n) to (0, 0) at row b BAND: the walk is pinned at both ends
    return max(1, min(c - Wc / 2, n - Wc + 1));
}

// trace byte of the 1-based cell (ci, cj), per lane, from memory (unsigned arithmetic: shifts, no sign fix-ups; the offset inside a
// band fits 32 bits; a global, not a flat, load)
template <int R>
__device__ __forceinline__ unsigned trace_byte_at(const uint8_t* __restrict__ trace, int64_t band_bytes, int ci, int cj)
{
    constexpr unsigned SPB = TraceCfg<R>::SPB, BAND = 64 * R;
    const unsigned u = (unsigned)(ci - 1);
    const unsigned cw = u / BAND, rr = u % BAND, cl = rr / R, ck = rr % R;
    const unsigned ct = (unsigned)(cj - 1) + cl;
    const unsigned off = ((ct / SPB) * 64u + cl) * 16u + (ct % SPB) * R + ck;
    typedef __attribute__((address_space(1))) const uint8_t GlobalByte;
    return *(GlobalByte*)(trace + (size_t)cw * (size_t)band_bytes + off);
}

// one step of the walk (psa.c:441-463, as psa_walk_kernel below takes it), per lane; `byte(ci, cj)` = trace byte of a cell; returns the op
template <class Fetch>
__device__ __forceinline__ int walk_one_step(Fetch&& byte, int& i, int& j, int& forced)
{
    int k;
    if (forced >= 0) { k = forced; forced = -1; }
    else { const unsigned c = byte(i, j); k = !(c & 1u) ? 1 : (!(c & 2u) ? 0 : 2); }
    if (k != 1) {
        // a gap step: continue it if it was an extension, or it was opened and the next cell in its direction is "opened & tied"
        // (left: F bits of (i, j-1), (i, j-2); up: E bits of (i-1, j), (i-2, j)) — one code path for both directions
        const int di = k == 2 ? 1 : 0, dj = k == 0 ? 1 : 0, sh = k == 2 ? 4 : 2;
        const int a = k == 2 ? i - 1 : j - 1;
        if (a >= 1) {
            const unsigned c1 = (byte(i - di, j - dj) >> sh) & 3u;
            bool open_n = true;
            if (a != 1) open_n = ((byte(i - 2 * di, j - 2 * dj) >> sh) & 3u) >= 1u;
            if (c1 < 1u || (open_n && c1 == 1u)) forced = k;
        }
    }
    i -= (k != 0) ? 1 : 0;                                   // (arithmetic, not branches: the compiler otherwise keeps {i, j} as an indexed array in scratch)
    j -= (k != 2) ? 1 : 0;
    return k;
}

// one thread per entry column of one boundary (the bytes come through the caches: walks of neighbouring columns share their lines;
// staging a workgroup's region in LDS first was measured at twice the time)
template <int R>
__global__ __launch_bounds__(256) void psa_walk_origin_kernel(const TraceJob* jobs, WalkTabArgs T)
{
    constexpr int BAND = 64 * R;
    const TraceJob& J = jobs[0];
    const int b = (int)blockIdx.y + 1;
    const int idx = (int)(blockIdx.x * 256 + threadIdx.x);
    if (idx >= T.Wc) return;
    const int n = J.n, m = J.m;
    const int lo = corridor_lo(b, BAND, n, m, T.Wc);
    if (idx == 0) T.lo[b] = lo;                              // (for the walker: one 64-bit division per boundary, here)
    const int j0 = lo + idx;
    unsigned out = 0u;
    if (j0 <= n) {
        int i = b * BAND, j = j0, forced = -1, cnt = 0;
        const int stop = i - BAND;
        bool ok = true;
        const uint8_t* __restrict__ trace = J.trace;
        const int64_t band_bytes = J.band_bytes;
        auto byte = [&](int ci, int cj) -> unsigned { return trace_byte_at<R>(trace, band_bytes, ci, cj); };
        while (i > stop) {
            if (j < 1 || cnt >= 512) { ok = false; break; }
            walk_one_step(byte, i, j, forced);
            ++cnt;
        }
        if (ok && j >= 1) out = (unsigned)cnt | ((unsigned)(j0 - j) << 10) | (forced == 2 ? 1u << 20 : 0u) | 0x80000000u;
    }
    // (a walk that enters a band in the middle of a vertical gap, forced = 2, is rare — about one band in a hundred — and not
    // tabulated: half the table's work for nothing; the walker crosses such a band step by step)
    T.tab[(size_t)b * T.Wc + idx] = out;
}

template <int R>
__global__ __launch_bounds__(64) void psa_walk_emit_kernel(const TraceJob* jobs, WalkTabArgs T)
{
    constexpr int BAND = 64 * R;
    const TraceJob& J = jobs[0];
    const int b = (int)(blockIdx.x * 64 + threadIdx.x) + 1;
    if (b > T.nb) return;
    const int sj = T.seg[2 * b];
    if (sj == 0) return;
    int i = b * BAND, j = sj, forced = -1, cnt = T.seg[2 * b + 1];
    const int stop = i - BAND;
    typedef __attribute__((address_space(1))) uint8_t GlobalByteW;
    GlobalByteW* ops = (GlobalByteW*)J.ops;
    const uint8_t* __restrict__ trace = J.trace;
    const int64_t band_bytes = J.band_bytes;
    auto byte = [&](int ci, int cj) -> unsigned { return trace_byte_at<R>(trace, band_bytes, ci, cj); };
    while (i > stop && j >= 1) ops[cnt++] = (uint8_t)walk_one_step(byte, i, j, forced);
}

// TAB: with the band table of K3t (a separate instantiation: the slices in flight cost registers the plain walker does without)
template <int R, int NQ, bool TAB>
__global__ __launch_bounds__(64) void psa_walk_kernel(const TraceJob* jobs, WalkTabArgs T)
{
    WalkArgs A;
    {
        const TraceJob& J = jobs[blockIdx.x];
        A.trace = J.trace; A.band_bytes = J.band_bytes; A.n = J.n; A.m = J.m; A.ops = J.ops; A.out = J.out;
    }
    constexpr int SPB = TraceCfg<R>::SPB;
    constexpr int BAND = 64 * R;
    __shared__ uint4 win[NQ * 64];
    const int lane = threadIdx.x;
    int i = A.m, j = A.n;            // 1-based current cell (wave-uniform)
    int forced = -1;
    int cnt = 0;
    // The window the walk will need NEXT is loaded while it crosses the current one (the trace is cold in HBM: ~2-3 us per
    // window otherwise, one window per band of a mostly diagonal walk): `pre` holds step groups [pre_qlo, pre_qhi] of band pre_w.
    uint4 pre[NQ];
    int pre_w = -1, pre_qlo = 0, pre_qhi = -1;
    const int groups_per_band = (int)(A.band_bytes / (64 * 16));
    int jumps = 0;
    while (i >= 1 && j >= 1) {
        if constexpr (TAB && NQ == 32) if (forced < 0 && i % BAND == 0 && i / BAND >= 1 && i / BAND <= T.nb) {
            // ---- at a band boundary, free: the slices of this and the next 31 boundaries around the diagonal through (i, j) — 256
            // entry columns, 1 KB each — go into the window's LDS, then up to 32 bands are crossed by look-up
            const int b0 = i / BAND;
            const int lo_l = T.lo[max(1, b0 - (lane & 31))];               // lane k: first corridor column of boundary b0 - k
            if ((unsigned)(j - __builtin_amdgcn_readfirstlane(lo_l)) < (unsigned)T.Wc) {
                unsigned* tw = reinterpret_cast<unsigned*>(win);
                int ws[32], los[32];
                uint4 tmp[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const int bb = max(1, b0 - k);
                    los[k] = __builtin_amdgcn_readlane(lo_l, k);
                    ws[k] = max(0, min(j - k * BAND - 128 - los[k], T.Wc - 256)) & ~3;
                    tmp[k] = ld_global_u4((GlobalConstU4*)(T.tab + (size_t)bb * T.Wc + ws[k]) + lane);
                }
#pragma unroll
                for (int k = 0; k < 32; ++k) win[k * 64 + lane] = tmp[k];
                __syncthreads();
                int jumped = 0;
                bool go = true;                               // (no break: the loop unrolls, ws[] stays in registers)
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const int bb = b0 - k;
                    const int idx = j - los[k] - ws[k];
                    go = go && forced < 0 && bb >= 1 && j >= 1 && (unsigned)idx < 256u;
                    unsigned en = 0u;
                    if (go) en = (unsigned)__builtin_amdgcn_readfirstlane((int)tw[k * 256 + idx]);
                    go = go && (en >> 31) != 0u;
                    if (go) {
                        if (lane == 0) { T.seg[2 * bb] = j; T.seg[2 * bb + 1] = cnt; }
                        cnt += (int)(en & 1023u); j -= (int)((en >> 10) & 1023u); forced = ((en >> 20) & 1u) ? 2 : -1; i -= BAND;
                        ++jumped;
                    }
                }
                __syncthreads();
                jumps += jumped;
                if (jumped) continue;
            }
        }
        // stage the window that ends at the current cell's step group (one band, NQ step groups of 1 KiB)
        const int w = (i - 1) / BAND;
        const int ln = ((i - 1) % BAND) / R;
        const int q_need = (j - 1 + ln) / SPB;
        int qlo, qhi;
        if (w == pre_w && q_need <= pre_qhi && q_need - pre_qlo >= 4) {
            qlo = pre_qlo; qhi = pre_qhi;                       // predicted well enough: no load on the path
#pragma unroll
            for (int x = 0; x < NQ; ++x) win[x * 64 + lane] = pre[x];
        } else {
            qhi = q_need; qlo = max(0, qhi - (NQ - 1));
            GlobalConstU4* src = (GlobalConstU4*)(reinterpret_cast<const uint4*>(A.trace + (size_t)w * A.band_bytes) + (size_t)qlo * 64);
            const int nq = qhi - qlo + 1;
            // all loads of the window in flight at once
            uint4 tmp[NQ];
#pragma unroll
            for (int x = 0; x < NQ; ++x) tmp[x] = (x < nq) ? ld_global_u4(src + x * 64 + lane) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int x = 0; x < NQ; ++x) win[x * 64 + lane] = tmp[x];
        }
        {
            // the next window, assuming the walk goes on diagonally: the band above if this window reaches the band's first row,
            // else the step groups just below this window (a walk leaves a window at step group qlo - 1 exactly)
            const int rows_left = i - w * BAND;                 // diagonal steps until row w*BAND (the last row of the band above)
            const int j_top = j - rows_left + 1;                // column of the band's first row on that diagonal
            const bool exits_band = j_top < 1 || (j_top - 1) / SPB >= qlo;
            if (exits_band && w >= 1) {
                pre_w = w - 1;
                pre_qhi = min(groups_per_band - 1, (max(j_top - 1, 1) - 1 + 63) / SPB + 4);      // (margin: gaps shift the crossing)
            } else if (!exits_band && qlo >= 1) {
                pre_w = w;
                pre_qhi = qlo - 1;
            } else pre_w = -1;
            if (pre_w >= 0) {
                pre_qlo = max(0, pre_qhi - (NQ - 1));
                GlobalConstU4* src = (GlobalConstU4*)(reinterpret_cast<const uint4*>(A.trace + (size_t)pre_w * A.band_bytes) + (size_t)pre_qlo * 64);
                const int nq = pre_qhi - pre_qlo + 1;
#pragma unroll
                for (int x = 0; x < NQ; ++x) pre[x] = (x < nq) ? ld_global_u4(src + x * 64 + lane) : make_uint4(0, 0, 0, 0);
            }
        }
        __syncthreads();
        const uint8_t* wb = reinterpret_cast<const uint8_t*>(win);
        // trace byte of cell (ci,cj), 1-based; *inwin tells whether it came from the staged window
        auto code = [&](int ci, int cj) -> unsigned {
            const int cw = (ci - 1) / BAND, rr = (ci - 1) % BAND, cl = rr / R, ck = rr % R;
            const int ct = cj - 1 + cl, cq = ct / SPB;
            const size_t off = ((size_t)cq * 64 + cl) * 16 + (ct % SPB) * R + ck;
            // (made uniform inside each branch: the LDS path then waits for the LDS only, not for the prefetch in flight)
            if (__builtin_expect(cw == w && cq >= qlo && cq <= qhi, 1)) return (unsigned)__builtin_amdgcn_readfirstlane((int)wb[off - (size_t)qlo * 64 * 16]);
            return (unsigned)__builtin_amdgcn_readfirstlane((int)*(GlobalConstU8*)(A.trace + (size_t)cw * A.band_bytes + off));
        };
        // byte of cell (ci,cj) inside the staged window, or -1
        auto win_off = [&](int ci, int cj) -> int {
            const int cw = (ci - 1) / BAND, rr = (ci - 1) % BAND, cl = rr / R, ck = rr % R;
            const int ct = cj - 1 + cl, cq = ct / SPB;
            return (cw == w && cq >= qlo && cq <= qhi) ? ((cq - qlo) * 64 + cl) * 16 + (ct % SPB) * R + ck : -1;
        };
        // two cells at once (the gap rules look at the next two cells in the gap's direction): both LDS reads go out together
        auto code2 = [&](int ai, int aj, int bi, int bj, unsigned& ca, unsigned& cb) {
            const int oa = win_off(ai, aj), ob = win_off(bi, bj);
            if (__builtin_expect(oa >= 0 && ob >= 0, 1)) {
                const int va = wb[oa], vb = wb[ob];
                ca = (unsigned)__builtin_amdgcn_readfirstlane(va); cb = (unsigned)__builtin_amdgcn_readfirstlane(vb);
            } else { ca = code(ai, aj); cb = code(bi, bj); }
        };
        for (;;) {
            if (__builtin_expect(i < 1 || j < 1, 0)) break;
            {
                const int cw = (i - 1) / BAND, cl = ((i - 1) % BAND) / R;
                const int cq = (j - 1 + cl) / SPB;
                if (__builtin_expect(cw != w || cq < qlo, 0)) break;                  // left the window: restage
            }
            unsigned c = 0;
            if (__builtin_expect(forced < 0, 1)) {
                // ---- diagonal run: lane l probes cell (i-l, j-l); the run of leading "diag" cells is taken at once
                const int ci = i - lane, cj = j - lane;
                bool isdiag = false, seen = false;
                int byte_l = 0;
                if (ci >= 1 && cj >= 1) {
                    const int cw = (ci - 1) / BAND, rr = (ci - 1) % BAND, cl = rr / R, ck = rr % R;
                    const int ct = cj - 1 + cl, cq = ct / SPB;
                    if (cw == w && cq >= qlo) {
                        const size_t off = ((size_t)(cq - qlo) * 64 + cl) * 16 + (ct % SPB) * R + ck;
                        byte_l = wb[off];
                        seen = true;
                        isdiag = !(byte_l & 1u);                  // psa.c:443: back == 1
                    }
                }
                const unsigned long long mask = __ballot(isdiag);
                const int run = (~mask == 0ull) ? 64 : __builtin_ctzll(~mask);
                if (__builtin_expect(run > 0, 1)) {
                    if (lane < run) *(GlobalU8*)(A.ops + cnt + lane) = 1;
                    i -= run; j -= run; cnt += run;
                    // the cell that ended the run was probed too (lane `run`): its step follows without another probe
                    if (run == 64 || i < 1 || j < 1 || !((__ballot(seen) >> run) & 1ull)) continue;
                    c = (unsigned)__builtin_amdgcn_readlane(byte_l, run);
                } else
                    c = (unsigned)__builtin_amdgcn_readfirstlane(byte_l); // lane 0 probed (i, j) itself (inside the window)
            }
            // ---- one non-diagonal (or forced) step, computed identically by every lane (a forced step does not look at its cell)
            int k;
            if (forced >= 0) { k = forced; forced = -1; }
            else k = !(c & 1u) ? 1 : (!(c & 2u) ? 0 : 2);         // psa.c:443-455: diag > left > up
            if (lane == 0) *(GlobalU8*)(A.ops + cnt) = (uint8_t)k;
            ++cnt;
            if (k == 1) { --i; --j; }
            else if (k == 0) {
                // psa.c:450-452: continue left if F(i,j) was an extension, or it was opened and the cell to the
                // left is "opened & tied"
                if (j - 1 >= 1) {
                    unsigned cl1, cl2 = ~0u;
                    if (j - 1 != 1) code2(i, j - 1, i, j - 2, cl1, cl2); else cl1 = code(i, j - 1);
                    const bool fopen_ij = ((cl1 >> 2) & 3u) >= 1u;                       // Fopen(i,j) stored at (i,j-1)
                    const bool ftie_l = ((cl1 >> 2) & 3u) == 1u;                         // Ftie(i,j-1)
                    const bool fopen_l = (j - 1 != 1) ? (((cl2 >> 2) & 3u) >= 1u) : true;   // Fopen(i,j-1)
                    if (!fopen_ij || (fopen_l && ftie_l)) forced = 0;
                }
                --j;
            } else {
                if (i - 1 >= 1) {                                                       // psa.c:457-459
                    unsigned cu1, cu2 = ~0u;
                    if (i - 1 != 1) code2(i - 1, j, i - 2, j, cu1, cu2); else cu1 = code(i - 1, j);
                    const bool eopen_ij = ((cu1 >> 4) & 3u) >= 1u;                       // Eopen(i,j) stored at (i-1,j)
                    const bool etie_u = ((cu1 >> 4) & 3u) == 1u;                         // Etie(i-1,j)
                    const bool eopen_u = (i - 1 != 1) ? (((cu2 >> 4) & 3u) >= 1u) : true;   // Eopen(i-1,j)
                    if (!eopen_ij || (eopen_u && etie_u)) forced = 2;
                }
                --i;
            }
        }
        __syncthreads();
    }
    if (lane == 0) { A.out[2] = cnt; A.out[3] = i; A.out[4] = j; A.out[5] = jumps; }
}

// One oriented pair on the host side of a group, and what comes back for it.
struct HostJob {
    const uint8_t* a; int n;        // columns (host pointer)
    const uint8_t* b; int m;        // rows
    int32_t maxscore = -100, end_score = 0;
    std::vector<uint8_t> ops;       // walker output, from the end of the alignment backwards
    int fi = 0, fj = 0;             // cell at which the walk stopped
};

// layout of one job's workspaces for a sweep of n_lay columns x m_lay rows (its own size, or its couple's)
template <int R> struct JobGeom {
    int W, S_total; int64_t row_stride, band_bytes;
    size_t rows_bytes, state_bytes, trace_bytes, ops_bytes, seq_bytes;
    JobGeom(int n, int m, int n_lay, int m_lay, bool trace) {
        W = (m_lay + 64 * R - 1) / (64 * R);
        S_total = (n_lay + 64 + 63) / 64 * 64;
        row_stride = S_total + 64;
        band_bytes = (int64_t)S_total * 64 * R;
        rows_bytes = (size_t)(W + 1) * row_stride * sizeof(int2);
        state_bytes = (size_t)W * TraceCfg<R>::NS * 64 * sizeof(int);
        trace_bytes = trace ? (size_t)W * band_bytes : 0;
        ops_bytes = trace ? ((size_t)n + m + 64 + 63) / 64 * 64 : 0;
        seq_bytes = ((size_t)n + m + 64 + 63) / 64 * 64;
    }
    JobGeom(int n, int m, bool trace) : JobGeom(n, m, n, m, trace) {}
    size_t total() const { return rows_bytes + state_bytes + trace_bytes + ops_bytes + seq_bytes + 256; }
};

static bool pure_acgt(const HostJob& J)
{
    for (int i = 0; i < J.n; ++i) { const uint8_t c = J.a[i]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return false; }
    for (int i = 0; i < J.m; ++i) { const uint8_t c = J.b[i]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return false; }
    return true;
}

// can the packed 16-bit trace fill (psa_trace_packed.hip) take this scoring?  (same domain as K1p at R = kPackedTraceR)
static bool packed_trace_fits(const tsta_params& p)
{
    if (p.E > 0 || p.O > 0 || p.M < 0 || p.M < p.X || p.M - p.X > 255) return false;
    if (p.E < -4096 || p.O < -4096 || p.M > 4096 || p.X < -4096) return false;
    return psa_packed_window(p.M, p.X, p.E, p.O, kPackedTraceR) < 28000;
}

// TRACE = true: fill with packed trace + on-device walk of every job of the group.  TRACE = false: the same pipelined
// fill, score only (psa_score_single_pair: a single long pair would otherwise occupy one wave of the batch kernel).
// All jobs of the group are in flight together: launch d advances every band of every job by one T-step tile.
// With R = kPackedTraceR and a scoring that fits 16-bit lanes, jobs of similar geometry are coupled and filled two per
// wave by K2p (psa_trace_packed.hip); the rest ("singles") by K2.
// single_launch = false: every fill through the launch-per-diagonal pipeline (what a call is run again with after a single
// launch abandoned a wait).
template <int R, bool TRACE>
static int run_group(Engine* e, const tsta_params* p, HostJob* jobs, int n_jobs, int T, bool single_launch = true)
{
    constexpr int NSP = packed_state_ints(kPackedTraceR);
    hipStream_t s = e->stream;
    // ---- couples: adjacent jobs in geometry order whose padded sweep stays below 1.25x their own cells
    std::vector<int> lay_n((size_t)n_jobs), lay_m((size_t)n_jobs), couple_of((size_t)n_jobs, -1);
    for (int j = 0; j < n_jobs; ++j) { lay_n[j] = jobs[j].n; lay_m[j] = jobs[j].m; }
    std::vector<TraceCouple> tc;
    if (TRACE && R == kPackedTraceR && n_jobs >= 2 && e->opt_trace_packed != 0 && packed_trace_fits(*p)) {
        std::vector<int> order((size_t)n_jobs);
        for (int j = 0; j < n_jobs; ++j) order[j] = j;
        std::sort(order.begin(), order.end(), [&](int x, int y) {
            if (jobs[x].n != jobs[y].n) return jobs[x].n > jobs[y].n;
            if (jobs[x].m != jobs[y].m) return jobs[x].m > jobs[y].m;
            return x < y;
        });
        for (int q = 0; q + 1 < n_jobs; ++q) {
            const int x = order[q], y = order[q + 1];
            const int64_t pn = std::max(jobs[x].n, jobs[y].n), pm = std::max(jobs[x].m, jobs[y].m);
            if (5 * pn * pm > 4 * ((int64_t)jobs[x].n * jobs[x].m + (int64_t)jobs[y].n * jobs[y].m)) continue;
            TraceCouple c; c.ja = x; c.jb = y; c.band0 = 0; c.W = 0; c.S_total = 0; c.state = nullptr;
            c.perm = (e->opt_psa_perm != 0 && pure_acgt(jobs[x]) && pure_acgt(jobs[y])) ? 1 : 0;
            couple_of[x] = couple_of[y] = (int)tc.size();
            lay_n[x] = lay_n[y] = (int)pn; lay_m[x] = lay_m[y] = (int)pm;
            tc.push_back(c);
            ++q;
        }
    }
    const int n_couples = (int)tc.size();

    std::vector<TraceJob> tj((size_t)n_jobs);
    size_t rows_b = 0, state_b = 0, trace_b = 0, ops_b = 0, seq_b = 0;
    int64_t single_bands = 0, couple_bands = 0, d_bound = 0;    // d_bound: tile diagonals ("launches") of the longest couple
    int n_singles = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const JobGeom<R> g(jobs[j].n, jobs[j].m, lay_n[j], lay_m[j], TRACE);
        if (couple_of[j] >= 0) d_bound = std::max<int64_t>(d_bound, ((int64_t)g.S_total + (int64_t)(g.W - 1) * (T + 64) + T - 1) / T);
        rows_b += g.rows_bytes; trace_b += g.trace_bytes; ops_b += g.ops_bytes; seq_b += g.seq_bytes;
        if (couple_of[j] < 0) { state_b += g.state_bytes; single_bands += g.W; ++n_singles; }
        else if (tc[couple_of[j]].ja == j) { state_b += (size_t)g.W * NSP * 64 * sizeof(int); couple_bands += g.W; }
    }
    if (single_bands > 0x7fffffff || couple_bands > 0x7fffffff) return TSTA_ERR_UNSUPPORTED;
    const size_t jobs_b = ((size_t)(n_jobs + n_singles) * sizeof(TraceJob) + 255) / 256 * 256;
    // split form (one pair: solvers + emitters, psa_trace_split_kernel): emitter tiles of kSplitC steps, numbered along u = w + c*K
    constexpr int kSplitC = 1024, kSplitK = 8;
    size_t split_tab_ints = 0;
    if (TRACE && n_jobs == 1) {
        const JobGeom<R> g0(jobs[0].n, jobs[0].m, true);
        split_tab_ints = (size_t)g0.W + (size_t)(g0.S_total / kSplitC + 2) * kSplitK + 8;
    }
    const size_t couples_b = ((size_t)n_couples * sizeof(TraceCouple) + 255) / 256 * 256;
    int rc;
    if ((rc = e->reserve(e->in_seqs, seq_b, "hipMalloc(seqs)")) ||
        (rc = e->reserve(e->rows, rows_b, "hipMalloc(rows)")) ||
        (rc = e->reserve(e->scratch, state_b + 256, "hipMalloc(state)")) ||
        (rc = e->reserve(e->in_meta, jobs_b + couples_b + ((size_t)n_jobs * 8 * sizeof(int) + 255) / 256 * 256 +
                                         (3 * (size_t)single_bands + (size_t)couple_bands + 2 * (size_t)d_bound + split_tab_ints + 256) * sizeof(int) + 256, "hipMalloc(jobs)")) ||
        (TRACE && (rc = e->reserve(e->trace, trace_b, "hipMalloc(trace)"))) ||
        (TRACE && (rc = e->reserve(e->scratch2, ops_b, "hipMalloc(ops)"))))
        return rc;
    char* d_seq = reinterpret_cast<char*>(e->in_seqs.p);
    char* d_rows = reinterpret_cast<char*>(e->rows.p);
    char* d_state = reinterpret_cast<char*>(e->scratch.p);
    char* d_trace = reinterpret_cast<char*>(e->trace.p);
    char* d_ops = reinterpret_cast<char*>(e->scratch2.p);
    TraceJob* d_jobs = reinterpret_cast<TraceJob*>(e->in_meta.p);            // all jobs, then the singles again
    TraceJob* d_singles = d_jobs + n_jobs;
    TraceCouple* d_couples = reinterpret_cast<TraceCouple*>(reinterpret_cast<char*>(e->in_meta.p) + jobs_b);
    int* d_out = reinterpret_cast<int*>(reinterpret_cast<char*>(e->in_meta.p) + jobs_b + couples_b);
    // single-launch fills: [0] ticket of K2, [1] status, [2] ticket of K2p, [64 ...] steps completed per band (K2's, then K2p's)
    int* d_sync = reinterpret_cast<int*>(reinterpret_cast<char*>(d_out) + ((size_t)n_jobs * 8 * sizeof(int) + 255) / 256 * 256);
    // one pair per wave: a wave owns a band from its first step to its last (psa_trace_tile_kernel<.., PERSIST>) — right for the
    // single-pair calls, whose bands are few and all on the critical path; many unpaired jobs keep the launch pipeline
    const bool persistent = n_singles == 1 && e->opt_trace_persistent != 0 && single_launch;
    const bool persistent_couples = n_couples > 0 && e->opt_trace_persistent != 0 && single_launch;
    int sband0 = 0, cband0 = 0, max_n = 0;
    std::vector<TraceJob> singles;
    std::vector<uint8_t> h_seq(seq_b);
    for (int j = 0; j < n_jobs; ++j) {
        const JobGeom<R> g(jobs[j].n, jobs[j].m, lay_n[j], lay_m[j], TRACE);
        TraceJob& J = tj[j];
        J.a = reinterpret_cast<const uint8_t*>(d_seq); J.b = J.a + jobs[j].n; J.n = jobs[j].n; J.m = jobs[j].m;
        {   // one staging buffer, one H2D for all jobs (thousands of short pairs would otherwise be thousands of copies)
            const size_t off = (size_t)(d_seq - reinterpret_cast<char*>(e->in_seqs.p));
            memcpy(h_seq.data() + off, jobs[j].a, (size_t)jobs[j].n);
            memcpy(h_seq.data() + off + jobs[j].n, jobs[j].b, (size_t)jobs[j].m);
        }
        J.rows = reinterpret_cast<int2*>(d_rows); J.row_stride = g.row_stride;
        J.trace = reinterpret_cast<uint8_t*>(d_trace); J.band_bytes = g.band_bytes;
        J.W = g.W; J.S_total = g.S_total;
        J.ops = reinterpret_cast<uint8_t*>(d_ops); J.out = d_out + 8 * j;
        J.state = nullptr; J.band0 = 0;
        if (couple_of[j] < 0) {
            J.state = reinterpret_cast<int*>(d_state); d_state += g.state_bytes;
            J.band0 = sband0; sband0 += g.W;
            singles.push_back(J);
        } else {
            TraceCouple& C = tc[couple_of[j]];
            if (C.ja == j) {                                    // the couple's geometry and state: set once
                C.W = g.W; C.S_total = g.S_total; C.band0 = cband0; cband0 += g.W;
                C.state = reinterpret_cast<int*>(d_state); d_state += (size_t)g.W * NSP * 64 * sizeof(int);
            }
        }
        d_seq += g.seq_bytes; d_rows += g.rows_bytes; d_trace += g.trace_bytes; d_ops += g.ops_bytes;
        max_n = std::max(max_n, jobs[j].n);
    }
    // couples were numbered in creation order but their bands in job order of their first member: renumber by band0
    std::sort(tc.begin(), tc.end(), [](const TraceCouple& x, const TraceCouple& y) { return x.band0 < y.band0; });
    if ((rc = e->check(hipMemcpyAsync(e->in_seqs.p, h_seq.data(), seq_b, hipMemcpyHostToDevice, s), "H2D sequences")) ||
        (rc = e->check(hipMemcpyAsync(d_jobs, tj.data(), tj.size() * sizeof(TraceJob), hipMemcpyHostToDevice, s), "H2D jobs")) ||
        (n_singles && (rc = e->check(hipMemcpyAsync(d_singles, singles.data(), singles.size() * sizeof(TraceJob), hipMemcpyHostToDevice, s), "H2D singles"))) ||
        (n_couples && (rc = e->check(hipMemcpyAsync(d_couples, tc.data(), tc.size() * sizeof(TraceCouple), hipMemcpyHostToDevice, s), "H2D couples"))))
        return rc;

    PsaTraceArgs A;
    A.jobs = d_singles; A.n_jobs = n_singles; A.T = T; A.M = p->M; A.X = p->X; A.E = p->E; A.O = p->O;
    A.ticket = reinterpret_cast<unsigned*>(d_sync); A.status = reinterpret_cast<unsigned*>(d_sync + 1); A.progress = d_sync + 64;
    A.test_abandon = e->opt_test_abandon;
    A.dbg = nullptr; A.eprog = nullptr; A.ckpt = nullptr; A.C = 0; A.NC = 0; A.K = 1; A.n_solver_wg = 0; A.tile_ticket = nullptr; A.diag_first = nullptr; A.n_diag = 0; A.n_tiles = 0;
    if ((persistent || persistent_couples) &&
        (rc = e->check(hipMemsetAsync(d_sync, 0, (2 * (size_t)single_bands + (size_t)couple_bands + 128) * sizeof(int), s), "hipMemsetAsync(sync)")))
        return rc;
    PsaTracePackedArgs B;
    B.jobs = d_jobs; B.couples = d_couples; B.n_couples = n_couples; B.T = T; B.M = p->M; B.X = p->X; B.E = p->E; B.O = p->O;
    {
        const uint32_t ae = (uint32_t)(-p->E), ao = (uint32_t)(-p->O), am = (uint32_t)p->M, amx = (uint32_t)(p->M - p->X);
        B.pe = ae | (ae << 16); B.po = ao | (ao << 16); B.pm = am | (am << 16); B.pmx = amx | (amx << 16);
        B.elide = e->opt_psa_elide != 0 &&
                  psa_packed_window(p->M, p->X, p->E, p->O, kPackedTraceR) + (int64_t)ae * (32 * (kPackedTraceR + 1) + 192) < 28000;
    }
    // The couples are independent of each other: split them into up to 4 chunks with their own band numbering and
    // their own stream, so that the launch sequences overlap (one sequence's launch gaps and half-empty tails are
    // filled by the others' waves).  Chunk 0 rides on the engine's stream.
    B.ticket = reinterpret_cast<unsigned*>(d_sync + 2); B.status = reinterpret_cast<unsigned*>(d_sync + 1); B.progress = d_sync + 64 + single_bands;
    B.tile_first = nullptr; B.tile_glo = nullptr; B.n_d = 0; B.n_tiles = 0; B.test_abandon = e->opt_test_abandon;
    int n_chunks = (n_couples >= 8 && e->opt_trace_streams != 1 && !persistent_couples) ? 4 : 1;
    for (int g = 0; g + 1 < n_chunks; ++g) {
        if (!e->side[g] && hipStreamCreateWithFlags(&e->side[g], hipStreamNonBlocking) != hipSuccess) { n_chunks = 1; break; }
        if (!e->ev_join[g] && hipEventCreateWithFlags(&e->ev_join[g], hipEventDisableTiming) != hipSuccess) { n_chunks = 1; break; }
    }
    if (n_chunks > 1 && !e->ev_fork && hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess) n_chunks = 1;
    int chunk_lo[5] = {0, n_couples, n_couples, n_couples, n_couples};
    if (n_chunks > 1) {
        int g = 1;
        for (int c = 0; c < n_couples && g < n_chunks; ++c)
            if ((int64_t)tc[c].band0 * n_chunks >= (int64_t)g * couple_bands) chunk_lo[g++] = c;
        for (; g < n_chunks; ++g) chunk_lo[g] = n_couples;
        chunk_lo[n_chunks] = n_couples;
        for (int q = 0; q < n_chunks; ++q) {                   // band numbers relative to the chunk
            if (chunk_lo[q] >= chunk_lo[q + 1]) continue;
            const int b0 = tc[chunk_lo[q]].band0;
            for (int c = chunk_lo[q]; c < chunk_lo[q + 1]; ++c) tc[c].band0 -= b0;
        }
        if ((rc = e->check(hipMemcpyAsync(d_couples, tc.data(), tc.size() * sizeof(TraceCouple), hipMemcpyHostToDevice, s), "H2D couples")))
            return rc;
    }

    e->timing_begin(s);
    hipLaunchKernelGGL(psa_trace_init_kernel, dim3((unsigned)std::min(64, (max_n + 255) / 256), (unsigned)n_jobs), dim3(256), 0, s,
                       d_jobs, p->E, p->O);
    int launches = 1;
    if (n_chunks > 1) {                                         // fork: the side streams start after the uploads + init
        if ((rc = e->check(hipEventRecord(e->ev_fork, s), "hipEventRecord(fork)"))) return rc;
        for (int q = 1; q < n_chunks; ++q)
            if ((rc = e->check(hipStreamWaitEvent(e->side[q - 1], e->ev_fork, 0), "hipStreamWaitEvent(fork)"))) return rc;
    }
    // band w of a job may run steps below S_w(d) = (d+1)T - w(T+64) after launch d
    int64_t D = 0;
    if (!persistent) for (const TraceJob& J : singles) D = std::max<int64_t>(D, ((int64_t)J.S_total + (int64_t)(J.W - 1) * (T + 64) + T - 1) / T);
    if (!persistent_couples) for (const TraceCouple& C : tc) D = std::max<int64_t>(D, ((int64_t)C.S_total + (int64_t)(C.W - 1) * (T + 64) + T - 1) / T);
    // range of bands with work in launch d: S_w(d) > 0 and S_w(d-1) < S_total
    auto active = [&](int64_t d, int W, int S_total, int band0, int64_t& g_lo, int64_t& g_hi) {
        int64_t w_hi = ((d + 1) * T + (T + 64) - 1) / (T + 64);               // first w with S_w(d) <= 0
        const int64_t w_lo = (d * T - S_total < 0) ? 0 : (d * T - S_total) / (T + 64);   // conservative
        w_hi = std::min<int64_t>(W, w_hi);
        if (w_lo >= w_hi) return;
        g_lo = std::min(g_lo, band0 + w_lo); g_hi = std::max(g_hi, band0 + w_hi);
    };
    // The odd job out of a group of couples runs BESIDE their launch, on a side stream (both single-launch forms take work as
    // tickets and wait only for work that is already running, so neither needs the other's waves to be resident); alone, it
    // has the chip to itself and may use the statically assigned workgroups of four linked bands.
    hipStream_t ss = s;
    bool single_on_side = false;
    if (persistent && persistent_couples) {
        if (!e->side[0] && hipStreamCreateWithFlags(&e->side[0], hipStreamNonBlocking) != hipSuccess) e->side[0] = nullptr;
        if (!e->ev_join[0] && hipEventCreateWithFlags(&e->ev_join[0], hipEventDisableTiming) != hipSuccess) e->ev_join[0] = nullptr;
        if (!e->ev_fork && hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess) e->ev_fork = nullptr;
        if (e->side[0] && e->ev_join[0] && e->ev_fork &&
            hipEventRecord(e->ev_fork, s) == hipSuccess && hipStreamWaitEvent(e->side[0], e->ev_fork, 0) == hipSuccess) { ss = e->side[0]; single_on_side = true; }
    }
    bool quad = false;
    e->last_trace_form = persistent_couples ? 5 : persistent ? 1 : 0;      // tsta_engine_stat("last_trace_form"); refined below
    if constexpr (TRACE && R <= 2) {
        quad = persistent && !persistent_couples && e->opt_trace_lds != 0 && ((single_bands + 3) / 4 <= e->n_cu || e->opt_trace_lds == 2);
        // split form: worth it once the pair has enough bands for the chain to matter (and only while every solver workgroup
        // gets a compute unit of its own)
        static const TraceJob none = {};
        const TraceJob& J0 = quad ? singles[0] : none;
        const int NC = (J0.S_total + kSplitC - 1) / kSplitC;
        bool split = false;
        if constexpr (R == 2) split = quad && e->opt_trace_split != 0 && (single_bands + 3) / 4 <= e->n_cu && (J0.W >= 16 || e->opt_trace_split == 2);
        const size_t ckpt_bytes = ((size_t)J0.W * NC * TraceCfg<R>::NS * 64 * sizeof(int) + 255) / 256 * 256;
        static const bool want_clock = getenv("TSTA_SPLIT_CLOCK") != nullptr;              // diagnostics
        if (split && e->reserve(e->ckpt, ckpt_bytes + (want_clock ? (size_t)J0.W * 16 : 0) + 256, "hipMalloc(checkpoints)") == TSTA_OK) {
            if (want_clock) A.dbg = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(e->ckpt.p) + ckpt_bytes);
            const int n_diag = (J0.W - 1) + (NC - 1) * kSplitK + 1;
            // (the table diag_first[u] = first ticket of diagonal u is built on the device, psa_split_diag_kernel: a copy from pageable
            // host memory here stalled the host for ~70 us between the init kernel and the fill)
            const int64_t total = (int64_t)J0.W * NC;
            if ((size_t)n_diag + 1 > split_tab_ints) return TSTA_ERR_INTERNAL;
            int* d_eprog = d_sync + 64 + single_bands + couple_bands;                       // cleared by the memset above
            int* d_diag = d_sync + 2 * single_bands + couple_bands + 128 + 2 * d_bound;     // behind the couples' tile table
            hipLaunchKernelGGL(psa_split_diag_kernel, dim3(1), dim3(256), 0, s, d_diag, n_diag, J0.W, NC, kSplitK, J0.state, TraceCfg<R>::NS, 2 * R + 4);
            ++launches;
            A.eprog = d_eprog; A.ckpt = reinterpret_cast<int*>(e->ckpt.p); A.C = kSplitC; A.NC = NC; A.K = kSplitK;
            A.n_solver_wg = (int)((single_bands + 3) / 4);
            A.tile_ticket = reinterpret_cast<unsigned*>(d_sync + 3); A.diag_first = d_diag; A.n_diag = n_diag; A.n_tiles = (int)total;
            A.d = 0; A.g_lo = 0; A.g_hi = (int)single_bands;
            // emitters: every compute unit ends up with two workgroups (a solver workgroup and an emitter workgroup, or two of the latter)
            // one workgroup per compute unit (LDS): the solvers' CUs carry four emitter waves each, the others eight
            int n_wg = std::max(e->n_cu, A.n_solver_wg);
#ifdef TSTA_EXPERIMENTS     // never in the shipped library: NOEMIT leaves the trace unwritten
            if (const char* v = getenv("TSTA_SPLIT_WG")) n_wg = std::max(A.n_solver_wg, atoi(v));
            if (getenv("TSTA_SPLIT_NOEMIT")) A.n_tiles = 0;
#endif
            if constexpr (R == 2) {
                // pure-ACGT pair and byte-sized substitution scores: the match / mismatch term as a table look-up (trace_step<.., ACGT>)
                const bool acgt = e->opt_psa_perm != 0 && p->M >= -128 && p->M <= 127 && p->X >= -128 && p->X <= 127 && pure_acgt(jobs[0]);
                if (acgt) hipLaunchKernelGGL((psa_trace_split_kernel<2, true>), dim3((unsigned)n_wg), dim3(kSplitThreads), 0, s, A);
                else      hipLaunchKernelGGL((psa_trace_split_kernel<2, false>), dim3((unsigned)n_wg), dim3(kSplitThreads), 0, s, A);
                e->last_trace_form = acgt ? 4 : 3;
            }
            ++launches;
        } else if (quad) {
            // one pair: bands linked through LDS inside a workgroup of four
            e->last_trace_form = 2;
            A.d = 0; A.g_lo = 0; A.g_hi = (int)single_bands;
            hipLaunchKernelGGL((psa_trace_quad_kernel<R>), dim3((unsigned)((single_bands + 3) / 4)), dim3(256), 0, s, A);
            ++launches;
        }
    }
    if (quad) {
    } else if (persistent) {                                    // the singles' whole fill in one launch (flag hand-off)
        A.d = 0; A.g_lo = 0; A.g_hi = (int)single_bands;
        hipLaunchKernelGGL((psa_trace_tile_kernel<R, TRACE, true>), dim3((unsigned)std::min<int64_t>((single_bands + 3) / 4, e->n_cu)), dim3(256), 0, ss, A);
        ++launches;
        if (single_on_side && ((rc = e->check(hipEventRecord(e->ev_join[0], ss), "hipEventRecord(join)")))) return rc;
    }
    std::vector<int> tile_tab;                                  // [first tile of d ...][D + 1], then [g_lo(d) ...][D]
    if (persistent_couples) {
        // every couple's whole fill as ONE launch in dataflow form: the tiles of the launch pipeline, enumerated d-major
        int64_t Dc = 0;
        for (const TraceCouple& C : tc) Dc = std::max<int64_t>(Dc, ((int64_t)C.S_total + (int64_t)(C.W - 1) * (T + 64) + T - 1) / T);
        if (Dc > d_bound) return TSTA_ERR_INTERNAL;
        tile_tab.assign((size_t)(2 * Dc + 1), 0);
        int64_t total = 0;
        for (int64_t d = 0; d < Dc; ++d) {
            int64_t g_lo = couple_bands, g_hi = 0;
            for (const TraceCouple& C : tc) active(d, C.W, C.S_total, C.band0, g_lo, g_hi);
            if (g_lo >= g_hi) { g_lo = 0; g_hi = 0; }
            tile_tab[(size_t)d] = (int)total; tile_tab[(size_t)(Dc + 1 + d)] = (int)g_lo;
            total += g_hi - g_lo;
            if (total > 0x7fffffff) return TSTA_ERR_UNSUPPORTED;
        }
        tile_tab[(size_t)Dc] = (int)total;
        int* d_tab = d_sync + 2 * single_bands + couple_bands + 128;      // behind the words the memset above clears
        if ((rc = e->check(hipMemcpyAsync(d_tab, tile_tab.data(), tile_tab.size() * sizeof(int), hipMemcpyHostToDevice, s), "H2D tile table"))) return rc;
        B.couples = d_couples; B.n_couples = n_couples; B.d = 0; B.g_lo = 0; B.g_hi = (int)couple_bands;
        B.tile_first = d_tab; B.tile_glo = d_tab + Dc + 1; B.n_d = (int)Dc; B.n_tiles = (int)total;
        launch_psa_trace_packed_persistent(B, e->n_cu, s);
        ++launches;
    }
    for (int64_t d = 0; d < D; ++d) {
        int64_t g_lo = single_bands, g_hi = 0;
        if (!persistent) for (const TraceJob& J : singles) active(d, J.W, J.S_total, J.band0, g_lo, g_hi);
        if (g_lo < g_hi) {
            A.d = (int)d; A.g_lo = (int)g_lo; A.g_hi = (int)g_hi;
            hipLaunchKernelGGL((psa_trace_tile_kernel<R, TRACE, false>), dim3((unsigned)((g_hi - g_lo + 3) / 4)), dim3(256), 0, s, A);
            ++launches;
        }
        for (int q = 0; q < n_chunks && !persistent_couples; ++q) {
            g_lo = couple_bands; g_hi = 0;
            for (int c = chunk_lo[q]; c < chunk_lo[q + 1]; ++c) active(d, tc[c].W, tc[c].S_total, tc[c].band0, g_lo, g_hi);
            if (g_lo >= g_hi) continue;
            B.couples = d_couples + chunk_lo[q]; B.n_couples = chunk_lo[q + 1] - chunk_lo[q];
            B.d = (int)d; B.g_lo = (int)g_lo; B.g_hi = (int)g_hi;
            launch_psa_trace_packed_tile(B, (int)((g_hi - g_lo + 3) / 4), q == 0 ? s : e->side[q - 1]);
            ++launches;
        }
    }
    for (int q = 1; q < n_chunks; ++q) {                         // join
        if ((rc = e->check(hipEventRecord(e->ev_join[q - 1], e->side[q - 1]), "hipEventRecord(join)")) ||
            (rc = e->check(hipStreamWaitEvent(s, e->ev_join[q - 1], 0), "hipStreamWaitEvent(join)")))
            return rc;
    }
    if (single_on_side && (rc = e->check(hipStreamWaitEvent(s, e->ev_join[0], 0), "hipStreamWaitEvent(join)"))) return rc;
    if (n_singles) hipLaunchKernelGGL(psa_trace_finish_kernel<R>, dim3(n_singles), dim3(n_singles == 1 ? 1024 : 256), 0, s, d_singles);
    if (n_couples) launch_psa_trace_packed_finish(d_jobs, d_couples, n_couples, s);     // (does not use band0)
    if constexpr (TRACE) {
        WalkTabArgs WT; WT.tab = nullptr; WT.Wc = 0; WT.nb = 0; WT.seg = nullptr; WT.lo = nullptr;
        // one long pair: the walk crosses its bands through a table (K3t above); worth its fixed cost (three launches, two of them bound
        // by the latency of ~150 dependent loads) from ~200 bands up
        const int nb = n_jobs == 1 ? (jobs[0].m - 1) / (64 * R) : 0;
        int Wc = std::min(e->opt_walk_corridor, jobs[0].n / 256 * 256);
        if constexpr (R == 2) if (n_jobs == 1 && Wc >= 256 && ((e->opt_walk_table == 1 && nb >= 192) || (e->opt_walk_table == 2 && nb >= 1))) {
            const size_t tab_b = ((size_t)(nb + 1) * Wc * sizeof(unsigned) + 255) / 256 * 256, seg_b = ((size_t)2 * (nb + 1) * sizeof(int) + 255) / 256 * 256;
            if (e->reserve(e->walk_tab, tab_b + seg_b + (size_t)(nb + 1) * sizeof(int) + 256, "hipMalloc(walk table)") == TSTA_OK) {
                WT.tab = reinterpret_cast<unsigned*>(e->walk_tab.p); WT.Wc = Wc; WT.nb = nb;
                WT.seg = reinterpret_cast<int*>(reinterpret_cast<char*>(e->walk_tab.p) + tab_b);
                WT.lo = reinterpret_cast<int*>(reinterpret_cast<char*>(e->walk_tab.p) + tab_b + seg_b);
                if ((rc = e->check(hipMemsetAsync(WT.seg, 0, seg_b, s), "hipMemsetAsync(walk segments)"))) return rc;
                hipLaunchKernelGGL((psa_walk_origin_kernel<R>), dim3((unsigned)((Wc + 255) / 256), (unsigned)nb), dim3(256), 0, s, d_jobs, WT);
                ++launches;
            } else { e->err[0] = '\0'; }
        }
        if (WT.tab) hipLaunchKernelGGL((psa_walk_kernel<R, 32, true>), dim3(n_jobs), dim3(64), 0, s, d_jobs, WT);
        else        hipLaunchKernelGGL((psa_walk_kernel<R, 32, false>), dim3(n_jobs), dim3(64), 0, s, d_jobs, WT);
        if constexpr (R == 2) if (WT.tab) { hipLaunchKernelGGL((psa_walk_emit_kernel<R>), dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, s, d_jobs, WT); ++launches; }
    }
    launches += 3;
    e->n_launches = launches;
    e->timing_end(s);
    if ((rc = e->check(hipGetLastError(), "psa_align launches"))) return rc;
    std::vector<int> h_out((size_t)n_jobs * 8);
    if ((rc = e->check(hipMemcpyAsync(h_out.data(), d_out, h_out.size() * sizeof(int), hipMemcpyDeviceToHost, s), "D2H out")) ||
        (rc = e->check(hipStreamSynchronize(s), "psa_align sync")))
        return rc;
    if (A.dbg) {
        std::vector<unsigned long long> h((size_t)2 * single_bands);
        if (e->copy_sync(h.data(), A.dbg, h.size() * 8, hipMemcpyDeviceToHost, "D2H clocks") == TSTA_OK) {
            const double us = 0.01;                             // s_memrealtime: 100 MHz
            fprintf(stderr, "[tsta] split solvers: band, start us, duration us, start - previous start us\n");
            for (int64_t b = 0; b < single_bands; ++b)
                if (b < 24 || b % 20 == 0 || b + 4 >= single_bands)
                    fprintf(stderr, "[tsta] %5lld %10.1f %10.1f %8.2f\n", (long long)b, (double)(h[2 * b] - h[0]) * us, (double)(h[2 * b + 1] - h[2 * b]) * us,
                            b ? (double)(h[2 * b] - h[2 * b - 2]) * us : 0.0);
            fprintf(stderr, "[tsta] last band ends at %.1f us\n", (double)(h[2 * single_bands - 1] - h[0]) * us);
        }
    }
    if (persistent || persistent_couples) {
        unsigned st = 0;
        if ((rc = e->copy_sync(&st, d_sync + 1, sizeof(st), hipMemcpyDeviceToHost, "D2H status"))) return rc;
        if (st) {
            // a wait was abandoned (the chip shared with other work, or the test hook): everything this launch left is
            // suspect.  Same behaviour as the other single-launch families (psa_api.hip, poa_fill.hip): the call is run
            // again, now, through the launch-per-diagonal pipeline — same kernels, the kernel boundary as only synchronisation.
            ++e->single_launch_reruns;
            return run_group<R, TRACE>(e, p, jobs, n_jobs, T, false);
        }
    }
    // the walkers' ops: per-job copies for a few long alignments, one copy of the whole region for many short ones
    std::vector<uint8_t> h_ops;
    const bool one_copy = TRACE && n_jobs > 16;
    if (one_copy) {
        h_ops.resize(ops_b);
        if ((rc = e->check(hipMemcpyAsync(h_ops.data(), e->scratch2.p, ops_b, hipMemcpyDeviceToHost, s), "D2H ops")) ||
            (rc = e->check(hipStreamSynchronize(s), "psa_align sync")))
            return rc;
    }
    for (int j = 0; j < n_jobs; ++j) {
        jobs[j].maxscore = h_out[8 * j]; jobs[j].end_score = h_out[8 * j + 1];
        if (!TRACE) continue;
        if (n_jobs == 1) e->last_walk_jumps = h_out[8 * j + 5];
        const int cnt = h_out[8 * j + 2]; jobs[j].fi = h_out[8 * j + 3]; jobs[j].fj = h_out[8 * j + 4];
        if (cnt < 0 || cnt > jobs[j].n + jobs[j].m) { e->set_error("walker returned %d ops", cnt); return TSTA_ERR_INTERNAL; }
        if (one_copy) {
            const uint8_t* src = h_ops.data() + (tj[j].ops - reinterpret_cast<const uint8_t*>(e->scratch2.p));
            jobs[j].ops.assign(src, src + cnt);
            continue;
        }
        jobs[j].ops.resize(cnt);
        if (cnt && (rc = e->check(hipMemcpyAsync(jobs[j].ops.data(), tj[j].ops, cnt, hipMemcpyDeviceToHost, s), "D2H ops"))) return rc;
    }
    return e->check(hipStreamSynchronize(s), "psa_align sync");
}

// Score-only fill of ONE oriented pair (a = columns, b = rows; host pointers) spread over the whole chip.
int psa_score_single_pair(Engine* e, const tsta_params* p, const uint8_t* a, int n, const uint8_t* b, int m, int32_t* maxscore)
{
    HostJob j; j.a = a; j.n = n; j.b = b; j.m = m;
    const int T = (e->opt_trace_tile >= 64 && e->opt_trace_tile % 64 == 0) ? e->opt_trace_tile : 64;
    const int rc = run_group<4, false>(e, p, &j, 1, T);
    *maxscore = j.maxscore;
    return rc;
}

// the two gapped rows of one pair from its walker ops (psa.c:464-516: leftover prefix of a against gaps, leftover
// prefix of b against gaps, then reverse); malloc'd, NUL-terminated
static int assemble_rows(Engine* e, const HostJob& J, char** row_a, char** row_b, int64_t* out_len)
{
    const uint8_t* a = J.a; const uint8_t* b = J.b;
    const int64_t len = (int64_t)J.ops.size() + J.fi + J.fj;
    char* ra = (char*)malloc((size_t)len + 1);
    char* rb = (char*)malloc((size_t)len + 1);
    if (!ra || !rb) { free(ra); free(rb); return TSTA_ERR_OOM; }
    ra[len] = rb[len] = '\0';
    int64_t q = len;
    int i = J.m, j = J.n;
    for (size_t x = 0; x < J.ops.size(); ++x) {
        const int k = J.ops[x];
        if (k == 1)      { ra[q - 1] = (char)a[j - 1]; rb[q - 1] = (char)b[i - 1]; --i; --j; }
        else if (k == 0) { ra[q - 1] = (char)a[j - 1]; rb[q - 1] = '-'; --j; }
        else             { ra[q - 1] = '-'; rb[q - 1] = (char)b[i - 1]; --i; }
        --q;
    }
    if (i != J.fi || j != J.fj) { free(ra); free(rb); e->set_error("walker/ops mismatch"); return TSTA_ERR_INTERNAL; }
    while (j >= 1) { ra[q - 1] = (char)a[j - 1]; rb[q - 1] = '-'; --j; --q; }
    while (i >= 1) { rb[q - 1] = (char)b[i - 1]; ra[q - 1] = '-'; --i; --q; }
    *row_a = ra; *row_b = rb; *out_len = len;
    return TSTA_OK;
}

}  // namespace tsta

extern "C" int tsta_psa_align(tsta_engine* eng, const tsta_params* p, const uint8_t* s1, int64_t l1,
                              const uint8_t* s2, int64_t l2, int32_t* maxscore, char** row_a, char** row_b,
                              int64_t* out_len)
{
    using namespace tsta;
    if (!eng || !p || l1 < 0 || l2 < 0 || !maxscore || !row_a || !row_b || !out_len) return TSTA_ERR_INVALID_ARG;
    if ((l1 && !s1) || (l2 && !s2)) return TSTA_ERR_INVALID_ARG;
    if (l1 > (1 << 24) || l2 > (1 << 24)) return TSTA_ERR_UNSUPPORTED;
    Engine* e = reinterpret_cast<Engine*>(eng);
    int rc = e->check(hipSetDevice(e->device), "hipSetDevice");
    if (rc != TSTA_OK) return rc;
    // psa.c:194-199: the strictly longer input becomes seq[0] = columns = ">1"
    const bool swap = l2 > l1;
    HostJob J;
    J.a = swap ? s2 : s1; J.n = (int)(swap ? l2 : l1);
    J.b = swap ? s1 : s2; J.m = (int)(swap ? l1 : l2);
    J.fi = J.m; J.fj = J.n;
    if (J.m > 0 && J.n > 0) {
        // rows per lane: small bands expose more concurrent bands of the single pair
        const int R_env = e->opt_trace_rows;                                                    // knob "trace_rows"
        const int Rsel = (R_env == 1 || R_env == 2 || R_env == 4) ? R_env : 2;   // measured on 100 kbp x 100 kbp: R=2,T=64 40 ms; R=4,T=128 50 ms
        const int T = (e->opt_trace_tile >= 64 && e->opt_trace_tile % 64 == 0) ? e->opt_trace_tile : 64;   // knob "trace_tile"
        if (Rsel == 1)      rc = run_group<1, true>(e, p, &J, 1, T);
        else if (Rsel == 4) rc = run_group<4, true>(e, p, &J, 1, T);
        else           rc = run_group<2, true>(e, p, &J, 1, T);
        if (rc != TSTA_OK) return rc;
    }
    if ((rc = assemble_rows(e, J, row_a, row_b, out_len)) != TSTA_OK) return rc;
    *maxscore = J.maxscore;
    return psa_reference_warning(e, p);       // TSTA_OK, or the scoring is outside the reference's int8 domain (tsta_psa_reference_exact)
}

extern "C" void tsta_alignments_free(tsta_alignment* out, int64_t n_pairs)
{
    if (!out) return;
    for (int64_t k = 0; k < n_pairs; ++k) { free(out[k].row_a); free(out[k].row_b); out[k].row_a = out[k].row_b = nullptr; }
}

extern "C" int tsta_psa_align_batch(tsta_engine* eng, const tsta_params* p, int64_t n_pairs, const uint8_t* seqs,
                                    int64_t seqs_bytes, const int64_t* off1, const int32_t* len1,
                                    const int64_t* off2, const int32_t* len2, tsta_alignment* out)
{
    using namespace tsta;
    if (!eng || !p || n_pairs < 0) return TSTA_ERR_INVALID_ARG;
    if (n_pairs == 0) return TSTA_OK;
    if (!off1 || !len1 || !off2 || !len2 || !out || seqs_bytes < 0 || (seqs_bytes && !seqs)) return TSTA_ERR_INVALID_ARG;
    if (n_pairs > (1 << 24)) return TSTA_ERR_UNSUPPORTED;
    for (int64_t k = 0; k < n_pairs; ++k) {
        if (len1[k] < 0 || len2[k] < 0 || off1[k] < 0 || off2[k] < 0 || off1[k] + len1[k] > seqs_bytes || off2[k] + len2[k] > seqs_bytes)
            return TSTA_ERR_INVALID_ARG;
        if (len1[k] > (1 << 24) || len2[k] > (1 << 24)) return TSTA_ERR_UNSUPPORTED;
    }
    Engine* e = reinterpret_cast<Engine*>(eng);
    int rc = e->check(hipSetDevice(e->device), "hipSetDevice");
    if (rc != TSTA_OK) return rc;
    for (int64_t k = 0; k < n_pairs; ++k) { out[k].maxscore = -100; out[k].len = 0; out[k].row_a = out[k].row_b = nullptr; }

    std::vector<HostJob> jobs((size_t)n_pairs);
    int64_t bands4 = 0;
    for (int64_t k = 0; k < n_pairs; ++k) {
        const bool swap = len2[k] > len1[k];                       // psa.c:194-199
        HostJob& J = jobs[(size_t)k];
        J.a = seqs + (swap ? off2[k] : off1[k]); J.n = swap ? len2[k] : len1[k];
        J.b = seqs + (swap ? off1[k] : off2[k]); J.m = swap ? len1[k] : len2[k];
        J.fi = J.m; J.fj = J.n;
        bands4 += (J.m + 255) / 256;
    }
    // rows per lane / tile length: with enough bands in flight to fill every SIMD several times over, taller bands and
    // longer tiles cost fewer instructions per cell; a few long pairs want many short bands instead (see tsta_psa_align)
    const int R_env = e->opt_trace_rows;
    const int Rsel = (R_env == 2 || R_env == 4) ? R_env : (bands4 >= (int64_t)e->n_cu * 16 ? 4 : 2);
    const int T = (e->opt_trace_tile >= 64 && e->opt_trace_tile % 64 == 0) ? e->opt_trace_tile : (bands4 >= (int64_t)e->n_cu * 16 ? 128 : 64);
    // groups of pairs whose workspaces (1 B per DP cell of trace, mostly) fit the device together
    size_t free_b = 0, total_b = 0;
    if ((rc = e->check(hipMemGetInfo(&free_b, &total_b), "hipMemGetInfo")) != TSTA_OK) return rc;
    const size_t held = e->in_seqs.bytes + e->rows.bytes + e->scratch.bytes + e->trace.bytes + e->scratch2.bytes + e->in_meta.bytes;
    size_t budget = (size_t)((double)(free_b + held) * 0.85);
    if (const char* v = getenv("TSTA_ALIGN_BUDGET_MB")) budget = std::min(budget, (size_t)atoll(v) << 20);
    const size_t hard_budget = budget;          // what the device (or TSTA_ALIGN_BUDGET_MB) allows at all
    // Device memory costs ~35 ms per GB to allocate on this platform (measured: hipMalloc 8 / 32 / 128 GB = 0.27 / 1.2 / 4.8 s,
    // the same through hipMemCreate; tests/tools/malloc_time.cpp), so a one-shot caller (TSTA_psa_batch -o) whose 64 pairs
    // would like 165 GB of trace must not pay 5.7 s for 70 ms of kernels.  The workspace therefore grows over the calls of
    // an engine: the first call works in groups of at most 16 GB (0.5 s to allocate), the next in 64 GB, then all of it;
    // option "align_ws_gb" fixes the cap.  A workspace the engine already holds is always used in full.
    {
        size_t cap = e->opt_align_ws_gb > 0 ? (size_t)e->opt_align_ws_gb << 30
                                            : (size_t)16 << (30 + 2 * std::min(e->align_calls, 8));
        cap = std::max(cap, held);
        budget = std::min(budget, cap);
        ++e->align_calls;
    }
    // (a coupled job is laid out for its couple's geometry: up to 1.25x its own cells)
    auto job_bytes = [&](const HostJob& J) { return Rsel == 4 ? JobGeom<4>(J.n, J.m, true).total() * 5 / 4 : JobGeom<2>(J.n, J.m, true).total(); };
    std::vector<HostJob> group;
    std::vector<int64_t> group_idx;
    size_t used = 0;
    auto flush = [&]() -> int {
        if (group.empty()) return TSTA_OK;
        int r = Rsel == 4 ? run_group<4, true>(e, p, group.data(), (int)group.size(), T)
                          : run_group<2, true>(e, p, group.data(), (int)group.size(), T);
        if (r != TSTA_OK) return r;
        for (size_t x = 0; x < group.size(); ++x) {
            tsta_alignment& o = out[group_idx[x]];
            if ((r = assemble_rows(e, group[x], &o.row_a, &o.row_b, &o.len)) != TSTA_OK) return r;
            o.maxscore = group[x].maxscore;
        }
        group.clear(); group_idx.clear(); used = 0;
        return TSTA_OK;
    };
    for (int64_t k = 0; k < n_pairs && rc == TSTA_OK; ++k) {
        HostJob& J = jobs[(size_t)k];
        if (J.m == 0 || J.n == 0) {                       // nothing to fill: one sequence against gaps
            tsta_alignment& o = out[k];
            rc = assemble_rows(e, J, &o.row_a, &o.row_b, &o.len);
            continue;
        }
        const size_t need = job_bytes(J);
        if (need > budget && need <= hard_budget) { rc = flush(); budget = need; if (rc != TSTA_OK) break; }   // one pair larger than the soft cap
        if (need > budget) { e->set_error("pair %lld needs %zu bytes of device workspace, %zu available", (long long)k, need, budget); rc = TSTA_ERR_OOM; break; }
        if (used + need > budget || group.size() >= 65535) rc = flush();
        if (rc != TSTA_OK) break;
        group.push_back(J); group_idx.push_back(k); used += need;
    }
    if (rc == TSTA_OK) rc = flush();
    if (rc != TSTA_OK) { tsta_alignments_free(out, n_pairs); return rc; }
    return psa_reference_warning(e, p);
}
