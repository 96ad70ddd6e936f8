// poa_dataflow.hip — K4d: the POA fill of poa_fill.hip as ONE launch, with the work of a row split over the waves of a
// workgroup.  Same results (reference: msa/c-t-simd.c:165-599, 602-657, 979-1045; see poa_fill.hip).
//
// Why.  The README example is a latency problem: 5 000 columns are only 40 tiles wide, so a fill is as long as
// the chain of its rows, V x (time per row), and a lone wave issues one instruction every ~8 cycles: the time per row IS the
// number of instructions on the wave that owns the chain.  K4 spends ~460 instructions per row and a launch per tile
// anti-diagonal.  Here:
//
//  1. One workgroup per chunk of 128 columns, one launch.  Workgroup q sweeps ALL the nodes of its chunk, 32 at a time (a
//     "block"); it takes over, per row, three values from the chunk on its left (F carry, open flag, H at the edge: `hand`,
//     sc1 stores / loads, handoff.h) once that chunk has published the block (prog[q]), and publishes its own.  A workgroup
//     waits only for its left neighbour, and roles are tickets drawn in the order the workgroups start (the neighbour is always
//     already running): the waits end whatever the dispatch order or the residency.  Column 0 — H(r, 0), a function of the
//     graph alone — is one more workgroup (a single wave) that plays "left neighbour" for chunk 0.
//  2. Roles.  Wave 0, the SOLVER, runs only what the next row needs — predecessor maxima, the max-plus scan of F, final
//     H / Eagg / F — and leaves six dwords per lane and row in an LDS ring of three blocks: ~55 instructions per backbone row
//     (one predecessor = the row before, tight loop), ~100 for a row with several / non-adjacent predecessors whose records
//     are in the ring.  Four EMITTER waves (rows i mod 4) work out again, from the records of a row's predecessors, everything
//     the reference's three trace codes record (c-t-simd.c:485-506, 521-538, 564-565) and do all the stores.  The FETCHER wave
//     reads what a block needs from global memory — DAG metadata, the left neighbour's hand-over — up to two blocks ahead into
//     LDS, polling the neighbour's flag: no global-memory latency sits on the chain of rows.
//  3. Nobody waits at a barrier: records are followed by a counter (the LDS performs one wave's operations in order), a block
//     is published by whichever wave finishes it last.
//
// Scope: 2 columns per lane, any length (with more chunks than CUs the later ones start as CUs come free), sequences without a
// literal 'N' (the N-column rule needs the segment-by-segment scan of poa_fill.hip: those run the launch-per-diagonal K4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

#include "../../include/tsta_engine.h"
#include "engine.h"
#include "handoff.h"
#include "poa_types.h"

namespace tsta {

constexpr int kDfKC = 2;                 // columns per lane
constexpr int kDfCW = 64 * kDfKC;        // columns per chunk
constexpr int kDfRB = 32;                // nodes per block
constexpr int kDfThreads = 448;          // solver (wave 0), emitters (waves 1-3, 5, 6), fetcher (wave 4)
constexpr int kDfEmit = 5;               // emitter waves
constexpr int kDfRing = 3 * kDfRB;       // rows whose records stay in LDS: three blocks (the solver writes one and reads the one before it,
                                         // the emitters of that one still read the third): 96 x 6 x 64 ints = 144 KB

// What the solver leaves per lane and row in LDS (rec[row][field][lane]: every access is 64 consecutive dwords,
// conflict-free; the fields of a row are reached from one address): H, Eagg and F of the lane's two columns — only what the
// NEXT rows need.  Everything else a trace code records (which predecessor won, whether a gap opened) the emitters work out
// again from the records of the row's predecessors.
enum { F_HS0, F_HS1, F_ES0, F_ES1, F_FL0, F_FL1, F_COUNT };
// Per-row side data next to the records (side[what][row slot]): H left of the chunk, "first F opens", and the row's
// predecessor list for the emitters (number, offset, the first four with their row slots).
enum { S_EDGE, S_FLAG, S_NP, S_POFF, S_PR0, S_PR1, S_PR2, S_PR3, S_PS0, S_PS1, S_PS2, S_PS3, S_COUNT };
// What the fetcher wave stages per row of a block: the DAG metadata (offset / number of predecessors, the first four of them
// with their row slots, the node's base) and what the chunk on the left handed over (F carry, open flag, edge H).
enum { FB_POFF, FB_NP, FB_PR0, FB_PR1, FB_PR2, FB_PR3, FB_PS0, FB_PS1, FB_PS2, FB_PS3, FB_BASE, FB_XIN, FB_FLAG, FB_EDGE, FB_COUNT };

__device__ __forceinline__ int df_shr1(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false); }

// bounded wait on an LDS word (solver -> emitters of one workgroup); acquires what was written before it
__device__ __forceinline__ bool lds_wait_gt(int* word, int v, unsigned* status, int lane)
{
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= v) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 24)) { if (lane == 0) atomicOr(status, 2u); return false; }      // ~1 s: far beyond any ramp-up of the chunks
        // once any wave of the launch has given up, nobody waits any more
        if ((spins & 4095) == 0 && __hip_atomic_load((const GlobalU32*)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}

// the same for a counter that only grows: `seen` keeps the last value read, so a reader that is behind does not look again
__device__ __forceinline__ bool lds_wait_gt_seen(int* word, int v, int& seen, unsigned* status, int lane)
{
    if (seen > v) return true;
    int spins = 0;
    for (;;) {
        seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (seen > v) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 24)) { if (lane == 0) atomicOr(status, 2u); return false; }
        if ((spins & 4095) == 0 && __hip_atomic_load((const GlobalU32*)status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}

// F of one row given the column maxima a[k] = max(diagonal, vertical-gap) candidates: local pass with no carry-in, max-plus
// inclusive scan over the 64 lanes (DPP, VALU only), fix-up pass with the true carry (poa_fill.hip has the derivation).
// Out: H, F per column, per column whether F(c+1) opens from H(c).  (The carry leaving the chunk is max(F, H+O) of the last
// column: the emitters take it from lane 63's record.)
__device__ __forceinline__ void df_scan_row(const int (&a)[kDfKC], int xin_row, int cl, int cl1,
                                            int E, int O, int (&hs)[kDfKC], int (&fl)[kDfKC], int (&opn_next)[kDfKC])
{
    // local pass, no carry-in: F of the lane's first column is -inf, so H = a there (every a is a real score)
    int fbar = a[0] + O;
#pragma unroll
    for (int k = 1; k < kDfKC; ++k) {
        const int f = fbar + E;
        const int h = max(a[k], f);
        fbar = max(f, h + O);
    }
    // Carry entering lane l = max(xin + l*aK, max_{j<l} fbar_j + (l-1-j)*aK).  In the frame u_j = fbar_j - (j+1)*aK the
    // max-plus scan is a plain max scan: t = u shifted one lane up with xin entering lane 0, inclusive max scan of t
    // (six v_max_i32_dpp: a lane without a source keeps its value), carry = scan + l*aK.
    const int kMin = (int)0x80000000;
    int t = df_shr1(xin_row, fbar - cl1);
    t = max(t, __builtin_amdgcn_update_dpp(kMin, t, 0x111, 0xf, 0xf, false));
    t = max(t, __builtin_amdgcn_update_dpp(kMin, t, 0x112, 0xf, 0xf, false));
    t = max(t, __builtin_amdgcn_update_dpp(kMin, t, 0x114, 0xf, 0xf, false));
    t = max(t, __builtin_amdgcn_update_dpp(kMin, t, 0x118, 0xf, 0xf, false));
    t = max(t, __builtin_amdgcn_update_dpp(kMin, t, 0x142, 0xa, 0xf, false));
    t = max(t, __builtin_amdgcn_update_dpp(kMin, t, 0x143, 0xc, 0xf, false));
    fbar = t + cl;
#pragma unroll
    for (int k = 0; k < kDfKC; ++k) {
        const int f = fbar + E;
        const int h = max(a[k], f);
        hs[k] = h; fl[k] = f;
        const int g = h + O;
        opn_next[k] = (g >= f) ? 1 : 0;                  // F(c+1) == H(c)+O+E  <=>  H(c)+O >= F(c)
        fbar = max(f, g);
    }
}

__global__ __launch_bounds__(kDfThreads) void poa_dataflow_kernel(PoaDfArgs P)
{
    const PoaArgs& A = P.A;
    __shared__ int rec[kDfRing][F_COUNT][64];            // 144 KB, block b in slots (b % 3) * 32 ..
    __shared__ int side[S_COUNT][kDfRing];
    __shared__ int produced[64];                         // rows the solver has finished since the kernel began (64 copies: every lane stores
                                                         // and polls its own, no exec juggling on the chain of rows)
    __shared__ int blk_done[4];                          // waves (of 4) that have finished block b, slot b & 3
    __shared__ int blk_pub;                              // blocks published (all four waves done, stores drained)
    __shared__ int fbuf[2][FB_COUNT][kDfRB];             // what the fetcher read for block b (slot b & 1), one row per entry
    __shared__ int fetched;                              // blocks the fetcher has put into fbuf
    __shared__ int taken;                                // blocks the solver has copied out of fbuf
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the roles' loops stay uniform)
    const int n = A.n, M = A.M, X = A.X, E = A.E, O = A.O, OE = A.O + A.E;
    const int aK = kDfKC * E;
    const int ncar = A.nchunks + 1;
    const int cl = lane * aK, cl1 = (lane + 1) * aK;

    if (threadIdx.x == 0) { blk_pub = 0; fetched = 0; taken = 0; blk_done[0] = blk_done[1] = blk_done[2] = blk_done[3] = 0; }
    if (threadIdx.x < 64) produced[threadIdx.x] = 0;
    // Roles are handed out as tickets in the order the workgroups actually start (HIP promises no dispatch order): ticket 0 is
    // column 0, ticket t chunk t-1.  A workgroup only ever waits for the holder of the ticket before its own, which is already
    // running: no deadlock whatever the residency — with more chunks than CUs the later chunks simply start as CUs come free.
    __shared__ int ticket_s;
    if (threadIdx.x == 0) ticket_s = (int)atomicAdd(P.ticket, 1u);
    __syncthreads();
    const int q = __builtin_amdgcn_readfirstlane(ticket_s) - 1;      // this workgroup's chunk; -1: the column-0 solver below
    const bool mute = P.test_abandon != 0 && q + 1 == P.test_abandon - 1 && q >= 0;   // test hook: this chunk never publishes a block
    if (q < 0) {
        // ================================================================ column 0 (c-t-simd.c:174-188, 194-200)
        // H(r, 0) = max over predecessors H(p, 0) + E (a source: O + E) depends on the graph only.  One wave walks the nodes
        // ahead of chunk 0 and hands it, per row, what every other chunk gets from the chunk on its left: the F carry
        // (F(r, 0) = -inf, so the carry is H + O and the first F opens), and H at the "edge".  Chunk 0 is then a chunk like
        // any other.
        // Nothing global sits on its per-block path either: the predecessor lists are loaded one block ahead (their offsets two),
        // the previous block's values stay in a register (older ones are read back), a block is published once its stores
        // have had the next block's time to drain.
        if (wave != 0) return;
        const int nblk = (A.V + kDfRB - 1) / kDfRB;
        struct Lists { int poff, np, pr[4]; };
        auto load_offsets = [&](int rb, int& poff, int& np) {
            const int r0 = rb * kDfRB, nrows = min(A.V, r0 + kDfRB) - r0;
            poff = 0; np = 0;
            if (rb < nblk && lane < nrows) { poff = A.pred_off[r0 + lane]; np = A.pred_off[r0 + lane + 1] - poff; }
        };
        auto load_lists = [&](int poff, int np) -> Lists {
            Lists L; L.poff = poff; L.np = np;
#pragma unroll
            for (int x = 0; x < 4; ++x) L.pr[x] = (x < np) ? A.pred[poff + x] : -1;
            return L;
        };
        int poff_n, np_n;
        load_offsets(0, poff_n, np_n);
        Lists cur = load_lists(poff_n, np_n);            // block 0
        load_offsets(1, poff_n, np_n);                   // block 1's offsets
        int h_prev = 0;                                  // lane i: H(r0 - 32 + i, 0), the previous block
        for (int rb = 0; rb < nblk; ++rb) {
            const int r0 = rb * kDfRB, nrows = min(A.V, r0 + kDfRB) - r0;
            __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): everything issued a block ago (lists, offsets, stores) has landed
            if (rb > 0) { if (lane == 0) __hip_atomic_store((GlobalI32*)P.prog, rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // block rb-1
            const Lists L = cur;
            cur = load_lists(poff_n, np_n);              // block rb+1 (offsets loaded a block ago)
            load_offsets(rb + 2, poff_n, np_n);
            // max H(p, 0) over the predecessors before this block, one row per lane
            int ext = kNeg;
            for (int j = 0; __ballot(lane < nrows && j < L.np) != 0ull; ++j) {       // (uniform loop: the shuffle needs every lane)
                const bool mine = lane < nrows && j < L.np;
                int p = (j == 0) ? L.pr[0] : (j == 1) ? L.pr[1] : (j == 2) ? L.pr[2] : (j == 3) ? L.pr[3] : -1;
                if (mine && j >= 4) p = A.pred[L.poff + j];
                int hv = __shfl(h_prev, (p - (r0 - kDfRB)) & 63, 64);
                if (mine && p < r0 - kDfRB) hv = load_sc1_int(&P.hand[((size_t)p * ncar) * 4 + 2]);
                if (mine && p < r0) ext = max(ext, hv);
            }
            int h_l = 0;                                 // lane i: H(r0 + i, 0)
            for (int i = 0; i < nrows; ++i) {
                const int npi = __builtin_amdgcn_readlane(L.np, i);
                int h = OE;
                if (npi > 0) {
                    h = __builtin_amdgcn_readlane(ext, i);
                    const int p0 = __builtin_amdgcn_readlane(L.poff, i);
                    for (int j = 0; j < npi; ++j) {
                        int p;
                        if (j == 0) p = __builtin_amdgcn_readlane(L.pr[0], i);
                        else if (j == 1) p = __builtin_amdgcn_readlane(L.pr[1], i);
                        else if (j == 2) p = __builtin_amdgcn_readlane(L.pr[2], i);
                        else if (j == 3) p = __builtin_amdgcn_readlane(L.pr[3], i);
                        else p = __builtin_amdgcn_readfirstlane(A.pred[p0 + j]);
                        if (p >= r0) h = max(h, __builtin_amdgcn_readlane(h_l, p - r0));
                    }
                    h += E;
                }
                h_l = (lane == i) ? h : h_l;
            }
            if (lane < nrows) {
                int* hd = P.hand + ((size_t)(r0 + lane) * ncar) * 4;
                store_sc1_int2(reinterpret_cast<int2*>(hd), h_l + O, 1);
                store_sc1_int2(reinterpret_cast<int2*>(hd + 2), h_l, 0);
            }
            h_prev = h_l;
        }
        publish_progress(P.prog, nblk, lane);
        return;
    }
    const int cbase = q * kDfCW, c0 = cbase + lane * kDfKC;
    int sch[kDfKC];
#pragma unroll
    for (int k = 0; k < kDfKC; ++k) sch[k] = (c0 + k < n) ? (int)A.seq[c0 + k] : -1;
    int rows_base = 0;                                   // rows the solver has produced in earlier blocks
    // solver state that lives across blocks: the row just finished (the backbone predecessor of the next one)
    int hs_prev[kDfKC] = {0, 0}, es_prev[kDfKC] = {0, 0}, prev_r = -2, edge_prev = 0;
    const int nblocks = (A.V + kDfRB - 1) / kDfRB;
    if (wave == 4) {
        // ================================================================ fetcher
        // Everything the solver needs from global memory at the start of a block — the DAG metadata of its rows and what the
        // chunk on the left handed over — is read by this wave up to two blocks ahead and left in LDS: no global-memory
        // latency (and no wait for a neighbour's flag) sits on the chain of rows.  The flag of the chunk on the left is
        // polled here (prog[q]; prog[0] belongs to column 0).
        // Software-pipelined like column 0: a block's predecessor lists are loaded while the block before it is handed over,
        // their offsets one block earlier still; per block only the neighbour's flag and its hand-over are waited for.
        struct FMeta { int poff, np, base, pr[4], ps[4]; };
        auto load_offsets = [&](int rb, int& poff, int& np, int& base) {
            const int r0 = rb * kDfRB, count = min(A.V, r0 + kDfRB) - r0;
            poff = 0; np = 0; base = 0;
            if (rb < nblocks && lane < count) { poff = A.pred_off[r0 + lane]; np = A.pred_off[r0 + lane + 1] - poff; base = A.base[r0 + lane]; }
        };
        auto load_lists = [&](int poff, int np, int base) -> FMeta {
            FMeta m; m.poff = poff; m.np = np; m.base = base;
#pragma unroll
            for (int x = 0; x < 4; ++x) { m.pr[x] = -1; m.ps[x] = 0; if (x < np) { m.pr[x] = A.pred[poff + x]; m.ps[x] = A.pred_slot[poff + x]; } }
            return m;
        };
        int poff_n, np_n, base_n;
        load_offsets(0, poff_n, np_n, base_n);
        FMeta nxt = load_lists(poff_n, np_n, base_n);    // block 0
        load_offsets(1, poff_n, np_n, base_n);
        for (int rb = 0; rb < nblocks; ++rb) {
            const int r0 = rb * kDfRB, count = min(A.V, r0 + kDfRB) - r0;
            __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): issued an iteration ago
            const FMeta m = nxt;
            nxt = load_lists(poff_n, np_n, base_n);      // block rb+1
            load_offsets(rb + 2, poff_n, np_n, base_n);
            const int poff = m.poff, np = m.np, base = m.base;
            const int (&pr)[4] = m.pr; const int (&ps)[4] = m.ps;
            if (rb >= 2 && !lds_wait_gt(&taken, rb - 2, P.status, lane)) break;     // the solver has copied block rb-2 out of this slot
            wait_progress(P.prog + q, rb + 1, P.status, lane);
            // a launch in which any wait was abandoned stops here: nothing more is staged, the solver and the emitters run
            // into their own (bounded) waits and return; the entry point runs the call again through K4
            if (__hip_atomic_load((const GlobalU32*)P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            int xin = 0, flag = 1, edge = 0;
            if (lane < count) {
                const int* hd = P.hand + ((size_t)(r0 + lane) * ncar + q) * 4;
                const int2 a = load_sc1_int2(reinterpret_cast<const int2*>(hd)), b = load_sc1_int2(reinterpret_cast<const int2*>(hd + 2));
                xin = a.x; flag = a.y; edge = b.x;
            }
            if (lane < kDfRB) {
                int (*fb)[kDfRB] = fbuf[rb & 1];
                fb[FB_POFF][lane] = poff; fb[FB_NP][lane] = np; fb[FB_BASE][lane] = base;
                fb[FB_PR0][lane] = pr[0]; fb[FB_PR1][lane] = pr[1]; fb[FB_PR2][lane] = pr[2]; fb[FB_PR3][lane] = pr[3];
                fb[FB_PS0][lane] = ps[0]; fb[FB_PS1][lane] = ps[1]; fb[FB_PS2][lane] = ps[2]; fb[FB_PS3][lane] = ps[3];
                fb[FB_XIN][lane] = xin; fb[FB_FLAG][lane] = flag; fb[FB_EDGE][lane] = edge;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the LDS performs one wave's operations in order
            if (lane == 0) __hip_atomic_store(&fetched, rb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
            return;
    }
    // One predecessor row as a lane sees it: H at the column left of the lane's first column, H and Eagg at its two columns.
    struct PredVals { int hl, h0, h1, e0, e1; };
    // ... from global memory: a row older than the ring (stored by this workgroup's emitters; its block was published before
    // the reader's block began), or the virtual init row (slot 0) of a source node
    auto pred_far = [&](int p, int ps, bool is_src) -> PredVals {
        const int* Hp = A.Hrow + (size_t)ps * A.pitch + c0 + 3;    // H(p, c0 .. c0+2)
        const int* Ep = A.Erow + (size_t)ps * A.pitch + c0;        // Eagg(p, c0+1 .. c0+2)
        PredVals v;
        // the column left of the chunk belongs to the chunk on the left: its per-(row, chunk) copy
        v.hl = (lane == 0 && !is_src) ? load_sc1_int(&P.hand[((size_t)p * ncar + q) * 4 + 2]) : load_sc1_int(Hp);
        const int2 ha = load_sc1_int2(reinterpret_cast<const int2*>(Hp + 1));
        const int2 ea = load_sc1_int2(reinterpret_cast<const int2*>(Ep));
        v.h0 = ha.x; v.h1 = ha.y; v.e0 = ea.x; v.e1 = ea.y;
        return v;
    };
    int blk3 = 0;                                        // rb % 3
    for (int rb = 0; rb < nblocks; ++rb, blk3 = (blk3 == 2 ? 0 : blk3 + 1)) {
        const int r0 = rb * kDfRB, r1 = min(A.V, r0 + kDfRB), nrows = r1 - r0;
        const int s0 = blk3 * kDfRB;                     // ring slot of the block's first row
        const int sprev = (blk3 == 0 ? 2 : blk3 - 1) * kDfRB;   // ... of the previous block's first row
        const int ring_lo = max(0, r0 - kDfRB);          // oldest row a row of this block reads from the ring
        auto ring_slot = [&](int p) -> int { return (p >= r0 ? s0 : sprev) + (p & (kDfRB - 1)); };

        if (wave == 0) {
            // ================================================================ solver
            if (!lds_wait_gt(&fetched, rb, P.status, lane)) return;      // (abandoned launch: fbuf was never staged)
            int poff_l, np_l, base_l, xin_l, flag_l, edge_l, pr_l[4], ps_l[4];
            {
                const int (*fb)[kDfRB] = fbuf[rb & 1];
                const int l = lane & (kDfRB - 1);
                poff_l = fb[FB_POFF][l]; np_l = fb[FB_NP][l]; base_l = fb[FB_BASE][l];
                pr_l[0] = fb[FB_PR0][l]; pr_l[1] = fb[FB_PR1][l]; pr_l[2] = fb[FB_PR2][l]; pr_l[3] = fb[FB_PR3][l];
                ps_l[0] = fb[FB_PS0][l]; ps_l[1] = fb[FB_PS1][l]; ps_l[2] = fb[FB_PS2][l]; ps_l[3] = fb[FB_PS3][l];
                xin_l = fb[FB_XIN][l]; flag_l = fb[FB_FLAG][l]; edge_l = fb[FB_EDGE][l];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (lane == 0) __hip_atomic_store(&taken, rb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // queued behind the reads
            // rows whose only predecessor is the row before them (the first row of a block: the row the previous block ended with)
            const unsigned long long bb_mask = __ballot(lane < nrows && np_l == 1 && pr_l[0] == r0 + lane - 1 && (lane > 0 || prev_r == r0 - 1));
            // rows all of whose (at most four) predecessors have their records in the ring
            const unsigned long long near_mask = __ballot(lane < nrows && np_l >= 1 && np_l <= 4 && pr_l[0] >= ring_lo &&
                                                          (np_l < 2 || pr_l[1] >= ring_lo) && (np_l < 3 || pr_l[2] >= ring_lo) && (np_l < 4 || pr_l[3] >= ring_lo));
            // blocks rb-3 and rb-2 are complete (all four waves, stores drained): the ring third of block rb-3 may be overwritten
            // (nobody reads it any more), and every row older than the ring can be read back from the H / Eagg rows
            if (rb >= 2 && !lds_wait_gt(&blk_pub, rb - 2, P.status, lane)) return;
            // per-row side data of the block's records, one row per lane
            if (lane < kDfRB) {
                const int sl = s0 + lane;
                side[S_EDGE][sl] = edge_l; side[S_FLAG][sl] = flag_l; side[S_NP][sl] = np_l; side[S_POFF][sl] = poff_l;
                side[S_PR0][sl] = pr_l[0]; side[S_PR1][sl] = pr_l[1]; side[S_PR2][sl] = pr_l[2]; side[S_PR3][sl] = pr_l[3];
                side[S_PS0][sl] = ps_l[0]; side[S_PS1][sl] = ps_l[1]; side[S_PS2][sl] = ps_l[2]; side[S_PS3][sl] = ps_l[3];
            }
            int Mv = M, Xv = X;                               // kept in VGPRs: a select between two SGPRs costs two extra moves per row
            asm volatile("" : "+v"(Mv), "+v"(Xv));
            __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): the staged values are in registers (or the row loop would
                                                              // wait, at every row, for the LDS writes of the row before)
            for (int i = 0; i < nrows; ) {
                if ((bb_mask >> i) & 1ull) {
                    // ---------------- a run of backbone rows: H / Eagg of the row above are in registers, its column on the left
                    // of this lane comes by one DPP shift; tight loop, state updated in place, one branch per row
                    const int i_end = i + __builtin_ctzll(~(bb_mask >> i));
                    do {
                        const int bs = __builtin_amdgcn_readlane(base_l, i);
                        const int xin_row = __builtin_amdgcn_readlane(xin_l, i), my_edge = __builtin_amdgcn_readlane(edge_l, i);
                        const int sc0 = (sch[0] == bs) ? Mv : Xv, sc1 = (sch[1] == bs) ? Mv : Xv;
                        int a[kDfKC], be[kDfKC], hs[kDfKC], fl[kDfKC], opn_next[kDfKC];
                        const int hv0 = df_shr1(edge_prev, hs_prev[kDfKC - 1]);
                        be[0] = max(es_prev[0] + E, hs_prev[0] + OE);                 // c-t-simd.c:524-528
                        be[1] = max(es_prev[1] + E, hs_prev[1] + OE);
                        a[0] = max(hv0 + sc0, be[0]); a[1] = max(hs_prev[0] + sc1, be[1]);
                        df_scan_row(a, xin_row, cl, cl1, E, O, hs, fl, opn_next);
                        const int si = s0 + i;
                        rec[si][F_HS0][lane] = hs[0]; rec[si][F_HS1][lane] = hs[1]; rec[si][F_ES0][lane] = be[0]; rec[si][F_ES1][lane] = be[1];
                        rec[si][F_FL0][lane] = fl[0]; rec[si][F_FL1][lane] = fl[1];
                        // the counter follows the record: the LDS performs one wave's operations in issue order, so no wait is
                        // needed (a workgroup-scope release fence would drain the writes first: ~150 cycles on the chain of rows)
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __hip_atomic_store(&produced[lane], rows_base + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        hs_prev[0] = hs[0]; hs_prev[1] = hs[1]; es_prev[0] = be[0]; es_prev[1] = be[1]; edge_prev = my_edge;
                        ++i;
                    } while (i < i_end);
                    prev_r = r0 + i - 1;
                    continue;
                }
                // ---------------- any other row: the maxima over its predecessors (values only)
                const int r = r0 + i;
                const int bs = __builtin_amdgcn_readlane(base_l, i);
                const int xin_row = __builtin_amdgcn_readlane(xin_l, i), my_edge = __builtin_amdgcn_readlane(edge_l, i);
                const int sc0 = (sch[0] == bs) ? Mv : Xv, sc1 = (sch[1] == bs) ? Mv : Xv;
                const int np = __builtin_amdgcn_readlane(np_l, i);
                int bq0 = kNeg, bq1 = kNeg, be0 = kNeg, be1 = kNeg;
                if ((near_mask >> i) & 1ull) {
                    // every predecessor's record is in the ring: the reads of the first two go out together, one wait
                    PredVals va, vb;
                    int eda, edb = 0;
                    { const int sl = ring_slot(__builtin_amdgcn_readlane(pr_l[0], i));
                      va.h0 = rec[sl][F_HS0][lane]; va.h1 = rec[sl][F_HS1][lane]; va.e0 = rec[sl][F_ES0][lane]; va.e1 = rec[sl][F_ES1][lane]; eda = side[S_EDGE][sl]; }
                    if (np > 1) {
                        const int sl = ring_slot(__builtin_amdgcn_readlane(pr_l[1], i));
                        vb.h0 = rec[sl][F_HS0][lane]; vb.h1 = rec[sl][F_HS1][lane]; vb.e0 = rec[sl][F_ES0][lane]; vb.e1 = rec[sl][F_ES1][lane]; edb = side[S_EDGE][sl];
                    }
                    bq0 = df_shr1(eda, va.h1) + sc0; bq1 = va.h0 + sc1;
                    be0 = max(va.e0 + E, va.h0 + OE); be1 = max(va.e1 + E, va.h1 + OE);   // c-t-simd.c:524-528
                    if (np > 1) {
                        bq0 = max(bq0, df_shr1(edb, vb.h1) + sc0); bq1 = max(bq1, vb.h0 + sc1);
                        be0 = max(be0, max(vb.e0 + E, vb.h0 + OE)); be1 = max(be1, max(vb.e1 + E, vb.h1 + OE));
                        for (int j = 2; j < np; ++j) {
                            const int sl = ring_slot(j == 2 ? __builtin_amdgcn_readlane(pr_l[2], i) : __builtin_amdgcn_readlane(pr_l[3], i));
                            const int h0 = rec[sl][F_HS0][lane], h1 = rec[sl][F_HS1][lane], e0 = rec[sl][F_ES0][lane], e1 = rec[sl][F_ES1][lane];
                            bq0 = max(bq0, df_shr1(side[S_EDGE][sl], h1) + sc0); bq1 = max(bq1, h0 + sc1);
                            be0 = max(be0, max(e0 + E, h0 + OE)); be1 = max(be1, max(e1 + E, h1 + OE));
                        }
                    }
                } else {
                    // a source (virtual init predecessor, c-t-simd.c:174-188), a predecessor older than the ring, more than four
                    const bool is_src = (np == 0);
                    const int p0 = __builtin_amdgcn_readlane(poff_l, i);
                    const int npp = is_src ? 1 : np;
                    for (int j = 0; j < npp; ++j) {
                        int p = A.V, ps = 0;
                        if (!is_src) {
                            if (j == 0) { p = __builtin_amdgcn_readlane(pr_l[0], i); ps = __builtin_amdgcn_readlane(ps_l[0], i); }
                            else if (j == 1) { p = __builtin_amdgcn_readlane(pr_l[1], i); ps = __builtin_amdgcn_readlane(ps_l[1], i); }
                            else if (j == 2) { p = __builtin_amdgcn_readlane(pr_l[2], i); ps = __builtin_amdgcn_readlane(ps_l[2], i); }
                            else if (j == 3) { p = __builtin_amdgcn_readlane(pr_l[3], i); ps = __builtin_amdgcn_readlane(ps_l[3], i); }
                            else { p = __builtin_amdgcn_readfirstlane(A.pred[p0 + j]); ps = __builtin_amdgcn_readfirstlane(A.pred_slot[p0 + j]); }
                        }
                        PredVals v;
                        if (!is_src && p >= ring_lo) {
                            const int sl = ring_slot(p);
                            v.h0 = rec[sl][F_HS0][lane]; v.h1 = rec[sl][F_HS1][lane]; v.e0 = rec[sl][F_ES0][lane]; v.e1 = rec[sl][F_ES1][lane];
                            v.hl = df_shr1(side[S_EDGE][sl], v.h1);
                        } else {
                            v = pred_far(p, ps, is_src);
                        }
                        bq0 = max(bq0, v.hl + sc0); bq1 = max(bq1, v.h0 + sc1);
                        be0 = max(be0, max(v.e0 + E, v.h0 + OE));
                        be1 = max(be1, max(v.e1 + E, v.h1 + OE));
                    }
                }
                int a[kDfKC], hs[kDfKC], fl[kDfKC], opn_next[kDfKC];
                a[0] = max(bq0, be0); a[1] = max(bq1, be1);
                df_scan_row(a, xin_row, cl, cl1, E, O, hs, fl, opn_next);
                const int si = s0 + i;
                rec[si][F_HS0][lane] = hs[0]; rec[si][F_HS1][lane] = hs[1]; rec[si][F_ES0][lane] = be0; rec[si][F_ES1][lane] = be1;
                rec[si][F_FL0][lane] = fl[0]; rec[si][F_FL1][lane] = fl[1];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __hip_atomic_store(&produced[lane], rows_base + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                hs_prev[0] = hs[0]; hs_prev[1] = hs[1]; es_prev[0] = be0; es_prev[1] = be1;
                prev_r = r; edge_prev = my_edge;
                ++i;
            }
        } else {
            // ================================================================ emitters: rows i = wave-1, wave+2, ...
            // A row's record holds H, Eagg, F; which predecessor gave the diagonal / the vertical gap and whether that gap was an
            // open (the reference's codes: c-t-simd.c:485-506, 521-538, 564-565) is worked out here from the records of the row's
            // predecessors, off the chain of rows.  Written for instruction count (three emitters must keep up with a solver
            // that spends ~55 instructions per backbone row): wave-uniform row index (scalar branches, scalar address
            // arithmetic), the values of the lane on the left by DPP, branch-free selects for the codes.
            int slot_l = 0, base_l = 0;
            if (lane < nrows) { slot_l = A.slot[r0 + lane]; base_l = A.base[r0 + lane]; }
            __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0) here: or every row would wait for the stores of the row before
            int spread_l = 0;
            // the block's side data (the solver wrote them before the block's first record), one row per lane
            const int em = (wave >= 5) ? wave - 2 : wave - 1; // emitter number: rows em, em + kDfEmit, ...
            int seen = 0;
            const bool live = (em >= nrows) || lds_wait_gt_seen(&produced[lane], rows_base + em, seen, P.status, lane);
            if (!live) return;                                // (abandoned launch)
            const int sl_b = s0 + (lane & (kDfRB - 1));
            const int edge_b = side[S_EDGE][sl_b], flag_b = side[S_FLAG][sl_b], np_b = side[S_NP][sl_b], poff_b = side[S_POFF][sl_b];
            const int pr_b[4] = {side[S_PR0][sl_b], side[S_PR1][sl_b], side[S_PR2][sl_b], side[S_PR3][sl_b]};
            const int ps_b[4] = {side[S_PS0][sl_b], side[S_PS1][sl_b], side[S_PS2][sl_b], side[S_PS3][sl_b]};
            const unsigned long long near_mask = __ballot(lane < nrows && np_b >= 1 && np_b <= 4 && pr_b[0] >= ring_lo && (np_b < 2 || pr_b[1] >= ring_lo) &&
                                                          (np_b < 3 || pr_b[2] >= ring_lo) && (np_b < 4 || pr_b[3] >= ring_lo));
            for (int i = em; live && i < nrows; i += kDfEmit) {
                if (!lds_wait_gt_seen(&produced[lane], rows_base + i, seen, P.status, lane)) return;
                const int r = r0 + i, si = s0 + i;
                const int hs0 = rec[si][F_HS0][lane], hs1 = rec[si][F_HS1][lane], es0 = rec[si][F_ES0][lane], es1 = rec[si][F_ES1][lane];
                const int fl0 = rec[si][F_FL0][lane], fl1 = rec[si][F_FL1][lane];
                const int myslot = __builtin_amdgcn_readlane(slot_l, i);
                const int bs = __builtin_amdgcn_readlane(base_l, i);
                const int np = __builtin_amdgcn_readlane(np_b, i);
                const bool mt0 = (sch[0] == bs), mt1 = (sch[1] == bs);
                const int sc0 = mt0 ? M : X, sc1 = mt1 ? M : X;
                int bq0 = kNeg, bq1 = kNeg, be0 = kNeg, be1 = kNeg, jq0 = 0, jq1 = 0, je0 = 0, je1 = 0;
                int hmx0 = kNeg, hmx1 = kNeg, hmn0 = -kNeg, hmn1 = -kNeg;
                int dl[4] = {0, 0, 0, 0};                       // links: distance to predecessor j (0: not representable / not known)
                // one predecessor's share: lowest predecessor index wins ties (strict >), je bit 6 = "the vertical gap opened"
                auto take = [&](int j, int hl, int h0, int h1, int e0, int e1) {
                    const int qv0 = hl + sc0, qv1 = h0 + sc1;                                  // diagonal from H(p, c-1)
                    if (qv0 > bq0) { bq0 = qv0; jq0 = j; }
                    if (qv1 > bq1) { bq1 = qv1; jq1 = j; }
                    const int eo0 = h0 + OE, eo1 = h1 + OE;                                     // c-t-simd.c:524-528
                    const int ei0 = max(e0 + E, eo0), ei1 = max(e1 + E, eo1);
                    if (ei0 > be0) { be0 = ei0; je0 = j | ((ei0 == eo0) ? 64 : 0); }
                    if (ei1 > be1) { be1 = ei1; je1 = j | ((ei1 == eo1) ? 64 : 0); }
                    hmx0 = max(hmx0, h0); hmn0 = min(hmn0, h0); hmx1 = max(hmx1, h1); hmn1 = min(hmn1, h1);
                };
                if ((near_mask >> i) & 1ull) {
                    // every predecessor's record is in the ring: the reads of the first two go out together
                    PredVals va, vb;
                    int eda, edb = 0;
                    { const int pa = __builtin_amdgcn_readlane(pr_b[0], i);
                      const int sl = ring_slot(pa);
                      dl[0] = (r - pa <= 31) ? r - pa : 0;
                      va.h0 = rec[sl][F_HS0][lane]; va.h1 = rec[sl][F_HS1][lane]; va.e0 = rec[sl][F_ES0][lane]; va.e1 = rec[sl][F_ES1][lane]; eda = side[S_EDGE][sl]; }
                    if (np > 1) {
                        const int pb = __builtin_amdgcn_readlane(pr_b[1], i);
                        const int sl = ring_slot(pb);
                        dl[1] = (r - pb <= 31) ? r - pb : 0;
                        vb.h0 = rec[sl][F_HS0][lane]; vb.h1 = rec[sl][F_HS1][lane]; vb.e0 = rec[sl][F_ES0][lane]; vb.e1 = rec[sl][F_ES1][lane]; edb = side[S_EDGE][sl];
                    }
                    {   // predecessor 0 (no comparison needed)
                        const int eo0 = va.h0 + OE, eo1 = va.h1 + OE;
                        bq0 = df_shr1(eda, va.h1) + sc0; bq1 = va.h0 + sc1;
                        be0 = max(va.e0 + E, eo0); be1 = max(va.e1 + E, eo1);
                        je0 = (be0 == eo0) ? 64 : 0; je1 = (be1 == eo1) ? 64 : 0;
                        hmx0 = hmn0 = va.h0; hmx1 = hmn1 = va.h1;
                    }
                    if (np > 1) {
                        take(1, df_shr1(edb, vb.h1), vb.h0, vb.h1, vb.e0, vb.e1);
                        for (int j = 2; j < np; ++j) {
                            const int pj = (j == 2) ? __builtin_amdgcn_readlane(pr_b[2], i) : __builtin_amdgcn_readlane(pr_b[3], i);
                            const int sl = ring_slot(pj);
                            if (j == 2) dl[2] = (r - pj <= 31) ? r - pj : 0; else dl[3] = (r - pj <= 31) ? r - pj : 0;
                            const int h0 = rec[sl][F_HS0][lane], h1 = rec[sl][F_HS1][lane], e0 = rec[sl][F_ES0][lane], e1 = rec[sl][F_ES1][lane];
                            take(j, df_shr1(side[S_EDGE][sl], h1), h0, h1, e0, e1);
                        }
                    }
                } else {
                    const bool is_src = (np == 0);
                    const int p0 = __builtin_amdgcn_readlane(poff_b, i);
                    const int npp = is_src ? 1 : np;
                    for (int j = 0; j < npp; ++j) {
                        int p = A.V, ps = 0;
                        if (!is_src) {
                            if (j == 0) { p = __builtin_amdgcn_readlane(pr_b[0], i); ps = __builtin_amdgcn_readlane(ps_b[0], i); }
                            else if (j == 1) { p = __builtin_amdgcn_readlane(pr_b[1], i); ps = __builtin_amdgcn_readlane(ps_b[1], i); }
                            else if (j == 2) { p = __builtin_amdgcn_readlane(pr_b[2], i); ps = __builtin_amdgcn_readlane(ps_b[2], i); }
                            else if (j == 3) { p = __builtin_amdgcn_readlane(pr_b[3], i); ps = __builtin_amdgcn_readlane(ps_b[3], i); }
                            else { p = __builtin_amdgcn_readfirstlane(A.pred[p0 + j]); ps = __builtin_amdgcn_readfirstlane(A.pred_slot[p0 + j]); }
                        }
                        PredVals v;
                        if (!is_src && p >= ring_lo) {
                            const int sl = ring_slot(p);
                            v.h0 = rec[sl][F_HS0][lane]; v.h1 = rec[sl][F_HS1][lane]; v.e0 = rec[sl][F_ES0][lane]; v.e1 = rec[sl][F_ES1][lane];
                            v.hl = df_shr1(side[S_EDGE][sl], v.h1);
                        } else {
                            v = pred_far(p, ps, is_src);
                        }
                        take(j, v.hl, v.h0, v.h1, v.e0, v.e1);
                    }
                }
                if (np > 1) {                                 // largest distance between two predecessor rows of a node (tsta_engine.h: warnings)
                    if (c0 < n) spread_l = max(spread_l, hmx0 - hmn0);
                    if (c0 + 1 < n) spread_l = max(spread_l, hmx1 - hmn1);
                }
                const int hO0 = hs0 + O, hO1 = hs1 + O, hOE0 = hO0 + E, hOE1 = hO1 + E;
                // F-open flag of a column: H(c-1) + O >= F(c-1); the lane's first column looks at the lane on the left / the chunk edge
                const int opn1 = (hO0 >= fl0) ? 1 : 0;
                const int opn0 = df_shr1(__builtin_amdgcn_readlane(flag_b, i), (hO1 >= fl1) ? 1 : 0);
                // match > mismatch > E > F (c-t-simd.c:485-506)
                const int cd0 = (mt0 ? 42 : 84) + jq0, cd1 = (mt1 ? 42 : 84) + jq1;           // the three candidates first: selects, no branches
                const int ce0 = (es0 == hs0) ? (je0 & 63) : 126, ce1 = (es1 == hs1) ? (je1 & 63) : 126;
                const int code0 = (bq0 == hs0) ? cd0 : ce0, code1 = (bq1 == hs1) ? cd1 : ce1;
                const int ep0 = ((je0 & 64) ? 43 : 1) + (je0 & 63), ep1 = ((je1 & 64) ? 43 : 1) + (je1 & 63);       // c-t-simd.c:530-538
                const int ec0 = (es0 + E == hOE0) ? -ep0 : ep0, ec1 = (es1 + E == hOE1) ? -ep1 : ep1;
                const int fp0 = opn0 + 1, fp1 = opn1 + 1;                                   // c-t-simd.c:521-522,564-565
                int fc0 = (fl0 + E == hOE0) ? -fp0 : fp0, fc1 = (fl1 + E == hOE1) ? -fp1 : fp1;
                if (A.links) {
                    // the cell's own move goes to predecessor jq (diagonal) or je (vertical gap): its distance rides on the F code
                    const int s0i = (bq0 == hs0) ? jq0 : (je0 & 63), s1i = (bq1 == hs1) ? jq1 : (je1 & 63);
                    int d0 = dl[0], d1 = dl[0];
                    if (np > 1) {
                        d0 = (s0i == 1) ? dl[1] : d0; d1 = (s1i == 1) ? dl[1] : d1;
                        if (np > 2) {
                            d0 = (s0i == 2) ? dl[2] : d0; d1 = (s1i == 2) ? dl[2] : d1;
                            d0 = (s0i == 3) ? dl[3] : d0; d1 = (s1i == 3) ? dl[3] : d1;
                            d0 = (s0i > 3) ? 0 : d0; d1 = (s1i > 3) ? 0 : d1;
                        }
                    }
                    fc0 = (fc0 & 7) | (d0 << 3); fc1 = (fc1 & 7) | (d1 << 3);
                }
                const unsigned w1 = (unsigned)(code0 & 0xff) | ((unsigned)(code1 & 0xff) << 8);
                const unsigned w2 = (unsigned)(ec0 & 0xff) | ((unsigned)(ec1 & 0xff) << 8);
                const unsigned w3 = (unsigned)(fc0 & 0xff) | ((unsigned)(fc1 & 0xff) << 8);
                // (row bases are wave-uniform: scalar address arithmetic, the lane only adds its index)
                int* Hr = A.Hrow + (size_t)myslot * A.pitch;
                int* Er = A.Erow + (size_t)myslot * A.pitch;
                if (c0 < n) {
                    store_sc1_int2(reinterpret_cast<int2*>(Hr + cbase + 4) + lane, hs0, hs1);     // H(r, c0+1..c0+2): read by later blocks of this launch
                    store_sc1_int2(reinterpret_cast<int2*>(Er + cbase) + lane, es0, es1);
                    if (A.source) {                                                           // trace: read after the launch only
                        const size_t toff = (size_t)r * A.tpitch + cbase;
                        reinterpret_cast<unsigned short*>(A.source + toff)[lane] = (unsigned short)w1;
                        reinterpret_cast<unsigned short*>(A.esource + toff)[lane] = (unsigned short)w2;
                        reinterpret_cast<unsigned short*>(A.fsource + toff)[lane] = (unsigned short)w3;
                    }
                }
                if (q == A.nchunks - 1 && c0 < n && c0 + kDfKC >= n) A.lastscore[r] = (n - 1 - c0 == 1) ? hs1 : hs0;   // owner of the last column
                if (q == 0 && lane == 0) store_sc1_int(&Hr[3], __builtin_amdgcn_readlane(edge_b, i));      // H(r, 0)
                if (lane == 63) {                             // what the chunk on the right needs of this row (c-t-simd.c: v0 / f0 carries)
                    int* hd = P.hand + ((size_t)r * ncar + q + 1) * 4;
                    store_sc1_int2(reinterpret_cast<int2*>(hd), max(fl1, hO1), (hO1 >= fl1) ? 1 : 0);   // Fbar leaving the chunk; its first F opens from H
                    store_sc1_int2(reinterpret_cast<int2*>(hd + 2), hs1, 0);                            // H at the chunk's right edge
                }
            }
            if (__ballot(spread_l > 0)) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) spread_l = max(spread_l, __shfl_xor(spread_l, off, 64));
                if (lane == 0) atomicMax(A.best + 8, spread_l);
            }
        }
        // No barrier: the solver runs on into the next block while the emitters finish this one.  Every wave, once ITS stores
        // of the block have left, counts itself in; the last one publishes the block — to the chunk on the right (prog[q+1],
        // sc1) and to this workgroup's solver (blk_pub: it may then recycle the block's ring third two blocks later).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const int before = __hip_atomic_fetch_add(&blk_done[rb & 3], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (before == kDfEmit) {
                __hip_atomic_store(&blk_done[rb & 3], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!mute) __hip_atomic_store((GlobalI32*)(P.prog + q + 1), rb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&blk_pub, rb + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        rows_base += nrows;
    }
}

// Launches the fill (A complete, inputs uploaded, virtual init row written).  `work` = device ints for this call:
// [1] status, [16 .. 16+nchunks] blocks of 32 nodes completed by the column-0 solver and by each chunk; hand =
// [V][nchunks+1][4] ints.
int launch_poa_dataflow(Engine* e, const PoaArgs& A, int* d_work, size_t work_ints, int* d_hand, hipStream_t s)
{
    const size_t head = 16 + (size_t)A.nchunks + 1;
    if (head > work_ints) return TSTA_ERR_INTERNAL;
    int rc;
    if ((rc = e->check(hipMemsetAsync(d_work, 0, head * sizeof(int), s), "hipMemsetAsync(poa dataflow)"))) return rc;
    PoaDfArgs P;
    P.A = A;
    P.ticket = reinterpret_cast<unsigned*>(d_work); P.status = reinterpret_cast<unsigned*>(d_work + 1); P.prog = d_work + 16;
    P.hand = d_hand;
    P.test_abandon = e->opt_test_abandon;
    hipLaunchKernelGGL(poa_dataflow_kernel, dim3(A.nchunks + 1), dim3(kDfThreads), 0, s, P);
    return e->check(hipGetLastError(), "poa dataflow launch");
}

}  // namespace tsta
