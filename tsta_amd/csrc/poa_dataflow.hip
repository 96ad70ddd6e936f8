// poa_dataflow.hip — K4d: the POA fill of poa_fill.hip as ONE launch, with the per-row work of a tile split over the four
// waves of a workgroup.  Same results (reference: msa/c-t-simd.c:165-599, 602-657, 979-1045; see poa_fill.hip).
//
// Why.  The README example is a latency problem: 5 000 columns are only 40 tiles wide, so a fill is as long as
// the chain of its rows, V x (time per row), and K4 spends ~0.9 us per row: one wave does everything a row needs (~460
// instructions at the single-wave issue rate) and every tile anti-diagonal is a kernel launch.  Two things change here:
//
//  1. Dataflow instead of launches.  The tiles (32 consecutive nodes x 128 columns) are tickets, enumerated in the order the
//     launch pipeline would run them (anti-diagonal d = chunk q + row-block rb, then q).  A workgroup takes a ticket, waits
//     until its two producers have published — the tile above (prog[q] >= rb) and the tile on the left (prog[q-1] >= rb+1);
//     both hold earlier tickets, so no deadlock whatever the number of resident workgroups — and publishes prog[q] = rb + 1.
//     What crosses tiles goes through sc1 stores / loads (handoff.h): the H / Eagg rows of the nodes, the per-row F carry,
//     open flag and H at the chunk's right edge (edge[]: kept per (row, chunk), so a recycled row slot is only ever read
//     and rewritten inside ONE chunk's chain of tiles).
//  2. Solver + emitters.  Wave 0 of the workgroup (the solver) runs only what the NEXT row needs: predecessor maxima, the
//     max-plus scan of F, final H / Eagg / F.  It leaves 7 dwords per lane and row in an LDS record; waves 1-3 (emitters, rows
//     i = e mod 3) turn records into the reference's three trace codes (c-t-simd.c:485-506, 521-538, 564-565) and do all the
//     stores.  The solver's chain is ~100 instructions per row instead of ~460.
//
// Scope: 2 columns per lane (sequences up to 16 384) and sequences without a literal 'N' (the N-column rule needs the
// segment-by-segment scan of poa_fill.hip); everything else runs the launch-per-diagonal K4.
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
constexpr int kDfCW = 64 * kDfKC;        // columns per tile

// what the solver leaves for the emitters, per lane and row: 7 dwords, kept field-major in LDS (rec[field][row][lane]:
// every access is 64 consecutive dwords, conflict-free; 56 KB for 32 rows)
struct DfRec {
    int hs0, hs1;                        // H of the lane's two columns
    int es0, es1;                        // Eagg
    int fl0, fl1;                        // F
    int meta;                            // per column (16 bits each): jq | je(7 bits, bit 6 = "opened") << 6 | (bq==H) << 13 | (be==H) << 14 | open-flag << 15
};
enum { F_HS0, F_HS1, F_ES0, F_ES1, F_FL0, F_FL1, F_META, F_COUNT };

__device__ __forceinline__ int df_shr1(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false); }

// bounded wait on an LDS word (solver -> emitters of one workgroup); acquires what was written before it
__device__ __forceinline__ bool lds_wait_gt(int* word, int v, unsigned* status, int lane)
{
    int spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= v) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 18)) { if (lane == 0) atomicOr(status, 2u); return false; }      // ~30 ms
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return true;
}

__global__ __launch_bounds__(256) void poa_dataflow_kernel(PoaDfArgs P)
{
    const PoaArgs& A = P.A;
    __shared__ int rec[F_COUNT][kRB][64];                // 56 KB
    __shared__ int row_edge[kRB];                        // H(row, cbase): the column left of the tile
    __shared__ int row_h0[kRB];                          // chunk 0: H(row, 0)
    __shared__ int produced;                             // rows of this tile the solver has finished
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = A.n, M = A.M, X = A.X, E = A.E, O = A.O, OE = A.O + A.E;
    const int aK = kDfKC * E;
    const int ncar = A.nchunks + 1;
    const int w15 = ((lane & 15) + 1) * aK, w31 = (lane - 31) * aK;

    // Tiles are dealt statically: workgroup b runs tiles b, b + G, b + 2G, ... of the d-major order (G = gridDim.x, all
    // resident).  The lowest unfinished tile always has finished producers and a workgroup that has nothing lower left to
    // do, so the waits below always end.  (No ticket atomics and no barrier at the top of a tile.)
    if (threadIdx.x == 0) produced = 0;
    __syncthreads();
    int rows_base = 0;                                   // rows this workgroup's solver has produced in earlier tiles
    for (int t = blockIdx.x; t < P.n_tiles; t += gridDim.x) {
        int lo = 0, hi = P.n_d - 1;                      // d = last anti-diagonal whose first tile is <= t
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (__builtin_amdgcn_readfirstlane(P.tile_first[mid]) <= t) lo = mid; else hi = mid - 1; }
        const int d = lo, q = __builtin_amdgcn_readfirstlane(P.tile_qlo[d] + (t - P.tile_first[d])), rb = d - q;
        const int r0 = rb * kRB, r1 = min(A.V, r0 + kRB), nrows = r1 - r0;
        const int cbase = q * kDfCW, c0 = cbase + lane * kDfKC;

        int sch[kDfKC];
#pragma unroll
        for (int k = 0; k < kDfKC; ++k) sch[k] = (c0 + k < n) ? (int)A.seq[c0 + k] : -1;
        int slot_l = 0, base_l = 0;
        if (lane < nrows) { slot_l = A.slot[r0 + lane]; base_l = A.base[r0 + lane]; }

        if (wave == 0) {
            // ================================================================ solver
            if (rb > 0) wait_progress(P.prog + q, rb, P.status, lane);                  // the tile above (and, by induction, all above it)
            if (q > 0)  wait_progress(P.prog + q - 1, rb + 1, P.status, lane);          // the tile on the left
            int xin_l = 0, flag_l = 1, edge_l = 0;
            if (q > 0 && lane < nrows) {
                xin_l = load_sc1_int(&A.carry[(size_t)(r0 + lane) * ncar + q]);
                flag_l = load_sc1_int(&P.cflag32[(size_t)(r0 + lane) * ncar + q]);
                edge_l = load_sc1_int(&P.edge[(size_t)(r0 + lane) * ncar + q]);
            }
            int poff_l = 0, np_l = 0, pfirst_l = -1, pfslot_l = 0;
            if (lane < nrows) {
                poff_l = A.pred_off[r0 + lane];
                np_l = A.pred_off[r0 + lane + 1] - poff_l;
                if (np_l > 0) { pfirst_l = A.pred[poff_l]; pfslot_l = A.pred_slot[poff_l]; }
            }
            int cout_l = 0, fout_l = 0, eout_l = 0, spread_l = 0;
            int hs_prev[kDfKC] = {0, 0}, es_prev[kDfKC] = {0, 0}, prev_r = -2, h0_prev = 0;

            for (int r = r0; r < r1; ++r) {
                const int i = r - r0;
                const int p0 = __builtin_amdgcn_readlane(poff_l, i), np = __builtin_amdgcn_readlane(np_l, i);
                const int pfirst = __builtin_amdgcn_readlane(pfirst_l, i), pfslot = __builtin_amdgcn_readlane(pfslot_l, i);
                const bool is_src = (np == 0);               // in==0 -> virtual init predecessor (c-t-simd.c:174-188)
                const int bs = __builtin_amdgcn_readlane(base_l, i);
                int h0 = 0, xin_row, flag_row;
                if (q == 0) {                                // first column (c-t-simd.c:187,194-200)
                    if (is_src) h0 = OE;
                    else {
                        h0 = kNeg;
                        for (int j = 0; j < np; ++j) {
                            const int p = (j == 0) ? pfirst : A.pred[p0 + j];
                            const int ps = (j == 0) ? pfslot : A.pred_slot[p0 + j];
                            int hp;
                            if (p == prev_r) hp = h0_prev;
                            else if (p >= r0) hp = row_h0[p - r0];
                            else hp = load_sc1_int(&A.Hrow[(size_t)ps * A.pitch + 3]);
                            h0 = max(h0, hp);
                        }
                        h0 += E;
                    }
                    xin_row = h0 + O; flag_row = 1;          // F(r,0) = -inf: the first F is an open
                } else {
                    xin_row = __builtin_amdgcn_readlane(xin_l, i);
                    flag_row = __builtin_amdgcn_readlane(flag_l, i);
                }
                const int my_edge = (q == 0) ? h0 : __builtin_amdgcn_readlane(edge_l, i);     // H(r, cbase)

                int bq[kDfKC], jq[kDfKC], be[kDfKC], je[kDfKC], sc[kDfKC];
#pragma unroll
                for (int k = 0; k < kDfKC; ++k) { sc[k] = (sch[k] == bs) ? M : X; bq[k] = kNeg; jq[k] = 0; be[k] = kNeg; je[k] = 0; }
                const int npp = is_src ? 1 : np;
                int hmx[kDfKC] = {kNeg, kNeg}, hmn[kDfKC] = {-kNeg, -kNeg};
                for (int j = 0; j < npp; ++j) {
                    const int p = is_src ? A.V : ((j == 0) ? pfirst : A.pred[p0 + j]);
                    const int ps = is_src ? 0 : ((j == 0) ? pfslot : A.pred_slot[p0 + j]);
                    int hv[kDfKC + 1], ev[kDfKC];
                    if (p == prev_r) {
                        // the row just finished: registers + one lane shift
                        const int edge = (q == 0) ? h0_prev : __builtin_amdgcn_readlane(edge_l, i - 1);
                        hv[0] = df_shr1(edge, hs_prev[kDfKC - 1]);
                        hv[1] = hs_prev[0]; hv[2] = hs_prev[1]; ev[0] = es_prev[0]; ev[1] = es_prev[1];
                    } else if (p >= r0 && p < r1) {
                        // a row finished earlier in this tile: its record
                        const int ip = p - r0;
                        hv[1] = rec[F_HS0][ip][lane]; hv[2] = rec[F_HS1][ip][lane]; ev[0] = rec[F_ES0][ip][lane]; ev[1] = rec[F_ES1][ip][lane];
                        hv[0] = (lane > 0) ? rec[F_HS1][ip][lane - 1] : row_edge[ip];
                    } else {
                        // a row of an earlier row-block (published before this tile started) or the virtual init row
                        const int* Hp = A.Hrow + (size_t)ps * A.pitch + c0 + 3;    // H(p, c0 .. c0+2)
                        const int* Ep = A.Erow + (size_t)ps * A.pitch + c0;        // Eagg(p, c0+1 .. c0+2)
                        // the column left of the chunk belongs to the chunk on the left: its per-(row, chunk) copy
                        hv[0] = (lane == 0 && q > 0 && !is_src) ? load_sc1_int(&P.edge[(size_t)p * ncar + q]) : load_sc1_int(Hp);
                        const int2 ha = load_sc1_int2(reinterpret_cast<const int2*>(Hp + 1));
                        const int2 ea = load_sc1_int2(reinterpret_cast<const int2*>(Ep));
                        hv[1] = ha.x; hv[2] = ha.y; ev[0] = ea.x; ev[1] = ea.y;
                    }
#pragma unroll
                    for (int k = 0; k < kDfKC; ++k) {
                        const int qv = hv[k] + sc[k];                              // diagonal from H(p, c-1)
                        if (qv > bq[k]) { bq[k] = qv; jq[k] = j; }                 // lowest predecessor index on ties
                        const int eo = hv[k + 1] + OE, ex = ev[k] + E;             // c-t-simd.c:524-528
                        const int ein = max(ex, eo);
                        if (ein > be[k]) { be[k] = ein; je[k] = j | ((ein == eo) ? 64 : 0); }
                        if (npp > 1) { hmx[k] = max(hmx[k], hv[k + 1]); hmn[k] = min(hmn[k], hv[k + 1]); }
                    }
                }
                if (npp > 1) {
#pragma unroll
                    for (int k = 0; k < kDfKC; ++k) if (c0 + k < n) spread_l = max(spread_l, hmx[k] - hmn[k]);
                }
                // ---- F: local pass with no carry-in, max-plus inclusive scan over the 64 lanes, fix-up pass (as poa_fill.hip)
                int fbar = kNeg;
#pragma unroll
                for (int k = 0; k < kDfKC; ++k) {
                    const int f = fbar + E;
                    const int h = max(max(bq[k], be[k]), f);
                    fbar = max(f, h + O);
                }
                int v = fbar;
                v = max(v, __builtin_amdgcn_update_dpp(kNeg, v, 0x111, 0xf, 0xf, false) + aK);
                v = max(v, __builtin_amdgcn_update_dpp(kNeg, v, 0x112, 0xf, 0xf, false) + 2 * aK);
                v = max(v, __builtin_amdgcn_update_dpp(kNeg, v, 0x114, 0xf, 0xf, false) + 4 * aK);
                v = max(v, __builtin_amdgcn_update_dpp(kNeg, v, 0x118, 0xf, 0xf, false) + 8 * aK);
                v = max(v, __builtin_amdgcn_update_dpp(kNeg, v, 0x142, 0xa, 0xf, false) + w15);
                v = max(v, __builtin_amdgcn_update_dpp(kNeg, v, 0x143, 0xc, 0xf, false) + w31);
                const int prev = df_shr1(kNeg, v);
                const int xin = max(prev, xin_row + lane * aK);
                const int xout = max(__builtin_amdgcn_readlane(v, 63), xin_row + 64 * aK);
                fbar = xin;
                int hs[kDfKC], fl[kDfKC], opn_next[kDfKC];
#pragma unroll
                for (int k = 0; k < kDfKC; ++k) {
                    const int f = fbar + E;
                    const int h = max(max(bq[k], be[k]), f);
                    hs[k] = h; fl[k] = f;
                    const int g = h + O;
                    opn_next[k] = (g >= f) ? 1 : 0;          // F(c+1) == H(c)+O+E  <=>  H(c)+O >= F(c)
                    fbar = max(f, g);
                }
                const int opn0 = df_shr1(flag_row, opn_next[kDfKC - 1]);   // open flag of the lane's first column: from the left
                const int flag_out = __builtin_amdgcn_readlane(opn_next[kDfKC - 1], 63);
                const int edge_out = __builtin_amdgcn_readlane(hs[kDfKC - 1], 63);
                if (lane == i) { cout_l = xout; fout_l = flag_out; eout_l = edge_out; }
                // ---- the record for the emitters (and for later rows of this tile that have this row as a predecessor)
                {
                    const unsigned m0 = (unsigned)jq[0] | ((unsigned)je[0] << 6) | ((bq[0] == hs[0]) ? (1u << 13) : 0u) | ((be[0] == hs[0]) ? (1u << 14) : 0u) | ((unsigned)opn0 << 15);
                    const unsigned m1 = (unsigned)jq[1] | ((unsigned)je[1] << 6) | ((bq[1] == hs[1]) ? (1u << 13) : 0u) | ((be[1] == hs[1]) ? (1u << 14) : 0u) | ((unsigned)opn_next[0] << 15);
                    rec[F_HS0][i][lane] = hs[0]; rec[F_HS1][i][lane] = hs[1]; rec[F_ES0][i][lane] = be[0]; rec[F_ES1][i][lane] = be[1];
                    rec[F_FL0][i][lane] = fl[0]; rec[F_FL1][i][lane] = fl[1]; rec[F_META][i][lane] = (int)(m0 | (m1 << 16));
                    if (lane == 0) { row_edge[i] = my_edge; row_h0[i] = h0; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the record is in LDS before the counter says so
                    if (lane == 0) __hip_atomic_store(&produced, rows_base + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                hs_prev[0] = hs[0]; hs_prev[1] = hs[1]; es_prev[0] = be[0]; es_prev[1] = be[1];
                prev_r = r; h0_prev = h0;
            }
            if (lane < nrows) {                               // what the tile on the right needs of these rows
                store_sc1_int(&A.carry[(size_t)(r0 + lane) * ncar + q + 1], cout_l);
                store_sc1_int(&P.cflag32[(size_t)(r0 + lane) * ncar + q + 1], fout_l);
                store_sc1_int(&P.edge[(size_t)(r0 + lane) * ncar + q + 1], eout_l);
            }
            if (__ballot(spread_l > 0)) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) spread_l = max(spread_l, __shfl_xor(spread_l, off, 64));
                if (lane == 0) atomicMax(A.best + 8, spread_l);
            }
        } else {
            // ================================================================ emitters: rows i = wave-1, wave+2, ...
            for (int i = wave - 1; i < nrows; i += 3) {
                if (!lds_wait_gt(&produced, rows_base + i, P.status, lane)) break;
                const int r = r0 + i;
                DfRec R;
                R.hs0 = rec[F_HS0][i][lane]; R.hs1 = rec[F_HS1][i][lane]; R.es0 = rec[F_ES0][i][lane]; R.es1 = rec[F_ES1][i][lane];
                R.fl0 = rec[F_FL0][i][lane]; R.fl1 = rec[F_FL1][i][lane]; R.meta = rec[F_META][i][lane];
                const int myslot = __builtin_amdgcn_readlane(slot_l, i);
                const int bs = __builtin_amdgcn_readlane(base_l, i);
                int* Hr = A.Hrow + (size_t)myslot * A.pitch;
                int* Er = A.Erow + (size_t)myslot * A.pitch;
                if (c0 < n) {
                    unsigned w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
                    for (int k = 0; k < kDfKC; ++k) {
                        const int h = k ? R.hs1 : R.hs0, f = k ? R.fl1 : R.fl0, be = k ? R.es1 : R.es0;
                        const int m = (int)(((unsigned)R.meta >> (16 * k)) & 0xffffu);
                        const int jq = m & 63, je = (m >> 6) & 127;
                        const bool mt = (sch[k] == bs);
                        int code;                            // match > mismatch > E > F (c-t-simd.c:485-506)
                        if (m & (1 << 13)) code = (mt ? 42 : 84) + jq;
                        else if (m & (1 << 14)) code = je & 63;
                        else code = 126;
                        int ecode = ((je & 64) ? 43 : 1) + (je & 63);              // c-t-simd.c:530-538
                        if (be + E == h + OE) ecode = -ecode;
                        int fcode = (m & (1 << 15)) ? 2 : 1;                       // c-t-simd.c:521-522,564-565
                        if (f + E == h + OE) fcode = -fcode;
                        w1 |= (unsigned)(uint8_t)code << (8 * k);
                        w2 |= (unsigned)(uint8_t)ecode << (8 * k);
                        w3 |= (unsigned)(uint8_t)fcode << (8 * k);
                    }
                    store_sc1_int2(reinterpret_cast<int2*>(Hr + c0 + 4), R.hs0, R.hs1);          // H(r, c0+1..c0+2): read by later tiles of this launch
                    store_sc1_int2(reinterpret_cast<int2*>(Er + c0), R.es0, R.es1);
                    if (A.source) {                                                           // trace: read after the launch only
                        *reinterpret_cast<unsigned short*>(A.source + (size_t)r * A.tpitch + c0) = (unsigned short)w1;
                        *reinterpret_cast<unsigned short*>(A.esource + (size_t)r * A.tpitch + c0) = (unsigned short)w2;
                        *reinterpret_cast<unsigned short*>(A.fsource + (size_t)r * A.tpitch + c0) = (unsigned short)w3;
                    }
                    if (c0 + kDfKC >= n) A.lastscore[r] = (n - 1 - c0 == 1) ? R.hs1 : R.hs0;     // owner of the last column
                }
                if (q == 0 && lane == 0) store_sc1_int(&Hr[3], row_h0[i]);                       // H(r, 0)
            }
        }
        // every wave's stores have left before the tile is published; the emitters are done with the records before the
        // solver overwrites them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store((GlobalI32*)(P.prog + q), rb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rows_base += nrows;
    }
}

// Launches the dataflow fill (A complete, inputs uploaded, virtual init row written).  `work` = device ints for this call:
// [0] ticket, [1] status, [16 .. 16+nchunks) progress per chunk, then the tile table; edge / cflag32 = [V][nchunks+1] ints.
int launch_poa_dataflow(Engine* e, const PoaArgs& A, int* d_work, size_t work_ints, int* d_edge, int* d_cflag32, hipStream_t s)
{
    const int D = A.nchunks + A.nrb - 1;
    std::vector<int> tab((size_t)2 * D + 1);
    int64_t total = 0;
    for (int d = 0; d < D; ++d) {
        const int q_lo = std::max(0, d - A.nrb + 1), q_hi = std::min(A.nchunks - 1, d);
        tab[(size_t)d] = (int)total; tab[(size_t)(D + 1 + d)] = q_lo;
        total += q_hi - q_lo + 1;
    }
    tab[(size_t)D] = (int)total;
    const size_t head = 16 + (size_t)A.nchunks;
    if (total > 0x7fffffff || head + tab.size() > work_ints) return TSTA_ERR_INTERNAL;
    int rc;
    if ((rc = e->check(hipMemsetAsync(d_work, 0, head * sizeof(int), s), "hipMemsetAsync(poa dataflow)")) ||
        (rc = e->check(hipMemcpyAsync(d_work + head, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, s), "H2D tile table")) ||
        (rc = e->check(hipStreamSynchronize(s), "poa dataflow upload")))         // `tab` lives on this stack frame
        return rc;
    PoaDfArgs P;
    P.A = A;
    P.ticket = reinterpret_cast<unsigned*>(d_work); P.status = reinterpret_cast<unsigned*>(d_work + 1); P.prog = d_work + 16;
    P.tile_first = d_work + head; P.tile_qlo = d_work + head + D + 1; P.n_d = D; P.n_tiles = (int)total;
    P.edge = d_edge; P.cflag32 = d_cflag32;
    const int blocks = (int)std::min<int64_t>(total, e->n_cu);               // one workgroup per CU
    hipLaunchKernelGGL(poa_dataflow_kernel, dim3(blocks), dim3(256), 0, s, P);
    return e->check(hipGetLastError(), "poa dataflow launch");
}

}  // namespace tsta
