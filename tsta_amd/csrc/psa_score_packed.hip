// psa_score_packed.hip — K1p: batched score-only fill, two pairs per wave in packed 16-bit lanes.
//
// Same path and same results as K1 (psa_score.hip; reference: psa/psa.c:219-232, 234-433, 593-639) but about
// half the VALU issue per cell.  Measured on gfx950 (tools/valu_microbench.hip): v_add/v_sub/v_xor issue at
// ~1.1 cycles per wave64, every max/min/cndmask/v_pk_* at ~2.1 — so the cell is rebuilt from
//   * plain 32-bit v_sub_u32 / v_add_u32 / v_xor_b32 acting on BOTH 16-bit halves at once (no borrow/carry
//     can cross the halves because scores are kept biased and non-negative), and
//   * v_pk_max_u16 / v_pk_min_u16 for the five max/min,
// i.e. 6 fast + 6 packed ops per TWO cells.  The two halves of every register belong to two different pairs
// (a "couple") of similar size; their lanes, skew, band hand-off and DPP traffic are shared.  Pairs of different
// size are padded to the larger one (cells outside the smaller pair are masked out of its running max).
//
// Scores are stored as  v = H - base + 0x8000  with an int32 `base` per pair that is re-anchored to the band's
// top row every 64 steps; all values live within (64R+192)*U of the base (U = per-step score bound), which the
// host checks against the 16-bit range before choosing this kernel (otherwise K1's int32 path runs).
// The match/mismatch select is  t = min_u16((a^b) << 8, M-X)  ->  D = (Hdiag + M) - t.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "wave_ops.h"
#include "packed_ops.h"
#include "packed_step.h"

namespace tsta {

// The two pairs of a couple may differ in size: the wave sweeps max(nA,nB) columns and max(mA,mB) rows; cells
// outside the smaller pair compute garbage that only feeds other such cells and is masked out of its running max.
// F3 (round 4): the kernel keeps its scores around the bias 0x4000 instead of 0x8000, so that they are bit patterns of positive
// normal binary16 numbers, and a steady 64-step block whose operands provably stay inside [0x0400, 0x7bff] takes the three-way
// maximum of every cell as ONE v_pk_maximum3_f16 (7 instead of 8 instructions per cell pair).  The proof is a look at the block's
// entering values (block_range) plus what 64 steps can move them by (`mdn` down, `mup` up): a block that fails it runs the
// two-instruction u16 maximum as before (any value of the 16-bit window is fine for that), and a block that would leave the 16-bit
// window altogether makes the band return false — the couple is then scored by the int32 kernel (PsaBatchArgs::redo).
// F3 == false is the round-3 kernel unchanged (bias 0x8000, window guaranteed by the host's bound).
template <int R, bool PARTIAL, bool PERM, bool FF, bool F3>
__device__ __forceinline__ bool pband(const uint8_t* __restrict__ aA, const uint8_t* __restrict__ aB, const int nA, const int nB,
                                      const uint8_t* __restrict__ bA, const uint8_t* __restrict__ bB, const int mA, const int mB,
                                      const bool store_bottom,
                                      const int band_row0, int2* __restrict__ rowA, int2* __restrict__ rowB,
                                      PairAcc& acc, const int lane, const int M, const int E, const int O,
                                      const unsigned pe_s, const unsigned po_s, const unsigned pm_s, const unsigned pmx_s,
                                      const bool elide_ok, const int f3_mode)
{
    constexpr int kBias = F3 ? 0x4000 : tsta::kBias;      // (shadows the namespace constant inside this band)
    const unsigned pe = in_vgpr(pe_s), po = in_vgpr(po_s), pm = in_vgpr(pm_s), pmx = in_vgpr(pmx_s);
    const unsigned pf = pe + pm;                   // |E| + M: the horizontal gap pays the column shift of the H' frame
    const unsigned pf2 = pf + pe;                  // 2|E| + M: the same in the frame of the elided blocks
    PLane<R> s;
    const int n = max(nA, nB), nmin = min(nA, nB);
    const int i0 = band_row0 + lane * R;
    const int nvA = mA - i0, nvB = mB - i0;
    // The band hand-off row is kept as the kernel holds it: per column one dword H' and one dword Ebar', both pairs' biased
    // 16-bit halves side by side (rowP), plus per 64-column block the two int32 bases those halves are relative to (rowS):
    // 8 B per column and couple instead of the two int2 {H, Ebar} of 16 B (the row is this kernel's only HBM traffic).
    // An entry read back differs from the reader's frame only by (writer's base - reader's base) per half.
    uint2* __restrict__ rowP = reinterpret_cast<uint2*>(rowA);
    int2* __restrict__ rowS = rowB;
    // anchor of the H' frame: the band's top row at the first column (H' = H - (column index + 1) * M)
    uint2 pt = make_uint2(0u, 0u);
    int naA = 0, naB = 0;
    { const int l0 = opaque(lane); if (l0 < n) pt = rowP[l0]; if (l0 < nA) naA = aA[l0]; if (l0 < nB) naB = aB[l0]; }
    int2 bw = rowS[0];                              // bases of the block the entries in `pt` belong to
    bw.x = __builtin_amdgcn_readfirstlane(bw.x); bw.y = __builtin_amdgcn_readfirstlane(bw.y);
    {
        const unsigned p0 = (unsigned)__builtin_amdgcn_readfirstlane((int)pt.x);
        acc.baseA = (int)(p0 & 0xffffu) - kBias + bw.x;     // = H'(band's top row, column 0)
        acc.baseB = (int)(p0 >> 16) - kBias + bw.y;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int hk = hcol0p(i0 + 1 + k, E, O);   // column 0: H' = H
        s.hl[k] = pack2(hk - acc.baseA + kBias, hk - acc.baseB + kBias);
        s.fb[k] = pack2(hk + O - acc.baseA + kBias, hk + O - acc.baseB + kBias);
        if (PERM) {                                 // byte selector {codeA, ->0, 4 + codeB, ->0}; rows beyond m select 0
            const unsigned cA = (i0 + k < mA) ? acgt_code(bA[i0 + k]) : 0x0cu;
            const unsigned cB = (i0 + k < mB) ? 4u + acgt_code(bB[i0 + k]) : 0x0cu;
            s.bch[k] = cA | 0x0c00u | (cB << 16) | 0x0c000000u;
        } else {
            const int cA = (i0 + k < mA) ? ((int)bA[i0 + k] << 8) : 0x1ff;
            const int cB = (i0 + k < mB) ? ((int)bB[i0 + k] << 8) : 0x1ff;
            s.bch[k] = pack2(cA, cB);
        }
    }
    {
        const int hd = hcol0p(i0, E, O);
        s.hd = pack2(hd - acc.baseA + kBias, hd - acc.baseB + kBias);
    }
    if (F3) {
        // the column-0 values just packed are exact ints: did they fit 16 bits at all?  (they fall by |E| per row from the band's top row;
        // a value that wrapped would look like a legitimate high one to the per-block look below)
        const int top = hcol0p(i0, E, O), bot = hcol0p(i0 + R, E, O) + O;      // highest / lowest of this lane's hd, hl[], fb[]
        const int lo_i = min(bot - acc.baseA, bot - acc.baseB) + kBias, hi_i = max(top - acc.baseA, top - acc.baseB) + kBias;
        if (__ballot(lo_i < 0 || hi_i > 0xffff) != 0ull) return false;
    }
    s.hbot = pack2(kBias, kBias); s.ebot = s.hbot; s.cch = 0u; s.sh = s.hbot; s.se = s.hbot; s.mx = 0u;
    s.cch2 = 0u; s.ach2 = 0u;
    const unsigned k4 = (pmx_s & 0xffu) * 0x01010101u;

    const int nsteps = n + 64;
    const int ae = -E;
    const EFrame F = {ae, FF ? ae + M : -ae};      // frame of the elided blocks: H+ (both gap extensions free) or H*
    const unsigned cm4 = (unsigned)(2 * ae + M) * 0x01010101u;   // FF: table of diagonal ADDENDS = cm4 - table of mismatch costs
    bool in_star = false;        // lane state currently in the frame of the elided blocks (kept across consecutive ones)
    for (int t0 = 0; t0 < nsteps; t0 += 64) {
        const int ln = opaque(lane);
        int lag = 64;            // columns the lane has moved since the H* frame's reference column was set
        if (t0 > 0 && t0 < n) {
            // re-anchor both pairs to the band's top row at column t0
            fold_max<R, kBias>(s, acc);
            // (a pair whose columns are exhausted keeps its last anchor: its lanes hold garbage from here on)
            const unsigned p0 = (unsigned)__builtin_amdgcn_readfirstlane((int)pt.x);
            const int nbA = (t0 < nA) ? (int)(p0 & 0xffffu) - kBias + bw.x : acc.baseA;
            const int nbB = (t0 < nB) ? (int)(p0 >> 16) - kBias + bw.y : acc.baseB;
            // (in the H* frame the reference column moves along with the block: +64|E| on everything)
            const int star = in_star ? -64 * F.col : 0;
            const unsigned pd = pack2(nbA - acc.baseA - star, nbB - acc.baseB - star);
            acc.baseA = nbA; acc.baseB = nbB;
            lag = 0;
#pragma unroll
            for (int k = 0; k < R; ++k) { s.hl[k] = pksub(s.hl[k], pd); s.fb[k] = pksub(s.fb[k], pd); }
            s.hd = pksub(s.hd, pd); s.hbot = pksub(s.hbot, pd); s.ebot = pksub(s.ebot, pd);
        } else if (t0 >= n) {
            fold_max<R, kBias>(s, acc);                   // last (drain) block keeps the previous anchor
        }
        // running-max anchor of this block: column index t0-63 in the H frame
        acc.ancA = acc.baseA + (t0 - 62) * M;
        acc.ancB = acc.baseB + (t0 - 62) * M;
        s.joff = pack2((63 - ln) * M, (63 - ln) * M);
        {
            const unsigned dlt = pack2(bw.x - acc.baseA, bw.y - acc.baseB);     // the writer's frame -> this block's frame, per half
            s.th = pkadd(pt.x, dlt);
            s.te = pkadd(pt.y, dlt);
        }
        unsigned tabA = 0u, tabB = 0u;                 // this block's column tables (PERM) / characters
        if (PERM) { tabA = acgt_table(naA, k4); tabB = acgt_table(naB, k4); }
        else      tabA = pack2(naA << 8, naB << 8);
        const int cn = t0 + 64 + ln;
        if (cn < n) pt = rowP[cn];
        if (cn < nA) naA = aA[cn];
        if (cn < nB) naB = aB[cn];
        if (t0 + 64 < n) { bw = rowS[(t0 + 64) >> 6]; bw.x = __builtin_amdgcn_readfirstlane(bw.x); bw.y = __builtin_amdgcn_readfirstlane(bw.y); }
        const bool steady = t0 >= 64 && t0 + 63 < nmin;
        bool elide = false;
        if (steady && elide_ok) {
            elide = block_cannot_raise_max<R, PARTIAL, kBias>(s, acc, in_star, ln, F, nvA, nvB, mA > band_row0, mB > band_row0, t0, M);
        }
        if (elide != in_star) {                        // frame change only where the kind of block changes
            if (elide) elided_frame<R>(s, ln, F, +1, 0);
            else       elided_frame<R>(s, ln, F, -1, lag);
            if (FF) { s.cch = cm4 - s.cch; s.cch2 = cm4 - s.cch2; }   // the column tables under way change meaning with the frame
            in_star = elide;
        }
        if (PERM) { s.ach = (FF && elide) ? cm4 - tabA : tabA; s.ach2 = (FF && elide) ? cm4 - tabB : tabB; }
        else      s.ach = tabA;
        if (elide) { const int v = eframe_top_shift<R>(F, ln); const unsigned pv = pack2(v, v); s.th = pkadd(s.th, pv); s.te = pkadd(s.te, pv); }
        bool fast = false;
        if (F3) {
            // where does the block start from, and can 64 steps take it out of range?  Down: a step lowers a value by at most a mismatch
            // plus a gap open, intermediates by a gap extension more; up: only the free-free frame's diagonal adds (2|E|+M per step)
            if (!steady && !elide && nA != nB) bound_garbage_columns<R, kBias>(s, t0, ln, nA, nB);   // (before the look: garbage halves at the bias)
            unsigned lo, hi;
            block_range<R>(s, t0 + ln < n, lo, hi);
            const int lo_min = min((int)(lo & 0xffffu), (int)(lo >> 16)), hi_max = max((int)(hi & 0xffffu), (int)(hi >> 16));
            const int dm = (int)(pmx_s & 0xffffu) + (int)(po_s & 0xffffu) + 2 * ae + M;
            // ... and what happens to every value BETWEEN this look and the next one: the next block's re-anchoring shifts by the top
            // row's change over 64 columns (at most |O| + |E| + M per column, plus 64 columns of the elided frame's tilt), a change of
            // frame adds the lane / row / column parts of elided_frame.  All of it inside the margins: no packed add can wrap unseen.
            const int reanchor = 64 * ((int)(po_s & 0xffffu) + ae + M) + 64 * max(ae, ae + M);
            const int reframe = 32 * abs(F.lane_coeff(R)) + ae * R + 64 * abs(F.col);
            const int mdn = 66 * dm + reanchor + reframe, mup = 66 * (2 * ae + M) + 130 * M + reanchor + reframe;           // (+ the running max's column offset, <= 127 M)
            if (lo_min < mdn || hi_max > 0xffff - mup || f3_mode == 3) return false;   // leaves the 16-bit window: the int32 kernel takes the couple
            fast = f3_mode != 2 && lo_min >= (int)kF16Lo + mdn && hi_max <= (int)kF16Hi - mup;
        }
        if (elide) {
            if (F3 && fast) {
#pragma unroll 2
                for (int q = 0; q < 64; ++q) pstep_elided<R, PERM, FF, true>(s, po, pmx, pf2);
            } else {
#pragma unroll 2
                for (int q = 0; q < 64; ++q) pstep_elided<R, PERM, FF>(s, po, pmx, pf2);
            }
            // the 64 captured bottom-row entries (column t0-64+lane): row 64R below the top, 63-lane columns behind t0-1
            { const int v = -eframe_bottom_shift<R>(F, ln); const unsigned pv = pack2(v, v); s.sh = pkadd(s.sh, pv); s.se = pkadd(s.se, pv); }
        } else if (steady) {
            if (F3 && fast) {
#pragma unroll 2
                for (int q = 0; q < 64; ++q) pstep<R, false, PARTIAL, PERM, true>(s, t0 + q, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf);
            } else {
#pragma unroll 2
                for (int q = 0; q < 64; ++q) pstep<R, false, PARTIAL, PERM>(s, t0 + q, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf);
            }
        } else {
            if (nA != nB) bound_garbage_columns<R, kBias>(s, t0, ln, nA, nB);   // beyond the shorter pair's end: keep its half in range
#pragma unroll 1
            for (int q = 0; q < 64; ++q) pstep<R, true, PARTIAL, PERM>(s, t0 + q, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf);
        }
        const int cs = t0 - 64 + opaque(lane);
        if (store_bottom && cs >= 0 && cs < n) rowP[cs] = make_uint2(s.sh, s.se);      // this block's captures, in this block's frame
        if (store_bottom && t0 >= 64 && ln == 0) rowS[(t0 - 64) >> 6] = make_int2(acc.baseA, acc.baseB);
    }
    fold_max<R, kBias>(s, acc);
    return true;
}

template <int R, bool PERM, bool FF, bool F3>
__device__ __forceinline__ bool pscore_couple(const uint8_t* aA, const uint8_t* bA, const uint8_t* aB, const uint8_t* bB,
                                              int nA, int mA, int nB, int mB, int2* rowA, int2* rowB, int lane,
                                              int M, int E, int O,
                                              unsigned pe, unsigned po, unsigned pm, unsigned pmx, bool elide_ok, int f3_mode,
                                              int& outA, int& outB)
{
    constexpr int kBias = F3 ? 0x4000 : tsta::kBias;      // as in pband
    const int n = max(nA, nB), mmax = max(mA, mB), mmin = min(mA, mB);
    {   // blockmatrix_init, psa.c:221-226: H(0,c+1) = O + (c+1)E, Ebar = H + O, in the packed row format of pband: H' = H - (c+1)M
        // relative to the block's first column
        uint2* rowP = reinterpret_cast<uint2*>(rowA);
        for (int c = opaque(lane); c < n; c += 64) {
            const int rel = (c & 63) * (E - M);                    // H'(c) - H'(first column of the block)
            rowP[c] = make_uint2(pack2(rel + kBias, rel + kBias), pack2(rel + O + kBias, rel + O + kBias));
            if ((c & 63) == 0) { const int b0 = O + (c + 1) * (E - M); rowB[c >> 6] = make_int2(b0, b0); }
        }
    }
    wave_mem_sync();
    PairAcc acc; acc.baseA = 0; acc.baseB = 0; acc.ancA = 0; acc.ancB = 0; acc.mxA = kNegInfP; acc.mxB = kNegInfP;
    constexpr int BAND = 64 * R;
    bool ok = true;
    for (int r0 = 0; r0 < mmax && ok; r0 += BAND) {
        const bool last = r0 + BAND >= mmax;
        const int rem = mmax - r0;
        // The LAST band need not be as tall as the others: the boundary-row format does not depend on R, so the rows that are
        // left are swept with the smallest band that covers them (50 000 rows = 24 bands of 2 048 + 848: a 1 024-row band
        // instead of a 2 048-row one, 0.35 % instead of 2.4 % of padded rows).  Its window is a subset of the tall band's.
        if (R >= 16 && rem <= 64 * 8) {
            if (r0 + 64 * 8 <= mmin) ok = pband<8, false, PERM, FF, F3>(aA, aB, nA, nB, bA, bB, mA, mB, false, r0, rowA, rowB, acc, lane, M, E, O, pe, po, pm, pmx, elide_ok, f3_mode);
            else                     ok = pband<8, true, PERM, FF, F3>(aA, aB, nA, nB, bA, bB, mA, mB, false, r0, rowA, rowB, acc, lane, M, E, O, pe, po, pm, pmx, elide_ok, f3_mode);
        } else if (R >= 32 && rem <= 64 * 16) {
            if (r0 + 64 * 16 <= mmin) ok = pband<16, false, PERM, FF, F3>(aA, aB, nA, nB, bA, bB, mA, mB, false, r0, rowA, rowB, acc, lane, M, E, O, pe, po, pm, pmx, elide_ok, f3_mode);
            else                      ok = pband<16, true, PERM, FF, F3>(aA, aB, nA, nB, bA, bB, mA, mB, false, r0, rowA, rowB, acc, lane, M, E, O, pe, po, pm, pmx, elide_ok, f3_mode);
        }
        else if (r0 + BAND <= mmin) ok = pband<R, false, PERM, FF, F3>(aA, aB, nA, nB, bA, bB, mA, mB, !last, r0, rowA, rowB, acc, lane, M, E, O, pe, po, pm, pmx, elide_ok, f3_mode);
        else                        ok = pband<R, true, PERM, FF, F3>(aA, aB, nA, nB, bA, bB, mA, mB, !last, r0, rowA, rowB, acc, lane, M, E, O, pe, po, pm, pmx, elide_ok, f3_mode);
        if (!last) wave_mem_sync();
    }
    outA = wave_max(acc.mxA);
    outB = wave_max(acc.mxB);
    return ok;
}

template <int R, int WAVES, bool F3>
__global__ __launch_bounds__(256, WAVES) void psa_score_packed_kernel(PsaPackedArgs P)
{
    const PsaBatchArgs& A = P.b;
    const int lane = threadIdx.x & 63;
    const int64_t slot = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int2* rowA = A.rows + slot * 2 * A.row_stride;
    int2* rowB = rowA + A.row_stride;
    const bool stamp = A.clock_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    if (stamp) { A.clock_out[0] = __builtin_amdgcn_s_memtime(); A.clock_out[1] = __builtin_amdgcn_s_memrealtime(); }
    for (;;) {
        unsigned int ticket = 0;
        if (lane == 0) ticket = atomicAdd(A.counter, 1u);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if ((int64_t)ticket >= A.packed_couples) break;         // every wave reaches this exit
        if (!psa_couple_is_packed(A.len1, A.len2, A.order, ticket)) continue;   // K1 (int32) takes these two pairs
        const int64_t pa = psa_pair_at(A.order, 2 * (int64_t)ticket), pb = psa_pair_at(A.order, 2 * (int64_t)ticket + 1);
        const int l1a = A.len1[pa], l2a = A.len2[pa], l1b = A.len1[pb], l2b = A.len2[pb];
        const uint8_t* s1a = A.seqs + A.off1[pa]; const uint8_t* s2a = A.seqs + A.off2[pa];
        const uint8_t* s1b = A.seqs + A.off1[pb]; const uint8_t* s2b = A.seqs + A.off2[pb];
        const bool swa = l2a > l1a, swb = l2b > l1b;            // psa.c:194-199
        const uint8_t* aA = swa ? s2a : s1a; const uint8_t* bA = swa ? s1a : s2a;
        const uint8_t* aB = swb ? s2b : s1b; const uint8_t* bB = swb ? s1b : s2b;
        const int nA = swa ? l2a : l1a, mA = swa ? l1a : l2a;
        const int nB = swb ? l2b : l1b, mB = swb ? l1b : l2b;
        int ra, rb;
        bool ok;
        const unsigned long long t_begin = P.wave_log ? __builtin_amdgcn_s_memrealtime() : 0ull;
        // both pairs pure ACGT (flags from psa_alphabet_flags_kernel): the mismatch term is one byte-permute lookup
        const bool perm = P.acgt != nullptr && __builtin_amdgcn_readfirstlane((int)(P.acgt[pa] & P.acgt[pb] & 1)) != 0;
        if (perm && P.ffree) ok = pscore_couple<R, true, true, F3>(aA, bA, aB, bB, nA, mA, nB, mB, rowA, rowB, lane, A.M, A.E, A.O, P.pe, P.po, P.pm, P.pmx, P.elide != 0, P.f3, ra, rb);
        else if (perm)       ok = pscore_couple<R, true, false, F3>(aA, bA, aB, bB, nA, mA, nB, mB, rowA, rowB, lane, A.M, A.E, A.O, P.pe, P.po, P.pm, P.pmx, P.elide != 0, P.f3, ra, rb);
        else                 ok = pscore_couple<R, false, false, F3>(aA, bA, aB, bB, nA, mA, nB, mB, rowA, rowB, lane, A.M, A.E, A.O, P.pe, P.po, P.pm, P.pmx, P.elide != 0, P.f3, ra, rb);
        if (lane == 0) {
            if (ok) { A.out[pa] = max(ra, -100); A.out[pb] = max(rb, -100); }    // ms floor, psa.c:110,590
            else    { A.redo[pa] = 1; A.redo[pb] = 1; }                         // left the 16-bit window: psa_score_batch_kernel takes both pairs
        }
        if (P.wave_log && lane == 0) {                            // where and when this couple ran (diagnostics)
            unsigned long long* w = P.wave_log + 4 * (size_t)ticket;
            w[0] = t_begin; w[1] = __builtin_amdgcn_s_memrealtime();
            w[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);   // HW_REG_HW_ID, HW_REG_XCC_ID
            w[3] = (unsigned long long)slot;
        }
        wave_mem_sync();
    }
    if (stamp) { A.clock_out[2] = __builtin_amdgcn_s_memtime(); A.clock_out[3] = __builtin_amdgcn_s_memrealtime(); }
}

// f3 (PsaPackedArgs::f3 != 0): the instantiations whose scores sit around 0x4000 and whose steady blocks may use v_pk_maximum3_f16
void launch_psa_score_packed(int R, const PsaPackedArgs& args, int blocks, hipStream_t stream)
{
    const bool f3 = args.f3 != 0;
    if (R == 8) {
        if (f3) hipLaunchKernelGGL((psa_score_packed_kernel<8, 4, true>), dim3(blocks), dim3(256), 0, stream, args);
        else    hipLaunchKernelGGL((psa_score_packed_kernel<8, 4, false>), dim3(blocks), dim3(256), 0, stream, args);
    } else if (R == 32) {
        if (f3) hipLaunchKernelGGL((psa_score_packed_kernel<32, 2, true>), dim3(blocks), dim3(256), 0, stream, args);
        else    hipLaunchKernelGGL((psa_score_packed_kernel<32, 2, false>), dim3(blocks), dim3(256), 0, stream, args);
    } else {
        if (f3) hipLaunchKernelGGL((psa_score_packed_kernel<16, 3, true>), dim3(blocks), dim3(256), 0, stream, args);
        else    hipLaunchKernelGGL((psa_score_packed_kernel<16, 3, false>), dim3(blocks), dim3(256), 0, stream, args);
    }
}

int psa_packed_blocks_per_cu(int R)
{
    int nb = 0;
    hipError_t e = (R == 8)  ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psa_score_packed_kernel<8, 4, true>, 256, 0)
                 : (R == 32) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psa_score_packed_kernel<32, 2, true>, 256, 0)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psa_score_packed_kernel<16, 3, true>, 256, 0);
    if (e != hipSuccess || nb < 1) nb = 1;
    return nb > 8 ? 8 : nb;
}

}  // namespace tsta
