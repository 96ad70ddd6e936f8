// psa_trace_packed.hip — K2p: the trace fill of psa_trace.hip for TWO pairs per wave in packed 16-bit lanes.
//
// Same path, same trace bytes, same band pipeline as K2 (reference: psa/psa.c:294-327,358-362 for what a cell
// records; :593-639 for the wavefront it replaces), with the arithmetic of K1p (psa_score_packed.hip): every
// register holds job A in its low and job B in its high 16 bits, scores are stored biased around an int32 base per
// job that is re-anchored to the band's top row every 64 steps, in the column-shifted frame H' = H - (column+1)*M.
// All comparisons a trace byte records are between values of ONE cell (H vs D, H vs F, H+O vs F, H+O vs E), so they
// are frame-independent.  Per cell pair: 11 VALU for the recurrence + running max, 13 for the two trace codes
// (min(h-d,1), min(h-f,1), two clamps of a packed signed difference to {-1,0,1}, three v_pk_mad_u16) and one
// v_perm_b32 + 1/2 add to sort the codes of 4 cells into one dword per job: ~12.5 per cell instead of K2's 25.7.
//
// The two jobs of a couple may differ in size: the wave sweeps max(n) columns and max(m) rows, both jobs' boundary
// rows and trace are laid out for that geometry, and cells outside a job compute garbage that feeds only other such
// cells, is masked out of the job's running max and lands in trace bytes the walker never visits.
//
// The same tile body without the trace (TRACE = false) is K1b, psa_score_bands_kernel: the score-only fill of a
// batch too small to give every SIMD two waves of K1p — there the bands of every couple run concurrently in the
// band pipeline instead of one after the other in one wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "kernels.h"
#include "packed_ops.h"
#include "packed_step.h"
#include "handoff.h"
#include "trace_types.h"
#include "wave_ops.h"

namespace tsta {

// The compiler expands a packed signed clamp into per-half compares and selects and a packed multiply-add into
// shift + add, so the four packed instructions the trace code is built from are spelled out (one asm statement per
// instruction: the scheduler stays free to interleave them).
// clamp(a - b, -1, 1) per signed 16-bit half (a, b biased scores: |a - b| is far below 2^15)
__device__ __forceinline__ unsigned pk_sign(unsigned a, unsigned b, unsigned mone, unsigned one)
{
    unsigned d, r;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(d), "v"(mone));
    asm("v_pk_min_i16 %0, %1, %2" : "=v"(d) : "v"(r), "v"(one));
    return d;
}
// a * k + c per 16-bit half (wraps); k replicated in both halves of a VGPR
__device__ __forceinline__ unsigned pk_mad(unsigned a, unsigned k, unsigned c)
{
    unsigned r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(k), "v"(c));
    return r;
}

// packed constants of the trace code, pinned in VGPRs
struct PkConst { unsigned one, mone, k2, k4, k16, k20, k21; };

// One step of the lane's R rows for both jobs WITH the trace codes (the score-only tiles use packed_step.h directly).
// code[k] = per half, the trace byte of row k: nd + 2nf + 4(fge+1) + 16(ege+1) = (nd + 20) + 2nf + 4fge + 16ege with
// fge, ege in {-1,0,1}; the +20 rides on nd so that every partial sum stays a non-negative byte.
// ELIDED: the block provably cannot raise either job's maximum (packed_step.h): no running max, and the frame in
// which the vertical gap costs nothing (pe unused, pf = 2|E|+M); implies !PRED, and needs no PARTIAL masks.
template <int R, bool PRED, bool PARTIAL, bool PERM, bool ELIDED>
__device__ __forceinline__ void ptstep(PLane<R>& s, const int t, const int lane, const int n, const int nA, const int nB,
                                       const int nvA, const int nvB, const unsigned pe, const unsigned po, const unsigned pm,
                                       const unsigned pmx, const unsigned pf, const PkConst& K, unsigned (&code)[R])
{
    unsigned hu, eu;
    if (PERM) lane_exchange2(s.cch, s.cch2, hu, eu, s.sh, s.se, s.ach, s.ach2, s.th, s.te, s.hbot, s.ebot);
    else      lane_exchange(s.cch, hu, eu, s.sh, s.se, s.ach, s.th, s.te, s.hbot, s.ebot);
    const bool act = !PRED || ((unsigned)(t - lane) < (unsigned)n);
    unsigned d = s.hd;
    unsigned h = 0;
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const unsigned e = ELIDED ? eu : eu - pe;        // E' = Ebar'(up) - |E|    (E* = Ebar*(up))
        const unsigned f = s.fb[k] - pf;                 // F' = Fbar'(left) - (|E|+M)   (F*: 2|E|+M)
        const unsigned tt = mismatch_term<PERM>(s.cch, s.cch2, s.bch[k], pmx);
        const unsigned dg = d - tt;                      // D' = H'diag - (match ? 0 : M-X)
        h = pkmax(pkmax(dg, e), f);
        const unsigned g = h - po;                       // H' + O
        {
            const unsigned nd = pkmin((h + K.k20) - dg, K.k21);   // psa.c:294-299: back; min(x,1) + 20 (x >= 0): bytes stay >= 0
            const unsigned nf = pkmin(h - f, K.one);
            const unsigned fge = pk_sign(g, f, K.mone, K.one);    // psa.c:302-313 (fback), one cell to the left of where it is used
            const unsigned ege = pk_sign(g, e, K.mone, K.one);    // psa.c:315-327 (eback), one cell above
            code[k] = pk_mad(ege, K.k16, pk_mad(fge, K.k4, pk_mad(nf, K.k2, nd)));
        }
        const unsigned eb = pkmax(e, g);
        const unsigned fbn = pkmax(f, g);
        d = s.hl[k];
        if (PRED) {
            s.hl[k] = act ? h : s.hl[k];
            s.fb[k] = act ? fbn : s.fb[k];
        } else {
            s.hl[k] = h;
            s.fb[k] = fbn;
        }
        if (!ELIDED) {
            unsigned hv = h;
            if (PARTIAL) hv &= ((k < nvA) ? 0xffffu : 0u) | ((k < nvB) ? 0xffff0000u : 0u);
            m = (k == 0) ? hv : pkmax(m, hv);
        }
        eu = eb;
    }
    s.hbot = h;
    s.ebot = eu;
    if (PRED) s.hd = act ? hu : s.hd; else s.hd = hu;
    if (!ELIDED) {
        unsigned cand = m + s.joff;                      // running max in the H frame (see pstep)
        if (PRED) {
            const unsigned c = (unsigned)(t - lane);
            cand &= ((c < (unsigned)nA) ? 0xffffu : 0u) | ((c < (unsigned)nB) ? 0xffff0000u : 0u);
        }
        if (PARTIAL) cand &= ((nvA > 0) ? 0xffffu : 0u) | ((nvB > 0) ? 0xffff0000u : 0u);
        s.mx = pkmax(s.mx, cand);
        s.joff += pm;
    }
}

// 64 steps with trace; the trace bytes of every 16/R steps leave as one uint4 per lane and job
template <int R, bool PRED, bool PARTIAL, bool PERM, bool ELIDED>
__device__ __forceinline__ void ptblock(PLane<R>& s, const int t0, const int lane, const int n, const int nA, const int nB,
                                        const int nvA, const int nvB, const unsigned pe, const unsigned po, const unsigned pm,
                                        const unsigned pmx, const unsigned pf, const PkConst& K,
                                        uint4* __restrict__ tqA, uint4* __restrict__ tqB)
{
    static_assert(R == 4, "byte gather below is written for 4 rows per lane");
    constexpr int SPB = TraceCfg<R>::SPB;
#pragma unroll 1
    for (int g = 0; g < 64 / SPB; ++g) {
        unsigned accA[4], accB[4];
#pragma unroll
        for (int q = 0; q < SPB; ++q) {
            unsigned code[R];
            ptstep<R, PRED, PARTIAL, PERM, ELIDED>(s, t0 + g * SPB + q, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf, K, code);
            // bytes 0 / 2 of the four codes -> one dword per job (v_perm_b32: selectors 0-3 second operand, 4-7 first)
            const unsigned p01 = __builtin_amdgcn_perm(code[1], code[0], 0x06020400u);   // A0 A1 B0 B1
            const unsigned p23 = __builtin_amdgcn_perm(code[3], code[2], 0x06020400u);   // A2 A3 B2 B3
            accA[q] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
            accB[q] = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
        }
        tqA[(size_t)g * 64] = make_uint4(accA[0], accA[1], accA[2], accA[3]);
        tqB[(size_t)g * 64] = make_uint4(accB[0], accB[1], accB[2], accB[3]);
    }
}

// everything one wave needs of its couple and band
struct CoupleView {
    const uint8_t *aA, *bA, *aB, *bB;           // columns / rows of job A and B
    int nA, mA, nB, mB;
    const int2 *topA, *topB;                    // top row of the band ({H, Ebar} per column); nullptr: band 0, the matrix
    int2 *botA, *botB;                          //   boundary row is computed (psa.c:221-226); bottom row out
    bool store_bottom;
    uint8_t *tbandA, *tbandB;                   // TRACE: this band's trace bytes
    int* st;                                    // this band's per-lane state between tiles
    int w;                                      // band index
};

// steps [s0, s1) of one band of one couple.  PERSIST: the tile runs inside the single-launch (dataflow) form: the tiles
// it depends on ran in OTHER waves of the same launch, so everything it exchanges with them — top row in, bottom row
// out, the band's lane state in and out — goes through sc1 loads / stores (handoff.h); the kernel around it waits for the
// two producer tiles before and publishes this band's progress after.
// F3 (score-only tiles): as in psa_score_packed.hip — scores around the bias 0x4000 (bit patterns of positive normal binary16
// numbers), a steady block whose operands provably stay in [0x0400, 0x7bff] takes max3 as one v_pk_maximum3_f16; returns false when a
// block would leave the 16-bit window (the caller flags the couple for the int32 kernel).  The lane state between tiles holds the
// same biased values: every tile of a launch uses the same instantiation.
template <int R, bool TRACE, bool PERM, bool FF = false, bool PERSIST = false, bool F3 = false>
__device__ __forceinline__ bool packed_tile(const CoupleView& V, const int s0, const int s1, const int lane,
                                            const int M, const int E, const int O,
                                            const unsigned pe_s, const unsigned po_s, const unsigned pm_s, const unsigned pmx_s,
                                            const bool elide_ok, const int f3_mode = 0)
{
    static_assert(!F3 || !TRACE, "the binary16-order maximum is for score-only tiles");
    constexpr int kBias = F3 ? 0x4000 : tsta::kBias;      // (shadows the namespace constant inside this tile)
    bool in_window = true;
    constexpr int SPB = TraceCfg<TRACE ? R : 4>::SPB;
    const int nA = V.nA, mA = V.mA, nB = V.nB, mB = V.mB, w = V.w;
    const int n = max(nA, nB), nmin = min(nA, nB), mmin = min(mA, mB);
    const uint8_t* __restrict__ aA = V.aA; const uint8_t* __restrict__ bA = V.bA;
    const uint8_t* __restrict__ aB = V.aB; const uint8_t* __restrict__ bB = V.bB;
    const unsigned pe = in_vgpr(pe_s), po = in_vgpr(po_s), pm = in_vgpr(pm_s), pmx = in_vgpr(pmx_s);
    PkConst K;
    K.one = in_vgpr(0x00010001u); K.mone = in_vgpr(0xffffffffu); K.k2 = in_vgpr(0x00020002u); K.k4 = in_vgpr(0x00040004u);
    K.k16 = in_vgpr(0x00100010u); K.k20 = in_vgpr(0x00140014u); K.k21 = in_vgpr(0x00150015u);
    const unsigned pf = pe + pm;
    const unsigned pf2 = pf + pe;                  // horizontal gap in the frame of the elided blocks
    const unsigned k4 = (pmx_s & 0xffu) * 0x01010101u;
    const int ae = -E;
    static_assert(!FF || (PERM && !TRACE), "free-free frame: score-only tiles of pure-ACGT couples");
    const EFrame F = {ae, FF ? ae + M : -ae};      // frame of the elided blocks (packed_step.h)
    const unsigned cm4 = (unsigned)(2 * ae + M) * 0x01010101u;
    constexpr int BAND = 64 * R;
    const int band_row0 = w * BAND;
    const bool partial = band_row0 + BAND > mmin;
    const int i0 = band_row0 + lane * R;
    const int nvA = mA - i0, nvB = mB - i0;
    int* st = V.st + lane;

    PLane<R> s;
    PairAcc acc;
    bool in_star = false;                          // lane state in the frame of the elided blocks (kept across blocks and tiles)
#pragma unroll
    for (int k = 0; k < R; ++k) {
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
    const int2* __restrict__ topA = V.topA;
    const int2* __restrict__ topB = V.topB;
    int2* __restrict__ botA = V.botA;
    int2* __restrict__ botB = V.botB;
    // top-row entry of column c: from the band above, or the matrix boundary row H(0,c+1) = O + (c+1)E, Ebar = H + O
    auto top_entry = [&](const int2* __restrict__ top, int c) -> int2 {
        if (PERSIST && top && w > 0) return load_sc1_int2(&top[c]);     // written inside this launch by another wave: never a plain load
        if (top) return top[c];
        const int h = O + (c + 1) * E;
        return make_int2(h, h + O);
    };

    if (s0 == 0) {
        // anchor of the H' frame: the band's top row at the first column (H' = H - (column index + 1) * M)
        acc.baseA = __builtin_amdgcn_readfirstlane(top_entry(topA, 0).x) - M;
        acc.baseB = __builtin_amdgcn_readfirstlane(top_entry(topB, 0).x) - M;
        acc.mxA = kNegInfP; acc.mxB = kNegInfP;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int hk = hcol0p(i0 + 1 + k, E, O);    // column 0: H' = H
            s.hl[k] = pack2(hk - acc.baseA + kBias, hk - acc.baseB + kBias);
            s.fb[k] = pack2(hk + O - acc.baseA + kBias, hk + O - acc.baseB + kBias);
        }
        const int hd = hcol0p(i0, E, O);
        s.hd = pack2(hd - acc.baseA + kBias, hd - acc.baseB + kBias);
        if (F3) {                                     // (as in psa_score_packed.hip: the exact column-0 values must fit 16 bits)
            const int bot = hcol0p(i0 + R, E, O) + O;
            const int lo_i = min(bot - acc.baseA, bot - acc.baseB) + kBias, hi_i = max(hd - acc.baseA, hd - acc.baseB) + kBias;
            if (__ballot(lo_i < 0 || hi_i > 0xffff) != 0ull) in_window = false;
        }
        s.hbot = pack2(kBias, kBias); s.ebot = s.hbot; s.cch = 0u; s.cch2 = 0u;
    } else {
        auto ld = [&](int idx) -> int { return PERSIST ? load_sc1_int(&st[idx * 64]) : st[idx * 64]; };
#pragma unroll
        for (int k = 0; k < R; ++k) { s.hl[k] = (unsigned)ld(2 * k); s.fb[k] = (unsigned)ld(2 * k + 1); }
        s.hd = (unsigned)ld(2 * R); s.hbot = (unsigned)ld(2 * R + 1); s.ebot = (unsigned)ld(2 * R + 2);
        s.cch = (unsigned)ld(2 * R + 3);
        if (PERM) s.cch2 = (unsigned)ld(2 * R + 9);
        in_star = __builtin_amdgcn_readfirstlane(ld(2 * R + 8)) != 0;
        acc.mxA = ld(2 * R + 4); acc.mxB = ld(2 * R + 5);
        acc.baseA = __builtin_amdgcn_readfirstlane(ld(2 * R + 6)); acc.baseB = __builtin_amdgcn_readfirstlane(ld(2 * R + 7));
    }
    s.sh = pack2(kBias, kBias); s.se = s.sh; s.mx = 0u; s.ach2 = 0u;
    if (!PERM) s.cch2 = 0u;
    acc.ancA = 0; acc.ancB = 0;

    // top-row entries and column characters of a block are loaded one block ahead (within the tile: everything this
    // launch may read of the band above was written by earlier launches)
    int2 ntA = make_int2(0, 0), ntB = make_int2(0, 0);
    int naA = 0, naB = 0;
    { const int c = s0 + opaque(lane); if (c < nA) { ntA = top_entry(topA, c); naA = aA[c]; } if (c < nB) { ntB = top_entry(topB, c); naB = aB[c]; } }
    int lag = 64;                                  // columns moved since the H* frame's reference column was set
    for (int t0 = s0; t0 < s1; t0 += 64) {
        const int ln = opaque(lane);
        lag = 64;
        if (t0 > 0 && t0 < n) {
            // re-anchor both jobs to the band's top row at column t0 (a job whose columns are exhausted keeps its anchor)
            fold_mx<kBias>(s.mx, acc);
            const int nbA = (t0 < nA) ? __builtin_amdgcn_readfirstlane(ntA.x) - (t0 + 1) * M : acc.baseA;
            const int nbB = (t0 < nB) ? __builtin_amdgcn_readfirstlane(ntB.x) - (t0 + 1) * M : acc.baseB;
            // (in the H* frame the reference column moves along with the block: +64|E| on everything)
            const int star = in_star ? -64 * F.col : 0;
            const unsigned pd = pack2(nbA - acc.baseA - star, nbB - acc.baseB - star);
            lag = 0;
            acc.baseA = nbA; acc.baseB = nbB;
#pragma unroll
            for (int k = 0; k < R; ++k) { s.hl[k] = pksub(s.hl[k], pd); s.fb[k] = pksub(s.fb[k], pd); }
            s.hd = pksub(s.hd, pd); s.hbot = pksub(s.hbot, pd); s.ebot = pksub(s.ebot, pd);
        } else if (t0 >= n) {
            fold_mx<kBias>(s.mx, acc);                    // drain block keeps the previous anchor
        }
        acc.ancA = acc.baseA + (t0 - 62) * M;      // running-max anchor of this block: column t0-63 in the H frame
        acc.ancB = acc.baseB + (t0 - 62) * M;
        s.joff = pack2((63 - ln) * M, (63 - ln) * M);
        {
            const int sh_m = (t0 + ln + 1) * M;    // column shift of this lane's top-row entry
            s.th = pack2(ntA.x - sh_m - acc.baseA + kBias, ntB.x - sh_m - acc.baseB + kBias);
            s.te = pack2(ntA.y - sh_m - acc.baseA + kBias, ntB.y - sh_m - acc.baseB + kBias);
        }
        unsigned tabA = 0u, tabB = 0u;                 // this block's column tables (PERM) / characters
        if (PERM) { tabA = acgt_table(naA, k4); tabB = acgt_table(naB, k4); }
        else      tabA = pack2(naA << 8, naB << 8);
        if (t0 + 64 < s1) {
            const int cn = t0 + 64 + ln;
            ntA = make_int2(0, 0); ntB = make_int2(0, 0); naA = 0; naB = 0;
            if (cn < nA) { ntA = top_entry(topA, cn); naA = aA[cn]; }
            if (cn < nB) { ntB = top_entry(topB, cn); naB = aB[cn]; }
        }
        uint4* tqA = nullptr; uint4* tqB = nullptr;
        if (TRACE) {
            tqA = reinterpret_cast<uint4*>(V.tbandA) + ((size_t)(t0 / SPB) * 64 + ln);
            tqB = reinterpret_cast<uint4*>(V.tbandB) + ((size_t)(t0 / SPB) * 64 + ln);
        }
        const bool steady = t0 >= 64 && t0 + 63 < nmin;
        // blocks that provably cannot raise either maximum run without the running max, in the H* frame (packed_step.h)
        bool elide = false;
        if (steady && elide_ok)
            elide = partial ? block_cannot_raise_max<R, true, kBias>(s, acc, in_star, ln, F, nvA, nvB, mA > band_row0, mB > band_row0, t0, M)
                            : block_cannot_raise_max<R, false, kBias>(s, acc, in_star, ln, F, nvA, nvB, true, true, t0, M);
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
        if (F3) {                                      // the block's range against what 64 steps can move it by (psa_score_packed.hip)
            if (!steady && !elide && nA != nB) bound_garbage_columns<R, kBias>(s, t0, ln, nA, nB);   // (before the look: garbage halves at the bias)
            unsigned lo, hi;
            block_range<R>(s, t0 + ln < n, lo, hi);            // (beyond both jobs' end the entry is loaded as 0: not a score; beyond ONE
                                                               //  job's end its half was just set to the bias)
            const int lo_min = min((int)(lo & 0xffffu), (int)(lo >> 16)), hi_max = max((int)(hi & 0xffffu), (int)(hi >> 16));
            const int dm = (int)(pmx_s & 0xffffu) + (int)(po_s & 0xffffu) + 2 * ae + M;
            // ... and what happens to every value BETWEEN this look and the next one: the next block's re-anchoring shifts by the top
            // row's change over 64 columns (at most |O| + |E| + M per column, plus 64 columns of the elided frame's tilt), a change of
            // frame adds the lane / row / column parts of elided_frame.  All of it inside the margins: no packed add can wrap unseen.
            const int reanchor = 64 * ((int)(po_s & 0xffffu) + ae + M) + 64 * max(ae, ae + M);
            const int reframe = 32 * abs(F.lane_coeff(R)) + ae * R + 64 * abs(F.col);
            const int mdn = 66 * dm + reanchor + reframe, mup = 66 * (2 * ae + M) + 130 * M + reanchor + reframe;
            if (lo_min < mdn || hi_max > 0xffff - mup || f3_mode == 3) in_window = false;   // (the tile is finished all the same: its
            fast = in_window && f3_mode != 2 && lo_min >= (int)kF16Lo + mdn && hi_max <= (int)kF16Hi - mup;   //  dependants must not wait)
        }
        if (elide) {
            if constexpr (TRACE) {
                ptblock<R, false, false, PERM, true>(s, t0, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf2, K, tqA, tqB);
            } else if (F3 && fast) {
#pragma unroll 2
                for (int q = 0; q < 64; ++q) pstep_elided<R, PERM, FF, true>(s, po, pmx, pf2);
            } else {
#pragma unroll 2
                for (int q = 0; q < 64; ++q) pstep_elided<R, PERM, FF>(s, po, pmx, pf2);
            }
            { const int v = -eframe_bottom_shift<R>(F, ln); const unsigned pv = pack2(v, v); s.sh = pkadd(s.sh, pv); s.se = pkadd(s.se, pv); }
        } else if constexpr (TRACE) {
            if (!steady && nA != nB) bound_garbage_columns<R, kBias>(s, t0, ln, nA, nB);   // beyond the shorter job's end: keep its half in range
            if (steady && !partial)      ptblock<R, false, false, PERM, false>(s, t0, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf, K, tqA, tqB);
            else if (steady)             ptblock<R, false, true, PERM, false>(s, t0, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf, K, tqA, tqB);
            else                         ptblock<R, true, true, PERM, false>(s, t0, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf, K, tqA, tqB);
        } else if (steady && !partial) {
            if (F3 && fast) {
#pragma unroll 2
                for (int q = 0; q < 64; ++q) pstep<R, false, false, PERM, true>(s, t0 + q, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf);
            } else {
#pragma unroll 2
                for (int q = 0; q < 64; ++q) pstep<R, false, false, PERM>(s, t0 + q, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf);
            }
        } else if (steady) {
            if (F3 && fast) {
#pragma unroll 2
                for (int q = 0; q < 64; ++q) pstep<R, false, true, PERM, true>(s, t0 + q, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf);
            } else {
#pragma unroll 2
                for (int q = 0; q < 64; ++q) pstep<R, false, true, PERM>(s, t0 + q, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf);
            }
        } else {
            if (nA != nB) bound_garbage_columns<R, kBias>(s, t0, ln, nA, nB);
#pragma unroll 1
            for (int q = 0; q < 64; ++q) pstep<R, true, true, PERM>(s, t0 + q, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf);
        }
        const int cs = t0 - 64 + ln;
        if (V.store_bottom && cs >= 0 && cs < n) {
            const int sh_m = (cs + 1) * M;
            const int hA = (int)(s.sh & 0xffffu) - kBias + acc.baseA + sh_m, eA = (int)(s.se & 0xffffu) - kBias + acc.baseA + sh_m;
            const int hB = (int)(s.sh >> 16) - kBias + acc.baseB + sh_m, eB = (int)(s.se >> 16) - kBias + acc.baseB + sh_m;
            if (PERSIST) { store_sc1_int2(&botA[cs], hA, eA); store_sc1_int2(&botB[cs], hB, eB); }
            else         { botA[cs] = make_int2(hA, eA); botB[cs] = make_int2(hB, eB); }
        }
    }
    fold_mx<kBias>(s.mx, acc);
    auto sv = [&](int idx, int v) { if (PERSIST) store_sc1_int(&st[idx * 64], v); else st[idx * 64] = v; };
#pragma unroll
    for (int k = 0; k < R; ++k) { sv(2 * k, (int)s.hl[k]); sv(2 * k + 1, (int)s.fb[k]); }
    sv(2 * R, (int)s.hd); sv(2 * R + 1, (int)s.hbot); sv(2 * R + 2, (int)s.ebot);
    sv(2 * R + 3, (int)s.cch);
    sv(2 * R + 4, acc.mxA); sv(2 * R + 5, acc.mxB);
    sv(2 * R + 6, acc.baseA); sv(2 * R + 7, acc.baseB);
    sv(2 * R + 8, in_star ? 1 : 0);
    if (PERM) sv(2 * R + 9, (int)s.cch2);
    return in_window;
}

// PERSIST = false: one tile of the launch-per-diagonal pipeline (launch d advances every band by one T-step tile).
// PERSIST = true: the SAME tiles as one launch, in dataflow form.  Tickets enumerate the tiles in the order the launch
// pipeline runs them (d-major: tile_first[d] .. tile_first[d+1] are the bands [g_lo(d), g_hi(d)) of "launch" d); a wave
// takes a ticket, waits until the two tiles it depends on have published — the band's own previous tile (progress[g] >= s0)
// and the tile of the band above that ends 64 steps later (progress[g-1] >= s1 + 64) — runs the tile and publishes
// progress[g] = s1.  Both producers belong to "launch" d-1, i.e. hold earlier tickets and are finished or running: no
// deadlock whatever the number of resident waves.  Unlike the launches, a slow tile delays only its dependents, and there
// are no launch gaps; unlike a wave-per-band form, every resident wave always has runnable work (measured on 64 pairs of
// 50 kbp: wave-per-band 88-99 ms, launches 72 ms on free hardware queues / 99 ms when the process has other streams).
template <int R, bool PERSIST>
__global__ __launch_bounds__(256, 4) void psa_trace_packed_tile_kernel(PsaTracePackedArgs A)
{
    constexpr int NS = packed_state_ints(R);
    const int lane = threadIdx.x & 63;
    for (;;) {
    int g, d = A.d;
    bool mute = false;                          // test hook: this tile is computed but never published
    if (PERSIST) {
        unsigned ticket = 0;
        if (lane == 0) ticket = atomicAdd(A.ticket, 1u);
        const int t = (int)__builtin_amdgcn_readfirstlane(ticket);
        if (t >= A.n_tiles) return;
        mute = A.test_abandon != 0 && t == A.test_abandon - 1;
        int lo = 0, hi = A.n_d - 1;             // d = last one whose first tile is <= t
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (A.tile_first[mid] <= t) lo = mid; else hi = mid - 1; }
        d = lo;
        g = A.tile_glo[d] + (t - A.tile_first[d]);
    } else {
        g = A.g_lo + blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        if (g >= A.g_hi) return;
    }
    int lo = 0, hi = A.n_couples - 1;           // couple = last one whose first band is <= g
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (A.couples[mid].band0 <= g) lo = mid; else hi = mid - 1; }
    const TraceCouple& C = A.couples[lo];
    const int w = g - C.band0;
    if (w >= C.W) { if (PERSIST) continue; return; }
    const int shift = w * (A.T + 64);
    int s0 = d * A.T - shift, s1 = (d + 1) * A.T - shift;
    s0 = max(0, min(s0, C.S_total));
    s1 = max(0, min(s1, C.S_total));
    if (s0 >= s1) { if (PERSIST) continue; return; }

    const TraceJob& JA = A.jobs[C.ja];
    const TraceJob& JB = A.jobs[C.jb];
    CoupleView V;
    V.aA = JA.a; V.bA = JA.b; V.aB = JB.a; V.bB = JB.b;
    V.nA = JA.n; V.mA = JA.m; V.nB = JB.n; V.mB = JB.m;
    V.topA = JA.rows + (size_t)w * JA.row_stride; V.topB = JB.rows + (size_t)w * JB.row_stride;
    V.botA = JA.rows + (size_t)(w + 1) * JA.row_stride; V.botB = JB.rows + (size_t)(w + 1) * JB.row_stride;
    V.store_bottom = true;
    V.tbandA = JA.trace + (size_t)w * JA.band_bytes; V.tbandB = JB.trace + (size_t)w * JB.band_bytes;
    V.st = C.state + (size_t)w * NS * 64;
    V.w = w;
    if (PERSIST) {
        if (s0 > 0) wait_progress(A.progress + g, s0, A.status, lane);                                  // this band's previous tile
        if (w > 0)  wait_progress(A.progress + g - 1, min(s1 + 64, C.S_total), A.status, lane);         // the band above, 64 steps further
    }
    if (C.perm) packed_tile<R, true, true, false, PERSIST>(V, s0, s1, lane, A.M, A.E, A.O, A.pe, A.po, A.pm, A.pmx, A.elide != 0);
    else        packed_tile<R, true, false, false, PERSIST>(V, s0, s1, lane, A.M, A.E, A.O, A.pe, A.po, A.pm, A.pmx, A.elide != 0);
    if (!PERSIST) return;
    if (!mute) publish_progress(A.progress + g, s1, lane);
    }
}

// per couple (blockIdx.x): out[0] of both jobs = max over all lanes' running maxima, floored at -100 (psa.c:590)
template <int R>
__global__ void psa_trace_packed_finish_kernel(const TraceJob* jobs, const TraceCouple* couples)
{
    constexpr int NS = packed_state_ints(R);
    const TraceCouple& C = couples[blockIdx.x];
    int mxA = kNegInfP, mxB = kNegInfP;
    for (int idx = threadIdx.x; idx < C.W * 64; idx += blockDim.x) {
        const int w = idx / 64, lane = idx % 64;
        mxA = max(mxA, C.state[(size_t)w * NS * 64 + (2 * R + 4) * 64 + lane]);
        mxB = max(mxB, C.state[(size_t)w * NS * 64 + (2 * R + 5) * 64 + lane]);
    }
    __shared__ int redA[256], redB[256];
    redA[threadIdx.x] = mxA; redB[threadIdx.x] = mxB;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { redA[threadIdx.x] = max(redA[threadIdx.x], redA[threadIdx.x + o]); redB[threadIdx.x] = max(redB[threadIdx.x], redB[threadIdx.x + o]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        jobs[C.ja].out[0] = max(redA[0], -100); jobs[C.ja].out[1] = 0;
        jobs[C.jb].out[0] = max(redB[0], -100); jobs[C.jb].out[1] = 0;
    }
}

// ---- K1b: score-only band pipeline for batches that cannot give every SIMD two waves of K1p ------------------------
// Couples and orientation exactly as in K1p (adjacent pairs of the geometry order, psa_couple_is_packed; K1 takes the
// rest); workspaces are laid out uniformly for max_len so that a wave finds its couple and band by division:
//   band g -> couple g / W_max, band g % W_max;  rows: [pair][W_max+1][row_stride] int2;  state: [couple][W_max][2R+8][64]
// PERSIST: the same tiles as ONE launch in dataflow form (see psa_trace_packed_tile_kernel<.., true>): tickets enumerate the
// tiles d-major, a wave waits for the band's previous tile and the upper band's tile 64 steps further, runs the tile with sc1
// loads / stores of rows and lane state, publishes.
template <int R, bool PERSIST, bool F3>
__global__ __launch_bounds__(256, (R == 16 ? 3 : 4)) void psa_score_bands_kernel(PsaBandsArgs P)
{
    constexpr int NS = packed_state_ints(R);
    const PsaBatchArgs& A = P.b;
    const int lane = threadIdx.x & 63;
    for (;;) {
    int64_t c; int w, d = P.d;
    bool mute = false;                          // test hook: this tile is computed but never published
    if (PERSIST) {
        unsigned ticket = 0;
        if (lane == 0) ticket = atomicAdd(P.ticket, 1u);
        const int t = (int)__builtin_amdgcn_readfirstlane(ticket);
        if (t >= P.n_tiles) return;
        mute = P.test_abandon != 0 && t == P.test_abandon - 1;
        int lo = 0, hi = P.n_d - 1;             // d = last one whose first tile is <= t
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (P.tile_first[mid] <= t) lo = mid; else hi = mid - 1; }
        d = lo;
        const int idx = t - P.tile_first[d], nw = P.tile_nw[d];
        c = idx / nw; w = P.tile_wlo[d] + idx % nw;
    } else {
        const int64_t g = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        c = g / P.W_max; w = (int)(g % P.W_max);
    }
    if (c >= A.packed_couples) { if (PERSIST) continue; return; }
    if (!psa_couple_is_packed(A.len1, A.len2, A.order, c)) { if (PERSIST) continue; return; }   // K1 (int32) takes these two pairs
    const int64_t pa = psa_pair_at(A.order, 2 * c), pb = psa_pair_at(A.order, 2 * c + 1);
    const int l1a = A.len1[pa], l2a = A.len2[pa], l1b = A.len1[pb], l2b = A.len2[pb];
    const uint8_t* s1a = A.seqs + A.off1[pa]; const uint8_t* s2a = A.seqs + A.off2[pa];
    const uint8_t* s1b = A.seqs + A.off1[pb]; const uint8_t* s2b = A.seqs + A.off2[pb];
    const bool swa = l2a > l1a, swb = l2b > l1b;                             // psa.c:194-199
    CoupleView V;
    V.aA = swa ? s2a : s1a; V.bA = swa ? s1a : s2a; V.aB = swb ? s2b : s1b; V.bB = swb ? s1b : s2b;
    V.nA = swa ? l2a : l1a; V.mA = swa ? l1a : l2a; V.nB = swb ? l2b : l1b; V.mB = swb ? l1b : l2b;
    const int n = max(V.nA, V.nB), m = max(V.mA, V.mB);
    const int W = (m + 64 * R - 1) / (64 * R);
    if (w >= W) { if (PERSIST) continue; return; }
    const int S_total = (n + 64 + 63) / 64 * 64;
    const int shift = w * (P.T + 64);
    int s0 = d * P.T - shift, s1 = (d + 1) * P.T - shift;
    s0 = max(0, min(s0, S_total));
    s1 = max(0, min(s1, S_total));
    if (s0 >= s1) { if (PERSIST) continue; return; }
    int2* rowsA = A.rows + (size_t)(2 * c) * P.rows_per_pair;
    int2* rowsB = rowsA + P.rows_per_pair;
    V.topA = w ? rowsA + (size_t)w * A.row_stride : nullptr; V.topB = w ? rowsB + (size_t)w * A.row_stride : nullptr;
    V.botA = rowsA + (size_t)(w + 1) * A.row_stride; V.botB = rowsB + (size_t)(w + 1) * A.row_stride;
    V.store_bottom = w + 1 < W;
    V.tbandA = nullptr; V.tbandB = nullptr;
    V.st = P.state + ((size_t)c * P.W_max + w) * NS * 64;
    V.w = w;
    int* prog = PERSIST ? P.progress + (size_t)c * P.W_max + w : nullptr;
    if (PERSIST) {
        if (s0 > 0) wait_progress(prog, s0, P.status, lane);                                   // this band's previous tile
        if (w > 0)  wait_progress(prog - 1, min(s1 + 64, S_total), P.status, lane);            // the band above, 64 steps further
    }
    // both pairs pure ACGT (flags from psa_alphabet_flags_kernel): the mismatch term is one byte-permute lookup
    const bool perm = P.acgt != nullptr && __builtin_amdgcn_readfirstlane((int)(P.acgt[pa] & P.acgt[pb] & 1)) != 0;
    bool ok;
    if (perm && P.ffree) ok = packed_tile<R, false, true, true, PERSIST, F3>(V, s0, s1, lane, A.M, A.E, A.O, P.pe, P.po, P.pm, P.pmx, P.elide != 0, P.f3);
    else if (perm)       ok = packed_tile<R, false, true, false, PERSIST, F3>(V, s0, s1, lane, A.M, A.E, A.O, P.pe, P.po, P.pm, P.pmx, P.elide != 0, P.f3);
    else                 ok = packed_tile<R, false, false, false, PERSIST, F3>(V, s0, s1, lane, A.M, A.E, A.O, P.pe, P.po, P.pm, P.pmx, P.elide != 0, P.f3);
    if (F3 && !ok && lane == 0) { A.redo[pa] = 1; A.redo[pb] = 1; }           // left the 16-bit window: psa_score_batch_kernel scores both pairs
    if (!PERSIST) return;
    if (!mute) publish_progress(prog, s1, lane);
    }
}

// per couple (blockIdx.x): out[pair] = max over the bands' lanes, floored at -100 (psa.c:110,590).  After a single launch
// that abandoned a wait (P.status set) every score of the call is poison (INT32_MIN): never a silently wrong answer; the
// host-pointer entry points then run the call again through the launch pipeline (psa_api.hip).
template <int R>
__global__ void psa_score_bands_finish_kernel(PsaBandsArgs P)
{
    constexpr int NS = packed_state_ints(R);
    const PsaBatchArgs& A = P.b;
    const int64_t c = blockIdx.x;
    if (!psa_couple_is_packed(A.len1, A.len2, A.order, c)) return;
    const int64_t pa = psa_pair_at(A.order, 2 * c), pb = psa_pair_at(A.order, 2 * c + 1);
    const int mA = min(A.len1[pa], A.len2[pa]), mB = min(A.len1[pb], A.len2[pb]);
    const int W = (max(mA, mB) + 64 * R - 1) / (64 * R);
    const int* state = P.state + (size_t)c * P.W_max * NS * 64;
    int mxA = kNegInfP, mxB = kNegInfP;
    for (int idx = threadIdx.x; idx < W * 64; idx += blockDim.x) {
        const int w = idx / 64, lane = idx % 64;
        mxA = max(mxA, state[(size_t)w * NS * 64 + (2 * R + 4) * 64 + lane]);
        mxB = max(mxB, state[(size_t)w * NS * 64 + (2 * R + 5) * 64 + lane]);
    }
    __shared__ int redA[256], redB[256];
    redA[threadIdx.x] = mxA; redB[threadIdx.x] = mxB;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { redA[threadIdx.x] = max(redA[threadIdx.x], redA[threadIdx.x + o]); redB[threadIdx.x] = max(redB[threadIdx.x], redB[threadIdx.x + o]); }
        __syncthreads();
    }
    const bool poisoned = P.status != nullptr && *P.status != 0u;
    if (threadIdx.x == 0) { A.out[pa] = poisoned ? INT32_MIN : max(redA[0], -100); A.out[pb] = poisoned ? INT32_MIN : max(redB[0], -100); }
}

void launch_psa_score_bands(int R, const PsaBandsArgs& args, int blocks, hipStream_t stream)
{
    const bool f3 = args.f3 != 0;
    if (R == 16) { if (f3) hipLaunchKernelGGL((psa_score_bands_kernel<16, false, true>), dim3(blocks), dim3(256), 0, stream, args);
                   else    hipLaunchKernelGGL((psa_score_bands_kernel<16, false, false>), dim3(blocks), dim3(256), 0, stream, args); }
    else         { if (f3) hipLaunchKernelGGL((psa_score_bands_kernel<8, false, true>), dim3(blocks), dim3(256), 0, stream, args);
                   else    hipLaunchKernelGGL((psa_score_bands_kernel<8, false, false>), dim3(blocks), dim3(256), 0, stream, args); }
}

void launch_psa_score_bands_persistent(int R, const PsaBandsArgs& args, int n_cu, hipStream_t stream)
{
    const bool f3 = args.f3 != 0;
    int nb = 0;
    hipError_t e = (R == 16) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psa_score_bands_kernel<16, true, true>, 256, 0)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psa_score_bands_kernel<8, true, true>, 256, 0);
    if (e != hipSuccess || nb < 1) nb = 1;
    // two workgroups per CU = two waves per SIMD: more resident waves only lengthen the pipeline's skew (measured in round 4 on
    // config #2 / 128 x 50 kbp: 3 per CU 4.41 / 51.2 ms, 4 per CU 4.56 / 51.5 ms against 4.16 / 40.5 — a SIMD serves its waves oldest-first)
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(((int64_t)args.n_tiles + 3) / 4, (int64_t)n_cu * std::min(nb, 2)));
    if (R == 16) { if (f3) hipLaunchKernelGGL((psa_score_bands_kernel<16, true, true>), dim3((unsigned)blocks), dim3(256), 0, stream, args);
                   else    hipLaunchKernelGGL((psa_score_bands_kernel<16, true, false>), dim3((unsigned)blocks), dim3(256), 0, stream, args); }
    else         { if (f3) hipLaunchKernelGGL((psa_score_bands_kernel<8, true, true>), dim3((unsigned)blocks), dim3(256), 0, stream, args);
                   else    hipLaunchKernelGGL((psa_score_bands_kernel<8, true, false>), dim3((unsigned)blocks), dim3(256), 0, stream, args); }
}

void launch_psa_score_bands_finish(int R, const PsaBandsArgs& args, hipStream_t stream)
{
    if (R == 16) hipLaunchKernelGGL((psa_score_bands_finish_kernel<16>), dim3((unsigned)args.b.packed_couples), dim3(256), 0, stream, args);
    else         hipLaunchKernelGGL((psa_score_bands_finish_kernel<8>), dim3((unsigned)args.b.packed_couples), dim3(256), 0, stream, args);
}

void launch_psa_trace_packed_tile(const PsaTracePackedArgs& args, int blocks, hipStream_t stream)
{
    hipLaunchKernelGGL((psa_trace_packed_tile_kernel<kPackedTraceR, false>), dim3(blocks), dim3(256), 0, stream, args);
}

// the single-launch (dataflow) form: as many 256-thread blocks as stay resident (or fewer, if there are fewer tiles)
void launch_psa_trace_packed_persistent(const PsaTracePackedArgs& args, int n_cu, hipStream_t stream)
{
    static int per_cu = 0;
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psa_trace_packed_tile_kernel<kPackedTraceR, true>, 256, 0) != hipSuccess || nb < 1) nb = 1;
        per_cu = nb > 2 ? 2 : nb;            // measured (64 pairs of 50 kbp): 2 blocks per CU 72-74 ms, 3: 75-81, 4: 81
    }
    const int64_t blocks = std::min<int64_t>(((int64_t)args.n_tiles + 3) / 4, (int64_t)n_cu * per_cu);
    hipLaunchKernelGGL((psa_trace_packed_tile_kernel<kPackedTraceR, true>), dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(256), 0, stream, args);
}

void launch_psa_trace_packed_finish(const TraceJob* jobs, const TraceCouple* couples, int n_couples, hipStream_t stream)
{
    hipLaunchKernelGGL((psa_trace_packed_finish_kernel<kPackedTraceR>), dim3(n_couples), dim3(256), 0, stream, jobs, couples);
}

}  // namespace tsta
