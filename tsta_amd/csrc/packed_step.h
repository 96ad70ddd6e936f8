// packed_step.h — the DP step of the packed 16-bit kernels (K1p psa_score_packed.hip, K1b psa_score_bands_kernel):
// lane state, the generic step with running max, the elided step in the row-shifted frame, the frame change and the
// bound that licenses it.
#pragma once
#include <hip/hip_runtime.h>
#include "packed_ops.h"
#include "wave_ops.h"

namespace tsta {

// 0 / M-X per half for (column character, row character):
//   generic bytes:  min_u16((a ^ b) << 8, M-X)                                              (2 VALU)
//   PERM couples (both pairs pure ACGT): one v_perm_b32 byte lookup.  The column carries, per pair, a 4-byte table
//   indexed by the ROW's 2-bit code ((c >> 1) & 3: A 0, C 1, T 2, G 3) holding 0 at the column's own code and M-X
//   elsewhere; the row register is the byte selector {codeA, 0x0c (-> 0x00), 4 + codeB, 0x0c}: selectors 0-3 read
//   the second table operand (pair A), 4-7 the first (pair B).
template <bool PERM>
__device__ __forceinline__ unsigned mismatch_term(unsigned cch, unsigned cch2, unsigned bch, unsigned pmx)
{
    if (PERM) return __builtin_amdgcn_perm(cch2, cch, bch);
    return pkmin(cch ^ bch, pmx);
}
__device__ __forceinline__ unsigned acgt_code(int c) { return ((unsigned)c >> 1) & 3u; }
// the column's table for one pair: M-X in the four bytes, 0 at the column's own code
__device__ __forceinline__ unsigned acgt_table(int c, unsigned k4) { return k4 & ~(0xffu << (8 * acgt_code(c))); }


template <int R>
struct PLane {
    unsigned hl[R];    // H'(row k, previous column)           (biased, packed A|B; H' = H - column*M, see pstep)
    unsigned fb[R];    // Fbar'(row k, previous column)
    unsigned bch[R];   // row characters << 8 in each half
    unsigned hd;       // H'(row above first row, previous column)
    unsigned hbot, ebot, cch, th, te, ach, sh, se, mx;
    unsigned cch2, ach2;   // PERM couples: the column character travels as two byte tables (pair A in cch/ach, pair B here)
    unsigned joff;     // (column of this step - (t0 - 63)) * M in both halves: turns H' back into H for the running max
};

// All scores are kept in the column-shifted frame  H'(i,j) = H(i,j) - j*M  (same for E, F, Ebar, Fbar).  The
// diagonal term then needs no "+M":  D' = H'(i-1,j-1) - (match ? 0 : M-X);  the vertical gap is unchanged
// (E' = Ebar'(i-1,j) - |E|, same column) and the horizontal gap pays the shift:  F' = Fbar'(i,j-1) - (|E|+M).
// That removes one add per cell; the price is one add + one max per STEP to feed the running max in the H frame.
// F3: the three-way maximum as ONE v_pk_maximum3_f16 (packed_ops.h: pkmax3f) — the caller guarantees that every operand of the
// block lies in the range where that is an exact u16 max3.
template <int R, bool PRED, bool PARTIAL, bool PERM, bool F3 = false>
__device__ __forceinline__ void pstep(PLane<R>& s, const int t, const int lane, const int n, const int nA, const int nB,
                                      const int nvA, const int nvB,
                                      const unsigned pe, const unsigned po, const unsigned pm, const unsigned pmx,
                                      const unsigned pf)
{
    unsigned hu, eu;
    if (PERM) lane_exchange2(s.cch, s.cch2, hu, eu, s.sh, s.se, s.ach, s.ach2, s.th, s.te, s.hbot, s.ebot);
    else      lane_exchange(s.cch, hu, eu, s.sh, s.se, s.ach, s.th, s.te, s.hbot, s.ebot);

    const bool act = !PRED || ((unsigned)(t - lane) < (unsigned)n);
    unsigned d = s.hd;
    unsigned h = 0;
    unsigned m = 0;                                     // max of this step's H' over the lane's rows
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const unsigned e = eu - pe;                      // E' = Ebar'(up) - |E|        (v_sub_u32 on both halves)
        const unsigned f = s.fb[k] - pf;                 // F' = Fbar'(left) - (|E|+M)
        const unsigned tt = mismatch_term<PERM>(s.cch, s.cch2, s.bch[k], pmx);   // 0 or M-X
        const unsigned dg = d - tt;                      // D' = H'diag - (match ? 0 : M-X)
        h = F3 ? pkmax3f(dg, e, f) : pkmax(pkmax(dg, e), f);
        const unsigned g = h - po;                       // H' + O
        const unsigned eb = pkmax(e, g);
        const unsigned fbn = pkmax(f, g);
        d = s.hl[k];
        unsigned hv = h;
        if (PARTIAL) hv &= ((k < nvA) ? 0xffffu : 0u) | ((k < nvB) ? 0xffff0000u : 0u);   // rows beyond m of either pair
        if (PRED) {
            s.hl[k] = act ? h : s.hl[k];
            s.fb[k] = act ? fbn : s.fb[k];
        } else {
            s.hl[k] = h;
            s.fb[k] = fbn;
        }
        m = (k == 0) ? hv : pkmax(m, hv);
        eu = eb;
    }
    s.hbot = h;
    s.ebot = eu;
    if (PRED) s.hd = act ? hu : s.hd; else s.hd = hu;
    // running max in the H frame: H - anchor = H' + (column - anchor column)*M ; joff >= 0, no carry across halves
    unsigned cand = m + s.joff;
    if (PRED) {                                         // columns outside either pair's sequence do not count
        const unsigned c = (unsigned)(t - lane);
        cand &= ((c < (unsigned)nA) ? 0xffffu : 0u) | ((c < (unsigned)nB) ? 0xffff0000u : 0u);
    }
    if (PARTIAL) cand &= ((nvA > 0) ? 0xffffu : 0u) | ((nvB > 0) ? 0xffff0000u : 0u);
    s.mx = pkmax(s.mx, cand);
    s.joff += pm;
}

// ---- steady-state step without the running max ("elided" blocks) -------------------------------------------------
// In the H' frame every transition is <= 0 (diagonal 0 / X-M, vertical -|E|, horizontal -|E|-M, opening O <= 0), so no
// cell of a 64-step block exceeds the largest H' that ENTERS the block (the lanes' previous-column H', the diagonal
// entries and the 64 top-row entries), and  H = H' + (column+1)*M  <=  that maximum + (t0+64)*M.  When this bound is
// <= the maximum the pair has ALREADY reached, the block cannot change maxsorce: every cell is still computed (the
// bands below need the boundary row), but nothing has to be compared against the running max, and with no maximum to
// keep the rows need not share a frame either.  Such a block runs in the frame
//     H*(i,c) = H'(i,c) + |E|*(i - top row of the band) - |E|*(c - (t0-1))  (- a per-band constant, for the u16 range)
// in which the VERTICAL gap costs nothing (E* = Ebar*(up)), the diagonal still costs 0 / X-M and the horizontal gap
// 2|E|+M:  8 + 1 VALU per cell pair instead of 11.  State is converted on entry and exit (2R+5 packed adds each, per
// 64 steps), so blocks of either kind can follow each other freely and results are identical by construction.
// FF ("free-free", pure-ACGT couples whose scoring has X >= -2|E|): the frame
//     H+(i,c) = H'(i,c) + |E|*(i - top) + (|E|+M)*(c - c_ref)
// in which BOTH gap extensions cost nothing; the diagonal then ADDS 2|E|+M (match) or 2|E|+X (mismatch, >= 0 by the
// condition), which the column's byte table delivers directly: 7 VALU per cell pair.
template <int R, bool PERM, bool FF = false, bool F3 = false>
__device__ __forceinline__ void pstep_elided(PLane<R>& s, const unsigned po, const unsigned pmx, const unsigned pf2)
{
    static_assert(!FF || PERM, "the free-free frame needs the byte-table lookup");
    unsigned hu, eu;
    if (PERM) lane_exchange2(s.cch, s.cch2, hu, eu, s.sh, s.se, s.ach, s.ach2, s.th, s.te, s.hbot, s.ebot);
    else      lane_exchange(s.cch, hu, eu, s.sh, s.se, s.ach, s.th, s.te, s.hbot, s.ebot);
    unsigned d = s.hd;
    unsigned h = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const unsigned f = FF ? s.fb[k] : s.fb[k] - pf2; // F* = Fbar*(left) - (2|E|+M);  F+ = Fbar+(left)
        const unsigned tt = mismatch_term<PERM>(s.cch, s.cch2, s.bch[k], pmx);
        const unsigned dg = FF ? d + tt : d - tt;        // FF: the table holds 2|E|+M / 2|E|+X
        h = F3 ? pkmax3f(dg, eu, f) : pkmax(pkmax(dg, eu), f);   // E* = Ebar*(up): no subtraction
        const unsigned g = h - po;
        d = s.hl[k];
        s.hl[k] = h;
        s.fb[k] = pkmax(f, g);
        eu = pkmax(eu, g);
    }
    s.hbot = h;
    s.ebot = eu;
    s.hd = hu;
}

// The frame of the elided blocks relative to H':  + ae*(row - top row) + col*(column - reference column), minus a
// per-band constant that centres it in the 16-bit window.  col = -|E| (H*), or |E|+M (H+, see pstep_elided<FF>).
struct EFrame {
    int ae, col;
    __device__ __forceinline__ int lane_coeff(int R) const { return ae * R - col; }   // step from lane to lane
};
// shift of this lane's top-row entry (column t0 + ln) / of its captured bottom-row entry (column t0 - 64 + ln)
template <int R> __device__ __forceinline__ int eframe_top_shift(const EFrame F, const int ln) { return F.col * (ln + 1) - 32 * F.lane_coeff(R); }
template <int R> __device__ __forceinline__ int eframe_bottom_shift(const EFrame F, const int ln) { return F.ae * 64 * R + F.col * (ln - 63) - 32 * F.lane_coeff(R); }

// frame change H' -> elided frame (sign = +1) or back (sign = -1) of everything a lane carries across a block boundary;
// `after` = 0 at the block's first step, 64 after its last one (the lane has moved 64 columns to the right).
template <int R>
__device__ __forceinline__ void elided_frame(PLane<R>& s, const int ln, const EFrame F, const int sign, const int after)
{
    const int lane_part = F.lane_coeff(R) * (ln - 32) + F.col * after;   // rows above this lane + its column lead, centred
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int v = sign * (lane_part + F.ae * (k + 1));
        const unsigned pv = pack2(v, v);
        s.hl[k] = pkadd(s.hl[k], pv);
        s.fb[k] = pkadd(s.fb[k], pv);
    }
    { const int v = sign * lane_part;              const unsigned pv = pack2(v, v); s.hd = pkadd(s.hd, pv); }
    { const int v = sign * (lane_part + F.ae * R); const unsigned pv = pack2(v, v); s.hbot = pkadd(s.hbot, pv); s.ebot = pkadd(s.ebot, pv); }
}

template <int R, int BIAS = kBias>
__device__ __forceinline__ void fold_max(PLane<R>& s, PairAcc& acc) { fold_mx<BIAS>(s.mx, acc); }



// The two pairs of a couple may have different numbers of columns.  Beyond the end of the shorter one its half of every
// register is garbage that nobody reads — but the plain 32-bit adds / subs of the packed step are only carry-free
// while BOTH halves stay inside the 16-bit window, so the garbage must stay bounded too.  (1) top-row entries of columns
// beyond a pair's end enter at the bias level; (2) once no lane of the wave is on a real column of that pair any more
// (t0 - 63 >= its n) its half of the lane state is flattened to the bias at the start of every block, so it can drift
// by at most 64 columns' worth of score.
template <int R, int BIAS = kBias>
__device__ __forceinline__ void flatten_half(PLane<R>& s, const bool low)
{
    const unsigned keep = low ? 0xffff0000u : 0x0000ffffu, fill = low ? (unsigned)BIAS : ((unsigned)BIAS << 16);
#pragma unroll
    for (int k = 0; k < R; ++k) { s.hl[k] = (s.hl[k] & keep) | fill; s.fb[k] = (s.fb[k] & keep) | fill; }
    // (hbot / ebot are left alone: lane 63's still hold the pair's LAST real column on their way into the bottom row;
    //  from the next step on they are recomputed from the flattened state)
    s.hd = (s.hd & keep) | fill;
    s.th = (s.th & keep) | fill; s.te = (s.te & keep) | fill;
}
template <int R, int BIAS = kBias>
__device__ __forceinline__ void bound_garbage_columns(PLane<R>& s, const int t0, const int ln, const int nA, const int nB)
{
    const int c = t0 + ln;                              // column of this lane's top-row entry
    if (c >= nA) { s.th = (s.th & 0xffff0000u) | (unsigned)BIAS; s.te = (s.te & 0xffff0000u) | (unsigned)BIAS; }
    if (c >= nB) { s.th = (s.th & 0x0000ffffu) | ((unsigned)BIAS << 16); s.te = (s.te & 0x0000ffffu) | ((unsigned)BIAS << 16); }
    if (t0 - 63 >= nA) flatten_half<R, BIAS>(s, true);
    if (t0 - 63 >= nB) flatten_half<R, BIAS>(s, false);
}

// Can the 64-step block starting at t0 raise either pair's maximum?  (see pstep_elided)  Entering H' values: the lanes'
// previous-column rows, the diagonal entries and the 64 top-row entries; rows beyond the end of a pair hold garbage
// that no real cell ever reads and do not count.  `in_star`: the lane state is currently in the H* frame.
template <int R, bool PARTIAL, int BIAS = kBias>
__device__ __forceinline__ bool block_cannot_raise_max(const PLane<R>& s, const PairAcc& acc, const bool in_star, const int ln,
                                                       const EFrame F, const int nvA, const int nvB, const bool rowsA, const bool rowsB,
                                                       const int t0, const int M)
{
    const int ae = F.ae;
    const unsigned lane_s = in_star ? pack2(F.lane_coeff(R) * (ln - 32), F.lane_coeff(R) * (ln - 32)) : 0u;
    unsigned v = pkmax(PARTIAL ? (pksub(s.hd, lane_s) & (((nvA > 0) ? 0xffffu : 0u) | ((nvB > 0) ? 0xffff0000u : 0u))) : pksub(s.hd, lane_s),
                       PARTIAL ? (s.th & ((rowsA ? 0xffffu : 0u) | (rowsB ? 0xffff0000u : 0u))) : s.th);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        unsigned hv = s.hl[k];
        if (in_star) hv = pksub(hv, pkadd(lane_s, pack2(ae * (k + 1), ae * (k + 1))));        // back to H'
        if (PARTIAL) hv &= ((k < nvA) ? 0xffffu : 0u) | ((k < nvB) ? 0xffff0000u : 0u);
        v = pkmax(v, hv);
    }
    const int ubA = wave_max((int)(v & 0xffffu)) - BIAS + acc.baseA + (t0 + 64) * M;
    const int ubB = wave_max((int)(v >> 16)) - BIAS + acc.baseB + (t0 + 64) * M;
    return ubA <= max(wave_max(acc.mxA), -100) && ubB <= max(wave_max(acc.mxB), -100);
}

// Range of everything a 64-step block starts from, per half, over the whole wave: the lanes' rows hl[], the diagonal entry and the
// 64 top-row entries (Fbar, Ebar and the bottom-row pair lie within |O| below their H).  lo = packed minima, hi = packed maxima.
// What the block computes from them moves by at most `per step` bounds the caller knows, so one look per block decides whether the
// block may run with pkmax3f (operands inside [kF16Lo, kF16Hi]), must run with the two-instruction u16 maximum, or has left the
// 16-bit window altogether.
// th_real: this lane's top-row entry belongs to a real column (beyond the sequences' end the register holds whatever was loaded
// last, which never reaches a real cell).
template <int R>
__device__ __forceinline__ void block_range(const PLane<R>& s, const bool th_real, unsigned& lo, unsigned& hi)
{
    const unsigned thv = th_real ? s.th : s.hd;
    lo = pkmin(s.hd, thv); hi = pkmax(s.hd, thv);
#pragma unroll
    for (int k = 0; k < R; ++k) { lo = pkmin(lo, s.hl[k]); hi = pkmax(hi, s.hl[k]); }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = pkmin(lo, (unsigned)__shfl_xor((int)lo, off, 64));
        hi = pkmax(hi, (unsigned)__shfl_xor((int)hi, off, 64));
    }
}

}  // namespace tsta
