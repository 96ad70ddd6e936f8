// psa_score_packed.hip — K1p: batched score-only fill, two pairs per wave in packed 16-bit lanes.
//
// Same path and same results as K1 (psa_score.hip; reference: psa/psa.c:219-232, 234-433, 593-639) but about
// half the VALU issue per cell.  Measured on gfx950 (tools/valu_microbench.hip): v_add/v_sub/v_xor issue at
// ~1.1 cycles per wave64, every max/min/cndmask/v_pk_* at ~2.1 — so the cell is rebuilt from
//   * plain 32-bit v_sub_u32 / v_add_u32 / v_xor_b32 acting on BOTH 16-bit halves at once (no borrow/carry
//     can cross the halves because scores are kept biased and non-negative), and
//   * v_pk_max_u16 / v_pk_min_u16 for the five max/min,
// i.e. 6 fast + 6 packed ops per TWO cells.  The two halves of every register belong to two different pairs
// (a "couple") with identical geometry; their lanes, skew, band hand-off and DPP traffic are shared.
//
// Scores are stored as  v = H - base + 0x8000  with an int32 `base` per pair that is re-anchored to the band's
// top row every 64 steps; all values live within (64R+192)*U of the base (U = per-step score bound), which the
// host checks against the 16-bit range before choosing this kernel (otherwise K1's int32 path runs).
// The match/mismatch select is  t = min_u16((a^b) << 8, M-X)  ->  D = (Hdiag + M) - t.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "wave_ops.h"

namespace tsta {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pkmax(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ unsigned pkmin(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ unsigned pksub(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, (us2)(__builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ unsigned pack2(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }
__device__ __forceinline__ unsigned ushr1(unsigned old, unsigned src) { return (unsigned)wave_shr1((int)old, (int)src); }
__device__ __forceinline__ unsigned ushl1(unsigned old, unsigned src) { return (unsigned)wave_shl1((int)old, (int)src); }

constexpr int kBias = 0x8000;
constexpr int kNegInfP = -(1 << 30);

template <int R>
struct PLane {
    unsigned hm[R];    // H(row k, previous column) + M        (biased, packed A|B)
    unsigned fb[R];    // Fbar(row k, previous column)
    unsigned bch[R];   // row characters << 8 in each half
    unsigned hdm;      // H(row above first row, previous column) + M
    unsigned hbot, ebot, cch, th, te, ach, sh, se, mx;
};

template <int R, bool PRED, bool PARTIAL>
__device__ __forceinline__ void pstep(PLane<R>& s, const int t, const int lane, const int n, const int nvalid,
                                      const unsigned pe, const unsigned po, const unsigned pm, const unsigned pmx)
{
    s.cch = ushr1(s.ach, s.cch);
    const unsigned hu = ushr1(s.th, s.hbot);
    unsigned eu = ushr1(s.te, s.ebot);
    s.sh = ushl1(s.hbot, s.sh);
    s.se = ushl1(s.ebot, s.se);
    s.th = ushl1(s.th, s.th);
    s.te = ushl1(s.te, s.te);
    s.ach = ushl1(s.ach, s.ach);

    const bool act = !PRED || ((unsigned)(t - lane) < (unsigned)n);
    unsigned d = s.hdm;
    unsigned h = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const unsigned e = eu - pe;                      // E = Ebar(up) + E          (v_sub_u32 on both halves)
        const unsigned f = s.fb[k] - pe;                 // F = Fbar(left) + E
        const unsigned x = s.cch ^ s.bch[k];             // 0 iff characters equal, else >= 256 per half
        const unsigned tt = pkmin(x, pmx);               // 0 or M-X
        const unsigned dg = d - tt;                      // (Hdiag + M) - (match ? 0 : M-X)
        h = pkmax(pkmax(dg, e), f);
        const unsigned g = h - po;                       // H + O
        const unsigned eb = pkmax(e, g);
        const unsigned fbn = pkmax(f, g);
        d = s.hm[k];
        unsigned hv = h;
        if (PARTIAL) hv = (k < nvalid) ? hv : 0u;
        if (PRED) {
            s.hm[k] = act ? h + pm : s.hm[k];
            s.fb[k] = act ? fbn : s.fb[k];
            hv = act ? hv : 0u;
        } else {
            s.hm[k] = h + pm;
            s.fb[k] = fbn;
        }
        s.mx = pkmax(s.mx, hv);
        eu = eb;
    }
    s.hbot = h;
    s.ebot = eu;
    if (PRED) s.hdm = act ? hu + pm : s.hdm; else s.hdm = hu + pm;
}

__device__ __forceinline__ int hcol0p(int i, int E, int O) { return i == 0 ? 0 : O + i * E; }

struct PairAcc { int baseA, baseB, mxA, mxB; };

template <int R>
__device__ __forceinline__ void fold_max(PLane<R>& s, PairAcc& acc)
{
    acc.mxA = max(acc.mxA, (int)(s.mx & 0xffffu) + acc.baseA - kBias);
    acc.mxB = max(acc.mxB, (int)(s.mx >> 16) + acc.baseB - kBias);
    s.mx = 0u;
}

// VALU instructions with an SGPR source issue in the slow class on gfx950 (tools/valu_microbench.hip: v_sub_u32
// 1.1 -> 2.0+ cycles), so the four packed constants are pinned in VGPRs.
__device__ __forceinline__ unsigned in_vgpr(unsigned x) { asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "s"(x)); return x; }

template <int R, bool PARTIAL>
__device__ __forceinline__ void pband(const uint8_t* __restrict__ aA, const uint8_t* __restrict__ aB, const int n,
                                      const uint8_t* __restrict__ bA, const uint8_t* __restrict__ bB, const int m,
                                      const int band_row0, int2* __restrict__ rowA, int2* __restrict__ rowB,
                                      PairAcc& acc, const int lane, const int M, const int E, const int O,
                                      const unsigned pe_s, const unsigned po_s, const unsigned pm_s, const unsigned pmx_s)
{
    const unsigned pe = in_vgpr(pe_s), po = in_vgpr(po_s), pm = in_vgpr(pm_s), pmx = in_vgpr(pmx_s);
    PLane<R> s;
    const int i0 = band_row0 + lane * R;
    const int nvalid = m - i0;
    // anchor: top row of the band at the first column
    int2 ntA = make_int2(0, 0), ntB = make_int2(0, 0);
    int naA = 0, naB = 0;
    { const int l0 = opaque(lane); if (l0 < n) { ntA = rowA[l0]; ntB = rowB[l0]; naA = aA[l0]; naB = aB[l0]; } }
    acc.baseA = __builtin_amdgcn_readfirstlane(ntA.x);
    acc.baseB = __builtin_amdgcn_readfirstlane(ntB.x);
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int hk = hcol0p(i0 + 1 + k, E, O);
        s.hm[k] = pack2(hk + M - acc.baseA + kBias, hk + M - acc.baseB + kBias);
        s.fb[k] = pack2(hk + O - acc.baseA + kBias, hk + O - acc.baseB + kBias);
        const int cA = (i0 + k < m) ? ((int)bA[i0 + k] << 8) : 0x1ff;
        const int cB = (i0 + k < m) ? ((int)bB[i0 + k] << 8) : 0x1ff;
        s.bch[k] = pack2(cA, cB);
    }
    {
        const int hd = hcol0p(i0, E, O);
        s.hdm = pack2(hd + M - acc.baseA + kBias, hd + M - acc.baseB + kBias);
    }
    s.hbot = pack2(kBias, kBias); s.ebot = s.hbot; s.cch = 0u; s.sh = s.hbot; s.se = s.hbot; s.mx = 0u;

    const int nsteps = n + 64;
    for (int t0 = 0; t0 < nsteps; t0 += 64) {
        const int ln = opaque(lane);
        if (t0 > 0 && t0 < n) {
            // re-anchor both pairs to the band's top row at column t0
            fold_max<R>(s, acc);
            const int nbA = __builtin_amdgcn_readfirstlane(ntA.x), nbB = __builtin_amdgcn_readfirstlane(ntB.x);
            const unsigned pd = pack2(nbA - acc.baseA, nbB - acc.baseB);
            acc.baseA = nbA; acc.baseB = nbB;
#pragma unroll
            for (int k = 0; k < R; ++k) { s.hm[k] = pksub(s.hm[k], pd); s.fb[k] = pksub(s.fb[k], pd); }
            s.hdm = pksub(s.hdm, pd); s.hbot = pksub(s.hbot, pd); s.ebot = pksub(s.ebot, pd);
        }
        s.th = pack2(ntA.x - acc.baseA + kBias, ntB.x - acc.baseB + kBias);
        s.te = pack2(ntA.y - acc.baseA + kBias, ntB.y - acc.baseB + kBias);
        s.ach = pack2(naA << 8, naB << 8);
        const int cn = t0 + 64 + ln;
        if (cn < n) { ntA = rowA[cn]; ntB = rowB[cn]; naA = aA[cn]; naB = aB[cn]; }
        if (t0 >= 64 && t0 + 63 < n) {
#pragma unroll 1
            for (int q = 0; q < 64; ++q) pstep<R, false, PARTIAL>(s, t0 + q, lane, n, nvalid, pe, po, pm, pmx);
        } else {
#pragma unroll 1
            for (int q = 0; q < 64; ++q) pstep<R, true, PARTIAL>(s, t0 + q, lane, n, nvalid, pe, po, pm, pmx);
        }
        const int cs = t0 - 64 + opaque(lane);
        if (!PARTIAL && cs >= 0 && cs < n) {
            rowA[cs] = make_int2((int)(s.sh & 0xffffu) - kBias + acc.baseA, (int)(s.se & 0xffffu) - kBias + acc.baseA);
            rowB[cs] = make_int2((int)(s.sh >> 16) - kBias + acc.baseB, (int)(s.se >> 16) - kBias + acc.baseB);
        }
    }
    fold_max<R>(s, acc);
}

template <int R>
__device__ __forceinline__ void pscore_couple(const uint8_t* aA, const uint8_t* bA, const uint8_t* aB, const uint8_t* bB,
                                              int n, int m, int2* rowA, int2* rowB, int lane, int M, int E, int O,
                                              unsigned pe, unsigned po, unsigned pm, unsigned pmx, int& outA, int& outB)
{
    for (int c = opaque(lane); c < n; c += 64) {          // blockmatrix_init, psa.c:221-226
        const int h = O + (c + 1) * E;
        rowA[c] = make_int2(h, h + O);
        rowB[c] = make_int2(h, h + O);
    }
    wave_mem_sync();
    PairAcc acc; acc.baseA = 0; acc.baseB = 0; acc.mxA = kNegInfP; acc.mxB = kNegInfP;
    constexpr int BAND = 64 * R;
    int r0 = 0;
    for (; r0 + BAND <= m; r0 += BAND) {
        pband<R, false>(aA, aB, n, bA, bB, m, r0, rowA, rowB, acc, lane, M, E, O, pe, po, pm, pmx);
        wave_mem_sync();
    }
    if (r0 < m) pband<R, true>(aA, aB, n, bA, bB, m, r0, rowA, rowB, acc, lane, M, E, O, pe, po, pm, pmx);
    outA = wave_max(acc.mxA);
    outB = wave_max(acc.mxB);
}

template <int R, int WAVES>
__global__ __launch_bounds__(256, WAVES) void psa_score_packed_kernel(PsaPackedArgs P)
{
    const PsaBatchArgs& A = P.b;
    const int lane = threadIdx.x & 63;
    const int64_t slot = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int2* rowA = A.rows + slot * 2 * A.row_stride;
    int2* rowB = rowA + A.row_stride;
    const bool stamp = A.clock_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    if (stamp) { A.clock_out[0] = __builtin_amdgcn_s_memtime(); A.clock_out[1] = __builtin_amdgcn_s_memrealtime(); }
    if (A.clock_out != nullptr && threadIdx.x == 0) A.clock_out[8 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        unsigned int ticket = 0;
        if (lane == 0) ticket = atomicAdd(A.counter, 1u);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if ((int64_t)ticket >= A.packed_couples) break;         // every wave reaches this exit
        if (!psa_couple_is_packed(A.len1, A.len2, ticket)) continue;   // K1 (int32) takes these two pairs
        const int64_t pa = 2 * (int64_t)ticket, pb = pa + 1;
        const int l1a = A.len1[pa], l2a = A.len2[pa], l1b = A.len1[pb], l2b = A.len2[pb];
        const uint8_t* s1a = A.seqs + A.off1[pa]; const uint8_t* s2a = A.seqs + A.off2[pa];
        const uint8_t* s1b = A.seqs + A.off1[pb]; const uint8_t* s2b = A.seqs + A.off2[pb];
        const bool swa = l2a > l1a, swb = l2b > l1b;            // psa.c:194-199
        const uint8_t* aA = swa ? s2a : s1a; const uint8_t* bA = swa ? s1a : s2a;
        const uint8_t* aB = swb ? s2b : s1b; const uint8_t* bB = swb ? s1b : s2b;
        const int n = swa ? l2a : l1a, m = swa ? l1a : l2a;
        int ra, rb;
        pscore_couple<R>(aA, bA, aB, bB, n, m, rowA, rowB, lane, A.M, A.E, A.O, P.pe, P.po, P.pm, P.pmx, ra, rb);
        if (lane == 0) { A.out[pa] = max(ra, -100); A.out[pb] = max(rb, -100); }    // ms floor, psa.c:110,590
        wave_mem_sync();
    }
    if (stamp) { A.clock_out[2] = __builtin_amdgcn_s_memtime(); A.clock_out[3] = __builtin_amdgcn_s_memrealtime(); }
    if (A.clock_out != nullptr && threadIdx.x == 0) A.clock_out[8 + gridDim.x + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

void launch_psa_score_packed(int R, const PsaPackedArgs& args, int blocks, hipStream_t stream)
{
    if (R == 8) hipLaunchKernelGGL((psa_score_packed_kernel<8, 4>), dim3(blocks), dim3(256), 0, stream, args);
    else        hipLaunchKernelGGL((psa_score_packed_kernel<16, 4>), dim3(blocks), dim3(256), 0, stream, args);
}

int psa_packed_blocks_per_cu(int R)
{
    int nb = 0;
    hipError_t e = (R == 8) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psa_score_packed_kernel<8, 4>, 256, 0)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psa_score_packed_kernel<16, 4>, 256, 0);
    if (e != hipSuccess || nb < 1) nb = 1;
    return nb > 8 ? 8 : nb;
}

}  // namespace tsta
