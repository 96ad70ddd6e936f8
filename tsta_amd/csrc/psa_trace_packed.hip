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
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "packed_ops.h"
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

template <int R>
struct PTLane {
    unsigned hl[R], fb[R], bch[R];
    unsigned hd, hbot, ebot, cch, th, te, ach, sh, se, mx, joff;
};

// One step of the lane's R rows for both jobs (see psa_score_packed.hip: pstep for the recurrence and the masks);
// code[k] = per half, the trace byte of row k: nd + 2nf + 4(fge+1) + 16(ege+1) = (nd + 20) + 2nf + 4fge + 16ege with
// fge, ege in {-1,0,1}; the +20 rides on nd so that every partial sum stays a non-negative byte.
template <int R, bool PRED, bool PARTIAL>
__device__ __forceinline__ void ptstep(PTLane<R>& s, const int t, const int lane, const int n, const int nA, const int nB,
                                       const int nvA, const int nvB, const unsigned pe, const unsigned po, const unsigned pm,
                                       const unsigned pmx, const unsigned pf, const PkConst& K, unsigned (&code)[R])
{
    unsigned hu, eu;
    lane_exchange(s.cch, hu, eu, s.sh, s.se, s.ach, s.th, s.te, s.hbot, s.ebot);
    const bool act = !PRED || ((unsigned)(t - lane) < (unsigned)n);
    unsigned d = s.hd;
    unsigned h = 0;
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const unsigned e = eu - pe;                      // E' = Ebar'(up) - |E|
        const unsigned f = s.fb[k] - pf;                 // F' = Fbar'(left) - (|E|+M)
        const unsigned tt = pkmin(s.cch ^ s.bch[k], pmx);
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
        unsigned hv = h;
        if (PARTIAL) hv &= ((k < nvA) ? 0xffffu : 0u) | ((k < nvB) ? 0xffff0000u : 0u);
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
    unsigned cand = m + s.joff;                          // running max in the H frame (see pstep)
    if (PRED) {
        const unsigned c = (unsigned)(t - lane);
        cand &= ((c < (unsigned)nA) ? 0xffffu : 0u) | ((c < (unsigned)nB) ? 0xffff0000u : 0u);
    }
    if (PARTIAL) cand &= ((nvA > 0) ? 0xffffu : 0u) | ((nvB > 0) ? 0xffff0000u : 0u);
    s.mx = pkmax(s.mx, cand);
    s.joff += pm;
}

// 64 steps; the trace bytes of every 16/R steps leave as one uint4 per lane and job
template <int R, bool PRED, bool PARTIAL>
__device__ __forceinline__ void ptblock(PTLane<R>& s, const int t0, const int lane, const int n, const int nA, const int nB,
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
            ptstep<R, PRED, PARTIAL>(s, t0 + g * SPB + q, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf, K, code);
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

template <int R>
__global__ __launch_bounds__(256) void psa_trace_packed_tile_kernel(PsaTracePackedArgs A)
{
    constexpr int NS = 2 * R + 8;
    constexpr int SPB = TraceCfg<R>::SPB;
    const int lane = threadIdx.x & 63;
    const int g = A.g_lo + blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (g >= A.g_hi) return;
    int lo = 0, hi = A.n_couples - 1;           // couple = last one whose first band is <= g
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (A.couples[mid].band0 <= g) lo = mid; else hi = mid - 1; }
    const TraceCouple& C = A.couples[lo];
    const int w = g - C.band0;
    if (w >= C.W) return;
    const int shift = w * (A.T + 64);
    int s0 = A.d * A.T - shift, s1 = (A.d + 1) * A.T - shift;
    s0 = max(0, min(s0, C.S_total));
    s1 = max(0, min(s1, C.S_total));
    if (s0 >= s1) return;

    const TraceJob& JA = A.jobs[C.ja];
    const TraceJob& JB = A.jobs[C.jb];
    const int nA = JA.n, mA = JA.m, nB = JB.n, mB = JB.m;
    const int n = max(nA, nB), nmin = min(nA, nB), mmin = min(mA, mB);
    const int M = A.M, E = A.E, O = A.O;
    const uint8_t* __restrict__ aA = JA.a; const uint8_t* __restrict__ bA = JA.b;
    const uint8_t* __restrict__ aB = JB.a; const uint8_t* __restrict__ bB = JB.b;
    const unsigned pe = in_vgpr(A.pe), po = in_vgpr(A.po), pm = in_vgpr(A.pm), pmx = in_vgpr(A.pmx);
    PkConst K;
    K.one = in_vgpr(0x00010001u); K.mone = in_vgpr(0xffffffffu); K.k2 = in_vgpr(0x00020002u); K.k4 = in_vgpr(0x00040004u);
    K.k16 = in_vgpr(0x00100010u); K.k20 = in_vgpr(0x00140014u); K.k21 = in_vgpr(0x00150015u);
    const unsigned pf = pe + pm;
    constexpr int BAND = 64 * R;
    const int band_row0 = w * BAND;
    const bool partial = band_row0 + BAND > mmin;
    const int i0 = band_row0 + lane * R;
    const int nvA = mA - i0, nvB = mB - i0;
    int* st = C.state + (size_t)w * NS * 64 + lane;

    PTLane<R> s;
    PairAcc acc;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int cA = (i0 + k < mA) ? ((int)bA[i0 + k] << 8) : 0x1ff;
        const int cB = (i0 + k < mB) ? ((int)bB[i0 + k] << 8) : 0x1ff;
        s.bch[k] = pack2(cA, cB);
    }
    const int2* __restrict__ topA = JA.rows + (size_t)w * JA.row_stride;
    const int2* __restrict__ topB = JB.rows + (size_t)w * JB.row_stride;
    int2* __restrict__ botA = JA.rows + (size_t)(w + 1) * JA.row_stride;
    int2* __restrict__ botB = JB.rows + (size_t)(w + 1) * JB.row_stride;
    uint8_t* tbandA = JA.trace + (size_t)w * JA.band_bytes;
    uint8_t* tbandB = JB.trace + (size_t)w * JB.band_bytes;

    if (s0 == 0) {
        // anchor of the H' frame: the band's top row at the first column (H' = H - (column index + 1) * M)
        acc.baseA = __builtin_amdgcn_readfirstlane(topA[0].x) - M;
        acc.baseB = __builtin_amdgcn_readfirstlane(topB[0].x) - M;
        acc.mxA = kNegInfP; acc.mxB = kNegInfP;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int hk = hcol0p(i0 + 1 + k, E, O);    // column 0: H' = H
            s.hl[k] = pack2(hk - acc.baseA + kBias, hk - acc.baseB + kBias);
            s.fb[k] = pack2(hk + O - acc.baseA + kBias, hk + O - acc.baseB + kBias);
        }
        const int hd = hcol0p(i0, E, O);
        s.hd = pack2(hd - acc.baseA + kBias, hd - acc.baseB + kBias);
        s.hbot = pack2(kBias, kBias); s.ebot = s.hbot; s.cch = 0u;
    } else {
#pragma unroll
        for (int k = 0; k < R; ++k) { s.hl[k] = (unsigned)st[(2 * k) * 64]; s.fb[k] = (unsigned)st[(2 * k + 1) * 64]; }
        s.hd = (unsigned)st[(2 * R) * 64]; s.hbot = (unsigned)st[(2 * R + 1) * 64]; s.ebot = (unsigned)st[(2 * R + 2) * 64];
        s.cch = (unsigned)st[(2 * R + 3) * 64];
        acc.mxA = st[(2 * R + 4) * 64]; acc.mxB = st[(2 * R + 5) * 64];
        acc.baseA = __builtin_amdgcn_readfirstlane(st[(2 * R + 6) * 64]); acc.baseB = __builtin_amdgcn_readfirstlane(st[(2 * R + 7) * 64]);
    }
    s.sh = pack2(kBias, kBias); s.se = s.sh; s.mx = 0u;
    acc.ancA = 0; acc.ancB = 0;

    // top-row entries and column characters of a block are loaded one block ahead (within the tile: everything this
    // launch may read of the band above was written by earlier launches)
    int2 ntA = make_int2(0, 0), ntB = make_int2(0, 0);
    int naA = 0, naB = 0;
    { const int c = s0 + opaque(lane); if (c < nA) { ntA = topA[c]; naA = aA[c]; } if (c < nB) { ntB = topB[c]; naB = aB[c]; } }
    for (int t0 = s0; t0 < s1; t0 += 64) {
        const int ln = opaque(lane);
        if (t0 > 0 && t0 < n) {
            // re-anchor both jobs to the band's top row at column t0 (a job whose columns are exhausted keeps its anchor)
            fold_mx(s.mx, acc);
            const int nbA = (t0 < nA) ? __builtin_amdgcn_readfirstlane(ntA.x) - (t0 + 1) * M : acc.baseA;
            const int nbB = (t0 < nB) ? __builtin_amdgcn_readfirstlane(ntB.x) - (t0 + 1) * M : acc.baseB;
            const unsigned pd = pack2(nbA - acc.baseA, nbB - acc.baseB);
            acc.baseA = nbA; acc.baseB = nbB;
#pragma unroll
            for (int k = 0; k < R; ++k) { s.hl[k] = pksub(s.hl[k], pd); s.fb[k] = pksub(s.fb[k], pd); }
            s.hd = pksub(s.hd, pd); s.hbot = pksub(s.hbot, pd); s.ebot = pksub(s.ebot, pd);
        } else if (t0 >= n) {
            fold_mx(s.mx, acc);                    // drain block keeps the previous anchor
        }
        acc.ancA = acc.baseA + (t0 - 62) * M;      // running-max anchor of this block: column t0-63 in the H frame
        acc.ancB = acc.baseB + (t0 - 62) * M;
        s.joff = pack2((63 - ln) * M, (63 - ln) * M);
        {
            const int sh_m = (t0 + ln + 1) * M;    // column shift of this lane's top-row entry
            s.th = pack2(ntA.x - sh_m - acc.baseA + kBias, ntB.x - sh_m - acc.baseB + kBias);
            s.te = pack2(ntA.y - sh_m - acc.baseA + kBias, ntB.y - sh_m - acc.baseB + kBias);
        }
        s.ach = pack2(naA << 8, naB << 8);
        if (t0 + 64 < s1) {
            const int cn = t0 + 64 + ln;
            ntA = make_int2(0, 0); ntB = make_int2(0, 0); naA = 0; naB = 0;
            if (cn < nA) { ntA = topA[cn]; naA = aA[cn]; }
            if (cn < nB) { ntB = topB[cn]; naB = aB[cn]; }
        }
        uint4* tqA = reinterpret_cast<uint4*>(tbandA) + ((size_t)(t0 / SPB) * 64 + ln);
        uint4* tqB = reinterpret_cast<uint4*>(tbandB) + ((size_t)(t0 / SPB) * 64 + ln);
        const bool steady = t0 >= 64 && t0 + 63 < nmin;
        if (steady && !partial)      ptblock<R, false, false>(s, t0, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf, K, tqA, tqB);
        else if (steady)             ptblock<R, false, true>(s, t0, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf, K, tqA, tqB);
        else                         ptblock<R, true, true>(s, t0, lane, n, nA, nB, nvA, nvB, pe, po, pm, pmx, pf, K, tqA, tqB);
        const int cs = t0 - 64 + ln;
        if (cs >= 0 && cs < n) {
            const int sh_m = (cs + 1) * M;
            botA[cs] = make_int2((int)(s.sh & 0xffffu) - kBias + acc.baseA + sh_m, (int)(s.se & 0xffffu) - kBias + acc.baseA + sh_m);
            botB[cs] = make_int2((int)(s.sh >> 16) - kBias + acc.baseB + sh_m, (int)(s.se >> 16) - kBias + acc.baseB + sh_m);
        }
    }
    fold_mx(s.mx, acc);
#pragma unroll
    for (int k = 0; k < R; ++k) { st[(2 * k) * 64] = (int)s.hl[k]; st[(2 * k + 1) * 64] = (int)s.fb[k]; }
    st[(2 * R) * 64] = (int)s.hd; st[(2 * R + 1) * 64] = (int)s.hbot; st[(2 * R + 2) * 64] = (int)s.ebot;
    st[(2 * R + 3) * 64] = (int)s.cch;
    st[(2 * R + 4) * 64] = acc.mxA; st[(2 * R + 5) * 64] = acc.mxB;
    st[(2 * R + 6) * 64] = acc.baseA; st[(2 * R + 7) * 64] = acc.baseB;
}

// per couple (blockIdx.x): out[0] of both jobs = max over all lanes' running maxima, floored at -100 (psa.c:590)
template <int R>
__global__ void psa_trace_packed_finish_kernel(const TraceJob* jobs, const TraceCouple* couples)
{
    constexpr int NS = 2 * R + 8;
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

void launch_psa_trace_packed_tile(const PsaTracePackedArgs& args, int blocks, hipStream_t stream)
{
    hipLaunchKernelGGL((psa_trace_packed_tile_kernel<kPackedTraceR>), dim3(blocks), dim3(256), 0, stream, args);
}

void launch_psa_trace_packed_finish(const TraceJob* jobs, const TraceCouple* couples, int n_couples, hipStream_t stream)
{
    hipLaunchKernelGGL((psa_trace_packed_finish_kernel<kPackedTraceR>), dim3(n_couples), dim3(256), 0, stream, jobs, couples);
}

}  // namespace tsta
