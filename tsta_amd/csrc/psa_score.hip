// psa_score.hip — K1: batched score-only global affine-gap alignment for gfx950.
//
// Replaces, for many independent pairs at once, the reference's blockmatrix_init + block-anti-diagonal
// wavefront of block_alignment()/row() + `ms` reduction (psa/psa.c:219-232, 234-433, 593-639).
// The arithmetic is the absolute-score form of row() (psa.c:250-354); results equal the reference's
// printed `maxsorce` (psa.c:666) bit for bit.  This is NOT a translation of the striped int8 code:
//
//   * one wave64 owns one pair (pairs are handed out by an atomic ticket, so the grid is persistent and
//     needs no host-side balancing); lane l owns R consecutive DP rows of the current 64*R-row band;
//   * lanes are skewed by one column: at step t lane l computes column t-l of its R rows, i.e. the wave
//     sweeps one anti-diagonal of Rx1 micro-tiles per step (TSTA's "anti-diagonal method" at lane
//     granularity instead of LxL blocks + a pthread barrier);
//   * the (H, Ebar) pair of the row above comes from lane l-1 by a DPP wave_shr:1 in the VALU; the column
//     character rides the same shift, so there is no query profile to gather;
//   * F (horizontal gap) never leaves the lane: no lazy-F fix-up loop (psa.c:266-274);
//   * the band's bottom row (8 B per column) is handed to the next band through an in-place per-wave
//     HBM/L2 row, loaded and stored 64 columns at a time (512 B coalesced), rotated through lanes by DPP.
//
// Per cell: Ebar = max(E, H+O) and Fbar = max(F, H+O) are carried instead of E/F so that the "open"
// candidate H+O is computed once and shared (10 VALU ops per cell, all int32, no memory traffic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "wave_ops.h"

namespace tsta {

constexpr int kNegInf = -(1 << 30);

template <int R>
struct LaneState {
    int hl[R];     // H(row k, previous column)    -> diagonal of row k+1
    int fb[R];     // Fbar(row k, previous column) = max(F, H+O)
    int bch[R];    // row characters
    int hd;        // H(row above the lane's first row, previous column)
    int hbot, ebot;   // this lane's bottom-row outputs of the previous step
    int cch;       // column character of the current step
    int th, te, ach;  // rotating 64-column blocks of the band's top row and of the column sequence
    int sh, se;    // rotating capture of lane 63's bottom-row outputs
    int mx;
};

template <int R, bool PRED, bool PARTIAL>
__device__ __forceinline__ void psa_step(LaneState<R>& s, const int t, const int lane, const int n,
                                         const int nvalid, const int M, const int X, const int E, const int O)
{
    // cross-lane traffic first, on all lanes (DPP must not run under a partial EXEC mask)
    // (one asm block of 8 DPP instructions, wave_ops.h: lane_exchange)
    unsigned hu_u, eu_u;
    lane_exchange(reinterpret_cast<unsigned&>(s.cch), hu_u, eu_u, reinterpret_cast<unsigned&>(s.sh), reinterpret_cast<unsigned&>(s.se),
                  reinterpret_cast<unsigned&>(s.ach), reinterpret_cast<unsigned&>(s.th), reinterpret_cast<unsigned&>(s.te),
                  (unsigned)s.hbot, (unsigned)s.ebot);
    const int hu = (int)hu_u;
    int eu = (int)eu_u;

    const bool act = !PRED || ((unsigned)(t - lane) < (unsigned)n);
    int d = s.hd;
    int h = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int e = eu + E;                                   // psa.c:316-318 (open candidate folded in Ebar)
        const int f = s.fb[k] + E;                              // psa.c:301-305
        const int dg = d + (s.cch == s.bch[k] ? M : X);         // psa.c:252-254
        h = max3i(dg, e, f);                                    // psa.c:289-290
        const int g = h + O;
        const int eb = max(e, g);
        const int fbn = max(f, g);
        d = s.hl[k];
        int hv = h;
        if (PARTIAL) hv = (k < nvalid) ? hv : kNegInf;
        if (PRED) {
            s.hl[k] = act ? h : s.hl[k];
            s.fb[k] = act ? fbn : s.fb[k];
            hv = act ? hv : kNegInf;
        } else {
            s.hl[k] = h;
            s.fb[k] = fbn;
        }
        s.mx = max(s.mx, hv);                                   // psa.c:333-353 (running max of H)
        eu = eb;
    }
    s.hbot = h;
    s.ebot = eu;
    if (PRED) s.hd = act ? hu : s.hd; else s.hd = hu;
}

__device__ __forceinline__ int hcol0(int i, int E, int O) { return i == 0 ? 0 : O + i * E; }   // psa.c:221-231

template <int R, bool PARTIAL>
__device__ __forceinline__ void psa_band(const uint8_t* __restrict__ a, const int n, const uint8_t* __restrict__ b,
                                         const int m, const int band_row0, int2* __restrict__ row, int& mx,
                                         const int lane, const int M, const int X, const int E, const int O)
{
    LaneState<R> s;
    const int i0 = band_row0 + lane * R;          // rows i0+1 .. i0+R (1-based) belong to this lane
    const int nvalid = m - i0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int hk = hcol0(i0 + 1 + k, E, O);
        s.hl[k] = hk;
        s.fb[k] = hk + O;                          // F(i,0) = -inf  =>  Fbar(i,0) = H(i,0)+O
        s.bch[k] = (i0 + k < m) ? (int)b[i0 + k] : 0x100;
    }
    s.hd = hcol0(i0, E, O);
    s.hbot = 0; s.ebot = 0; s.cch = 0; s.sh = 0; s.se = 0;
    s.mx = mx;

    int2 nt = make_int2(0, 0);
    int na = 0;
    if (lane < n) { nt = row[lane]; na = a[lane]; }
    const int nsteps = n + 64;                     // n+63 compute steps + 1 to capture lane 63's last output
    for (int t0 = 0; t0 < nsteps; t0 += 64) {
        s.th = nt.x; s.te = nt.y; s.ach = na;
        const int cn = t0 + 64 + lane;
        if (cn < n) { nt = row[cn]; na = a[cn]; }
        if (t0 >= 64 && t0 + 63 < n) {
#pragma unroll 2
            for (int q = 0; q < 64; ++q) psa_step<R, false, PARTIAL>(s, t0 + q, lane, n, nvalid, M, X, E, O);
        } else {
#pragma unroll 1
            for (int q = 0; q < 64; ++q) psa_step<R, true, PARTIAL>(s, t0 + q, lane, n, nvalid, M, X, E, O);
        }
        // lanes now hold the bottom row of columns t0-64 .. t0-1
        const int cs = t0 - 64 + lane;
        if (!PARTIAL && cs >= 0 && cs < n) row[cs] = make_int2(s.sh, s.se);
    }
    mx = s.mx;
}

template <int R>
__device__ __forceinline__ int psa_score_pair(const uint8_t* a, int n, const uint8_t* b, int m, int2* row,
                                              int lane, int M, int X, int E, int O)
{
    // boundary row of the matrix (blockmatrix_init, psa.c:221-226): H(0,j) = O + j*E; no vertical gap is
    // open on the boundary, so Ebar(0,j) = H(0,j) + O.
    for (int c = lane; c < n; c += 64) {
        const int h = O + (c + 1) * E;
        row[c] = make_int2(h, h + O);
    }
    wave_mem_sync();
    int mx = kNegInf;
    constexpr int BAND = 64 * R;
    int r0 = 0;
    for (; r0 + BAND <= m; r0 += BAND) {
        psa_band<R, false>(a, n, b, m, r0, row, mx, lane, M, X, E, O);
        wave_mem_sync();
    }
    if (r0 < m) psa_band<R, true>(a, n, b, m, r0, row, mx, lane, M, X, E, O);
    return wave_max(mx);
}

template <int R>
__global__ __launch_bounds__(256) void psa_score_batch_kernel(PsaBatchArgs A)
{
    const int lane = threadIdx.x & 63;
    const int64_t slot = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int2* row = A.rows + slot * A.row_stride;
    const bool stamp = A.clock_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    if (stamp) { A.clock_out[0] = __builtin_amdgcn_s_memtime(); A.clock_out[1] = __builtin_amdgcn_s_memrealtime(); }
    for (;;) {
        unsigned int ticket = 0;
        if (lane == 0) ticket = atomicAdd(A.counter, 1u);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if ((int64_t)ticket >= A.n_pairs) break;               // every wave reaches this exit
        const int64_t pi = psa_pair_at(A.order, ticket);
        if ((int64_t)(ticket >> 1) < A.packed_couples && psa_couple_is_packed(A.len1, A.len2, A.order, ticket >> 1) &&
            !(A.redo != nullptr && A.redo[pi] != 0))
            continue;                                           // scored by the packed-int16 kernel (unless it gave the pair up)
        const int l1 = A.len1[pi], l2 = A.len2[pi];
        const uint8_t* s1 = A.seqs + A.off1[pi];
        const uint8_t* s2 = A.seqs + A.off2[pi];
        // psa.c:194-199: the strictly longer sequence becomes seq[0] (columns)
        const bool swap = l2 > l1;
        const uint8_t* a = swap ? s2 : s1; const int n = swap ? l2 : l1;
        const uint8_t* b = swap ? s1 : s2; const int m = swap ? l1 : l2;
        int ms = kNegInf;
        if (m > 0 && n > 0) ms = psa_score_pair<R>(a, n, b, m, row, lane, A.M, A.X, A.E, A.O);
        if (lane == 0) A.out[pi] = max(ms, -100);               // ms = MIN (-100) floor, psa.c:110,590
        wave_mem_sync();                                        // row buffer is reused by the next pair
    }
    if (stamp) { A.clock_out[2] = __builtin_amdgcn_s_memtime(); A.clock_out[3] = __builtin_amdgcn_s_memrealtime(); }
}

// DPP semantics self-test (hardware contract the DP kernels rely on).
__global__ void dpp_selftest_kernel(int* out)
{
    const int lane = threadIdx.x;
    const int a = wave_shr1(1000 + lane, lane);     // lane0 -> 1000, lane l -> l-1
    const int b = wave_shl1(2000 + lane, lane);     // lane63 -> 2063, lane l -> l+1
    out[lane] = a;
    out[64 + lane] = b;
    // lane_exchange (v_cndmask_b32_dpp with bound_ctrl) as the packed kernel uses it
    unsigned cch = 100 + lane, hu, eu, sh = 200 + lane, se = 300 + lane, ach = 400 + lane, th = 500 + lane, te = 600 + lane;
    const unsigned hbot = 700 + lane, ebot = 800 + lane;
    asm volatile("s_nop 1");
    lane_exchange(cch, hu, eu, sh, se, ach, th, te, hbot, ebot);
    out[128 + lane] = cch; out[192 + lane] = hu; out[256 + lane] = eu; out[320 + lane] = sh;
    out[384 + lane] = se; out[448 + lane] = ach; out[512 + lane] = th; out[576 + lane] = te;
    // v_perm_b32 byte selectors as the packed kernel's mismatch lookup assumes: 0-3 second operand, 4-7 first, 0x0c -> 0
    out[640 + lane] = (int)__builtin_amdgcn_perm(0x44332211u, 0x88776655u + (unsigned)lane, 0x0c060c01u);
}

void launch_psa_score_batch(int R, const PsaBatchArgs& args, int blocks, hipStream_t stream)
{
    switch (R) {
        case 4:  hipLaunchKernelGGL(psa_score_batch_kernel<4>, dim3(blocks), dim3(256), 0, stream, args); break;
        case 16: hipLaunchKernelGGL(psa_score_batch_kernel<16>, dim3(blocks), dim3(256), 0, stream, args); break;
        default: hipLaunchKernelGGL(psa_score_batch_kernel<8>, dim3(blocks), dim3(256), 0, stream, args); break;
    }
}

int psa_score_blocks_per_cu(int R)
{
    int nb = 0;
    hipError_t e;
    switch (R) {
        case 4:  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psa_score_batch_kernel<4>, 256, 0); break;
        case 16: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psa_score_batch_kernel<16>, 256, 0); break;
        default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, psa_score_batch_kernel<8>, 256, 0); break;
    }
    if (e != hipSuccess || nb < 1) nb = 1;
    return nb > 8 ? 8 : nb;
}

void launch_dpp_selftest(int* d_out, hipStream_t stream)
{
    hipLaunchKernelGGL(dpp_selftest_kernel, dim3(1), dim3(64), 0, stream, d_out);
}

}  // namespace tsta
