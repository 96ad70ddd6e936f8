// psa_trace.hip — K2 (fill with packed trace) + K3 (traceback walker) for ONE pair: tsta_psa_align.
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
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/tsta_engine.h"
#include "engine.h"
#include "wave_ops.h"

namespace tsta {

constexpr int kNegInfT = -(1 << 30);

struct PsaTraceArgs {
    const uint8_t* a; int n;        // columns (longer sequence)
    const uint8_t* b; int m;        // rows
    int2*     rows;                 // [W+1][row_stride]: rows[w] = top row of band w ({H, Ebar} per column)
    int64_t   row_stride;
    int*      state;                // [W][NS][64] per-lane DP state between tiles
    uint8_t*  trace;                // [W][band_bytes]
    int64_t   band_bytes;
    int       W, T, d, S_total;     // bands, tile length (steps), launch index, total steps per band (mult. of 64)
    int       w_lo, w_hi;           // active band range of this launch
    int       M, X, E, O;
};

template <int R> struct TraceCfg {
    static constexpr int SPB = 16 / R;          // steps per 16-byte trace group
    static constexpr int NS = 2 * R + 5;        // state ints per lane
};

__device__ __forceinline__ int hcol0t(int i, int E, int O) { return i == 0 ? 0 : O + i * E; }

template <int R>
struct TLane {
    int hl[R], fb[R], bch[R];
    int hd, hbot, ebot, cch, mx;
    int th, te, ach, sh, se;
};

// one DP step of the lane's R rows; returns the R trace bytes packed little-endian into `tb` (R<=4: one dword,
// R==8: two dwords)
template <int R, bool PRED, bool TRACE>
__device__ __forceinline__ void trace_step(TLane<R>& s, const int t, const int lane, const int n, const int nvalid,
                                           const int M, const int X, const int E, const int O, unsigned (&tb)[(R + 3) / 4])
{
    s.cch = wave_shr1(s.ach, s.cch);
    const int hu = wave_shr1(s.th, s.hbot);
    int eu = wave_shr1(s.te, s.ebot);
    s.sh = wave_shl1(s.hbot, s.sh);
    s.se = wave_shl1(s.ebot, s.se);
    s.th = wave_shl1(s.th, s.th);
    s.te = wave_shl1(s.te, s.te);
    s.ach = wave_shl1(s.ach, s.ach);

    const bool act = !PRED || ((unsigned)(t - lane) < (unsigned)n);
    int d = s.hd;
    int h = 0;
#pragma unroll
    for (int q = 0; q < (R + 3) / 4; ++q) tb[q] = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int e = eu + E;
        const int f = s.fb[k] + E;
        const int dg = d + (s.cch == s.bch[k] ? M : X);
        h = max3i(dg, e, f);
        const int g = h + O;
        if (TRACE) {                                        // trace code (see file header)
            const int nd = min(h - dg, 1);
            const int nf = min(h - f, 1);
            const int fge = min(max(g - f, -1), 1) + 1;
            const int ege = min(max(g - e, -1), 1) + 1;
            const unsigned code = (unsigned)(nd | (nf << 1) | (fge << 2) | (ege << 4));
            tb[k / 4] |= code << (8 * (k % 4));
        }
        const int eb = max(e, g);
        const int fbn = max(f, g);
        d = s.hl[k];
        int hv = (k < nvalid) ? h : kNegInfT;
        if (PRED) {
            s.hl[k] = act ? h : s.hl[k];
            s.fb[k] = act ? fbn : s.fb[k];
            hv = act ? hv : kNegInfT;
        } else {
            s.hl[k] = h;
            s.fb[k] = fbn;
        }
        s.mx = max(s.mx, hv);
        eu = eb;
    }
    s.hbot = h;
    s.ebot = eu;
    if (PRED) s.hd = act ? hu : s.hd; else s.hd = hu;
}

template <int R, bool PRED, bool TRACE>
__device__ __forceinline__ void trace_block(TLane<R>& s, const int t0, const int lane, const int n, const int nvalid,
                                            const int M, const int X, const int E, const int O, uint4* __restrict__ tq)
{
    constexpr int SPB = TraceCfg<R>::SPB;
    // tq -> trace group of step t0 for this lane (stride 64 uint4 per group)
#pragma unroll 1
    for (int g = 0; g < 64 / SPB; ++g) {
        unsigned acc[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < SPB; ++q) {
            unsigned tb[(R + 3) / 4];
            trace_step<R, PRED, TRACE>(s, t0 + g * SPB + q, lane, n, nvalid, M, X, E, O, tb);
            if constexpr (!TRACE) { (void)acc; }
            else if constexpr (R == 8) { acc[2 * q] = tb[0]; acc[2 * q + 1] = tb[1]; }
            else if constexpr (R == 4) { acc[q] = tb[0]; }
            else { acc[(q * R) / 4] |= tb[0] << (8 * ((q * R) % 4)); }
        }
        if (TRACE) tq[(size_t)g * 64] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
    }
}

template <int R, bool TRACE>
__global__ __launch_bounds__(256) void psa_trace_tile_kernel(PsaTraceArgs A)
{
    constexpr int NS = TraceCfg<R>::NS;
    constexpr int SPB = TraceCfg<R>::SPB;
    const int lane = threadIdx.x & 63;
    const int w = A.w_lo + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= A.w_hi) return;
    const int shift = w * (A.T + 64);
    int s0 = A.d * A.T - shift, s1 = (A.d + 1) * A.T - shift;
    s0 = max(0, min(s0, A.S_total));
    s1 = max(0, min(s1, A.S_total));
    if (s0 >= s1) return;

    const int n = A.n, m = A.m, M = A.M, X = A.X, E = A.E, O = A.O;
    const int i0 = w * 64 * R + lane * R;
    const int nvalid = m - i0;
    int* st = A.state + (size_t)w * NS * 64 + lane;
    TLane<R> s;
#pragma unroll
    for (int k = 0; k < R; ++k) s.bch[k] = (i0 + k < m) ? (int)A.b[i0 + k] : 0x100;
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

    const int2* __restrict__ top = A.rows + (size_t)w * A.row_stride;
    int2* __restrict__ bot = A.rows + (size_t)(w + 1) * A.row_stride;
    uint8_t* tband = A.trace + (size_t)w * A.band_bytes;

    for (int t0 = s0; t0 < s1; t0 += 64) {
        const int c = t0 + lane;
        int2 nt = make_int2(0, 0); int na = 0;
        if (c < n) { nt = top[c]; na = A.a[c]; }
        s.th = nt.x; s.te = nt.y; s.ach = na;
        uint4* tq = reinterpret_cast<uint4*>(tband) + ((size_t)(t0 / SPB) * 64 + lane);
        if (t0 >= 64 && t0 + 63 < n) trace_block<R, false, TRACE>(s, t0, lane, n, nvalid, M, X, E, O, tq);
        else                         trace_block<R, true, TRACE>(s, t0, lane, n, nvalid, M, X, E, O, tq);
        const int cs = t0 - 64 + lane;
        if (cs >= 0 && cs < n) bot[cs] = make_int2(s.sh, s.se);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) { st[(2 * k) * 64] = s.hl[k]; st[(2 * k + 1) * 64] = s.fb[k]; }
    st[(2 * R) * 64] = s.hd; st[(2 * R + 1) * 64] = s.hbot; st[(2 * R + 2) * 64] = s.ebot;
    st[(2 * R + 3) * 64] = s.cch; st[(2 * R + 4) * 64] = s.mx;
}

// boundary row of the matrix (psa.c:221-226) into rows[0]
__global__ void psa_trace_init_kernel(int2* row0, int n, int E, int O)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) { const int h = O + (c + 1) * E; row0[c] = make_int2(h, h + O); }
}

// out[0] = max over all lanes' running maxima (floored at -100, psa.c:590), out[1] = H(m,n)
template <int R>
__global__ void psa_trace_finish_kernel(const int* state, int W, int m, int* out)
{
    constexpr int NS = TraceCfg<R>::NS;
    int mx = kNegInfT;
    for (int idx = threadIdx.x; idx < W * 64; idx += blockDim.x) {
        const int w = idx / 64, lane = idx % 64;
        mx = max(mx, state[(size_t)w * NS * 64 + (2 * R + 4) * 64 + lane]);
    }
    __shared__ int red[256];
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = max(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) {
        out[0] = max(red[0], -100);
        const int rr = (m - 1) % (64 * R), w = (m - 1) / (64 * R);
        out[1] = state[(size_t)w * NS * 64 + (2 * (rr % R)) * 64 + rr / R];
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

template <int R, int NQ>
__global__ __launch_bounds__(64) void psa_walk_kernel(WalkArgs A)
{
    constexpr int SPB = TraceCfg<R>::SPB;
    constexpr int BAND = 64 * R;
    __shared__ uint4 win[NQ * 64];
    const int lane = threadIdx.x;
    int i = A.m, j = A.n;            // 1-based current cell (wave-uniform)
    int forced = -1;
    int cnt = 0;
    while (i >= 1 && j >= 1) {
        // stage the window that ends at the current cell's step group (one band, NQ step groups of 1 KiB)
        const int w = (i - 1) / BAND;
        const int ln = ((i - 1) % BAND) / R;
        const int qhi = (j - 1 + ln) / SPB;
        const int qlo = max(0, qhi - (NQ - 1));
        const uint4* src = reinterpret_cast<const uint4*>(A.trace + (size_t)w * A.band_bytes) + (size_t)qlo * 64;
        const int nq = qhi - qlo + 1;
        {
            // all loads of the window in flight at once (the trace is cold in HBM: ~2 us per dependent round trip)
            uint4 tmp[NQ];
#pragma unroll
            for (int x = 0; x < NQ; ++x) tmp[x] = (x < nq) ? src[x * 64 + lane] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int x = 0; x < NQ; ++x) win[x * 64 + lane] = tmp[x];
        }
        __syncthreads();
        const uint8_t* wb = reinterpret_cast<const uint8_t*>(win);
        // trace byte of cell (ci,cj), 1-based; *inwin tells whether it came from the staged window
        auto code = [&](int ci, int cj) -> unsigned {
            const int cw = (ci - 1) / BAND, rr = (ci - 1) % BAND, cl = rr / R, ck = rr % R;
            const int ct = cj - 1 + cl, cq = ct / SPB;
            const size_t off = ((size_t)cq * 64 + cl) * 16 + (ct % SPB) * R + ck;
            if (cw == w && cq >= qlo && cq <= qhi) return wb[off - (size_t)qlo * 64 * 16];
            return A.trace[(size_t)cw * A.band_bytes + off];
        };
        for (;;) {
            if (i < 1 || j < 1) break;
            {
                const int cw = (i - 1) / BAND, cl = ((i - 1) % BAND) / R;
                const int cq = (j - 1 + cl) / SPB;
                if (cw != w || cq < qlo) break;                  // left the window: restage
            }
            if (forced < 0) {
                // ---- diagonal run: lane l probes cell (i-l, j-l); the run of leading "diag" cells is taken at once
                const int ci = i - lane, cj = j - lane;
                bool isdiag = false;
                if (ci >= 1 && cj >= 1) {
                    const int cw = (ci - 1) / BAND, rr = (ci - 1) % BAND, cl = rr / R, ck = rr % R;
                    const int ct = cj - 1 + cl, cq = ct / SPB;
                    if (cw == w && cq >= qlo) {
                        const size_t off = ((size_t)(cq - qlo) * 64 + cl) * 16 + (ct % SPB) * R + ck;
                        isdiag = !(wb[off] & 1u);                 // psa.c:443: back == 1
                    }
                }
                const unsigned long long mask = __ballot(isdiag);
                const int run = (~mask == 0ull) ? 64 : __builtin_ctzll(~mask);
                if (run > 0) {
                    if (lane < run) A.ops[cnt + lane] = 1;
                    i -= run; j -= run; cnt += run;
                    continue;
                }
            }
            // ---- one non-diagonal (or forced) step, computed identically by every lane
            const unsigned c = __builtin_amdgcn_readfirstlane(code(i, j));
            int k;
            if (forced >= 0) { k = forced; forced = -1; }
            else k = !(c & 1u) ? 1 : (!(c & 2u) ? 0 : 2);         // psa.c:443-455: diag > left > up
            if (lane == 0) A.ops[cnt] = (uint8_t)k;
            ++cnt;
            if (k == 1) { --i; --j; }
            else if (k == 0) {
                // psa.c:450-452: continue left if F(i,j) was an extension, or it was opened and the cell to the
                // left is "opened & tied"
                if (j - 1 >= 1) {
                    const unsigned cl1 = __builtin_amdgcn_readfirstlane(code(i, j - 1));
                    const bool fopen_ij = ((cl1 >> 2) & 3u) >= 1u;                       // Fopen(i,j) stored at (i,j-1)
                    const bool ftie_l = ((cl1 >> 2) & 3u) == 1u;                         // Ftie(i,j-1)
                    bool fopen_l = true;                                                 // Fopen(i,j-1)
                    if (j - 1 != 1) fopen_l = ((__builtin_amdgcn_readfirstlane(code(i, j - 2)) >> 2) & 3u) >= 1u;
                    if (!fopen_ij || (fopen_l && ftie_l)) forced = 0;
                }
                --j;
            } else {
                if (i - 1 >= 1) {                                                       // psa.c:457-459
                    const unsigned cu1 = __builtin_amdgcn_readfirstlane(code(i - 1, j));
                    const bool eopen_ij = ((cu1 >> 4) & 3u) >= 1u;                       // Eopen(i,j) stored at (i-1,j)
                    const bool etie_u = ((cu1 >> 4) & 3u) == 1u;                         // Etie(i-1,j)
                    bool eopen_u = true;                                                 // Eopen(i-1,j)
                    if (i - 1 != 1) eopen_u = ((__builtin_amdgcn_readfirstlane(code(i - 2, j)) >> 4) & 3u) >= 1u;
                    if (!eopen_ij || (eopen_u && etie_u)) forced = 2;
                }
                --i;
            }
        }
        __syncthreads();
    }
    if (lane == 0) { A.out[2] = cnt; A.out[3] = i; A.out[4] = j; }
}

// TRACE = true: fill with packed trace + on-device walk (tsta_psa_align).  TRACE = false: the same pipelined fill of
// ONE pair over many waves, score only (a single long pair would otherwise occupy one wave of the batch kernel).
template <int R, bool TRACE>
static int run_align(Engine* e, const tsta_params* p, const uint8_t* a, int n, const uint8_t* b, int m,
                     int32_t* maxscore, int32_t* end_score, std::vector<uint8_t>& ops, int& fi, int& fj)
{
    constexpr int NS = TraceCfg<R>::NS;
    const int T = (e->opt_trace_tile >= 64 && e->opt_trace_tile % 64 == 0) ? e->opt_trace_tile : 64;   // knob "trace_tile"
    const int W = (m + 64 * R - 1) / (64 * R);
    const int S_total = (n + 64 + 63) / 64 * 64;
    const int64_t row_stride = S_total + 64;
    const int64_t band_bytes = (int64_t)S_total * 64 * R;
    hipStream_t s = e->stream;
    int rc;
    if ((rc = e->reserve(e->in_seqs, (size_t)n + m + 64, "hipMalloc(seqs)")) ||
        (rc = e->reserve(e->rows, (size_t)(W + 1) * row_stride * sizeof(int2), "hipMalloc(rows)")) ||
        (rc = e->reserve(e->scratch, (size_t)W * NS * 64 * sizeof(int), "hipMalloc(state)")) ||
        (TRACE && (rc = e->reserve(e->trace, (size_t)W * band_bytes, "hipMalloc(trace)"))) ||
        (TRACE && (rc = e->reserve(e->scratch2, (size_t)n + m + 64, "hipMalloc(ops)"))))
        return rc;
    uint8_t* d_a = reinterpret_cast<uint8_t*>(e->in_seqs.p);
    uint8_t* d_b = d_a + n;
    int* d_out = reinterpret_cast<int*>(e->counter.p);
    if ((rc = e->check(hipMemcpyAsync(d_a, a, n, hipMemcpyHostToDevice, s), "H2D a")) ||
        (rc = e->check(hipMemcpyAsync(d_b, b, m, hipMemcpyHostToDevice, s), "H2D b")))
        return rc;

    PsaTraceArgs A;
    A.a = d_a; A.n = n; A.b = d_b; A.m = m;
    A.rows = reinterpret_cast<int2*>(e->rows.p); A.row_stride = row_stride;
    A.state = reinterpret_cast<int*>(e->scratch.p);
    A.trace = reinterpret_cast<uint8_t*>(e->trace.p); A.band_bytes = band_bytes;
    A.W = W; A.T = T; A.S_total = S_total; A.M = p->M; A.X = p->X; A.E = p->E; A.O = p->O;

    e->timing_begin(s);
    hipLaunchKernelGGL(psa_trace_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, A.rows, n, p->E, p->O);
    int launches = 1;
    // band w may run steps below S_w(d) = (d+1)T - w(T+64) after launch d
    const int64_t D = ((int64_t)S_total + (int64_t)(W - 1) * (T + 64) + T - 1) / T;
    for (int64_t d = 0; d < D; ++d) {
        // active bands: S_w(d) > 0  and  S_w(d-1) < S_total
        int64_t w_hi = ((d + 1) * T + (T + 64) - 1) / (T + 64);           // first w with S_w(d) <= 0
        int64_t w_lo = (d * T - S_total) / (T + 64);                       // conservative lower bound
        if (d * T - S_total < 0) w_lo = 0;
        w_lo = std::max<int64_t>(0, w_lo); w_hi = std::min<int64_t>(W, w_hi);
        if (w_lo >= w_hi) continue;
        A.d = (int)d; A.w_lo = (int)w_lo; A.w_hi = (int)w_hi;
        const int blocks = (int)((w_hi - w_lo + 3) / 4);
        hipLaunchKernelGGL((psa_trace_tile_kernel<R, TRACE>), dim3(blocks), dim3(256), 0, s, A);
        ++launches;
    }
    hipLaunchKernelGGL(psa_trace_finish_kernel<R>, dim3(1), dim3(256), 0, s, A.state, W, m, d_out);
    WalkArgs Wk;
    Wk.trace = A.trace; Wk.band_bytes = band_bytes; Wk.n = n; Wk.m = m;
    Wk.ops = reinterpret_cast<uint8_t*>(e->scratch2.p); Wk.out = d_out;
    if (TRACE) hipLaunchKernelGGL((psa_walk_kernel<R, 32>), dim3(1), dim3(64), 0, s, Wk);
    launches += 2;
    e->n_launches = launches;
    e->timing_end(s);
    if ((rc = e->check(hipGetLastError(), "psa_align launches"))) return rc;
    int h_out[8];
    if ((rc = e->check(hipMemcpyAsync(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost, s), "D2H out")) ||
        (rc = e->check(hipStreamSynchronize(s), "psa_align sync")))
        return rc;
    *maxscore = h_out[0]; *end_score = h_out[1];
    if (!TRACE) return TSTA_OK;
    const int cnt = h_out[2]; fi = h_out[3]; fj = h_out[4];
    if (cnt < 0 || cnt > n + m) { e->set_error("walker returned %d ops", cnt); return TSTA_ERR_INTERNAL; }
    ops.resize(cnt);
    if (cnt && (rc = e->check(hipMemcpy(ops.data(), Wk.ops, cnt, hipMemcpyDeviceToHost), "D2H ops"))) return rc;
    return TSTA_OK;
}

// Score-only fill of ONE oriented pair (a = columns, b = rows; host pointers) spread over the whole chip.
int psa_score_single_pair(Engine* e, const tsta_params* p, const uint8_t* a, int n, const uint8_t* b, int m, int32_t* maxscore)
{
    std::vector<uint8_t> ops;
    int fi = 0, fj = 0;
    int32_t es = 0;
    return run_align<4, false>(e, p, a, n, b, m, maxscore, &es, ops, fi, fj);
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
    const uint8_t* a = swap ? s2 : s1; const int n = (int)(swap ? l2 : l1);
    const uint8_t* b = swap ? s1 : s2; const int m = (int)(swap ? l1 : l2);

    std::vector<uint8_t> ops;
    int fi = m, fj = n;
    int32_t ms = -100, es = 0;
    if (m > 0 && n > 0) {
        // rows per lane: small bands expose more concurrent bands of the single pair
        const int R_env = e->opt_trace_rows;                                                    // knob "trace_rows"
        const int Rsel = (R_env == 2 || R_env == 4) ? R_env : 2;   // measured on 100 kbp x 100 kbp: R=2,T=64 40 ms; R=4,T=128 50 ms
        if (Rsel == 4) rc = run_align<4, true>(e, p, a, n, b, m, &ms, &es, ops, fi, fj);
        else           rc = run_align<2, true>(e, p, a, n, b, m, &ms, &es, ops, fi, fj);
        if (rc != TSTA_OK) return rc;
    }
    // psa.c:464-516: leftover prefix of a against gaps, leftover prefix of b against gaps, then reverse
    const int64_t len = (int64_t)ops.size() + fi + fj;
    char* ra = (char*)malloc((size_t)len + 1);
    char* rb = (char*)malloc((size_t)len + 1);
    if (!ra || !rb) { free(ra); free(rb); return TSTA_ERR_OOM; }
    ra[len] = rb[len] = '\0';
    int64_t q = len;
    int i = m, j = n;
    for (size_t x = 0; x < ops.size(); ++x) {
        const int k = ops[x];
        if (k == 1)      { ra[q - 1] = (char)a[j - 1]; rb[q - 1] = (char)b[i - 1]; --i; --j; }
        else if (k == 0) { ra[q - 1] = (char)a[j - 1]; rb[q - 1] = '-'; --j; }
        else             { ra[q - 1] = '-'; rb[q - 1] = (char)b[i - 1]; --i; }
        --q;
    }
    if (i != fi || j != fj) { free(ra); free(rb); e->set_error("walker/ops mismatch"); return TSTA_ERR_INTERNAL; }
    while (j >= 1) { ra[q - 1] = (char)a[j - 1]; rb[q - 1] = '-'; --j; --q; }
    while (i >= 1) { rb[q - 1] = (char)b[i - 1]; ra[q - 1] = '-'; --i; --q; }
    *maxscore = ms; *row_a = ra; *row_b = rb; *out_len = len;
    return TSTA_OK;
}
