// poa_ref8.hip — K4e: the POA fill in the reference's own number representation, for the inputs on which the reference's output
// is a property of that representation rather than of the alignment problem.
//
// The reference (msa/c-t-simd.c:165-599) keeps every row as int8 horizontal differences and tracks, per predecessor, the
// saturating int8 offset of the row above it (with two overflow reservoirs and a cutoff at 60); a block of L = 64 W columns is
// swept as 64 byte lanes of W columns each: a speculative first pass (every lane from a guessed start, WITHOUT the 'N' rule),
// a lazy-F fix-up across the lanes, a second pass from the handed-over lane ends WITH the rule (:461), absolute scores as
// running sums of the second pass's differences.  Inside the documented domain this is the plain affine DP (what K4 / K4d
// compute).  Outside it — a literal 'N' in the sequence, a sequence spanning more blocks than the graph (:194 unguarded), long
// 'N' padding at large -W, predecessor rows further apart than the trackers reach — what the reference prints is whatever this
// arithmetic yields, -W-dependent, and the only way to print the same bytes is to do the same arithmetic.  This kernel does:
// ONE WAVE IS ONE OF THE REFERENCE'S 512-bit VECTORS (lane b = byte b = columns [bW, bW + W) of the block), every vector
// statement becomes a per-lane integer statement with the saturations written out, the byte shift is a lane shift, the three
// scalar loops over the 64 bytes are wave scans.  The test infrastructure holds the same restatement on the CPU, pinned
// against the compiled reference on 200 'N' outputs and on -W 32 runs.
// Mapping: one wave per block of columns, sweeping the nodes in topological order (the reference's anti-diagonal task order is
// a schedule, not a dependency: a block needs, per row, what the block on its left hands over — the row's running sum, the F
// carries row->f0 — and the rows above in its own block); blocks are tickets in start order, hand-over through sc1 stores and
// progress words (handoff.h).  Rows live in HBM as packed int8 (2 B per cell).  Speed is not the point of this path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>

#include "../../include/tsta_engine.h"
#include "engine.h"
#include "handoff.h"
#include "poa_types.h"

namespace tsta {

namespace {

constexpr int kSmin = -120, kLogoLow = -6, kLogoCut = 60, kLogoTop = 100, kFClamp = 125;      // poa.h:8, c-t-simd.c:169-171,413-427

__device__ __forceinline__ int sat8(int x) { return min(127, max(-128, x)); }
__device__ __forceinline__ int adds8(int a, int b) { return sat8(a + b); }
__device__ __forceinline__ int subs8(int a, int b) { return sat8(a - b); }
__device__ __forceinline__ int wrap8(int x) { return (int)(int8_t)x; }
__device__ __forceinline__ int subus8(int a, int b) { const int ua = a & 255, ub = b & 255; return wrap8(ua > ub ? ua - ub : 0); }
__device__ __forceinline__ int byte_of(int w, int i) { return (int)(int8_t)(w >> ((i & 3) * 8)); }
__device__ __forceinline__ int pack4(int a, int b, int c, int d) { return (a & 255) | ((b & 255) << 8) | ((c & 255) << 16) | (int)((unsigned)d << 24); }

// one predecessor's offset tracker in one lane (c-t-simd.c:385-403 = :541-558)
__device__ __forceinline__ void tracker_step(int best, int t, int& v, int& vc1, int& vc2)
{
    int d = subs8(best, t);
    const int vc0 = adds8(v, d);
    d = max(d, 0);
    const int y = subus8(d, subs8(127, v));
    v = adds8(vc0, vc1);
    const int diff = (vc1 == 0) ? 0 : subus8(127, vc0);
    const int old1 = vc1;
    vc1 = adds8(subus8(vc1, diff), vc2);
    const int moved = subus8(vc1, old1);
    vc2 = subus8(adds8(subus8(vc2, diff), y), moved);
}

// an offset that may exceed a byte, split at a block's first column (c-t-simd.c:203-222,236-256,318-337)
__device__ __forceinline__ void split_offset(int te, int& v, int& c1, int& c2)
{
    if (te > kLogoTop) {
        v = kLogoTop;
        if (te - kLogoTop > 127) { c2 = min(te - kLogoTop - 127, 127); c1 = 127; }
        else { c2 = 0; c1 = te - kLogoTop; }
    } else { v = wrap8(te); c2 = 0; c1 = 0; }
}

// the scans over the 64 byte lanes and the one-lane shift as VALU-only DPP (row_shr 1 / 2 / 4 / 8 inside the 16-lane rows, then
// row_bcast:15 / :31 across them) instead of six ds_bpermute round trips each: a block is a lone wave, for which an LDS instruction
// costs far more than an issue slot
__device__ __forceinline__ int dpp_scan_add(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}
__device__ __forceinline__ int dpp_scan_max(int x)
{
    constexpr int kNeg = INT_MIN / 2;
    x = max(x, __builtin_amdgcn_update_dpp(kNeg, x, 0x111, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(kNeg, x, 0x112, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(kNeg, x, 0x114, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(kNeg, x, 0x118, 0xf, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(kNeg, x, 0x142, 0xa, 0xf, false));
    x = max(x, __builtin_amdgcn_update_dpp(kNeg, x, 0x143, 0xc, 0xf, false));
    return x;
}
__device__ __forceinline__ int dpp_up1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138 /*wave_shr:1*/, 0xf, 0xf, false); }
__device__ __forceinline__ int scan_add(int x, int) { return dpp_scan_add(x); }      // inclusive prefix sum / maximum over the lanes
__device__ __forceinline__ int scan_max(int x, int) { return dpp_scan_max(x); }
// sum over the wave (every lane gets lane 63's inclusive prefix)
__device__ __forceinline__ int wave_sum(int x) { return __builtin_amdgcn_readlane(dpp_scan_add(x), 63); }

// tracker_step for a predecessor whose two reservoirs are empty (vc1 = vc2 = 0, the state every row with ONE predecessor starts from,
// c-t-simd.c:339-346): v moves as before, vc1 stays 0, and the new vc2 is y — collected in `spill`; a row in which any lane's y was
// non-zero is run again with the full tracker.
__device__ __forceinline__ void tracker_step_empty(int best, int t, int& v, int& spill)
{
    const int d = subs8(best, t);
    const int vc0 = adds8(v, d);
    // y = subus8(max(d, 0), subs8(127, v)) is non-zero exactly when v + d exceeds 127, i.e. when the add above saturated upwards; any
    // saturation of it (either side) sends the row to the general sweeps
    spill |= (v + d) ^ vc0;
    v = vc0;
}

// The row of a node with ONE predecessor at -W 10, every statement of the three sweeps below specialised for it: the predecessor's
// bytes, the sequence characters and the lane state {v, f} live in registers (reservoirs empty: tracker_step_empty), the W loops are
// unrolled, the scans are DPP.  Same arithmetic, statement for statement (c-t-simd.c:357-565 with pre_num = 1).  Returns false when a
// reservoir would have filled: nothing of the row is final then and the caller runs the general sweeps.
template <int W>
struct RowBytes { int t[(W + 3) / 4], e[(W + 3) / 4]; };

template <int W, bool ALLIN, class Mid>
__device__ __forceinline__ bool row_one_fast(Mid&& mid, const RowBytes<W>& pr, const int rs_lane, const int v0, const int f0v, const int (&chr)[W],
                                             const int base, const int M, const int X, const int E, const int O, const int lane,
                                             RowBytes<W>& out, int& sum, int& f_last,
                                             int8_t* __restrict__ src, int8_t* __restrict__ esrc, int8_t* __restrict__ fsrc, const int n_left)
{
    const int E8 = wrap8(E), OE8 = wrap8(O + E);
    int v = lane == 0 ? v0 : E8, f = lane == 0 ? f0v : kSmin, spill = 0;
    // ---- first pass (:357-404)
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int h = (chr[i] == base) ? M : X;
        int t1 = max(max(f, h), byte_of(pr.e[i >> 2], i));
        t1 = subs8(t1, v);
        if (v > kLogoCut) t1 = OE8;
        const int s1 = max(kSmin, t1);
        const int t = byte_of(pr.t[i >> 2], i);
        const int a = subs8(f, wrap8(t - E8));
        const int c = adds8(v, subs8(adds8(s1, OE8), t));
        f = max(a, c);
        tracker_step_empty(s1, t, v, spill);
    }
    // ---- lazy F across the lanes, every lane's end state to its right neighbour (:406-446)
    {
        const bool mid = lane >= 1 && lane <= 62;
        const int cx = mid ? W * E - rs_lane : 0;
        const int C = dpp_scan_add(cx);
        const int te = C + dpp_scan_max(f - C);
        const int ftn = mid ? min(te, kFClamp) : f;
        const int nv0 = max(subs8(ftn, E8), v);
        const int f_up = dpp_up1(ftn), v_up = dpp_up1(nv0);
        v = lane ? v_up : v0;                                  // adds8(.., nc1 = 0)
        f = lane ? f_up : f0v;
    }
    mid();                                                     // (the caller's deferred publication of the previous row: its stores have long landed)
    // ---- second pass (:447-565)
    sum = 0;
#pragma unroll
    for (int q = 0; q < (W + 3) / 4; ++q) { out.t[q] = 0; out.e[q] = 0; }
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int ch = chr[i];
        const int h = (ch == base) ? M : X;
        const int t = byte_of(pr.t[i >> 2], i), e = byte_of(pr.e[i >> 2], i);
        const int fv = subs8(f, v), eu = subs8(e, v), q = subs8(h, v), ev = subs8(e, t);
        int c = max(max(fv, eu), q);
        if (v > kLogoCut) c = OE8;
        int best = max(kSmin, c);
        const int eumax = max(kSmin, eu);
        if (ch == 'N') best = 0;
        sum = wrap8(sum + best);
        const int open = adds8(best, OE8);
        const bool j_eu = best == eu, j_q = best == q;
        f = adds8(f, E8);
        const int s1 = adds8(open, v);
        const bool m1 = f == s1;
        f = subs8(max(f, s1), t);
        int fs = (fv == OE8) ? 2 : 1;
        const int e2 = subs8(adds8(e, E8), v);
        const int emax = max(kSmin, max(open, e2));
        int es = 1;
        if (eu == eumax) { es = (ev == OE8) ? 43 : 1; if (open == e2) es = -es; }
        tracker_step_empty(best, t, v, spill);
        if (m1) fs = -fs;
        int code = 126;
        if (j_eu) code = 0;
        if (j_q && h == X) code = 84;
        if (j_q && h == M) code = 42;
        out.t[i >> 2] |= (best & 255) << ((i & 3) * 8);
        out.e[i >> 2] |= (emax & 255) << ((i & 3) * 8);
        if (ALLIN || i < n_left) { src[i] = (int8_t)code; esrc[i] = (int8_t)es; fsrc[i] = (int8_t)fs; }   // (ALLIN: the lane's W columns all exist
                                                                                                       //  and the trace arrays are wanted: no branch)
    }
    f_last = __builtin_amdgcn_readlane(f, 63);               // row->f0 (:566-567): lane 63's F
    return __builtin_amdgcn_readfirstlane((int)(__ballot(spill != 0) != 0ull)) == 0;
}

// The row of a node with NP = 2 .. 4 predecessors at -W 10 in the same register form (round 4): the general sweeps of the kernel below
// (c-t-simd.c:290-565), statement for statement, with the per-predecessor lane state {v, vc1, vc2, f}, the predecessors' bytes and
// the per-column temporaries in registers instead of LDS, the W and predecessor loops unrolled, the full tracker (the reservoirs of
// such a row are not empty at its start).  fc_*: the block's first-column values per predecessor (what the kernel keeps in s_v0 ...).
template <int W, int NP, bool ALLIN, class Mid>
__device__ __forceinline__ void row_multi_fast(Mid&& mid, const RowBytes<W> (&pr)[NP], const int (&rs)[NP],
                                               const int (&fc_v0)[NP], const int (&fc_c1)[NP], const int (&fc_c2)[NP], const int (&fc_te)[NP],
                                               const int (&fc_f0)[NP], const int (&chr)[W],
                                               const int base, const int M, const int X, const int E, const int O, const int lane,
                                               RowBytes<W>& out, int& sum, int (&f_last)[NP],
                                               int8_t* __restrict__ src, int8_t* __restrict__ esrc, int8_t* __restrict__ fsrc, const int n_left)
{
    const int E8 = wrap8(E), OE8 = wrap8(O + E);
    int v[NP], c1[NP], c2[NP], f[NP];
    // ---- what every lane starts its first pass with (:290-356)
    {
        int te[NP], m1 = INT_MAX;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int x = W * E - rs[j];
            te[j] = fc_te[j] + dpp_scan_add(x) - x;              // te_j at the start of lane b, under the guess that this row falls by E per column
            m1 = min(m1, te[j]);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            split_offset(te[j] + kLogoLow - m1, v[j], c1[j], c2[j]);
            if (lane == 0) { v[j] = fc_v0[j]; c1[j] = fc_c1[j]; c2[j] = fc_c2[j]; }
            f[j] = lane == 0 ? fc_f0[j] : kSmin;
        }
    }
    // ---- first pass (:357-404)
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int h = (chr[i] == base) ? M : X;
        int s1 = kSmin;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            int t = max(max(f[j], h), byte_of(pr[j].e[i >> 2], i));
            t = subs8(t, v[j]);
            if (v[j] > kLogoCut) t = OE8;
            s1 = max(s1, t);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int t = byte_of(pr[j].t[i >> 2], i);
            const int a = subs8(f[j], wrap8(t - E8));
            const int c = adds8(v[j], subs8(adds8(s1, OE8), t));
            f[j] = wrap8(max(a, c));
            tracker_step(s1, t, v[j], c1[j], c2[j]);
        }
    }
    // ---- lazy F across the lanes and the hand-over of every lane's end state to its right neighbour (:406-446)
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int ft = f[j];
        const bool mid_lane = lane >= 1 && lane <= 62;
        const int cx = mid_lane ? W * E - rs[j] : 0;
        const int C = dpp_scan_add(cx);
        const int te = C + dpp_scan_max(ft - C);                 // te_x = max(te_{x-1} + c_x, ft_x)
        const int ftn = mid_lane ? min(te, kFClamp) : ft;
        const int nv0 = max(subs8(ftn, E8), v[j]);
        const int f_up = dpp_up1(ftn), v_up = dpp_up1(nv0), c1_up = dpp_up1(c1[j]), c2_up = dpp_up1(c2[j]);
        const int nc1 = lane ? c1_up : fc_c1[j], nc2 = lane ? c2_up : fc_c2[j];
        v[j] = adds8(lane ? v_up : fc_v0[j], nc1); c1[j] = nc1; c2[j] = nc2; f[j] = wrap8(lane ? f_up : fc_f0[j]);
    }
    mid();
    // ---- second pass (:447-565)
    sum = 0;
#pragma unroll
    for (int q = 0; q < (W + 3) / 4; ++q) { out.t[q] = 0; out.e[q] = 0; }
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const int ch = chr[i];
        const int h = (ch == base) ? M : X;
        int fv[NP], eu[NP], qd[NP], ev[NP];
        int best = kSmin, eumax = kSmin;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int t = byte_of(pr[j].t[i >> 2], i), e = byte_of(pr[j].e[i >> 2], i);
            fv[j] = subs8(f[j], v[j]); eu[j] = subs8(e, v[j]); qd[j] = subs8(h, v[j]); ev[j] = subs8(e, t);
            int c = max(max(fv[j], eu[j]), qd[j]);
            if (v[j] > kLogoCut) c = OE8;
            best = max(best, c); eumax = max(eumax, eu[j]);
        }
        if (ch == 'N') best = 0;
        sum = wrap8(sum + best);
        const int open = adds8(best, OE8);
        int es = 1, fs = 1, m1 = 0, emax = kSmin, j_eu = -1, j_q = -1;
#pragma unroll
        for (int j = NP - 1; j >= 0; --j) {
            const int t = byte_of(pr[j].t[i >> 2], i), e = byte_of(pr[j].e[i >> 2], i);
            if (best == eu[j]) j_eu = j;                          // lowest j wins (:485-506)
            if (best == qd[j]) j_q = j;
            int fj = adds8(f[j], E8);
            const int s1 = adds8(open, v[j]);
            m1 = (fj == s1);
            fj = subs8(max(fj, s1), t);
            if (fv[j] == OE8) fs = 2;
            const int e2 = subs8(adds8(e, E8), v[j]);
            emax = max(emax, max(open, e2));
            if (eu[j] == eumax) { es = wrap8(((ev[j] == OE8) ? 43 : 1) + j); if (open == e2) es = wrap8(-es); }
            tracker_step(best, t, v[j], c1[j], c2[j]);
            f[j] = fj;
        }
        if (m1) fs = wrap8(-fs);
        int code = 126;
        if (j_eu >= 0) code = j_eu;
        if (j_q >= 0 && h == X) code = 84 + j_q;
        if (j_q >= 0 && h == M) code = 42 + j_q;
        out.t[i >> 2] |= (best & 255) << ((i & 3) * 8);
        out.e[i >> 2] |= (emax & 255) << ((i & 3) * 8);
        if (ALLIN || i < n_left) { src[i] = (int8_t)code; esrc[i] = (int8_t)es; fsrc[i] = (int8_t)fs; }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) f_last[j] = __builtin_amdgcn_readlane(f[j], 63);   // row->f0 (:566-567): lane 63's F per predecessor
}

}  // namespace

__global__ __launch_bounds__(64) void poa_ref8_kernel(PoaRef8Args P)
{
    const PoaArgs& A = P.A;
    extern __shared__ int dyn[];
    const int maxp = P.maxp, Wq = P.Wq, W = P.W, L = 64 * W;
    int* pt = dyn;                                   // [maxp][Wq][64]: the predecessors' differences of this block, 4 columns per dword
    int* pe = pt + maxp * Wq * 64;                   //                 ... their E differences
    int* st = pe + maxp * Wq * 64;                   // [maxp][64]: per predecessor and lane {v, vc1, vc2, f}
    int* tmp = st + maxp * 64;                       // [maxp][64]: scratch ({fv, eu, q, ev} of the second pass; te of the lane starts)
    int* rs = tmp + maxp * 64;                       // [maxp][64]: the predecessor's differences summed over the lane's W columns
    __shared__ int s_v0[64], s_c1[64], s_c2[64], s_te[64], s_f0[64];     // per predecessor: the block's first-column values
    __shared__ int outb[2][8][64];                   // this row's differences / E differences, packed like pt / pe
    __shared__ uint8_t sqc[32][64];                  // the lane's W sequence characters ('N' beyond the end: readseq, :37-58)
    const int lane = threadIdx.x;
    unsigned ticket = 0;
    if (lane == 0) ticket = atomicAdd(P.ticket, 1u);
    const int k = P.k_fixed >= 0 ? P.k_fixed : (int)__builtin_amdgcn_readfirstlane(ticket);
    if (k >= P.nblk) return;
    const bool mute = P.test_abandon != 0 && k == P.test_abandon - 1;
    const int n = A.n, V = A.V, M = A.M, X = A.X, E = A.E, O = A.O;
    const int E8 = wrap8(E), OE8 = wrap8(O + E);
    const int nblk = P.nblk;
    for (int i = 0; i < W; ++i) { const int c0 = k * L + lane * W + i; sqc[i][lane] = c0 < n ? A.seq[c0] : (uint8_t)'N'; }
    const size_t blk_dw = (size_t)2 * Wq * 64;       // dwords of one (row, block) in P.rows
    int seen_left = 0;
    // ---- round 4: what the fast row (row_one_fast, -W 10) keeps in registers across rows
    constexpr int FW = 10;
    const bool fastW = (W == FW);
    int chr[FW];
#pragma unroll
    for (int i = 0; i < FW; ++i) chr[i] = fastW ? (int)sqc[i][lane] : 0;
    RowBytes<FW> prev;                               // the previous row's bytes of this block (the backbone predecessor never goes through memory)
#pragma unroll
    for (int q = 0; q < (FW + 3) / 4; ++q) { prev.t[q] = 0; prev.e[q] = 0; }
    int prev_r = -2, prev_sum = 0, prev_fc = 0, prev_simple_k = 0;
    // A row's stores are drained and its progress published by the NEXT row: half-way through it when that row takes everything of this
    // one from registers (the stores have long landed then: the s_waitcnt costs nothing), at its start when it reads rows of this block
    // back from memory.  rows_done = rows of this block that are complete.
    bool pending_pub = false;
    auto flush_pub = [&](int rows_done) {
        if (!pending_pub) return;
        if (k + 1 < nblk && !mute) publish_progress(&P.prog[k], rows_done, lane);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pending_pub = false;
    };

    // the row's graph metadata is fetched one row ahead (scalar loads whose latency a lone wave would otherwise wait out at every row's
    // start); the row block advances by counting instead of dividing
    const int n_pred_total = A.pred_off[V];
    int po_cur = A.pred_off[0], po_next = A.pred_off[V > 0 ? 1 : 0], base_next = V > 0 ? A.base[0] : 0;
    int p0_next = (po_cur < po_next) ? A.pred[po_cur] : V;
    int rb = 0, r_in_blk = 0;
    for (int r = 0; r < V; ++r) {
        const int poff = po_cur;
        int np = po_next - poff;
        const bool is_src = (np == 0);
        if (is_src) np = 1;
        const int base = base_next;
        const int pj0 = is_src ? V : p0_next;
        if (r + 1 < V) {                               // next row's: issued now, consumed at the top of the next iteration
            po_cur = po_next;
            po_next = A.pred_off[r + 2];
            base_next = A.base[r + 1];
            p0_next = po_cur < n_pred_total ? A.pred[po_cur] : V;       // (the first predecessor, if the row has one)
        }
        // where control() runs this (row block, column block) pair: task (block_i, block_j, block_l) (:1020-1044, :611-638)
        const int rb_row = rb, bi = rb_row + k;         // rb_row = r / L
        if (++r_in_blk == L) { r_in_blk = 0; ++rb; }
        const int bj = bi <= P.fmaxtag ? bi + 1 : (bi <= P.lmaxtag ? P.fmaxtag + 1 : P.fmaxtag + 1 - (bi - P.lmaxtag));
        const int bl = rb_row - max(bi - P.maxtag, 0);
        const bool first_task = bi <= P.lmaxtag && bl == bj - 1;          // :194
        const bool f0_init = first_task && bi < P.nrowblk;                // :263

        const bool one_fast = fastW && np == 1;
        const bool multi_fast = fastW && (np == 2 || np == 3);   // row_multi_fast<10, np>
        const bool from_prev = one_fast && !is_src && pj0 == prev_r;
        // (-W 10: the previous row is taken from registers wherever it is a predecessor, every older row's stores were drained half-way
        //  through the previous row: nothing to wait for here.  Other -W: rows of this block are read back from memory)
        if (!fastW) flush_pub(r);
        RowBytes<FW> pr;
        int rs_lane = 0;
        if (from_prev) { pr = prev; rs_lane = prev_sum; }
        else if (one_fast) {                       // the one predecessor's bytes straight into registers (constant indices: no scratch)
            int sm = 0;
#pragma unroll
            for (int q = 0; q < (FW + 3) / 4; ++q) {
                int tw, ew;
                if (pj0 == V) {                                            // the virtual first row (:82-94; its first byte :176-185)
                    const int t0 = (bi == 0) ? wrap8(O + E) : E8, e0 = (bi == 0) ? wrap8(2 * (O + E)) : wrap8(E + O + E);
                    tw = pack4(E8, E8, E8, E8); ew = pack4(wrap8(E + E + O), wrap8(E + E + O), wrap8(E + E + O), wrap8(E + E + O));
                    if (q == 0 && lane == 0) { tw = (tw & ~255) | (t0 & 255); ew = (ew & ~255) | (e0 & 255); }
                } else {
                    const int* src = P.rows + ((size_t)pj0 * nblk + k) * blk_dw + (size_t)q * 64 + lane;
                    tw = load_sc1_int(src); ew = load_sc1_int(src + (size_t)Wq * 64);
                }
                pr.t[q] = tw; pr.e[q] = ew;
#pragma unroll
                for (int i = 4 * q; i < (4 * q + 4 < FW ? 4 * q + 4 : FW); ++i) sm = wrap8(sm + byte_of(tw, i));
            }
            rs_lane = sm;
        }
        // ---- the predecessors' bytes of this block into LDS; their sums per lane (:279-288).  Rows of this block: nothing here
        // waits for the block on the left
        RowBytes<FW> prm[3];
        int rsm[3] = {0, 0, 0};
        if (multi_fast) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j < np) {
                    const int pj = A.pred[poff + j];
                    int sm = 0;
#pragma unroll
                    for (int q = 0; q < (FW + 3) / 4; ++q) {
                        int tw, ew;
                        if (pj == prev_r) { tw = prev.t[q]; ew = prev.e[q]; }           // the row just finished: still in registers
                        else {
                            const int* src = P.rows + ((size_t)pj * nblk + k) * blk_dw + (size_t)q * 64 + lane;
                            tw = load_sc1_int(src); ew = load_sc1_int(src + (size_t)Wq * 64);
                        }
                        prm[j].t[q] = tw; prm[j].e[q] = ew;
#pragma unroll
                        for (int i = 4 * q; i < (4 * q + 4 < FW ? 4 * q + 4 : FW); ++i) sm = wrap8(sm + byte_of(tw, i));
                    }
                    rsm[j] = sm;
                }
            }
        }
        for (int j = 0; j < np && !one_fast && !multi_fast; ++j) {
            const int pj = is_src ? V : A.pred[poff + j];
            int sum = 0;
            for (int q = 0; q < Wq; ++q) {
                int tw, ew;
                if (pj == V) {                                             // the virtual first row (:82-94; its first byte :176-185)
                    const int t0 = (bi == 0) ? wrap8(O + E) : E8, e0 = (bi == 0) ? wrap8(2 * (O + E)) : wrap8(E + O + E);
                    tw = pack4(E8, E8, E8, E8); ew = pack4(wrap8(E + E + O), wrap8(E + E + O), wrap8(E + E + O), wrap8(E + E + O));
                    if (q == 0 && lane == 0) { tw = (tw & ~255) | (t0 & 255); ew = (ew & ~255) | (e0 & 255); }
                } else if (fastW && pj == prev_r) {                        // the row just finished: still in registers (its stores may be under way)
                    tw = q == 0 ? prev.t[0] : (q == 1 ? prev.t[1] : prev.t[2]);
                    ew = q == 0 ? prev.e[0] : (q == 1 ? prev.e[1] : prev.e[2]);
                } else {
                    const int* src = P.rows + ((size_t)pj * nblk + k) * blk_dw + (size_t)q * 64 + lane;
                    tw = load_sc1_int(src); ew = load_sc1_int(src + (size_t)Wq * 64);
                }
                pt[(j * Wq + q) * 64 + lane] = tw; pe[(j * Wq + q) * 64 + lane] = ew;
                for (int i = 4 * q; i < min(W, 4 * q + 4); ++i) sum = wrap8(sum + byte_of(tw, i));
            }
            rs[j * 64 + lane] = sum;
        }
        // ---- the block on the left has finished this row (its running sum, its F carries); how far it is is remembered
        if (k > 0 && seen_left <= r) {
            int cur = __builtin_amdgcn_readfirstlane(load_sc1_int(&P.prog[k - 1]));
            if (cur < r + 1) { wait_progress(&P.prog[k - 1], r + 1, P.status, lane); cur = r + 1; }
            asm volatile("" ::: "memory");
            seen_left = cur;
        }
        // ---- the block's first column: this row's offset above every predecessor (lane j = predecessor j); all loads first
        const int pj_l = lane < np ? (is_src ? V : A.pred[poff + lane]) : -1;
        const bool has_p = lane < np && !is_src;
        int simple_rk = (k > 0) ? load_sc1_int(&P.simple[(size_t)r * (nblk + 1) + k]) : 0;
        const bool p_prev = pj_l == prev_r;          // (what the previous row left is still in registers)
        const int fcp = (has_p && first_task) ? ((p_prev && k == 0) ? prev_fc : load_sc1_int(&P.first_col[pj_l])) : INT_MIN / 2;
        const int spk = (has_p && !first_task) ? (p_prev ? prev_simple_k : load_sc1_int(&P.simple[(size_t)pj_l * (nblk + 1) + k])) : 0;
        const int f0_in = (!f0_init && lane < np) ? load_sc1_int(&P.f0[(size_t)r * maxp + lane]) : 0;
        int v0_b = 0, f0_b = 0, fc_row = E + O;      // lane 0's first-column v and F (a row with one predecessor needs nothing else); first_col of this row
        {
            int te = 0, v0 = 0, c1 = 0, c2 = 0;
            if (is_src) {                                                  // :174-188, :227-232
                if (k == 0) simple_rk = E + O;
                v0 = wrap8(simple_rk - (k * L * E + (k > 0 ? O : 0)));
                if (k == 0 && lane == 0) { store_sc1_int(&P.first_col[r], E + O); store_sc1_int(&P.simple[(size_t)r * (nblk + 1)], E + O); }
            } else if (first_task) {                                       // :194-224
                int fc = fcp;
                if (np == 1) fc = __builtin_amdgcn_readfirstlane(fcp);      // (lanes beyond the predecessors hold minus infinity)
                else {
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) fc = max(fc, __shfl_xor(fc, off, 64));
                }
                fc += E;
                te = fc - fcp;
                split_offset(te, v0, c1, c2);
                fc_row = fc;
                if (k == 0) {
                    simple_rk = fc;
                    if (lane == 0) { store_sc1_int(&P.first_col[r], fc); store_sc1_int(&P.simple[(size_t)r * (nblk + 1)], fc); }
                }
            } else {                                                       // :233-261
                te = simple_rk - spk;
                split_offset(te, v0, c1, c2);
            }
            if (np == 1) { c1 = 0; c2 = 0; }                               // :339-346
            const int f0v = f0_init ? wrap8(v0 + E + O) : f0_in;          // :263-271
            s_v0[lane] = v0; s_c1[lane] = c1; s_c2[lane] = c2; s_te[lane] = te;
            s_f0[lane] = f0v;
            v0_b = __builtin_amdgcn_readfirstlane(v0); f0_b = __builtin_amdgcn_readfirstlane(f0v);
        }
        int sum = 0, f_last = 0;
        // the three sweeps over the row; ONE: a node with a single predecessor keeps its lane state in a register
        auto sweeps = [&](auto one_c) {
            constexpr bool ONE = decltype(one_c)::value;
            const int npj = ONE ? 1 : np;
            int S1 = 0, T1 = 0;
            auto ld = [&](int j) -> int { if constexpr (ONE) return S1; else return st[j * 64 + lane]; };
            auto sd = [&](int j, int v) { if constexpr (ONE) S1 = v; else st[j * 64 + lane] = v; };
            // ---- what every lane starts its first pass with (:290-356)
            if constexpr (!ONE) {
                int m1 = INT_MAX;
                for (int j = 0; j < npj; ++j) {
                    const int x = W * E - rs[j * 64 + lane];
                    const int te = s_te[j] + scan_add(x, lane) - x;        // te_j at the start of lane b: the first column's, carried on
                    tmp[j * 64 + lane] = te; m1 = min(m1, te);             // under the guess that this row falls by E per column
                }
                for (int j = 0; j < npj; ++j) {
                    int v, c1, c2;
                    split_offset(tmp[j * 64 + lane] + kLogoLow - m1, v, c1, c2);
                    if (lane == 0) { v = s_v0[j]; c1 = s_c1[j]; c2 = s_c2[j]; }
                    sd(j, pack4(v, c1, c2, lane == 0 ? s_f0[j] : kSmin));
                }
            } else sd(0, pack4(lane == 0 ? s_v0[0] : E8, 0, 0, lane == 0 ? s_f0[0] : kSmin));
            // ---- first pass (:357-404)
            for (int i = 0; i < W; ++i) {
                const int h = ((int)sqc[i][lane] == base) ? M : X;
                const int q4 = (i >> 2) * 64 + lane;
                int s1 = kSmin;
                for (int j = 0; j < npj; ++j) {
                    const int S = ld(j);
                    const int v = wrap8(S), f = S >> 24;
                    int t = max(max(f, h), byte_of(pe[j * Wq * 64 + q4], i));
                    t = subs8(t, v);
                    if (v > kLogoCut) t = OE8;
                    s1 = max(s1, t);
                }
                for (int j = 0; j < npj; ++j) {
                    const int S = ld(j);
                    int v = wrap8(S), c1 = wrap8(S >> 8), c2 = wrap8(S >> 16), f = S >> 24;
                    const int t = byte_of(pt[j * Wq * 64 + q4], i);
                    const int a = subs8(f, wrap8(t - E8));
                    const int c = adds8(v, subs8(adds8(s1, OE8), t));
                    f = max(a, c);
                    tracker_step(s1, t, v, c1, c2);
                    sd(j, pack4(v, c1, c2, f));
                }
            }
            // ---- lazy F across the lanes and the hand-over of every lane's end state to its right neighbour (:406-446)
            for (int j = 0; j < npj; ++j) {
                const int S = ld(j);
                const int v = wrap8(S), c1 = wrap8(S >> 8), c2 = wrap8(S >> 16), ft = S >> 24;
                const bool mid = lane >= 1 && lane <= 62;
                const int cx = mid ? W * E - rs[j * 64 + lane] : 0;
                const int C = scan_add(cx, lane);
                const int te = C + scan_max(ft - C, lane);                 // te_x = max(te_{x-1} + c_x, ft_x)
                const int ftn = mid ? min(te, kFClamp) : ft;
                const int nv0 = max(subs8(ftn, E8), v);
                const int f_up = dpp_up1(ftn), v_up = dpp_up1(nv0), c1_up = dpp_up1(c1), c2_up = dpp_up1(c2);
                const int nc1 = lane ? c1_up : s_c1[j], nc2 = lane ? c2_up : s_c2[j];
                sd(j, pack4(adds8(lane ? v_up : s_v0[j], nc1), nc1, nc2, lane ? f_up : s_f0[j]));
            }
            flush_pub(r);                          // (the previous row's stores have long landed: see flush_pub)
            // ---- second pass (:447-565)
            for (int i = 0; i < W; ++i) {
                const int ch = sqc[i][lane];
                const bool isn = (ch == 'N');
                const int h = (ch == base) ? M : X;
                const int q4 = (i >> 2) * 64 + lane;
                int best = kSmin, eumax = kSmin;
                for (int j = 0; j < npj; ++j) {
                    const int S = ld(j);
                    const int v = wrap8(S), f = S >> 24;
                    const int t = byte_of(pt[j * Wq * 64 + q4], i), e = byte_of(pe[j * Wq * 64 + q4], i);
                    const int fv = subs8(f, v), eu = subs8(e, v), q = subs8(h, v), ev = subs8(e, t);
                    int c = max(max(fv, eu), q);
                    if (v > kLogoCut) c = OE8;
                    best = max(best, c); eumax = max(eumax, eu);
                    if constexpr (ONE) T1 = pack4(fv, eu, q, ev); else tmp[j * 64 + lane] = pack4(fv, eu, q, ev);
                }
                if (isn) best = 0;
                sum = wrap8(sum + best);
                const int open = adds8(best, OE8);
                int es = 1, fs = 1, es_num = 0, m1 = 0, emax = kSmin, j_eu = -1, j_q = -1;
                for (int j = npj - 1; j >= 0; --j) {
                    const int S = ld(j);
                    int Tm;
                    if constexpr (ONE) Tm = T1; else Tm = tmp[j * 64 + lane];
                    int v = wrap8(S), c1 = wrap8(S >> 8), c2 = wrap8(S >> 16), f = S >> 24;
                    const int fv = wrap8(Tm), eu = wrap8(Tm >> 8), q = wrap8(Tm >> 16), ev = Tm >> 24;
                    const int t = byte_of(pt[j * Wq * 64 + q4], i), e = byte_of(pe[j * Wq * 64 + q4], i);
                    if (best == eu) j_eu = j;                             // lowest j wins (:485-506)
                    if (best == q) j_q = j;
                    f = adds8(f, E8);
                    const int s1 = adds8(open, v);
                    m1 = (f == s1);
                    f = subs8(max(f, s1), t);
                    if (fv == OE8) fs = 2;
                    const int e2 = subs8(adds8(e, E8), v);
                    emax = max(emax, max(open, e2));
                    const bool m3 = (eu == eumax);
                    if (m3) { es_num = j; es = wrap8(((ev == OE8) ? 43 : 1) + es_num); if (open == e2) es = wrap8(-es); }
                    tracker_step(best, t, v, c1, c2);
                    sd(j, pack4(v, c1, c2, f));
                }
                if (m1) fs = wrap8(-fs);
                int code = 126;
                if (j_eu >= 0) code = j_eu;
                if (j_q >= 0 && h == X) code = 84 + j_q;
                if (j_q >= 0 && h == M) code = 42 + j_q;
                reinterpret_cast<int8_t*>(&outb[0][i >> 2][lane])[i & 3] = (int8_t)best;
                reinterpret_cast<int8_t*>(&outb[1][i >> 2][lane])[i & 3] = (int8_t)emax;
                const int c0 = k * L + lane * W + i;
                if (A.source && c0 < n) {
                    const size_t o = (size_t)r * A.tpitch + c0;
                    A.source[o] = (int8_t)code; A.esource[o] = (int8_t)es; A.fsource[o] = (int8_t)fs;
                }
            }
            if constexpr (ONE) f_last = __shfl(S1 >> 24, 63, 64);         // row->f0 (:566-567): lane 63's F
            else f_last = lane < np ? (st[lane * 64 + 63] >> 24) : 0;
        };
        bool fast_done = false;
        RowBytes<FW> ob;
        if (one_fast) {
            const int c0 = k * L + lane * FW;
            int8_t* src = A.source ? A.source + (size_t)r * A.tpitch + c0 : nullptr;
            int8_t* esrc = A.source ? A.esource + (size_t)r * A.tpitch + c0 : nullptr;
            int8_t* fsrc = A.source ? A.fsource + (size_t)r * A.tpitch + c0 : nullptr;
            int fl = 0;
            auto mid = [&] { flush_pub(r); };
            // (whole wave inside the sequence and trace arrays wanted: the stores need no per-column test; a wave-uniform choice)
            if (A.source && k * L + 64 * FW <= n)
                fast_done = row_one_fast<FW, true>(mid, pr, rs_lane, v0_b, f0_b, chr, base, M, X, E, O, lane, ob, sum, fl, src, esrc, fsrc, FW);
            else
                fast_done = row_one_fast<FW, false>(mid, pr, rs_lane, v0_b, f0_b, chr, base, M, X, E, O, lane, ob, sum, fl, src, esrc, fsrc,
                                                    A.source ? n - c0 : 0);
            if (fast_done) f_last = fl;               // (lane 0 stores it: one predecessor)
            else {                                    // a reservoir filled: the general sweeps, from LDS as usual
#pragma unroll
                for (int q = 0; q < (FW + 3) / 4; ++q) { pt[q * 64 + lane] = pr.t[q]; pe[q * 64 + lane] = pr.e[q]; }
                rs[lane] = rs_lane;
            }
            if (!fast_done) sum = 0;
        }
        if (multi_fast) {
            const int c0 = k * L + lane * FW;
            int8_t* src = A.source ? A.source + (size_t)r * A.tpitch + c0 : nullptr;
            int8_t* esrc = A.source ? A.esource + (size_t)r * A.tpitch + c0 : nullptr;
            int8_t* fsrc = A.source ? A.fsource + (size_t)r * A.tpitch + c0 : nullptr;
            auto mid = [&] { flush_pub(r); };
            const bool allin = A.source && k * L + 64 * FW <= n;
            const int nl = A.source ? n - c0 : 0;
            auto run = [&](auto np_c) {
                constexpr int NP = decltype(np_c)::value;
                RowBytes<FW> prj[NP]; int rsj[NP], a_v0[NP], a_c1[NP], a_c2[NP], a_te[NP], a_f0[NP], fl[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    prj[j] = prm[j]; rsj[j] = rsm[j];
                    a_v0[j] = s_v0[j]; a_c1[j] = s_c1[j]; a_c2[j] = s_c2[j]; a_te[j] = s_te[j]; a_f0[j] = s_f0[j];
                }
                if (allin) row_multi_fast<FW, NP, true>(mid, prj, rsj, a_v0, a_c1, a_c2, a_te, a_f0, chr, base, M, X, E, O, lane, ob, sum, fl, src, esrc, fsrc, FW);
                else       row_multi_fast<FW, NP, false>(mid, prj, rsj, a_v0, a_c1, a_c2, a_te, a_f0, chr, base, M, X, E, O, lane, ob, sum, fl, src, esrc, fsrc, nl);
                f_last = 0;
#pragma unroll
                for (int j = 0; j < NP; ++j) if (lane == j) f_last = fl[j];
            };
            if (np == 2) run(std::integral_constant<int, 2>{}); else run(std::integral_constant<int, 3>{});
            fast_done = true;
        }
        if (!fast_done) {
            if (np == 1) sweeps(std::true_type{}); else sweeps(std::false_type{});
            if (fastW) {
#pragma unroll
                for (int q = 0; q < (FW + 3) / 4; ++q) { ob.t[q] = outb[0][q][lane]; ob.e[q] = outb[1][q][lane]; }
            }
        }
        // ---- what the row leaves: its bytes of this block, the F carries, the running sum (:568-575)
        if (fastW) {
#pragma unroll
            for (int q = 0; q < (FW + 3) / 4; ++q) {
                int* dst = P.rows + ((size_t)r * nblk + k) * blk_dw + (size_t)q * 64 + lane;
                store_sc1_int(dst, ob.t[q]); store_sc1_int(dst + (size_t)Wq * 64, ob.e[q]);
            }
        } else {
            for (int q = 0; q < Wq; ++q) {
                int* dst = P.rows + ((size_t)r * nblk + k) * blk_dw + (size_t)q * 64 + lane;
                store_sc1_int(dst, outb[0][q][lane]); store_sc1_int(dst + (size_t)Wq * 64, outb[1][q][lane]);
            }
        }
        if (lane < np) store_sc1_int(&P.f0[(size_t)r * maxp + lane], f_last);
        const int red = wave_sum(sum);
        const int total = simple_rk + (int)(int16_t)red;
        if (lane == 0) {
            store_sc1_int(&P.simple[(size_t)r * (nblk + 1) + k + 1], total);
            if (k == nblk - 1) A.lastscore[r] = total;
        }
        // the row's stores are drained, and the block on the right told, by the next row (flush_pub)
        if (fastW) { prev = ob; prev_r = r; prev_sum = sum; prev_fc = fc_row; prev_simple_k = simple_rk; }
        pending_pub = true;
    }
    flush_pub(V);
}

size_t poa_ref8_lds_bytes(int maxp, int Wq) { return (size_t)maxp * 64 * sizeof(int) * (2 * (size_t)Wq + 3); }

// single = true: all blocks in one launch (tickets, progress words); false: one launch per block, left to right (what a call
// is run again with after the single launch abandoned a wait: the kernel boundary is then the only synchronisation)
int launch_poa_ref8(PoaRef8Args P, bool single, hipStream_t s)
{
    const size_t lds = poa_ref8_lds_bytes(P.maxp, P.Wq);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(poa_ref8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    if (single) { P.k_fixed = -1; hipLaunchKernelGGL(poa_ref8_kernel, dim3((unsigned)P.nblk), dim3(64), lds, s, P); return 1; }
    P.test_abandon = 0;
    for (int k = 0; k < P.nblk; ++k) { P.k_fixed = k; hipLaunchKernelGGL(poa_ref8_kernel, dim3(1), dim3(64), lds, s, P); }
    return P.nblk;
}

}  // namespace tsta
