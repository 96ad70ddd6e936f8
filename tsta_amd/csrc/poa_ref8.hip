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

__device__ __forceinline__ int scan_add(int x, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(x, off, 64); if (lane >= off) x += o; }
    return x;
}
__device__ __forceinline__ int scan_max(int x, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(x, off, 64); if (lane >= off) x = max(x, o); }
    return x;
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

    for (int r = 0; r < V; ++r) {
        const int poff = A.pred_off[r];
        int np = A.pred_off[r + 1] - poff;
        const bool is_src = (np == 0);
        if (is_src) np = 1;
        const int base = A.base[r];
        // where control() runs this (row block, column block) pair: task (block_i, block_j, block_l) (:1020-1044, :611-638)
        const int rb = r / L, bi = rb + k;
        const int bj = bi <= P.fmaxtag ? bi + 1 : (bi <= P.lmaxtag ? P.fmaxtag + 1 : P.fmaxtag + 1 - (bi - P.lmaxtag));
        const int bl = rb - max(bi - P.maxtag, 0);
        const bool first_task = bi <= P.lmaxtag && bl == bj - 1;          // :194
        const bool f0_init = first_task && bi < P.nrowblk;                // :263

        // ---- the predecessors' bytes of this block into LDS; their sums per lane (:279-288).  Rows of this block: nothing here
        // waits for the block on the left
        for (int j = 0; j < np; ++j) {
            const int pj = is_src ? V : A.pred[poff + j];
            int sum = 0;
            for (int q = 0; q < Wq; ++q) {
                int tw, ew;
                if (pj == V) {                                             // the virtual first row (:82-94; its first byte :176-185)
                    const int t0 = (bi == 0) ? wrap8(O + E) : E8, e0 = (bi == 0) ? wrap8(2 * (O + E)) : wrap8(E + O + E);
                    tw = pack4(E8, E8, E8, E8); ew = pack4(wrap8(E + E + O), wrap8(E + E + O), wrap8(E + E + O), wrap8(E + E + O));
                    if (q == 0 && lane == 0) { tw = (tw & ~255) | (t0 & 255); ew = (ew & ~255) | (e0 & 255); }
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
        const int fcp = (has_p && first_task) ? load_sc1_int(&P.first_col[pj_l]) : INT_MIN / 2;
        const int spk = (has_p && !first_task) ? load_sc1_int(&P.simple[(size_t)pj_l * (nblk + 1) + k]) : 0;
        const int f0_in = (!f0_init && lane < np) ? load_sc1_int(&P.f0[(size_t)r * maxp + lane]) : 0;
        {
            int te = 0, v0 = 0, c1 = 0, c2 = 0;
            if (is_src) {                                                  // :174-188, :227-232
                if (k == 0) simple_rk = E + O;
                v0 = wrap8(simple_rk - (k * L * E + (k > 0 ? O : 0)));
                if (k == 0 && lane == 0) { store_sc1_int(&P.first_col[r], E + O); store_sc1_int(&P.simple[(size_t)r * (nblk + 1)], E + O); }
            } else if (first_task) {                                       // :194-224
                int fc = fcp;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) fc = max(fc, __shfl_xor(fc, off, 64));
                fc += E;
                te = fc - fcp;
                split_offset(te, v0, c1, c2);
                if (k == 0) {
                    simple_rk = fc;
                    if (lane == 0) { store_sc1_int(&P.first_col[r], fc); store_sc1_int(&P.simple[(size_t)r * (nblk + 1)], fc); }
                }
            } else {                                                       // :233-261
                te = simple_rk - spk;
                split_offset(te, v0, c1, c2);
            }
            if (np == 1) { c1 = 0; c2 = 0; }                               // :339-346
            s_v0[lane] = v0; s_c1[lane] = c1; s_c2[lane] = c2; s_te[lane] = te;
            s_f0[lane] = f0_init ? wrap8(v0 + E + O) : f0_in;              // :263-271
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
                const int f_up = __shfl_up(ftn, 1, 64), v_up = __shfl_up(nv0, 1, 64), c1_up = __shfl_up(c1, 1, 64), c2_up = __shfl_up(c2, 1, 64);
                const int nc1 = lane ? c1_up : s_c1[j], nc2 = lane ? c2_up : s_c2[j];
                sd(j, pack4(adds8(lane ? v_up : s_v0[j], nc1), nc1, nc2, lane ? f_up : s_f0[j]));
            }
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
        if (np == 1) sweeps(std::true_type{}); else sweeps(std::false_type{});
        // ---- what the row leaves: its bytes of this block, the F carries, the running sum (:568-575)
        for (int q = 0; q < Wq; ++q) {
            int* dst = P.rows + ((size_t)r * nblk + k) * blk_dw + (size_t)q * 64 + lane;
            store_sc1_int(dst, outb[0][q][lane]); store_sc1_int(dst + (size_t)Wq * 64, outb[1][q][lane]);
        }
        if (lane < np) store_sc1_int(&P.f0[(size_t)r * maxp + lane], f_last);
        int red = sum;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) red += __shfl_xor(red, off, 64);
        const int total = simple_rk + (int)(int16_t)red;
        if (lane == 0) {
            store_sc1_int(&P.simple[(size_t)r * (nblk + 1) + k + 1], total);
            if (k == nblk - 1) A.lastscore[r] = total;
        }
        // the row's stores have left before anything of the next row is loaded (later rows of this block read them back); the
        // block on the right is told
        if (k + 1 < nblk && !mute) publish_progress(&P.prog[k], r + 1, lane);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
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
