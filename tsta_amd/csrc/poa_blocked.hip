// poa_blocked.hip — K4n: the POA fill for sequences with a literal 'N', in the blocking of the reference's AVX512 build.
//
// Why a kernel of its own.  The reference pads sequences with 'N' and forces the column delta to 0 in such columns, but only
// in its SECOND pass (msa/c-t-simd.c:461,481); the first pass (:357-404), which produces what one SIMD lane segment hands to
// the next (the F carry-out and the offsets v_j = H(r, c) - H(p_j, c) at the segment's end, :433-446), does not.  A block of
// L = 64 x W columns is 64 lane segments of W columns (NUM2, :14), so on a sequence with a literal 'N' the reference's
// output depends on where the segment boundaries fall, i.e. on -W.  What it computes there is still a function of
// (input, W) as long as its saturating int8 trackers stay in range, and this kernel computes exactly that function in plain
// int32 (the test infrastructure holds a CPU restatement of the same; measured against the compiled reference: 179 of 200 outputs
// of tests/golden/msa_n_golden.json, all of those on which the trackers do not saturate):
//   per node r and block k, lane b = segment b:
//   pass 1   the segment on its own, WITHOUT the rule, lanes >= 1 with no F carry-in: H1(end), F1out;
//   lazy F   Fout(b) = max(F1out(b), Fout(b-1) + W E), b = 1..62                                   (:408-432)
//   pass 2   the segment again, WITH the rule, from Hs = max(H1(end of b-1), Fout(b-1) - E) and F = Fout(b-1): trace codes
//            and per-column DELTAS; the row's absolute scores are the running sum of the deltas (simple_sorce, :572-575),
//            so a segment whose pass-1 hand-over differs from what pass 2 computed for the same cell shifts the rest of the
//            row's block by D = H_pass2 - H_pass1 ("shadow" frame Hs = H - D).  At the next block D = 0 again (:196-261),
//            the F carry (row->f0) keeps its shadow-frame value.
// Mapping: one wave per block, sweeping the nodes in topological order; a block waits for the block on its left to have
// finished the row (progress word, sc1 hand-off as everywhere: handoff.h; blocks are tickets in start order).  Rows are kept
// as full int32 rows H / Eo in HBM ((V + 1) x (n + 1) x 8 B), so the host uses this kernel only where that fits and W <= 32;
// otherwise, and for 'N'-free sequences always, the other fills run.  Speed is not the point of this path (sequences with a
// literal 'N' are rare and small next to the fills around them): ~5 us per node and block.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "../../include/tsta_engine.h"
#include "engine.h"
#include "handoff.h"
#include "poa_types.h"

namespace tsta {

constexpr int kBlkMaxW = 32;

__global__ __launch_bounds__(64) void poa_blocked_kernel(PoaBlockedArgs P)
{
    const PoaArgs& A = P.A;
    __shared__ int c_bq[kBlkMaxW][64], c_be[kBlkMaxW][64], c_jj[kBlkMaxW][64];      // per column of the lane's segment: best diagonal /
                                                                                     // E candidate, jq | je << 8 | E-open << 16
    const int lane = threadIdx.x;
    unsigned ticket = 0;
    if (lane == 0) ticket = atomicAdd(P.ticket, 1u);
    const int k = P.k_fixed >= 0 ? P.k_fixed : (int)__builtin_amdgcn_readfirstlane(ticket);     // (k_fixed: one launch per block, left to right)
    if (k >= P.nblk) return;
    const bool mute = P.test_abandon != 0 && k == P.test_abandon - 1;
    const int n = A.n, V = A.V, M = A.M, X = A.X, E = A.E, OE = A.O + A.E, W = P.W, L = 64 * W;
    const int k0 = k * L;
    const int c_lo = k0 + lane * W + 1;                          // first column (1-based) of this lane's segment
    const int cnt = max(0, min(W, n - c_lo + 1));                // its real columns
    const int64_t hp = P.hpitch;
    int fcarry = kNeg;                                           // F of the block's first column (shadow frame of the previous block's end), lane 0
    int seen_left = 0;
    int maxd = 0;
    for (int r = 0; r < V; ++r) {
        const int poff = A.pred_off[r];
        int np = A.pred_off[r + 1] - poff;
        const bool is_src = (np == 0);
        if (is_src) np = 1;
        const int base = A.base[r];
        int* Hr = P.Hf + (size_t)r * hp; int* Er = P.Ef + (size_t)r * hp;
        int h_start;
        if (k > 0 && seen_left <= r) {                           // the block on the left has finished this row (its Hr[k0], its F carry)
            wait_progress(&P.prog[k - 1], r + 1, P.status, lane);
            seen_left = r + 1;
        }
        if (k == 0) {                                            // first column (c-t-simd.c:187,194-200)
            int h0 = kNeg;
            for (int j = 0; j < np; ++j) { const int pj = is_src ? V : A.pred[poff + j]; h0 = max(h0, load_sc1_int(&P.Hf[(size_t)pj * hp])); }
            const int hv = is_src ? OE : h0 + E;
            if (lane == 0) store_sc1_int(&Hr[0], hv);
            fcarry = hv + OE;                                    // :264-271: f0 = v0 + E + O
            h_start = hv;
        } else {
            fcarry = load_sc1_int(&P.fc[(size_t)r * P.nblk + k]);
            h_start = load_sc1_int(&Hr[k0]);                     // H(r, last column of the block on the left)
        }
        // ---- candidates of every column of the segment: independent of the row's own past, gathered first
        for (int i = 0; i < cnt; ++i) {
            const int c = c_lo + i;
            const int sc = ((int)A.seq[c - 1] == base) ? M : X;
            int bq = kNeg, jq = 0, be = kNeg, je = 0, jo = 0;
            for (int j = 0; j < np; ++j) {
                const int pj = is_src ? V : A.pred[poff + j];
                const int* Hp = P.Hf + (size_t)pj * hp; const int* Ep = P.Ef + (size_t)pj * hp;
                const int2 hh = make_int2(load_sc1_int(&Hp[c - 1]), load_sc1_int(&Hp[c]));
                const int ein = load_sc1_int(&Ep[c]);
                const int q = hh.x + sc;
                if (q > bq) { bq = q; jq = j; }
                if (ein > be) { be = ein; je = j; jo = (ein == hh.y + OE) ? 1 : 0; }
            }
            c_bq[i][lane] = bq; c_be[i][lane] = be; c_jj[i][lane] = jq | (je << 8) | (jo << 16);
        }
        // ---- pass 1: no 'N' rule; lane 0 from the block's true start, the others with no F carry-in
        int h1 = (lane == 0) ? h_start : kNeg, f1 = (lane == 0) ? fcarry : kNeg;
        bool have = (lane == 0);
        for (int i = 0; i < cnt; ++i) {
            const int a = max(c_bq[i][lane], c_be[i][lane]);
            const int fcur = have ? f1 : kNeg;
            const int h = max(a, fcur);
            f1 = (fcur <= kNeg / 2) ? h + OE : max(fcur + E, h + OE);
            h1 = h; have = true;
        }
        if (cnt == 0) { h1 = kNeg; f1 = kNeg; }
        // ---- lazy F across the lanes: Fout(b) = max(F1out(b), Fout(b-1) + W E)   (max-plus scan; lane 63's is not used)
        int fo = f1;
        {
            const int wE = W * E;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(fo, off, 64);
                if (lane >= off && o > kNeg / 2) fo = max(fo, o + off * wE);
            }
        }
        // ---- hand-over into this lane's segment
        const int h1_prev = __shfl_up(h1, 1, 64), fo_prev = __shfl_up(fo, 1, 64);
        int hs = (lane == 0) ? h_start : max(h1_prev, fo_prev > kNeg / 2 ? fo_prev - E : kNeg);        // :437-440
        int fs = (lane == 0) ? fcarry : fo_prev;
        const int hs0 = hs;
        // ---- pass 2: with the rule; trace codes; shadow-frame scores kept in c_bq / c_be for the fix-up below
        for (int i = 0; i < cnt; ++i) {
            const int c = c_lo + i;
            const int ch = A.seq[c - 1];
            const bool is_n = (ch == 'N');
            const int bq = c_bq[i][lane], be = c_be[i][lane], jj = c_jj[i][lane];
            const int jq = jj & 0xff, je = (jj >> 8) & 0xff, jo = (jj >> 16) & 1;
            const int f = fs, f_open = hs + OE;
            const int h = is_n ? hs : max(bq, max(be, f));
            int code;
            if (is_n) {
                // equality with the forced value: lowest j whose diagonal / E candidate equals it (c-t-simd.c:485-506)
                const int sc = (ch == base) ? M : X;
                int jq_n = -1, je_n = -1;
                for (int j = 0; j < np; ++j) {
                    const int pj = is_src ? V : A.pred[poff + j];
                    const int q = load_sc1_int(&P.Hf[(size_t)pj * hp + c - 1]) + sc, ein = load_sc1_int(&P.Ef[(size_t)pj * hp + c]);
                    if (jq_n < 0 && q == hs) jq_n = j;
                    if (je_n < 0 && ein == hs) je_n = j;
                }
                code = jq_n >= 0 ? ((ch == base ? 42 : 84) + jq_n) : (je_n >= 0 ? je_n : 126);
            }
            else if (bq == h) code = (ch == base ? 42 : 84) + jq;
            else if (be == h) code = je;
            else              code = 126;
            int ev = (jo ? 43 : 1) + je;
            if (be + E == h + OE) ev = -ev;
            int fv = (f == f_open) ? 2 : 1;
            if (f + E == h + OE) fv = -fv;
            if (A.source) {
                const size_t o = (size_t)r * A.tpitch + (c - 1);
                A.source[o] = (int8_t)code; A.esource[o] = (int8_t)ev; A.fsource[o] = (int8_t)fv;
            }
            c_bq[i][lane] = h;                                   // shadow H(r, c)
            c_be[i][lane] = max(h + OE, be + E);                 // shadow E a successor sees through this node
            fs = max(f + E, h + OE);
            hs = h;
        }
        // ---- absolute scores = running sum of the deltas: D of this segment = (sum of the deltas before it) - (its shadow start)
        int delta = (cnt > 0) ? hs - hs0 : 0, run = delta;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(run, off, 64); if (lane >= off) run += o; }
        const int D = h_start + (run - delta) - hs0;
        if (cnt > 0) maxd = max(maxd, D < 0 ? -D : D);
        for (int i = 0; i < cnt; ++i) {
            const int c = c_lo + i;
            store_sc1_int(&Hr[c], c_bq[i][lane] + D);
            store_sc1_int(&Er[c], c_be[i][lane] + D);
        }
        if (cnt > 0 && c_lo + cnt - 1 == n) A.lastscore[r] = hs + D;
        // F carry into the next block: lane 63's F after its last column (shadow frame)
        if (lane == 63 && k + 1 < P.nblk) store_sc1_int(&P.fc[(size_t)r * P.nblk + k + 1], fs);
        // the row's stores have left before anything of the next row is loaded (later rows of this block read them back, other
        // lanes' segments included); the block on the right is told
        if (k + 1 < P.nblk && !mute) publish_progress(&P.prog[k], r + 1, lane);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (maxd > 0) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) maxd = max(maxd, __shfl_xor(maxd, off, 64));
        if (lane == 0) atomicMax(A.best + 9, maxd);
    }
}

// virtual init row (index V): H = 0, O + cE; the E a successor sees through it = H + O + E
__global__ void poa_blocked_init_kernel(PoaBlockedArgs P)
{
    int* Hi = P.Hf + (size_t)P.A.V * P.hpitch; int* Ei = P.Ef + (size_t)P.A.V * P.hpitch;
    const int OE = P.A.O + P.A.E;
    if (blockIdx.x == 0 && threadIdx.x == 0) { P.A.best[8] = 0; P.A.best[9] = 0; }
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c <= P.A.n; c += gridDim.x * blockDim.x) {
        const int h = c == 0 ? 0 : P.A.O + c * P.A.E;
        Hi[c] = h; Ei[c] = c == 0 ? kNeg : h + OE;
    }
}

// single = true: all blocks in one launch (tickets, progress words); false: one launch per block, left to right (what a call
// is run again with after the single launch abandoned a wait: the kernel boundary is then the only synchronisation)
int launch_poa_blocked(PoaBlockedArgs P, bool single, hipStream_t s)
{
    hipLaunchKernelGGL(poa_blocked_init_kernel, dim3(std::min(64, (P.A.n + 256) / 256)), dim3(256), 0, s, P);
    if (single) { P.k_fixed = -1; hipLaunchKernelGGL(poa_blocked_kernel, dim3((unsigned)P.nblk), dim3(64), 0, s, P); return 2; }
    P.test_abandon = 0;
    for (int k = 0; k < P.nblk; ++k) { P.k_fixed = k; hipLaunchKernelGGL(poa_blocked_kernel, dim3(1), dim3(64), 0, s, P); }
    return 1 + P.nblk;
}

}  // namespace tsta
