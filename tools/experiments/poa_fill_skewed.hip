// poa_fill_skewed.hip — K4s: the sequence-vs-POA-graph fill with the lane-skewed sweep of the pairwise kernels.
//
// Same recurrence, same trace codes, same outputs as K4 (poa_fill.hip; reference: msa/c-t-simd.c:165-599 driven by
// :602-657, trace codes :485-506, 530-538, 521-522/564-565).  Mapping: a band is 64 consecutive nodes of the
// topological order, lane l owns node 64b+l and is at sequence position x = t - l in step t, so
//   * the horizontal gap F is a lane-local carry (no prefix scan over the wave),
//   * the backbone predecessor (node r-1) is the lane on the left: its H/Eagg of the same column arrive by ONE DPP
//     shift from the previous step (as in the pairwise kernels),
//   * every other predecessor row is read from an LDS "slot", a 128-column circular buffer of {H, Eagg}:
//       - STAGED slot (predecessor in an earlier band, or the virtual init row): refilled from HBM once per 64 steps;
//       - RING slot (predecessor in the same band, not the left neighbour): written by its owner lane every step;
//     the value a lane needs in step t+1 is in LDS by the end of step t-1, so the reads are issued one step ahead;
//   * bands run concurrently, band b trailing band b-1 by T+64 steps, one launch per T-step tile (the kernel boundary
//     is the only synchronisation, as in K2); per-lane state between tiles is 5 ints.
// H/Eagg rows and the three trace arrays are stored SKEWED (row r's column c at index c + (r & 63)), so that what a
// wave writes in a step group is aligned: 16 B per lane and array every 4 (H, Eagg) / 16 (trace bytes) steps.
// Fast-path limits, checked on the host: in-degree <= 4, at most kSkSlots slots per band; other graphs run K4.
//
// STATUS: parity-green (tests/test_poa_fill_gpu.py with poa_skewed=1) but OFF by default: measured on gfx950 it is
// slower than K4's tile wavefront at every size tried (README example 9.5 vs 4.8 ms for the first fill, 4 x 60 kbp
// 0.83 vs 0.38 s): a step costs ~125 VALU + ~45 SALU/waits for 64 cells — the same instructions per cell as K4's row
// sweep (240 VALU per 128 cells), because the predecessor selection replaces the scan it saves — while the band
// pipeline is deeper (2 steps of skew per node).  Kept as a documented experiment; see DESIGN.md section 5.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/tsta_engine.h"
#include "engine.h"
#include "poa_types.h"
#include "wave_ops.h"

namespace tsta {

constexpr int kSkSlots = 32;            // LDS predecessor slots per band (1 KiB each)
constexpr int kSkMaxDeg = 4;
constexpr int kSkState = 5;             // ints of per-lane state between tiles

struct PoaSkArgs {
    PoaArgs a;
    const uint32_t* desc;               // [V]  slot byte of predecessor j in bits 8j..8j+7: 0xFE left neighbour (DPP), 0xFF none
    const int* h0;                      // [V]  H(r, column 0)      (c-t-simd.c:187,194-200)
    const int8_t* ring_of;              // [V]  ring slot this node's lane feeds, -1: none
    const int4* bslot;                  // [bands][kSkSlots]: x HBM row slot of the predecessor, y its lane (row skew),
                                        //                    z kind (0 staged, 1 ring), w reader lane (staged) / owner lane (ring)
    const uint8_t* nbslot;              // [bands] slots in use
    const uint8_t* maxdeg;              // [bands] largest in-degree (>= 1)
    int* state;                         // [bands][kSkState][64]
    int T, d, skew, nbands, b_lo;
};

__device__ __forceinline__ int sk_shr1(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int sk_shl1(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, 0x130, 0xf, 0xf, false); }

__global__ __launch_bounds__(64) void poa_skewed_kernel(PoaSkArgs P)
{
    __shared__ int2 ring[kSkSlots][128];
    const PoaArgs& A = P.a;
    const int lane = threadIdx.x;
    const int b = P.b_lo + blockIdx.x;
    const int S_total = (A.n + 64 + 15) / 16 * 16;                        // steps of a band, multiple of 16
    int s0 = P.d * P.T - b * P.skew, s1 = s0 + P.T;
    s0 = max(0, min(s0, S_total)); s1 = max(0, min(s1, S_total));
    if (s0 >= s1) return;
    const int n = A.n, M = A.M, X = A.X, E = A.E, O = A.O, OE = A.O + A.E;
    const int r = b * 64 + lane;
    const bool valid = r < A.V;
    const int myslot = valid ? A.slot[r] : 0;
    const uint32_t desc = valid ? P.desc[r] : 0xffffffffu;
    const int rbase = valid ? (int)A.base[r] : -2;
    const int h0 = valid ? P.h0[r] : 0;
    const int myring = valid ? (int)P.ring_of[r] : -1;
    const int nsl = P.nbslot[b], mdeg = P.maxdeg[b];
    int* Hr = A.Hrow + (size_t)myslot * A.pitch + lane;                   // skewed: H(r, col) at Hr[col + 3]
    int* st = P.state + (size_t)b * kSkState * 64 + lane;

    // ---- lane state
    int fbar, opn, hlast, elast, hd;
    if (s0 == 0) {
        fbar = h0 + O; opn = 1;                                           // F(r,0) = -inf: the first F is an open
        hlast = h0; elast = kNeg;                                         // "last computed" = column 0
        hd = 0;
        if (valid) Hr[3] = h0;                                            // H(r, 0)
        if (myring >= 0) ring[myring][0] = make_int2(h0, kNeg);
    } else {
        fbar = st[0]; opn = st[64]; hlast = st[128]; elast = st[192]; hd = st[256];
    }
    // ---- slot descriptors: one per lane, broadcast with v_readlane (no dependent global loads in the slot loops)
    int4 myd = make_int4(0, 0, 0, 0);
    if (lane < nsl) myd = P.bslot[(size_t)b * kSkSlots + lane];
    auto slot_desc = [&](int sl) {
        return make_int4(__builtin_amdgcn_readlane(myd.x, sl), __builtin_amdgcn_readlane(myd.y, sl),
                         __builtin_amdgcn_readlane(myd.z, sl), __builtin_amdgcn_readlane(myd.w, sl));
    };
    // ---- slots: the columns the readers will touch first.  Staged: reader lane w reads columns x, x+1 now and x+2
    // onwards later (x = s0 - w).  Ring: everything the owner (lane w) has produced so far that a reader may still
    // want (<= 66 columns back), read back from the rows in HBM when a tile resumes.  Loads of 4 slots in flight at once.
    for (int sl0 = 0; sl0 < nsl; sl0 += 4) {
        int2 val[4][2]; int col[4][2]; bool ok[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sl = sl0 + u;
#pragma unroll
            for (int v = 0; v < 2; ++v) { ok[u][v] = false; col[u][v] = 0; val[u][v] = make_int2(0, 0); }
            if (sl < nsl) {
                const int4 d4 = slot_desc(sl);
                const int* Hp = A.Hrow + (size_t)d4.x * A.pitch + d4.y;
                const int* Ep = A.Erow + (size_t)d4.x * A.pitch + d4.y;
                const int c_lo = (d4.z == 0) ? s0 - d4.w : s0 - 65;
                const int c_hi = (d4.z == 0) ? s0 - d4.w + 1 : min(s0 - d4.w, n);
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int c = c_lo + lane + 64 * v;
                    if (c <= c_hi && c >= 0 && c <= n && (d4.z == 0 || s0 > 0)) {
                        ok[u][v] = true; col[u][v] = c;
                        val[u][v] = make_int2(Hp[c + 3], c >= 1 ? Ep[c - 1] : kNeg);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)
                if (ok[u][v]) ring[sl0 + u][col[u][v] & 127] = val[u][v];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0xc07f);                                    // lgkmcnt(0): slot contents written
    // prefetched predecessor values for the first step: pre0[j] = H(p_j, x), pre1[j] = {H(p_j, x+1), Eagg(p_j, x+1)}
    int pre0[kSkMaxDeg]; int2 pre1[kSkMaxDeg];
    {
        const int x = s0 - lane;
#pragma unroll
        for (int j = 0; j < kSkMaxDeg; ++j) {
            const unsigned sl = (desc >> (8 * j)) & 0xffu;
            pre0[j] = 0; pre1[j] = make_int2(0, 0);
            if (sl < (unsigned)kSkSlots) {
                if (x >= 0) pre0[j] = ring[sl][x & 127].x;
                if (x + 1 >= 0) pre1[j] = ring[sl][(x + 1) & 127];
            }
        }
    }
    unsigned tr1[4], tr2[4], tr3[4];
    int cch = 0, ach = 0;
    for (int t0 = s0; t0 < s1; t0 += 64) {
        // ---- per 64 steps: the next 64 sequence characters (rotate through lane 0), fresh columns for the staged slots
        { const int c = t0 + lane; ach = (c < n) ? (int)A.seq[c] : -1; }
        if (t0 == s0 && s0 > 0) {                                        // resume: the characters already under way
            const int x = s0 - 1 - lane; cch = (x >= 0 && x < n) ? (int)A.seq[x] : -1;
        }
        for (int sl0 = 0; sl0 < nsl; sl0 += 4) {                         // loads of 4 slots in flight at once
            int2 val[4]; int col[4]; bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int sl = sl0 + u;
                ok[u] = false; col[u] = 0; val[u] = make_int2(0, 0);
                if (sl < nsl) {
                    const int4 d4 = slot_desc(sl);
                    const int c = t0 - d4.w + 2 + lane;                  // columns the reader prefetches during this block
                    if (d4.z == 0 && c >= 0 && c <= n) {
                        const int* Hp = A.Hrow + (size_t)d4.x * A.pitch + d4.y;
                        const int* Ep = A.Erow + (size_t)d4.x * A.pitch + d4.y;
                        ok[u] = true; col[u] = c;
                        val[u] = make_int2(Hp[c + 3], c >= 1 ? Ep[c - 1] : kNeg);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (ok[u]) ring[sl0 + u][col[u] & 127] = val[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int tend = min(t0 + 64, s1);
#pragma unroll 1
        for (int tq = t0; tq < tend; tq += 16) {                          // s0, s1, t0 are multiples of 16
#pragma unroll
            for (int k16 = 0; k16 < 4; ++k16) {
                unsigned w1 = 0, w2 = 0, w3 = 0;
                int hq[4], eq[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int t = tq + 4 * k16 + k4;
                    const int x = t - lane;
                    const bool act = valid && x >= 0 && x < n;
                    cch = sk_shr1(ach, cch);
                    ach = sk_shl1(ach, ach);
                    const int hvB = sk_shr1(0, hlast), evB = sk_shr1(kNeg, elast);   // left neighbour: H / Eagg(p, x+1)
                    const bool mt = (cch == rbase);
                    const int sc = mt ? M : X;
                    int bq = kNeg, jq = 0, be = kNeg, je = 0;
#pragma unroll
                    for (int j = 0; j < kSkMaxDeg; ++j) {
                        if (j < mdeg) {
                            const unsigned sl = (desc >> (8 * j)) & 0xffu;
                            const bool bb = sl == 0xfeu;
                            const int hv0 = bb ? hd : pre0[j];
                            const int hv1 = bb ? hvB : pre1[j].x;
                            const int ev = bb ? evB : pre1[j].y;
                            if (sl != 0xffu) {
                                const int qv = hv0 + sc;                               // diagonal from H(p, x)
                                if (qv > bq) { bq = qv; jq = j; }                      // lowest predecessor index on ties
                                const int eo = hv1 + OE, ex = ev + E;                  // c-t-simd.c:524-528
                                const int ein = max(ex, eo);
                                if (ein > be) { be = ein; je = j | ((ein == eo) ? 64 : 0); }
                            }
                            // next step's values: H(p, x+1) becomes the diagonal source, column x+2 is read now
                            pre0[j] = pre1[j].x;
                            if (sl < (unsigned)kSkSlots && x + 2 >= 0) pre1[j] = ring[sl][(x + 2) & 127];
                        }
                    }
                    const int f = fbar + E;
                    const int h = max(max(bq, be), f);
                    const int g = h + O;
                    const int opn_next = (g >= f) ? 1 : 0;                             // F(x+1) opens  <=>  H(x)+O >= F(x)
                    int code;                                                          // match > mismatch > E > F (c-t-simd.c:485-506)
                    if (bq == h) code = (mt ? 42 : 84) + jq;
                    else if (be == h) code = je & 63;
                    else code = 126;
                    int ecode = ((je & 64) ? 43 : 1) + (je & 63);                      // c-t-simd.c:530-538
                    if (be + E == h + OE) ecode = -ecode;
                    int fcode = opn ? 2 : 1;                                           // c-t-simd.c:521-522,564-565
                    if (f + E == h + OE) fcode = -fcode;
                    if (act) {
                        hd = hvB;
                        fbar = max(f, g); opn = opn_next; hlast = h; elast = be;
                        if (myring >= 0) ring[myring][(x + 1) & 127] = make_int2(h, be);
                        if (x == n - 1) A.lastscore[r] = h;
                    } else if (valid && x < 0) {
                        hd = hvB;                                                      // the left neighbour's H(., 0) on the way in
                    }
                    // outputs, aligned in step space; before the first / after the last column the row repeats its edge value
                    hq[k4] = hlast; eq[k4] = elast;
                    w1 |= (unsigned)(uint8_t)code << (8 * k4);
                    w2 |= (unsigned)(uint8_t)ecode << (8 * k4);
                    w3 |= (unsigned)(uint8_t)fcode << (8 * k4);
                }
                {
                    const int tb = tq + 4 * k16;                                       // first step of the group of 4
                    if (valid && tb - lane + 3 >= 0 && tb - lane < n) {
                        // H(r, x+1) of step t sits at index lane + (x+1) + 3 = t + 4 of the row, Eagg(r, x+1) at index t
                        *reinterpret_cast<int4*>(A.Hrow + (size_t)myslot * A.pitch + tb + 4) = make_int4(hq[0], hq[1], hq[2], hq[3]);
                        *reinterpret_cast<int4*>(A.Erow + (size_t)myslot * A.pitch + tb) = make_int4(eq[0], eq[1], eq[2], eq[3]);
                    }
                    tr1[k16] = w1; tr2[k16] = w2; tr3[k16] = w3;
                }
            }
            if (A.source && valid && tq - lane + 15 >= 0 && tq - lane < n) {
                *reinterpret_cast<uint4*>(A.source + (size_t)r * A.tpitch + tq) = make_uint4(tr1[0], tr1[1], tr1[2], tr1[3]);
                *reinterpret_cast<uint4*>(A.esource + (size_t)r * A.tpitch + tq) = make_uint4(tr2[0], tr2[1], tr2[2], tr2[3]);
                *reinterpret_cast<uint4*>(A.fsource + (size_t)r * A.tpitch + tq) = make_uint4(tr3[0], tr3[1], tr3[2], tr3[3]);
            }
        }
    }
    st[0] = fbar; st[64] = opn; st[128] = hlast; st[192] = elast; st[256] = hd;
}

}  // namespace tsta

namespace tsta {

int poa_run_skewed(Engine* e, const tsta_params* p, const tsta_dag_csr* dag, const uint8_t* seq, int32_t n,
                   bool want_trace, PoaArgs& A, bool* used)
{
    *used = false;
    const int V = dag->n_nodes;
    const int nbands = (V + 63) / 64;
    // ---- per-band predecessor tables; bail out (tile kernel) when the graph leaves the fast path
    std::vector<uint32_t> desc((size_t)V, 0xffffffffu);
    std::vector<int8_t> ring_of((size_t)V, -1);
    std::vector<uint8_t> nbslot((size_t)nbands, 0), maxdeg((size_t)nbands, 1);
    struct SlotRef { int pred, lp, kind, who; };               // pred: node index, -1 = the virtual init row
    std::vector<SlotRef> slots((size_t)nbands * kSkSlots);
    std::vector<int> h0((size_t)V);
    for (int b = 0; b < nbands; ++b) {
        int ns = 0, md = 1;
        int ring_slot_of_lane[64];
        for (int l = 0; l < 64; ++l) ring_slot_of_lane[l] = -1;
        for (int l = 0; l < 64; ++l) {
            const int r = b * 64 + l;
            if (r >= V) break;
            const int k0 = dag->pred_off[r], np = dag->pred_off[r + 1] - k0;
            const int deg = np == 0 ? 1 : np;                  // a source reads the virtual init row (c-t-simd.c:174-188)
            if (deg > kSkMaxDeg) return TSTA_OK;
            md = std::max(md, deg);
            uint32_t d = 0xffffffffu;
            for (int j = 0; j < deg; ++j) {
                const int pr = np == 0 ? -1 : dag->pred[k0 + j];
                int sl;
                if (pr >= 0 && pr == r - 1 && l > 0) sl = 0xfe;
                else if (pr >= 0 && pr / 64 == b) {            // same band, not the left neighbour: ring slot of its lane
                    const int lp = pr % 64;
                    if (ring_slot_of_lane[lp] < 0) {
                        if (ns >= kSkSlots) return TSTA_OK;
                        slots[(size_t)b * kSkSlots + ns] = SlotRef{pr, lp, 1, lp};
                        ring_slot_of_lane[lp] = ns++;
                    }
                    sl = ring_slot_of_lane[lp];
                } else {                                       // earlier band / init row: one staged slot per reference
                    if (ns >= kSkSlots) return TSTA_OK;
                    slots[(size_t)b * kSkSlots + ns] = SlotRef{pr, pr < 0 ? 0 : pr % 64, 0, l};
                    sl = ns++;
                }
                d = (d & ~(0xffu << (8 * j))) | ((uint32_t)sl << (8 * j));
            }
            desc[r] = d;
            // first column (c-t-simd.c:187,194-200): sources O+E, else max over the predecessors' + E
            if (np == 0) h0[r] = p->O + p->E;
            else {
                int hm = kNeg;
                for (int j = 0; j < np; ++j) hm = std::max(hm, h0[dag->pred[k0 + j]]);
                h0[r] = hm + p->E;
            }
        }
        for (int l = 0; l < 64; ++l) if (ring_slot_of_lane[l] >= 0 && b * 64 + l < V) ring_of[(size_t)b * 64 + l] = (int8_t)ring_slot_of_lane[l];
        nbslot[b] = (uint8_t)ns; maxdeg[b] = (uint8_t)md;
    }

    int rc = e->check(hipSetDevice(e->device), "hipSetDevice");
    if (rc != TSTA_OK) return rc;
    hipStream_t s = e->stream;
    const int T = 64, skew = T + 64;
    const int S_total = (n + 64 + 15) / 16 * 16;
    const int64_t pitch = ((int64_t)S_total + 16 + 3) / 4 * 4 + 8;        // row index = step + 4 (H) / step (Eagg)
    const int64_t tpitch = ((int64_t)S_total + 15) / 16 * 16 + 16;
    // Row slots: node r's row is read by the bands of its successors while they sweep; a slot is handed to a new row
    // only once every such band has finished, i.e. to bands that START after the last reader band ENDS.
    std::vector<int> slot((size_t)V);
    int nslots = 1;                                                       // slot 0 = virtual init row
    {
        const int lag = (S_total + T + skew - 1) / skew + 1;              // bands between a band's start and its end, in starts
        std::vector<int> last_use((size_t)V);
        for (int r = 0; r < V; ++r) last_use[r] = r;
        for (int r = 0; r < V; ++r)
            for (int k = dag->pred_off[r]; k < dag->pred_off[r + 1]; ++k) last_use[dag->pred[k]] = std::max(last_use[dag->pred[k]], r);
        std::vector<std::pair<int, int>> heap;                            // (first band allowed to reuse, slot), min-heap
        auto cmp = [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; };
        for (int r = 0; r < V; ++r) {
            const int b = r / 64;
            if (!heap.empty() && heap.front().first <= b) {
                std::pop_heap(heap.begin(), heap.end(), cmp);
                slot[r] = heap.back().second;
                heap.pop_back();
            } else slot[r] = nslots++;
            heap.emplace_back(last_use[r] / 64 + lag + 1, slot[r]);
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
    }
    std::vector<int4> bslot((size_t)nbands * kSkSlots);
    for (int b = 0; b < nbands; ++b)
        for (int i = 0; i < nbslot[b]; ++i) {
            const SlotRef& sr = slots[(size_t)b * kSkSlots + i];
            bslot[(size_t)b * kSkSlots + i] = make_int4(sr.pred < 0 ? 0 : slot[sr.pred], sr.lp, sr.kind, sr.who);
        }

    const int npred = dag->pred_off[V];
    auto up16 = [](size_t x) { return (x + 15) / 16 * 16; };
    const size_t o_off = 0, o_pred = o_off + up16((size_t)(V + 1) * 4), o_slot = o_pred + up16((size_t)std::max(npred, 1) * 4),
                 o_desc = o_slot + up16((size_t)V * 4), o_h0 = o_desc + up16((size_t)V * 4), o_bslot = o_h0 + up16((size_t)V * 4),
                 o_ring = o_bslot + up16(bslot.size() * sizeof(int4)), o_nbs = o_ring + up16((size_t)V), o_mdg = o_nbs + up16((size_t)nbands),
                 o_base = o_mdg + up16((size_t)nbands), o_sink = o_base + up16((size_t)V), o_seq = o_sink + up16((size_t)V),
                 meta = o_seq + up16((size_t)n) + 64;
    const size_t rows_bytes = (size_t)nslots * pitch * sizeof(int);
    if ((rc = e->reserve(e->in_meta, meta, "hipMalloc(dag)")) ||
        (rc = e->reserve(e->rows, 2 * rows_bytes, "hipMalloc(poa rows)")) ||
        (rc = e->reserve(e->out_scores, (size_t)V * 4 + 64, "hipMalloc(lastscore)")) ||
        (rc = e->reserve(e->scratch, (size_t)nbands * kSkState * 64 * sizeof(int) + 64, "hipMalloc(poa state)")))
        return rc;
    if (want_trace && (rc = e->reserve(e->trace, 3 * (size_t)V * tpitch + 64, "hipMalloc(poa trace)"))) return rc;
    char* m = reinterpret_cast<char*>(e->in_meta.p);
    auto up = [&](size_t off, const void* src, size_t bytes, const char* what) {
        return bytes ? e->check(hipMemcpyAsync(m + off, src, bytes, hipMemcpyHostToDevice, s), what) : (int)TSTA_OK;
    };
    if ((rc = up(o_off, dag->pred_off, (size_t)(V + 1) * 4, "H2D pred_off")) || (rc = up(o_pred, dag->pred, (size_t)npred * 4, "H2D pred")) ||
        (rc = up(o_slot, slot.data(), (size_t)V * 4, "H2D slots")) || (rc = up(o_desc, desc.data(), (size_t)V * 4, "H2D desc")) ||
        (rc = up(o_h0, h0.data(), (size_t)V * 4, "H2D h0")) || (rc = up(o_bslot, bslot.data(), bslot.size() * sizeof(int4), "H2D band slots")) ||
        (rc = up(o_ring, ring_of.data(), (size_t)V, "H2D ring")) || (rc = up(o_nbs, nbslot.data(), (size_t)nbands, "H2D nbslot")) ||
        (rc = up(o_mdg, maxdeg.data(), (size_t)nbands, "H2D maxdeg")) || (rc = up(o_base, dag->base, (size_t)V, "H2D base")) ||
        (rc = up(o_sink, dag->is_sink, (size_t)V, "H2D is_sink")) || (rc = up(o_seq, seq, (size_t)n, "H2D seq")))
        return rc;
    if ((rc = e->check(hipStreamSynchronize(s), "poa upload sync"))) return rc;   // the host vectors must outlive the copies

    A.V = V; A.n = n;
    A.pred_off = reinterpret_cast<const int*>(m + o_off); A.pred = reinterpret_cast<const int*>(m + o_pred);
    A.base = reinterpret_cast<const uint8_t*>(m + o_base); A.is_sink = reinterpret_cast<const uint8_t*>(m + o_sink);
    A.seq = reinterpret_cast<const uint8_t*>(m + o_seq);
    A.slot = reinterpret_cast<const int*>(m + o_slot); A.pred_slot = nullptr;
    A.Hrow = reinterpret_cast<int*>(e->rows.p);
    A.Erow = A.Hrow + (size_t)nslots * pitch;
    A.pitch = pitch; A.tpitch = tpitch;
    A.carry = nullptr; A.cflag = nullptr; A.nchunks = 0; A.nrb = 0; A.kc = 0; A.d = 0; A.q_lo = 0;
    A.source = want_trace ? reinterpret_cast<int8_t*>(e->trace.p) : nullptr;
    A.esource = want_trace ? A.source + (size_t)V * tpitch : nullptr;
    A.fsource = want_trace ? A.esource + (size_t)V * tpitch : nullptr;
    A.lastscore = reinterpret_cast<int*>(e->out_scores.p);
    A.best = reinterpret_cast<int*>(e->counter.p);
    A.M = p->M; A.X = p->X; A.E = p->E; A.O = p->O;
    A.skew = 1;

    PoaSkArgs P;
    P.a = A;
    P.desc = reinterpret_cast<const uint32_t*>(m + o_desc); P.h0 = reinterpret_cast<const int*>(m + o_h0);
    P.ring_of = reinterpret_cast<const int8_t*>(m + o_ring); P.bslot = reinterpret_cast<const int4*>(m + o_bslot);
    P.nbslot = reinterpret_cast<const uint8_t*>(m + o_nbs); P.maxdeg = reinterpret_cast<const uint8_t*>(m + o_mdg);
    P.state = reinterpret_cast<int*>(e->scratch.p);
    P.T = T; P.skew = skew; P.nbands = nbands;

    e->timing_begin(s);
    launch_poa_init(A, s);
    int launches = 1;
    const int64_t D = ((int64_t)S_total + (int64_t)(nbands - 1) * skew + T - 1) / T;
    for (int64_t d = 0; d < D; ++d) {
        // bands with work: S_b(d) = (d+1)T - b*skew > 0  and  dT - b*skew < S_total
        int64_t b_hi = ((d + 1) * T + skew - 1) / skew;                   // first b with S_b(d) <= 0
        int64_t b_lo = (d * T - S_total < 0) ? 0 : (d * T - S_total) / skew;
        b_hi = std::min<int64_t>(b_hi, nbands);
        if (b_lo >= b_hi) continue;
        P.d = (int)d; P.b_lo = (int)b_lo;
        hipLaunchKernelGGL(poa_skewed_kernel, dim3((unsigned)(b_hi - b_lo)), dim3(64), 0, s, P);
        ++launches;
    }
    launch_poa_best(A, s);
    e->n_launches = launches + 1;
    *used = true;
    return e->check(hipGetLastError(), "poa_fill (skewed) launch");
}

}  // namespace tsta
