// poa_fill.hip — K4: sequence-vs-POA-graph affine fill for gfx950 (tsta_poa_fill).
//
// Replaces one control() wavefront of TSTA_msa: block_alignment()/block_line_alignment()
// (msa/c-t-simd.c:602-657, 165-599) over all nodes in topological order, producing the reference's trace codes
// source/esource/fsource (c-t-simd.c:485-506, 530-538, 521-522/564-565) and the sinks' lastsorce (:577-578).
// Arithmetic is the plain int32 partial-order DP the reference's saturating-int8 machinery encodes.
//
// v1 mapping (round 1): rows (graph nodes) are inherently ordered by the DAG, columns are not — so one
// workgroup sweeps the nodes in topological order and spreads the COLUMNS of a node over its threads, 8
// consecutive columns per thread:
//   * predecessor rows H(p,.) / Eagg(p,.) are read back from HBM/L2 as two aligned 16-byte loads per array
//     (rows are stored with a one-column shift for Eagg so both are 16-B aligned), any number of
//     predecessors, lowest predecessor index wins ties (strict > update in ascending order);
//   * the horizontal gap F is a max-plus prefix scan: thread-local pass, wave scan by __shfl_up, cross-wave
//     through LDS, then a fix-up pass that also emits the three trace codes;
//   * one __syncthreads() per row publishes the row to the later rows (same CU, write-through L1).
// The reference's lazy-F scalar fix-up loops (c-t-simd.c:293-329, 408-432) have no counterpart.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/tsta_engine.h"
#include "engine.h"

namespace tsta {

constexpr int kNeg = -1000000000;
constexpr int kCols = 8;                 // columns per thread per segment

struct PoaArgs {
    int V, n;
    const int* pred_off; const int* pred; const uint8_t* base; const uint8_t* is_sink;
    const uint8_t* seq;                  // n bytes
    int* Hrow;                           // [V+1][pitch]: Hrow[r][c] = H(r,c), c = 0..n      (row V = virtual init)
    int* Erow;                           // [V+1][pitch]: Erow[r][c] = Eagg(r,c+1), c = 0..n-1
    int64_t pitch;                       // ints per row, multiple of 8, >= n + 8
    int8_t* source; int8_t* esource; int8_t* fsource; int64_t tpitch;   // [V][tpitch] device layout (tpitch mult. of 8)
    int* lastscore;                      // [V]
    int* best;                           // [0] = best sink
    int M, X, E, O;
};

__global__ __launch_bounds__(1024) void poa_fill_kernel(PoaArgs A)
{
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wv = tid >> 6;
    const int n = A.n, M = A.M, X = A.X, E = A.E, O = A.O, OE = A.O + A.E;
    const int a8 = kCols * E;
    __shared__ int wtot[16];
    __shared__ int segcarry;             // Fbar entering the current segment
    __shared__ int openflag[1025];       // "F at next column is an open" of each thread's last column; [0] = segment-in

    // virtual init row (c-t-simd.c:82-94,174-188): H = 0, O+c*E ; no vertical gap state
    {
        int* Hi = A.Hrow + (size_t)A.V * A.pitch; int* Ei = A.Erow + (size_t)A.V * A.pitch;
        for (int c = tid; c <= n; c += T) { Hi[c] = c == 0 ? 0 : O + c * E; Ei[c] = kNeg; }
    }
    __syncthreads();

    const int seg_cols = T * kCols;
    for (int r = 0; r < A.V; ++r) {
        int p0 = A.pred_off[r], np = A.pred_off[r + 1] - p0;
        const bool is_src = (np == 0);                       // in==0 -> virtual init predecessor (c-t-simd.c:174-188)
        int* Hr = A.Hrow + (size_t)r * A.pitch;
        int* Er = A.Erow + (size_t)r * A.pitch;
        const int bs = A.base[r];
        // first column (c-t-simd.c:187,194-200)
        int h0;
        if (is_src) h0 = OE;
        else {
            h0 = kNeg;
            for (int j = 0; j < np; ++j) h0 = max(h0, A.Hrow[(size_t)A.pred[p0 + j] * A.pitch]);
            h0 += E;
        }
        if (tid == 0) { Hr[0] = h0; segcarry = h0 + O; openflag[0] = 1; }   // F(r,0) = -inf: first F is an open
        __syncthreads();

        for (int sb = 0; sb < n; sb += seg_cols) {
            const int c0 = sb + tid * kCols;                 // this thread owns columns c0+1 .. c0+8 (1-based)
            int bq[kCols], jq[kCols], be[kCols], je[kCols], sc[kCols];
            bool mt[kCols];
#pragma unroll
            for (int k = 0; k < kCols; ++k) {
                const int c = c0 + k;                        // 0-based column index of seq
                const int ch = (c < n) ? (int)A.seq[c] : -1;
                mt[k] = (ch == bs);
                sc[k] = mt[k] ? M : X;
                bq[k] = kNeg; jq[k] = 0; be[k] = kNeg; je[k] = 0;
            }
            if (c0 < n) {
                const int npp = is_src ? 1 : np;
                for (int j = 0; j < npp; ++j) {
                    const int p = is_src ? A.V : A.pred[p0 + j];
                    const int* Hp = A.Hrow + (size_t)p * A.pitch + c0;     // H(p, c0 .. c0+8)
                    const int* Ep = A.Erow + (size_t)p * A.pitch + c0;     // Eagg(p, c0+1 .. c0+8)
                    const int4 ha = *reinterpret_cast<const int4*>(Hp);
                    const int4 hb = *reinterpret_cast<const int4*>(Hp + 4);
                    const int h8 = Hp[8];
                    const int4 ea = *reinterpret_cast<const int4*>(Ep);
                    const int4 eb = *reinterpret_cast<const int4*>(Ep + 4);
                    const int hv[9] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w, h8};
                    const int ev[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
#pragma unroll
                    for (int k = 0; k < kCols; ++k) {
                        const int q = hv[k] + sc[k];                       // diagonal from H(p, c-1)
                        if (q > bq[k]) { bq[k] = q; jq[k] = j; }
                        const int eo = hv[k + 1] + OE, ex = ev[k] + E;     // c-t-simd.c:524-528
                        const int ein = max(ex, eo);
                        if (ein > be[k]) { be[k] = ein; je[k] = j | ((ein == eo) ? 64 : 0); }
                    }
                }
            }
            // ---- F: local pass with no carry-in
            int fbar = kNeg;
#pragma unroll
            for (int k = 0; k < kCols; ++k) {
                const int f = fbar + E;
                const int h = max(max(bq[k], be[k]), f);
                fbar = max(f, h + O);
            }
            // ---- max-plus inclusive scan of (b, shift a8 per thread)
            int v = fbar;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(v, off, 64);
                if (lane >= off) v = max(v, u + off * a8);
            }
            if (lane == 63) wtot[wv] = v;
            __syncthreads();
            int pw = kNeg;                                   // prefix of the previous waves, aligned to lane -1
            for (int w2 = 0; w2 < wv; ++w2) pw = max(pw, wtot[w2] + (wv - 1 - w2) * 64 * a8);
            const int incl = max(v, pw + (lane + 1) * a8);   // I_g
            int prev = __shfl_up(incl, 1, 64);               // I_{g-1}
            if (lane == 0) prev = pw;                        // previous wave's inclusive value at its lane 63
            const int sc_in = segcarry;
            int xin = max(prev, sc_in + tid * a8);           // Fbar entering this thread's first column
            if (tid == 0) xin = sc_in;
            __syncthreads();                                 // everyone has read segcarry / wtot
            if (tid == T - 1) segcarry = max(incl, sc_in + T * a8);
            // ---- fix-up pass with the true carry: final H, Eagg, trace codes
            fbar = xin;
            int hl[kCols], fl[kCols];
            bool opn[kCols + 1];                             // opn[k]: F at column k is an open (from column k-1)
            opn[0] = false;
#pragma unroll
            for (int k = 0; k < kCols; ++k) {
                const int f = fbar + E;
                const int h = max(max(bq[k], be[k]), f);
                hl[k] = h; fl[k] = f;
                const int g = h + O;
                opn[k + 1] = (g >= f);                       // F(c+1) == H(c)+O+E  <=>  H(c)+O >= F(c)
                fbar = max(f, g);
            }
            openflag[tid + 1] = opn[kCols] ? 1 : 0;
            bool f0neg = false;                              // sign of fsource at the thread's first column
            if (c0 < n) {
                int hs[kCols], es[kCols];
                int8_t s1[kCols], s2[kCols], s3[kCols];
#pragma unroll
                for (int k = 0; k < kCols; ++k) {
                    const int h = hl[k], f = fl[k];
                    hs[k] = h; es[k] = be[k];
                    // source: match > mismatch > E > F, lowest predecessor index (c-t-simd.c:485-506)
                    int code;
                    if (bq[k] == h) code = (mt[k] ? 42 : 84) + jq[k];
                    else if (be[k] == h) code = je[k] & 63;
                    else code = 126;
                    s1[k] = (int8_t)code;
                    int ecode = ((je[k] & 64) ? 43 : 1) + (je[k] & 63);            // c-t-simd.c:530-538
                    if (be[k] + E == h + OE) ecode = -ecode;
                    s2[k] = (int8_t)ecode;
                    int fcode = (k > 0 && opn[k]) ? 2 : 1;                          // k==0 patched after the barrier
                    if (f + E == h + OE) fcode = -fcode;
                    if (k == 0) f0neg = (fcode < 0);
                    s3[k] = (int8_t)fcode;
                }
                // rows for the later nodes (H at c0+1..c0+8 -> index c0+1.., Eagg at c0+1.. -> index c0..)
#pragma unroll
                for (int k = 0; k < kCols; ++k) if (c0 + k < n) { Hr[c0 + 1 + k] = hs[k]; Er[c0 + k] = es[k]; }
                if (A.source) {
                    int8_t* d1 = A.source + (size_t)r * A.tpitch + c0;
                    int8_t* d2 = A.esource + (size_t)r * A.tpitch + c0;
                    int8_t* d3 = A.fsource + (size_t)r * A.tpitch + c0;
                    uint2 w1, w2, w3;
                    w1.x = (uint8_t)s1[0] | ((uint8_t)s1[1] << 8) | ((uint8_t)s1[2] << 16) | ((unsigned)(uint8_t)s1[3] << 24);
                    w1.y = (uint8_t)s1[4] | ((uint8_t)s1[5] << 8) | ((uint8_t)s1[6] << 16) | ((unsigned)(uint8_t)s1[7] << 24);
                    w2.x = (uint8_t)s2[0] | ((uint8_t)s2[1] << 8) | ((uint8_t)s2[2] << 16) | ((unsigned)(uint8_t)s2[3] << 24);
                    w2.y = (uint8_t)s2[4] | ((uint8_t)s2[5] << 8) | ((uint8_t)s2[6] << 16) | ((unsigned)(uint8_t)s2[7] << 24);
                    w3.x = (uint8_t)s3[0] | ((uint8_t)s3[1] << 8) | ((uint8_t)s3[2] << 16) | ((unsigned)(uint8_t)s3[3] << 24);
                    w3.y = (uint8_t)s3[4] | ((uint8_t)s3[5] << 8) | ((uint8_t)s3[6] << 16) | ((unsigned)(uint8_t)s3[7] << 24);
                    *reinterpret_cast<uint2*>(d1) = w1;
                    *reinterpret_cast<uint2*>(d2) = w2;
                    *reinterpret_cast<uint2*>(d3) = w3;
                }
                if (c0 + kCols >= n && c0 < n) {             // owner of the last column
                    const int kl = n - 1 - c0;
                    int hlast = hs[0];
#pragma unroll
                    for (int k = 1; k < kCols; ++k) if (k == kl) hlast = hs[k];
                    A.lastscore[r] = hlast;
                }
            }
            __syncthreads();                                 // row (segment) + open flags published
            if (c0 < n && A.fsource) {
                // first column of the thread: the open flag comes from the thread on the left
                if (openflag[tid]) A.fsource[(size_t)r * A.tpitch + c0] = (int8_t)(f0neg ? -2 : 2);   // +-1 -> +-2
            }
            __syncthreads();
            if (tid == 0) openflag[0] = openflag[T];         // carry the flag into the next segment
            __syncthreads();
        }
    }
    // best sink: max lastscore, ties -> lowest index (c-t-simd.c:737-750)
    __syncthreads();
    if (tid == 0) {
        int best = -1, bv = 0;
        for (int r = 0; r < A.V; ++r)
            if (A.is_sink[r] && (best < 0 || A.lastscore[r] > bv)) { best = r; bv = A.lastscore[r]; }
        A.best[0] = best;
    }
}

}  // namespace tsta

extern "C" int tsta_poa_fill(tsta_engine* eng, const tsta_params* p, const tsta_dag_csr* dag, const uint8_t* seq,
                             int32_t n, int8_t* source, int8_t* esource, int8_t* fsource, int32_t* lastscore,
                             int32_t* best_sink)
{
    using namespace tsta;
    if (!eng || !p || !dag || n < 0 || dag->n_nodes < 0) return TSTA_ERR_INVALID_ARG;
    Engine* e = reinterpret_cast<Engine*>(eng);
    const int V = dag->n_nodes;
    if (best_sink) *best_sink = -1;
    if (V == 0 || n == 0) return TSTA_OK;
    if (!dag->pred_off || !dag->base || !dag->is_sink || !seq) return TSTA_ERR_INVALID_ARG;
    const bool want_trace = source || esource || fsource;
    if (want_trace && !(source && esource && fsource)) return TSTA_ERR_INVALID_ARG;
    const int npred = dag->pred_off[V];
    for (int r = 0; r < V; ++r) {
        const int k0 = dag->pred_off[r], k1 = dag->pred_off[r + 1];
        if (k1 < k0 || k1 - k0 >= 42) { e->set_error("node %d: %d predecessors (must be < 42)", r, k1 - k0); return TSTA_ERR_UNSUPPORTED; }
        for (int k = k0; k < k1; ++k)
            if (dag->pred[k] < 0 || dag->pred[k] >= r) { e->set_error("node %d: predecessor %d not earlier in topological order", r, dag->pred[k]); return TSTA_ERR_INVALID_ARG; }
    }
    int rc = e->check(hipSetDevice(e->device), "hipSetDevice");
    if (rc != TSTA_OK) return rc;
    hipStream_t s = e->stream;

    const int64_t pitch = ((int64_t)n + 8 + 7) / 8 * 8 + 8;
    const int64_t tpitch = ((int64_t)n + 7) / 8 * 8;
    const size_t meta = (size_t)(V + 1) * 4 + (size_t)std::max(npred, 1) * 4 + (size_t)V * 2 + (size_t)n + 64;
    const size_t rows_bytes = (size_t)(V + 1) * pitch * sizeof(int);
    if ((rc = e->reserve(e->in_meta, meta + 64, "hipMalloc(dag)")) ||
        (rc = e->reserve(e->rows, 2 * rows_bytes, "hipMalloc(poa rows)")) ||
        (rc = e->reserve(e->out_scores, (size_t)V * 4 + 64, "hipMalloc(lastscore)")))
        return rc;
    if (want_trace && (rc = e->reserve(e->trace, 3 * (size_t)V * tpitch, "hipMalloc(poa trace)"))) return rc;

    char* m = reinterpret_cast<char*>(e->in_meta.p);
    int* d_off = reinterpret_cast<int*>(m);
    int* d_pred = d_off + (V + 1);
    uint8_t* d_base = reinterpret_cast<uint8_t*>(d_pred + std::max(npred, 1));
    uint8_t* d_sink = d_base + V;
    uint8_t* d_seq = d_sink + V;
    if ((rc = e->check(hipMemcpyAsync(d_off, dag->pred_off, (size_t)(V + 1) * 4, hipMemcpyHostToDevice, s), "H2D pred_off")) ||
        (npred && (rc = e->check(hipMemcpyAsync(d_pred, dag->pred, (size_t)npred * 4, hipMemcpyHostToDevice, s), "H2D pred"))) ||
        (rc = e->check(hipMemcpyAsync(d_base, dag->base, V, hipMemcpyHostToDevice, s), "H2D base")) ||
        (rc = e->check(hipMemcpyAsync(d_sink, dag->is_sink, V, hipMemcpyHostToDevice, s), "H2D is_sink")) ||
        (rc = e->check(hipMemcpyAsync(d_seq, seq, n, hipMemcpyHostToDevice, s), "H2D seq")))
        return rc;

    PoaArgs A;
    A.V = V; A.n = n; A.pred_off = d_off; A.pred = d_pred; A.base = d_base; A.is_sink = d_sink; A.seq = d_seq;
    A.Hrow = reinterpret_cast<int*>(e->rows.p);
    A.Erow = A.Hrow + (size_t)(V + 1) * pitch;
    A.pitch = pitch;
    A.tpitch = tpitch;
    A.source = want_trace ? reinterpret_cast<int8_t*>(e->trace.p) : nullptr;
    A.esource = want_trace ? A.source + (size_t)V * tpitch : nullptr;
    A.fsource = want_trace ? A.esource + (size_t)V * tpitch : nullptr;
    A.lastscore = reinterpret_cast<int*>(e->out_scores.p);
    A.best = reinterpret_cast<int*>(e->counter.p);
    A.M = p->M; A.X = p->X; A.E = p->E; A.O = p->O;

    int threads = ((n + kCols - 1) / kCols + 63) / 64 * 64;
    threads = std::max(64, std::min(1024, threads));
    e->timing_begin(s);
    hipLaunchKernelGGL(poa_fill_kernel, dim3(1), dim3(threads), 0, s, A);
    e->n_launches = 1;
    e->timing_end(s);
    if ((rc = e->check(hipGetLastError(), "poa_fill launch"))) return rc;

    int h_best = -1;
    if ((rc = e->check(hipMemcpyAsync(&h_best, A.best, 4, hipMemcpyDeviceToHost, s), "D2H best"))) return rc;
    if (lastscore && (rc = e->check(hipMemcpyAsync(lastscore, A.lastscore, (size_t)V * 4, hipMemcpyDeviceToHost, s), "D2H lastscore"))) return rc;
    if (want_trace) {
        if ((rc = e->check(hipMemcpy2DAsync(source, n, A.source, tpitch, n, V, hipMemcpyDeviceToHost, s), "D2H source")) ||
            (rc = e->check(hipMemcpy2DAsync(esource, n, A.esource, tpitch, n, V, hipMemcpyDeviceToHost, s), "D2H esource")) ||
            (rc = e->check(hipMemcpy2DAsync(fsource, n, A.fsource, tpitch, n, V, hipMemcpyDeviceToHost, s), "D2H fsource")))
            return rc;
    }
    if ((rc = e->check(hipStreamSynchronize(s), "poa_fill sync"))) return rc;
    if (best_sink) *best_sink = h_best;
    return TSTA_OK;
}
