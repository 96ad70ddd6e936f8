/*
 * poa_oracle.c — scalar restatement of TSTA_msa's sequence-vs-DAG affine fill
 * (msa/c-t-simd.c:165-599 block_line_alignment, driven by :602-657 / :979-1049).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Plain int32 partial-order DP; the reference's saturating
 * int8 offset tracker (c-t-simd.c:196-261,370-404) is an encoding device that is exact inside the
 * domain documented in DESIGN.md (predecessor rows within 60 of the best, 'N'-free sequences, sequence
 * not spanning more L-blocks than the graph).  Parity: PINNED on the README example (README.md:26-40:
 * per-step lastsorce) and on trace-array dumps of the compiled reference (tests/golden/).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-1000000000)

static inline int32_t max2(int32_t a, int32_t b) { return a > b ? a : b; }

int32_t oracle_poa_fill(const oracle_dag* g, const uint8_t* seq, int32_t n, const oracle_params* p,
                        int8_t* source, int8_t* esource, int8_t* fsource, int32_t* lastscore)
{
    const int32_t V = g->n_nodes, M = p->M, X = p->X, E = p->E, OE = p->O + p->E;
    const size_t W = (size_t)n + 1;
    int32_t* H  = (int32_t*)malloc((size_t)(V + 1) * W * sizeof(int32_t));   /* row V = virtual `init` */
    int32_t* Ea = (int32_t*)malloc((size_t)(V + 1) * W * sizeof(int32_t));
    /* virtual predecessor of source nodes: the boundary row (c-t-simd.c:82-94,174-188,663-680) */
    int32_t* Hi = H + (size_t)V * W; int32_t* Ei = Ea + (size_t)V * W;
    Hi[0] = 0; Ei[0] = NEG_INF;
    for (int32_t c = 1; c <= n; c++) { Hi[c] = p->O + c * E; Ei[c] = NEG_INF; }

    for (int32_t r = 0; r < V; r++) {
        int32_t np = g->pred_off[r + 1] - g->pred_off[r];
        const int32_t* pl = g->pred + g->pred_off[r];
        int32_t init_idx = V;
        if (np == 0) { np = 1; pl = &init_idx; }           /* in==0 -> virtual init, c-t-simd.c:174-188 */
        int32_t* Hr = H + (size_t)r * W; int32_t* Er = Ea + (size_t)r * W;
        /* first column: frist_col_sorce = max_j(pre_j.frist_col_sorce) + E; sources E+O (c-t-simd.c:187,194-200) */
        int32_t h0 = NEG_INF;
        for (int32_t j = 0; j < np; j++) h0 = max2(h0, H[(size_t)pl[j] * W]);
        Hr[0] = (pl == &init_idx) ? OE : h0 + E;
        Er[0] = NEG_INF;
        int32_t f = NEG_INF;
        const uint8_t base = g->base[r];
        for (int32_t c = 1; c <= n; c++) {
            const int32_t s = (seq[c - 1] == base) ? M : X;
            /* a literal 'N' in the sequence is indistinguishable from the reference's pad character (readseq,
             * c-t-simd.c:46): the column's delta H(r,c)-H(r,c-1) is forced to 0 for EVERY node (`max = blendv(max,
             * zero, SN)`, c-t-simd.c:461,481) and the trace codes are decided by equality with that forced value
             * (:485-506: lowest j with Q_j == H, else lowest j with Ein_j == H, else F).  Everything downstream (F, Eagg
             * of the successors, the signs of esource / fsource, :508-538) uses the forced H.  NOTE: the reference
             * applies this in its second pass only; its speculative first pass (:357-404) does not, so its own output
             * on inputs with a literal N depends on -W / the SIMD width; this is the rule applied
             * consistently. */
            const int is_n = (seq[c - 1] == 'N');
            const int32_t h_forced = Hr[c - 1];
            int32_t bestq = NEG_INF, jq = 0, beste = NEG_INF, je = 0, je_open = 0, jq_n = -1, je_n = -1;
            for (int32_t j = 0; j < np; j++) {
                const int32_t* Hp = H + (size_t)pl[j] * W; const int32_t* Ep = Ea + (size_t)pl[j] * W;
                const int32_t q = Hp[c - 1] + s;
                if (q > bestq) { bestq = q; jq = j; }                   /* lowest j on ties */
                const int32_t eo = Hp[c] + OE, ex = Ep[c] + E;
                const int32_t ein = max2(ex, eo);
                if (ein > beste) { beste = ein; je = j; je_open = (ein == eo); }
                if (is_n && jq_n < 0 && q == h_forced) jq_n = j;
                if (is_n && je_n < 0 && ein == h_forced) je_n = j;
            }
            const int32_t f_open = Hr[c - 1] + OE;
            f = max2(f + E, f_open);
            const int32_t h = is_n ? h_forced : max2(bestq, max2(beste, f));
            Hr[c] = h; Er[c] = beste;
            const size_t o = (size_t)r * (size_t)n + (size_t)(c - 1);
            if (source) {                                               /* c-t-simd.c:485-506 */
                int8_t code;
                if (is_n)            code = jq_n >= 0 ? (int8_t)((seq[c - 1] == base ? 42 : 84) + jq_n) : (je_n >= 0 ? (int8_t)je_n : 126);
                else if (bestq == h) code = (int8_t)((seq[c - 1] == base ? 42 : 84) + jq);
                else if (beste == h) code = (int8_t)(0 + je);
                else                 code = 126;
                source[o] = code;
            }
            if (esource) {                                              /* c-t-simd.c:530-538 */
                int32_t v = (je_open ? 43 : 1) + je;
                if (beste + E == h + OE) v = -v;
                esource[o] = (int8_t)v;
            }
            if (fsource) {                                              /* c-t-simd.c:521-522,564-565 */
                int32_t v = (f == f_open) ? 2 : 1;
                if (f + E == h + OE) v = -v;
                fsource[o] = (int8_t)v;
            }
        }
        if (lastscore) lastscore[r] = Hr[n];
    }
    int32_t best = -1;                                                  /* c-t-simd.c:737-750 */
    for (int32_t r = 0; r < V; r++)
        if (g->is_sink[r] && (best < 0 || H[(size_t)r * W + n] > H[(size_t)best * W + n])) best = r;
    free(H); free(Ea);
    return best;
}

/*
 * The same fill as the reference's AVX512 build performs it on sequences with a literal 'N', i.e. with its blocking:
 * a block of L = 64 x W columns is 64 lane segments of W columns (NUM2, c-t-simd.c:14: column x of a block sits in
 * lane x / W).  Per node and block the reference runs
 *   pass 1 (c-t-simd.c:357-404): every segment on its own, WITHOUT the 'N' rule, lanes >= 1 with no F carry-in; what a
 *          segment hands to the next lane is its last H (as offsets v_j to the predecessors) and its F carry-out;
 *   the lazy-F loop (:408-432): Fout(b) = max(F1out(b), Fout(b-1) + W*E) for b = 1..62;
 *   pass 2 (:447-565): every segment again, WITH the rule (max := 0 in an 'N' column, :461), starting from
 *          v = max(Fout(b-1) - E, v_end(b-1)) and F = Fout(b-1) (:433-446) — values of PASS 1.
 * Pass 2 stores per-column DELTAS (sorce) and adds them up (simple_sorce, :572-575), so a segment whose pass-1 hand-over
 * differs from what pass 2 computed for the same cell (only possible after an 'N' column) shifts the row's absolute
 * scores from there to the end of the block by D = H_pass2(end) - H_pass1(end): the candidates of pass 2 live in the
 * frame of the pass-1 value ("shadow" frame Hs = H - D).  At the next block the offsets are rebuilt from the absolute
 * sums (:196-261: D = 0 again), while the F carry (row->f0, :567) keeps its shadow-frame value.
 * Restated in plain int32 (no saturation, no cutoff): equal to oracle_poa_fill on 'N'-free input for every W.
 */
static int32_t g_blocked_max_shift = 0;
/* largest |D| (shift of a row's scores between the reference's two passes) any call has seen since the last reset: the
 * reference's int8 offset trackers hold about +-60 before their cutoff (c-t-simd.c:370-404) */
int32_t oracle_poa_blocked_max_shift(int reset) { const int32_t v = g_blocked_max_shift; if (reset) g_blocked_max_shift = 0; return v; }

int32_t oracle_poa_fill_blocked(const oracle_dag* g, const uint8_t* seq, int32_t n, const oracle_params* p, int32_t Wv,
                                int8_t* source, int8_t* esource, int8_t* fsource, int32_t* lastscore)
{
    const int32_t V = g->n_nodes, M = p->M, X = p->X, E = p->E, OE = p->O + p->E;
    const int32_t B = 64, L = B * Wv;
    const size_t W = (size_t)n + 1;
    int32_t* H  = (int32_t*)malloc((size_t)(V + 1) * W * sizeof(int32_t));   /* row V = virtual `init` */
    int32_t* Eo = (int32_t*)malloc((size_t)(V + 1) * W * sizeof(int32_t));   /* E a successor sees through this node */
    int32_t* Hi = H + (size_t)V * W; int32_t* Ei = Eo + (size_t)V * W;
    Hi[0] = 0; Ei[0] = NEG_INF;
    for (int32_t c = 1; c <= n; c++) { Hi[c] = p->O + c * E; Ei[c] = Hi[c] + OE; }
    int32_t H1end[64], F1out[64], Fout[64];

    for (int32_t r = 0; r < V; r++) {
        int32_t np = g->pred_off[r + 1] - g->pred_off[r];
        const int32_t* pl = g->pred + g->pred_off[r];
        int32_t init_idx = V;
        if (np == 0) { np = 1; pl = &init_idx; }
        int32_t* Hr = H + (size_t)r * W; int32_t* Er = Eo + (size_t)r * W;
        int32_t h0 = NEG_INF;
        for (int32_t j = 0; j < np; j++) h0 = max2(h0, H[(size_t)pl[j] * W]);
        Hr[0] = (pl == &init_idx) ? OE : h0 + E;
        Er[0] = NEG_INF;
        const uint8_t base = g->base[r];
        int32_t fcarry = NEG_INF;                           /* F of the block's first column, shadow frame of the previous block's end */
        for (int32_t k0 = 0; k0 < n; k0 += L) {             /* block: columns c = k0+1 .. min(n, k0+L) */
            if (k0 == 0) fcarry = Hr[0] + OE;               /* :264-271: f0 = v0 + E + O */
            const int32_t nseg = B;
            /* ---- pass 1 + lazy F */
            for (int32_t b = 0; b < nseg; b++) {
                const int32_t c_lo = k0 + b * Wv + 1, c_hi = c_lo + Wv - 1;     /* segment columns (may lie beyond n: padding) */
                int32_t hprev = (b == 0) ? Hr[k0] : NEG_INF, f = (b == 0) ? fcarry : NEG_INF;
                int have_prev = (b == 0);
                for (int32_t c = c_lo; c <= c_hi && c <= n; c++) {
                    const int32_t sc = (seq[c - 1] == base) ? M : X;
                    int32_t a = NEG_INF;
                    for (int32_t j = 0; j < np; j++) {
                        const int32_t* Hp = H + (size_t)pl[j] * W; const int32_t* Ep = Eo + (size_t)pl[j] * W;
                        a = max2(a, max2(Hp[c - 1] + sc, Ep[c]));
                    }
                    const int32_t fc = have_prev ? f : NEG_INF;
                    const int32_t h = max2(a, fc);
                    f = max2(fc + E, h + OE);
                    if (fc <= NEG_INF / 2) f = h + OE;
                    hprev = h; have_prev = 1;
                }
                H1end[b] = hprev; F1out[b] = have_prev ? f : NEG_INF;
                if (c_lo > n) { H1end[b] = NEG_INF; F1out[b] = NEG_INF; }
            }
            Fout[0] = F1out[0];
            for (int32_t b = 1; b < nseg; b++) Fout[b] = max2(F1out[b], Fout[b - 1] > NEG_INF / 2 ? Fout[b - 1] + Wv * E : NEG_INF);
            /* ---- pass 2 */
            int32_t hs = Hr[k0], fs = fcarry, D = 0;
            for (int32_t b = 0; b < nseg; b++) {
                const int32_t c_lo = k0 + b * Wv + 1, c_hi = c_lo + Wv - 1;
                if (c_lo > n) break;
                if (b >= 1) {
                    const int32_t hnew = max2(H1end[b - 1], Fout[b - 1] - E);      /* :437-440 */
                    D = Hr[c_lo - 1] - hnew;
                    if ((D < 0 ? -D : D) > g_blocked_max_shift) g_blocked_max_shift = D < 0 ? -D : D;
                    hs = hnew; fs = Fout[b - 1];
                }
                for (int32_t c = c_lo; c <= c_hi && c <= n; c++) {
                    const int32_t sc = (seq[c - 1] == base) ? M : X;
                    const int is_n = (seq[c - 1] == 'N');
                    int32_t bestq = NEG_INF, jq = 0, beste = NEG_INF, je = 0, je_open = 0, jq_n = -1, je_n = -1;
                    for (int32_t j = 0; j < np; j++) {
                        const int32_t* Hp = H + (size_t)pl[j] * W; const int32_t* Ep = Eo + (size_t)pl[j] * W;
                        const int32_t q = Hp[c - 1] + sc;
                        if (q > bestq) { bestq = q; jq = j; }
                        const int32_t ein = Ep[c];
                        if (ein > beste) { beste = ein; je = j; je_open = (ein == Hp[c] + OE); }
                        if (is_n && jq_n < 0 && q == hs) jq_n = j;
                        if (is_n && je_n < 0 && ein == hs) je_n = j;
                    }
                    const int32_t f_open = hs + OE;
                    const int32_t f = fs;
                    const int32_t h = is_n ? hs : max2(bestq, max2(beste, f));
                    const size_t o = (size_t)r * (size_t)n + (size_t)(c - 1);
                    if (source) {
                        int8_t code;
                        if (is_n)            code = jq_n >= 0 ? (int8_t)((seq[c - 1] == base ? 42 : 84) + jq_n) : (je_n >= 0 ? (int8_t)je_n : 126);
                        else if (bestq == h) code = (int8_t)((seq[c - 1] == base ? 42 : 84) + jq);
                        else if (beste == h) code = (int8_t)(0 + je);
                        else                 code = 126;
                        source[o] = code;
                    }
                    if (esource) {
                        int32_t v = (je_open ? 43 : 1) + je;
                        if (beste + E == h + OE) v = -v;
                        esource[o] = (int8_t)v;
                    }
                    if (fsource) {
                        int32_t v = (f == f_open) ? 2 : 1;
                        if (f + E == h + OE) v = -v;
                        fsource[o] = (int8_t)v;
                    }
                    Hr[c] = h + D;
                    Er[c] = max2(h + OE, beste + E) + D;
                    fs = max2(f + E, h + OE);
                    hs = h;
                }
            }
            fcarry = fs;
        }
        if (lastscore) lastscore[r] = Hr[n];
    }
    int32_t best = -1;
    for (int32_t r = 0; r < V; r++)
        if (g->is_sink[r] && (best < 0 || H[(size_t)r * W + n] > H[(size_t)best * W + n])) best = r;
    free(H); free(Eo);
    return best;
}
