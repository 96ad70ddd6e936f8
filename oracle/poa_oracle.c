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
