/* poa_graph.c — see poa_graph.h.  Host-side POA graph for the TSTA_msa drop-in.
 *
 * The reference keeps the graph as a web of malloc'd `struct poa` nodes (msa/poa.h:100-126) and frees/relinks them
 * while walking the trace; here nodes live in one pool and are addressed by index, chain nodes that fuse into an
 * existing node are simply abandoned, and the reference's in-place overwrite of the next trace cell
 * (c-t-simd.c:767-769, 776-782) is a `forced` variable.  Observable behaviour (graph after every step, output
 * file bytes) is what is reproduced; it is checked against the compiled reference in tests/test_msa_host.py.
 *
 * Inputs on which the reference itself has undefined behaviour (use of a freed chain node after a consecutive
 * match at the last character, c-t-simd.c:788-795 then :905-917; sort[-1] at :790) are handled in the evident
 * intent (the fused graph node stands in for the freed chain node).
 */
#include "poa_graph.h"

#include <limits.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ pool */

static int node_new(pg_graph* g)
{
    if (g->n_nodes == g->cap_nodes) {
        g->cap_nodes = g->cap_nodes ? g->cap_nodes * 2 : 1024;
        g->nodes = (pg_node*)realloc(g->nodes, (size_t)g->cap_nodes * sizeof(pg_node));
    }
    pg_node* nd = &g->nodes[g->n_nodes];
    memset(nd, 0, sizeof(*nd));
    nd->pre = (int*)malloc(sizeof(int));  nd->ew = (int*)malloc(sizeof(int));  nd->pre_cap = 1;
    nd->next = (int*)malloc(sizeof(int)); nd->next_cap = 1;
    nd->pre[0] = PG_INIT; nd->ew[0] = 0;
    nd->sub = -1;
    return g->n_nodes++;
}

static void add_pre(pg_graph* g, int id, int pred, int weight)          /* in++, pre[in-1] = pred, edge_weight = weight */
{
    pg_node* nd = &g->nodes[id];
    if (nd->in + 1 > nd->pre_cap) {
        nd->pre_cap = (nd->in + 1) * 2;
        nd->pre = (int*)realloc(nd->pre, (size_t)nd->pre_cap * sizeof(int));
        nd->ew = (int*)realloc(nd->ew, (size_t)nd->pre_cap * sizeof(int));
    }
    nd->pre[nd->in] = pred; nd->ew[nd->in] = weight; nd->in++;
}

static void add_next(pg_graph* g, int id, int succ)                      /* out++, next[out-1] = succ */
{
    pg_node* nd = &g->nodes[id];
    if (nd->out + 1 > nd->next_cap) {
        nd->next_cap = (nd->out + 1) * 2;
        nd->next = (int*)realloc(nd->next, (size_t)nd->next_cap * sizeof(int));
    }
    nd->next[nd->out++] = succ;
}

static inline int sub_of(const pg_graph* g, int id) { return id == PG_INIT ? -1 : g->nodes[id].sub; }
static inline int cs_score_of(const pg_graph* g, int id) { return id == PG_INIT ? 0 : g->nodes[id].cs_score; }
static inline int cs_len_of(const pg_graph* g, int id) { return id == PG_INIT ? 0 : g->nodes[id].cs_len; }

/* ------------------------------------------------------------------- poa_build_init (c-t-simd.c:72-163) */

pg_graph* pg_build_init(const char* s0, int sum)
{
    pg_graph* g = (pg_graph*)calloc(1, sizeof(pg_graph));
    const int len = (int)strlen(s0);
    g->sum = sum;
    g->len = len;
    g->sort = (int*)malloc((size_t)(len > 0 ? len : 1) * sizeof(int));
    g->unsort = (int*)malloc((size_t)(len > 0 ? len : 1) * sizeof(int));
    g->last_node_num = 1;
    for (int i = 0; i < len; i++) {
        const int id = node_new(g);
        pg_node* nd = &g->nodes[id];
        nd->base = (unsigned char)s0[i];
        nd->pass = (unsigned char*)calloc((size_t)sum, 1);
        nd->pass[0] = 1;
        nd->sub = i;
        if (i == 0) { nd->in = 0; nd->pre[0] = PG_INIT; nd->ew[0] = 0; nd->cs_len = 1; }      /* head: edge_weight[0] = 0 */
        else        { nd->in = 1; nd->pre[0] = id - 1; nd->ew[0] = 1; }
        if (i + 1 < len) { nd->out = 1; nd->next[0] = id + 1; } else nd->out = 0;
        g->unsort[i] = g->sort[i] = id;
    }
    return g;
}

/* ------------------------------------------------------------------------ node_fuse (c-t-simd.c:659-977) */

static void become_member(pg_graph* g, int id, int num, int slot, int set_sub)
{
    pg_node* nd = &g->nodes[id];
    nd->pass = (unsigned char*)calloc((size_t)g->sum, 1);
    nd->pass[num] = 1;
    nd->ew[0] = 1;                                        /* edge_weight = malloc(int); [0] = 1 */
    g->unsort[slot] = id;
    if (set_sub) nd->sub = slot;
}

static int iabs(int v) { return v < 0 ? -v : v; }

/* The trace walk of node_fuse, separated from the graph surgery: visits cells from (best sink, last character)
 * following the trace codes and the forced-continuation rules, and records one pg_step per visited cell.
 * This host version reads full [V][n] trace arrays (used with a fill that returns them, e.g. the test oracle);
 * the GPU runs the same walk next to the arrays (poa_walk_kernel) so that they never leave HBM. */
static pg_step* walk_arrays(const pg_graph* g, int len_b, const int32_t* pred_off, const int32_t* pred,
                            const int8_t* source, const int8_t* esource, const int8_t* fsource,
                            const int32_t* lastscore, int64_t* n_steps, int32_t* end_col)
{
    const size_t W = (size_t)len_b;
#define SRC(row, col) source[(size_t)(row) * W + (size_t)(col)]
#define ESRC(row, col) esource[(size_t)(row) * W + (size_t)(col)]
#define FSRC(row, col) fsource[(size_t)(row) * W + (size_t)(col)]
    pg_step* steps = (pg_step*)malloc(((size_t)g->len + (size_t)len_b + 2) * sizeof(pg_step));
    int64_t ns = 0;
    int num1 = g->len - 1, num2 = len_b - 1, forced = -1;
    /* start at the best sink: max lastsorce, the lowest sort index wins ties, index 0 is never looked at
     * (c-t-simd.c:737-750) */
    {
        int s1 = INT_MIN, seen = 0;
        for (int i = g->len - 1; i > 0; i--) {
            const pg_node* nd = &g->nodes[g->sort[i]];
            if (nd->out == 0) {
                if (s1 <= lastscore[i]) { s1 = lastscore[i]; num1 = nd->sub; }
                seen++;
            }
            if (seen >= g->last_node_num) break;
        }
    }
    while (num1 != -1 && num2 != -1) {
        int code;
        if (forced >= 0) { code = forced; forced = -1; } else code = SRC(num1, num2);
        const int kind = code / 42, pidx = code % 42;
        const int np = pred_off[num1 + 1] - pred_off[num1];
        const int Psub = (pidx < np) ? pred[pred_off[num1] + pidx] : -1;
        pg_step st; st.node = num1; st.col = num2; st.code = (int8_t)code; st.peek = 0; st.pad = 0;
        if (kind == 3) {                                   /* F (:753-772) */
            const int fs = FSRC(num1, num2);
            if (num2 - 1 > 0 && (iabs(fs) == 1 || (iabs(fs) == 2 && FSRC(num1, num2 - 1) < 0))) forced = 126;
            steps[ns++] = st;
            num2--;
        } else if (kind == 0) {                            /* E (:773-785) */
            const int es = ESRC(num1, num2);
            if (Psub > 0 && ((es <= 42 && es >= -42) || ((es > 42 || es < -42) && ESRC(Psub, num2) < 0))) {
                int s5 = ESRC(Psub, num2) % 42;
                forced = iabs(s5) - 1;
            }
            steps[ns++] = st;
            num1 = Psub;
        } else {                                           /* match / mismatch (:786-860, :861-956) */
            st.peek = (int8_t)(Psub >= 0 && num2 - 1 >= 0 && SRC(Psub, num2 - 1) / 42 == 1);
            steps[ns++] = st;
            num1 = Psub;
            num2--;
        }
    }
#undef SRC
#undef ESRC
#undef FSRC
    *n_steps = ns;
    *end_col = num2;
    return steps;
}

/* The graph surgery of node_fuse (c-t-simd.c:659-977) driven by the recorded walk. */
/* returns 0, or PG_ERR_RING_FULL when a column would need a sixth distinct character (the reference's mismatch_node[4],
 * poa.h:109, overflows there: undefined behaviour; here it is an error the caller reports) */
static int apply_path(pg_graph* g, const char* b, int num, const pg_step* steps, int64_t n_steps, int end_col)
{
    const int len_b = (int)strlen(b);
    g->unsort = (int*)realloc(g->unsort, (size_t)(len_b + g->len) * sizeof(int));

    /* fresh linear chain for the sequence (c-t-simd.c:685-726) */
    int* seq = (int*)malloc((size_t)len_b * sizeof(int));
    for (int i = 0; i < len_b; i++) {
        const int id = node_new(g);
        seq[i] = id;
        pg_node* nd = &g->nodes[id];
        nd->base = (unsigned char)b[i];
        if (i == 0) { nd->in = 0; nd->pre[0] = PG_INIT; nd->cs_len = 1; }
        else        { nd->in = 1; nd->pre[0] = seq[i - 1]; g->nodes[seq[i - 1]].next[0] = id; }
        nd->out = (i + 1 < len_b) ? 1 : 0;
    }

    int l = 0, cont = 0, ring_full = 0;
    for (int64_t si = 0; si < n_steps; si++) {
        const int num1 = steps[si].node, num2 = steps[si].col;
        const int code = steps[si].code, peek = steps[si].peek;
        const int R = g->sort[num1];
        const int kind = code / 42, pidx = code % 42;
        const int P = g->nodes[R].pre[pidx];
        const int Psub = sub_of(g, P);

        if (kind == 3) {                                   /* F: the character becomes a new node (:753-772) */
            cont = 0;
            become_member(g, seq[num2], num, g->len + l, 1);
            l++;
            continue;
        }
        if (kind == 0) { cont = 3; continue; }             /* E: graph node R is skipped (:773-785) */
        if (kind == 1) {                                   /* match: the character fuses into R (:786-860) */
            if (num2 == len_b - 1) {
                if (peek) g->nodes[R].ew[pidx]++;
                else if (num2 - 1 >= 0) {
                    add_pre(g, R, seq[num2 - 1], 1);
                    pg_node* pv = &g->nodes[seq[num2 - 1]];
                    pv->next[pv->out - 1] = R;
                }
                seq[num2] = R;
            } else if (num2 == 0) {
                if (cont == 1 || cont == 5) seq[0] = R;
                else {
                    add_next(g, R, seq[1]);
                    pg_node* nx = &g->nodes[seq[1]];
                    nx->pre[nx->in - 1] = R;
                }
            } else {
                if (Psub != -1 && peek) g->nodes[R].ew[pidx]++;
                else {
                    add_pre(g, R, seq[num2 - 1], 1);
                    pg_node* pv = &g->nodes[seq[num2 - 1]];
                    pv->next[pv->out - 1] = R;
                }
                if (!(cont == 1 || cont == 5)) {
                    add_next(g, R, seq[num2 + 1]);
                    pg_node* nx = &g->nodes[seq[num2 + 1]];
                    nx->pre[nx->in - 1] = R;
                }
                seq[num2] = R;
            }
            cont = 1;
            g->nodes[R].pass[num] = 1;
            continue;
        }
        /* mismatch: the character is aligned to R's column (:861-956) */
        {
            int s4 = 0;
            for (int s = 0; s < g->nodes[R].ring_n; s++) {
                const int Mn = g->nodes[R].ring[s];
                if (g->nodes[seq[num2]].base != g->nodes[Mn].base) continue;
                int reused = 0;
                if (num2 != 0) {
                    if (Psub != -1 && peek) {
                        for (int ss = 0; ss < g->nodes[Mn].in; ss++)
                            if (g->nodes[Mn].pre[ss] == P) { g->nodes[Mn].ew[ss]++; reused = 1; }
                    }
                    if (!reused) {
                        add_pre(g, Mn, seq[num2 - 1], 1);
                        pg_node* pv = &g->nodes[seq[num2 - 1]];
                        pv->next[pv->out - 1] = Mn;
                    }
                }
                s4 = 1;
                if (cont == 1 || cont == 4) {
                    pg_node* nx = &g->nodes[seq[num2 + 1]];
                    for (int ss = 0; ss < nx->in; ss++)
                        if (nx->pre[ss] == Mn) { s4 = 2; nx->ew[ss]++; nx->in--; }       /* drops the placeholder in-edge */
                }
                if (s4 == 1 && num2 != len_b - 1) {
                    pg_node* nx = &g->nodes[seq[num2 + 1]];
                    nx->pre[nx->in - 1] = Mn;
                    add_next(g, Mn, seq[num2 + 1]);
                }
                g->nodes[Mn].pass[num] = 1;
                cont = reused ? 5 : 4;
                seq[num2] = Mn;
            }
            if (s4 == 0) {                                 /* a new member of R's ring (:927-953) */
                cont = 2;
                const int Xn = seq[num2];
                become_member(g, Xn, num, g->len + l, 1);
                l++;
                pg_node* r = &g->nodes[R];
                pg_node* x = &g->nodes[Xn];
                if (r->ring_n >= 4) { ring_full = 1; break; }
                r->ring_n++;
                r->ring[r->ring_n - 1] = Xn;
                x->ring_n = r->ring_n;
                x->ring[x->ring_n - 1] = R;
                for (int s = 0; s < r->ring_n - 1; s++) {
                    pg_node* mn = &g->nodes[r->ring[s]];
                    mn->ring_n++;
                    mn->ring[r->ring_n - 1] = Xn;
                    x->ring[s] = r->ring[s];
                }
            }
        }
    }
    for (int num2 = end_col; num2 > -1; num2--) {          /* leftover prefix: new nodes (:959-972) */
        become_member(g, seq[num2], num, g->len + l, 0);
        l++;
    }
    g->len += l;
    g->unsort = (int*)realloc(g->unsort, (size_t)g->len * sizeof(int));
    g->sort = (int*)realloc(g->sort, (size_t)g->len * sizeof(int));
    free(seq);
    if (ring_full) {
        fprintf(stderr, "TSTA_msa: sequence %d puts a sixth distinct character into one alignment column; the reference's "
                        "mismatch ring holds four aligned nodes besides the column's own (poa.h:109)\n", num + 1);
        return PG_ERR_RING_FULL;
    }
    return 0;
}

/* ------------------------------------------------------------------------------ control (c-t-simd.c:979-1049) */

static void build_csr(const pg_graph* g, int32_t** pred_off, int32_t** pred, uint8_t** base, uint8_t** is_sink)
{
    const int V = g->len;
    *pred_off = (int32_t*)malloc((size_t)(V + 1) * sizeof(int32_t));
    size_t npred = 0;
    for (int r = 0; r < V; r++) npred += (size_t)g->nodes[g->sort[r]].in;
    *pred = (int32_t*)malloc((npred ? npred : 1) * sizeof(int32_t));
    *base = (uint8_t*)malloc((size_t)(V > 0 ? V : 1));
    *is_sink = (uint8_t*)malloc((size_t)(V > 0 ? V : 1));
    (*pred_off)[0] = 0;
    for (int r = 0; r < V; r++) {                          /* CSR of the DAG in p->sort[] order, predecessors in pre[] order */
        const pg_node* nd = &g->nodes[g->sort[r]];
        (*base)[r] = nd->base;
        (*is_sink)[r] = (nd->out == 0);
        int k = (*pred_off)[r];
        for (int j = 0; j < nd->in; j++) (*pred)[k++] = sub_of(g, nd->pre[j]);
        (*pred_off)[r + 1] = k;
    }
}

int pg_align_and_fuse(pg_graph* g, const char* seq, int num, pg_fill_fn fill, void* ctx)
{
    const int V = g->len, n = (int)strlen(seq);
    int32_t *pred_off, *pred; uint8_t *base, *is_sink;
    build_csr(g, &pred_off, &pred, &base, &is_sink);
    const size_t cells = (size_t)V * (size_t)n;
    int8_t* source = (int8_t*)malloc(cells ? cells : 1);
    int8_t* esource = (int8_t*)malloc(cells ? cells : 1);
    int8_t* fsource = (int8_t*)malloc(cells ? cells : 1);
    int32_t* lastscore = (int32_t*)calloc((size_t)(V > 0 ? V : 1), sizeof(int32_t));
    int rc = fill(ctx, V, pred_off, pred, base, is_sink, (const uint8_t*)seq, n, source, esource, fsource, lastscore);
    if (rc == 0) {
        int64_t ns = 0; int32_t end_col = -1;
        pg_step* steps = walk_arrays(g, n, pred_off, pred, source, esource, fsource, lastscore, &ns, &end_col);
        rc = apply_path(g, seq, num, steps, ns, end_col);
        free(steps);
    }
    free(pred_off); free(pred); free(base); free(is_sink);
    free(source); free(esource); free(fsource); free(lastscore);
    return rc;
}

int pg_align_and_fuse_path(pg_graph* g, const char* seq, int num, pg_path_fn path, void* ctx)
{
    const int V = g->len, n = (int)strlen(seq);
    int32_t *pred_off, *pred; uint8_t *base, *is_sink;
    build_csr(g, &pred_off, &pred, &base, &is_sink);
    pg_step* steps = NULL; int64_t ns = 0; int32_t end_col = -1;
    void (*free_steps)(void*) = free;
    int rc = path(ctx, V, pred_off, pred, base, is_sink, (const uint8_t*)seq, n, &steps, &ns, &end_col, &free_steps);
    if (rc == 0) rc = apply_path(g, seq, num, steps, ns, end_col);
    if (steps) free_steps(steps);
    free(pred_off); free(pred); free(base); free(is_sink);
    return rc;
}

/* ---------------------------------------------------------------------------------- t_sort (topo.c) */

typedef struct sort_ctx { pg_graph* g; int final_pass; } sort_ctx;

/* tp(), topo.c:179-203: place p, then first its ring-aligned sinks whose last predecessor was just placed, then
 * depth-first every successor that became ready (sinks whose whole ring is sinks are deferred: passing == 2) */
static int tp(pg_graph* g, int p, int subs)
{
    pg_node* nd = &g->nodes[p];
    g->sort[subs] = p;
    nd->logo = 0;
    nd->sub = subs;
    nd->in_temp = -1;
    subs++;
    for (int j = 0; j < nd->out; j++) {
        pg_node* q = &g->nodes[nd->next[j]];
        if (q->out == 0 && q->passing == 1 && q->in_temp - 1 == 0) {
            q->in_temp--;
            if (q->in_temp == 0) { subs = tp(g, nd->next[j], subs); nd = &g->nodes[p]; }
        }
    }
    for (int j = 0; j < nd->out; j++) {
        pg_node* q = &g->nodes[nd->next[j]];
        q->in_temp--;
        if (q->in_temp == 0 && q->passing != 2) { subs = tp(g, nd->next[j], subs); nd = &g->nodes[p]; }
    }
    return subs;
}

static void mark_sink_rings(pg_graph* g, int with_passing1)               /* topo.c:207-223 / :69-83 */
{
    for (int i = 0; i < g->len; i++) {
        pg_node* u = &g->nodes[g->unsort[i]];
        u->in_temp = u->in;
        u->passing = 0;
        if (u->out == 0 && u->ring_n > 0) {
            int any = 0;
            for (int j = 0; j < u->ring_n; j++)
                if (g->nodes[u->ring[j]].out != 0) { any = 1; if (with_passing1) u->passing = 1; }
            if (!any) u->passing = 2;
        }
    }
}

static void toposort(pg_graph* g)                                         /* topo.c:205-241 */
{
    mark_sink_rings(g, 1);
    int subs = 0;
    while (subs < g->len) {
        int progressed = 0;
        for (int i = 0; i < g->len; i++)
            if (g->nodes[g->unsort[i]].in_temp == 0) {
                subs = tp(g, g->unsort[i], subs);
                if (subs + g->last_node_num == g->len) {
                    for (int k = 0; k < g->len; k++)
                        if (g->nodes[g->unsort[k]].in_temp == 0) subs = tp(g, g->unsort[k], subs);
                }
                progressed = 1;
                break;
            }
        if (!progressed) break;                            /* cannot happen on a DAG; avoids the reference's endless loop */
    }
}

/* tp1(), topo.c:7-66: as tp() but a ring is emitted as a consecutive group only when all members are ready, and
 * the heaviest-edge consensus DP (node_sorce / node_base_len / node_sorce_source) is filled in on the way */
static int tp1(pg_graph* g, int p, int subs)
{
    pg_node* nd = &g->nodes[p];
    g->sort[subs] = p;
    nd->logo = 0;
    nd->sub = subs;
    int max = 0, max_i = 0;
    for (int i = 0; i < nd->in; i++) {
        if (cs_score_of(g, nd->pre[i]) >= 0) {
            if (max < nd->ew[i]) { max = nd->ew[i]; max_i = i; }
            else if (max == nd->ew[i] && cs_score_of(g, nd->pre[max_i]) <= cs_score_of(g, nd->pre[i])) { max = nd->ew[i]; max_i = i; }
        }
    }
    nd->cs_score = cs_score_of(g, nd->pre[max_i]) + max;
    nd->cs_len = cs_len_of(g, nd->pre[max_i]) + 1;
    nd->cs_src = sub_of(g, nd->pre[max_i]);
    nd->in_temp = -1;
    subs++;
    for (int j = 0; j < g->nodes[p].out; j++) {
        const int qi = g->nodes[p].next[j];
        pg_node* q = &g->nodes[qi];
        q->in_temp--;
        if (q->in_temp == 0 && q->ring_n == 0 && q->passing != 2) subs = tp1(g, qi, subs);
        else if (q->in_temp == 0 && q->ring_n > 0 && q->passing != 2) {
            int c = 0;
            for (int s = 0; s < q->ring_n; s++) if (g->nodes[q->ring[s]].in_temp == 0) c++;
            if (c == g->nodes[qi].ring_n) {
                subs = tp1(g, qi, subs);
                for (int ss = 0; ss < g->nodes[qi].ring_n; ss++) {
                    const int m = g->nodes[qi].ring[ss];
                    if (g->nodes[m].in_temp == 0) subs = tp1(g, m, subs);
                }
            }
        }
    }
    return subs;
}

static void toposort1(pg_graph* g)                                        /* topo.c:68-117 */
{
    mark_sink_rings(g, 0);
    int subs = 0;
    while (subs < g->len) {
        int progressed = 0;
        for (int i = 0; i < g->len && !progressed; i++) {
            const int ui = g->unsort[i];
            if (g->nodes[ui].in_temp != 0) continue;
            if (g->nodes[ui].ring_n == 0) { subs = tp1(g, ui, subs); progressed = 1; }
            else {
                int c = 0;
                for (int j = 0; j < g->nodes[ui].ring_n; j++) if (g->nodes[g->nodes[ui].ring[j]].in_temp == 0) c++;
                if (c == g->nodes[ui].ring_n) {
                    subs = tp1(g, ui, subs);
                    for (int ss = 0; ss < g->nodes[ui].ring_n; ss++) {
                        const int m = g->nodes[ui].ring[ss];
                        if (g->nodes[m].in_temp == 0) subs = tp1(g, m, subs);
                    }
                    progressed = 1;
                }
            }
        }
        if (!progressed) break;
    }
}

/* modify(), topo.c:119-177: while the heaviest consensus path does not end in a sink, knock out the lighter
 * alternatives below its end and re-run the DP downstream */
static void modify(pg_graph* g)
{
    for (;;) {
        int max = INT_MIN, max_i = 0;
        for (int i = 0; i < g->len; i++) {
            const pg_node* nd = &g->nodes[g->sort[i]];
            if (max <= nd->cs_score) { max = nd->cs_score; max_i = nd->sub; }
        }
        pg_node* top = &g->nodes[g->sort[max_i]];
        if (top->out == 0) return;
        for (int i = 0; i < top->out; i++) {
            pg_node* nx = &g->nodes[top->next[i]];
            for (int j = 0; j < nx->in; j++) {
                if (nx->pre[j] == PG_INIT) continue;
                pg_node* pj = &g->nodes[nx->pre[j]];
                if (pj->cs_score < top->cs_score && pj->cs_score > 0) pj->cs_score = -pj->cs_score;
            }
            nx->logo = 4;
        }
        int Max = 0, Max_i = 0;
        for (int i = max_i + 1; i < g->len; i++) {
            pg_node* nd = &g->nodes[g->sort[i]];
            if (nd->cs_score >= 0 || nd->logo == 4) {
                for (int j = 0; j < nd->in; j++) {
                    if (cs_score_of(g, nd->pre[j]) >= 0) {
                        if (Max < nd->ew[j]) { Max = nd->ew[j]; Max_i = j; }
                        else if (Max == nd->ew[j] && cs_score_of(g, nd->pre[Max_i]) <= cs_score_of(g, nd->pre[j])) { Max = nd->ew[j]; Max_i = j; }
                    }
                }
                nd->cs_score = cs_score_of(g, nd->pre[Max_i]) + Max;
                nd->cs_len = cs_len_of(g, nd->pre[Max_i]) + 1;
                nd->cs_src = sub_of(g, nd->pre[Max_i]);
                nd->logo = 0;
                Max = Max_i = 0;
            }
        }
    }
}

static void* sort_thread(void* arg)
{
    sort_ctx* c = (sort_ctx*)arg;
    pg_graph* g = c->g;
    g->last_node_num = 0;                                               /* topo.c:246-249 */
    for (int i = 0; i < g->len; i++) if (g->nodes[g->unsort[i]].out == 0) g->last_node_num++;
    if (!c->final_pass) toposort(g);
    else { toposort1(g); modify(g); }
    for (int i = 0; i < g->len; i++) g->unsort[i] = g->sort[i];          /* topo.c:257-258 */
    return NULL;
}

/* The reference's tp()/tp1() recurse once per node along a chain (depth = sequence length); run them on a thread
 * with a stack sized for that instead of the process stack. */
void pg_sort(pg_graph* g, int final_pass)
{
    sort_ctx c = { g, final_pass };
    pthread_attr_t at;
    pthread_t th;
    pthread_attr_init(&at);
    size_t want = (size_t)256 * 1024 * 1024 + (size_t)g->len * 256;
    pthread_attr_setstacksize(&at, want);
    if (pthread_create(&th, &at, sort_thread, &c) == 0) pthread_join(th, NULL);
    else sort_thread(&c);
    pthread_attr_destroy(&at);
}

/* --------------------------------------------------------------------------- printf_result (result.c) */

void pg_write_result(pg_graph* g, int num, FILE* out)
{
    char** r = (char**)malloc((size_t)num * sizeof(char*));
    for (int i = 0; i < num; i++) {
        r[i] = (char*)malloc((size_t)g->len + 1);
        memset(r[i], '-', (size_t)g->len);
    }
    int len = 0;
    for (int i = 0; i < g->len; i++) {                                   /* one output column per ring (result.c:53-68) */
        pg_node* nd = &g->nodes[g->sort[i]];
        if (nd->logo != 0) continue;
        for (int j = 0; j < num; j++) if (nd->pass[j] == 1) r[j][len] = (char)nd->base;
        for (int l = 0; l < nd->ring_n; l++) {
            pg_node* mn = &g->nodes[nd->ring[l]];
            for (int j = 0; j < num; j++)
                if (mn->pass[j] == 1) { r[j][len] = (char)mn->base; mn->logo = 1; }
        }
        len++;
    }
    for (int i = 0; i < num; i++) {
        r[i][len] = '\0';
        fprintf(out, ">%d\n", i + 1);
        fputs(r[i], out);
        fputs("\n", out);
        free(r[i]);
    }
    free(r);
    /* consensus: heaviest path ending in the best sink (result.c:6-40) */
    int s = 0;
    int max = g->nodes[g->sort[g->len - 1]].cs_score, max_i = g->nodes[g->sort[g->len - 1]].sub;
    for (int i = g->len - 2; i >= 0; i--) {
        const pg_node* nd = &g->nodes[g->sort[i]];
        if (nd->out == 0) {
            if (max <= nd->cs_score) { max = nd->cs_score; max_i = nd->sub; }
            s++;
        }
        if (s >= g->last_node_num) break;
    }
    int clen = g->nodes[g->sort[max_i]].cs_len;
    char* c = (char*)malloc((size_t)clen + 1);
    c[clen] = '\0';
    for (int k = clen - 1; k > -1; k--) {
        c[k] = (char)g->nodes[g->sort[max_i]].base;
        max_i = g->nodes[g->sort[max_i]].cs_src;
    }
    fprintf(out, "consensus:\n");
    fputs(c, out);
    free(c);
}

void pg_free(pg_graph* g)
{
    if (!g) return;
    for (int i = 0; i < g->n_nodes; i++) { free(g->nodes[i].pre); free(g->nodes[i].ew); free(g->nodes[i].next); free(g->nodes[i].pass); }
    free(g->nodes); free(g->sort); free(g->unsort); free(g);
}
