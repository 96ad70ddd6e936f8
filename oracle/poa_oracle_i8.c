/*
 * poa_oracle_i8.c — the sequence-vs-DAG fill of TSTA_msa restated WITH the reference's number representation:
 * int8 horizontal differences per row (`sorce` / `esorce`), saturating int8 offsets to every predecessor with their two
 * overflow reservoirs, 64 byte lanes per vector (the AVX512 build: B = 64, lane = column / W inside a block of L = 64 W
 * columns), the speculative first pass, the lazy-F fix-up and the second pass (msa/c-t-simd.c:165-599), driven by the
 * anti-diagonal block schedule of control() / block_alignment() (:602-657, :979-1049).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Why it exists beside poa_oracle.c: on 'N'-free input inside the documented
 * domain the int8 encoding is exact and the plain int32 DP of poa_oracle.c is the same function.  On sequences with a
 * literal 'N' the reference's output is an artefact of this very encoding (its first pass ignores the 'N' rule, its
 * second pass applies it, the offsets saturate and are cut off at 60), so the only faithful statement of "what the
 * reference prints" for such input is the encoding itself.  Every vector statement of the reference is written here
 * as what it does to ONE byte lane (the functions below take the lane's values), the cross-lane parts (byte shift, the
 * three scalar loops over lanes) are explicit.
 * Parity: PINNED by tests/golden/msa_n_golden.json (40 inputs x 5 values of -W, outputs of the compiled reference) and by
 * the 'N'-free fixtures (equal to oracle_poa_fill there).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

enum { NB = 64, SMIN = -120, LOGO_LOW = -6, LOGO_CUT = 60, LOGO_TOP = 100, FCLAMP = 125 };

static inline int sat8(int x) { return x > 127 ? 127 : (x < -128 ? -128 : x); }
static inline int adds8(int a, int b) { return sat8(a + b); }                   /* _mm512_adds_epi8 */
static inline int subs8(int a, int b) { return sat8(a - b); }                   /* _mm512_subs_epi8 */
static inline int wrap8(int x) { return (int8_t)x; }                            /* _mm512_add_epi8 / sub_epi8 */
static inline int subus8(int a, int b)                                          /* _mm512_subs_epu8 on the same bytes */
{
    const int ua = (uint8_t)a, ub = (uint8_t)b;
    return (int8_t)(ua > ub ? ua - ub : 0);
}
static inline int max8(int a, int b) { return a > b ? a : b; }

/* one predecessor's offset tracker in one lane: v = offset of this row above the predecessor (saturating), vc1 / vc2 = what
 * did not fit (c-t-simd.c:385-403 and :541-558: the same statements in both passes); `best` = the row's delta of this column,
 * t = the predecessor's */
static inline void tracker_step(int best, int t, int* v, int* vc1, int* vc2)
{
    int d = subs8(best, t);
    const int vc0 = adds8(*v, d);
    if (!(d > 0)) d = 0;
    const int room = subs8(127, *v);
    const int y = subus8(d, room);
    *v = adds8(vc0, *vc1);
    const int diff = (*vc1 == 0) ? 0 : subus8(127, vc0);
    const int old1 = *vc1;
    *vc1 = subus8(*vc1, diff);
    *vc1 = adds8(*vc1, *vc2);
    const int moved = subus8(*vc1, old1);
    *vc2 = subus8(*vc2, diff);
    *vc2 = adds8(*vc2, y);
    *vc2 = subus8(*vc2, moved);
}

/* how an offset that may exceed a byte is split at a block's first column (c-t-simd.c:203-222, :236-256, :318-337) */
static inline void split_offset(int te, int* v, int* c1, int* c2)
{
    if (te > LOGO_TOP) {
        *v = LOGO_TOP;
        if (te - LOGO_TOP > 127) { *c2 = (te - LOGO_TOP - 127) > 127 ? 127 : (te - LOGO_TOP - 127); *c1 = 127; }
        else { *c2 = 0; *c1 = te - LOGO_TOP; }
    } else { *v = wrap8(te); *c2 = 0; *c1 = 0; }
}

typedef struct {
    int32_t V, n, W, L, nblk, maxp;
    int32_t length1, length2, maxtag, fmaxtag, lmaxtag;
    const oracle_dag* g; const oracle_params* p;
    uint8_t* seq;                 /* padded with 'N', striped per block (readseq, c-t-simd.c:37-58) */
    int8_t *sorce, *esorce;       /* [(V+1)][length1], striped; row V = the virtual `init` (one block only) */
    int8_t *source, *esource, *fsource;   /* [V][length1], striped */
    int32_t* simple;              /* [(V+1)][nblk + 1] */
    int32_t* first_col;           /* [(V+1)] */
    int8_t* f0;                   /* [V][maxp] */
} st8;

/* per-task scratch of block_alignment (c-t-simd.c:616-637): lives across the rows of one task */
typedef struct { int8_t *f_temp, *VC1, *VC2, *r_s; int8_t *v0, *vc_1, *vc_2; int* te; } task8;

static void line_i8(st8* S, task8* T, int block_i, int block_j, int block_l, int r, int nv)
{
    const oracle_params* p = S->p;
    const int M = p->M, X = p->X, E = p->E, O = p->O, OE = wrap8(p->O + p->E);
    const int W = S->W, L = S->L, V = S->V;
    int np = S->g->pred_off[r + 1] - S->g->pred_off[r];
    const int32_t* pl = S->g->pred + S->g->pred_off[r];
    int32_t init_idx = V;
    const int is_source = (np == 0);
    int8_t* init_s = S->sorce + (size_t)V * S->length1; int8_t* init_e = S->esorce + (size_t)V * S->length1;
    int32_t* simple = S->simple + (size_t)r * (S->nblk + 1);
    if (is_source) {                                                   /* :174-188 */
        if (block_i == 0) { init_s[0] = (int8_t)(O + E); init_e[0] = (int8_t)(2 * (O + E)); }
        else { init_s[0] = (int8_t)E; init_e[0] = (int8_t)(E + O + E); }
        np = 1; pl = &init_idx;
        S->first_col[r] = simple[0] = E + O;
    }
    /* where this block's L bytes of a row start (the reference's pd / pc1 juggling is memory management: a row that is only
     * read inside its own block of rows keeps one block of bytes) */
    const size_t boff = (size_t)nv * L;
    const int8_t* ps[64]; const int8_t* pe[64];
    for (int j = 0; j < np; j++) {
        const size_t o = (size_t)pl[j] * S->length1 + (pl[j] == V ? 0 : boff);
        ps[j] = S->sorce + o; pe[j] = S->esorce + o;
    }
    int8_t* my_s = S->sorce + (size_t)r * S->length1 + boff; int8_t* my_e = S->esorce + (size_t)r * S->length1 + boff;
    int8_t* o_src = S->source + (size_t)r * S->length1 + boff; int8_t* o_es = S->esource + (size_t)r * S->length1 + boff;
    int8_t* o_fs = S->fsource + (size_t)r * S->length1 + boff;
    const uint8_t* sq = S->seq + boff;
    int8_t* f0 = S->f0 + (size_t)r * S->maxp;
    const int first_block_task = (block_i <= S->lmaxtag && block_l == block_j - 1);

    if (first_block_task && !is_source) {                              /* :194-224 */
        int fc = S->first_col[pl[0]] + E;
        for (int j = 1; j < np; j++) if (fc < S->first_col[pl[j]] + E) fc = S->first_col[pl[j]] + E;
        S->first_col[r] = fc; simple[0] = fc;
        for (int j = 0; j < np; j++) {
            T->te[j] = fc - S->first_col[pl[j]];
            int v, c1, c2; split_offset(T->te[j], &v, &c1, &c2);
            T->v0[j] = (int8_t)v; T->vc_1[j] = T->VC1[j * NB] = (int8_t)c1; T->vc_2[j] = T->VC2[j * NB] = (int8_t)c2;
        }
    } else if (pl[0] == V) {                                           /* :227-232 */
        T->v0[0] = (int8_t)(simple[nv] - (nv * L * E + (nv > 0 ? O : 0)));
        T->vc_2[0] = T->VC2[0] = 0; T->vc_1[0] = T->VC1[0] = 0;
    } else {                                                           /* :233-261 */
        for (int j = 0; j < np; j++) {
            T->te[j] = simple[nv] - S->simple[(size_t)pl[j] * (S->nblk + 1) + nv];
            int v, c1, c2; split_offset(T->te[j], &v, &c1, &c2);
            T->v0[j] = (int8_t)v; T->vc_1[j] = T->VC1[j * NB] = (int8_t)c1; T->vc_2[j] = T->VC2[j * NB] = (int8_t)c2;
        }
    }
    if (first_block_task && block_i < S->length2 / L)                  /* :264-271 */
        for (int j = 0; j < (is_source ? 1 : np); j++) f0[j] = (int8_t)(T->v0[j] + E + O);

    /* what the predecessor's deltas add up to over each lane's W columns (:279-288) */
    for (int j = 0; j < np; j++)
        for (int b = 0; b < NB; b++) {
            int s = 0;
            for (int i = 0; i < W; i++) s = wrap8(s + ps[j][i * NB + b]);
            T->r_s[j * NB + b] = (int8_t)s;
        }
    /* the offsets every lane STARTS its first pass with: the first column's, carried on under the guess that this row falls by
     * E per column, renormalised so that the smallest is -6 (:290-346) */
    int v[64][NB], vc1[64][NB], vc2[64][NB], f[64][NB];
    if (np != 1) {
        for (int j = 0; j < np; j++) T->f_temp[j * NB] = T->v0[j];
        for (int b = 1; b < NB; b++) {
            for (int j = 0; j < np; j++) T->te[j] = T->te[j] - T->r_s[j * NB + b - 1] + W * E;
            int m1 = T->te[0];
            for (int j = 1; j < np; j++) if (T->te[j] < m1) m1 = T->te[j];
            const int m2 = LOGO_LOW - m1;
            for (int j = 0; j < np; j++) {
                int vv, c1, c2; split_offset(T->te[j] + m2, &vv, &c1, &c2);
                T->f_temp[j * NB + b] = (int8_t)vv; T->VC1[j * NB + b] = (int8_t)c1; T->VC2[j * NB + b] = (int8_t)c2;
            }
        }
        for (int j = 0; j < np; j++) for (int b = 0; b < NB; b++) v[j][b] = T->f_temp[j * NB + b];
    } else {
        for (int b = 0; b < NB; b++) { T->VC2[b] = 0; T->VC1[b] = 0; v[0][b] = wrap8(E); }
        T->vc_1[0] = T->vc_2[0] = 0;
        v[0][0] = T->v0[0];
    }
    for (int j = 0; j < np; j++)
        for (int b = 0; b < NB; b++) { vc2[j][b] = T->VC2[j * NB + b]; vc1[j][b] = T->VC1[j * NB + b]; f[j][b] = (b == 0) ? f0[j] : SMIN; }

    /* ---- first pass (:357-404): no 'N' rule, every lane from its guessed start */
    int8_t hg[32 * NB];
    for (int i = 0; i < W; i++)
        for (int b = 0; b < NB; b++) {
            const int h = (sq[i * NB + b] == S->g->base[r]) ? M : X;
            hg[i * NB + b] = (int8_t)h;
            int s1 = SMIN;
            for (int j = 0; j < np; j++) {
                int t = max8(f[j][b], h);
                t = max8(pe[j][i * NB + b], t);
                t = subs8(t, v[j][b]);
                if (v[j][b] > LOGO_CUT) t = OE;
                s1 = max8(s1, t);
            }
            for (int j = 0; j < np; j++) {
                const int t = ps[j][i * NB + b];
                const int a = subs8(f[j][b], wrap8(t - E));
                int c = adds8(s1, OE);
                c = subs8(c, t);
                c = adds8(v[j][b], c);
                f[j][b] = max8(a, c);
                tracker_step(s1, t, &v[j][b], &vc1[j][b], &vc2[j][b]);
            }
        }
    /* ---- lazy F across the lanes, the hand-over of every lane's end state to its right neighbour (:406-446) */
    for (int j = 0; j < np; j++) {
        int ft[NB];
        for (int b = 0; b < NB; b++) ft[b] = f[j][b];
        int te = ft[0];
        for (int x = 1; x < NB - 1; x++) {
            te = te - T->r_s[j * NB + x] + W * E;
            if (te > ft[x] && te > FCLAMP) ft[x] = FCLAMP;
            else if (te > ft[x]) ft[x] = te;
            else if (ft[x] > FCLAMP) { te = ft[x]; ft[x] = FCLAMP; }
            else te = ft[x];
        }
        T->te[j] = te;
        int nf[NB], nv0[NB], n1[NB], n2[NB];
        for (int b = 0; b < NB; b++) {
            const int fe = subs8(ft[b], wrap8(E));
            nf[b] = ft[b]; nv0[b] = max8(fe, v[j][b]); n1[b] = vc1[j][b]; n2[b] = vc2[j][b];
        }
        for (int b = NB - 1; b >= 1; b--) {                              /* the byte shift */
            f[j][b] = nf[b - 1]; vc1[j][b] = n1[b - 1]; vc2[j][b] = n2[b - 1];
            v[j][b] = adds8(nv0[b - 1], n1[b - 1]);
        }
        f[j][0] = f0[j]; vc1[j][0] = T->vc_1[j]; vc2[j][0] = T->vc_2[j];
        v[j][0] = adds8(T->v0[j], T->vc_1[j]);
    }
    /* ---- second pass (:447-565): the 'N' rule, the trace codes, the deltas */
    int sum[NB];
    for (int b = 0; b < NB; b++) sum[b] = 0;
    for (int i = 0; i < W; i++)
        for (int b = 0; b < NB; b++) {
            const int isn = (sq[i * NB + b] == 'N');
            const int h = hg[i * NB + b];
            const int sm = (h == M), sx = (h == X);
            int best = SMIN, eumax = SMIN;
            int fv[64], eu[64], q[64], ev[64], tt[64], ee[64];
            for (int j = 0; j < np; j++) {
                tt[j] = ps[j][i * NB + b]; ee[j] = pe[j][i * NB + b];
                fv[j] = subs8(f[j][b], v[j][b]);
                eu[j] = subs8(ee[j], v[j][b]);
                q[j] = subs8(h, v[j][b]);
                int t = max8(max8(fv[j], eu[j]), q[j]);
                if (v[j][b] > LOGO_CUT) t = OE;
                best = max8(best, t);
                ev[j] = subs8(ee[j], tt[j]);
                eumax = max8(eumax, eu[j]);
            }
            if (isn) best = 0;
            sum[b] = wrap8(sum[b] + best);
            int code = 126, num = 0;
            for (int j = np - 1; j >= 0; j--) if (best == eu[j]) { code = 0; num = j; }
            for (int j = np - 1; j >= 0; j--) if (best == q[j] && sx) { code = 84; num = j; }
            for (int j = np - 1; j >= 0; j--) if (best == q[j] && sm) { code = 42; num = j; }
            o_src[i * NB + b] = (int8_t)wrap8(code + num);
            my_s[i * NB + b] = (int8_t)best;
            int es = 1, fs = 1, es_num = 0, m1 = 0;
            const int open = adds8(best, OE);
            int emax = SMIN;
            for (int j = np - 1; j >= 0; j--) {
                f[j][b] = adds8(f[j][b], wrap8(E));
                const int s1 = adds8(open, v[j][b]);
                m1 = (f[j][b] == s1);
                f[j][b] = max8(f[j][b], s1);
                f[j][b] = subs8(f[j][b], tt[j]);
                if (fv[j] == OE) fs = 2;
                int e2 = adds8(ee[j], wrap8(E));
                e2 = subs8(e2, v[j][b]);
                const int m2 = (open == e2);
                emax = max8(emax, max8(open, e2));
                const int m3 = (eu[j] == eumax);
                if (m3) es_num = j;
                const int cand = wrap8(((ev[j] == OE) ? 43 : 1) + es_num);
                if (m3) es = cand;
                if (m3 && m2) es = wrap8(-es);
                tracker_step(best, tt[j], &v[j][b], &vc1[j][b], &vc2[j][b]);
            }
            if (m1) fs = wrap8(-fs);
            o_fs[i * NB + b] = (int8_t)fs; o_es[i * NB + b] = (int8_t)es; my_e[i * NB + b] = (int8_t)emax;
        }
    for (int j = 0; j < np; j++) f0[j] = (int8_t)f[j][NB - 1];          /* :566-567 */
    int red = 0;
    for (int b = 0; b < NB; b++) red += sum[b];
    simple[nv + 1] = simple[nv] + (int16_t)red;                          /* :568-572 */
}

int32_t oracle_poa_fill_i8(const oracle_dag* g, const uint8_t* seq, int32_t n, const oracle_params* p, int32_t Wv,
                           int8_t* source, int8_t* esource, int8_t* fsource, int32_t* lastscore)
{
    st8 S; memset(&S, 0, sizeof S);
    const int32_t V = g->n_nodes, L = NB * Wv;
    if (Wv < 1 || Wv > 32 || V < 1 || n < 1) return -1;
    S.V = V; S.n = n; S.W = Wv; S.L = L; S.g = g; S.p = p;
    S.length1 = (n % L) ? n + (L - n % L) : n;
    S.length2 = (V % L) ? V + (L - V % L) : V;
    S.nblk = S.length1 / L;
    S.maxtag = S.length1 / L - 1;
    if (S.length1 >= S.length2) { S.fmaxtag = S.length2 / L - 1; S.lmaxtag = S.length1 / L - 1; }
    else { S.fmaxtag = S.length1 / L - 1; S.lmaxtag = S.length2 / L - 1; }
    S.maxp = 1;
    for (int32_t r = 0; r < V; r++) { const int32_t k = g->pred_off[r + 1] - g->pred_off[r]; if (k > S.maxp) S.maxp = k; }
    if (S.maxp > 64) return -1;
    const size_t rowb = (size_t)S.length1;
    S.seq = (uint8_t*)malloc(rowb);
    for (int32_t i = 0; i < S.length1; i++) {                           /* readseq: pad with 'N', stripe */
        const int32_t x = i % L, col = i / L * L + (x % NB) * Wv + x / NB;
        S.seq[i] = col < n ? seq[col] : (uint8_t)'N';
    }
    S.sorce = (int8_t*)calloc((size_t)(V + 1) * rowb, 1); S.esorce = (int8_t*)calloc((size_t)(V + 1) * rowb, 1);
    S.source = (int8_t*)calloc((size_t)V * rowb, 1); S.esource = (int8_t*)calloc((size_t)V * rowb, 1); S.fsource = (int8_t*)calloc((size_t)V * rowb, 1);
    S.simple = (int32_t*)calloc((size_t)(V + 1) * (S.nblk + 1), sizeof(int32_t));
    S.first_col = (int32_t*)calloc((size_t)V + 1, sizeof(int32_t));
    S.f0 = (int8_t*)calloc((size_t)V * S.maxp, 1);
    memset(S.sorce + (size_t)V * rowb, p->E, (size_t)L);                /* the virtual init row (:82-94) */
    memset(S.esorce + (size_t)V * rowb, p->E + p->E + p->O, (size_t)L);

    task8 T;
    const int num = S.maxp;
    T.f_temp = (int8_t*)malloc((size_t)num * NB); T.VC1 = (int8_t*)malloc((size_t)num * NB); T.VC2 = (int8_t*)malloc((size_t)num * NB);
    T.r_s = (int8_t*)malloc((size_t)num * NB); T.v0 = (int8_t*)malloc(num); T.vc_1 = (int8_t*)malloc(num); T.vc_2 = (int8_t*)malloc(num);
    T.te = (int*)malloc((size_t)num * sizeof(int));

    /* control(): anti-diagonals of (row block, column block) tasks (:1020-1044); inside one, block_alignment (:602-657) */
    const int tsl = (S.length1 + S.length2) / L - 1;
    int j = 0;
    for (int i = 0; i < tsl; i++) {
        if (i <= S.fmaxtag) j++;
        else if (i <= S.lmaxtag) ;
        else j--;
        for (int l = 0; l < j; l++) {
            const int nconv = S.maxtag > 0 ? ((i / S.maxtag) ? S.maxtag : i % S.maxtag) : 0;
            const int nv = nconv - l;
            const int a1 = (((i - S.maxtag) > 0 ? (i - S.maxtag) : 0) + l) * L;
            memset(T.f_temp, 0, (size_t)num * NB); memset(T.VC1, 0, (size_t)num * NB); memset(T.VC2, 0, (size_t)num * NB);
            memset(T.r_s, 0, (size_t)num * NB); memset(T.v0, 0, num); memset(T.vc_1, 0, num); memset(T.vc_2, 0, num);
            memset(T.te, 0, (size_t)num * sizeof(int));
            if (nv < 0 || nv >= S.nblk) continue;                        /* (never with the schedule above) */
            for (int k = 0; k < L; k++) {
                const int a2 = a1 + k;
                if (a2 >= V) break;
                line_i8(&S, &T, i, j, l, a2, nv);
            }
        }
    }
    /* plain (row, column) layout for the caller; NUM2 (:14): column x of a block sits at (x % W) * 64 + x / W */
    for (int32_t r = 0; r < V; r++) {
        for (int32_t c = 0; c < n; c++) {
            const int32_t x = c % L; const size_t o = (size_t)r * rowb + (size_t)(c / L) * L + (size_t)(x % Wv) * NB + x / Wv;
            if (source) source[(size_t)r * n + c] = S.source[o];
            if (esource) esource[(size_t)r * n + c] = S.esource[o];
            if (fsource) fsource[(size_t)r * n + c] = S.fsource[o];
        }
        if (lastscore) lastscore[r] = S.simple[(size_t)r * (S.nblk + 1) + S.nblk];
    }
    int32_t best = -1;                                                  /* c-t-simd.c:737-750 */
    for (int32_t r = 0; r < V; r++)
        if (g->is_sink[r] && (best < 0 || S.simple[(size_t)r * (S.nblk + 1) + S.nblk] > S.simple[(size_t)best * (S.nblk + 1) + S.nblk])) best = r;
    free(S.seq); free(S.sorce); free(S.esorce); free(S.source); free(S.esource); free(S.fsource); free(S.simple); free(S.first_col); free(S.f0);
    free(T.f_temp); free(T.VC1); free(T.VC2); free(T.r_s); free(T.v0); free(T.vc_1); free(T.vc_2); free(T.te);
    return best;
}
