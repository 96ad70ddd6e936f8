/*
 * psa_oracle.c — scalar restatement of TSTA's pairwise global affine-gap alignment (psa/psa.c).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Plain int32 Gotoh with the reference's exact conventions;
 * every rule cites the reference line it follows.  Parity: PINNED (tests/test_oracle_golden.py).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-1000000000)         /* plays the role of MIN=-100 in the int8 difference form, psa.c:110,244-246 */
#define MS_FLOOR (-100)               /* ms = MIN before any block reports, psa.c:110,590 */

static inline int32_t max2(int32_t a, int32_t b) { return a > b ? a : b; }

/* psa.c:194-199: if the -2 sequence is strictly longer the two are swapped, so that seq[0] (columns, ">1")
 * is the longer one; on equal length the -1 sequence stays seq[0]. */
static void orient(const uint8_t* s1, int64_t l1, const uint8_t* s2, int64_t l2,
                   const uint8_t** a, int64_t* n, const uint8_t** b, int64_t* m)
{
    if (l2 > l1) { *a = s2; *n = l2; *b = s1; *m = l1; }
    else         { *a = s1; *n = l1; *b = s2; *m = l2; }
}

/* Score-only fill: O(n) memory.  a = columns (n), b = rows (m).
 * Boundary (blockmatrix_init, psa.c:219-232): H(0,0)=0, H(0,j)=O+j*E, H(i,0)=O+i*E, no gap state on the
 * boundary (first E/F of a row/column is always an "open": esorce=2E+O / F=2E+O relative to the boundary).
 * Recurrence (row(), psa.c:250-354, in absolute form):
 *   Ev(i,j) = max(Ev(i-1,j)+E, H(i-1,j)+O+E)     psa.c:316-318
 *   Fh(i,j) = max(Fh(i,j-1)+E, H(i,j-1)+O+E)     psa.c:301-305 (and the lazy-F pass psa.c:250-274)
 *   H(i,j)  = max(H(i-1,j-1)+(a_j==b_i?M:X), Ev, Fh)   psa.c:252-259,289-290
 * maxsorce = max(-100, max over i,j>=1 of H(i,j))  psa.c:333-357,416-424,590,666 */
static int32_t fill_score(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                          const oracle_params* p, int32_t* end_score)
{
    const int32_t M = p->M, X = p->X, E = p->E, OE = p->O + p->E;
    int32_t ms = MS_FLOOR;
    if (n <= 0 || m <= 0) { if (end_score) *end_score = 0; return ms; }
    int32_t* H  = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
    int32_t* Ev = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
    H[0] = 0;
    for (int64_t j = 1; j <= n; j++) { H[j] = p->O + (int32_t)j * E; Ev[j] = NEG_INF; }
    for (int64_t i = 1; i <= m; i++) {
        int32_t diag = H[0];
        int32_t hleft = p->O + (int32_t)i * E;    /* H(i,0) */
        int32_t f = NEG_INF;                      /* Fh(i,0) */
        H[0] = hleft;
        const uint8_t bi = b[i - 1];
        for (int64_t j = 1; j <= n; j++) {
            int32_t e = max2(Ev[j] + E, H[j] + OE);
            f = max2(f + E, hleft + OE);
            int32_t d = diag + (a[j - 1] == bi ? M : X);
            int32_t h = max2(d, max2(e, f));
            diag = H[j];
            H[j] = h; Ev[j] = e; hleft = h;
            if (h > ms) ms = h;
        }
    }
    if (end_score) *end_score = H[n];
    free(H); free(Ev);
    return ms;
}

int32_t oracle_psa_maxscore(const uint8_t* s1, int64_t l1, const uint8_t* s2, int64_t l2,
                            const oracle_params* p)
{
    const uint8_t *a, *b; int64_t n, m;
    orient(s1, l1, s2, l2, &a, &n, &b, &m);
    return fill_score(a, n, b, m, p, NULL);
}

/* maxsorce and the global end score H(m,n) in one O(n)-memory pass */
void oracle_psa_scores(const uint8_t* s1, int64_t l1, const uint8_t* s2, int64_t l2, const oracle_params* p,
                       int32_t* maxscore, int32_t* end_score)
{
    const uint8_t *a, *b; int64_t n, m;
    orient(s1, l1, s2, l2, &a, &n, &b, &m);
    *maxscore = fill_score(a, n, b, m, p, end_score);
}

void oracle_psa_maxscore_batch(int64_t n_pairs, const uint8_t* seqs, const int64_t* off1,
                               const int32_t* len1, const int64_t* off2, const int32_t* len2,
                               const oracle_params* p, int32_t* out, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int64_t k = 0; k < n_pairs; k++)
        out[k] = oracle_psa_maxscore(seqs + off1[k], len1[k], seqs + off2[k], len2[k], p);
}

/* Fill with the three trace matrices of the TRACE build (psa.c:294-327). */
int32_t oracle_psa_fill_trace(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                              const oracle_params* p, int8_t* back, int8_t* fback, int8_t* eback,
                              int32_t* end_score)
{
    const int32_t M = p->M, X = p->X, E = p->E, OE = p->O + p->E;
    int32_t ms = MS_FLOOR;
    if (n <= 0 || m <= 0) { if (end_score) *end_score = 0; return ms; }
    int32_t* H  = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
    int32_t* Ev = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
    H[0] = 0;
    for (int64_t j = 1; j <= n; j++) { H[j] = p->O + (int32_t)j * E; Ev[j] = NEG_INF; }
    for (int64_t i = 1; i <= m; i++) {
        int32_t diag = H[0];
        int32_t hleft = p->O + (int32_t)i * E;
        int32_t f = NEG_INF;
        H[0] = hleft;
        const uint8_t bi = b[i - 1];
        int8_t* bk = back  + (size_t)(i - 1) * (size_t)n;
        int8_t* fb = fback + (size_t)(i - 1) * (size_t)n;
        int8_t* eb = eback + (size_t)(i - 1) * (size_t)n;
        for (int64_t j = 1; j <= n; j++) {
            const int32_t hup = H[j];
            const int32_t e_open = hup + OE,   e_ext = Ev[j] + E;
            const int32_t f_open = hleft + OE, f_ext = f + E;
            const int32_t e = max2(e_ext, e_open);
            f = max2(f_ext, f_open);
            const int32_t d = diag + (a[j - 1] == bi ? M : X);
            const int32_t h = max2(d, max2(e, f));
            /* psa.c:294-299: trace=2; s==f -> 0; s==h -> 1   (priority diag > left > up) */
            bk[j - 1] = (h == d) ? 1 : (h == f ? 0 : 2);
            /* psa.c:302-313: fv==ogap <=> F(i,j)==H(i,j-1)+O+E ("opened"); mask1 <=> F+E == H(i,j)+O+E */
            if (f != f_open) fb[j - 1] = 1;
            else             fb[j - 1] = (f + E == h + OE) ? -2 : 2;
            /* psa.c:315-327: same with the vertical gap */
            if (e != e_open) eb[j - 1] = 1;
            else             eb[j - 1] = (e + E == h + OE) ? -2 : 2;
            diag = hup;
            H[j] = h; Ev[j] = e; hleft = h;
            if (h > ms) ms = h;
        }
    }
    if (end_score) *end_score = H[n];
    free(H); free(Ev);
    return ms;
}

/* trace(), psa.c:435-518.  The reference overwrites back[] of the NEXT visited cell to force a gap
 * continuation (psa.c:450-459) and then walks the (modified) matrices a second time to emit; we do the
 * same literally: pass 1 mutates `back`, pass 2 emits. */
int32_t oracle_psa_align(const uint8_t* s1, int64_t l1, const uint8_t* s2, int64_t l2,
                         const oracle_params* p, char** row_a, char** row_b, int64_t* out_len,
                         int32_t* end_score)
{
    const uint8_t *a, *b; int64_t n, m;
    orient(s1, l1, s2, l2, &a, &n, &b, &m);
    size_t cells = (size_t)(n > 0 ? n : 0) * (size_t)(m > 0 ? m : 0);
    int8_t* back  = (int8_t*)malloc(cells ? cells : 1);
    int8_t* fback = (int8_t*)malloc(cells ? cells : 1);
    int8_t* eback = (int8_t*)malloc(cells ? cells : 1);
    int32_t ms = oracle_psa_fill_trace(a, n, b, m, p, back, fback, eback, end_score);
#define AT(mat, i, j) (mat)[(size_t)(i) * (size_t)n + (size_t)(j)]     /* 0-based like the reference */
    int64_t i = m - 1, j = n - 1, cnt = 0;
    while (i >= 0 && j >= 0) {                                           /* psa.c:441-463 */
        int8_t k = AT(back, i, j);
        if (k == 1) { i--; j--; }
        else if (k == 0) {
            int8_t fb = AT(fback, i, j);
            if (j - 1 >= 0 && ((fb == 1 || fb == -1) || ((fb == 2 || fb == -2) && AT(fback, i, j - 1) < 0)))
                AT(back, i, j - 1) = 0;
            j--;
        } else {
            int8_t eb = AT(eback, i, j);
            if (i - 1 >= 0 && ((eb == 1 || eb == -1) || ((eb == 2 || eb == -2) && AT(eback, i - 1, j) < 0)))
                AT(back, i - 1, j) = 2;
            i--; cnt++;
        }
    }
    if (i >= 0) cnt += i + 1;                                            /* psa.c:464-466 */
    int64_t len = n + cnt;
    char* ra = (char*)malloc((size_t)len + 1);
    char* rb = (char*)malloc((size_t)len + 1);
    ra[len] = rb[len] = '\0';
    int64_t q = len;
    i = m - 1; j = n - 1;
    while (i >= 0 && j >= 0) {                                           /* psa.c:472-496 */
        int8_t k = AT(back, i, j);
        if (k == 1)      { ra[q - 1] = (char)a[j]; rb[q - 1] = (char)b[i]; i--; j--; }
        else if (k == 0) { ra[q - 1] = (char)a[j]; rb[q - 1] = '-';        j--; }
        else             { ra[q - 1] = '-';        rb[q - 1] = (char)b[i]; i--; }
        q--;
    }
    while (j >= 0) { ra[q - 1] = (char)a[j]; rb[q - 1] = '-'; j--; q--; }    /* psa.c:497-503 */
    while (i >= 0) { rb[q - 1] = (char)b[i]; ra[q - 1] = '-'; i--; q--; }    /* psa.c:504-510 */
#undef AT
    free(back); free(fback); free(eback);
    *row_a = ra; *row_b = rb; *out_len = len;
    return ms;
}

void oracle_free(void* p) { free(p); }
