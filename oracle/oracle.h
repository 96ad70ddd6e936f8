/*
 * oracle.h — scalar CPU restatement of TSTA's alignment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by or executed from the
 * product (tsta_amd/, include/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it, and there only as the checker / reported CPU baseline.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_golden.py) against
 *   - the reference's README known answers (README.md:16-40 of bxskdh/TSTA), and
 *   - golden vectors produced by the unmodified reference compiled from /root/reference by
 *     oracle/Makefile (`make ref`), committed under tests/golden/ with the generating script
 *     tests/golden/make_golden.py.
 *
 * All arithmetic is plain int32 (the reference's int8 difference encoding, striping and blocking are
 * implementation devices that do not change results inside the domain documented in DESIGN.md).
 */
#ifndef TSTA_ORACLE_H
#define TSTA_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_params {
    int32_t M;   /* match      (psa.c:114, default  2) */
    int32_t X;   /* mismatch   (psa.c:115, default -5) */
    int32_t E;   /* gap extend (psa.c:116, default -2) */
    int32_t O;   /* gap open   (psa.c:117, default -4) */
} oracle_params;

/* ------------------------------- PSA (psa/psa.c) ------------------------------------------ */

/* `maxsorce` as printed by TSTA_psa / TSTA_psa_notrace (psa.c:666): max(-100, max_{i,j>=1} H(i,j)).
 * s1/s2 are given in CLI order (-1, -2); the longer one becomes the column sequence (psa.c:194-199). */
int32_t oracle_psa_maxscore(const uint8_t* s1, int64_t l1, const uint8_t* s2, int64_t l2,
                            const oracle_params* p);

void oracle_psa_scores(const uint8_t* s1, int64_t l1, const uint8_t* s2, int64_t l2, const oracle_params* p,
                       int32_t* maxscore, int32_t* end_score);      /* + H(m,n), O(n) memory */

/* Batch of independent pairs (OpenMP over pairs when threads > 1).  seqs = concatenated bytes,
 * off1/off2 = start offsets, len1/len2 = lengths. */
void oracle_psa_maxscore_batch(int64_t n_pairs, const uint8_t* seqs, const int64_t* off1,
                               const int32_t* len1, const int64_t* off2, const int32_t* len2,
                               const oracle_params* p, int32_t* out, int threads);

/* Full TSTA_psa behaviour: fill (psa.c:234-363) + trace() (psa.c:435-518).
 * Outputs (malloc'd, caller frees with oracle_free): the two gapped rows; row_a belongs to the LONGER
 * input (">1" in the output file), row_b to the shorter.  *out_len = alignment length.
 * If end_score != NULL it receives H(m,n).  Returns maxsorce.  Memory: 3 bytes per DP cell. */
int32_t oracle_psa_align(const uint8_t* s1, int64_t l1, const uint8_t* s2, int64_t l2,
                         const oracle_params* p, char** row_a, char** row_b, int64_t* out_len,
                         int32_t* end_score);

/* The three trace matrices exactly as the reference's fill defines them (row-major [m][n], cell (i,j)
 * 1-based at [(i-1)*n + (j-1)]): back in {0,1,2} (psa.c:294-299), fback/eback in {1,2,-2}
 * (psa.c:302-327).  a = column sequence (length n), b = row sequence (length m): NO swap here. */
int32_t oracle_psa_fill_trace(const uint8_t* a, int64_t n, const uint8_t* b, int64_t m,
                              const oracle_params* p, int8_t* back, int8_t* fback, int8_t* eback,
                              int32_t* end_score);

/* --------------------------- POA fill (msa/c-t-simd.c) -------------------------------------- */

/* Flattened DAG in topological order (the order of p->sort[], topo.c:243-260).
 * Node r has predecessors pred[pred_off[r] .. pred_off[r+1]) (indices < r, in the order of
 * poa::pre[], poa.h:105); base[r] = poa::base; is_sink[r] = (out == 0). */
typedef struct oracle_dag {
    int32_t        n_nodes;
    const int32_t* pred_off;   /* n_nodes + 1 */
    const int32_t* pred;       /* pred_off[n_nodes] */
    const uint8_t* base;       /* n_nodes */
    const uint8_t* is_sink;    /* n_nodes */
} oracle_dag;

/* Sequence-vs-DAG affine fill (c-t-simd.c:165-599 via SURVEY Appendix A.2).
 * source/esource/fsource: int8 [n_nodes][n] row-major, column c (1-based) at [r*n + c-1] (plain
 * layout, NOT the reference's striped NUM2 layout).  Any of them may be NULL.
 * lastscore[r] = H(r,n) for every node (the reference keeps it for sinks only, c-t-simd.c:577-578).
 * Returns the index of the best sink (max lastscore, ties -> lowest index, c-t-simd.c:737-750),
 * or -1 if there is no sink. */
int32_t oracle_poa_fill(const oracle_dag* g, const uint8_t* seq, int32_t n, const oracle_params* p,
                        int8_t* source, int8_t* esource, int8_t* fsource, int32_t* lastscore);

/* The same fill in the blocking of the reference's AVX512 build at -W `W` (L = 64 W columns per block, 64 lane segments of W
 * columns): identical to oracle_poa_fill on 'N'-free sequences; on sequences with a literal 'N' it reproduces what the
 * reference computes at that -W (its first pass ignores the 'N' rule; see poa_oracle.c). */
int32_t oracle_poa_fill_blocked(const oracle_dag* g, const uint8_t* seq, int32_t n, const oracle_params* p, int32_t W,
                                int8_t* source, int8_t* esource, int8_t* fsource, int32_t* lastscore);

int32_t oracle_poa_blocked_max_shift(int reset);
/* The same fill in the reference's own number representation (int8 differences, saturating offsets with their overflow
 * reservoirs, 64 byte lanes, both passes, the block schedule of control()): poa_oracle_i8.c.  What the compiled AVX512
 * reference computes at -W W, including where that depends on its encoding (literal 'N', offsets beyond the cutoff). */
int32_t oracle_poa_fill_i8(const oracle_dag* g, const uint8_t* seq, int32_t n, const oracle_params* p, int32_t W,
                           int8_t* source, int8_t* esource, int8_t* fsource, int32_t* lastscore);

void oracle_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
