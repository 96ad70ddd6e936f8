/* poa_graph.h — host side of the TSTA_msa drop-in: the partial-order graph, the fusion of an aligned sequence
 * into it, the topological orders the fill consumes, and the result writer.
 *
 * Behavioural restatement (own data structures: an index-based node pool instead of malloc'd pointer webs) of
 *   poa_build_init   msa/c-t-simd.c:72-163
 *   control          msa/c-t-simd.c:979-1049   (the fill itself is delegated to a callback: the GPU's tsta_poa_fill)
 *   node_fuse        msa/c-t-simd.c:659-977
 *   t_sort           msa/topo.c:243-260  (toposort/tp :179-241, toposort1/tp1 :7-117, modify :119-177)
 *   printf_result    msa/result.c:6-79
 * These are serial, pointer-chasing O(V+E) steps that stay on the host; they consume the
 * trace codes the fill produces, which is what pins the fill's encoding.
 */
#ifndef TSTA_HOST_POA_GRAPH_H
#define TSTA_HOST_POA_GRAPH_H
#define PG_ERR_RING_FULL (-5)                     /* == TSTA_ERR_UNSUPPORTED: a sixth distinct character in one column */
#include <stdint.h>
#include <stdio.h>

#define PG_INIT (-1)          /* the virtual predecessor of source nodes ("init": sub -1, consensus score 0) */

typedef struct pg_node {
    int* pre;  int* ew;  int in;   int pre_cap;   /* predecessors (node ids, PG_INIT allowed in slot 0) + edge weights */
    int* next; int out;  int next_cap;            /* successors */
    int  ring[4]; int ring_n;                     /* mutually aligned ("mismatch") nodes, poa.h:109 */
    unsigned char* pass;                          /* passing_seq[sum] */
    unsigned char base;
    int  sub;                                     /* position in sort[] (-1 while outside the graph) */
    int  in_temp, passing, logo;                  /* scratch of the sorts / writer */
    int  cs_score, cs_len, cs_src;                /* consensus DP: node_sorce, node_base_len, node_sorce_source */
} pg_node;

typedef struct pg_graph {
    pg_node* nodes; int n_nodes, cap_nodes;       /* pool; ids are indices, never reused */
    int* unsort; int* sort; int len;              /* graph members, discovery order / topological order */
    int  last_node_num;                           /* number of sinks */
    int  sum;                                     /* number of sequences (width of pass[]) */
} pg_graph;

/* The fill: given the DAG in sort[] order (CSR predecessors in pre[] order) and the sequence, produce the three
 * trace arrays [V][n] (plain layout) and lastscore[V].  Returns 0 on success. */
typedef int (*pg_fill_fn)(void* ctx, int n_nodes, const int32_t* pred_off, const int32_t* pred, const uint8_t* base,
                          const uint8_t* is_sink, const uint8_t* seq, int32_t n, int8_t* source, int8_t* esource,
                          int8_t* fsource, int32_t* lastscore);

/* One visited cell of the traceback walk of node_fuse (c-t-simd.c:751-957): sort index of the node, sequence
 * position, the EFFECTIVE trace code there (after the forced-continuation overwrites, :767-769, :776-782) and, for
 * match / mismatch cells, whether the diagonal predecessor cell is itself a match (:790, :827, :873), which is all
 * the graph surgery needs to know about the trace arrays.  Same layout as tsta_poa_step (include/tsta_engine.h). */
typedef struct pg_step { int32_t node, col; int8_t code, peek; int16_t pad; } pg_step;

/* The walk done elsewhere (on the GPU, next to the trace arrays): returns malloc'd steps (freed by the caller with
 * free_steps), their count and the sequence position at which the walk stopped (-1 if the sequence was consumed). */
typedef int (*pg_path_fn)(void* ctx, int n_nodes, const int32_t* pred_off, const int32_t* pred, const uint8_t* base,
                          const uint8_t* is_sink, const uint8_t* seq, int32_t n, pg_step** steps, int64_t* n_steps,
                          int32_t* end_col, void (**free_steps)(void*));

pg_graph* pg_build_init(const char* s0, int sum);
int       pg_align_and_fuse(pg_graph* g, const char* seq, int num, pg_fill_fn fill, void* ctx);
int       pg_align_and_fuse_path(pg_graph* g, const char* seq, int num, pg_path_fn path, void* ctx);
void      pg_sort(pg_graph* g, int final_pass);
void      pg_write_result(pg_graph* g, int num, FILE* out);
void      pg_free(pg_graph* g);

#endif
