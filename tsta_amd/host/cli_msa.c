/* cli_msa.c — drop-in for the reference's TSTA_msa executable (msa/msa.c:62-151): same flags, defaults, stderr /
 * stdout lines and output-file bytes.  The sequence-vs-graph DP fill of every step runs on the MI355X
 * (tsta_poa_fill); graph fusion, topological sorting and the result writer are host C (poa_graph.c). */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>

#include "../../include/tsta_engine.h"
#include "msa_driver.h"

static void print_usage(void)
{   /* the reference's help text, byte for byte (msa.c:62-73) */
    printf("-M                      the sorce of match [default: 2]\n");
    printf("-X                      the sorce of dismatch [default: -5]\n");
    printf("-E                      the sorce of extend-gap [default: -2]\n");
    printf("-O                      the sorce of open-gap [default: -4]\n");
    printf("-T                      the number of threads [default: 10]\n");
    printf("-W                      the width of block(Multiplication of simd data width) [default: 16]\n");
    printf("-i                      the input sequence(fasta format)\n");
    printf("-o                      the output file [default: output.txt]\n");
    printf("example:\n./TSTA_msa -i seq.fa -o output.txt\n");
}

typedef struct gpu_ctx { tsta_engine* eng; tsta_params prm; double gpu_s; int warned; } gpu_ctx;

/* TSTA_WARN_REFERENCE_UNDEFINED: the result is valid, but the reference binary's own output for this input depends on its
 * -W / SIMD width (include/tsta_engine.h, tsta_poa_fill).  Said once per run on stderr; not an error. */
static int absorb_warning(gpu_ctx* c, int rc)
{
    if (rc <= 0) return rc;
    if (!c->warned) fprintf(stderr, "warning: %s\n", tsta_last_warning(c->eng));
    c->warned = 1;
    return 0;
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static int gpu_fill(void* vctx, int n_nodes, const int32_t* pred_off, const int32_t* pred, const uint8_t* base,
                    const uint8_t* is_sink, const uint8_t* seq, int32_t n, int8_t* source, int8_t* esource,
                    int8_t* fsource, int32_t* lastscore)
{
    gpu_ctx* c = (gpu_ctx*)vctx;
    tsta_dag_csr dag = { n_nodes, pred_off, pred, base, is_sink };
    int32_t best = -1;
    const double t0 = now_s();
    int rc = tsta_poa_fill(c->eng, &c->prm, &dag, seq, n, source, esource, fsource, lastscore, &best);
    c->gpu_s += now_s() - t0;
    rc = absorb_warning(c, rc);
    if (rc != TSTA_OK) fprintf(stderr, "tsta_poa_fill: %s: %s\n", tsta_strerror(rc), tsta_last_error(c->eng));
    return rc;
}

/* fill + traceback walk on the GPU: only the visited cells come back (tsta_poa_step == pg_step) */
static void free_gpu_steps(void* p) { tsta_free(p); }

static int gpu_path(void* vctx, int n_nodes, const int32_t* pred_off, const int32_t* pred, const uint8_t* base,
                    const uint8_t* is_sink, const uint8_t* seq, int32_t n, pg_step** steps, int64_t* n_steps,
                    int32_t* end_col, void (**free_steps)(void*))
{
    gpu_ctx* c = (gpu_ctx*)vctx;
    tsta_dag_csr dag = { n_nodes, pred_off, pred, base, is_sink };
    tsta_poa_step* st = NULL;
    const double t0 = now_s();
    int rc = tsta_poa_align(c->eng, &c->prm, &dag, seq, n, &st, n_steps, end_col, NULL);
    c->gpu_s += now_s() - t0;
    rc = absorb_warning(c, rc);
    if (rc != TSTA_OK) fprintf(stderr, "tsta_poa_align: %s: %s\n", tsta_strerror(rc), tsta_last_error(c->eng));
    *steps = (pg_step*)st;
    *free_steps = free_gpu_steps;
    return rc;
}

int main(int argc, char* argv[])
{
    gpu_ctx ctx = { NULL, { 2, -5, -2, -4 }, 0.0, 0 };
    int bS = 10;                                                     /* msa.c:21: default -W (the help text says 16) */
    const double t_start = now_s();
    const char *input = NULL, *output = "output.txt";
    int c;
    while ((c = getopt(argc, argv, "M:X:E:O:T:W:i:o:")) != -1) {
        switch (c) {
            case 'M': ctx.prm.M = atoi(optarg); break;
            case 'X': ctx.prm.X = atoi(optarg); break;
            case 'E': ctx.prm.E = atoi(optarg); break;
            case 'O': ctx.prm.O = atoi(optarg); break;
            case 'T': break;
            case 'W': bS = atoi(optarg); break;                      /* results never depend on it; it only decides what counts as reference-undefined */
            case 'i': input = optarg; break;
            case 'o': output = optarg; break;
            default: print_usage(); return 0;                          /* msa.c:109-111 */
        }
    }
    if (input == NULL) {                                             /* msa.c:115-119 */
        printf("input file is not specified\n");
        print_usage();
        return 0;
    }
    const char* dev = getenv("TSTA_DEVICE");
    int rc = tsta_engine_create(dev ? atoi(dev) : 0, &ctx.eng);
    if (rc != TSTA_OK) { fprintf(stderr, "tsta_engine_create: %s\n", tsta_strerror(rc)); return 1; }
    if (bS >= 1 && bS <= 16384) tsta_engine_set_option(ctx.eng, "poa_ref_block", 64 * (int64_t)bS);   /* L = block x W of the AVX512 reference build */
    const double t_init = now_s();
    /* default: the walk runs on the GPU next to the trace arrays; TSTA_MSA_HOST_WALK=1 copies the arrays back instead */
    if (getenv("TSTA_MSA_HOST_WALK")) rc = msa_run(input, output, gpu_fill, NULL, &ctx);
    else                              rc = msa_run(input, output, NULL, gpu_path, &ctx);
    if (getenv("TSTA_MSA_TIMING"))      /* diagnostics only: where the wall time of this run went */
        fprintf(stderr, "[tsta] engine create %.3f s, GPU fill+walk calls %.3f s, host (FASTA, graph fusion, sorts, writer) %.3f s\n",
                t_init - t_start, ctx.gpu_s, now_s() - t_init - ctx.gpu_s);
    tsta_engine_destroy(ctx.eng);
    return rc ? 1 : 0;
}

_Static_assert(sizeof(pg_step) == sizeof(tsta_poa_step), "pg_step and tsta_poa_step must have the same layout");
