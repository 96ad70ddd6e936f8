/* cli_msa.c — drop-in for the reference's TSTA_msa executable (msa/msa.c:62-151): same flags, defaults, stderr /
 * stdout lines and output-file bytes.  The sequence-vs-graph DP fill of every step runs on the MI355X
 * (tsta_poa_fill); graph fusion, topological sorting and the result writer are host C (poa_graph.c). */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>

#include "../../include/tsta_engine.h"
#include "cli_common.h"
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

typedef struct gpu_ctx { tsta_engine* eng; tsta_params prm; double gpu_s; int warned; cli_engine boot; int bS; double t_ready; } gpu_ctx;

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

/* the engine is created by a helper thread while the FASTA is read and the first graph built: the first GPU call waits for it */
static int engine_ready(gpu_ctx* c)
{
    if (c->eng) return TSTA_OK;
    int rc = cli_engine_get(&c->boot, &c->eng);
    c->t_ready = now_s();
    if (rc != TSTA_OK) { fprintf(stderr, "tsta_engine_create: %s\n", tsta_strerror(rc)); return rc; }
    if (c->bS >= 1 && c->bS <= 16384) tsta_engine_set_option(c->eng, "poa_ref_block", 64 * (int64_t)c->bS);   /* L = block x W of the AVX512 reference build */
    return TSTA_OK;
}

static int gpu_fill(void* vctx, int n_nodes, const int32_t* pred_off, const int32_t* pred, const uint8_t* base,
                    const uint8_t* is_sink, const uint8_t* seq, int32_t n, int8_t* source, int8_t* esource,
                    int8_t* fsource, int32_t* lastscore)
{
    gpu_ctx* c = (gpu_ctx*)vctx;
    tsta_dag_csr dag = { n_nodes, pred_off, pred, base, is_sink };
    int32_t best = -1;
    int rc = engine_ready(c);
    if (rc != TSTA_OK) return rc;
    const double t0 = now_s();
    rc = tsta_poa_fill(c->eng, &c->prm, &dag, seq, n, source, esource, fsource, lastscore, &best);
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
    int rc = engine_ready(c);
    if (rc != TSTA_OK) return rc;
    const double t0 = now_s();
    rc = tsta_poa_align(c->eng, &c->prm, &dag, seq, n, &st, n_steps, end_col, NULL);
    c->gpu_s += now_s() - t0;
    if (getenv("TSTA_MSA_SPREAD")) {                /* diagnostics: how far apart the predecessor rows lay, which path served the step */
        int64_t sp = 0, rw = 0;
        tsta_engine_stat(c->eng, "poa_last_spread", &sp); tsta_engine_stat(c->eng, "poa_last_ref_w", &rw);
        fprintf(stderr, "[tsta] step: %d nodes x %d columns, predecessor spread %lld, %s\n", n_nodes, n, (long long)sp,
                rw > 0 ? "reference arithmetic (K4e)" : "plain kernels");
    }
    rc = absorb_warning(c, rc);
    if (rc != TSTA_OK) fprintf(stderr, "tsta_poa_align: %s: %s\n", tsta_strerror(rc), tsta_last_error(c->eng));
    *steps = (pg_step*)st;
    *free_steps = free_gpu_steps;
    return rc;
}

int main(int argc, char* argv[])
{
    gpu_ctx ctx = { NULL, { 2, -5, -2, -4 }, 0.0, 0, { 0 }, 10, 0.0 };
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
    ctx.bS = bS;                                                     /* results never depend on it; it only decides what counts as reference-undefined */
    cli_engine_start(&ctx.boot);                                     /* HIP start-up + engine in a helper thread, from here on */
    int rc;
    /* default: the walk runs on the GPU next to the trace arrays; TSTA_MSA_HOST_WALK=1 copies the arrays back instead */
    if (getenv("TSTA_MSA_HOST_WALK")) rc = msa_run(input, output, gpu_fill, NULL, &ctx);
    else                              rc = msa_run(input, output, NULL, gpu_path, &ctx);
    if (!ctx.eng) { tsta_engine* e = NULL; (void)cli_engine_get(&ctx.boot, &e); ctx.eng = e; ctx.t_ready = now_s(); }   /* (fewer than two sequences: nothing was aligned) */
    if (getenv("TSTA_MSA_TIMING"))      /* diagnostics only: where the wall time of this run went */
        fprintf(stderr, "[tsta] engine ready %.3f s after start, GPU fill+walk calls %.3f s, everything else (FASTA, graph fusion, sorts, writer) %.3f s\n",
                ctx.t_ready - t_start, ctx.gpu_s, now_s() - t_start - ctx.gpu_s - (ctx.t_ready - t_start));
    /* the output file is written and closed (msa_run): no need to walk the runtime's exit handlers */
    cli_quick_exit(rc ? 1 : 0);
    return rc ? 1 : 0;
}

_Static_assert(sizeof(pg_step) == sizeof(tsta_poa_step), "pg_step and tsta_poa_step must have the same layout");
