/* msa_oracle_main.c — TEST DRIVER (not product): runs the host POA graph code (tsta_amd/host/poa_graph.c,
 * msa_driver.c) with the CPU oracle's fill in place of the GPU's, so that node_fuse / t_sort / printf_result can be
 * checked against the compiled reference in a container without a GPU.
 *   usage: msa_oracle_main M X E O input.fa output.txt [W]      (W: fill in the reference's AVX512 blocking at -W W) */
#include <stdio.h>
#include <stdlib.h>
#include "../../oracle/oracle.h"
#include "../../tsta_amd/host/msa_driver.h"

static int g_w = 0;      /* > 0: the fill in the reference's blocking at that -W (oracle_poa_fill_blocked); < 0: the same in the
                          * reference's own int8 number representation (oracle_poa_fill_i8) at -W |g_w| */

static int oracle_fill(void* vctx, int n_nodes, const int32_t* pred_off, const int32_t* pred, const uint8_t* base,
                       const uint8_t* is_sink, const uint8_t* seq, int32_t n, int8_t* source, int8_t* esource,
                       int8_t* fsource, int32_t* lastscore)
{
    oracle_dag dag = { n_nodes, pred_off, pred, base, is_sink };
    if (g_w < 0) oracle_poa_fill_i8(&dag, seq, n, (const oracle_params*)vctx, -g_w, source, esource, fsource, lastscore);
    else if (g_w > 0) oracle_poa_fill_blocked(&dag, seq, n, (const oracle_params*)vctx, g_w, source, esource, fsource, lastscore);
    else         oracle_poa_fill(&dag, seq, n, (const oracle_params*)vctx, source, esource, fsource, lastscore);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 7) return 2;
    oracle_params p = { atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]) };
    if (argc > 7) g_w = atoi(argv[7]);
    const int rc = msa_run(argv[5], argv[6], oracle_fill, NULL, &p);
    if (g_w > 0) fprintf(stderr, "max_shift: %d\n", (int)oracle_poa_blocked_max_shift(0));
    return rc;
}
