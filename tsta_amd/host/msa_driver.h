/* msa_driver.h — the TSTA_msa main loop (msa/msa.c:36-60 readseq, :123-144 driver) over poa_graph + a fill. */
#ifndef TSTA_HOST_MSA_DRIVER_H
#define TSTA_HOST_MSA_DRIVER_H
#include "poa_graph.h"
/* Reads every FASTA record of `input`, builds the POA progressively and writes the result file.
 * stderr "seq_num: %d\n" (msa.c:47), stdout progress "\r[i/N]" every 100 sequences, "\r[N/N]\n" (msa.c:135-139).
 * Returns 0, or the fill's error code. */
/* Exactly one of `fill` (returns the full trace arrays; the host walks them) and `path` (the walk was done next to
 * the arrays, e.g. on the GPU) is non-NULL. */
int msa_run(const char* input, const char* output, pg_fill_fn fill, pg_path_fn path, void* ctx);
#endif
