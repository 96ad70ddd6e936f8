#include "msa_driver.h"

#include <stdio.h>
#include <stdlib.h>

#include "fasta.h"

static int step(pg_graph* g, const char* seq, int num, pg_fill_fn fill, pg_path_fn path, void* ctx)
{
    return path ? pg_align_and_fuse_path(g, seq, num, path, ctx) : pg_align_and_fuse(g, seq, num, fill, ctx);
}

int msa_run(const char* input, const char* output, pg_fill_fn fill, pg_path_fn path, void* ctx)
{
    fasta_file* f = fasta_read(input, 0);
    const int N = (int)f->count;
    fprintf(stderr, "seq_num: %d\n", N);                                   /* msa.c:47 */
    if (N == 0) { fasta_free(f); return 0; }
    pg_graph* g = pg_build_init(f->seqs[0].data, N);                       /* msa.c:128 */
    int rc = 0;
    for (int i = 1; i < N - 1 && rc == 0; i++) {                           /* msa.c:130-137 */
        rc = step(g, f->seqs[i].data, i, fill, path, ctx);
        if (rc) break;
        pg_sort(g, 0);
        if (i % 100 == 0) printf("\r[%d/%d]", i, N);
    }
    if (rc == 0) {
        printf("\r[%d/%d]", N, N);
        printf("\n");
        fflush(stdout);
        rc = step(g, f->seqs[N - 1].data, N - 1, fill, path, ctx);   /* msa.c:140 (N == 1: S[0] against its own chain) */
    }
    if (rc == 0) {
        pg_sort(g, 1);                                                     /* msa.c:141 */
        FILE* res = fopen(output, "w");
        if (!res) { fprintf(stderr, "cannot open %s\n", output); rc = 1; }
        else { pg_write_result(g, N, res); fclose(res); }
    }
    pg_free(g);
    fasta_free(f);
    return rc;
}
