/* cli_psa_batch.c — batch front-end the reference does not have: scores record k of the -1
 * multi-FASTA against record k of the -2 multi-FASTA, all pairs in ONE tsta_psa_score_batch call (packed kernel,
 * pairs ordered by geometry on the device).  Same scoring flags as TSTA_psa_notrace; prints one line per pair,
 * "maxsorce=%d" exactly as the reference prints it for a single pair (psa.c:666).
 * With -o <file>: every pair is aligned WITH traceback (tsta_psa_align_batch, the pairs' fills in flight together)
 * and <file> receives, per pair, exactly the bytes TSTA_psa writes to its output file (">1\n" row ">\n>2\n" row,
 * psa.c:512-516) followed by one "\n" as record separator.
 * With -A: ALL pairs — every record of -1 against every record of -2 (row-major: -2 varies fastest); each sequence is
 * stored once on the device, the pairs only carry offsets.  Lines then read "<i>\t<j>\tmaxsorce=%d" (1-based records).
 * With -G <n>: the batch is sharded over n GPUs of this node (0 = all of them; tsta_psa_score_batch_multi /
 * tsta_psa_align_batch_multi: contiguous shards balanced by DP cells, one engine per device; -R gathers the scores with
 * RCCL over xGMI instead of one copy per device).  Output is identical whatever -G says.
 *   TSTA_psa_batch -1 a.fa -2 b.fa [-M -X -E -O] [-A] [-G n [-R]] [-o alignments.txt]    env TSTA_DEVICE = GPU index (without -G) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/tsta_engine.h"
#include "fasta.h"

int main(int argc, char* argv[])
{
    tsta_params prm = {2, -5, -2, -4};
    const char *input1 = NULL, *input2 = NULL, *output = NULL;
    int all_pairs = 0, n_gpus = -1, rccl = 0;
    int c;
    while ((c = getopt(argc, argv, "M:X:E:O:T:W:1:2:o:AG:R")) != -1) {
        switch (c) {
            case 'M': prm.M = atoi(optarg); break;
            case 'X': prm.X = atoi(optarg); break;
            case 'E': prm.E = atoi(optarg); break;
            case 'O': prm.O = atoi(optarg); break;
            case 'T': case 'W': break;
            case '1': input1 = optarg; break;
            case '2': input2 = optarg; break;
            case 'o': output = optarg; break;
            case 'A': all_pairs = 1; break;
            case 'G': n_gpus = atoi(optarg); break;
            case 'R': rccl = 1; break;
            default: fprintf(stderr, "usage: TSTA_psa_batch -1 a.fa -2 b.fa [-M 2 -X -5 -E -2 -O -4] [-A] [-G gpus [-R]] [-o alignments.txt]\n"); return 2;
        }
    }
    if (!input1 || !input2) { fprintf(stderr, "usage: TSTA_psa_batch -1 a.fa -2 b.fa [-M 2 -X -5 -E -2 -O -4] [-A] [-G gpus [-R]] [-o alignments.txt]\n"); return 2; }
    fasta_file* f1 = fasta_read(input1, 0);
    fasta_file* f2 = fasta_read(input2, 0);
    if (!all_pairs && f1->count != f2->count) { fprintf(stderr, "record counts differ: %zu vs %zu (use -A for all pairs)\n", f1->count, f2->count); return 1; }
    const size_t n1 = f1->count, n2 = f2->count;
    const size_t n = all_pairs ? n1 * n2 : n1;
    if (n > ((size_t)1 << 24)) { fprintf(stderr, "too many pairs: %zu\n", n); return 1; }
    /* every sequence once: all -1 records, then all -2 records */
    size_t total = 0;
    for (size_t k = 0; k < n1; k++) total += f1->seqs[k].length;
    for (size_t k = 0; k < n2; k++) total += f2->seqs[k].length;
    uint8_t* seqs = (uint8_t*)malloc(total ? total : 1);
    int64_t* start1 = (int64_t*)malloc((n1 ? n1 : 1) * sizeof(int64_t)); int64_t* start2 = (int64_t*)malloc((n2 ? n2 : 1) * sizeof(int64_t));
    size_t pos = 0;
    for (size_t k = 0; k < n1; k++) { start1[k] = (int64_t)pos; memcpy(seqs + pos, f1->seqs[k].data, f1->seqs[k].length); pos += f1->seqs[k].length; }
    for (size_t k = 0; k < n2; k++) { start2[k] = (int64_t)pos; memcpy(seqs + pos, f2->seqs[k].data, f2->seqs[k].length); pos += f2->seqs[k].length; }
    int64_t* off1 = (int64_t*)malloc((n ? n : 1) * sizeof(int64_t)); int64_t* off2 = (int64_t*)malloc((n ? n : 1) * sizeof(int64_t));
    int32_t* len1 = (int32_t*)malloc((n ? n : 1) * sizeof(int32_t)); int32_t* len2 = (int32_t*)malloc((n ? n : 1) * sizeof(int32_t));
    int32_t* out = (int32_t*)malloc((n ? n : 1) * sizeof(int32_t));
    for (size_t k = 0; k < n; k++) {
        const size_t i = all_pairs ? k / n2 : k, j = all_pairs ? k % n2 : k;
        off1[k] = start1[i]; len1[k] = (int32_t)f1->seqs[i].length;
        off2[k] = start2[j]; len2[k] = (int32_t)f2->seqs[j].length;
    }
    const char* dev = getenv("TSTA_DEVICE");
    tsta_engine* eng = NULL;
    tsta_multi* multi = NULL;
    int rc;
    if (n_gpus >= 0) {
        rc = tsta_multi_create(NULL, n_gpus, &multi);
        if (rc != TSTA_OK) { fprintf(stderr, "tsta_multi_create(%d GPUs): %s\n", n_gpus, tsta_strerror(rc)); return 1; }
        if (rccl) tsta_multi_set_option(multi, "gather", 1);
    } else {
        rc = tsta_engine_create(dev ? atoi(dev) : 0, &eng);
        if (rc != TSTA_OK) { fprintf(stderr, "tsta_engine_create: %s\n", tsta_strerror(rc)); return 1; }
    }
    if (output) {
        tsta_alignment* al = (tsta_alignment*)calloc(n ? n : 1, sizeof(tsta_alignment));
        FILE* fo = fopen(output, "w");
        if (!fo) { fprintf(stderr, "cannot open %s\n", output); return 1; }
        rc = multi ? tsta_psa_align_batch_multi(multi, &prm, (int64_t)n, seqs, (int64_t)total, off1, len1, off2, len2, al)
                   : tsta_psa_align_batch(eng, &prm, (int64_t)n, seqs, (int64_t)total, off1, len1, off2, len2, al);
        if (TSTA_FAILED(rc)) { fprintf(stderr, "tsta_psa_align_batch: %s: %s\n", tsta_strerror(rc), multi ? tsta_multi_last_error(multi) : tsta_last_error(eng)); return 1; }
        for (size_t k = 0; k < n; k++) {
            fputs(">1\n", fo); fwrite(al[k].row_a, 1, (size_t)al[k].len, fo);
            fputs("\n>2\n", fo); fwrite(al[k].row_b, 1, (size_t)al[k].len, fo);
            fputc('\n', fo);
            out[k] = al[k].maxscore;
        }
        fclose(fo);
        tsta_alignments_free(al, (int64_t)n);
        free(al);
    } else {
        rc = multi ? tsta_psa_score_batch_multi(multi, &prm, (int64_t)n, seqs, (int64_t)total, off1, len1, off2, len2, out)
                   : tsta_psa_score_batch(eng, &prm, (int64_t)n, seqs, (int64_t)total, off1, len1, off2, len2, out);
        if (TSTA_FAILED(rc)) { fprintf(stderr, "tsta_psa_score_batch: %s: %s\n", tsta_strerror(rc), multi ? tsta_multi_last_error(multi) : tsta_last_error(eng)); return 1; }
    }
    if (rc == TSTA_WARN_REFERENCE_UNDEFINED)      /* valid results; the reference's own output for this scoring depends on SIMD width / -W */
        fprintf(stderr, "warning: %s\n", tsta_last_warning(multi ? tsta_multi_engine(multi, 0) : eng));
    for (size_t k = 0; k < n; k++) {
        if (all_pairs) printf("%zu\t%zu\tmaxsorce=%d\n", k / n2 + 1, k % n2 + 1, out[k]);
        else           printf("maxsorce=%d\n", out[k]);
    }
    tsta_engine_destroy(eng);
    tsta_multi_destroy(multi);
    fasta_free(f1); fasta_free(f2);
    free(seqs); free(off1); free(off2); free(len1); free(len2); free(out); free(start1); free(start2);
    return 0;
}
