/* cli_psa.c — drop-in for the reference's TSTA_psa (-DTRACE) and TSTA_psa_notrace executables
 * (psa/psa.c:539-668): same flags, defaults, stdout and output-file bytes; the DP fill and traceback run on the
 * MI355X through the C-ABI (include/tsta_engine.h).  Host code stays C, as in the reference.
 *
 *   -M -X -E -O   scoring (defaults 2 -5 -2 -4, psa.c:113-117)
 *   -T -W         accepted for compatibility (CPU thread count / block width: results do not depend on them)
 *   -1 -2         input FASTA (first record of each, psa.c:192-193); -o output file (TRACE build only)
 *   env TSTA_DEVICE = GPU index (default 0)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/tsta_engine.h"
#include "cli_common.h"
#include "fasta.h"

static void print_usage(void)
{   /* the reference's help text, byte for byte (psa.c:521-537), incl. its "-W default 16" inaccuracy */
    printf("-M                      the sorce of match [default: 2]\n");
    printf("-X                      the sorce of dismatch [default: -5]\n");
    printf("-E                      the sorce of extend-gap [default: -2]\n");
    printf("-O                      the sorce of open-gap [default: -4]\n");
    printf("-T                      the number of threads [default: 10]\n");
    printf("-W                      the width of block(Multiplication of simd data width) [default: 16]\n");
    printf("-1                      the input sequence1(fasta)\n");
    printf("-2                      the input sequence2(fasta)\n");
#ifdef TRACE
    printf("-o                      the output file [default: output.txt]\n");
    printf("example:\n./TSTA_psa -1 seq1.fa -2 seq2.fa -o output.txt\n");
#else
    printf("example:\n./TSTA_psa -1 seq1.fa -2 seq2.fa\n");
#endif
}

static int fail(tsta_engine* eng, int rc, const char* what)
{
    fprintf(stderr, "%s: %s%s%s\n", what, tsta_strerror(rc), eng ? ": " : "", eng ? tsta_last_error(eng) : "");
    cli_quick_exit(1);
    return 1;
}

int main(int argc, char* argv[])
{
    tsta_params prm = {2, -5, -2, -4};
    const char *input1 = NULL, *input2 = NULL, *output = "output.txt";
    int c;
    while ((c = getopt(argc, argv, "M:X:E:O:T:W:1:2:o:")) != -1) {
        switch (c) {
            case 'M': prm.M = atoi(optarg); break;
            case 'X': prm.X = atoi(optarg); break;
            case 'E': prm.E = atoi(optarg); break;
            case 'O': prm.O = atoi(optarg); break;
            case 'T': break;
            case 'W': break;
            case '1': input1 = optarg; break;
            case '2': input2 = optarg; break;
            case 'o': output = optarg; break;
            default: print_usage(); return 0;                    /* psa.c:577-579 */
        }
    }
    if (input1 == NULL || input2 == NULL) {                      /* psa.c:583-587 */
        printf("input file is not specified\n");
        print_usage();
        return 0;
    }
    cli_engine boot;
    cli_engine_start(&boot);                                     /* HIP start-up + engine in a helper thread while the FASTA files are read */
    fasta_file* f1 = fasta_read(input1, 1);
    fasta_file* f2 = fasta_read(input2, 1);
    if (f1->count == 0 || f2->count == 0) { fprintf(stderr, "no FASTA record found\n"); cli_quick_exit(1); }
    const fasta_seq* s1 = &f1->seqs[0];
    const fasta_seq* s2 = &f2->seqs[0];

    tsta_engine* eng = NULL;
    int rc = cli_engine_get(&boot, &eng);
    if (rc != TSTA_OK) return fail(NULL, rc, "tsta_engine_create");
    int32_t ms = 0;
#ifdef TRACE
    char *row_a = NULL, *row_b = NULL;
    int64_t alen = 0;
    rc = tsta_psa_align(eng, &prm, (const uint8_t*)s1->data, (int64_t)s1->length, (const uint8_t*)s2->data,
                        (int64_t)s2->length, &ms, &row_a, &row_b, &alen);
    if (TSTA_FAILED(rc)) return fail(eng, rc, "tsta_psa_align");
    FILE* fptr = fopen(output, "w");                             /* psa.c:641, 512-515 */
    if (!fptr) { fprintf(stderr, "cannot open %s\n", output); return 1; }
    fputs(">1\n", fptr); fputs(row_a, fptr); fputs("\n>2\n", fptr); fputs(row_b, fptr);
    fclose(fptr);
    tsta_free(row_a); tsta_free(row_b);
#else
    (void)output;
    {
        size_t tot = s1->length + s2->length;
        uint8_t* both = (uint8_t*)malloc(tot ? tot : 1);
        memcpy(both, s1->data, s1->length);
        memcpy(both + s1->length, s2->data, s2->length);
        int64_t off1 = 0, off2 = (int64_t)s1->length;
        int32_t l1 = (int32_t)s1->length, l2 = (int32_t)s2->length;
        rc = tsta_psa_score_batch(eng, &prm, 1, both, (int64_t)tot, &off1, &l1, &off2, &l2, &ms);
        free(both);
        if (TSTA_FAILED(rc)) return fail(eng, rc, "tsta_psa_score_batch");
    }
#endif
    /* rc > 0: valid result, but for this scoring the reference binary's own output is a property of its int8 encoding (SIMD width,
     * -W): stdout and the output file keep the reference's format, the note goes to stderr */
    if (rc == TSTA_WARN_REFERENCE_UNDEFINED) fprintf(stderr, "warning: %s\n", tsta_last_warning(eng));
    printf("maxsorce=%d\n", ms);                                 /* psa.c:666 */
    /* every output is written and closed: no need to walk the runtime's exit handlers (cli_common.h) */
    cli_quick_exit(0);
    return 0;
}
