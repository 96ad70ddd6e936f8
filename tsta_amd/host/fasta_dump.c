/* fasta_dump.c — test helper: prints "<count>\n" then "<length> <sequence>\n" per record of a FASTA file. */
#include <stdio.h>
#include <stdlib.h>
#include "fasta.h"
int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    fasta_file* f = fasta_read(argv[1], argc > 2 ? (size_t)atoi(argv[2]) : 0);
    printf("%zu\n", f->count);
    for (size_t i = 0; i < f->count; i++) printf("%zu %s\n", f->seqs[i].length, f->seqs[i].data);
    fasta_free(f);
    return 0;
}
