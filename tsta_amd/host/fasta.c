#include "fasta.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* the reference exit()s on unreadable input (seqio.c:38,47,180); here another thread may be in the middle of bringing the HIP
 * runtime up (cli_common.h), so leave without running exit handlers */
#include <unistd.h>
static void fasta_die(void) { fflush(stdout); fflush(stderr); _exit(1); }

static char* slurp(const char* path, size_t* out_len)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) {
        fprintf(stderr, "File %s does not exist.\n", path);
        fasta_die();
    }
    unsigned char magic[2] = {0, 0};
    size_t got = fread(magic, 1, 2, fp);
    fclose(fp);
    size_t cap = 1 << 20, len = 0;
    char* buf = (char*)malloc(cap);
    if (!buf) { fprintf(stderr, "out of memory\n"); fasta_die(); }
    /* gzread on a plain file passes the bytes through, so one code path serves both */
    gzFile gz = gzopen(path, "rb");
    if (!gz) { fprintf(stderr, "File %s does not exist.\n", path); fasta_die(); }
    (void)got; (void)magic;
    for (;;) {
        if (cap - len < (1 << 16)) {
            cap *= 2;
            buf = (char*)realloc(buf, cap);
            if (!buf) { fprintf(stderr, "out of memory\n"); fasta_die(); }
        }
        int n = gzread(gz, buf + len, (unsigned)(cap - len - 1));
        if (n <= 0) break;
        len += (size_t)n;
    }
    gzclose(gz);
    buf[len] = '\0';
    *out_len = len;
    return buf;
}

fasta_file* fasta_read(const char* path, size_t max_records)
{
    size_t len = 0;
    char* buf = slurp(path, &len);
    /* record type guess: first '>' or '@' anywhere (seqio.c:274-309) */
    size_t p = 0;
    while (p < len && buf[p] != '>' && buf[p] != '@') p++;
    fasta_file* f = (fasta_file*)calloc(1, sizeof(fasta_file));
    if (p < len && buf[p] == '@') {
        fprintf(stderr, "Cannot read fasta record from a fastq file.\n");
        fasta_die();
    }
    size_t cap = 0;
    while (p < len) {
        /* buf[p] == '>' : header line */
        while (p < len && buf[p] != '\n') p++;
        if (p < len) p++;
        size_t sl = 0, scap = 256;
        char* s = (char*)malloc(scap);
        while (p < len && buf[p] != '>') {
            size_t e = p;
            while (e < len && buf[e] != '\n') e++;
            size_t ll = e - p;
            if (ll && buf[p + ll - 1] == '\r') ll--;
            if (ll) {
                if (sl + ll + 1 > scap) { while (sl + ll + 1 > scap) scap *= 2; s = (char*)realloc(s, scap); }
                memcpy(s + sl, buf + p, ll);
                sl += ll;
            }
            p = (e < len) ? e + 1 : e;
        }
        s[sl] = '\0';
        if (f->count == cap) { cap = cap ? cap * 2 : 8; f->seqs = (fasta_seq*)realloc(f->seqs, cap * sizeof(fasta_seq)); }
        f->seqs[f->count].data = s;
        f->seqs[f->count].length = sl;
        f->count++;
        if (max_records && f->count >= max_records) break;
    }
    free(buf);
    return f;
}

void fasta_free(fasta_file* f)
{
    if (!f) return;
    for (size_t i = 0; i < f->count; i++) free(f->seqs[i].data);
    free(f->seqs);
    free(f);
}
