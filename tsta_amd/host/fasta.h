/* fasta.h — minimal FASTA reader for the drop-in CLIs (observable behaviour of the reference's seqio reader for
 * the cases TSTA uses: seqio.c:185-246 open + gzip sniff, :463-541 seqioReadFasta, :426-461 readUntil). */
#ifndef TSTA_HOST_FASTA_H
#define TSTA_HOST_FASTA_H
#include <stddef.h>

typedef struct fasta_seq {
    char*  data;     /* NUL-terminated */
    size_t length;
} fasta_seq;

typedef struct fasta_file {
    fasta_seq* seqs;
    size_t     count;
} fasta_file;

/* Reads every record of a FASTA file (plain or gzip, detected by the 1f 8b magic like seqio.c:203-210).
 * Sequence = concatenation of the lines after the header up to the next line starting with '>', with the line's
 * trailing '\r' removed and empty lines skipped; bytes are kept as they are (no case folding).
 * Missing file: prints "File <name> does not exist." to stderr and exits 1 (seqio.c:175-183); a FASTQ file
 * ('@' seen before '>'): "Cannot read fasta record from a fastq file." and exit 1 (seqio.c:416-423).
 * max_records = 0 reads all records. */
fasta_file* fasta_read(const char* path, size_t max_records);
void        fasta_free(fasta_file* f);

#endif
