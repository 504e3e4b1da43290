/* sa_io.h -- text-format loaders and writers used by the signalMachine drop-in (host only). */
#ifndef SA_IO_H_
#define SA_IO_H_
#include <stdint.h>
#include <stdio.h>
#include "signalalign_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* whole line without the newline, NULL at end of file (heap; caller frees) */
char *sa_read_line(FILE *f);
/* splits on blanks/tabs in place; returns token count, *toks is heap (caller frees the array only) */
int64_t sa_split_ws(char *line, char ***toks);

/* ---- .npRead (impl/nanopore.c:145-521; writer src/signalalign/nanoporeRead.py:438-540) ---- */
typedef struct sa_strand_params {
    double scale, shift, var, scale_sd, var_sd, drift, shift_sd;
} sa_strand_params_t;

typedef struct sa_npread {
    int64_t read_length, n_template_events, n_complement_events, template_read_length, complement_read_length;
    sa_strand_params_t template_params, complement_params;
    int two_d;
    char *two_d_read, *template_read, *complement_read;
    int64_t *template_strand_event_map, *complement_strand_event_map; /* per base of the strand read     */
    int64_t *template_event_map, *complement_event_map;               /* per base of the 2D read         */
    double *template_events, *complement_events;                      /* 4 doubles per event             */
} sa_npread_t;
int sa_npread_load(const char *path, sa_npread_t **out);
void sa_npread_free(sa_npread_t *r);

/* ---- guide alignment in exonerate cigar format (sonLib cigarRead; written by utils/bwaWrapper.py:213) ---- */
typedef struct sa_cigar {
    char *contig1, *contig2;     /* reference name, query name                  */
    int64_t start1, end1, start2, end2;
    int strand1, strand2;        /* 1 = '+'                                     */
    double score;
    int64_t n_ops;
    int32_t *op_type;            /* 0 = M, 1 = D (reference only), 2 = I (read only) */
    int64_t *op_len;
} sa_cigar_t;
int sa_cigar_load(const char *path, sa_cigar_t **out);
void sa_cigar_free(sa_cigar_t *c);

/* ---- FASTA + .fai: closed interval [start, end] like faidx_fetch_seq (impl/fasta_handler.c:19-39).
 * Returns a heap string, NULL if the file/index is unreadable; *err = -2 if the name is not in the index. */
char *sa_fasta_fetch(const char *fasta_path, const char *name, int64_t start, int64_t end_inclusive, int *err);

/* reverse complement of a nucleotide string (sonLib stString_reverseComplementString: A<->T, C<->G, rest kept) */
char *sa_reverse_complement(const char *s);
char *sa_complement(const char *s);
void sa_reverse_in_place(char *s);

#ifdef __cplusplus
}
#endif
#endif
