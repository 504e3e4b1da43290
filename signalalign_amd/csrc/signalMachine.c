/* signalMachine -- drop-in replacement of signalAlign's per-read aligner executable (impl/signalMachine.c).
 *
 * Same argv (getopt table impl/signalMachine.c:514-543), same input files (.model, .nhdp, .npRead, exonerate
 * cigar, indexed FASTA), same outputs: the three TSV renderings appended to -u / -i, the summary line on
 * stdout and the "SUCCESS" line on stderr that signalAlignment.py keys on (src/signalalign/signalAlignment.py:480).
 * The banded pair-HMM itself runs on the MI355X through libsignalalign_hip.so; there is no CPU fallback.
 *
 * Expectations mode (-t/-c) runs sa_expect_batch and writes the .expectations files of impl/continuousHmm.c.
 *
 * Batch front door (not in the reference): --batch <manifest> aligns many reads in ONE
 * process and ONE GPU batch per strand model -- models are parsed once, HIP is initialised once, and the reads run
 * side by side on the device.  Manifest: one read per line, tab separated, '#' comments,
 *     label  npRead  cigar_file  posteriors_out  [posteriors_out2|-]  [sequence_name|-]  [template_expectations|-]  [complement_expectations|-]
 * Everything else (models, references, thresholds, output format) comes from the usual options; --device <n> picks the
 * GPU (one process per GPU, each with its share of the manifest: reads are independent).  Per read the same
 * files, stdout summary line and stderr SUCCESS line are produced as by one single-read invocation.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <inttypes.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "sa_io.h"
#include "signalalign_hip.h"

#define PROB_1 10000000.0

static void die(const char *fmt, const char *a) { /* st_errAbort: message to stderr, non-zero exit */
    fprintf(stderr, fmt, a ? a : "");
    fputc('\n', stderr);
    exit(1);
}

static void usage(void) {
    fprintf(stderr, "\n\tsignalMachine - Align ONT ionic current to a reference sequence\n\n");
    fprintf(stderr, "--help: Display this super useful message and exit\n");
    fprintf(stderr, "--sm3Hdp, -d: Flag, enable HMM-HDP model\n");
    fprintf(stderr, "--twoD, -e: Flag, use 2D workflow (enables complement alignment)\n");
    fprintf(stderr, "-s: Output format, 0=full, 1=variantCaller, 2=assignments\n");
    fprintf(stderr, "-o: Degernate, 0=C/E, 1=C/E/O, 2=A/I, 3=A/C/G/T, 4=J/T, 5=A/F");
    fprintf(stderr, "-T: Template HMM model\n");
    fprintf(stderr, "-C: Complement HMM model\n");
    fprintf(stderr, "-L: Read (output) label\n");
    fprintf(stderr, "-q: NanoporeRead (in npRead format)\n");
    fprintf(stderr, "-f: Forward reference to align to as a flat file\n");
    fprintf(stderr, "-b: Backward reference to align to as a flat file\n");
    fprintf(stderr, "-p: Guide alignment file, containing CIGARs in EXONERATE format\n");
    fprintf(stderr, "-u: Posteriors (output) file path, place to put the output\n");
    fprintf(stderr, "-v: TemplateHDP file\n");
    fprintf(stderr, "-w: Complement HDP file\n");
    fprintf(stderr, "-t: Template expectations (HMM transitions) output location\n");
    fprintf(stderr, "-c: Complement expectations (HMM transitions) output location\n");
    fprintf(stderr, "-x: Diagonal expansion, how much to expand the dynamic programming envelope\n");
    fprintf(stderr, "-D: Posterior probability threshold, keep aligned pairs with posterior prob >= this\n");
    fprintf(stderr, "-m: Constranint trim, how much to trim the guide alignment anchors by\n");
    fprintf(stderr, "-g: traceBackDiagonals, how many backward diagonals to calculate during traceback\n");
    fprintf(stderr, "-r: boolean option if read is RNA\n");
    fprintf(stderr, "--batch <manifest>: align many reads in one process (one GPU batch per strand model)\n");
    fprintf(stderr, "--batch-reads <n>: reads per GPU batch of a manifest (default 2048)\n");
    fprintf(stderr, "--emission <meanOnly|twoDist>: match emission of a Gaussian model (default meanOnly, what the reference's\n"
                    "                  signalMachine installs; twoDist adds the inverse Gaussian on the event noise; single read only)\n");
    fprintf(stderr, "--device <n>: GPU to use\n");
    fprintf(stderr, "--mea: also write <posteriors file>.mea, the rows of the full output on the maximum expected accuracy path\n\n");
}

static double descale(double e, double level, double scale, double shift, double var) {
    return (e + var * level - scale * level - shift) / var;
}

typedef struct {
    sa_model_t *model;
    double *table;          /* EMISSION_MATCH_MATRIX as the DP uses it (HDP expected means once they are set)  */
    double *table_orig;     /* as loaded: what the per-read parameter estimation starts from                  */
    char alphabet[64];
    int n_alpha, k;
} strand_model_t;

static void kmer_string(const strand_model_t *sm, int32_t id, char *out) {
    for (int i = sm->k - 1; i >= 0; i--) {
        out[i] = sm->alphabet[id % sm->n_alpha];
        id /= sm->n_alpha;
    }
    out[sm->k] = 0;
}

/* adjustReferenceCoordinate, impl/signalMachine.c:54-62 */
static int64_t adjust_ref(int64_t x, int64_t off, int64_t len_kmers, int64_t len, int is_template, int forward) {
    if ((is_template && forward) || (!is_template && !forward)) return x + off;
    return len_kmers - (x + (len - off));
}

typedef struct {
    const char *label, *contig;
    const strand_model_t *sm;
    sa_strand_params_t npp;
    const double *events;   /* all events of the strand, 4 doubles each */
    const char *target;
    int forward, is_template, rna;
    int64_t event_offset, ref_offset;
    const sa_pair_t *pairs;
    int64_t n_pairs;
    double score;
} out_ctx_t;

/* Rows are put together in a line buffer and handed to stdio as bytes: "%f" through sa_format_f6 (the same characters as
 * printf, tests/test_host_ambig.py), integers and strings by hand.  With fprintf a row of the full format cost 0.8 us of
 * formatter on every one of 11 million rows of a 1000-read batch; rendering was the largest stage of the front door. */
static char *put_s(char *p, const char *s) {
    while (*s) *p++ = *s++;
    return p;
}
static char *put_i64(char *p, int64_t v) {
    char t[24];
    int n = 0;
    uint64_t u = v < 0 ? (uint64_t) 0 - (uint64_t) v : (uint64_t) v;
    if (v < 0) *p++ = '-';
    do { t[n++] = (char) ('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = t[--n];
    return p;
}
static char *put_f(char *p, double v) { return p + sa_format_f6(p, v); }
static void reverse_complement_into(const char *s, int k, char *out) {   /* sa_reverse_complement of a k-mer, no allocation */
    for (int i = 0; i < k; i++) {
        const char c = s[k - 1 - i];
        out[i] = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C'
               : c == 'a' ? 't' : c == 't' ? 'a' : c == 'c' ? 'g' : c == 'g' ? 'c' : c;
    }
    out[k] = 0;
}
static FILE *open_rows(const char *path) {
    FILE *fh = fopen(path, "a");
    if (!fh) die("signalMachine: cannot open output %s", path);
    setvbuf(fh, NULL, _IOFBF, 1 << 20);
    return fh;
}

/* writePosteriorProbsFull, impl/signalMachine.c:89-159 */
static void write_full(const char *path, const out_ctx_t *o) {
    FILE *fh = open_rows(path);
    const size_t fixed = strlen(o->contig) + strlen(o->label);
    char *line = (char *) malloc(fixed + 640);   /* nine "%f" of at most 320 characters... in theory; 24 for sane values */
    if (!line) die("signalMachine: out of memory%s", "");
    const int k = o->sm->k;
    int64_t ref_len = (int64_t) strlen(o->target), ref_len_kmers = ref_len - k;
    char k_i[16], path_kmer[16];
    for (int64_t i = 0; i < o->n_pairs; i++) {
        const sa_pair_t *p = &o->pairs[i];
        int64_t x_adj = adjust_ref(p->x, o->ref_offset, ref_len_kmers, ref_len, o->is_template, o->forward);
        int64_t y = p->y + o->event_offset;
        double prob = ((double) p->prob_e7) / PROB_1;
        double ev_mean = o->events[y * 4], ev_noise = o->events[y * 4 + 1], ev_dur = o->events[y * 4 + 2];
        memcpy(k_i, o->target + p->x, k);
        k_i[k] = 0;
        kmer_string(o->sm, p->kmer_id, path_kmer);
        double E_mean = o->sm->table[(int64_t) p->kmer_id * 5];
        /* emissions_signal_scaleNoise (impl/stateMachine.c:721-741) rescales the table per read; applied on the fly here */
        double E_noise = o->sm->table[(int64_t) p->kmer_id * 5 + 2] * o->npp.scale_sd;
        double scaled_Emean = E_mean * o->npp.scale + o->npp.shift;
        double scaled_Enoise = E_noise * o->npp.scale_sd;
        double descaled = descale(ev_mean, E_mean, o->npp.scale, o->npp.shift, o->npp.var);
        char ref_kmer[16], tmp[16];
        if ((o->is_template && o->forward) || (!o->is_template && !o->forward)) memcpy(ref_kmer, k_i, (size_t) k + 1);
        else reverse_complement_into(k_i, k, ref_kmer);
        if (o->rna) {
            reverse_complement_into(ref_kmer, k, tmp);
            memcpy(ref_kmer, tmp, (size_t) k + 1);
        }
        const double vals[9] = {ev_mean, ev_noise, ev_dur, scaled_Emean, scaled_Enoise, prob, descaled, E_mean, 0.0};
        int wide = 0;   /* a value the 24-character budget does not hold: the formatter's own path */
        for (int q = 0; q < 8; q++) wide |= !(fabs(vals[q]) < 9.0e15);
        if (wide) {
            fprintf(fh, "%s\t%" PRId64 "\t%s\t%s\t%s\t%" PRId64 "\t%f\t%f\t%f\t%s\t%f\t%f\t%f\t%f\t%f\t%s\n", o->contig, x_adj,
                    ref_kmer, o->label, o->is_template ? "t" : "c", y, ev_mean, ev_noise, ev_dur, k_i, scaled_Emean,
                    scaled_Enoise, prob, descaled, E_mean, path_kmer);
            continue;
        }
        char *w = line;
        w = put_s(w, o->contig); *w++ = '\t';
        w = put_i64(w, x_adj); *w++ = '\t';
        w = put_s(w, ref_kmer); *w++ = '\t';
        w = put_s(w, o->label); *w++ = '\t';
        *w++ = o->is_template ? 't' : 'c'; *w++ = '\t';
        w = put_i64(w, y); *w++ = '\t';
        w = put_f(w, ev_mean); *w++ = '\t';
        w = put_f(w, ev_noise); *w++ = '\t';
        w = put_f(w, ev_dur); *w++ = '\t';
        w = put_s(w, k_i); *w++ = '\t';
        w = put_f(w, scaled_Emean); *w++ = '\t';
        w = put_f(w, scaled_Enoise); *w++ = '\t';
        w = put_f(w, prob); *w++ = '\t';
        w = put_f(w, descaled); *w++ = '\t';
        w = put_f(w, E_mean); *w++ = '\t';
        w = put_s(w, path_kmer); *w++ = '\n';
        fwrite(line, 1, (size_t) (w - line), fh);
    }
    free(line);
    fclose(fh);
}

/* writePosteriorProbsVC, impl/signalMachine.c:161-232: only k-mers holding the internal ambiguity letter X */
static void write_vc(const char *path, const out_ctx_t *o) {
    int forward = o->forward;
    int label_forward = (o->rna || !o->is_template) ? !forward : forward;
    FILE *fh = open_rows(path);
    char *line = (char *) malloc(strlen(o->contig) + strlen(o->label) + 768);
    if (!line) die("signalMachine: out of memory%s", "");
    const int k = o->sm->k;
    int64_t ref_len = (int64_t) strlen(o->target), ref_len_kmers = ref_len - k;
    char k_i[16], path_kmer[16];
    for (int64_t i = 0; i < o->n_pairs; i++) {
        const sa_pair_t *p = &o->pairs[i];
        memcpy(k_i, o->target + p->x, k);
        k_i[k] = 0;
        int same = (o->is_template && forward) || (!o->is_template && !forward);
        char ref_kmer[16];
        if (same) memcpy(ref_kmer, k_i, (size_t) k + 1);
        else reverse_complement_into(k_i, k, ref_kmer);
        if (!strchr(ref_kmer, 'X')) continue;
        int64_t x_adj = adjust_ref(p->x, o->ref_offset, ref_len_kmers, ref_len, o->is_template, forward);
        int64_t y = p->y + o->event_offset;
        double prob = ((double) p->prob_e7) / PROB_1;
        kmer_string(o->sm, p->kmer_id, path_kmer);
        for (int q = 0; q < k; q++) {
            if (ref_kmer[q] != 'X') continue;
            int qp = same ? q : (k - 1) - q; /* adjustQueryPosition :81-87 */
            char *w = line;
            w = put_i64(w, y); *w++ = '\t';
            w = put_i64(w, x_adj + q); *w++ = '\t';
            *w++ = path_kmer[qp]; *w++ = '\t';
            w = put_f(w, prob); *w++ = '\t';
            *w++ = o->is_template ? 't' : 'c'; *w++ = '\t';
            w = put_s(w, label_forward ? "forward" : "backward"); *w++ = '\t';
            w = put_s(w, o->label); *w++ = '\t';
            w = put_f(w, o->score); *w++ = '\t';
            w = put_s(w, o->contig); *w++ = '\n';
            fwrite(line, 1, (size_t) (w - line), fh);
        }
    }
    free(line);
    fclose(fh);
}

/* writeAssignments, impl/signalMachine.c:234-270 */
static void write_assignments(const char *path, const out_ctx_t *o) {
    FILE *fh = open_rows(path);
    char path_kmer[16], line[704];
    for (int64_t i = 0; i < o->n_pairs; i++) {
        const sa_pair_t *p = &o->pairs[i];
        int64_t y = p->y + o->event_offset;
        double prob = ((double) p->prob_e7) / PROB_1;
        kmer_string(o->sm, p->kmer_id, path_kmer);
        double E_mean = o->sm->table[(int64_t) p->kmer_id * 5];
        double descaled = descale(o->events[y * 4], E_mean, o->npp.scale, o->npp.shift, o->npp.var);
        char *w = line;
        w = put_s(w, path_kmer); *w++ = '\t';
        *w++ = o->is_template ? 't' : 'c'; *w++ = '\t';
        w = put_f(w, descaled); *w++ = '\t';
        w = put_f(w, prob); *w++ = '\n';
        fwrite(line, 1, (size_t) (w - line), fh);
    }
    fclose(fh);
}

static void output_alignment(int64_t fmt, const char *f1, const char *f2, const out_ctx_t *o) {
    switch (fmt) {
        case 0: write_full(f1, o); break;
        case 1: write_vc(f1, o); break;
        case 2: write_assignments(f1, o); break;
        case 3: write_full(f1, o); write_vc(f2, o); break;
        default: fprintf(stderr, "signalAlign - No valid output format provided\n");
    }
}

/* continuousPairHmm_writeToFile (impl/continuousHmm.c:352-408) / hdpHmm_writeToFile (:572-623): the library's Hmm object
 * (sa_hmm_*), filled with what sa_expect_batch returned for the read.  `trans` already holds the transition pseudocount. */
static void write_expectations(const char *path, const strand_model_t *sm, const sa_strand_params_t *npp, int hdp, double threshold,
                               const double *trans, double lik, const sa_job_t *job, const sa_assignment_t *as,
                               int64_t n_as) {
    sa_hmm_t *h = NULL;
    int rc = sa_hmm_create(&h, sm->model, hdp ? SA_HMM_HDP : SA_HMM_GAUSSIAN, threshold, 0.0, 0.001 /* emissionsPseudocount, :785 */);
    if (rc != SA_OK) die("signalMachine: cannot build the expectations object: %s", sa_strerror(rc));
    int64_t n_kmers = 1;
    for (int i = 0; i < sm->k; i++) n_kmers *= sm->n_alpha;
    /* the event model is the state machine's table after this read's emissions_signal_scaleNoise
     * (impl/stateMachine.c:721-741: noise_mean *= scale_sd, noise_lambda *= var_sd, noise_sd = sqrt(mean^3 / lambda)) */
    double *em = malloc(sizeof(double) * 5 * (size_t) n_kmers);
    if (!em) die("signalMachine: out of memory%s", "");
    for (int64_t i = 0; i < n_kmers * 5; i += 5) {
        const double nm = sm->table[i + 2] * npp->scale_sd, nl = sm->table[i + 4] * npp->var_sd;
        em[i] = sm->table[i]; em[i + 1] = sm->table[i + 1]; em[i + 2] = nm; em[i + 3] = sqrt(pow(nm, 3.0) / nl); em[i + 4] = nl;
    }
    sa_hmm_set_event_model(h, em);
    free(em);
    sa_hmm_add_expectations(h, trans, lik);
    for (int64_t i = 0; hdp && i < n_as; i++)
        sa_hmm_add_assignment(h, job->ref + as[i].ref_pos, job->events[as[i].event * job->event_stride]);
    rc = sa_hmm_write(h, path);
    sa_hmm_destroy(h);
    if (rc != SA_OK) die("signalMachine: cannot open %s for writing", path);
}

static int load_strand_model(strand_model_t *sm, const char *model_path, const char *nhdp_path) {
    int rc = sa_model_load(&sm->model, model_path, nhdp_path);
    if (rc) return rc;
    sa_model_alphabet(sm->model, sm->alphabet, &sm->n_alpha, &sm->k);
    int64_t n = 5;
    for (int i = 0; i < sm->k; i++) n *= sm->n_alpha;
    sm->table = malloc(sizeof(double) * (size_t) n);
    sm->table_orig = malloc(sizeof(double) * (size_t) n);
    memcpy(sm->table, sa_model_table5(sm->model), sizeof(double) * (size_t) n);
    memcpy(sm->table_orig, sm->table, sizeof(double) * (size_t) n);
    return SA_OK;
}

/* options shared by every read of a run */
typedef struct {
    int hdp, two_d, rna, expect_mode;
    int mea; /* --mea: also write the maximum-expected-accuracy path of every read (not in the reference binary) */
    int two_dist; /* --emission twoDist: the two-distribution emission (not an option of the reference binary: it is what its
                   * state machine carried when the reference's shipped output files were written); one read per process */
    int64_t out_fmt, constraint_trim;
    const char *fwd_ref, *bwd_ref;
    sa_params_t p;
    strand_model_t smt, smc;
    const char *ambig[256];
} run_t;

/* one read: its inputs, the two alignment jobs, where its outputs go */
typedef struct {
    char *label, *npread_path, *cigar_path, *post_path, *post_path2, *seq_name, *t_expect, *c_expect;
    sa_cigar_t *pA;
    sa_npread_t *np;
    char *forward_seq, *backward_seq;
    const char *template_target, *complement_target;
    int64_t t_lo, t_hi, c_lo, c_hi, r_shift_t, r_shift_c, n_guide;
    int forward;
    int64_t *ax[2], *ay[2];
    sa_job_t jobs[2];
    sa_model_t *model[2]; /* --emission twoDist: the strand models with this read's noise scaling */
    int failed;
    char err[512];
} read_t;

/* single-read mode keeps the reference's abort-with-message behaviour; in batch mode a bad read is reported and skipped */
static int fail(read_t *rd, int fatal, const char *fmt, const char *a) {
    if (fatal) die(fmt, a);
    snprintf(rd->err, sizeof(rd->err), fmt, a ? a : "");
    rd->failed = 1;
    return -1;
}

static int estimate_strand(const strand_model_t *sm, const int64_t *strand_map, double *events, int64_t n_events,
                           const char *read, int64_t read_len, sa_strand_params_t *out, sa_model_t **read_model) {
    int64_t n = 5;
    for (int i = 0; i < sm->k; i++) n *= sm->n_alpha;
    double *scratch = malloc(sizeof(double) * (size_t) n); /* the estimation rescales the noise columns in place */
    if (!scratch) return SA_ENOMEM;
    memcpy(scratch, sm->table_orig, sizeof(double) * (size_t) n);
    double est[7];
    int rc = sa_estimate_params(sm->model, scratch, strand_map, events, n_events, read, read_len, est);
    if (rc == SA_OK && read_model) {   /* the model this read is aligned with: the rescaled noise columns, two-distribution emission */
        rc = sa_model_clone_with_table(read_model, sm->model, scratch);
        if (rc == SA_OK) rc = sa_model_set_emission(*read_model, SA_EMISSION_TWO_DIST);
    }
    free(scratch);
    if (rc != SA_OK) return rc;
    out->scale = est[0]; out->shift = est[1]; out->var = est[2]; out->drift = est[3];
    out->scale_sd = est[4]; out->var_sd = est[5]; out->shift_sd = est[6];
    return SA_OK;
}

/* everything of impl/signalMachine.c:main between option parsing and performSignalAlignment, for one read */
/* SA_CLI_TIMING=1: wall time of the three stages of every slice and the thread-seconds spent inside the host stage, printed
 * at the end of the run (probes/batch_cli_timing.sh; INTEGRATION.md quotes them) */
static double g_t_prep, g_t_gpu, g_t_render;                 /* wall seconds, summed over slices */
static double g_ts_parse, g_ts_fetch, g_ts_estimate;         /* thread-seconds inside the host stage */
static pthread_mutex_t g_t_mu = PTHREAD_MUTEX_INITIALIZER;
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}
static void t_add(double *acc, double dt) {
    pthread_mutex_lock(&g_t_mu);
    *acc += dt;
    pthread_mutex_unlock(&g_t_mu);
}

static int validate_read(const run_t *R, read_t *rd);
static int prepare_read(const run_t *R, read_t *rd, int fatal) {
    const double tp0 = now_s();
    if (rd->cigar_path == NULL) return fail(rd, fatal, "[signalMachine]ERROR: Need to provide input guide alignments, exiting", NULL);
    if (sa_cigar_load(rd->cigar_path, &rd->pA) != SA_OK)
        return fail(rd, fatal, "[signalMachine]ERROR: Didn't find input alignment file, looked %s", rd->cigar_path);
    fprintf(stderr, "[signalMachine]NOTICE: Using guide alignments from %s\n", rd->cigar_path);
    sa_cigar_t *pA = rd->pA;
    if (rd->npread_path == NULL || sa_npread_load(rd->npread_path, &rd->np) != SA_OK)
        return fail(rd, fatal, "signalMachine: could not load the nanopore read %s", rd->npread_path);
    sa_npread_t *np = rd->np;
    if (pA->start2 < 0 || pA->end2 <= pA->start2 || pA->end2 > (R->two_d ? np->read_length : np->template_read_length))
        return fail(rd, fatal, "signalMachine: guide alignment of %s does not fit the read", rd->label);
    if (R->rna) {
        int64_t tmp = pA->start2;
        pA->start2 = np->template_read_length - pA->end2;
        pA->end2 = np->template_read_length - tmp;
    }
    const double tp1 = now_s();
    t_add(&g_ts_parse, tp1 - tp0);
    const char *seq_name = rd->seq_name ? rd->seq_name : pA->contig1;
    if (R->fwd_ref == NULL || seq_name == NULL)
        return fail(rd, fatal, "[signalMachine] ERROR: need -f <fasta> and -n <sequence name>", NULL);

    /* fastaHandler_ReferenceSequenceConstructFull, impl/fasta_handler.c:47-102 */
    if (R->rna) { /* listReverse(pA->operationList) */
        for (int64_t i = 0, j = pA->n_ops - 1; i < j; i++, j--) {
            int32_t t = pA->op_type[i]; pA->op_type[i] = pA->op_type[j]; pA->op_type[j] = t;
            int64_t l = pA->op_len[i]; pA->op_len[i] = pA->op_len[j]; pA->op_len[j] = l;
        }
    }
    int ferr = 0;
    rd->forward_seq = pA->strand1 ? sa_fasta_fetch(R->fwd_ref, seq_name, pA->start1, pA->end1 - 1, &ferr)
                                  : sa_fasta_fetch(R->fwd_ref, seq_name, pA->end1, pA->start1 - 1, &ferr);
    if (ferr == -2) {
        fprintf(stderr, "[signalMachine] ERROR %d: sequence name: %s is not in reference fasta: %s \n", ferr, seq_name, R->fwd_ref);
        if (fatal) exit(1);
        return fail(rd, 0, "sequence name %s is not in the reference fasta", seq_name);
    }
    if (rd->forward_seq == NULL) return fail(rd, fatal, "[signalMachine] ERROR: Unable to fetch reference sequence.  ", NULL);
    if (R->bwd_ref) {
        rd->backward_seq = pA->strand1 ? sa_fasta_fetch(R->bwd_ref, seq_name, pA->start1, pA->end1 - 1, &ferr)
                                       : sa_fasta_fetch(R->bwd_ref, seq_name, pA->end1, pA->start1 - 1, &ferr);
        if (rd->backward_seq == NULL) return fail(rd, fatal, "[signalMachine] ERROR: Unable to fetch reference sequence.  ", NULL);
        sa_reverse_in_place(rd->backward_seq);
    } else {
        rd->backward_seq = sa_complement(rd->forward_seq);
        sa_reverse_in_place(rd->backward_seq);
    }
    int strand1 = pA->strand1;
    if (R->rna) {
        char *tmp = rd->backward_seq;
        rd->backward_seq = strdup(rd->forward_seq);
        sa_reverse_in_place(rd->backward_seq);
        free(rd->forward_seq);
        rd->forward_seq = tmp;
        sa_reverse_in_place(rd->forward_seq);
        int64_t t2 = pA->start1;
        pA->start1 = pA->end1;
        pA->end1 = t2;
        pA->strand1 = !pA->strand1;
        strand1 = pA->strand1;
    }
    rd->template_target = strand1 ? rd->forward_seq : rd->backward_seq;
    rd->complement_target = strand1 ? rd->backward_seq : rd->forward_seq;

    /* event slices and coordinate shifts (impl/signalMachine.c:726-750) */
    const int64_t *t_map = R->two_d ? np->template_event_map : np->template_strand_event_map;
    rd->t_lo = t_map[pA->start2];
    rd->t_hi = t_map[pA->end2 - 1];
    if (R->two_d) { rd->c_lo = np->complement_event_map[pA->start2]; rd->c_hi = np->complement_event_map[pA->end2 - 1]; }
    rd->r_shift_t = pA->start1;
    rd->r_shift_c = R->two_d ? pA->end1 : 0;
    rd->forward = pA->strand1;

    const double tp2 = now_s();
    t_add(&g_ts_fetch, tp2 - tp1);
    /* anchors from the guide alignment (pA is rebased inside, impl/signalMachineUtils.c:142-164) */
    int64_t cap = 0;
    for (int64_t i = 0; i < pA->n_ops; i++) cap += pA->op_len[i];
    int64_t *gx = malloc(sizeof(int64_t) * (size_t) (cap + 1)), *gy = malloc(sizeof(int64_t) * (size_t) (cap + 1));
    rd->n_guide = sa_guide_to_anchors(pA->start1, pA->end1, pA->strand1, pA->start2, pA->op_type, pA->op_len, pA->n_ops,
                                      R->constraint_trim, gx, gy, cap + 1);
    if (rd->n_guide < 0) { free(gx); free(gy); return fail(rd, fatal, "signalMachine: could not convert the guide alignment", NULL); }

    /* per-strand: estimate the read's parameters (signalUtils_estimateNanoporeParams), build the job */
    if (estimate_strand(&R->smt, np->template_strand_event_map, np->template_events, np->n_template_events,
                        np->template_read, np->template_read_length, &np->template_params,
                        R->two_dist ? &rd->model[0] : NULL) != SA_OK) {
        free(gx); free(gy);
        return fail(rd, fatal, "Cannot get scale params with no assignments", NULL);
    }
    memset(rd->jobs, 0, sizeof(rd->jobs));
    rd->ax[0] = malloc(sizeof(int64_t) * (size_t) (rd->n_guide + 1));
    rd->ay[0] = malloc(sizeof(int64_t) * (size_t) (rd->n_guide + 1));
    int64_t na0 = sa_remap_anchors(gx, gy, rd->n_guide, t_map, pA->start2, rd->ax[0], rd->ay[0]);
    rd->jobs[0].ref = rd->template_target;
    rd->jobs[0].ref_len = (int64_t) strlen(rd->template_target);
    rd->jobs[0].events = np->template_events + 4 * rd->t_lo;
    rd->jobs[0].event_stride = 4;
    rd->jobs[0].n_events = rd->t_hi - rd->t_lo;
    rd->jobs[0].anchor_x = rd->ax[0]; rd->jobs[0].anchor_y = rd->ay[0]; rd->jobs[0].n_anchors = na0;
    rd->jobs[0].scale = np->template_params.scale; rd->jobs[0].shift = np->template_params.shift;
    rd->jobs[0].var = np->template_params.var;
    if (R->two_d) {
        if (estimate_strand(&R->smc, np->complement_strand_event_map, np->complement_events, np->n_complement_events,
                            np->complement_read, np->complement_read_length, &np->complement_params,
                            R->two_dist ? &rd->model[1] : NULL) != SA_OK) {
            free(gx); free(gy);
            return fail(rd, fatal, "Cannot get scale params with no assignments", NULL);
        }
        rd->ax[1] = malloc(sizeof(int64_t) * (size_t) (rd->n_guide + 1));
        rd->ay[1] = malloc(sizeof(int64_t) * (size_t) (rd->n_guide + 1));
        int64_t na1 = sa_remap_anchors(gx, gy, rd->n_guide, np->complement_event_map, pA->start2, rd->ax[1], rd->ay[1]);
        rd->jobs[1].ref = rd->complement_target;
        rd->jobs[1].ref_len = (int64_t) strlen(rd->complement_target);
        rd->jobs[1].events = np->complement_events + 4 * rd->c_lo;
        rd->jobs[1].event_stride = 4;
        rd->jobs[1].n_events = rd->c_hi - rd->c_lo;
        rd->jobs[1].anchor_x = rd->ax[1]; rd->jobs[1].anchor_y = rd->ay[1]; rd->jobs[1].n_anchors = na1;
        rd->jobs[1].scale = np->complement_params.scale; rd->jobs[1].shift = np->complement_params.shift;
        rd->jobs[1].var = np->complement_params.var;
    }
    free(gx);
    free(gy);
    t_add(&g_ts_estimate, now_s() - tp2);
    /* batch mode: the reads of a slice share one GPU batch, and the planner rejects a whole batch for one bad job (a
     * reference window with a letter outside the alphabet, anchors that give an invalid diagonal).  The reference runs one
     * process per read, so only that read may fail.  Alignment runs find the offender when -- and only when -- a batch is
     * turned down (validate_reads below); the expectation routine writes files strand by strand and cannot be re-run, so
     * its jobs are planned alone on the host here (integer geometry only, no GPU). */
    if (!fatal && R->expect_mode) return validate_read(R, rd);
    return 0;
}

/* plans every job of the read alone on the host; marks the read failed when the planner rejects one */
static int validate_read(const run_t *R, read_t *rd) {
    for (int s = 0; s < (R->two_d ? 2 : 1); s++) {
        int rc = sa_plan_describe(s == 0 ? R->smt.model : R->smc.model, &R->p, &rd->jobs[s], R->ambig, 0, NULL, NULL, 0, NULL, 0,
                                  NULL, 0);
        if (rc != SA_OK) return fail(rd, 0, "alignment job rejected: %s", sa_strerror(rc));
    }
    return 0;
}

typedef struct { const run_t *R; read_t *reads; const int64_t *who; } validate_ctx_t;
static void validate_one(int64_t j, void *ctx) {
    validate_ctx_t *v = ctx;
    read_t *rd = &v->reads[v->who[j]];
    if (validate_read(v->R, rd) != 0)
        fprintf(stderr, "[signalMachine] ERROR: read %s skipped: %s\n", rd->label, rd->err);
}

static void set_hdp_expected(strand_model_t *sm) { /* stateMachine3_setModelToHdpExpectedValues, once per run */
    sa_model_set_to_hdp_expected_values(sm->model);
    int64_t n = 5;
    for (int i = 0; i < sm->k; i++) n *= sm->n_alpha;
    const double *mt = sa_model_table5(sm->model);
    for (int64_t i = 0; i < n; i += 5) { sm->table[i] = mt[i]; sm->table[i + 1] = mt[i + 1]; }
}

static char *dup_field(const char *s) { return (s == NULL || s[0] == 0 || strcmp(s, "-") == 0) ? NULL : strdup(s); }

/* manifest of --batch: label, npRead, cigar, posteriors [, posteriors2, sequence_name, template expectations, complement expectations] */
static int64_t load_manifest(const char *path, read_t **out) {
    FILE *fh = fopen(path, "r");
    if (!fh) return -1;
    int64_t n = 0, cap = 0;
    read_t *reads = NULL;
    char *line = NULL;
    size_t lcap = 0;
    while (getline(&line, &lcap, fh) >= 0) {
        size_t len = strlen(line);
        while (len && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        if (len == 0 || line[0] == '#') continue;
        char *f[8] = {0};
        int nf = 0;
        for (char *tok = line; tok && nf < 8;) {
            char *tab = strchr(tok, '\t');
            if (tab) *tab = 0;
            f[nf++] = tok;
            tok = tab ? tab + 1 : NULL;
        }
        if (nf < 4) { fprintf(stderr, "[signalMachine] batch manifest: line with %d fields ignored (need at least 4)\n", nf); continue; }
        if (n == cap) {
            cap = cap ? cap * 2 : 64;
            reads = realloc(reads, sizeof(read_t) * (size_t) cap);
        }
        read_t *rd = &reads[n++];
        memset(rd, 0, sizeof(*rd));
        rd->label = strdup(f[0]);
        rd->npread_path = dup_field(f[1]);
        rd->cigar_path = dup_field(f[2]);
        rd->post_path = dup_field(f[3]);
        rd->post_path2 = dup_field(f[4]);
        rd->seq_name = dup_field(f[5]);
        rd->t_expect = dup_field(f[6]);
        rd->c_expect = dup_field(f[7]);
    }
    free(line);
    fclose(fh);
    *out = reads;
    return n;
}

/* host work per read (parsing, parameter estimation, TSV rendering) is independent: a small pthread parallel-for */
typedef struct {
    void (*fn)(int64_t i, void *ctx);
    void *ctx;
    int64_t n;
    int64_t next;
    pthread_mutex_t mu;
} pfor_t;

static void *pfor_worker(void *arg) {
    pfor_t *pf = arg;
    for (;;) {
        pthread_mutex_lock(&pf->mu);
        int64_t i = pf->next++;
        pthread_mutex_unlock(&pf->mu);
        if (i >= pf->n) return NULL;
        pf->fn(i, pf->ctx);
    }
}

static void parallel_for(int64_t n, void (*fn)(int64_t, void *), void *ctx) {
    const char *e = getenv("SA_HOST_THREADS");
    long t = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    if (!e) {   /* a container's CPU quota (cgroup v2 cpu.max): more runnable threads than that only get the whole group throttled */
        static long quota = -1;
        if (quota < 0) {
            quota = 0;
            FILE *fq = fopen("/sys/fs/cgroup/cpu.max", "r");
            if (fq) {
                char a[32];
                long per = 0;
                if (fscanf(fq, "%31s %ld", a, &per) == 2 && strcmp(a, "max") != 0 && per > 0) quota = (atol(a) + per - 1) / per;
                fclose(fq);
            }
        }
        if (quota > 0 && t > quota) t = quota;
    }
    if (t > 32) t = 32;
    if (t > n) t = n;
    if (t <= 1) {
        for (int64_t i = 0; i < n; i++) fn(i, ctx);
        return;
    }
    pfor_t pf = {fn, ctx, n, 0, PTHREAD_MUTEX_INITIALIZER};
    pthread_t th[32];
    int started[32];
    for (long k = 0; k < t; k++) started[k] = pthread_create(&th[k], NULL, pfor_worker, &pf) == 0;
    if (!started[0]) pfor_worker(&pf); /* no threads at all: do the work here */
    for (long k = 0; k < t; k++)
        if (started[k]) pthread_join(th[k], NULL);
}

typedef struct {
    const run_t *R;
    read_t *reads;
    int fatal;
} prep_ctx_t;

static void prep_one(int64_t i, void *ctx) {
    prep_ctx_t *c = ctx;
    if (prepare_read(c->R, &c->reads[i], c->fatal) != 0)
        fprintf(stderr, "[signalMachine] ERROR: read %s skipped: %s\n", c->reads[i].label, c->reads[i].err);
}

typedef struct {
    const run_t *R;
    read_t *reads;
    const int64_t *who;
    sa_pair_t ***pairs;   /* [strand][job] */
    int64_t **n_pairs;
    double (*score)[2];
    sa_mea_pair_t ***mea; /* [strand][job], --mea only */
    int64_t **n_mea;
    sa_batch_t *const *batch;   /* [strand]: the batch is still alive and pairs[strand][job] is NULL -- a job's rows are expanded
                                 * from the batch's packed records (sa_batch_pairs16) by the thread that renders the job */
    int64_t *const *all_n;      /* [strand][job], -s 1 only (SA_FLAG_VC_ROWS): number and prob_e7 sum of ALL pairs of the job -- the */
    int64_t *const *all_sum;    /* rows the variant-caller output does not print were dropped on the device                        */
    const int *p8;              /* [strand]: the batch holds 8-byte records (SA_FLAG_PAIRS8): path 0, the reference's k-mer at x          */
} out_job_t;

/* kmer_id of the k letters at s (sorted alphabet, first letter most significant), -1 for a letter outside it */
static int32_t kmer_id_of(const strand_model_t *sm, const char *s) {
    int32_t id = 0;
    for (int i = 0; i < sm->k; i++) {
        const char *q = memchr(sm->alphabet, s[i], (size_t) sm->n_alpha);
        if (!q || !s[i]) return -1;
        id = id * sm->n_alpha + (int32_t) (q - sm->alphabet);
    }
    return id;
}

/* The rows of the full output that lie on the maximum-expected-accuracy path -- what mea_alignment_from_signal_align
 * (src/signalalign/mea_algorithm.py:323-341) returns as its final event table, here as a TSV next to the posteriors file.
 * The path holds one (x, y) per event; where several rows share a cell (ambiguous positions) the row with the lowest
 * posterior is the one the reference's matrix keeps (get_mea_params_from_events :305-318). */
static void write_mea(const char *post_path, const out_ctx_t *o, const sa_mea_pair_t *path, int64_t n_path) {
    char *out_path = malloc(strlen(post_path) + 8);
    sprintf(out_path, "%s.mea", post_path);
    int64_t y_max = -1;
    for (int64_t i = 0; i < n_path; i++) y_max = path[i].event_idx > y_max ? path[i].event_idx : y_max;
    int64_t *x_of = malloc(sizeof(int64_t) * (size_t) (y_max + 2)), *best = malloc(sizeof(int64_t) * (size_t) (y_max + 2));
    for (int64_t y = 0; y <= y_max; y++) { x_of[y] = -1; best[y] = -1; }
    for (int64_t i = 0; i < n_path; i++) x_of[path[i].event_idx] = path[i].ref_idx;
    for (int64_t i = 0; i < o->n_pairs; i++) {
        const sa_pair_t *p = &o->pairs[i];
        if (p->y > y_max || x_of[p->y] != p->x) continue;
        if (best[p->y] < 0 || p->prob_e7 < o->pairs[best[p->y]].prob_e7) best[p->y] = i;
    }
    sa_pair_t *rows = malloc(sizeof(sa_pair_t) * (size_t) (n_path > 0 ? n_path : 1));
    int64_t n = 0;
    for (int64_t i = 0; i < o->n_pairs; i++) {   /* output order of the posteriors file */
        const sa_pair_t *p = &o->pairs[i];
        if (p->y <= y_max && best[p->y] == i) rows[n++] = *p;
    }
    out_ctx_t m = *o;
    m.pairs = rows;
    m.n_pairs = n;
    write_full(out_path, &m);
    free(rows); free(x_of); free(best); free(out_path);
}

static void output_one(int64_t j, void *ctx) {
    out_job_t *c = ctx;
    const run_t *R = c->R;
    read_t *rd = &c->reads[c->who[j]];
    const int n_strands = R->two_d ? 2 : 1;
    if (R->out_fmt == 3 && rd->post_path2 == NULL) {
        fprintf(stderr, "[signalMachine] ERROR: read %s: 'both' output format needs a second output file\n", rd->label);
        rd->failed = 1;
        return;
    }
    /* Round 4: the GPU stage no longer expands every job's pairs into freshly allocated sa_pair_t arrays on the main thread (523 MB
     * per slice of 2048 long reads); the rows are expanded here, job by job on the rendering threads, from the packed 16-byte
     * records the batch holds in page-locked memory.  (Measured: no difference in wall time -- 2.42-2.55 s against 2.49 s per 6144
     * long reads, SA_CLI_EXPAND_EARLY=1 -- the front door is bound by the CPU time of parsing and rendering, not by this.) */
    sa_pair_t *mine[2] = {NULL, NULL};
    const sa_pair_t *pp[2] = {NULL, NULL};
    for (int s = 0; s < n_strands; s++) {
        pp[s] = c->pairs[s][j];
        if (pp[s] == NULL && c->batch && c->batch[s] && c->p8 && c->p8[s]) {
            /* 8-byte records (round 6: -s 0 / 2 on reads without ambiguity letters -- half the bytes over PCIe where the pairs outweigh
             * the kernels, e.g. --sm3Hdp -D 0.01): x, y, probability; the pair's k-mer is the reference's at x, its path 0 */
            const sa_pair8_t *pk8 = NULL;
            int64_t n = 0;
            const strand_model_t *sm = s == 0 ? &R->smt : &R->smc;
            const char *target = s == 0 ? rd->template_target : rd->complement_target;
            if (sa_batch_pairs8(c->batch[s], j, &pk8, &n) != SA_OK || n != c->n_pairs[s][j]) {
                fprintf(stderr, "[signalMachine] ERROR: read %s: results of the batch are not readable\n", rd->label);
                rd->failed = 1;
                free(mine[0]);
                return;
            }
            mine[s] = malloc(sizeof(sa_pair_t) * (size_t) (n > 0 ? n : 1));
            int bad_kmer = 0;
            for (int64_t i = 0; i < n; i++) {
                sa_pair_t *q = &mine[s][i];
                sa_pair8_unpack(pk8[i], &q->prob_e7, &q->x, &q->y);
                q->path = 0;
                q->kmer_id = kmer_id_of(sm, target + q->x);
                bad_kmer |= q->kmer_id < 0;
            }
            if (bad_kmer) {   /* (cannot happen: the planner refuses a reference with a letter outside the model's alphabet) */
                fprintf(stderr, "[signalMachine] ERROR: read %s: a pair names a k-mer outside the model's alphabet\n", rd->label);
                rd->failed = 1;
                free(mine[0]); free(mine[1]);
                return;
            }
            pp[s] = mine[s];
        } else if (pp[s] == NULL && c->batch && c->batch[s]) {
            const sa_pair16_t *pk = NULL;
            int64_t n = 0;
            if (sa_batch_pairs16(c->batch[s], j, &pk, &n) != SA_OK || n != c->n_pairs[s][j]) {
                fprintf(stderr, "[signalMachine] ERROR: read %s: results of the batch are not readable\n", rd->label);
                rd->failed = 1;
                free(mine[0]);
                return;
            }
            mine[s] = malloc(sizeof(sa_pair_t) * (size_t) (n > 0 ? n : 1));
            for (int64_t i = 0; i < n; i++) mine[s][i] = sa_pair16_unpack(pk[i]);
            pp[s] = mine[s];
        }
    }
    for (int s = 0; s < n_strands; s++) {
        double tot = 0.0;
        int64_t n_all = c->n_pairs[s][j];
        if (c->all_n && c->all_n[s]) {   /* (prob_e7 sums are integers below 2^53: the same double as the loop below gives) */
            n_all = c->all_n[s][j];
            tot = (double) c->all_sum[s][j];
        } else {
            for (int64_t i = 0; i < c->n_pairs[s][j]; i++) tot += (double) pp[s][i].prob_e7;
        }
        c->score[j][s] = 100.0 * tot / ((double) n_all * PROB_1); /* scoreByPosteriorProbabilityIgnoringGaps :407-412 */
    }
    if (rd->post_path != NULL) {
        out_ctx_t o;
        o.label = rd->label; o.contig = rd->pA->contig1; o.sm = &R->smt; o.npp = rd->np->template_params;
        o.events = rd->np->template_events; o.target = rd->template_target; o.forward = rd->forward; o.is_template = 1;
        o.rna = R->rna; o.event_offset = rd->t_lo; o.ref_offset = rd->r_shift_t; o.pairs = pp[0];
        o.n_pairs = c->n_pairs[0][j]; o.score = c->score[j][0];
        output_alignment(R->out_fmt, rd->post_path, rd->post_path2, &o);
        if (R->mea) write_mea(rd->post_path, &o, c->mea[0][j], c->n_mea[0][j]);
        if (R->two_d) {
            o.sm = &R->smc; o.npp = rd->np->complement_params; o.events = rd->np->complement_events;
            o.target = rd->complement_target; o.is_template = 0; o.event_offset = rd->c_lo; o.ref_offset = rd->r_shift_c;
            o.pairs = pp[1]; o.n_pairs = c->n_pairs[1][j]; o.score = c->score[j][1];
            output_alignment(R->out_fmt, rd->post_path, rd->post_path2, &o);
            if (R->mea) write_mea(rd->post_path, &o, c->mea[1][j], c->n_mea[1][j]);
        }
    }
    free(mine[0]); free(mine[1]);
}

static int cmp_str(const void *a, const void *b) { return strcmp(*(const char *const *) a, *(const char *const *) b); }

/* appending from several threads is only safe when no two reads share an output file */
static int outputs_distinct(const read_t *reads, const int64_t *who, int64_t n) {
    const char **v = malloc(sizeof(char *) * (size_t) (2 * n + 1));
    int64_t m = 0;
    for (int64_t j = 0; j < n; j++) {
        if (reads[who[j]].post_path) v[m++] = reads[who[j]].post_path;
        if (reads[who[j]].post_path2) v[m++] = reads[who[j]].post_path2;
    }
    qsort(v, (size_t) m, sizeof(char *), cmp_str);
    int ok = 1;
    for (int64_t i = 1; i < m && ok; i++) ok = strcmp(v[i], v[i - 1]) != 0;
    free(v);
    return ok;
}

/* One slice of the run's reads: host side of every read, one GPU batch per strand model, outputs.  Returns the number
 * of reads that failed.  (The whole manifest used to be one batch: fine for thousands of reads, not for a flow cell.) */
/* host side of every read of a slice (files, parameter estimation, anchors): all host threads */
typedef struct { const run_t *R; read_t *reads; int64_t n; int batch_mode; } slice_prep_t;
static void *slice_prepare(void *arg) {
    slice_prep_t *sp = arg;
    const double ts0 = now_s();
    prep_ctx_t pc = {sp->R, sp->reads, !sp->batch_mode};
    parallel_for(sp->n, prep_one, &pc);
    t_add(&g_t_prep, now_s() - ts0);
    return NULL;
}

/* What the GPU stage of a slice leaves for its rendering: the outputs of slice k are written (on a thread of their own) while
 * the GPU stage of slice k + 1 runs -- 30 000 long reads: GPU stage 3.8 s, rendering 3.9 s, one after the other before. */
static void release_read(read_t *rd);
typedef struct {
    run_t *Rp;
    read_t *reads;
    int64_t n_reads, n_ok;
    int64_t *who;
    sa_job_t *bj;
    sa_pair_t **pairs_s[2];
    int64_t *n_pairs_s[2];
    sa_mea_pair_t **mea_s[2];
    int64_t *n_mea_s[2];
    sa_batch_t *batch[2];  /* alive until the slice is rendered (their packed records are what the rendering reads) */
    int64_t *all_n_s[2], *all_sum_s[2];   /* -s 1: see out_job_t */
    int p8_s[2];          /* see out_job_t */
    int64_t n_failed;     /* out */
} render_job_t;
static void *render_slice(void *arg);

/* GPU stage of a slice; returns the rendering job (NULL: nothing left to render -- the expectations mode writes its files
 * here -- with the number of failed reads in *n_failed_now) */
static render_job_t *run_slice(run_t *Rp, read_t *reads, int64_t n_reads, int batch_mode, int device, int64_t *n_failed_now) {
#define R (*Rp)
    *n_failed_now = 0;
    /* (the host side of the slice's reads has run: slice_prepare, a slice ahead of this function) */
    int64_t n_ok = 0;
    for (int64_t i = 0; i < n_reads; i++) n_ok += reads[i].failed ? 0 : 1;
    const double ts1 = now_s();
    const strand_model_t *sms[2] = {&R.smt, &R.smc};
    const int n_strands = R.two_d ? 2 : 1;

    /* ---- the pair-HMM on the GPU: one batch per strand model, all reads side by side ---- */
    sa_job_t *bj = malloc(sizeof(sa_job_t) * (size_t) (n_ok > 0 ? n_ok : 1));
    int64_t *who = malloc(sizeof(int64_t) * (size_t) (n_ok > 0 ? n_ok : 1));
    int64_t k = 0;
    for (int64_t i = 0; i < n_reads; i++)
        if (!reads[i].failed) who[k++] = i;

    if (R.expect_mode) { /* impl/signalMachine.c:772-848 */
        if (n_ok > 0) fprintf(stderr, "Starting expectations routine\n");
        for (int s = 0; s < n_strands && n_ok > 0; s++) {
            fprintf(stderr, "signalAlign - getting expectations for %s\n", s == 0 ? "template" : "complement");
            for (int64_t j = 0; j < n_ok; j++) bj[j] = reads[who[j]].jobs[s];
            double *trans = malloc(sizeof(double) * 9 * (size_t) n_ok), *lik = calloc((size_t) n_ok, sizeof(double));
            for (int64_t j = 0; j < 9 * n_ok; j++) trans[j] = 0.001; /* transitionsPseudocount, :785 */
            sa_assignment_t **as = calloc((size_t) n_ok, sizeof(*as));
            int64_t *n_as = calloc((size_t) n_ok, sizeof(int64_t));
            int rc = sa_expect_batch(sms[s]->model, &R.p, bj, n_ok, R.ambig, device, 0, trans, lik, as, n_as);
            if (rc != SA_OK) {
                fprintf(stderr, "signalMachine: expectations failed: %s\n", sa_strerror(rc));
                exit(1);
            }
            for (int64_t j = 0; j < n_ok; j++) {
                read_t *rd = &reads[who[j]];
                const char *path = s == 0 ? rd->t_expect : rd->c_expect;
                if (R.hdp)
                    fprintf(stderr, s == 0 ? "signalAlign - got %" PRId64 " template HDP assignments\n"
                                           : "signalAlign - got %" PRId64 "complement HDP assignments\n", n_as[j]);
                if (path != NULL) {
                    fprintf(stderr, "signalAlign - writing expectations to file: %s\n", path);
                    write_expectations(path, sms[s], s == 0 ? &rd->np->template_params : &rd->np->complement_params, R.hdp,
                                       R.p.threshold, trans + 9 * j, lik[j], &rd->jobs[s], as[j], n_as[j]);
                }
                sa_free(as[j]);
            }
            free(trans); free(lik); free(as); free(n_as);
        }
        for (int64_t j = 0; j < n_ok; j++)
            fprintf(stderr, "signalAlign - SUCCESS: finished alignment of query %s, exiting\n", reads[who[j]].label);
        free(bj); free(who);
        *n_failed_now = n_reads - n_ok;
        for (int64_t i = 0; i < n_reads; i++) release_read(&reads[i]);
        return NULL;
    }

    sa_pair_t **pairs_s[2] = {NULL, NULL};
    int64_t *n_pairs_s[2] = {NULL, NULL};
    sa_pair_t ***pairs = pairs_s;
    int64_t **n_pairs = n_pairs_s;
    sa_mea_pair_t **mea_s[2] = {NULL, NULL};
    int64_t *n_mea_s[2] = {NULL, NULL};
    sa_mea_pair_t ***mea = mea_s;
    int64_t **n_mea = n_mea_s;
    int validated = !batch_mode;   /* a single-read run has nobody to isolate a bad job from */
    sa_batch_t *batches[2] = {NULL, NULL};
    int64_t *all_n[2] = {NULL, NULL}, *all_sum[2] = {NULL, NULL};
    /* -s 1 prints only the rows whose reference k-mer holds an X: the others stay on the device (SA_FLAG_VC_ROWS), the run's pair
     * count and score come from the totals the device kept */
    const unsigned vc_flag = (R.out_fmt == 1 && !R.mea && !getenv("SA_CLI_EXPAND_EARLY") && !getenv("SA_CLI_VC_ON_HOST")) ? SA_FLAG_VC_ROWS : 0u;
    /* -s 0 / -s 2 without --mea: 8-byte result records where the batch allows them (one path per cell: no ambiguity letter in any
     * read's reference; fewer than 2^20 positions and events per read) -- the planner says SA_EUNSUPPORTED otherwise and the strand's
     * batch is made again with 16-byte records.  SA_CLI_PAIRS16=1: always 16-byte records (the test's checker). */
    const unsigned p8_want = ((R.out_fmt == 0 || R.out_fmt == 2) && !R.mea && !getenv("SA_CLI_EXPAND_EARLY") && !getenv("SA_CLI_PAIRS16")) ? SA_FLAG_PAIRS8 : 0u;
    int p8_used[2] = {0, 0};
    for (int s = 0; s < n_strands; s++) {
        pairs[s] = calloc((size_t) (n_ok > 0 ? n_ok : 1), sizeof(sa_pair_t *));
        n_pairs[s] = calloc((size_t) (n_ok > 0 ? n_ok : 1), sizeof(int64_t));
        if (n_ok == 0) continue;
        fprintf(stderr, s == 0 ? "signalAlign - starting template alignment\n" : "signalAlign - starting complement alignment\n");
        for (int64_t j = 0; j < n_ok; j++) bj[j] = reads[who[j]].jobs[s];
        int rc;
        if (!R.mea && getenv("SA_CLI_EXPAND_EARLY")) {   /* (A/B hook: the one-shot call of rounds 1-3) */
            rc = sa_align_batch(R.two_dist ? reads[who[0]].model[s] : sms[s]->model, &R.p, bj, n_ok, R.ambig, device, 0, pairs[s],
                                n_pairs[s]);
        } else if (!R.mea) {   /* the batch stays alive for the rendering, which expands its packed records job by job */
            sa_batch_t *b = NULL;
            rc = sa_batch_create(&b, R.two_dist ? reads[who[0]].model[s] : sms[s]->model, &R.p, bj, n_ok, R.ambig, device, vc_flag | p8_want);
            p8_used[s] = rc == SA_OK && p8_want != 0;
            if (rc == SA_EUNSUPPORTED && p8_want)
                rc = sa_batch_create(&b, R.two_dist ? reads[who[0]].model[s] : sms[s]->model, &R.p, bj, n_ok, R.ambig, device, vc_flag);
            if (rc == SA_OK) rc = sa_batch_run(b);
            for (int64_t j = 0; j < n_ok && rc == SA_OK; j++) rc = sa_batch_n_pairs(b, j, &n_pairs[s][j]);
            if (vc_flag && rc == SA_OK) {
                free(all_n[s]); free(all_sum[s]);
                all_n[s] = calloc((size_t) n_ok, sizeof(int64_t));
                all_sum[s] = calloc((size_t) n_ok, sizeof(int64_t));
                for (int64_t j = 0; j < n_ok && rc == SA_OK; j++) rc = sa_batch_all_pairs_summary(b, j, &all_n[s][j], &all_sum[s][j]);
            }
            /* only the packed pairs (pinned host memory) are read from here on: the batch's HBM goes back now, so that the
             * complement strand's batch -- and, with a render thread, the next slice's -- plans into the whole card */
            if (rc == SA_OK) rc = sa_batch_release_device(b);
            if (rc == SA_OK) batches[s] = b;
            else sa_batch_destroy(b);
        } else { /* the same batch, kept alive for the path step: its pairs are still on the device */
            sa_batch_t *b = NULL;
            mea[s] = calloc((size_t) n_ok, sizeof(sa_mea_pair_t *));
            n_mea[s] = calloc((size_t) n_ok, sizeof(int64_t));
            rc = sa_batch_create(&b, R.two_dist ? reads[who[0]].model[s] : sms[s]->model, &R.p, bj, n_ok, R.ambig, device, 0);
            if (rc == SA_OK) rc = sa_batch_run(b);
            for (int64_t j = 0; j < n_ok && rc == SA_OK; j++) {
                sa_batch_n_pairs(b, j, &n_pairs[s][j]);
                pairs[s][j] = malloc(sizeof(sa_pair_t) * (size_t) (n_pairs[s][j] > 0 ? n_pairs[s][j] : 1));
                rc = sa_batch_pairs(b, j, pairs[s][j], n_pairs[s][j]);
            }
            if (rc == SA_OK) rc = sa_batch_mea(b, 0, mea[s], n_mea[s], NULL, NULL, NULL);
            sa_batch_destroy(b);
        }
        if (!validated && rc != SA_OK && rc != SA_ENODEVICE && rc != SA_ENOMEM) {
            /* the planner turned the batch down (a letter outside the alphabet, anchors that give no band, a cell of more paths
             * or a matrix larger than the result records can name -- SA_EUNSUPPORTED --, ...): find the reads whose jobs it
             * rejects (each planned alone on the host, all host threads), let them fail alone as the reference's
             * one-process-per-read runs would, and start over.  Only when no read is to blame does the run end below. */
            validated = 1;
            validate_ctx_t vc = {&R, reads, who};
            parallel_for(n_ok, validate_one, &vc);
            int64_t k2 = 0;
            for (int64_t j = 0; j < n_ok; j++)
                if (!reads[who[j]].failed) who[k2++] = who[j];
            if (k2 < n_ok) {
                for (int q = 0; q <= s; q++) {
                    sa_batch_destroy(batches[q]);
                    batches[q] = NULL;
                    for (int64_t j = 0; j < n_ok; j++) { sa_free(pairs[q][j]); if (R.mea && mea[q]) sa_free(mea[q][j]); }
                    free(pairs[q]); free(n_pairs[q]);
                    if (R.mea) { free(mea[q]); free(n_mea[q]); mea[q] = NULL; n_mea[q] = NULL; }
                    pairs[q] = NULL; n_pairs[q] = NULL;
                }
                n_ok = k2;
                s = -1;   /* both strands again, without the offenders */
                continue;
            }
        }
        if (rc != SA_OK) {
            fprintf(stderr, "signalMachine: alignment failed: %s\n", sa_strerror(rc));
            exit(1);
        }
    }

    /* ---- outputs: rendered in parallel (one file per read), summary lines in read order ---- */
    g_t_gpu += now_s() - ts1;
    render_job_t *job = calloc(1, sizeof(*job));
    job->Rp = Rp; job->reads = reads; job->n_reads = n_reads; job->n_ok = n_ok; job->who = who; job->bj = bj;
    for (int s = 0; s < 2; s++) { job->pairs_s[s] = pairs[s]; job->n_pairs_s[s] = n_pairs[s]; job->mea_s[s] = mea[s]; job->n_mea_s[s] = n_mea[s]; job->batch[s] = batches[s]; job->all_n_s[s] = all_n[s]; job->all_sum_s[s] = all_sum[s]; job->p8_s[s] = p8_used[s]; }
    return job;
#undef R
}

/* rendering of a slice (one file per read, in parallel), summary lines in read order, then the reads' memory goes back */
static void *render_slice(void *arg) {
    render_job_t *job = arg;
    run_t *Rp = job->Rp;
#define R (*Rp)
    read_t *reads = job->reads;
    const int64_t n_reads = job->n_reads, n_ok = job->n_ok;
    int64_t *who = job->who;
    sa_job_t *bj = job->bj;
    sa_pair_t ***pairs = job->pairs_s;
    int64_t **n_pairs = job->n_pairs_s;
    sa_mea_pair_t ***mea = job->mea_s;
    int64_t **n_mea = job->n_mea_s;
    const int n_strands = R.two_d ? 2 : 1;
    const double ts2 = now_s();
    double (*score)[2] = calloc((size_t) (n_ok > 0 ? n_ok : 1), sizeof(*score));
    {
        out_job_t oc = {&R, reads, who, pairs, n_pairs, score, mea, n_mea, job->batch, job->all_n_s, job->all_sum_s, job->p8_s};
        if (outputs_distinct(reads, who, n_ok)) parallel_for(n_ok, output_one, &oc);
        else for (int64_t j = 0; j < n_ok; j++) output_one(j, &oc);
    }
    for (int s = 0; s < 2; s++) { sa_batch_destroy(job->batch[s]); job->batch[s] = NULL; }
    for (int64_t j = 0; j < n_ok; j++) {
        read_t *rd = &reads[who[j]];
        if (rd->failed) continue;
        fprintf(stdout, "%s %" PRId64 "\t%" PRId64 "(%f)\t", rd->label, rd->n_guide, job->all_n_s[0] ? job->all_n_s[0][j] : n_pairs[0][j], score[j][0]);
        if (R.two_d) fprintf(stdout, "%" PRId64 "(%f)\n", job->all_n_s[1] ? job->all_n_s[1][j] : n_pairs[1][j], score[j][1]);
        else fprintf(stdout, "\n");
        fprintf(stderr, "signalAlign - SUCCESS: finished alignment of query %s, exiting\n", rd->label);
        for (int s = 0; s < n_strands; s++) {
            sa_free(pairs[s][j]);
            if (R.mea) sa_free(mea[s][j]);
        }
    }
    t_add(&g_t_render, now_s() - ts2);
    int64_t n_failed = 0;
    for (int64_t i = 0; i < n_reads; i++) n_failed += reads[i].failed ? 1 : 0;
    for (int s = 0; s < n_strands; s++) { free(pairs[s]); free(n_pairs[s]); free(job->all_n_s[s]); free(job->all_sum_s[s]); if (R.mea) { free(mea[s]); free(n_mea[s]); } }
    free(score); free(bj); free(who);
    for (int64_t i = 0; i < n_reads; i++) release_read(&reads[i]);
    job->n_failed = n_failed;
    return NULL;
#undef R
}

/* what a read holds once its outputs are written */
static void release_read(read_t *rd) {
    if (rd->pA) sa_cigar_free(rd->pA);
    if (rd->np) sa_npread_free(rd->np);
    free(rd->forward_seq); free(rd->backward_seq);
    for (int s = 0; s < 2; s++) { free(rd->ax[s]); free(rd->ay[s]); rd->ax[s] = rd->ay[s] = NULL; }
    for (int s = 0; s < 2; s++) { if (rd->model[s]) sa_model_destroy(rd->model[s]); rd->model[s] = NULL; }
    rd->pA = NULL; rd->np = NULL; rd->forward_seq = rd->backward_seq = NULL;
}

int main(int argc, char **argv) {
    run_t R;
    memset(&R, 0, sizeof(R));
    int64_t diag_expansion = 50, trace_back = 50, batch_reads = 2048;
    double threshold = 0.01;
    int device = 0; /* --device: which GPU of the node (one process per GPU; reads shard across processes) */
    R.constraint_trim = 14;
    char *t_model = NULL, *c_model = NULL, *label = NULL, *npread_path = NULL, *cigar_path = NULL, *post_path = NULL;
    char *t_expect = NULL, *c_expect = NULL, *t_hdp = NULL, *c_hdp = NULL, *fwd_ref = NULL, *bwd_ref = NULL,
         *post_path2 = NULL, *seq_name = NULL, *ambig_model = NULL, *manifest = NULL;
    static struct option long_options[] = {{"help", no_argument, 0, 'h'},
                                           {"sm3Hdp", no_argument, 0, 'd'},
                                           {"sparse_output", no_argument, 0, 's'},
                                           {"twoD", no_argument, 0, 'e'},
                                           {"rna", no_argument, 0, 'r'},
                                           {"templateModel", required_argument, 0, 'T'},
                                           {"complementModel", required_argument, 0, 'C'},
                                           {"readLabel", required_argument, 0, 'L'},
                                           {"npRead", required_argument, 0, 'q'},
                                           {"exonerate_cigar_file", required_argument, 0, 'p'},
                                           {"posteriors", required_argument, 0, 'u'},
                                           {"templateHdp", required_argument, 0, 'v'},
                                           {"complementHdp", required_argument, 0, 'w'},
                                           {"templateExpectations", required_argument, 0, 't'},
                                           {"complementExpectations", required_argument, 0, 'c'},
                                           {"diagonalExpansion", required_argument, 0, 'x'},
                                           {"threshold", required_argument, 0, 'D'},
                                           {"constraintTrim", required_argument, 0, 'm'},
                                           {"forward_reference_path", required_argument, 0, 'f'},
                                           {"backward_reference_path", optional_argument, 0, 'b'},
                                           {"sequence_name", required_argument, 0, 'n'},
                                           {"traceBackDiagonals", optional_argument, 0, 'g'},
                                           {"posteriorProbsFile2", optional_argument, 0, 'i'},
                                           {"ambig_model", optional_argument, 0, 'a'},
                                           {"batch", required_argument, 0, 1000},
                                           {"device", required_argument, 0, 1001},
                                           {"mea", no_argument, 0, 1002},
                                           {"batch-reads", required_argument, 0, 1003},
                                           {"emission", required_argument, 0, 1004},
                                           {0, 0, 0, 0}};
    for (;;) {
        int idx = 0;
        int key = getopt_long(argc, argv, "h:d:e:s:r:o:a:T:C:a:L:q:f:b:g:i:p:u:v:w:t:c:x:D:m:n:", long_options, &idx);
        if (key == -1) break;
        switch (key) {
            case 'h': usage(); return 1;
            case 's': if (optarg) sscanf(optarg, "%" SCNd64, &R.out_fmt); break;
            case 'e': R.two_d = 1; break;
            case 'a': ambig_model = optarg ? strdup(optarg) : NULL; break;
            case 'r': R.rna = 1; break;
            case 'd': R.hdp = 1; break;
            case 'T': t_model = strdup(optarg); break;
            case 'C': c_model = strdup(optarg); break;
            case 'L': label = strdup(optarg); break;
            case 'q': npread_path = strdup(optarg); break;
            case 'p': cigar_path = strdup(optarg); break;
            case 'u': post_path = strdup(optarg); break;
            case 't': t_expect = strdup(optarg); break;
            case 'c': c_expect = strdup(optarg); break;
            case 'v': t_hdp = strdup(optarg); break;
            case 'w': c_hdp = strdup(optarg); break;
            case 'x': sscanf(optarg, "%" SCNd64, &diag_expansion); break;
            case 'D': sscanf(optarg, "%lf", &threshold); break;
            case 'm': sscanf(optarg, "%" SCNd64, &R.constraint_trim); break;
            case 'f': fwd_ref = strdup(optarg); break;
            case 'b': bwd_ref = optarg ? strdup(optarg) : NULL; break;
            case 'n': seq_name = strdup(optarg); break;
            case 'g': if (optarg) sscanf(optarg, "%" SCNd64, &trace_back); break;
            case 'i': post_path2 = optarg ? strdup(optarg) : NULL; break;
            case 1000: manifest = strdup(optarg); break;
            case 1001: device = atoi(optarg); break;
            case 1002: R.mea = 1; break;
            case 1003: batch_reads = atoll(optarg) > 0 ? atoll(optarg) : batch_reads; break;
            case 1004:
                if (!strcmp(optarg, "twoDist")) R.two_dist = 1;
                else if (strcmp(optarg, "meanOnly")) die("signalMachine: --emission takes meanOnly or twoDist, not %s", optarg);
                break;
            default: usage(); return 1;
        }
    }
    if (!label) label = strdup("");
    if (t_model == NULL || (c_model == NULL && R.two_d)) die("Missing model files, exiting", NULL);
    if (R.out_fmt == 3 && post_path2 == NULL && manifest == NULL) die("Must pass in posteriorProbsFile2 if using 'both' outFmt", NULL);
    if (cigar_path == NULL && manifest == NULL) die("[signalMachine]ERROR: Need to provide input guide alignments, exiting", NULL);
    R.fwd_ref = fwd_ref;
    R.bwd_ref = bwd_ref;

    /* the reads of this run */
    read_t *reads = NULL;
    int64_t n_reads = 0;
    const int batch_mode = manifest != NULL;
    if (R.two_dist && (batch_mode || R.hdp || R.expect_mode))
        die("signalMachine: --emission twoDist aligns one read per process with a Gaussian model%s", "");
    if (batch_mode) {
        n_reads = load_manifest(manifest, &reads);
        if (n_reads < 0) die("[signalMachine]ERROR: cannot read the batch manifest %s", manifest);
        for (int64_t i = 0; i < n_reads; i++) {
            if (reads[i].seq_name == NULL && seq_name != NULL) reads[i].seq_name = strdup(seq_name);
            if (reads[i].t_expect != NULL || reads[i].c_expect != NULL) R.expect_mode = 1;
        }
        if (R.expect_mode)
            for (int64_t i = 0; i < n_reads; i++)
                if (reads[i].t_expect == NULL && reads[i].c_expect == NULL)
                    die("[signalMachine]ERROR: batch manifest mixes expectation and alignment reads (%s)", reads[i].label);
    } else {
        reads = calloc(1, sizeof(read_t));
        n_reads = 1;
        reads[0].label = label; reads[0].npread_path = npread_path; reads[0].cigar_path = cigar_path;
        reads[0].post_path = post_path; reads[0].post_path2 = post_path2; reads[0].seq_name = seq_name;
        reads[0].t_expect = t_expect; reads[0].c_expect = c_expect;
        R.expect_mode = t_expect != NULL || c_expect != NULL;
        if (fwd_ref == NULL || seq_name == NULL) {
            /* the reference needs -n; kept after the cigar check so that the error order matches (impl/signalMachine.c:642-663) */
            sa_cigar_t *probe = NULL;
            if (sa_cigar_load(cigar_path, &probe) != SA_OK)
                die("[signalMachine]ERROR: Didn't find input alignment file, looked %s", cigar_path);
            sa_cigar_free(probe);
            die("[signalMachine] ERROR: need -f <fasta> and -n <sequence name>", NULL);
        }
    }

    R.p.threshold = threshold;
    R.p.diagonal_expansion = diag_expansion % 2 == 0 ? diag_expansion : diag_expansion + 1;
    R.p.trace_back_diagonals = trace_back;
    R.p.min_diags_between_trace_back = 1000;
    R.p.split_matrix_bigger_than_this = (int64_t) 3000 * 3000;

    if (t_hdp != NULL || c_hdp != NULL) {
        if (t_hdp == NULL || (c_hdp == NULL && R.two_d)) die("Need to have template and complement HDPs", NULL);
        if (!R.hdp) {
            R.hdp = 1;
            fprintf(stderr, "[signalAlign] - Using threeStateHdp stateMachine since you pass in an HDP file\n");
        } else {
            fprintf(stderr, "[signalAlign] - using NanoporeHDPs\n");
        }
    }
    if (R.hdp && t_hdp == NULL) die("signalAlign - ERROR: --sm3Hdp needs -v <template .nhdp>", NULL);

    if (load_strand_model(&R.smt, t_model, R.hdp ? t_hdp : NULL) != SA_OK)
        die("signalAlign - ERROR: couldn't find model file here: %s", t_model);
    if (R.two_d && load_strand_model(&R.smc, c_model, R.hdp ? c_hdp : NULL) != SA_OK)
        die("signalAlign - ERROR: couldn't find model file here: %s", c_model);
    if (ambig_model) {
        if (sa_load_ambig(ambig_model, R.ambig) != SA_OK) {
            printf("Couldn't open %s for reading\n", ambig_model);
            return 1;
        }
    } else {
        sa_default_ambig(R.ambig);
    }

    if (R.hdp && !R.expect_mode) { /* the alignment branch sets the HDP expected values (impl/signalMachine.c:861-863), the expectation branch does not */
        set_hdp_expected(&R.smt);
        if (R.two_d) set_hdp_expected(&R.smc);
    }
    /* the reads go through in slices of --batch-reads (default 2048): bounded host and device memory for any manifest */
    /* Two slices are in the air: while the GPU stage and the rendering of slice k run here, a second thread does the host side
     * of slice k+1 (10 000 short reads: host stage 0.39 s, GPU 0.28 s, rendering 0.25 s, one after the other before).  A third
     * stage -- rendering on a thread of its own -- exists behind SA_CLI_RENDER_THREAD=1 and does not pay (see below). */
    int64_t n_failed = 0;
    pthread_t render_th;
    render_job_t *render_prev = NULL;
    int rendering = 0;
    slice_prep_t cur = {&R, reads, n_reads < batch_reads ? n_reads : batch_reads, batch_mode}, nxt;
    slice_prepare(&cur);
    for (int64_t off = 0; off < n_reads; off += batch_reads) {
        const int64_t n = n_reads - off < batch_reads ? n_reads - off : batch_reads;
        pthread_t th;
        int started = 0;
        if (off + n < n_reads) {
            const int64_t n2 = n_reads - off - n < batch_reads ? n_reads - off - n : batch_reads;
            nxt = (slice_prep_t) {&R, reads + off + n, n2, batch_mode};
            started = pthread_create(&th, NULL, slice_prepare, &nxt) == 0;
            if (!started) slice_prepare(&nxt);
        }
        int64_t failed_now = 0;
        render_job_t *job = run_slice(&R, reads + off, n, batch_mode, device, &failed_now);
        n_failed += failed_now;
        /* the previous slice's rendering has had this slice's GPU stage to finish in; slices are rendered one after the other,
         * so the summary lines stay in read order */
        if (rendering) {
            pthread_join(render_th, NULL);
            n_failed += render_prev->n_failed;
            free(render_prev);
            rendering = 0;
        }
        if (job) {
            /* SA_CLI_RENDER_THREAD=1: the slice is rendered on a thread of its own while the next slice's GPU stage runs.
             * Measured on the 16-CPU quota of a GPU box and found SLOWER (30 000 long reads: 9.9 against 8.5 s; 100 000 short
             * reads: no difference): the front door is bound by host CPU time -- text parsing and TSV rendering, 110
             * thread-seconds per 30 000 long reads on 16 CPUs -- not by the order of its stages. */
            const char *ert = getenv("SA_CLI_RENDER_THREAD");
            if (ert && atoi(ert) == 1) {
                render_prev = job;
                rendering = pthread_create(&render_th, NULL, render_slice, job) == 0;
            }
            if (!rendering) { render_slice(job); n_failed += job->n_failed; free(job); }
        }
        if (started) pthread_join(th, NULL);
    }
    if (rendering) {
        pthread_join(render_th, NULL);
        n_failed += render_prev->n_failed;
        free(render_prev);
    }
    if (batch_mode)
        fprintf(stderr, "[signalMachine] batch: %" PRId64 " of %" PRId64 " reads aligned\n", n_reads - n_failed, n_reads);
    if (getenv("SA_CLI_TIMING"))
        fprintf(stderr, "[signalMachine] timing: host stage %.3f s wall (thread-seconds: npRead+cigar parse %.3f, reference fetch "
                        "%.3f, parameter estimation+anchors %.3f), GPU stage %.3f s wall, render+write %.3f s wall\n",
                g_t_prep, g_ts_parse, g_ts_fetch, g_ts_estimate, g_t_gpu, g_t_render);
    return n_failed == 0 ? 0 : 1;
}
