/* signalMachine -- drop-in replacement of signalAlign's per-read aligner executable (impl/signalMachine.c).
 *
 * Same argv (getopt table impl/signalMachine.c:514-543), same input files (.model, .nhdp, .npRead, exonerate
 * cigar, indexed FASTA), same outputs: the three TSV renderings appended to -u / -i, the summary line on
 * stdout and the "SUCCESS" line on stderr that signalAlignment.py keys on (src/signalalign/signalAlignment.py:480).
 * The banded pair-HMM itself runs on the MI355X through libsignalalign_hip.so; there is no CPU fallback.
 *
 * Expectations mode (-t/-c) runs sa_expect_batch and writes the .expectations files of impl/continuousHmm.c.
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <inttypes.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
    fprintf(stderr, "-r: boolean option if read is RNA\n\n");
}

static double descale(double e, double level, double scale, double shift, double var) {
    return (e + var * level - scale * level - shift) / var;
}

typedef struct {
    sa_model_t *model;
    double *table;          /* working copy of EMISSION_MATCH_MATRIX (noise columns rescaled per read) */
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

/* writePosteriorProbsFull, impl/signalMachine.c:89-159 */
static void write_full(const char *path, const out_ctx_t *o) {
    FILE *fh = fopen(path, "a");
    if (!fh) die("signalMachine: cannot open output %s", path);
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
        double E_mean = o->sm->table[(int64_t) p->kmer_id * 5], E_noise = o->sm->table[(int64_t) p->kmer_id * 5 + 2];
        double scaled_Emean = E_mean * o->npp.scale + o->npp.shift;
        double scaled_Enoise = E_noise * o->npp.scale_sd;
        double descaled = descale(ev_mean, E_mean, o->npp.scale, o->npp.shift, o->npp.var);
        char *ref_kmer = ((o->is_template && o->forward) || (!o->is_template && !o->forward)) ? strdup(k_i)
                                                                                              : sa_reverse_complement(k_i);
        if (o->rna) {
            char *t = sa_reverse_complement(ref_kmer);
            free(ref_kmer);
            ref_kmer = t;
        }
        fprintf(fh, "%s\t%" PRId64 "\t%s\t%s\t%s\t%" PRId64 "\t%f\t%f\t%f\t%s\t%f\t%f\t%f\t%f\t%f\t%s\n", o->contig, x_adj,
                ref_kmer, o->label, o->is_template ? "t" : "c", y, ev_mean, ev_noise, ev_dur, k_i, scaled_Emean,
                scaled_Enoise, prob, descaled, E_mean, path_kmer);
        free(ref_kmer);
    }
    fclose(fh);
}

/* writePosteriorProbsVC, impl/signalMachine.c:161-232: only k-mers holding the internal ambiguity letter X */
static void write_vc(const char *path, const out_ctx_t *o) {
    int forward = o->forward;
    int label_forward = (o->rna || !o->is_template) ? !forward : forward;
    FILE *fh = fopen(path, "a");
    if (!fh) die("signalMachine: cannot open output %s", path);
    const int k = o->sm->k;
    int64_t ref_len = (int64_t) strlen(o->target), ref_len_kmers = ref_len - k;
    char k_i[16], path_kmer[16];
    for (int64_t i = 0; i < o->n_pairs; i++) {
        const sa_pair_t *p = &o->pairs[i];
        memcpy(k_i, o->target + p->x, k);
        k_i[k] = 0;
        int same = (o->is_template && forward) || (!o->is_template && !forward);
        char *ref_kmer = same ? strdup(k_i) : sa_reverse_complement(k_i);
        if (!strchr(ref_kmer, 'X')) {
            free(ref_kmer);
            continue;
        }
        int64_t x_adj = adjust_ref(p->x, o->ref_offset, ref_len_kmers, ref_len, o->is_template, forward);
        int64_t y = p->y + o->event_offset;
        double prob = ((double) p->prob_e7) / PROB_1;
        kmer_string(o->sm, p->kmer_id, path_kmer);
        for (int q = 0; q < k; q++) {
            if (ref_kmer[q] != 'X') continue;
            int qp = same ? q : (k - 1) - q; /* adjustQueryPosition :81-87 */
            fprintf(fh, "%" PRId64 "\t%" PRId64 "\t%c\t%f\t%s\t%s\t%s\t%f\t%s\n", y, x_adj + q, path_kmer[qp], prob,
                    o->is_template ? "t" : "c", label_forward ? "forward" : "backward", o->label, o->score, o->contig);
        }
        free(ref_kmer);
    }
    fclose(fh);
}

/* writeAssignments, impl/signalMachine.c:234-270 */
static void write_assignments(const char *path, const out_ctx_t *o) {
    FILE *fh = fopen(path, "a");
    if (!fh) die("signalMachine: cannot open output %s", path);
    char path_kmer[16];
    for (int64_t i = 0; i < o->n_pairs; i++) {
        const sa_pair_t *p = &o->pairs[i];
        int64_t y = p->y + o->event_offset;
        double prob = ((double) p->prob_e7) / PROB_1;
        kmer_string(o->sm, p->kmer_id, path_kmer);
        double E_mean = o->sm->table[(int64_t) p->kmer_id * 5];
        double descaled = descale(o->events[y * 4], E_mean, o->npp.scale, o->npp.shift, o->npp.var);
        fprintf(fh, "%s\t%s\t%lf\t%lf\n", path_kmer, o->is_template ? "t" : "c", descaled, prob);
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

/* continuousPairHmm_writeToFile (impl/continuousHmm.c:352-408) / hdpHmm_writeToFile (:572-623) */
static void write_expectations(const char *path, const strand_model_t *sm, int hdp, const double *trans, double lik,
                               const sa_job_t *job, const sa_assignment_t *as, int64_t n_as) {
    for (int i = 0; i < 9; i++)
        if (isnan(trans[i])) { /* hmmContinuous_checkTransitions: an empty file is left behind */
            fprintf(stderr, "GOT NaN TRANS\n");
            FILE *f0 = fopen(path, "w");
            if (f0) fclose(f0);
            return;
        }
    FILE *fh = fopen(path, "w");
    if (!fh) die("signalMachine: cannot open %s for writing", path);
    int64_t n_kmers = 1;
    for (int i = 0; i < sm->k; i++) n_kmers *= sm->n_alpha;
    fprintf(fh, "%d\t%d\t%s\t%d\t\n", 3, sm->n_alpha, sm->alphabet, sm->k);
    for (int i = 0; i < 9; i++) fprintf(fh, "%f\t", trans[i]);
    fprintf(fh, "%f\n", lik);
    for (int64_t i = 0; i < n_kmers * 5; i++) fprintf(fh, "%lf\t", sm->table[i]);
    fprintf(fh, "\n");
    if (!hdp) {
        for (int64_t i = 0; i < n_kmers * 2; i++) fprintf(fh, "%lf\t", 0.0);   /* eventExpectations: never updated */
        fprintf(fh, "\n");
        for (int64_t i = 0; i < n_kmers; i++) fprintf(fh, "%lf\t", 0.001);     /* posteriors = emissionsPseudocount */
        fprintf(fh, "\n");
        for (int64_t i = 0; i < n_kmers; i++) fprintf(fh, "%d\t", 0);          /* observed mask */
        fprintf(fh, "\n");
    } else {
        for (int64_t i = 0; i < n_as; i++) fprintf(fh, "%lf\t", job->events[as[i].event * job->event_stride]);
        fprintf(fh, "\n");
        for (int64_t i = 0; i < n_as; i++) {
            for (int n = 0; n < sm->k; n++) fputc(job->ref[as[i].ref_pos + n], fh);
            fputc('\t', fh);
        }
        fprintf(fh, "\n");
    }
    fclose(fh);
}

static int load_strand_model(strand_model_t *sm, const char *model_path, const char *nhdp_path) {
    int rc = sa_model_load(&sm->model, model_path, nhdp_path);
    if (rc) return rc;
    sa_model_alphabet(sm->model, sm->alphabet, &sm->n_alpha, &sm->k);
    int64_t n = 5;
    for (int i = 0; i < sm->k; i++) n *= sm->n_alpha;
    sm->table = malloc(sizeof(double) * (size_t) n);
    memcpy(sm->table, sa_model_table5(sm->model), sizeof(double) * (size_t) n);
    return SA_OK;
}

int main(int argc, char **argv) {
    int hdp = 0, two_d = 0, rna = 0;
    int64_t diag_expansion = 50, constraint_trim = 14, trace_back = 50, out_fmt = 0;
    double threshold = 0.01;
    char *t_model = NULL, *c_model = NULL, *label = NULL, *npread_path = NULL, *cigar_path = NULL, *post_path = NULL;
    char *t_expect = NULL, *c_expect = NULL, *t_hdp = NULL, *c_hdp = NULL, *fwd_ref = NULL, *bwd_ref = NULL,
         *post_path2 = NULL, *seq_name = NULL, *ambig_model = NULL;
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
                                           {0, 0, 0, 0}};
    for (;;) {
        int idx = 0;
        int key = getopt_long(argc, argv, "h:d:e:s:r:o:a:T:C:a:L:q:f:b:g:i:p:u:v:w:t:c:x:D:m:n:", long_options, &idx);
        if (key == -1) break;
        switch (key) {
            case 'h': usage(); return 1;
            case 's': if (optarg) sscanf(optarg, "%" SCNd64, &out_fmt); break;
            case 'e': two_d = 1; break;
            case 'a': ambig_model = optarg ? strdup(optarg) : NULL; break;
            case 'r': rna = 1; break;
            case 'd': hdp = 1; break;
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
            case 'm': sscanf(optarg, "%" SCNd64, &constraint_trim); break;
            case 'f': fwd_ref = strdup(optarg); break;
            case 'b': bwd_ref = optarg ? strdup(optarg) : NULL; break;
            case 'n': seq_name = strdup(optarg); break;
            case 'g': if (optarg) sscanf(optarg, "%" SCNd64, &trace_back); break;
            case 'i': post_path2 = optarg ? strdup(optarg) : NULL; break;
            default: usage(); return 1;
        }
    }
    if (!label) label = strdup("");
    if (t_model == NULL || (c_model == NULL && two_d)) die("Missing model files, exiting", NULL);
    if (out_fmt == 3 && post_path2 == NULL) die("Must pass in posteriorProbsFile2 if using 'both' outFmt", NULL);
    if (cigar_path == NULL) die("[signalMachine]ERROR: Need to provide input guide alignments, exiting", NULL);
    sa_cigar_t *pA = NULL;
    if (sa_cigar_load(cigar_path, &pA) != SA_OK)
        die("[signalMachine]ERROR: Didn't find input alignment file, looked %s", cigar_path);
    fprintf(stderr, "[signalMachine]NOTICE: Using guide alignments from %s\n", cigar_path);
    const int expect_mode = t_expect != NULL || c_expect != NULL;

    sa_params_t p;
    p.threshold = threshold;
    p.diagonal_expansion = diag_expansion % 2 == 0 ? diag_expansion : diag_expansion + 1;
    p.trace_back_diagonals = trace_back;
    p.min_diags_between_trace_back = 1000;
    p.split_matrix_bigger_than_this = (int64_t) 3000 * 3000;

    if (t_hdp != NULL || c_hdp != NULL) {
        if (t_hdp == NULL || (c_hdp == NULL && two_d)) die("Need to have template and complement HDPs", NULL);
        if (!hdp) {
            hdp = 1;
            fprintf(stderr, "[signalAlign] - Using threeStateHdp stateMachine since you pass in an HDP file\n");
        } else {
            fprintf(stderr, "[signalAlign] - using NanoporeHDPs\n");
        }
    }
    if (hdp && t_hdp == NULL) die("signalAlign - ERROR: --sm3Hdp needs -v <template .nhdp>", NULL);

    strand_model_t smt, smc;
    memset(&smt, 0, sizeof(smt));
    memset(&smc, 0, sizeof(smc));
    if (load_strand_model(&smt, t_model, hdp ? t_hdp : NULL) != SA_OK)
        die("signalAlign - ERROR: couldn't find model file here: %s", t_model);
    if (two_d && load_strand_model(&smc, c_model, hdp ? c_hdp : NULL) != SA_OK)
        die("signalAlign - ERROR: couldn't find model file here: %s", c_model);

    sa_npread_t *np = NULL;
    if (npread_path == NULL || sa_npread_load(npread_path, &np) != SA_OK)
        die("signalMachine: could not load the nanopore read %s", npread_path);
    if (rna) {
        int64_t tmp = pA->start2;
        pA->start2 = np->template_read_length - pA->end2;
        pA->end2 = np->template_read_length - tmp;
    }
    if (fwd_ref == NULL || seq_name == NULL) die("[signalMachine] ERROR: need -f <fasta> and -n <sequence name>", NULL);

    /* fastaHandler_ReferenceSequenceConstructFull, impl/fasta_handler.c:47-102 */
    if (rna) { /* listReverse(pA->operationList) */
        for (int64_t i = 0, j = pA->n_ops - 1; i < j; i++, j--) {
            int32_t t = pA->op_type[i]; pA->op_type[i] = pA->op_type[j]; pA->op_type[j] = t;
            int64_t l = pA->op_len[i]; pA->op_len[i] = pA->op_len[j]; pA->op_len[j] = l;
        }
    }
    int ferr = 0;
    char *forward_seq = pA->strand1 ? sa_fasta_fetch(fwd_ref, seq_name, pA->start1, pA->end1 - 1, &ferr)
                                    : sa_fasta_fetch(fwd_ref, seq_name, pA->end1, pA->start1 - 1, &ferr);
    if (ferr == -2) {
        fprintf(stderr, "[signalMachine] ERROR %d: sequence name: %s is not in reference fasta: %s \n", ferr, seq_name, fwd_ref);
        return 1;
    }
    if (forward_seq == NULL) die("[signalMachine] ERROR: Unable to fetch reference sequence.  ", NULL);
    char *backward_seq;
    if (bwd_ref) {
        backward_seq = pA->strand1 ? sa_fasta_fetch(bwd_ref, seq_name, pA->start1, pA->end1 - 1, &ferr)
                                   : sa_fasta_fetch(bwd_ref, seq_name, pA->end1, pA->start1 - 1, &ferr);
        if (backward_seq == NULL) die("[signalMachine] ERROR: Unable to fetch reference sequence.  ", NULL);
        sa_reverse_in_place(backward_seq);
    } else {
        backward_seq = sa_complement(forward_seq);
        sa_reverse_in_place(backward_seq);
    }
    int strand1 = pA->strand1;
    if (rna) {
        char *tmp = backward_seq;
        backward_seq = strdup(forward_seq);
        sa_reverse_in_place(backward_seq);
        free(forward_seq);
        forward_seq = tmp;
        sa_reverse_in_place(forward_seq);
        int64_t t2 = pA->start1;
        pA->start1 = pA->end1;
        pA->end1 = t2;
        pA->strand1 = !pA->strand1;
        strand1 = pA->strand1;
    }
    const char *template_target = strand1 ? forward_seq : backward_seq;
    const char *complement_target = strand1 ? backward_seq : forward_seq;

    /* event slices and coordinate shifts (impl/signalMachine.c:726-750) */
    const int64_t *t_map = two_d ? np->template_event_map : np->template_strand_event_map;
    int64_t t_lo = t_map[pA->start2], t_hi = t_map[pA->end2 - 1];
    int64_t c_lo = 0, c_hi = 0;
    if (two_d) { c_lo = np->complement_event_map[pA->start2]; c_hi = np->complement_event_map[pA->end2 - 1]; }
    int64_t r_shift_t = pA->start1, r_shift_c = two_d ? pA->end1 : 0;
    int forward = pA->strand1;

    /* anchors from the guide alignment (pA is rebased inside, impl/signalMachineUtils.c:142-164) */
    int64_t cap = 0;
    for (int64_t i = 0; i < pA->n_ops; i++) cap += pA->op_len[i];
    int64_t *gx = malloc(sizeof(int64_t) * (size_t) (cap + 1)), *gy = malloc(sizeof(int64_t) * (size_t) (cap + 1));
    int64_t n_guide = sa_guide_to_anchors(pA->start1, pA->end1, pA->strand1, pA->start2, pA->op_type, pA->op_len,
                                          pA->n_ops, constraint_trim, gx, gy, cap + 1);
    if (n_guide < 0) die("signalMachine: could not convert the guide alignment", NULL);

    const char *ambig[256];
    if (ambig_model) {
        if (sa_load_ambig(ambig_model, ambig) != SA_OK) {
            printf("Couldn't open %s for reading\n", ambig_model);
            return 1;
        }
    } else {
        sa_default_ambig(ambig);
    }

    /* ---- per-strand work: estimate parameters, build the job ---- */
    if (expect_mode) fprintf(stderr, "Starting expectations routine\n");
    else fprintf(stderr, "signalAlign - starting template alignment\n");
    double est[7];
    if (sa_estimate_params(smt.model, smt.table, np->template_strand_event_map, np->template_events,
                           np->n_template_events, np->template_read, np->template_read_length, est) != SA_OK)
        die("Cannot get scale params with no assignments", NULL);
    np->template_params.scale = est[0]; np->template_params.shift = est[1]; np->template_params.var = est[2];
    np->template_params.drift = est[3]; np->template_params.scale_sd = est[4]; np->template_params.var_sd = est[5];
    np->template_params.shift_sd = est[6];
    if (hdp && !expect_mode) {
        sa_model_set_to_hdp_expected_values(smt.model);
        int64_t n = 5;
        for (int i = 0; i < smt.k; i++) n *= smt.n_alpha;
        const double *mt = sa_model_table5(smt.model);
        for (int64_t i = 0; i < n; i += 5) { smt.table[i] = mt[i]; smt.table[i + 1] = mt[i + 1]; }
    }
    sa_job_t jobs[2];
    const strand_model_t *sms[2] = {&smt, &smc};
    int n_jobs = 1;
    int64_t *ax[2] = {NULL, NULL}, *ay[2] = {NULL, NULL};
    ax[0] = malloc(sizeof(int64_t) * (size_t) (n_guide + 1));
    ay[0] = malloc(sizeof(int64_t) * (size_t) (n_guide + 1));
    int64_t na0 = sa_remap_anchors(gx, gy, n_guide, t_map, pA->start2, ax[0], ay[0]);
    memset(jobs, 0, sizeof(jobs));
    jobs[0].ref = template_target;
    jobs[0].ref_len = (int64_t) strlen(template_target);
    jobs[0].events = np->template_events + 4 * t_lo;
    jobs[0].event_stride = 4;
    jobs[0].n_events = t_hi - t_lo;
    jobs[0].anchor_x = ax[0]; jobs[0].anchor_y = ay[0]; jobs[0].n_anchors = na0;
    jobs[0].scale = np->template_params.scale; jobs[0].shift = np->template_params.shift; jobs[0].var = np->template_params.var;
    if (two_d) {
        if (sa_estimate_params(smc.model, smc.table, np->complement_strand_event_map, np->complement_events,
                               np->n_complement_events, np->complement_read, np->complement_read_length, est) != SA_OK)
            die("Cannot get scale params with no assignments", NULL);
        np->complement_params.scale = est[0]; np->complement_params.shift = est[1]; np->complement_params.var = est[2];
        np->complement_params.drift = est[3]; np->complement_params.scale_sd = est[4]; np->complement_params.var_sd = est[5];
        np->complement_params.shift_sd = est[6];
        if (hdp && !expect_mode) {
            sa_model_set_to_hdp_expected_values(smc.model);
            int64_t n = 5;
            for (int i = 0; i < smc.k; i++) n *= smc.n_alpha;
            const double *mt = sa_model_table5(smc.model);
            for (int64_t i = 0; i < n; i += 5) { smc.table[i] = mt[i]; smc.table[i + 1] = mt[i + 1]; }
        }
        ax[1] = malloc(sizeof(int64_t) * (size_t) (n_guide + 1));
        ay[1] = malloc(sizeof(int64_t) * (size_t) (n_guide + 1));
        int64_t na1 = sa_remap_anchors(gx, gy, n_guide, np->complement_event_map, pA->start2, ax[1], ay[1]);
        jobs[1].ref = complement_target;
        jobs[1].ref_len = (int64_t) strlen(complement_target);
        jobs[1].events = np->complement_events + 4 * c_lo;
        jobs[1].event_stride = 4;
        jobs[1].n_events = c_hi - c_lo;
        jobs[1].anchor_x = ax[1]; jobs[1].anchor_y = ay[1]; jobs[1].n_anchors = na1;
        jobs[1].scale = np->complement_params.scale; jobs[1].shift = np->complement_params.shift; jobs[1].var = np->complement_params.var;
        n_jobs = 2;
    }

    if (expect_mode) { /* impl/signalMachine.c:772-848 */
        const char *paths[2] = {t_expect, c_expect};
        for (int s = 0; s < n_jobs; s++) {
            fprintf(stderr, "signalAlign - getting expectations for %s\n", s == 0 ? "template" : "complement");
            double trans[9], lik = 0.0;
            for (int i = 0; i < 9; i++) trans[i] = 0.001; /* transitionsPseudocount, :785 */
            sa_assignment_t *as = NULL;
            int64_t n_as = 0;
            int rc = sa_expect_batch(sms[s]->model, &p, &jobs[s], 1, ambig, 0, 0, trans, &lik, &as, &n_as);
            if (rc != SA_OK) {
                fprintf(stderr, "signalMachine: expectations failed: %s\n", sa_strerror(rc));
                return 1;
            }
            if (hdp)
                fprintf(stderr, s == 0 ? "signalAlign - got %" PRId64 " template HDP assignments\n"
                                       : "signalAlign - got %" PRId64 "complement HDP assignments\n", n_as);
            if (paths[s] != NULL) {
                fprintf(stderr, "signalAlign - writing expectations to file: %s\n", paths[s]);
                write_expectations(paths[s], sms[s], hdp, trans, lik, &jobs[s], as, n_as);
            }
            sa_free(as);
        }
        fprintf(stderr, "signalAlign - SUCCESS: finished alignment of query %s, exiting\n", label);
        return 0;
    }

    /* ---- the pair-HMM on the GPU: one batch per strand model ---- */
    sa_pair_t *pairs[2] = {NULL, NULL};
    int64_t n_pairs[2] = {0, 0};
    for (int s = 0; s < n_jobs; s++) {
        if (s == 1) fprintf(stderr, "signalAlign - starting complement alignment\n");
        int rc = sa_align_batch(sms[s]->model, &p, &jobs[s], 1, ambig, 0, 0, &pairs[s], &n_pairs[s]);
        if (rc != SA_OK) {
            fprintf(stderr, "signalMachine: alignment failed: %s\n", sa_strerror(rc));
            return 1;
        }
    }
    double score[2] = {0, 0};
    for (int s = 0; s < n_jobs; s++) {
        double tot = 0.0;
        for (int64_t i = 0; i < n_pairs[s]; i++) tot += (double) pairs[s][i].prob_e7;
        score[s] = 100.0 * tot / ((double) n_pairs[s] * PROB_1); /* scoreByPosteriorProbabilityIgnoringGaps :407-412 */
    }
    if (post_path != NULL) {
        out_ctx_t o;
        o.label = label; o.contig = pA->contig1; o.sm = &smt; o.npp = np->template_params; o.events = np->template_events;
        o.target = template_target; o.forward = forward; o.is_template = 1; o.rna = rna; o.event_offset = t_lo;
        o.ref_offset = r_shift_t; o.pairs = pairs[0]; o.n_pairs = n_pairs[0]; o.score = score[0];
        output_alignment(out_fmt, post_path, post_path2, &o);
        if (two_d) {
            o.sm = &smc; o.npp = np->complement_params; o.events = np->complement_events; o.target = complement_target;
            o.is_template = 0; o.event_offset = c_lo; o.ref_offset = r_shift_c; o.pairs = pairs[1];
            o.n_pairs = n_pairs[1]; o.score = score[1];
            output_alignment(out_fmt, post_path, post_path2, &o);
        }
    }
    fprintf(stdout, "%s %" PRId64 "\t%" PRId64 "(%f)\t", label, n_guide, n_pairs[0], score[0]);
    if (two_d) fprintf(stdout, "%" PRId64 "(%f)\n", n_pairs[1], score[1]);
    else fprintf(stdout, "\n");
    fprintf(stderr, "signalAlign - SUCCESS: finished alignment of query %s, exiting\n", label);
    return 0;
}
