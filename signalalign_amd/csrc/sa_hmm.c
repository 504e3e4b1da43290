/* sa_hmm.c -- the expectations objects of the EM loop, host only: Hmm / ContinuousPairHmm / HdpHmm
 * (inc/stateMachine.h:64-83, inc/continuousHmm.h:8-75; impl/continuousHmm.c).
 *
 * The accumulators getExpectationsUsingAnchors adds to are filled from sa_expect_batch's results (the transition sums and
 * likelihoods of a batch of reads, HDP assignments as (reference position, event) pairs); what this file holds is everything
 * around that call: construction from a model with pseudocounts (:83-144, :220-231, :522-569, :787-794), the writers
 * (:353-407, :571-628), the READERS (:409-507, :630-785), normalisation (:282-308, impl/discreteHmm.c:125-137) and the M-step's
 * load into the state machine (:320-351).
 *
 * Two places where this file does not follow the reference letter by letter, both in code no caller of signalMachine reaches
 * (loadHmmRoutine is commented out, impl/signalMachine.c:376-379):
 *   - continuousPairHmm_loadTransitionsIntoStateMachine sets TRANSITION_GAP_SWITCH_TO_X = log(T[gapY][gapX]) (:336).  A state
 *     machine loaded from a .model file keeps that transition at log(0) (stateMachine3_loadTransitionsFromFile stores token 7 in the
 *     unused SWITCH_TO_Y, impl/stateMachine.c:1246-1251) and every kernel of this library carries the eight live transitions of
 *     such a machine: sa_hmm_load_into_model loads the seven non-constant ones and leaves gapY -> gapX dead;
 *   - continuousPairHmm_loadEmissionsIntoStateMachine writes the gapY mean to EMISSION_GAP_Y_MATRIX[i + MODEL_PARAMS] (:347,
 *     a typo for i * MODEL_PARAMS that scrambles the gapY table): here the gapY table stays what stateMachine3_loadFromFile makes of
 *     the match table (same mean, 1.75 x the sd).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sa_internal.h"
#include "sa_io.h"

struct sa_hmm {
    int type;                 /* SA_HMM_GAUSSIAN / SA_HMM_HDP */
    int n_alpha, k;
    char alphabet[64];
    int64_t n_kmers;
    double transitions[9];
    double likelihood;
    double *event_model;      /* 5 per k-mer */
    int has_model;
    /* ContinuousPairHmm */
    double *event_expectations; /* 2 per k-mer: sum p * mean, sum p * (mean - running mean)^2 */
    double *posteriors;         /* per k-mer: sum p (starts at the emission pseudocount)      */
    uint8_t *observed;
    /* HdpHmm */
    double threshold;
    int64_t n_assign, cap_assign;
    double *assign_event;
    char *assign_kmer;        /* k characters per assignment */
};

static void hmm_free(sa_hmm_t *h) {
    if (!h) return;
    free(h->event_model); free(h->event_expectations); free(h->posteriors); free(h->observed);
    free(h->assign_event); free(h->assign_kmer);
    free(h);
}

static sa_hmm_t *hmm_new(int type, int n_alpha, const char *alphabet, int k, double t_pc, double e_pc) {
    if (n_alpha < 1 || n_alpha > 60 || k < 1 || k > 12) return NULL;
    sa_hmm_t *h = calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->type = type;
    h->n_alpha = n_alpha;
    h->k = k;
    memcpy(h->alphabet, alphabet, (size_t) n_alpha);   /* (already sorted when it comes from a model; sorted below otherwise) */
    for (int i = 1; i < n_alpha; i++)                   /* sequence_prepareAlphabet (impl/pairwiseAligner.c:366-395) */
        for (int j = i; j > 0 && h->alphabet[j] < h->alphabet[j - 1]; j--) {
            char c = h->alphabet[j]; h->alphabet[j] = h->alphabet[j - 1]; h->alphabet[j - 1] = c;
        }
    h->n_kmers = 1;
    for (int i = 0; i < k; i++) h->n_kmers *= n_alpha;
    for (int i = 0; i < 9; i++) h->transitions[i] = t_pc;
    h->event_model = calloc((size_t) h->n_kmers * 5, sizeof(double));
    if (!h->event_model) { hmm_free(h); return NULL; }
    if (type == SA_HMM_GAUSSIAN) {
        h->event_expectations = calloc((size_t) h->n_kmers * 2, sizeof(double));
        h->posteriors = malloc(sizeof(double) * (size_t) h->n_kmers);
        h->observed = calloc((size_t) h->n_kmers, 1);
        if (!h->event_expectations || !h->posteriors || !h->observed) { hmm_free(h); return NULL; }
        for (int64_t i = 0; i < h->n_kmers; i++) h->posteriors[i] = e_pc;
    }
    return h;
}

/* hmmContinuous_getExpectationsHmm (:841-856) = continuousPairHmm_makeExpectationsHmm (:220-231) / hdpHmm_makeExpectationsHmm (:787-794) */
int sa_hmm_create(sa_hmm_t **out, const sa_model_t *m, int type, double threshold, double transitions_pseudocount,
                  double emissions_pseudocount) {
    if (!out || !m || (type != SA_HMM_GAUSSIAN && type != SA_HMM_HDP)) return SA_EINVAL;
    sa_hmm_t *h = hmm_new(type, m->n_alpha, m->alphabet, m->k, transitions_pseudocount, emissions_pseudocount);
    if (!h) return SA_ENOMEM;
    memcpy(h->event_model, m->table5, sizeof(double) * 5 * (size_t) m->n_kmers);   /* hmmContinuous_loadEventModel */
    h->has_model = 1;
    h->threshold = threshold;
    *out = h;
    return SA_OK;
}

void sa_hmm_destroy(sa_hmm_t *h) { hmm_free(h); }

int sa_hmm_view(sa_hmm_t *h, sa_hmm_view_t *v) {
    if (!h || !v) return SA_EINVAL;
    memset(v, 0, sizeof(*v));
    v->type = h->type; v->n_states = 3; v->n_alpha = h->n_alpha; v->k = h->k;
    memcpy(v->alphabet, h->alphabet, (size_t) h->n_alpha);
    v->n_kmers = h->n_kmers;
    v->transitions = h->transitions;
    v->likelihood = &h->likelihood;
    v->event_model = h->event_model;
    v->event_expectations = h->event_expectations;
    v->posteriors = h->posteriors;
    v->observed = h->observed;
    v->threshold = h->threshold;
    v->n_assignments = h->n_assign;
    v->assignment_events = h->assign_event;
    v->assignment_kmers = h->assign_kmer;
    v->has_model = h->has_model;
    return SA_OK;
}

int sa_hmm_set_event_model(sa_hmm_t *h, const double *table5) {
    if (!h || !table5) return SA_EINVAL;
    memcpy(h->event_model, table5, sizeof(double) * 5 * (size_t) h->n_kmers);
    h->has_model = 1;
    return SA_OK;
}

/* what getExpectationsUsingAnchors leaves in the Hmm for one read: hmm_addToTransitionsExpectation (:147) per cell and
 * transition, hmm->likelihood += total per diagonal (impl/pairwiseAligner.c:1432) -- here the sums sa_expect_batch returns */
int sa_hmm_add_expectations(sa_hmm_t *h, const double *trans9, double likelihood) {
    if (!h || !trans9) return SA_EINVAL;
    for (int i = 0; i < 9; i++) h->transitions[i] += trans9[i];
    h->likelihood += likelihood;
    return SA_OK;
}

/* continuousPairHmm_addToEmissionExpectation (:159-168) */
int sa_hmm_add_emission_expectation(sa_hmm_t *h, int64_t kmer_index, double mean, double p) {
    if (!h || h->type != SA_HMM_GAUSSIAN || kmer_index < 0 || kmer_index >= h->n_kmers) return SA_EINVAL;
    double *e = h->event_expectations + 2 * kmer_index;
    e[0] += (p * mean);
    h->posteriors[kmer_index] += p;
    const double uK = e[0] / h->posteriors[kmer_index];
    e[1] += p * (mean - uK) * (mean - uK);
    h->observed[kmer_index] = 1;
    return SA_OK;
}

/* hdpHmm_addToAssignment (:510-514): the k-mer the cell's pointer names and the event's mean */
int sa_hmm_add_assignment(sa_hmm_t *h, const char *kmer, double event_mean) {
    if (!h || h->type != SA_HMM_HDP || !kmer) return SA_EINVAL;
    if (h->n_assign == h->cap_assign) {
        const int64_t nc = h->cap_assign ? 2 * h->cap_assign : 1024;
        double *ne = realloc(h->assign_event, sizeof(double) * (size_t) nc);
        if (!ne) return SA_ENOMEM;
        h->assign_event = ne;
        char *nk = realloc(h->assign_kmer, (size_t) nc * (size_t) h->k);
        if (!nk) return SA_ENOMEM;
        h->assign_kmer = nk;
        h->cap_assign = nc;
    }
    for (int i = 0; i < h->k; i++)
        if (kmer[i] == 0) return SA_EINVAL;
    h->assign_event[h->n_assign] = event_mean;
    memcpy(h->assign_kmer + h->n_assign * h->k, kmer, (size_t) h->k);
    h->n_assign++;
    return SA_OK;
}

/* continuousPairHmm_writeToFile (:353-407) / hdpHmm_writeToFile (:571-628).  hmmContinuous_checkTransitions: a NaN among the
 * transitions leaves an empty file behind (the caller opened it) */
int sa_hmm_write(const sa_hmm_t *h, const char *path) {
    if (!h || !path) return SA_EINVAL;
    FILE *fh = fopen(path, "w");
    if (!fh) return SA_EIO;
    for (int i = 0; i < 9; i++)
        if (isnan(h->transitions[i])) {
            fprintf(stderr, "GOT NaN TRANS\n");
            fclose(fh);
            return SA_OK;
        }
    fprintf(fh, "%d\t%d\t%s\t%d\t\n", 3, h->n_alpha, h->alphabet, h->k);
    for (int i = 0; i < 9; i++) fprintf(fh, "%f\t", h->transitions[i]);
    fprintf(fh, "%f\n", h->likelihood);
    for (int64_t i = 0; i < h->n_kmers * 5; i++) fprintf(fh, "%lf\t", h->event_model[i]);
    fprintf(fh, "\n");
    if (h->type == SA_HMM_GAUSSIAN) {
        for (int64_t i = 0; i < h->n_kmers * 2; i++) fprintf(fh, "%lf\t", h->event_expectations[i]);
        fprintf(fh, "\n");
        for (int64_t i = 0; i < h->n_kmers; i++) fprintf(fh, "%lf\t", h->posteriors[i]);
        fprintf(fh, "\n");
        for (int64_t i = 0; i < h->n_kmers; i++) fprintf(fh, "%d\t", (int) h->observed[i]);
        fprintf(fh, "\n");
    } else {
        for (int64_t i = 0; i < h->n_assign; i++) fprintf(fh, "%lf\t", h->assign_event[i]);
        fprintf(fh, "\n");
        for (int64_t i = 0; i < h->n_assign; i++) {
            fwrite(h->assign_kmer + i * h->k, 1, (size_t) h->k, fh);
            fputc('\t', fh);
        }
        fprintf(fh, "\n");
    }
    return fclose(fh) == 0 ? SA_OK : SA_EIO;
}

static int parse_double(const char *s, double *out) {
    char *end = NULL;
    *out = strtod(s, &end);
    return end != s && *end == 0;
}

/* continuousPairHmm_loadFromFile (:409-507): header, transitions + likelihood, event model -- the three lines it reads (the
 * expectation, posterior and mask lines of the file are not read back by the reference either: a loaded Hmm starts with empty
 * accumulators at the pseudocounts).  hdpHmm_loadFromFile (:630-785): the same three lines, then the assignment events and
 * k-mers (the reference parses those two lines only when it is handed a NanoporeHDP to pass them to, :722-780; here they are
 * always kept and sa_hdp_state_pass_data is the call that hands them on).  Where the reference aborts the process: SA_EIO. */
int sa_hmm_load(sa_hmm_t **out, const char *path, int type, double transitions_pseudocount, double emissions_pseudocount) {
    if (!out || !path || (type != SA_HMM_GAUSSIAN && type != SA_HMM_HDP)) return SA_EINVAL;
    FILE *f = fopen(path, "r");
    if (!f) return SA_EIO;
    sa_hmm_t *h = NULL;
    char *line = NULL, **tok = NULL;
    int rc = SA_EIO;
    int64_t n;
    /* line 0: stateNumber alphabetSize alphabet kmerLength */
    line = sa_read_line(f);
    if (!line) goto done;
    n = sa_split_ws(line, &tok);
    if (n != 4) goto done;
    {
        char *e1, *e2, *e3;
        const long long states = strtoll(tok[0], &e1, 10), na = strtoll(tok[1], &e2, 10), k = strtoll(tok[3], &e3, 10);
        if (e1 == tok[0] || e2 == tok[1] || e3 == tok[3] || states != 3 || na < 1 || na > 60 || k < 1 || k > 12 ||
            (long long) strlen(tok[2]) != na)
            goto done;
        double total = 1.0;
        for (int i = 0; i < k; i++) total *= (double) na;
        if (total > 2e8) goto done;
        /* (hdpHmm_loadFromFile constructs with a transition pseudocount of 0: :658) */
        h = hmm_new(type, (int) na, tok[2], (int) k, type == SA_HMM_HDP ? 0.0 : transitions_pseudocount, emissions_pseudocount);
        if (!h) { rc = SA_ENOMEM; goto done; }
    }
    free(tok); tok = NULL; free(line);
    /* line 1: nine transitions and the likelihood */
    line = sa_read_line(f);
    if (!line) goto done;
    n = sa_split_ws(line, &tok);
    if (n != 10) goto done;
    for (int i = 0; i < 9; i++)
        if (!parse_double(tok[i], &h->transitions[i])) goto done;
    if (!parse_double(tok[9], &h->likelihood)) goto done;
    free(tok); tok = NULL; free(line);
    /* line 2: the event model */
    line = sa_read_line(f);
    if (!line) goto done;
    n = sa_split_ws(line, &tok);
    if (n != h->n_kmers * 5) goto done;
    for (int64_t i = 0; i < n; i++)
        if (!parse_double(tok[i], &h->event_model[i])) goto done;
    h->has_model = 1;
    free(tok); tok = NULL; free(line); line = NULL;
    if (type == SA_HMM_HDP) {
        /* lines 3, 4: assignment events, assignment k-mers (both may be empty) */
        line = sa_read_line(f);
        if (line) {
            n = sa_split_ws(line, &tok);
            const int64_t na = n;
            double *ev = malloc(sizeof(double) * (size_t) (na > 0 ? na : 1));
            if (!ev) { rc = SA_ENOMEM; goto done; }
            for (int64_t i = 0; i < na; i++)
                if (!parse_double(tok[i], &ev[i])) { free(ev); goto done; }
            free(tok); tok = NULL; free(line);
            line = sa_read_line(f);
            n = line ? sa_split_ws(line, &tok) : 0;
            if (n != na) { free(ev); goto done; }   /* "Incorrect number of events" (:756-759) */
            for (int64_t i = 0; i < na; i++) {
                if ((int) strlen(tok[i]) != h->k || sa_hmm_add_assignment(h, tok[i], ev[i]) != SA_OK) { free(ev); goto done; }
            }
            free(ev);
        }
    }
    rc = SA_OK;
done:
    free(tok);
    free(line);
    fclose(f);
    if (rc != SA_OK) { hmm_free(h); return rc; }
    *out = h;
    return SA_OK;
}

/* continuousPairHmm_normalize (:282-308): hmmDiscrete_normalizeTransitions (impl/discreteHmm.c:125-137), then the event model of
 * every OBSERVED k-mer from its expectations (an unobserved one keeps what it had; a zero keeps the previous value).  An HdpHmm
 * has transitions only (the reference's tests normalise it with hmmDiscrete_normalizeTransitions, tests/stateMachineTests.c:1212). */
int sa_hmm_normalize(sa_hmm_t *h) {
    if (!h) return SA_EINVAL;
    for (int from = 0; from < 3; from++) {
        double total = 0.0;
        for (int to = 0; to < 3; to++) total += h->transitions[from * 3 + to];
        for (int to = 0; to < 3; to++) h->transitions[from * 3 + to] = h->transitions[from * 3 + to] / total;
    }
    if (h->type != SA_HMM_GAUSSIAN) return SA_OK;
    for (int64_t i = 0; i < h->n_kmers; i++) {
        if (!h->observed[i]) continue;
        const double sigma = h->posteriors[i];
        const double u_k = h->event_expectations[2 * i] / sigma;
        const double o_k = sqrt(h->event_expectations[2 * i + 1] / sigma);
        double *em = h->event_model + 5 * i;
        em[0] = u_k == 0.0 ? em[0] : u_k;
        em[1] = o_k == 0.0 ? em[1] : o_k;
    }
    return SA_OK;
}

/* The M-step: continuousPairHmm_loadTransitionsIntoStateMachine (:320-338) and, for a ContinuousPairHmm,
 * continuousPairHmm_loadEmissionsIntoStateMachine (:340-351: level mean and sd of every k-mer; the gapY table follows as 1.75 x
 * the sd -- see the head of this file for the two deviations).  The model's other columns (noise) stay. */
int sa_hmm_load_into_model(sa_model_t *m, const sa_hmm_t *h) {
    if (!m || !h || m->n_kmers != h->n_kmers || m->n_alpha != h->n_alpha || m->k != h->k ||
        memcmp(m->alphabet, h->alphabet, (size_t) m->n_alpha) != 0)
        return SA_EINVAL;
    const double *T = h->transitions;
    m->t_mm = log(T[0]); m->t_mx = log(T[1]); m->t_my = log(T[2]);
    m->t_xm = log(T[3]); m->t_xx = log(T[4]);
    m->t_ym = log(T[6]); m->t_yy = log(T[8]);
    if (h->type == SA_HMM_GAUSSIAN && !m->hdp) {
        if (!h->has_model) return SA_ESTATE;
        for (int64_t i = 0; i < m->n_kmers; i++) {
            m->table5[5 * i] = h->event_model[5 * i];
            m->table5[5 * i + 1] = h->event_model[5 * i + 1];
        }
    }
    m->uid = sa_model_next_uid();   /* (what the library remembers per model -- candidate capacity -- starts over) */
    return SA_OK;
}

/* HMM.add_expectations_file (src/signalalign/hiddenMarkovModel.py:424-486), the accumulating reader of trainModels.py: a read's
 * .expectations file ADDED to this object -- likelihood and transitions (line 1), the event-model line only checked for its
 * length (line 2), and for a ContinuousPairHmm file the emission expectations, the k-mer posteriors and the observed mask
 * (lines 3-5: summed, summed, or-ed); for an HdpHmm file the assignments of lines 3-4 appended.  An empty file (what a NaN
 * transition leaves behind) or a malformed one changes nothing: SA_EIO. */
int sa_hmm_add_expectations_file(sa_hmm_t *h, const char *path) {
    if (!h || !path) return SA_EINVAL;
    FILE *f = fopen(path, "r");
    if (!f) return SA_EIO;
    char *line[6] = {NULL, NULL, NULL, NULL, NULL, NULL}, **tok[6] = {NULL, NULL, NULL, NULL, NULL, NULL};
    int64_t n[6] = {0, 0, 0, 0, 0, 0};
    const int want = h->type == SA_HMM_GAUSSIAN ? 6 : 5;
    int rc = SA_EIO, got = 0;
    for (; got < want; got++) {
        line[got] = sa_read_line(f);
        if (!line[got]) break;
        n[got] = sa_split_ws(line[got], &tok[got]);
    }
    fclose(f);
    double t[10], *ex = NULL, *po = NULL, *ev = NULL;
    if (got < want || n[0] != 4 || n[1] != 10 || n[2] != 5 * h->n_kmers) goto done;
    if (atoll(tok[0][0]) != 3 || atoll(tok[0][1]) != h->n_alpha || atoll(tok[0][3]) != h->k ||
        strlen(tok[0][2]) != (size_t) h->n_alpha)   /* check_header_line */
        goto done;
    for (int i = 0; i < 10; i++)
        if (!parse_double(tok[1][i], &t[i])) goto done;
    if (h->type == SA_HMM_GAUSSIAN) {
        if (n[3] != 2 * h->n_kmers || n[4] != h->n_kmers || n[5] != h->n_kmers) goto done;
        ex = malloc(sizeof(double) * 2 * (size_t) h->n_kmers);
        po = malloc(sizeof(double) * (size_t) h->n_kmers);
        if (!ex || !po) { rc = SA_ENOMEM; goto done; }
        for (int64_t i = 0; i < 2 * h->n_kmers; i++)
            if (!parse_double(tok[3][i], &ex[i])) goto done;
        for (int64_t i = 0; i < h->n_kmers; i++)
            if (!parse_double(tok[4][i], &po[i])) goto done;
        for (int64_t i = 0; i < h->n_kmers; i++)
            if (strcmp(tok[5][i], "0") != 0 && strcmp(tok[5][i], "1") != 0) goto done;
        for (int64_t i = 0; i < 2 * h->n_kmers; i++) h->event_expectations[i] += ex[i];
        for (int64_t i = 0; i < h->n_kmers; i++) h->posteriors[i] += po[i];
        for (int64_t i = 0; i < h->n_kmers; i++) h->observed[i] = h->observed[i] || tok[5][i][0] == '1';
    } else {
        if (n[3] != n[4]) goto done;
        ev = malloc(sizeof(double) * (size_t) (n[3] > 0 ? n[3] : 1));
        if (!ev) { rc = SA_ENOMEM; goto done; }
        for (int64_t i = 0; i < n[3]; i++)
            if (!parse_double(tok[3][i], &ev[i]) || (int) strlen(tok[4][i]) != h->k) goto done;
        for (int64_t i = 0; i < n[3]; i++) {
            rc = sa_hmm_add_assignment(h, tok[4][i], ev[i]);
            if (rc != SA_OK) goto done;
        }
        rc = SA_EIO;
    }
    for (int i = 0; i < 9; i++) h->transitions[i] += t[i];
    h->likelihood += t[9];
    rc = SA_OK;
done:
    free(ex); free(po); free(ev);
    for (int i = 0; i < 6; i++) { free(tok[i]); free(line[i]); }
    return rc;
}

int sa_model_transitions10(const sa_model_t *m, double *out10) {
    if (!m || !out10) return SA_EINVAL;
    const double v[10] = {exp(m->t_mm), exp(m->t_mx), exp(m->t_my), exp(m->t_xm), exp(m->t_xx), 0.0, exp(m->t_ym), 0.0, exp(m->t_yy), 0.0};
    memcpy(out10, v, sizeof(v));
    return SA_OK;
}
