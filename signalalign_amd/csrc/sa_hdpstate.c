/* The whole state of a serialised NanoporeHDP (.nhdp), for the deterministic pieces of the HDP rebuild:
 *
 *   sa_hdp_state_load      deserialize_nhdp + deserialize_hdp            impl/nanopore_hdp.c:1088-1115, impl/hdp.c:3052-3322
 *   sa_hdp_state_write     serialize_nhdp + serialize_hdp                impl/nanopore_hdp.c:1077-1086, impl/hdp.c:2868-3050
 *   sa_hdp_state_weights   cache_base_factor_weight / cache_prior_contribution   impl/hdp.c:2001-2044 (take_distr_sample :2067-2092)
 *
 * The alignment path reads a slice of the same file (sa_io.c:nhdp_read: grid, parents, observed marks, densities, slopes); this
 * file keeps everything -- data, DP assignments, base parameters, gamma vectors, the factor tree with the base factors' cached
 * normal-inverse-gamma parameters -- and writes it back in the reference's format ("%.17lg", tabs, one factor per line in
 * tree order), so that a file the reference wrote goes through load + write byte for byte (tests/test_host_hdp_state.py).
 * The Gibbs sweep (sample_dp_factors, RNG-driven) is NOT here: parity unpinned, host-side, out of this round's scope.
 * Host code only; the grid evaluation and the spline slopes run on the GPU (sa_hdpgrid.hip). */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "signalalign_hip.h"
#include "sa_hdpstate.h"
#include "sa_io.h"

void sa_hdp_state_free(sa_hdp_state_t *s) {
    if (!s) return;
    free(s->data); free(s->data_dp); free(s->gamma); free(s->gamma_alpha); free(s->gamma_beta); free(s->w_aux); free(s->s_aux);
    free(s->dp_parent); free(s->dp_num_factor_children); free(s->dp_depth); free(s->observed); free(s->row_of_dp);
    free(s->post); free(s->slope); free(s->has_post); free(s->has_slope); free(s->grid);
    free(s->f_type); free(s->f_parent); free(s->f_ref); free(s->f_params); free(s->f_n_children);
    free(s);
}

static int64_t *parse_int64s(char *line, int64_t *n_out) {
    char **tok;
    int64_t n = sa_split_ws(line, &tok);
    if (n < 0) return NULL;
    int64_t *v = malloc(sizeof(int64_t) * (size_t) (n > 0 ? n : 1));
    for (int64_t i = 0; v && i < n; i++) v[i] = strtoll(tok[i], NULL, 10);
    free(tok);
    *n_out = n;
    return v;
}
static double *parse_f64s(char *line, int64_t *n_out) {
    char **tok;
    int64_t n = sa_split_ws(line, &tok);
    if (n < 0) return NULL;
    double *v = malloc(sizeof(double) * (size_t) (n > 0 ? n : 1));
    for (int64_t i = 0; v && i < n; i++) v[i] = strtod(tok[i], NULL);   /* (sscanf "%lf" in the reference: the same conversion) */
    free(tok);
    *n_out = n;
    return v;
}

/* linspace, impl/hdp_math_utils.c:497-510 */
static double *grid_linspace(double start, double stop, int64_t length) {
    double *lin = malloc(sizeof(double) * (size_t) length);
    if (!lin) return NULL;
    const int64_t n = length - 1;
    const double dx = (stop - start) / ((double) n);
    for (int64_t i = 0; i < n; i++) lin[i] = start + (double) i * dx;
    lin[n] = stop;
    return lin;
}

int sa_hdp_state_load(sa_hdp_state_t **out, const char *path) {
    if (!out || !path) return SA_EINVAL;
    *out = NULL;
    FILE *f = fopen(path, "r");
    if (!f) return SA_EIO;
    sa_hdp_state_t *s = calloc(1, sizeof(*s));
    if (!s) { fclose(f); return SA_ENOMEM; }
    char *ln = NULL;
    int rc = SA_EIO;
    int64_t n = 0;
#define NEXT() do { free(ln); ln = sa_read_line(f); if (!ln) goto bad; } while (0)
    NEXT(); s->alphabet_size = strtoll(ln, NULL, 10);
    NEXT(); if (sscanf(ln, "%63s", s->alphabet) != 1) goto bad;
    NEXT(); s->kmer_length = strtoll(ln, NULL, 10);
    NEXT(); s->splines_finalized = strtol(ln, NULL, 10) != 0;
    NEXT(); s->has_data = strtol(ln, NULL, 10) != 0;
    NEXT(); s->sample_gamma = strtol(ln, NULL, 10) != 0;
    NEXT(); s->num_dps = strtoll(ln, NULL, 10);
    if (s->num_dps <= 0 || s->num_dps > ((int64_t) 1 << 31) || s->alphabet_size < 1 || s->alphabet_size > 60 ||
        (int64_t) strlen(s->alphabet) != s->alphabet_size || s->kmer_length < 1 || s->kmer_length > 12)
        goto bad;
    {   /* every model constructor of impl/nanopore_hdp.c:413-1060 lays out one leaf DP per k-mer plus internal DPs that group
         * k-mers (at most one per k-mer prefix / multiset and the base): a count far beyond that is a malformed header, refused
         * before seven arrays of that size are allocated */
        double leaves = 1.0;
        for (int64_t i = 0; i < s->kmer_length; i++) leaves *= (double) s->alphabet_size;
        /* (sa_hdp_state_new_tree writes a plain DP tree under the placeholder header "A", k = 1 -- any number of processes:
         * bounded by what a file of that many lines could hold at all, checked against the arrays below) */
        const int placeholder = s->alphabet_size == 1 && s->kmer_length == 1;
        if (!placeholder && (double) s->num_dps > 3.0 * leaves + 64.0) goto bad;
        if (placeholder && s->num_dps > ((int64_t) 1 << 24)) goto bad;
    }
    if (s->has_data) {
        NEXT(); s->data = parse_f64s(ln, &s->n_data);
        NEXT(); s->data_dp = parse_int64s(ln, &n);
        if (!s->data || !s->data_dp || n != s->n_data || n < 1) goto bad;
        for (int64_t i = 0; i < n; i++)
            if (s->data_dp[i] < 0 || s->data_dp[i] >= s->num_dps) goto bad;
    }
    NEXT(); if (sscanf(ln, "%lg %lg %lg %lg", &s->mu, &s->nu, &s->alpha, &s->beta) != 4) goto bad;
    NEXT();
    {
        long long gl = 0;
        if (sscanf(ln, "%lg %lg %lld", &s->grid_start, &s->grid_stop, &gl) != 3 || gl < 2 || gl > (1 << 24) ||
            !(s->grid_start < s->grid_stop))
            goto bad;
        s->grid_length = gl;
    }
    NEXT(); s->gamma = parse_f64s(ln, &s->depth);
    if (!s->gamma || s->depth < 1 || s->depth > 64) goto bad;
    if (s->sample_gamma) {
        NEXT(); s->gamma_alpha = parse_f64s(ln, &n); if (!s->gamma_alpha || n != s->depth) goto bad;
        NEXT(); s->gamma_beta = parse_f64s(ln, &n);  if (!s->gamma_beta || n != s->depth) goto bad;
        NEXT(); s->w_aux = parse_f64s(ln, &n);       if (!s->w_aux || n != s->num_dps) goto bad;
        NEXT(); s->s_aux = parse_int64s(ln, &n);     if (!s->s_aux || n != s->num_dps) goto bad;
    }
    s->dp_parent = malloc(sizeof(int64_t) * (size_t) s->num_dps);
    s->dp_num_factor_children = malloc(sizeof(int64_t) * (size_t) s->num_dps);
    s->dp_depth = malloc(sizeof(int64_t) * (size_t) s->num_dps);
    s->observed = calloc((size_t) s->num_dps, 1);
    s->row_of_dp = malloc(sizeof(int64_t) * (size_t) s->num_dps);
    s->has_post = calloc((size_t) s->num_dps, 1);
    s->has_slope = calloc((size_t) s->num_dps, 1);
    if (!s->dp_parent || !s->dp_num_factor_children || !s->dp_depth || !s->observed || !s->row_of_dp || !s->has_post || !s->has_slope) {
        rc = SA_ENOMEM;
        goto bad;
    }
    s->base_dp = -1;
    for (int64_t id = 0; id < s->num_dps; id++) {
        NEXT();
        long long pa = -1, nc = 0;
        if (ln[0] == '-') {
            if (sscanf(ln, "- %lld", &nc) != 1 || s->base_dp >= 0) goto bad;   /* one root (establish_base_dp, impl/hdp.c) */
            s->base_dp = id;
        } else if (sscanf(ln, "%lld %lld", &pa, &nc) != 2 || pa < 0 || pa >= s->num_dps || pa == id) {
            goto bad;
        }
        if (nc < 0) goto bad;   /* (gamma + num_factor_children is a denominator of the sample weights) */
        s->dp_parent[id] = pa;
        s->dp_num_factor_children[id] = nc;
    }
    if (s->base_dp < 0) goto bad;
    /* depth of every DP (verify_tree_depth, impl/hdp.c:1086-1104: gamma by depth); a parent chain longer than `depth` is a cycle or
     * a tree the gamma vector does not cover */
    for (int64_t id = 0; id < s->num_dps; id++) {
        int64_t d = 0;
        for (int64_t a = id; s->dp_parent[a] >= 0; a = s->dp_parent[a])
            if (++d >= s->depth) goto bad;
        s->dp_depth[id] = d;
    }
    s->grid = grid_linspace(s->grid_start, s->grid_stop, s->grid_length);
    if (!s->grid) { rc = SA_ENOMEM; goto bad; }
    if (s->has_data) {   /* mark_observed_dps, impl/hdp.c:1132-1160: every DP that holds data and all its ancestors */
        for (int64_t i = 0; i < s->n_data; i++)
            for (int64_t a = s->data_dp[i]; a >= 0 && !s->observed[a]; a = s->dp_parent[a]) s->observed[a] = 1;
    }
    for (int64_t id = 0; id < s->num_dps; id++) s->row_of_dp[id] = s->observed[id] ? s->n_observed++ : -1;
    const size_t plane = (size_t) (s->n_observed > 0 ? s->n_observed : 1) * (size_t) s->grid_length;
    s->post = calloc(plane, sizeof(double));
    s->slope = calloc(plane, sizeof(double));
    if (!s->post || !s->slope) { rc = SA_ENOMEM; goto bad; }
    for (int pass = 0; pass < 2; pass++) {   /* posterior predictives (with data), then spline slopes (when finalised) */
        if (pass == 0 ? !s->has_data : !s->splines_finalized) continue;
        for (int64_t id = 0; id < s->num_dps; id++) {
            NEXT();
            double *v = parse_f64s(ln, &n);
            if (!v) { rc = SA_ENOMEM; goto bad; }
            if (n != 0) {
                /* a row belongs to an observed DP and is complete */
                if (n != s->grid_length || !s->observed[id]) { free(v); goto bad; }
                memcpy((pass == 0 ? s->post : s->slope) + (size_t) s->row_of_dp[id] * (size_t) s->grid_length, v,
                       sizeof(double) * (size_t) n);
                (pass == 0 ? s->has_post : s->has_slope)[id] = 1;
            }
            free(v);
        }
    }
    if (s->has_data) {   /* the factor tree, one factor per line in tree order (serialize_factor_tree_internal, impl/hdp.c:2868-2917) */
        int64_t cap = 1024;
        s->f_type = malloc(sizeof(int64_t) * (size_t) cap);
        s->f_parent = malloc(sizeof(int64_t) * (size_t) cap);
        s->f_ref = malloc(sizeof(int64_t) * (size_t) cap);
        s->f_params = malloc(sizeof(double) * 5 * (size_t) cap);
        if (!s->f_type || !s->f_parent || !s->f_ref || !s->f_params) { rc = SA_ENOMEM; goto bad; }
        for (;;) {
            free(ln);
            ln = sa_read_line(f);
            if (!ln) break;
            char **tok;
            const int64_t nt = sa_split_ws(ln, &tok);
            if (nt == 0) { free(tok); continue; }
            if (nt != 3) { free(tok); goto bad; }
            if (s->n_factors == cap) {
                cap *= 2;
                int64_t *a = realloc(s->f_type, sizeof(int64_t) * (size_t) cap); if (a) s->f_type = a;
                int64_t *b = realloc(s->f_parent, sizeof(int64_t) * (size_t) cap); if (b) s->f_parent = b;
                int64_t *c = realloc(s->f_ref, sizeof(int64_t) * (size_t) cap); if (c) s->f_ref = c;
                double *d = realloc(s->f_params, sizeof(double) * 5 * (size_t) cap); if (d) s->f_params = d;
                if (!a || !b || !c || !d) { free(tok); rc = SA_ENOMEM; goto bad; }
            }
            const int64_t id = s->n_factors;
            const int64_t type = strtoll(tok[0], NULL, 10);
            int64_t parent = -1, ref = -1;
            double *pp = s->f_params + 5 * id;
            for (int i = 0; i < 5; i++) pp[i] = 0.0;
            int ok = type >= 0 && type <= 2;
            if (ok && type == 0) {
                ok = tok[1][0] == '-';
                char *q = tok[2];
                for (int i = 0; ok && i < 5; i++) {
                    char *e = NULL;
                    pp[i] = strtod(q, &e);
                    ok = e != q && (i == 4 ? *e == 0 : *e == ';');
                    q = e + 1;
                }
                ref = s->base_dp;
                s->n_base_factors++;
            } else if (ok) {
                parent = strtoll(tok[1], NULL, 10);
                ref = strtoll(tok[2], NULL, 10);
                ok = tok[1][0] != '-' && parent >= 0 && parent < id && s->f_type[parent] != 2 &&
                     ref >= 0 && ref < (type == 1 ? s->num_dps : s->n_data);
                /* a middle factor sits in a child DP of its parent's DP; a data point under a factor of its own DP */
                if (ok && type == 1) ok = s->dp_parent[ref] == s->f_ref[parent];
                if (ok && type == 2) ok = s->data_dp[ref] == s->f_ref[parent];
            }
            free(tok);
            if (!ok) goto bad;
            s->f_type[id] = type; s->f_parent[id] = parent; s->f_ref[id] = ref;
            s->n_factors++;
        }
        s->f_n_children = calloc((size_t) (s->n_factors > 0 ? s->n_factors : 1), sizeof(int64_t));
        if (!s->f_n_children) { rc = SA_ENOMEM; goto bad; }
        for (int64_t i = 0; i < s->n_factors; i++)
            if (s->f_parent[i] >= 0) s->f_n_children[s->f_parent[i]]++;
        /* num_factor_children of a DP counts the factors (middle factors, data points) whose PARENT factor sits in that DP
         * (impl/hdp.c:346, :1368, :1413, :1682-1720): the two halves of the file must agree */
        int64_t *cnt = calloc((size_t) s->num_dps, sizeof(int64_t));
        if (!cnt) { rc = SA_ENOMEM; goto bad; }
        for (int64_t i = 0; i < s->n_factors; i++)
            if (s->f_parent[i] >= 0) cnt[s->f_ref[s->f_parent[i]]]++;
        int same = 1;
        for (int64_t id = 0; id < s->num_dps && same; id++) same = cnt[id] == s->dp_num_factor_children[id];
        free(cnt);
        if (!same) goto bad;
    }
    free(ln);
    fclose(f);
    *out = s;
    return SA_OK;
bad:
    free(ln);
    fclose(f);
    sa_hdp_state_free(s);
    return rc;
#undef NEXT
}

int sa_hdp_state_write(const sa_hdp_state_t *s, const char *path) {
    if (!s || !path) return SA_EINVAL;
    FILE *o = fopen(path, "w");
    if (!o) return SA_EIO;
    /* serialize_nhdp, impl/nanopore_hdp.c:1077-1086 */
    fprintf(o, "%lld\n%s\n%lld\n", (long long) s->alphabet_size, s->alphabet, (long long) s->kmer_length);
    /* serialize_hdp, impl/hdp.c:2919-3050 */
    fprintf(o, "%d\n%d\n%d\n%lld\n", s->splines_finalized, s->has_data, s->sample_gamma, (long long) s->num_dps);
    if (s->has_data) {
        for (int64_t i = 0; i < s->n_data - 1; i++) fprintf(o, "%.17lg\t", s->data[i]);
        fprintf(o, "%.17lg\n", s->data[s->n_data - 1]);
        for (int64_t i = 0; i < s->n_data - 1; i++) fprintf(o, "%lld\t", (long long) s->data_dp[i]);
        fprintf(o, "%lld\n", (long long) s->data_dp[s->n_data - 1]);
    }
    fprintf(o, "%.17lg\t%.17lg\t%.17lg\t%.17lg\n", s->mu, s->nu, s->alpha, s->beta);
    fprintf(o, "%.17lg\t%.17lg\t%lld\n", s->grid[0], s->grid[s->grid_length - 1], (long long) s->grid_length);
    for (int64_t i = 0; i < s->depth - 1; i++) fprintf(o, "%.17lg\t", s->gamma[i]);
    fprintf(o, "%.17lg\n", s->gamma[s->depth - 1]);
    if (s->sample_gamma) {
        for (int64_t i = 0; i < s->depth - 1; i++) fprintf(o, "%.17lg\t", s->gamma_alpha[i]);
        fprintf(o, "%.17lg\n", s->gamma_alpha[s->depth - 1]);
        for (int64_t i = 0; i < s->depth - 1; i++) fprintf(o, "%.17lg\t", s->gamma_beta[i]);
        fprintf(o, "%.17lg\n", s->gamma_beta[s->depth - 1]);
        for (int64_t i = 0; i < s->num_dps - 1; i++) fprintf(o, "%.17lg\t", s->w_aux[i]);
        fprintf(o, "%.17lg\n", s->w_aux[s->num_dps - 1]);
        for (int64_t i = 0; i < s->num_dps - 1; i++) fprintf(o, "%lld\t", (long long) (s->s_aux[i] != 0));
        fprintf(o, "%lld\n", (long long) (s->s_aux[s->num_dps - 1] != 0));
    }
    for (int64_t id = 0; id < s->num_dps; id++) {
        if (id == s->base_dp) fprintf(o, "-\t%lld\n", (long long) s->dp_num_factor_children[id]);
        else fprintf(o, "%lld\t%lld\n", (long long) s->dp_parent[id], (long long) s->dp_num_factor_children[id]);
    }
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 0 ? !s->has_data : !s->splines_finalized) continue;
        const uint8_t *has = pass == 0 ? s->has_post : s->has_slope;
        const double *pl = pass == 0 ? s->post : s->slope;
        for (int64_t id = 0; id < s->num_dps; id++) {
            if (has[id]) {
                const double *v = pl + (size_t) s->row_of_dp[id] * (size_t) s->grid_length;
                for (int64_t j = 0; j < s->grid_length - 1; j++) fprintf(o, "%.17lg\t", v[j]);
                fprintf(o, "%.17lg", v[s->grid_length - 1]);
            }
            fputc('\n', o);
        }
    }
    if (s->has_data) {
        for (int64_t i = 0; i < s->n_factors; i++) {
            const double *pp = s->f_params + 5 * i;
            if (s->f_type[i] == 0)
                fprintf(o, "0\t-\t%.17lg;%.17lg;%.17lg;%.17lg;%.17lg\n", pp[0], pp[1], pp[2], pp[3], pp[4]);
            else
                fprintf(o, "%lld\t%lld\t%lld\n", (long long) s->f_type[i], (long long) s->f_parent[i], (long long) s->f_ref[i]);
        }
    }
    const int bad = ferror(o);
    if (fclose(o) != 0 || bad) return SA_EIO;
    return SA_OK;
}

int sa_hdp_state_info(const sa_hdp_state_t *s, sa_hdp_state_info_t *info) {
    if (!s || !info) return SA_EINVAL;
    memset(info, 0, sizeof(*info));
    info->num_dps = s->num_dps; info->depth = s->depth; info->grid_length = s->grid_length; info->n_data = s->n_data;
    info->n_factors = s->n_factors; info->n_base_factors = s->n_base_factors; info->n_observed = s->n_observed;
    info->base_dp = s->base_dp; info->alphabet_size = s->alphabet_size; info->kmer_length = s->kmer_length;
    info->mu = s->mu; info->nu = s->nu; info->alpha = s->alpha; info->beta = s->beta;
    info->grid_start = s->grid_start; info->grid_stop = s->grid_stop;
    info->splines_finalized = s->splines_finalized; info->has_data = s->has_data; info->sample_gamma = s->sample_gamma;
    info->data = s->data; info->data_dp = s->data_dp; info->gamma = s->gamma; info->grid = s->grid;
    info->dp_parent = s->dp_parent; info->dp_num_factor_children = s->dp_num_factor_children; info->dp_depth = s->dp_depth;
    info->observed = s->observed; info->row_of_dp = s->row_of_dp; info->post = s->post; info->slope = s->slope;
    info->f_type = s->f_type; info->f_parent = s->f_parent; info->f_ref = s->f_ref; info->f_params = s->f_params;
    info->f_n_children = s->f_n_children;
    return SA_OK;
}

/* Weights of one distribution sample (take_distr_sample, impl/hdp.c:2067-2092): for every base factor F, in tree order, what
 * cache_base_factor_weight(F) leaves in every observed DP's base_factor_wt before push_factor_distr adds wt * pdf_F to the DP's
 * collector -- and at the end the prior's share (cache_prior_contribution(base_dp, 1.0)).  The recursions of the reference are
 * walked here with an explicit stack over the factor tree (children in file order, which is the order the reference's sets had
 * when the file was written) and over the DP tree (children in id order); sums are taken in the reference's order.
 * Output, CSR by observed-DP row: entry = (column, weight); column < n_base_factors names the base factor (in tree order),
 * column == n_base_factors the prior.  Entries of a row are in column order: the order in which the reference adds them. */
typedef struct { int64_t *first, *next; } child_list_t;

int sa_hdp_state_weights(const sa_hdp_state_t *s, int64_t **row_start_out, int64_t **col_out, double **w_out, int64_t *nnz_out) {
    if (!s || !row_start_out || !col_out || !w_out || !nnz_out) return SA_EINVAL;
    if (!s->has_data) return SA_ESTATE;
    const int64_t nd = s->num_dps, nf = s->n_factors, nb = s->n_base_factors, nrow = s->n_observed;
    int rc = SA_ENOMEM;
    /* child lists: DPs by parent (ascending id), factors by parent (file order) */
    int64_t *dp_first = malloc(sizeof(int64_t) * (size_t) (nd + 1)), *dp_child = malloc(sizeof(int64_t) * (size_t) (nd > 0 ? nd : 1));
    int64_t *f_first = malloc(sizeof(int64_t) * (size_t) (nf + 1)), *f_child = malloc(sizeof(int64_t) * (size_t) (nf > 0 ? nf : 1));
    double *acc = calloc((size_t) nd, sizeof(double));           /* base_factor_wt of every DP */
    int64_t *touched = malloc(sizeof(int64_t) * (size_t) nd);    /* DPs with acc != 0 for the current column, in first-touch order */
    uint8_t *is_touched = calloc((size_t) nd, 1);
    int64_t *stack = malloc(sizeof(int64_t) * (size_t) (nf + nd + 2));
    double *stack_w = malloc(sizeof(double) * (size_t) (nf + nd + 2));
    int64_t cap = 4 * (nrow + nb) + 16, nnz = 0;
    int64_t *e_row = malloc(sizeof(int64_t) * (size_t) cap), *e_col = malloc(sizeof(int64_t) * (size_t) cap);
    double *e_w = malloc(sizeof(double) * (size_t) cap);
    int64_t *row_start = calloc((size_t) (nrow + 2), sizeof(int64_t)), *col = NULL;
    double *w = NULL;
    if (!dp_first || !dp_child || !f_first || !f_child || !acc || !touched || !is_touched || !stack || !stack_w || !e_row || !e_col ||
        !e_w || !row_start)
        goto done;
    {   /* counting sort of children */
        for (int64_t i = 0; i <= nd; i++) dp_first[i] = 0;
        for (int64_t i = 0; i < nd; i++) if (s->dp_parent[i] >= 0) dp_first[s->dp_parent[i] + 1]++;
        for (int64_t i = 0; i < nd; i++) dp_first[i + 1] += dp_first[i];
        int64_t *fill = calloc((size_t) (nd > nf ? nd : nf) + 1, sizeof(int64_t));
        if (!fill) goto done;
        for (int64_t i = 0; i < nd; i++) if (s->dp_parent[i] >= 0) dp_child[dp_first[s->dp_parent[i]] + fill[s->dp_parent[i]]++] = i;
        memset(fill, 0, sizeof(int64_t) * ((size_t) (nd > nf ? nd : nf) + 1));
        for (int64_t i = 0; i <= nf; i++) f_first[i] = 0;
        for (int64_t i = 0; i < nf; i++) if (s->f_parent[i] >= 0) f_first[s->f_parent[i] + 1]++;
        for (int64_t i = 0; i < nf; i++) f_first[i + 1] += f_first[i];
        for (int64_t i = 0; i < nf; i++) if (s->f_parent[i] >= 0) f_child[f_first[s->f_parent[i]] + fill[s->f_parent[i]]++] = i;
        free(fill);
    }
#define TOUCH(d_) do { if (!is_touched[d_]) { is_touched[d_] = 1; touched[n_touched++] = (d_); } } while (0)
#define GAMMA(d_) (s->gamma[s->dp_depth[d_]])
    /* cache_prior_contribution(dp, parent_prior_prod), impl/hdp.c:2001-2017: depth-first, children in order */
#define PRIOR_WALK(d0_, pp0_) do {                                                                                  \
        int64_t sp_ = 0;                                                                                            \
        stack[sp_] = (d0_); stack_w[sp_++] = (pp0_);                                                                \
        while (sp_ > 0) {                                                                                           \
            const int64_t d_ = stack[--sp_];                                                                        \
            const double pp_ = stack_w[sp_];                                                                        \
            if (!s->observed[d_]) continue;                                                                         \
            const double g_ = GAMMA(d_);                                                                            \
            const double prod_ = (g_ / (g_ + (double) s->dp_num_factor_children[d_])) * pp_;                       \
            acc[d_] += prod_;                                                                                       \
            TOUCH(d_);                                                                                              \
            for (int64_t c_ = dp_first[d_ + 1] - 1; c_ >= dp_first[d_]; c_--) {   /* pushed in reverse: popped in order */ \
                stack[sp_] = dp_child[c_]; stack_w[sp_++] = prod_;                                                  \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)
    int64_t column = 0;
    for (int64_t F = 0; F <= nf; F++) {
        int64_t n_touched = 0;
        if (F < nf) {
            if (s->f_type[F] != 0) continue;
            /* cache_base_factor_weight(F), impl/hdp.c:2019-2044: the factor, then its child factors (recursively), then the prior
             * contributions to the child DPs of the factor's DP.  Explicit stack of (factor, phase). */
            int64_t *fs = malloc(sizeof(int64_t) * 2 * (size_t) (nf + 1));
            double *fw = malloc(sizeof(double) * (size_t) (nf + 1));
            if (!fs || !fw) { free(fs); free(fw); goto done; }
            int64_t sp = 0;
            fs[0] = F; fs[1] = 0; sp = 1;
            while (sp > 0) {
                const int64_t fc = fs[2 * (sp - 1)], phase = fs[2 * (sp - 1) + 1];
                const int64_t d = s->f_ref[fc];
                const int has_child_dps = dp_first[d + 1] > dp_first[d];
                if (phase == 0) {
                    const double g = GAMMA(d);
                    const double wt = ((double) s->f_n_children[fc]) / (g + (double) s->dp_num_factor_children[d]);
                    acc[d] += wt;
                    TOUCH(d);
                    fw[sp - 1] = wt;
                    if (!has_child_dps) { sp--; continue; }
                    fs[2 * (sp - 1) + 1] = 1;
                    /* child factors, in order: pushed in reverse */
                    for (int64_t c = f_first[fc + 1] - 1; c >= f_first[fc]; c--) {
                        if (s->f_type[f_child[c]] == 2) continue;   /* (data points hang under factors of leaf DPs only) */
                        fs[2 * sp] = f_child[c]; fs[2 * sp + 1] = 0; sp++;
                    }
                } else {
                    const double wt = fw[sp - 1];
                    sp--;
                    for (int64_t c = dp_first[d]; c < dp_first[d + 1]; c++) PRIOR_WALK(dp_child[c], wt);
                }
            }
            free(fs); free(fw);
        } else {
            PRIOR_WALK(s->base_dp, 1.0);
        }
        /* push_factor_distr, impl/hdp.c:2046-2065: every observed DP takes acc * pdf and resets acc */
        for (int64_t t = 0; t < n_touched; t++) {
            const int64_t d = touched[t];
            is_touched[d] = 0;
            if (s->observed[d]) {
                if (nnz == cap) {
                    cap *= 2;
                    int64_t *a = realloc(e_row, sizeof(int64_t) * (size_t) cap); if (a) e_row = a;
                    int64_t *b = realloc(e_col, sizeof(int64_t) * (size_t) cap); if (b) e_col = b;
                    double *c = realloc(e_w, sizeof(double) * (size_t) cap); if (c) e_w = c;
                    if (!a || !b || !c) goto done;
                }
                e_row[nnz] = s->row_of_dp[d]; e_col[nnz] = column; e_w[nnz] = acc[d]; nnz++;
            }
            acc[d] = 0.0;
        }
        column++;
    }
#undef PRIOR_WALK
#undef TOUCH
#undef GAMMA
    /* to CSR (stable: a row's entries stay in column order) */
    col = malloc(sizeof(int64_t) * (size_t) (nnz > 0 ? nnz : 1));
    w = malloc(sizeof(double) * (size_t) (nnz > 0 ? nnz : 1));
    if (!col || !w) goto done;
    for (int64_t i = 0; i < nnz; i++) row_start[e_row[i] + 2]++;
    for (int64_t r = 0; r < nrow; r++) row_start[r + 2] += row_start[r + 1];
    for (int64_t i = 0; i < nnz; i++) {
        const int64_t p = row_start[e_row[i] + 1]++;
        col[p] = e_col[i]; w[p] = e_w[i];
    }
    *row_start_out = row_start; *col_out = col; *w_out = w; *nnz_out = nnz;
    row_start = NULL; col = NULL; w = NULL;
    rc = SA_OK;
done:
    free(dp_first); free(dp_child); free(f_first); free(f_child); free(acc); free(touched); free(is_touched); free(stack); free(stack_w);
    free(e_row); free(e_col); free(e_w); free(row_start); free(col); free(w);
    return rc;
}

int sa_hdp_state_sample_weights(const sa_hdp_state_t *s, int64_t **row_start, int64_t **col, double **w, int64_t *nnz) {
    return sa_hdp_state_weights(s, row_start, col, w, nnz);
}
