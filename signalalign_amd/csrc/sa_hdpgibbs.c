/* sa_hdpgibbs.c -- the HDP rebuild loop, host side: what buildHdpUtil / trainModels.py --hdp do
 * around the deterministic pieces of sa_hdpstate.c / sa_hdpgrid.hip.
 *
 *   sa_hdp_state_new          new_hier_dir_proc[_2] + the NanoporeHDP tree layouts + finalize_hdp_structure
 *                             (impl/hdp.c:879-1015, :1572-1582; impl/nanopore_hdp.c:489-1060: flat, multiset, middle two
 *                             nucleotides, purine composition, group multiset -- one parent rule per leaf k-mer)
 *   sa_hdp_nig_params_from_model   normal_inverse_gamma_params_from_minION (impl/nanopore_hdp.c:122-176) =
 *                             mle_normal_inverse_gamma_params (impl/hdp_math_utils.c:751-810) over a lookup table's level means
 *                             and precisions
 *   sa_hdp_state_pass_data    reset_hdp_data + pass_data_to_hdp -> finalize_data (impl/hdp.c:1591-1660, :1549-1570):
 *                             verify_valid_dp_assignments, mark_observed_dps, init_factors (:1440-1547: one chain of factors per
 *                             observed DP under ONE base factor, whose normal-inverse-gamma parameters take all the data)
 *   sa_hdp_state_pass_assignment_file   update_nhdp_from_alignment_with_filter (impl/nanopore_hdp.c:206-297): the assignments
 *                             (4 columns) or full alignment (15 columns) table, optionally one strand only
 *   sa_hdp_state_gibbs        execute_gibbs_sampling (impl/hdp.c:2486-2549): sweeps over the shuffled Dirichlet processes
 *                             (get_shuffled_dps :2094-2106), sample_dp_factors (:2108-2163), gibbs_factor_iteration (:1993-1998) =
 *                             unassign_from_parent (:1664-1704) + sample_factor (:1794-1991, with the unobserved-factor
 *                             likelihoods :645-798) + assign_to_parent (:1706-1737); the concentration parameters when the HDP
 *                             holds a Gamma prior on them (sample_gamma_params :2165-2300: Escobar & West's auxiliary variables)
 *   sa_hdp_state_finalize     finalize_distributions (:2551-2584)
 *
 * The sweep is a sequential, random-number-driven walk over a pointer tree: it runs HERE, on the host, in plain C.  What is O(observed
 * DPs x grid points x factors) -- every kept sample's take_distr_sample (:2067-2092) and the finalisation -- runs on the GPU
 * (sa_hdpgrid.hip: the collectors stay in HBM for the whole run, a sample uploads its weights and the base factors' parameters).
 *
 * PARITY UNPINNED, by construction: the reference draws from rand() and ranlib (genbet, gengam), iterates its factor sets in the
 * order of a pointer-hashed stSet, and its own tests of this code are properties (tests/hdpTests.c:109-233, tests/nanoporeHdpTests.c:
 * 272-480).  Here one seeded generator (splitmix64 -> xoshiro256**) stands for all of them and factor sets are insertion-ordered
 * lists; the arithmetic of every likelihood, parameter update and weight follows the reference line by line.  tests/
 * test_gpu_hdp_rebuild.py checks the reference's properties.
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "signalalign_hip.h"
#include "sa_hdpstate.h"
#include "sa_io.h"

#define GIBBS_MINUS_INF (-0.5 * DBL_MAX)

/* sa_hdpgrid.hip: the collectors of a sampling run on the device */
typedef struct sa_hdp_sampler sa_hdp_sampler_t;
int sa_hdp_sampler_open(sa_hdp_sampler_t **out, const sa_hdp_state_t *s, int device);
int sa_hdp_sampler_add(sa_hdp_sampler_t *h, const sa_hdp_state_t *s);
int sa_hdp_sampler_finish(sa_hdp_sampler_t *h, double *sum_out);   /* n_observed x grid_length; closes the session */
void sa_hdp_sampler_close(sa_hdp_sampler_t *h);

/* ---------------------------------------------------------------------------------------------------------------------------- */
/* construction                                                                                                                 */
/* ---------------------------------------------------------------------------------------------------------------------------- */
static int64_t ipow(int64_t n, int64_t k) { int64_t v = 1; for (int64_t i = 0; i < k; i++) v *= n; return v; }
static int64_t multiset_number(int64_t n, int64_t k) {   /* ((n k)), impl/nanopore_hdp.c:312-321 */
    int64_t num = 1;
    for (int64_t m = n + k - 1; m >= n; m--) num *= m;
    for (int64_t m = k; m >= 2; m--) num /= m;
    return num;
}
static void word_of(int64_t id, int64_t a, int64_t k, int64_t *w) {
    for (int64_t i = 0; i < k; i++) { w[k - i - 1] = id % a; id /= a; }
}
static void sort_small(int64_t *w, int64_t k) {
    for (int64_t i = 1; i < k; i++)
        for (int64_t j = i; j > 0 && w[j] < w[j - 1]; j--) { int64_t t = w[j]; w[j] = w[j - 1]; w[j - 1] = t; }
}
static int64_t multiset_id_internal(const int64_t *tail, int64_t len, int64_t amin, int64_t a) {   /* :355-372 */
    const int64_t head = tail[0];
    if (len == 1) return head - amin;
    int64_t step = 0;
    for (int64_t i = amin; i < a; i++) {
        if (head > i) step += multiset_number(a - i, len - 1);
        else return step + multiset_id_internal(tail + 1, len - 1, i, a);
    }
    return -1;
}

static double *linspace(double start, double stop, int64_t length) {   /* impl/hdp_math_utils.c:497-510 */
    double *lin = malloc(sizeof(double) * (size_t) length);
    if (!lin) return NULL;
    const int64_t n = length - 1;
    const double dx = (stop - start) / ((double) n);
    for (int64_t i = 0; i < n; i++) lin[i] = start + (double) i * dx;
    lin[n] = stop;
    return lin;
}

/* new_hier_dir_proc / new_hier_dir_proc_2 (impl/hdp.c:879-995) + set_dir_proc_parent for every DP + finalize_hdp_structure
 * (:1572-1582: one base DP, a tree, every leaf at depth - 1): a plain HierarchicalDirichletProcess over any tree */
int sa_hdp_state_new_tree(sa_hdp_state_t **out, int64_t num_dps, int64_t depth, const int64_t *parents, const double *gamma,
                          const double *gamma_alpha, const double *gamma_beta, double grid_start, double grid_stop, int64_t grid_length,
                          double mu, double nu, double alpha, double beta) {
    if (!out || !parents || num_dps < 2 || num_dps > ((int64_t) 1 << 31) || depth < 1 || depth > 64 || grid_length < 2 ||
        grid_length > (1 << 24) || !(grid_start < grid_stop))
        return SA_EINVAL;
    const int prior = gamma == NULL;
    if (prior && (!gamma_alpha || !gamma_beta)) return SA_EINVAL;
    /* new_hier_dir_proc's checks (impl/hdp.c:882-918) */
    if (!(nu > 0.0) || !(beta > 0.0)) return SA_EINVAL;
    if (alpha <= 1.0) alpha = 1.00001;
    for (int64_t i = 0; i < depth; i++) {
        if (!prior && !(gamma[i] > 0.0)) return SA_EINVAL;
        if (prior && (!(gamma_alpha[i] > 0.0) || !(gamma_beta[i] > 0.0))) return SA_EINVAL;
    }
    sa_hdp_state_t *s = calloc(1, sizeof(*s));
    if (!s) return SA_ENOMEM;
    s->alphabet_size = 1; s->kmer_length = 1; s->alphabet[0] = 'A';   /* (a NanoporeHDP header for sa_hdp_state_write: sa_hdp_state_new sets the real one) */
    s->num_dps = num_dps;
    s->depth = depth;
    s->base_dp = -1;
    s->mu = mu; s->nu = nu; s->alpha = alpha; s->beta = beta;
    s->grid_start = grid_start; s->grid_stop = grid_stop; s->grid_length = grid_length;
    s->sample_gamma = prior;
    s->grid = linspace(grid_start, grid_stop, grid_length);
    s->gamma = malloc(sizeof(double) * (size_t) depth);
    s->dp_parent = malloc(sizeof(int64_t) * (size_t) num_dps);
    s->dp_num_factor_children = calloc((size_t) num_dps, sizeof(int64_t));
    s->dp_depth = malloc(sizeof(int64_t) * (size_t) num_dps);
    s->observed = calloc((size_t) num_dps, 1);
    s->row_of_dp = malloc(sizeof(int64_t) * (size_t) num_dps);
    s->has_post = calloc((size_t) num_dps, 1);
    s->has_slope = calloc((size_t) num_dps, 1);
    s->post = calloc(1, sizeof(double));
    s->slope = calloc(1, sizeof(double));
    uint8_t *has_child = calloc((size_t) num_dps, 1);
    if (!s->grid || !s->gamma || !s->dp_parent || !s->dp_num_factor_children || !s->dp_depth || !s->observed || !s->row_of_dp ||
        !s->has_post || !s->has_slope || !s->post || !s->slope || !has_child) {
        free(has_child);
        sa_hdp_state_free(s);
        return SA_ENOMEM;
    }
    int rc = SA_OK;
    if (prior) {   /* new_hier_dir_proc_2 :949-995: gamma starts at the prior's expected value, w = 1, s = false */
        s->gamma_alpha = malloc(sizeof(double) * (size_t) depth);
        s->gamma_beta = malloc(sizeof(double) * (size_t) depth);
        s->w_aux = malloc(sizeof(double) * (size_t) num_dps);
        s->s_aux = calloc((size_t) num_dps, sizeof(int64_t));
        if (!s->gamma_alpha || !s->gamma_beta || !s->w_aux || !s->s_aux) rc = SA_ENOMEM;
        for (int64_t i = 0; rc == SA_OK && i < depth; i++) {
            s->gamma_alpha[i] = gamma_alpha[i]; s->gamma_beta[i] = gamma_beta[i];
            s->gamma[i] = gamma_alpha[i] / gamma_beta[i];
        }
        for (int64_t i = 0; rc == SA_OK && i < num_dps; i++) s->w_aux[i] = 1.0;
    } else {
        for (int64_t i = 0; i < depth; i++) s->gamma[i] = gamma[i];
    }
    /* establish_base_dp, verify_dp_tree, verify_tree_depth (:1017-1104) */
    for (int64_t d = 0; rc == SA_OK && d < num_dps; d++) {
        const int64_t pa = parents[d];
        if (pa < -1 || pa >= num_dps || pa == d) { rc = SA_EINVAL; break; }
        s->dp_parent[d] = pa;
        s->row_of_dp[d] = -1;
        if (pa < 0) {
            if (s->base_dp >= 0) rc = SA_EINVAL;   /* "contains orphaned Dirichlet process" */
            s->base_dp = d;
        } else {
            has_child[pa] = 1;
        }
    }
    if (rc == SA_OK && s->base_dp < 0) rc = SA_EINVAL;
    for (int64_t d = 0; rc == SA_OK && d < num_dps; d++) {
        int64_t dd = 0;
        for (int64_t a = d; s->dp_parent[a] >= 0; a = s->dp_parent[a])
            if (++dd >= depth) { rc = SA_EINVAL; break; }   /* a cycle, or deeper than the gamma vector */
        s->dp_depth[d] = dd;
        if (rc == SA_OK && !has_child[d] && dd != depth - 1) rc = SA_EINVAL;   /* "leaf Dirichlet process at incorrect depth" */
    }
    free(has_child);
    if (rc != SA_OK) { sa_hdp_state_free(s); return rc; }
    *out = s;
    return SA_OK;
}

int sa_hdp_state_new(sa_hdp_state_t **out, int layout, const char *alphabet, int64_t kmer_length, const int64_t *groups,
                     const double *gamma, const double *gamma_alpha, const double *gamma_beta, double grid_start, double grid_stop,
                     int64_t grid_length, double mu, double nu, double alpha, double beta) {
    if (!out || !alphabet || kmer_length < 1 || kmer_length > 12) return SA_EINVAL;
    const int64_t a = (int64_t) strlen(alphabet);
    if (a < 1 || a > 60) return SA_EINVAL;
    const int two_level = layout == SA_HDP_LAYOUT_FLAT;
    if (layout < SA_HDP_LAYOUT_FLAT || layout > SA_HDP_LAYOUT_GROUP_MULTISET) return SA_EINVAL;
    if ((layout == SA_HDP_LAYOUT_COMPOSITION || layout == SA_HDP_LAYOUT_GROUP_MULTISET) && !groups) return SA_EINVAL;
    if (layout == SA_HDP_LAYOUT_MIDDLE_NTS && kmer_length <= 2) return SA_EINVAL;
    const int64_t depth = two_level ? 2 : 3;
    double leaves_d = 1.0;
    for (int64_t i = 0; i < kmer_length; i++) leaves_d *= (double) a;
    if (leaves_d > 2e8) return SA_EINVAL;
    const int64_t leaves = ipow(a, kmer_length);
    /* the alphabet in sorted order, and the letter groups with it (package_nanopore_hdp :34-77, alphabet_sort_groups :788-815) */
    char sorted[64];
    int64_t grp[64];
    memcpy(sorted, alphabet, (size_t) a);
    sorted[a] = 0;
    for (int64_t i = 0; i < a; i++) grp[i] = groups ? groups[i] : 0;
    for (int64_t i = 1; i < a; i++)
        for (int64_t j = i; j > 0 && sorted[j] < sorted[j - 1]; j--) {
            char c = sorted[j]; sorted[j] = sorted[j - 1]; sorted[j - 1] = c;
            int64_t g = grp[j]; grp[j] = grp[j - 1]; grp[j - 1] = g;
        }
    for (int64_t i = 1; i < a; i++)
        if (sorted[i] == sorted[i - 1]) return SA_EINVAL;   /* "Characters of alphabet must be distinct." */
    int64_t n_groups = 0, middle = 0;
    if (groups) {
        for (int64_t i = 0; i < a; i++) {
            if (grp[i] < 0) return SA_EINVAL;
            if (grp[i] + 1 > n_groups) n_groups = grp[i] + 1;
        }
        if (layout == SA_HDP_LAYOUT_GROUP_MULTISET)   /* confirm_valid_groupings :758-786: consecutively numbered from 0 */
            for (int64_t g = 0; g < n_groups; g++) {
                int found = 0;
                for (int64_t i = 0; i < a; i++) found = found || grp[i] == g;
                if (!found) return SA_EINVAL;
            }
    }
    switch (layout) {
        case SA_HDP_LAYOUT_FLAT: middle = 0; break;
        case SA_HDP_LAYOUT_MULTISET: middle = multiset_number(a, kmer_length); break;
        case SA_HDP_LAYOUT_MIDDLE_NTS: middle = a * a; break;
        case SA_HDP_LAYOUT_COMPOSITION: middle = kmer_length + 1; break;
        default: middle = multiset_number(n_groups, kmer_length); break;
    }
    const int64_t num_dps = leaves + middle + 1, base = num_dps - 1;
    int64_t *parents = malloc(sizeof(int64_t) * (size_t) num_dps);
    if (!parents) return SA_ENOMEM;
    /* the tree: one parent rule per leaf (k-mer id = word id), middle DPs under the base DP */
    int64_t w[16];
    for (int64_t id = 0; id < leaves; id++) {
        int64_t parent = base;
        if (!two_level) {
            word_of(id, a, kmer_length, w);
            int64_t mid = 0;
            switch (layout) {
                case SA_HDP_LAYOUT_MULTISET:   /* word_id_to_multiset_id :379-384 */
                    sort_small(w, kmer_length);
                    mid = multiset_id_internal(w, kmer_length, 0, a);
                    break;
                case SA_HDP_LAYOUT_MIDDLE_NTS:   /* kmer_id_to_middle_nts_id :635-640 */
                    mid = a * w[kmer_length / 2 - 1] + w[kmer_length / 2];
                    break;
                case SA_HDP_LAYOUT_COMPOSITION:   /* purine_composition_hdp_model_internal :947-974: number of purines */
                    for (int64_t i = 0; i < kmer_length; i++) mid += grp[w[i]] != 0;
                    break;
                default:   /* word_id_to_group_multiset_id :702-727 */
                    for (int64_t i = 0; i < kmer_length; i++) w[i] = grp[w[i]];
                    sort_small(w, kmer_length);
                    mid = multiset_id_internal(w, kmer_length, 0, n_groups);
                    break;
            }
            if (mid < 0 || mid >= middle) { free(parents); return SA_EINVAL; }
            parent = leaves + mid;
        }
        parents[id] = parent;
    }
    for (int64_t id = leaves; id < leaves + middle; id++) parents[id] = base;
    parents[base] = -1;
    sa_hdp_state_t *s = NULL;
    const int rc = sa_hdp_state_new_tree(&s, num_dps, depth, parents, gamma, gamma_alpha, gamma_beta, grid_start, grid_stop, grid_length,
                                         mu, nu, alpha, beta);
    free(parents);
    if (rc != SA_OK) return rc;
    s->alphabet_size = a;
    s->kmer_length = kmer_length;
    memcpy(s->alphabet, sorted, (size_t) a + 1);
    *out = s;
    return SA_OK;
}

/* ---- digamma / trigamma for the maximum-likelihood alpha (the reference carries SciPy's cephes routines; these are the textbook
 * recurrence + asymptotic series, good to 1e-14 for x > 0) ---- */
static double digamma_pos(double x) {
    double r = 0.0;
    while (x < 12.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + log(x) - 0.5 / x - f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f * (1.0 / 132.0)))));
}
static double trigamma_pos(double x) {
    double r = 0.0;
    while (x < 12.0) { r += 1.0 / (x * x); x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + 1.0 / x + 0.5 * f + (1.0 / x) * f * (1.0 / 6.0 - f * (1.0 / 30.0 - f * (1.0 / 42.0 - f * (1.0 / 30.0 - f * (5.0 / 66.0)))));
}
double sa_hdp_digamma(double x) { return digamma_pos(x); }
double sa_hdp_trigamma(double x) { return trigamma_pos(x); }

int sa_hdp_nig_params_from_table(const double *table5, int64_t n_kmers, double *mu_out, double *nu_out, double *alpha_out,
                                 double *beta_out) {
    if (!table5 || n_kmers < 2 || !mu_out || !nu_out || !alpha_out || !beta_out) return SA_EINVAL;
    /* mle_normal_inverse_gamma_params over (level mean, 1 / level sd^2) */
    double sum_tau = 0.0, sum_log_tau = 0.0, mu_0 = 0.0;
    for (int64_t i = 0; i < n_kmers; i++) {
        const double noise = table5[5 * i + 1];
        if (!(noise > 0.0)) return SA_EINVAL;
        const double tau = 1.0 / (noise * noise);
        sum_tau += tau;
        sum_log_tau += log(tau);
    }
    for (int64_t i = 0; i < n_kmers; i++) { const double noise = table5[5 * i + 1]; mu_0 += table5[5 * i] * (1.0 / (noise * noise)); }
    mu_0 /= sum_tau;
    double sw = 0.0;
    for (int64_t i = 0; i < n_kmers; i++) {
        const double noise = table5[5 * i + 1], dev = table5[5 * i] - mu_0;
        sw += (1.0 / (noise * noise)) * dev * dev;
    }
    const double nu = ((double) n_kmers) / sw;
    /* newton_approx_alpha :751-774, tolerance 1e-9, from alpha = 1 */
    const double constant = sum_log_tau / n_kmers - log(sum_tau / n_kmers);
    double alpha = 1.0;
    for (int it = 0; it < 10000; it++) {
        const double f = log(alpha) - digamma_pos(alpha) + constant, df = 1.0 / alpha - trigamma_pos(alpha);
        if (df == 0.0 || df != df) return SA_EINVAL;
        const double next = alpha - f / df;
        if (!(next > 0.0)) return SA_EINVAL;
        if (fabs(alpha - next) < .000000001) { alpha = next; break; }
        alpha = next;
    }
    *mu_out = mu_0; *nu_out = nu; *alpha_out = alpha; *beta_out = n_kmers * alpha / sum_tau;
    return SA_OK;
}

/* ---------------------------------------------------------------------------------------------------------------------------- */
/* the mutable factor tree of a sampling run                                                                                    */
/* ---------------------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int type;                 /* 0 base, 1 middle, 2 data point; -1: a free slot */
    int parent;               /* factor, -1 */
    int dp;                   /* the factor's DP (data points: -1) */
    int child_head, n_children;
    int sib_next, sib_prev;   /* in the parent's child list */
    int dp_next, dp_prev;     /* in the DP's factor list */
    int data;                 /* data index (data points) */
    double par[5];            /* base factors: mu, nu, two_alpha, beta, log posterior term */
} gfac_t;

typedef struct {
    sa_hdp_state_t *s;
    gfac_t *f;
    int n_f, cap_f, free_head;
    int *dp_fhead, *dp_nf;
    double *c_mean, *c_ssd;   /* cached statistics of the factor being reassigned, per DP (impl/hdp.c:51-53) */
    int64_t *c_size;
    int64_t *ch_first, *ch;   /* child DPs by parent */
    uint64_t rng[4];
    double two_alpha;
    int oom;
} gibbs_t;

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t rng_next(gibbs_t *g) {   /* xoshiro256** */
    uint64_t *s = g->rng;
    const uint64_t r = rotl64(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return r;
}
static void rng_seed(gibbs_t *g, uint64_t seed) {   /* splitmix64 */
    for (int i = 0; i < 4; i++) {
        seed += 0x9e3779b97f4a7c15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        g->rng[i] = z ^ (z >> 31);
    }
}
static double rng_unit(gibbs_t *g) { return (double) (rng_next(g) >> 11) * (1.0 / 9007199254740992.0); }   /* [0, 1) */
static double rng_normal(gibbs_t *g) {
    double u, v, q;
    do { u = 2.0 * rng_unit(g) - 1.0; v = 2.0 * rng_unit(g) - 1.0; q = u * u + v * v; } while (q >= 1.0 || q == 0.0);
    return u * sqrt(-2.0 * log(q) / q);
}
static double rng_gamma(gibbs_t *g, double shape, double rate) {   /* Marsaglia & Tsang; stands for ranlib's gengam(rate, shape) */
    if (shape < 1.0) {
        const double u = rng_unit(g);
        return rng_gamma(g, shape + 1.0, rate) * pow(u > 0.0 ? u : DBL_MIN, 1.0 / shape);
    }
    const double d = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x, v;
        do { x = rng_normal(g); v = 1.0 + c * x; } while (v <= 0.0);
        v = v * v * v;
        const double u = rng_unit(g);
        if (u < 1.0 - 0.0331 * x * x * x * x || log(u > 0.0 ? u : DBL_MIN) < 0.5 * x * x + d * (1.0 - v + log(v))) return d * v / rate;
    }
}
static double rng_beta(gibbs_t *g, double a, double b) {   /* stands for ranlib's genbet */
    const double x = rng_gamma(g, a, 1.0), y = rng_gamma(g, b, 1.0);
    return x / (x + y);
}

static inline double gamma_of(const gibbs_t *g, int dp) { return g->s->gamma[g->s->dp_depth[dp]]; }

static int fac_alloc(gibbs_t *g) {
    if (g->free_head >= 0) {
        const int i = g->free_head;
        g->free_head = g->f[i].sib_next;
        return i;
    }
    if (g->n_f == g->cap_f) {
        const int nc = g->cap_f * 2 + 1024;
        gfac_t *nf = realloc(g->f, sizeof(gfac_t) * (size_t) nc);
        if (!nf) { g->oom = 1; return -1; }
        g->f = nf;
        g->cap_f = nc;
    }
    return g->n_f++;
}
static void fac_init(gibbs_t *g, int i, int type, int dp) {
    gfac_t *F = &g->f[i];
    F->type = type; F->parent = -1; F->dp = dp; F->child_head = -1; F->n_children = 0; F->sib_next = F->sib_prev = -1;
    F->dp_next = F->dp_prev = -1; F->data = -1;
    for (int k = 0; k < 5; k++) F->par[k] = 0.0;
    if (dp >= 0) {   /* stSet_insert(dp->factors, fctr) */
        F->dp_next = g->dp_fhead[dp];
        if (g->dp_fhead[dp] >= 0) g->f[g->dp_fhead[dp]].dp_prev = i;
        g->dp_fhead[dp] = i;
        g->dp_nf[dp]++;
    }
}
static int new_base_factor(gibbs_t *g) {   /* impl/hdp.c:283-299: the cached log term starts at 1.0 */
    const int i = fac_alloc(g);
    if (i < 0) return -1;
    fac_init(g, i, 0, (int) g->s->base_dp);
    gfac_t *F = &g->f[i];
    F->par[0] = g->s->mu; F->par[1] = g->s->nu; F->par[2] = g->two_alpha; F->par[3] = g->s->beta; F->par[4] = 1.0;
    return i;
}
static int new_middle_factor(gibbs_t *g, int dp) {
    const int i = fac_alloc(g);
    if (i < 0) return -1;
    fac_init(g, i, 1, dp);
    return i;
}
static void child_link(gibbs_t *g, int parent, int child) {
    gfac_t *P = &g->f[parent], *C = &g->f[child];
    C->parent = parent;
    C->sib_prev = -1;
    C->sib_next = P->child_head;
    if (P->child_head >= 0) g->f[P->child_head].sib_prev = child;
    P->child_head = child;
    P->n_children++;
}
static void child_unlink(gibbs_t *g, int parent, int child) {
    gfac_t *P = &g->f[parent], *C = &g->f[child];
    if (C->sib_prev >= 0) g->f[C->sib_prev].sib_next = C->sib_next; else P->child_head = C->sib_next;
    if (C->sib_next >= 0) g->f[C->sib_next].sib_prev = C->sib_prev;
    C->sib_next = C->sib_prev = -1;
    C->parent = -1;
    P->n_children--;
}
static void destroy_factor(gibbs_t *g, int i) {   /* impl/hdp.c:331-365 (a factor without children) */
    gfac_t *F = &g->f[i];
    const int parent = F->parent;
    if (parent >= 0) {
        child_unlink(g, parent, i);
        g->s->dp_num_factor_children[g->f[parent].dp]--;
        if (g->f[parent].n_children == 0) destroy_factor(g, parent);
    }
    F = &g->f[i];
    if (F->dp >= 0 && F->type != 2) {
        if (F->dp_prev >= 0) g->f[F->dp_prev].dp_next = F->dp_next; else g->dp_fhead[F->dp] = F->dp_next;
        if (F->dp_next >= 0) g->f[F->dp_next].dp_prev = F->dp_prev;
        g->dp_nf[F->dp]--;
    }
    F->type = -1;
    F->sib_next = g->free_head;
    g->free_head = i;
}
static int base_of(const gibbs_t *g, int i) {
    while (i >= 0 && g->f[i].type != 0) i = g->f[i].parent;
    return i;
}

/* get_factor_stats :414-422 (sum, then the squared deviations around the mean) */
static void stats_sum(const gibbs_t *g, int i, double *sum, int64_t *n) {
    const gfac_t *F = &g->f[i];
    if (F->type == 2) { *sum += g->s->data[F->data]; (*n)++; return; }
    for (int c = F->child_head; c >= 0; c = g->f[c].sib_next) stats_sum(g, c, sum, n);
}
static void stats_ssd(const gibbs_t *g, int i, double center, double *ssd) {
    const gfac_t *F = &g->f[i];
    if (F->type == 2) { const double dev = g->s->data[F->data] - center; *ssd += dev * dev; return; }
    for (int c = F->child_head; c >= 0; c = g->f[c].sib_next) stats_ssd(g, c, center, ssd);
}
static void factor_stats(const gibbs_t *g, int i, double *mean, double *ssd, int64_t *n) {
    *mean = 0.0; *ssd = 0.0; *n = 0;
    stats_sum(g, i, mean, n);
    *mean /= (double) *n;
    stats_ssd(g, i, *mean, ssd);
}

static double log_post_term(double nu_post, double two_alpha_post, double beta_post) {   /* impl/hdp_math_utils.c:532-538 */
    return lgamma(0.5 * two_alpha_post) - .5 * (log(nu_post) + two_alpha_post * log(beta_post));
}
static void add_update(gfac_t *B, double mean, double ssd, double n) {   /* add_update_base_factor_params :424-445 */
    const double mu_prev = B->par[0], nu_prev = B->par[1], ta_prev = B->par[2], beta_prev = B->par[3];
    const double nu_post = nu_prev + n;
    const double mu_post = (mu_prev * nu_prev + mean * n) / nu_post;
    const double ta_post = ta_prev + n;
    const double mean_dev = mean - mu_prev;
    const double sq_mean_dev = nu_prev * n * mean_dev * mean_dev / nu_post;
    const double beta_post = beta_prev + .5 * (ssd + sq_mean_dev);
    B->par[0] = mu_post; B->par[1] = nu_post; B->par[2] = ta_post; B->par[3] = beta_post;
    B->par[4] = log_post_term(nu_post, ta_post, beta_post);
}
static void remove_update(gfac_t *B, double mean, double ssd, double n) {   /* remove_update_base_factor_params :447-468 */
    const double mu_post = B->par[0], nu_post = B->par[1], ta_post = B->par[2], beta_post = B->par[3];
    const double nu_prev = nu_post - n;
    const double mu_prev = (mu_post * nu_post - mean * n) / nu_prev;
    const double ta_prev = ta_post - n;
    const double mean_dev = mean - mu_prev;
    const double sq_mean_dev = nu_prev * n * mean_dev * mean_dev / nu_post;
    const double beta_prev = beta_post - 0.5 * (ssd + sq_mean_dev);
    B->par[0] = mu_prev; B->par[1] = nu_prev; B->par[2] = ta_prev; B->par[3] = beta_prev;
    B->par[4] = log_post_term(nu_prev, ta_prev, beta_prev);
}

/* data_pt_factor_parent_likelihood :502-528 */
static double data_parent_likelihood(const gibbs_t *g, int f, int parent) {
    const double x = g->s->data[g->f[f].data];
    const gfac_t *B = &g->f[base_of(g, parent)];
    const double mu_d = B->par[0], nu_d = B->par[1], ta_d = B->par[2], beta_d = B->par[3];
    const double nu_n = nu_d + 1.0;
    const double mean_dev = x - mu_d;
    const double sq_mean_dev = nu_d * mean_dev * mean_dev / nu_n;
    const double ta_n = ta_d + 1.0;
    const double beta_n = beta_d + 0.5 * sq_mean_dev;
    return (1.0 / sqrt(2.0 * M_PI)) * exp(log_post_term(nu_n, ta_n, beta_n) - B->par[4]);
}
/* factor_parent_joint_log_likelihood :470-500 (the statistics cached in the factor's DP) */
static double joint_parent_log_likelihood(const gibbs_t *g, int f, int parent) {
    const gfac_t *B = &g->f[base_of(g, parent)];
    const int dp = g->f[f].dp;
    const double n = (double) g->c_size[dp], mean = g->c_mean[dp], ssd = g->c_ssd[dp];
    const double mu_d = B->par[0], nu_d = B->par[1], ta_d = B->par[2], beta_d = B->par[3];
    const double nu_n = nu_d + n, ta_n = ta_d + n;
    const double mean_dev = mean - mu_d;
    const double sq_mean_dev = nu_d * n * mean_dev * mean_dev / nu_n;
    const double beta_n = beta_d + 0.5 * (ssd + sq_mean_dev);
    return -0.5 * n * log(2.0 * M_PI) + log_post_term(nu_n, ta_n, beta_n) - B->par[4];
}
static double prior_likelihood(const gibbs_t *g, int f) {   /* :587-612 */
    const sa_hdp_state_t *s = g->s;
    const double dev = s->data[g->f[f].data] - s->mu, ta = g->two_alpha;
    const double alpha_term = exp(lgamma(.5 * (ta + 1.0)) - lgamma(.5 * ta));
    const double nu_term = s->nu / (2.0 * (s->nu + 1.0) * s->beta);
    const double beta_term = pow(1.0 + nu_term * dev * dev, -0.5 * (ta + 1.0));
    return alpha_term * sqrt(nu_term / M_PI) * beta_term;
}
static double prior_joint_log_likelihood(const gibbs_t *g, int f) {   /* :614-643 */
    const sa_hdp_state_t *s = g->s;
    const int dp = g->f[f].dp;
    const double n = (double) g->c_size[dp], mean = g->c_mean[dp], ssd = g->c_ssd[dp], ta = g->two_alpha;
    const double mean_dev = mean - s->mu;
    const double sq_mean_dev = s->nu * n * mean_dev * mean_dev / (s->nu + n);
    const double log_alpha_term = lgamma(.5 * (ta + n)) - lgamma(.5 * ta);
    const double log_nu_term = 0.5 * (log(s->nu) - log(s->nu + n));
    const double log_pi_term = 0.5 * n * log(2.0 * M_PI);
    const double log_beta_term_1 = ta * log(s->beta);
    const double log_beta_term_2 = (ta + n) * log(s->beta + 0.5 * (ssd + sq_mean_dev));
    return log_alpha_term + log_nu_term - log_pi_term + 0.5 * (log_beta_term_1 - log_beta_term_2);
}
static double add_logs(double a, double b) { return a > b ? a + log(1.0 + exp(b - a)) : b + log(1.0 + exp(a - b)); }

static double unobserved_likelihood(const gibbs_t *g, int f, int dp) {   /* unobserved_factor_likelihood :645-692 */
    const int pd = (int) g->s->dp_parent[dp];
    if (pd < 0) return prior_likelihood(g, f);
    const double pg = gamma_of(g, pd);
    double lik = 0.0;
    for (int q = g->dp_fhead[pd]; q >= 0; q = g->f[q].dp_next) lik += g->f[q].n_children * data_parent_likelihood(g, f, q);
    lik += pg * unobserved_likelihood(g, f, pd);
    lik /= (pg + (double) g->s->dp_num_factor_children[pd]);
    return lik;
}
static double unobserved_joint_log_likelihood(const gibbs_t *g, int f, int dp) {   /* :724-776 */
    const int pd = (int) g->s->dp_parent[dp];
    if (pd < 0) return prior_joint_log_likelihood(g, f);
    const double pg = gamma_of(g, pd);
    double ll = GIBBS_MINUS_INF;
    for (int q = g->dp_fhead[pd]; q >= 0; q = g->f[q].dp_next)
        ll = add_logs(ll, log((double) g->f[q].n_children) + joint_parent_log_likelihood(g, f, q));
    ll = add_logs(ll, log(pg) + unobserved_joint_log_likelihood(g, f, pd));
    ll -= log(pg + (double) g->s->dp_num_factor_children[pd]);
    return ll;
}

/* bisect_left, impl/hdp_math_utils.c:380-400 */
static int64_t bisect_left(double x, const double *arr, int64_t length) {
    if (x <= arr[0]) return 0;
    int64_t low = 0, hi = length - 1;
    while (hi > low + 1) {
        const int64_t mid = (hi + low) / 2;
        if (x <= arr[mid]) hi = mid; else low = mid;
    }
    return hi;
}

static void assign_to_parent(gibbs_t *g, int f, int parent, int update) {   /* :1706-1737 */
    child_link(g, parent, f);
    g->s->dp_num_factor_children[g->f[parent].dp]++;
    if (!update) return;
    gfac_t *B = &g->f[base_of(g, parent)];
    if (g->f[f].type == 2) add_update(B, g->s->data[g->f[f].data], 0.0, 1.0);
    else { const int dp = g->f[f].dp; add_update(B, g->c_mean[dp], g->c_ssd[dp], (double) g->c_size[dp]); }
}

/* sample_from_data_pt_factor :1794-1860 / sample_from_middle_factor :1918-1984: the factors of `dp` weighted by their sizes and
 * the likelihood of the moving factor under them, or a new factor -- whose own parent is then sampled one level up */
static int sample_factor(gibbs_t *g, int f, int dp) {
    const int n = g->dp_nf[dp], is_data = g->f[f].type == 2;
    int *order = malloc(sizeof(int) * (size_t) (n > 0 ? n : 1));
    double *cdf = malloc(sizeof(double) * (size_t) (n + 1));
    if (!order || !cdf) { free(order); free(cdf); g->oom = 1; return -1; }
    int k = 0;
    for (int q = g->dp_fhead[dp]; q >= 0; q = g->f[q].dp_next) order[k++] = q;
    double total;
    if (is_data) {
        const double new_prob = gamma_of(g, dp) * unobserved_likelihood(g, f, dp);
        double cumul = 0.0;
        for (int i = 0; i < n; i++) { cumul += g->f[order[i]].n_children * data_parent_likelihood(g, f, order[i]); cdf[i] = cumul; }
        cdf[n] = cumul + new_prob;
        total = cdf[n];
    } else {
        const double new_lp = log(gamma_of(g, dp)) + unobserved_joint_log_likelihood(g, f, dp);
        double mx = new_lp;
        for (int i = 0; i < n; i++) {
            cdf[i] = log((double) g->f[order[i]].n_children) + joint_parent_log_likelihood(g, f, order[i]);
            if (cdf[i] > mx) mx = cdf[i];
        }
        cdf[n] = new_lp;
        double cumul = 0.0;
        for (int i = 0; i <= n; i++) { cumul += exp(cdf[i] - mx); cdf[i] = cumul; }
        total = cdf[n];
    }
    const int64_t choice = bisect_left(rng_unit(g) * total, cdf, n + 1);   /* rand_uniform(cdf[n]) */
    int chosen;
    if (choice == n) {
        const int pd = (int) g->s->dp_parent[dp];
        if (pd < 0) {
            chosen = new_base_factor(g);
        } else {
            chosen = new_middle_factor(g, dp);
            if (chosen >= 0) {
                const int up = sample_factor(g, f, pd);
                if (up < 0) chosen = -1; else assign_to_parent(g, chosen, up, 0);
            }
        }
    } else {
        chosen = order[choice];
    }
    free(order);
    free(cdf);
    return chosen;
}

static int gibbs_factor_iteration(gibbs_t *g, int f) {   /* :1993-1998 with unassign_from_parent :1664-1704 */
    const int parent = g->f[f].parent;
    const int parent_dp = g->f[parent].dp;
    const int base = base_of(g, parent);
    child_unlink(g, parent, f);
    g->s->dp_num_factor_children[parent_dp]--;
    if (g->f[parent].n_children == 0) destroy_factor(g, parent);
    double mean, ssd;
    int64_t n;
    factor_stats(g, f, &mean, &ssd, &n);
    if (g->f[base].type == 0) remove_update(&g->f[base], mean, ssd, (double) n);   /* (not when the base factor went with its last child) */
    const int dp = g->f[f].dp;
    if (dp >= 0) { g->c_mean[dp] = mean; g->c_size[dp] = n; g->c_ssd[dp] = ssd; }
    const int np = sample_factor(g, f, parent_dp);
    if (np < 0) return SA_ENOMEM;
    assign_to_parent(g, f, np, 1);
    return SA_OK;
}

/* the tree as the state's flat arrays, parents in front of their children (the order serialize_factor_tree_internal writes and
 * sa_hdp_state_load checks): base factors in list order, depth first */
static int export_tree(gibbs_t *g) {
    sa_hdp_state_t *s = g->s;
    int64_t n = 0, nb = 0;
    for (int i = 0; i < g->n_f; i++) n += g->f[i].type >= 0;
    int64_t *ft = malloc(sizeof(int64_t) * (size_t) (n + 1)), *fp = malloc(sizeof(int64_t) * (size_t) (n + 1));
    int64_t *fr = malloc(sizeof(int64_t) * (size_t) (n + 1)), *fn = malloc(sizeof(int64_t) * (size_t) (n + 1));
    double *pa = calloc((size_t) (n + 1) * 5, sizeof(double));
    int *stack = malloc(sizeof(int) * 2 * (size_t) (n + 1));
    if (!ft || !fp || !fr || !fn || !pa || !stack) { free(ft); free(fp); free(fr); free(fn); free(pa); free(stack); return SA_ENOMEM; }
    int64_t k = 0;
    for (int b = g->dp_fhead[s->base_dp]; b >= 0; b = g->f[b].dp_next) {
        int sp = 0;
        stack[0] = b; stack[1] = -1; sp = 1;
        while (sp > 0) {
            sp--;
            const int i = stack[2 * sp], parent_new = stack[2 * sp + 1];
            const gfac_t *F = &g->f[i];
            const int64_t me = k++;
            ft[me] = F->type; fp[me] = parent_new; fn[me] = F->n_children;
            fr[me] = F->type == 2 ? F->data : F->dp;
            if (F->type == 0) { nb++; for (int q = 0; q < 5; q++) pa[5 * me + q] = F->par[q]; }
            for (int c = F->child_head; c >= 0; c = g->f[c].sib_next) { stack[2 * sp] = c; stack[2 * sp + 1] = (int) me; sp++; }
        }
    }
    free(stack);
    free(s->f_type); free(s->f_parent); free(s->f_ref); free(s->f_params); free(s->f_n_children);
    s->f_type = ft; s->f_parent = fp; s->f_ref = fr; s->f_params = pa; s->f_n_children = fn;
    s->n_factors = k; s->n_base_factors = nb;
    return k == n ? SA_OK : SA_ESTATE;
}

static void gibbs_free(gibbs_t *g) {
    free(g->f); free(g->dp_fhead); free(g->dp_nf); free(g->c_mean); free(g->c_ssd); free(g->c_size); free(g->ch_first); free(g->ch);
}
static int gibbs_open(gibbs_t *g, sa_hdp_state_t *s, uint64_t seed) {
    memset(g, 0, sizeof(*g));
    g->s = s;
    g->two_alpha = 2.0 * s->alpha;
    g->free_head = -1;
    g->cap_f = (int) (s->n_factors + s->n_factors / 2 + 1024);
    g->f = malloc(sizeof(gfac_t) * (size_t) g->cap_f);
    g->dp_fhead = malloc(sizeof(int) * (size_t) s->num_dps);
    g->dp_nf = calloc((size_t) s->num_dps, sizeof(int));
    g->c_mean = calloc((size_t) s->num_dps, sizeof(double));
    g->c_ssd = calloc((size_t) s->num_dps, sizeof(double));
    g->c_size = calloc((size_t) s->num_dps, sizeof(int64_t));
    if (!g->f || !g->dp_fhead || !g->dp_nf || !g->c_mean || !g->c_ssd || !g->c_size) return SA_ENOMEM;
    for (int64_t d = 0; d < s->num_dps; d++) g->dp_fhead[d] = -1;
    /* the state's factors, in REVERSE file order: lists are built by insertion at the head, so they come out in file order */
    for (int64_t i = 0; i < s->n_factors; i++) g->f[i].type = -1;
    g->n_f = (int) s->n_factors;
    for (int64_t i = s->n_factors - 1; i >= 0; i--) {
        const int type = (int) s->f_type[i];
        fac_init(g, (int) i, type, type == 2 ? -1 : (int) s->f_ref[i]);
        if (type == 2) g->f[i].data = (int) s->f_ref[i];
        if (type == 0) for (int q = 0; q < 5; q++) g->f[i].par[q] = s->f_params[5 * i + q];
    }
    for (int64_t i = s->n_factors - 1; i >= 0; i--)
        if (s->f_parent[i] >= 0) child_link(g, (int) s->f_parent[i], (int) i);
    rng_seed(g, seed);
    return SA_OK;
}

/* init_factors :1440-1547 as flat arrays: one base factor, under it one middle factor per observed child DP of the base DP and so on
 * down to the leaves, whose factors hold the DP's data points; the base factor's parameters take all the data at once */
static int init_factors(sa_hdp_state_t *s) {
    const int64_t nd = s->num_dps, n = s->n_data;
    int64_t n_obs_nonbase = 0;
    for (int64_t d = 0; d < nd; d++) n_obs_nonbase += s->observed[d] && d != s->base_dp;
    const int64_t nf = 1 + n_obs_nonbase + n;
    free(s->f_type); free(s->f_parent); free(s->f_ref); free(s->f_params); free(s->f_n_children);
    s->f_type = malloc(sizeof(int64_t) * (size_t) nf);
    s->f_parent = malloc(sizeof(int64_t) * (size_t) nf);
    s->f_ref = malloc(sizeof(int64_t) * (size_t) nf);
    s->f_params = calloc((size_t) nf * 5, sizeof(double));
    s->f_n_children = calloc((size_t) nf, sizeof(int64_t));
    int64_t *factor_of_dp = malloc(sizeof(int64_t) * (size_t) nd);
    int64_t *by_depth = malloc(sizeof(int64_t) * (size_t) nd);
    if (!s->f_type || !s->f_parent || !s->f_ref || !s->f_params || !s->f_n_children || !factor_of_dp || !by_depth) {
        free(factor_of_dp); free(by_depth);
        return SA_ENOMEM;
    }
    for (int64_t d = 0; d < nd; d++) { factor_of_dp[d] = -1; s->dp_num_factor_children[d] = 0; }
    int64_t k = 0;
    s->f_type[0] = 0; s->f_parent[0] = -1; s->f_ref[0] = s->base_dp;
    factor_of_dp[s->base_dp] = 0;
    k = 1;
    /* observed DPs by depth (parents first), ascending id inside a depth */
    int64_t m = 0;
    for (int64_t depth = 1; depth < s->depth; depth++)
        for (int64_t d = 0; d < nd; d++)
            if (s->observed[d] && s->dp_depth[d] == depth) by_depth[m++] = d;
    for (int64_t q = 0; q < m; q++) {
        const int64_t d = by_depth[q], pf = factor_of_dp[s->dp_parent[d]];
        s->f_type[k] = 1; s->f_parent[k] = pf; s->f_ref[k] = d;
        s->f_n_children[pf]++;
        factor_of_dp[d] = k++;
    }
    for (int64_t i = 0; i < n; i++) {
        const int64_t pf = factor_of_dp[s->data_dp[i]];
        s->f_type[k] = 2; s->f_parent[k] = pf; s->f_ref[k] = i;
        s->f_n_children[pf]++;
        k++;
    }
    s->n_factors = k;
    s->n_base_factors = 1;
    /* num_factor_children of a DP: children of the factors that sit in it */
    for (int64_t i = 0; i < k; i++)
        if (s->f_type[i] != 2) s->dp_num_factor_children[s->f_ref[i]] += s->f_n_children[i];
    /* the base factor's parameters: new_base_factor, then add_update with the statistics of all the data */
    double mean = 0.0, ssd = 0.0;
    for (int64_t i = 0; i < n; i++) mean += s->data[i];   /* (tree order of the reference: any order sums the same set) */
    mean /= (double) n;
    for (int64_t i = 0; i < n; i++) { const double dev = s->data[i] - mean; ssd += dev * dev; }
    gfac_t B;
    memset(&B, 0, sizeof(B));
    B.par[0] = s->mu; B.par[1] = s->nu; B.par[2] = 2.0 * s->alpha; B.par[3] = s->beta; B.par[4] = 1.0;
    add_update(&B, mean, ssd, (double) n);
    for (int q = 0; q < 5; q++) s->f_params[q] = B.par[q];
    free(factor_of_dp);
    free(by_depth);
    return SA_OK;
}

int sa_hdp_state_pass_data(sa_hdp_state_t *s, const double *data, const int64_t *dp_ids, int64_t n) {
    if (!s || !data || !dp_ids || n < 1) return SA_EINVAL;
    /* verify_valid_dp_assignments :1106-1130: an existing DP without child DPs */
    uint8_t *has_child = calloc((size_t) s->num_dps, 1);
    if (!has_child) return SA_ENOMEM;
    for (int64_t d = 0; d < s->num_dps; d++)
        if (s->dp_parent[d] >= 0) has_child[s->dp_parent[d]] = 1;
    for (int64_t i = 0; i < n; i++)
        if (dp_ids[i] < 0 || dp_ids[i] >= s->num_dps || has_child[dp_ids[i]] || !isfinite(data[i])) { free(has_child); return SA_EINVAL; }
    free(has_child);
    double *nd = malloc(sizeof(double) * (size_t) n);
    int64_t *ni = malloc(sizeof(int64_t) * (size_t) n);
    if (!nd || !ni) { free(nd); free(ni); return SA_ENOMEM; }
    memcpy(nd, data, sizeof(double) * (size_t) n);
    memcpy(ni, dp_ids, sizeof(int64_t) * (size_t) n);
    /* reset_hdp_data :1591-1660 */
    free(s->data); free(s->data_dp);
    s->data = nd; s->data_dp = ni; s->n_data = n;
    s->has_data = 1;
    s->splines_finalized = 0;
    s->samples_taken = 0;
    if (s->sample_gamma) {
        for (int64_t d = 0; d < s->depth; d++) s->gamma[d] = s->gamma_alpha[d] / s->gamma_beta[d];
        for (int64_t d = 0; d < s->num_dps; d++) { s->w_aux[d] = 1.0; s->s_aux[d] = 0; }
    }
    /* mark_observed_dps :1132-1160 */
    memset(s->observed, 0, (size_t) s->num_dps);
    memset(s->has_post, 0, (size_t) s->num_dps);
    memset(s->has_slope, 0, (size_t) s->num_dps);
    for (int64_t i = 0; i < n; i++)
        for (int64_t a = ni[i]; a >= 0 && !s->observed[a]; a = s->dp_parent[a]) s->observed[a] = 1;
    s->n_observed = 0;
    for (int64_t d = 0; d < s->num_dps; d++) s->row_of_dp[d] = s->observed[d] ? s->n_observed++ : -1;
    free(s->post); free(s->slope);
    const size_t plane = (size_t) (s->n_observed > 0 ? s->n_observed : 1) * (size_t) s->grid_length;
    s->post = calloc(plane, sizeof(double));
    s->slope = calloc(plane, sizeof(double));
    if (!s->post || !s->slope) return SA_ENOMEM;
    for (int64_t d = 0; d < s->num_dps; d++) s->has_post[d] = s->observed[d];   /* (collectors start at zero: mark_observed_dps) */
    return init_factors(s);
}

int sa_hdp_state_kmer_dp(const sa_hdp_state_t *s, const char *kmer) {   /* kmer_id, impl/nanopore_hdp.c:405-410 */
    if (!s || !kmer) return -1;
    int64_t id = 0;
    for (int64_t i = 0; i < s->kmer_length; i++) {
        const char *hit = kmer[i] ? memchr(s->alphabet, kmer[i], (size_t) s->alphabet_size) : NULL;
        if (!hit) return -1;
        id = id * s->alphabet_size + (hit - s->alphabet);
    }
    return (int) id;
}

int sa_hdp_state_pass_assignments(sa_hdp_state_t *s, const char *kmers, const double *events, int64_t n) {
    if (!s || !kmers || !events || n < 1) return SA_EINVAL;
    int64_t *ids = malloc(sizeof(int64_t) * (size_t) n);
    if (!ids) return SA_ENOMEM;
    for (int64_t i = 0; i < n; i++) {
        char km[16];
        memcpy(km, kmers + i * s->kmer_length, (size_t) s->kmer_length);
        km[s->kmer_length] = 0;
        ids[i] = sa_hdp_state_kmer_dp(s, km);
        if (ids[i] < 0) { free(ids); return SA_EALPHABET; }   /* ("K-mer contains character outside alphabet": the reference exits) */
    }
    const int rc = sa_hdp_state_pass_data(s, events, ids, n);
    free(ids);
    return rc;
}

/* update_nhdp_from_alignment_with_filter :206-297: a 4-column assignments table (k-mer, strand, signal, probability) or a
 * 15-column alignment table (k-mer in column 9, strand in 4, signal in 13); strand_filter NULL takes every row */
int sa_hdp_state_pass_assignment_file(sa_hdp_state_t *s, const char *path, const char *strand_filter, int64_t *n_out) {
    if (!s || !path) return SA_EINVAL;
    FILE *f = fopen(path, "r");
    if (!f) return SA_EIO;
    int64_t cap = 4096, n = 0;
    double *ev = malloc(sizeof(double) * (size_t) cap);
    char *km = malloc((size_t) cap * (size_t) s->kmer_length);
    int rc = SA_OK;
    char *line;
    while (ev && km && (line = sa_read_line(f)) != NULL) {
        char **tok;
        const int64_t nt = sa_split_ws(line, &tok);
        if (nt == 0) { free(tok); free(line); continue; }
        if (nt != 15 && nt != 4) { free(tok); free(line); rc = SA_EIO; break; }
        const int kc = nt == 15 ? 9 : 0, sc = nt == 15 ? 4 : 1, vc = nt == 15 ? 13 : 2;
        if (!strand_filter || strcmp(tok[sc], strand_filter) == 0) {
            char *end = NULL;
            const double v = strtod(tok[vc], &end);
            if (end == tok[vc] || (int64_t) strlen(tok[kc]) != s->kmer_length) { free(tok); free(line); rc = SA_EIO; break; }
            if (n == cap) {
                cap *= 2;
                double *ne = realloc(ev, sizeof(double) * (size_t) cap);
                char *nk = realloc(km, (size_t) cap * (size_t) s->kmer_length);
                if (ne) ev = ne;
                if (nk) km = nk;
                if (!ne || !nk) { free(tok); free(line); rc = SA_ENOMEM; break; }
            }
            ev[n] = v;
            memcpy(km + n * s->kmer_length, tok[kc], (size_t) s->kmer_length);
            n++;
        }
        free(tok);
        free(line);
    }
    fclose(f);
    if (!ev || !km) rc = SA_ENOMEM;
    if (rc == SA_OK && n == 0) rc = SA_EINVAL;
    if (rc == SA_OK) rc = sa_hdp_state_pass_assignments(s, km, ev, n);
    if (n_out) *n_out = n;
    free(ev);
    free(km);
    return rc;
}

/* ---------------------------------------------------------------------------------------------------------------------------- */
/* the sweeps                                                                                                                   */
/* ---------------------------------------------------------------------------------------------------------------------------- */
typedef struct {
    gibbs_t *g;
    sa_hdp_sampler_t *sampler;
    int64_t iter, samples, burn_in, thinning, num_samples;
    int rc;
} sweep_t;

static void take_sample(sweep_t *w) {   /* take_distr_sample :2067-2092: weights on the host, grid evaluation and mixing on the GPU */
    int rc = export_tree(w->g);
    if (rc == SA_OK) rc = sa_hdp_sampler_add(w->sampler, w->g->s);
    if (rc != SA_OK && w->rc == SA_OK) w->rc = rc;
    w->samples++;
    w->g->s->samples_taken++;
}

static void sample_dp_factors(sweep_t *w, int dp) {   /* :2108-2163 */
    gibbs_t *g = w->g;
    if (!g->s->observed[dp]) return;
    const int64_t nc = g->s->dp_num_factor_children[dp];
    int *todo = malloc(sizeof(int) * (size_t) (nc > 0 ? nc : 1));
    if (!todo) { w->rc = SA_ENOMEM; return; }
    int64_t i = 0;
    for (int q = g->dp_fhead[dp]; q >= 0; q = g->f[q].dp_next)
        for (int c = g->f[q].child_head; c >= 0 && i < nc; c = g->f[c].sib_next) todo[i++] = c;
    for (int64_t j = 0; j < i && w->rc == SA_OK; j++) {
        const int rc = gibbs_factor_iteration(g, todo[j]);
        if (rc != SA_OK || g->oom) { w->rc = SA_ENOMEM; break; }
        w->iter++;
        if (w->iter % w->thinning == 0 && w->iter > w->burn_in) {
            take_sample(w);
            if (w->samples >= w->num_samples) break;
        }
    }
    free(todo);
}

static void sample_gamma_params(sweep_t *w) {   /* :2165-2300 */
    gibbs_t *g = w->g;
    sa_hdp_state_t *s = g->s;
    /* sample_gamma_aux_vars: w ~ Beta(gamma + 1, children), s ~ Bernoulli(children / (children + gamma)) per observed DP */
    for (int64_t d = 0; d < s->num_dps; d++) {
        if (!s->observed[d]) continue;
        const double nc = (double) s->dp_num_factor_children[d], gm = gamma_of(g, (int) d);
        s->w_aux[d] = rng_beta(g, gm + 1.0, nc);
        s->s_aux[d] = rng_unit(g) < nc / (nc + gm);
    }
    int64_t n_fctrs[64];
    double sum_log_w[64];
    int64_t sum_s[64];
    for (int64_t d = 0; d < s->depth; d++) { n_fctrs[d] = 0; sum_log_w[d] = 0.0; sum_s[d] = 0; }
    for (int64_t d = 0; d < s->num_dps; d++) {
        if (!s->observed[d]) continue;
        const int64_t depth = s->dp_depth[d];
        n_fctrs[depth] += g->dp_nf[d];
        sum_log_w[depth] += log(s->w_aux[d]);
        if (s->s_aux[d]) sum_s[depth]++;
    }
    for (int64_t depth = 0; depth < s->depth; depth++) {
        if (depth == 0) {   /* sample_base_gamma_internal: Escobar & West (1995) */
            const double nc = (double) s->dp_num_factor_children[s->base_dp];
            const double beta_post = s->gamma_beta[0] - sum_log_w[0];
            const double alpha_post = s->gamma_alpha[0] + (double) n_fctrs[0];
            const double frac = (alpha_post - 1.0) / (nc * beta_post);
            const double wt = frac / (1.0 + frac);
            s->gamma[0] = wt * rng_gamma(g, alpha_post, beta_post) + (1 - wt) * rng_gamma(g, alpha_post - 1.0, beta_post);
        } else {            /* sample_middle_gammas_internal */
            const double alpha_post = s->gamma_alpha[depth] + (double) (n_fctrs[depth] - sum_s[depth]);
            const double beta_post = s->gamma_beta[depth] - sum_log_w[depth];
            s->gamma[depth] = rng_gamma(g, alpha_post, beta_post);
        }
        w->iter++;
        if (w->iter % w->thinning == 0 && w->iter > w->burn_in) {
            take_sample(w);
            if (w->samples >= w->num_samples) break;
        }
    }
}

int sa_hdp_state_gibbs(sa_hdp_state_t *s, int64_t num_samples, int64_t burn_in, int64_t thinning, uint64_t seed, int device, int verbose) {
    if (!s || num_samples < 1 || burn_in < 0 || thinning < 1) return SA_EINVAL;
    if (!s->has_data || s->n_factors < 1) return SA_ESTATE;   /* "Cannot perform Gibbs sampling before passing data to HDP." */
    if (s->splines_finalized) return SA_ESTATE;
    gibbs_t g;
    int rc = gibbs_open(&g, s, seed);
    sa_hdp_sampler_t *sampler = NULL;
    if (rc == SA_OK) rc = sa_hdp_sampler_open(&sampler, s, device);
    if (rc != SA_OK) { gibbs_free(&g); return rc; }
    sweep_t w = {&g, sampler, 0, 0, burn_in, thinning, num_samples, SA_OK};
    int64_t sweep = 1, prev_iter = 0;
    int64_t *order = malloc(sizeof(int64_t) * (size_t) s->num_dps);
    if (!order) w.rc = SA_ENOMEM;
    while (w.rc == SA_OK && w.samples < num_samples) {
        if (verbose) {
            fprintf(stderr, "Beginning sweep %lld. Performed %lld sampling iterations. Previous sweep sampled from ~%lld non-data point "
                            "factors. Collected %lld of %lld distribution samples.\n", (long long) sweep, (long long) w.iter,
                    (long long) (sweep > 1 ? w.iter - prev_iter - s->n_data : 0), (long long) w.samples, (long long) num_samples);
            prev_iter = w.iter;
        }
        sweep++;
        for (int64_t i = 0; i < s->num_dps; i++) {   /* get_shuffled_dps :2094-2106 (Knuth) */
            const int64_t pos = (int64_t) (rng_next(&g) % (uint64_t) (i + 1));
            order[i] = order[pos];
            order[pos] = i;
        }
        for (int64_t i = 0; i < s->num_dps && w.rc == SA_OK; i++) {
            sample_dp_factors(&w, (int) order[i]);
            if (w.samples >= num_samples) break;
        }
        if (w.rc == SA_OK && s->sample_gamma && w.samples < num_samples) sample_gamma_params(&w);
    }
    free(order);
    /* the tree as it stands, and the collectors: added to what earlier runs left in the state */
    int rc2 = export_tree(&g);
    if (w.rc == SA_OK) w.rc = rc2;
    if (w.rc == SA_OK) {
        const size_t plane = (size_t) s->n_observed * (size_t) s->grid_length;
        double *sum = malloc(sizeof(double) * (plane > 0 ? plane : 1));
        if (!sum) w.rc = SA_ENOMEM;
        else {
            w.rc = sa_hdp_sampler_finish(sampler, sum);
            sampler = NULL;
            if (w.rc == SA_OK)
                for (size_t i = 0; i < plane; i++) s->post[i] += sum[i];
            free(sum);
        }
    }
    if (sampler) sa_hdp_sampler_close(sampler);
    gibbs_free(&g);
    return w.rc;
}

/* finalize_distributions :2551-2584: collectors / samples, then the spline slopes (on the GPU) */
int sa_hdp_state_finalize(sa_hdp_state_t *s, int device) {
    if (!s) return SA_EINVAL;
    if (s->samples_taken <= 0 || s->splines_finalized) return SA_ESTATE;
    const size_t plane = (size_t) (s->n_observed > 0 ? s->n_observed : 1) * (size_t) s->grid_length;
    double *y = malloc(sizeof(double) * plane), *k = malloc(sizeof(double) * plane);
    if (!y || !k) { free(y); free(k); return SA_ENOMEM; }
    const int rc = sa_hdp_finalize_distributions(s->grid, s->grid_length, s->post, s->n_observed, s->samples_taken, device, y, k);
    if (rc == SA_OK) {
        memcpy(s->post, y, sizeof(double) * (size_t) s->n_observed * (size_t) s->grid_length);
        memcpy(s->slope, k, sizeof(double) * (size_t) s->n_observed * (size_t) s->grid_length);
        for (int64_t d = 0; d < s->num_dps; d++) s->has_slope[d] = s->has_post[d] = s->observed[d];
        s->splines_finalized = 1;
    }
    free(y);
    free(k);
    return rc;
}

int64_t sa_hdp_state_samples_taken(const sa_hdp_state_t *s) { return s ? s->samples_taken : -1; }
