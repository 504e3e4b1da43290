/*
 * sa_hdp_oracle.c -- CPU restatement of the deterministic pieces of the reference's HDP rebuild.
 *
 * TEST INFRASTRUCTURE ONLY: tests/ and nothing else call it (the product's host code and kernels are sa_hdpstate.c and
 * sa_hdpgrid.hip).  Plain C11, -ffp-contract=off.  Written as the reference writes these functions -- recursions over the factor
 * tree and the DP tree, one collector row per DP:
 *   sao_hdp_linspace                    linspace                             impl/hdp_math_utils.c:497-510
 *   sao_hdp_spline_knot_slopes          spline_knot_slopes[_internal]        impl/hdp_math_utils.c:402-442
 *   sao_hdp_posterior_predictive        evaluate_posterior_predictive        impl/hdp.c:530-562
 *                                       log_posterior_conditional_term       impl/hdp_math_utils.c:532-538
 *   sao_hdp_prior_predictive            evaluate_prior_predictive            impl/hdp.c:564-585
 *   sao_hdp_distr_sample                take_distr_sample                    impl/hdp.c:2067-2092
 *                                       cache_base_factor_weight             impl/hdp.c:2019-2044
 *                                       cache_prior_contribution             impl/hdp.c:2001-2017
 *                                       push_factor_distr                    impl/hdp.c:2046-2065
 *   sao_hdp_nig_posterior               add_update_base_factor_params with get_factor_stats (one batch)  impl/hdp.c:414-445
 *
 * What pins it (tests/test_oracle_hdp_rebuild.py): the reference cannot be compiled here (sonLib is an empty submodule), so the
 * pins are the numbers a file the reference WROTE holds (tests/golden/models/templateSingleLevelFixed.nhdp, 352 observed DPs):
 *   - spline slopes: the file stores every observed DP's density AND the slopes the reference computed from it ("%.17lg": exact):
 *     sao_hdp_spline_knot_slopes reproduces all 352 x 100 slopes bit for bit;
 *   - base-factor parameters: the file stores the cached normal-inverse-gamma parameters of every base factor (updated
 *     incrementally during sampling) and the data under it: the batch posterior reproduces them to 1e-9;
 *   - num_factor_children of every DP against the factor tree, observed marks against the data assignments.
 * The one-sample posterior predictive (take_distr_sample) is PARITY UNPINNED: the file's densities are averages over the samples of
 * a Gibbs run, not the contribution of its final state; checked by properties only (weights of every DP sum to one, densities
 * integrate to one over a grid that covers them, a DP without data of its own equals its parent's mixture).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void sao_hdp_linspace(double start, double stop, int64_t length, double *lin) {
    int64_t n = length - 1;
    double dx = (stop - start) / ((double) n);
    for (int64_t i = 0; i < n; i++) lin[i] = start + i * dx;
    lin[n] = stop;
}

static void slopes_internal(const double *x, const double *y, double *k, int64_t idx, double center_coef_prev, double right_coef_prev,
                            double rhs_prev, int64_t final_idx) {
    if (idx == final_idx) {
        double left_coef = 1.0 / (x[idx] - x[idx - 1]);
        double center_coef = 2.0 * left_coef;
        double rhs = 3.0 * (y[idx] - y[idx - 1]) * left_coef * left_coef;
        k[idx] = (rhs * center_coef_prev - rhs_prev * left_coef) / (center_coef * center_coef_prev - right_coef_prev * left_coef);
        return;
    }
    double left_coef = 1.0 / (x[idx] - x[idx - 1]);
    double right_coef = 1.0 / (x[idx + 1] - x[idx]);
    double center_coef = 2.0 * (left_coef + right_coef);
    double rhs = 3.0 * ((y[idx] - y[idx - 1]) * left_coef * left_coef + (y[idx + 1] - y[idx]) * right_coef * right_coef);
    center_coef -= left_coef * right_coef_prev / center_coef_prev;
    rhs -= left_coef * rhs_prev / center_coef_prev;
    slopes_internal(x, y, k, idx + 1, center_coef, right_coef, rhs, final_idx);
    k[idx] = (rhs - right_coef * k[idx + 1]) / center_coef;
}

void sao_hdp_spline_knot_slopes(const double *x, const double *y, int64_t length, double *k) {
    double right_coef = 1.0 / (x[1] - x[0]);
    double center_coef = 2.0 * right_coef;
    double rhs = 3.0 * (y[1] - y[0]) * right_coef * right_coef;
    slopes_internal(x, y, k, 1, center_coef, right_coef, rhs, length - 1);
    k[0] = (rhs - right_coef * k[1]) / center_coef;
}

static double log_posterior_conditional_term(double nu_post, double two_alpha_post, double beta_post) {
    return lgamma(0.5 * two_alpha_post) - .5 * (log(nu_post) + two_alpha_post * log(beta_post));
}

/* params5: mu, nu, two_alpha, beta, log posterior term of a base factor */
void sao_hdp_posterior_predictive(const double *param_array, const double *x, double *pdf_out, int64_t length) {
    double mu_denom = param_array[0], nu_denom = param_array[1], two_alpha_denom = param_array[2], beta_denom = param_array[3];
    double log_denom = param_array[4];
    double nu_numer = nu_denom + 1.0;
    double two_alpha_numer = two_alpha_denom + 1.0;
    double nu_ratio = nu_denom / nu_numer;
    double pi_factor = 1.0 / sqrt(2.0 * M_PI);
    for (int64_t i = 0; i < length; i++) {
        double mean_dev = x[i] - mu_denom;
        double sq_mean_dev = nu_ratio * mean_dev * mean_dev;
        double beta_numer = beta_denom + 0.5 * sq_mean_dev;
        double log_numer = log_posterior_conditional_term(nu_numer, two_alpha_numer, beta_numer);
        pdf_out[i] = pi_factor * exp(log_numer - log_denom);
    }
}

void sao_hdp_prior_predictive(double mu, double nu, double two_alpha, double beta, const double *x, double *pdf_out, int64_t length) {
    double nu_factor = nu / (2.0 * (nu + 1.0) * beta);
    double alpha_term = exp(lgamma(.5 * (two_alpha + 1.0)) - lgamma(.5 * two_alpha));
    double beta_term = sqrt(nu_factor / M_PI);
    double constant_term = alpha_term * beta_term;
    double alpha_power = -0.5 * (two_alpha + 1.0);
    for (int64_t i = 0; i < length; i++) {
        double dev = x[i] - mu;
        double var_term = pow(1.0 + nu_factor * dev * dev, alpha_power);
        pdf_out[i] = constant_term * var_term;
    }
}

/* the batch form of the normal-inverse-gamma update: prior (mu, nu, two_alpha, beta) + the data of one base factor */
void sao_hdp_nig_posterior(double mu, double nu, double two_alpha, double beta, const double *data, int64_t n, double *params5_out) {
    double mean = 0.0;
    for (int64_t i = 0; i < n; i++) mean += data[i];
    mean /= (double) n;
    double ssd = 0.0;
    for (int64_t i = 0; i < n; i++) ssd += (data[i] - mean) * (data[i] - mean);
    double num_data = (double) n;
    double nu_post = nu + num_data;
    double mu_post = (mu * nu + mean * num_data) / nu_post;
    double two_alpha_post = two_alpha + num_data;
    double mean_dev = mean - mu;
    double sq_mean_dev = nu * num_data * mean_dev * mean_dev / nu_post;
    double beta_post = beta + .5 * (ssd + sq_mean_dev);
    params5_out[0] = mu_post; params5_out[1] = nu_post; params5_out[2] = two_alpha_post; params5_out[3] = beta_post;
    params5_out[4] = log_posterior_conditional_term(nu_post, two_alpha_post, beta_post);
}

typedef struct {
    int64_t num_dps, n_factors, grid_length;
    const int64_t *dp_parent, *dp_nfc, *dp_depth, *f_type, *f_parent, *f_dp;
    const uint8_t *observed;
    const double *gamma;
    int64_t *dp_child_first, *dp_child, *f_child_first, *f_child, *f_nchild;
    double *base_factor_wt;   /* per DP */
    double *collector;        /* num_dps x grid_length */
} tree_t;

static void cache_prior_contribution(tree_t *t, int64_t dp, double parent_prior_prod) {
    if (!t->observed[dp]) return;
    double gamma_param = t->gamma[t->dp_depth[dp]];
    double total_children = (double) t->dp_nfc[dp];
    double prior_prod = (gamma_param / (gamma_param + total_children)) * parent_prior_prod;
    t->base_factor_wt[dp] += prior_prod;
    for (int64_t c = t->dp_child_first[dp]; c < t->dp_child_first[dp + 1]; c++) cache_prior_contribution(t, t->dp_child[c], prior_prod);
}

static void cache_base_factor_weight(tree_t *t, int64_t fctr) {
    int64_t dp = t->f_dp[fctr];
    double gamma_param = t->gamma[t->dp_depth[dp]];
    double total_children = (double) t->dp_nfc[dp];
    double wt = ((double) t->f_nchild[fctr]) / (gamma_param + total_children);
    t->base_factor_wt[dp] += wt;
    if (t->dp_child_first[dp + 1] > t->dp_child_first[dp]) {
        for (int64_t c = t->f_child_first[fctr]; c < t->f_child_first[fctr + 1]; c++) cache_base_factor_weight(t, t->f_child[c]);
        for (int64_t c = t->dp_child_first[dp]; c < t->dp_child_first[dp + 1]; c++) cache_prior_contribution(t, t->dp_child[c], wt);
    }
}

static void push_factor_distr(tree_t *t, int64_t dp, const double *distr) {
    double *sample_collector = t->collector + dp * t->grid_length;
    double wt = t->base_factor_wt[dp];
    for (int64_t i = 0; i < t->grid_length; i++) sample_collector[i] += wt * distr[i];
    t->base_factor_wt[dp] = 0.0;
    for (int64_t c = t->dp_child_first[dp]; c < t->dp_child_first[dp + 1]; c++)
        if (t->observed[t->dp_child[c]]) push_factor_distr(t, t->dp_child[c], distr);
}

static int child_lists(int64_t n, const int64_t *parent, int64_t **first_out, int64_t **child_out) {
    int64_t *first = calloc((size_t) n + 2, sizeof(int64_t)), *child = malloc(sizeof(int64_t) * (size_t) (n > 0 ? n : 1));
    int64_t *fill = calloc((size_t) n + 1, sizeof(int64_t));
    if (!first || !child || !fill) { free(first); free(child); free(fill); return -1; }
    for (int64_t i = 0; i < n; i++) if (parent[i] >= 0) first[parent[i] + 1]++;
    for (int64_t i = 0; i < n; i++) first[i + 1] += first[i];
    for (int64_t i = 0; i < n; i++) if (parent[i] >= 0) child[first[parent[i]] + fill[parent[i]]++] = i;
    free(fill);
    *first_out = first; *child_out = child;
    return 0;
}

/* collector_out: num_dps x grid_length, zeroed here; rows of unobserved DPs stay zero.  f_dp: the DP of a base / middle factor
 * (anything for data points).  f_params: 5 per factor (base factors).  Base factors are taken in index order. */
int sao_hdp_distr_sample(int64_t num_dps, const int64_t *dp_parent, const int64_t *dp_nfc, const int64_t *dp_depth, const uint8_t *observed,
                         const double *gamma, int64_t n_factors, const int64_t *f_type, const int64_t *f_parent, const int64_t *f_dp,
                         const double *f_params, double mu, double nu, double two_alpha, double beta, const double *grid,
                         int64_t grid_length, double *collector_out) {
    tree_t t;
    memset(&t, 0, sizeof(t));
    t.num_dps = num_dps; t.n_factors = n_factors; t.grid_length = grid_length;
    t.dp_parent = dp_parent; t.dp_nfc = dp_nfc; t.dp_depth = dp_depth; t.f_type = f_type; t.f_parent = f_parent; t.f_dp = f_dp;
    t.observed = observed; t.gamma = gamma; t.collector = collector_out;
    int rc = -1;
    double *pdf = malloc(sizeof(double) * (size_t) grid_length);
    t.base_factor_wt = calloc((size_t) num_dps, sizeof(double));
    t.f_nchild = calloc((size_t) (n_factors > 0 ? n_factors : 1), sizeof(int64_t));
    /* child factors that are not data points take part in the recursion; all children count */
    int64_t *fp = malloc(sizeof(int64_t) * (size_t) (n_factors > 0 ? n_factors : 1));
    if (!pdf || !t.base_factor_wt || !t.f_nchild || !fp) goto done;
    for (int64_t i = 0; i < n_factors; i++) {
        if (f_parent[i] >= 0) t.f_nchild[f_parent[i]]++;
        fp[i] = f_type[i] == 2 ? -1 : f_parent[i];
    }
    if (child_lists(num_dps, dp_parent, &t.dp_child_first, &t.dp_child)) goto done;
    if (child_lists(n_factors, fp, &t.f_child_first, &t.f_child)) goto done;
    memset(collector_out, 0, sizeof(double) * (size_t) num_dps * (size_t) grid_length);
    int64_t base_dp = -1;
    for (int64_t i = 0; i < num_dps; i++) if (dp_parent[i] < 0) base_dp = i;
    for (int64_t f = 0; f < n_factors; f++) {
        if (f_type[f] != 0) continue;
        cache_base_factor_weight(&t, f);
        sao_hdp_posterior_predictive(f_params + 5 * f, grid, pdf, grid_length);
        push_factor_distr(&t, base_dp, pdf);
    }
    cache_prior_contribution(&t, base_dp, 1.0);
    sao_hdp_prior_predictive(mu, nu, two_alpha, beta, grid, pdf, grid_length);
    push_factor_distr(&t, base_dp, pdf);
    rc = 0;
done:
    free(pdf); free(t.base_factor_wt); free(t.f_nchild); free(fp);
    free(t.dp_child_first); free(t.dp_child); free(t.f_child_first); free(t.f_child);
    return rc;
}
