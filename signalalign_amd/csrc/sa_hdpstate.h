/* internal layout of sa_hdp_state_t (sa_hdpstate.c, sa_hdpgrid.hip) */
#ifndef SA_HDPSTATE_H
#define SA_HDPSTATE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
struct sa_hdp_state {
    int64_t alphabet_size, kmer_length;
    char alphabet[64];
    int splines_finalized, has_data, sample_gamma;
    int64_t num_dps, depth, grid_length, n_data, base_dp;
    double mu, nu, alpha, beta;      /* the file holds alpha = two_alpha / 2 (impl/hdp.c:2960) */
    double grid_start, grid_stop;
    double *grid;                    /* linspace(grid_start, grid_stop, grid_length) */
    double *data;
    int64_t *data_dp;
    double *gamma, *gamma_alpha, *gamma_beta, *w_aux;
    int64_t *s_aux;
    int64_t *dp_parent, *dp_num_factor_children, *dp_depth;
    uint8_t *observed, *has_post, *has_slope;
    int64_t *row_of_dp;              /* observed DP -> row of post / slope, -1 otherwise */
    int64_t n_observed;
    double *post, *slope;            /* n_observed x grid_length */
    int64_t n_factors, n_base_factors;
    int64_t *f_type;                 /* 0 base, 1 middle, 2 data point */
    int64_t *f_parent;               /* factor id, -1 for base factors */
    int64_t *f_ref;                  /* the factor's DP (base, middle) or its data index (data point) */
    double *f_params;                /* 5 per factor: mu, nu, two_alpha, beta, log posterior term (base factors) */
    int64_t *f_n_children;
    int64_t samples_taken;           /* distribution samples in `post` since the data were passed (not serialised: impl/hdp.c:2919-3050) */
};
/* CSR weights of one distribution sample (see sa_hdpstate.c); the three arrays are malloc'ed */
int sa_hdp_state_weights(const struct sa_hdp_state *s, int64_t **row_start_out, int64_t **col_out, double **w_out, int64_t *nnz_out);
#ifdef __cplusplus
}
#endif
#endif
