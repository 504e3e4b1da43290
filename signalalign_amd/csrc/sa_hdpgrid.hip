// HDP rebuild, the deterministic pieces on the GPU (include/signalalign_hip.h):
//   k_hdp_pdf       evaluate_posterior_predictive (impl/hdp.c:530-562) for every base factor and evaluate_prior_predictive
//                   (:564-585) as the last row, on the sampling grid: rows x grid points
//   k_hdp_mix       push_factor_distr (:2046-2065) for every observed DP: collector += weight * pdf, base factors in tree order, the
//                   prior last -- the order in which take_distr_sample (:2067-2092) adds them
//   k_hdp_finalize  finalize_distributions (:2551-2584): collector / samples and spline_knot_slopes
//                   (impl/hdp_math_utils.c:402-442: forward elimination, Cramer's rule on the last two rows, back substitution --
//                   the reference writes it as a recursion; same operations in the same order here)
// Built with -ffp-contract=off like everything else: a * b + c stays two roundings, as in the reference's x86-64 build.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "signalalign_hip.h"
#include "sa_hdpstate.h"

namespace {

struct HdpCol {       // per base factor (host-side constants of evaluate_posterior_predictive)
    double mu, nu_ratio, beta_denom, two_alpha_numer, lg_half, log_nu_numer, log_denom;
    double pad;
};

__global__ __launch_bounds__(256) void k_hdp_pdf(const HdpCol *__restrict__ cols, int n_base, const double *__restrict__ grid,
                                                 int grid_length, double prior_mu, double prior_nu_factor, double prior_constant,
                                                 double prior_alpha_power, double *__restrict__ P) {
    const int i = (int) (blockIdx.x * blockDim.x + threadIdx.x);
    const int c = (int) blockIdx.y;
    if (i >= grid_length) return;
    const double x = grid[i];
    double v;
    if (c < n_base) {
        const HdpCol k = cols[c];
        const double mean_dev = x - k.mu;
        const double sq_mean_dev = k.nu_ratio * mean_dev * mean_dev;
        const double beta_numer = k.beta_denom + 0.5 * sq_mean_dev;
        // log_posterior_conditional_term (impl/hdp_math_utils.c:532-538): lgamma(two_alpha / 2) - (log nu + two_alpha log beta) / 2
        const double log_numer = k.lg_half - .5 * (k.log_nu_numer + k.two_alpha_numer * log(beta_numer));
        v = 0.3989422804014327 * exp(log_numer - k.log_denom);   // 1 / sqrt(2 pi)
    } else {
        const double dev = x - prior_mu;
        v = prior_constant * pow(1.0 + prior_nu_factor * dev * dev, prior_alpha_power);
    }
    P[(size_t) c * (size_t) grid_length + (size_t) i] = v;
}

__global__ __launch_bounds__(256) void k_hdp_mix(const long long *__restrict__ row_start, const long long *__restrict__ col,
                                                 const double *__restrict__ w, const double *__restrict__ P, int grid_length,
                                                 double *__restrict__ out) {
    const int i = (int) (blockIdx.x * blockDim.x + threadIdx.x);
    const long long r = blockIdx.y;
    if (i >= grid_length) return;
    double acc = 0.0;
    for (long long e = row_start[r]; e < row_start[r + 1]; e++) acc += w[e] * P[(size_t) col[e] * (size_t) grid_length + (size_t) i];
    out[(size_t) r * (size_t) grid_length + (size_t) i] = acc;
}

// one DP per thread; c_scr: n_rows x grid_length scratch for the eliminated centre coefficients (the eliminated right-hand sides
// are kept in the slope row itself until the back substitution overwrites them)
__global__ __launch_bounds__(64) void k_hdp_finalize(const double *__restrict__ x, int n, const double *__restrict__ sum, long long n_rows,
                                                     double inv_samples, double *__restrict__ y_out, double *__restrict__ k_out,
                                                     double *__restrict__ c_scr) {
    const long long r = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const double *s = sum + (size_t) r * (size_t) n;
    double *y = y_out + (size_t) r * (size_t) n;
    double *k = k_out + (size_t) r * (size_t) n;
    double *cc = c_scr + (size_t) r * (size_t) n;
    for (int i = 0; i < n; i++) y[i] = s[i] * inv_samples;
    double right_prev = 1.0 / (x[1] - x[0]);
    double center_prev = 2.0 * right_prev;
    double rhs_prev = 3.0 * (y[1] - y[0]) * right_prev * right_prev;
    cc[0] = center_prev; k[0] = rhs_prev;
    for (int idx = 1; idx < n - 1; idx++) {
        const double left = 1.0 / (x[idx] - x[idx - 1]);
        const double right = 1.0 / (x[idx + 1] - x[idx]);
        double center = 2.0 * (left + right);
        double rhs = 3.0 * ((y[idx] - y[idx - 1]) * left * left + (y[idx + 1] - y[idx]) * right * right);
        center -= left * right_prev / center_prev;
        rhs -= left * rhs_prev / center_prev;
        cc[idx] = center; k[idx] = rhs;
        right_prev = right; center_prev = center; rhs_prev = rhs;
    }
    {   // the last row against the one before it: Cramer's rule
        const double left = 1.0 / (x[n - 1] - x[n - 2]);
        const double center = 2.0 * left;
        const double rhs = 3.0 * (y[n - 1] - y[n - 2]) * left * left;
        k[n - 1] = (rhs * center_prev - rhs_prev * left) / (center * center_prev - right_prev * left);
    }
    for (int idx = n - 2; idx >= 0; idx--) {
        const double right = 1.0 / (x[idx + 1] - x[idx]);
        k[idx] = (k[idx] - right * k[idx + 1]) / cc[idx];
    }
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void) hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes > 0 ? bytes : 8) == hipSuccess ? SA_OK : SA_ENOMEM; }
    int put(const void *src, size_t bytes) {
        if (alloc(bytes)) return SA_ENOMEM;
        return (bytes == 0 || hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) == hipSuccess) ? SA_OK : SA_ENODEVICE;
    }
};

int use_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        (void) hipGetLastError();
        fprintf(stderr, "[signalalign_hip] no HIP device available; this library has no CPU fallback\n");
        return SA_ENODEVICE;
    }
    return hipSetDevice(device) == hipSuccess ? SA_OK : SA_ENODEVICE;
}

}  // namespace

extern "C" int sa_hdp_state_distr_sample(const sa_hdp_state_t *s, int device, double *out) {
    if (!s || !out) return SA_EINVAL;
    if (!s->has_data) return SA_ESTATE;
    int rc = use_device(device);
    if (rc) return rc;
    int64_t *row_start = nullptr, *col = nullptr, nnz = 0;
    double *w = nullptr;
    rc = sa_hdp_state_weights(s, &row_start, &col, &w, &nnz);
    if (rc) return rc;
    struct Free3 { int64_t *a, *b; double *c; ~Free3() { free(a); free(b); free(c); } } free3{row_start, col, w};
    const int64_t nb = s->n_base_factors, G = s->grid_length, nrow = s->n_observed;
    std::vector<HdpCol> cols((size_t) (nb > 0 ? nb : 1));
    int64_t c = 0;
    for (int64_t f = 0; f < s->n_factors; f++) {
        if (s->f_type[f] != 0) continue;
        const double *pa = s->f_params + 5 * f;   // mu, nu, two_alpha, beta, log posterior term (cache_base_factor_params, impl/hdp.c:269-281)
        HdpCol k;
        const double nu_numer = pa[1] + 1.0;
        k.mu = pa[0];
        k.nu_ratio = pa[1] / nu_numer;
        k.beta_denom = pa[3];
        k.two_alpha_numer = pa[2] + 1.0;
        k.lg_half = lgamma(0.5 * k.two_alpha_numer);
        k.log_nu_numer = log(nu_numer);
        k.log_denom = pa[4];
        k.pad = 0.0;
        cols[(size_t) c++] = k;
    }
    // evaluate_prior_predictive (impl/hdp.c:564-585)
    const double two_alpha = 2.0 * s->alpha;
    const double nu_factor = s->nu / (2.0 * (s->nu + 1.0) * s->beta);
    const double alpha_term = exp(lgamma(.5 * (two_alpha + 1.0)) - lgamma(.5 * two_alpha));
    const double beta_term = sqrt(nu_factor / M_PI);
    const double constant_term = alpha_term * beta_term;
    const double alpha_power = -0.5 * (two_alpha + 1.0);

    DevBuf d_cols, d_grid, d_P, d_rs, d_col, d_w, d_out;
    if ((rc = d_cols.put(cols.data(), sizeof(HdpCol) * cols.size()))) return rc;
    if ((rc = d_grid.put(s->grid, sizeof(double) * (size_t) G))) return rc;
    if ((rc = d_P.alloc(sizeof(double) * (size_t) (nb + 1) * (size_t) G))) return rc;
    if ((rc = d_rs.put(row_start, sizeof(int64_t) * (size_t) (nrow + 1)))) return rc;
    if ((rc = d_col.put(col, sizeof(int64_t) * (size_t) nnz))) return rc;
    if ((rc = d_w.put(w, sizeof(double) * (size_t) nnz))) return rc;
    if ((rc = d_out.alloc(sizeof(double) * (size_t) (nrow > 0 ? nrow : 1) * (size_t) G))) return rc;
    const unsigned gx = (unsigned) ((G + 255) / 256);
    // (grid.y is limited to 65535: rows in slices)
    for (int64_t c0 = 0; c0 <= nb; c0 += 65535) {
        const int64_t n = nb + 1 - c0 < 65535 ? nb + 1 - c0 : 65535;
        hipLaunchKernelGGL(k_hdp_pdf, dim3(gx, (unsigned) n), dim3(256), 0, 0, (const HdpCol *) d_cols.p + c0, (int) (nb - c0 > 0 ? nb - c0 : 0),
                           (const double *) d_grid.p, (int) G, s->mu, nu_factor, constant_term, alpha_power,
                           (double *) d_P.p + (size_t) c0 * (size_t) G);
    }
    for (int64_t r0 = 0; r0 < nrow; r0 += 65535) {
        const int64_t n = nrow - r0 < 65535 ? nrow - r0 : 65535;
        hipLaunchKernelGGL(k_hdp_mix, dim3(gx, (unsigned) n), dim3(256), 0, 0, (const long long *) d_rs.p + r0, (const long long *) d_col.p,
                           (const double *) d_w.p, (const double *) d_P.p, (int) G, (double *) d_out.p + (size_t) r0 * (size_t) G);
    }
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return SA_ENODEVICE;
    if (nrow > 0 && hipMemcpy(out, d_out.p, sizeof(double) * (size_t) nrow * (size_t) G, hipMemcpyDeviceToHost) != hipSuccess) return SA_ENODEVICE;
    return SA_OK;
}


// ---- the collectors of a sampling run (sa_hdpgibbs.c: sa_hdp_state_gibbs) ---------------------------------------------------
// take_distr_sample (impl/hdp.c:2067-2092) once per kept sample of the Gibbs run: the collectors (observed DPs x grid points) stay
// in HBM for the whole run, a sample brings its weights (CSR, from the host: sa_hdp_state_weights) and the base factors'
// normal-inverse-gamma parameters, k_hdp_pdf evaluates the posterior predictives and the prior on the grid and k_hdp_mix_add adds
// every observed DP's mixture to its collector.
__global__ __launch_bounds__(256) void k_hdp_mix_add(const long long *__restrict__ row_start, const long long *__restrict__ col,
                                                     const double *__restrict__ w, const double *__restrict__ P, int grid_length,
                                                     double *__restrict__ out) {
    const int i = (int) (blockIdx.x * blockDim.x + threadIdx.x);
    const long long r = blockIdx.y;
    if (i >= grid_length) return;
    double acc = 0.0;
    for (long long e = row_start[r]; e < row_start[r + 1]; e++) acc += w[e] * P[(size_t) col[e] * (size_t) grid_length + (size_t) i];
    out[(size_t) r * (size_t) grid_length + (size_t) i] += acc;
}

struct sa_hdp_sampler {
    int device;
    int64_t G, nrow;
    double *d_grid = nullptr, *d_sum = nullptr, *d_P = nullptr, *d_w = nullptr;
    HdpCol *d_cols = nullptr;
    long long *d_rs = nullptr, *d_col = nullptr;
    int64_t cap_cols = 0, cap_nnz = 0;
};

extern "C" void sa_hdp_sampler_close(sa_hdp_sampler *h) {
    if (!h) return;
    (void) hipSetDevice(h->device);
    void *ptrs[] = {h->d_grid, h->d_sum, h->d_P, h->d_w, h->d_cols, h->d_rs, h->d_col};
    for (void *p : ptrs)
        if (p) (void) hipFree(p);
    delete h;
}

extern "C" int sa_hdp_sampler_open(sa_hdp_sampler **out, const sa_hdp_state_t *s, int device) {
    if (!out || !s) return SA_EINVAL;
    if (!s->has_data) return SA_ESTATE;
    int rc = use_device(device);
    if (rc) return rc;
    sa_hdp_sampler *h = new (std::nothrow) sa_hdp_sampler();
    if (!h) return SA_ENOMEM;
    h->device = device; h->G = s->grid_length; h->nrow = s->n_observed;
    const size_t plane = sizeof(double) * (size_t) (h->nrow > 0 ? h->nrow : 1) * (size_t) h->G;
    if (hipMalloc((void **) &h->d_grid, sizeof(double) * (size_t) h->G) != hipSuccess || hipMalloc((void **) &h->d_sum, plane) != hipSuccess ||
        hipMalloc((void **) &h->d_rs, sizeof(long long) * (size_t) (h->nrow + 2)) != hipSuccess ||
        hipMemcpy(h->d_grid, s->grid, sizeof(double) * (size_t) h->G, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(h->d_sum, 0, plane) != hipSuccess) {
        sa_hdp_sampler_close(h);
        return SA_ENOMEM;
    }
    *out = h;
    return SA_OK;
}

extern "C" int sa_hdp_sampler_add(sa_hdp_sampler *h, const sa_hdp_state_t *s) {
    if (!h || !s || s->n_observed != h->nrow || s->grid_length != h->G) return SA_EINVAL;
    if (hipSetDevice(h->device) != hipSuccess) return SA_ENODEVICE;
    int64_t *row_start = nullptr, *col = nullptr, nnz = 0;
    double *w = nullptr;
    int rc = sa_hdp_state_weights(s, &row_start, &col, &w, &nnz);
    if (rc) return rc;
    struct Free3 { int64_t *a, *b; double *c; ~Free3() { free(a); free(b); free(c); } } free3{row_start, col, w};
    const int64_t nb = s->n_base_factors, G = h->G, nrow = h->nrow;
    std::vector<HdpCol> cols((size_t) (nb > 0 ? nb : 1));
    int64_t c = 0;
    for (int64_t f = 0; f < s->n_factors; f++) {
        if (s->f_type[f] != 0) continue;
        const double *pa = s->f_params + 5 * f;
        HdpCol k;
        const double nu_numer = pa[1] + 1.0;
        k.mu = pa[0]; k.nu_ratio = pa[1] / nu_numer; k.beta_denom = pa[3]; k.two_alpha_numer = pa[2] + 1.0;
        k.lg_half = lgamma(0.5 * k.two_alpha_numer); k.log_nu_numer = log(nu_numer); k.log_denom = pa[4]; k.pad = 0.0;
        cols[(size_t) c++] = k;
    }
    if (nb + 1 > h->cap_cols) {   // (the number of base factors moves from sample to sample: grow in steps)
        if (h->d_cols) (void) hipFree(h->d_cols);
        if (h->d_P) (void) hipFree(h->d_P);
        h->d_cols = nullptr; h->d_P = nullptr;
        h->cap_cols = 2 * (nb + 1) + 16;
        if (hipMalloc((void **) &h->d_cols, sizeof(HdpCol) * (size_t) h->cap_cols) != hipSuccess ||
            hipMalloc((void **) &h->d_P, sizeof(double) * (size_t) h->cap_cols * (size_t) G) != hipSuccess) { h->cap_cols = 0; return SA_ENOMEM; }
    }
    if (nnz > h->cap_nnz) {
        if (h->d_col) (void) hipFree(h->d_col);
        if (h->d_w) (void) hipFree(h->d_w);
        h->d_col = nullptr; h->d_w = nullptr;
        h->cap_nnz = 2 * nnz + 1024;
        if (hipMalloc((void **) &h->d_col, sizeof(long long) * (size_t) h->cap_nnz) != hipSuccess ||
            hipMalloc((void **) &h->d_w, sizeof(double) * (size_t) h->cap_nnz) != hipSuccess) { h->cap_nnz = 0; return SA_ENOMEM; }
    }
    if (hipMemcpy(h->d_cols, cols.data(), sizeof(HdpCol) * (size_t) (nb > 0 ? nb : 1), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_rs, row_start, sizeof(long long) * (size_t) (nrow + 1), hipMemcpyHostToDevice) != hipSuccess ||
        (nnz > 0 && (hipMemcpy(h->d_col, col, sizeof(long long) * (size_t) nnz, hipMemcpyHostToDevice) != hipSuccess ||
                     hipMemcpy(h->d_w, w, sizeof(double) * (size_t) nnz, hipMemcpyHostToDevice) != hipSuccess)))
        return SA_ENODEVICE;
    const double two_alpha = 2.0 * s->alpha;
    const double nu_factor = s->nu / (2.0 * (s->nu + 1.0) * s->beta);
    const double constant_term = exp(lgamma(.5 * (two_alpha + 1.0)) - lgamma(.5 * two_alpha)) * sqrt(nu_factor / M_PI);
    const double alpha_power = -0.5 * (two_alpha + 1.0);
    const unsigned gx = (unsigned) ((G + 255) / 256);
    for (int64_t c0 = 0; c0 <= nb; c0 += 65535) {
        const int64_t n = nb + 1 - c0 < 65535 ? nb + 1 - c0 : 65535;
        hipLaunchKernelGGL(k_hdp_pdf, dim3(gx, (unsigned) n), dim3(256), 0, 0, (const HdpCol *) h->d_cols + c0, (int) (nb - c0 > 0 ? nb - c0 : 0),
                           (const double *) h->d_grid, (int) G, s->mu, nu_factor, constant_term, alpha_power, h->d_P + (size_t) c0 * (size_t) G);
    }
    for (int64_t r0 = 0; r0 < nrow; r0 += 65535) {
        const int64_t n = nrow - r0 < 65535 ? nrow - r0 : 65535;
        hipLaunchKernelGGL(k_hdp_mix_add, dim3(gx, (unsigned) n), dim3(256), 0, 0, (const long long *) h->d_rs + r0, (const long long *) h->d_col,
                           (const double *) h->d_w, (const double *) h->d_P, (int) G, h->d_sum + (size_t) r0 * (size_t) G);
    }
    // (the next sample's uploads are blocking copies on the same stream: they order themselves behind these kernels)
    return hipGetLastError() == hipSuccess ? SA_OK : SA_ENODEVICE;
}

extern "C" int sa_hdp_sampler_finish(sa_hdp_sampler *h, double *sum_out) {
    if (!h || !sum_out) { sa_hdp_sampler_close(h); return SA_EINVAL; }
    int rc = SA_OK;
    if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        (h->nrow > 0 && hipMemcpy(sum_out, h->d_sum, sizeof(double) * (size_t) h->nrow * (size_t) h->G, hipMemcpyDeviceToHost) != hipSuccess))
        rc = SA_ENODEVICE;
    sa_hdp_sampler_close(h);
    return rc;
}

extern "C" int sa_hdp_finalize_distributions(const double *grid, int64_t grid_length, const double *sum, int64_t n_rows, int64_t samples,
                                             int device, double *y_out, double *slope_out) {
    if (!grid || !sum || !slope_out || grid_length < 2 || grid_length > (1 << 24) || n_rows < 0 || samples <= 0) return SA_EINVAL;
    for (int64_t i = 1; i < grid_length; i++)
        if (!(grid[i] > grid[i - 1])) return SA_EINVAL;
    int rc = use_device(device);
    if (rc) return rc;
    if (n_rows == 0) return SA_OK;
    const size_t plane = sizeof(double) * (size_t) n_rows * (size_t) grid_length;
    DevBuf d_grid, d_sum, d_y, d_k, d_c;
    if ((rc = d_grid.put(grid, sizeof(double) * (size_t) grid_length))) return rc;
    if ((rc = d_sum.put(sum, plane))) return rc;
    if ((rc = d_y.alloc(plane)) || (rc = d_k.alloc(plane)) || (rc = d_c.alloc(plane))) return rc;
    const double inv = 1.0 / ((double) samples);
    hipLaunchKernelGGL(k_hdp_finalize, dim3((unsigned) ((n_rows + 63) / 64)), dim3(64), 0, 0, (const double *) d_grid.p, (int) grid_length,
                       (const double *) d_sum.p, (long long) n_rows, inv, (double *) d_y.p, (double *) d_k.p, (double *) d_c.p);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return SA_ENODEVICE;
    if (hipMemcpy(slope_out, d_k.p, plane, hipMemcpyDeviceToHost) != hipSuccess) return SA_ENODEVICE;
    if (y_out && hipMemcpy(y_out, d_y.p, plane, hipMemcpyDeviceToHost) != hipSuccess) return SA_ENODEVICE;
    return SA_OK;
}
