#!/bin/bash
# The host sources (planner, loaders, anchors, parameter estimation: sa_plan.c, sa_io.c) built with gcc under
# AddressSanitizer + UBSan and driven by the host test-suite.  The GPU entry points are stubs that return SA_ENODEVICE:
# this build exists to check host memory safety only (GPU sanitizers are not available on the pool).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=/tmp/sa_host_asan
mkdir -p $OUT
cat > $OUT/stubs.c <<'EOS'
#include "signalalign_hip.h"
#include "sa_internal.h"
#include <stdlib.h>
#define NODEV { return SA_ENODEVICE; }
int sa_device_count(void) { return 0; }
int sa_device_memory(int d, int64_t *f, int64_t *t) NODEV
int sa_batch_create(sa_batch_t **b, const sa_model_t *m, const sa_params_t *p, const sa_job_t *j, int64_t n, const char *const *a, int d, unsigned f) NODEV
int sa_batch_run(sa_batch_t *b) NODEV
int sa_dplan_compare(const sa_model_t *m, const sa_params_t *p, const sa_job_t *j, int64_t n, const char *const *a, int d, unsigned f) NODEV
int sa_batch_start(sa_batch_t *b) NODEV
int sa_batch_wait(sa_batch_t *b) NODEV
int sa_batch_n_pairs(const sa_batch_t *b, int64_t j, int64_t *n) NODEV
int sa_batch_pairs(const sa_batch_t *b, int64_t j, sa_pair_t *o, int64_t c) NODEV
int sa_batch_stats(const sa_batch_t *b, sa_batch_stats_t *s) NODEV
int sa_batch_job_cells(const sa_batch_t *b, int64_t j, double *f, double *k) NODEV
void sa_batch_destroy(sa_batch_t *b) { (void) b; }
int sa_align_batch(const sa_model_t *m, const sa_params_t *p, const sa_job_t *j, int64_t n, const char *const *a, int d, unsigned f, sa_pair_t **po, int64_t *no) NODEV
int sa_expect_batch(const sa_model_t *m, const sa_params_t *p, const sa_job_t *j, int64_t n, const char *const *a, int d, unsigned f, double *t, double *l, sa_assignment_t **as, int64_t *na) NODEV
int sa_scalings_mom(const sa_model_t *m, const char *s, int64_t n, const double *e, int64_t ne, unsigned f, double *a, double *b) NODEV
int sa_event_align_batch(const sa_model_t *m, const sa_ea_job_t *j, int64_t n, int d, unsigned f, sa_ea_pair_t **p, int64_t *np, int32_t *st, double *c, double *k) NODEV
void sa_event_align_release(void) {}
int sa_mea_batch(const sa_mea_job_t *j, int64_t n, int d, unsigned f, sa_mea_pair_t **p, int64_t *np, double *s, int32_t *st, int32_t *ne, double *k) NODEV
void sa_mea_release(void) {}
void sa_pool_release(void) { sa_plan_pool_release(); }
int sa_batch_mea(sa_batch_t *b, unsigned f, sa_mea_pair_t **p, int64_t *np, double *s, int32_t *st, double *k) NODEV
double sa_mea_printed_posterior(int64_t p) { return (double) p; }
int sa_mea_printed_posterior_device(int64_t a, int64_t n, double *o, int d) NODEV
int64_t sa_mea_params(const int64_t *r, const int64_t *e, const double *p, int64_t n, int32_t *a, int32_t *b, double *c, int32_t *d, int64_t *ne) NODEV
EOS
gcc -O1 -g -std=c11 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off \
    -I"$ROOT/include" -I"$ROOT/signalalign_amd/csrc" -o $OUT/libsignalalign_hip.so \
    "$ROOT/signalalign_amd/csrc/sa_plan.c" "$ROOT/signalalign_amd/csrc/sa_io.c" $OUT/stubs.c -lm -lpthread
cd "$ROOT"
SA_LIBRARY=$OUT/libsignalalign_hip.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) \
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_host_plan.py -x -q -p no:cacheprovider -k "not scalings_by_method_of_moments" "$@"   # that one lives in a HIP source
