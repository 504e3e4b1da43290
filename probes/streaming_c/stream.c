/* A C caller streaming batches of new reads through the library: create -> start -> (create the next) -> wait -> read the
 * pairs -> destroy.  Prints the steady-state time per batch, everything included, without any Python in the loop. */
#include "signalalign_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

static sa_job_t *load_jobs(const char *path, int64_t *n_out) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    int64_t n;
    if (fread(&n, 8, 1, f) != 1) exit(2);
    sa_job_t *jobs = calloc((size_t) n, sizeof(sa_job_t));
    for (int64_t j = 0; j < n; j++) {
        int64_t h[3];
        if (fread(h, 8, 3, f) != 3) exit(2);
        char *ref = malloc((size_t) h[0] + 1);
        double *ev = malloc(sizeof(double) * 4 * (size_t) h[1]);
        int64_t *ax = malloc(8 * (size_t) h[2] + 8), *ay = malloc(8 * (size_t) h[2] + 8);
        double p3[3];
        if (fread(ref, 1, (size_t) h[0], f) != (size_t) h[0] || fread(ev, 8, 4 * (size_t) h[1], f) != 4 * (size_t) h[1] ||
            fread(ax, 8, (size_t) h[2], f) != (size_t) h[2] || fread(ay, 8, (size_t) h[2], f) != (size_t) h[2] ||
            fread(p3, 8, 3, f) != 3)
            exit(2);
        ref[h[0]] = 0;
        jobs[j].ref = ref; jobs[j].ref_len = h[0]; jobs[j].events = ev; jobs[j].event_stride = 4; jobs[j].n_events = h[1];
        jobs[j].anchor_x = ax; jobs[j].anchor_y = ay; jobs[j].n_anchors = h[2];
        jobs[j].scale = p3[0]; jobs[j].shift = p3[1]; jobs[j].var = p3[2];
    }
    fclose(f);
    *n_out = n;
    return jobs;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: stream <model> <jobs_0.bin> <jobs_1.bin> [batches]\n"); return 2; }
    int64_t n[2];
    sa_job_t *sets[2] = {load_jobs(argv[2], &n[0]), load_jobs(argv[3], &n[1])};
    const int n_batches = argc > 4 ? atoi(argv[4]) : 16;
    sa_model_t *m;
    if (sa_model_load(&m, argv[1], NULL)) return 3;
    sa_params_t p = {0.01, 50, 100, 1000, 3000LL * 3000LL};
    const char *ambig[256];
    sa_default_ambig(ambig);
    for (int overlapped = 0; overlapped < 2; overlapped++) {
        sa_batch_t *prev = NULL;
        double t0 = 0;
        int64_t pairs_total = 0;
        for (int i = 0; i < n_batches + 4; i++) {
            if (i == 4) t0 = now_ms();   /* the first batches fill the caches */
            sa_batch_t *b = NULL;
            int rc = sa_batch_create(&b, m, &p, sets[i & 1], n[i & 1], ambig, 0, 0);
            if (rc) { fprintf(stderr, "create: %s\n", sa_strerror(rc)); return 4; }
            if (!overlapped) {
                if ((rc = sa_batch_run(b))) return 5;
                int64_t k; sa_batch_n_pairs(b, 0, &k); pairs_total += k;
                sa_batch_destroy(b);
            } else {
                if ((rc = sa_batch_start(b))) return 5;
                if (prev) {
                    if ((rc = sa_batch_wait(prev))) return 6;
                    int64_t k; sa_batch_n_pairs(prev, 0, &k); pairs_total += k;
                    sa_batch_destroy(prev);
                }
                prev = b;
            }
        }
        if (prev) { if (sa_batch_wait(prev)) return 6; sa_batch_destroy(prev); }
        const double per = (now_ms() - t0) / n_batches;
        printf("%s: %.1f ms per batch of %lld new reads (%lld pairs of read 0 summed)\n", overlapped ? "overlapped" : "serial    ", per,
               (long long) n[0], (long long) pairs_total);
    }
    return 0;
}
