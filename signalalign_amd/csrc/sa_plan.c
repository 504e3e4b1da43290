/* sa_plan.c -- host-side planning for the MI355X pair-HMM kernels.
 *
 * Everything that is integer geometry stays on the host and is computed once per batch:
 *   - per reference position: the list of path k-mers of ambiguous windows (HDCell paths),
 *   - split regions at large anchor gaps, band table (one (xmyL,width) per anti-diagonal),
 *   - the traceback schedule (which diagonals start a backward sweep, what each sweep emits),
 *   - the total-probability checkpoints (every 10th posterior diagonal of a sweep).
 * The device then only does floating-point work on flat arrays.
 *
 * Reference behaviour restated here (paths relative to the upstream signalAlign tree):
 *   band_construct                      impl/pairwiseAligner.c:195-246
 *   getSplitPoints                      impl/pairwiseAligner.c:1886-1937
 *   getPosteriorProbsWithBanding        impl/pairwiseAligner.c:1450-1590 (the schedule, not the maths)
 *   hdCell_construct2                   impl/pairwiseAligner.c:723-801
 *   stateMachine3_loadFromFile          impl/stateMachine.c:1440-1538
 *   filterToRemoveOverlap               impl/pairwiseAligner.c:1755-1796
 *   convertPairwise...ToAnchorPairs     impl/pairwiseAligner.c:1624-1658
 *   signalUtils_estimateNanoporeParams  impl/signalMachineUtils.c:186-225, impl/nanopore.c:601-954
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "sa_internal.h"

/* ------------------------------------------------------------------------------------------------ */
const char *sa_strerror(int code) {
    switch (code) {
        case SA_OK: return "ok";
        case SA_EINVAL: return "invalid argument";
        case SA_ENOMEM: return "out of memory";
        case SA_ENODEVICE: return "no usable HIP device (this library has no CPU fallback)";
        case SA_EALPHABET: return "k-mer contains a character outside the model alphabet";
        case SA_EBAND: return "anchor pairs give an invalid diagonal";
        case SA_EIO: return "file could not be read or parsed";
        case SA_ESTATE: return "call out of order";
        case SA_EUNSUPPORTED: return "unsupported";
    }
    return "unknown error";
}
const char *sa_version(void) { return "signalalign_hip 0.3 (gfx950)"; }
void sa_free(void *p) { free(p); }

/* test hook (host only): pairs through the 16-byte record they cross PCIe in (sa_internal.h) and back */
int sa_pair_roundtrip(const sa_pair_t *in, sa_pair_t *out, int64_t n) {
    if (!in || !out || n < 0) return SA_EINVAL;
    for (int64_t i = 0; i < n; i++) out[i] = sa_pair16_unpack(sa_pair16_pack(in[i].prob_e7, in[i].x, in[i].y, in[i].path, in[i].kmer_id));
    return SA_OK;
}

/* ---- model ------------------------------------------------------------------------------------ */
static void sort_chars(char *s, int n) {
    for (int i = 1; i < n; i++) {
        char c = s[i];
        int j = i - 1;
        while (j >= 0 && s[j] > c) {
            s[j + 1] = s[j];
            j--;
        }
        s[j + 1] = c;
    }
}

static void hdp_free(sa_hdp_t *h) {
    if (!h) return;
    free(h->grid); free(h->parent); free(h->observed); free(h->resolved); free(h->slot); free(h->y); free(h->slope);
    free(h);
}

static sa_hdp_t *hdp_from_desc(const sa_hdp_desc_t *d) {
    sa_hdp_t *h = calloc(1, sizeof(*h));
    if (!h) return NULL;
    h->num_dps = d->num_dps;
    h->grid_length = d->grid_length;
    h->grid_start = d->grid_start;
    h->grid_stop = d->grid_stop;
    int64_t n = d->num_dps, g = d->grid_length;
    h->grid = malloc(sizeof(double) * g);
    h->parent = malloc(sizeof(int64_t) * n);
    h->observed = malloc(n);
    h->resolved = malloc(sizeof(int64_t) * n);
    h->slot = malloc(sizeof(int64_t) * n);
    /* linspace: every point start + i*dx except the last, which is exactly stop */
    double dx = (d->grid_stop - d->grid_start) / (double) (g - 1);
    for (int64_t i = 0; i + 1 < g; i++) h->grid[i] = d->grid_start + i * dx;
    h->grid[g - 1] = d->grid_stop;
    int64_t ns = 0;
    for (int64_t i = 0; i < n; i++) {
        h->parent[i] = d->parent[i];
        h->observed[i] = d->observed[i] ? 1 : 0;
        h->slot[i] = (h->observed[i] && d->post_pred[i] && d->slopes[i]) ? ns++ : -1;
    }
    h->n_slots = ns;
    h->y = malloc(sizeof(double) * (ns > 0 ? ns : 1) * g);
    h->slope = malloc(sizeof(double) * (ns > 0 ? ns : 1) * g);
    for (int64_t i = 0; i < n; i++) {
        if (h->slot[i] >= 0) {
            memcpy(h->y + h->slot[i] * g, d->post_pred[i], sizeof(double) * g);
            memcpy(h->slope + h->slot[i] * g, d->slopes[i], sizeof(double) * g);
        }
    }
    for (int64_t i = 0; i < n; i++) { /* walk to the first observed ancestor */
        int64_t a = i, guard = 0;
        while (a >= 0 && !h->observed[a] && guard++ < n) a = h->parent[a];
        h->resolved[i] = (a >= 0 && h->observed[a]) ? a : -1;
    }
    return h;
}

uint64_t sa_model_next_uid(void) {
    static uint64_t next = 1;
    return __atomic_fetch_add(&next, 1, __ATOMIC_RELAXED);
}

int sa_model_create(sa_model_t **out, int n_states, const char *alphabet, int k, const double *t10,
                    const double *table5, const sa_hdp_desc_t *hdp) {
    if (!out || !alphabet || !t10 || !table5) return SA_EINVAL;
    if (n_states != 3) return SA_EUNSUPPORTED; /* the 5-state machine aborts in the reference too */
    int na = (int) strlen(alphabet);
    if (na < 1 || na > 60 || k < 1 || k > 12) return SA_EINVAL;
    sa_model_t *m = calloc(1, sizeof(*m));
    if (!m) return SA_ENOMEM;
    m->n_alpha = na;
    m->k = k;
    m->uid = sa_model_next_uid();
    memcpy(m->alphabet, alphabet, na);
    sort_chars(m->alphabet, na);
    for (int i = 1; i < na; i++)
        if (m->alphabet[i] == m->alphabet[i - 1]) {
            free(m);
            return SA_EINVAL;
        }
    m->n_kmers = 1;
    for (int i = 0; i < k; i++) m->n_kmers *= na;
    m->pow_km1 = m->n_kmers / na;
    /* token order of the .model transition line: mm mx my xm xx (xy) ym (yx) yy (+1 extra);
     * tokens 5 and 7 never reach a live transition (gapX<->gapY stay log(0)). */
    m->t_mm = log(t10[0]);
    m->t_mx = log(t10[1]);
    m->t_my = log(t10[2]);
    m->t_xm = log(t10[3]);
    m->t_xx = log(t10[4]);
    m->t_ym = log(t10[6]);
    m->t_yy = log(t10[8]);
    m->table5 = malloc(sizeof(double) * 5 * m->n_kmers);
    if (!m->table5) {
        free(m);
        return SA_ENOMEM;
    }
    memcpy(m->table5, table5, sizeof(double) * 5 * m->n_kmers);
    if (hdp) {
        if (hdp->num_dps < m->n_kmers || hdp->grid_length < 2) {
            sa_model_destroy(m);
            return SA_EINVAL;
        }
        m->hdp = hdp_from_desc(hdp);
        if (!m->hdp) {
            sa_model_destroy(m);
            return SA_ENOMEM;
        }
    }
    *out = m;
    return SA_OK;
}

void sa_model_destroy(sa_model_t *m) {
    if (!m) return;
    hdp_free(m->hdp);
    free(m->table5);
    free(m);
}

int sa_model_alphabet(const sa_model_t *m, char *out64, int *n_alpha, int *k) {
    if (!m) return SA_EINVAL;
    if (out64) {
        memcpy(out64, m->alphabet, m->n_alpha);
        out64[m->n_alpha] = 0;
    }
    if (n_alpha) *n_alpha = m->n_alpha;
    if (k) *k = m->k;
    return SA_OK;
}
const double *sa_model_table5(const sa_model_t *m) { return m ? m->table5 : NULL; }

int64_t sa_model_kmer_id(const sa_model_t *m, const char *kmer) {
    int64_t id = 0;
    for (int i = 0; i < m->k; i++) {
        const char *hit = memchr(m->alphabet, kmer[i], m->n_alpha);
        if (!hit || kmer[i] == 0) return -1;
        id = id * m->n_alpha + (hit - m->alphabet);
    }
    return id;
}
int64_t sa_kmer_id(const sa_model_t *m, const char *kmer) { return (m && kmer) ? sa_model_kmer_id(m, kmer) : -1; }

/* expected value / variance of an observed DP's posterior predictive on the grid */
/* a model with the same alphabet, k-mer length and transitions and another emission table (per-read noise scaling) */
int sa_model_clone_with_table(sa_model_t **out, const sa_model_t *m, const double *table5) {
    if (!out || !m || !table5 || m->hdp) return SA_EINVAL;
    sa_model_t *c = calloc(1, sizeof(*c));
    if (!c) return SA_ENOMEM;
    *c = *m;
    c->uid = sa_model_next_uid();
    c->hdp = NULL;
    c->table5 = malloc(sizeof(double) * 5 * (size_t) m->n_kmers);
    if (!c->table5) { free(c); return SA_ENOMEM; }
    memcpy(c->table5, table5, sizeof(double) * 5 * (size_t) m->n_kmers);
    *out = c;
    return SA_OK;
}

int sa_model_set_emission(sa_model_t *m, int emission) {
    if (!m || (emission != SA_EMISSION_MEAN_ONLY && emission != SA_EMISSION_TWO_DIST && emission != SA_EMISSION_TWO_DIST_SCALED_MODEL))
        return SA_EINVAL;
    if (emission != SA_EMISSION_MEAN_ONLY && m->hdp) return SA_EUNSUPPORTED;
    m->emission = emission;
    return SA_OK;
}

int sa_model_set_to_hdp_expected_values(sa_model_t *m) {
    if (!m || !m->hdp) return SA_EINVAL;
    const sa_hdp_t *h = m->hdp;
    int64_t g = h->grid_length;
    for (int64_t id = 0; id < m->n_kmers; id++) {
        if (!h->observed[id] || h->slot[id] < 0) continue;
        const double *distr = h->y + h->slot[id] * g;
        double ev = 0.0;
        for (int64_t i = 1; i < g; i++) {
            double dx = h->grid[i] - h->grid[i - 1];
            ev += h->grid[i] * distr[i] * dx;
        }
        double var = 0.0;
        for (int64_t i = 1; i < g; i++) {
            double dx = h->grid[i] - h->grid[i - 1];
            double dev = h->grid[i] - ev;
            var += dev * dev * distr[i] * dx;
        }
        m->table5[id * 5] = ev;
        m->table5[id * 5 + 1] = sqrt(var);
    }
    return SA_OK;
}

/* ---- ambiguity -------------------------------------------------------------------------------- */
void sa_default_ambig(const char **map) {
    static const struct { char c; const char *r; } tab[] = {
        {'R', "AG"}, {'Y', "CT"}, {'S', "CG"}, {'W', "AT"}, {'K', "GT"}, {'M', "AC"}, {'B', "CGT"}, {'D', "AGT"},
        {'H', "ACT"}, {'V', "ACG"}, {'X', "ACGT"}, {'L', "CEO"}, {'P', "CE"}, {'Q', "AI"}, {'f', "AF"},
        {'U', "ACEGOT"}, {'Z', "JT"}, {'j', "Tp"}, {'k', "Gb"}, {'l', "Gd"}, {'m', "Ce"}, {'n', "Th"}, {'o', "Ai"}};
    memset(map, 0, sizeof(char *) * 256);
    for (size_t i = 0; i < sizeof(tab) / sizeof(tab[0]); i++) map[(unsigned char) tab[i].c] = tab[i].r;
}

int sa_load_ambig(const char *path, const char **map) {
    /* two whitespace-separated columns: symbol, replacement letters; at most 300 lines of < 100 chars,
     * tokens of at most 9 characters (buffers of create_ambig_bases2). Only 1-character symbols can match. */
    FILE *f = fopen(path, "r");
    if (!f) return SA_EIO;
    memset(map, 0, sizeof(char *) * 256);
    char line[100], enc[16], rep[16];
    int n = 0;
    while (n < 300 && fgets(line, sizeof(line), f)) {
        if (sscanf(line, "%9s %9s", enc, rep) == 2 && enc[1] == 0) {
            char *copy = strdup(rep); /* lives for the process, like the reference's hash values */
            map[(unsigned char) enc[0]] = copy;
        }
        n++;
    }
    fclose(f);
    return SA_OK;
}

/* ---- growable arrays --------------------------------------------------------------------------- */
#define GROW(pl, arr, n, cap, need, type)                                  \
    do {                                                                   \
        if ((pl)->n + (need) > (pl)->cap) {                                \
            if ((pl)->borrowed && (void *) (pl)->arr != (void *) (pl)->regions && (void *) (pl)->arr != (void *) (pl)->segs && \
                (void *) (pl)->arr != (void *) (pl)->cks)                  \
                return SA_EINVAL; /* the counting pass undercounted */     \
            int64_t nc = (pl)->cap ? (pl)->cap * 2 : 1024;                 \
            while (nc < (pl)->n + (need)) nc *= 2;                         \
            void *np_ = realloc((pl)->arr, sizeof(type) * (size_t) nc);    \
            if (!np_) return SA_ENOMEM;                                    \
            (pl)->arr = np_;                                               \
            (pl)->cap = nc;                                                \
        }                                                                  \
    } while (0)

/* ---- band ------------------------------------------------------------------------------------- */
static inline int64_t clampz(int64_t z, int64_t hi) { return z < 0 ? 0 : (z > hi ? hi : z); }

/* One anti-diagonal of the band: the stretch of x-y between the lower corner (xL,yL) and the upper
 * corner (xU,yU) of the current anchor-to-anchor box, snapped to the parity of xay. */
static int clip_row(int64_t xay, int64_t xL, int64_t yL, int64_t xU, int64_t yU, int64_t *lo, int64_t *hi) {
    int64_t a = xL - yL, b = xU - yU;
    if ((xay + a) % 2 != 0) a++;
    if ((xay + b) % 2 != 0) b++;
    int64_t x = (xay + a) / 2;
    if (x < xL) a += 2 * (xL - x);
    int64_t y = (xay - a) / 2;
    if (yL < y) a += 2 * (y - yL);
    x = (xay + b) / 2;
    if (xU < x) b -= 2 * (x - xU);
    y = (xay - b) / 2;
    if (y < yU) b -= 2 * (yU - y);
    if ((xay + a) % 2 != 0 || (xay + b) % 2 != 0 || a > b) return SA_EBAND;
    *lo = a;
    *hi = b;
    return SA_OK;
}

int sa_band_rows(const int64_t *ax, const int64_t *ay, int64_t n, int64_t lX, int64_t lY, int64_t e, int64_t *lo,
                 int64_t *hi) {
    int64_t N = lX + lY;
    int rc = clip_row(0, 0, 0, 0, 0, &lo[0], &hi[0]);
    if (rc) return rc;
    int64_t p_sum = 0, p_dif = 0; /* previous anchor in matrix coordinates, as x+y and x-y */
    int64_t d = 1;
    for (int64_t i = 0; d <= N; i++) {
        int64_t x = lX, y = lY;
        if (i < n) {
            x = ax[i] + 1;
            y = ay[i] + 1;
            if (x <= (p_sum + p_dif) / 2 || y <= (p_sum - p_dif) / 2 || x > lX || y > lY) return SA_EBAND;
        }
        int64_t n_sum = x + y, n_dif = x - y;
        int64_t xL = clampz((p_sum + p_dif - e) / 2, lX);
        int64_t yL = clampz((n_sum - n_dif + e) / 2, lY);
        int64_t xU = clampz((n_sum + n_dif + e) / 2, lX);
        int64_t yU = clampz((p_sum - p_dif - e) / 2, lY);
        int64_t last = n_sum < N ? n_sum : N;
        for (; d <= last; d++) {
            rc = clip_row(d, xL, yL, xU, yU, &lo[d], &hi[d]);
            if (rc) return rc;
        }
        p_sum = n_sum;
        p_dif = n_dif;
        if (i >= n && d <= N) return SA_EBAND; /* cannot happen: the last box ends at (lX,lY) */
    }
    return SA_OK;
}

/* ---- split regions ---------------------------------------------------------------------------- */
typedef struct { int64_t x1, y1, x2, y2; } rect_t;

/* Gaps between consecutive anchors whose rectangle is larger than `limit` cut the alignment in two;
 * each side keeps at most floor(sqrt(limit)) rows/columns of the gap. */
static int64_t split_regions(const int64_t *ax, const int64_t *ay, int64_t n, int64_t lX, int64_t lY, int64_t limit,
                             int ragged_l, int ragged_r, rect_t *out) {
    int64_t cnt = 0, x1 = 0, y1 = 0, px = 0, py = 0;
    int64_t side = (int64_t) sqrt((double) limit);
    for (int64_t i = 0; i <= n; i++) {
        int64_t nx = i < n ? ax[i] : lX, ny = i < n ? ay[i] : lY;
        int64_t gx = nx - px, gy = ny - py;
        int cut = gx * gy > limit;
        if (cut) {
            int64_t hx = gx / 2 > side ? side : gx / 2, hy = gy / 2 > side ? side : gy / 2;
            int skip = ragged_l && ((i < n && i == 0) || (i == n && n == 0));
            if (!skip) out[cnt++] = (rect_t){x1, y1, px + hx, py + hy};
            x1 = nx - hx;
            y1 = ny - hy;
        }
        if (i == n) {
            if (!cut || !ragged_r) out[cnt++] = (rect_t){x1, y1, lX, lY};
        }
        px = nx + 1;
        py = ny + 1;
    }
    return cnt;
}

/* ---- per-position path tables ------------------------------------------------------------------ */
/* Expands the k-mer starting at s into all substitutions of its ambiguous letters, first ambiguous
 * position varying slowest. Returns count, or SA_EALPHABET. ids may be NULL to only count. */
static int64_t expand_kmer(const sa_model_t *m, const char *s, const char *const *ambig, int32_t *ids, int64_t cap) {
    int k = m->k;
    const char *opt[16];
    int nopt[16];
    int64_t total = 1;
    for (int i = 0; i < k; i++) {
        const char *r = ambig ? ambig[(unsigned char) s[i]] : NULL;
        opt[i] = r;
        nopt[i] = r ? (int) strlen(r) : 1;
        total *= nopt[i];
        if (total > (1 << 24)) return SA_EUNSUPPORTED;
    }
    if (!ids) return total;
    if (total > cap) return SA_EINVAL;
    char buf[16];
    for (int64_t j = 0; j < total; j++) {
        int64_t rem = j;
        for (int i = k - 1; i >= 0; i--) { /* last position varies fastest */
            int c = (int) (rem % nopt[i]);
            rem /= nopt[i];
            buf[i] = opt[i] ? opt[i][c] : s[i];
        }
        int64_t id = sa_model_kmer_id(m, buf);
        if (id < 0) return SA_EALPHABET;
        ids[j] = (int32_t) id;
    }
    return total;
}

/* ---- plan -------------------------------------------------------------------------------------- */
/* ---- host blocks kept between batches ---------------------------------------------------------------------------
 * The big arrays of a plan (0.8 GB for 2000 reads) are fresh memory for every batch otherwise: first-touch page faults
 * while the planner threads fill them and 80 ms of munmap when the batch is destroyed.  Freed blocks are parked here
 * and handed out again when they fit (at most twice the request); SA_POOL=0 disables it, sa_pool_release() empties it. */
#define PLAN_POOL_SLOTS 48
static struct { void *p; size_t bytes; } plan_pool[PLAN_POOL_SLOTS];
static size_t plan_pool_held = 0;
static pthread_mutex_t plan_pool_mu = PTHREAD_MUTEX_INITIALIZER;
static int plan_pool_on(void) {
    const char *e = getenv("SA_POOL");
    return !(e && atoi(e) == 0);
}
static __thread void *(*plan_alloc_hook)(size_t) = NULL;
static __thread void (*plan_free_hook)(void *, size_t) = NULL;
void sa_plan_use_allocator(void *(*alloc)(size_t bytes), void (*release)(void *p, size_t bytes)) {
    plan_alloc_hook = alloc;
    plan_free_hook = release;
}

static void *plan_big_alloc(size_t bytes) {
    if (bytes == 0) bytes = 8;
    if (plan_pool_on()) {
        pthread_mutex_lock(&plan_pool_mu);
        int best = -1;
        for (int i = 0; i < PLAN_POOL_SLOTS; i++)
            if (plan_pool[i].p && plan_pool[i].bytes >= bytes && plan_pool[i].bytes / 2 <= bytes + (1 << 20) &&
                (best < 0 || plan_pool[i].bytes < plan_pool[best].bytes))
                best = i;
        if (best >= 0) {
            void *p = plan_pool[best].p;
            plan_pool_held -= plan_pool[best].bytes;
            plan_pool[best].p = NULL;
            pthread_mutex_unlock(&plan_pool_mu);
            return p;
        }
        pthread_mutex_unlock(&plan_pool_mu);
    }
    return malloc(bytes);
}
static void plan_big_free(void *p, size_t bytes) {
    if (!p) return;
    if (plan_pool_on() && bytes >= (1 << 20)) {
        pthread_mutex_lock(&plan_pool_mu);
        if (plan_pool_held + bytes <= ((size_t) 8 << 30))
            for (int i = 0; i < PLAN_POOL_SLOTS; i++)
                if (!plan_pool[i].p) {
                    plan_pool[i].p = p;
                    plan_pool[i].bytes = bytes;
                    plan_pool_held += bytes;
                    pthread_mutex_unlock(&plan_pool_mu);
                    return;
                }
        pthread_mutex_unlock(&plan_pool_mu);
    }
    free(p);
}
void sa_plan_pool_release(void) {
    pthread_mutex_lock(&plan_pool_mu);
    for (int i = 0; i < PLAN_POOL_SLOTS; i++) {
        free(plan_pool[i].p);
        plan_pool[i].p = NULL;
    }
    plan_pool_held = 0;
    pthread_mutex_unlock(&plan_pool_mu);
}

void sa_plan_free(sa_plan_t *pl) {
    if (!pl) return;
    free(pl->jobs); free(pl->regions);
    if (pl->pooled && pl->big_free) {
        pl->big_free(pl->rows, sizeof(sa_row_t) * (size_t) (pl->cap_rows > 0 ? pl->cap_rows : 1));
        pl->big_free(pl->pk, sizeof(int32_t) * (size_t) (pl->cap_pk > 0 ? pl->cap_pk : 1));
        pl->big_free(pl->poff, sizeof(int32_t) * (size_t) (pl->cap_poff > 0 ? pl->cap_poff : 1));
        pl->big_free(pl->pid, sizeof(int32_t) * (size_t) (pl->cap_pid > 0 ? pl->cap_pid : 1));
        if (pl->xc) pl->big_free(pl->xc, sizeof(double) * 4 * (size_t) (pl->cap_pid > 0 ? pl->cap_pid : 1));
        if (pl->prec) pl->big_free(pl->prec, sizeof(sa_prec_t) * (size_t) (pl->cap_pid > 0 ? pl->cap_pid : 1));
        pl->big_free(pl->ev, sizeof(double) * (size_t) (pl->cap_ev > 0 ? pl->cap_ev : 1));
    } else if (pl->pooled) { /* sizes as allocated by sa_plan_build */
        plan_big_free(pl->rows, sizeof(sa_row_t) * (size_t) (pl->cap_rows > 0 ? pl->cap_rows : 1));
        plan_big_free(pl->pk, sizeof(int32_t) * (size_t) (pl->cap_pk > 0 ? pl->cap_pk : 1));
        plan_big_free(pl->poff, sizeof(int32_t) * (size_t) (pl->cap_poff > 0 ? pl->cap_poff : 1));
        plan_big_free(pl->pid, sizeof(int32_t) * (size_t) (pl->cap_pid > 0 ? pl->cap_pid : 1));
        plan_big_free(pl->xc, sizeof(double) * 4 * (size_t) (pl->cap_pid > 0 ? pl->cap_pid : 1));
        plan_big_free(pl->prec, sizeof(sa_prec_t) * (size_t) (pl->cap_pid > 0 ? pl->cap_pid : 1));
        plan_big_free(pl->ev, sizeof(double) * (size_t) (pl->cap_ev > 0 ? pl->cap_ev : 1));
    } else if (!pl->borrowed) {
        free(pl->rows); free(pl->pk); free(pl->poff); free(pl->pid); free(pl->xc); free(pl->ev); free(pl->prec);
    }
    free(pl->segs); free(pl->cks);
    free(pl);
}

static int ring_env_on(void) { /* SA_RING=0: never use the ring kernels (test / comparison hook) */
    const char *e = getenv("SA_RING");
    return !(e && atoi(e) == 0);
}
static int ring_wide_env_on(void) { /* SA_RING_WIDE=0: one-path regions stay on the register kernels whatever their band */
    const char *e = getenv("SA_RING_WIDE");
    return !(e && atoi(e) == 0);
}
/* The index form of path legality (fill_prec) needs the options of every ambiguity letter to be distinct characters */
static int ambig_options_distinct(const char *const *ambig) {
    if (!ambig) return 1;
    for (int ch = 0; ch < 256; ch++) {
        const char *r = ambig[ch];
        if (!r) continue;
        for (int i = 0; r[i]; i++)
            for (int j = i + 1; r[j]; j++)
                if (r[i] == r[j]) return 0;
    }
    return 1;
}

/* Per cell-path records of a SA_KIND_RING region with ambiguous positions.  Paths of column x enumerate the substitutions of
 * the window s[x-1 .. x+k-2] with the LAST position varying fastest (expand_kmer), so with n(c) options for letter c,
 *   shared(x) = product of n over the first k-1 letters of the window = P(x) / n(last letter),
 * path p of column x and path q of column x-1 are a legal step (k-1 shared letters, path_checkLegal) iff
 *   q mod shared(x) == p / n(last letter of x):
 * the legal predecessors of p are q = j * shared(x) + p / n_last(x), j < n(first letter of x-1) (strided), and the legal
 * successors of q in column x+1 are the n_last(x+1) consecutive paths from (q mod shared(x+1)) * n_last(x+1).  The NULL
 * k-mer of column 0 is a legal neighbour of everything (path_checkLegal with a NULL k-mer). */
static int fill_prec(sa_plan_t *pl, const sa_region_t *R, const char *s, const char *const *ambig) {
    const int k = pl->model->k;
    const int64_t lX = R->lX;
    if (!pl->prec) {
        if (pl->borrowed) return SA_EINVAL; /* sized by sa_plan_build whenever a thread counted several paths */
        pl->prec = calloc((size_t) (pl->cap_pid > 0 ? pl->cap_pid : 1), sizeof(sa_prec_t));
        if (!pl->prec) return SA_ENOMEM;
        pl->prec_cap = pl->cap_pid;
    } else if (!pl->borrowed && pl->prec_cap < pl->cap_pid) {
        sa_prec_t *np_ = realloc(pl->prec, sizeof(sa_prec_t) * (size_t) pl->cap_pid);
        if (!np_) return SA_ENOMEM;
        memset(np_ + pl->prec_cap, 0, sizeof(sa_prec_t) * (size_t) (pl->cap_pid - pl->prec_cap));
        pl->prec = np_;
        pl->prec_cap = pl->cap_pid;
    }
    const int32_t *poff = pl->poff + R->poff_off;
    sa_prec_t *pr = pl->prec + R->pid_off;
#define NOPT(c) ((ambig && ambig[(unsigned char) (c)]) ? (int64_t) strlen(ambig[(unsigned char) (c)]) : 1)
    /* column 0: the NULL k-mer; every path of column 1 is a successor */
    {
        int64_t P1 = lX >= 1 ? poff[2] - poff[1] : 0;
        if (P1 > 255) return SA_EUNSUPPORTED;
        pr[0].x = 0; pr[0].pred0 = -1; pr[0].succ0 = lX >= 1 ? poff[1] : -1;
        pr[0].meta = (uint32_t) P1;
    }
    for (int64_t x = 1; x <= lX; x++) {
        const char *w = s + (x - 1);
        const int64_t P = poff[x + 1] - poff[x];
        const int64_t n_last = NOPT(w[k - 1]);
        const int64_t shared = P / n_last;
        int64_t npred = 1, stride = 0;
        if (x >= 2) { npred = NOPT(w[-1]); stride = shared; }
        int64_t nsucc = 0, shared_n = 1, n_last_n = 1;
        if (x < lX) {
            n_last_n = NOPT(w[k]);
            shared_n = P / NOPT(w[0]);
            nsucc = n_last_n;
        }
        if (npred > 255 || nsucc > 255 || stride > 65535) return SA_EUNSUPPORTED;
        for (int64_t p = 0; p < P; p++) {
            sa_prec_t *o = &pr[poff[x] + p];
            o->x = (int32_t) x;
            o->pred0 = (int32_t) (x >= 2 ? poff[x - 1] + p / n_last : 0);
            o->succ0 = x < lX ? (int32_t) (poff[x + 1] + (p % shared_n) * n_last_n) : -1;
            o->meta = (uint32_t) (stride << 16) | (uint32_t) (npred << 8) | (uint32_t) nsucc;
        }
    }
#undef NOPT
    return SA_OK;
}

static int add_region(sa_plan_t *pl, int64_t job, const sa_job_t *jb, rect_t rc, const int64_t *ax, const int64_t *ay,
                      int64_t na, int ragged_l, int ragged_r, const char *const *ambig, int64_t job_ev_off) {
    const sa_model_t *m = pl->model;
    const sa_params_t *p = &pl->params;
    int64_t lX = rc.x2 - rc.x1, lY = rc.y2 - rc.y1, N = lX + lY;
    if (N == 0) return SA_OK; /* "Deal with trivial case": impl/pairwiseAligner.c:1467-1469 */
    GROW(pl, regions, n_regions, cap_regions, 1, sa_region_t);
    sa_region_t *R = &pl->regions[pl->n_regions];
    memset(R, 0, sizeof(*R));
    R->job = (int32_t) job;
    R->ragged_l = ragged_l;
    R->ragged_r = ragged_r;
    R->x1 = rc.x1; R->y1 = rc.y1; R->lX = lX; R->lY = lY; R->N = N;
    R->scale = jb->scale; R->shift = jb->shift; R->var = jb->var;
    R->lvar = log((1 / jb->var));
    R->ev_off = job_ev_off + rc.y1;

    /* paths per x: cell x = 0 is the NULL k-mer (one path, id -1) */
    GROW(pl, poff, n_poff, cap_poff, lX + 2, int32_t);
    R->poff_off = pl->n_poff;
    R->pid_off = pl->n_pid;
    int32_t *poff = pl->poff + pl->n_poff;
    pl->n_poff += lX + 2;
    GROW(pl, pid, n_pid, cap_pid, 1, int32_t);
    pl->pid[pl->n_pid++] = -1;
    poff[0] = 0;
    poff[1] = 1;
    int maxP = 1;
    /* results travel as 16-byte records (sa_pair16_t: 28 bits of x and y, 16 bits of path index) whatever kernels and
     * finalisation produce them: a matrix or a cell they cannot name is refused here, as the device planner refuses it */
    if (rc.x1 + lX >= SA_PAIR16_MAX_COORD || rc.y1 + lY >= SA_PAIR16_MAX_COORD) return SA_EUNSUPPORTED;
    for (int64_t x = 1; x <= lX; x++) {
        const char *s = jb->ref + rc.x1 + (x - 1);
        int64_t cnt = expand_kmer(m, s, ambig, NULL, 0);
        if (cnt < 0) return (int) cnt;
        if (cnt > SA_PAIR16_MAX_PATHS) return SA_EUNSUPPORTED;
        GROW(pl, pid, n_pid, cap_pid, cnt, int32_t);
        int64_t got = expand_kmer(m, s, ambig, pl->pid + pl->n_pid, cnt);
        if (got < 0) return (int) got;
        pl->n_pid += cnt;
        if (cnt > maxP) maxP = (int) cnt;
        int64_t next = (int64_t) poff[x] + cnt;
        if (next > INT32_MAX) return SA_EUNSUPPORTED;
        poff = pl->poff + R->poff_off; /* (GROW of pid does not move poff; kept for clarity) */
        poff[x + 1] = (int32_t) next;
    }

    /* band */
    int64_t *lo = malloc(sizeof(int64_t) * (N + 1)), *hi = malloc(sizeof(int64_t) * (N + 1));
    int32_t *span3 = malloc(sizeof(int32_t) * (N + 1));
    if (!lo || !hi || !span3) {
        free(lo); free(hi); free(span3);
        return SA_ENOMEM;
    }
    int rcode = sa_band_rows(ax, ay, na, lX, lY, p->diagonal_expansion, lo, hi);
    if (rcode) {
        free(lo); free(hi); free(span3);
        return rcode;
    }
    if (pl->n_rows + N + 2 > pl->cap_rows) {
        if (pl->borrowed) {
            free(lo); free(hi); free(span3);
            return SA_EINVAL; /* the counting pass undercounted */
        }
        int64_t nc = pl->cap_rows ? pl->cap_rows * 2 : 4096;
        while (nc < pl->n_rows + N + 2) nc *= 2;
        void *np_ = realloc(pl->rows, sizeof(sa_row_t) * (size_t) nc);
        if (!np_) {
            free(lo); free(hi); free(span3);
            return SA_ENOMEM;
        }
        pl->rows = np_;
        pl->cap_rows = nc;
    }
    R->row_off = pl->n_rows;
    sa_row_t *rows = pl->rows + pl->n_rows;
    pl->n_rows += N + 2; /* one sentinel row behind diagonal N (SA_KIND_RING: its offset closes the last diagonal) */
    poff = pl->poff + R->poff_off;
    int64_t K = lY + (lY & 1) + 2;
    R->K = (int32_t) K;
    int64_t foff = 0, max_rowpaths = 0, span = 0;
    double cf = 0;
    /* register-kernel regions start every diagonal on a 128-byte boundary of its plane: a cache line then belongs to
     * one store instruction of one diagonal and never has to be merged with the next diagonal's bytes */
    int fast_ok = maxP == 1 && !(pl->flags & (SA_FLAG_EXACT | SA_FLAG_FORCE_GENERIC));
    /* HDP emissions on the register kernels: the {y, slope} table is addressed with 32-bit byte offsets */
    if (m->hdp != NULL && (m->hdp->grid_length < 2 || m->hdp->n_slots * m->hdp->grid_length * 16 >= SA_HDP_FAST_MAX_BYTES))
        fast_ok = 0;
    const int64_t row_align = fast_ok ? SA_FAST_ROW_ALIGN : 1;
    for (int64_t d = 0; d <= N; d++) {
        int64_t w = (hi[d] - lo[d]) / 2 + 1;
        int64_t x0 = (d + lo[d]) / 2, xe = x0 + w; /* cells cover x0 .. xe-1 */
        int64_t paths = (int64_t) poff[xe] - poff[x0];
        rows[d].xmyL = (int32_t) lo[d];
        rows[d].width = (int32_t) w;
        rows[d].foff = foff;
        foff += (paths + row_align - 1) / row_align * row_align;
        if (paths > max_rowpaths) max_rowpaths = paths;
        if (d >= 1) cf += (double) paths;
        /* widest window of (x-y+K)>>1 over three consecutive diagonals plus one neighbour each side */
        int64_t uL = (lo[d] + K) >> 1, uR = uL + w - 1;
        int64_t wl = uL - 1, wr = uR + 1;
        for (int64_t b = 1; b <= 2 && d - b >= 0; b++) {
            int64_t l2 = (lo[d - b] + K) >> 1, r2 = l2 + (hi[d - b] - lo[d - b]) / 2;
            if (l2 < wl) wl = l2;
            if (r2 > wr) wr = r2;
        }
        span3[d] = (int32_t) (wr - wl + 1); /* lanes needed to hold this diagonal, the two before it and one neighbour each side */
        if (wr - wl + 1 > span) span = wr - wl + 1;
    }
    /* packed band words for the register kernels */
    GROW(pl, pk, n_pk, cap_pk, N + 1 + SA_PK_PAD + 160, int32_t);
    R->pk_off = pl->n_pk;
    {
        int32_t *pk = pl->pk + pl->n_pk;
        memset(pk, 0, sizeof(int32_t) * (size_t) (N + 1 + SA_PK_PAD + 160));
        rows = pl->rows + R->row_off;
        for (int64_t d = 0; d <= N; d++) {
            int64_t uL = ((int64_t) rows[d].xmyL + K) >> 1;
            int64_t w = rows[d].width;
            int32_t word = (int32_t) ((w > SA_PK_WIDTH_MASK ? SA_PK_WIDTH_MASK : w) | ((uint32_t) uL << SA_PK_SHIFT));
            if (span3[d] <= 64) word |= SA_PK_FWD;
            if (span3[d + 2 <= N ? d + 2 : N] <= 64) word |= SA_PK_BWD;
            pk[SA_PK_PAD + d] = word;
        }
        pl->n_pk += N + 1 + SA_PK_PAD + 160;
    }
    R->f_cellpaths = foff + 1; /* last cell of the match plane: a -inf sentinel the backward kernel reads for lanes without a cell */
    R->max_rowpaths = (int32_t) max_rowpaths;
    R->slots = (int32_t) ((span + 63) / 64);
    if (span > pl->max_span) pl->max_span = span;
    if (foff + 1 > SA_FAST_MAX_CELLS || ((lX + lY + K) >> 1) >= (1ll << (31 - SA_PK_SHIFT))) fast_ok = 0;
    rows[N + 1].xmyL = 0; rows[N + 1].width = 0; rows[N + 1].foff = foff;
    /* ring kernels (sa_ring.inc): several paths per cell, or one path and a band mostly wider than a wave */
    /* (HDP models: both read the emission plane k_emit_hdp_ring fills, one value per cell-path; same table limit as above) */
    const int hdp_plane_ok = m->hdp == NULL ||
                             !(m->hdp->grid_length < 2 || m->hdp->n_slots * m->hdp->grid_length * 16 >= SA_HDP_FAST_MAX_BYTES);
    /* (the expectation pass: ring kernels for regions with several paths per cell under a Gaussian model -- k_bwd_ring<EXPECT> --,
     * never for one-path regions, which keep the register kernels' expectation variant) */
    const int expect_ = (pl->flags & SA_FLAG_EXPECT_INTERNAL) != 0;
    /* (the two-distribution emission exists in the register kernels and the reference-ordered ones: a one-path region with a wide
     * band stays a register-kernel region -- their in-kernel memory-resident path --, one with several paths per cell is not
     * SA_KIND_FAST and sends the batch to the reference-ordered kernels, sa_hip.hip batch_prepare_body) */
    int ring_ok = !(pl->flags & (SA_FLAG_EXACT | SA_FLAG_FORCE_GENERIC)) && hdp_plane_ok && m->emission == 0 &&
                  (!expect_ || (maxP > 1 && m->hdp == NULL)) &&
                  max_rowpaths <= SA_RING_MAX_ROWPATHS && foff + 1 <= SA_FAST_MAX_CELLS && ring_env_on() &&
                  (maxP == 1 || (maxP <= 255 && ambig_options_distinct(ambig)));
    int use_ring = 0;
    if (ring_ok && maxP > 1) use_ring = 1;
    if (ring_ok && maxP == 1 && fast_ok) {
        double wide_cells = 0;
        for (int64_t d = 1; d <= N; d++)
            if (span3[d] > 64) wide_cells += (double) rows[d].width;
        use_ring = wide_cells > SA_RING_WIDE_FRACTION * cf && ring_wide_env_on();
    }
    if (use_ring) {
        fast_ok = 0;
        for (int64_t d = 0; d <= N + 1; d++) { /* (g0 << 32) | offset: see sa_internal.h */
            int64_t g0 = d <= N ? (int64_t) poff[(d + lo[d]) / 2] : 0;
            rows[d].foff |= g0 << 32;
        }
        if (maxP > 1) {
            int rcp = fill_prec(pl, R, jb->ref + rc.x1, ambig);
            if (rcp) { free(lo); free(hi); free(span3); return rcp; }
        }
        pl->n_ring_regions++;
    }
    R->kind = use_ring ? SA_KIND_RING : (fast_ok ? SA_KIND_FAST : SA_KIND_GENERIC);
    R->max_p = maxP; /* 1: the ring kernels skip the per-path records */
    if (fast_ok) pl->n_fast_regions++;

    /* traceback schedule */
    R->seg_off = pl->n_segs;
    int64_t traced_to = 0;
    double cb = 0;
    for (int64_t d = 1; d <= N; d++) {
        int at_end = d == N;
        int tb = d >= traced_to + p->min_diags_between_trace_back && rows[d].width <= p->diagonal_expansion * 2 + 1;
        if (!(at_end || tb)) continue;
        GROW(pl, segs, n_segs, cap_segs, 1, sa_seg_t);
        sa_seg_t *S = &pl->segs[pl->n_segs++];
        memset(S, 0, sizeof(*S));
        S->region = (int32_t) pl->n_regions;
        S->at_end = at_end;
        S->start = d;
        /* the diagonal a traceback starts on keeps all three forward planes: its total there -- forward state x end state --
         * is the backward sweep's candidate bound (k_spec_match, sa_hip.hip) */
        pl->pk[R->pk_off + SA_PK_PAD + d] |= SA_PK_FULL;
        S->from = d - (at_end ? 0 : p->trace_back_diagonals + 1);
        S->to = traced_to;
        if (S->from <= S->to) { /* would violate traceBackDiagonals+1 < minDiagsBetweenTraceBack */
            free(lo); free(hi); free(span3);
            return SA_EINVAL;
        }
        for (int64_t e = S->to + 2; e <= S->start; e++) {
            int64_t x0 = (e + rows[e].xmyL) / 2;
            cb += (double) (poff[x0 + rows[e].width] - poff[x0]);
        }
        /* checkpoints: diagonals from, from-10, ... > to */
        S->ck_base = pl->n_cks;
        int64_t nck = (S->from - S->to + SA_CKPT_EVERY - 1) / SA_CKPT_EVERY;
        S->n_ck = (int32_t) nck;
        GROW(pl, cks, n_cks, cap_cks, nck, sa_ck_t);
        for (int64_t c = 0; c < nck; c++) {
            int64_t e = S->from - SA_CKPT_EVERY * c;
            sa_ck_t *ck = &pl->cks[pl->n_cks++];
            ck->voff = pl->n_vbuf;
            pl->pk[R->pk_off + SA_PK_PAD + e] |= SA_PK_CK; /* dot(F,B) over all three states is taken here */
            ck->nA = rows[e].width;
            ck->nB = e < S->start ? rows[e + 1].width : 0;
            pl->n_vbuf += ck->nA + ck->nB;
        }
        /* measured: 0.54 pairs per diagonal at threshold 0.01 with Gaussian emissions; HDP densities as broad as the bundled
         * model's leave the posteriors flat across the band (3-6 candidates per diagonal at threshold 0.1); overflow re-runs
         * the pass with 4x */
        int64_t cap = (m->hdp ? SA_CAND_PER_DIAG_HDP : SA_CAND_PER_DIAG) * (S->from - S->to + 32);   /* (the 32: short tracebacks at a read's end) */
        if (p->threshold <= 0.0) { /* everything passes: every cell-path of the posterior diagonals */
            cap = 64;
            for (int64_t e = S->to + 1; e <= S->from; e++) {
                int64_t x0 = (e + rows[e].xmyL) / 2;
                cap += poff[x0 + rows[e].width] - poff[x0];
            }
        }
        if (cap > INT32_MAX) cap = INT32_MAX;
        S->cand_cap = (int32_t) cap;
        S->cand_off = pl->n_cand;
        pl->n_cand += cap;
        S->bscratch_off = pl->n_bscratch;
        if (R->kind != SA_KIND_RING) pl->n_bscratch += 12 * (int64_t) max_rowpaths; /* the ring kernels keep backward rows in LDS */
        traced_to = S->from;
    }
    R->n_seg = (int32_t) (pl->n_segs - R->seg_off);
    {   /* derived flags (the words behind diagonal N are zero) */
        int32_t *pk = pl->pk + R->pk_off + SA_PK_PAD;
        for (int64_t d = 0; d <= N; d++) {
            /* (the expectation pass reads all three forward states of every diagonal: impl/pairwiseAligner.c:1423-1443) */
            if ((pk[d] & SA_PK_CK) || !(pk[d + 1] & SA_PK_FWD) || !(pk[d + 2] & SA_PK_FWD) || d + 2 > N ||
                (pl->flags & SA_FLAG_EXPECT_INTERNAL))
                pk[d] |= SA_PK_FULL;
            if (d < N && (pk[d + 1] & SA_PK_FWD)) pk[d] |= SA_PK_FWD_MORE;
            if (d >= 1 && (pk[d - 1] & SA_PK_BWD)) pk[d] |= SA_PK_BWD_MORE;
        }
    }
    free(lo);
    free(hi);
    free(span3);
    pl->jobs[job].cells_fwd += cf;
    pl->jobs[job].cells_bwd += cb;
    pl->cells_fwd += cf;
    pl->cells_bwd += cb;
    pl->n_regions++;
    return SA_OK;
}

/* per-(x,path) emission constants with the read's scale/shift/var folded in:
 *   a  = (event - m) * inv_s            m = scale*mu + shift, inv_s = 1/(var*sd)
 *   lM = cM - a*a/2                     cM = log(1/var) - log(sqrt(2 pi)) - log(sd)
 *   lY = cY - (a/1.75)^2/2              cY = cM - log(1.75)
 * which is emissions_signal_strawManGetKmerEventMatchProbWithDescaling_MeanOnly
 * (impl/stateMachine.c:557-605) with the descaling (e + var*mu - scale*mu - shift)/var - mu = (e - m)/var
 * carried out symbolically. */
static int fill_xc(sa_plan_t *pl) {
    const sa_model_t *m = pl->model;
    if (!pl->borrowed) pl->xc = malloc(sizeof(double) * 4 * (size_t) (pl->n_pid > 0 ? pl->n_pid : 1));
    if (!pl->xc) return SA_ENOMEM;
    for (int64_t r = 0; r < pl->n_regions; r++) {
        const sa_region_t *R = &pl->regions[r];
        const int32_t *poff = pl->poff + R->poff_off;
        int64_t n = poff[R->lX + 1];
        for (int64_t i = 0; i < n; i++) {
            int32_t id = pl->pid[R->pid_off + i];
            double *o = pl->xc + 4 * (R->pid_off + i);
            if (m->hdp != NULL) {
                /* HDP (register kernels only): e' = e/var - o[0]; o[1] = byte offset of the k-mer's {y, slope} row, or
                 * an offset beyond the table when there is no density (NULL k-mer, no observed ancestor) */
                const sa_hdp_t *h = m->hdp;
                int64_t slot = -1;
                if (id >= 0) {
                    int64_t rs = h->resolved[id];
                    slot = rs >= 0 ? h->slot[rs] : -1;
                }
                double mu = id >= 0 ? m->table5[5 * (int64_t) id] : 0.0;
                o[0] = ((R->scale - R->var) * mu + R->shift) / R->var;
                o[1] = slot >= 0 ? (double) (slot * h->grid_length * 16) : (double) SA_HDP_FAST_MAX_BYTES;
                o[2] = 0.0; o[3] = 0.0;
                continue;
            }
            if (id < 0) { /* NULL k-mer: both emissions are log(0); inv_s = 1 keeps (e - m) * inv_s finite */
                o[0] = 0.0; o[1] = 1.0; o[2] = SA_NEG_INF; o[3] = SA_NEG_INF;
                continue;
            }
            double mu = m->table5[5 * (int64_t) id], sd = m->table5[5 * (int64_t) id + 1];
            o[0] = R->scale * mu + R->shift;
            if (sd == 0.0) { /* emissions_signal_logGaussPdf returns LOG_ZERO */
                o[1] = 1.0; o[2] = SA_NEG_INF; o[3] = SA_NEG_INF;
            } else {
                o[1] = 1.0 / (R->var * sd);
                o[2] = R->lvar + (-0.91893853320467267 - log(sd));
                o[3] = R->lvar + (-0.91893853320467267 - log(sd * SA_GAPY_SD_MULT));
            }
        }
    }
    return SA_OK;
}

/* plans job jb as job number j of pl (appends its regions, rows, segments ... to pl's arrays) */
static int plan_job(sa_plan_t *pl, int64_t j, const sa_job_t *jb, const char *const *ambig) {
    const sa_model_t *m = pl->model;
    const sa_params_t *p = &pl->params;
    if (!jb->ref || jb->ref_len < 0 || jb->n_events < 0 || jb->n_anchors < 0 || (jb->n_events && !jb->events) ||
        (jb->n_anchors && (!jb->anchor_x || !jb->anchor_y)) || !(jb->var > 0.0) ||
        (jb->ends & ~(SA_JOB_LEFT_END_NOT_RAGGED | SA_JOB_RIGHT_END_NOT_RAGGED)))
        return SA_EINVAL;
    /* alignmentHasRaggedLeftEnd / alignmentHasRaggedRightEnd of getAlignedPairsUsingAnchors (impl/pairwiseAligner.c:2052-2080);
     * signalMachine passes 1, 1 (impl/signalMachine.c:436-437) = a zeroed `ends` */
    const int ragged_l = !(jb->ends & SA_JOB_LEFT_END_NOT_RAGGED), ragged_r = !(jb->ends & SA_JOB_RIGHT_END_NOT_RAGGED);
    int64_t lX = jb->ref_len == 0 ? 0 : jb->ref_len - (m->k - 1); /* sequence_correctSeqLength */
    if (lX < 0) lX = 0;
    int64_t lY = jb->n_events;
    for (int64_t i = 0; i < jb->n_anchors; i++)
        if (jb->anchor_x[i] < 0 || jb->anchor_y[i] < 0 || jb->anchor_x[i] >= lX || jb->anchor_y[i] >= lY ||
            (i > 0 && (jb->anchor_x[i] <= jb->anchor_x[i - 1] || jb->anchor_y[i] <= jb->anchor_y[i - 1])))
            return SA_EBAND;
    /* events: keep only the mean column */
    if (pl->borrowed) {
        if (pl->n_ev + lY > pl->cap_ev) return SA_EINVAL; /* the counting pass undercounted */
    } else if (pl->n_ev + lY + 1 > pl->cap_ev) {
        int64_t nc = pl->cap_ev ? pl->cap_ev * 2 : 4096;
        while (nc < pl->n_ev + lY + 1) nc *= 2;
        void *np_ = realloc(pl->ev, sizeof(double) * (size_t) nc);
        if (!np_) return SA_ENOMEM;
        pl->ev = np_;
        pl->cap_ev = nc;
    }
    pl->jobs[j].ev_off = pl->n_ev;
    pl->jobs[j].n_events = lY;
    int64_t st = jb->event_stride > 0 ? jb->event_stride : 1;
    for (int64_t i = 0; i < lY; i++) pl->ev[pl->n_ev + i] = jb->events[i * st];
    pl->n_ev += lY;
    pl->jobs[j].region_off = pl->n_regions;

    rect_t *rects = malloc(sizeof(rect_t) * (size_t) (jb->n_anchors + 2));
    int64_t *sx = malloc(sizeof(int64_t) * (size_t) (jb->n_anchors + 1));
    int64_t *sy = malloc(sizeof(int64_t) * (size_t) (jb->n_anchors + 1));
    if (!rects || !sx || !sy) {
        free(rects); free(sx); free(sy);
        return SA_ENOMEM;
    }
    int64_t nr = split_regions(jb->anchor_x, jb->anchor_y, jb->n_anchors, lX, lY, p->split_matrix_bigger_than_this, ragged_l,
                               ragged_r, rects);
    int rc = SA_OK;
    int64_t a = 0;
    for (int64_t i = 0; i < nr && rc == SA_OK; i++) {
        int64_t a0 = a;
        while (a < jb->n_anchors && jb->anchor_x[a] + jb->anchor_y[a] < rects[i].x2 + rects[i].y2) a++;
        for (int64_t t = a0; t < a; t++) {
            sx[t - a0] = jb->anchor_x[t] - rects[i].x1;
            sy[t - a0] = jb->anchor_y[t] - rects[i].y1;
        }
        /* sub-regions behind / in front of a cut are ragged there whatever the caller said (impl/pairwiseAligner.c:2001-2002) */
        rc = add_region(pl, j, jb, rects[i], sx, sy, a - a0, ragged_l || i > 0, ragged_r || i < nr - 1, ambig,
                        pl->jobs[j].ev_off);
    }
    free(rects); free(sx); free(sy);
    pl->jobs[j].n_regions = (int32_t) (pl->n_regions - pl->jobs[j].region_off);
    return rc;
}

static sa_plan_t *plan_new(const sa_model_t *m, const sa_params_t *p, unsigned flags, int64_t n_jobs) {
    sa_plan_t *pl = calloc(1, sizeof(*pl));
    if (!pl) return NULL;
    pl->model = m;
    pl->params = *p;
    pl->flags = flags;
    pl->n_jobs = n_jobs;
    pl->jobs = calloc(n_jobs > 0 ? n_jobs : 1, sizeof(sa_jobinfo_t));
    if (!pl->jobs) { free(pl); return NULL; }
    return pl;
}

/* planning is integer work per read and the reads are independent: worker t plans a contiguous range of jobs into
 * a plan of its own, the ranges are then concatenated (every cross-array offset shifts by what precedes the range) */
typedef struct {
    sa_plan_t *pl;
    const sa_job_t *jobs;
    int64_t n;
    const char *const *ambig;
    int rc;
} plan_worker_t;

static void *plan_worker(void *arg) {
    plan_worker_t *w = arg;
    w->rc = SA_OK;
    for (int64_t j = 0; j < w->n && w->rc == SA_OK; j++) w->rc = plan_job(w->pl, j, &w->jobs[j], w->ambig);
    if (w->rc == SA_OK && !(w->pl->flags & SA_FLAG_DEVICE_XC_INTERNAL)) w->rc = fill_xc(w->pl);
    if (w->rc == SA_OK && w->pl->borrowed &&
        (w->pl->n_rows != w->pl->cap_rows || w->pl->n_pk != w->pl->cap_pk || w->pl->n_poff != w->pl->cap_poff ||
         w->pl->n_pid != w->pl->cap_pid || w->pl->n_ev != w->pl->cap_ev))
        w->rc = SA_EINVAL; /* the counting pass overcounted: the slices would not be contiguous */
    return NULL;
}

/* Counting pass of the threaded planner: how many entries of the big arrays (band rows, packed words, path offsets and
 * ids, emission constants, events) a range of jobs will occupy -- exactly, so that every thread can then write its
 * sub-plan straight into its slice of the final arrays (copying several hundred MB of sub-plans and handing them back
 * to the kernel cost more than planning them).  Follows plan_job / add_region; a job those would reject counts as
 * whatever it counts, the planning pass then fails with its error code. */
static void *count_worker(void *arg) {
    plan_worker_t *w = arg;
    const sa_model_t *m = w->pl->model;
    const sa_params_t *p = &w->pl->params;
    sa_plan_t *c = w->pl; /* cap_* fields receive the counts */
    int nopt[256];
    for (int ch = 0; ch < 256; ch++) nopt[ch] = (w->ambig && w->ambig[ch]) ? (int) strlen(w->ambig[ch]) : 1;
    w->rc = SA_OK;
    for (int64_t j = 0; j < w->n; j++) {
        const sa_job_t *jb = &w->jobs[j];
        if (!jb->ref || jb->ref_len < 0 || jb->n_events < 0 || jb->n_anchors < 0 || (jb->n_events && !jb->events) ||
            (jb->n_anchors && (!jb->anchor_x || !jb->anchor_y)))
            return NULL; /* plan_job rejects it */
        int64_t lX = jb->ref_len == 0 ? 0 : jb->ref_len - (m->k - 1);
        if (lX < 0) lX = 0;
        const int64_t lY = jb->n_events;
        for (int64_t i = 0; i < jb->n_anchors; i++)
            if (jb->anchor_x[i] < 0 || jb->anchor_y[i] < 0 || jb->anchor_x[i] >= lX || jb->anchor_y[i] >= lY ||
                (i > 0 && (jb->anchor_x[i] <= jb->anchor_x[i - 1] || jb->anchor_y[i] <= jb->anchor_y[i - 1])))
                return NULL;
        c->cap_ev += lY;
        rect_t *rects = malloc(sizeof(rect_t) * (size_t) (jb->n_anchors + 2));
        if (!rects) { w->rc = SA_ENOMEM; return NULL; }
        const int64_t nr = split_regions(jb->anchor_x, jb->anchor_y, jb->n_anchors, lX, lY, p->split_matrix_bigger_than_this,
                                         !(jb->ends & SA_JOB_LEFT_END_NOT_RAGGED), !(jb->ends & SA_JOB_RIGHT_END_NOT_RAGGED), rects);
        for (int64_t i = 0; i < nr; i++) {
            const int64_t rX = rects[i].x2 - rects[i].x1, rY = rects[i].y2 - rects[i].y1, N = rX + rY;
            if (N == 0) continue;
            c->cap_rows += N + 2;
            c->cap_pk += N + 1 + SA_PK_PAD + 160;
            c->cap_poff += rX + 2;
            int64_t paths = 1; /* the NULL k-mer of x = 0 */
            for (int64_t x = 1; x <= rX; x++) {
                const unsigned char *s = (const unsigned char *) jb->ref + rects[i].x1 + (x - 1);
                int64_t total = 1;
                for (int q = 0; q < m->k; q++) total *= nopt[s[q]];
                paths += total;
            }
            c->cap_pid += paths;
            if (paths > rX + 1) c->prec_cap = 1; /* several paths somewhere: the planning pass may write per-path records */
        }
        free(rects);
    }
    return NULL;
}

/* Merge of the per-thread sub-plans into one plan.  The destination arrays are sized from the sub-plans' counts
 * (ALLOC_CAT), then every thread copies ITS sub-plan to its place and shifts its offsets by what the threads before it
 * hold (merge_worker): the copy is several hundred MB per batch and runs at memory bandwidth only when it is spread. */
#define ALLOC_CAT(dst, field, count, type)                                                              \
    do {                                                                                                \
        int64_t tot_ = 0;                                                                               \
        for (int t_ = 0; t_ < T; t_++) tot_ += W[t_].pl->count;                                         \
        dst->field = malloc(sizeof(type) * (size_t) (tot_ > 0 ? tot_ : 1));                             \
        if (!dst->field) rc = SA_ENOMEM;                                                                \
    } while (0)
#define COPY_CAT(field, count, type)                                                                    \
    do {                                                                                                \
        if (s->count) memcpy(pl->field + b->count, s->field, sizeof(type) * (size_t) s->count);         \
    } while (0)

typedef struct {
    sa_plan_t *pl;        /* destination */
    sa_plan_t *s;         /* this thread's sub-plan; freed by the worker once copied (munmap of several hundred MB is
                           * 80 ms on one thread) */
    sa_plan_t base;       /* counts held by the threads before it */
    int64_t job_base;
} merge_task_t;

static void *merge_worker(void *arg) {
    merge_task_t *k = arg;
    sa_plan_t *pl = k->pl;
    const sa_plan_t *s = k->s;
    const sa_plan_t *b = &k->base;
    COPY_CAT(regions, n_regions, sa_region_t);
    COPY_CAT(segs, n_segs, sa_seg_t);
    COPY_CAT(cks, n_cks, sa_ck_t);
    if (s->n_jobs) memcpy(pl->jobs + k->job_base, s->jobs, sizeof(sa_jobinfo_t) * (size_t) s->n_jobs);
    /* rows, pk, poff, pid, xc, ev were written in place (borrowed slices) */
    for (int64_t i = 0; i < s->n_jobs; i++) {
        sa_jobinfo_t *J = &pl->jobs[k->job_base + i];
        J->region_off += b->n_regions;
        J->ev_off += b->n_ev;
    }
    for (int64_t i = 0; i < s->n_regions; i++) {
        sa_region_t *R = &pl->regions[b->n_regions + i];
        R->job += (int32_t) k->job_base;
        R->row_off += b->n_rows; R->pk_off += b->n_pk; R->poff_off += b->n_poff; R->pid_off += b->n_pid;
        R->ev_off += b->n_ev; R->seg_off += b->n_segs;
    }
    for (int64_t i = 0; i < s->n_segs; i++) {
        sa_seg_t *S = &pl->segs[b->n_segs + i];
        S->region += (int32_t) b->n_regions;
        S->ck_base += b->n_cks; S->cand_off += b->n_cand; S->bscratch_off += b->n_bscratch;
    }
    for (int64_t i = 0; i < s->n_cks; i++) pl->cks[b->n_cks + i].voff += b->n_vbuf;
    sa_plan_free(k->s);
    k->s = NULL;
    return NULL;
}

static __thread int plan_threads_override = 0; /* sa_plan_digest */
static int plan_threads(int64_t n_jobs) {
    const char *e = getenv("SA_PLAN_THREADS");
    long t = plan_threads_override > 0 ? plan_threads_override : (e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN));
    if (t > 32) t = 32; /* measured on a 256-thread host: 16 -> 48 ms, 32 -> 37 ms, 64 -> 34 ms, 128 -> 38 ms per serial batch cycle */
    if (t > n_jobs / 4) t = n_jobs / 4; /* a handful of reads is not worth a thread */
    return t < 1 ? 1 : (int) t;
}

int sa_plan_build(sa_plan_t **out, const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs,
                  const char *const *ambig, unsigned flags, int64_t chunk_budget) {
    if (!out || !m || !p || (!jobs && n_jobs > 0) || n_jobs < 0) return SA_EINVAL;
    if (p->diagonal_expansion < 0 || p->diagonal_expansion % 2 != 0 || p->trace_back_diagonals < 1 ||
        p->min_diags_between_trace_back < 2 || p->trace_back_diagonals + 1 >= p->min_diags_between_trace_back ||
        !(p->threshold >= 0.0 && p->threshold <= 1.0))
        return SA_EINVAL; /* the asserts of impl/pairwiseAligner.c:1460-1464, :1358-1359 */
    const int T = plan_threads(n_jobs);
    sa_plan_t *pl = NULL;
    int rc = SA_OK;
    const int trace = getenv("SA_TRACE") != NULL;
    struct timespec ts0_;
    clock_gettime(CLOCK_MONOTONIC, &ts0_);
#define PLAN_TRACE(what)                                                                                   \
    do {                                                                                                   \
        if (trace) {                                                                                       \
            struct timespec t_;                                                                            \
            clock_gettime(CLOCK_MONOTONIC, &t_);                                                           \
            fprintf(stderr, "[trace] plan (%d threads): %s at %.1f ms\n", T, what,                         \
                    (t_.tv_sec - ts0_.tv_sec) * 1e3 + (t_.tv_nsec - ts0_.tv_nsec) * 1e-6);                   \
        }                                                                                                  \
    } while (0)
    if (T == 1) {
        pl = plan_new(m, p, flags, n_jobs);
        if (!pl) return SA_ENOMEM;
        plan_worker_t w = {pl, jobs, n_jobs, ambig, SA_OK};
        plan_worker(&w);
        rc = w.rc;
    } else {
        /* contiguous ranges balanced by events (planning cost ~ diagonals ~ events) */
        plan_worker_t *W = calloc((size_t) T, sizeof(*W));
        pthread_t *th = calloc((size_t) T, sizeof(*th));
        int *started = calloc((size_t) T, sizeof(int));
        if (!W || !th || !started) { free(W); free(th); free(started); return SA_ENOMEM; }
        double total = 0, acc = 0;
        for (int64_t j = 0; j < n_jobs; j++) total += (double) (jobs[j].n_events > 0 ? jobs[j].n_events : 0) + 64.0;
        int64_t j = 0;
        for (int t = 0; t < T; t++) {
            int64_t j0 = j;
            double target = total * (double) (t + 1) / (double) T;
            while (j < n_jobs && (t == T - 1 || acc < target)) {
                acc += (double) (jobs[j].n_events > 0 ? jobs[j].n_events : 0) + 64.0;
                j++;
            }
            W[t].jobs = jobs + j0;
            W[t].n = j - j0;
            W[t].ambig = ambig;
            W[t].pl = plan_new(m, p, flags, W[t].n);
            if (!W[t].pl) rc = SA_ENOMEM;
        }
        /* round 1: exact sizes of the big arrays per thread (cap_* of the sub-plans receive the counts) */
        for (int t = 0; t < T && rc == SA_OK; t++) {
            if (pthread_create(&th[t], NULL, count_worker, &W[t]) == 0) started[t] = 1;
            else count_worker(&W[t]);
        }
        for (int t = 0; t < T; t++)
            if (started[t]) { pthread_join(th[t], NULL); started[t] = 0; }
        for (int t = 0; t < T && rc == SA_OK; t++) rc = W[t].rc;
        PLAN_TRACE("sizes counted");
        /* the final plan, its big arrays sized once; every sub-plan gets its slices */
        if (rc == SA_OK) {
            pl = plan_new(m, p, flags, n_jobs);
            if (!pl) rc = SA_ENOMEM;
        }
        if (rc == SA_OK) {
            int64_t tr = 0, tk = 0, to = 0, ti = 0, te = 0;
            for (int t = 0; t < T; t++) {
                tr += W[t].pl->cap_rows; tk += W[t].pl->cap_pk; to += W[t].pl->cap_poff; ti += W[t].pl->cap_pid; te += W[t].pl->cap_ev;
            }
            pl->pooled = 1;
            pl->cap_rows = tr; pl->cap_pk = tk; pl->cap_poff = to; pl->cap_pid = ti; pl->cap_ev = te;
            void *(*get)(size_t) = plan_alloc_hook ? plan_alloc_hook : plan_big_alloc;
            pl->big_free = plan_alloc_hook ? plan_free_hook : NULL;
            pl->rows = get(sizeof(sa_row_t) * (size_t) (tr > 0 ? tr : 1));
            pl->pk = get(sizeof(int32_t) * (size_t) (tk > 0 ? tk : 1));
            pl->poff = get(sizeof(int32_t) * (size_t) (to > 0 ? to : 1));
            pl->pid = get(sizeof(int32_t) * (size_t) (ti > 0 ? ti : 1));
            pl->xc = (flags & SA_FLAG_DEVICE_XC_INTERNAL) ? NULL : get(sizeof(double) * 4 * (size_t) (ti > 0 ? ti : 1));
            pl->ev = get(sizeof(double) * (size_t) (te > 0 ? te : 1));
            int want_prec = 0;
            for (int t = 0; t < T; t++) want_prec |= W[t].pl->prec_cap != 0;
            want_prec = want_prec && !(flags & (SA_FLAG_EXACT | SA_FLAG_FORCE_GENERIC)) &&
                        (!(flags & SA_FLAG_EXPECT_INTERNAL) || m->hdp == NULL) &&
                        (m->hdp == NULL ||   /* (HDP regions take the ring kernels too when the emission plane can be built) */
                         !(m->hdp->grid_length < 2 || m->hdp->n_slots * m->hdp->grid_length * 16 >= SA_HDP_FAST_MAX_BYTES)) &&
                        ring_env_on();
            if (want_prec) {
                pl->prec = get(sizeof(sa_prec_t) * (size_t) (ti > 0 ? ti : 1));
                if (!pl->prec) rc = SA_ENOMEM;
                else memset(pl->prec, 0, sizeof(sa_prec_t) * (size_t) (ti > 0 ? ti : 1));
                pl->prec_cap = ti;
            }
            if (!pl->rows || !pl->pk || !pl->poff || !pl->pid || (!pl->xc && !(flags & SA_FLAG_DEVICE_XC_INTERNAL)) || !pl->ev)
                rc = SA_ENOMEM;
            tr = tk = to = ti = te = 0;
            for (int t = 0; t < T && rc == SA_OK; t++) {
                sa_plan_t *s = W[t].pl;
                s->borrowed = 1;
                s->rows = pl->rows + tr; s->pk = pl->pk + tk; s->poff = pl->poff + to; s->pid = pl->pid + ti;
                s->xc = pl->xc ? pl->xc + 4 * ti : NULL; s->ev = pl->ev + te;
                s->prec = pl->prec ? pl->prec + ti : NULL; s->prec_cap = pl->prec ? s->cap_pid : 0;
                tr += s->cap_rows; tk += s->cap_pk; to += s->cap_poff; ti += s->cap_pid; te += s->cap_ev;
            }
        }
        /* round 2: the planning itself, in place */
        for (int t = 0; t < T && rc == SA_OK; t++) {
            if (pthread_create(&th[t], NULL, plan_worker, &W[t]) == 0) started[t] = 1;
            else plan_worker(&W[t]); /* no thread to be had: do it here */
        }
        for (int t = 0; t < T; t++)
            if (started[t]) pthread_join(th[t], NULL);
        PLAN_TRACE("sub-plans built");
        for (int t = 0; t < T && rc == SA_OK; t++) rc = W[t].rc; /* the first failing job in job order decides */
        /* round 3: the small arrays (jobs, regions, segments, checkpoints) are concatenated, offsets shifted */
        if (rc == SA_OK) {
            free(pl->jobs);
            pl->jobs = NULL;
            ALLOC_CAT(pl, jobs, n_jobs, sa_jobinfo_t);
            ALLOC_CAT(pl, regions, n_regions, sa_region_t);
            ALLOC_CAT(pl, segs, n_segs, sa_seg_t);
            ALLOC_CAT(pl, cks, n_cks, sa_ck_t);
        }
        if (rc == SA_OK) {
            merge_task_t *M = calloc((size_t) T, sizeof(*M));
            if (!M) rc = SA_ENOMEM;
            sa_plan_t b; /* running bases */
            memset(&b, 0, sizeof(b));
            int64_t job_base = 0;
            for (int t = 0; t < T && rc == SA_OK; t++) {
                sa_plan_t *s = W[t].pl;
                M[t].pl = pl; M[t].s = s; M[t].base = b; M[t].job_base = job_base;
                job_base += s->n_jobs;
                b.n_regions += s->n_regions; b.n_rows += s->n_rows; b.n_pk += s->n_pk; b.n_poff += s->n_poff;
                b.n_pid += s->n_pid; b.n_ev += s->n_ev; b.n_segs += s->n_segs; b.n_cks += s->n_cks;
                b.n_vbuf += s->n_vbuf; b.n_cand += s->n_cand; b.n_bscratch += s->n_bscratch;
                b.cells_fwd += s->cells_fwd; b.cells_bwd += s->cells_bwd; b.n_fast_regions += s->n_fast_regions;
                b.n_ring_regions += s->n_ring_regions;
                if (s->max_span > b.max_span) b.max_span = s->max_span;
            }
            if (rc == SA_OK) {
                for (int t = 0; t < T; t++) {
                    started[t] = 0;
                    if (pthread_create(&th[t], NULL, merge_worker, &M[t]) == 0) started[t] = 1;
                    else merge_worker(&M[t]);
                }
                for (int t = 0; t < T; t++)
                    if (started[t]) pthread_join(th[t], NULL);
                for (int t = 0; t < T; t++) W[t].pl = NULL; /* freed by their merge workers */
                pl->n_regions = pl->cap_regions = b.n_regions; pl->n_rows = pl->cap_rows = b.n_rows;
                pl->n_pk = pl->cap_pk = b.n_pk; pl->n_poff = pl->cap_poff = b.n_poff; pl->n_pid = pl->cap_pid = b.n_pid;
                pl->n_ev = pl->cap_ev = b.n_ev; pl->n_segs = pl->cap_segs = b.n_segs; pl->n_cks = pl->cap_cks = b.n_cks;
                pl->n_vbuf = b.n_vbuf; pl->n_cand = b.n_cand; pl->n_bscratch = b.n_bscratch;
                pl->cells_fwd = b.cells_fwd; pl->cells_bwd = b.cells_bwd; pl->n_fast_regions = b.n_fast_regions;
                pl->n_ring_regions = b.n_ring_regions;
                pl->max_span = b.max_span;
            }
            free(M);
        }
        PLAN_TRACE("merged");
        for (int t = 0; t < T; t++) sa_plan_free(W[t].pl);
        free(W); free(th); free(started);
        PLAN_TRACE("sub-plans freed");
    }
    if (rc != SA_OK) {
        sa_plan_free(pl);
        return rc;
    }
    sa_plan_repack(pl, chunk_budget);
    *out = pl;
    return SA_OK;
}

/* forward storage: pack regions into passes of at most chunk_budget cell-paths (a region larger than the budget gets a pass of
 * its own).  Also what a batch does again, with a smaller budget, when its working storage turns out not to fit the device. */
void sa_plan_repack(sa_plan_t *pl, int64_t chunk_budget) {
    int32_t chunk = 0;
    int64_t used = 0;
    pl->max_chunk_cellpaths = 0;
    for (int64_t r = 0; r < pl->n_regions; r++) {
        sa_region_t *R = &pl->regions[r];
        if (used > 0 && chunk_budget > 0 && used + R->f_cellpaths > chunk_budget) {
            chunk++;
            used = 0;
        }
        R->chunk = chunk;
        R->f_base = used;
        used += R->f_cellpaths;
        if (used > pl->max_chunk_cellpaths) pl->max_chunk_cellpaths = used;
    }
    pl->n_chunks = pl->n_regions ? chunk + 1 : 0;
}

void sa_plan_grow_candidates(sa_plan_t *pl, int factor) {
    int64_t off = 0;
    for (int64_t s = 0; s < pl->n_segs; s++) {
        int64_t cap = (int64_t) pl->segs[s].cand_cap * factor;
        if (cap > INT32_MAX) cap = INT32_MAX;
        pl->segs[s].cand_cap = (int32_t) cap;
        pl->segs[s].cand_off = off;
        off += cap;
    }
    pl->n_cand = off;
}

/* ---- finalisation: candidates + totals -> the reference's pair list ---------------------------- */
static int cmp_pair_out(const void *a, const void *b) {
    const sa_pair_t *p = a, *q = b;
    int64_t sp = (int64_t) p->x + p->y, sq = (int64_t) q->x + q->y;
    if (sp != sq) return sp < sq ? -1 : 1;
    if (p->x != q->x) return p->x > q->x ? -1 : 1;          /* within a diagonal: x descending  */
    if (p->path != q->path) return p->path > q->path ? -1 : 1; /* within a cell: path descending */
    return 0;
}

int sa_plan_finalize(const sa_plan_t *pl, const sa_cand_t *cands, const int32_t *cand_count, const double *totals,
                     sa_pair_t **pairs_out, int64_t *n_pairs_out) {
    double thr = pl->params.threshold;
    for (int64_t j = 0; j < pl->n_jobs; j++) {
        const sa_jobinfo_t *J = &pl->jobs[j];
        int64_t cap = 0;
        for (int64_t r = J->region_off; r < J->region_off + J->n_regions; r++) {
            const sa_region_t *R = &pl->regions[r];
            for (int64_t s = R->seg_off; s < R->seg_off + R->n_seg; s++) cap += cand_count[s];
        }
        sa_pair_t *out = malloc(sizeof(sa_pair_t) * (size_t) (cap > 0 ? cap : 1));
        if (!out) return SA_ENOMEM;
        int64_t n = 0;
        for (int64_t r = J->region_off; r < J->region_off + J->n_regions; r++) {
            const sa_region_t *R = &pl->regions[r];
            const int32_t *poff = pl->poff + R->poff_off;
            for (int64_t s = R->seg_off; s < R->seg_off + R->n_seg; s++) {
                const sa_seg_t *S = &pl->segs[s];
                const sa_cand_t *c = cands + S->cand_off;
                for (int32_t i = 0; i < cand_count[s]; i++) {
                    int64_t e = (int64_t) c[i].x + c[i].y + 2; /* diagonal in matrix coordinates */
                    int64_t ck = S->ck_base + (S->from - e) / SA_CKPT_EVERY;
                    double pp = exp(c[i].fb - totals[ck]); /* impl/pairwiseAligner.c:1385-1386 */
                    if (pp >= thr) {
                        if (pp > 1.0) pp = 1.0;
                        pp = floor(pp * SA_PROB_1);
                        sa_pair_t *o = &out[n++];
                        o->prob_e7 = (int64_t) pp;
                        o->x = (int32_t) (c[i].x + R->x1);
                        o->y = (int32_t) (c[i].y + R->y1);
                        o->path = c[i].path;
                        o->kmer_id = pl->pid[R->pid_off + poff[c[i].x + 1] + c[i].path];
                    }
                }
            }
        }
        qsort(out, (size_t) n, sizeof(sa_pair_t), cmp_pair_out);
        pairs_out[j] = out;
        n_pairs_out[j] = n;
    }
    return SA_OK;
}

/* ---- plan introspection ------------------------------------------------------------------------ */
int sa_plan_describe(const sa_model_t *m, const sa_params_t *p, const sa_job_t *job, const char *const *ambig,
                     unsigned flags, sa_plan_info_t *info, int64_t *regions4, int64_t regions_cap, int64_t *rows3,
                     int64_t rows_cap, int64_t *segs4, int64_t segs_cap) {
    sa_plan_t *pl = NULL;
    int rc = sa_plan_build(&pl, m, p, job, 1, ambig, flags, 0);
    if (rc) return rc;
    if (info) {
        info->n_regions = pl->n_regions;
        info->n_segments = pl->n_segs;
        info->n_checkpoints = pl->n_cks;
        info->cells_forward = pl->cells_fwd;
        info->cells_backward = pl->cells_bwd;
        info->f_cellpaths = pl->max_chunk_cellpaths;
        info->max_span = pl->max_span;
        info->n_fast_regions = pl->n_fast_regions;
        info->n_ring_regions = pl->n_ring_regions;
    }
    int64_t nrow = 0;
    for (int64_t r = 0; r < pl->n_regions; r++) {
        const sa_region_t *R = &pl->regions[r];
        if (regions4 && r < regions_cap) {
            regions4[4 * r] = R->x1; regions4[4 * r + 1] = R->y1;
            regions4[4 * r + 2] = R->x1 + R->lX; regions4[4 * r + 3] = R->y1 + R->lY;
        }
        for (int64_t d = 0; d <= R->N; d++, nrow++)
            if (rows3 && nrow < rows_cap) {
                const sa_row_t *w = &pl->rows[R->row_off + d];
                rows3[3 * nrow] = r;
                rows3[3 * nrow + 1] = w->xmyL;
                rows3[3 * nrow + 2] = w->xmyL + 2 * ((int64_t) w->width - 1);
            }
    }
    for (int64_t s = 0; s < pl->n_segs; s++)
        if (segs4 && s < segs_cap) {
            segs4[4 * s] = pl->segs[s].region; segs4[4 * s + 1] = pl->segs[s].start;
            segs4[4 * s + 2] = pl->segs[s].from; segs4[4 * s + 3] = pl->segs[s].to;
        }
    sa_plan_free(pl);
    return SA_OK;
}

/* Test hook (host only): plans one job and checks the per-path records of its ring-kernel regions against the definition
 * they abbreviate -- path q of column x-1 and path p of column x are neighbours iff q's k-mer minus its first letter equals
 * p's k-mer minus its last (path_checkKmerLegalTransition, impl/pairwiseAligner.c:595-608), the NULL k-mer of column 0
 * being everybody's neighbour.  Returns the number of (column, path) entries whose predecessor or successor set differs,
 * or a negative error; *n_checked receives the number of entries looked at (0: the job has no such region). */
int64_t sa_plan_check_path_records(const sa_model_t *m, const sa_params_t *p, const sa_job_t *job, const char *const *ambig,
                                   int64_t *n_checked) {
    sa_plan_t *pl = NULL;
    int rc = sa_plan_build(&pl, m, p, job, 1, ambig, 0, 0);
    if (rc) return rc;
    int64_t bad = 0, seen = 0;
    for (int64_t r = 0; r < pl->n_regions; r++) {
        const sa_region_t *R = &pl->regions[r];
        if (R->kind != SA_KIND_RING || R->max_p <= 1) continue;
        const int32_t *poff = pl->poff + R->poff_off, *pid = pl->pid + R->pid_off;
        const sa_prec_t *pr = pl->prec + R->pid_off;
        for (int64_t x = 0; x <= R->lX; x++)
            for (int32_t g = poff[x]; g < poff[x + 1]; g++, seen++) {
                const sa_prec_t *o = &pr[g];
                int ok = o->x == x;
                const int npred = (int) ((o->meta >> 8) & 255u), nsucc = (int) (o->meta & 255u), stride = (int) (o->meta >> 16);
                /* predecessors: column x-1 */
                if (x >= 1) {
                    int64_t cnt = 0;
                    for (int32_t q = poff[x - 1]; q < poff[x]; q++) {
                        const int legal = pid[q] < 0 || pid[g] < 0 || (pid[q] % m->pow_km1) == (pid[g] / m->n_alpha);
                        int listed = 0;
                        for (int i = 0; i < npred; i++) listed |= (o->pred0 + i * stride) == q;
                        ok &= legal == listed;
                        cnt += legal;
                    }
                    ok &= cnt == npred;
                } else {
                    ok &= o->pred0 < 0 || npred == 0;
                }
                if (x < R->lX) {
                    int64_t cnt = 0;
                    for (int32_t q = poff[x + 1]; q < poff[x + 2]; q++) {
                        const int legal = pid[q] < 0 || pid[g] < 0 || (pid[g] % m->pow_km1) == (pid[q] / m->n_alpha);
                        const int listed = q >= o->succ0 && q < o->succ0 + nsucc;
                        ok &= legal == listed;
                        cnt += legal;
                    }
                    ok &= cnt == nsucc;
                } else {
                    ok &= o->succ0 < 0 || nsucc == 0;
                }
                bad += !ok;
            }
    }
    if (n_checked) *n_checked = seen;
    sa_plan_free(pl);
    return bad;
}

static uint64_t fnv1a(uint64_t h, const void *data, size_t n) {
    const unsigned char *b = data;
    for (size_t i = 0; i < n; i++) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

int sa_plan_digest(const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs,
                   const char *const *ambig, unsigned flags, int threads, sa_plan_info_t *info, uint64_t *digest) {
    sa_plan_t *pl = NULL;
    plan_threads_override = threads;
    int rc = sa_plan_build(&pl, m, p, jobs, n_jobs, ambig, flags, 0);
    plan_threads_override = 0;
    if (rc) return rc;
    if (info) {
        info->n_regions = pl->n_regions;
        info->n_segments = pl->n_segs;
        info->n_checkpoints = pl->n_cks;
        info->cells_forward = pl->cells_fwd;
        info->cells_backward = pl->cells_bwd;
        info->f_cellpaths = pl->max_chunk_cellpaths;
        info->max_span = pl->max_span;
        info->n_fast_regions = pl->n_fast_regions;
        info->n_ring_regions = pl->n_ring_regions;
    }
    if (digest) {
        uint64_t h = 1469598103934665603ull;
        h = fnv1a(h, pl->jobs, sizeof(sa_jobinfo_t) * (size_t) pl->n_jobs);
        h = fnv1a(h, pl->regions, sizeof(sa_region_t) * (size_t) pl->n_regions);
        h = fnv1a(h, pl->rows, sizeof(sa_row_t) * (size_t) pl->n_rows);
        h = fnv1a(h, pl->pk, sizeof(int32_t) * (size_t) pl->n_pk);
        h = fnv1a(h, pl->poff, sizeof(int32_t) * (size_t) pl->n_poff);
        h = fnv1a(h, pl->pid, sizeof(int32_t) * (size_t) pl->n_pid);
        h = fnv1a(h, pl->xc, sizeof(double) * 4 * (size_t) pl->n_pid);
        h = fnv1a(h, pl->ev, sizeof(double) * (size_t) pl->n_ev);
        h = fnv1a(h, pl->segs, sizeof(sa_seg_t) * (size_t) pl->n_segs);
        h = fnv1a(h, pl->cks, sizeof(sa_ck_t) * (size_t) pl->n_cks);
        for (int64_t r = 0; r < pl->n_regions; r++) { /* per-path records exist for these regions only */
            const sa_region_t *R = &pl->regions[r];
            if (R->kind == SA_KIND_RING && R->max_p > 1)
                h = fnv1a(h, pl->prec + R->pid_off, sizeof(sa_prec_t) * (size_t) pl->poff[R->poff_off + R->lX + 1]);
        }
        int64_t tail[4] = {pl->n_vbuf, pl->n_cand, pl->n_bscratch, pl->n_chunks};
        h = fnv1a(h, tail, sizeof(tail));
        *digest = h;
    }
    sa_plan_free(pl);
    return SA_OK;
}

/* ---- anchors ----------------------------------------------------------------------------------- */
typedef struct { int64_t x, y; } pt_t;
static int cmp_pt(const void *a, const void *b) {
    const pt_t *p = a, *q = b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    return p->y < q->y ? -1 : (p->y > q->y ? 1 : 0);
}

/* Keeps the pairs that are strictly below-left of everything after them (backward pass: a pair
 * survives if both coordinates are smaller than every later minimum) and strictly above-right of
 * everything before them (forward pass over running maxima). Input sorted lexicographically. */
static int64_t drop_overlaps(const pt_t *in, int64_t n, int64_t *ox, int64_t *oy) {
    uint8_t *ok = calloc((size_t) (n > 0 ? n : 1), 1);
    int64_t mx = INT64_MAX, my = INT64_MAX;
    for (int64_t i = n - 1; i >= 0; i--) {
        /* value-keyed membership: an equal pair right after i that was accepted makes i "found" too */
        if (in[i].x < mx && in[i].y < my) ok[i] = 1;
        else if (i + 1 < n && ok[i + 1] && in[i + 1].x == in[i].x && in[i + 1].y == in[i].y) ok[i] = 1;
        if (in[i].x < mx) mx = in[i].x;
        if (in[i].y < my) my = in[i].y;
    }
    int64_t m = 0, hx = INT64_MIN, hy = INT64_MIN;
    for (int64_t i = 0; i < n; i++) {
        if (ok[i] && in[i].x > hx && in[i].y > hy) {
            ox[m] = in[i].x;
            oy[m] = in[i].y;
            m++;
        }
        if (in[i].x > hx) hx = in[i].x;
        if (in[i].y > hy) hy = in[i].y;
    }
    free(ok);
    return m;
}

int64_t sa_guide_to_anchors(int64_t start1, int64_t end1, int strand1, int64_t start2, const int32_t *op_type,
                            const int64_t *op_len, int64_t n_ops, int64_t trim, int64_t *ax, int64_t *ay, int64_t cap) {
    /* rebase the reference interval to 0; a minus-strand hit is flipped so it reads forward */
    int64_t origin = strand1 ? start1 : end1;
    int64_t ref_end = strand1 ? end1 - origin : start1 - origin;
    int64_t ref_pos = strand1 ? start1 - origin : end1 - origin; /* == 0 */
    int64_t read_pos = start2;
    int64_t total = 0;
    for (int64_t i = 0; i < n_ops; i++)
        if (op_type[i] == 0 && op_len[i] > 2 * trim) total += op_len[i] - 2 * trim;
    pt_t *pts = malloc(sizeof(pt_t) * (size_t) (total > 0 ? total : 1));
    if (!pts) return SA_ENOMEM;
    int64_t n = 0;
    for (int64_t i = 0; i < n_ops; i++) {
        if (op_type[i] == 0)
            for (int64_t l = trim; l < op_len[i] - trim; l++)
                if (ref_end >= ref_pos + l + 6) { /* hard-coded +6: impl/pairwiseAligner.c:1642 */
                    pts[n].x = ref_pos + l;
                    pts[n].y = read_pos + l;
                    n++;
                }
        if (op_type[i] != 2) ref_pos += op_len[i];
        if (op_type[i] != 1) read_pos += op_len[i];
    }
    qsort(pts, (size_t) n, sizeof(pt_t), cmp_pt);
    int64_t *tx = malloc(sizeof(int64_t) * (size_t) (n + 1)), *ty = malloc(sizeof(int64_t) * (size_t) (n + 1));
    int64_t m = drop_overlaps(pts, n, tx, ty);
    if (m > cap) m = cap;
    memcpy(ax, tx, sizeof(int64_t) * (size_t) m);
    memcpy(ay, ty, sizeof(int64_t) * (size_t) m);
    free(pts); free(tx); free(ty);
    return m;
}

int64_t sa_remap_anchors(const int64_t *ax, const int64_t *ay, int64_t n, const int64_t *event_map, int64_t map_offset,
                         int64_t *ox, int64_t *oy) {
    pt_t *pts = malloc(sizeof(pt_t) * (size_t) (n > 0 ? n : 1));
    if (!pts) return SA_ENOMEM;
    for (int64_t i = 0; i < n; i++) {
        pts[i].x = ax[i];
        pts[i].y = event_map[ay[i]] - event_map[map_offset];
    }
    int64_t m = drop_overlaps(pts, n, ox, oy);
    free(pts);
    return m;
}

/* ---- per-read parameter re-estimation ---------------------------------------------------------- */
/* Gaussian elimination as the reference writes it, including the pivot swap that copies instead of
 * swapping the right-hand side (impl/nanopore.c:692-753). */
static int solve_small(const double *A, const double *b, double *x, int n) {
    double M[9];
    for (int i = 0; i < n; i++) {
        x[i] = b[i];
        for (int j = 0; j < n; j++) M[i * n + j] = A[i * n + j];
    }
    const double eps = 1.11022302462515654042E-16;
    for (int i = 0; i < n; i++) {
        if (fabs(M[i * n + i]) < eps) {
            int s = i + 1;
            while (s < n && M[s * n + i] < eps) s++;
            if (s >= n) return SA_EINVAL; /* "Matrix is not invertible." */
            for (int j = 0; j < n; j++) {
                double t = M[i * n + j];
                M[i * n + j] = M[s * n + j];
                M[s * n + j] = t;
            }
            x[i] = x[s];
        }
        double f = 1.0 / M[i * n + i];
        x[i] *= f;
        for (int j = 0; j < n; j++) M[i * n + j] *= f;
        for (int r = i + 1; r < n; r++) {
            f = M[r * n + i];
            x[r] -= f * x[i];
            for (int j = 0; j < n; j++) M[r * n + j] -= f * M[i * n + j];
        }
    }
    for (int i = n - 1; i >= 0; i--)
        for (int r = i - 1; r >= 0; r--) {
            double f = M[r * n + i];
            x[r] -= f * x[i];
            for (int j = 0; j < n; j++) M[r * n + j] -= f * M[i * n + j];
        }
    return SA_OK;
}

int sa_estimate_params(const sa_model_t *m, double *tab, const int64_t *emap, double *ev4, int64_t n_events,
                       const char *read, int64_t read_len, double *out7) {
    if (!m || !tab || !emap || !ev4 || !read || !out7) return SA_EINVAL;
    int64_t rows = read_len - (m->k - 1);
    if (rows <= 0) return SA_EINVAL;
    /* first event of every k-mer of the strand read (1-D assignments) */
    int64_t *kid = malloc(sizeof(int64_t) * (size_t) rows), *eid = malloc(sizeof(int64_t) * (size_t) rows);
    if (!kid || !eid) { free(kid); free(eid); return SA_ENOMEM; }
    int64_t n = 0, last = -1;
    for (int64_t i = 0; i < rows; i++) {
        int64_t id = sa_model_kmer_id(m, read + i);
        if (id < 0) { free(kid); free(eid); return SA_EALPHABET; }
        if (emap[i] > last) {
            if (emap[i] >= n_events) { free(kid); free(eid); return SA_EINVAL; }
            kid[n] = id;
            eid[n] = emap[i];
            n++;
            last = emap[i];
        }
    }
    if (n == 0) { free(kid); free(eid); return SA_EINVAL; } /* "Cannot get scale params with no assignments" */
    /* weighted least squares: event_mean ~ shift + scale*level_mean + drift*start_time, weights 1/level_sd^2 */
    double G[9] = {0}, g[3] = {0}, beta[3];
    for (int64_t i = 0; i < n; i++) {
        double mu = tab[kid[i] * 5], sd = tab[kid[i] * 5 + 1];
        double e = ev4[eid[i] * 4], t = ev4[eid[i] * 4 + 3];
        double w = 1.0 / (sd * sd);
        double wm = mu * w, wt = t * w;
        G[0] += w; G[1] += wm; G[2] += wt; G[4] += wm * mu; G[5] += wm * t; G[8] += wt * t;
        g[0] += w * e; g[1] += wm * e; g[2] += wt * e;
    }
    G[3] = G[1]; G[6] = G[2]; G[7] = G[5];
    int rc = solve_small(G, g, beta, 3);
    if (rc) { free(kid); free(eid); return rc; }
    double disp = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double mu = tab[kid[i] * 5], sd = tab[kid[i] * 5 + 1];
        double pred = beta[0] + beta[1] * mu + beta[2] * ev4[eid[i] * 4 + 3];
        double res = ev4[eid[i] * 4] - pred;
        disp += (res * res) / (sd * sd);
    }
    double var = sqrt(disp / n);
    /* noise: event_sd ~ shift_sd + scale_sd*noise_mean, weights 1/noise_sd^2 */
    double H[4] = {0}, h[2] = {0}, gam[2];
    for (int64_t i = 0; i < n; i++) {
        double nm = tab[kid[i] * 5 + 2], ns = tab[kid[i] * 5 + 3];
        double w = 1.0 / (ns * ns), wm = nm * w, s = ev4[eid[i] * 4 + 1];
        H[0] += w; H[1] += wm; H[3] += wm * nm;
        h[0] += w * s; h[1] += wm * s;
    }
    H[2] = H[1];
    rc = solve_small(H, h, gam, 2);
    if (rc) { free(kid); free(eid); return rc; }
    disp = 0.0;
    for (int64_t i = 0; i < n; i++) {
        double nm = tab[kid[i] * 5 + 2], ns = tab[kid[i] * 5 + 3];
        double res = ev4[eid[i] * 4 + 1] - (gam[0] + gam[1] * nm);
        disp += (res * res) / (ns * ns);
    }
    double var_sd = sqrt(disp / n);
    out7[0] = beta[1]; out7[1] = beta[0]; out7[2] = var; out7[3] = beta[2];
    out7[4] = gam[1]; out7[5] = var_sd; out7[6] = gam[0];
    /* drift correction of every template event, then the noise columns of the table */
    for (int64_t i = 0; i < n_events; i++) ev4[i * 4] = ev4[i * 4] - (ev4[i * 4 + 3] * beta[2]);
    for (int64_t i = 0; i < m->n_kmers * 5; i += 5) {
        tab[i + 2] = tab[i + 2] * gam[1];
        tab[i + 4] = tab[i + 4] * var_sd;
        tab[i + 3] = sqrt(pow(tab[i + 2], 3.0) / tab[i + 4]);
    }
    free(kid);
    free(eid);
    return SA_OK;
}
