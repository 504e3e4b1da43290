/* sa_io.c -- loaders for the reference's text formats (host only).
 *
 *   .model   3 whitespace-split lines (stateMachine3_loadFromFile, impl/stateMachine.c:1440-1538)
 *   .nhdp    serialize_nhdp/serialize_hdp text (impl/nanopore_hdp.c:1088-1115, impl/hdp.c:2919-3322);
 *            only the fields alignment reads are kept: grid, parents, observed flags (recomputed from the
 *            dp-id line as mark_observed_dps does, impl/hdp.c:1132-1160), posterior predictives, slopes.
 */
#define _GNU_SOURCE
#include "sa_io.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/types.h>
#include <string.h>

#include "sa_internal.h"

/* one line without its terminator (NULL at end of file); getline() scans the stdio buffer with memchr -- a character
 * at a time through fgetc() cost 10 ms per 600 KB .npRead, most of the CLI's host stage */
char *sa_read_line(FILE *f) {
    char *s = NULL;
    size_t cap = 0;
    ssize_t n = getline(&s, &cap, f);
    if (n < 0) { free(s); return NULL; }
    if (n > 0 && s[n - 1] == '\n') s[--n] = 0;
    return s;
}

/* Decimal text to double, exactly as strtod rounds it, for the numbers the formats hold: up to 15 significant digits and a
 * decimal exponent within +-22 convert with ONE correctly rounded multiplication or division (both operands are exact
 * doubles: Clinger's fast path); anything else -- more digits, hex, inf/nan, huge exponents -- goes to strtod. */
/* printf("%f") of a double, digit for digit (glibc: the exact binary value rounded to six decimals, ties to even), without
 * going through the stdio formatter: the row writers of signalMachine print nine of these per aligned pair, and the formatter
 * was most of their time.  |v| < 2^53: the integer part and the fraction are exact doubles; the fraction is m * 2^-sh with a
 * 53-bit m, so m * 10^6 fits 128 bits and the shift's remainder decides the rounding exactly.  Anything else (inf, nan, huge
 * values) goes to snprintf.  Returns the number of characters written (no terminator needed by the callers, one is written). */
int sa_format_f6(char *out, double v) {
    if (!(fabs(v) < 9.0e15)) return sprintf(out, "%f", v);
    char *p = out;
    if (signbit(v)) { *p++ = '-'; v = -v; }
    uint64_t ip = (uint64_t) v;
    const double fr = v - (double) ip;
    uint64_t q = 0;
    if (fr > 0.0) {
        int ex;
        const double mant = frexp(fr, &ex);                 /* fr = mant * 2^ex, mant in [0.5, 1), ex <= 0 */
        const uint64_t m = (uint64_t) ldexp(mant, 53);
        const int sh = 53 - ex;                             /* fr = m * 2^-sh, sh >= 53 */
        if (sh < 127) {
            const unsigned __int128 num = (unsigned __int128) m * 1000000u, one = (unsigned __int128) 1 << sh;
            const unsigned __int128 rem = num & (one - 1), half = one >> 1;
            q = (uint64_t) (num >> sh);
            if (rem > half || (rem == half && (q & 1))) q++;
        }                                                   /* (else: below 2^-73, rounds to 0) */
        if (q == 1000000) { q = 0; ip++; }
    }
    char t[24];
    int n = 0;
    do { t[n++] = (char) ('0' + ip % 10); ip /= 10; } while (ip);
    while (n) *p++ = t[--n];
    *p++ = '.';
    for (int i = 5; i >= 0; i--) { p[i] = (char) ('0' + q % 10); q /= 10; }
    p += 6;
    *p = 0;
    return (int) (p - out);
}

static double sa_atod(const char *p) {
    static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                                   1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char *s = p;
    int neg = 0;
    if (*s == '-') { neg = 1; s++; } else if (*s == '+') s++;
    uint64_t mant = 0;
    int digits = 0, exp10 = 0, any = 0;
    while (*s >= '0' && *s <= '9') {
        if (mant || *s != '0') { mant = mant * 10 + (uint64_t) (*s - '0'); digits++; }
        any = 1; s++;
        if (digits > 15) return strtod(p, NULL);
    }
    if (*s == '.') {
        s++;
        while (*s >= '0' && *s <= '9') {
            if (mant || *s != '0') { mant = mant * 10 + (uint64_t) (*s - '0'); digits++; }
            exp10--; any = 1; s++;
            if (digits > 15) return strtod(p, NULL);
        }
    }
    if (!any) return strtod(p, NULL);
    if (*s == 'e' || *s == 'E') {
        s++;
        int eneg = 0, e = 0, edig = 0;
        if (*s == '-') { eneg = 1; s++; } else if (*s == '+') s++;
        while (*s >= '0' && *s <= '9' && edig < 5) { e = e * 10 + (*s - '0'); s++; edig++; }
        if (!edig || (*s >= '0' && *s <= '9')) return strtod(p, NULL);
        exp10 += eneg ? -e : e;
    }
    if (*s != 0 || exp10 < -22 || exp10 > 22) return strtod(p, NULL);
    double v = (double) mant;
    v = exp10 < 0 ? v / p10[-exp10] : v * p10[exp10];
    return neg ? -v : v;
}

int64_t sa_split_ws(char *line, char ***toks) {
    int64_t cap = 256, n = 0;
    char **t = malloc(sizeof(char *) * (size_t) cap);
    *toks = NULL;
    if (!t) return -1;
    char *p = line;
    for (;;) {
        while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
        if (!*p) break;
        if (n == cap) {
            cap *= 2;
            char **t2 = realloc(t, sizeof(char *) * (size_t) cap);
            if (!t2) { free(t); return -1; }
            t = t2;
        }
        t[n++] = p;
        while (*p && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') p++;
        if (*p) *p++ = 0;
    }
    *toks = t;
    return n;
}

typedef struct {
    int64_t num_dps, grid_length;
    double grid_start, grid_stop;
    int64_t *parent;
    uint8_t *observed;
    double **post, **slope;
    char alphabet[64];
    int n_alpha, k;
} nhdp_file_t;

static void nhdp_file_free(nhdp_file_t *h) {
    if (!h) return;
    if (h->post) for (int64_t i = 0; i < h->num_dps; i++) free(h->post[i]);
    if (h->slope) for (int64_t i = 0; i < h->num_dps; i++) free(h->slope[i]);
    free(h->post); free(h->slope); free(h->parent); free(h->observed);
    free(h);
}

static double *parse_doubles(char *line, int64_t expect) {
    char **tok;
    int64_t n = sa_split_ws(line, &tok);
    double *v = NULL;
    if (n == expect && n >= 0) {
        v = malloc(sizeof(double) * (size_t) (n > 0 ? n : 1));
        for (int64_t i = 0; v && i < n; i++) v[i] = sa_atod(tok[i]);
    }
    free(tok);
    return v;
}

static nhdp_file_t *nhdp_read(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    nhdp_file_t *h = calloc(1, sizeof(*h));
    if (!h) { fclose(f); return NULL; }
    char *ln;
    int64_t n_data = 0, *dp_ids = NULL;
#define NEXT() do { ln = sa_read_line(f); if (!ln) goto bad; } while (0)
    NEXT(); h->n_alpha = (int) strtol(ln, NULL, 10); free(ln);
    NEXT(); sscanf(ln, "%63s", h->alphabet); free(ln);
    NEXT(); h->k = (int) strtol(ln, NULL, 10); free(ln);
    NEXT(); int splines = strtol(ln, NULL, 10) != 0; free(ln);
    NEXT(); int has_data = strtol(ln, NULL, 10) != 0; free(ln);
    NEXT(); int sample_gamma = strtol(ln, NULL, 10) != 0; free(ln);
    NEXT(); h->num_dps = strtoll(ln, NULL, 10); free(ln);
    if (h->num_dps <= 0 || h->num_dps > ((int64_t) 1 << 31) || h->n_alpha < 1 || h->n_alpha > 60 || h->k < 1 || h->k > 12)
        goto bad;
    if (has_data) {
        NEXT(); free(ln); /* data values */
        NEXT();
        char **tok;
        n_data = sa_split_ws(ln, &tok);
        if (n_data < 0) { free(ln); goto bad; }
        dp_ids = malloc(sizeof(int64_t) * (size_t) (n_data > 0 ? n_data : 1));
        if (!dp_ids) { free(tok); free(ln); goto bad; }
        for (int64_t i = 0; i < n_data; i++) dp_ids[i] = strtoll(tok[i], NULL, 10);
        free(tok);
        free(ln);
    }
    NEXT(); free(ln); /* mu nu alpha beta */
    NEXT();
    long long gl = 0;
    if (sscanf(ln, "%lg %lg %lld", &h->grid_start, &h->grid_stop, &gl) != 3 || gl < 2 || gl > (1 << 24)) { free(ln); goto bad; }
    h->grid_length = gl;
    free(ln);
    NEXT(); free(ln); /* gamma */
    if (sample_gamma) for (int i = 0; i < 4; i++) { NEXT(); free(ln); }
    h->parent = malloc(sizeof(int64_t) * (size_t) h->num_dps);
    h->observed = calloc((size_t) h->num_dps, 1);
    h->post = calloc((size_t) h->num_dps, sizeof(double *));
    h->slope = calloc((size_t) h->num_dps, sizeof(double *));
    if (!h->parent || !h->observed || !h->post || !h->slope) goto bad;
    for (int64_t id = 0; id < h->num_dps; id++) {
        NEXT();
        h->parent[id] = ln[0] == '-' ? -1 : strtoll(ln, NULL, 10);
        free(ln);
    }
    if (has_data) {
        for (int64_t i = 0; i < n_data; i++) /* every named DP and all its ancestors */
            for (int64_t a = dp_ids[i], guard = 0; a >= 0 && a < h->num_dps && !h->observed[a] && guard <= h->num_dps;
                 a = h->parent[a], guard++)
                h->observed[a] = 1;
        for (int64_t id = 0; id < h->num_dps; id++) {
            NEXT();
            h->post[id] = parse_doubles(ln, h->grid_length);
            if (!h->post[id] && h->observed[id]) h->post[id] = calloc((size_t) h->grid_length, sizeof(double));
            free(ln);
        }
    }
    if (splines)
        for (int64_t id = 0; id < h->num_dps; id++) {
            NEXT();
            h->slope[id] = parse_doubles(ln, h->grid_length);
            free(ln);
        }
    free(dp_ids);
    fclose(f);
    return h;
bad:
    free(dp_ids);
    fclose(f);
    nhdp_file_free(h);
    return NULL;
#undef NEXT
}

int sa_model_load(sa_model_t **out, const char *model_path, const char *nhdp_path) {
    if (!out || !model_path) return SA_EINVAL;
    FILE *f = fopen(model_path, "r");
    if (!f) return SA_EIO;
    char *l0 = sa_read_line(f), *l1 = sa_read_line(f), *l2 = sa_read_line(f);
    fclose(f);
    int rc = SA_EIO;
    char **t0 = NULL, **t1 = NULL, **t2 = NULL;
    double *table = NULL;
    nhdp_file_t *h = NULL;
    if (!l0 || !l1 || !l2) goto done;
    if (sa_split_ws(l0, &t0) != 4) goto done; /* stateNumber alphabetSize alphabet kmerLength */
    int n_states = (int) strtol(t0[0], NULL, 10), n_alpha = (int) strtol(t0[1], NULL, 10), k = (int) strtol(t0[3], NULL, 10);
    if ((int) strlen(t0[2]) != n_alpha || n_alpha < 1 || n_alpha > 60 || k < 1 || k > 12) goto done;
    if (n_states != 3) { rc = SA_EUNSUPPORTED; goto done; } /* the 5-state machine aborts in the reference (impl/stateMachine.c:907) */
    if (sa_split_ws(l1, &t1) != 10) goto done;               /* 3 x 3 transitions + 1 */
    double t10[10];
    for (int i = 0; i < 10; i++) t10[i] = strtod(t1[i], NULL);
    int64_t nk = 1;
    for (int i = 0; i < k; i++) nk *= n_alpha;
    int64_t n2 = sa_split_ws(l2, &t2);
    if (n2 != nk * 5) goto done;
    table = malloc(sizeof(double) * (size_t) n2);
    if (!table) { rc = SA_ENOMEM; goto done; }
    for (int64_t i = 0; i < n2; i++) table[i] = sa_atod(t2[i]);
    if (nhdp_path) {
        h = nhdp_read(nhdp_path);
        if (!h) goto done;
        sa_hdp_desc_t d = {h->num_dps, h->grid_length, h->grid_start, h->grid_stop, h->parent, h->observed,
                           (const double *const *) h->post, (const double *const *) h->slope};
        rc = sa_model_create(out, n_states, t0[2], k, t10, table, &d);
        if (rc == SA_OK) { /* stateMachine3_setModelToHdpExpectedValues checks (impl/stateMachine.c:1277-1288) */
            char a[64]; int na, kk;
            sa_model_alphabet(*out, a, &na, &kk);
            char b[64];
            strncpy(b, h->alphabet, 63); b[63] = 0;
            for (int i = 1; i < h->n_alpha; i++) { char c = b[i]; int j = i - 1; while (j >= 0 && b[j] > c) { b[j + 1] = b[j]; j--; } b[j + 1] = c; }
            if (strcmp(a, b) != 0 || kk != h->k) { sa_model_destroy(*out); *out = NULL; rc = SA_EINVAL; }
        }
    } else {
        rc = sa_model_create(out, n_states, t0[2], k, t10, table, NULL);
    }
done:
    free(t0); free(t1); free(t2); free(l0); free(l1); free(l2); free(table);
    nhdp_file_free(h);
    return rc;
}

/* ---- .npRead --------------------------------------------------------------------------------- */
static int parse_i64_line(char *line, int64_t expect, int64_t **out) {
    char **tok;
    int64_t n = sa_split_ws(line, &tok);
    if (n != expect || n < 0) { free(tok); return SA_EIO; }
    int64_t *v = malloc(sizeof(int64_t) * (size_t) (n > 0 ? n : 1));
    if (!v) { free(tok); return SA_ENOMEM; }
    for (int64_t i = 0; i < n; i++) v[i] = strtoll(tok[i], NULL, 10);
    free(tok);
    *out = v;
    return SA_OK;
}

void sa_npread_free(sa_npread_t *r) {
    if (!r) return;
    free(r->two_d_read); free(r->template_read); free(r->complement_read);
    free(r->template_strand_event_map); free(r->complement_strand_event_map);
    free(r->template_event_map); free(r->complement_event_map);
    free(r->template_events); free(r->complement_events);
    free(r);
}

/* A sequence line of the .npRead: its first token, cut at the declared length as the reference does.  NULL when the
 * declared length is negative or the line holds fewer characters than declared (the k-mer walk of the parameter estimation
 * reads declared-length - k + 1 k-mers). */
static char *first_token_copy(const char *line, int64_t want_len) {
    if (want_len < 0) return NULL;
    while (*line == ' ' || *line == '\t') line++;
    size_t n = strcspn(line, " \t\r\n");
    if ((int64_t) n < want_len) return NULL;
    char *s = malloc((size_t) want_len + 1);
    if (!s) return NULL;
    memcpy(s, line, (size_t) want_len);
    s[want_len] = 0;
    return s;
}

int sa_npread_load(const char *path, sa_npread_t **out) {
    FILE *f = fopen(path, "r");
    if (!f) return SA_EIO;
    sa_npread_t *r = calloc(1, sizeof(*r));
    if (!r) { fclose(f); return SA_ENOMEM; }
    char *ln[14];
    memset(ln, 0, sizeof(ln));
    int rc = SA_EIO;
    for (int i = 0; i < 14; i++) {
        ln[i] = sa_read_line(f);
        if (!ln[i] && i < 10) goto done; /* the model_state / p_model lines are not used by the aligner */
    }
    {
        char **t;
        char *h = strdup(ln[0]);
        int64_t n = sa_split_ws(h, &t);
        if (n != 18) { free(t); free(h); goto done; }
        r->read_length = strtoll(t[0], NULL, 10);
        r->n_template_events = strtoll(t[1], NULL, 10);
        r->n_complement_events = strtoll(t[2], NULL, 10);
        r->template_read_length = strtoll(t[3], NULL, 10);
        r->complement_read_length = strtoll(t[4], NULL, 10);
        sa_strand_params_t *p[2] = {&r->template_params, &r->complement_params};
        for (int s = 0; s < 2; s++) {
            p[s]->scale = strtod(t[5 + 6 * s], NULL);
            p[s]->shift = strtod(t[6 + 6 * s], NULL);
            p[s]->var = strtod(t[7 + 6 * s], NULL);
            p[s]->scale_sd = strtod(t[8 + 6 * s], NULL);
            p[s]->var_sd = strtod(t[9 + 6 * s], NULL);
            p[s]->drift = strtod(t[10 + 6 * s], NULL);
            p[s]->shift_sd = 0.0;
        }
        r->two_d = (int) strtol(t[17], NULL, 10);
        free(t);
        free(h);
    }
    if (r->read_length < 0 || r->n_template_events < 0 || r->n_complement_events < 0 || r->template_read_length < 0 ||
        r->complement_read_length < 0)
        goto done;
    r->two_d_read = r->two_d ? first_token_copy(ln[1], r->read_length) : strdup("");
    r->template_read = first_token_copy(ln[2], r->template_read_length);
    if (!r->two_d_read || !r->template_read) goto done;
    if (parse_i64_line(ln[3], r->template_read_length, &r->template_strand_event_map)) goto done;
    r->complement_read = r->two_d ? first_token_copy(ln[4], r->complement_read_length) : strdup("");
    if (!r->complement_read) goto done;
    if (parse_i64_line(ln[5], r->complement_read_length, &r->complement_strand_event_map)) goto done;
    if (parse_i64_line(ln[6], r->read_length, &r->template_event_map)) goto done;
    r->template_events = parse_doubles(ln[7], r->n_template_events * 4);
    if (!r->template_events && r->n_template_events) goto done;
    if (parse_i64_line(ln[8], r->read_length, &r->complement_event_map)) goto done;
    r->complement_events = parse_doubles(ln[9], r->n_complement_events * 4);
    if (!r->complement_events && r->n_complement_events) goto done;
    rc = SA_OK;
done:
    for (int i = 0; i < 14; i++) free(ln[i]);
    fclose(f);
    if (rc) { sa_npread_free(r); return rc; }
    *out = r;
    return SA_OK;
}

/* ---- cigar -------------------------------------------------------------------------------------- */
void sa_cigar_free(sa_cigar_t *c) {
    if (!c) return;
    free(c->contig1); free(c->contig2); free(c->op_type); free(c->op_len);
    free(c);
}

int sa_cigar_load(const char *path, sa_cigar_t **out) {
    FILE *f = fopen(path, "r");
    if (!f) return SA_EIO;
    char *line = sa_read_line(f);
    fclose(f);
    if (!line) return SA_EIO;
    char **t;
    int64_t n = sa_split_ws(line, &t);
    int rc = SA_EIO;
    sa_cigar_t *c = calloc(1, sizeof(*c));
    /* cigar: query qstart qend qstrand target tstart tend tstrand score (op len)* */
    if (n >= 10 && strcmp(t[0], "cigar:") == 0 && (n - 10) % 2 == 0) {
        c->contig2 = strdup(t[1]);
        c->start2 = strtoll(t[2], NULL, 10);
        c->end2 = strtoll(t[3], NULL, 10);
        c->strand2 = t[4][0] == '+';
        c->contig1 = strdup(t[5]);
        c->start1 = strtoll(t[6], NULL, 10);
        c->end1 = strtoll(t[7], NULL, 10);
        c->strand1 = t[8][0] == '+';
        c->score = strtod(t[9], NULL);
        c->n_ops = (n - 10) / 2;
        c->op_type = malloc(sizeof(int32_t) * (size_t) (c->n_ops > 0 ? c->n_ops : 1));
        c->op_len = malloc(sizeof(int64_t) * (size_t) (c->n_ops > 0 ? c->n_ops : 1));
        rc = SA_OK;
        for (int64_t i = 0; i < c->n_ops; i++) {
            char op = t[10 + 2 * i][0];
            c->op_type[i] = op == 'M' ? 0 : (op == 'D' ? 1 : (op == 'I' ? 2 : -1));
            c->op_len[i] = strtoll(t[11 + 2 * i], NULL, 10);
            if (c->op_type[i] < 0 || c->op_len[i] < 0) rc = SA_EIO;
        }
    }
    free(t);
    free(line);
    if (rc) { sa_cigar_free(c); return rc; }
    *out = c;
    return SA_OK;
}

/* ---- FASTA through its .fai index ---------------------------------------------------------------- */
/* fastaHandler_getSubSequence (impl/fasta_handler.c:15-44): [start, end) of the named record on the forward strand, the
 * htslib interval [end, start - 1] (what the reference asks for) otherwise */
int sa_fasta_subsequence(const char *fasta_path, const char *name, int64_t start, int64_t end, int strand, char **out) {
    if (!fasta_path || !name || !out) return SA_EINVAL;
    int err = 0;
    *out = strand ? sa_fasta_fetch(fasta_path, name, start, end - 1, &err) : sa_fasta_fetch(fasta_path, name, end, start - 1, &err);
    if (*out == NULL) return err == -2 ? SA_EINVAL : SA_EIO;
    return SA_OK;
}

/* A batch front door fetches one window per read from the same FASTA: the record's index line (from <path>.fai, or from a scan
 * of the file) is remembered per (path, record) -- validated against the file's size and modification time -- and the bases are
 * read with one pread instead of a character at a time (100 000 short reads: 39.5 thread-seconds of 93 in the host stage were
 * this function re-opening and re-parsing the index). */
#include <fcntl.h>
#include <pthread.h>
#include <sys/stat.h>
#include <unistd.h>
typedef struct { char *key; long long len, off, bases, width; long long size; long long mtime_ns; long long fai_sig; } fai_memo_t;
static fai_memo_t g_fai_memo[16];
static int g_fai_next = 0;
static pthread_mutex_t g_fai_mu = PTHREAD_MUTEX_INITIALIZER;
static int fai_memo_get(const char *key, const struct stat *st, long long fai_sig, long long *len, long long *off, long long *bases, long long *width) {
    int hit = 0;
    pthread_mutex_lock(&g_fai_mu);
    for (int i = 0; i < 16; i++)
        if (g_fai_memo[i].key && strcmp(g_fai_memo[i].key, key) == 0 && g_fai_memo[i].size == (long long) st->st_size &&
            g_fai_memo[i].mtime_ns == (long long) st->st_mtim.tv_sec * 1000000000ll + st->st_mtim.tv_nsec &&
            g_fai_memo[i].fai_sig == fai_sig) {
            *len = g_fai_memo[i].len; *off = g_fai_memo[i].off; *bases = g_fai_memo[i].bases; *width = g_fai_memo[i].width;
            hit = 1;
            break;
        }
    pthread_mutex_unlock(&g_fai_mu);
    return hit;
}
static void fai_memo_put(const char *key, const struct stat *st, long long fai_sig, long long len, long long off, long long bases, long long width) {
    pthread_mutex_lock(&g_fai_mu);
    fai_memo_t *e = &g_fai_memo[g_fai_next];
    g_fai_next = (g_fai_next + 1) % 16;
    free(e->key);
    e->key = strdup(key);
    e->len = len; e->off = off; e->bases = bases; e->width = width;
    e->size = (long long) st->st_size;
    e->mtime_ns = (long long) st->st_mtim.tv_sec * 1000000000ll + st->st_mtim.tv_nsec;
    e->fai_sig = fai_sig;
    pthread_mutex_unlock(&g_fai_mu);
}
static char *fasta_window(const char *fasta_path, long long len, long long off, long long bases, long long width, int64_t start,
                          int64_t end_incl, int *err) {
    if (start < 0) start = 0;
    if (end_incl >= len) end_incl = len - 1;
    const int64_t n = end_incl >= start ? end_incl - start + 1 : 0;
    char *seq = malloc((size_t) n + 1);
    if (!seq) { if (err) *err = -1; return NULL; }
    int64_t got = 0;
    if (n > 0 && bases > 0) {
        const int fd = open(fasta_path, O_RDONLY);
        if (fd < 0) { free(seq); if (err) *err = -1; return NULL; }
        const long long pos = off + (start / bases) * width + (start % bases);
        const long long extra = width > bases ? width - bases : 1;
        const size_t want = (size_t) (n + (n / bases + 2) * extra + 2);
        char *raw = malloc(want);
        if (!raw) { close(fd); free(seq); if (err) *err = -1; return NULL; }
        /* (pread may return less than asked for -- a signal, a network file system: until `want` bytes or the end of the file) */
        size_t have = 0;
        int failed = 0;
        while (have < want) {
            const ssize_t r = pread(fd, raw + have, want - have, (off_t) (pos + (long long) have));
            if (r < 0) { failed = 1; break; }
            if (r == 0) break;
            have += (size_t) r;
        }
        close(fd);
        for (size_t i = 0; i < have && got < n; i++)
            if (raw[i] != '\n' && raw[i] != '\r') seq[got++] = raw[i];
        free(raw);
        /* fewer bases than the index promises (a read error, a FASTA shorter than its .fai says): an error, not a shorter window */
        if (failed || got < n) { free(seq); if (err) *err = -1; return NULL; }
    }
    seq[got] = 0;
    return seq;
}

char *sa_fasta_fetch(const char *fasta_path, const char *name, int64_t start, int64_t end_incl, int *err) {
    if (err) *err = 0;
    struct stat st_fa;
    char *memo_key = NULL;
    /* the index's own size and modification time are part of what a remembered line is valid for (a regenerated .fai -- another
     * line width -- must not be answered from memory); 0: no index next to the FASTA */
    long long fai_sig = 0;
    {
        const size_t pl0 = strlen(fasta_path);
        char *fai0 = malloc(pl0 + 5);
        struct stat st_fi;
        if (fai0) {
            memcpy(fai0, fasta_path, pl0);
            memcpy(fai0 + pl0, ".fai", 5);
            if (stat(fai0, &st_fi) == 0)
                fai_sig = ((long long) st_fi.st_mtim.tv_sec * 1000000000ll + st_fi.st_mtim.tv_nsec) ^ ((long long) st_fi.st_size << 20) ^ 1;
            free(fai0);
        }
    }
    if (stat(fasta_path, &st_fa) == 0) {
        memo_key = malloc(strlen(fasta_path) + strlen(name) + 2);
        if (memo_key) {
            sprintf(memo_key, "%s\n%s", fasta_path, name);
            long long l_, o_, b_, w_;
            if (fai_memo_get(memo_key, &st_fa, fai_sig, &l_, &o_, &b_, &w_)) {
                free(memo_key);
                return fasta_window(fasta_path, l_, o_, b_, w_, start, end_incl, err);
            }
        }
    }
    size_t pl = strlen(fasta_path);
    char *fai = malloc(pl + 5);
    memcpy(fai, fasta_path, pl);
    memcpy(fai + pl, ".fai", 5);
    FILE *fi = fopen(fai, "r");
    free(fai);
    long long len = -1, off = 0, bases = 0, width = 0;
    if (!fi) {
        /* no index next to the FASTA: htslib's fai_load (impl/fasta_handler.c:19) would build one; here the record is
         * located by scanning, nothing is written */
        FILE *fs = fopen(fasta_path, "r");
        if (!fs) { free(memo_key); if (err) *err = -1; return NULL; }
        const size_t name_len = strlen(name);
        long long pos = 0, rec_len = 0, rec_off = 0, rec_bases = 0, rec_width = 0, line_start = 0, line_bases = 0;
        int in_header = 0, match = 0, hdr_i = 0, hdr_ok = 1, first_line = 0, ch;
        char hdr[256];
        while (1) {
            ch = fgetc(fs);
            if (ch == EOF || (ch == '>' && pos == line_start)) {
                if (match) { len = rec_len; off = rec_off; bases = rec_bases; width = rec_width; break; }
                if (ch == EOF) break;
                in_header = 1; hdr_i = 0; hdr_ok = 1;
                pos++;
                continue;
            }
            pos++;
            if (in_header) {
                if (ch == '\n') {
                    hdr[hdr_i < 255 ? hdr_i : 255] = 0;
                    match = hdr_ok && strlen(hdr) == name_len && strcmp(hdr, name) == 0;
                    in_header = 0;
                    rec_len = 0; rec_off = pos; rec_bases = 0; rec_width = 0; first_line = 1;
                    line_start = pos; line_bases = 0;
                } else if (ch == ' ' || ch == '\t' || ch == '\r') {
                    hdr_ok = hdr_ok && 1;
                    if (hdr_i < 255) hdr[hdr_i] = 0;
                    hdr_i = 256; /* the name ends at the first white space */
                } else if (hdr_i < 255) {
                    hdr[hdr_i++] = (char) ch;
                    hdr[hdr_i] = 0;
                }
                continue;
            }
            if (ch == '\n') {
                if (first_line && line_bases > 0) { rec_bases = line_bases; rec_width = pos - line_start; first_line = 0; }
                line_start = pos; line_bases = 0;
            } else if (ch != '\r') {
                line_bases++;
                rec_len++;
            }
        }
        if (match && rec_bases == 0) { len = rec_len; off = rec_off; bases = rec_len > 0 ? rec_len : 1; width = bases + 1; }
        fclose(fs);
    } else {
    char *line;
    while ((line = sa_read_line(fi)) != NULL) {
        char **t;
        int64_t n = sa_split_ws(line, &t);
        if (n >= 5 && strcmp(t[0], name) == 0) {
            len = strtoll(t[1], NULL, 10); off = strtoll(t[2], NULL, 10);
            bases = strtoll(t[3], NULL, 10); width = strtoll(t[4], NULL, 10);
            free(t); free(line);
            break;
        }
        free(t);
        free(line);
    }
    fclose(fi);
    }
    if (len < 0) { free(memo_key); if (err) *err = -2; return NULL; }
    if (memo_key) { fai_memo_put(memo_key, &st_fa, fai_sig, len, off, bases, width); free(memo_key); }
    return fasta_window(fasta_path, len, off, bases, width, start, end_incl, err);
}

static char comp_char(char c) {
    switch (c) {
        case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
        case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
        default: return c;
    }
}
char *sa_complement(const char *s) {
    size_t n = strlen(s);
    char *o = malloc(n + 1);
    for (size_t i = 0; i < n; i++) o[i] = comp_char(s[i]);
    o[n] = 0;
    return o;
}
void sa_reverse_in_place(char *s) {
    size_t n = strlen(s);
    for (size_t i = 0; i + 1 < n - i; i++) { char t = s[i]; s[i] = s[n - 1 - i]; s[n - 1 - i] = t; }
}
char *sa_reverse_complement(const char *s) {
    char *o = sa_complement(s);
    sa_reverse_in_place(o);
    return o;
}
