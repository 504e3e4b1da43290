/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's maximum-expected-accuracy path step
 * (src/signalalign/mea_algorithm.py), the step immediately downstream of the pair-HMM posteriors.
 * Nothing in the product may call this; tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it as
 * the checker.
 *
 * Pinned by the reference's own known-answer test (src/signalalign/tests/test_mea_algorithm.py:22-69: a 5x5 posterior
 * matrix, the three / four surviving forward edges and their sums) and, as that file does (:87-123), by agreement of
 * the best sum with an independent exhaustive formulation (mea_slow, restated in oracle/sa_oracle_py.py) on random
 * matrices; see tests/test_oracle_mea.py.
 *
 * The reference keeps Python lists of edges [ref, event, posterior, sum, previous edge]; here edges live in an arena
 * and the two lists ("forward edges" of the previous events, "new edges" of the current one) hold arena indices.
 * Branch order, strictness of every comparison and the order of the floating-point additions follow the reference
 * line by line, including the behaviours that look accidental (see the comments), because the GPU path is checked
 * against this bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "sa_oracle.h"

typedef struct {
    int32_t ref, event;
    double p, sum;
    int32_t prev; /* arena index or -1 (None) */
} mea_edge_t;

typedef struct {
    mea_edge_t *a;
    int64_t n, cap;
} arena_t;
typedef struct {
    int32_t *v;
    int64_t n, cap;
} list_t;

static int32_t arena_push(arena_t *A, int32_t ref, int32_t event, double p, double sum, int32_t prev) {
    if (A->n == A->cap) {
        A->cap = A->cap ? 2 * A->cap : 1024;
        A->a = (mea_edge_t *) realloc(A->a, sizeof(mea_edge_t) * (size_t) A->cap);
    }
    mea_edge_t *e = &A->a[A->n];
    e->ref = ref; e->event = event; e->p = p; e->sum = sum; e->prev = prev;
    return (int32_t) A->n++;
}
static void list_push(list_t *L, int32_t id) {
    if (L->n == L->cap) {
        L->cap = L->cap ? 2 * L->cap : 256;
        L->v = (int32_t *) realloc(L->v, sizeof(int32_t) * (size_t) L->cap);
    }
    L->v[L->n++] = id;
}

/* maximum_expected_accuracy_alignment (mea_algorithm.py:25-197) on a COO matrix (row = event, col = reference position,
 * in the order given: the reference takes scipy's row-major order for granted) and shortest_ref_per_event (INT32_MAX =
 * the reference's inf).  Returns the status (SAO_MEA_*); on SAO_MEA_OK path_ref and path_event (malloc, freed with
 * sao_free) hold get_indexes_from_best_path (:248-264) of the best edge, *best_sum its sum; edge_sums/n_edges (may be
 * NULL) receive the sums of ALL final forward edges (return_all=True), which is what the reference's test asserts on. */
int sao_mea(const int32_t *rows, const int32_t *cols, const double *data, int64_t n, const int32_t *shortest,
            int64_t n_shortest, int32_t **path_ref, int32_t **path_event, int64_t *n_path, double *best_sum,
            double **edge_sums, int64_t *n_edges) {
    if (path_ref) *path_ref = NULL;
    if (path_event) *path_event = NULL;
    if (n_path) *n_path = 0;
    if (edge_sums) *edge_sums = NULL;
    if (n_edges) *n_edges = 0;
    if (n <= 0) return SAO_MEA_EMPTY; /* min() of an empty sequence raises */
    arena_t A = {0};
    list_t F = {0}, N = {0};
    int status = SAO_MEA_OK;
    /* :41-58 first event: entries up to (and including) its largest posterior, kept while non-decreasing */
    int32_t smallest = rows[0];
    for (int64_t j = 1; j < n; j++)
        if (rows[j] < smallest) smallest = rows[j];
    int64_t num_first = 0, arg = -1, seen = 0;
    double best = 0.0;
    for (int64_t j = 0; j < n; j++)
        if (rows[j] == smallest) {
            if (arg < 0 || data[j] > best) { best = data[j]; arg = seen; } /* np.argmax: first maximum of the masked data */
            seen++;
            num_first++;
        }
    double max_prob = 0;
    for (int64_t x = 0; x <= arg; x++) /* x indexes the unmasked arrays, as the reference does */
        if (data[x] >= max_prob) {
            list_push(&F, arena_push(&A, cols[x], rows[x], data[x], data[x], -1));
            max_prob = data[x];
        }
    if (num_first >= n) { status = SAO_MEA_SINGLE_EVENT; goto out; } /* row[num_first_event] raises IndexError (:61) */
    {
        int32_t prev_event = rows[num_first];
        int first_pass = 1;
        int64_t i = 0, max_i = -1;
        max_prob = 0;
        for (int64_t j = num_first; j < n; j++) {
            const int32_t e = rows[j], r = cols[j];
            const double p = data[j];
            if (prev_event != e) { /* :76-93 a new event: what is left of the old front survives if it raises the maximum */
                prev_event = e;
                for (; i < F.n; i++)
                    if (A.a[F.v[i]].sum > max_prob) {
                        list_push(&N, F.v[i]);
                        max_prob = A.a[F.v[i]].sum;
                    }
                first_pass = 1;
                list_t t = F; F = N; N = t; /* forward_edges = new_edges (N is re-created below) */
            }
            if (first_pass) { /* :95-118 */
                first_pass = 0;
                max_i = -1;
                N.n = 0;
                i = 0;
                max_prob = 0;
                int found = 0;
                if (F.n == 0) { status = SAO_MEA_NO_FRONT; goto out; } /* forward_edges[0] raises IndexError */
                if (e < 0 || e >= n_shortest) { status = SAO_MEA_BAD_EVENT; goto out; }
                while (A.a[F.v[i]].ref < shortest[e]) {
                    i++;
                    found = 1;
                    if (i == F.n) break;
                }
                if (found) { /* the last edge below every future reference position stays reachable */
                    list_push(&N, F.v[i - 1]);
                    max_prob = A.a[F.v[i - 1]].sum;
                }
                i = 0;
            }
            for (;;) { /* :120-171 */
                if (i < F.n) {
                    const mea_edge_t *fi = &A.a[F.v[i]];
                    if (fi->ref < r) {
                        if (i > max_i && max_prob < fi->sum) {
                            list_push(&N, F.v[i]);
                            max_prob = fi->sum;
                            max_i = i;
                        }
                        i++;
                    } else if (fi->ref == r) {
                        if (i == 0) {
                            if (fi->sum > max_prob) { /* stay: the sum does not grow */
                                list_push(&N, arena_push(&A, r, e, p, fi->sum, F.v[i]));
                                max_prob = A.a[F.v[i]].sum;
                            }
                        } else {
                            const double via = A.a[F.v[i - 1]].sum + p;
                            const double stay = A.a[F.v[i]].sum;
                            if (stay > via) {
                                if (stay > max_prob) {
                                    list_push(&N, arena_push(&A, r, e, p, stay, F.v[i]));
                                    max_prob = stay;
                                }
                            } else if (via > max_prob) {
                                list_push(&N, arena_push(&A, r, e, p, via, F.v[i - 1]));
                                max_prob = via;
                            }
                        }
                        max_i = i;
                        break;
                    } else {
                        if (i == 0) {
                            if (p > max_prob) {
                                list_push(&N, arena_push(&A, r, e, p, p, -1));
                                max_prob = p;
                            }
                        } else {
                            const double via = A.a[F.v[i - 1]].sum + p;
                            if (via > max_prob) {
                                list_push(&N, arena_push(&A, r, e, p, via, F.v[i - 1]));
                                max_prob = via;
                            }
                        }
                        break;
                    }
                } else { /* the reference position lies past every edge */
                    const double via = A.a[F.v[i - 1]].sum + p;
                    if (via > max_prob) {
                        list_push(&N, arena_push(&A, r, e, p, via, F.v[i - 1]));
                        max_prob = via;
                    }
                    break;
                }
            }
        }
        /* :174-180 trailing edges; max_prob is NOT raised here */
        for (; i < F.n; i++)
            if (A.a[F.v[i]].sum > max_prob) list_push(&N, F.v[i]);
        list_t t = F; F = N; N = t;
    }
    if (edge_sums && n_edges) {
        *edge_sums = (double *) malloc(sizeof(double) * (size_t) (F.n ? F.n : 1));
        for (int64_t q = 0; q < F.n; q++) (*edge_sums)[q] = A.a[F.v[q]].sum;
        *n_edges = F.n;
    }
    {
        /* :186-196 the first edge with the strictly highest sum above 0 */
        double highest = 0;
        int32_t best_id = -1;
        for (int64_t q = 0; q < F.n; q++)
            if (A.a[F.v[q]].sum > highest) { highest = A.a[F.v[q]].sum; best_id = F.v[q]; }
        if (best_id < 0) { status = SAO_MEA_NO_PATH; goto out; } /* best_forward_edge stays the int 0 */
        if (best_sum) *best_sum = highest;
        int64_t len = 0;
        for (int32_t q = best_id; q >= 0; q = A.a[q].prev) len++;
        if (path_ref && path_event && n_path) {
            *path_ref = (int32_t *) malloc(sizeof(int32_t) * (size_t) len);
            *path_event = (int32_t *) malloc(sizeof(int32_t) * (size_t) len);
            int64_t w = len;
            for (int32_t q = best_id; q >= 0; q = A.a[q].prev) {
                w--;
                (*path_ref)[w] = A.a[q].ref;
                (*path_event)[w] = A.a[q].event;
            }
            *n_path = len;
        }
    }
out:
    free(A.a);
    free(F.v);
    free(N.v);
    return status;
}

/* get_mea_params_from_events (mea_algorithm.py:267-320) without the dense matrices: from the (reference_index,
 * event_index, posterior_probability) columns of a signalAlign event table, in table order, to the COO matrix
 * scipy.sparse.coo_matrix(posterior_matrix) would hold (row-major, explicit zeros dropped) and shortest_ref_per_event.
 * rows_out/cols_out/data_out need room for n entries, shortest_out for (max event - min event + 1) (INT32_MAX = inf);
 * returns the number of COO entries, *n_events_out the length of shortest_out, or a negative SAO_MEA_* on bad input. */
typedef struct {
    int32_t ev, ref;
    double p;
    int64_t order;
} mea_row_t;
/* np.sort(events, order=['event_index']) on a structured array breaks ties with the remaining fields in dtype order
 * (contig, reference_index, ..., posterior_probability, ...): within an event rows end up by reference index, and rows
 * of one cell that differ only in their posterior by ascending posterior. */
static int cmp_cell(const void *a, const void *b) {
    const mea_row_t *x = (const mea_row_t *) a, *y = (const mea_row_t *) b;
    if (x->ev != y->ev) return x->ev < y->ev ? -1 : 1;
    if (x->ref != y->ref) return x->ref < y->ref ? -1 : 1;
    if (x->p != y->p) return x->p < y->p ? -1 : 1;
    return x->order < y->order ? -1 : (x->order > y->order ? 1 : 0);
}
int64_t sao_mea_params(const int64_t *reference_index, const int64_t *event_index, const double *posterior, int64_t n,
                       int32_t *rows_out, int32_t *cols_out, double *data_out, int32_t *shortest_out,
                       int64_t *n_events_out) {
    if (n <= 0) return -SAO_MEA_EMPTY;
    int64_t ref_start = reference_index[0], ref_end = reference_index[0];
    for (int64_t i = 1; i < n; i++) {
        if (reference_index[i] < ref_start) ref_start = reference_index[i];
        if (reference_index[i] > ref_end) ref_end = reference_index[i];
    }
    mea_row_t *t = (mea_row_t *) malloc(sizeof(mea_row_t) * (size_t) n);
    int64_t ev_start = event_index[0], ev_end = event_index[0];
    for (int64_t i = 0; i < n; i++) {
        if (event_index[i] < ev_start) ev_start = event_index[i];
        if (event_index[i] > ev_end) ev_end = event_index[i];
    }
    /* :288-292 minus strand when the first sorted row (smallest event, its smallest reference index) lies above the
     * last one (largest event, its largest reference index) */
    int64_t first = -1, last = -1;
    for (int64_t i = 0; i < n; i++) {
        if (event_index[i] == ev_start && (first < 0 || reference_index[i] < reference_index[first])) first = i;
        if (event_index[i] == ev_end && (last < 0 || reference_index[i] > reference_index[last])) last = i;
    }
    const int minus = reference_index[first] > reference_index[last];
    const int64_t n_events = ev_end - ev_start + 1;
    for (int64_t i = 0; i < n; i++) {
        t[i].ev = (int32_t) (event_index[i] - ev_start);
        t[i].ref = (int32_t) (minus ? ref_end - reference_index[i] : reference_index[i] - ref_start);
        t[i].p = posterior[i];
        t[i].order = i;
    }
    qsort(t, (size_t) n, sizeof(mea_row_t), cmp_cell);
    for (int64_t e = 0; e < n_events; e++) shortest_out[e] = INT32_MAX;
    /* :305-318 backwards through the sorted table: a cell is written last by its FIRST row, and shortest_ref_per_event
     * takes the running minimum whenever a row lowers its own event's entry (zero posteriors count here) */
    int64_t min_shortest = INT64_MAX;
    for (int64_t i = n - 1; i >= 0; i--) {
        const int32_t ev = t[i].ev, ref = t[i].ref;
        if ((int64_t) shortest_out[ev] > ref) {
            if (min_shortest > ref) min_shortest = ref;
            shortest_out[ev] = (int32_t) min_shortest;
        }
    }
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) {
        if (i > 0 && t[i].ev == t[i - 1].ev && t[i].ref == t[i - 1].ref) continue; /* later duplicates are overwritten */
        if (t[i].p == 0.0) continue;                                                /* coo_matrix keeps non-zeros only  */
        rows_out[m] = t[i].ev;
        cols_out[m] = t[i].ref;
        data_out[m] = t[i].p;
        m++;
    }
    free(t);
    if (n_events_out) *n_events_out = n_events;
    return m;
}
