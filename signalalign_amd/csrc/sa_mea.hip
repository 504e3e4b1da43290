// Maximum-expected-accuracy path over a read's posteriors on gfx950 -- SURVEY.md §8(f) row 3, the step immediately
// downstream of the pair-HMM: maximum_expected_accuracy_alignment + get_indexes_from_best_path
// (src/signalalign/mea_algorithm.py:25-197, :248-264).
//
// The reference walks the sparse posterior matrix (COO, row = event, column = reference position) in row-major order
// and keeps a short front of "forward edges" (reference position, best sum so far, back pointer) whose sums rise with
// the reference position; every entry extends the best edge to its left ("move", the sum grows by the posterior) or
// the edge in its own column ("stay", the sum is kept).  The recurrence is a chain: entry j needs the front entry j-1
// left behind, and the result must equal the reference's choice among ties, so the work inside one read is serial.
// The parallelism is across reads: ONE WAVE PER READ, run wave-uniformly -- 64 entries are fetched with one coalesced
// load and handed to the serial loop by v_readlane, the two fronts live in LDS (ping-pong), the edge arena (reference
// position, event, back pointer) streams out to HBM and is read back in 64-record blocks for the traceback, so the
// chain of dependent HBM loads a pointer walk would be becomes one load per 64 arena records.  A read whose front
// outgrows its LDS share is re-run by the same kernel body with its fronts in global memory (k_mea<true>).
//
// Branch order, comparison strictness and the order of the double additions follow the reference line by line
// (there are no multiplications to contract): paths and sums are bit-identical to oracle/sa_mea_oracle.c.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <vector>

#include "sa_internal.h"
#include "sa_scratch.h"

#define MEA_FRONT_CAP 256     // entries per LDS front (2 fronts x 16 bytes x 256 = 8 KB per wave)
#define MEA_ST_OVERFLOW 100   // internal: front outgrew LDS, the read is re-run with global fronts

struct MeaJob {
    long long off;      // first COO entry (and first arena record) of the read
    long long sh_off;   // first shortest_ref_per_event entry
    long long out_off;  // first output pair
    long long gf_off;   // first global-front entry (second pass only)
    int n, n_sh, out_cap, pad;
};

struct MeaPlan {
    const MeaJob *jobs;
    const int *rows, *cols;
    const double *data;
    const int *shortest;
    int *a_ref, *a_ev, *a_prev;  // edge arena, one record per COO entry at most
    int2 *out;                   // path, written backwards from out_off + out_cap
    int *n_out, *n_edges, *status;
    double *sum;
    struct MeaEdge *gf;          // global fronts (k_mea<true>)
    const int *n_dev;            // entries per read when the matrices were built on the device (sa_batch_mea), else NULL
    const int *mins;             // sa_batch_mea: (x, y) origin of each read's matrix, added back to the path; else NULL
};
struct __attribute__((aligned(16))) MeaEdge {   // one front entry, read and written as a single 16-byte word
    double sum;
    int ref, id;
};

__device__ __forceinline__ int rl(int v, int k) { return __builtin_amdgcn_readlane(v, k); }
__device__ __forceinline__ double rld(double v, int k) {
    return __hiloint2double(rl(__double2hiint(v), k), rl(__double2loint(v), k));
}
// Wave reductions.  The result is the same in every lane, but the compiler cannot know that of a cross-lane shuffle:
// v_readfirstlane tells it, and everything computed from these values (loop bounds, front lengths) stays in SGPRs with
// scalar branches instead of exec-masked vector code.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(uni(__double2hiint(v)), uni(__double2loint(v)));
}
__device__ __forceinline__ int wave_min(int v) {
    for (int o = 32; o; o >>= 1) v = min(v, __shfl_xor(v, o));
    return uni(v);
}
__device__ __forceinline__ int wave_sum(int v) {
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return uni(v);
}
__device__ __forceinline__ double wave_max(double v) {
    for (int o = 32; o; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return uni(v);
}

// :41-58 which entries belong to the first event: their count and the (masked) index of their first largest posterior
__device__ __forceinline__ void mea_first_event(const int *__restrict__ rows, const double *__restrict__ data, int n, int lane,
                                                int &num_first, int &arg) {
    int smallest = 0x7fffffff;
    for (int j = lane; j < n; j += 64) smallest = min(smallest, rows[j]);
    smallest = wave_min(smallest);
    double mx = -INFINITY;
    for (int j = lane; j < n; j += 64)
        if (rows[j] == smallest) mx = fmax(mx, data[j]);
    mx = wave_max(mx);
    int jstar = 0x7fffffff;   // np.argmax: the first maximum of the masked data
    for (int j = lane; j < n; j += 64)
        if (rows[j] == smallest && data[j] == mx) jstar = min(jstar, j);
    jstar = wave_min(jstar);
    int nf = 0, a = 0;
    for (int j = lane; j < n; j += 64)
        if (rows[j] == smallest) { nf++; a += j < jstar; }
    num_first = wave_sum(nf);
    arg = min(wave_sum(a), n - 1);
}

// :248-264 traceback: the arena comes back in blocks of 64 consecutive records (one coalesced load each; back pointers
// are short, so a block serves many steps), the walk inside a block is v_readlane.  Returns the path length, or -1 if
// the arena is inconsistent (cannot happen: back pointers fall strictly, a path holds one pair per event at most).
__device__ __forceinline__ int mea_traceback(const int *a_ref, const int *a_ev, const int *a_prev, int best_id, int2 *out,
                                             int out_cap, int lane, int x0, int y0) {
    __threadfence();
    int q = best_id, w = out_cap, n_path = 0;
    while (q >= 0) {
        const int lo = max(0, q - 63);
        const int idx = lo + lane;
        int rr = 0, ee = 0, pp = -1;
        if (idx <= q) { rr = a_ref[idx]; ee = a_ev[idx]; pp = a_prev[idx]; }
        while (q >= lo) {
            const int l = q - lo;
            const int r_ = rl(rr, l), e_ = rl(ee, l);
            const int nq = rl(pp, l);
            if (w <= 0 || nq >= q) return -1;
            q = nq;
            w--;
            if (lane == 0) out[w] = make_int2(r_ + x0, e_ + y0);
            n_path++;
        }
    }
    return n_path;
}

// ---- fronts in registers: lane l holds forward edge l -------------------------------------------------------------------
// The serial kernel below walks the front entry by entry; a lone wave issues about one instruction per 2.3 ns, and with a
// front of 5-10 edges every matrix entry costs 1.4 us.  Here both fronts live in VGPRs, one edge per lane (a front is
// rarely longer than a handful of edges; 64 is the limit of this kernel), and each of the reference's loops over the
// front becomes a few wave operations:
//   * "while forward_edges[i][0] < x: i += 1"  ->  count-trailing-zeros of a ballot (first lane whose edge fails);
//   * "append every edge that raises max_prob" ->  a ballot, because sums never fall along a front (every edge of a
//     front was appended because its sum exceeded the running maximum; the first event's edges are kept while they do
//     not fall): an edge raises the running maximum iff it exceeds the value the loop started with and the edge to its
//     left, which is one DPP wave rotate away;
//   * appending = writing the lane whose index equals the length (v_cndmask), copying an old edge = v_readlane.
// Order of appends, comparisons and additions are the reference's; a front that outgrows 64 edges sends the read to the
// serial kernel.
__device__ __forceinline__ unsigned long long lanes_below(int k) { return k >= 64 ? ~0ull : ((1ull << k) - 1ull); }
__device__ __forceinline__ double ror1(double v) {   // lane l receives lane l-1 (lane 0: lane 63)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x13C, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x13C, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void k_mea_wave(MeaPlan P, int n_jobs) {
    const int jb = blockIdx.x;
    const int lane = threadIdx.x;
    const MeaJob J = P.jobs[jb];
    const int *__restrict__ rows = P.rows + J.off;
    const int *__restrict__ cols = P.cols + J.off;
    const double *__restrict__ data = P.data + J.off;
    const int *__restrict__ shortest = P.shortest + J.sh_off;
    int *a_ref = P.a_ref + J.off, *a_ev = P.a_ev + J.off, *a_prev = P.a_prev + J.off;
    const int n = P.n_dev ? P.n_dev[jb] : J.n;
    int fref = 0, fid = 0, nref = 0, nid = 0;   // F[lane] (lane < nF), N[lane] (lane < nN)
    double fsum = 0, nsum = 0;
    int nF = 0, nN = 0, na = 0, status = SA_MEA_OK, n_edges = 0, n_path = 0;
    double best_sum = 0.0;

#define WN_SET(ref_, id_, sum_)                                         \
    do {                                                                \
        if (nN >= 64) { status = MEA_ST_OVERFLOW; goto done; }          \
        if (lane == nN) { nref = (ref_); nid = (id_); nsum = (sum_); }  \
        nN++;                                                           \
    } while (0)
#define WN_COPY_SET(mask_)                                              \
    do {                                                                \
        unsigned long long m_ = (mask_);                                \
        while (m_) {                                                    \
            const int l_ = __builtin_ctzll(m_);                         \
            m_ &= m_ - 1;                                               \
            WN_SET(rl(fref, l_), rl(fid, l_), rld(fsum, l_));           \
        }                                                               \
    } while (0)
#define WARENA_PUSH(ref_, ev_, prev_)                                                  \
    do {                                                                               \
        if (lane == 0) { a_ref[na] = (ref_); a_ev[na] = (ev_); a_prev[na] = (prev_); } \
        na++;                                                                          \
    } while (0)
#define WSWAP()                                      \
    do {                                             \
        fref = nref; fid = nid; fsum = nsum; nF = nN; \
    } while (0)

    if (n <= 0) { status = SA_MEA_EMPTY; goto done; }
    {
        int num_first, arg;
        mea_first_event(rows, data, n, lane, num_first, arg);
        double max_prob = 0;
        for (int x = 0; x <= arg; x++) {   // x indexes the unmasked arrays, as the reference does
            const double d = data[x];
            if (d >= max_prob) {
                WN_SET(cols[x], na, d);
                WARENA_PUSH(cols[x], rows[x], -1);
                max_prob = d;
            }
        }
        WSWAP();
        nN = 0;
        if (num_first >= n) { status = SA_MEA_SINGLE_EVENT; goto done; }

        int prev_event = rows[num_first];
        bool first_pass = true;
        int i = 0, max_i = -1;
        max_prob = 0;
        for (int base = num_first; base < n; base += 64) {
            const int cnt = min(64, n - base);
            int rv = 0, cv = 0, sv = 0;
            double dv = 0;
            if (lane < cnt) {
                rv = rows[base + lane]; cv = cols[base + lane]; dv = data[base + lane];
                if (rv >= 0 && rv < J.n_sh) sv = shortest[rv];
            }
            for (int k = 0; k < cnt; k++) {
                const int e = rl(rv, k), r = rl(cv, k);
                const double p = rld(dv, k);
                if (prev_event != e) {   // :76-93 what is left of the old front survives where it raises the maximum
                    prev_event = e;
                    if (i < nF) {
                        const double left = ror1(fsum);
                        const unsigned long long cand = lanes_below(nF) & ~lanes_below(i);
                        const unsigned long long up =
                            __builtin_amdgcn_ballot_w64(fsum > max_prob && (lane == i || fsum > left)) & cand;
                        if (up) max_prob = rld(fsum, 63 - __builtin_clzll(up));
                        WN_COPY_SET(up);
                    }
                    first_pass = true;
                    WSWAP();
                }
                if (first_pass) {        // :95-118
                    first_pass = false;
                    max_i = -1;
                    nN = 0;
                    i = 0;
                    max_prob = 0;
                    if (nF == 0) { status = SA_MEA_NO_FRONT; goto done; }
                    if (e < 0 || e >= J.n_sh) { status = SA_MEA_BAD_EVENT; goto done; }
                    const int sh = rl(sv, k);
                    const unsigned long long stop = __builtin_amdgcn_ballot_w64(!(fref < sh)) | ~lanes_below(nF);
                    const int i_stop = stop ? __builtin_ctzll(stop) : 64;
                    if (i_stop > 0) {    // the last edge below every future reference position stays reachable
                        const double ks = rld(fsum, i_stop - 1);
                        WN_SET(rl(fref, i_stop - 1), rl(fid, i_stop - 1), ks);
                        max_prob = ks;
                    }
                }
                // :120-171  edges left of r: those that raise the maximum are carried over ...
                {
                    const unsigned long long stop =
                        (__builtin_amdgcn_ballot_w64(!(fref < r)) | ~lanes_below(nF)) & ~lanes_below(i);
                    const int i_end = stop ? __builtin_ctzll(stop) : 64;
                    const int c0 = max(i, max_i + 1);
                    if (c0 < i_end) {
                        const double left = ror1(fsum);
                        const unsigned long long cand = lanes_below(i_end) & ~lanes_below(c0);
                        const unsigned long long up =
                            __builtin_amdgcn_ballot_w64(fsum > max_prob && (lane == c0 || fsum > left)) & cand;
                        if (up) {
                            max_i = 63 - __builtin_clzll(up);
                            max_prob = rld(fsum, max_i);
                        }
                        WN_COPY_SET(up);
                    }
                    if (i_end > i) i = i_end;
                }
                // ... then the entry itself: stay in its column, or move in from the edge to its left
                if (i < nF) {
                    const int fr = rl(fref, i);
                    if (fr == r) {
                        const double stay = rld(fsum, i);
                        if (i == 0) {
                            if (stay > max_prob) {
                                WN_SET(r, na, stay);
                                WARENA_PUSH(r, e, rl(fid, i));
                                max_prob = stay;
                            }
                        } else {
                            const double via = rld(fsum, i - 1) + p;
                            if (stay > via) {
                                if (stay > max_prob) {
                                    WN_SET(r, na, stay);
                                    WARENA_PUSH(r, e, rl(fid, i));
                                    max_prob = stay;
                                }
                            } else if (via > max_prob) {
                                WN_SET(r, na, via);
                                WARENA_PUSH(r, e, rl(fid, i - 1));
                                max_prob = via;
                            }
                        }
                        max_i = i;
                    } else if (i == 0) {
                        if (p > max_prob) {
                            WN_SET(r, na, p);
                            WARENA_PUSH(r, e, -1);
                            max_prob = p;
                        }
                    } else {
                        const double via = rld(fsum, i - 1) + p;
                        if (via > max_prob) {
                            WN_SET(r, na, via);
                            WARENA_PUSH(r, e, rl(fid, i - 1));
                            max_prob = via;
                        }
                    }
                } else {                 // the reference position lies past every edge
                    const double via = rld(fsum, i - 1) + p;
                    if (via > max_prob) {
                        WN_SET(r, na, via);
                        WARENA_PUSH(r, e, rl(fid, i - 1));
                        max_prob = via;
                    }
                }
            }
        }
        // :174-180 trailing edges; max_prob is NOT raised here
        if (i < nF) {
            const unsigned long long up =
                __builtin_amdgcn_ballot_w64(fsum > max_prob) & lanes_below(nF) & ~lanes_below(i);
            WN_COPY_SET(up);
        }
        WSWAP();
        n_edges = nF;
        // :186-196 the first edge with the strictly highest sum above 0
        const double highest = wave_max(lane < nF ? fsum : -INFINITY);
        if (!(highest > 0)) { status = SA_MEA_NO_PATH; goto done; }
        const unsigned long long at = __builtin_amdgcn_ballot_w64(fsum == highest) & lanes_below(nF);
        const int best_id = rl(fid, __builtin_ctzll(at));
        best_sum = highest;
        n_path = mea_traceback(a_ref, a_ev, a_prev, best_id, P.out + J.out_off, J.out_cap, lane, P.mins ? P.mins[2 * jb] : 0,
                               P.mins ? P.mins[2 * jb + 1] : 0);
        if (n_path < 0) { n_path = 0; status = SA_MEA_NO_PATH; }
    }
done:
    if (lane == 0) {
        P.status[jb] = status;
        P.n_out[jb] = status == SA_MEA_OK ? n_path : 0;
        P.n_edges[jb] = n_edges;
        P.sum[jb] = best_sum;
    }
#undef WN_SET
#undef WN_COPY_SET
#undef WARENA_PUSH
#undef WSWAP
}

__device__ __forceinline__ MeaEdge uni_edge(const MeaEdge &e) {   // every lane read the same entry
    MeaEdge u;
    u.sum = uni(e.sum); u.ref = uni(e.ref); u.id = uni(e.id);
    return u;
}

template <bool GLOBAL_FRONT>
__global__ __launch_bounds__(64) void k_mea(MeaPlan P, const int *__restrict__ ids, int n_ids) {
    const int jb = ids ? ids[blockIdx.x] : (int) blockIdx.x;
    const int lane = threadIdx.x;
    const MeaJob J = P.jobs[jb];
    const int *__restrict__ rows = P.rows + J.off;
    const int *__restrict__ cols = P.cols + J.off;
    const double *__restrict__ data = P.data + J.off;
    const int *__restrict__ shortest = P.shortest + J.sh_off;
    int *a_ref = P.a_ref + J.off, *a_ev = P.a_ev + J.off, *a_prev = P.a_prev + J.off;
    const int n = P.n_dev ? P.n_dev[jb] : J.n;

    __shared__ MeaEdge s_front[2 * MEA_FRONT_CAP];
    const int cap = GLOBAL_FRONT ? n + 2 : MEA_FRONT_CAP;
    // F = forward edges of the previous events, N = new edges: two halves of one array, told apart by an offset (not
    // by swapping pointers: the LDS accesses must stay ds_read/ds_write -- a generic pointer turns them into flat
    // loads, which wait for the outstanding arena stores as well)
    MeaEdge *const gfront = GLOBAL_FRONT ? P.gf + J.gf_off : nullptr;
    int fo = 0, no = cap;
#define FRONT(idx_) (*(GLOBAL_FRONT ? &gfront[idx_] : &s_front[idx_]))
#define F_AT(i_) uni_edge(FRONT(fo + (i_)))
    int nF = 0, nN = 0, na = 0, status = SA_MEA_OK, n_edges = 0, n_path = 0;
    double best_sum = 0.0;

// every lane holds the same values and writes them: a lane reads back what it wrote itself (LDS or global)
#define N_PUSH(ref_, id_, sum_)                                   \
    do {                                                          \
        if (nN >= cap) { status = MEA_ST_OVERFLOW; goto done; }   \
        { MeaEdge t_; t_.sum = (sum_); t_.ref = (ref_); t_.id = (id_); FRONT(no + nN) = t_; } \
        nN++;                                                     \
    } while (0)
#define ARENA_PUSH(ref_, ev_, prev_)                                                 \
    do {                                                                             \
        if (lane == 0) { a_ref[na] = (ref_); a_ev[na] = (ev_); a_prev[na] = (prev_); } \
        na++;                                                                        \
    } while (0)
#define SWAP_FRONTS()                                                        \
    do {                                                                     \
        { const int t_ = fo; fo = no; no = t_; }                             \
        nF = nN;                                                             \
    } while (0)

    if (n <= 0) { status = SA_MEA_EMPTY; goto done; }
    {
        // :41-58 the first event: its entries up to the largest posterior, kept while they do not fall
        int num_first, arg;
        mea_first_event(rows, data, n, lane, num_first, arg);
        double max_prob = 0;
        for (int x = 0; x <= arg; x++) {   // x indexes the unmasked arrays, as the reference does
            const double d = data[x];
            if (d >= max_prob) {
                N_PUSH(cols[x], na, d);
                ARENA_PUSH(cols[x], rows[x], -1);
                max_prob = d;
            }
        }
        SWAP_FRONTS();
        nN = 0;
        if (num_first >= n) { status = SA_MEA_SINGLE_EVENT; goto done; }

        int prev_event = rows[num_first];
        bool first_pass = true;
        int i = 0, max_i = -1;
        max_prob = 0;
        for (int base = num_first; base < n; base += 64) {
            const int cnt = min(64, n - base);
            int rv = 0, cv = 0, sv = 0;
            double dv = 0;
            if (lane < cnt) {
                rv = rows[base + lane]; cv = cols[base + lane]; dv = data[base + lane];
                if (rv >= 0 && rv < J.n_sh) sv = shortest[rv];   // fetched with the entry: no load on the serial chain
            }
            for (int k = 0; k < cnt; k++) {
                const int e = rl(rv, k), r = rl(cv, k);
                const double p = rld(dv, k);
                if (prev_event != e) {   // :76-93 what is left of the old front survives where it raises the maximum
                    prev_event = e;
                    for (; i < nF; i++) {
                        const MeaEdge fe = F_AT(i);
                        if (fe.sum > max_prob) { N_PUSH(fe.ref, fe.id, fe.sum); max_prob = fe.sum; }
                    }
                    first_pass = true;
                    SWAP_FRONTS();
                }
                if (first_pass) {        // :95-118
                    first_pass = false;
                    max_i = -1;
                    nN = 0;
                    i = 0;
                    max_prob = 0;
                    bool found = false;
                    if (nF == 0) { status = SA_MEA_NO_FRONT; goto done; }
                    if (e < 0 || e >= J.n_sh) { status = SA_MEA_BAD_EVENT; goto done; }
                    const int sh = rl(sv, k);
                    MeaEdge last = F_AT(0), fe = last;
                    while (fe.ref < sh) {
                        last = fe;
                        i++;
                        found = true;
                        if (i == nF) break;
                        fe = F_AT(i);
                    }
                    if (found) {         // the last edge below every future reference position stays reachable
                        N_PUSH(last.ref, last.id, last.sum);
                        max_prob = last.sum;
                    }
                    i = 0;
                }
                for (;;) {               // :120-171
                    if (i < nF) {
                        const MeaEdge fe = F_AT(i);
                        if (fe.ref < r) {
                            if (i > max_i && max_prob < fe.sum) {
                                N_PUSH(fe.ref, fe.id, fe.sum);
                                max_prob = fe.sum;
                                max_i = i;
                            }
                            i++;
                        } else if (fe.ref == r) {
                            const double stay = fe.sum;
                            if (i == 0) {
                                if (stay > max_prob) {   // stay: the sum does not grow
                                    N_PUSH(r, na, stay);
                                    ARENA_PUSH(r, e, fe.id);
                                    max_prob = stay;
                                }
                            } else {
                                const MeaEdge left = F_AT(i - 1);
                                const double via = left.sum + p;
                                if (stay > via) {
                                    if (stay > max_prob) {
                                        N_PUSH(r, na, stay);
                                        ARENA_PUSH(r, e, fe.id);
                                        max_prob = stay;
                                    }
                                } else if (via > max_prob) {
                                    N_PUSH(r, na, via);
                                    ARENA_PUSH(r, e, left.id);
                                    max_prob = via;
                                }
                            }
                            max_i = i;
                            break;
                        } else {
                            if (i == 0) {
                                if (p > max_prob) {
                                    N_PUSH(r, na, p);
                                    ARENA_PUSH(r, e, -1);
                                    max_prob = p;
                                }
                            } else {
                                const MeaEdge left = F_AT(i - 1);
                                const double via = left.sum + p;
                                if (via > max_prob) {
                                    N_PUSH(r, na, via);
                                    ARENA_PUSH(r, e, left.id);
                                    max_prob = via;
                                }
                            }
                            break;
                        }
                    } else {             // the reference position lies past every edge
                        const MeaEdge left = F_AT(i - 1);
                        const double via = left.sum + p;
                        if (via > max_prob) {
                            N_PUSH(r, na, via);
                            ARENA_PUSH(r, e, left.id);
                            max_prob = via;
                        }
                        break;
                    }
                }
            }
        }
        // :174-180 trailing edges; max_prob is NOT raised here
        for (; i < nF; i++) {
            const MeaEdge fe = F_AT(i);
            if (fe.sum > max_prob) N_PUSH(fe.ref, fe.id, fe.sum);
        }
        SWAP_FRONTS();
        n_edges = nF;
        // :186-196 the first edge with the strictly highest sum above 0
        double highest = 0;
        int best_id = -1;
        for (int q = 0; q < nF; q++) {
            const MeaEdge fe = F_AT(q);
            if (fe.sum > highest) { highest = fe.sum; best_id = fe.id; }
        }
        if (best_id < 0) { status = SA_MEA_NO_PATH; goto done; }
        best_sum = highest;
        n_path = mea_traceback(a_ref, a_ev, a_prev, best_id, P.out + J.out_off, J.out_cap, lane, P.mins ? P.mins[2 * jb] : 0,
                               P.mins ? P.mins[2 * jb + 1] : 0);
        if (n_path < 0) { n_path = 0; status = SA_MEA_NO_PATH; }
    }
done:
    if (lane == 0) {
        P.status[jb] = status;
        P.n_out[jb] = status == SA_MEA_OK ? n_path : 0;
        P.n_edges[jb] = n_edges;
        P.sum[jb] = best_sum;
    }
#undef N_PUSH
#undef ARENA_PUSH
#undef SWAP_FRONTS
#undef F_AT
#undef FRONT
}

struct MeaWorkspace : SaScratch {
    void *d_ws = nullptr, *d_gf = nullptr, *h_in = nullptr, *h_res = nullptr;
    size_t d_ws_cap = 0, d_gf_cap = 0, h_in_cap = 0, h_res_cap = 0;
};
static MeaWorkspace g_mea_ws;

static void mea_chain_release();
extern "C" void sa_mea_release(void) {
    {
        std::lock_guard<std::mutex> guard(g_mea_ws.mu);
        g_mea_ws.release();
    }
    mea_chain_release();
}

#define MEACHK(call)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            fprintf(stderr, "[signalalign_hip] %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            rc = e_ == hipErrorOutOfMemory ? SA_ENOMEM : SA_ENODEVICE;                                      \
            goto done;                                                                                      \
        }                                                                                                   \
    } while (0)

// Three tiers of the same algorithm: fronts in registers (64 edges), in LDS (256), in global memory (any length); a read
// whose front outgrows a tier is handed to the next one.  SA_MEA_TIER=1|2 starts lower (tests).  h_status: pinned, one
// word per read.
static int mea_run_tiers(MeaWorkspace &W, MeaPlan &P, std::vector<MeaJob> &hj, char *d, size_t o_jobs, size_t o_ids,
                         int *h_status, int device, float *kms_out) {
    const size_t nj = hj.size();
    int rc = SA_OK;
    float kms2 = 0;
    std::vector<int> redo;
    const char *tier_env = getenv("SA_MEA_TIER");
    int tier = tier_env ? atoi(tier_env) : 0;
    tier = tier < 0 ? 0 : (tier > 2 ? 2 : tier);
    bool all = true;                       // first launch: every read, no id list
    for (; tier < 3; tier++) {
        const unsigned grid = all ? (unsigned) nj : (unsigned) redo.size();
        const int *d_ids = all ? nullptr : (const int *) (d + o_ids);
        if (tier == 2) {                   // two lists of n + 2 entries per read
            size_t gf_tot = 0;
            for (size_t q = 0; q < (all ? nj : redo.size()); q++) {
                const size_t j = all ? q : (size_t) redo[q];
                hj[j].gf_off = (long long) gf_tot;
                gf_tot += 2 * ((size_t) hj[j].n + 2);
            }
            if ((rc = W.dev(&W.d_gf, &W.d_gf_cap, sizeof(MeaEdge) * gf_tot, device)) != SA_OK) goto done;
            P.gf = (MeaEdge *) W.d_gf;
            MEACHK(hipMemcpyAsync(d + o_jobs, hj.data(), sizeof(MeaJob) * nj, hipMemcpyHostToDevice, 0));
        }
        if (!all) MEACHK(hipMemcpyAsync(d + o_ids, redo.data(), 4 * redo.size(), hipMemcpyHostToDevice, 0));
        MEACHK(hipEventRecord(W.e0, 0));
        if (tier == 0) {
            if (!all) { rc = SA_EINVAL; goto done; }   // the register tier is only ever the first
            hipLaunchKernelGGL(k_mea_wave, dim3(grid), dim3(64), 0, 0, P, (int) nj);
        } else if (tier == 1) {
            hipLaunchKernelGGL(k_mea<false>, dim3(grid), dim3(64), 0, 0, P, d_ids, (int) grid);
        } else {
            hipLaunchKernelGGL(k_mea<true>, dim3(grid), dim3(64), 0, 0, P, d_ids, (int) grid);
        }
        MEACHK(hipEventRecord(W.e1, 0));
        MEACHK(hipGetLastError());
        MEACHK(hipMemcpyAsync(h_status, P.status, 4 * nj, hipMemcpyDeviceToHost, 0));
        MEACHK(hipStreamSynchronize(0));
        MEACHK(hipEventElapsedTime(&kms2, W.e0, W.e1));
        *kms_out += kms2;
        redo.clear();
        for (size_t j = 0; j < nj; j++)
            if (h_status[j] == MEA_ST_OVERFLOW) redo.push_back((int) j);
        all = false;
        if (redo.empty()) break;
    }
done:
    return rc;
}

extern "C" int sa_mea_batch(const sa_mea_job_t *jobs, int64_t n_jobs, int device, unsigned flags, sa_mea_pair_t **path_out,
                            int64_t *n_path_out, double *sum_out, int32_t *status_out, int32_t *n_edges_out,
                            double *kernel_ms_out) {
    (void) flags;
    if ((!jobs && n_jobs > 0) || n_jobs < 0 || !path_out || !n_path_out) return SA_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[signalalign_hip] no HIP device available; this library has no CPU fallback\n");
        return SA_ENODEVICE;
    }
    if (device < 0 || device >= ndev) return SA_EINVAL;
    for (int64_t j = 0; j < n_jobs; j++) {
        path_out[j] = nullptr; n_path_out[j] = 0;
        if (status_out) status_out[j] = 0;
        if (sum_out) sum_out[j] = 0.0;
        if (n_edges_out) n_edges_out[j] = 0;
    }
    if (kernel_ms_out) *kernel_ms_out = 0.0;
    if (n_jobs == 0) return SA_OK;
    const size_t nj = (size_t) n_jobs;
    std::vector<MeaJob> hj(nj);
    size_t n_tot = 0, sh_tot = 0, out_tot = 0;
    for (size_t j = 0; j < nj; j++) {
        const sa_mea_job_t *jb = &jobs[j];
        if (jb->n < 0 || jb->n_events < 0 || jb->n > (1ll << 30) || jb->n_events > (1ll << 30) ||
            (jb->n > 0 && (!jb->event_idx || !jb->ref_idx || !jb->posterior)) || (jb->n_events > 0 && !jb->shortest_ref_per_event))
            return SA_EINVAL;
        MeaJob &J = hj[j];
        memset(&J, 0, sizeof(J));
        J.off = (long long) n_tot; J.sh_off = (long long) sh_tot; J.out_off = (long long) out_tot;
        J.n = (int) jb->n; J.n_sh = (int) jb->n_events;
        J.out_cap = (int) (jb->n < jb->n_events ? jb->n : jb->n_events);   // one pair per event at most
        n_tot += (size_t) jb->n; sh_tot += (size_t) jb->n_events; out_tot += (size_t) J.out_cap;
    }
    MeaWorkspace &W = g_mea_ws;
    std::lock_guard<std::mutex> guard(W.mu);
    int rc = SA_OK;
    // pinned upload image: data (f64) | rows | cols | shortest (i32)
    const size_t o_h_rows = sizeof(double) * n_tot, o_h_cols = o_h_rows + 4 * n_tot, o_h_sh = o_h_cols + 4 * n_tot,
                 in_bytes = o_h_sh + 4 * sh_tot;
    // device: [jobs | upload image | arena ref, ev, prev | out | sum | n_out, n_edges, status | ids]
    const size_t o_jobs = 0, o_in = sa_up256(sizeof(MeaJob) * nj), o_arena = sa_up256(o_in + in_bytes),
                 o_out = sa_up256(o_arena + 12 * n_tot), o_sum = sa_up256(o_out + 8 * out_tot), o_res = o_sum + 8 * nj,
                 o_ids = sa_up256(o_res + 12 * nj), dev_bytes = o_ids + 4 * nj;
    const size_t res_bytes = o_ids - o_out;
    MeaPlan P;
    memset(&P, 0, sizeof(P));
    float kms = 0;
    if ((rc = W.pin(&W.h_in, &W.h_in_cap, in_bytes ? in_bytes : 8, device)) != SA_OK) return rc;
    {
        char *h = (char *) W.h_in;
        sa_parallel_for(nj, [&](size_t j) {
            const sa_mea_job_t *jb = &jobs[j];
            const size_t n = (size_t) jb->n, ns = (size_t) jb->n_events, a = (size_t) hj[j].off, s = (size_t) hj[j].sh_off;
            if (n) {
                memcpy(h + 8 * a, jb->posterior, 8 * n);
                memcpy(h + o_h_rows + 4 * a, jb->event_idx, 4 * n);
                memcpy(h + o_h_cols + 4 * a, jb->ref_idx, 4 * n);
            }
            if (ns) memcpy(h + o_h_sh + 4 * s, jb->shortest_ref_per_event, 4 * ns);
        });
    }
    if ((rc = W.dev(&W.d_ws, &W.d_ws_cap, dev_bytes, device)) != SA_OK) goto done;
    if ((rc = W.pin(&W.h_res, &W.h_res_cap, res_bytes, device)) != SA_OK) goto done;
    if ((rc = W.events()) != SA_OK) goto done;
    {
        char *d = (char *) W.d_ws;
        MEACHK(hipMemcpyAsync(d + o_jobs, hj.data(), sizeof(MeaJob) * nj, hipMemcpyHostToDevice, 0));
        if (in_bytes) MEACHK(hipMemcpyAsync(d + o_in, W.h_in, in_bytes, hipMemcpyHostToDevice, 0));
        P.jobs = (const MeaJob *) (d + o_jobs);
        P.data = (const double *) (d + o_in);
        P.rows = (const int *) (d + o_in + o_h_rows);
        P.cols = (const int *) (d + o_in + o_h_cols);
        P.shortest = (const int *) (d + o_in + o_h_sh);
        P.a_ref = (int *) (d + o_arena); P.a_ev = P.a_ref + n_tot; P.a_prev = P.a_ev + n_tot;
        P.out = (int2 *) (d + o_out);
        P.sum = (double *) (d + o_sum);
        P.n_out = (int *) (d + o_res); P.n_edges = P.n_out + nj; P.status = P.n_edges + nj;
        if ((rc = mea_run_tiers(W, P, hj, d, o_jobs, o_ids, (int *) ((char *) W.h_res + (o_res - o_out)) + 2 * nj, device, &kms)) != SA_OK) goto done;
        MEACHK(hipMemcpyAsync(W.h_res, d + o_out, res_bytes, hipMemcpyDeviceToHost, 0));
        MEACHK(hipStreamSynchronize(0));
    }
    if (kernel_ms_out) *kernel_ms_out = (double) kms;
    {
        const char *hr = (const char *) W.h_res;
        const int2 *h_out = (const int2 *) hr;
        const double *h_sum = (const double *) (hr + (o_sum - o_out));
        const int *h_n = (const int *) (hr + (o_res - o_out)), *h_edges = h_n + nj, *h_status = h_edges + nj;
        std::atomic<bool> oom(false);
        sa_parallel_for(nj, [&](size_t j) {
            const int n = h_n[j];
            if (status_out) status_out[j] = h_status[j];
            if (sum_out) sum_out[j] = h_sum[j];
            if (n_edges_out) n_edges_out[j] = h_edges[j];
            n_path_out[j] = n;
            path_out[j] = (sa_mea_pair_t *) malloc(sizeof(sa_mea_pair_t) * (size_t) (n > 0 ? n : 1));
            if (!path_out[j]) { oom = true; return; }
            if (n > 0) memcpy(path_out[j], h_out + hj[j].out_off + hj[j].out_cap - n, sizeof(sa_mea_pair_t) * (size_t) n);
        });
        if (oom) rc = SA_ENOMEM;
    }
done:
    if (rc != SA_OK)
        for (int64_t j = 0; j < n_jobs; j++) { free(path_out[j]); path_out[j] = nullptr; n_path_out[j] = 0; }
    return rc;
}

// get_mea_params_from_events (mea_algorithm.py:267-320) without its dense matrices.  np.sort(events,
// order=['event_index']) breaks ties with the table's remaining fields in dtype order, so within an event rows come by
// reference index and rows of one cell by ascending posterior; the reference fills its matrices walking that order
// BACKWARDS, so a cell keeps the posterior of its first row, and coo_matrix(dense) then drops exact zeros.
extern "C" int64_t sa_mea_params(const int64_t *reference_index, const int64_t *event_index, const double *posterior,
                                 int64_t n, int32_t *event_idx_out, int32_t *ref_idx_out, double *posterior_out,
                                 int32_t *shortest_out, int64_t *n_events_out) {
    if (n <= 0 || !reference_index || !event_index || !posterior || !event_idx_out || !ref_idx_out || !posterior_out ||
        !shortest_out)
        return SA_EINVAL;
    int64_t r0 = reference_index[0], r1 = r0, e0 = event_index[0], e1 = e0;
    for (int64_t i = 1; i < n; i++) {
        r0 = reference_index[i] < r0 ? reference_index[i] : r0;
        r1 = reference_index[i] > r1 ? reference_index[i] : r1;
        e0 = event_index[i] < e0 ? event_index[i] : e0;
        e1 = event_index[i] > e1 ? event_index[i] : e1;
    }
    if (e1 - e0 >= (1ll << 30) || r1 - r0 >= (1ll << 30)) return SA_EINVAL;
    // :288-292 minus strand: the first sorted row (first event, its lowest reference index) lies above the last one
    // (last event, its highest reference index)
    int64_t lo_first = INT64_MAX, hi_last = INT64_MIN;
    for (int64_t i = 0; i < n; i++) {
        if (event_index[i] == e0 && reference_index[i] < lo_first) lo_first = reference_index[i];
        if (event_index[i] == e1 && reference_index[i] > hi_last) hi_last = reference_index[i];
    }
    const bool minus = lo_first > hi_last;
    const int64_t n_ev = e1 - e0 + 1;
    // bucket the rows by event (counting sort), order each bucket by (reference index, posterior)
    std::vector<int64_t> start((size_t) n_ev + 1, 0);
    for (int64_t i = 0; i < n; i++) start[(size_t) (event_index[i] - e0) + 1]++;
    for (int64_t e = 0; e < n_ev; e++) start[(size_t) e + 1] += start[(size_t) e];
    struct Cell {
        int32_t ref;
        double p;
    };
    std::vector<Cell> cells((size_t) n);
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t i = 0; i < n; i++) {
            Cell &c = cells[(size_t) fill[(size_t) (event_index[i] - e0)]++];
            c.ref = (int32_t) (minus ? r1 - reference_index[i] : reference_index[i] - r0);
            c.p = posterior[i];
        }
    }
    int64_t m = 0;
    for (int64_t e = 0; e < n_ev; e++) {
        Cell *b = cells.data() + start[(size_t) e], *end = cells.data() + start[(size_t) e + 1];
        // on the minus strand the flipped index falls where the table's rises; ties of one cell stay by posterior
        std::sort(b, end, [](const Cell &x, const Cell &y) { return x.ref != y.ref ? x.ref < y.ref : x.p < y.p; });
        for (Cell *c = b; c < end; c++) {
            if (c > b && c->ref == c[-1].ref) continue;   // a later row of the same cell is overwritten
            if (c->p == 0.0) continue;                    // coo_matrix keeps non-zeros only
            event_idx_out[m] = (int32_t) e;
            ref_idx_out[m] = c->ref;
            posterior_out[m] = c->p;
            m++;
        }
    }
    // :305-318 shortest_ref_per_event: the lowest reference position of this and every later event (zeros count);
    // an event without rows keeps inf
    int64_t lowest = INT64_MAX;
    for (int64_t e = n_ev - 1; e >= 0; e--) {
        const int64_t b = start[(size_t) e], end = start[(size_t) e + 1];
        if (b == end) { shortest_out[e] = SA_MEA_INF; continue; }
        for (int64_t q = b; q < end; q++) lowest = cells[(size_t) q].ref < lowest ? cells[(size_t) q].ref : lowest;
        shortest_out[e] = (int32_t) lowest;
    }
    if (n_events_out) *n_events_out = n_ev;
    return m;
}

// ---- chained onto a finished batch: the posterior matrices are built on the device ---------------------------------
// sa_batch_run leaves every read's aligned pairs in HBM.  mea_alignment_from_signal_align (mea_algorithm.py:323-341)
// would read them back from the TSV / fast5 table and run get_mea_params_from_events; here one wave per read turns the
// pairs into the COO matrix and shortest_ref_per_event in place (k_mea_from_pairs) and the MEA kernels follow, so
// nothing but the final paths crosses PCIe.
//
// The event table's posterior_probability is what the TSV prints, "%f" of prob_e7 / 1e7, six decimals: a decimal
// rounding of a binary double.  Only a last digit of 5 can tie; then the sign of q * 1e7 - prob_e7 (one fma, exact in
// sign) says on which side of the tie the double q = prob_e7 / 1e7 lies, and an exact tie goes to even as glibc's
// printf does.  tests/test_host_mea.py checks every value of prob_e7 against Python's "%f".
__host__ __device__ static inline double mea_printed_posterior(long long prob_e7) {
    long long k = prob_e7 / 10;
    const long long rem = prob_e7 % 10;
    if (rem > 5) {
        k++;
    } else if (rem == 5) {
        const double q = (double) prob_e7 / 1e7;
        const double side = fma(q, 1e7, -(double) prob_e7);
        if (side > 0 || (side == 0 && (k & 1))) k++;
    }
    return (double) k / 1e6;
}
extern "C" double sa_mea_printed_posterior(int64_t prob_e7) { return mea_printed_posterior((long long) prob_e7); }

struct MeaChain {
    long long pair_off;   // first pair of the read in the batch's device results
    int n_raw, n_events;  // pairs and events of the read
};

__device__ __forceinline__ int wave_max_i(int v) {
    for (int o = 32; o; o >>= 1) v = max(v, __shfl_xor(v, o));
    return uni(v);
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// next cell of an event's bucket [b0, b1) in reference order: the smallest reference position above `last`, and the
// smallest prob_e7 among its rows (numpy's field-order tie break leaves the lowest posterior first, and the first row
// of a cell is the one get_mea_params_from_events keeps)
__device__ __forceinline__ bool mea_next_cell(const int *sr, const int *sp, int b0, int b1, int last, int &ref, int &prob) {
    int best = 0x7fffffff;
    for (int t = b0; t < b1; t++) {
        const int r = sr[t];
        if (r > last && r < best) best = r;
    }
    if (best == 0x7fffffff) return false;
    int mp = 0x7fffffff;
    for (int t = b0; t < b1; t++)
        if (sr[t] == best) mp = min(mp, sp[t]);
    ref = best;
    prob = mp;
    return true;
}

__global__ __launch_bounds__(64) void k_mea_from_pairs(const sa_pair16_t *__restrict__ pairs, const MeaChain *__restrict__ chain,
                                                       MeaPlan P, int *fill, int *s_ref, int *s_prob, int *n_dev, int *mins,
                                                       int n_jobs) {
    const int jb = blockIdx.x, lane = threadIdx.x;
    const MeaChain C = chain[jb];
    const MeaJob J = P.jobs[jb];
    const sa_pair16_t *pr = pairs + C.pair_off;   // packed result records (sa_internal.h)
    int *fl = fill + J.sh_off, *sr = s_ref + J.off, *sp = s_prob + J.off;
    int *rows = const_cast<int *>(P.rows) + J.off, *cols = const_cast<int *>(P.cols) + J.off;
    double *data = const_cast<double *>(P.data) + J.off;
    int *sh = const_cast<int *>(P.shortest) + J.sh_off;
    const int n = C.n_raw;
    int xmin = 0x7fffffff, ymin = 0x7fffffff, ymax = -0x7fffffff;
    for (int j = lane; j < n; j += 64) {
        const sa_pair_t pj = sa_pair16_unpack(pr[j]);
        const int x = (int) pj.x, y = (int) pj.y;
        xmin = min(xmin, x); ymin = min(ymin, y); ymax = max(ymax, y);
    }
    xmin = wave_min(xmin); ymin = wave_min(ymin); ymax = wave_max_i(ymax);
    const int n_ev = n > 0 ? ymax - ymin + 1 : 0;
    if (n <= 0 || n_ev > J.n_sh) {   // no pairs (the MEA kernels report SA_MEA_EMPTY); the second cannot happen
        if (lane == 0) { n_dev[jb] = 0; mins[2 * jb] = 0; mins[2 * jb + 1] = 0; }
        return;
    }
    // rows per event -> bucket starts -> rows placed by event (order inside a bucket is irrelevant: mea_next_cell)
    for (int e = lane; e < n_ev; e += 64) fl[e] = 0;
    __threadfence();
    for (int j = lane; j < n; j += 64) atomicAdd(&fl[(int) sa_pair16_unpack(pr[j]).y - ymin], 1);
    __threadfence();
    int carry = 0;
    for (int base = 0; base < n_ev; base += 64) {
        const int e = base + lane;
        const int c = e < n_ev ? fl[e] : 0;
        const int incl = wave_incl_scan(c, lane);
        if (e < n_ev) fl[e] = carry + incl - c;
        carry += rl(incl, 63);
    }
    __threadfence();
    for (int j = lane; j < n; j += 64) {
        const sa_pair_t pj = sa_pair16_unpack(pr[j]);
        const int pos = atomicAdd(&fl[(int) pj.y - ymin], 1);   // fl[e] ends as the END of bucket e
        sr[pos] = (int) pj.x - xmin;
        sp[pos] = (int) pj.prob_e7;
    }
    __threadfence();
    // the COO matrix, events ascending, reference positions ascending inside an event, zeros dropped
    int n_out = 0;
    for (int base = 0; base < n_ev; base += 64) {
        const int e = base + lane;
        int b0 = 0, b1 = 0;
        if (e < n_ev) { b1 = fl[e]; b0 = e ? fl[e - 1] : 0; }
        int kept = 0, last = -1, ref = 0, prob = 0;
        while (mea_next_cell(sr, sp, b0, b1, last, ref, prob)) {
            last = ref;
            kept += mea_printed_posterior(prob) != 0.0;
        }
        const int incl = wave_incl_scan(kept, lane);
        int w = n_out + incl - kept;
        last = -1;
        while (mea_next_cell(sr, sp, b0, b1, last, ref, prob)) {
            last = ref;
            const double p = mea_printed_posterior(prob);
            if (p != 0.0) { rows[w] = e; cols[w] = ref; data[w] = p; w++; }
        }
        n_out += rl(incl, 63);
    }
    // :305-318 shortest_ref_per_event: the lowest reference position of this and every later event (zeros count), inf for
    // an event without rows
    int below = SA_MEA_INF;
    for (int base = ((n_ev - 1) / 64) * 64; base >= 0; base -= 64) {
        const int e = base + lane;
        int mn = SA_MEA_INF;
        bool any = false;
        if (e < n_ev) {
            const int b1 = fl[e], b0 = e ? fl[e - 1] : 0;
            any = b1 > b0;
            for (int t = b0; t < b1; t++) mn = min(mn, sr[t]);
        }
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_down(mn, o);
            if (lane + o < 64) mn = min(mn, t);
        }
        mn = min(mn, below);
        if (e < n_ev) sh[e] = any ? mn : SA_MEA_INF;
        below = rl(mn, 0);
    }
    if (lane == 0) { n_dev[jb] = n_out; mins[2 * jb] = xmin; mins[2 * jb + 1] = ymin; }
}

struct MeaChainWorkspace : MeaWorkspace {
    void *d_aux = nullptr;
    size_t d_aux_cap = 0;
};
static MeaChainWorkspace g_mea_chain_ws;
static void mea_chain_release() {
    std::lock_guard<std::mutex> guard(g_mea_chain_ws.mu);
    g_mea_chain_ws.release();
}

extern "C" int sa_batch_mea(sa_batch_t *b, unsigned flags, sa_mea_pair_t **path_out, int64_t *n_path_out, double *sum_out,
                            int32_t *status_out, double *kernel_ms_out) {
    (void) flags;
    if (!b || !path_out || !n_path_out) return SA_EINVAL;
    const bool trace = getenv("SA_TRACE") != nullptr;
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t0 = now_ms();
    const sa_pair16_t *d_pairs = nullptr;
    std::vector<long long> first, count, n_events;
    int device = 0;
    int rc = sa_batch_device_view(b, &d_pairs, &first, &count, &n_events, &device);
    if (rc) return rc;
    const size_t nj = first.size();
    for (size_t j = 0; j < nj; j++) {
        path_out[j] = nullptr; n_path_out[j] = 0;
        if (status_out) status_out[j] = 0;
        if (sum_out) sum_out[j] = 0.0;
    }
    if (kernel_ms_out) *kernel_ms_out = 0.0;
    if (nj == 0) return SA_OK;
    std::vector<MeaJob> hj(nj);
    std::vector<MeaChain> hc(nj);
    size_t n_tot = 0, sh_tot = 0, out_tot = 0;
    for (size_t j = 0; j < nj; j++) {
        if (count[j] > (1ll << 30) || n_events[j] > (1ll << 30)) return SA_EINVAL;
        MeaJob &J = hj[j];
        memset(&J, 0, sizeof(J));
        J.off = (long long) n_tot; J.sh_off = (long long) sh_tot; J.out_off = (long long) out_tot;
        J.n = (int) count[j]; J.n_sh = (int) n_events[j] + 1;
        J.out_cap = (int) (count[j] < n_events[j] ? count[j] : n_events[j]);
        hc[j].pair_off = first[j]; hc[j].n_raw = (int) count[j]; hc[j].n_events = (int) n_events[j];
        n_tot += (size_t) count[j]; sh_tot += (size_t) n_events[j] + 1; out_tot += (size_t) J.out_cap;
    }
    if (trace) fprintf(stderr, "[trace] mea: view+layout %.3f ms\n", now_ms() - t0);
    MeaChainWorkspace &W = g_mea_chain_ws;
    std::lock_guard<std::mutex> guard(W.mu);
    // device: [jobs | chain | data f64 | rows | cols | shortest | arena ref, ev, prev | out | sum | n_out, n_edges, status | mins | ids]
    // aux (dead after k_mea_from_pairs): [fill | sorted ref | sorted prob]
    const size_t o_jobs = 0, o_chain = sa_up256(sizeof(MeaJob) * nj), o_data = sa_up256(o_chain + sizeof(MeaChain) * nj),
                 o_rows = sa_up256(o_data + 8 * n_tot), o_cols = sa_up256(o_rows + 4 * n_tot), o_sh = sa_up256(o_cols + 4 * n_tot),
                 o_arena = sa_up256(o_sh + 4 * sh_tot), o_out = sa_up256(o_arena + 12 * n_tot),
                 o_sum = sa_up256(o_out + 8 * out_tot), o_res = o_sum + 8 * nj, o_mins = o_res + 12 * nj,
                 o_ndev = o_mins + 8 * nj, o_ids = sa_up256(o_ndev + 4 * nj), dev_bytes = o_ids + 4 * nj;
    const size_t res_bytes = o_ids - o_out;
    const size_t a_fill = 0, a_sref = sa_up256(4 * sh_tot), a_sprob = sa_up256(a_sref + 4 * n_tot), aux_bytes = a_sprob + 4 * n_tot;
    MeaPlan P;
    memset(&P, 0, sizeof(P));
    float kms = 0, kms0 = 0;
    if ((rc = W.dev(&W.d_ws, &W.d_ws_cap, dev_bytes, device)) != SA_OK) return rc;
    if ((rc = W.dev(&W.d_aux, &W.d_aux_cap, aux_bytes ? aux_bytes : 256, device)) != SA_OK) return rc;
    if ((rc = W.pin(&W.h_res, &W.h_res_cap, res_bytes, device)) != SA_OK) return rc;
    if ((rc = W.events()) != SA_OK) return rc;
    {
        char *d = (char *) W.d_ws, *a = (char *) W.d_aux;
        MEACHK(hipMemcpyAsync(d + o_jobs, hj.data(), sizeof(MeaJob) * nj, hipMemcpyHostToDevice, 0));
        MEACHK(hipMemcpyAsync(d + o_chain, hc.data(), sizeof(MeaChain) * nj, hipMemcpyHostToDevice, 0));
        P.jobs = (const MeaJob *) (d + o_jobs);
        P.data = (const double *) (d + o_data);
        P.rows = (const int *) (d + o_rows);
        P.cols = (const int *) (d + o_cols);
        P.shortest = (const int *) (d + o_sh);
        P.a_ref = (int *) (d + o_arena); P.a_ev = P.a_ref + n_tot; P.a_prev = P.a_ev + n_tot;
        P.out = (int2 *) (d + o_out);
        P.sum = (double *) (d + o_sum);
        P.n_out = (int *) (d + o_res); P.n_edges = P.n_out + nj; P.status = P.n_edges + nj;
        P.n_dev = (const int *) (d + o_ndev);
        P.mins = (const int *) (d + o_mins);
        MEACHK(hipEventRecord(W.e0, 0));
        hipLaunchKernelGGL(k_mea_from_pairs, dim3((unsigned) nj), dim3(64), 0, 0, d_pairs, (const MeaChain *) (d + o_chain), P,
                           (int *) (a + a_fill), (int *) (a + a_sref), (int *) (a + a_sprob), (int *) (d + o_ndev),
                           (int *) (d + o_mins), (int) nj);
        MEACHK(hipEventRecord(W.e1, 0));
        MEACHK(hipGetLastError());
        MEACHK(hipStreamSynchronize(0));
        MEACHK(hipEventElapsedTime(&kms0, W.e0, W.e1));
        if (trace) fprintf(stderr, "[trace] mea: matrices built at %.3f ms\n", now_ms() - t0);
        if ((rc = mea_run_tiers(W, P, hj, d, o_jobs, o_ids, (int *) ((char *) W.h_res + (o_res - o_out)) + 2 * nj, device, &kms)) != SA_OK)
            goto done;
        if (trace) fprintf(stderr, "[trace] mea: paths done at %.3f ms\n", now_ms() - t0);
        MEACHK(hipMemcpyAsync(W.h_res, d + o_out, res_bytes, hipMemcpyDeviceToHost, 0));
        MEACHK(hipStreamSynchronize(0));
        if (trace) fprintf(stderr, "[trace] mea: %zu bytes on the host at %.3f ms\n", res_bytes, now_ms() - t0);
    }
    if (kernel_ms_out) *kernel_ms_out = (double) kms + (double) kms0;
    {
        const char *hr = (const char *) W.h_res;
        const int2 *h_out = (const int2 *) hr;
        const double *h_sum = (const double *) (hr + (o_sum - o_out));
        const int *h_n = (const int *) (hr + (o_res - o_out)), *h_status = h_n + 2 * nj;
        std::atomic<bool> oom(false);
        sa_parallel_for(nj, [&](size_t j) {
            const int n = h_n[j];
            if (status_out) status_out[j] = h_status[j];
            if (sum_out) sum_out[j] = h_sum[j];
            n_path_out[j] = n;
            path_out[j] = (sa_mea_pair_t *) malloc(sizeof(sa_mea_pair_t) * (size_t) (n > 0 ? n : 1));
            if (!path_out[j]) { oom = true; return; }
            // already in the batch's own coordinates: the traceback added the matrix origin back
            if (n > 0) memcpy(path_out[j], h_out + hj[j].out_off + hj[j].out_cap - n, sizeof(sa_mea_pair_t) * (size_t) n);
        });
        if (oom) rc = SA_ENOMEM;
    }
done:
    if (trace) fprintf(stderr, "[trace] mea: outputs built at %.3f ms\n", now_ms() - t0);
    if (rc != SA_OK)
        for (size_t j = 0; j < nj; j++) { free(path_out[j]); path_out[j] = nullptr; n_path_out[j] = 0; }
    return rc;
}

// the device's evaluation of mea_printed_posterior for prob_e7 = first .. first + n - 1 (tests: it must agree with the host's
// for every value; both are the same source, but the division and the fma are the device's own)
__global__ void k_printed_posterior(long long first, long long n, double *out) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mea_printed_posterior(first + i);
}
extern "C" int sa_mea_printed_posterior_device(int64_t first, int64_t n, double *out, int device) {
    if (n < 0 || (n > 0 && !out) || n > (1ll << 31)) return SA_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SA_ENODEVICE;
    if (device < 0 || device >= ndev) return SA_EINVAL;
    if (n == 0) return SA_OK;
    double *d = nullptr;
    if (hipSetDevice(device) != hipSuccess) return SA_ENODEVICE;
    if (hipMalloc((void **) &d, sizeof(double) * (size_t) n) != hipSuccess) return SA_ENOMEM;
    hipLaunchKernelGGL(k_printed_posterior, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, (long long) first, (long long) n, d);
    hipError_t e = hipMemcpy(out, d, sizeof(double) * (size_t) n, hipMemcpyDeviceToHost);
    (void) hipFree(d);
    return e == hipSuccess ? SA_OK : SA_ENODEVICE;
}
