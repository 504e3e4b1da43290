// sa_ea.hip -- event <-> k-mer pre-alignment on the GPU.
//
// What it replaces: adaptive_banded_simple_event_align2 (impl/eventAligner.c:899-1235), the step immediately upstream
// of the pair-HMM: Suzuki-Kasahara adaptive banding as used by nanopolish.  Bands are anti-diagonals of the
// (event+1) x (k-mer+1) matrix, 100 cells wide; every new band steps right or down from the previous one depending on
// which end of it scores higher; Viterbi scores over three moves (step, stay, skip) kept as floats; traceback from the
// best (event, last k-mer) cell; three quality checks.
//
// Mapping: one wave per read.  The recurrence is serial in the band index (~ events + k-mers steps), the 100 cells of a
// band are independent: lane l computes offsets l and l+64.  The three live bands sit in LDS (the neighbours of a
// cell are at the same or an adjacent offset of the two previous bands, which one depends on the moves taken); the
// trace (one byte per cell) and the band origins go to global memory for the traceback, which lane 0 walks at the end.
// Arithmetic follows the reference's order of operations and float casts (the file is compiled with
// -ffp-contract=off), with the emission of the memory-resident EXACT kernels: results are bit-identical to the CPU
// restatement in oracle/sa_oracle.c.  PARITY UNPINNED against the reference itself: its tests of this function need
// fast5 files (tests/eventAlignerTests.c:223-320, :404-430).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "sa_internal.h"
#include "sa_scratch.h"

#define EA_BW 100
#define EA_HALF 50
#define EA_NEG_INF (-__builtin_inf())

struct EaJob {
    long long ev_off, kc_off, trace_off, ll_off, col_off, out_off;  // offsets into the shared arrays
    int n_events, n_kmers;
    double lp_skip, lp_stay, lp_step, lp_trim, events_per_kmer;
    double scale, shift, var, lvar;
};

struct EaPlan {
    const EaJob *jobs;
    const double *ev;        // event means
    const double *kc;        // per k-mer position: mu, sd, c = -log(sqrt(2 pi)) - log(sd)   (3 doubles)
    unsigned char *trace;    // [n_bands][100] per read
    int *ll;                 // [n_bands][2] per read: event / k-mer index of the band's offset 0
    double *col;             // [n_events] per read: score of (event, last k-mer)
    int *out;                // [cap][2] per read: (k-mer, event) pairs, traceback order
    int *n_out;              // per read
    int *status;             // per read
    int *fills;              // per read: cells filled (the reference's `fills` counter)
};

// emissions_signal_strawManGetKmerEventMatchProbWithDescaling_MeanOnly (impl/stateMachine.c:557-605), the operation
// order of oracle/sa_oracle.c:emit
__device__ __forceinline__ double ea_emit(const double *kc, double e, const EaJob &J) {
    const double mu = kc[0], sd = kc[1], c = kc[2];
    const double en = (e + J.var * mu - J.scale * mu - J.shift) / J.var;
    const double a = (en - mu) / sd;
    return J.lvar + (c + (-0.5 * a * a));
}

// One block of 128 threads (two waves) per read: thread o owns band offset o (100 of them).  What a band cell needs
// from memory -- its event's mean, its k-mer's three constants -- is data dependent (the band's position is decided
// band by band) but only ever advances by one event or one k-mer per band, so both streams are staged ahead of the band
// in LDS circular buffers (256 slots each, refilled 64 at a time by coalesced loads well before the band arrives) and
// the serial chain of a band never waits for HBM: per band it is LDS reads, the emission, three scores and a barrier.
#define EA_THREADS 128
#define EA_SLOTS 256   // power of two, >= band width + refill chunk + lead
// UNIT_VAR: every read of the launch has var == 1 (the reference always aligns with var = 1, impl/eventAligner.c:845-849):
// x / 1.0 == x, one division per cell less.
template <bool UNIT_VAR>
__global__ __launch_bounds__(EA_THREADS) void k_event_align(EaPlan P, int n_jobs) {
    __shared__ double ring[3][EA_BW];
    __shared__ double kmu[EA_SLOTS], ksd[EA_SLOTS], kcc[EA_SLOTS], ebuf[EA_SLOTS];
    __shared__ int s_fills;
    const int job = blockIdx.x;
    if (job >= n_jobs) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const EaJob J = P.jobs[job];
    const double *ev = P.ev + J.ev_off;
    const double *kc = P.kc + 3 * J.kc_off;
    unsigned char *trace = P.trace + J.trace_off;
    int *ll = P.ll + 2 * J.ll_off;
    double *col = P.col + J.col_off;
    const int n_events = J.n_events, n_kmers = J.n_kmers;
    const long long n_bands = (long long) (n_events + 1) + (n_kmers + 1);

    int my_fills = 0;
    if (tid == 0) s_fills = 0;
    for (int i = tid; i < n_events; i += EA_THREADS) col[i] = EA_NEG_INF;
    for (int o = tid; o < EA_BW; o += EA_THREADS) {
        ring[0][o] = (o == EA_HALF) ? 0.0 : EA_NEG_INF;        // band 0: (event -1, k-mer -1) at offset 50
        ring[1][o] = (o == EA_HALF) ? J.lp_trim : EA_NEG_INF;  // band 1: first event trimmed
    }
    // first 192 k-mers and events (the first bands reach k-mer 48 and event 50 at most)
    int k_loaded = 0, e_loaded = 0;
    for (; k_loaded < n_kmers && k_loaded < 192; k_loaded += 64)
        if (tid < 64 && k_loaded + tid < n_kmers) {
            const int idx = k_loaded + tid;
            kmu[idx & (EA_SLOTS - 1)] = kc[3ll * idx]; ksd[idx & (EA_SLOTS - 1)] = kc[3ll * idx + 1];
            kcc[idx & (EA_SLOTS - 1)] = kc[3ll * idx + 2];
        }
    for (; e_loaded < n_events && e_loaded < 192; e_loaded += 64)
        if (tid >= 64 && e_loaded + tid - 64 < n_events) ebuf[(e_loaded + tid - 64) & (EA_SLOTS - 1)] = ev[e_loaded + tid - 64];
    int ll_ev1 = EA_HALF, ll_km1 = -1 - EA_HALF;   // band b-1
    int ll_ev2 = EA_HALF - 1, ll_km2 = -1 - EA_HALF;  // band b-2
    if (tid == 0) {
        ll[0] = ll_ev2; ll[1] = ll_km2;
        ll[2] = ll_ev1; ll[3] = ll_km1;
    }
    __syncthreads();
    const int o = tid;
    // ring rows of bands b-2, b-1, b (rotated, not recomputed with % 3); clamped neighbour offsets
    int r2 = 0, r1 = 1, r0 = 2;
    const int om1 = o > 0 ? o - 1 : 0, oc = o < EA_BW ? o : EA_BW - 1, op1 = o + 1 < EA_BW ? o + 1 : EA_BW - 1;
    for (int b = 2; b < (int) n_bands; b++) {
        const double *prev1 = ring[r1], *prev2 = ring[r2];
        double *cur = ring[r0];
        // Everything this band can need from LDS is requested in ONE batch before the band's position is known: the two
        // band ends that decide the move, the three neighbours of either move in band b-1 and the two possible diagonal
        // neighbours in band b-2, and the operands of both candidate cells (k-mer km or km + 1, event e or e + 1).  One
        // LDS round trip per band instead of five dependent ones.
        const int dk = ll_km1 - ll_km2;                      // 0 or 1: the move that produced band b-1
        const double lo = prev1[0], hi = prev1[EA_BW - 1];
        const double p1m = prev1[om1], p1c = prev1[oc], p1p = prev1[op1];
        const int od0 = o - 1 + dk;                          // diagonal neighbour if this band moves down, +1 if right
        const double p2a = prev2[od0 < 0 ? 0 : (od0 < EA_BW ? od0 : EA_BW - 1)];
        const double p2b = prev2[od0 + 1 < 0 ? 0 : (od0 + 1 < EA_BW ? od0 + 1 : EA_BW - 1)];
        const int km_d = ll_km1 + o, e_r = ll_ev1 - o;      // cell of offset o after a down move / a right move
        const double mu_d = kmu[km_d & (EA_SLOTS - 1)], sd_d = ksd[km_d & (EA_SLOTS - 1)], c_d = kcc[km_d & (EA_SLOTS - 1)];
        const double mu_r = kmu[(km_d + 1) & (EA_SLOTS - 1)], sd_r = ksd[(km_d + 1) & (EA_SLOTS - 1)],
                     c_r = kcc[(km_d + 1) & (EA_SLOTS - 1)];
        const double ev_r = ebuf[e_r & (EA_SLOTS - 1)], ev_d = ebuf[(e_r + 1) & (EA_SLOTS - 1)];
        bool right;
        if (lo == EA_NEG_INF && hi == EA_NEG_INF) right = (b % 2) == 1;  // both ends outside the matrix: alternate
        else right = lo < hi;                                           // Suzuki's rule
        const int ll_ev = ll_ev1 + (right ? 0 : 1), ll_km = ll_km1 + (right ? 1 : 0);
        if (tid == 0) { ll[2 * b] = ll_ev; ll[2 * b + 1] = ll_km; }
        // keep the streams ahead of the band: k-mers up to ll_km + 99 and events up to ll_ev are needed now; a chunk
        // lands at least 32 bands before its first use and overwrites slots the band left 60 and more bands ago
        if (k_loaded < n_kmers && ll_km + EA_BW + 32 > k_loaded) {
            if (tid < 64 && k_loaded + tid < n_kmers) {
                const int idx = k_loaded + tid;
                kmu[idx & (EA_SLOTS - 1)] = kc[3ll * idx]; ksd[idx & (EA_SLOTS - 1)] = kc[3ll * idx + 1];
                kcc[idx & (EA_SLOTS - 1)] = kc[3ll * idx + 2];
            }
            k_loaded += 64;
        }
        if (e_loaded < n_events && ll_ev + 1 + 32 > e_loaded) {
            if (tid >= 64 && e_loaded + tid - 64 < n_events) ebuf[(e_loaded + tid - 64) & (EA_SLOTS - 1)] = ev[e_loaded + tid - 64];
            e_loaded += 64;
        }
        int o_min = 0 - ll_km, o_max = n_kmers - ll_km;
        const int e_min = ll_ev - (n_events - 1), e_max = ll_ev + 1;
        o_min = e_min > o_min ? e_min : o_min;
        o_min = o_min < 0 ? 0 : o_min;
        o_max = e_max < o_max ? e_max : o_max;
        o_max = o_max > EA_BW ? EA_BW : o_max;
        const int trim_o = -1 - ll_km;
        if (o < EA_BW) {
            double val = EA_NEG_INF;
            if (o == trim_o) {  // k-mer -1: every event so far trimmed
                const int e = ll_ev - o;
                if (e >= 0 && e < n_events) val = J.lp_trim * (double) (e + 1);
            }
            if (o >= o_min && o < o_max) {
                const int e = ll_ev - o, km = ll_km + o;
                // o_up = ll_ev1 - (e - 1) = o + 1 (right) or o (down); o_left = (km - 1) - ll_km1 = o (right) or o - 1
                // (down); o_diag = (km - 1) - ll_km2 = o - 1 + dk (+ 1 if right)
                const int o_up = right ? o + 1 : o, o_left = right ? o : o - 1, o_diag = od0 + (right ? 1 : 0);
                const float up = (o_up >= 0 && o_up < EA_BW) ? (float) (right ? p1p : p1c) : -__builtin_inff();
                const float left = (o_left >= 0 && o_left < EA_BW) ? (float) (right ? p1c : p1m) : -__builtin_inff();
                const float diag = (o_diag >= 0 && o_diag < EA_BW) ? (float) (right ? p2b : p2a) : -__builtin_inff();
                // ea_emit on the staged operands (same operations, same order)
                const double mu = right ? mu_r : mu_d, sd = right ? sd_r : sd_d, c = right ? c_r : c_d;
                const double num = (right ? ev_r : ev_d) + J.var * mu - J.scale * mu - J.shift;
                const double en = UNIT_VAR ? num : num / J.var;
                const double a = (en - mu) / sd;
                const double lp_em = J.lvar + (c + (-0.5 * a * a));
                const float s_d = (float) ((double) diag + J.lp_step + lp_em);
                const float s_u = (float) ((double) up + J.lp_stay + lp_em);
                const float s_l = (float) ((double) left + J.lp_skip);
                float best = s_d;
                unsigned char from = 0;               // FROM_D
                best = s_u > best ? s_u : best;
                from = best == s_u ? 1 : from;        // FROM_U
                best = s_l > best ? s_l : best;
                from = best == s_l ? 2 : from;        // FROM_L
                val = (double) best;
                trace[(long long) b * EA_BW + o] = from;
                if (km == n_kmers - 1) col[e] = val;
                my_fills++;
            }
            cur[o] = val;
        }
        ll_ev2 = ll_ev1; ll_km2 = ll_km1;
        ll_ev1 = ll_ev; ll_km1 = ll_km;
        { const int t = r2; r2 = r1; r1 = r0; r0 = t; }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) my_fills += __shfl_xor(my_fills, off, 64);
    if (lane == 0) atomicAdd(&s_fills, my_fills);
    __threadfence();   // the second wave's col[] entries, read by the first below
    __syncthreads();
    if (tid >= 64) return;
    my_fills = s_fills;
    // best (event, last k-mer) cell with the events behind it trimmed: first maximum, as the reference's scan
    float best = -__builtin_inff();
    int best_ev = 0x7fffffff;
    for (int e = lane; e < n_events; e += 64) {
        const float s = (float) (col[e] + (double) (n_events - e) * J.lp_trim);
        if (s > best) { best = s; best_ev = e; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oe = __shfl_xor(best_ev, off, 64);
        if (ob > best || (ob == best && oe < best_ev)) { best = ob; best_ev = oe; }
    }
    if (lane != 0) return;
    P.fills[job] = my_fills;
    // traceback (lane 0): the scan starts from event 0 when nothing scored (best == -inf), like the reference
    int cur_ev = (best > -__builtin_inff()) ? best_ev : 0, cur_km = n_kmers - 1;
    int *out = P.out + 2 * J.out_off;
    int n = 0, cur_gap = 0, max_gap = 0;
    double sum_em = 0;
    while (cur_km >= 0 && cur_ev >= 0) {
        out[2 * n] = cur_km; out[2 * n + 1] = cur_ev; n++;
        sum_em += ea_emit(kc + 3ll * cur_km, ev[cur_ev], J);
        const long long b = (long long) (cur_ev + 1) + (cur_km + 1);
        const int o = ll[2 * b] - cur_ev;
        const unsigned char from = trace[b * EA_BW + o];
        if (from == 0) { cur_km--; cur_ev--; cur_gap = 0; }
        else if (from == 1) { cur_ev--; cur_gap = 0; }
        else { cur_km--; cur_gap++; max_gap = cur_gap > max_gap ? cur_gap : max_gap; }
    }
    int st = 0;
    const double avg = sum_em / (double) n;
    if (avg < -5.2) st |= 1;
    // pairs are in traceback order: the last one is the front of the reversed list
    if (!(n > 0 && out[2 * (n - 1)] == 0 && out[0] == n_kmers - 1)) st |= 2;
    if (max_gap > 50) st |= 4;
    if (J.events_per_kmer > 5.0) st |= 8;
    P.status[job] = st;
    P.n_out[job] = st ? 0 : n;
}

// k-mer ids of every position as build_kmer_list (impl/eventAligner.c:772-790) lists them (for RNA, U reads as T and
// every k-mer is reversed), computed by rolling the id along the sequence.  Returns SA_EALPHABET for a foreign letter.
static int ea_kmer_ids(const sa_model_t *m, const char *seq, int64_t n_kmers, bool rna, int32_t *out) {
    int8_t digit[256];
    memset(digit, -1, sizeof(digit));
    for (int i = 0; i < m->n_alpha; i++) digit[(unsigned char) m->alphabet[i]] = (int8_t) i;
    if (rna && digit[(unsigned char) 'T'] >= 0) digit[(unsigned char) 'U'] = digit[(unsigned char) 'T'];
    const int k = m->k;
    const int64_t A = m->n_alpha, top = m->pow_km1, full = top * A;  // A^(k-1), A^k
    const int64_t n = n_kmers + k - 1;
    for (int64_t i = 0; i < n; i++)
        if (digit[(unsigned char) seq[i]] < 0) return SA_EALPHABET;
    int64_t id = 0;
    if (!rna) {  // forward k-mers, left to right: shift in the new letter, take out the one that left the window
        for (int64_t i = 0; i < n; i++) {
            id = id * A + digit[(unsigned char) seq[i]];
            if (i >= k) id -= digit[(unsigned char) seq[i - k]] * full;
            if (i >= k - 1) out[i - (k - 1)] = (int32_t) id;
        }
    } else {     // reversed k-mers (letter j of the window weighs A^j), right to left for the same division-free roll
        for (int64_t i = n - 1; i >= 0; i--) {
            id = id * A + digit[(unsigned char) seq[i]];
            if (i + k < n) id -= digit[(unsigned char) seq[i + k]] * full;
            if (i < n_kmers) out[i] = (int32_t) id;
        }
    }
    return SA_OK;
}

#define EACHK(call)                                                                                         \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            fprintf(stderr, "[signalalign_hip] %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            rc = e_ == hipErrorOutOfMemory ? SA_ENOMEM : SA_ENODEVICE;                                      \
            goto done;                                                                                      \
        }                                                                                                   \
    } while (0)

extern "C" int sa_scalings_mom(const sa_model_t *m, const char *sequence, int64_t seq_len, const double *event_mean,
                               int64_t n_events, unsigned flags, double *shift_out, double *scale_out) {
    if (!m || !sequence || !event_mean || !shift_out || !scale_out) return SA_EINVAL;
    const int64_t n_kmers = seq_len - (m->k - 1);
    if (n_kmers <= 0 || n_events <= 0) return SA_EINVAL;
    double ev_sum = 0.0f;
    for (int64_t i = 0; i < n_events; i++) ev_sum += event_mean[i];
    double km_sum = 0.0f, km_sq = 0.0f;
    std::vector<int32_t> ids;
    ids.resize((size_t) n_kmers);
    int rck = ea_kmer_ids(m, sequence, n_kmers, (flags & SA_FLAG_RNA) != 0, ids.data());
    if (rck) return rck;
    for (int64_t i = 0; i < n_kmers; i++) {
        int64_t id = ids[(size_t) i];
        double level = m->table5[5 * id];
        km_sum += level;
        km_sq += pow(level, 2.0f);
    }
    double shift = ev_sum / (double) n_events - km_sum / (double) n_kmers;
    double ev_sq = 0.0f;
    for (int64_t i = 0; i < n_events; i++) ev_sq += pow(event_mean[i] - shift, 2.0);
    *shift_out = shift;
    *scale_out = (ev_sq / (double) n_events) / (km_sq / (double) n_kmers);
    return SA_OK;
}

// kc[i] = constants of the k-mer at position i (24 bytes per position written on the device instead of uploaded)
__global__ void k_ea_expand(const double *__restrict__ kt, const int32_t *__restrict__ ids, double *__restrict__ kc, long long n) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *src = kt + 3ll * ids[i];
    kc[3 * i] = src[0];
    kc[3 * i + 1] = src[1];
    kc[3 * i + 2] = src[2];
}

// Device and pinned-host scratch of sa_event_align_batch, kept between calls (sa_scratch.h).
struct EaWorkspace : SaScratch {
    void *d_ws = nullptr, *d_trace = nullptr, *h_in = nullptr, *h_res = nullptr;
    size_t d_ws_cap = 0, d_trace_cap = 0, h_in_cap = 0, h_res_cap = 0;
};
static EaWorkspace g_ea_ws;
static inline size_t ea_up(size_t x) { return sa_up256(x); }

extern "C" void sa_event_align_release(void) {
    std::lock_guard<std::mutex> guard(g_ea_ws.mu);
    g_ea_ws.release();
}

extern "C" int sa_event_align_batch(const sa_model_t *m, const sa_ea_job_t *jobs, int64_t n_jobs, int device, unsigned flags,
                                    sa_ea_pair_t **pairs_out, int64_t *n_pairs_out, int32_t *status_out, double *cells_out,
                                    double *kernel_ms_out) {
    if (!m || (!jobs && n_jobs > 0) || n_jobs < 0 || !pairs_out || !n_pairs_out) return SA_EINVAL;
    if (m->hdp) return SA_EUNSUPPORTED;  // the reference builds this aligner's state machine without an HDP
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[signalalign_hip] no HIP device available; this library has no CPU fallback\n");
        return SA_ENODEVICE;
    }
    if (device < 0 || device >= ndev) return SA_EINVAL;
    for (int64_t j = 0; j < n_jobs; j++) { pairs_out[j] = nullptr; n_pairs_out[j] = 0; if (status_out) status_out[j] = 0; }
    if (n_jobs == 0) return SA_OK;
    // host side: k-mer constants and offsets
    std::vector<EaJob> hj((size_t) n_jobs);
    size_t ev_tot = 0, kc_tot = 0;
    for (int64_t j = 0; j < n_jobs; j++) {
        const sa_ea_job_t *jb = &jobs[j];
        const int64_t n_kmers = jb->seq_len - (m->k - 1);
        if (!jb->sequence || !jb->event_mean || n_kmers <= 0 || jb->n_events <= 0 || !(jb->var > 0.0) ||
            jb->n_events > (1 << 24) || n_kmers > (1 << 24))
            return SA_EINVAL;
        ev_tot += (size_t) jb->n_events;
        kc_tot += 3 * (size_t) n_kmers;
    }
    EaWorkspace &W = g_ea_ws;
    std::lock_guard<std::mutex> guard(W.mu);
    int rc = SA_OK;
    if (hipSetDevice(device) != hipSuccess) return SA_ENODEVICE;
    // pinned upload image: event means | per-model k-mer constants {level mean, level sd, -log(sqrt(2 pi)) - log(sd)} |
    // k-mer id of every position (the device expands ids to per-position constants, k_ea_expand)
    const size_t kt_n = 3 * (size_t) m->n_kmers, in_bytes = sizeof(double) * (ev_tot + kt_n) + sizeof(int32_t) * (kc_tot / 3);
    if ((rc = W.pin(&W.h_in, &W.h_in_cap, in_bytes, device)) != SA_OK) return rc;
    double *ev = (double *) W.h_in, *kt = ev + ev_tot;
    int32_t *ids = (int32_t *) (kt + kt_n);
    for (int64_t id = 0; id < m->n_kmers; id++) {
        const double mu = m->table5[5 * id], sd = m->table5[5 * id + 1];
        kt[3 * (size_t) id] = mu;
        kt[3 * (size_t) id + 1] = sd == 0.0 ? 1.0 : sd;
        kt[3 * (size_t) id + 2] = sd == 0.0 ? -INFINITY : (-0.91893853320467267 - log(sd));
    }
    size_t ev_n = 0, kc_n = 0;
    long long trace_tot = 0, ll_tot = 0, col_tot = 0, out_tot = 0;
    bool all_unit_var = true;
    for (int64_t j = 0; j < n_jobs; j++) {
        const sa_ea_job_t *jb = &jobs[j];
        const int64_t n_kmers = jb->seq_len - (m->k - 1);
        EaJob &J = hj[(size_t) j];
        J.ev_off = (long long) ev_n;
        J.kc_off = (long long) (kc_n / 3);
        J.n_events = (int) jb->n_events;
        J.n_kmers = (int) n_kmers;
        memcpy(ev + ev_n, jb->event_mean, sizeof(double) * (size_t) jb->n_events);
        ev_n += (size_t) jb->n_events;
        int rck = ea_kmer_ids(m, jb->sequence, n_kmers, (flags & SA_FLAG_RNA) != 0, ids + kc_n / 3);
        if (rck) return rck;
        kc_n += 3 * (size_t) n_kmers;
        const long long n_bands = (long long) (jb->n_events + 1) + (n_kmers + 1);
        J.trace_off = trace_tot; trace_tot += n_bands * EA_BW;
        J.ll_off = ll_tot; ll_tot += n_bands;
        J.col_off = col_tot; col_tot += jb->n_events;
        J.out_off = out_tot; out_tot += jb->n_events + n_kmers + 2;
        // transition penalties (impl/eventAligner.c:928-940)
        J.events_per_kmer = (double) jb->n_events / (double) n_kmers;
        const double p_stay = 1 - (1 / (J.events_per_kmer + 1));
        J.lp_skip = log(1e-10);
        J.lp_stay = log(p_stay);
        J.lp_step = log(1.0 - exp(J.lp_skip) - exp(J.lp_stay));
        J.lp_trim = log(0.01);
        J.scale = jb->scale; J.shift = jb->shift; J.var = jb->var;
        all_unit_var = all_unit_var && jb->var == 1.0;
        J.lvar = log((1 / jb->var));
    }
    EaPlan P;
    memset(&P, 0, sizeof(P));
    float kms = 0;
    const size_t nj = (size_t) n_jobs;
    // device workspace, kept between calls (grow only): [jobs | upload image | kc | ll | col | out | n,status,fills] and the trace plane
    const size_t o_jobs = 0, o_in = ea_up(o_jobs + sizeof(EaJob) * nj), o_kc = ea_up(o_in + in_bytes),
                 o_ll = ea_up(o_kc + sizeof(double) * kc_tot),
                 o_col = ea_up(o_ll + sizeof(int) * 2 * (size_t) ll_tot), o_out = ea_up(o_col + sizeof(double) * (size_t) col_tot),
                 o_res = ea_up(o_out + sizeof(int) * 2 * (size_t) out_tot), dev_bytes = o_res + sizeof(int) * 3 * nj;
    const size_t res_bytes = dev_bytes - o_out;  // pair lists and the three per-read result words come back in one copy
    const int *h_out, *h_n, *h_st, *h_fills;
    if ((rc = W.dev(&W.d_ws, &W.d_ws_cap, dev_bytes, device)) != SA_OK) goto done;
    if ((rc = W.dev(&W.d_trace, &W.d_trace_cap, (size_t) trace_tot, device)) != SA_OK) goto done;
    if ((rc = W.pin(&W.h_res, &W.h_res_cap, res_bytes, device)) != SA_OK) goto done;
    if ((rc = W.events()) != SA_OK) goto done;
    {
        char *d = (char *) W.d_ws;
        EACHK(hipMemcpyAsync(d + o_jobs, hj.data(), sizeof(EaJob) * nj, hipMemcpyHostToDevice, 0));
        EACHK(hipMemcpyAsync(d + o_in, W.h_in, in_bytes, hipMemcpyHostToDevice, 0));
        P.jobs = (EaJob *) (d + o_jobs); P.ev = (double *) (d + o_in); P.kc = (double *) (d + o_kc);
        P.trace = (unsigned char *) W.d_trace; P.ll = (int *) (d + o_ll); P.col = (double *) (d + o_col);
        P.out = (int *) (d + o_out); P.n_out = (int *) (d + o_res); P.status = P.n_out + nj; P.fills = P.status + nj;
        EACHK(hipEventRecord(W.e0, 0));
        {
            const long long n_pos = (long long) (kc_tot / 3);
            const double *d_kt = P.ev + ev_tot;
            hipLaunchKernelGGL(k_ea_expand, dim3((unsigned) ((n_pos + 255) / 256)), dim3(256), 0, 0, d_kt,
                               (const int32_t *) (d_kt + kt_n), (double *) (d + o_kc), n_pos);
        }
        if (all_unit_var)
            hipLaunchKernelGGL(k_event_align<true>, dim3((unsigned) n_jobs), dim3(EA_THREADS), 0, 0, P, (int) n_jobs);
        else
            hipLaunchKernelGGL(k_event_align<false>, dim3((unsigned) n_jobs), dim3(EA_THREADS), 0, 0, P, (int) n_jobs);
        EACHK(hipEventRecord(W.e1, 0));
        EACHK(hipGetLastError());
        EACHK(hipMemcpyAsync(W.h_res, d + o_out, res_bytes, hipMemcpyDeviceToHost, 0));
        EACHK(hipStreamSynchronize(0));
        EACHK(hipEventElapsedTime(&kms, W.e0, W.e1));
    }
    if (kernel_ms_out) *kernel_ms_out = (double) kms;
    h_out = (const int *) W.h_res;
    h_n = (const int *) ((const char *) W.h_res + (o_res - o_out));
    h_st = h_n + nj; h_fills = h_st + nj;
    if (cells_out)
        for (int64_t j = 0; j < n_jobs; j++) cells_out[j] = (double) h_fills[(size_t) j];
    for (int64_t j = 0; j < n_jobs; j++) {
        const int n = h_n[(size_t) j];
        if (status_out) status_out[j] = h_st[(size_t) j];
        n_pairs_out[j] = n;
        pairs_out[j] = (sa_ea_pair_t *) malloc(sizeof(sa_ea_pair_t) * (size_t) (n > 0 ? n : 1));
        if (!pairs_out[j]) { rc = SA_ENOMEM; goto done; }
        const int *src = h_out + 2 * hj[(size_t) j].out_off;
        for (int i = 0; i < n; i++) {  // stList_reverse: ascending order
            pairs_out[j][i].kmer_idx = src[2 * (n - 1 - i)];
            pairs_out[j][i].event_idx = src[2 * (n - 1 - i) + 1];
        }
    }
done:
    if (rc != SA_OK)
        for (int64_t j = 0; j < n_jobs; j++) { free(pairs_out[j]); pairs_out[j] = nullptr; n_pairs_out[j] = 0; }
    return rc;
}
