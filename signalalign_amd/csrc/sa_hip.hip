// sa_hip.hip -- gfx950 kernels and the batch runtime behind include/signalalign_hip.h.
//
// Device work per batch (all inputs resident in HBM before the first launch):
//   1. forward sweep      one wavefront per split region, anti-diagonal after anti-diagonal
//   2. backward sweep     one wavefront per traceback segment (independent of each other), fused with the
//                         posterior numerator f.match+b.match and the per-cell terms of totalProbability
//   3. fold               one lane per checkpoint: the reference's strictly sequential logAdd fold of the
//                         per-cell terms (kept sequential on purpose: logAdd is a piecewise cubic, not associative)
//   4. finalize/scan/gather   posterior = exp(fb - total), threshold, floor(p*1e7), compaction into the
//                         reference's output order
//
// Two kernel families share the data layout decisions but not the code:
//   * generic (k_*_generic): any band width, any number of paths per cell, HDP emissions, reference-ordered
//     un-contracted arithmetic.  State lives in memory.  This is the exactness baseline.
//   * fast (k_*_fast): one path per cell, Gaussian emissions.  The two previous anti-diagonals live in
//     registers; lane = ((x-y+K)>>1) mod 64, so a cell's middle neighbour is in the same lane and its
//     lower/upper neighbours are in the same or an adjacent lane, alternating with the diagonal's parity.
//
// The file is compiled with -ffp-contract=off; where fused multiply-add is wanted it is spelled fma().
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <new>
#include <thread>
#include <time.h>
#include <vector>

#include "sa_internal.h"
#include "sa_scratch.h"

#define NEG_INF (-__builtin_inf())

// ---------------------------------------------------------------------------------------------------
// device-side views
// ---------------------------------------------------------------------------------------------------
struct DevModel {
    double t_mm, t_mx, t_my, t_xm, t_xx, t_ym, t_yy;
    const double *tab6;     // per k-mer: mu, sd, c(sd), sdY, c(sdY), 0   with c(s) = -log(sqrt(2 pi)) - log(s)
    long long pow_km1;
    int n_alpha;
    int hdp;
    int emission;           // 0: MeanOnly (what signalMachine installs); 1 / 2: the two-distribution emission with / without descaling (sa_model_set_emission)
    const double *noise3;   // emission 1: per k-mer noise mean, noise lambda, log(lambda)  (columns 2 and 4 of the model table)
    const int *hdp_slot;    // per k-mer: row of y/slope tables of the first observed ancestor, -1 if none
    const double *hdp_y, *hdp_slope, *hdp_grid;
    const double *hdp_tab;  // register kernels: {y[i], slope[i]} interleaved, one row per observed process
    const double *hdp_coef; // k_emit_hdp: the four cubic coefficients of interval i of every row (twice the size of hdp_tab)
    double hdp_g0, hdp_gN, hdp_dx;
    unsigned hdp_tab_bytes;
    unsigned hdp_hot;       // byte offset in hdp_tab of the row most k-mers resolve to (0xffffffff: no such row)
    int grid_len;
};

struct DevPlan {
    const sa_region_t *regions;
    const sa_row_t *rows;
    const int *pk;
    const int *poff;
    const int *pid;
    const int *px;        // reference position (region-relative x) of every pid entry: cell-path -> cell
    const double *xc;
    const sa_prec_t *prec; // per cell-path records of SA_KIND_RING regions with several paths per cell
    const double *ev;
    const double *evn;    // two-distribution emission only: per event its noise and log(noise) (C library's log, from the host)
    const sa_seg_t *segs;
    const sa_ck_t *cks;
    double *F;
    double *E;        // HDP models: emission plane of the register-kernel regions (k_emit_hdp), max_chunk_cellpaths doubles
    const double *two;     // two-distribution emission on the register kernels (k_*_fast_two): the noise constants, else nullptr
    long long two_xn_off;  // ... first entry (double4) of the per-position part (FastT.two_xn_off)
    double *vbuf;
    sa_cand_t *cands;
    int *cand_count;
    int *overflow;
    double *totals;
    double *bscratch;
    double *gsum;     // EXPECT: 8 doubles per checkpoint group (7 live transitions)
    double *gmc;      // EXPECT: the group's scaling maximum
    DevModel m;
    double log_thr;   // log(threshold)
    double threshold;
    double *spec;     // ring / strip kernels: per traceback segment its speculative total (NaN: a segment of another kernel family)
    double spec_slack;
    int expect;       // the expectation pass (sa_expect_batch)
};

// ---------------------------------------------------------------------------------------------------
// logAdd: impl/pairwiseAligner.c:298-318.  The coefficients are float literals promoted to double.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double la_lookup(double x) {
    if (x <= 1.00f)
        return ((-0.009350833524763f * x + 0.130659527668286f) * x + 0.498799810682272f) * x + 0.693203116424741f;
    if (x <= 2.50f)
        return ((-0.014532321752540f * x + 0.139942324101744f) * x + 0.495635523139337f) * x + 0.692140569840976f;
    if (x <= 4.50f)
        return ((-0.004605031767994f * x + 0.063427417320019f) * x + 0.695956496475118f) * x + 0.514272634594009f;
    return ((-0.000458661602210f * x + 0.009695946122598f) * x + 0.930734667215156f) * x + 0.168037164329057f;
}

__device__ __forceinline__ double la_exact(double x, double y) {
    if (x < y) return (x == NEG_INF || y - x >= 7.5) ? y : la_lookup(y - x) + x;
    return (y == NEG_INF || x - y >= 7.5) ? x : la_lookup(x - y) + y;
}

// ---------------------------------------------------------------------------------------------------
// emissions in reference order (impl/stateMachine.c:296-306, :344-348, :527-605)
// ---------------------------------------------------------------------------------------------------
struct ReadPar {
    double scale, shift, var, lvar;
    const double *evn;   // the region's {noise, log noise} pairs (two-distribution emission), else nullptr
};

__device__ __forceinline__ double hdp_interp(const DevModel &m, int slot, double q) {
    const double *x = m.hdp_grid;
    const double *y = m.hdp_y + (long long) slot * m.grid_len;
    const double *s = m.hdp_slope + (long long) slot * m.grid_len;
    int n = m.grid_len;
    if (q <= x[0]) return y[0] - s[0] * (x[0] - q);
    if (q >= x[n - 1]) return y[n - 1] + s[n - 1] * (q - x[n - 1]);
    double dx = x[1] - x[0];
    long long il = (long long) ((q - x[0]) / dx);
    long long ir = il + 1;
    double dy = y[ir] - y[il];
    double a = s[il] * dx - dy;
    double b = dy - s[ir] * dx;
    double tl = (q - x[il]) / dx;
    double tr = 1.0 - tl;
    return tr * y[il] + tl * y[ir] + tl * tr * (a * tr + b * tl);
}

// match != 0: EMISSION_MATCH_MATRIX; == 0: EMISSION_GAP_Y_MATRIX (sd * 1.75).  id < 0: NULL k-mer.
// yi: index of the event (matrix row - 1), read by the two-distribution emission only
__device__ __forceinline__ double emit_ref(const DevModel &m, const ReadPar &rp, int id, double e, int match, long long yi) {
    if (id < 0) return NEG_INF;
    const double *t = m.tab6 + 6ll * id;
    double mu = t[0];
    double en = (e + rp.var * mu - rp.scale * mu - rp.shift) / rp.var;
    if (m.emission != 0) {
        // emissions_signal_strawManGetKmerEventMatchProbWithDescaling (impl/stateMachine.c:607-650): logGaussPdf of the descaled
        // mean + logInvGaussPdf of the event noise (:296-306, :320-330), in the reference's order of operations; emission 2 is
        // emissions_signal_strawManGetKmerEventMatchProb (:659-700): the same on the event mean as it is (the MODEL was scaled)
        if (m.emission == 2) en = e;
        double sd = match ? t[1] : t[3];
        double c = match ? t[2] : t[4];  // -log(sqrt(2 pi)) - log(sd); -inf when sd == 0
        double a = (en - mu) / sd;
        double l1 = c + (-0.5 * a * a);
        double n = rp.evn[2 * yi], l_n = rp.evn[2 * yi + 1];
        const double *nz = m.noise3 + 3ll * id;
        double a2 = (n - nz[0]) / nz[0];
        double l2 = (nz[2] - 1.8378770664093453 - 3 * l_n - nz[1] * a2 * a2 / n) / 2;
        return l1 + l2;
    }
    if (m.hdp) {
        int slot = m.hdp_slot[id];
        if (slot < 0) return NEG_INF;
        double d = hdp_interp(m, slot, en);
        d = d > 0.0 ? d : 0.0;
        double density = (1 / rp.var) * d;
        return log(density);
    }
    double sd = match ? t[1] : t[3];
    double c = match ? t[2] : t[4];  // -inf when sd == 0
    double a = (en - mu) / sd;
    return rp.lvar + (c + (-0.5 * a * a));
}

__device__ __forceinline__ bool legal_step(const DevModel &m, int from, int to) {
    if (from < 0 || to < 0) return true;
    return (from % m.pow_km1) == (to / m.n_alpha);
}

// Maximum over the 64 lanes, wave-uniform (values may be -inf, never NaN).  Six DPP steps -- butterflies inside a row of 16
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then row_bcast15 into rows 1 and 3 and row_bcast31 into rows 2 and 3 --
// leave the maximum in lane 63, which v_readlane hands to every lane as a scalar: 33 issue slots.  The generic __shfl_xor
// butterfly costs 85 per reduction (ds_bpermute pairs, lane arithmetic, selects, their waits and hazards), and the backward
// sweeps reduce twice per checkpoint, i.e. twice per ten diagonals: an eighth of k_bwd_fast's instructions.
// A lane without a source (or outside the row mask) keeps `old` = its own value: max(v, v).
__device__ __forceinline__ double wave_max(double v) {
    int lo_, hi_;
    double o_;
    // (butterflies: every lane has a source, the move needs no old value -- no copy in front of it)
#define SA_WMAX_BFLY(CTRL)                                                                       \
    lo_ = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);                    \
    hi_ = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);                    \
    o_ = __hiloint2double(hi_, lo_);                                                             \
    asm("v_max_f64 %0, %1, %2" : "=v"(v) : "v"(v), "v"(o_));
    // (broadcasts: a lane outside the row mask keeps what the previous step left in the same registers -- a value that
    // already went into its maximum)
#define SA_WMAX_BCAST(CTRL, ROWS)                                                                \
    lo_ = __builtin_amdgcn_update_dpp(lo_, __double2loint(v), CTRL, ROWS, 0xF, false);           \
    hi_ = __builtin_amdgcn_update_dpp(hi_, __double2hiint(v), CTRL, ROWS, 0xF, false);           \
    o_ = __hiloint2double(hi_, lo_);                                                             \
    asm("v_max_f64 %0, %1, %2" : "=v"(v) : "v"(v), "v"(o_));
    SA_WMAX_BFLY(0xB1)           // quad_perm:[1,0,3,2]
    SA_WMAX_BFLY(0x4E)           // quad_perm:[2,3,0,1]
    SA_WMAX_BFLY(0x141)          // row_half_mirror
    SA_WMAX_BFLY(0x140)          // row_mirror: every lane holds its row's maximum
    SA_WMAX_BCAST(0x142, 0xA)    // row_bcast15 -> rows 1, 3
    SA_WMAX_BCAST(0x143, 0xC)    // row_bcast31 -> rows 2, 3: lane 63 holds the wave's
#undef SA_WMAX_BFLY
#undef SA_WMAX_BCAST
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
// Sum over the 64 lanes, wave-uniform, with the same six DPP steps (the butterflies are true pairings, so every lane of a row
// ends with its row's sum; a lane outside a broadcast's row mask adds 0.0).  The order of the additions differs from a serial
// sum's, as the __shfl_xor butterfly's did.
__device__ __forceinline__ double wave_sum(double v) {
    int lo_, hi_;
#define SA_WSUM_BFLY(CTRL)                                                                       \
    lo_ = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);                    \
    hi_ = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);                    \
    v += __hiloint2double(hi_, lo_);
#define SA_WSUM_BCAST(CTRL, ROWS)                                                                \
    lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xF, false);             \
    hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xF, false);             \
    v += __hiloint2double(hi_, lo_);
    SA_WSUM_BFLY(0xB1)
    SA_WSUM_BFLY(0x4E)
    SA_WSUM_BFLY(0x141)
    SA_WSUM_BFLY(0x140)
    SA_WSUM_BCAST(0x142, 0xA)
    SA_WSUM_BCAST(0x143, 0xC)
#undef SA_WSUM_BFLY
#undef SA_WSUM_BCAST
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ int wave_max_i(int v) {
    for (int off = 32; off > 0; off >>= 1) {
        int o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

#include "sa_fast.inc"
#include "sa_ring.inc"
#include "sa_strip.inc"

// ---------------------------------------------------------------------------------------------------
// The memory-resident kernels come in two flavours.  EXACT (SA_FLAG_EXACT, the expectation pass, HDP with several
// paths per cell): the reference's arithmetic in the reference's order, rows read back from global memory.
// RELAX (default for everything the register kernels cannot take: several paths per cell, windows wider than 64
// lanes): the same recurrence with the register kernels' arithmetic -- logAdd from the LDS table, Gaussian emissions
// from the folded per-position constants, legality by a float-reciprocal division -- and the three live diagonals in
// an LDS ring (dynamic shared memory: 68 doubles of logAdd table + 3 x ring_cap x 3 doubles; ring_cap == 0 keeps the
// rows in global memory).  Results agree with EXACT to ~1e-9 on a posterior (bar: 1e-5).
// ---------------------------------------------------------------------------------------------------
template <bool RELAX>
__device__ __forceinline__ double la_any(const double *LT, double x, double y) {
    return RELAX ? la_fast(LT, x, y) : la_exact(x, y);
}
// k-mer ids are < 2^24: exact in float; one correction step makes the truncated quotient exact
__device__ __forceinline__ int div_small(int a, int d, float inv_d) {
    int q = (int) ((float) a * inv_d);
    int r = a - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}
template <bool RELAX>
__device__ __forceinline__ bool legal_any(const DevModel &m, float inv_pow, float inv_alpha, int from, int to) {
    if (!RELAX) return legal_step(m, from, to);
    if (from < 0 || to < 0) return true;
    int fq = div_small(from, (int) m.pow_km1, inv_pow);
    return from - fq * (int) m.pow_km1 == div_small(to, m.n_alpha, inv_alpha);
}

// One cell-path of forward diagonal d.  P1 / P2 are rows d-1 / d-2; the function is instantiated twice so that, when both
// rows sit in the LDS ring (the rule), the compiler sees shared-memory pointers and emits ds_read: a pointer that may
// be either LDS or global is a FLAT access, and flat loads wait on the vector-memory counter as well, i.e. on the
// stores of the previous diagonal to the forward storage.
template <bool RELAX>
__device__ __forceinline__ void fwd_generic_cellpath(const DevModel &m, const ReadPar &rp, const double *LT, float inv_pow,
                                                     float inv_alpha, const sa_row_t &rd, const sa_row_t &r1, const sa_row_t &r2,
                                                     long long d, long long x01, long long x02, const int *poff, const int *pid,
                                                     const int *px, const double *ev, const double4 *xc4, const double *P1,
                                                     const double *P2, double *F, double *L0, int g0, int j, bool lds0) {
    const int g = g0 + j;
    const long long x = px[g];
    const int p = g - poff[x];
    const long long xmy = 2 * x - d, y = d - x;
    double e = y >= 1 ? ev[y - 1] : NEG_INF;
    const int id = pid[g];
    double *cur = F + 3 * (rd.foff + j);
    double *lcur = L0 + 3 * j;
    long long i_lo = xmy - 1 - r1.xmyL, i_up = xmy + 1 - r1.xmyL, i_mid = xmy - r2.xmyL;
    bool has_lo = x >= 1 && i_lo >= 0 && (i_lo >> 1) < r1.width;
    bool has_up = i_up >= 0 && (i_up >> 1) < r1.width;
    bool has_mid = d >= 2 && x >= 1 && i_mid >= 0 && (i_mid >> 1) < r2.width;
    const double *lo = has_lo ? P1 + 3 * (poff[x - 1] - poff[x01]) : nullptr;
    const double *up = has_up ? P1 + 3 * (poff[x] - poff[x01]) : nullptr;
    const double *mid = has_mid ? P2 + 3 * (poff[x - 1] - poff[x02]) : nullptr;
    int nq = x >= 1 ? poff[x] - poff[x - 1] : 0;
    const int *idq = x >= 1 ? pid + poff[x - 1] : nullptr;
    {
        double sm = NEG_INF, sx = NEG_INF, sy = NEG_INF;
        double eM, eY;  // match / gapY emission of this cell-path
        if (RELAX) {
            emit_gauss(xc4[g], e, eM, eY);
        } else {
            eM = has_mid ? emit_ref(m, rp, id, e, 1, y - 1) : NEG_INF;
            eY = has_up ? emit_ref(m, rp, id, e, 0, y - 1) : NEG_INF;
        }
        if (has_lo) {
            double eP = (m.hdp || id >= 0) ? SA_LOG_GAPX : NEG_INF;
            for (int q = 0; q < nq; q++)
                if (legal_any<RELAX>(m, inv_pow, inv_alpha, idq[q], id)) {
                    sx = la_any<RELAX>(LT, sx, lo[3 * q + 0] + (eP + m.t_mx));
                    sx = la_any<RELAX>(LT, sx, lo[3 * q + 1] + (eP + m.t_xx));
                }
        }
        if (has_mid) {
            double eP = eM;
            for (int q = 0; q < nq; q++)
                if (legal_any<RELAX>(m, inv_pow, inv_alpha, idq[q], id)) {
                    sm = la_any<RELAX>(LT, sm, mid[3 * q + 0] + (eP + m.t_mm));
                    sm = la_any<RELAX>(LT, sm, mid[3 * q + 1] + (eP + m.t_xm));
                    sm = la_any<RELAX>(LT, sm, mid[3 * q + 2] + (eP + m.t_ym));
                }
        }
        if (has_up) {
            double eP = eY;
            sy = la_any<RELAX>(LT, sy, up[3 * p + 0] + (eP + m.t_my));
            sy = la_any<RELAX>(LT, sy, up[3 * p + 2] + (eP + m.t_yy));
        }
        cur[0] = sm;
        cur[1] = sx;
        cur[2] = sy;
        if (lds0) { lcur[0] = sm; lcur[1] = sx; lcur[2] = sy; }
    }

}

// ---------------------------------------------------------------------------------------------------
// generic forward: cellCalculate with doTransitionForward (impl/stateMachine.c:1306-1437,
// impl/pairwiseAligner.c:852-858, :1280-1322).  Row layout: [cell-path][3].
// ---------------------------------------------------------------------------------------------------
template <bool RELAX>
__global__ __launch_bounds__(128) void k_fwd_generic(DevPlan P, const int *region_ids, int n, int ring_cap) {
    extern __shared__ __attribute__((aligned(32))) double dyn_lds[];
    double *LT = dyn_lds;                          // RELAX only
    double *lring = dyn_lds + LA_TAB_DOUBLES;      // RELAX && ring_cap > 0: rows d, d-1, d-2 as [cell-path][3]
    int w = blockIdx.x;
    if (w >= n) return;
    // one lane per cell-path of a diagonal: 64 threads, or 128 (two waves sharing the LDS ring) when some diagonal of the
    // launch holds more than 64 cell-paths -- ambiguous positions put ~70 on a 51-cell band, and a second pass of one
    // wave over the last few would double the time of every diagonal
    const int lane = threadIdx.x, nthr = blockDim.x;
    if (RELAX) {
        la_tab_init(LT, lane);
        __syncthreads();
    }
    const bool use_ring = RELAX && ring_cap > 0;
    const float inv_pow = 1.0f / (float) P.m.pow_km1, inv_alpha = 1.0f / (float) P.m.n_alpha;
    const sa_region_t *R = &P.regions[region_ids[w]];
    const double4 *xc4 = reinterpret_cast<const double4 *>(P.xc) + R->pid_off;
    const sa_row_t *rows = P.rows + R->row_off;
    const int *poff = P.poff + R->poff_off;
    const int *pid = P.pid + R->pid_off;
    const int *px = P.px + R->pid_off;
    const double *ev = P.ev + R->ev_off;
    double *F = P.F + 3 * R->f_base;
    const DevModel &m = P.m;
    ReadPar rp = {R->scale, R->shift, R->var, R->lvar, P.evn ? P.evn + 2 * R->ev_off : nullptr};
    const long long N = R->N;

    {   // diagonal 0: startStateProb / raggedStartStateProb (impl/stateMachine.c:1134-1143)
        sa_row_t r0 = rows[0];
        long long x0 = (0 + r0.xmyL) / 2;
        for (int i = lane; i < r0.width; i += nthr) {
            long long x = x0 + i;
            int np = poff[x + 1] - poff[x];
            double *c = F + 3 * (r0.foff + poff[x] - poff[x0]);
            double *lc = lring + 3 * (poff[x] - poff[x0]);
            const bool row_in_lds = use_ring && poff[x0 + r0.width] - poff[x0] <= ring_cap;
            for (int p = 0; p < np; p++) {
                c[3 * p + 0] = R->ragged_l ? NEG_INF : 0.0;
                c[3 * p + 1] = R->ragged_l ? 0.0 : NEG_INF;
                c[3 * p + 2] = R->ragged_l ? 0.0 : NEG_INF;
                if (row_in_lds) { lc[3 * p + 0] = c[3 * p + 0]; lc[3 * p + 1] = c[3 * p + 1]; lc[3 * p + 2] = c[3 * p + 2]; }
            }
        }
    }
    __syncthreads();
    for (long long d = 1; d <= N; d++) {
        sa_row_t rd = rows[d], r1 = rows[d - 1];
        sa_row_t r2 = {0, 0, 0};
        if (d >= 2) r2 = rows[d - 2];
        long long x0 = (d + rd.xmyL) >> 1;
        long long x01 = (d - 1 + r1.xmyL) >> 1;
        long long x02 = d >= 2 ? ((d - 2 + r2.xmyL) >> 1) : 0;
        // previous diagonals: the LDS ring, or the forward storage itself
        // a diagonal lives in the ring if it fits (ring_cap cell-paths); the few that do not are read back from F
        const bool lds1 = use_ring && poff[x01 + r1.width] - poff[x01] <= ring_cap;
        const bool lds2 = use_ring && d >= 2 && poff[x02 + r2.width] - poff[x02] <= ring_cap;
        const double *P1 = lds1 ? lring + ((d - 1) % 3) * (long long) ring_cap * 3 : F + 3 * r1.foff;
        const double *P2 = lds2 ? lring + ((d + 1) % 3) * (long long) ring_cap * 3 : F + 3 * r2.foff;
        double *L0 = lring + (d % 3) * (long long) ring_cap * 3;
        // one lane per cell-path of the diagonal (cells with many paths would otherwise serialise the whole wave)
        const int g0 = poff[x0];
        const int rowpaths = poff[x0 + rd.width] - g0;
        const bool lds0 = use_ring && rowpaths <= ring_cap;
        if (lds1 && (lds2 || d < 2)) {   // both previous rows in the ring: shared-memory accesses
            const double *Q1 = lring + ((d - 1) % 3) * (long long) ring_cap * 3;
            const double *Q2 = lring + ((d + 1) % 3) * (long long) ring_cap * 3;
            for (int j = lane; j < rowpaths; j += nthr)
                fwd_generic_cellpath<RELAX>(m, rp, LT, inv_pow, inv_alpha, rd, r1, r2, d, x01, x02, poff, pid, px, ev, xc4, Q1, Q2, F,
                                            L0, g0, j, lds0);
        } else {
            for (int j = lane; j < rowpaths; j += nthr)
                fwd_generic_cellpath<RELAX>(m, rp, LT, inv_pow, inv_alpha, rd, r1, r2, d, x01, x02, poff, pid, px, ev, xc4, P1, P2, F,
                                            L0, g0, j, lds0);
        }
        __syncthreads();
    }
}

// EXPECT mode: close a checkpoint group -- lane 0 stores the wave sums of exp(term - Mc) and Mc itself; the host
// rescales by exp(Mc - totalProbability) once the exact fold of the group's total is known.
__device__ __forceinline__ void expect_flush(const DevPlan &P, long long ck, double Mc, double *acc, int lane) {
    for (int k = 0; k < 7; k++) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) P.gsum[ck * 8 + k] = v;
        acc[k] = 0.0;
    }
    if (lane == 0) { P.gsum[ck * 8 + 7] = 0.0; P.gmc[ck] = Mc; }
}

// ---------------------------------------------------------------------------------------------------
// generic backward + posterior numerators + checkpoint terms.
// The reference scatters (doTransitionBackward, impl/pairwiseAligner.c:866-871); here each cell GATHERS
// the same terms in the same order: first from (x+1,y+1) (it was that cell's "middle"), then from (x,y+1)
// (its "upper"), then from (x+1,y) (its "lower").  Backward rows live in a 3-row ring in memory.
// ---------------------------------------------------------------------------------------------------
template <bool EXPECT, bool RELAX>
__global__ __launch_bounds__(128) void k_bwd_generic(DevPlan P, const int *seg_ids, int n, int ring_cap) {
    extern __shared__ __attribute__((aligned(32))) double dyn_lds[];
    double *LT = dyn_lds;                          // RELAX only
    double *lring = dyn_lds + LA_TAB_DOUBLES;      // RELAX && ring_cap > 0: backward rows e, e+1, e+2
    int w = blockIdx.x;
    if (w >= n) return;
    // 64 or 128 threads (see k_fwd_generic): the cell-path sweep of a diagonal is shared by all threads; what follows a
    // diagonal (checkpoint terms, candidates, expectations: per cell, wave-wide scans) is the first wave's alone, while
    // the second goes on to the barrier of the next diagonal
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const bool first_wave = tid < 64;
    if (RELAX) {
        la_tab_init(LT, tid);
        __syncthreads();
    }
    const bool use_ring = RELAX && ring_cap > 0;
    const float inv_pow = 1.0f / (float) P.m.pow_km1, inv_alpha = 1.0f / (float) P.m.n_alpha;
    const int seg = seg_ids[w];
    const sa_seg_t *S = &P.segs[seg];
    const sa_region_t *R = &P.regions[S->region];
    const sa_row_t *rows = P.rows + R->row_off;
    const int *poff = P.poff + R->poff_off;
    const int *pid = P.pid + R->pid_off;
    const double *ev = P.ev + R->ev_off;
    const double *F = P.F + 3 * R->f_base;
    const DevModel &m = P.m;
    ReadPar rp = {R->scale, R->shift, R->var, R->lvar, P.evn ? P.evn + 2 * R->ev_off : nullptr};
    // backward rows: the LDS ring for diagonals of at most ring_cap cell-paths, the global ring for the others
    const long long grow = R->max_rowpaths;
    double *gring = P.bscratch + S->bscratch_off;  // 3 rows x grow x 3
    const double4 *xc4 = reinterpret_cast<const double4 *>(P.xc) + R->pid_off;
    const int *px = P.px + R->pid_off;
    const long long start = S->start, from = S->from, to = S->to;
    double end_m, end_x, end_y;  // endStateProb / raggedEndStateProb (impl/stateMachine.c:1145-1173)
    if (S->at_end && R->ragged_r) {
        end_m = (m.t_mx + m.t_my) / 2.0; end_x = m.t_xx; end_y = m.t_yy;
    } else {
        end_m = m.t_mm; end_x = m.t_xm; end_y = m.t_ym;
    }
    int count = 0;
    double Mc = NEG_INF;
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};  // EXPECT: sum of exp(term - Mc) per live transition, current checkpoint group
    for (long long e = start; e > to; e--) {
        sa_row_t re = rows[e];
        long long x0 = (e + re.xmyL) >> 1;
        auto row_ptr = [&](long long row, long long xfirst, int width) -> double * {
            const bool in_lds = use_ring && poff[xfirst + width] - poff[xfirst] <= ring_cap;
            return in_lds ? lring + (row % 3) * (long long) ring_cap * 3 : gring + (row % 3) * grow * 3;
        };
        double *Be = row_ptr(e, x0, re.width);
        sa_row_t r1 = {0, 0, 0}, r2 = {0, 0, 0};
        long long x01 = 0, x02 = 0;
        const double *B1 = nullptr, *B2 = nullptr;
        if (e + 1 <= start) {
            r1 = rows[e + 1];
            x01 = (e + 1 + r1.xmyL) >> 1;
            B1 = row_ptr(e + 1, x01, r1.width);
        }
        if (e + 2 <= start) {
            r2 = rows[e + 2];
            x02 = (e + 2 + r2.xmyL) >> 1;
            B2 = row_ptr(e + 2, x02, r2.width);
        }
        // one lane per cell-path of the diagonal
        const int g0 = poff[x0];
        const int rowpaths = poff[x0 + re.width] - g0;
        for (int j = tid; j < rowpaths; j += nthr) {
            const int g = g0 + j;
            const long long x = px[g];
            const int q = g - poff[x];
            const long long xmy = 2 * x - e, y = e - x;
            double *cur = Be + 3 * j;
            if (e == start) {
                cur[0] = end_m; cur[1] = end_x; cur[2] = end_y;
                continue;
            }
            long long i_mid = xmy - r2.xmyL, i_up = xmy - 1 - r1.xmyL, i_lo = xmy + 1 - r1.xmyL;
            bool has_mid = B2 && i_mid >= 0 && (i_mid >> 1) < r2.width && x + 1 <= R->lX;
            bool has_up = B1 && i_up >= 0 && (i_up >> 1) < r1.width;                    // cell (x, y+1)
            bool has_lo = B1 && i_lo >= 0 && (i_lo >> 1) < r1.width && x + 1 <= R->lX;  // cell (x+1, y)
            const double *cm = has_mid ? B2 + 3 * (poff[x + 1] - poff[x02]) : nullptr;
            const double *cu = has_up ? B1 + 3 * (poff[x] - poff[x01]) : nullptr;
            const double *cl = has_lo ? B1 + 3 * (poff[x + 1] - poff[x01]) : nullptr;
            int nn = (x + 1 <= R->lX) ? poff[x + 2] - poff[x + 1] : 0;
            const int *idn = (x + 1 <= R->lX) ? pid + poff[x + 1] : nullptr;
            double e_next = (y < R->lY) ? ev[y] : NEG_INF;  // event of matrix row y+1
            {
                int idq = pid[g];
                double tm = NEG_INF, tx = NEG_INF, ty = NEG_INF;
                if (has_mid)
                    for (int p = 0; p < nn; p++)
                        if (legal_any<RELAX>(m, inv_pow, inv_alpha, idq, idn[p])) {
                            double eP, eU;
                            if (RELAX) emit_gauss(xc4[poff[x + 1] + p], e_next, eP, eU);
                            else eP = emit_ref(m, rp, idn[p], e_next, 1, y);
                            double c = cm[3 * p + 0];
                            tm = la_any<RELAX>(LT, tm, c + (eP + m.t_mm));
                            tx = la_any<RELAX>(LT, tx, c + (eP + m.t_xm));
                            ty = la_any<RELAX>(LT, ty, c + (eP + m.t_ym));
                        }
                if (has_up) {
                    double eP, eU;
                    if (RELAX) emit_gauss(xc4[g], e_next, eU, eP);
                    else eP = emit_ref(m, rp, idq, e_next, 0, y);
                    double c = cu[3 * q + 2];
                    tm = la_any<RELAX>(LT, tm, c + (eP + m.t_my));
                    ty = la_any<RELAX>(LT, ty, c + (eP + m.t_yy));
                }
                if (has_lo)
                    for (int p = 0; p < nn; p++)
                        if (legal_any<RELAX>(m, inv_pow, inv_alpha, idq, idn[p])) {
                            double eP = (m.hdp || idn[p] >= 0) ? SA_LOG_GAPX : NEG_INF;
                            double c = cl[3 * p + 1];
                            tm = la_any<RELAX>(LT, tm, c + (eP + m.t_mx));
                            tx = la_any<RELAX>(LT, tx, c + (eP + m.t_xx));
                        }
                cur[0] = tm; cur[1] = tx; cur[2] = ty;
            }
        }
        __syncthreads();
        if (e > from || !first_wave) continue;
        // ---- checkpoint: per-cell terms of diagonalCalculationTotalProbability (impl/pairwiseAligner.c:1335-1353)
        if ((from - e) % SA_CKPT_EVERY == 0) {
            if (EXPECT && e != from) expect_flush(P, S->ck_base + (from - e) / SA_CKPT_EVERY - 1, Mc, acc, lane);
            const sa_ck_t ck = P.cks[S->ck_base + (from - e) / SA_CKPT_EVERY];
            double mx = NEG_INF;
            for (int i = lane; i < re.width; i += 64) {
                long long x = x0 + i;
                int np = poff[x + 1] - poff[x];
                const double *cf = F + 3 * (re.foff + poff[x] - poff[x0]);
                const double *cb = Be + 3 * (poff[x] - poff[x0]);
                double cell = NEG_INF;
                for (int q = 0; q < np; q++) {
                    double t = cf[3 * q] + cb[3 * q];
                    t = la_any<RELAX>(LT, t, cf[3 * q + 1] + cb[3 * q + 1]);
                    t = la_any<RELAX>(LT, t, cf[3 * q + 2] + cb[3 * q + 2]);
                    cell = la_any<RELAX>(LT, cell, t);
                }
                P.vbuf[ck.voff + i] = cell;
                mx = cell > mx ? cell : mx;
            }
            if (ck.nB > 0) {  // match-only forward step into diagonal e+1 == F[e+1].match (same arithmetic, same band)
                for (int i = lane; i < r1.width; i += 64) {
                    long long x = x01 + i;
                    int np = poff[x + 1] - poff[x];
                    const double *cf = F + 3 * (r1.foff + poff[x] - poff[x01]);
                    const double *cb = B1 + 3 * (poff[x] - poff[x01]);
                    double cell = NEG_INF;
                    for (int q = 0; q < np; q++) cell = la_any<RELAX>(LT, cell, cf[3 * q] + cb[3 * q]);
                    P.vbuf[ck.voff + ck.nA + i] = cell;
                    mx = cell > mx ? cell : mx;
                }
            }
            Mc = wave_max(mx);
        }
        if (!EXPECT) {
        // ---- posterior candidates of this diagonal (impl/pairwiseAligner.c:1355-1421); total >= Mc
        int nchunks = (re.width + 63) >> 6;
        for (int c = 0; c < nchunks; c++) {
            int i = c * 64 + lane;
            bool in = i < re.width;
            long long x = x0 + (in ? i : 0), y = e - x;
            int np = (in && x > 0 && y > 0) ? poff[x + 1] - poff[x] : 0;
            const double *cf = F + 3 * (re.foff + poff[x] - poff[x0]);
            const double *cb = Be + 3 * (poff[x] - poff[x0]);
            const double lim = Mc + P.log_thr - SA_CAND_EPS;
            int mine = 0;
            if (Mc > NEG_INF)
                for (int q = 0; q < np; q++) mine += (cf[3 * q] + cb[3 * q] >= lim) ? 1 : 0;
            // exclusive prefix over lanes: candidates are laid out cell by cell, path by path
            int incl = mine;
            for (int off = 1; off < 64; off <<= 1) {
                int o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            int total = __shfl(incl, 63, 64);
            int pos = count + incl - mine;
            if (mine > 0)
                for (int q = 0; q < np; q++) {
                    double fb = cf[3 * q] + cb[3 * q];
                    if (fb >= lim) {
                        if (pos < S->cand_cap) {
                            sa_cand_t cd;
                            cd.x = (int) (x - 1); cd.y = (int) (y - 1); cd.path = q; cd.pad = 0; cd.fb = fb;
                            P.cands[S->cand_off + pos] = cd;
                        } else {
                            P.overflow[0] = 1;
                        }
                        pos++;
                    }
                }
            count += total;
        }
        } else {
        // ---- EXPECT: diagonalCalculation_Expectations (impl/pairwiseAligner.c:1423-1443): the cell calculation with
        // current = backward diagonal e, lower/upper = forward diagonal e-1, middle = forward diagonal e-2, every
        // transition adding exp(F[from] + B[to] + (eP + tP) - total) (cell_signal_updateExpectations :914-944).
        // Forward diagonal e-2 has already been deleted for the first diagonal of a traceback (:1563-1578).
        {
            const sa_row_t rm1 = rows[e - 1];
            const bool have2 = e - 2 >= to && e - 2 >= 0;
            sa_row_t rm2 = {0, 0, 0};
            if (have2) rm2 = rows[e - 2];
            const long long x0m1 = (e - 1 + rm1.xmyL) >> 1, x0m2 = have2 ? (e - 2 + rm2.xmyL) >> 1 : 0;
            const double lim = Mc + P.log_thr - SA_CAND_EPS;
            const bool live = Mc > NEG_INF;
            int nchunks = (re.width + 63) >> 6;
            for (int c = 0; c < nchunks; c++) {
                int i = c * 64 + lane;
                bool in = live && i < re.width;
                long long xmy = (long long) re.xmyL + 2 * (in ? i : 0);
                long long x = x0 + (in ? i : 0), y = e - x;
                int np = in ? poff[x + 1] - poff[x] : 0;
                const int *idc = pid + poff[x];
                const double *cb = Be + 3 * (poff[x] - poff[x0]);
                long long il = xmy - 1 - rm1.xmyL, iu = xmy + 1 - rm1.xmyL, im = xmy - rm2.xmyL;
                bool has_lo = in && il >= 0 && (il >> 1) < rm1.width && x >= 1;
                bool has_up = in && iu >= 0 && (iu >> 1) < rm1.width && y >= 1;
                bool has_mid = in && have2 && im >= 0 && (im >> 1) < rm2.width && x >= 1 && y >= 1;
                int nl = (has_lo || has_mid) ? poff[x] - poff[x - 1] : 0;
                const int *idl = pid + poff[x >= 1 ? x - 1 : 0];
                const double *fl = has_lo ? F + 3 * (rm1.foff + poff[x - 1] - poff[x0m1]) : nullptr;
                const double *fm = has_mid ? F + 3 * (rm2.foff + poff[x - 1] - poff[x0m2]) : nullptr;
                const double *fu = has_up ? F + 3 * (rm1.foff + poff[x] - poff[x0m1]) : nullptr;
                double e_cur = (y >= 1) ? ev[y - 1] : NEG_INF;  // NULLEVENT for y == 0 (impl/pairwiseAligner.c:509-512)
                int mine = 0;
                for (int p = 0; p < np; p++) {
                    int idp = idc[p];
                    if (has_lo) {
                        double eP = (m.hdp || idp >= 0) ? SA_LOG_GAPX : NEG_INF;
                        for (int q = 0; q < nl; q++)
                            if (legal_step(m, idl[q], idp)) {
                                acc[0] += exp(fl[3 * q + 0] + cb[3 * p + 1] + (eP + m.t_mx) - Mc);
                                acc[1] += exp(fl[3 * q + 1] + cb[3 * p + 1] + (eP + m.t_xx) - Mc);
                            }
                    }
                    if (has_mid) {
                        double eP = emit_ref(m, rp, idp, e_cur, 1, y - 1);
                        for (int q = 0; q < nl; q++)
                            if (legal_step(m, idl[q], idp)) {
                                double v2 = fm[3 * q + 0] + cb[3 * p + 0] + (eP + m.t_mm);
                                double v3 = fm[3 * q + 1] + cb[3 * p + 0] + (eP + m.t_xm);
                                double v4 = fm[3 * q + 2] + cb[3 * p + 0] + (eP + m.t_ym);
                                acc[2] += exp(v2 - Mc);
                                acc[3] += exp(v3 - Mc);
                                acc[4] += exp(v4 - Mc);
                                if (m.hdp) mine += (v2 >= lim ? 1 : 0) + (v3 >= lim ? 1 : 0) + (v4 >= lim ? 1 : 0);
                            }
                    }
                    if (has_up) {
                        double eP = emit_ref(m, rp, idp, e_cur, 0, y - 1);
                        acc[5] += exp(fu[3 * p + 0] + cb[3 * p + 2] + (eP + m.t_my) - Mc);
                        acc[6] += exp(fu[3 * p + 2] + cb[3 * p + 2] + (eP + m.t_yy) - Mc);
                    }
                }
                if (!m.hdp) continue;
                // assignment candidates (cell_signal_updateExpectationsAndAssignments :946-968), reference order
                int incl = mine;
                for (int off = 1; off < 64; off <<= 1) {
                    int o = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += o;
                }
                int total = __shfl(incl, 63, 64);
                int pos = count + incl - mine;
                if (mine > 0)
                    for (int p = 0; p < np; p++) {
                        int idp = idc[p];
                        double eP = emit_ref(m, rp, idp, e_cur, 1, y - 1);
                        for (int q = 0; q < nl; q++)
                            if (legal_step(m, idl[q], idp)) {
                                double v[3] = {fm[3 * q + 0] + cb[3 * p + 0] + (eP + m.t_mm),
                                               fm[3 * q + 1] + cb[3 * p + 0] + (eP + m.t_xm),
                                               fm[3 * q + 2] + cb[3 * p + 0] + (eP + m.t_ym)};
                                for (int t = 0; t < 3; t++)
                                    if (v[t] >= lim) {
                                        if (pos < S->cand_cap) {
                                            sa_cand_t cd;
                                            cd.x = (int) (x - 1); cd.y = (int) (y - 1); cd.path = p; cd.pad = t; cd.fb = v[t];
                                            P.cands[S->cand_off + pos] = cd;
                                        } else {
                                            P.overflow[0] = 1;
                                        }
                                        pos++;
                                    }
                            }
                    }
                count += total;
            }
        }
        }
    }
    if (EXPECT && S->n_ck > 0 && first_wave) expect_flush(P, S->ck_base + S->n_ck - 1, Mc, acc, lane);
    if (tid == 0) P.cand_count[seg] = count < S->cand_cap ? count : S->cand_cap;
}

// ---------------------------------------------------------------------------------------------------
// fold: totalProbability of every checkpoint, folded exactly as dpDiagonal_dotProduct does
// (impl/pairwiseAligner.c:1167-1180): a left fold over the cells in ascending x-y.
// ---------------------------------------------------------------------------------------------------
// branch-free form of la_exact (same comparisons, same un-contracted polynomial), coefficients from LDS
__device__ __forceinline__ double la_exact_bf(const double *tab, double x, double y) {
    double mx = __builtin_fmax(x, y);
    double mn = __builtin_fmin(x, y);
    double d = mx - mn;
    int idx = (d > 1.0 ? 1 : 0) + (d > 2.5 ? 1 : 0) + (d > 4.5 ? 1 : 0);
    const double4 c = *reinterpret_cast<const double4 *>(tab + 4 * idx);
    double r = ((c.x * d + c.y) * d + c.z) * d + c.w;
    r = r + mn;
    return (d < 7.5) ? r : mx;
}

// One wave folds 64 consecutive checkpoints, one per lane.  A checkpoint's terms are contiguous in vbuf -- nA terms of its own
// diagonal, then nB of the diagonal above -- and its total is logAdd(fold(A), fold(B)): the two folds are independent chains, walked
// side by side (round 4: two logAdds in flight per lane instead of one; the chain is latency, ~25 dependent instructions and an LDS
// read per term).  Terms come in with coalesced loads, FOLD_TW per checkpoint and chain (a load instruction serves 64 / FOLD_TW
// checkpoints), and are transposed through LDS so that every lane then walks its own checkpoint.  Two tiles of 64 x (FOLD_TW + 1)
// doubles: 9 KB per wave at FOLD_TW 8, seventeen waves per CU (FOLD_TW 4 / 8 / 16 / 32: 0.88 / 0.57 / 0.60 / 1.14 ms on the headline batch, 2.04 / 1.31 / 1.58 / 2.99 on the realistic one) (the 64-term tile of rounds 1-3 took 33 KB: four waves per CU, one
// chain each -- k_fold 0.84 ms of the headline batch's 9.9 and 2.4 of the realistic batch's 29).
#define FOLD_TW 8
#define FOLD_LD (FOLD_TW + 1)
__global__ __launch_bounds__(64) void k_fold(DevPlan P, long long ck0, long long ck1) {
    __shared__ double tileA[64 * FOLD_LD], tileB[64 * FOLD_LD];
    __shared__ __attribute__((aligned(32))) double LT[16];
    const int lane = threadIdx.x;
    if (lane < 4) {
        const float a3[4] = {-0.009350833524763f, -0.014532321752540f, -0.004605031767994f, -0.000458661602210f};
        const float a2[4] = {0.130659527668286f, 0.139942324101744f, 0.063427417320019f, 0.009695946122598f};
        const float a1[4] = {0.498799810682272f, 0.495635523139337f, 0.695956496475118f, 0.930734667215156f};
        const float a0[4] = {0.693203116424741f, 0.692140569840976f, 0.514272634594009f, 0.168037164329057f};
        LT[4 * lane + 0] = (double) a3[lane]; LT[4 * lane + 1] = (double) a2[lane];
        LT[4 * lane + 2] = (double) a1[lane]; LT[4 * lane + 3] = (double) a0[lane];
    }
    const long long ckid = ck0 + (long long) blockIdx.x * 64 + lane;
    sa_ck_t ck = {0, 0, 0};
    if (ckid < ck1) ck = P.cks[ckid];
    const int nA = ck.nA, nB = ck.nB;
    const int maxlen = wave_max_i(nA > nB ? nA : nB);
    const int vo_lo = (int) (ck.voff & 0xffffffffll), vo_hi = (int) (ck.voff >> 32);
    double tA = NEG_INF, tB = NEG_INF;
    constexpr int CPL = 64 / FOLD_TW;              // checkpoints per load instruction
    const int sub = lane / FOLD_TW, t = lane % FOLD_TW;
    __syncthreads();
    for (int j0 = 0; j0 < maxlen; j0 += FOLD_TW) {
#pragma unroll 4
        for (int c0 = 0; c0 < 64; c0 += CPL) {
            const int c = c0 + sub;                // this lane's checkpoint of the load
            const int cnA = __shfl(nA, c), cnB = __shfl(nB, c);
            const long long vo = ((long long) __shfl(vo_hi, c) << 32) | (unsigned int) __shfl(vo_lo, c);
            const int j = j0 + t;
            double va = NEG_INF, vb = NEG_INF;     // -inf past the end: logAdd(t, -inf) == t
            if (j < cnA) va = P.vbuf[vo + j];
            if (j < cnB) vb = P.vbuf[vo + cnA + j];
            tileA[c * FOLD_LD + t] = va;
            tileB[c * FOLD_LD + t] = vb;
        }
        __syncthreads();
        const int lim = maxlen - j0 < FOLD_TW ? maxlen - j0 : FOLD_TW;
        for (int i = 0; i < lim; i++) {
            tA = la_exact_bf(LT, tA, tileA[lane * FOLD_LD + i]);
            tB = la_exact_bf(LT, tB, tileB[lane * FOLD_LD + i]);
        }
        __syncthreads();
    }
    if (ckid < ck1) P.totals[ckid] = (nB > 0) ? la_exact_bf(LT, tA, tB) : tA;
}

// ---------------------------------------------------------------------------------------------------
// finalize: posterior, threshold, floor; count survivors per segment
// ---------------------------------------------------------------------------------------------------
// The speculative total of every traceback segment of the ring / strip kernels (sa_strip.inc, "the speculative total of a
// traceback"): log-sum-exp over the cell-paths and states of the segment's first diagonal of forward state + end state, from the
// three planes the forward sweeps of these kernels keep on such diagonals.  One wave per segment, between the two sweeps of a pass;
// segments of other kernel families keep their NaN.
__global__ __launch_bounds__(64) void k_spec_match(DevPlan P, int seg0, int n_segs, double *__restrict__ spec) {
    if ((int) blockIdx.x >= n_segs) return;
    const int seg = seg0 + blockIdx.x;
    const sa_seg_t *S = &P.segs[seg];
    const sa_region_t *R = &P.regions[S->region];
    if (R->kind != SA_KIND_RING && R->kind != SA_KIND_FAST) return;
    const sa_row_t *rows = P.rows + R->row_off;
    const long long start = S->start;
    const long long o0 = rows[start].foff & 0xffffffffll, o1 = rows[start + 1].foff & 0xffffffffll;   // (g0 << 32 | offset; row N + 1 closes)
    const long long C = R->f_cellpaths;
    const double *Fm = P.F + 3 * R->f_base + o0;   // planes [match | gapX | gapY] of C cell-paths each
    const int np = (int) (o1 - o0), lane = threadIdx.x;
    const bool ragged_end = S->at_end && R->ragged_r;   // endStateProb / raggedEndStateProb (impl/stateMachine.c:1145-1173)
    const double em = ragged_end ? (P.m.t_mx + P.m.t_my) / 2.0 : P.m.t_mm, ex = ragged_end ? P.m.t_xx : P.m.t_xm,
                 ey = ragged_end ? P.m.t_yy : P.m.t_ym;
    double mx = NEG_INF;
    for (int j = lane; j < np; j += 64) {
        const double a = Fm[j] + em, b_ = Fm[C + j] + ex, c = Fm[2 * C + j] + ey;
        const double v = a > b_ ? (a > c ? a : c) : (b_ > c ? b_ : c);
        mx = v > mx ? v : mx;
    }
    mx = wave_max(mx);
    double sum = 0.0;
    if (mx > NEG_INF)
        for (int j = lane; j < np; j += 64) sum += exp(Fm[j] + em - mx) + exp(Fm[C + j] + ex - mx) + exp(Fm[2 * C + j] + ey - mx);
    sum = wave_sum(sum);
    if (lane == 0) {
        double r = (mx > NEG_INF && sum > 0.0) ? mx + log(sum) : NEG_INF;
        // NaN means "a segment of another kernel family" to the kernels that read this array: a NaN that comes out of the DATA (an
        // event mean or a model entry that is not a number poisons the forward values) must not pass for that -- the traceback
        // would return nothing without a word.  It is reported instead (sa_batch_run: SA_EINVAL).
        if (!(sum == sum) || !(mx == mx)) { r = NEG_INF; P.overflow[2] = 1; }
        spec[seg] = r;
    }
}

// An event mean that is not a finite number: the reference's logAdd turns such a cell's NaN into NaN everywhere (every comparison
// with it is false), the kernels' max/min drop it silently -- the read would come back with an alignment that steps around the
// event, or with none.  One coalesced pass over the batch's event means per run (80 MB per 2000 x 5000-event reads: ~0.03 ms)
// raises P.overflow[2] instead, whatever path brought the events here (packed by the host, gathered from the caller's block).
__global__ __launch_bounds__(256) void k_check_events(const double *__restrict__ ev, long long n, int *flag) {
    bool bad = false;
    for (long long i = (long long) blockIdx.x * 256 + threadIdx.x; i < n; i += (long long) gridDim.x * 256) {
        const unsigned hi = (unsigned) __double2hiint(ev[i]);
        bad = bad || ((hi >> 20) & 0x7ffu) == 0x7ffu;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) flag[2] = 1;
}

// spec (one-pass strip sweep, sa_strip.inc): per segment the speculative total its candidate bound was derived from, NaN for every
// other segment.  The bound is only valid while no exact total of the segment lies below spec - slack: checked here, raised in
// P.overflow[1] (the pass is then repeated with the two-pass sweep).
// vc_bits (SA_FLAG_VC_ROWS): one bit per reference position of every job (job j's from bit vc_off[j] on): set where the k-mer that
// starts there holds the ambiguity letter 'X' -- the rows writePosteriorProbsVC prints (impl/signalMachine.c:161-232).  Pairs
// elsewhere are counted and summed into seg_all (their number and the sum of their floor(p 1e7), what
// scoreByPosteriorProbabilityIgnoringGaps needs) and dropped here, on the device.
__global__ __launch_bounds__(64) void k_finalize(DevPlan P, int seg0, int n_segs, long long *prob_e7, int *seg_pass,
                                                 const double *__restrict__ spec, double spec_slack,
                                                 const unsigned long long *__restrict__ vc_bits, const long long *__restrict__ vc_off,
                                                 long long *__restrict__ seg_all) {
    if ((int) blockIdx.x >= n_segs) return;
    int seg = seg0 + blockIdx.x;
    const sa_seg_t *S = &P.segs[seg];
    int n = P.cand_count[seg];
    int lane = threadIdx.x;
    int cnt = 0;
    long long all_n = 0, all_sum = 0;
    const sa_region_t *Rv = &P.regions[S->region];
    const long long vc_base = vc_bits ? vc_off[Rv->job] : 0;
    if (spec) {
        const double sp = spec[seg];
        if (sp == sp && sp > NEG_INF) {
            bool bad = false;
            for (int c = lane; c < S->n_ck; c += 64) bad = bad || (P.totals[S->ck_base + c] < sp - spec_slack + 1e-9);
            if (__ballot(bad) && lane == 0) P.overflow[1] = 1;
        }
    }
    for (int i = lane; i < ((n + 63) & ~63); i += 64) {
        bool pass = false;
        if (i < n) {
            sa_cand_t c = P.cands[S->cand_off + i];
            long long e = (long long) c.x + c.y + 2;
            double total = P.totals[S->ck_base + (S->from - e) / SA_CKPT_EVERY];
            double p = exp(c.fb - total);
            long long v = -1;
            if (p >= P.threshold) {
                if (p > 1.0) p = 1.0;
                v = (long long) floor(p * SA_PROB_1);
                pass = true;
            }
            if (vc_bits && pass) {
                all_n++; all_sum += v;
                const long long bit = vc_base + (long long) c.x + Rv->x1;
                if (!((vc_bits[bit >> 6] >> (bit & 63)) & 1ull)) { pass = false; v = -1; }
            }
            prob_e7[S->cand_off + i] = v;
        }
        cnt += __popcll(__ballot(pass));
    }
    if (lane == 0) seg_pass[seg] = cnt;
    if (vc_bits) {
        for (int off = 32; off > 0; off >>= 1) { all_n += __shfl_xor(all_n, off, 64); all_sum += __shfl_xor(all_sum, off, 64); }
        if (lane == 0) { seg_all[2ll * seg] = all_n; seg_all[2ll * seg + 1] = all_sum; }
    }
}

// Expectation pass: the per-read sums on the device.  Every checkpoint group holds its seven transition sums scaled by its
// maximum (gsum / gmc) and its exact total (k_fold); a read's expectations are sum_groups gsum * exp(gmc - total), its
// likelihood the totals once per diagonal (hmm->likelihood += totalProbability, impl/pairwiseAligner.c:1432).  One wave per
// region, a lane per checkpoint group; 8 doubles per read come back instead of 80 bytes per group (130 MB per 2000 reads).
// Bit-reproducible from run to run: a read's regions (consecutive in the plan) are summed by ONE wave in region order -- the
// wave of the read's first region; the others return -- with a fixed lane assignment and a fixed butterfly, no atomics.  What is
// NOT the reference's order of additions: it adds cell by cell and the likelihood once per diagonal (:1432) where this adds
// total * rows; transition expectations agree with the restatement to 1e-9 relative and the likelihood to 1e-12
// (tests/test_gpu_expectations.py) -- that tolerance, not bit equality, is the parity statement of this entry point.
__global__ __launch_bounds__(64) void k_expect_reduce(DevPlan P, double *__restrict__ red, int n_regions) {
    const int r0 = (int) blockIdx.x;
    const int job = P.regions[r0].job;
    if (r0 > 0 && P.regions[r0 - 1].job == job) return;
    const int lane = threadIdx.x;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = r0; r < n_regions && P.regions[r].job == job; r++) {
        const sa_region_t *R = &P.regions[r];
        for (long long sg = R->seg_off; sg < R->seg_off + R->n_seg; sg++) {
            const sa_seg_t *S = &P.segs[sg];
            const long long nrows = S->from - S->to;
            for (int c = lane; c < S->n_ck; c += 64) {
                const double total = P.totals[S->ck_base + c];
                long long rows_here = nrows - (long long) c * SA_CKPT_EVERY;
                if (rows_here > SA_CKPT_EVERY) rows_here = SA_CKPT_EVERY;
                if (rows_here > 0) acc[7] += total * (double) rows_here;
                if (!(total > NEG_INF)) continue;
                const double sc = exp(P.gmc[S->ck_base + c] - total);
                for (int k = 0; k < 7; k++) acc[k] += P.gsum[8 * (S->ck_base + c) + k] * sc;
            }
        }
    }
    for (int k = 0; k < 8; k++) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[8ll * job + k] = v;
    }
}

// exclusive scan of seg_pass (single block)
// out_host (pinned host memory, written straight from the kernel) spares a copy-engine transfer: a queued copy that
// waits for a kernel blocks every later copy on the engine, including the pair copies of groups already finished
__global__ __launch_bounds__(1024) void k_scan(const int *in, long long *out, long long *out_host, int n) {
    __shared__ long long part[1024];
    int t = threadIdx.x;
    int per = (n + 1023) / 1024;
    int lo = t * per, hi = lo + per < n ? lo + per : n;
    long long s = 0;
    for (int i = lo; i < hi; i++) s += in[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        long long v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    long long base = t ? part[t - 1] : 0;
    for (int i = lo; i < hi; i++) {
        out[i] = base;
        out_host[i] = base;
        base += in[i];
    }
    if (t == 1023) { out[n] = part[1023]; out_host[n] = part[1023]; }
}

// gather survivors of a segment in REVERSE candidate order (=> ascending diagonals, x descending, path descending:
// the order of stList_pop + stable sort by x+y, impl/pairwiseAligner.c:2043-2050, impl/signalMachine.c:872)
// seg_off: exclusive scan over the n_segs segments starting at seg0 (indexed from 0); out: first slot of that range
// strip segment: a one-path ring-kernel region that the strip kernels sweep (the host's strip_region(), sa_hip.hip)
__device__ __forceinline__ bool seg_is_strip(const sa_region_t *R, int strip_on) {
    return strip_on && R->kind == SA_KIND_RING && R->max_p == 1 && R->lX < 64ll * STRIP_NS_MAX && R->N >= 1;
}
// spec: per segment its speculative total where the ring / strip kernels produced the candidates (NaN elsewhere), or nullptr.
//   * a strip segment's candidates arrive strip by strip: k_gather_sorted writes it;
//   * a ring segment's candidates arrive diagonal by diagonal (the workgroup's barrier separates diagonals) but, inside a
//     diagonal, in the order the waves got there: a survivor's place is the number of survivors ahead of it in the list, minus
//     those of its own diagonal among them, plus those of its own diagonal with a smaller (column, path) -- its neighbours in the
//     list, a handful.
// a result record at slot `pos` of a group's range: 16 bytes, or (SA_FLAG_PAIRS8: one path per cell, coordinates below 2^20) 8
__device__ __forceinline__ void put_pair(sa_pair16_t *out, long long pos, int p8, long long pe, int x, int y, int path, int kmer) {
    if (p8) reinterpret_cast<unsigned long long *>(out)[pos] = sa_pair8_pack(pe, x, y);
    else out[pos] = sa_pair16_pack(pe, x, y, path, kmer);
}
__global__ __launch_bounds__(64) void k_gather(DevPlan P, int seg0, int n_segs, const long long *prob_e7,
                                               const long long *seg_off, sa_pair16_t *out, const double *__restrict__ spec, int strip_on,
                                               int p8) {
    if ((int) blockIdx.x >= n_segs) return;
    const int lseg = blockIdx.x;
    int seg = seg0 + lseg;
    const sa_seg_t *S = &P.segs[seg];
    const sa_region_t *R = &P.regions[S->region];
    const int *poff = P.poff + R->poff_off;
    const int *pid = P.pid + R->pid_off;
    int n = P.cand_count[seg];
    int lane = threadIdx.x;
    long long total = seg_off[lseg + 1] - seg_off[lseg];
    long long done = 0;
    bool unordered = false;
    if (spec) {
        const double sp = spec[seg];
        if (sp == sp) {
            if (seg_is_strip(R, strip_on)) return;
            unordered = R->kind == SA_KIND_RING;   // (a register-kernel segment is one wave: its candidates are in order)
        }
    }
    const sa_cand_t *cd = P.cands + S->cand_off;
    const long long *pe = prob_e7 + S->cand_off;
    for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        bool pass = i < n && pe[i] >= 0;
        unsigned long long mask = __ballot(pass);
        int rank = __popcll(mask & ((1ull << lane) - 1ull));
        if (pass) {
            sa_cand_t c = cd[i];
            long long k = done + rank;              // index in candidate order
            if (unordered) {
                const int e_i = c.x + c.y;
                const long long key_i = ((long long) c.x << 20) | c.path;
                int same_before = 0, less = 0;
                for (int j = i - 1; j >= 0; j--) {
                    const sa_cand_t q = cd[j];
                    if (q.x + q.y != e_i) break;
                    if (pe[j] >= 0) { same_before++; less += (((long long) q.x << 20) | q.path) < key_i ? 1 : 0; }
                }
                for (int j = i + 1; j < n; j++) {
                    const sa_cand_t q = cd[j];
                    if (q.x + q.y != e_i) break;
                    if (pe[j] >= 0) less += (((long long) q.x << 20) | q.path) < key_i ? 1 : 0;
                }
                k += less - same_before;
            }
            long long pos = seg_off[lseg] + (total - 1 - k);
            put_pair(out, pos, p8, pe[i], (int) (c.x + R->x1), (int) (c.y + R->y1), c.path, pid[poff[c.x + 1] + c.path]);
        }
        done += __popcll(mask);
    }
}


// The same for the segments of the ring kernels and of the one-pass strip sweep (sa_ring.inc, sa_strip.inc), whose candidates are
// appended in the order the waves / strips get to them instead of in candidate order (diagonals downwards, columns upwards, a
// cell's paths upwards): the survivors are put in candidate order first -- a counting sort by diagonal (histogram of the
// segment's diagonals in LDS, GATHER_H at a time), then every diagonal's few survivors by (column, path) -- and written as k_gather
// writes them.  A survivor's key: diagonals below the start << 40 | column << 12 | path (28 and 12 bits: the planners' limits are
// 2^28 columns and 255 paths per cell on these kernels); its candidate slot travels beside the key.  keys / idx: 12 bytes of
// scratch per candidate slot.  The result does not depend on the order the candidates arrived in.
#define GATHER_H 1024   // (4 KB of LDS per wave: 8192 entries held a wave to four per CU and cost the realistic batch 0.97 ms)
__global__ __launch_bounds__(64) void k_gather_sorted(DevPlan P, int seg0, int n_segs, const long long *prob_e7, const long long *seg_off,
                                                      sa_pair16_t *out, const double *__restrict__ spec,
                                                      unsigned long long *keys_all, unsigned *idx_all, int p8) {
    __shared__ int H[GATHER_H + 64];
    if ((int) blockIdx.x >= n_segs) return;
    const int lseg = blockIdx.x, seg = seg0 + lseg;
    { const double sp = spec[seg]; if (!(sp == sp)) return; }   // not a segment of these kernels: k_gather wrote it
    const sa_seg_t *S = &P.segs[seg];
    const sa_region_t *R = &P.regions[S->region];
    if (!seg_is_strip(R, 1)) return;                            // a ring segment: k_gather wrote it
    const int *poff = P.poff + R->poff_off;
    const int *pid = P.pid + R->pid_off;
    const int n = P.cand_count[seg];
    const int lane = threadIdx.x;
    const long long total = seg_off[lseg + 1] - seg_off[lseg];
    if (total <= 0) return;
    volatile unsigned long long *keys = keys_all + S->cand_off;   // (written and read by different lanes: not through this CU's L1)
    volatile unsigned *idx = idx_all + S->cand_off;
    const long long start = S->start, span = S->start - S->to;   // diagonals below the start: 0 .. span - 1
    long long placed = 0;   // survivors on diagonals above the current range (all in place)
    for (long long r0 = 0; r0 < span && placed < total; r0 += GATHER_H) {
        const int hn = (int) (span - r0 < GATHER_H ? span - r0 : GATHER_H);
        for (int i = lane; i < hn + 1; i += 64) H[i] = 0;
        __syncthreads();
        for (int i = lane; i < n; i += 64) {
            if (prob_e7[S->cand_off + i] < 0) continue;
            const sa_cand_t c = P.cands[S->cand_off + i];
            const long long de = start - ((long long) c.x + c.y + 2);
            if (de >= r0 && de < r0 + hn) atomicAdd(&H[(int) (de - r0)], 1);
        }
        __syncthreads();
        // exclusive scan of H[0 .. hn) in place (a wave scan per 64 entries, carried), H[hn] = the range's count
        int carry = 0;
        for (int b0 = 0; b0 < hn; b0 += 64) {
            const int i = b0 + lane;
            const int v = i < hn ? H[i] : 0;
            int incl = v;
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            if (i < hn) H[i] = carry + incl - v;
            carry += __shfl(incl, 63, 64);
        }
        if (lane == 0) H[hn] = carry;
        __syncthreads();
        const int in_range = H[hn];
        if (in_range > 0) {
            // placement: any order inside a diagonal (sorted below); the cursor of diagonal i runs from H[i] up to the old H[i + 1]
            for (int i = lane; i < n; i += 64) {
                if (prob_e7[S->cand_off + i] < 0) continue;
                const sa_cand_t c = P.cands[S->cand_off + i];
                const long long de = start - ((long long) c.x + c.y + 2);
                if (de < r0 || de >= r0 + hn) continue;
                const int slot = atomicAdd(&H[(int) (de - r0)], 1);
                keys[placed + slot] = ((unsigned long long) de << 40) | ((unsigned long long) (unsigned) c.x << 12) | (unsigned long long) (unsigned) c.path;
                idx[placed + slot] = (unsigned) i;
            }
            __threadfence_block();
            __syncthreads();
            // H[i] is now the END of diagonal i's group (= the old start of i + 1): sort every group (a handful of entries)
            for (int i = lane; i < hn; i += 64) {
                const int ge = H[i], gs = i == 0 ? 0 : H[i - 1];
                for (int a = gs + 1; a < ge; a++) {
                    const unsigned long long k = keys[placed + a];
                    const unsigned ki = idx[placed + a];
                    int b = a - 1;
                    while (b >= gs && keys[placed + b] > k) { keys[placed + b + 1] = keys[placed + b]; idx[placed + b + 1] = idx[placed + b]; b--; }
                    keys[placed + b + 1] = k;
                    idx[placed + b + 1] = ki;
                }
            }
            __threadfence_block();
            __syncthreads();
        }
        placed += in_range;
    }
    // candidate order is ascending key order; written in reverse, as k_gather does
    for (long long k = lane; k < total; k += 64) {
        const unsigned i = idx[k];
        const sa_cand_t c = P.cands[S->cand_off + i];
        put_pair(out, seg_off[lseg] + (total - 1 - k), p8, prob_e7[S->cand_off + i], (int) (c.x + R->x1), (int) (c.y + R->y1), c.path,
                 pid[poff[c.x + 1] + c.path]);
    }
}

// Emission constants per (reference position, path) with the read's scale / shift / var folded in -- what fill_xc of the
// planner computes (sa_plan.c), here on the device: 32 bytes per path that the host neither has to write nor to upload.
// One block per region.  The per-k-mer logarithms come from tab6 (computed once per batch on the host with the C
// library's log), so the values are bit-identical to the host's.
__global__ __launch_bounds__(256) void k_fill_xc(const sa_region_t *__restrict__ regions, const int *__restrict__ poff_all,
                                                 const int *__restrict__ pid_all, const double *__restrict__ tab6,
                                                 const int *__restrict__ hdp_slot, long long hdp_grid_length, double4 *xc, int emission) {
    const sa_region_t *R = &regions[blockIdx.x];
    const int *poff = poff_all + R->poff_off;
    const int *pid = pid_all + R->pid_off;
    const long long n = poff[R->lX + 1];
    double4 *o = xc + R->pid_off;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const int id = pid[i];
        double4 v;
        if (hdp_slot) {   // e' = e/var - v.x; v.y = byte offset of the k-mer's {y, slope} row (or past the table: no density)
            const double mu = id >= 0 ? tab6[6ll * id] : 0.0;
            const int slot = id >= 0 ? hdp_slot[id] : -1;
            v.x = ((R->scale - R->var) * mu + R->shift) / R->var;
            v.y = slot >= 0 ? (double) ((long long) slot * hdp_grid_length * 16) : (double) SA_HDP_FAST_MAX_BYTES;
            v.z = 0.0; v.w = 0.0;
        } else if (id < 0) {   // NULL k-mer: both emissions are log(0); inv_s = 1 keeps (e - m) * inv_s finite
            v.x = 0.0; v.y = 1.0; v.z = NEG_INF; v.w = NEG_INF;
        } else {
            const double mu = tab6[6ll * id], sd = tab6[6ll * id + 1], c = tab6[6ll * id + 2], cy = tab6[6ll * id + 4];
            // (the two-distribution emissions carry no log(1 / var) -- impl/stateMachine.c:607-700 against :557-605 --, and the one on the
            // scaled model, emission 2, takes the event as it is: scale 1, shift 0, var 1)
            const double sc = emission == 2 ? 1.0 : R->scale, sh = emission == 2 ? 0.0 : R->shift, va = emission == 2 ? 1.0 : R->var;
            const double lv = emission != 0 ? 0.0 : R->lvar;
            v.x = sc * mu + sh;
            if (c == NEG_INF) {   // sd == 0: emissions_signal_logGaussPdf returns LOG_ZERO
                v.y = 1.0; v.z = NEG_INF; v.w = NEG_INF;
            } else {
                v.y = 1.0 / (va * sd);
                v.z = lv + c;
                v.w = lv + cy;
            }
        }
        o[i] = v;
    }
}

// ===================================================================================================
// host runtime
// ===================================================================================================
#define HIPCHK(call)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "[signalalign_hip] %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, \
                    __LINE__);                                                                 \
            return e_ == hipErrorOutOfMemory ? SA_ENOMEM : SA_ENODEVICE;                       \
        }                                                                                      \
    } while (0)

struct sa_launch_chunk {
    long long ids_gr, ids_fr;  // offsets into d_ids: memory-resident / register-kernel regions
    int ngr, nfr;
    long long ids_rr[16];      // ring-kernel regions by class: [multi * 8 + cap class], cap = 64 * (class + 1)
    int nrr[16];
    long long ids_st;          // one-path ring-kernel regions taken by the strip kernels (sa_strip.inc)
    int nst;
    int g0, g1;                // groups [g0, g1)
};
struct sa_launch_group {
    long long seg0, seg1, ck0, ck1;
    long long ids_gs, ids_fs;           // segments of memory-resident / register-kernel regions
    int ngs, nfs;
    long long ids_rs[16];               // segments of ring-kernel regions, by the class of their region
    int nrs[16];
    long long ids_ss;                   // segments of strip-kernel regions
    int nss;
    unsigned seam_first;                // their first wave slot in the seam storage
};

struct sa_batch {
    sa_plan_t *plan;
    int device;
    unsigned flags;
    hipStream_t stream;            // == cstream[0]
    hipStream_t cstream[2];        // compute streams; groups alternate between them
    hipStream_t xstream[2];        // two more, for the forward launches of the ring-kernel classes
    // device buffers
    sa_region_t *d_regions; sa_row_t *d_rows; int *d_pk; int *d_poff; int *d_pid; int *d_px; double *d_xc; double *d_ev;
    sa_prec_t *d_prec;
    size_t d_blk_bytes;
    char *d_blk;           // SA_FLAG_INPUTS_IN_HOST_BLOCK: the image of the caller's block (its event records are gathered on this
                           // batch's own stream, possibly after sa_batch_create has returned: kept until the batch goes)
    sa_seg_t *d_segs; sa_ck_t *d_cks;
    double *d_F; double *d_E; double *d_vbuf; sa_cand_t *d_cands; int *d_cand_count; int *d_overflow; double *d_totals;
    double *d_bscratch;
    double *d_gsum, *d_gmc;  // expectation mode only
    bool expect;
    bool relax;              // memory-resident kernels in their RELAX flavour
    int ring_cap;            // cell-paths per diagonal of their LDS ring (0: rows stay in global memory)
    int wide_cap;            // cells per row of the register kernels' LDS ring for wide diagonals (0: every diagonal fits)
    int gen_threads;         // 64, or 128 when a diagonal of a memory-resident region holds more than 64 cell-paths
    bool strip_on;           // one-path ring-kernel regions run on the strip kernels (default; SA_STRIP=0: ring kernels)
    double *d_ckxy;          // ... and the (two-pass) backward kernel's side buffer (2 x n_vbuf doubles)
    bool strip_one_pass;     // strip segments run the one-pass backward sweep (k_bwd_strip1; SA_STRIP_PASSES=2: the two-pass one)
    double *d_spec;          // ring / strip kernels: speculative totals, one per segment (NaN: a segment of another kernel family)
    double spec_slack;       // candidates: forward + backward >= spec - slack + log(threshold); grows when a pass has to be repeated
    int spec_repeats;        // passes repeated because of it (the second repeat drops the bound altogether)
    bool released;           // sa_batch_release_device: the working storage went back to the pool, the results stay
    unsigned long long *d_sortkey;   // k_gather_sorted's scratch: 8 + 4 bytes per candidate slot
    unsigned *d_sortidx;
    unsigned long long *d_vc_bits = nullptr;   // SA_FLAG_VC_ROWS: see k_finalize
    long long *d_vc_off = nullptr, *d_seg_all = nullptr;
    std::vector<unsigned long long> h_vc_bits;  // (the same on the host, for SA_FLAG_EXACT's host finalisation)
    std::vector<long long> h_vc_off, job_all_n, job_all_sum;
    bool plan_hdp = false;                      // the batch's model holds an HDP
    unsigned hdp_hot = 0xffffffffu;             // DevModel.hdp_hot
    char *d_seam;            // their seam storage: per wave two arrays of seam_cap records of 16 bytes
    unsigned seam_cap;
    unsigned seam_cap_bwd;   // records per seam array of the backward launches (a traceback segment is shorter than a region)
    long long seam_bwd_off;  // bytes: the forward launch's slots come first, then those of a pass's backward launches
    double *d_two = nullptr; long long two_xn_off = 0;   // two-distribution emission on the register kernels (DevPlan.two)
    double *d_tab6; double *d_noise3; double *d_evn; int *d_hdp_slot; double *d_hdp_y, *d_hdp_slope, *d_hdp_grid, *d_hdp_tab, *d_hdp_coef;
    long long *d_prob; int *d_seg_pass; long long *d_seg_off; sa_pair16_t *d_out;
    int *d_ids;  // region / segment id lists per launch
    long long cand_alloc;
    long long out_alloc;
    int cand_factor;               // candidate capacity relative to the planner's 2 per posterior diagonal (overflow re-runs)
    // launch lists (host): a chunk is one forward-storage pass; its traceback segments are cut into groups of
    // consecutive reads so that the result copy of one group overlaps the backward kernels of the next
    std::vector<sa_launch_chunk> chunks;
    std::vector<sa_launch_group> groups;
    std::vector<int> ids_flat;
    hipStream_t pair_stream;       // the pairs themselves
    std::vector<hipEvent_t> gev;   // per group: backward start, backward end, results ready, (unused)
    std::vector<hipEvent_t> cev;   // per chunk: forward start, forward end
    long long *h_seg_off;          // pinned: per group n+1 exclusive offsets
    int *h_overflow;               // pinned
    // results
    sa_pair16_t *h_pairs;    // pinned host copy of all pairs (packed, sa_internal.h), job after job
    bool p8 = false;         // SA_FLAG_PAIRS8: the records are 8 bytes (sa_pair8_t), in h_pairs and in d_out alike
    size_t rec() const { return p8 ? sizeof(sa_pair8_t) : sizeof(sa_pair16_t); }
    sa_pair16_t *out_at(long long slot) const { return reinterpret_cast<sa_pair16_t *>(reinterpret_cast<char *>(d_out) + rec() * (size_t) slot); }
    sa_pair16_t *host_at(long long slot) const { return reinterpret_cast<sa_pair16_t *>(reinterpret_cast<char *>(h_pairs) + rec() * (size_t) slot); }
    long long h_pairs_cap, n_pairs_total;
    std::vector<long long> job_off;
    std::vector<long long> job_dev_off;   // where a job's pairs start in d_out (device finalisation only)
    sa_pair16_t *d_pairs_up;              // host-finalised pairs uploaded for a downstream device step (sa_batch_mea)
    long long d_pairs_up_cap;
    bool ran;
    bool quiet;            // the last run returned SA_OK: it waited for everything it had queued, the batch's streams are idle
    bool dev_planned;      // the plan was built on the device (sa_dplan.inc): its big arrays exist in HBM only
    std::thread *runner;   // sa_batch_start .. sa_batch_wait
    int runner_rc;
    sa_batch_stats_t stats;
    hipEvent_t ev[8];
    // Creation in two halves (sa_batch_create_deferred): what the second half needs.  `pending` is the device plan whose kernels
    // are queued; the caller's arrays (c_jobs, c_ambig) are only touched again if that plan turns a read down and the host
    // planner takes over -- which is why a deferred batch asks the caller to keep them until its first run has returned.
    struct DPlanPending *pending;
    const sa_model_t *c_m;
    sa_params_t c_p;
    const sa_job_t *c_jobs;
    int64_t c_n;
    const char *const *c_ambig;
    long long c_budget;
    double c_t0;
    bool c_deferred;
    char *held_stage;      // (deferred batches: see dplan_back)
    bool finished;
    int prepare_rc;
    bool prepared;           // plan collected and launch lists built (sa_batch_prepare, or the first step of finishing)
    long long lw_strip_max_n, lw_strip_max_seg, lw_strip_fwd_slots, lw_strip_bwd_slots;   // from the launch lists: seam storage
    int finish_rc;
    std::mutex fin_mu;
};
static int batch_finish(sa_batch *b);

static size_t g_sa_pool_idle_bytes(int device);
int sa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sa_device_memory(int device, int64_t *free_bytes, int64_t *total_bytes) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SA_ENODEVICE;
    if (device < 0 || device >= n) return SA_EINVAL;
    HIPCHK(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t) (f + g_sa_pool_idle_bytes(device));   // what the caching allocator holds is available
    if (total_bytes) *total_bytes = (int64_t) t;
    return SA_OK;
}

static DevPlan make_devplan(const sa_batch *b) {
    const sa_plan_t *pl = b->plan;
    const sa_model_t *m = pl->model;
    DevPlan P;
    memset(&P, 0, sizeof(P));
    P.regions = b->d_regions; P.rows = b->d_rows; P.pk = b->d_pk; P.poff = b->d_poff; P.pid = b->d_pid; P.px = b->d_px; P.xc = b->d_xc; P.ev = b->d_ev;
    P.prec = b->d_prec;
    P.segs = b->d_segs; P.cks = b->d_cks; P.F = b->d_F; P.E = b->d_E; P.vbuf = b->d_vbuf; P.cands = b->d_cands;
    P.cand_count = b->d_cand_count; P.overflow = b->h_overflow; P.totals = b->d_totals; P.bscratch = b->d_bscratch;
    P.gsum = b->d_gsum; P.gmc = b->d_gmc;
    P.two = b->d_two; P.two_xn_off = b->two_xn_off;
    P.m.t_mm = m->t_mm; P.m.t_mx = m->t_mx; P.m.t_my = m->t_my; P.m.t_xm = m->t_xm; P.m.t_xx = m->t_xx;
    P.m.t_ym = m->t_ym; P.m.t_yy = m->t_yy;
    P.m.tab6 = b->d_tab6; P.m.emission = m->emission; P.m.noise3 = b->d_noise3; P.evn = b->d_evn; P.m.pow_km1 = m->pow_km1; P.m.n_alpha = m->n_alpha; P.m.hdp = m->hdp ? 1 : 0;
    P.m.hdp_slot = b->d_hdp_slot; P.m.hdp_y = b->d_hdp_y; P.m.hdp_slope = b->d_hdp_slope; P.m.hdp_grid = b->d_hdp_grid;
    P.m.grid_len = m->hdp ? (int) m->hdp->grid_length : 0;
    if (m->hdp) {
        const sa_hdp_t *h = m->hdp;
        P.m.hdp_tab = b->d_hdp_tab;
        P.m.hdp_coef = b->d_hdp_coef;
        P.m.hdp_g0 = h->grid[0];
        P.m.hdp_gN = h->grid[h->grid_length - 1];
        P.m.hdp_dx = h->grid[1] - h->grid[0];  // grid_spline_interp: dx = x[1] - x[0]
        P.m.hdp_tab_bytes = (unsigned) (h->n_slots * h->grid_length * 16);
        P.m.hdp_hot = b->hdp_hot;
    }
    P.threshold = pl->params.threshold;
    P.log_thr = log(pl->params.threshold);
    P.spec = b->d_spec;
    P.spec_slack = b->spec_slack;
    P.expect = b->expect ? 1 : 0;
    return P;
}

// Host -> device copies of the plan (several hundred MB per batch) through a persistent ring of pinned buffers: the
// runtime's own staging of pageable memory moves about 3 GB/s; here the CPU copy into a pinned slot (all host threads)
// overlaps the DMA of the previous slots.  One ring per process and device, calls serialise on it.
static hipError_t sa_sync_stream_fwd(hipStream_t s, int dev);   // sa_sync_stream (below)
struct SaUploader {
    std::mutex mu;
    int device = -1;
    static const int SLOTS = 4;
    static const size_t SLOT_BYTES = (size_t) 16 << 20;
    void *slot[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t done[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t stream = nullptr;
    int next = 0;
    int bind(int dev) {
        if (device == dev && stream) return SA_OK;
        // (a process normally drives one GPU; a change of device rebuilds the ring)
        for (int i = 0; i < SLOTS; i++) {
            if (slot[i]) (void) hipHostFree(slot[i]);
            if (done[i]) (void) hipEventDestroy(done[i]);
            slot[i] = nullptr; done[i] = nullptr;
        }
        if (stream) (void) hipStreamDestroy(stream);
        stream = nullptr;
        device = dev;
        // highest priority: the uploads and the device planner of the NEXT batch run while the current batch's sweeps fill the
        // chip; at normal priority their (short) kernels wait for wave slots behind thousands of long-running waves and
        // sa_batch_create takes 18 ms instead of 8
        int prio_lo = 0, prio_hi = 0;
        (void) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if (hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio_hi) != hipSuccess) return SA_ENODEVICE;
        for (int i = 0; i < SLOTS; i++) {
            if (hipHostMalloc(&slot[i], SLOT_BYTES, hipHostMallocDefault) != hipSuccess) return SA_ENOMEM;
            if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) return SA_ENODEVICE;
        }
        return SA_OK;
    }
    int copy_pinned(void *dst, const void *src, size_t bytes) {   // the source is pinned: plain DMA
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
        return SA_OK;
    }
    int copy(void *dst, const void *src, size_t bytes) {
        const char *s = (const char *) src;
        char *d = (char *) dst;
        while (bytes > 0) {
            const size_t n = bytes < SLOT_BYTES ? bytes : SLOT_BYTES;
            const int k = next;
            next = (next + 1) % SLOTS;
            HIPCHK(hipEventSynchronize(done[k]));            // the slot's previous DMA has left it
            char *buf = (char *) slot[k];
            const size_t piece = (size_t) 1 << 20;
            sa_parallel_for((n + piece - 1) / piece, [&](size_t q) {
                const size_t a = q * piece, len = a + piece < n ? piece : n - a;
                memcpy(buf + a, s + a, len);
            });
            HIPCHK(hipMemcpyAsync(d, buf, n, hipMemcpyHostToDevice, stream));
            HIPCHK(hipEventRecord(done[k], stream));
            s += n; d += n; bytes -= n;
        }
        return SA_OK;
    }
    int drain() {
        HIPCHK(sa_sync_stream_fwd(stream, device));
        return SA_OK;
    }
};
static SaUploader g_uploader;
// A second one for the second half of a batch's creation (batch_finish_body): with sa_batch_create_deferred that half runs on the
// batch's runner thread while the caller's thread is inside the NEXT batch's first half, which holds g_uploader for as long as it
// packs and uploads the reads (60 ms for a 10k-event slice) -- the batch that is ready to run would wait for it.
static SaUploader g_uploader_tail;
static thread_local SaUploader *tl_uploader = &g_uploader;   // the one upload() uses on this thread
SaPool g_sa_pool;
SaWorkers g_sa_workers;
static size_t g_sa_pool_idle_bytes(int device) { return g_sa_pool.idle_bytes(SaPool::DEVICE, device); }

// Streams and events of destroyed batches, kept per device for the next batch (creating three streams and ~50 events is
// 10 ms per batch).  Handles are only parked after the batch has drained them.
struct SaHandles {
    std::mutex mu;
    struct S { hipStream_t s; int dev; int kind; };   // kind 0: compute, 1: high priority
    struct E { hipEvent_t e; int dev; };
    std::vector<S> streams;
    std::vector<E> events;
    hipError_t stream(hipStream_t *out, int dev, int kind) {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < streams.size(); i++)
                if (streams[i].dev == dev && streams[i].kind == kind) {
                    *out = streams[i].s;
                    streams.erase(streams.begin() + (long) i);
                    return hipSuccess;
                }
        }
        if (kind == 0) return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
        int lo = 0, hi = 0;
        (void) hipDeviceGetStreamPriorityRange(&lo, &hi);
        return hipStreamCreateWithPriority(out, hipStreamNonBlocking, hi);
    }
    hipError_t event(hipEvent_t *out, int dev) {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = events.size(); i-- > 0;)
                if (events[i].dev == dev) {
                    *out = events[i].e;
                    events.erase(events.begin() + (long) i);
                    return hipSuccess;
                }
        }
        // blocking: a host thread that waits on one of these sleeps instead of spinning (see sa_sync_stream)
        return hipEventCreateWithFlags(out, hipEventBlockingSync);
    }
    void park(hipStream_t s, int dev, int kind) {
        if (!s) return;
        if (!SaPool::enabled()) { (void) hipStreamDestroy(s); return; }
        std::lock_guard<std::mutex> g(mu);
        streams.push_back(S{s, dev, kind});
    }
    void park(hipEvent_t e, int dev) {
        if (!e) return;
        if (SaPool::enabled()) {
            std::lock_guard<std::mutex> g(mu);   // events.size() is read under the lock: batches may be destroyed from several threads
            if (events.size() <= 4096) {
                events.push_back(E{e, dev});
                return;
            }
        }
        (void) hipEventDestroy(e);
    }
    void release() {
        std::lock_guard<std::mutex> g(mu);
        for (S &x : streams) (void) hipStreamDestroy(x.s);
        for (E &x : events) (void) hipEventDestroy(x.e);
        streams.clear();
        events.clear();
    }
};
static SaHandles g_handles;

// Waits for a stream without spinning: hipStreamSynchronize busy-waits by default, and a pipeline with several batches in
// flight then burns one CPU per waiting thread -- inside a container with a CPU quota that pushes the process over its
// share and the kernel throttles ALL its threads for the rest of the accounting period (measured: 40 ms stalls in
// sa_batch_create).  An event created with hipEventBlockingSync sleeps on an interrupt instead.
static hipError_t sa_sync_stream(hipStream_t s, int dev);
static hipError_t sa_sync_stream_fwd(hipStream_t s, int dev) { return sa_sync_stream(s, dev); }
static hipError_t sa_sync_stream(hipStream_t s, int dev) {
    hipEvent_t e = nullptr;
    if (g_handles.event(&e, dev) != hipSuccess) { (void) hipGetLastError(); return hipStreamSynchronize(s); }
    hipError_t r = hipEventRecord(e, s);
    if (r == hipSuccess) r = hipEventSynchronize(e);
    g_handles.park(e, dev);
    return r;
}

// the planner's big arrays as pinned memory of the caching allocator (the device then reads them by plain DMA)
static void *plan_pinned_alloc(size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    void *p = nullptr;
    if (g_sa_pool.get(SaPool::PINNED, &p, bytes, dev) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    return p;
}
static void plan_pinned_free(void *p, size_t bytes) { (void) bytes; g_sa_pool.put(SaPool::PINNED, p); }

extern "C" int sa_pool_configure(int64_t device_limit_bytes, int64_t pinned_limit_bytes) {
    if (device_limit_bytes >= 0) SaPool::configured(SaPool::DEVICE).store((long long) device_limit_bytes);
    if (pinned_limit_bytes >= 0) SaPool::configured(SaPool::PINNED).store((long long) pinned_limit_bytes);
    g_sa_pool.trim(SaPool::DEVICE);
    g_sa_pool.trim(SaPool::PINNED);
    return SA_OK;
}

// sa_host_alloc: page-locked blocks a caller fills with its reads' arrays (SA_FLAG_INPUTS_IN_HOST_BLOCK).  hipHostMalloc's default
// flags make them visible to every device of the process; the registry is what lets sa_batch_create check that a job's pointers
// really lie in such a block before a DMA is pointed at them.
static std::mutex g_host_blocks_mu;
static std::map<const char *, size_t> g_host_blocks;   // first byte -> bytes
extern "C" void *sa_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes > 0 ? bytes : 8, hipHostMallocDefault) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> g(g_host_blocks_mu);
    g_host_blocks[(const char *) p] = bytes > 0 ? bytes : 8;
    return p;
}
extern "C" void sa_host_free(void *block) {
    if (!block) return;
    {
        std::lock_guard<std::mutex> g(g_host_blocks_mu);
        auto it = g_host_blocks.find((const char *) block);
        if (it == g_host_blocks.end()) return;
        g_host_blocks.erase(it);
    }
    (void) hipHostFree(block);
}
// the block that holds `p`, if any
static bool sa_host_block_of(const char *p, const char **base, size_t *bytes) {
    std::lock_guard<std::mutex> g(g_host_blocks_mu);
    auto it = g_host_blocks.upper_bound(p);
    if (it == g_host_blocks.begin()) return false;
    --it;
    if (p >= it->first + it->second) return false;
    *base = it->first;
    *bytes = it->second;
    return true;
}

extern "C" void sa_pool_release_device(void) {
    g_sa_pool.release(SaPool::DEVICE);
}
extern "C" void sa_pool_release(void) {
    g_sa_pool.release(SaPool::DEVICE);
    g_sa_pool.release(SaPool::PINNED);
    sa_plan_pool_release();
    g_handles.release();
}

template <typename T>
static int upload(T **dst, const T *src, long long n, long long pad = 0, bool src_pinned = false) {
    // pad: extra zeroed elements behind the data (kernels that clamp an index may read one element past the end)
    size_t bytes = sizeof(T) * (size_t) (n + pad > 0 ? n + pad : 1);
    SaUploader &U = *tl_uploader;
    HIPCHK(g_sa_pool.get(SaPool::DEVICE, (void **) dst, bytes, U.device));
    if (pad > 0) HIPCHK(hipMemsetAsync((char *) *dst + sizeof(T) * (size_t) n, 0, sizeof(T) * (size_t) pad, U.stream));
    if (n > 0) return src_pinned ? U.copy_pinned(*dst, src, sizeof(T) * (size_t) n) : U.copy(*dst, src, sizeof(T) * (size_t) n);
    return SA_OK;
}

// What the last overflow taught: batches of one stream resemble each other, so the next batch of the same model and threshold
// starts with the candidate capacity the previous one had to grow to (a re-run of the whole pass costs a batch's kernel
// time again: the HDP workload at threshold 0.1 ran 73 instead of 37 ms per batch until it stopped overflowing every time).
// Keyed on the model OBJECT (its uid, not its address: the CLI clones and destroys a model per read and addresses come back), the
// threshold and the device; a few entries, least recently used out.  The factor is not for ever: after `patience` batches in a
// row without an overflow the next batch is planned one step (x4) lower; if that one overflows the old factor is back and the
// patience is four times longer -- one outlier batch no longer inflates every later batch's candidate, probability and result
// slots 4-16x.
struct SaCandMemo {
    struct Entry { uint64_t uid; double threshold; int device; int factor; int quiet; int patience; bool probing; uint64_t used; };
    std::mutex mu;
    std::vector<Entry> e;
    uint64_t clock = 0;
    Entry *find(uint64_t uid, double thr, int dev) {
        for (auto &x : e)
            if (x.uid == uid && x.threshold == thr && x.device == dev) { x.used = ++clock; return &x; }
        return nullptr;
    }
};
static SaCandMemo g_cand_memo;
static int cand_memo_factor(const sa_model_t *m, double threshold, int device) {   // once per batch created
    std::lock_guard<std::mutex> g(g_cand_memo.mu);
    SaCandMemo::Entry *x = g_cand_memo.find(m->uid, threshold, device);
    if (!x) return 1;
    if (x->probing && x->quiet >= 8) x->probing = false;   // the lower capacity held for eight batches
    if (x->factor > 1 && ++x->quiet >= x->patience) { x->factor /= 4; if (x->factor < 1) x->factor = 1; x->quiet = 0; x->probing = true; }
    return x->factor;
}
static void cand_memo_note(const sa_model_t *m, double threshold, int device, int factor) {   // a batch overflowed and grew to `factor`
    std::lock_guard<std::mutex> g(g_cand_memo.mu);
    SaCandMemo::Entry *x = g_cand_memo.find(m->uid, threshold, device);
    if (!x) {
        if (g_cand_memo.e.size() >= 32) {
            size_t lru = 0;
            for (size_t i = 1; i < g_cand_memo.e.size(); i++) if (g_cand_memo.e[i].used < g_cand_memo.e[lru].used) lru = i;
            g_cand_memo.e.erase(g_cand_memo.e.begin() + (long) lru);
        }
        g_cand_memo.e.push_back({m->uid, threshold, device, factor, 0, 64, false, ++g_cand_memo.clock});
        return;
    }
    if (factor > x->factor) x->factor = factor;
    if (x->probing && x->patience < (1 << 20)) x->patience *= 4;   // the lower capacity did not hold
    x->probing = false;
    x->quiet = 0;
}

// The slack of the speculative candidate bound (sa_strip.inc: STRIP_SPEC_SLACK) a model had to grow to on a device is remembered
// too: a stream of batches whose totals drift further than the default allows (longer tracebacks, densities broader than the
// bundled HDP's) would otherwise run every batch's pass twice.  Keyed like the candidate capacity; +inf is remembered as well.
struct SaSpecMemo {
    struct Entry { uint64_t uid; int device; double slack; uint64_t used; };
    std::mutex mu;
    std::vector<Entry> e;
    uint64_t clock = 0;
};
static SaSpecMemo g_spec_memo;
static double spec_memo_slack(const sa_model_t *m, int device, double dflt) {
    std::lock_guard<std::mutex> g(g_spec_memo.mu);
    for (auto &x : g_spec_memo.e)
        if (x.uid == m->uid && x.device == device) { x.used = ++g_spec_memo.clock; return x.slack > dflt ? x.slack : dflt; }
    return dflt;
}
static void spec_memo_note(const sa_model_t *m, int device, double slack) {
    std::lock_guard<std::mutex> g(g_spec_memo.mu);
    for (auto &x : g_spec_memo.e)
        if (x.uid == m->uid && x.device == device) { if (slack > x.slack) x.slack = slack; x.used = ++g_spec_memo.clock; return; }
    if (g_spec_memo.e.size() >= 32) {
        size_t lru = 0;
        for (size_t i = 1; i < g_spec_memo.e.size(); i++) if (g_spec_memo.e[i].used < g_spec_memo.e[lru].used) lru = i;
        g_spec_memo.e.erase(g_spec_memo.e.begin() + (long) lru);
    }
    g_spec_memo.e.push_back({m->uid, device, slack, ++g_spec_memo.clock});
}

static std::atomic<int> g_batches_started(0);
static void dplan_release_fwd(sa_batch *b, struct DPlanPending *P);   // sa_dplan.inc (below)   // batches between sa_batch_start and sa_batch_wait (this process)

void sa_batch_destroy(sa_batch_t *b) {
    if (!b) return;
    if (b->runner) { b->runner->join(); delete b->runner; b->runner = nullptr; g_batches_started.fetch_sub(1); }
    if (b->device >= 0) (void) hipSetDevice(b->device);
    if (b->pending) { dplan_release_fwd(b, b->pending); b->pending = nullptr; }   // created, never used
    // the storage goes back to the caching allocators without the implicit synchronisation of hipFree: nothing of this
    // batch may still be in flight (only possible after an error inside a run)
    const bool trace_d = getenv("SA_TRACE") != nullptr;
    auto now_ms_d = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double td0 = now_ms_d();
    if (!b->quiet) {   // (a failed or interrupted run, or a batch that never ran: 0.6-0.9 ms of API calls otherwise, per batch,
                       // on the thread that is about to plan the next one)
        for (int i = 0; i < 2; i++)
            if (b->cstream[i]) (void) hipStreamSynchronize(b->cstream[i]);
        for (int i = 0; i < 2; i++)
            if (b->xstream[i]) (void) hipStreamSynchronize(b->xstream[i]);
        if (b->pair_stream) (void) hipStreamSynchronize(b->pair_stream);
    }
    const double td1 = now_ms_d();
    void *ptrs[] = {b->d_regions, b->d_rows, b->d_pk, b->d_poff, b->d_pid, b->d_px, b->d_xc, b->d_prec, b->d_ev, b->d_segs, b->d_cks, b->d_F, b->d_E,
                    b->d_vbuf, b->d_cands, b->d_cand_count, b->d_overflow, b->d_totals, b->d_bscratch, b->d_tab6, b->d_noise3, b->d_evn, b->d_two,
                    b->d_hdp_slot, b->d_hdp_y, b->d_hdp_slope, b->d_hdp_grid, b->d_hdp_tab, b->d_hdp_coef, b->d_prob, b->d_seg_pass, b->d_seg_off,
                    b->d_out, b->d_ids, b->d_gsum, b->d_gmc, b->d_seam, b->d_ckxy, b->d_blk, b->d_spec, b->d_sortkey, b->d_sortidx,
                    b->d_vc_bits, b->d_vc_off, b->d_seg_all};
    for (void *p : ptrs)
        if (p) g_sa_pool.put(SaPool::DEVICE, p);
    for (int i = 0; i < 8; i++)
        g_handles.park(b->ev[i], b->device);
    for (hipEvent_t e : b->gev) g_handles.park(e, b->device);
    for (hipEvent_t e : b->cev) g_handles.park(e, b->device);
    for (int i = 0; i < 2; i++)
        g_handles.park(b->cstream[i], b->device, 0);
    for (int i = 0; i < 2; i++)
        g_handles.park(b->xstream[i], b->device, 0);
    g_handles.park(b->pair_stream, b->device, 1);
    g_sa_pool.put(SaPool::PINNED, b->h_pairs);
    g_sa_pool.put(SaPool::PINNED, b->held_stage);
    g_sa_pool.put(SaPool::DEVICE, b->d_pairs_up);
    g_sa_pool.put(SaPool::PINNED, b->h_seg_off);
    g_sa_pool.put(SaPool::PINNED, b->h_overflow);
    const double td2 = now_ms_d();
    sa_plan_free(b->plan);
    delete b;
    if (trace_d) fprintf(stderr, "[trace] destroy: streams idle after %.2f ms, blocks parked after %.2f ms, done after %.2f ms\n", td1 - td0, td2 - td0, now_ms_d() - td0);
}

#include "sa_dplan.inc"
static void dplan_release_fwd(sa_batch *b, DPlanPending *P) { dplan_release(b, P, true); }

// Pairs per event of the last finished batch of the same MODEL (its uid: a broad HDP at threshold 0.01 returns 17.8 pairs per event, a
// narrow model beside it 0.9), device and threshold, process-wide: the estimate the NEXT such batch's pinned result block is sized
// from (a batch whose estimate is short copies its pairs after its kernels instead of beside them).  Clamped to [1.5, 64] pairs per
// event; a pinned block that cannot be had at the estimated size is not an error (the run copies after its kernels, as without one).
struct SaPairsMemo {
    std::mutex mu;
    struct E { uint64_t uid; int device; double thr, ratio; };
    E e[8] = {};
    int next = 0;
    void note(uint64_t uid, int device, double threshold, double pairs, double events) {
        if (!(events > 0)) return;
        std::lock_guard<std::mutex> g(mu);
        for (int i = 0; i < 8; i++)
            if (e[i].uid == uid && e[i].device == device && e[i].thr == threshold && e[i].ratio > 0) { e[i].ratio = pairs / events > 1e-9 ? pairs / events : 1e-9; return; }
        e[next] = E{uid, device, threshold, pairs / events > 1e-9 ? pairs / events : 1e-9};
        next = (next + 1) & 7;
    }
    double estimate(uint64_t uid, int device, double threshold) {
        std::lock_guard<std::mutex> g(mu);
        for (int i = 0; i < 8; i++)
            if (e[i].uid == uid && e[i].device == device && e[i].thr == threshold && e[i].ratio > 0) {
                const double r = e[i].ratio * 1.1;
                return r < 1.5 ? 1.5 : (r > 64.0 ? 64.0 : r);
            }
        return 1.5;
    }
};
static SaPairsMemo g_pairs_memo;

static int batch_create_impl(sa_batch_t **out, const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs,
                             const char *const *ambig, int device, unsigned flags, bool deferred) {
    if (!out || !m || !p) return SA_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[signalalign_hip] no HIP device available; this library has no CPU fallback\n");
        return SA_ENODEVICE;
    }
    if (device < 0 || device >= ndev) return SA_EINVAL;
    // Threshold 0 keeps every band cell, the ones of posterior 0 included: the default kernels' candidate filter (forward +
    // backward >= checkpoint maximum + log threshold) has no lower bound then and would pass lanes that hold no cell.  Such a
    // batch takes the reference-ordered kernels with host finalisation, which list a diagonal's cells explicitly.
    if (!(p->threshold > 0.0)) flags |= SA_FLAG_EXACT;
    // The two-distribution emission (sa_model_set_emission) exists in the reference-ordered memory-resident kernels and, since round
    // 6, in the register kernels (k_fwd_fast_two / k_bwd_fast_two: one path per cell; wide stretches through their in-kernel
    // memory-resident path).  A batch whose regions are not ALL register-kernel regions (an ambiguity letter, a matrix that the
    // planner splits beyond their limits) is planned again as with SA_FLAG_EXACT (batch_prepare_body); the expectation pass keeps the
    // reference-ordered kernels.  (SA_TWO_DIST_FAST_OFF=1: always the reference-ordered kernels, as up to round 5.)
    if (m->emission != 0 && ((flags & (SA_FLAG_EXPECT_INTERNAL | SA_FLAG_FORCE_GENERIC)) || getenv("SA_TWO_DIST_FAST_OFF")))
        flags |= SA_FLAG_EXACT;
    const bool trace_c = getenv("SA_TRACE") != nullptr;
    auto now_ms_c = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double tc0 = now_ms_c();
    HIPCHK(hipSetDevice(device));
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    if (trace_c) fprintf(stderr, "[trace] create: memory queried at %.1f ms\n", now_ms_c() - tc0);
    free_b += g_sa_pool.idle_bytes(SaPool::DEVICE, device);   // what destroyed batches left parked is available to this one
    if (deferred && (flags & SA_FLAG_DEVICE_TO_ITSELF)) free_b += g_sa_pool.live_bytes(SaPool::DEVICE, device);   // ... and what the running ones hold
    // forward storage gets at most 60% of what is free; 24 B per cell-path (HDP models: 8 B more, the emission plane)
    long long budget = (long long) ((double) free_b * 0.60 / (m->hdp ? 32.0 : 24.0));
    const char *envb = getenv("SA_F_BUDGET_CELLPATHS");  // test hook: force several passes
    if (envb && atoll(envb) > 0) budget = atoll(envb);

    sa_batch *b = new sa_batch();
    b->plan = nullptr;
    b->pending = nullptr; b->finished = false; b->finish_rc = SA_OK; b->c_deferred = false; b->held_stage = nullptr;
    b->c_m = m; b->c_p = *p; b->c_jobs = jobs; b->c_n = n_jobs; b->c_ambig = ambig; b->c_budget = budget; b->c_t0 = tc0;
    b->dev_planned = false;
    b->device = device;
    b->flags = flags;
    b->stream = nullptr;
    b->cstream[0] = b->cstream[1] = nullptr;
    b->xstream[0] = b->xstream[1] = nullptr;
    b->pair_stream = nullptr;
    b->h_seg_off = nullptr;
    b->h_overflow = nullptr;
    b->ran = false; b->released = false;
    b->quiet = false;
    b->runner = nullptr; b->runner_rc = SA_OK;
    b->d_regions = nullptr; b->d_rows = nullptr; b->d_pk = nullptr; b->d_poff = nullptr; b->d_pid = nullptr; b->d_px = nullptr; b->d_xc = nullptr;
    b->d_prec = nullptr; b->d_blk = nullptr;
    b->d_ev = nullptr; b->d_segs = nullptr; b->d_cks = nullptr; b->d_F = nullptr; b->d_E = nullptr; b->d_vbuf = nullptr;
    b->d_cands = nullptr; b->d_cand_count = nullptr; b->d_overflow = nullptr; b->d_totals = nullptr;
    b->d_seam = nullptr; b->d_ckxy = nullptr; b->seam_cap = 0; b->seam_cap_bwd = 0; b->seam_bwd_off = 0; b->strip_on = false;
    b->prepared = false; b->prepare_rc = SA_OK; b->lw_strip_max_n = b->lw_strip_max_seg = b->lw_strip_fwd_slots = b->lw_strip_bwd_slots = 0;
    b->strip_one_pass = false; b->d_spec = nullptr; b->d_sortkey = nullptr; b->d_sortidx = nullptr;
    // the exact totals of a traceback drift away from its speculative total diagonal by diagonal (1.6e-4 per diagonal with the flat
    // HDP fixture: sa_strip.inc), so the slack is sized for the traceback's length -- the default 0.5 at the default 1100 diagonals --
    // and starts from what earlier batches of this model on this device had to grow to
    b->spec_slack = STRIP_SPEC_SLACK * std::max(1.0, (double) (p->min_diags_between_trace_back + p->trace_back_diagonals) / 1100.0);
    b->spec_slack = spec_memo_slack(m, device, b->spec_slack);
    if (const char *ets = getenv("SA_TEST_SPEC_SLACK")) {   // test hook: a slack the totals' drift exceeds, so that the repeat below is exercised
        const double v_ = atof(ets);
        if (v_ > 0.0) b->spec_slack = v_;
    }
    b->d_bscratch = nullptr; b->d_tab6 = nullptr; b->d_noise3 = nullptr; b->d_evn = nullptr; b->d_two = nullptr; b->two_xn_off = 0; b->d_hdp_slot = nullptr; b->d_hdp_y = nullptr;
    b->d_hdp_slope = nullptr; b->d_hdp_grid = nullptr; b->d_hdp_tab = nullptr; b->d_hdp_coef = nullptr; b->d_prob = nullptr; b->d_seg_pass = nullptr;
    b->d_seg_off = nullptr; b->d_out = nullptr; b->d_ids = nullptr; b->d_gsum = nullptr; b->d_gmc = nullptr;
    b->cand_alloc = 0; b->out_alloc = 0; b->cand_factor = 1; b->spec_repeats = 0;
    b->h_pairs = nullptr; b->h_pairs_cap = 0; b->n_pairs_total = 0;
    b->d_pairs_up = nullptr; b->d_pairs_up_cap = 0;
    memset(&b->stats, 0, sizeof(b->stats));
    for (int i = 0; i < 8; i++) b->ev[i] = nullptr;
#define TRY(x) do { int rc_ = (x); if (rc_) { sa_batch_destroy(b); return rc_; } } while (0)
    if (g_handles.stream(&b->cstream[0], device, 0) != hipSuccess || g_handles.stream(&b->cstream[1], device, 0) != hipSuccess) {
        sa_batch_destroy(b);
        return SA_ENODEVICE;
    }
    b->stream = b->cstream[0];
    {   // the copy stream outranks the compute streams
        if (g_handles.stream(&b->pair_stream, device, 1) != hipSuccess) {
            sa_batch_destroy(b);
            return SA_ENODEVICE;
        }
    }
    for (int i = 0; i < 8; i++)
        if (g_handles.event(&b->ev[i], device) != hipSuccess) { sa_batch_destroy(b); return SA_ENODEVICE; }
    // ---- the plan: on the device when the batch allows it (sa_dplan.inc: its first half here), else on the host ----
    {
        std::unique_lock<std::mutex> dp_lock(g_uploader.mu);
        TRY(g_uploader.bind(device));
        const int rcd = dplan_front(b, m, p, jobs, n_jobs, ambig, flags, budget, &b->pending);
        if (rcd < 0) { dp_lock.unlock(); sa_batch_destroy(b); return rcd; }
    }
#undef TRY
    b->c_deferred = deferred && b->pending != nullptr;
    if (!b->c_deferred) {   // (a batch the device planner does not take is planned on the host right away)
        const int rc = batch_finish(b);
        if (rc) { sa_batch_destroy(b); return rc; }
    }
    *out = b;
    return SA_OK;
}

int sa_batch_create(sa_batch_t **out, const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs,
                    const char *const *ambig, int device, unsigned flags) {
    return batch_create_impl(out, m, p, jobs, n_jobs, ambig, device, flags, false);
}
int sa_batch_create_deferred(sa_batch_t **out, const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs,
                             const char *const *ambig, int device, unsigned flags) {
    return batch_create_impl(out, m, p, jobs, n_jobs, ambig, device, flags, true);
}

// Second half of a batch's creation: the plan (the device planner's results, or the host planner), the remaining uploads, the
// working buffers and the launch lists.  Runs once, on the batch's first use (run, statistics, accessors) or at the end of
// sa_batch_create; a failure is remembered and returned to every later caller.
static int batch_finish_body(sa_batch *b);
static int batch_prepare_body(sa_batch *b);
static int batch_finish(sa_batch *b) {
    std::lock_guard<std::mutex> g(b->fin_mu);
    if (!b->finished) {
        b->finish_rc = b->prepared ? b->prepare_rc : batch_prepare_body(b);
        if (b->finish_rc == SA_OK) b->finish_rc = batch_finish_body(b);
        b->finished = true;
        if (b->finish_rc != SA_OK) {
            // Memsets, uploads and k_fill_xc of this batch may still be queued on the upload stream it used; sa_batch_destroy
            // only waits for the batch's own streams before its blocks go back to the caching allocator, where a batch being
            // created on the other uploader could receive them while that work still writes.  Drain the stream here.
            SaUploader *const U = b->c_deferred ? &g_uploader_tail : &g_uploader;
            if (U->stream && U->device == b->device) (void) sa_sync_stream(U->stream, b->device);
        }
    }
    return b->finish_rc;
}
// Launch lists: regions per forward-storage pass, traceback segments per result group (needs the plan, no device memory)
static int batch_build_lists(sa_batch *b) {
    sa_plan_t *pl = b->plan;
    const unsigned flags = b->flags;
    const bool host_finalize = (flags & SA_FLAG_EXACT) || b->expect;
    int want = 1;
    const char *envg = getenv("SA_GROUPS");  // test hook
    if (envg && atoi(envg) > 0) want = atoi(envg);
    else if (!host_finalize) want = pl->n_chunks == 1 ? 8 : (pl->n_chunks < 4 ? 4 : 2);
    // A caller that keeps batches in flight (sa_batch_start: another batch of this process is running while this one is
    // created) already overlaps a batch's result copy with its neighbours' kernels; what it wants is few, large launches:
    // 2000 x 5000-event reads, three in flight, step time with 1 / 2 / 3 / 8 groups: 13.9 / 13.1 / 12.9 / 14.9 ms.
    if (!(envg && atoi(envg) > 0) && !host_finalize && g_batches_started.load() > 0 && !(flags & SA_FLAG_DEVICE_TO_ITSELF) && want > 3) want = 3;
    b->ids_flat.clear(); b->chunks.clear(); b->groups.clear();
    // One-path ring-kernel regions go to the strip kernels (sa_strip.inc): Gaussian emissions, default arithmetic,
    // device-side finalisation, reference windows of fewer than 64 * STRIP_NS_MAX positions.  SA_STRIP=0: ring kernels.
    b->strip_on = !host_finalize && !(getenv("SA_STRIP") && atoi(getenv("SA_STRIP")) == 0);   // (HDP regions too: they read the emission plane)
    auto strip_region = [&](const sa_region_t &Rq) {
        return b->strip_on && Rq.kind == SA_KIND_RING && Rq.max_p == 1 && Rq.lX < 64ll * STRIP_NS_MAX && Rq.N >= 1;
    };
    long long strip_max_n = 0, strip_max_seg = 0, strip_fwd_slots = 0, strip_bwd_slots = 0;
    long long r = 0;
    for (int c = 0; c < pl->n_chunks; c++) {
        long long ra = r;
        while (r < pl->n_regions && pl->regions[r].chunk == c) r++;
        long long rb = r;
        sa_launch_chunk C;
        std::vector<int> gr, fr, sr_;
        double work = 0;
        std::vector<int> rr[16];
        auto ring_class = [](const sa_region_t &Rq) {
            const int cl = Rq.max_rowpaths <= 64 ? 0 : (int) ((Rq.max_rowpaths - 1) / 64);   // <= 7 (SA_RING_MAX_ROWPATHS)
            return (Rq.max_p > 1 ? 8 : 0) + (cl > 7 ? 7 : cl);
        };
        for (long long q = ra; q < rb; q++) {
            const sa_region_t &Rq = pl->regions[q];
            if (strip_region(Rq)) { sr_.push_back((int) q); strip_max_n = Rq.N > strip_max_n ? Rq.N : strip_max_n; }
            else if (Rq.kind == SA_KIND_RING) rr[ring_class(Rq)].push_back((int) q);
            else if (Rq.kind != SA_KIND_FAST) gr.push_back((int) q);
            else fr.push_back((int) q);
            work += (double) Rq.N;
        }
        auto by_len_r = [&](int a, int d) { return pl->regions[a].N > pl->regions[d].N; };
        // longest first inside each launch: the tail of a launch is then made of short waves
        std::stable_sort(gr.begin(), gr.end(), by_len_r);
        std::stable_sort(fr.begin(), fr.end(), by_len_r);
        C.ids_gr = (long long) b->ids_flat.size(); C.ngr = (int) gr.size();
        b->ids_flat.insert(b->ids_flat.end(), gr.begin(), gr.end());
        C.ids_fr = (long long) b->ids_flat.size(); C.nfr = (int) fr.size();
        b->ids_flat.insert(b->ids_flat.end(), fr.begin(), fr.end());
        for (int cl = 0; cl < 16; cl++) {
            std::stable_sort(rr[cl].begin(), rr[cl].end(), by_len_r);
            C.ids_rr[cl] = (long long) b->ids_flat.size(); C.nrr[cl] = (int) rr[cl].size();
            b->ids_flat.insert(b->ids_flat.end(), rr[cl].begin(), rr[cl].end());
        }
        std::stable_sort(sr_.begin(), sr_.end(), by_len_r);
        C.ids_st = (long long) b->ids_flat.size(); C.nst = (int) sr_.size();
        b->ids_flat.insert(b->ids_flat.end(), sr_.begin(), sr_.end());
        strip_fwd_slots = (long long) sr_.size() > strip_fwd_slots ? (long long) sr_.size() : strip_fwd_slots;
        long long chunk_bwd_slots = 0;
        C.g0 = (int) b->groups.size();
        // a group should still be a sizeable launch: at least 2048 segments each (measured optimum 6-8 groups
        // for 18000 segments; 16 and more lose to launch gaps)
        long long nseg_chunk = 0, nseg_wide = 0;
        for (long long q = ra; q < rb; q++) {
            nseg_chunk += pl->regions[q].n_seg;
            if ((pl->regions[q].kind == SA_KIND_FAST && pl->regions[q].slots >= 2) ||
                (pl->regions[q].kind == SA_KIND_RING && pl->regions[q].max_rowpaths > 64))
                nseg_wide += pl->regions[q].n_seg;
        }
        // segments of wide-band regions live three to four times longer than those of dense anchors (4 ms against
        // 1.2 ms), and so do the tails of their launches: fewer, larger groups.  2000 reads with realistic anchors,
        // 17 300 segments, step time with 1 / 2 / 3 / 4 / 6 / 8 groups: 70.2 / 67.8 / 69.3 / 73.1 / 80.2 / 87 ms
        // (strip-kernel segments: 1 / 2 / 3 / 4 groups give 39.5 / 38.3 / 37.5 / 42.9 ms per step of fresh reads)
        const long long min_per_group = (2 * nseg_wide > nseg_chunk) ? (b->strip_on ? 5500 : 8192) : 2048;
        int ng = want;
        if (!(envg && atoi(envg) > 0))
            while (ng > 1 && nseg_chunk / ng < min_per_group) ng--;
        long long q = ra;
        double acc = 0;
        for (int g = 0; g < ng && q < rb; g++) {
            long long qa = q;
            double target = work * (double) (g + 1) / (double) ng;
            while (q < rb && (g == ng - 1 || acc < target)) { acc += (double) pl->regions[q].N; q++; }
            if (q == qa) continue;
            // a read's regions stay in one group so that its pairs are contiguous in the output
            while (q < rb && pl->regions[q].job == pl->regions[q - 1].job) { acc += (double) pl->regions[q].N; q++; }
            sa_launch_group G;
            G.seg0 = G.seg1 = G.ck0 = G.ck1 = 0;
            std::vector<int> gs, fs, rs[16], ss;
            bool any = false;
            for (long long t = qa; t < q; t++) {
                const sa_region_t *R = &pl->regions[t];
                for (long long sg = R->seg_off; sg < R->seg_off + R->n_seg; sg++) {
                    if (strip_region(*R)) {
                        ss.push_back((int) sg);
                        const long long span = pl->segs[sg].start - pl->segs[sg].to;
                        strip_max_seg = span > strip_max_seg ? span : strip_max_seg;
                    }
                    else if (R->kind == SA_KIND_RING) rs[ring_class(*R)].push_back((int) sg);
                    else (R->kind != SA_KIND_FAST ? gs : fs).push_back((int) sg);
                    const sa_seg_t *S = &pl->segs[sg];
                    if (!any) { G.seg0 = sg; G.ck0 = S->ck_base; any = true; }
                    G.seg1 = sg + 1;
                    G.ck1 = S->ck_base + S->n_ck;
                }
            }
            if (!any) continue;
            auto by_len_s = [&](int a, int d) {
                return pl->segs[a].start - pl->segs[a].to > pl->segs[d].start - pl->segs[d].to;
            };
            std::stable_sort(gs.begin(), gs.end(), by_len_s);
            std::stable_sort(fs.begin(), fs.end(), by_len_s);
            G.ids_gs = (long long) b->ids_flat.size(); G.ngs = (int) gs.size();
            b->ids_flat.insert(b->ids_flat.end(), gs.begin(), gs.end());
            G.ids_fs = (long long) b->ids_flat.size(); G.nfs = (int) fs.size();
            b->ids_flat.insert(b->ids_flat.end(), fs.begin(), fs.end());
            for (int cl = 0; cl < 16; cl++) {
                std::stable_sort(rs[cl].begin(), rs[cl].end(), by_len_s);
                G.ids_rs[cl] = (long long) b->ids_flat.size(); G.nrs[cl] = (int) rs[cl].size();
                b->ids_flat.insert(b->ids_flat.end(), rs[cl].begin(), rs[cl].end());
            }
            std::stable_sort(ss.begin(), ss.end(), by_len_s);
            G.ids_ss = (long long) b->ids_flat.size(); G.nss = (int) ss.size();
            b->ids_flat.insert(b->ids_flat.end(), ss.begin(), ss.end());
            G.seam_first = (unsigned) chunk_bwd_slots;   // (rebased behind the forward slots below)
            chunk_bwd_slots += (long long) ss.size();
            b->groups.push_back(G);
        }
        strip_bwd_slots = chunk_bwd_slots > strip_bwd_slots ? chunk_bwd_slots : strip_bwd_slots;
        C.g1 = (int) b->groups.size();
        b->chunks.push_back(C);
    }
    b->lw_strip_max_n = strip_max_n; b->lw_strip_max_seg = strip_max_seg;
    b->lw_strip_fwd_slots = strip_fwd_slots; b->lw_strip_bwd_slots = strip_bwd_slots;
    return SA_OK;
}

// First step of finishing a batch, everything that needs no working storage: the plan (from the device planner, or built on the host)
// and the launch lists.  sa_batch_prepare runs it ahead of time for a deferred batch, while the batch before it is on the device.
static int batch_prepare_body(sa_batch *b) {
    const sa_model_t *m = b->c_m;
    const sa_params_t *p = &b->c_p;
    const sa_job_t *jobs = b->c_jobs;
    const int64_t n_jobs = b->c_n;
    const char *const *ambig = b->c_ambig;
    unsigned flags = b->flags;   // (gains SA_FLAG_EXACT when a two-distribution batch is planned again below)
    const long long budget = b->c_budget;
    const int device = b->device;
    const bool trace_c = getenv("SA_TRACE") != nullptr;
    auto now_ms_c = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double tc0 = b->c_t0;
    HIPCHK(hipSetDevice(device));
    // (only a deferred batch: with several batches in flight and creation in one piece the second upload stream measured
    // 1.5-3 ms per step slower than one)
    SaUploader *const UPT = b->c_deferred ? &g_uploader_tail : &g_uploader;
    struct UseTail { SaUploader *prev; UseTail(SaUploader *u) : prev(tl_uploader) { tl_uploader = u; } ~UseTail() { tl_uploader = prev; } } use_tail_(UPT);
#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
    sa_plan_t *pl = nullptr;
    if (b->pending) {
        DPlanPending *P = b->pending;
        b->pending = nullptr;
        const int rcd = dplan_back(b, P);
        if (rcd < 0) return rcd;
        if (rcd == SA_OK) pl = b->plan;
    }
    if (!pl) {
        // pinning memory costs about 0.25 ms per MB: it pays for a process that streams batches (the blocks are reused), not
        // for the one or two batches of a command-line run, which stage their plan through the uploader's ring instead
        static std::atomic<int> batches_created(0);
        if (SaPool::enabled() && batches_created.fetch_add(1) >= 2) sa_plan_use_allocator(plan_pinned_alloc, plan_pinned_free);
        int rc = sa_plan_build(&pl, m, p, jobs, n_jobs, ambig, flags | SA_FLAG_DEVICE_XC_INTERNAL, budget);
        if (rc == SA_OK && m->emission != 0 && !(flags & SA_FLAG_EXACT) && pl->n_fast_regions != pl->n_regions) {
            // the two-distribution emission off the register kernels: the reference-ordered kernels for the whole batch
            sa_plan_free(pl);
            pl = nullptr;
            b->flags |= SA_FLAG_EXACT;
            flags = b->flags;
            rc = sa_plan_build(&pl, m, p, jobs, n_jobs, ambig, flags | SA_FLAG_DEVICE_XC_INTERNAL, budget);
        }
        sa_plan_use_allocator(nullptr, nullptr);
        if (rc) return rc;
        if (trace_c) fprintf(stderr, "[trace] create: planned at %.1f ms\n", now_ms_c() - tc0);
        b->plan = pl;
    }
    if (trace_c) fprintf(stderr, "[trace] create: planned (%s) at %.1f ms\n", b->dev_planned ? "device" : "host", now_ms_c() - tc0);

    b->expect = (flags & SA_FLAG_EXPECT_INTERNAL) != 0;
    b->plan_hdp = m->hdp != nullptr;
    b->p8 = (flags & SA_FLAG_PAIRS8) != 0 && !b->expect;
    if (b->p8) {   // 20 bits per coordinate, no path index, no k-mer: only what one path per cell and short matrices allow
        if (flags & SA_FLAG_VC_ROWS) return SA_EINVAL;
        for (int64_t j = 0; j < n_jobs; j++)
            if (jobs[j].ref_len >= SA_PAIR8_MAX_COORD || jobs[j].n_events >= SA_PAIR8_MAX_COORD) return SA_EUNSUPPORTED;
        for (long long r = 0; r < pl->n_regions; r++)
            if (pl->regions[r].max_p > 1) return SA_EUNSUPPORTED;
    }
    b->relax = !(flags & SA_FLAG_EXACT) && !b->expect && m->hdp == nullptr;
    b->ring_cap = 0;
    b->gen_threads = 64;
    b->wide_cap = 0;
    {   // register-kernel regions with diagonals too wide for the registers: an LDS ring of up to 256 cells per row
        long long widest = 0;
        for (long long r = 0; r < pl->n_regions; r++)
            if (pl->regions[r].kind == SA_KIND_FAST && pl->regions[r].slots > 1 && pl->regions[r].max_rowpaths > widest)
                widest = pl->regions[r].max_rowpaths;
        if (widest > 0) b->wide_cap = (int) (widest < 256 ? (widest + 31) / 32 * 32 : 256);
        if (const char *envw = getenv("SA_WIDE_CAP")) b->wide_cap = atoi(envw) < 0 ? 0 : atoi(envw);  // tuning / test hook
    }
    for (long long r = 0; r < pl->n_regions; r++)
        if (pl->regions[r].kind == SA_KIND_GENERIC && pl->regions[r].max_rowpaths > 64) b->gen_threads = 128;
    if (const char *envt = getenv("SA_GENERIC_THREADS")) b->gen_threads = atoi(envt) == 128 ? 128 : 64;  // test hook
    if (b->relax) {
        long long cap = 0;
        for (long long r = 0; r < pl->n_regions; r++)
            if (pl->regions[r].kind == SA_KIND_GENERIC && pl->regions[r].max_rowpaths > cap) cap = pl->regions[r].max_rowpaths;
        // diagonals wider than the ring go through global memory one by one; a small ring keeps many waves per CU
        long long lim = 128;
        const char *envr = getenv("SA_RING_CAP");  // tuning hook
        if (envr && atoll(envr) > 0) lim = atoll(envr);
        if (lim > 900) lim = 900;                   // 64 KB of dynamic LDS
        b->ring_cap = (int) (cap < lim ? cap : lim);
    }
    // (round 4: the model tables, the plan arrays of a host-built plan and the emission constants go up here too -- with
    // sa_batch_prepare that is while the batch before this one still runs; nothing of it needs the working storage)
    std::unique_lock<std::mutex> up_lock((*UPT).mu);
    TRY((*UPT).bind(device));
    if (trace_c) fprintf(stderr, "[trace] create: upload ring ready at %.1f ms\n", now_ms_c() - tc0);
    {   // candidate capacity an earlier batch of this stream had to grow to
        const int f = pl->params.threshold > 0.0 ? cand_memo_factor(m, pl->params.threshold, device) : 1;
        if (f > 1) {
            sa_plan_grow_candidates(pl, f);
            b->cand_factor = f;
            if (b->dev_planned && pl->n_segs > 0)   // its segments are in HBM already: the copy is stream-ordered behind the planner
                TRY((*UPT).copy(b->d_segs, pl->segs, sizeof(sa_seg_t) * (size_t) pl->n_segs));
        }
    }
    const bool big_pinned = pl->pooled && pl->big_free == plan_pinned_free;   // the big arrays are pinned: no staging
    if (!b->dev_planned) {
        TRY(upload(&b->d_regions, pl->regions, pl->n_regions));
        TRY(upload(&b->d_rows, pl->rows, pl->n_rows, 192, big_pinned));   // (the ring kernels read row tiles up to 127 entries past a region)
        TRY(upload(&b->d_pk, pl->pk, pl->n_pk, 0, big_pinned));
        TRY(upload(&b->d_poff, pl->poff, pl->n_poff, 0, big_pinned));
        TRY(upload(&b->d_pid, pl->pid, pl->n_pid, 0, big_pinned));
    } else {   // the scan kernel set f_base / chunk / seg_off on the device; the host copy has them too
    }
    if (!b->dev_planned) {   // per-path records: only batches that hold ring-kernel regions with several paths per cell have (and read) them
        bool need = false;
        for (long long r = 0; r < pl->n_regions && !need; r++) need = pl->regions[r].kind == SA_KIND_RING && pl->regions[r].max_p > 1;
        if (need && pl->prec) TRY(upload(&b->d_prec, pl->prec, pl->n_pid, 0, big_pinned));
    }
    std::vector<int> px;   // (alive until the uploader has drained)
    {   // cell-path -> reference position, for the memory-resident kernels (one lane per cell-path); register-kernel
        // regions never read it
        bool any_generic = false;
        for (long long r = 0; r < pl->n_regions && !any_generic && !b->dev_planned; r++) any_generic = pl->regions[r].kind == SA_KIND_GENERIC;
        if (any_generic) {
            px.assign((size_t) (pl->n_pid > 0 ? pl->n_pid : 1), 0);
            for (long long r = 0; r < pl->n_regions; r++) {
                const sa_region_t *R = &pl->regions[r];
                if (R->kind != SA_KIND_GENERIC) continue;
                const int32_t *po = pl->poff + R->poff_off;
                for (long long x = 0; x <= R->lX; x++)
                    for (int g = po[x]; g < po[x + 1]; g++) px[(size_t) (R->pid_off + g)] = (int) x;
            }
            TRY(upload(&b->d_px, px.data(), pl->n_pid));
        } else {
            TRY(upload(&b->d_px, (const int *) nullptr, 0));
        }
    }
    // readable padding behind the events: the kernels clamp event indices to 0 even for reads without events
    if (!b->dev_planned) {
        TRY(upload(&b->d_ev, pl->ev, pl->n_ev, 8, big_pinned));
        TRY(upload(&b->d_segs, pl->segs, pl->n_segs));
        TRY(upload(&b->d_cks, pl->cks, pl->n_cks));
    }
    {   // model tables
        std::vector<double> tab6((size_t) m->n_kmers * 6);
        for (long long i = 0; i < m->n_kmers; i++) {
            double mu = m->table5[5 * i], sd = m->table5[5 * i + 1];
            double sdy = sd * SA_GAPY_SD_MULT;  // stateMachine3_loadFromFile multiplies the loaded sd (impl/stateMachine.c:1530-1532)
            tab6[6 * i + 0] = mu;
            tab6[6 * i + 1] = sd == 0.0 ? 1.0 : sd;
            tab6[6 * i + 2] = sd == 0.0 ? -INFINITY : (-0.91893853320467267 - log(sd));
            tab6[6 * i + 3] = sdy == 0.0 ? 1.0 : sdy;
            tab6[6 * i + 4] = sdy == 0.0 ? -INFINITY : (-0.91893853320467267 - log(sdy));
            tab6[6 * i + 5] = 0.0;
        }
        TRY(upload(&b->d_tab6, tab6.data(), (long long) tab6.size()));
        if (m->emission != 0) {   // noise columns of the table, and every event's noise with its logarithm (C library's log)
            if (m->hdp) return SA_EUNSUPPORTED;
            std::vector<double> nz((size_t) m->n_kmers * 3);
            for (long long i = 0; i < m->n_kmers; i++) {
                nz[3 * i] = m->table5[5 * i + 2];
                nz[3 * i + 1] = m->table5[5 * i + 4];
                nz[3 * i + 2] = log(m->table5[5 * i + 4]);
            }
            TRY(upload(&b->d_noise3, nz.data(), (long long) nz.size()));
            std::vector<double> evn((size_t) (2 * (pl->n_ev + 8)), 1.0);
            for (int64_t j = 0; j < n_jobs; j++) {
                const sa_job_t *jb = &jobs[j];
                if (jb->n_events > 0 && jb->event_stride < 2) return SA_EINVAL;   // the noise is the record's second value
                const sa_jobinfo_t *J = &pl->jobs[j];
                for (int64_t i = 0; i < J->n_events; i++) {
                    double n = jb->events[i * jb->event_stride + 1];
                    if (n == 0 && m->emission == SA_EMISSION_TWO_DIST) n = 0.000000001;   // (impl/stateMachine.c:619-621; :659-700 has no such guard)
                    evn[(size_t) (2 * (J->ev_off + i))] = n;
                    evn[(size_t) (2 * (J->ev_off + i) + 1)] = log(n);
                }
            }
            TRY(upload(&b->d_evn, evn.data(), (long long) evn.size()));
            if (!(flags & SA_FLAG_EXACT)) {
                // the register kernels' form of the same numbers (FastT.two_xn_off): per event {n, 1 / n, 1.5 log n, 0}, then per
                // path-space index {(log lambda - log 2 pi) / 2, 1 / noise mean, lambda / 2, 0} of the position's k-mer (zeros for the
                // NULL entry, whose Gaussian part is -inf already)
                const long long ne = pl->n_ev + 8, np_ = pl->n_pid > 0 ? pl->n_pid : 1;
                std::vector<double> two((size_t) (4 * (ne + np_)), 0.0);
                for (long long y = 0; y < ne; y++) {
                    const double n = evn[(size_t) (2 * y)];
                    two[(size_t) (4 * y)] = n; two[(size_t) (4 * y + 1)] = 1.0 / n; two[(size_t) (4 * y + 2)] = 1.5 * evn[(size_t) (2 * y + 1)];
                }
                for (long long i = 0; i < pl->n_pid; i++) {
                    const int id = pl->pid[i];
                    if (id < 0) continue;
                    double *q = &two[(size_t) (4 * (ne + i))];
                    q[0] = 0.5 * (nz[(size_t) (3 * id + 2)] - 1.8378770664093453);
                    q[1] = 1.0 / nz[(size_t) (3 * id)];
                    q[2] = 0.5 * nz[(size_t) (3 * id + 1)];
                }
                b->two_xn_off = ne;
                TRY(upload(&b->d_two, two.data(), (long long) two.size()));
            }
        }
        if (m->hdp) {
            const sa_hdp_t *h = m->hdp;
            std::vector<int> slot((size_t) m->n_kmers);
            for (long long i = 0; i < m->n_kmers; i++) {
                long long r = h->resolved[i];
                slot[i] = (r >= 0 && h->slot[r] >= 0) ? (int) h->slot[r] : -1;
            }
            {   // the row most k-mers resolve to (k_emit_hdp stages it in LDS): worth it from a quarter of the k-mers on
                std::vector<long long> cnt((size_t) (h->n_slots > 0 ? h->n_slots : 1), 0);
                for (long long i = 0; i < m->n_kmers; i++)
                    if (slot[(size_t) i] >= 0) cnt[(size_t) slot[(size_t) i]]++;
                long long best = 0;
                for (long long s_ = 1; s_ < h->n_slots; s_++)
                    if (cnt[(size_t) s_] > cnt[(size_t) best]) best = s_;
                b->hdp_hot = (h->n_slots > 0 && 4 * cnt[(size_t) best] >= m->n_kmers) ? (unsigned) (best * h->grid_length * 16) : 0xffffffffu;
                if (getenv("SA_HDP_HOT") && atoi(getenv("SA_HDP_HOT")) == 0) b->hdp_hot = 0xffffffffu;   // test hook: the flavours without a hot row
            }
            TRY(upload(&b->d_hdp_slot, slot.data(), (long long) slot.size()));
            TRY(upload(&b->d_hdp_y, h->y, h->n_slots * h->grid_length));
            TRY(upload(&b->d_hdp_slope, h->slope, h->n_slots * h->grid_length));
            TRY(upload(&b->d_hdp_grid, h->grid, h->grid_length));
            std::vector<double> tab((size_t) (h->n_slots * h->grid_length * 2));
            for (long long i = 0; i < h->n_slots * h->grid_length; i++) {
                tab[2 * i] = h->y[i];
                tab[2 * i + 1] = h->slope[i];
            }
            TRY(upload(&b->d_hdp_tab, tab.data(), (long long) tab.size()));
            // the same spline as a cubic in the position inside interval i (k_emit_hdp, sa_fast.inc): c0 + c1 t + c2 t^2 + c3 t^3 with
            // the combinations formed in long double; the last entry of a row (no interval to its right) stays zero
            std::vector<double> coef((size_t) (h->n_slots * h->grid_length * 4), 0.0);
            const long double dxl = (long double) h->grid[1] - (long double) h->grid[0];
            for (long long s = 0; s < h->n_slots; s++)
                for (long long i = 0; i + 1 < h->grid_length; i++) {
                    const long long k = s * h->grid_length + i;
                    const long double y0 = h->y[k], y1 = h->y[k + 1], s0 = h->slope[k], s1 = h->slope[k + 1], dy = y1 - y0;
                    coef[4 * k] = (double) y0;
                    coef[4 * k + 1] = (double) (s0 * dxl);
                    coef[4 * k + 2] = (double) (3.0L * dy - (2.0L * s0 + s1) * dxl);
                    coef[4 * k + 3] = (double) ((s0 + s1) * dxl - 2.0L * dy);
                }
            TRY(upload(&b->d_hdp_coef, coef.data(), (long long) coef.size()));
        }
    }
    if ((flags & SA_FLAG_VC_ROWS) && !b->expect) {   // which reference positions the variant-caller output reports on (k_finalize)
        const int kk = m->k;
        b->h_vc_off.assign((size_t) n_jobs + 1, 0);
        for (int64_t j = 0; j < n_jobs; j++) b->h_vc_off[(size_t) j + 1] = b->h_vc_off[(size_t) j] + ((jobs[j].ref_len + 63) / 64 + 1) * 64;
        b->h_vc_bits.assign((size_t) (b->h_vc_off[(size_t) n_jobs] / 64 + 1), 0ull);
        for (int64_t j = 0; j < n_jobs; j++) {
            const char *ref = jobs[j].ref;
            const long long base = b->h_vc_off[(size_t) j];
            long long last_x = -1;   // the last 'X' at or in front of position i + k - 1
            for (long long i = 0; i < kk - 1 && i < jobs[j].ref_len; i++)
                if (ref[i] == 'X') last_x = i;
            for (long long i = 0; i + kk <= jobs[j].ref_len; i++) {
                if (ref[i + kk - 1] == 'X') last_x = i + kk - 1;
                if (last_x >= i) b->h_vc_bits[(size_t) ((base + i) >> 6)] |= 1ull << ((base + i) & 63);
            }
        }
        TRY(upload(&b->d_vc_bits, b->h_vc_bits.data(), (long long) b->h_vc_bits.size()));
        TRY(upload(&b->d_vc_off, b->h_vc_off.data(), (long long) b->h_vc_off.size()));
        if (g_sa_pool.get(SaPool::DEVICE, (void **) &b->d_seg_all, sizeof(long long) * 2 * (size_t) (pl->n_segs > 0 ? pl->n_segs : 1), device) !=
            hipSuccess)
            return SA_ENOMEM;
    }
    {   // emission constants, on the device (same stream as the uploads they read)
        if (g_sa_pool.get(SaPool::DEVICE, (void **) &b->d_xc, sizeof(double) * 4 * (size_t) (pl->n_pid > 0 ? pl->n_pid : 1), device) !=
            hipSuccess) {
            return SA_ENOMEM;
        }
        if (pl->n_regions > 0)
            hipLaunchKernelGGL(k_fill_xc, dim3((unsigned) pl->n_regions), dim3(256), 0, (*UPT).stream, b->d_regions, b->d_poff,
                               b->d_pid, b->d_tab6, m->hdp ? b->d_hdp_slot : (const int *) nullptr,
                               m->hdp ? (long long) m->hdp->grid_length : 0ll, reinterpret_cast<double4 *>(b->d_xc), m->emission);
        if (hipGetLastError() != hipSuccess) return SA_ENODEVICE;
    }
    TRY((*UPT).drain());
    up_lock.unlock();
    if (trace_c) fprintf(stderr, "[trace] create: inputs uploaded at %.1f ms\n", now_ms_c() - tc0);
    TRY(batch_build_lists(b));
    if (trace_c) fprintf(stderr, "[trace] create: launch lists at %.1f ms\n", now_ms_c() - tc0);
    return SA_OK;
#undef TRY
}

static int batch_finish_body(sa_batch *b) {
    const sa_model_t *m = b->c_m;
    const sa_params_t *p = &b->c_p;
    const unsigned flags = b->flags;
    const int device = b->device;
    const bool trace_c = getenv("SA_TRACE") != nullptr;
    auto now_ms_c = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double tc0 = b->c_t0;
    (void) p;
    HIPCHK(hipSetDevice(device));
    SaUploader *const UPT = b->c_deferred ? &g_uploader_tail : &g_uploader;
    struct UseTail { SaUploader *prev; UseTail(SaUploader *u) : prev(tl_uploader) { tl_uploader = u; } ~UseTail() { tl_uploader = prev; } } use_tail_(UPT);
#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
    sa_plan_t *pl = b->plan;
    // Working buffers and launch lists.  What does not fit is planned again: a deferred batch's storage budget dates from its first
    // half -- other batches may have taken the memory since --, SA_FLAG_DEVICE_TO_ITSELF is a promise the caller can break, and
    // candidate / result slots (HDP models, low thresholds) are sized after the budget was set.  When an allocation fails, what
    // this attempt took goes back, the forward storage is re-packed into more passes of half the size at most (regions keep
    // everything else of their plan: only regions[].chunk / f_base change) and the attempt is repeated; SA_ENOMEM only when a
    // single region's planes and the fixed buffers do not fit together.
    double working_bytes = 0.0;
    bool quiet_fail = false;
    auto dalloc = [&](void **p_, long long bytes) -> int {
        const hipError_t e_ = g_sa_pool.get(SaPool::DEVICE, p_, (size_t) (bytes > 0 ? bytes : 8), device);
        if (e_ != hipSuccess) {
            (void) hipGetLastError();
            if (!quiet_fail) fprintf(stderr, "[signalalign_hip] working storage: %lld bytes: %s\n", bytes, hipGetErrorString(e_));
            return e_ == hipErrorOutOfMemory ? SA_ENOMEM : SA_ENODEVICE;
        }
        working_bytes += (double) (bytes > 0 ? bytes : 8);
        return SA_OK;
    };
    // SA_TEST_FAIL_WORKING_ALLOC=n (test hook): an attempt fails as if out of memory while the plan has fewer than n passes
    const int test_min_passes = getenv("SA_TEST_FAIL_WORKING_ALLOC") ? atoi(getenv("SA_TEST_FAIL_WORKING_ALLOC")) : 0;
    auto build_working = [&]() -> int {
        working_bytes = b->d_blk ? (double) b->d_blk_bytes : 0.0;   // (the image of the caller's block stays until the batch goes)
        if (pl->n_chunks < test_min_passes && pl->n_regions > pl->n_chunks) return SA_ENOMEM;
        TRY(dalloc((void **) &b->d_F, 24 * pl->max_chunk_cellpaths));
        // HDP: the emission plane of the register-, ring- and strip-kernel regions (one value per cell-path, laid out like the match plane)
        if (m->hdp && pl->n_fast_regions + pl->n_ring_regions > 0) TRY(dalloc((void **) &b->d_E, 8 * pl->max_chunk_cellpaths));
        TRY(dalloc((void **) &b->d_vbuf, 8 * pl->n_vbuf));
        TRY(dalloc((void **) &b->d_cands, (long long) sizeof(sa_cand_t) * pl->n_cand));
        TRY(dalloc((void **) &b->d_prob, 8 * pl->n_cand));
        b->cand_alloc = pl->n_cand;
        TRY(dalloc((void **) &b->d_cand_count, 4 * pl->n_segs));
        TRY(dalloc((void **) &b->d_seg_pass, 4 * pl->n_segs));
        TRY(dalloc((void **) &b->d_seg_off, 8 * (2 * pl->n_segs + 8)));  // n+1 offsets per group
        TRY(dalloc((void **) &b->d_overflow, 4));
        TRY(dalloc((void **) &b->d_totals, 8 * pl->n_cks));
        TRY(dalloc((void **) &b->d_bscratch, 8 * pl->n_bscratch));
        if (b->expect) {
            TRY(dalloc((void **) &b->d_gsum, 64 * pl->n_cks));
            TRY(dalloc((void **) &b->d_gmc, 8 * pl->n_cks));
        }
        // launch lists (batch_build_lists: with the plan): seam storage, speculative totals, sort keys, events
        {
            const bool host_finalize = (flags & SA_FLAG_EXACT) || b->expect;
            const long long strip_max_n = b->lw_strip_max_n, strip_max_seg = b->lw_strip_max_seg;
            const long long strip_fwd_slots = b->lw_strip_fwd_slots, strip_bwd_slots = b->lw_strip_bwd_slots;
            if (strip_fwd_slots + strip_bwd_slots > 0) {
                // seam storage: per wave two arrays of (diagonals of the longest strip-kernel region / traceback segment + lead-in
                // + sentinels) records; the groups of a pass run side by side, every segment has its own slot behind the forward
                // launch's
                b->seam_cap = (unsigned) (strip_max_n + 16);
                b->seam_cap_bwd = (unsigned) (strip_max_seg + 16);
                b->seam_bwd_off = strip_fwd_slots * 32ll * (long long) b->seam_cap;
                TRY(dalloc((void **) &b->d_seam, b->seam_bwd_off + strip_bwd_slots * 32ll * (long long) b->seam_cap_bwd));
                // side buffer of the (two-pass) backward strip kernel: the two backward gap sums of every checkpoint cell, laid out like vbuf
                TRY(dalloc((void **) &b->d_ckxy, 16ll * (pl->n_vbuf > 0 ? pl->n_vbuf : 1)));
                // the one-pass sweep (default; SA_STRIP_PASSES=2: the two-pass sweep of round 2): speculative totals per segment, sort keys
                // per candidate slot.  Its 64-bit sort key holds 24 bits of diagonals below a traceback's start (de << 40).
                b->strip_one_pass = !(getenv("SA_STRIP_PASSES") && atoi(getenv("SA_STRIP_PASSES")) == 2);
            }
            if ((pl->n_ring_regions + pl->n_fast_regions > 0 && !host_finalize) || (b->expect && pl->n_ring_regions > 0)) {
                // register, ring and (one-pass) strip kernels: candidates against the traceback's speculative total (one per segment)
                TRY(dalloc((void **) &b->d_spec, 8ll * (pl->n_segs > 0 ? pl->n_segs : 1)));
            }
            if (b->strip_one_pass && strip_fwd_slots > 0 && !host_finalize) {   // k_gather_sorted: sort keys per candidate slot
                TRY(dalloc((void **) &b->d_sortkey, 8ll * (pl->n_cand > 0 ? pl->n_cand : 1)));
                TRY(dalloc((void **) &b->d_sortidx, 4ll * (pl->n_cand > 0 ? pl->n_cand : 1)));
            }
            b->gev.resize(4 * b->groups.size(), nullptr);
            b->cev.resize(2 * b->chunks.size(), nullptr);
            for (auto &e : b->gev)
                if (g_handles.event(&e, device) != hipSuccess) return SA_ENODEVICE;
            for (auto &e : b->cev)
                if (g_handles.event(&e, device) != hipSuccess) return SA_ENODEVICE;
            if (g_sa_pool.get(SaPool::PINNED, (void **) &b->h_seg_off, 8 * (size_t) (pl->n_segs + (long long) b->groups.size() + 1),
                              device) != hipSuccess ||
                g_sa_pool.get(SaPool::PINNED, (void **) &b->h_overflow, 64, device) != hipSuccess) {
                return SA_ENOMEM;
            }
            if (!host_finalize) {
                TRY(dalloc((void **) &b->d_out, (long long) sizeof(sa_pair16_t) * pl->n_cand));
                b->out_alloc = pl->n_cand;
            }
        }
        return SA_OK;
    };
    auto release_working = [&]() {
        void **ptrs[] = {(void **) &b->d_F, (void **) &b->d_E, (void **) &b->d_vbuf, (void **) &b->d_cands, (void **) &b->d_prob,
                         (void **) &b->d_cand_count, (void **) &b->d_seg_pass, (void **) &b->d_seg_off, (void **) &b->d_overflow,
                         (void **) &b->d_totals, (void **) &b->d_bscratch, (void **) &b->d_gsum, (void **) &b->d_gmc, (void **) &b->d_seam,
                         (void **) &b->d_ckxy, (void **) &b->d_spec, (void **) &b->d_sortkey, (void **) &b->d_sortidx,
                         (void **) &b->d_out};
        for (void **q : ptrs)
            if (*q) { g_sa_pool.put(SaPool::DEVICE, *q); *q = nullptr; }
        if (b->h_seg_off) { g_sa_pool.put(SaPool::PINNED, b->h_seg_off); b->h_seg_off = nullptr; }
        if (b->h_overflow) { g_sa_pool.put(SaPool::PINNED, b->h_overflow); b->h_overflow = nullptr; }
        for (hipEvent_t e : b->gev) if (e) g_handles.park(e, device);
        for (hipEvent_t e : b->cev) if (e) g_handles.park(e, device);
        b->gev.clear(); b->cev.clear();
        b->seam_cap = 0; b->seam_cap_bwd = 0; b->seam_bwd_off = 0; b->strip_one_pass = false;
    };
    // A batch whose RESULTS take longer to cross PCIe than its kernels take to run (broad HDP densities at a low threshold: hundreds of
    // millions of pairs) ends when its last copy ends, and its first copy cannot start before the forward sweep of its first pass has
    // finished: such a batch sweeps in four passes instead of one, so that the first groups' pairs travel while the later passes
    // compute (5000 HDP reads at threshold 0.01, 8-byte records: 87 -> 76 ms per step; the kernels themselves lose 5 ms to the
    // smaller launches).  The estimate is the pairs-per-event of the last finished batch of this kind (g_pairs_memo).
    if (!(flags & SA_FLAG_EXACT) && !b->expect && pl->n_chunks == 1 && pl->n_regions >= 64 && !getenv("SA_F_BUDGET_CELLPATHS")) {
        const double est_bytes = g_pairs_memo.estimate(pl->model->uid, b->device, pl->params.threshold) * (double) pl->n_ev * (double) b->rec();
        if (est_bytes > 2.0e9) {
            long long total = 0, largest = 1;
            for (long long r = 0; r < pl->n_regions; r++) {
                total += pl->regions[r].f_cellpaths;
                largest = pl->regions[r].f_cellpaths > largest ? pl->regions[r].f_cellpaths : largest;
            }
            sa_plan_repack(pl, (total + 3) / 4 + largest);   // (a pass closes before the region that would overflow it: four at most)
            TRY(batch_build_lists(b));
            if (pl->n_regions > 0 && hipMemcpy(b->d_regions, pl->regions, sizeof(sa_region_t) * (size_t) pl->n_regions, hipMemcpyHostToDevice) != hipSuccess) {
                (void) hipGetLastError();
                return SA_ENODEVICE;
            }
            if (trace_c) fprintf(stderr, "[trace] create: %.1f GB of pairs expected: forward storage in %d passes\n", est_bytes / 1e9, (int) pl->n_chunks);
        }
    }
    {
        int rcw = SA_OK;
        for (int attempt = 0; attempt < 6; attempt++) {
            quiet_fail = true;
            rcw = build_working();
            if (rcw != SA_ENOMEM) break;
            release_working();
            long long largest = 1;
            for (long long r = 0; r < pl->n_regions; r++) largest = pl->regions[r].f_cellpaths > largest ? pl->regions[r].f_cellpaths : largest;
            if (pl->max_chunk_cellpaths <= largest) break;   // one region per pass already: nothing left to give
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); free_b = 0; }
            free_b += g_sa_pool.idle_bytes(SaPool::DEVICE, device);
            long long budget2 = pl->max_chunk_cellpaths / 2;
            const long long by_free = (long long) (0.5 * (double) free_b / (m->hdp ? 32.0 : 24.0));
            if (by_free > 0 && by_free < budget2) budget2 = by_free;
            if (budget2 < largest) budget2 = largest;
            sa_plan_repack(pl, budget2);
            TRY(batch_build_lists(b));   // (the passes changed)
            if (trace_c || !test_min_passes)
                fprintf(stderr, "[signalalign_hip] working storage did not fit: forward storage re-packed into %d passes of at most %.1f GB\n",
                        (int) pl->n_chunks, (m->hdp ? 32.0 : 24.0) * (double) pl->max_chunk_cellpaths / 1e9);
            // the kernels read chunk / f_base from the device copy of the regions
            if (pl->n_regions > 0 && hipMemcpy(b->d_regions, pl->regions, sizeof(sa_region_t) * (size_t) pl->n_regions, hipMemcpyHostToDevice) != hipSuccess) {
                (void) hipGetLastError();
                return SA_ENODEVICE;
            }
        }
        if (rcw == SA_ENOMEM) fprintf(stderr, "[signalalign_hip] working storage does not fit the device\n");
        if (rcw) return rcw;
    }
    {   // the launch lists (small)
        std::lock_guard<std::mutex> g_((*UPT).mu);
        TRY((*UPT).bind(device));
        TRY(upload(&b->d_ids, b->ids_flat.data(), (long long) b->ids_flat.size()));
        TRY((*UPT).drain());
    }
    if (trace_c) fprintf(stderr, "[trace] create: buffers allocated at %.1f ms\n", now_ms_c() - tc0);
    b->stats.cells_forward = pl->cells_fwd;
    b->stats.cells_backward = pl->cells_bwd;
    b->stats.n_regions = pl->n_regions;
    b->stats.n_segments = pl->n_segs;
    b->stats.n_checkpoints = pl->n_cks;
    b->stats.n_fast_regions = pl->n_fast_regions;
    b->stats.n_ring_regions = pl->n_ring_regions;
    b->stats.n_strip_regions = 0;
    for (const auto &C_ : b->chunks) b->stats.n_strip_regions += C_.nst;
    b->stats.n_chunks = pl->n_chunks;
    b->stats.n_groups = (int64_t) b->groups.size();
    double fb = 0;
    for (long long r = 0; r < pl->n_regions; r++)   // (HDP register-kernel regions: 8 B more per cell, the emission plane)
        fb += (m->hdp && pl->regions[r].kind != SA_KIND_GENERIC ? 32.0 : 24.0) * (double) pl->regions[r].f_cellpaths;
    b->stats.f_bytes = fb;
    b->stats.device_bytes = working_bytes;
#undef TRY
    // The pinned result buffer, from an estimate of the result size (measured: 0.9 pairs per event at the default threshold): taken
    // here and not at the start of the run, so that a stream of batches asks the pinned cache for its blocks in the same order in
    // every step.  Taken by the runner thread, the third buffer of a three-deep pipeline was first needed whenever three runs
    // happened to overlap -- sometimes during the caller's warm-up, sometimes in the middle of its timed loop: a 100 ms
    // hipHostMalloc that also held up every other thread's HIP calls (12.5 against 16-19 ms per step, run to run).
    if (!(flags & SA_FLAG_EXACT) && !b->expect && b->h_pairs_cap == 0 && pl->params.threshold >= 0.005) {
        const long long total = (long long) (g_pairs_memo.estimate(m->uid, device, pl->params.threshold) * (double) pl->n_ev) + 4096;
        const long long cap = total + total / 8 + 1024;
        if (g_sa_pool.get(SaPool::PINNED, (void **) &b->h_pairs, b->rec() * (size_t) cap, device) == hipSuccess) b->h_pairs_cap = cap;
        else { (void) hipGetLastError(); b->h_pairs = nullptr; }   // (the run asks again)
    }
    if (trace_c) fprintf(stderr, "[trace] create: done at %.1f ms\n", now_ms_c() - tc0);
    return SA_OK;
}

// Test hook: plans the batch twice -- on the device (sa_dplan.inc) and with sa_plan.c -- and compares every array the kernels
// read, byte for byte.  Returns 0 when all agree, a bit mask of the arrays that differ (1 regions, 2 rows, 4 packed words,
// 8 path offsets, 16 k-mer ids, 32 events, 64 segments, 128 checkpoints, 256 totals, 512 per-path records), 1 << 30 when the batch is not one the
// device planner takes, or a negative SA_E* code.
int sa_dplan_compare(const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs, const char *const *ambig,
                     int device, unsigned flags) {
    if (!m || !p) return SA_EINVAL;
    HIPCHK(hipSetDevice(device));
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    free_b += g_sa_pool.idle_bytes(SaPool::DEVICE, device);
    long long budget = (long long) ((double) free_b * 0.60 / 24.0);
    const char *envb = getenv("SA_F_BUDGET_CELLPATHS");
    if (envb && atoll(envb) > 0) budget = atoll(envb);
    sa_batch *b = new sa_batch();
    b->device = device;
    b->plan = nullptr;
    b->dev_planned = false;
    b->runner = nullptr;
    b->cstream[0] = b->cstream[1] = nullptr; b->xstream[0] = b->xstream[1] = nullptr; b->pair_stream = nullptr; b->stream = nullptr;
    b->h_pairs = nullptr; b->d_pairs_up = nullptr; b->h_seg_off = nullptr; b->h_overflow = nullptr;
    for (int i = 0; i < 8; i++) b->ev[i] = nullptr;
    b->d_regions = nullptr; b->d_rows = nullptr; b->d_pk = nullptr; b->d_poff = nullptr; b->d_pid = nullptr; b->d_px = nullptr; b->d_xc = nullptr;
    b->d_prec = nullptr; b->d_blk = nullptr; b->d_ev = nullptr; b->d_segs = nullptr; b->d_cks = nullptr; b->d_F = nullptr; b->d_E = nullptr; b->d_vbuf = nullptr;
    b->d_cands = nullptr; b->d_cand_count = nullptr; b->d_overflow = nullptr; b->d_totals = nullptr; b->d_bscratch = nullptr;
    b->d_tab6 = nullptr; b->d_hdp_slot = nullptr; b->d_hdp_y = nullptr; b->d_hdp_slope = nullptr; b->d_hdp_grid = nullptr;
    b->d_hdp_tab = nullptr; b->d_hdp_coef = nullptr; b->d_prob = nullptr; b->d_seg_pass = nullptr; b->d_seg_off = nullptr; b->d_out = nullptr;
    b->d_ids = nullptr; b->d_gsum = nullptr; b->d_gmc = nullptr; b->d_seam = nullptr; b->d_ckxy = nullptr;
    b->d_spec = nullptr; b->d_sortkey = nullptr; b->d_sortidx = nullptr;
    int rcd;
    {
        std::unique_lock<std::mutex> lk(g_uploader.mu);
        rcd = g_uploader.bind(device);
        if (rcd == SA_OK) rcd = dplan_build(b, m, p, jobs, n_jobs, ambig, flags, budget);
    }
    if (rcd != SA_OK) {
        sa_batch_destroy(b);
        return rcd < 0 ? rcd : (1 << 30);
    }
    sa_plan_t *hp = nullptr;
    int rc = sa_plan_build(&hp, m, p, jobs, n_jobs, ambig, flags | SA_FLAG_DEVICE_XC_INTERNAL, budget);
    if (rc) { sa_batch_destroy(b); return rc; }
    const sa_plan_t *dp = b->plan;
    int mask = 0;
    auto differs = [&](const void *dev, const void *host, size_t bytes) -> bool {
        if (bytes == 0) return false;
        std::vector<char> tmp(bytes);
        if (hipMemcpy(tmp.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return true;
        return memcmp(tmp.data(), host, bytes) != 0;
    };
    if (dp->n_regions != hp->n_regions || dp->n_segs != hp->n_segs || dp->n_cks != hp->n_cks || dp->n_rows != hp->n_rows ||
        dp->n_pk != hp->n_pk || dp->n_poff != hp->n_poff || dp->n_pid != hp->n_pid || dp->n_ev != hp->n_ev ||
        dp->n_vbuf != hp->n_vbuf || dp->n_cand != hp->n_cand || dp->n_bscratch != hp->n_bscratch ||
        dp->n_chunks != hp->n_chunks || dp->max_chunk_cellpaths != hp->max_chunk_cellpaths ||
        dp->n_fast_regions != hp->n_fast_regions || dp->n_ring_regions != hp->n_ring_regions || dp->cells_fwd != hp->cells_fwd ||
        dp->cells_bwd != hp->cells_bwd)
        mask |= 256;
    if (!(mask & 256)) {
        if (differs(b->d_regions, hp->regions, sizeof(sa_region_t) * (size_t) hp->n_regions) ||
            memcmp(dp->regions, hp->regions, sizeof(sa_region_t) * (size_t) hp->n_regions) != 0)
            mask |= 1;
        if (differs(b->d_rows, hp->rows, sizeof(sa_row_t) * (size_t) hp->n_rows)) mask |= 2;
        if (differs(b->d_pk, hp->pk, 4 * (size_t) hp->n_pk)) mask |= 4;
        if (differs(b->d_poff, hp->poff, 4 * (size_t) hp->n_poff)) mask |= 8;
        if (differs(b->d_pid, hp->pid, 4 * (size_t) hp->n_pid)) mask |= 16;
        if (differs(b->d_ev, hp->ev, 8 * (size_t) hp->n_ev)) mask |= 32;
        if (differs(b->d_segs, hp->segs, sizeof(sa_seg_t) * (size_t) hp->n_segs) ||
            memcmp(dp->segs, hp->segs, sizeof(sa_seg_t) * (size_t) hp->n_segs) != 0)
            mask |= 64;
        if (differs(b->d_cks, hp->cks, sizeof(sa_ck_t) * (size_t) hp->n_cks)) mask |= 128;
        for (int64_t r = 0; r < hp->n_regions; r++) {   // per-path records: they exist for these regions only
            const sa_region_t *R = &hp->regions[r];
            if (R->kind != SA_KIND_RING || R->max_p <= 1) continue;
            if (!b->d_prec || !hp->prec ||
                differs(b->d_prec + R->pid_off, hp->prec + R->pid_off, sizeof(sa_prec_t) * (size_t) hp->poff[R->poff_off + R->lX + 1]))
                mask |= 512;
        }
        for (int64_t j = 0; j < hp->n_jobs; j++)
            if (memcmp(&dp->jobs[j], &hp->jobs[j], sizeof(sa_jobinfo_t)) != 0) mask |= 256;
    }
    sa_plan_free(hp);
    sa_batch_destroy(b);
    return mask;
}

// One pass = per chunk the forward sweeps (stream 0), then per group the backward/posterior kernels, the exact fold
// of its checkpoints and -- with `finalize` -- the on-device finalisation (k_scan also writes the segment offsets
// straight into pinned host memory).  Consecutive groups alternate between two compute streams: the next group's
// waves move in while the previous group's last waves drain, so cutting the traceback work into groups costs no
// idle tail.  Everything is queued up front; `after_group` waits for one group and requests its pairs.  No copy that
// depends on a kernel is ever queued: the copy engine works in order, and a transfer waiting for a kernel would hold
// back the pair copies of groups that are already finished (measured: probes/copy_overlap.hip).
static int submit_group(sa_batch *b, const DevPlan &P, int g, int which_stream, bool finalize) {
    sa_plan_t *pl = b->plan;
    const sa_launch_group &G = b->groups[g];
    hipStream_t st = b->cstream[which_stream];
    HIPCHK(hipEventRecord(b->gev[4 * g], st));
    const size_t relax_lds = sizeof(double) * (size_t) (LA_TAB_DOUBLES + 9 * b->ring_cap);
    if (G.ngs && b->expect)
        hipLaunchKernelGGL((k_bwd_generic<true, false>), dim3(G.ngs), dim3(b->gen_threads), 0, st, P, b->d_ids + G.ids_gs, G.ngs, 0);
    else if (G.ngs && b->relax)
        hipLaunchKernelGGL((k_bwd_generic<false, true>), dim3(G.ngs), dim3(b->gen_threads), relax_lds, st, P, b->d_ids + G.ids_gs, G.ngs,
                           b->ring_cap);
    else if (G.ngs)
        hipLaunchKernelGGL((k_bwd_generic<false, false>), dim3(G.ngs), dim3(b->gen_threads), 0, st, P, b->d_ids + G.ids_gs, G.ngs, 0);
    if (G.nss) {
        StripT ST;
        ST.ev_total = pl->n_ev + 8; ST.seam_cap = b->seam_cap_bwd; ST.seam_stride = 32ull * b->seam_cap_bwd; ST.seam_first = G.seam_first;
        ST.ck_half = pl->n_vbuf;
        ST.spec = b->strip_one_pass ? b->d_spec : nullptr;
        ST.slack = b->spec_slack;
        if (b->strip_one_pass) launch_bwd_strip1(P, b->d_ids + G.ids_ss, G.nss, st, b->d_seam + b->seam_bwd_off, ST);
        else launch_bwd_strip(P, b->d_ids + G.ids_ss, G.nss, st, b->d_seam + b->seam_bwd_off, b->d_ckxy, ST);
    }
    for (int cl = 15; cl >= 0; cl--)   // widest (longest-running) classes first
        if (G.nrs[cl]) launch_bwd_ring(P, b->d_ids + G.ids_rs[cl], G.nrs[cl], st, 64 * ((cl & 7) + 1), cl >= 8, b->expect);
    if (G.nfs) { const int rcl = launch_bwd_fast(P, b->d_ids + G.ids_fs, G.nfs, st, b->expect); if (rcl) return rcl; }
    HIPCHK(hipEventRecord(b->gev[4 * g + 1], st));
    if (G.ck1 > G.ck0)
        hipLaunchKernelGGL(k_fold, dim3((unsigned) ((G.ck1 - G.ck0 + 63) / 64)), dim3(64), 0, st, P, G.ck0, G.ck1);
    if (finalize) {
        const int n = (int) (G.seg1 - G.seg0);
        long long *soff = b->d_seg_off + G.seg0 + g;
        bool any_ring = G.nfs > 0;
        for (int cl = 0; cl < 16; cl++) any_ring = any_ring || G.nrs[cl] > 0;
        // (groups without register / ring / one-pass strip segments: no look at the speculative totals)
        const double *spec = (b->d_spec && (any_ring || (b->strip_one_pass && G.nss > 0))) ? b->d_spec : nullptr;
        hipLaunchKernelGGL(k_finalize, dim3((unsigned) n), dim3(64), 0, st, P, (int) G.seg0, n, b->d_prob, b->d_seg_pass, spec,
                           b->spec_slack, (const unsigned long long *) b->d_vc_bits, (const long long *) b->d_vc_off, b->d_seg_all);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, b->d_seg_pass + G.seg0, soff, b->h_seg_off + G.seg0 + g, n);
        sa_pair16_t *const gout = b->out_at(pl->segs[G.seg0].cand_off);
        hipLaunchKernelGGL(k_gather, dim3((unsigned) n), dim3(64), 0, st, P, (int) G.seg0, n, b->d_prob, soff,
                           gout, spec, (b->strip_on && b->strip_one_pass && b->d_sortkey) ? 1 : 0, b->p8 ? 1 : 0);
        if (spec && G.nss > 0 && b->d_sortkey)
            hipLaunchKernelGGL(k_gather_sorted, dim3((unsigned) n), dim3(64), 0, st, P, (int) G.seg0, n, b->d_prob, soff,
                               gout, spec, b->d_sortkey, b->d_sortidx, b->p8 ? 1 : 0);
        HIPCHK(hipEventRecord(b->gev[4 * g + 2], st));
    } else {
        HIPCHK(hipEventRecord(b->gev[4 * g + 2], st));
    }
    return SA_OK;
}

template <typename AfterGroup>
static int enqueue_pass(sa_batch *b, bool finalize, AfterGroup after_group) {
    sa_plan_t *pl = b->plan;
    DevPlan P = make_devplan(b);
    hipStream_t s0 = b->cstream[0], s1 = b->cstream[1];
    HIPCHK(hipMemsetAsync(b->d_cand_count, 0, 4 * (size_t) (pl->n_segs > 0 ? pl->n_segs : 1), s0));
    if (b->d_spec)   // all bits set = NaN: "not a segment of the ring / strip kernels" until their forward sweep says otherwise
        HIPCHK(hipMemsetAsync(b->d_spec, 0xff, 8 * (size_t) (pl->n_segs > 0 ? pl->n_segs : 1), s0));
    b->h_overflow[0] = 0;  // pinned host word the kernels raise directly
    b->h_overflow[1] = 0;  // ... and the one k_finalize raises when a speculative candidate bound turns out too high
    b->h_overflow[2] = 0;  // ... and k_check_events' / k_spec_match's: an event mean or a forward value that is not a number
    if (pl->n_ev > 0)
        hipLaunchKernelGGL(k_check_events, dim3((unsigned) std::min<long long>((pl->n_ev + 255) / 256, 2048)), dim3(256), 0, s0,
                           (const double *) b->d_ev, (long long) pl->n_ev, b->h_overflow);
    HIPCHK(hipEventRecord(b->ev[0], s0));
    for (size_t c = 0; c < b->chunks.size(); c++) {
        const sa_launch_chunk &C = b->chunks[c];
        HIPCHK(hipEventRecord(b->cev[2 * c], s0));
        if (C.ngr && b->relax)
            hipLaunchKernelGGL(k_fwd_generic<true>, dim3(C.ngr), dim3(b->gen_threads), sizeof(double) * (size_t) (LA_TAB_DOUBLES + 9 * b->ring_cap),
                               s0, P, b->d_ids + C.ids_gr, C.ngr, b->ring_cap);
        else if (C.ngr)
            hipLaunchKernelGGL(k_fwd_generic<false>, dim3(C.ngr), dim3(b->gen_threads), 0, s0, P, b->d_ids + C.ids_gr, C.ngr, 0);
        {   // ring-kernel regions, one launch per class of row capacity.  A forward launch holds one workgroup per read and
            // lasts as long as its longest read's serial chain, so launches that follow each other on one stream leave the chip
            // mostly empty three times over: the classes alternate between the two compute streams and run side by side
            int n_cl = C.nst > 0;
            for (int cl = 0; cl < 16; cl++) n_cl += C.nrr[cl] > 0;
            hipStream_t lanes[4] = {s0, s1, b->xstream[0], b->xstream[1]};
            int n_lanes = n_cl < 4 ? n_cl : 4;
            for (int q = 2; q < n_lanes; q++)   // the two extra streams are made on first need
                if (!lanes[q]) {
                    if (g_handles.stream(&b->xstream[q - 2], b->device, 0) != hipSuccess) { n_lanes = q; break; }
                    lanes[q] = b->xstream[q - 2];
                }
            if (P.m.hdp) {   // the emission plane of this pass's ring / strip regions, ahead of the sweeps that read it (on s0: the
                             // other lanes wait for the event recorded below)
                if (C.nst) launch_emit_hdp_ring(P, b->d_ids + C.ids_st, C.nst, pl->regions[b->ids_flat[(size_t) C.ids_st]].N, s0, false);
                for (int cl = 0; cl < 16; cl++)
                    if (C.nrr[cl])
                        launch_emit_hdp_ring(P, b->d_ids + C.ids_rr[cl], C.nrr[cl], pl->regions[b->ids_flat[(size_t) C.ids_rr[cl]]].N, s0, cl >= 8);
            }
            if (n_lanes > 1) {
                HIPCHK(hipEventRecord(b->ev[1], s0));
                for (int q = 1; q < n_lanes; q++) HIPCHK(hipStreamWaitEvent(lanes[q], b->ev[1], 0));
            }
            int which = 0;
            if (C.nst) {
                StripT ST;
                ST.ev_total = pl->n_ev + 8; ST.seam_cap = b->seam_cap; ST.seam_stride = 32ull * b->seam_cap; ST.seam_first = 0;
                ST.spec = b->strip_one_pass ? b->d_spec : nullptr;
                ST.slack = b->spec_slack;
                launch_fwd_strip(P, b->d_ids + C.ids_st, C.nst, lanes[0], b->d_seam, ST);
                which = n_lanes > 1 ? 1 : 0;
            }
            for (int cl = 15; cl >= 0; cl--)
                if (C.nrr[cl]) {
                    launch_fwd_ring(P, b->d_ids + C.ids_rr[cl], C.nrr[cl], lanes[n_lanes > 1 ? which : 0], 64 * ((cl & 7) + 1), cl >= 8);
                    which = (which + 1) % (n_lanes > 1 ? n_lanes : 1);
                }
            for (int q = 1; q < n_lanes; q++) {
                HIPCHK(hipEventRecord(b->ev[1 + q], lanes[q]));
                HIPCHK(hipStreamWaitEvent(s0, b->ev[1 + q], 0));
            }
        }
        // (Round 5, measured and dropped: the pass's HDP regions in 2 / 4 / 8 slices on the two compute streams alternately, so that the
        // forward sweep of slice k runs beside the emission kernel of slice k + 1 -- neither keeps the chip busy alone --: forward
        // stage 15.8 -> 18.6 / 17.1 / 20.0 ms per 5000 reads.  A forward launch of fewer reads lasts as long as its longest chain and
        // the emission kernel slows down beside it by more than the overlap gives.)
        if (C.nfr && P.m.hdp) launch_emit_hdp(P, b->d_ids + C.ids_fr, C.nfr, pl->regions[b->ids_flat[(size_t) C.ids_fr]].N, s0);
        if (C.nfr) launch_fwd_fast(P, b->d_ids + C.ids_fr, C.nfr, s0, b->wide_cap);
        if (b->d_spec && C.g1 > C.g0) {   // the candidate bounds of this pass's ring / strip tracebacks (k_spec_match)
            bool any = C.nst > 0 || C.nfr > 0;
            for (int cl = 0; cl < 16; cl++) any = any || C.nrr[cl] > 0;
            const long long sa_ = b->groups[(size_t) C.g0].seg0, sb_ = b->groups[(size_t) C.g1 - 1].seg1;
            if (any && sb_ > sa_)
                hipLaunchKernelGGL(k_spec_match, dim3((unsigned) (sb_ - sa_)), dim3(64), 0, s0, P, (int) sa_, (int) (sb_ - sa_), b->d_spec);
        }
        HIPCHK(hipEventRecord(b->cev[2 * c + 1], s0));
        if (C.g1 - C.g0 > 1) HIPCHK(hipStreamWaitEvent(s1, b->cev[2 * c + 1], 0));
        int submitted = C.g0, completed = C.g0;
        while (completed < C.g1) {
            while (submitted < C.g1) {
                int rc = submit_group(b, P, submitted, (submitted - C.g0) & 1, finalize);
                if (rc) return rc;
                submitted++;
            }
            int rcg = after_group((size_t) completed);
            if (rcg) return rcg;
            completed++;
        }
        // the next chunk's forward sweep reuses the forward storage: both streams must be done with it
        for (int g = C.g0; g < C.g1; g++)
            if ((g - C.g0) & 1) HIPCHK(hipStreamWaitEvent(s0, b->gev[4 * g + 2], 0));
    }
    HIPCHK(hipEventRecord(b->ev[5], s0));
    HIPCHK(hipGetLastError());
    return SA_OK;
}

// after the stream has drained: kernel times from the events
static int collect_times(sa_batch *b) {
    float ms_f = 0, ms_b = 0, ms_tot = 0, t = 0;
    for (size_t c = 0; c < b->chunks.size(); c++) {
        const sa_launch_chunk &C = b->chunks[c];
        HIPCHK(hipEventElapsedTime(&t, b->cev[2 * c], b->cev[2 * c + 1]));
        ms_f += t;
        // backward stage of the chunk: from the end of its forward sweep to the last backward kernel's end (the
        // groups' kernels overlap on two streams; finalisation kernels of earlier groups run inside this window)
        float last = 0;
        for (int g = C.g0; g < C.g1; g++) {
            HIPCHK(hipEventElapsedTime(&t, b->cev[2 * c + 1], b->gev[4 * g + 1]));
            last = t > last ? t : last;
        }
        ms_b += last;
    }
    HIPCHK(hipEventElapsedTime(&ms_tot, b->ev[0], b->ev[5]));
    b->stats.ms_forward = ms_f;
    b->stats.ms_backward = ms_b;
    b->stats.ms_fold = ms_tot - ms_f - ms_b;  // what follows the last backward kernel: fold (+ finalisation) of the last group
    b->stats.ms_total_device = ms_tot;
    return SA_OK;
}

static int grow_after_overflow(sa_batch *b) {
    sa_plan_t *pl = b->plan;
    // a traceback segment produced more candidates than planned: enlarge and redo the pass
    sa_plan_grow_candidates(pl, 4);
    b->cand_factor = (b->cand_factor > 0 ? b->cand_factor : 1) * 4;
    cand_memo_note(pl->model, pl->params.threshold, b->device, b->cand_factor);
    g_sa_pool.put(SaPool::DEVICE, b->d_cands);
    g_sa_pool.put(SaPool::DEVICE, b->d_prob);
    b->d_cands = nullptr;
    b->d_prob = nullptr;
    HIPCHK(g_sa_pool.get(SaPool::DEVICE, (void **) &b->d_cands, sizeof(sa_cand_t) * (size_t) pl->n_cand, b->device));
    HIPCHK(g_sa_pool.get(SaPool::DEVICE, (void **) &b->d_prob, 8 * (size_t) pl->n_cand, b->device));
    b->cand_alloc = pl->n_cand;
    if (b->d_out) {
        g_sa_pool.put(SaPool::DEVICE, b->d_out);
        b->d_out = nullptr;
        HIPCHK(g_sa_pool.get(SaPool::DEVICE, (void **) &b->d_out, sizeof(sa_pair16_t) * (size_t) pl->n_cand, b->device));
        b->out_alloc = pl->n_cand;
    }
    if (b->d_sortkey) {
        g_sa_pool.put(SaPool::DEVICE, b->d_sortkey);
        g_sa_pool.put(SaPool::DEVICE, b->d_sortidx);
        b->d_sortkey = nullptr; b->d_sortidx = nullptr;
        HIPCHK(g_sa_pool.get(SaPool::DEVICE, (void **) &b->d_sortkey, 8 * (size_t) pl->n_cand, b->device));
        HIPCHK(g_sa_pool.get(SaPool::DEVICE, (void **) &b->d_sortidx, 4 * (size_t) pl->n_cand, b->device));
    }
    HIPCHK(hipMemcpy(b->d_segs, pl->segs, sizeof(sa_seg_t) * (size_t) pl->n_segs, hipMemcpyHostToDevice));
    return SA_OK;
}

// kernels only (host finalisation follows): SA_FLAG_EXACT and the expectation pass
static int run_passes(sa_batch_t *b) {
    for (int attempt = 0; attempt < 6; attempt++) {
        int rc = enqueue_pass(b, false, [](size_t) { return (int) SA_OK; });
        if (rc) return rc;
        HIPCHK(sa_sync_stream(b->cstream[1], b->device));
        HIPCHK(sa_sync_stream(b->cstream[0], b->device));
        rc = collect_times(b);
        if (rc) return rc;
        if (b->h_overflow[2]) return SA_EINVAL;   // (k_spec_match: a forward value that is not a number)
        if (!b->h_overflow[0]) return SA_OK;
        rc = grow_after_overflow(b);
        if (rc) return rc;
    }
    return SA_ENOMEM;
}

static int batch_run_body(sa_batch_t *b);
int sa_batch_run(sa_batch_t *b) {
    if (!b) return SA_EINVAL;
    if (b->released) return SA_ESTATE;   // (sa_batch_release_device: nothing left to run on)
    { const int rcf = batch_finish(b); if (rcf) return rcf; }   // (a deferred batch: the second half of its creation)
    b->quiet = false;
    const int rc = batch_run_body(b);
    b->quiet = rc == SA_OK;
    return rc;
}
static int batch_run_body(sa_batch_t *b) {
    if (b->expect) return SA_ESTATE;
    HIPCHK(hipSetDevice(b->device));
    sa_plan_t *pl = b->plan;
    long long n_segs = pl->n_segs;
    b->n_pairs_total = 0;
    b->job_off.assign((size_t) pl->n_jobs + 1, 0);
    auto reserve_pairs = [&](long long total) -> int {
        if (total > b->h_pairs_cap) {
            g_sa_pool.put(SaPool::PINNED, b->h_pairs);
            b->h_pairs = nullptr;
            long long cap = total + total / 8 + 1024;
            HIPCHK(g_sa_pool.get(SaPool::PINNED, (void **) &b->h_pairs, b->rec() * (size_t) cap, b->device));
            b->h_pairs_cap = cap;
        }
        return SA_OK;
    };
    // A first estimate of the result size (measured: 0.9 pairs per event at the default threshold) lets even the FIRST run
    // of a batch overlap its copies with the kernels; with the caching allocator the buffer is a reused block.  If the
    // estimate is short the run falls back to copying afterwards, as before.
    if (!(b->flags & SA_FLAG_EXACT) && b->h_pairs_cap == 0 && pl->params.threshold >= 0.005) {
        // (an estimate: when the pinned block cannot be had at that size the run copies after its kernels, exactly as with no estimate)
        if (reserve_pairs((long long) (g_pairs_memo.estimate(pl->model->uid, b->device, pl->params.threshold) * (double) pl->n_ev) + 4096) != SA_OK) {
            (void) hipGetLastError();
            b->h_pairs = nullptr;
            b->h_pairs_cap = 0;
        }
    }
    if (b->flags & SA_FLAG_EXACT) {
        int rcp0 = run_passes(b);
        if (rcp0) return rcp0;
        // host finalisation with the C library's exp(): bit-identical to the reference's posterior arithmetic
        std::vector<sa_cand_t> cands((size_t) (pl->n_cand > 0 ? pl->n_cand : 1));
        std::vector<int> counts((size_t) (n_segs > 0 ? n_segs : 1));
        std::vector<double> totals((size_t) (pl->n_cks > 0 ? pl->n_cks : 1));
        if (pl->n_cand) HIPCHK(hipMemcpy(cands.data(), b->d_cands, sizeof(sa_cand_t) * (size_t) pl->n_cand, hipMemcpyDeviceToHost));
        if (n_segs) HIPCHK(hipMemcpy(counts.data(), b->d_cand_count, 4 * (size_t) n_segs, hipMemcpyDeviceToHost));
        if (pl->n_cks) HIPCHK(hipMemcpy(totals.data(), b->d_totals, 8 * (size_t) pl->n_cks, hipMemcpyDeviceToHost));
        std::vector<sa_pair_t *> pp((size_t) (pl->n_jobs > 0 ? pl->n_jobs : 1), nullptr);
        std::vector<int64_t> np((size_t) (pl->n_jobs > 0 ? pl->n_jobs : 1), 0);
        int rc = sa_plan_finalize(pl, cands.data(), counts.data(), totals.data(), pp.data(), np.data());
        if (rc) return rc;
        if (!b->h_vc_bits.empty()) {   // SA_FLAG_VC_ROWS on host-finalised pairs: the same test as k_finalize's
            b->job_all_n.assign((size_t) pl->n_jobs, 0);
            b->job_all_sum.assign((size_t) pl->n_jobs, 0);
            for (long long j = 0; j < pl->n_jobs; j++) {
                const long long base = b->h_vc_off[(size_t) j];
                int64_t kept = 0;
                for (int64_t q = 0; q < np[j]; q++) {
                    b->job_all_n[(size_t) j]++;
                    b->job_all_sum[(size_t) j] += pp[j][q].prob_e7;
                    const long long bit = base + pp[j][q].x;
                    if ((b->h_vc_bits[(size_t) (bit >> 6)] >> (bit & 63)) & 1ull) pp[j][kept++] = pp[j][q];
                }
                np[j] = kept;
            }
        }
        long long total = 0;
        for (long long j = 0; j < pl->n_jobs; j++) total += np[j];
        rc = reserve_pairs(total);
        if (rc) return rc;
        total = 0;
        for (long long j = 0; j < pl->n_jobs; j++) {
            b->job_off[j] = total;
            for (int64_t q = 0; q < np[j]; q++) {
                if (b->p8) reinterpret_cast<sa_pair8_t *>(b->h_pairs)[total + q] = sa_pair8_pack(pp[j][q].prob_e7, pp[j][q].x, pp[j][q].y);
                else b->h_pairs[total + q] = sa_pair16_pack(pp[j][q].prob_e7, pp[j][q].x, pp[j][q].y, pp[j][q].path, pp[j][q].kmer_id);
            }
            total += np[j];
            free(pp[j]);
        }
        b->job_off[pl->n_jobs] = total;
        b->n_pairs_total = total;
        b->job_dev_off.clear();
        b->ran = true;
        return SA_OK;
    }
    // default: finalisation on the device, group after group; the pairs of group g travel to the pinned host buffer
    // on the copy stream while the kernels of group g+1 run
    const size_t ng = b->groups.size();
    std::vector<long long> gbase(ng + 1, 0);
    bool done = false;
    const bool trace = getenv("SA_TRACE") != nullptr;
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    for (int attempt = 0; attempt < 6 && !done; attempt++) {
        double t0 = now_ms();
        bool piped = true;
        long long running = 0;
        auto after_group = [&](size_t g) -> int {
            const sa_launch_group &G = b->groups[g];
            HIPCHK(hipEventSynchronize(b->gev[4 * g + 2]));
            long long tg = b->h_seg_off[G.seg0 + g + (G.seg1 - G.seg0)];
            if (trace) fprintf(stderr, "[trace] group %zu ready at %.3f ms, %lld pairs\n", g, now_ms() - t0, tg);
            gbase[g] = running;
            if (piped && running + tg <= b->h_pairs_cap) {
                if (tg > 0)
                    HIPCHK(hipMemcpyAsync(b->host_at(running), b->out_at(pl->segs[G.seg0].cand_off),
                                          b->rec() * (size_t) tg, hipMemcpyDeviceToHost, b->pair_stream));
            } else {
                piped = false;  // first run (or a larger result than last time): size the pinned buffer afterwards
            }
            running += tg;
            return SA_OK;
        };
        int rc = enqueue_pass(b, true, after_group);
        if (rc) return rc;
        gbase[ng] = running;
        HIPCHK(sa_sync_stream(b->cstream[1], b->device));
        HIPCHK(sa_sync_stream(b->cstream[0], b->device));
        if (trace) fprintf(stderr, "[trace] compute stream drained at %.3f ms\n", now_ms() - t0);
        HIPCHK(sa_sync_stream(b->pair_stream, b->device));
        if (trace) fprintf(stderr, "[trace] copies drained at %.3f ms (piped %d)\n", now_ms() - t0, (int) piped);
        rc = collect_times(b);
        if (rc) return rc;
        if (b->d_spec && getenv("SA_SPEC_DEBUG")) {
            // diagnostic: how far the exact totals of a traceback lie from its speculative total (expected: ~1e-3)
            std::vector<double> sp((size_t) n_segs), tt((size_t) (pl->n_cks > 0 ? pl->n_cks : 1));
            HIPCHK(hipMemcpy(sp.data(), b->d_spec, 8 * (size_t) n_segs, hipMemcpyDeviceToHost));
            if (pl->n_cks) HIPCHK(hipMemcpy(tt.data(), b->d_totals, 8 * (size_t) pl->n_cks, hipMemcpyDeviceToHost));
            double worst_lo = 0, worst_hi = 0;
            long long n_spec = 0, shown = 0;
            for (long long sg = 0; sg < n_segs; sg++) {
                if (!(sp[(size_t) sg] == sp[(size_t) sg]) || !(sp[(size_t) sg] > -INFINITY)) continue;
                n_spec++;
                const sa_seg_t *S = &pl->segs[sg];
                for (int c = 0; c < S->n_ck; c++) {
                    const double dlt = tt[(size_t) (S->ck_base + c)] - sp[(size_t) sg];
                    if (dlt < worst_lo) worst_lo = dlt;
                    if (dlt > worst_hi) worst_hi = dlt;
                    if (dlt < -b->spec_slack && shown < 6) {
                        shown++;
                        fprintf(stderr, "[spec] segment %lld (region %d, start %lld from %lld to %lld at_end %d) checkpoint %d of %d: total %.6f spec %.6f\n",
                                sg, S->region, (long long) S->start, (long long) S->from, (long long) S->to, S->at_end, c, S->n_ck,
                                tt[(size_t) (S->ck_base + c)], sp[(size_t) sg]);
                    }
                }
            }
            fprintf(stderr, "[spec] %lld segments with a speculative total; exact total - speculative total in [%.3e, %.3e]; slack %.3g\n",
                    n_spec, worst_lo, worst_hi, b->spec_slack);
        }
        if (b->h_overflow[2]) {
            fprintf(stderr, "[signalalign_hip] an event mean (or a model entry: a traceback's forward values) is not a finite number: no result\n");
            return SA_EINVAL;
        }
        if (b->h_overflow[1] && b->d_spec) {
            // a traceback's exact totals fell below its speculative total minus the slack (not seen with the default slack: the
            // totals of one traceback agree to ~1e-2, the flat HDP fixture's to 0.15; tests/test_gpu_parity.py forces it with
            // SA_TEST_SPEC_SLACK): its candidates may be incomplete -- the pass is repeated with a bound far lower (more candidates,
            // same survivors): x4, then x256 more, and if that is not enough with no bound at all (slack +inf: every posterior
            // cell-path is a candidate and k_finalize's check cannot fire again) -- never accepted as it is
            if (std::isinf(b->spec_slack)) return SA_ESTATE;   // (cannot happen: with an infinite slack the flag is never raised)
            b->spec_repeats++;
            b->spec_slack = b->spec_repeats == 1 ? b->spec_slack * 4.0 : (b->spec_repeats == 2 ? b->spec_slack * 256.0 : (double) INFINITY);
            if (!getenv("SA_TEST_SPEC_SLACK")) spec_memo_note(pl->model, b->device, b->spec_slack);
            fprintf(stderr, "[signalalign_hip] a speculative candidate bound was too high; repeating the pass with slack %g\n", b->spec_slack);
            continue;
        }
        if (b->h_overflow[0]) {
            rc = grow_after_overflow(b);
            if (rc) return rc;
            continue;
        }
        if (!piped) {
            rc = reserve_pairs(running);
            if (rc) return rc;
            for (size_t g = 0; g < ng; g++) {
                const sa_launch_group &G = b->groups[g];
                long long tg = gbase[g + 1] - gbase[g];
                if (tg > 0)
                    HIPCHK(hipMemcpyAsync(b->host_at(gbase[g]), b->out_at(pl->segs[G.seg0].cand_off),
                                          b->rec() * (size_t) tg, hipMemcpyDeviceToHost, b->pair_stream));
            }
            HIPCHK(sa_sync_stream(b->pair_stream, b->device));
        }
        done = true;
    }
    if (!done) return SA_ENOMEM;
    // job offsets: a job's pairs start where its first segment's do
    {
        std::vector<int> seg_group((size_t) (n_segs > 0 ? n_segs : 1), 0);
        for (size_t g = 0; g < ng; g++)
            for (long long sg = b->groups[g].seg0; sg < b->groups[g].seg1; sg++) seg_group[sg] = (int) g;
        long long next = gbase[ng];
        b->job_dev_off.assign((size_t) pl->n_jobs, 0);
        for (long long j = pl->n_jobs - 1; j >= 0; j--) {
            const sa_jobinfo_t *J = &pl->jobs[j];
            long long first_seg = -1;
            for (long long r = J->region_off; r < J->region_off + J->n_regions && first_seg < 0; r++)
                if (pl->regions[r].n_seg > 0) first_seg = pl->regions[r].seg_off;
            if (first_seg >= 0) {
                int g = seg_group[first_seg];
                next = gbase[g] + b->h_seg_off[first_seg + g];
                // the group's pairs sit in d_out from its first segment's candidate slot on
                b->job_dev_off[j] = pl->segs[b->groups[g].seg0].cand_off + b->h_seg_off[first_seg + g];
            }
            b->job_off[j] = next;  // jobs without segments are empty ranges in front of the next job
        }
        b->job_off[pl->n_jobs] = gbase[ng];
        b->n_pairs_total = gbase[ng];
    }
    if (b->d_seg_all) {   // SA_FLAG_VC_ROWS: what the dropped rows would have added to a job's count and score
        std::vector<long long> sa_((size_t) (2 * (n_segs > 0 ? n_segs : 1)), 0);
        if (n_segs) HIPCHK(hipMemcpy(sa_.data(), b->d_seg_all, sizeof(long long) * 2 * (size_t) n_segs, hipMemcpyDeviceToHost));
        b->job_all_n.assign((size_t) pl->n_jobs, 0);
        b->job_all_sum.assign((size_t) pl->n_jobs, 0);
        for (long long sg = 0; sg < n_segs; sg++) {
            const long long j = pl->regions[pl->segs[sg].region].job;
            b->job_all_n[(size_t) j] += sa_[(size_t) (2 * sg)];
            b->job_all_sum[(size_t) j] += sa_[(size_t) (2 * sg + 1)];
        }
    }
    g_pairs_memo.note(pl->model->uid, b->device, pl->params.threshold, (double) b->n_pairs_total, (double) pl->n_ev);
    b->ran = true;
    return SA_OK;
}

// Device-side view of the results for a downstream device step (sa_mea.hip): per job the first pair in *pairs and the
// number of pairs and of events.  After host finalisation (SA_FLAG_EXACT) the pairs are uploaded once.
int sa_batch_device_view(sa_batch_t *b, const sa_pair16_t **pairs, std::vector<long long> *first, std::vector<long long> *count,
                         std::vector<long long> *n_events, int *device) {
    if (!b || !pairs || !first || !count || !n_events || !device) return SA_EINVAL;
    // (8-byte records name neither path nor k-mer, and a batch filtered for the variant-caller output holds only the rows of X positions:
    // nothing a downstream device step -- the MEA path over ALL posteriors -- may read)
    if (!b->ran || b->p8 || (b->flags & SA_FLAG_VC_ROWS)) return SA_ESTATE;
    const sa_plan_t *pl = b->plan;
    const size_t nj = (size_t) pl->n_jobs;
    first->assign(nj, 0); count->assign(nj, 0); n_events->assign(nj, 0);
    for (size_t j = 0; j < nj; j++) {
        (*count)[j] = b->job_off[j + 1] - b->job_off[j];
        (*n_events)[j] = pl->jobs[j].n_events;
    }
    *device = b->device;
    HIPCHK(hipSetDevice(b->device));
    if (b->job_dev_off.size() == nj && !b->released) {
        *pairs = b->d_out;
        for (size_t j = 0; j < nj; j++) (*first)[j] = b->job_dev_off[j];
        return SA_OK;
    }
    if (b->n_pairs_total > b->d_pairs_up_cap) {
        g_sa_pool.put(SaPool::DEVICE, b->d_pairs_up);
        b->d_pairs_up = nullptr; b->d_pairs_up_cap = 0;
        HIPCHK(g_sa_pool.get(SaPool::DEVICE, (void **) &b->d_pairs_up, sizeof(sa_pair16_t) * (size_t) b->n_pairs_total, b->device));
        b->d_pairs_up_cap = b->n_pairs_total;
    }
    if (b->n_pairs_total)
        HIPCHK(hipMemcpy(b->d_pairs_up, b->h_pairs, sizeof(sa_pair16_t) * (size_t) b->n_pairs_total, hipMemcpyHostToDevice));
    *pairs = b->d_pairs_up;
    for (size_t j = 0; j < nj; j++) (*first)[j] = b->job_off[j];
    return SA_OK;
}

// sa_batch_run on a thread of the library's own, so that the caller can plan the next batch (sa_batch_create is host
// work) while this one is on the GPU; sa_batch_wait joins it and returns sa_batch_run's code.
int sa_batch_prepare(sa_batch_t *b) {
    if (!b) return SA_EINVAL;
    std::lock_guard<std::mutex> g(b->fin_mu);
    if (!b->finished && !b->prepared) {
        b->prepare_rc = batch_prepare_body(b);
        b->prepared = true;
    }
    return b->prepare_rc;
}
// Returns a finished batch's working storage in HBM (forward planes, candidate and result slots, plan arrays, seams: everything
// sa_batch_stats_t.device_bytes counts) to the caching allocator and keeps what the caller reads: the packed pairs in pinned host
// memory, the per-job offsets, the statistics.  For a caller that holds batches for their RESULTS while it creates further ones
// (signalMachine --twoD: the template batch while the complement batch runs; a render thread formatting the previous slice): the
// next batch then plans into the whole card.  Afterwards sa_batch_run / sa_batch_start return SA_ESTATE; sa_batch_mea still works
// (the pairs go up again, as after SA_FLAG_EXACT).
int sa_batch_release_device(sa_batch_t *b) {
    if (!b) return SA_EINVAL;
    if (!b->ran || b->runner) return SA_ESTATE;   // (between sa_batch_start and sa_batch_wait: the run is still using it)
    if (b->released) return SA_OK;
    if (hipSetDevice(b->device) != hipSuccess) return SA_ENODEVICE;
    for (int i = 0; i < 2; i++) {   // (a finished run has drained them; a failed one may not have)
        if (b->cstream[i]) HIPCHK(sa_sync_stream(b->cstream[i], b->device));
        if (b->xstream[i]) HIPCHK(sa_sync_stream(b->xstream[i], b->device));
    }
    if (b->pair_stream) HIPCHK(sa_sync_stream(b->pair_stream, b->device));
    void **ptrs[] = {(void **) &b->d_regions, (void **) &b->d_rows, (void **) &b->d_pk, (void **) &b->d_poff, (void **) &b->d_pid, (void **) &b->d_px,
                     (void **) &b->d_xc, (void **) &b->d_prec, (void **) &b->d_ev, (void **) &b->d_segs, (void **) &b->d_cks, (void **) &b->d_F,
                     (void **) &b->d_E, (void **) &b->d_vbuf, (void **) &b->d_cands, (void **) &b->d_cand_count, (void **) &b->d_overflow,
                     (void **) &b->d_totals, (void **) &b->d_bscratch, (void **) &b->d_tab6, (void **) &b->d_noise3, (void **) &b->d_evn, (void **) &b->d_two,
                     (void **) &b->d_hdp_slot, (void **) &b->d_hdp_y, (void **) &b->d_hdp_slope, (void **) &b->d_hdp_grid, (void **) &b->d_hdp_tab, (void **) &b->d_hdp_coef,
                     (void **) &b->d_prob, (void **) &b->d_seg_pass, (void **) &b->d_seg_off, (void **) &b->d_out, (void **) &b->d_ids,
                     (void **) &b->d_gsum, (void **) &b->d_gmc, (void **) &b->d_seam, (void **) &b->d_ckxy, (void **) &b->d_blk, (void **) &b->d_spec,
                     (void **) &b->d_sortkey, (void **) &b->d_sortidx, (void **) &b->d_vc_bits, (void **) &b->d_vc_off,
                     (void **) &b->d_seg_all};
    for (void **pp : ptrs)
        if (*pp) { g_sa_pool.put(SaPool::DEVICE, *pp); *pp = nullptr; }
    if (b->held_stage) { g_sa_pool.put(SaPool::PINNED, b->held_stage); b->held_stage = nullptr; }
    b->released = true;
    return SA_OK;
}

int sa_batch_start(sa_batch_t *b) {
    if (!b) return SA_EINVAL;
    if (b->runner || b->released) return SA_ESTATE;   // (sa_batch_release_device: nothing left to run on)
    b->runner_rc = SA_OK;
    g_batches_started.fetch_add(1);
    b->runner = new (std::nothrow) std::thread([b]() { b->runner_rc = sa_batch_run(b); });
    if (!b->runner) g_batches_started.fetch_sub(1);
    return b->runner ? SA_OK : SA_ENOMEM;
}
int sa_batch_wait(sa_batch_t *b) {
    if (!b) return SA_EINVAL;
    if (!b->runner) return SA_ESTATE;
    b->runner->join();
    delete b->runner;
    b->runner = nullptr;
    g_batches_started.fetch_sub(1);
    return b->runner_rc;
}

int sa_batch_n_pairs(const sa_batch_t *b, int64_t job, int64_t *n) {
    if (!b || !n || job < 0 || job >= b->c_n) return SA_EINVAL;
    if (!b->ran) return SA_ESTATE;
    *n = b->job_off[job + 1] - b->job_off[job];
    return SA_OK;
}
int sa_batch_all_pairs_summary(const sa_batch_t *b, int64_t job, int64_t *n_all, int64_t *sum_prob_e7) {
    if (!b || job < 0 || job >= b->c_n) return SA_EINVAL;
    if (!b->ran) return SA_ESTATE;
    if (b->job_all_n.size() == (size_t) b->c_n) {   // SA_FLAG_VC_ROWS: counted before the rows were dropped
        if (n_all) *n_all = b->job_all_n[(size_t) job];
        if (sum_prob_e7) *sum_prob_e7 = b->job_all_sum[(size_t) job];
        return SA_OK;
    }
    long long s = 0;
    for (long long i = b->job_off[job]; i < b->job_off[job + 1]; i++)
        s += b->p8 ? (long long) (reinterpret_cast<const sa_pair8_t *>(b->h_pairs)[i] >> 40) : (long long) ((b->h_pairs[i].b >> 32) & 0xffffffull);
    if (n_all) *n_all = b->job_off[job + 1] - b->job_off[job];
    if (sum_prob_e7) *sum_prob_e7 = s;
    return SA_OK;
}
int sa_batch_pairs8(const sa_batch_t *b, int64_t job, const sa_pair8_t **out, int64_t *n) {
    if (!b || !out || !n || job < 0 || job >= b->c_n) return SA_EINVAL;
    if (!b->ran || !b->p8) return SA_ESTATE;
    *out = reinterpret_cast<const sa_pair8_t *>(b->h_pairs) + b->job_off[job];
    *n = b->job_off[job + 1] - b->job_off[job];
    return SA_OK;
}
int sa_batch_pairs8_all(const sa_batch_t *b, const sa_pair8_t **out, int64_t *first) {
    if (!b || !out) return SA_EINVAL;
    if (!b->ran || !b->p8) return SA_ESTATE;
    *out = reinterpret_cast<const sa_pair8_t *>(b->h_pairs);
    if (first)
        for (int64_t j = 0; j <= b->c_n; j++) first[j] = b->job_off[(size_t) j];
    return SA_OK;
}
int sa_batch_pairs(const sa_batch_t *b, int64_t job, sa_pair_t *out, int64_t cap) {
    if (!b || job < 0 || job >= b->c_n) return SA_EINVAL;
    if (!b->ran || b->p8) return SA_ESTATE;
    long long n = b->job_off[job + 1] - b->job_off[job];
    if (n > cap) return SA_EINVAL;
    const sa_pair16_t *src = b->h_pairs + b->job_off[job];
    for (long long i = 0; i < n; i++) out[i] = sa_pair16_unpack(src[i]);
    return SA_OK;
}
int sa_batch_pairs16(const sa_batch_t *b, int64_t job, const sa_pair16_t **out, int64_t *n) {
    if (!b || !out || !n || job < 0 || job >= b->c_n) return SA_EINVAL;
    if (!b->ran || b->p8) return SA_ESTATE;
    *n = b->job_off[job + 1] - b->job_off[job];
    *out = b->h_pairs + b->job_off[job];
    return SA_OK;
}
int sa_batch_pairs16_all(const sa_batch_t *b, const sa_pair16_t **out, int64_t *first) {
    if (!b || !out) return SA_EINVAL;
    if (!b->ran || b->p8) return SA_ESTATE;
    *out = b->h_pairs;
    if (first)
        for (long long j = 0; j <= (long long) b->c_n; j++) first[j] = b->job_off[(size_t) j];
    return SA_OK;
}
int sa_batch_pairs_all(const sa_batch_t *b, sa_pair_t *out, int64_t cap, int64_t *first) {
    if (!b || (!out && cap > 0)) return SA_EINVAL;
    if (!b->ran || b->p8) return SA_ESTATE;
    const long long nj = b->c_n, total = b->n_pairs_total;
    if (first)
        for (long long j = 0; j <= nj; j++) first[j] = b->job_off[(size_t) j];
    if (total > cap) return SA_EINVAL;
    // the records are contiguous in job order: equal slices to the host threads (9 million pairs per headline batch, 143 MB read
    // and 215 MB written -- a memory-bound loop, first touch of the caller's buffer included)
    const sa_pair16_t *src = b->h_pairs;
    const size_t piece = 1 << 16, np_ = ((size_t) total + piece - 1) / piece;
    sa_parallel_for(np_, [&](size_t q) {
        const size_t i0 = q * piece, i1 = i0 + piece < (size_t) total ? i0 + piece : (size_t) total;
        for (size_t i = i0; i < i1; i++) out[i] = sa_pair16_unpack(src[i]);
    });
    return SA_OK;
}
int sa_batch_stats(const sa_batch_t *b, sa_batch_stats_t *out) {
    if (!b || !out) return SA_EINVAL;
    { const int rcf = batch_finish(const_cast<sa_batch_t *>(b)); if (rcf) return rcf; }
    *out = b->stats;
    return SA_OK;
}
int sa_batch_job_cells(const sa_batch_t *b, int64_t job, double *cf, double *cb) {
    if (!b || job < 0 || job >= b->c_n) return SA_EINVAL;
    { const int rcf = batch_finish(const_cast<sa_batch_t *>(b)); if (rcf) return rcf; }
    if (cf) *cf = b->plan->jobs[job].cells_fwd;
    if (cb) *cb = b->plan->jobs[job].cells_bwd;
    return SA_OK;
}

int sa_align_batch(const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs,
                   const char *const *ambig, int device, unsigned flags, sa_pair_t **pairs_out, int64_t *n_pairs_out) {
    sa_batch_t *b = nullptr;
    int rc = sa_batch_create(&b, m, p, jobs, n_jobs, ambig, device, flags);
    if (rc) return rc;
    rc = sa_batch_run(b);
    if (rc == SA_OK)
        for (int64_t j = 0; j < n_jobs; j++) {
            int64_t n = 0;
            sa_batch_n_pairs(b, j, &n);
            pairs_out[j] = (sa_pair_t *) malloc(sizeof(sa_pair_t) * (size_t) (n > 0 ? n : 1));
            if (!pairs_out[j]) { rc = SA_ENOMEM; break; }
            sa_batch_pairs(b, j, pairs_out[j], n);
            n_pairs_out[j] = n;
        }
    sa_batch_destroy(b);
    return rc;
}

// getExpectationsUsingAnchors for a batch of reads.  The pass runs on the memory-resident kernels (every region is
// planned SA_KIND_GENERIC); the per-group sums are rescaled here with the exact totals of the fold kernel.
static thread_local sa_batch_stats_t tl_expect_stats;   // of this thread's last sa_expect_batch (sa_expect_last_stats)
int sa_expect_last_stats(sa_batch_stats_t *out) {
    if (!out) return SA_EINVAL;
    *out = tl_expect_stats;
    return SA_OK;
}
int sa_expect_batch(const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs,
                    const char *const *ambig, int device, unsigned flags, double *trans9_out, double *likelihood_out,
                    sa_assignment_t **assign_out, int64_t *n_assign_out) {
    if (!trans9_out || !likelihood_out) return SA_EINVAL;
    sa_batch_t *b = nullptr;
    // regions with one path per cell take the register kernels' expectation variant (k_bwd_fast_expect, sa_fast.inc); several
    // paths per cell, SA_FLAG_EXACT or SA_FLAG_FORCE_GENERIC: the memory-resident kernels (the checker of the former)
    if (m && m->emission != 0) flags |= SA_FLAG_FORCE_GENERIC;
    int rc = sa_batch_create(&b, m, p, jobs, n_jobs, ambig, device, flags | SA_FLAG_EXPECT_INTERNAL);
    if (rc) return rc;
    rc = run_passes(b);
    if (rc) { sa_batch_destroy(b); return rc; }
    const sa_plan_t *pl = b->plan;
    tl_expect_stats = b->stats;
    // The per-read sums are taken on the device (k_expect_reduce): 8 doubles per read come back.  HDP models also return the
    // assignment candidates (24 B x slots per posterior diagonal) with the totals they are tested against -- into ONE pinned
    // block (a copy to pageable memory moves 3 GB/s), tested on all host threads.
    const bool want_cands = m->hdp != nullptr && assign_out != nullptr;
    const size_t n_ck = (size_t) (pl->n_cks > 0 ? pl->n_cks : 1), n_sg = (size_t) (pl->n_segs > 0 ? pl->n_segs : 1);
    const size_t o_red = 0, o_tot = o_red + sa_up256(64 * (size_t) (n_jobs > 0 ? n_jobs : 1));
    const size_t o_cnt = o_tot + (want_cands ? sa_up256(8 * n_ck) : 0);
    const size_t o_cand = o_cnt + (want_cands ? sa_up256(4 * n_sg) : 0);
    const size_t host_bytes = o_cand + (want_cands ? sizeof(sa_cand_t) * (size_t) (pl->n_cand > 0 ? pl->n_cand : 1) : 256);
    char *hb = nullptr;
    double *d_red = nullptr;
    if (g_sa_pool.get(SaPool::PINNED, (void **) &hb, host_bytes, b->device) != hipSuccess) { (void) hipGetLastError(); sa_batch_destroy(b); return SA_ENOMEM; }
    if (g_sa_pool.get(SaPool::DEVICE, (void **) &d_red, 64 * (size_t) (n_jobs > 0 ? n_jobs : 1), b->device) != hipSuccess) {
        (void) hipGetLastError();
        g_sa_pool.put(SaPool::PINNED, hb);
        sa_batch_destroy(b);
        return SA_ENOMEM;
    }
    auto dl = [&](void *dst, const void *src, size_t bytes) -> int {
        if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, b->stream));
        return SA_OK;
    };
    auto tail = [&]() -> int {
        HIPCHK(hipMemsetAsync(d_red, 0, 64 * (size_t) (n_jobs > 0 ? n_jobs : 1), b->stream));
        if (pl->n_regions > 0) {
            hipLaunchKernelGGL(k_expect_reduce, dim3((unsigned) pl->n_regions), dim3(64), 0, b->stream, make_devplan(b), d_red,
                               (int) pl->n_regions);
            HIPCHK(hipGetLastError());
        }
        int rc_ = dl(hb + o_red, d_red, 64 * (size_t) n_jobs);
        if (!rc_ && want_cands) rc_ = dl(hb + o_tot, b->d_totals, 8 * (size_t) pl->n_cks);
        if (!rc_ && want_cands) rc_ = dl(hb + o_cnt, b->d_cand_count, 4 * (size_t) pl->n_segs);
        if (!rc_ && want_cands) rc_ = dl(hb + o_cand, b->d_cands, sizeof(sa_cand_t) * (size_t) pl->n_cand);
        if (!rc_ && sa_sync_stream(b->stream, b->device) != hipSuccess) rc_ = SA_ENODEVICE;
        return rc_;
    };
    rc = tail();
    g_sa_pool.put(SaPool::DEVICE, d_red);
    if (rc) { g_sa_pool.put(SaPool::PINNED, hb); sa_batch_destroy(b); return rc; }
    const double *red = reinterpret_cast<const double *>(hb + o_red), *totals = reinterpret_cast<const double *>(hb + o_tot);
    const int *counts = reinterpret_cast<const int *>(hb + o_cnt);
    const sa_cand_t *cands = reinterpret_cast<const sa_cand_t *>(hb + o_cand);
    // (from, to) slots of hmm->transitions[from * 3 + to] in the order the kernels accumulate them
    static const int slot[7] = {0 * 3 + 1, 1 * 3 + 1, 0 * 3 + 0, 1 * 3 + 0, 2 * 3 + 0, 0 * 3 + 2, 2 * 3 + 2};
    const double thr = pl->params.threshold;
    std::atomic<int> oom(0);
    sa_parallel_for((size_t) n_jobs, [&](size_t jj) {
        const int64_t j = (int64_t) jj;
        const sa_jobinfo_t *J = &pl->jobs[j];
        for (int k = 0; k < 7; k++) trans9_out[j * 9 + slot[k]] += red[8 * j + k];
        likelihood_out[j] += red[8 * j + 7];
        std::vector<sa_assignment_t> as;
        for (long long r = J->region_off; want_cands && r < J->region_off + J->n_regions; r++) {
            const sa_region_t *R = &pl->regions[r];
            for (long long sg = R->seg_off; sg < R->seg_off + R->n_seg; sg++) {
                const sa_seg_t *S = &pl->segs[sg];
                for (int i = 0; i < counts[sg]; i++) {
                    const sa_cand_t &cd = cands[S->cand_off + i];
                    long long e = (long long) cd.x + cd.y + 2;
                    double total = totals[S->ck_base + (S->from - e) / SA_CKPT_EVERY];
                    if (exp(cd.fb - total) >= thr) {
                        sa_assignment_t a;
                        a.ref_pos = cd.x + R->x1;
                        a.event = cd.y + R->y1;
                        as.push_back(a);
                    }
                }
            }
        }
        if (assign_out) {
            assign_out[j] = (sa_assignment_t *) malloc(sizeof(sa_assignment_t) * (as.size() ? as.size() : 1));
            if (!assign_out[j]) { oom.store(1); return; }
            if (!as.empty()) memcpy(assign_out[j], as.data(), sizeof(sa_assignment_t) * as.size());
            if (n_assign_out) n_assign_out[j] = (int64_t) as.size();
        }
    });
    g_sa_pool.put(SaPool::PINNED, hb);
    if (oom.load()) { sa_batch_destroy(b); return SA_ENOMEM; }
    sa_batch_destroy(b);
    return SA_OK;
}
