// sa_ea.hip -- event <-> k-mer pre-alignment on the GPU.
//
// What it replaces: adaptive_banded_simple_event_align2 (impl/eventAligner.c:899-1235), the step immediately upstream
// of the pair-HMM: Suzuki-Kasahara adaptive banding as used by nanopolish.  Bands are anti-diagonals of the
// (event+1) x (k-mer+1) matrix, 100 cells wide; every new band steps right or down from the previous one depending on
// which end of it scores higher; Viterbi scores over three moves (step, stay, skip) kept as floats; traceback from the
// best (event, last k-mer) cell; three quality checks.
//
// Mapping: one wave per read.  The recurrence is serial in the band index (~ events + k-mers steps), the 100 cells of a
// band are independent: lane l computes offsets l and l+64.  The live bands and every cell's operands stay in registers
// (neighbours are DPP wave rotates, the move taken is a wave-uniform branch; see k_event_align); the trace (one byte per
// cell) and the band origins go to global memory for the traceback, which the wave walks at the end through 64-band
// blocks staged in LDS.
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

// ONE WAVE PER READ, THE BAND IN REGISTERS.  The band is 100 offsets: lane l owns offset l ("rep 0") and, for l < 36,
// offset 64 + l ("rep 1").  A band differs from the previous one by a single move, right (k-mer + 1) or down (event + 1),
// decided from the two end cells (Suzuki-Kasahara); which move it is is a wave-uniform scalar, so the loop body exists
// twice, once per move, each with STATIC neighbour shifts:
//   right:  up = prev[o + 1], left = prev[o],     the lane's k-mer operands shift one offset down, a fresh k-mer enters at 99
//   down:   up = prev[o],     left = prev[o - 1], the lane's event shifts one offset up, a fresh event enters at 0
// and the diagonal neighbour (band b-2) is one of the two shifted copies the band before already made.  Neighbours are
// DPP wave rotates (with the lane 63 -> rep 1 hand-over patched by one v_cndmask), fresh operands come out of a 64-entry
// register buffer per stream by v_readlane (refilled with one coalesced load every 64 moves of its kind): the serial
// chain of a band holds no LDS access, no barrier and no memory load.  An earlier version (two waves per read, band and
// operand streams in LDS, one barrier per band) ran at 1.1 us per band, LDS throughput (30 wave-wide reads per band,
// 8 reads resident per CU) and the barrier being the limit.
// UNIT_VAR: every read of the launch has var == 1 (the reference always aligns with var = 1, impl/eventAligner.c:845-849):
// x / 1.0 == x, one division per cell less.
#define EA_THREADS 64
#define EA_LAST1 (EA_BW - 65)   // lane of offset 99 in rep 1

__device__ __forceinline__ int ea_from_next(int v) { return __builtin_amdgcn_mov_dpp(v, 0x134, 0xF, 0xF, false); }  // lane l <- l+1
__device__ __forceinline__ int ea_from_prev(int v) { return __builtin_amdgcn_mov_dpp(v, 0x13C, 0xF, 0xF, false); }  // lane l <- l-1
// x[o] <- x[o + 1] over the 100 offsets, `fill` enters at offset 99
__device__ __forceinline__ void ea_shl(int &x0, int &x1, int fill, int lane) {
    const int n0 = ea_from_next(x0), n1 = ea_from_next(x1);   // n1[63] = x1[0]
    x0 = lane == 63 ? n1 : n0;
    x1 = lane == EA_LAST1 ? fill : n1;
}
// x[o] <- x[o - 1], `fill` enters at offset 0
__device__ __forceinline__ void ea_shr(int &x0, int &x1, int fill, int lane) {
    const int p0 = ea_from_prev(x0), p1 = ea_from_prev(x1);   // p0[0] = x0[63]
    x1 = lane == 0 ? p0 : p1;
    x0 = lane == 0 ? fill : p0;
}
__device__ __forceinline__ void ea_shl(double &x0, double &x1, double fill, int lane) {
    int a0 = __double2loint(x0), a1 = __double2loint(x1), b0 = __double2hiint(x0), b1 = __double2hiint(x1);
    ea_shl(a0, a1, __double2loint(fill), lane);
    ea_shl(b0, b1, __double2hiint(fill), lane);
    x0 = __hiloint2double(b0, a0); x1 = __hiloint2double(b1, a1);
}
__device__ __forceinline__ void ea_shr(double &x0, double &x1, double fill, int lane) {
    int a0 = __double2loint(x0), a1 = __double2loint(x1), b0 = __double2hiint(x0), b1 = __double2hiint(x1);
    ea_shr(a0, a1, __double2loint(fill), lane);
    ea_shr(b0, b1, __double2hiint(fill), lane);
    x0 = __hiloint2double(b0, a0); x1 = __hiloint2double(b1, a1);
}
__device__ __forceinline__ void ea_shl(float &x0, float &x1, float fill, int lane) {
    int a0 = __float_as_int(x0), a1 = __float_as_int(x1);
    ea_shl(a0, a1, __float_as_int(fill), lane);
    x0 = __int_as_float(a0); x1 = __int_as_float(a1);
}
__device__ __forceinline__ void ea_shr(float &x0, float &x1, float fill, int lane) {
    int a0 = __float_as_int(x0), a1 = __float_as_int(x1);
    ea_shr(a0, a1, __float_as_int(fill), lane);
    x0 = __int_as_float(a0); x1 = __int_as_float(a1);
}
// A loaded value passed through a VALU move: the wait for the load is then placed HERE (inside the refill branch that
// executes once per 64 moves) and not before the buffer's use in every band, where "s_waitcnt vmcnt(0)" would also wait
// for all the trace stores of the previous band (one counter orders loads and stores).
__device__ __forceinline__ double ea_settle(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v), lo2, hi2;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lo2) : "v"(lo));
    asm volatile("v_mov_b32 %0, %1" : "=v"(hi2) : "v"(hi));
    return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double ea_rld(double v, int k) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), k), __builtin_amdgcn_readlane(__double2loint(v), k));
}

// one cell: the reference's three scores in float, emission in double (ea_emit's operations in ea_emit's order)
template <bool UNIT_VAR>
__device__ __forceinline__ double ea_cell(const EaJob &J, float up, float left, float diag, double mu, double sd, double c, double e,
                                          unsigned char &from) {
    const double num = e + J.var * mu - J.scale * mu - J.shift;
    const double en = UNIT_VAR ? num : num / J.var;
    const double a = (en - mu) / sd;
    const double lp_em = J.lvar + (c + (-0.5 * a * a));
    const float s_d = (float) ((double) diag + J.lp_step + lp_em);
    const float s_u = (float) ((double) up + J.lp_stay + lp_em);
    const float s_l = (float) ((double) left + J.lp_skip);
    float best = s_d;
    from = 0;                             // FROM_D
    best = s_u > best ? s_u : best;
    from = best == s_u ? 1 : from;        // FROM_U
    best = s_l > best ? s_l : best;
    from = best == s_l ? 2 : from;        // FROM_L
    return (double) best;
}

template <bool UNIT_VAR>
__global__ __launch_bounds__(EA_THREADS) void k_event_align(EaPlan P, int n_jobs) {
    const int job = blockIdx.x;
    if (job >= n_jobs) return;
    const int lane = threadIdx.x;
    const EaJob J = P.jobs[job];
    const double *ev = P.ev + J.ev_off;
    const double *kc = P.kc + 3 * J.kc_off;
    unsigned char *trace = P.trace + J.trace_off;
    int *ll = P.ll + 2 * J.ll_off;
    double *col = P.col + J.col_off;
    const int n_events = J.n_events, n_kmers = J.n_kmers;
    const int n_bands = (n_events + 1) + (n_kmers + 1);
    const float NINF = -__builtin_inff();

    int my_fills = 0;
    for (int i = lane; i < n_events; i += 64) col[i] = EA_NEG_INF;
    // band 1 (the band "b-1" of the first iteration): offset 50 holds the first event trimmed; band 0: 0.0 at offset 50
    int ll_ev1 = EA_HALF, ll_km1 = -1 - EA_HALF;
    if (lane == 0) {
        ll[0] = EA_HALF - 1; ll[1] = -1 - EA_HALF;
        ll[2] = ll_ev1; ll[3] = ll_km1;
    }
    float vf0 = lane == EA_HALF ? (float) J.lp_trim : NINF, vf1 = NINF;   // band b-1 as its readers see it
    // band b-2 as band b's diagonal: band 0 -> band 1 was a DOWN move (ll_ev 49 -> 50), so o_diag = o - 1 if band 2 moves
    // down, o if it moves right; band 0 is 0.0 at offset 50
    float dd0 = lane == EA_HALF + 1 ? 0.0f : NINF, dd1 = NINF;            // diagonal neighbour if this band moves down
    float dr0 = lane == EA_HALF ? 0.0f : NINF, dr1 = NINF;                // ... if it moves right
    double lo = EA_NEG_INF, hi = EA_NEG_INF;                              // band b-1 at offsets 0 and 99
    // operands of band b-1's cells: k-mer ll_km1 + o, event ll_ev1 - o
    double mu0 = 0, sd0 = 1, c0 = 0, mu1 = 0, sd1 = 1, c1 = 0, e0 = 0, e1 = 0;
    {
        const int ka = ll_km1 + lane, kb = ll_km1 + 64 + lane, ea = ll_ev1 - lane, eb = ll_ev1 - 64 - lane;
        if (ka >= 0 && ka < n_kmers) { mu0 = kc[3ll * ka]; sd0 = kc[3ll * ka + 1]; c0 = kc[3ll * ka + 2]; }
        if (kb >= 0 && kb < n_kmers && lane <= EA_LAST1) { mu1 = kc[3ll * kb]; sd1 = kc[3ll * kb + 1]; c1 = kc[3ll * kb + 2]; }
        if (ea >= 0 && ea < n_events) e0 = ev[ea];
        if (eb >= 0 && eb < n_events) e1 = ev[eb];
    }
    // streams: the next k-mer to enter at offset 99 is ll_km1 + 100, the next event to enter at offset 0 is ll_ev1 + 1
    int ks = ll_km1 + EA_BW, kb_base = ks, es = ll_ev1 + 1, eb_base = es;
    double kb_mu = 0, kb_sd = 1, kb_c = 0, eb_v = 0;
    {
        const int k = kb_base + lane, e = eb_base + lane;
        if (k >= 0 && k < n_kmers) { kb_mu = kc[3ll * k]; kb_sd = kc[3ll * k + 1]; kb_c = kc[3ll * k + 2]; }
        if (e >= 0 && e < n_events) eb_v = ev[e];
    }
    kb_mu = ea_settle(kb_mu); kb_sd = ea_settle(kb_sd); kb_c = ea_settle(kb_c); eb_v = ea_settle(eb_v);
    mu0 = ea_settle(mu0); sd0 = ea_settle(sd0); c0 = ea_settle(c0); mu1 = ea_settle(mu1); sd1 = ea_settle(sd1); c1 = ea_settle(c1);
    e0 = ea_settle(e0); e1 = ea_settle(e1);
    for (int b = 2; b < n_bands; b++) {
        bool right;
        if (lo == EA_NEG_INF && hi == EA_NEG_INF) right = (b % 2) == 1;  // both ends outside the matrix: alternate
        else right = lo < hi;                                           // Suzuki's rule
        const int ll_ev = ll_ev1 + (right ? 0 : 1), ll_km = ll_km1 + (right ? 1 : 0);
        if (lane == 0) { ll[2 * b] = ll_ev; ll[2 * b + 1] = ll_km; }
        int o_min = 0 - ll_km, o_max = n_kmers - ll_km;
        const int e_min = ll_ev - (n_events - 1), e_max = ll_ev + 1;
        o_min = e_min > o_min ? e_min : o_min;
        o_min = o_min < 0 ? 0 : o_min;
        o_max = e_max < o_max ? e_max : o_max;
        o_max = o_max > EA_BW ? EA_BW : o_max;
        const int trim_o = -1 - ll_km;
        float up0, up1, lf0, lf1, dg0, dg1;
        if (right) {
            // a fresh k-mer enters at offset 99
            if (ks - kb_base == 64) {
                kb_base += 64;
                const int k = kb_base + lane;
                kb_mu = 0; kb_sd = 1; kb_c = 0;
                if (k >= 0 && k < n_kmers) { kb_mu = kc[3ll * k]; kb_sd = kc[3ll * k + 1]; kb_c = kc[3ll * k + 2]; }
                kb_mu = ea_settle(kb_mu); kb_sd = ea_settle(kb_sd); kb_c = ea_settle(kb_c);
            }
            const int q = ks - kb_base;
            ks++;
            ea_shl(mu0, mu1, ea_rld(kb_mu, q), lane);
            ea_shl(sd0, sd1, ea_rld(kb_sd, q), lane);
            ea_shl(c0, c1, ea_rld(kb_c, q), lane);
            lf0 = vf0; lf1 = vf1;
            up0 = vf0; up1 = vf1;
            ea_shl(up0, up1, NINF, lane);          // prev[o + 1]; offset 100 does not exist
            dg0 = dr0; dg1 = dr1;
            dr0 = up0; dr1 = up1;                  // for the NEXT band: band b-1 at o + 1 (dk = 1, right) ...
            dd0 = lf0; dd1 = lf1;                  // ... and at o (dk = 1, down)
        } else {
            if (es - eb_base == 64) {
                eb_base += 64;
                const int e = eb_base + lane;
                eb_v = (e >= 0 && e < n_events) ? ev[e] : 0.0;
                eb_v = ea_settle(eb_v);
            }
            const int q = es - eb_base;
            es++;
            ea_shr(e0, e1, ea_rld(eb_v, q), lane);
            up0 = vf0; up1 = vf1;
            lf0 = vf0; lf1 = vf1;
            ea_shr(lf0, lf1, NINF, lane);          // prev[o - 1]; offset -1 does not exist
            dg0 = dd0; dg1 = dd1;
            dr0 = up0; dr1 = up1;                  // for the NEXT band: band b-1 at o (dk = 0, right) ...
            dd0 = lf0; dd1 = lf1;                  // ... and at o - 1 (dk = 0, down)
        }
        double val0 = EA_NEG_INF, val1 = EA_NEG_INF;
        {
            const int o = lane;
            if (o == trim_o) {  // k-mer -1: every event so far trimmed
                const int e = ll_ev - o;
                if (e >= 0 && e < n_events) val0 = J.lp_trim * (double) (e + 1);
            }
            if (o >= o_min && o < o_max) {
                unsigned char from;
                val0 = ea_cell<UNIT_VAR>(J, up0, lf0, dg0, mu0, sd0, c0, e0, from);
                trace[(long long) b * EA_BW + o] = from;
                if (ll_km + o == n_kmers - 1) col[ll_ev - o] = val0;
                my_fills++;
            }
        }
        if (lane <= EA_LAST1) {
            const int o = 64 + lane;
            if (o == trim_o) {
                const int e = ll_ev - o;
                if (e >= 0 && e < n_events) val1 = J.lp_trim * (double) (e + 1);
            }
            if (o >= o_min && o < o_max) {
                unsigned char from;
                val1 = ea_cell<UNIT_VAR>(J, up1, lf1, dg1, mu1, sd1, c1, e1, from);
                trace[(long long) b * EA_BW + o] = from;
                if (ll_km + o == n_kmers - 1) col[ll_ev - o] = val1;
                my_fills++;
            }
        }
        vf0 = (float) val0; vf1 = (float) val1;
        lo = ea_rld(val0, 0); hi = ea_rld(val1, EA_LAST1);
        ll_ev1 = ll_ev; ll_km1 = ll_km;
    }
    for (int off = 32; off > 0; off >>= 1) my_fills += __shfl_xor(my_fills, off, 64);
    __threadfence();   // col[] entries written by other lanes, read below
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
    if (lane == 0) P.fills[job] = my_fills;
    // Traceback by the whole wave.  A pointer walk would be two dependent HBM loads per step (the band's origin, then
    // the trace byte at an offset that depends on it).  Instead the trace rows and origins of 64 consecutive bands are
    // staged with coalesced loads (rows into LDS, origins one per lane) and the walk inside the block is LDS reads and
    // v_readlane; a step moves one or two bands down, so a block serves 32-64 steps.  The scan starts from event 0 when
    // nothing scored (best == -inf), like the reference.
    __shared__ unsigned int tr_rows[64 * EA_BW / 4];
    const float best_u = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(best)));
    int cur_ev = (best_u > -__builtin_inff()) ? __builtin_amdgcn_readfirstlane(best_ev) : 0, cur_km = n_kmers - 1;
    int *out = P.out + 2 * J.out_off;
    int n = 0, cur_gap = 0, max_gap = 0, first_km = -1, last_km = -1;
    while (cur_km >= 0 && cur_ev >= 0) {
        const int b_hi = cur_ev + cur_km + 2, b_lo = b_hi - 63 > 0 ? b_hi - 63 : 0;
        const unsigned int *src = (const unsigned int *) (trace + (long long) b_lo * EA_BW);   // 100-byte rows: 4-byte aligned
        const int n_words = (b_hi - b_lo + 1) * (EA_BW / 4);
        {   // all 25 loads of a lane in flight together, then the LDS writes
            unsigned int w[EA_BW / 4];
#pragma unroll
            for (int q = 0; q < EA_BW / 4; q++) w[q] = lane + 64 * q < n_words ? src[lane + 64 * q] : 0u;
#pragma unroll
            for (int q = 0; q < EA_BW / 4; q++) tr_rows[lane + 64 * q] = w[q];
        }
        const int llv = (b_hi - lane >= b_lo) ? ll[2 * (b_hi - lane)] : 0;                      // origin (event) of band b_hi - lane
        const unsigned char *rows = (const unsigned char *) tr_rows;
        while (cur_km >= 0 && cur_ev >= 0) {
            const int b = cur_ev + cur_km + 2;
            if (b < b_lo) break;
            const int o = __builtin_amdgcn_readlane(llv, b_hi - b) - cur_ev;
            const int from = __builtin_amdgcn_readfirstlane((int) rows[(b - b_lo) * EA_BW + o]);
            if (lane == 0) { out[2 * n] = cur_km; out[2 * n + 1] = cur_ev; }
            if (n == 0) first_km = cur_km;
            last_km = cur_km;
            n++;
            if (from == 0) { cur_km--; cur_ev--; cur_gap = 0; }
            else if (from == 1) { cur_ev--; cur_gap = 0; }
            else { cur_km--; cur_gap++; max_gap = cur_gap > max_gap ? cur_gap : max_gap; }
        }
    }
    // the emissions along the path, summed in path order as the reference does while it walks: computed 64 at a time,
    // folded by v_readlane
    __threadfence();
    double sum_em = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane, cnt = n - base < 64 ? n - base : 64;
        double em = 0;
        if (i < n) em = ea_emit(kc + 3ll * out[2 * i], ev[out[2 * i + 1]], J);
        for (int k = 0; k < cnt; k++) sum_em += ea_rld(em, k);
    }
    int st = 0;
    const double avg = sum_em / (double) n;
    if (avg < -5.2) st |= 1;
    // pairs are in traceback order: the last one is the front of the reversed list
    if (!(n > 0 && last_km == 0 && first_km == n_kmers - 1)) st |= 2;
    if (max_gap > 50) st |= 4;
    if (J.events_per_kmer > 5.0) st |= 8;
    if (lane == 0) {
        P.status[job] = st;
        P.n_out[job] = st ? 0 : n;
    }
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
        ev_n += (size_t) jb->n_events;
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
    {   // the upload image, one read per task: event means copied, k-mer ids rolled
        std::atomic<int> bad(SA_OK);
        sa_parallel_for((size_t) n_jobs, [&](size_t j) {
            const sa_ea_job_t *jb = &jobs[j];
            const EaJob &J = hj[j];
            memcpy(ev + J.ev_off, jb->event_mean, sizeof(double) * (size_t) jb->n_events);
            const int rck = ea_kmer_ids(m, jb->sequence, J.n_kmers, (flags & SA_FLAG_RNA) != 0, ids + J.kc_off);
            if (rck) bad = rck;
        });
        if (bad != SA_OK) return bad;
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
    {
        std::atomic<bool> oom(false);
        sa_parallel_for(nj, [&](size_t j) {
            const int n = h_n[j];
            if (status_out) status_out[j] = h_st[j];
            n_pairs_out[j] = n;
            pairs_out[j] = (sa_ea_pair_t *) malloc(sizeof(sa_ea_pair_t) * (size_t) (n > 0 ? n : 1));
            if (!pairs_out[j]) { oom = true; return; }
            const int *src = h_out + 2 * hj[j].out_off;
            for (int i = 0; i < n; i++) {  // stList_reverse: ascending order
                pairs_out[j][i].kmer_idx = src[2 * (n - 1 - i)];
                pairs_out[j][i].event_idx = src[2 * (n - 1 - i) + 1];
            }
        });
        if (oom) rc = SA_ENOMEM;
    }
done:
    if (rc != SA_OK)
        for (int64_t j = 0; j < n_jobs; j++) { free(pairs_out[j]); pairs_out[j] = nullptr; n_pairs_out[j] = 0; }
    return rc;
}
