/* sa_internal.h -- structures shared by the host planner (sa_plan.c), the loaders (sa_io.c) and the
 * HIP runtime (sa_hip.hip).  Everything that is uploaded verbatim to HBM is a POD with fixed layout. */
#ifndef SA_INTERNAL_H_
#define SA_INTERNAL_H_

#include <stddef.h>
#include <stdint.h>

#include "signalalign_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SA_NEG_INF (-__builtin_inf())
#define SA_PROB_1 10000000.0          /* PAIR_ALIGNMENT_PROB_1, inc/pairwiseAligner.h:27 */
#define SA_CKPT_EVERY 10              /* totalProbability refresh period, impl/pairwiseAligner.c:1538 */
#define SA_LOG_GAPX (-2.3025850929940455) /* log(0.1): impl/stateMachine.c:1584-1586, :1394 */
#define SA_GAPY_SD_MULT 1.75          /* EXTRA_EVENT_NOISE_MULTIPLIER, inc/stateMachine.h:34 */
#define SA_FLAG_EXPECT_INTERNAL 0x10000u /* sa_expect_batch: expectation pass instead of posteriors */
#define SA_FLAG_DEVICE_XC_INTERNAL 0x20000u /* sa_batch_create: the emission constants (xc) are filled on the device; the
                                             * planner leaves pl->xc NULL */
#define SA_CAND_PER_DIAG 2            /* planned candidate slots per posterior diagonal of a traceback (+32 diagonals' worth per segment) */
#define SA_CAND_PER_DIAG_HDP 8
#define SA_CAND_EPS 1e-6              /* slack of the on-device candidate filter (see sa_hip.hip) */

/* ---- model (host) ---------------------------------------------------------------------------- */
typedef struct sa_hdp {
    int64_t num_dps, grid_length;
    double grid_start, grid_stop;
    double *grid;      /* linspace, impl/hdp_math_utils.c:497-510 */
    int64_t *parent;
    uint8_t *observed;
    int64_t *resolved; /* dp -> nearest observed ancestor (impl/hdp.c:2600-2602), -1 if none   */
    int64_t *slot;     /* observed dp -> row in y/slope tables, -1 otherwise                     */
    int64_t n_slots;
    double *y;         /* n_slots * grid_length posterior predictive                            */
    double *slope;     /* n_slots * grid_length spline slopes                                   */
} sa_hdp_t;

struct sa_model {
    int n_alpha, k;
    char alphabet[64]; /* sorted: sequence_prepareAlphabet, impl/pairwiseAligner.c:366-395 */
    int64_t n_kmers;
    int64_t pow_km1;   /* A^(k-1) */
    /* log transitions, named by (from -> to) */
    double t_mm, t_mx, t_my; /* match->match, match->gapX (open X), match->gapY (open Y)   */
    double t_xm, t_xx;       /* gapX->match, gapX->gapX                                    */
    double t_ym, t_yy;       /* gapY->match, gapY->gapY                                    */
    double *table5;          /* EMISSION_MATCH_MATRIX                                      */
    sa_hdp_t *hdp;
    int emission;            /* SA_EMISSION_* (signalalign_hip.h) */
    uint64_t uid;            /* unique per model object of the process, never reused (an address is: the CLI clones and destroys a
                              * model per read): what the library keys per-model memory on (candidate capacity, sa_hip.hip) */
};
uint64_t sa_model_next_uid(void);

/* ---- plan (host arrays, uploaded as they are) -------------------------------------------------- */
typedef struct sa_row {
    int32_t xmyL;  /* smallest x-y on this anti-diagonal */
    int32_t width; /* cells                               */
    int64_t foff;  /* offset (cell-paths) of the row inside the region's forward storage */
} sa_row_t;        /* 16 bytes per anti-diagonal: with 9000 diagonals per read the largest array of a plan */

/* SA_KIND_RING: lane-per-cell-path kernels with the three live diagonals' messages in an LDS ring (sa_ring.inc): regions with
 * several paths per cell (ambiguous positions) and one-path regions whose band is mostly wider than a wave.  Forward storage
 * in planes like SA_KIND_FAST; rows[d].foff holds (g0 << 32) | offset, g0 = poff[first x of the diagonal] (so that a
 * diagonal's record is one 16-byte scalar load), and rows[N + 1] is a sentinel whose offset closes the last diagonal. */
enum { SA_KIND_GENERIC = 0, SA_KIND_FAST = 1, SA_KIND_RING = 2 };
#define SA_RING_MAX_ROWPATHS 512      /* widest diagonal (cell-paths) the LDS ring takes: 3 rows x 3 messages x 8 B x 512 = 36 KB */
#define SA_RING_WIDE_FRACTION 0.5     /* one-path regions go to the ring kernels when more than this share of their cells lies on
                                       * diagonals the register kernels cannot hold (SA_PK_FWD clear) */

/* per cell-path record of SA_KIND_RING regions with several paths per cell (index: pid_off + poff[x] + path).
 * Legal predecessors in column x-1 (path_checkLegal, impl/pairwiseAligner.c:595-621: k-1 shared letters) are the paths
 * pred0 + i * stride, i < npred; legal successors in column x+1 the contiguous paths succ0 .. succ0 + nsucc - 1
 * (both as region-relative path-space indices poff[.] + path; -1 / 0 when there is none). */
typedef struct sa_prec {
    int32_t x;
    int32_t pred0;
    int32_t succ0;
    uint32_t meta;     /* stride << 16 | npred << 8 | nsucc */
} sa_prec_t;

typedef struct sa_region {
    int32_t job, kind;
    int32_t ragged_l, ragged_r;
    int64_t x1, y1, lX, lY, N;
    int64_t row_off;  /* rows[row_off + d], d = 0..N                                   */
    int64_t pk_off;   /* pk[pk_off + SA_PK_PAD + d]: packed band word of diagonal d     */
    int64_t poff_off; /* poff[poff_off + x], x = 0..lX+1 (region-relative path offsets) */
    int64_t pid_off;  /* pid / xc arrays: index pid_off + poff[x] + p                   */
    int64_t ev_off;   /* ev[ev_off + y-1] is the event of matrix row y                  */
    int64_t f_base;   /* cell-path offset of the forward storage inside the chunk       */
    int64_t seg_off;
    int32_t n_seg;
    int32_t K;        /* even offset so that (x-y+K)>>1 >= 0                            */
    int32_t max_rowpaths;
    int32_t slots;    /* 64-lane slots the widest 3-diagonal window needs               */
    int32_t chunk;    /* which forward-storage pass handles this region                 */
    int32_t max_p;    /* paths of the cell with the most                                */
    double scale, shift, var, lvar; /* lvar = log(1/var), impl/stateMachine.c:602       */
    int64_t f_cellpaths;
} sa_region_t;

/* packed band word (register kernels): width | flags | ((x-y+K)>>1 of the first cell) << SA_PK_SHIFT.
 * Everything wave-uniform the inner loops would otherwise derive from neighbouring words is a flag here. */
#define SA_PK_WIDTH_MASK 127
#define SA_PK_FWD 128        /* diagonal d, d-1, d-2 (+1 cell each side) fit in 64 lanes                       */
#define SA_PK_BWD 256        /* diagonal e, e+1, e+2 fit in 64 lanes                                           */
#define SA_PK_FULL 512       /* forward sweep stores all three planes of this diagonal: it is a checkpoint, or
                                one of the next two diagonals is memory-resident, or the matrix ends            */
#define SA_PK_FWD_MORE 1024  /* d < N and diagonal d+1 has SA_PK_FWD                                           */
#define SA_PK_BWD_MORE 2048  /* diagonal e-1 has SA_PK_BWD                                                     */
#define SA_PK_CK 4096        /* total-probability checkpoint of the traceback that owns this diagonal          */
#define SA_PK_SHIFT 13
#define SA_PK_PAD 64         /* readable (zero) words in front of diagonal 0; 160 behind diagonal N             */
#define SA_FAST_ROW_ALIGN 1 /* cells; 16 (128-byte rows) was measured: no gain, packed rows write better (probes/store_probe.hip) */
#define SA_HDP_FAST_MAX_BYTES 0x7f000000ll /* {y, slope} table of the observed processes, register kernels */
#define SA_FAST_MAX_CELLS (1ll << 27) /* register kernels address a region's forward planes with 32-bit byte offsets */

typedef struct sa_seg {
    int32_t region, at_end;
    int64_t start; /* diagonal the traceback starts on (end-state init)      */
    int64_t from;  /* tracedBackFrom: last diagonal that emits posteriors    */
    int64_t to;    /* tracedBackTo: exclusive lower end                      */
    int64_t ck_base;
    int32_t n_ck, cand_cap;
    int64_t cand_off;
    int64_t bscratch_off; /* 3 rows of backward state for the memory-resident path */
} sa_seg_t;

typedef struct sa_ck {
    int64_t voff;   /* vbuf[voff .. voff+nA) = per-cell dot(F,B) of the checkpoint diagonal,
                       vbuf[voff+nA .. voff+nA+nB) = match-through terms of the next diagonal */
    int32_t nA, nB;
} sa_ck_t;

typedef struct sa_cand {
    int32_t x, y;    /* sequence coordinates inside the region (matrix coordinate - 1) */
    int32_t path, pad;
    double fb;       /* forward.match + backward.match (log space)                    */
} sa_cand_t;

/* (sa_pair16_t, the 16-byte result record, and its pack / unpack functions are part of the public header: a caller may read
 * the packed records in place, sa_batch_pairs16) */

typedef struct sa_jobinfo {
    int64_t region_off;
    int32_t n_regions, pad;
    int64_t ev_off, n_events;
    double cells_fwd, cells_bwd;
} sa_jobinfo_t;

typedef struct sa_plan {
    const sa_model_t *model;
    sa_params_t params;
    unsigned flags;
    int64_t n_jobs;
    sa_jobinfo_t *jobs;
    sa_region_t *regions; int64_t n_regions, cap_regions;
    sa_row_t *rows;       int64_t n_rows, cap_rows;
    int32_t *pk;          int64_t n_pk, cap_pk;
    int32_t *poff;        int64_t n_poff, cap_poff;
    int32_t *pid;         int64_t n_pid, cap_pid;
    double *xc;           /* 4 doubles per pid entry: m, inv_s, cM, cY (read-params folded in) */
    sa_prec_t *prec;      /* one per pid entry (allocated when the batch may hold SA_KIND_RING regions with several paths) */
    int64_t prec_cap;
    double *ev;           int64_t n_ev, cap_ev;
    sa_seg_t *segs;       int64_t n_segs, cap_segs;
    sa_ck_t *cks;         int64_t n_cks, cap_cks;
    int64_t n_vbuf;       /* doubles */
    int64_t n_cand;       /* candidate slots */
    int64_t n_bscratch;   /* doubles */
    int64_t max_chunk_cellpaths;
    int32_t n_chunks;
    double cells_fwd, cells_bwd;
    int64_t n_fast_regions;
    int64_t n_ring_regions;
    int64_t max_span;
    int32_t pooled;       /* rows, pk, poff, pid, xc, ev came from plan_big_alloc and go back to its cache */
    void (*big_free)(void *p, size_t bytes); /* ... or from the caller's allocator (sa_plan_use_allocator): pinned host
                                              * memory the device reads directly */
    int32_t borrowed;     /* planner thread's sub-plan: rows, pk, poff, pid, xc, ev are slices of the final plan's arrays (sized
                           * exactly by a counting pass): never grown, never freed here */
} sa_plan_t;

/* sa_plan.c */
int sa_plan_build(sa_plan_t **out, const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs,
                  const char *const *ambig256, unsigned flags, int64_t chunk_budget_cellpaths);
void sa_plan_free(sa_plan_t *pl);
void sa_plan_repack(sa_plan_t *pl, int64_t chunk_budget_cellpaths); /* regions[].chunk / f_base, max_chunk_cellpaths, n_chunks */
void sa_plan_pool_release(void); /* frees the host blocks the planner keeps between batches */
/* The calling thread's next sa_plan_build takes the big arrays of a threaded plan from `alloc` and returns them with
 * `release` (sa_batch_create: pinned memory from the caching allocator, so that the upload is a plain DMA); NULL, NULL
 * restores the planner's own block cache. */
void sa_plan_use_allocator(void *(*alloc)(size_t bytes), void (*release)(void *p, size_t bytes));
/* grows every segment's candidate capacity by `factor` and re-lays out cand_off (overflow retry) */
void sa_plan_grow_candidates(sa_plan_t *pl, int factor);
/* after the device pass: turn candidates + totals into the reference's pair list for every job */
int sa_plan_finalize(const sa_plan_t *pl, const sa_cand_t *cands, const int32_t *cand_count, const double *totals,
                     sa_pair_t **pairs_out, int64_t *n_pairs_out);
int64_t sa_model_kmer_id(const sa_model_t *m, const char *kmer);
int sa_band_rows(const int64_t *ax, const int64_t *ay, int64_t n, int64_t lX, int64_t lY, int64_t expansion,
                 int64_t *xmyL, int64_t *xmyR);

#ifdef __cplusplus
}
#endif
#endif
