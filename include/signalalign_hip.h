/*
 * signalalign_hip.h -- C ABI of libsignalalign_hip.so: the MI355X-native replacement for
 * signalAlign's banded pair-HMM forward/backward/posterior path.
 *
 * Plain C: pointers and sizes only.  Every entry point names the reference interface it replaces
 * (paths relative to the upstream signalAlign tree).  The library never exits the process; it
 * returns 0 or a negative SA_E* code (sa_strerror()).  All device work is hand-written HIP for
 * gfx950; there is NO CPU fallback: without a usable GPU sa_batch_create()/sa_batch_run() fail with
 * SA_ENODEVICE.
 *
 * The seam is getAlignedPairsUsingAnchors()/getExpectationsUsingAnchors()
 * (inc/pairwiseAligner.h:406-429): a state machine, a k-mer sequence, an event sequence, a list of
 * anchor pairs and banding parameters go in; (floor(p*1e7), x, y, path k-mer) tuples come out.
 * Here the same call is batched over many reads so one launch fills the GPU.
 */
#ifndef SIGNALALIGN_HIP_H_
#define SIGNALALIGN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SA_OK 0
#define SA_EINVAL (-1)     /* bad argument                                        */
#define SA_ENOMEM (-2)     /* host or device allocation failed                    */
#define SA_ENODEVICE (-3)  /* no usable HIP device / HIP runtime error            */
#define SA_EALPHABET (-4)  /* reference k-mer has a character outside the alphabet
                              (the reference aborts in kmer_to_word, impl/nanopore_hdp.c:387-403) */
#define SA_EBAND (-5)      /* anchors produce an invalid diagonal (diagonal_construct throws,
                              impl/pairwiseAligner.c:98-103)                       */
#define SA_EIO (-6)        /* file could not be read / parsed                     */
#define SA_ESTATE (-7)     /* call out of order (e.g. results before run)         */
#define SA_EUNSUPPORTED (-8)

/* flags for sa_batch_create */
#define SA_FLAG_EXACT 1u          /* reference-ordered, un-contracted fp64 arithmetic on the device
                                     (slow kernels; bit-identical posteriors for Gaussian emissions) */
#define SA_FLAG_FORCE_GENERIC 2u  /* never pick the register-resident fast kernels */
#define SA_FLAG_RNA 4u            /* event alignment only: k-mers as build_kmer_list(..., rna=true) makes them (U -> T, reversed) */
#define SA_FLAG_DEVICE_TO_ITSELF 8u /* sa_batch_create_deferred only: every other batch on this device will have been destroyed
                                     before this one's first use -- its forward storage is sized for the memory they hold too
                                     (one batch at a time, the next one packed and planned while the current one runs).
                                     The promise is the caller's: if other batches still hold their storage at that first use
                                     (sa_batch_run / sa_batch_start, but also sa_batch_stats and sa_batch_job_cells, which
                                     complete the creation too), the working buffers may not fit and that call returns
                                     SA_ENOMEM; the batch can then only be destroyed */

#define SA_FLAG_VC_ROWS 32u        /* keep only the rows the variant-caller output prints (writePosteriorProbsVC, impl/signalMachine.c:161-232:
                                     pairs whose reference k-mer holds the ambiguity letter 'X'); every other pair is counted and
                                     summed on the device (sa_batch_all_pairs_summary: what the run's pair count and
                                     scoreByPosteriorProbabilityIgnoringGaps need) and never crosses PCIe.  signalMachine -s 1 sets it.
                                     Not for a batch that feeds sa_batch_mea (the path needs every pair: sa_batch_mea returns SA_ESTATE
                                     for such a batch) and not together with SA_FLAG_PAIRS8 (SA_EINVAL: an 8-byte record does not
                                     name the k-mer the filter looks at). */
#define SA_FLAG_PAIRS8 64u         /* the batch holds its pairs as 8-byte records (sa_pair8_t below) instead of 16-byte ones: half the
                                     bytes over PCIe for results of hundreds of millions of pairs (broad HDP densities at a low
                                     threshold).  Only for batches with ONE path per cell (no ambiguity letters: the pair's k-mer is
                                     the reference's at x, the path index 0 -- neither is stored) and fewer than 2^20 reference
                                     positions and events per job: SA_EUNSUPPORTED otherwise.  Read with sa_batch_pairs8 /
                                     sa_batch_pairs8_all; the 16-byte accessors, sa_batch_pairs(_all) and sa_batch_mea return
                                     SA_ESTATE on such a batch. */
#define SA_FLAG_INPUTS_IN_HOST_BLOCK 16u /* every job's `events`, `anchor_x` and `anchor_y` point into ONE block from sa_host_alloc
                                     (8-byte aligned inside it).  The library then does not read them on the host at all: the
                                     part of the block that holds them crosses PCIe with one DMA and a kernel checks the
                                     anchors, narrows them and gathers the event means out of their records -- what
                                     sa_batch_create otherwise does with host cores (416 MB per 2000 reads of 50 000 events
                                     in the reference's four-double records: two host threads cannot keep up with the GPU).
                                     A block laid out with all event records apart from all anchor arrays is sent in two
                                     pieces, and sa_batch_create returns while the event records are still travelling (on
                                     the batch's own stream, ahead of its kernels): the block must stay unchanged until the
                                     batch has RUN (sa_batch_run / sa_batch_wait has returned) or has been destroyed.  Pointers
                                     outside such a block: SA_EINVAL.  Everything else is as without the flag -- a read the
                                     device checks turn down sends the batch to the host planner, which reads the same
                                     memory and names the error -- and the flag is ignored by batches the host plans
                                     (SA_FLAG_EXACT, ...) and by sa_expect_batch's SA_EMISSION_TWO_DIST models.
                                     Layout: what crosses PCIe (and is held in HBM until the batch goes, counted in
                                     sa_batch_stats_t.device_bytes) is the COVERING RANGE of the batch's arrays -- of its event
                                     records and of its anchor arrays when the two lie apart, of everything otherwise -- so a
                                     batch's arrays should be contiguous in the block.  A batch whose covering ranges hold more
                                     than twice the bytes its jobs name (one arena shared by several batches, gaps) is packed
                                     by host threads instead, as without the flag: same results, none of the saving */

typedef struct sa_model sa_model_t; /* replaces StateMachine3 / StateMachine3_HDP (inc/stateMachine.h:150-190) */
typedef struct sa_batch sa_batch_t;

/* PairwiseAlignmentParameters (inc/pairwiseAligner.h, defaults impl/pairwiseAligner.c:2022-2037) */
typedef struct sa_params {
    double threshold;                      /* -D */
    int64_t diagonal_expansion;            /* -x, must be even (signalMachine rounds up: impl/signalMachine.c:675) */
    int64_t trace_back_diagonals;          /* -g */
    int64_t min_diags_between_trace_back;  /* 1000 */
    int64_t split_matrix_bigger_than_this; /* 3000*3000 */
} sa_params_t;

/* the slice of a deserialised .nhdp that alignment reads (impl/hdp.c:2588-2612, :2777-2806) */
typedef struct sa_hdp_desc {
    int64_t num_dps;
    int64_t grid_length;
    double grid_start, grid_stop;
    const int64_t *parent;        /* num_dps, -1 for the root                      */
    const uint8_t *observed;      /* num_dps (mark_observed_dps, impl/hdp.c:1132)  */
    const double *const *post_pred; /* num_dps pointers, NULL where unobserved     */
    const double *const *slopes;    /* num_dps pointers, NULL where absent         */
} sa_hdp_desc_t;

/* one alignment = one strand of one read: the arguments of getAlignedPairsUsingAnchors() */
typedef struct sa_job {
    const char *ref;        /* target nucleotides (may hold ambiguity letters); lX = ref_len - (k-1) */
    int64_t ref_len;
    const double *events;   /* event i's mean at events[i*event_stride]; already sliced to the guide alignment
                               (makeEventSequenceFromPairwiseAlignment, impl/signalMachine.c:442) and drift-adjusted */
    int64_t event_stride;   /* in doubles: 4 for the NB_EVENT_PARAMS layout, 1 for a dense mean vector */
    int64_t n_events;
    const int64_t *anchor_x; /* remapped + filtered anchors, (k-mer index, event index)  */
    const int64_t *anchor_y;
    int64_t n_anchors;
    double scale, shift, var; /* sM->scale/shift/var for this read (impl/signalMachine.c:755-757) */
    unsigned ends;          /* SA_JOB_*_END_NOT_RAGGED bits: the last two arguments of getAlignedPairsUsingAnchors(...,
                               bool alignmentHasRaggedLeftEnd, bool alignmentHasRaggedRightEnd) (inc/pairwiseAligner.h:406-414,
                               impl/pairwiseAligner.c:2052-2080), INVERTED so that a zeroed job is signalMachine's call (ragged on
                               both sides, impl/signalMachine.c:436-437).  A bit set = that end is NOT ragged: the alignment starts
                               in (0,0) in the match state / ends in (lX,lY) with the end-state transitions
                               (stateMachine3_startStateProb / _endStateProb, impl/stateMachine.c:1134-1173) and getSplitPoints
                               keeps the outer rectangle of a gap that is cut at that end (impl/pairwiseAligner.c:1910-1937);
                               the reference's own known-answer tests call it with (0, 0) (tests/stateMachineTests.c:943-947) */
} sa_job_t;
#define SA_JOB_LEFT_END_NOT_RAGGED 1u   /* alignmentHasRaggedLeftEnd == false  */
#define SA_JOB_RIGHT_END_NOT_RAGGED 2u  /* alignmentHasRaggedRightEnd == false */

/* stIntTuple4(floor(p*1e7), x, y, (char*)pathKmer)  (impl/pairwiseAligner.c:1405-1408) */
typedef struct sa_pair {
    int64_t prob_e7;
    int32_t x, y;
    int32_t path;    /* index of the path inside the cell (order of hdCell_construct2) */
    int32_t kmer_id; /* kmer_id() of the path's k-mer                                  */
} sa_pair_t;

/* The same tuple as it lives in HBM and crosses PCIe: 16 bytes instead of 24 (2000 x 5000-event reads return 9 million pairs
 * per batch; broad HDP densities at threshold 0.01 four hundred million).  This is what a finished batch HOLDS, in pinned host
 * memory, job after job in output order: sa_batch_pairs16 hands out a job's records in place (no copy), sa_batch_pairs and
 * sa_batch_pairs_all expand them into sa_pair_t.
 *   a = x (28 bits) | y (28 bits) << 28 | path bits 0..7 << 56        (the planners refuse matrices of 2^28 rows or columns
 *   b = kmer_id (32 bits) | prob_e7 (24 bits: <= 1e7) << 32 | path bits 8..15 << 56        and cells of more than 65535 paths) */
typedef struct sa_pair16 {
    uint64_t a, b;
} sa_pair16_t;
#if defined(__HIPCC__) || defined(__HIP__)
#define SA_PAIR16_ATTR __host__ __device__
#else
#define SA_PAIR16_ATTR
#endif
#define SA_PAIR16_FN static inline
#define SA_PAIR16_MAX_COORD (1ll << 28) /* x, y below this */
#define SA_PAIR16_MAX_PATHS 65536       /* paths per cell at most this */
SA_PAIR16_ATTR SA_PAIR16_FN sa_pair16_t sa_pair16_pack(int64_t prob_e7, int32_t x, int32_t y, int32_t path, int32_t kmer_id) {
    sa_pair16_t r;
    r.a = ((uint64_t) (uint32_t) x & 0xfffffffull) | (((uint64_t) (uint32_t) y & 0xfffffffull) << 28) |
          ((uint64_t) ((uint32_t) path & 0xffu) << 56);
    r.b = (uint64_t) (uint32_t) kmer_id | (((uint64_t) prob_e7 & 0xffffffull) << 32) |
          ((uint64_t) (((uint32_t) path >> 8) & 0xffu) << 56);
    return r;
}
SA_PAIR16_ATTR SA_PAIR16_FN sa_pair_t sa_pair16_unpack(sa_pair16_t r) {
    sa_pair_t o;
    o.x = (int32_t) (r.a & 0xfffffffull);
    o.y = (int32_t) ((r.a >> 28) & 0xfffffffull);
    o.path = (int32_t) (((r.a >> 56) & 0xffull) | (((r.b >> 56) & 0xffull) << 8));
    o.kmer_id = (int32_t) (uint32_t) (r.b & 0xffffffffull);
    o.prob_e7 = (int64_t) ((r.b >> 32) & 0xffffffull);
    return o;
}

/* SA_FLAG_PAIRS8: prob_e7 (24 bits: <= 1e7) << 40 | y (20 bits) << 20 | x (20 bits) */
typedef uint64_t sa_pair8_t;
#define SA_PAIR8_MAX_COORD (1ll << 20)
SA_PAIR16_ATTR SA_PAIR16_FN sa_pair8_t sa_pair8_pack(int64_t prob_e7, int32_t x, int32_t y) {
    return ((uint64_t) prob_e7 << 40) | (((uint64_t) (uint32_t) y & 0xfffffull) << 20) | ((uint64_t) (uint32_t) x & 0xfffffull);
}
SA_PAIR16_ATTR SA_PAIR16_FN void sa_pair8_unpack(sa_pair8_t r, int64_t *prob_e7, int32_t *x, int32_t *y) {
    *x = (int32_t) (r & 0xfffffull);
    *y = (int32_t) ((r >> 20) & 0xfffffull);
    *prob_e7 = (int64_t) (r >> 40);
}

/* Test hook (host only): `in` through the packed 16-byte record and back into `out`. */
int sa_pair_roundtrip(const sa_pair_t *in, sa_pair_t *out, int64_t n);

typedef struct sa_batch_stats {
    double cells_forward;    /* sum over forward diagonals of width*paths           */
    double cells_backward;   /* ditto for backward diagonals actually computed      */
    double ms_forward;       /* HIP-event time of the forward kernels, last run     */
    double ms_backward;      /* traceback stage: end of the forward sweep -> end of the last backward+posterior
                                kernel (the groups' launches overlap on two streams; fold/finalisation kernels of
                                earlier groups run inside this window)                                        */
    double ms_fold;          /* rest of the pass: fold (+ on-device finalisation) of the last group          */
    double ms_total_device;  /* first launch -> last kernel end                     */
    double f_bytes;          /* bytes of forward storage written (== read back)     */
    int64_t n_regions, n_segments, n_checkpoints;
    int64_t n_fast_regions;  /* regions handled by the register-resident kernels    */
    int64_t n_chunks;        /* passes needed to fit forward storage in HBM         */
    int64_t n_groups;        /* backward/posterior launches per pass (result copy of one overlaps the next) */
    int64_t n_ring_regions;  /* regions handled by the LDS-ring kernels (several paths per cell, or a band mostly wider
                                than a wave)                                           */
    int64_t n_strip_regions; /* of those: one-path regions swept in strips of 64 reference columns (register-resident,
                                no barrier; sa_strip.inc)                               */
    double device_bytes;     /* working storage of the batch in HBM: forward planes, emission plane (HDP), candidate and
                                result slots, checkpoint buffers -- what a caller sizes its pipeline depth with */
} sa_batch_stats_t;

/* ---- model -------------------------------------------------------------------------------------
 * Replaces stateMachine3_loadFromFile()'s result (impl/stateMachine.c:1440-1538):
 *   transitions10: the 10 tokens of the transition line, linear space;
 *   table5: 5*A^k doubles [level_mean level_sd noise_mean noise_sd noise_lambda] in kmer_id order.
 * hdp != NULL selects the HDP emission (stateMachine3HDP_cellCalculate, impl/stateMachine.c:1371);
 * the caller applies stateMachine3_setModelToHdpExpectedValues with sa_model_set_to_hdp_expected_values. */
int sa_model_create(sa_model_t **out, int n_states, const char *alphabet, int k, const double *transitions10,
                    const double *table5, const sa_hdp_desc_t *hdp);
int sa_model_load(sa_model_t **out, const char *model_path, const char *nhdp_path_or_null);
void sa_model_destroy(sa_model_t *m);
int sa_model_alphabet(const sa_model_t *m, char *out64, int *n_alpha, int *k);
const double *sa_model_table5(const sa_model_t *m);        /* EMISSION_MATCH_MATRIX view */
int sa_model_set_to_hdp_expected_values(sa_model_t *m);    /* impl/stateMachine.c:1275-1304 */
int64_t sa_kmer_id(const sa_model_t *m, const char *kmer); /* impl/nanopore_hdp.c:405-410 */
/* Which match / gapY emission a Gaussian model uses (the function pointers stateMachine3_construct takes, inc/stateMachine.h):
 *   SA_EMISSION_MEAN_ONLY  emissions_signal_strawManGetKmerEventMatchProbWithDescaling_MeanOnly (impl/stateMachine.c:557-605),
 *                          what signalMachine installs (impl/signalMachine.c:337); every kernel family
 *   SA_EMISSION_TWO_DIST   emissions_signal_strawManGetKmerEventMatchProbWithDescaling (:607-650): Gaussian on the descaled mean
 *                          x inverse Gaussian on the event noise -- what the reference's shipped output files were written with
 *                          (tests/test_oracle_reference_outputs.py).  Register kernels (round 6: k_fwd_fast_two / k_bwd_fast_two) when
 *                          every region of the batch holds one path per cell (sa_align_batch / sa_batch_*; not the expectation
 *                          pass); otherwise, and with SA_FLAG_EXACT, the reference-ordered memory-resident kernels (the batch then
 *                          behaves as with SA_FLAG_EXACT); jobs must hand over event records (event_stride >= 2: the
 *                          noise is a record's second value); the noise columns are the MODEL's, so the reads of a batch share
 *                          one noise scaling (the reference rescales them per read, emissions_signal_scaleNoise: create the
 *                          model from the rescaled table sa_estimate_params leaves). */
#define SA_EMISSION_MEAN_ONLY 0
#define SA_EMISSION_TWO_DIST 1
/*   SA_EMISSION_TWO_DIST_SCALED_MODEL  emissions_signal_strawManGetKmerEventMatchProb (:659-700): the same two distributions on the
 *                          event mean AS IT IS -- no descaling, the job's scale / shift / var are not read; the model table was scaled
 *                          to the read instead (emissions_signal_scaleModel :743-779).  What getStateMachine3 installs (:1757-1765):
 *                          the emission of the reference's literal-data known answers (tests/stateMachineTests.c:441-698) and of its
 *                          scaled-model whole-read test (:842-852).  Same kernels and conditions as SA_EMISSION_TWO_DIST. */
#define SA_EMISSION_TWO_DIST_SCALED_MODEL 2
int sa_model_set_emission(sa_model_t *m, int emission);
/* A Gaussian model with the same alphabet, k-mer length, transitions and emission kind as `m` and the emission table `table5`
 * (5 * A^k doubles, copied): the per-read model the reference gets from emissions_signal_scaleNoise (impl/stateMachine.c:721-741)
 * -- sa_estimate_params leaves the rescaled table in its table5_inout argument. */
int sa_model_clone_with_table(sa_model_t **out, const sa_model_t *m, const double *table5);

/* ambiguity table: 256 entries (index = character), NULL = not ambiguous.
 * sa_default_ambig fills create_ambig_bases() (impl/pairwiseAligner.c:32-65);
 * sa_load_ambig reads the -a file (create_ambig_bases2, impl/pairwiseAligner.c:68-92); strings are
 * owned by a static/heap pool that lives until the process ends. */
void sa_default_ambig(const char **map256);
int sa_load_ambig(const char *path, const char **map256);

/* ---- batched getAlignedPairsUsingAnchors (impl/pairwiseAligner.c:2052-2080) --------------------
 * create: host-side planning (split regions, band tables, traceback segments) + upload to HBM.
 * run:    forward, backward/posterior and fold kernels; inputs are already resident.
 * pairs:  rows in the order signalMachine writes them (stable sort by x+y of the reference's list). */
int sa_batch_create(sa_batch_t **out, const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs,
                    int64_t n_jobs, const char *const *ambig256, int device, unsigned flags);
int sa_batch_run(sa_batch_t *b);
/* sa_batch_create in two halves, for a caller that streams batches (sa_batch_start / sa_batch_wait below).  The first half --
 * input checks, packing and upload of the reads, the planning kernels queued -- runs here; the second -- waiting for the plan,
 * working buffers, launch lists -- on the batch's first use (sa_batch_run or the thread of sa_batch_start, sa_batch_stats,
 * sa_batch_job_cells), so the calling thread does not wait for the GPU and can pack the next batch.  Same arguments and
 * results as sa_batch_create; what differs: `jobs` and `ambig256` must stay valid until that first use has returned (a read
 * the planning kernels turn down sends the batch to the host planner, which reads them again), and an error found by the
 * second half is returned by that first use instead.  A batch the planning kernels do not take at all (an ambiguity letter
 * with repeated options, SA_FLAG_EXACT, ...) is created completely, as by sa_batch_create. */
int sa_batch_create_deferred(sa_batch_t **out, const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs,
                             int64_t n_jobs, const char *const *ambig256, int device, unsigned flags);
/* Optional, between the two halves: waits for the plan and builds the launch lists -- the part of the second half that needs no
 * working storage -- so that a caller whose batches take the device one at a time (SA_FLAG_DEVICE_TO_ITSELF) can have it done
 * while the batch before is still running; the batch's first use then only takes its buffers.  Counts as a first use for the
 * lifetime of `jobs` / `ambig256`; returns what that part of the second half returns (and the first use returns it again). */
int sa_batch_prepare(sa_batch_t *b);
/* The same on a thread of the library's own: sa_batch_start returns at once, sa_batch_wait returns sa_batch_run's code.
 * Lets one caller thread plan the next batch (sa_batch_create is host work) while this one is on the GPU. */
int sa_batch_start(sa_batch_t *b);
int sa_batch_wait(sa_batch_t *b);
int sa_batch_n_pairs(const sa_batch_t *b, int64_t job, int64_t *n);
int sa_batch_pairs(const sa_batch_t *b, int64_t job, sa_pair_t *out, int64_t cap);
/* A job's number of pairs and the sum of their prob_e7 over ALL pairs above the threshold -- with SA_FLAG_VC_ROWS including the rows
 * that were dropped on the device (without the flag: of the rows the batch holds).  100 * sum / (n * 1e7) is
 * scoreByPosteriorProbabilityIgnoringGaps (impl/pairwiseAligner.c:407-412). */
int sa_batch_all_pairs_summary(const sa_batch_t *b, int64_t job, int64_t *n_all, int64_t *sum_prob_e7);
/* A job's pairs as the batch holds them: *out points at *n packed records (sa_pair16_t above) inside the batch's pinned result
 * block, valid until the batch is run again or destroyed.  Nothing is copied: this is where sa_batch_run / sa_batch_wait
 * leaves the results, and the form a caller that formats or forwards them reads (signalMachine's TSV writers, sa_batch_mea). */
int sa_batch_pairs16(const sa_batch_t *b, int64_t job, const sa_pair16_t **out, int64_t *n);
/* The whole batch at once: the records of all jobs are contiguous in job order; *out is the first record of job 0 and job j's
 * are (*out)[first[j] .. first[j + 1]) (`first`: n_jobs + 1 entries, may be NULL). */
int sa_batch_pairs16_all(const sa_batch_t *b, const sa_pair16_t **out, int64_t *first);
/* the same two for a batch created with SA_FLAG_PAIRS8 (8-byte records, in place, job after job) */
int sa_batch_pairs8(const sa_batch_t *b, int64_t job, const sa_pair8_t **out, int64_t *n);
int sa_batch_pairs8_all(const sa_batch_t *b, const sa_pair8_t **out, int64_t *first);
/* Every job's pairs expanded into sa_pair_t, job after job, on the library's host threads: out[first[j] .. first[j + 1]) are
 * job j's rows (`first` has n_jobs + 1 entries; may be NULL).  cap < total number of pairs: SA_EINVAL and *first is still
 * filled, so first[n_jobs] says how much room is needed. */
int sa_batch_pairs_all(const sa_batch_t *b, sa_pair_t *out, int64_t cap, int64_t *first);
/* A finished batch's working storage in HBM (everything sa_batch_stats_t.device_bytes counts) back to the library's allocator,
 * the results kept: the packed pairs in pinned host memory, the per-job offsets and the statistics stay readable until
 * sa_batch_destroy.  For a caller that holds batches for their results while further ones are created (signalMachine --twoD
 * keeps the template strand's batch while the complement's runs).  Afterwards sa_batch_run / sa_batch_start: SA_ESTATE;
 * sa_batch_mea still works (the pairs are uploaded again).  Before the batch has run, or between sa_batch_start and
 * sa_batch_wait: SA_ESTATE. */
int sa_batch_release_device(sa_batch_t *b);
int sa_batch_stats(const sa_batch_t *b, sa_batch_stats_t *out);
int sa_batch_job_cells(const sa_batch_t *b, int64_t job, double *cells_fwd, double *cells_bwd);
void sa_batch_destroy(sa_batch_t *b);

/* one-shot convenience: create + run + copy out.  pairs_out[j] is malloc'd (sa_free). */
int sa_align_batch(const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs,
                   const char *const *ambig256, int device, unsigned flags, sa_pair_t **pairs_out,
                   int64_t *n_pairs_out);

/* ---- batched getExpectationsUsingAnchors (impl/pairwiseAligner.c:2164-2184) --------------------
 * trans9_out[j*9 + from*3 + to] and likelihood_out[j] are ADDED to (the caller seeds pseudocounts, as
 * hmmContinuous_getExpectationsHmm does); HDP assignments (to==match && p>=threshold,
 * impl/pairwiseAligner.c:946-968) come back as (reference position, event index) pairs. */
typedef struct sa_assignment {
    int64_t ref_pos;   /* index into job.ref of the cell's k-mer pointer (cX) */
    int64_t event;     /* index into the job's event array (cY)               */
} sa_assignment_t;
int sa_expect_batch(const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs,
                    const char *const *ambig256, int device, unsigned flags, double *trans9_out,
                    double *likelihood_out, sa_assignment_t **assign_out, int64_t *n_assign_out);

/* Test / measurement hook: the batch statistics (regions per kernel family, kernel times, passes) of the calling thread's last
 * sa_expect_batch. */
int sa_expect_last_stats(sa_batch_stats_t *out);

/* ---- planning introspection (host only, no GPU needed; used by the CPU test-suite) --------------
 * Builds the plan for ONE job and reports its geometry. */
typedef struct sa_plan_info {
    int64_t n_regions, n_segments, n_checkpoints;
    double cells_forward, cells_backward;
    int64_t f_cellpaths;   /* cell-paths of forward storage */
    int64_t max_span;      /* widest 3-row window in (x-y)/2 units          */
    int64_t n_fast_regions;
    int64_t n_ring_regions;
} sa_plan_info_t;
int sa_plan_describe(const sa_model_t *m, const sa_params_t *p, const sa_job_t *job, const char *const *ambig256,
                     unsigned flags, sa_plan_info_t *info,
                     int64_t *regions4_out, int64_t regions_cap,   /* x1,y1,x2,y2 per region             */
                     int64_t *rows3_out, int64_t rows_cap,         /* region,xmyL,xmyR per diagonal      */
                     int64_t *segs4_out, int64_t segs_cap);        /* region,start,from,to per traceback */

/* Test hook (host only): the per-path neighbour records the planner writes for regions with ambiguous positions, checked
 * against path_checkLegal (impl/pairwiseAligner.c:595-621) pair by pair.  Returns the number of wrong entries (0 = all
 * good) or a negative SA_E* code; *n_checked = entries examined. */
int64_t sa_plan_check_path_records(const sa_model_t *m, const sa_params_t *p, const sa_job_t *job,
                                   const char *const *ambig256, int64_t *n_checked);

/* Test hook (needs a GPU): plans the batch on the device and on the host and compares every array the kernels read.
 * 0: identical; > 0: bit mask of the arrays that differ (1 << 30: the batch is not one the device planner takes);
 * < 0: SA_E* code. */
int sa_dplan_compare(const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs,
                     const char *const *ambig256, int device, unsigned flags);

/* ---- event <-> k-mer pre-alignment (the step upstream of the pair-HMM) ------------
 * adaptive_banded_simple_event_align (impl/eventAligner.c:899-1235): adaptive banded Viterbi of a raw event table
 * against the k-mers of the basecalled sequence, with the state machine's MeanOnly match emission
 * (impl/eventAligner.c:1245-1247).  One wave per read on the GPU.  pairs_out[j] (malloc'd, sa_free) holds the
 * struct AlignedPair list {ref_pos = k-mer index, read_pos = event index} in ascending order; it is empty and
 * status_out[j] != 0 when the reference would have rejected the alignment (bit 0: average log emission < -5.2, bit 1:
 * first/last k-mer not reached, bit 2: more than 50 skipped k-mers in a row, bit 3: more than 5 events per k-mer).
 * Parity of this entry point is pinned by the CPU restatement only (the reference's tests of it need fast5 files). */
typedef struct sa_ea_job {
    const char *sequence;      /* nucleotides; a k-mer at every position (build_kmer_list, impl/eventAligner.c:755-782) */
    int64_t seq_len;
    const double *event_mean;  /* event_t.mean of the raw event table                                                  */
    int64_t n_events;
    double scale, shift, var;  /* update_SignalMachineWithNanoporeParameters (:845-849); see sa_scalings_mom           */
} sa_ea_job_t;
typedef struct sa_ea_pair {
    int32_t kmer_idx, event_idx;
} sa_ea_pair_t;
/* estimate_scalings_using_mom (impl/eventAligner.c:784-843): method of moments; the reference then uses var = 1 */
int sa_scalings_mom(const sa_model_t *m, const char *sequence, int64_t seq_len, const double *event_mean,
                    int64_t n_events, unsigned flags, double *shift_out, double *scale_out);
/* cells_out[j] (may be NULL): band cells filled for job j (the reference's `fills`); kernel_ms_out (may be NULL):
 * HIP-event time of the kernel */
int sa_event_align_batch(const sa_model_t *m, const sa_ea_job_t *jobs, int64_t n_jobs, int device, unsigned flags,
                         sa_ea_pair_t **pairs_out, int64_t *n_pairs_out, int32_t *status_out, double *cells_out,
                         double *kernel_ms_out);
/* sa_event_align_batch keeps its device and pinned-host scratch between calls (grow only, one workspace per process,
 * calls serialise on it); this returns the memory.  Safe to call at any time, also when nothing is held. */
void sa_event_align_release(void);

/* fastaHandler_getSubSequence (impl/fasta_handler.c:15-44, htslib faidx underneath): bases [start, end) of the record
 * `name` (the header's first word) of a FASTA file; strand == 0 asks htslib for [end, start - 1] as the reference does.
 * Uses <path>.fai when it exists, otherwise scans the file (nothing is written).  *out is freed with sa_free.
 * SA_EIO: file unreadable; SA_EINVAL: no such record. */
int sa_fasta_subsequence(const char *fasta_path, const char *name, int64_t start, int64_t end, int strand, char **out);
/* printf("%f") of v, character for character (the exact binary value rounded to six decimals, ties to even -- glibc's result),
 * without the stdio formatter: what the TSV writers (impl/signalMachine.c:89-270: nine "%f" per aligned pair) spend their time
 * in.  `out` needs 32 bytes for |v| < 9e15 (larger values, inf and nan go through snprintf: up to 320); a terminator is
 * written; returns the length. */
int sa_format_f6(char *out, double v);

/* Batches take their device and pinned-host storage from a caching allocator: what a destroyed batch held is kept and
 * handed to the next one (a pipeline that sees every read once creates and destroys a batch per few thousand reads;
 * allocation and release of its 25 GB cost more than its kernels).  What may stay PARKED between batches is bounded:
 * by default 90 % of the device's memory and 32 GB of pinned host memory (parked device blocks are handed back to the
 * runtime whenever an allocation of the process fails, so the cache never causes an out-of-memory by itself).
 * An embedding caller that shares the device or the host with other code sets its own bounds:
 *   sa_pool_configure(device_limit_bytes, pinned_limit_bytes)   a negative value leaves that bound as it is; 0 keeps nothing
 *       parked; blocks above a lowered bound are freed at once.  Wins over the environment.
 *   sa_pool_release()                                           returns everything that is parked right now.
 *   sa_pool_release_device()                                    ... the device blocks only (page-locked blocks stay parked)
 * sa_host_alloc / sa_host_free: page-locked host memory for a caller's own input arrays (SA_FLAG_INPUTS_IN_HOST_BLOCK); a
 * block belongs to the caller until sa_host_free, which must not be called before every batch created from it has run or has
 * been destroyed.
 * Environment (read when no limit was configured): SA_POOL=0 disables the cache, SA_POOL_LIMIT_GB bounds both kinds. */
int sa_pool_configure(int64_t device_limit_bytes, int64_t pinned_limit_bytes);
void *sa_host_alloc(size_t bytes);   /* NULL: no memory (or no device) */
void sa_host_free(void *block);      /* NULL or a pointer sa_host_alloc did not return: ignored */
void sa_pool_release(void);
void sa_pool_release_device(void);

/* ---- maximum-expected-accuracy path over a read's posteriors ----------------------------------
 * Replaces maximum_expected_accuracy_alignment + get_indexes_from_best_path (src/signalalign/mea_algorithm.py:25-197,
 * :248-264; called by mea_alignment_from_signal_align :323-341): the best monotone path through the sparse posterior
 * matrix, one (reference position, event) pair per event, bit-identical to the reference's choice among ties.
 * A job is what that function receives: the COO form of posterior_matrix (row = event, column = reference position,
 * row-major as scipy.sparse.coo_matrix(dense) yields it) and shortest_ref_per_event (INT32_MAX where the reference
 * holds inf).  sa_mea_params builds both from the columns of a signalAlign event table.  Pinned by the reference's
 * known-answer matrix (tests/golden/mea/kat_5x5.json). */
#define SA_MEA_OK 0
#define SA_MEA_EMPTY 1        /* no entries: the reference raises ValueError (min of an empty sequence, :42)   */
#define SA_MEA_SINGLE_EVENT 2 /* every entry belongs to the first event: IndexError at :61                      */
#define SA_MEA_NO_FRONT 3     /* an event starts with no forward edge left: IndexError at :106                  */
#define SA_MEA_NO_PATH 4      /* no final edge with a sum above 0: the reference returns the int 0 (:188-196)   */
#define SA_MEA_BAD_EVENT 5    /* event index outside shortest_ref_per_event: IndexError at :106                 */
#define SA_MEA_INF 2147483647 /* shortest_ref_per_event of an event without rows                                */
typedef struct sa_mea_job {
    const int32_t *event_idx;               /* COO row    */
    const int32_t *ref_idx;                 /* COO column */
    const double *posterior;                /* COO data   */
    int64_t n;
    const int32_t *shortest_ref_per_event;  /* indexed by event_idx */
    int64_t n_events;
} sa_mea_job_t;
typedef struct sa_mea_pair {
    int32_t ref_idx, event_idx;             /* get_indexes_from_best_path: [ref_pos, event_pos] */
} sa_mea_pair_t;
/* One wave per read.  path_out[j] (malloc'd, sa_free) holds n_path_out[j] pairs in path order; status_out[j] one of
 * SA_MEA_* (a failed read has no path and does not fail the call); sum_out[j] the best edge's sum; n_edges_out[j] the
 * number of final forward edges (return_all=True); any of the last four pointers may be NULL. */
int sa_mea_batch(const sa_mea_job_t *jobs, int64_t n_jobs, int device, unsigned flags, sa_mea_pair_t **path_out,
                 int64_t *n_path_out, double *sum_out, int32_t *status_out, int32_t *n_edges_out, double *kernel_ms_out);
void sa_mea_release(void); /* returns the scratch sa_mea_batch and sa_batch_mea keep between calls */
/* The same step chained onto a finished batch (after sa_batch_run), as mea_alignment_from_signal_align
 * (mea_algorithm.py:323-341) chains it onto signalAlign's output: every read's aligned pairs are still in HBM, one wave
 * per read builds the posterior matrix and shortest_ref_per_event from them there (what get_mea_params_from_events does
 * with the event table; the posterior is the one the TSV prints, six decimals) and the path kernels follow; only the
 * paths come back.  path_out[j] holds (ref_idx = x, event_idx = y) in the coordinates of sa_pair_t.  One entry per job
 * of the batch in every output array. */
int sa_batch_mea(sa_batch_t *b, unsigned flags, sa_mea_pair_t **path_out, int64_t *n_path_out, double *sum_out,
                 int32_t *status_out, double *kernel_ms_out);
/* "%f" of prob_e7 / 1e7 read back as a double: the posterior_probability column of the event table */
double sa_mea_printed_posterior(int64_t prob_e7);
/* the same, evaluated by the GPU for prob_e7 = first .. first + n - 1 (test hook: host and device must agree) */
int sa_mea_printed_posterior_device(int64_t first, int64_t n, double *out, int device);
/* get_mea_params_from_events (mea_algorithm.py:267-320), host side, sparse: from the reference_index, event_index and
 * posterior_probability columns of an event table (any row order) to the COO entries and shortest_ref_per_event.
 * The outputs need room for n entries / (max event - min event + 1) events; returns the number of COO entries and the
 * number of events in *n_events_out, or a negative SA_E* code. */
int64_t sa_mea_params(const int64_t *reference_index, const int64_t *event_index, const double *posterior, int64_t n,
                      int32_t *event_idx_out, int32_t *ref_idx_out, double *posterior_out, int32_t *shortest_out,
                      int64_t *n_events_out);

/* Plans a whole batch on the host (no GPU needed) with `threads` planner threads (0 = as sa_batch_create would) and
 * returns aggregate geometry plus a 64-bit FNV-1a digest over every array that would be uploaded.  The digest must
 * not depend on the number of threads: the CPU test-suite checks exactly that. */
int sa_plan_digest(const sa_model_t *m, const sa_params_t *p, const sa_job_t *jobs, int64_t n_jobs,
                   const char *const *ambig256, unsigned flags, int threads, sa_plan_info_t *info, uint64_t *digest);

/* ---- host-side helpers that signalMachine needs around the seam --------------------------------- */
/* signalUtils_guideAlignmentToRebasedAnchorPairs (impl/signalMachineUtils.c:142-164). op types:
 * 0 = match, 1 = reference-only (PAIRWISE_INDEL_X), 2 = read-only (PAIRWISE_INDEL_Y). */
int64_t sa_guide_to_anchors(int64_t start1, int64_t end1, int strand1, int64_t start2, const int32_t *op_type,
                            const int64_t *op_len, int64_t n_ops, int64_t trim, int64_t *ax, int64_t *ay,
                            int64_t cap);
/* signalUtils_getRemappedAnchorPairs (impl/signalMachineUtils.c:166-170) */
int64_t sa_remap_anchors(const int64_t *ax, const int64_t *ay, int64_t n, const int64_t *event_map,
                         int64_t map_offset, int64_t *ox, int64_t *oy);
/* signalUtils_estimateNanoporeParams (impl/signalMachineUtils.c:186-225): events (4 doubles each) are
 * drift-corrected in place, table5 noise columns rescaled; out7 = scale shift var drift scale_sd var_sd shift_sd */
int sa_estimate_params(const sa_model_t *m, double *table5_inout, const int64_t *strand_event_map, double *events4,
                       int64_t n_events, const char *strand_read, int64_t read_len, double *out7);

/* ---- HDP rebuild, the deterministic pieces --------------------------------------------------------
 * The state of a serialised NanoporeHDP as the reference's Gibbs sampler leaves it, and what is computed FROM a state without
 * random numbers.  The sampling sweep itself (sample_dp_factors / gibbs_factor_iteration, impl/hdp.c:2110-2260, rand()-driven)
 * is not part of this library.
 *   sa_hdp_state_load / _write   deserialize_nhdp + deserialize_hdp / serialize_nhdp + serialize_hdp
 *                                (impl/nanopore_hdp.c:1077-1115, impl/hdp.c:2868-3322): host code; a file the reference wrote
 *                                goes through load + write byte for byte
 *   sa_hdp_state_distr_sample    take_distr_sample (impl/hdp.c:2067-2092): what ONE sample of the state adds to every observed
 *                                DP's collector -- the posterior predictive of every base factor on the sampling grid
 *                                (evaluate_posterior_predictive :530-562), the prior's (evaluate_prior_predictive :564-585),
 *                                mixed with the weights of cache_base_factor_weight / cache_prior_contribution (:2001-2044).
 *                                Grid evaluation and mixing run on the GPU (observed DPs x grid points x factors).
 *   sa_hdp_finalize_distributions  finalize_distributions (impl/hdp.c:2551-2584): collector / samples, then the slopes of the
 *                                natural cubic spline through it (spline_knot_slopes, impl/hdp_math_utils.c:402-442), one DP
 *                                per GPU thread, the reference's elimination order (bit-identical slopes for the densities of a
 *                                file the reference wrote).
 * Errors: SA_EIO (unreadable / malformed file), SA_ESTATE (a state without data has no factors), SA_ENODEVICE, SA_ENOMEM. */
typedef struct sa_hdp_state sa_hdp_state_t;
typedef struct sa_hdp_state_info {
    int64_t num_dps, depth, grid_length, n_data, n_factors, n_base_factors, n_observed, base_dp, alphabet_size, kmer_length;
    double mu, nu, alpha, beta, grid_start, grid_stop;
    int splines_finalized, has_data, sample_gamma;
    /* views into the state (valid until sa_hdp_state_free) */
    const double *data;                 /* n_data                                                        */
    const int64_t *data_dp;             /* n_data: the leaf DP of every data point                       */
    const double *gamma;                /* depth: concentration by depth of the DP                       */
    const double *grid;                 /* grid_length (linspace, impl/hdp_math_utils.c:497-510)         */
    const int64_t *dp_parent;           /* num_dps, -1 for the base DP                                   */
    const int64_t *dp_num_factor_children, *dp_depth;
    const uint8_t *observed;            /* num_dps (mark_observed_dps, impl/hdp.c:1132-1160)             */
    const int64_t *row_of_dp;           /* num_dps: row of an observed DP in post / slope, -1 otherwise  */
    const double *post, *slope;         /* n_observed x grid_length                                      */
    const int64_t *f_type, *f_parent;   /* n_factors, tree order: 0 base / 1 middle / 2 data point       */
    const int64_t *f_ref;               /* the factor's DP, or its data index for a data point           */
    const double *f_params;             /* 5 per factor (base factors): mu nu two_alpha beta log-term    */
    const int64_t *f_n_children;
} sa_hdp_state_info_t;
int sa_hdp_state_load(sa_hdp_state_t **out, const char *nhdp_path);
int sa_hdp_state_write(const sa_hdp_state_t *s, const char *nhdp_path);
int sa_hdp_state_info(const sa_hdp_state_t *s, sa_hdp_state_info_t *info);
void sa_hdp_state_free(sa_hdp_state_t *s);
/* out: n_observed x grid_length (rows as sa_hdp_state_info_t.row_of_dp), overwritten */
int sa_hdp_state_distr_sample(const sa_hdp_state_t *s, int device, double *out);
/* The host half of sa_hdp_state_distr_sample on its own: the weights of one sample as CSR over the observed DPs' rows --
 * row r holds entries [row_start[r], row_start[r + 1]) of (col, w); col < n_base_factors names a base factor (in tree order),
 * col == n_base_factors the prior; entries of a row are in the order the reference adds them.  The three arrays are the
 * library's (sa_free). */
int sa_hdp_state_sample_weights(const sa_hdp_state_t *s, int64_t **row_start, int64_t **col, double **w, int64_t *nnz);
/* sum: n_rows x grid_length collectors after `samples` samples; y_out = sum / samples, slope_out the spline slopes (both n_rows x
 * grid_length; y_out may be NULL) */
int sa_hdp_finalize_distributions(const double *grid, int64_t grid_length, const double *sum, int64_t n_rows, int64_t samples,
                                  int device, double *y_out, double *slope_out);

/* ---- the expectations objects of the EM loop, host only -----------------------------------------
 * Hmm / ContinuousPairHmm / HdpHmm (inc/stateMachine.h:64-83, inc/continuousHmm.h:8-75, impl/continuousHmm.c): what
 * getExpectationsUsingAnchors accumulates into, the .expectations files trainModels.py exchanges, and the M-step.
 *   sa_hmm_create            hmmContinuous_getExpectationsHmm (:841-856): an empty accumulator that "fits" the model -- its event
 *                            model loaded (hmmContinuous_loadEventModel), transitions at the transition pseudocount, and for
 *                            SA_HMM_GAUSSIAN (continuousPairHmm_construct :83-144) k-mer posteriors at the emission pseudocount;
 *                            SA_HMM_HDP (hdpHmm_constructEmpty :522-569) carries the assignment threshold and the assignment lists
 *   sa_hmm_add_expectations  adds one read's transition sums and likelihood -- sa_expect_batch's trans9_out / likelihood_out --
 *                            as hmm_addToTransitionsExpectation (:147) and `hmm->likelihood +=` do cell by cell
 *   sa_hmm_add_emission_expectation  continuousPairHmm_addToEmissionExpectation (:159-168)
 *   sa_hmm_add_assignment    hdpHmm_addToAssignment (:510-514): `kmer` points at k characters (not terminated)
 *   sa_hmm_write             continuousPairHmm_writeToFile (:353-407) / hdpHmm_writeToFile (:571-628), byte for byte; a NaN
 *                            transition leaves an empty file (hmmContinuous_checkTransitions)
 *   sa_hmm_load              continuousPairHmm_loadFromFile (:409-507) / hdpHmm_loadFromFile (:630-785): header, transitions and
 *                            likelihood, event model (the reference reads no further for a ContinuousPairHmm: the accumulators of
 *                            a loaded object are empty, at the pseudocounts), and for SA_HMM_HDP the two assignment lines.  What the
 *                            reference answers with st_errAbort is SA_EIO here
 *   sa_hmm_add_expectations_file  HMM.add_expectations_file (src/signalalign/hiddenMarkovModel.py:424-486), trainModels.py's accumulating
 *                            reader: a read's .expectations file added to this object -- transitions and likelihood, and all of a
 *                            ContinuousPairHmm file's accumulator lines (expectations and posteriors summed, the mask or-ed) or an
 *                            HdpHmm file's assignments; an empty or malformed file adds nothing (SA_EIO)
 *   sa_hmm_normalize         continuousPairHmm_normalize (:282-308) = hmmDiscrete_normalizeTransitions (impl/discreteHmm.c:125-137)
 *                            + the event model of every observed k-mer from its expectations; SA_HMM_HDP: the transitions
 *   sa_hmm_load_into_model   the M-step, continuousPairHmm_loadTransitionsIntoStateMachine (:320-338) and (SA_HMM_GAUSSIAN, Gaussian
 *                            model) continuousPairHmm_loadEmissionsIntoStateMachine (:340-351).  gapY -> gapX stays at log 0 as in a
 *                            machine loaded from a .model file, and the gapY table stays 1.75 x the match sd (the reference's :347
 *                            writes it to the wrong index): signalalign_amd/csrc/sa_hmm.c says why.  No batch created from `m`
 *                            may be alive across the call.
 * Views stay valid until the next call that adds an assignment or destroys the object. */
typedef struct sa_hmm sa_hmm_t;
#define SA_HMM_GAUSSIAN 0   /* ContinuousPairHmm, StateMachineType threeState    */
#define SA_HMM_HDP 1        /* HdpHmm, threeStateHdp                             */
typedef struct sa_hmm_view {
    int type, n_states, n_alpha, k;
    char alphabet[64];
    int64_t n_kmers;
    double *transitions;            /* 9: from * 3 + to (match, gapX, gapY), linear space        */
    double *likelihood;
    double *event_model;            /* 5 per k-mer                                               */
    double *event_expectations;     /* SA_HMM_GAUSSIAN: 2 per k-mer                              */
    double *posteriors;             /* SA_HMM_GAUSSIAN: per k-mer                                */
    uint8_t *observed;              /* SA_HMM_GAUSSIAN: per k-mer                                */
    double threshold;               /* SA_HMM_HDP                                                */
    int64_t n_assignments;          /* SA_HMM_HDP                                                */
    const double *assignment_events;
    const char *assignment_kmers;   /* k characters per assignment, not terminated               */
    int has_model;
} sa_hmm_view_t;
int sa_hmm_create(sa_hmm_t **out, const sa_model_t *m, int type, double threshold, double transitions_pseudocount,
                  double emissions_pseudocount);
void sa_hmm_destroy(sa_hmm_t *h);
int sa_hmm_view(sa_hmm_t *h, sa_hmm_view_t *v);
int sa_hmm_set_event_model(sa_hmm_t *h, const double *table5);
int sa_hmm_add_expectations(sa_hmm_t *h, const double *trans9, double likelihood);
int sa_hmm_add_emission_expectation(sa_hmm_t *h, int64_t kmer_index, double mean, double p);
int sa_hmm_add_assignment(sa_hmm_t *h, const char *kmer, double event_mean);
int sa_hmm_write(const sa_hmm_t *h, const char *path);
int sa_hmm_load(sa_hmm_t **out, const char *path, int type, double transitions_pseudocount, double emissions_pseudocount);
int sa_hmm_add_expectations_file(sa_hmm_t *h, const char *path);
int sa_hmm_normalize(sa_hmm_t *h);
int sa_hmm_load_into_model(sa_model_t *m, const sa_hmm_t *h);
/* the model's transitions as the ten tokens of a .model file's second line (linear space; tokens 5, 7 and 9 are 0) */
int sa_model_transitions10(const sa_model_t *m, double *out10);

/* ---- HDP rebuild: model construction, data, the Gibbs sweeps ----------------------------------------
 * What buildHdpUtil (impl/buildHdpUtil.c) and updateHdpFromAssignments (impl/signalMachine.c:384-397) do around the pieces above.
 * The sweep itself is sequential, random-number-driven host code (signalalign_amd/csrc/sa_hdpgibbs.c); every kept sample's
 * take_distr_sample and the finalisation run on the GPU.  PARITY UNPINNED by construction: the reference draws from rand() / ranlib
 * and walks pointer-hashed sets; here one seeded generator and insertion-ordered lists -- its tests of this code are properties
 * (tests/hdpTests.c:109-233, tests/nanoporeHdpTests.c:272-480), checked in tests/test_gpu_hdp_rebuild.py.
 *   sa_hdp_state_new       a NanoporeHDP without data: new_hier_dir_proc / new_hier_dir_proc_2 (impl/hdp.c:879-995) + one of the tree
 *                          layouts of impl/nanopore_hdp.c:489-1060 + finalize_hdp_structure.  `gamma` (depth values: 2 for the flat
 *                          layout, 3 otherwise) fixes the concentration parameters; gamma == NULL takes a Gamma prior on them
 *                          (gamma_alpha, gamma_beta per depth) and the sweeps sample them.  `groups`: per letter of `alphabet` (as
 *                          given, unsorted) -- SA_HDP_LAYOUT_COMPOSITION: non-zero = purine; SA_HDP_LAYOUT_GROUP_MULTISET: group number
 *   sa_hdp_nig_params_from_table  normal_inverse_gamma_params_from_minION (impl/nanopore_hdp.c:122-176): mu, nu, alpha, beta by maximum
 *                          likelihood from a lookup table's level means and level sds
 *   sa_hdp_state_pass_data / _pass_assignments / _pass_assignment_file   reset_hdp_data + pass_data_to_hdp (impl/hdp.c:1549-1660:
 *                          observed DPs marked, one chain of factors per observed DP under one base factor); from (k-mer, event)
 *                          assignments as hdpHmm_loadFromFile hands them on (impl/continuousHmm.c:722-780: sa_hmm_view's arrays);
 *                          from an assignments / alignment table (update_nhdp_from_alignment_with_filter, impl/nanopore_hdp.c:206-297;
 *                          strand_filter NULL: every row).  A k-mer outside the alphabet: SA_EALPHABET (the reference exits)
 *   sa_hdp_state_gibbs     execute_gibbs_sampling (impl/hdp.c:2486-2549): num_samples distribution samples, one per `thinning`
 *                          iterations behind `burn_in`; the collectors are ADDED to the state's (several calls continue one run)
 *   sa_hdp_state_finalize  finalize_distributions (:2551-2584); afterwards sa_hdp_state_write gives a .nhdp the aligner loads
 * SA_ESTATE: sampling without data or after finalisation, finalising without samples. */
#define SA_HDP_LAYOUT_FLAT 0            /* flat_hdp_model: every k-mer under the base DP (singleLevel*)        */
#define SA_HDP_LAYOUT_MULTISET 1        /* multiset_hdp_model: k-mers grouped by their multiset of letters     */
#define SA_HDP_LAYOUT_MIDDLE_NTS 2      /* middle_2_nts_hdp_model: by their two middle letters                 */
#define SA_HDP_LAYOUT_COMPOSITION 3     /* purine_composition_hdp_model: by their number of purines            */
#define SA_HDP_LAYOUT_GROUP_MULTISET 4  /* group_multiset_hdp_model: by the multiset of their letters' groups  */
int sa_hdp_state_new(sa_hdp_state_t **out, int layout, const char *alphabet, int64_t kmer_length, const int64_t *groups,
                     const double *gamma, const double *gamma_alpha, const double *gamma_beta, double grid_start, double grid_stop,
                     int64_t grid_length, double mu, double nu, double alpha, double beta);
/* the same over any tree of Dirichlet processes (set_dir_proc_parent for every DP, parents[base] = -1, every leaf at depth - 1): a
 * plain HierarchicalDirichletProcess, what the reference's own HDP tests build (tests/nanoporeHdpTests.c:272-345) */
int sa_hdp_state_new_tree(sa_hdp_state_t **out, int64_t num_dps, int64_t depth, const int64_t *parents, const double *gamma,
                          const double *gamma_alpha, const double *gamma_beta, double grid_start, double grid_stop, int64_t grid_length,
                          double mu, double nu, double alpha, double beta);
int sa_hdp_nig_params_from_table(const double *table5, int64_t n_kmers, double *mu_out, double *nu_out, double *alpha_out,
                                 double *beta_out);
int sa_hdp_state_pass_data(sa_hdp_state_t *s, const double *data, const int64_t *dp_ids, int64_t n);
int sa_hdp_state_pass_assignments(sa_hdp_state_t *s, const char *kmers /* k characters each */, const double *events, int64_t n);
int sa_hdp_state_pass_assignment_file(sa_hdp_state_t *s, const char *path, const char *strand_filter, int64_t *n_out);
int sa_hdp_state_kmer_dp(const sa_hdp_state_t *s, const char *kmer);   /* kmer_id (impl/nanopore_hdp.c:405-410), -1 outside the alphabet */
int sa_hdp_state_gibbs(sa_hdp_state_t *s, int64_t num_samples, int64_t burn_in, int64_t thinning, uint64_t seed, int device, int verbose);
int sa_hdp_state_finalize(sa_hdp_state_t *s, int device);
int64_t sa_hdp_state_samples_taken(const sa_hdp_state_t *s);
double sa_hdp_digamma(double x);    /* test hooks: the two special functions of the maximum-likelihood alpha (x > 0) */
double sa_hdp_trigamma(double x);

int sa_device_count(void);
/* HBM of `device`: bytes free (what the library's caching allocator holds counts as free) and in total; a caller that keeps
 * several batches in flight sizes its pipeline with this (sa_batch_stats_t.f_bytes is the bulk of a batch) */
int sa_device_memory(int device, int64_t *free_bytes, int64_t *total_bytes);
const char *sa_strerror(int code);
const char *sa_version(void);
void sa_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
