/* rankfm_hip.h -- C ABI of the MI355X-native RankFM training engine (librankfm_hip.so).
 *
 * Drop-in boundary for the reference's private operator functions (all paths relative to the
 * etlundquist/rankfm tree):
 *
 *   rfm_fit_host / rfm_fit_device      replace  `_fit`        rankfm/_rankfm.pyx:122-342
 *                                      called from            rankfm/rankfm.py:304-324
 *   rfm_predict_host / _device         replace  `_predict`    rankfm/_rankfm.pyx:345-390
 *                                      called from            rankfm/rankfm.py:347-357
 *   rfm_recommend_host / _device       replace  `_recommend`  rankfm/_rankfm.pyx:393-460
 *                                      called from            rankfm/rankfm.py:381-394
 *
 * Conventions kept from the reference boundary: interactions int32 [N,2] C-contiguous; every
 * weight / feature / sample-weight array float32 C-contiguous; the six weight arrays
 * w_i[I], w_if[Q], v_u[U,F], v_i[I,F], v_uf[P,F], v_if[Q,F] are updated IN PLACE and nothing is
 * returned; absent features are the [U,1] / [I,1] all-zero placeholders with [1,F] / [1] zero
 * tables (rankfm/rankfm.py:199,211,224,236,244) and `has_*_features` carries the reference's
 * `x.any()` test (rankfm/_rankfm.pyx:193-194).
 *
 * Differences, all forced by the C boundary or by running on a GPU:
 *   - the per-user item dict (rankfm/rankfm.py:174) is passed as CSR: offsets int64 [U+1] and the
 *     users' sorted item indexes int32 [nnz] back to back;
 *   - exceptions become status codes (see rfm_status); the caller turns RFM_ERR_NONFINITE + k into
 *     the reference's AssertionError for array k (order of rankfm/_rankfm.pyx:98-103) and
 *     RFM_ERR_UNKNOWN_SCHEDULE into its ValueError (rankfm/_rankfm.pyx:225);
 *   - the epoch shuffle (np.random.shuffle, rankfm/_rankfm.pyx:227) is either passed in explicitly
 *     (`perms`, int32 [epochs,N]) or generated on the device by a keyed bijection (include/rfm_rng.h);
 *   - nothing is printed: per-epoch log-likelihood and L2 penalty come back in rfm_fit_report and
 *     the caller prints them when verbose (rankfm/_rankfm.pyx:332-336).
 *
 * No torch / numpy types appear here: plain pointers, sizes and a HIP stream handle as void*.
 */
#ifndef RANKFM_HIP_H
#define RANKFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFM_ABI_VERSION 6

typedef enum rfm_status {
    RFM_OK = 0,
    RFM_ERR_BAD_ARG = -1,           /* null pointer, negative size, max_samples < 1, ... */
    RFM_ERR_UNKNOWN_SCHEDULE = -2,  /* reference: ValueError('unknown [learning_schedule]') */
    RFM_ERR_NO_DEVICE = -3,         /* no HIP device / wrong architecture: the engine never falls back to a CPU */
    RFM_ERR_HIP = -4,               /* a HIP runtime call failed; rfm_last_error() has the text */
    RFM_ERR_UNSUPPORTED = -5,       /* shape outside the compiled kernel set (see rfm_fit_supported) */
    RFM_ERR_USER_SATURATED = -6,    /* a user has interacted with every item: rejection sampling cannot end */
    RFM_ERR_WORKSPACE = -7,         /* workspace missing or smaller than rfm_fit_workspace_bytes() */
    RFM_ERR_NONFINITE = 100         /* + k, k = 0..5 -> w_i, w_if, v_u, v_i, v_uf, v_if (assert_finite order) */
} rfm_status;

/* learning_schedule (rankfm/_rankfm.pyx:220-225) */
#define RFM_SCHEDULE_CONSTANT 0
#define RFM_SCHEDULE_INVSCALING 1

/* execution mode */
#define RFM_MODE_HOGWILD 0  /* production: thousands of wavefronts, fp32 atomic updates, counter RNG only */
#define RFM_MODE_SERIAL 1   /* one wavefront walks the shuffled rows in order with plain read-modify-write:
                               the reference's sequential semantics (parity / debugging mode) */

/* negative-draw stream.  The SAMPLER is always the reference's: every draw uniform over the whole catalogue, the user's own items
 * drawn again (rankfm/_rankfm.pyx:250-253). */
#define RFM_RNG_MT19937 0   /* reference stream: MT19937, `% I` (serial mode only; needs explicit perms) */
#define RFM_RNG_COUNTER 1   /* counter-based draws keyed by (seed, epoch, row, attempt): include/rfm_rng.h */

/* Experiments and tests steer the engine through THIS struct and nothing else (the library reads no environment variable); a
 * binding that replaces the reference's `_fit` passes rfm_fit_config.tuning = NULL and never needs it.  0 = automatic everywhere. */
typedef struct rfm_fit_tuning {
    int32_t n_workgroups;          /* Hogwild grid size (lifts the concurrency caps); ignored in serial mode */
    int32_t rows_per_launch;       /* > 0 splits an epoch into several launches */
    int32_t debug_shape;           /* 1-based index into the row-group shape table */
    int32_t debug_flags;           /* bit 0: run the Hogwild kernel on ONE row group (sequential; parity tests),
                                      bit 1: factor-row loads bypass the per-CU L1,
                                      bit 2: no LDS accumulation of hot item rows,
                                      bit 5: models with features: the dense feature tables are NOT trained (no table trainer); with
                                             bit 0 the row loop of the features kernel runs on one row group -- parity tests,
                                      bit 6: models with features: no table-friendly opening launch in the fit's first epoch,
                                      bit 7: row groups stride the epoch's segment order statically instead of taking tickets,
                                      bit 8: the item damping scales an item's step only when it is the POSITIVE item (the round-3 rule),
                                      bit 9: chip-filling BPR launches keep the item factor rows row-major (no segment-major working copy) */
    int32_t segment_rows;          /* longest user segment, 1..32 (auto: 32) */
    int32_t hot_publications;      /* publications of a hot row per epoch and workgroup (auto: 24 for BPR, 48 for WARP) */
    int32_t feature_waves;         /* wavefronts per workgroup of the features kernels, 2..16 (auto: 16; the pipelined row loop 12) */
    int32_t table_producers;       /* features: step-producer workgroups feeding the table trainer, 1..16 (auto: 3 on a full chip, 2 / 1 on
                                      small launches) */
    int32_t table_every;           /* features: the table trainer applies rows-of-the-launch / this many staged steps per launch (auto: 2.4 x
                                      the launch's row groups / 64 -- every 446th row on a full chip -- on launches of at least 4096 row
                                      groups, 1.8 x on smaller ones and in the opening launch: DESIGN.md 3.3) */
    int32_t table_step_pct;        /* features: the table trainer's step length in percent of the epoch's learning rate (auto: 100) */
    int32_t table_batch;           /* features: staged steps per batch of a producer, a multiple of 4 up to the row groups of the tables
                                      kernel's workgroup (auto: all of them, 64 with 16-lane row groups) -- a batch is scored on ONE state of
                                      the tables, DESIGN.md 3.3 */
    int32_t hot_sweep_every;       /* BPR segments kernel: a workgroup's hot-row bin lines are swept every this many row steps (auto: see
                                      rfm_api.hip kHotSweepEvery) */
    int32_t hot_slots;             /* most hot item rows a workgroup accumulates in LDS, up to 128 (auto: 64) */
    int32_t table_pace_pct;        /* features: the trainer's quota is spread over this percentage of a launch's segments (auto: 65 -- where the default
                                      quota ends by itself; -1: unpaced, as fast as the producers stage it: rounds 3 - 5) */
} rfm_fit_tuning;

/* What a binding of the reference's `_fit` fills in: its 19 arguments' scalars (rankfm/_rankfm.pyx:122-142), the engine's mode and
 * seed, and the two tokens a resident caller hands back. */
typedef struct rfm_fit_config {
    int64_t n_interactions;        /* N */
    int32_t n_users;               /* U */
    int32_t n_items;               /* I */
    int32_t n_user_features;       /* P  (1 when absent) */
    int32_t n_item_features;       /* Q  (1 when absent) */
    int32_t n_factors;             /* F */
    int32_t has_user_features;     /* reference: int(x_uf.any()) */
    int32_t has_item_features;     /* reference: int(x_if.any()) */
    float alpha;                   /* L2 on w_i, v_u, v_i */
    float beta;                    /* L2 on w_if, v_uf, v_if */
    float learning_rate;
    int32_t learning_schedule;     /* RFM_SCHEDULE_* */
    float learning_exponent;
    int32_t max_samples;           /* 1 = BPR (rankfm/rankfm.py:294-295), >1 = WARP */
    int32_t epochs;                /* epochs to run in this call */
    int32_t epoch_begin;           /* index of the first one for the learning-rate schedule (the reference restarts its
                                      schedule at 0 on every call, rankfm/_rankfm.pyx:218-223) */
    int32_t rng_epoch_offset;      /* added to the epoch index that keys the counter RNG and the device-generated order:
                                      a resumed fit (fit_partial) passes the epochs already trained so that it does not
                                      replay the first call's order and draws while its schedule restarts at 0 */
    int32_t mode;                  /* RFM_MODE_* */
    int32_t rng;                   /* RFM_RNG_* */
    uint32_t seed;                 /* MT seed (reference: 1492) or counter seed */
    int32_t check_finite;          /* 1: epoch-end finiteness check (reference behaviour), 0: skip */
    int32_t want_penalty;          /* 1: also return the L2 penalty per epoch (verbose printing) */
    float hogwild_damping;         /* M: a row touched by n in-flight updates at once is stepped with min(1, M/n) of the
                                      learning rate (n = in-flight rows x the row's share of the data).  0 = default (128),
                                      < 0 = off.  Ignored in serial mode. */
    int32_t epoch_part_index;      /* with epoch_parts > 1: run only part k (0-based) of each epoch's visiting order -- lets a */
    int32_t epoch_parts;           /* multi-GPU caller exchange item deltas several times per epoch; 0 or 1 = whole epochs    */
    int32_t keep_layout;           /* rfm_fit_device only.  1: the call may leave the item-side weights (v_i, w_i) in the ENGINE'S working
                                      layout inside the workspace when it returns -- segment-major factor rows, one 64-byte line per
                                      bias, pending hot-row sums -- instead of converting them back into the caller's arrays; the
                                      caller's v_i / w_i are then STALE until rfm_fit_export_weights() or a later call with
                                      keep_layout = 0 on the same workspace.  rfm_fit_report.layout_token says whether it did.  A
                                      resident training loop (one call per epoch or per exchange window) saves two passes over the item
                                      tables per call. */
    int64_t plan_token;            /* 0: build the Hogwild plan (user segments, CSR-ordered sample weights, per-item step
                                      scales) into the head of `workspace`; > 0: the value rfm_fit_report.plan_token returned
                                      by an earlier call on the SAME workspace, interactions, geometry and damping -- the
                                      plan is reused and the planning pass is skipped */
    int64_t layout_token;          /* 0: the caller's v_i / w_i arrays are current (always, unless the previous call on this workspace
                                      returned a non-zero rfm_fit_report.layout_token: then pass that value, together with its
                                      plan_token -- the workspace holds the current item-side weights) */
    const struct rfm_fit_tuning *tuning;   /* NULL = production */
} rfm_fit_config;

/* All pointers of one struct live in the same memory space: device memory for the *_device entry
 * points, host memory for the *_host ones. */
typedef struct rfm_fit_buffers {
    const int32_t *interactions;   /* [N,2] (user index, item index) */
    const float *sample_weight;    /* [N] */
    const int64_t *csr_offsets;    /* [U+1] */
    const int32_t *csr_items;      /* [csr_offsets[U]] sorted within each user */
    const float *x_uf;             /* [U,P] */
    const float *x_if;             /* [I,Q] */
    float *w_i;                    /* [I]    in/out */
    float *w_if;                   /* [Q]    in/out */
    float *v_u;                    /* [U,F]  in/out */
    float *v_i;                    /* [I,F]  in/out */
    float *v_uf;                   /* [P,F]  in/out */
    float *v_if;                   /* [Q,F]  in/out */
    const int32_t *perms;          /* [epochs,N] visiting order per epoch, or NULL = device-generated */
    void *workspace;               /* device scratch of >= rfm_fit_workspace_bytes() (device entry only) */
    size_t workspace_bytes;
} rfm_fit_buffers;

/* Host-side results; every pointer may be NULL. */
typedef struct rfm_fit_report {
    double *log_likelihood;        /* [epochs] sum over updates of log(sigmoid(pairwise utility)) */
    double *reg_penalty;           /* [epochs] alpha*(|w_i|^2+|v_u|^2+|v_i|^2) + beta*(|w_if|^2+|v_uf|^2+|v_if|^2) */
    float *sgd_kernel_ms;          /* [epochs] HIP-event time of the epoch's SGD launch(es) */
    int64_t *n_draws;              /* [epochs] accepted negative draws (== N for BPR) */
    int32_t epochs_done;           /* epochs completed before an error, or `epochs` */
    int32_t nonfinite_array;       /* -1, or 0..5 when the status is RFM_ERR_NONFINITE + k */
    int32_t launches_per_epoch;
    int32_t waves_per_launch;
    int64_t plan_token;            /* pass back as rfm_fit_config.plan_token to reuse the plan held in `workspace` */
    /* launch geometry of the Hogwild kernel: what a host program needs to replay the engine's draws (rankfm_amd/order.py) */
    int32_t workgroups;            /* grid size */
    int32_t groups_per_workgroup;  /* row groups (one interaction each) per workgroup */
    int64_t working_groups;        /* row groups that work (the concurrency cap can be below the grid's capacity) */
    int64_t units_per_launch;      /* user segments (or rows) per launch */
    int64_t n_units;               /* user segments (or rows) per epoch */
    int32_t segment_rows;          /* longest user segment of the plan (32), 0 = rows kernel */
    int32_t table_producers;       /* features kernel: step-producer workgroups beside the table trainer, 0 = none */
    int64_t layout_token;          /* non-zero: the item-side weights were LEFT in the workspace in the engine's layout (keep_layout): pass it
                                      back as rfm_fit_config.layout_token, or call rfm_fit_export_weights() before reading v_i / w_i */
    int64_t table_steps;           /* features kernel: staged steps the table trainer applied over the call (it sees every
                                      (epochs x N / table_steps)-th row of the stream) */
    int64_t feat_diag[8];          /* features kernel, microseconds over the call: the trainer waited for a batch | ran in all |
                                      the producers waited for a free slot (summed) | ran in all (summed) | the trainer's apply |
                                      publication | batch into LDS | slot release */
    int64_t table_overlap_us;      /* features kernel, last launch of the call: microseconds during which the table trainer's kernel and the
                                      row-loop kernel (two streams) were BOTH running; -1 = no trainer.  Near 0 = the two did not overlap
                                      (a profiler that serialises kernels, a device without room): the tables then were trained before the rows */
    int64_t table_span_us[2];      /* ... and how long each ran: tables kernel | row-loop kernel */
    float shader_mhz;              /* shader clock the call's last SGD launch ran at (its workgroup 0's cycle counter against the 100 MHz
                                      wall clock); 0 = not measured (serial / rows kernels).  The same binary differs by 10 - 25 % between
                                      boxes of one pool: quote timings with this next to them */
    int32_t reserved_report;       /* 0 */
} rfm_fit_report;

int rfm_abi_version(void);
const char *rfm_status_string(int status);
const char *rfm_last_error(void);            /* text of the last RFM_ERR_HIP on this thread */
int rfm_device_count(void);                  /* number of gfx950 devices visible, 0 if none */
int rfm_fit_supported(const rfm_fit_config *cfg);          /* RFM_OK or the error rfm_fit_* would return */
size_t rfm_fit_workspace_bytes(const rfm_fit_config *cfg); /* 0 on a bad config */

/* Multi-GPU exchange step (rankfm_amd/distributed.py; no reference counterpart -- the reference is single-process): the item-side
 * tables of a rank live in ONE flat fp32 bucket, and the per-epoch exchange is  delta = flat - start;  all-reduce(delta);
 * flat = start + scale .* delta.  These two entry points are the single pass over the bucket on each side of the collective
 * (the all-reduce itself is RCCL's, driven by the caller).  dev_scale: per-element damping of the summed deltas, or NULL for the
 * uniform factor `uniform_scale` (1 = plain sum, 1 / ranks = average). */
int rfm_delta_begin(float *dev_flat, const float *dev_start, size_t n, void *hip_stream);
int rfm_delta_finish(float *dev_flat, const float *dev_start, const float *dev_scale, float uniform_scale, size_t n, void *hip_stream);

/* Measurement aid (bench.py `roofline.peak_measured`): the rate at which this box's HBM serves a plain streaming kernel over
 * `bytes` of device memory -- a read-only pass and a copy (read + write), best of `iters` launches each, in GB/s.  The SGD
 * path's roofline is quoted against the 8 TB/s data-sheet peak AND against this achievable figure. */
int rfm_hbm_probe(size_t bytes, int iters, double *read_gbps, double *copy_gbps);

/* Converts item-side weights that an rfm_fit_device call with keep_layout = 1 left in `dev->workspace` back into the caller's dev->v_i /
 * dev->w_i (the reference's row-major layout), draining the pending hot-row sums first.  `cfg`: the configuration of that call with
 * plan_token / layout_token as its report returned them.  Enqueued on `hip_stream`, no synchronisation; idempotent; the workspace stays
 * current (a later call may still pass the tokens).  A layout_token of 0 is a no-op. */
int rfm_fit_export_weights(const rfm_fit_config *cfg, const rfm_fit_buffers *dev, void *hip_stream);

/* `_fit` on buffers already resident in HBM.  Work is enqueued on `hip_stream` (a hipStream_t; NULL =
 * the default stream); the call returns after one stream synchronisation at the end, when the report
 * is filled.  Weights are updated in place in device memory. */
int rfm_fit_device(const rfm_fit_config *cfg, const rfm_fit_buffers *dev, void *hip_stream, rfm_fit_report *report);

/* `_fit` on host (numpy) buffers: uploads to `device`, runs rfm_fit_device, downloads the six weight
 * arrays back into the caller's memory.  This is the 1:1 replacement of the reference call site.
 * Threading: calls on ONE device are serialised inside the library for their whole duration (the reference's `_fit` holds the GIL and
 * is not re-entrant either, rankfm/_rankfm.pyx:122 + mt19937ar.c:56-57): the staging allocation is one per device.  rfm_fit_device on
 * caller-owned buffers and streams is re-entrant except for models with features, which share one side stream per device and are
 * serialised the same way. */
int rfm_fit_host(const rfm_fit_config *cfg, const rfm_fit_buffers *host, int device, rfm_fit_report *report);

/* rfm_fit_host keeps its device staging allocation between calls (one per device, grown on demand; an allocation above 1 GiB is
 * freed when its call returns), and so do rfm_predict_host / rfm_recommend_host (one serving allocation per device for the model,
 * the inputs and the workspace; calls on one device are serialised); this releases them all.  The Python binding calls it at
 * interpreter exit. */
void rfm_release_cache(void);

/* ---- `_predict` (rankfm/_rankfm.pyx:345-390): pairs are float32 [n,2] indexes, NaN = unknown id ---- */
typedef struct rfm_model_view {
    int32_t n_users, n_items, n_user_features, n_item_features, n_factors;
    int32_t has_user_features, has_item_features;
    const float *x_uf, *x_if, *w_i, *w_if, *v_u, *v_i, *v_uf, *v_if;
} rfm_model_view;

int rfm_predict_device(const rfm_model_view *dev_model, int64_t n_pairs, const float *dev_pairs, float *dev_scores,
                       void *hip_stream);
int rfm_predict_host(const rfm_model_view *host_model, int64_t n_pairs, const float *pairs, float *scores, int device);

/* ---- `_recommend` (rankfm/_rankfm.pyx:393-460): users float32 [n] indexes (NaN = unknown), output float32
 * [n, n_rec] item indexes ranked by descending utility, optionally skipping the user's CSR items ---- */
int rfm_recommend_device(const rfm_model_view *dev_model, int64_t n_rec_users, const float *dev_users,
                         const int64_t *dev_csr_offsets, const int32_t *dev_csr_items, int32_t n_rec,
                         int32_t filter_previous, float *dev_rec_items, void *workspace, size_t workspace_bytes,
                         void *hip_stream);
size_t rfm_recommend_workspace_bytes(const rfm_model_view *model, int64_t n_rec_users, int32_t n_rec);
int rfm_recommend_host(const rfm_model_view *host_model, int64_t n_rec_users, const float *users,
                       const int64_t *csr_offsets, const int32_t *csr_items, int32_t n_rec, int32_t filter_previous,
                       float *rec_items, int device);

/* ---- `similar_items` / `similar_users` (rankfm/rankfm.py:405-428 / 431-454): the n rows whose latent representation
 * v[r] + x[r] . v_f has the largest dot product with that of row `index`, the row itself excluded; out: float32 [n] row
 * indexes by descending similarity.  kind: RFM_SIMILAR_ITEMS (v_i, x_if, v_if) or RFM_SIMILAR_USERS (v_u, x_uf, v_uf). ---- */
#define RFM_SIMILAR_ITEMS 0
#define RFM_SIMILAR_USERS 1
int rfm_similar_host(const rfm_model_view *host_model, int32_t kind, int32_t index, int32_t n, float *out, int device);

#ifdef __cplusplus
}
#endif
#endif /* RANKFM_HIP_H */
