/*
 * knncf.h — C ABI of the MI355X-native kNN collaborative-filtering engine.
 *
 * The reference (EloDoyard/movie-recommender-system) has no FFI boundary: its
 * hot path is a set of Scala functions in
 *   src/main/scala/shared/predictions.scala
 * that take a Seq[Rating]/RDD[Rating] and return (Int,Int)=>Double closures,
 * called by predict.Baseline, predict.Personalized, predict.kNN and
 * distributed.DistributedBaseline.  This header is the boundary a JNI shim
 *  binds instead: one handle == one set of those closures
 * (including their memo state), plain pointers and sizes only.
 *
 * Conventions: every function returns a status (0 ok, negative error; text via
 * knncf_last_error).  Ids are the RAW user/item ids of the rating files, in
 * and out.  Host-pointer and device-pointer variants exist for the bulk calls;
 * "_device" pointers must live on the handle's HIP device, and their contents
 * must be COMPLETE when the call is made: the engine works on private
 * non-blocking HIP streams and cannot order itself after the caller's
 * producer stream (synchronise that stream, or wait on its event, first);
 * results written to caller-provided device buffers are complete on return.
 * A handle is not thread-safe; different handles may be used from different
 * threads and on different devices (per-device kernel state is keyed by device
 * ordinal); the calling thread's current device is restored on return.  The library owns all device memory it allocates.  There is no
 * CPU fallback: without a usable gfx950 device knncf_create fails.
 */
#ifndef KNNCF_H
#define KNNCF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KNNCF_OK 0
#define KNNCF_E_INVALID (-1)     /* bad argument */
#define KNNCF_E_NONFINITE (-2)   /* scale() == 0: non-finite deviation (reference would emit NaN) */
#define KNNCF_E_DUPLICATE (-3)   /* duplicate (user,item) training rows */
#define KNNCF_E_NOMEM (-4)
#define KNNCF_E_HIP (-5)         /* HIP runtime error */
#define KNNCF_E_STATE (-6)       /* call order, e.g. query before fit */
#define KNNCF_E_UNSUPPORTED (-7)
#define KNNCF_E_NODEVICE (-8)    /* no gfx950 device / HIP runtime unavailable */

/* similarity functions of shared/predictions.scala */
#define KNNCF_SIM_COSINE 0  /* adjustedCosineSimilarityFunction :407-433 */
#define KNNCF_SIM_ONE 1     /* similarityOne :400 */
#define KNNCF_SIM_JACCARD 2 /* jaccardCoefficient :440-464 */

/* predictors (the closures the entry points build) */
#define KNNCF_PRED_GLOBAL_AVG 0   /* computeAvgRating :101 */
#define KNNCF_PRED_USER_AVG 1     /* computeUserAvg :120 */
#define KNNCF_PRED_ITEM_AVG 2     /* computeItemAvg :141 */
#define KNNCF_PRED_BASELINE 3     /* computePrediction :205-237 */
#define KNNCF_PRED_BASELINE_RDD 4 /* baselinePredictorSpark :362-391 */
#define KNNCF_PRED_KNN 5          /* predictor(train, weightedSumDeviation(train, getSimilarity(train, k, sim))) predict/kNN.scala:43-44;
                                     sim = the handle's similarity: the adjusted cosine or the Jaccard coefficient (any number of users:
                                     both go through the MFMA GEMM + sparse tail + exact re-rank); similarityOne: KNNCF_E_UNSUPPORTED */
#define KNNCF_PRED_PERSONALIZED 6 /* predictor(train, weightedSumDeviation(train, sim)) predict/Personalized.scala:61-72, sim = the
                                     handle's similarity itself, no neighbourhood cut.  Cosine / Jaccard keep U x U values: U <= 2048;
                                     cosine additionally needs > 4 ratings per user, else KNNCF_E_UNSUPPORTED */

#define KNNCF_FLAG_VERIFY_BOUND 1u /* check |approx - exact| <= eps on every re-ranked pair (debug) */
/* The similarity GEMM is only a filter in front of the exact fp64 re-rank.  Default operand type is fp16
 * (11-bit significand, same MFMA rate as bf16): its rigorous error band is 8x narrower, so the shortlists
 * are ~k instead of ~3k and the re-rank is 3.6x cheaper (measured).  |pre| <= 1, so fp16's range is ample.
 * This flag selects bf16 operands (the north_star's literal wording); results are identical either way. */
#define KNNCF_FLAG_BF16_FILTER 4u
#define KNNCF_FLAG_F32_PANEL 8u /* keep the similarity panel in fp32 (default fp16: half the HBM traffic, band + 2^-11) */
#define KNNCF_FLAG_OVERLAP 2u      /* double-buffer the row blocks: GEMM/tail of block b+1 overlap select/re-rank of block b */

typedef struct knncf_handle knncf_handle;

typedef struct knncf_config {
    uint32_t struct_size;    /* = sizeof(knncf_config) */
    int32_t device;          /* HIP device ordinal */
    int32_t k;               /* neighbourhood size of KNNCF_PRED_KNN (predict/kNN.scala:44 uses 300) */
    int32_t similarity;      /* KNNCF_SIM_* */
    int32_t shard_rank;      /* users are block-partitioned over shard_count handles (one per GPU); */
    int32_t shard_count;     /* this handle owns block shard_rank.  1 = everything. */
    int64_t workspace_bytes; /* cap for the similarity panel + dense operand panels; 0 = auto */
    uint32_t flags;          /* KNNCF_FLAG_* */
    uint32_t head_items;     /* hybrid similarity: the head_items most-rated items go through the dense MFMA
                                GEMM, the sparse tail is accumulated per panel row in LDS (fixed point).  0 = cost model,
                                KNNCF_HEAD_ALL = every item dense */
} knncf_config;
#define KNNCF_HEAD_ALL 0xffffffffu

/* per-stage device timings of the last fit / neighbour build / predict, milliseconds */
typedef struct knncf_timings {
    double prep_ms;     /* K0-K3 (and K4 when a call needed it): id compaction, CSR/CSC, means, deviations, norms */
    double densify_ms;  /* CSR -> 16-bit operand panels */
    double gemm_ms;     /* K5 similarity GEMM (all launches) */
    double tail_ms;     /* K5 sparse tail when run as its own pass (0: fused into select_ms) */
    double select_ms;   /* K6 threshold + shortlist */
    double rerank_ms;   /* K6b exact fp64 re-rank + top-k sort */
    double predict_ms;  /* K7-K9 prediction + MAE */
    int64_t gemm_launches;
    double gemm_flops_executed;    /* 2*M*N*K summed over launches */
    double gemm_flops_algorithmic; /* 2 * pairs * I_c for the rows built */
    int64_t shortlist_total;       /* sum of shortlist sizes */
    int64_t fallback_rows;         /* rows re-done by the exact fallback */
    double max_bound_violation;    /* KNNCF_FLAG_VERIFY_BOUND: max(|approx-exact| - eps), <= 0 when the bound holds */
    int64_t head_items;            /* dense head width used by the last build */
    double tail_pair_updates;      /* sum over tail items of (raters in panel) x (raters) */
    double rerank_row_bytes;       /* K6b algorithmic traffic: 12 B x ratings of every re-ranked candidate */
    double select_row_bytes;       /* K6 algorithmic traffic: panel entry size x (rows x users) similarity panel entries read */
    int64_t select_launches;       /* row-block launches of select / re-rank (the symmetric GEMM is ONE launch for all of them) */
} knncf_timings;

const char* knncf_version(void);
const char* knncf_status_string(int status);

int knncf_create(const knncf_config* cfg, knncf_handle** out);
void knncf_destroy(knncf_handle* h);
const char* knncf_last_error(const knncf_handle* h);

/* ---- fit: everything the reference's kNN closures compute eagerly (K0-K3) -- */
/* Rows are the collected Array[Rating] in FILE ORDER (the reference's summation
 * order and fallbacks depend on it).  Arrays are copied.
 * K4 — the per-item maps of the baseline predictors (itemsAvg :134, itemsAvgDev
 * :176-186, getItemsAvgDev :336-343) — is not part of the kNN closures
 * (weightedSumDeviation :489-548, predictor :557-585 never evaluate them): the
 * reference builds those maps when computeItemAvg / computePrediction / the Spark
 * forms are constructed, and the handle builds them on the first call that reads
 * them (knncf_item_avg*, KNNCF_PRED_ITEM_AVG / BASELINE / BASELINE_RDD, and
 * PERSONALIZED with similarityOne), charged to prep_ms of that call. */
int knncf_fit(knncf_handle* h, const int32_t* users, const int32_t* items,
              const double* ratings, int64_t n);
int knncf_fit_device(knncf_handle* h, const int32_t* d_users, const int32_t* d_items,
                     const double* d_ratings, int64_t n);

int knncf_num_users(const knncf_handle* h, int32_t* out);
int knncf_num_items(const knncf_handle* h, int32_t* out);

/* ---- scalar queries mirroring the JSON answers ---------------------------- */
int knncf_global_avg(knncf_handle* h, double* out);                 /* average :94 */
int knncf_user_avg(knncf_handle* h, int32_t user, double* out);     /* computeUserAvg(train)(user, _) */
int knncf_item_avg(knncf_handle* h, int32_t item, double* out);     /* computeItemAvg(train)(_, item) */
int knncf_item_avg_dev(knncf_handle* h, int32_t item, double* out); /* computeItemAvgDev(train)(_, item) :193 */
int knncf_item_avg_dev_rdd(knncf_handle* h, int32_t item, double* out); /* itemsAvgDevSpark(train)(_, item) :350 */
/* the similarity function on a fresh closure: sim(train)(u, v) */
int knncf_similarity(knncf_handle* h, int32_t u, int32_t v, double* out);
/* getSimilarity(train, k, sim)(u, v): sim if v is one of u's k nearest, else 0 :634-648 */
int knncf_knn_similarity(knncf_handle* h, int32_t u, int32_t v, double* out);
/* getNeighbors(train, k, sim)(u): ids and similarities in reference order :603-616 */
int knncf_neighbors(knncf_handle* h, int32_t u, int32_t cap, int32_t* ids, double* sims,
                    int32_t* count);
/* getNeighbors for users[0..n) at once ("as if called in this order"): row j of ids / sims ([n * cap]) receives
 * min(counts[j], cap) entries.  Neighbourhoods that do not exist yet are built in ONE batch on the device — the bulk
 * door for exporting or verifying whole neighbour tables (knncf_neighbors costs a device round trip per user). */
int knncf_neighbors_batch(knncf_handle* h, const int32_t* users, int64_t n, int32_t cap, int32_t* ids,
                          double* sims, int32_t* counts);
int knncf_predict(knncf_handle* h, int predictor, int32_t user, int32_t item, double* out);

/* recommendations(train, predictor)(user, n) shared/predictions.scala:651-674 (called by
 * recommend/Recommender.scala:85-88 with n = 3): every train item `user` has not rated, predicted with `predictor`,
 * ordered by (prediction descending, raw item id ascending); the first min(n, #unrated) are written, *count of them.
 * An unknown user has rated nothing (every prediction is the global average: pure id order). */
int knncf_recommend(knncf_handle* h, int predictor, int32_t user, int32_t n, int32_t* items,
                    double* predictions, int32_t* count);

/* ---- batch ---------------------------------------------------------------- */
int knncf_predict_batch(knncf_handle* h, int predictor, const int32_t* users,
                        const int32_t* items, int64_t n, double* out);
int knncf_predict_batch_device(knncf_handle* h, int predictor, const int32_t* d_users,
                               const int32_t* d_items, int64_t n, double* d_out);
/* MAE :69-73 over (users, items, ratings) in file order */
int knncf_mae(knncf_handle* h, int predictor, const int32_t* users, const int32_t* items,
              const double* ratings, int64_t n, double* mae);
/* device variant; returns the partial sums of the rows this shard owns
 * (rows of users outside the shard are skipped): mae = sum_abs_err / count
 * after an all-reduce over the shards.  d_pred (optional, may be NULL) receives
 * the per-row predictions of the owned rows (others untouched). */
int knncf_mae_device(knncf_handle* h, int predictor, const int32_t* d_users,
                     const int32_t* d_items, const double* d_ratings, int64_t n,
                     double* sum_abs_err, int64_t* count, double* d_pred);

/* ---- multi-GPU exchange (one handle per GPU, collectives done by the host) - */
/* After knncf_fit* on every shard, each shard holds the per-user means / norms
 * (the order-sensitive fp64 folds of K2 / K3) of ITS users only.  The host
 * all-gathers d_user_avg[user_begin, user_end) and d_user_norm[...] (RCCL
 * all-gather over xGMI: 16 B per user) in place, then calls knncf_shard_commit,
 * which recomputes the other users' normalized deviations and preprocessed
 * ratings — elementwise functions of (rating, mean) and (deviation, norm) — bit
 * for bit, so nothing per RATING travels.  (d_dev / d_pre stay in the view: a host
 * that gathers them as well, as the round-1 protocol did, gets the same values
 * written twice.)  Users are block-partitioned in ascending dense order,
 * ceil(num_users / shard_count) per shard.  With shard_count == 1 these are no-ops.
 * One process driving all GPUs binds knncf_group_* below instead. */
typedef struct knncf_shard_view {
    int32_t user_begin, user_end; /* owned dense users [begin, end) */
    int64_t nnz_begin, nnz_end;   /* their entries in the user-major rating arrays */
    int32_t num_users;
    int64_t num_ratings;
    double* d_user_avg;  /* [num_users]   */
    double* d_user_norm; /* [num_users]   */
    double* d_dev;       /* [num_ratings] normalized deviations, user-major order */
    double* d_pre;       /* [num_ratings] preprocessed ratings, user-major order */
} knncf_shard_view;
int knncf_shard_view_get(knncf_handle* h, knncf_shard_view* out);
int knncf_shard_commit(knncf_handle* h);

/* ---- one process, several GPUs: the collectives inside the library ---------- */
/* A group = n shard handles (shard_rank i on devices[i]) + one RCCL communicator per device (ncclCommInitAll) + one
 * HIP stream per device for the collectives; every entry point drives the n GPUs from n host threads.  This is what a
 * JVM binds: ONE call per step instead of a re-implementation of the exchange —
 * distributed/DistributedBaseline.scala:41-47 hands one RDD to Spark the same way.
 *   knncf_group_fit  : knncf_fit on every shard (the host arrays are copied to every device) -> collective status ->
 *                      ncclAllGather of the padded {mean, norm} segments -> knncf_shard_commit on every shard.
 *   knncf_group_mae  : knncf_mae_device on every shard (each predicts the test rows of its own users) -> collective
 *                      status -> ncclAllReduce (sum) of (sum |r - p|, rows) -> mae; shared/predictions.scala:246-268's
 *                      `sum` / `count` actions.
 * cfg->device, shard_rank and shard_count are ignored (the group sets them).  RCCL is loaded at the first
 * knncf_group_create (dlopen of librccl.so.1: the library itself does not link it); KNNCF_E_UNSUPPORTED if it is absent,
 * KNNCF_E_RCCL for a failing RCCL call.  A group is not thread-safe. */
#define KNNCF_E_RCCL (-9)
typedef struct knncf_group knncf_group;
int knncf_group_create(const knncf_config* cfg, const int32_t* devices, int32_t n_devices, knncf_group** out);
void knncf_group_destroy(knncf_group* g);
const char* knncf_group_last_error(const knncf_group* g);
int knncf_group_size(const knncf_group* g, int32_t* n_devices);
/* the shard handle of rank i (owned by the group): scalar queries, knncf_neighbors of ITS users, timings */
int knncf_group_handle(knncf_group* g, int32_t rank, knncf_handle** out);
int knncf_group_fit(knncf_group* g, const int32_t* users, const int32_t* items, const double* ratings, int64_t n);
int knncf_group_mae(knncf_group* g, int predictor, const int32_t* users, const int32_t* items, const double* ratings,
                    int64_t n, double* mae);
/* every shard predicts the rows of its own users; out[0..n) receives all of them */
int knncf_group_predict_batch(knncf_group* g, int predictor, const int32_t* users, const int32_t* items, int64_t n, double* out);

/* ---- loader and on-disk cache --------------------------------- */
/* `load` shared/predictions.scala:35-49 as a multithreaded host parser: the line is split on `separator` (literal),
 * columns are trimmed, a line is kept iff column 0 parses as an Int (headers are dropped silently); columns 1 and 2
 * of a kept line must parse (the reference throws; here: KNNCF_E_INVALID with "<path>:<line>: ..." in err).  Rows
 * come back in FILE ORDER.  threads <= 0: one per hardware thread.  Release with knncf_free_ratings. */
typedef struct knncf_ratings {
    int64_t n;
    int32_t* users;
    int32_t* items;
    double* ratings;
} knncf_ratings;
int knncf_load_file(const char* path, const char* separator, int threads, knncf_ratings* out, char* err, int err_cap);
void knncf_free_ratings(knncf_ratings* r);
/* The same with a binary cache beside it (at ml-25m the text parse takes seconds, the fit 6 ms): the parsed
 * triples in FILE ORDER — the order is part of the semantics — stamped with the source file's size, modification time and
 * the separator, closed by a checksum.  A cache that matches `path` as it is now is read instead of parsing (*from_cache
 * = 1); a missing, stale, truncated or corrupt one is ignored and rewritten (tmp file + rename) after the parse.  The CSR /
 * CSC are not cached: K0 rebuilds them on the GPU faster than they could be read back.  cache_path == NULL: plain
 * knncf_load_file.  A cache that cannot be written is not an error. */
int knncf_load_file_cached(const char* path, const char* separator, int threads, const char* cache_path, knncf_ratings* out,
                           int* from_cache, char* err, int err_cap);

/* The Recommender's personal-ratings file, recommend/Recommender.scala:40-54 ("id,title,rating" CSV): every row's
 * (id, title) in file order — the header row as (0, "header") — and the rows with a non-zero rating as ratings of
 * `user` (the reference uses 944), ready to be appended to the training rows (`data.union(personal)` :68).  Rows with
 * an empty rating column are unrated; a non-numeric id / rating fails loudly (the reference throws).  Release with
 * knncf_free_personal. */
typedef struct knncf_personal {
    int64_t n_rows;
    int32_t* row_ids;      /* [n_rows] */
    char** row_names;      /* [n_rows] NUL-terminated, owned by the struct */
    char* name_storage;
    knncf_ratings ratings; /* the non-zero ratings, file order */
} knncf_personal;
int knncf_load_personal(const char* path, int32_t user, knncf_personal* out, char* err, int err_cap);
void knncf_free_personal(knncf_personal* p);

/* Checkpoint / resume of the expensive part of a fit: the U x k neighbour table (ids, fp64 similarities, build
 * sequence numbers).  save: every neighbourhood built so far.  load: the handle must be fitted on the same training
 * rows with the same k and similarity (checked with a fingerprint of the users, row extents and means:
 * KNNCF_E_STATE otherwise); afterwards getNeighbors / getSimilarity / predictions use the loaded lists and only
 * users that were not built at save time are built on demand. */
int knncf_neighbors_save(knncf_handle* h, const char* path);
int knncf_neighbors_load(knncf_handle* h, const char* path);

/* ---- introspection for bench / tests -------------------------------------- */
int knncf_get_timings(const knncf_handle* h, knncf_timings* out);
int knncf_reset_timings(knncf_handle* h);
/* drop the getNeighbors/getSimilarity memo (== constructing fresh closures) */
int knncf_reset_neighbors(knncf_handle* h);
/* change k (== getSimilarity(train, k, ...) with a new k); drops the memo */
int knncf_set_k(knncf_handle* h, int32_t k);

#ifdef __cplusplus
}
#endif
#endif
