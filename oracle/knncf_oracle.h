/*
 * knncf_oracle.h — CPU restatement (plain C, fp64) of the reference's kNN
 * collaborative-filtering path, src/main/scala/shared/predictions.scala.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: the reference cannot be executed in this environment (no
 * JVM/Scala/Spark) and the MovieLens files its committed answer JSONs refer
 * to are absent, so those goldens are data-gated (tests/test_goldens.py).
 * Executable pins: Scala-2.11 HashSet iteration-order known answers
 * (Set(1..10), Set(1..20)), the hand-derived micro-fixture of SURVEY.md §4,
 * and structural invariants.  Against the real ml-100k/ml-25m goldens the
 * oracle is therefore "parity unpinned" until someone supplies the data.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/src/main/scala/).
 */
#ifndef KNNCF_ORACLE_H
#define KNNCF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (mirrors include/knncf.h) */
#define ORC_OK 0
#define ORC_E_INVALID (-1)
#define ORC_E_NONFINITE (-2) /* scale() == 0 -> non-finite deviation */
#define ORC_E_DUPLICATE (-3) /* duplicate (user,item) training rows */
#define ORC_E_NOMEM (-4)

/* similarity kinds */
#define ORC_SIM_COSINE 0  /* adjustedCosineSimilarityFunction shared/predictions.scala:407-433 */
#define ORC_SIM_ONE 1     /* similarityOne :400 */
#define ORC_SIM_JACCARD 2 /* jaccardCoefficient :440-464 */

typedef struct orc_model orc_model;
typedef struct orc_pipeline orc_pipeline;

/* ---- Scala 2.11 immutable collection iteration order -- */
/* HashSet/HashMap.improve (scala/collection/immutable/HashSet.scala) */
uint32_t orc_improve(uint32_t hcode);
/* key whose ascending unsigned order == HashTrieSet/HashTrieMap iteration
 * order of an element with *improved* hash h (5-bit digits, LSB digit first) */
uint32_t orc_trie_key(uint32_t improved);
/* Tuple2[Int,Int].hashCode = MurmurHash3.productHash(_, 0xcafebabe) */
uint32_t orc_tuple2_hash(int32_t a, int32_t b);
/* iteration order of an immutable.Set[Int] built by inserting `ids`
 * (distinct, insertion order) one by one: Set1..Set4 keep insertion order,
 * >=5 elements is a HashTrieSet.  out[] receives the ids in iteration order */
void orc_int_set_order(const int32_t* ids, int32_t n, int32_t* out);

/* scale :57-61 */
double orc_scale(double x, double y);

/* ---- model = everything the closures capture eagerly --------------------- */
/* load :35-49 has already happened: the arrays are the collected Array[Rating]
 * in file order. */
orc_model* orc_fit(const int32_t* users, const int32_t* items,
                   const double* ratings, int64_t n, int* status);
void orc_free(orc_model*);

int32_t orc_num_users(const orc_model*);
int32_t orc_num_items(const orc_model*);
/* N3: users in the iteration order of `ratings.map(_.user).toSet` :599 */
void orc_user_iteration_order(const orc_model*, int32_t* out_raw_ids);

double orc_average(const orc_model*);                                /* :94 */
int orc_users_avg(const orc_model*, int32_t user, double* out);      /* :113, 1 if present */
int orc_items_avg(const orc_model*, int32_t item, double* out);      /* :134 */
int orc_items_avg_dev(const orc_model*, int32_t item, double* out);  /* :176-186 (HashMap order, N4) */
int orc_items_avg_dev_spark(const orc_model*, int32_t item, double* out); /* :336-343 (file order) */
/* per training row (file order): computeNormalizeDeviation :155-169 and
 * preprocessedRating :470-481 */
const double* orc_normalized_deviations(const orc_model*);
const double* orc_preprocessed(const orc_model*);
int orc_user_weight(const orc_model*, int32_t user, double* out);    /* :474 */

/* the four non-personalised predictors */
double orc_predict_global(const orc_model*, int32_t u, int32_t i);       /* computeAvgRating :101 */
double orc_predict_user_avg(const orc_model*, int32_t u, int32_t i);     /* computeUserAvg :120 */
double orc_predict_item_avg(const orc_model*, int32_t u, int32_t i);     /* computeItemAvg :141 */
double orc_predict_item_avg_dev(const orc_model*, int32_t u, int32_t i); /* computeItemAvgDev :193 */
double orc_predict_baseline(const orc_model*, int32_t u, int32_t i);     /* computePrediction :205-237 */
double orc_predict_baseline_spark(const orc_model*, int32_t u, int32_t i); /* baselinePredictorSpark :362-391 */

/* kind: 0 global, 1 user, 2 item, 3 baseline, 4 baseline_spark.
 * MAE :69-73 / applyAndMean :80-86 (left fold, file order).  per_pred may be NULL. */
double orc_mae_simple(const orc_model*, int kind, const int32_t* users,
                      const int32_t* items, const double* ratings, int64_t n,
                      double* per_pred);

/* ---- personalised / kNN closures ----------------------------------------- */
/* predictor(train, weightedSumDeviation(train, F)) where
 *   F = sim                         if k < 0   (predict/Personalized.scala:61-72)
 *   F = getSimilarity(train,k,sim)  if k >= 0  (predict/kNN.scala:43-44)
 * The pipeline owns the closures' memo state (which only influences results
 * through summation order). */
orc_pipeline* orc_pipeline_create(const orc_model*, int sim_kind, int32_t k);
void orc_pipeline_free(orc_pipeline*);

/* the underlying similarity closure (u,v) (stateful memo :414-432) */
double orc_pipeline_raw_similarity(orc_pipeline*, int32_t u, int32_t v);
/* stateless evaluation of the similarity on a fresh closure: owner order = u */
double orc_fresh_similarity(const orc_model*, int sim_kind, int32_t u, int32_t v);
/* getSimilarity closure :634-648 (k >= 0 only) */
double orc_pipeline_knn_similarity(orc_pipeline*, int32_t u, int32_t v);
/* getNeighbors closure :603-616; returns count (<= cap written) */
int32_t orc_pipeline_neighbors(orc_pipeline*, int32_t u, int32_t cap,
                               int32_t* ids, double* sims);
double orc_pipeline_wsd(orc_pipeline*, int32_t u, int32_t i);     /* :504-548 */
double orc_pipeline_predict(orc_pipeline*, int32_t u, int32_t i); /* :568-585 */
double orc_pipeline_mae(orc_pipeline*, const int32_t* users, const int32_t* items,
                        const double* ratings, int64_t n, double* per_pred);

/* recommendations(ratings, predictor)(user, n) shared/predictions.scala:651-674:
 * every train item the user has not rated, predicted, ordered by (prediction
 * descending, item id ascending), first n.  `p` non-NULL: the kNN / personalised
 * predictor of the pipeline; p == NULL: the closed-form predictor `simple_kind`
 * (0 global, 1 user, 2 item, 3 baseline, 4 baseline_spark) of the model.  Returns the number of entries written (<= n). */
int32_t orc_recommend(const orc_model*, orc_pipeline* p, int simple_kind, int32_t user, int32_t n,
                      int32_t* out_items, double* out_preds);

/* ---- bulk evaluation on all cores (OpenMP) --------------------------------
 * The kNN closures (cosine similarity, k >= 0) evaluated for many users at once,
 * one user per thread; refused (ORC_E_INVALID) when some user has <= 4 ratings,
 * because then values depend on the memo history (use orc_pipeline).  Same
 * operands in the same order as the per-pair closures — see the comment in
 * knncf_oracle.c; tests/test_oracle_bulk.py pins the two forms to each other. */
typedef struct orc_knn_table orc_knn_table;
/* users_raw == NULL: every user, row r = r-th smallest raw id.  threads <= 0: all cores */
orc_knn_table* orc_knn_table_build(const orc_model*, int32_t k, const int32_t* users_raw, int32_t n_users,
                                   int threads, int* status);
void orc_knn_table_free(orc_knn_table*);
int32_t orc_knn_table_width(const orc_knn_table*); /* min(k, U-1) */
int32_t orc_knn_table_rows(const orc_knn_table*);
const double* orc_knn_table_sims(const orc_knn_table*); /* [rows * width], reference order */
void orc_knn_table_export_ids(const orc_knn_table*, int32_t* out_ids /*[rows*width] raw*/, int32_t* out_row_user /*[rows] raw*/);
/* predictions of predictor(train, weightedSumDeviation(train, getSimilarity(train,k,cos))) for every row; returns
 * the MAE (left fold, file order) when ratings != NULL */
double orc_knn_table_predict(const orc_knn_table*, const int32_t* users, const int32_t* items, const double* ratings,
                             int64_t n, int threads, double* out_pred, int* status);
int orc_max_threads(void);
void orc_set_threads(int n); /* cap for orc_fit's parallel loops and the bulk form; <= 0: all cores */

#ifdef __cplusplus
}
#endif
#endif
