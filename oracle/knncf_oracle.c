/*
 * knncf_oracle.c — CPU restatement (plain C, fp64, reference summation order)
 * of /root/reference/src/main/scala/shared/predictions.scala.
 *
 * TEST INFRASTRUCTURE ONLY (see knncf_oracle.h).  PARITY STATUS: pinned by the
 * Scala HashSet iteration-order known answers, the hand-derived micro-fixture
 * and structural invariants; "parity unpinned" against the committed answer
 * JSONs because the MovieLens inputs are absent from this environment.
 *
 * Data structures are sane (CSR, no per-pair maps) but every floating-point
 * operation happens with the same operands in the same order as the Scala
 * code would perform it; the order comes from Scala 2.11.12 immutable
 * HashSet/HashMap trie iteration and from the closures'
 * memo history (N6), both modelled explicitly below.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction, no fast-math).
 */
#define _GNU_SOURCE
#include "knncf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* Scala 2.11 collection order policies                                      */
/* ------------------------------------------------------------------------ */

/* scala.collection.immutable.HashSet.improve / HashMap.improve (2.11.12):
 *   var h = hcode + ~(hcode << 9); h ^= h >>> 14; h += h << 4; h ^ (h >>> 10) */
uint32_t orc_improve(uint32_t h) {
    h = h + ~(h << 9);
    h ^= h >> 14;
    h += h << 4;
    h ^= h >> 10;
    return h;
}

/* HashTrieSet/HashTrieMap iterate children by bitmap position = 5-bit digit of
 * the improved hash, least-significant digit at the root.  Concatenating the
 * digits root-first gives a key whose unsigned order is the iteration order. */
uint32_t orc_trie_key(uint32_t h) {
    return ((h & 31u) << 27) | (((h >> 5) & 31u) << 22) | (((h >> 10) & 31u) << 17) |
           (((h >> 15) & 31u) << 12) | (((h >> 20) & 31u) << 7) | (((h >> 25) & 31u) << 2) |
           ((h >> 30) & 3u);
}

static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* scala.util.hashing.MurmurHash3: mix / mixLast / finalizeHash / productHash */
static uint32_t mm3_mix_last(uint32_t hash, uint32_t data) {
    uint32_t k = data;
    k *= 0xcc9e2d51u;
    k = rotl32(k, 15);
    k *= 0x1b873593u;
    return hash ^ k;
}
static uint32_t mm3_mix(uint32_t hash, uint32_t data) {
    uint32_t h = mm3_mix_last(hash, data);
    h = rotl32(h, 13);
    return h * 5u + 0xe6546b64u;
}
static uint32_t mm3_finalize(uint32_t hash, uint32_t length) {
    uint32_t h = hash ^ length;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
uint32_t orc_tuple2_hash(int32_t a, int32_t b) {
    uint32_t h = 0xcafebabeu; /* MurmurHash3.productSeed */
    h = mm3_mix(h, (uint32_t)a);
    h = mm3_mix(h, (uint32_t)b);
    return mm3_finalize(h, 2u);
}

/* iteration key of a Map[(Int,Int),_] entry (N4) */
static uint32_t tuple_trie_key(int32_t a, int32_t b) {
    return orc_trie_key(orc_improve(orc_tuple2_hash(a, b)));
}
static uint32_t int_trie_key(int32_t a) { return orc_trie_key(orc_improve((uint32_t)a)); }

typedef struct {
    uint32_t key;
    int64_t idx;
} keyed;

static int cmp_keyed(const void* pa, const void* pb) {
    const keyed* a = (const keyed*)pa;
    const keyed* b = (const keyed*)pb;
    if (a->key != b->key) return a->key < b->key ? -1 : 1;
    if (a->idx != b->idx) return a->idx < b->idx ? -1 : 1;
    return 0;
}

void orc_int_set_order(const int32_t* ids, int32_t n, int32_t* out) {
    if (n <= 4) { /* Set1..Set4: insertion order */
        for (int32_t i = 0; i < n; ++i) out[i] = ids[i];
        return;
    }
    keyed* ks = (keyed*)malloc(sizeof(keyed) * (size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        ks[i].key = int_trie_key(ids[i]);
        ks[i].idx = i;
    }
    qsort(ks, (size_t)n, sizeof(keyed), cmp_keyed);
    for (int32_t i = 0; i < n; ++i) out[i] = ids[ks[i].idx];
    free(ks);
}

/* scale shared/predictions.scala:57-61 */
double orc_scale(double x, double y) {
    if (x > y) return 5 - y;
    else if (x < y) return y - 1;
    else return 1;
}

/* ------------------------------------------------------------------------ */
/* model                                                                     */
/* ------------------------------------------------------------------------ */

struct orc_model {
    int64_t n;
    int32_t* user;
    int32_t* item;
    double* rating;
    int32_t U, I;
    int32_t* uid; /* distinct raw user ids ascending; dense index = position */
    int32_t* iid;
    int32_t* du; /* per row dense user / item */
    int32_t* di;
    int64_t* u_ptr; /* groupBy(_.user): rows per user, file order */
    int64_t* u_rows;
    int64_t* i_ptr; /* groupBy(_.item): rows per item, file order */
    int64_t* i_rows;
    uint32_t* item_key; /* per dense item: trie key of the raw id (N2) */
    int64_t* u_sorted;  /* rows per user sorted by item_key (set intersection) */
    int32_t* user_pos;  /* N3: position of the user in allUsers' iteration */
    int32_t* user_by_pos;
    double global_avg;
    double* user_avg;
    double* item_avg;
    double* dev;    /* computeNormalizeDeviation, per row */
    double* weight; /* usersWeights, per user */
    double* pre;    /* preprocessedRating, per row */
    double* item_avg_dev;
    double* item_avg_dev_spark;
};

static int cmp_i32(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static int32_t lookup(const int32_t* ids, int32_t n, int32_t raw) {
    int32_t lo = 0, hi = n;
    while (lo < hi) {
        int32_t mid = lo + (hi - lo) / 2;
        if (ids[mid] < raw) lo = mid + 1;
        else hi = mid;
    }
    return (lo < n && ids[lo] == raw) ? lo : -1;
}

/* distinct values ascending.  Non-negative ids (every MovieLens id) go through a presence bitmap, O(n + max/64);
 * anything else through qsort.  Plumbing only: no arithmetic of the reference happens here. */
static int32_t* distinct_sorted(const int32_t* v, int64_t n, int32_t* count) {
    int32_t lo = 0, hi = -1;
    for (int64_t i = 0; i < n; ++i) {
        if (i == 0 || v[i] < lo) lo = v[i];
        if (i == 0 || v[i] > hi) hi = v[i];
    }
    if (n > 0 && lo >= 0) {
        const size_t words = ((size_t)hi >> 6) + 1;
        uint64_t* bits = (uint64_t*)calloc(words, sizeof(uint64_t));
        for (int64_t i = 0; i < n; ++i) bits[(uint32_t)v[i] >> 6] |= 1ull << ((uint32_t)v[i] & 63u);
        int64_t c = 0;
        for (size_t w = 0; w < words; ++w) c += __builtin_popcountll(bits[w]);
        int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (size_t)(c > 0 ? c : 1));
        int64_t k = 0;
        for (size_t w = 0; w < words; ++w) {
            uint64_t x = bits[w];
            while (x) {
                out[k++] = (int32_t)((w << 6) + (size_t)__builtin_ctzll(x));
                x &= x - 1;
            }
        }
        free(bits);
        *count = (int32_t)c;
        return out;
    }
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    memcpy(tmp, v, sizeof(int32_t) * (size_t)n);
    qsort(tmp, (size_t)n, sizeof(int32_t), cmp_i32);
    int64_t w = 0;
    for (int64_t i = 0; i < n; ++i)
        if (i == 0 || tmp[i] != tmp[i - 1]) tmp[w++] = tmp[i];
    *count = (int32_t)w;
    return tmp;
}

/* stable grouping (Seq.groupBy keeps the order of appearance inside a group) */
static void group_rows(const int32_t* dense, int64_t n, int32_t groups, int64_t** ptr_out,
                       int64_t** rows_out) {
    int64_t* ptr = (int64_t*)calloc((size_t)groups + 1, sizeof(int64_t));
    int64_t* rows = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t t = 0; t < n; ++t) ptr[dense[t] + 1]++;
    for (int32_t g = 0; g < groups; ++g) ptr[g + 1] += ptr[g];
    int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * ((size_t)groups + 1));
    memcpy(fill, ptr, sizeof(int64_t) * ((size_t)groups + 1));
    for (int64_t t = 0; t < n; ++t) rows[fill[dense[t]]++] = t;
    free(fill);
    *ptr_out = ptr;
    *rows_out = rows;
}

void orc_free(orc_model* m) {
    if (!m) return;
    free(m->user); free(m->item); free(m->rating); free(m->uid); free(m->iid);
    free(m->du); free(m->di); free(m->u_ptr); free(m->u_rows); free(m->i_ptr);
    free(m->i_rows); free(m->item_key); free(m->u_sorted); free(m->user_pos);
    free(m->user_by_pos); free(m->user_avg); free(m->item_avg); free(m->dev);
    free(m->weight); free(m->pre); free(m->item_avg_dev); free(m->item_avg_dev_spark);
    free(m);
}

orc_model* orc_fit(const int32_t* users, const int32_t* items, const double* ratings,
                   int64_t n, int* status) {
    int st = ORC_OK;
    if (n < 0 || (n > 0 && (!users || !items || !ratings))) {
        if (status) *status = ORC_E_INVALID;
        return NULL;
    }
    orc_model* m = (orc_model*)calloc(1, sizeof(orc_model));
    m->n = n;
    size_t nn = (size_t)(n > 0 ? n : 1);
    m->user = (int32_t*)malloc(sizeof(int32_t) * nn);
    m->item = (int32_t*)malloc(sizeof(int32_t) * nn);
    m->rating = (double*)malloc(sizeof(double) * nn);
    memcpy(m->user, users, sizeof(int32_t) * (size_t)n);
    memcpy(m->item, items, sizeof(int32_t) * (size_t)n);
    memcpy(m->rating, ratings, sizeof(double) * (size_t)n);

    m->uid = distinct_sorted(users, n, &m->U);
    m->iid = distinct_sorted(items, n, &m->I);
    m->du = (int32_t*)malloc(sizeof(int32_t) * nn);
    m->di = (int32_t*)malloc(sizeof(int32_t) * nn);
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < n; ++t) {
        m->du[t] = lookup(m->uid, m->U, users[t]);
        m->di[t] = lookup(m->iid, m->I, items[t]);
    }
    group_rows(m->du, n, m->U, &m->u_ptr, &m->u_rows);
    group_rows(m->di, n, m->I, &m->i_ptr, &m->i_rows);

    m->item_key = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(m->I > 0 ? m->I : 1));
    for (int32_t i = 0; i < m->I; ++i) m->item_key[i] = int_trie_key(m->iid[i]);

    /* per-user rows sorted by item trie key; duplicate (user,item) -> error */
    m->u_sorted = (int64_t*)malloc(sizeof(int64_t) * nn);
    int64_t max_row = 1, max_col = 1;
    for (int32_t u = 0; u < m->U; ++u)
        if (m->u_ptr[u + 1] - m->u_ptr[u] > max_row) max_row = m->u_ptr[u + 1] - m->u_ptr[u];
    for (int32_t i = 0; i < m->I; ++i)
        if (m->i_ptr[i + 1] - m->i_ptr[i] > max_col) max_col = m->i_ptr[i + 1] - m->i_ptr[i];
    {
        int dup = 0;
#pragma omp parallel reduction(| : dup)
        {
            keyed* ks = (keyed*)malloc(sizeof(keyed) * (size_t)max_row);
#pragma omp for schedule(dynamic, 256)
            for (int32_t u = 0; u < m->U; ++u) {
                int64_t b = m->u_ptr[u], e = m->u_ptr[u + 1];
                for (int64_t p = b; p < e; ++p) {
                    ks[p - b].key = m->item_key[m->di[m->u_rows[p]]];
                    ks[p - b].idx = m->u_rows[p];
                }
                qsort(ks, (size_t)(e - b), sizeof(keyed), cmp_keyed);
                for (int64_t p = b; p < e; ++p) {
                    m->u_sorted[p] = ks[p - b].idx;
                    if (p > b && ks[p - b].key == ks[p - b - 1].key) dup = 1;
                }
            }
            free(ks);
        }
        if (dup) st = ORC_E_DUPLICATE;
    }

    /* N3: iteration order of ratings.map(_.user).toSet (:599) */
    m->user_pos = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m->U > 0 ? m->U : 1));
    m->user_by_pos = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m->U > 0 ? m->U : 1));
    {
        /* distinct users in first-occurrence order */
        int32_t* first = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m->U > 0 ? m->U : 1));
        uint8_t* seen = (uint8_t*)calloc((size_t)(m->U > 0 ? m->U : 1), 1);
        int32_t c = 0;
        for (int64_t t = 0; t < n; ++t)
            if (!seen[m->du[t]]) {
                seen[m->du[t]] = 1;
                first[c++] = users[t];
            }
        int32_t* ord = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m->U > 0 ? m->U : 1));
        orc_int_set_order(first, m->U, ord);
        for (int32_t p = 0; p < m->U; ++p) {
            int32_t d = lookup(m->uid, m->U, ord[p]);
            m->user_pos[d] = p;
            m->user_by_pos[p] = d;
        }
        free(first); free(seen); free(ord);
    }

    /* average :94 = mean(ratings.map(_.rating)) :18 (left fold / length) */
    {
        double s = 0.0;
        for (int64_t t = 0; t < n; ++t) s = s + ratings[t];
        m->global_avg = n > 0 ? s / (double)n : 0.0;
    }
    /* usersAvg :113 / itemsAvg :134 — groupBy + average, file order */
    m->user_avg = (double*)malloc(sizeof(double) * (size_t)(m->U > 0 ? m->U : 1));
#pragma omp parallel for schedule(dynamic, 1024)
    for (int32_t u = 0; u < m->U; ++u) {
        double s = 0.0;
        for (int64_t p = m->u_ptr[u]; p < m->u_ptr[u + 1]; ++p) s = s + ratings[m->u_rows[p]];
        m->user_avg[u] = s / (double)(m->u_ptr[u + 1] - m->u_ptr[u]);
    }
    m->item_avg = (double*)malloc(sizeof(double) * (size_t)(m->I > 0 ? m->I : 1));
#pragma omp parallel for schedule(dynamic, 64)
    for (int32_t i = 0; i < m->I; ++i) {
        double s = 0.0;
        for (int64_t p = m->i_ptr[i]; p < m->i_ptr[i + 1]; ++p) s = s + ratings[m->i_rows[p]];
        m->item_avg[i] = s / (double)(m->i_ptr[i + 1] - m->i_ptr[i]);
    }

    /* computeNormalizeDeviation :155-169 */
    m->dev = (double*)malloc(sizeof(double) * nn);
    {
        int nonfinite = 0;
#pragma omp parallel for schedule(static) reduction(| : nonfinite)
        for (int64_t t = 0; t < n; ++t) {
            double ua = m->user_avg[m->du[t]];
            m->dev[t] = (ratings[t] - ua) / orc_scale(ratings[t], ua);
            if (!isfinite(m->dev[t])) nonfinite = 1;
        }
        if (nonfinite && st == ORC_OK) st = ORC_E_NONFINITE;
    }

    /* preprocessedRating :470-481.  usersWeights :474 folds x.map(pow(_,2)).sum over
     * the user's sub-map of the HashMap[(Int,Int),Double]: trie order of the tuple
     * hash (N4); hash collisions keep insertion (= file) order.  pow(x,2) == x*x. */
    m->weight = (double*)malloc(sizeof(double) * (size_t)(m->U > 0 ? m->U : 1));
    m->pre = (double*)malloc(sizeof(double) * nn);
#pragma omp parallel
    {
        keyed* ks = (keyed*)malloc(sizeof(keyed) * (size_t)max_row);
#pragma omp for schedule(dynamic, 256)
        for (int32_t u = 0; u < m->U; ++u) {
            int64_t b = m->u_ptr[u], e = m->u_ptr[u + 1];
            for (int64_t p = b; p < e; ++p) {
                int64_t t = m->u_rows[p];
                /* a Map of <= 4 entries (Map1..Map4) iterates in insertion = file order */
                ks[p - b].key = n <= 4 ? 0u : tuple_trie_key(users[t], items[t]);
                ks[p - b].idx = t;
            }
            qsort(ks, (size_t)(e - b), sizeof(keyed), cmp_keyed);
            double s = 0.0;
            for (int64_t p = 0; p < e - b; ++p) {
                double d = m->dev[ks[p].idx];
                s = s + d * d;
            }
            m->weight[u] = sqrt(s);
        }
        free(ks);
    }
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < n; ++t) {
        double w = m->weight[m->du[t]];
        m->pre[t] = (w != 0) ? m->dev[t] / w : 0.0;
    }

    /* itemsAvgDev :176-186: foldLeft over the whole HashMap (trie order of tuple
     * hashes), acc(item) = (dev + sum, 1 + count); then sum / count.
     * getItemsAvgDev :336-343 (Spark): reduceByKey, modelled in file order. */
    m->item_avg_dev = (double*)malloc(sizeof(double) * (size_t)(m->I > 0 ? m->I : 1));
    m->item_avg_dev_spark = (double*)malloc(sizeof(double) * (size_t)(m->I > 0 ? m->I : 1));
#pragma omp parallel
    {
        keyed* ks = (keyed*)malloc(sizeof(keyed) * (size_t)max_col);
#pragma omp for schedule(dynamic, 16)
        for (int32_t i = 0; i < m->I; ++i) {
            int64_t b = m->i_ptr[i], e = m->i_ptr[i + 1];
            double sf = 0.0;
            for (int64_t p = b; p < e; ++p) {
                int64_t t = m->i_rows[p];
                ks[p - b].key = n <= 4 ? 0u : tuple_trie_key(users[t], items[t]);
                ks[p - b].idx = t;
                sf = sf + m->dev[t];
            }
            m->item_avg_dev_spark[i] = sf / (double)(e - b);
            qsort(ks, (size_t)(e - b), sizeof(keyed), cmp_keyed);
            double s = 0.0;
            for (int64_t p = 0; p < e - b; ++p) s = m->dev[ks[p].idx] + s;
            m->item_avg_dev[i] = s / (double)(e - b);
        }
        free(ks);
    }

    if (status) *status = st;
    if (st != ORC_OK) {
        orc_free(m);
        return NULL;
    }
    return m;
}

int32_t orc_num_users(const orc_model* m) { return m->U; }
int32_t orc_num_items(const orc_model* m) { return m->I; }
void orc_user_iteration_order(const orc_model* m, int32_t* out) {
    for (int32_t p = 0; p < m->U; ++p) out[p] = m->uid[m->user_by_pos[p]];
}
double orc_average(const orc_model* m) { return m->global_avg; }
int orc_users_avg(const orc_model* m, int32_t user, double* out) {
    int32_t d = lookup(m->uid, m->U, user);
    if (d < 0) return 0;
    *out = m->user_avg[d];
    return 1;
}
int orc_items_avg(const orc_model* m, int32_t item, double* out) {
    int32_t d = lookup(m->iid, m->I, item);
    if (d < 0) return 0;
    *out = m->item_avg[d];
    return 1;
}
int orc_items_avg_dev(const orc_model* m, int32_t item, double* out) {
    int32_t d = lookup(m->iid, m->I, item);
    if (d < 0) return 0;
    *out = m->item_avg_dev[d];
    return 1;
}
int orc_items_avg_dev_spark(const orc_model* m, int32_t item, double* out) {
    int32_t d = lookup(m->iid, m->I, item);
    if (d < 0) return 0;
    *out = m->item_avg_dev_spark[d];
    return 1;
}
const double* orc_normalized_deviations(const orc_model* m) { return m->dev; }
const double* orc_preprocessed(const orc_model* m) { return m->pre; }
int orc_user_weight(const orc_model* m, int32_t user, double* out) {
    int32_t d = lookup(m->uid, m->U, user);
    if (d < 0) return 0;
    *out = m->weight[d];
    return 1;
}

/* computeAvgRating :101-106 */
double orc_predict_global(const orc_model* m, int32_t u, int32_t i) {
    (void)u; (void)i;
    return m->global_avg;
}
/* computeUserAvg :120-127 */
double orc_predict_user_avg(const orc_model* m, int32_t u, int32_t i) {
    (void)i;
    double v;
    return orc_users_avg(m, u, &v) ? v : m->global_avg;
}
/* computeItemAvg :141-148 */
double orc_predict_item_avg(const orc_model* m, int32_t u, int32_t i) {
    (void)u;
    double v;
    return orc_items_avg(m, i, &v) ? v : m->global_avg;
}
/* computeItemAvgDev :193-198 */
double orc_predict_item_avg_dev(const orc_model* m, int32_t u, int32_t i) {
    (void)u;
    double v;
    return orc_items_avg_dev(m, i, &v) ? v : 0.0;
}
/* computePrediction :205-237 (the (u,i) memo never changes a value) */
double orc_predict_baseline(const orc_model* m, int32_t u, int32_t i) {
    double ua;
    if (!orc_users_avg(m, u, &ua)) ua = -1.0;
    if (ua < 0.0) return m->global_avg;
    double d = orc_predict_item_avg_dev(m, u, i);
    return ua + d * orc_scale(ua + d, ua);
}
/* baselinePredictorSpark :362-391 */
double orc_predict_baseline_spark(const orc_model* m, int32_t u, int32_t i) {
    double ua;
    if (!orc_users_avg(m, u, &ua)) ua = -1.0;
    if (ua < 0.0) return m->global_avg;
    double d;
    if (!orc_items_avg_dev_spark(m, i, &d)) d = 0.0;
    return ua + d * orc_scale(ua + d, ua);
}

double orc_mae_simple(const orc_model* m, int kind, const int32_t* users, const int32_t* items,
                      const double* ratings, int64_t n, double* per_pred) {
    /* MAE :69-73, applyAndMean :80-86: foldLeft((0.0,0)) (f(x)+acc._1, acc._2+1) */
    double s = 0.0;
    for (int64_t t = 0; t < n; ++t) {
        double p;
        switch (kind) {
            case 0: p = orc_predict_global(m, users[t], items[t]); break;
            case 1: p = orc_predict_user_avg(m, users[t], items[t]); break;
            case 2: p = orc_predict_item_avg(m, users[t], items[t]); break;
            case 3: p = orc_predict_baseline(m, users[t], items[t]); break;
            default: p = orc_predict_baseline_spark(m, users[t], items[t]); break;
        }
        if (per_pred) per_pred[t] = p;
        s = fabs(ratings[t] - p) + s;
    }
    return s / (double)n; /* Double / Int; n == 0 -> NaN as in Scala (0.0/0) */
}

/* ------------------------------------------------------------------------ */
/* similarity closures                                                       */
/* ------------------------------------------------------------------------ */

static int64_t row_count(const orc_model* m, int32_t d) { return m->u_ptr[d + 1] - m->u_ptr[d]; }
/* immutable.Set of <= 4 elements keeps insertion order (Set1..Set4) */
static int small_row(const orc_model* m, int32_t d) { return row_count(m, d) <= 4; }

/* find the row of dense user d that rates dense item it (binary search in the
 * key-sorted row); -1 if absent */
static int64_t find_in_row(const orc_model* m, int32_t d, int32_t it) {
    uint32_t key = m->item_key[it];
    int64_t lo = m->u_ptr[d], hi = m->u_ptr[d + 1];
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        if (m->item_key[m->di[m->u_sorted[mid]]] < key) lo = mid + 1;
        else hi = mid;
    }
    if (lo < m->u_ptr[d + 1] && m->di[m->u_sorted[lo]] == it) return m->u_sorted[lo];
    return -1;
}

/* adjustedCosineSimilarityFunction :418-426 evaluated with `w` as the first
 * argument ("owner" of the iteration order): uItems.intersect(vItems) iterates
 * in the order of uItems — file order for <= 4 items, trie order otherwise —
 * and `.sum` is a left fold from 0.0. */
static double cos_value(const orc_model* m, int32_t w, int32_t o) {
    if (w < 0 || o < 0) return 0.0; /* getOrElse(_, Nil): empty intersection */
    double s = 0.0;
    if (small_row(m, w)) {
        for (int64_t p = m->u_ptr[w]; p < m->u_ptr[w + 1]; ++p) {
            int64_t t = m->u_rows[p];
            int64_t q = find_in_row(m, o, m->di[t]);
            if (q >= 0) s = s + m->pre[t] * m->pre[q];
        }
        return s;
    }
    int64_t a = m->u_ptr[w], ae = m->u_ptr[w + 1];
    int64_t b = m->u_ptr[o], be = m->u_ptr[o + 1];
    while (a < ae && b < be) {
        int64_t ta = m->u_sorted[a], tb = m->u_sorted[b];
        uint32_t ka = m->item_key[m->di[ta]], kb = m->item_key[m->di[tb]];
        if (ka == kb) {
            s = s + m->pre[ta] * m->pre[tb];
            ++a; ++b;
        } else if (ka < kb) ++a;
        else ++b;
    }
    return s;
}

/* jaccardCoefficient :446-463 */
static double jaccard_value(const orc_model* m, int32_t du, int32_t dv) {
    int64_t nu = du >= 0 ? row_count(m, du) : 0, nv = dv >= 0 ? row_count(m, dv) : 0;
    int64_t both = 0;
    if (du >= 0 && dv >= 0) {
        int64_t a = m->u_ptr[du], ae = m->u_ptr[du + 1];
        int64_t b = m->u_ptr[dv], be = m->u_ptr[dv + 1];
        while (a < ae && b < be) {
            uint32_t ka = m->item_key[m->di[m->u_sorted[a]]], kb = m->item_key[m->di[m->u_sorted[b]]];
            if (ka == kb) { ++both; ++a; ++b; }
            else if (ka < kb) ++a;
            else ++b;
        }
    }
    return (double)both / (double)(nu + nv - both); /* Double / Int; 0/0 -> NaN */
}

/* memo of the cosine closure (:414, :428), kept only for pairs whose result can
 * depend on the evaluation history, i.e. pairs with a <= 4-item user */
typedef struct {
    uint64_t* keys;
    int32_t* owner;
    size_t cap, used;
} pairmap;

static uint64_t pair_key(int32_t a, int32_t b) {
    uint32_t lo = (uint32_t)(a < b ? a : b), hi = (uint32_t)(a < b ? b : a);
    return (((uint64_t)hi) << 32 | lo) + 1; /* 0 = empty slot */
}
static size_t pm_slot(const pairmap* pm, uint64_t key) {
    uint64_t h = key * 0x9E3779B97F4A7C15ull;
    size_t s = (size_t)(h >> 20) & (pm->cap - 1);
    while (pm->keys[s] != 0 && pm->keys[s] != key) s = (s + 1) & (pm->cap - 1);
    return s;
}
static void pm_grow(pairmap* pm) {
    pairmap np;
    np.cap = pm->cap ? pm->cap * 2 : 1024;
    np.used = 0;
    np.keys = (uint64_t*)calloc(np.cap, sizeof(uint64_t));
    np.owner = (int32_t*)malloc(np.cap * sizeof(int32_t));
    for (size_t i = 0; i < pm->cap; ++i)
        if (pm->keys[i]) {
            size_t s = pm_slot(&np, pm->keys[i]);
            np.keys[s] = pm->keys[i];
            np.owner[s] = pm->owner[i];
            np.used++;
        }
    free(pm->keys); free(pm->owner);
    *pm = np;
}
static int32_t pm_get(const pairmap* pm, uint64_t key) {
    if (!pm->cap) return -1;
    size_t s = pm_slot(pm, key);
    return pm->keys[s] ? pm->owner[s] : -1;
}
static void pm_set(pairmap* pm, uint64_t key, int32_t owner) {
    if ((pm->used + 1) * 2 > pm->cap) pm_grow(pm);
    size_t s = pm_slot(pm, key);
    if (!pm->keys[s]) { pm->keys[s] = key; pm->used++; }
    pm->owner[s] = owner;
}

struct orc_pipeline {
    const orc_model* m;
    int32_t n_users; /* = m->U, kept so that freeing the pipeline never reads the model (which may be gone already) */
    int sim_kind;
    int32_t k;
    pairmap memo;
    /* getNeighbors memo :601 */
    uint8_t* nn_built;
    int32_t* nn_cnt;
    int32_t** nn_ids; /* dense user ids */
    double** nn_sims;
    /* fast getSimilarity lookup for the most recent user */
    int32_t mask_user;
    double* mask_sim;
    uint8_t* mask_has;
    /* scratch */
    double* row_sims;  /* U */
    double* dense_pre; /* I, scatter of one user's preprocessed row */
    uint8_t* dense_has;
    /* cached raw-similarity row (Personalized, k < 0) */
    int32_t cached_user;
    double* cached_row;
    uint8_t* cached_ok;
};

orc_pipeline* orc_pipeline_create(const orc_model* m, int sim_kind, int32_t k) {
    orc_pipeline* p = (orc_pipeline*)calloc(1, sizeof(orc_pipeline));
    size_t U = (size_t)(m->U > 0 ? m->U : 1), I = (size_t)(m->I > 0 ? m->I : 1);
    p->m = m;
    p->n_users = m->U;
    p->sim_kind = sim_kind;
    p->k = k;
    p->nn_built = (uint8_t*)calloc(U, 1);
    p->nn_cnt = (int32_t*)calloc(U, sizeof(int32_t));
    p->nn_ids = (int32_t**)calloc(U, sizeof(int32_t*));
    p->nn_sims = (double**)calloc(U, sizeof(double*));
    p->mask_user = -1;
    p->mask_sim = (double*)calloc(U, sizeof(double));
    p->mask_has = (uint8_t*)calloc(U, 1);
    p->row_sims = (double*)calloc(U, sizeof(double));
    p->dense_pre = (double*)calloc(I, sizeof(double));
    p->dense_has = (uint8_t*)calloc(I, 1);
    p->cached_user = -1;
    p->cached_row = (double*)calloc(U, sizeof(double));
    p->cached_ok = (uint8_t*)calloc(U, 1);
    return p;
}

void orc_pipeline_free(orc_pipeline* p) {
    if (!p) return;
    for (int32_t u = 0; u < p->n_users; ++u) { free(p->nn_ids[u]); free(p->nn_sims[u]); }
    free(p->memo.keys); free(p->memo.owner);
    free(p->nn_built); free(p->nn_cnt); free(p->nn_ids); free(p->nn_sims);
    free(p->mask_sim); free(p->mask_has); free(p->row_sims); free(p->dense_pre);
    free(p->dense_has); free(p->cached_row); free(p->cached_ok);
    free(p);
}

/* cosine closure :415-432 on dense ids (-1 = user absent from train) */
static double cos_closure(orc_pipeline* p, int32_t du, int32_t dv) {
    const orc_model* m = p->m;
    if (du < 0 || dv < 0) return 0.0;
    if (!small_row(m, du) && !small_row(m, dv)) return cos_value(m, du, dv); /* order-independent of owner */
    uint64_t key = pair_key(du, dv);
    int32_t owner = pm_get(&p->memo, key);
    if (owner >= 0) { /* similarities.getOrElse((u,v), -1.0); reused unless < 0.0 (:416-417) */
        double c = cos_value(m, owner, owner == du ? dv : du);
        if (!(c < 0.0)) return c;
    }
    double c = cos_value(m, du, dv);
    pm_set(&p->memo, key, du); /* + ((u,v)->sim) + ((v,u)->sim) :428 */
    return c;
}

static double raw_similarity_dense(orc_pipeline* p, int32_t du, int32_t dv) {
    switch (p->sim_kind) {
        case ORC_SIM_ONE: return 1.0;
        case ORC_SIM_JACCARD: return jaccard_value(p->m, du, dv);
        default: return cos_closure(p, du, dv);
    }
}

double orc_pipeline_raw_similarity(orc_pipeline* p, int32_t u, int32_t v) {
    return raw_similarity_dense(p, lookup(p->m->uid, p->m->U, u), lookup(p->m->uid, p->m->U, v));
}

double orc_fresh_similarity(const orc_model* m, int sim_kind, int32_t u, int32_t v) {
    int32_t du = lookup(m->uid, m->U, u), dv = lookup(m->uid, m->U, v);
    switch (sim_kind) {
        case ORC_SIM_ONE: return 1.0;
        case ORC_SIM_JACCARD: return jaccard_value(m, du, dv);
        default: return cos_value(m, du, dv);
    }
}

/* similarity of dense user du (known, > 4 items) against every user with > 4
 * items, all in trie order of the common items (identical to cos_value for
 * those pairs); ok[x] = 0 where the generic path must be used */
static void cos_row_fast(orc_pipeline* p, int32_t du, double* out, uint8_t* ok) {
    const orc_model* m = p->m;
    for (int64_t q = m->u_ptr[du]; q < m->u_ptr[du + 1]; ++q) {
        int64_t t = m->u_rows[q];
        p->dense_pre[m->di[t]] = m->pre[t];
        p->dense_has[m->di[t]] = 1;
    }
    for (int32_t x = 0; x < m->U; ++x) {
        if (small_row(m, x)) { ok[x] = 0; continue; }
        double s = 0.0;
        for (int64_t q = m->u_ptr[x]; q < m->u_ptr[x + 1]; ++q) {
            int64_t t = m->u_sorted[q];
            int32_t it = m->di[t];
            if (p->dense_has[it]) s = s + p->dense_pre[it] * m->pre[t];
        }
        out[x] = s;
        ok[x] = 1;
    }
    for (int64_t q = m->u_ptr[du]; q < m->u_ptr[du + 1]; ++q) p->dense_has[m->di[m->u_rows[q]]] = 0;
}

/* stable merge sort of idx[] by sims[idx] descending: sortWith(_._2 > _._2) :610 is
 * java.util.Arrays.sort(Object[], Comparator) = stable TimSort */
static void stable_sort_desc(int32_t* idx, int32_t n, const double* sims, int32_t* tmp) {
    if (n < 2) return;
    int32_t h = n / 2;
    stable_sort_desc(idx, h, sims, tmp);
    stable_sort_desc(idx + h, n - h, sims, tmp);
    int32_t a = 0, b = h, w = 0;
    while (a < h && b < n) {
        /* take from the right run only if it is strictly greater (stability) */
        if (sims[idx[b]] > sims[idx[a]]) tmp[w++] = idx[b++];
        else tmp[w++] = idx[a++];
    }
    while (a < h) tmp[w++] = idx[a++];
    while (b < n) tmp[w++] = idx[b++];
    memcpy(idx, tmp, sizeof(int32_t) * (size_t)n);
}

/* getNeighbors closure :603-616 for dense user du (or an unseen raw user when du < 0).
 * Returns malloc'ed arrays of length *cnt. */
static void build_neighbors(orc_pipeline* p, int32_t du, int32_t** ids_out, double** sims_out,
                            int32_t* cnt_out) {
    const orc_model* m = p->m;
    int32_t U = m->U;
    double* sims = p->row_sims;
    uint8_t* ok = (uint8_t*)calloc((size_t)(U > 0 ? U : 1), 1);
    if (p->sim_kind == ORC_SIM_COSINE && du >= 0 && !small_row(m, du)) cos_row_fast(p, du, sims, ok);
    /* others = (allUsers - u).toSeq :608, in Set iteration order (N3) */
    int32_t* others = (int32_t*)malloc(sizeof(int32_t) * (size_t)(U > 0 ? U : 1));
    int32_t c = 0;
    for (int32_t pos = 0; pos < U; ++pos) {
        int32_t x = m->user_by_pos[pos];
        if (x == du) continue;
        if (!ok[x]) sims[x] = raw_similarity_dense(p, du, x);
        others[c++] = x;
    }
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)(c > 0 ? c : 1));
    stable_sort_desc(others, c, sims, tmp);
    free(tmp);
    int32_t k = p->k < c ? p->k : c; /* take(k) */
    if (k < 0) k = 0;
    int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)(k > 0 ? k : 1));
    double* ss = (double*)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
    for (int32_t j = 0; j < k; ++j) { ids[j] = others[j]; ss[j] = sims[others[j]]; }
    free(others); free(ok);
    *ids_out = ids; *sims_out = ss; *cnt_out = k;
}

static void ensure_neighbors(orc_pipeline* p, int32_t du) {
    if (p->nn_built[du] && p->nn_cnt[du] > 0) return; /* `if (uNeighbors.isEmpty)` :606 */
    free(p->nn_ids[du]); free(p->nn_sims[du]);
    build_neighbors(p, du, &p->nn_ids[du], &p->nn_sims[du], &p->nn_cnt[du]);
    p->nn_built[du] = 1;
}

int32_t orc_pipeline_neighbors(orc_pipeline* p, int32_t u, int32_t cap, int32_t* ids, double* sims) {
    const orc_model* m = p->m;
    int32_t du = lookup(m->uid, m->U, u);
    int32_t* di; double* ds; int32_t cnt;
    if (du >= 0) {
        ensure_neighbors(p, du);
        di = p->nn_ids[du]; ds = p->nn_sims[du]; cnt = p->nn_cnt[du];
    } else {
        build_neighbors(p, -1, &di, &ds, &cnt);
    }
    for (int32_t j = 0; j < cnt && j < cap; ++j) { ids[j] = m->uid[di[j]]; sims[j] = ds[j]; }
    if (du < 0) { free(di); free(ds); }
    return cnt;
}

static void load_mask(orc_pipeline* p, int32_t du) {
    if (p->mask_user == du) return;
    if (p->mask_user >= 0)
        for (int32_t j = 0; j < p->nn_cnt[p->mask_user]; ++j) p->mask_has[p->nn_ids[p->mask_user][j]] = 0;
    ensure_neighbors(p, du);
    for (int32_t j = 0; j < p->nn_cnt[du]; ++j) {
        p->mask_has[p->nn_ids[du][j]] = 1;
        p->mask_sim[p->nn_ids[du][j]] = p->nn_sims[du][j];
    }
    p->mask_user = du;
}

/* getSimilarity closure :634-648: nn(user1).map(x => if (x._1==user2) x._2 else 0.0).sum */
static double knn_similarity_dense(orc_pipeline* p, int32_t du, int32_t dv) {
    load_mask(p, du);
    if (dv >= 0 && p->mask_has[dv]) return 0.0 + p->mask_sim[dv];
    return 0.0;
}

double orc_pipeline_knn_similarity(orc_pipeline* p, int32_t u, int32_t v) {
    const orc_model* m = p->m;
    int32_t du = lookup(m->uid, m->U, u), dv = lookup(m->uid, m->U, v);
    if (du < 0) { /* unseen user1: neighbours are recomputed each call, all sims are 0.0 */
        int32_t* di; double* ds; int32_t cnt;
        build_neighbors(p, -1, &di, &ds, &cnt);
        double s = 0.0;
        for (int32_t j = 0; j < cnt; ++j) s = s + (di[j] == dv ? ds[j] : 0.0);
        free(di); free(ds);
        return s;
    }
    return knn_similarity_dense(p, du, dv);
}

/* weightedSumDeviation closure :504-548 for a user present in train */
static double wsd_dense(orc_pipeline* p, int32_t du, int32_t item_raw) {
    const orc_model* m = p->m;
    int32_t it = lookup(m->iid, m->I, item_raw);
    if (it < 0) return 0.0; /* ratedI.getOrElse(i, Seq()) empty -> den = 0 -> 0.0 */
    double num = 0.0, den = 0.0;
    if (p->k < 0 && p->sim_kind == ORC_SIM_COSINE && !small_row(m, du) && p->cached_user != du) {
        cos_row_fast(p, du, p->cached_row, p->cached_ok);
        p->cached_user = du;
    }
    for (int64_t q = m->i_ptr[it]; q < m->i_ptr[it + 1]; ++q) {
        int64_t t = m->i_rows[q];
        int32_t x = m->du[t];
        double d = m->dev[t]; /* (x.rating-avgU)/scale(x.rating, avgU) :516, same expression as :167 */
        double s;
        if (p->k >= 0) s = knn_similarity_dense(p, du, x);
        else if (p->sim_kind == ORC_SIM_COSINE && p->cached_user == du && p->cached_ok[x]) s = p->cached_row[x];
        else s = raw_similarity_dense(p, du, x);
        num = num + d * s;
        den = den + fabs(s);
    }
    return den > 0 ? num / den : 0.0;
}

double orc_pipeline_wsd(orc_pipeline* p, int32_t u, int32_t i) {
    const orc_model* m = p->m;
    int32_t du = lookup(m->uid, m->U, u);
    if (du >= 0) return wsd_dense(p, du, i);
    /* user absent from train: F(u, x) with an empty item set */
    int32_t it = lookup(m->iid, m->I, i);
    if (it < 0) return 0.0;
    double num = 0.0, den = 0.0;
    for (int64_t q = m->i_ptr[it]; q < m->i_ptr[it + 1]; ++q) {
        int64_t t = m->i_rows[q];
        double s = p->k >= 0 ? orc_pipeline_knn_similarity(p, u, m->user[t])
                             : raw_similarity_dense(p, -1, m->du[t]);
        num = num + m->dev[t] * s;
        den = den + fabs(s);
    }
    return den > 0 ? num / den : 0.0;
}

/* predictor closure :568-585 */
double orc_pipeline_predict(orc_pipeline* p, int32_t u, int32_t i) {
    const orc_model* m = p->m;
    int32_t du = lookup(m->uid, m->U, u);
    double ua = du >= 0 ? m->user_avg[du] : -1.0;
    if (ua < 0.0) return m->global_avg;
    double w = wsd_dense(p, du, i);
    return ua + w * orc_scale(ua + w, ua);
}

double orc_pipeline_mae(orc_pipeline* p, const int32_t* users, const int32_t* items,
                        const double* ratings, int64_t n, double* per_pred) {
    double s = 0.0;
    for (int64_t t = 0; t < n; ++t) {
        double pr = orc_pipeline_predict(p, users[t], items[t]);
        if (per_pred) per_pred[t] = pr;
        s = fabs(ratings[t] - pr) + s;
    }
    return s / (double)n;
}

/* ---- recommendations :651-674 --------------------------------------------- */
typedef struct { int32_t item; double pred; } reco_t;

/* the reference's `order`: x before y iff (x._2 == y._2) ? x._1 < y._1 : x._2 > y._2.  Item ids are distinct,
 * so this is a strict total order and the stable sortWith has exactly one outcome. */
static int cmp_reco(const void* a, const void* b) {
    const reco_t* x = (const reco_t*)a;
    const reco_t* y = (const reco_t*)b;
    if (x->pred == y->pred) return x->item < y->item ? -1 : (x->item > y->item ? 1 : 0);
    return x->pred > y->pred ? -1 : 1;
}

int32_t orc_recommend(const orc_model* m, orc_pipeline* p, int simple_kind, int32_t user, int32_t n,
                      int32_t* out_items, double* out_preds) {
    if (n <= 0) return 0;
    /* notRated = ratings.map(_.item).toSet.diff(items of `user`) :667 */
    char* rated = (char*)calloc((size_t)m->I > 0 ? (size_t)m->I : 1, 1);
    reco_t* r = (reco_t*)malloc(sizeof(reco_t) * (size_t)(m->I > 0 ? m->I : 1));
    if (!rated || !r) { free(rated); free(r); return -1; }
    int32_t du = lookup(m->uid, m->U, user);
    if (du >= 0)
        for (int64_t q = m->u_ptr[du]; q < m->u_ptr[du + 1]; ++q) rated[m->di[m->u_rows[q]]] = 1;
    int32_t cnt = 0;
    for (int32_t i = 0; i < m->I; ++i) {
        if (rated[i]) continue;
        const int32_t item = m->iid[i];
        double pr;
        if (p) pr = orc_pipeline_predict(p, user, item);
        else {
            switch (simple_kind) {
                case 0: pr = orc_predict_global(m, user, item); break;
                case 1: pr = orc_predict_user_avg(m, user, item); break;
                case 2: pr = orc_predict_item_avg(m, user, item); break;
                case 4: pr = orc_predict_baseline_spark(m, user, item); break;
                default: pr = orc_predict_baseline(m, user, item); break;
            }
        }
        r[cnt].item = item;
        r[cnt].pred = pr;
        ++cnt;
    }
    qsort(r, (size_t)cnt, sizeof(reco_t), cmp_reco);
    if (cnt > n) cnt = n;
    for (int32_t j = 0; j < cnt; ++j) { out_items[j] = r[j].item; out_preds[j] = r[j].pred; }
    free(rated); free(r);
    return cnt;
}

/* ------------------------------------------------------------------------ */
/* bulk evaluation on all cores                                              */
/* ------------------------------------------------------------------------ */
/*
 * The same closures — adjustedCosineSimilarityFunction :407-433 under getNeighbors
 * :596-617, getSimilarity :626-649, weightedSumDeviation :489-549, predictor
 * :557-586, MAE :69-73 — evaluated for MANY users at once, one user per thread.
 * This is legal exactly when no user has <= 4 ratings: then no value depends on
 * the memo history (N6) and every neighbourhood is a pure function of the model.
 * orc_knn_table_build refuses (ORC_E_INVALID) otherwise.
 *
 * Evaluation order.  The per-pair form (cos_value above) walks both users' item
 * sets in trie order and adds pre(u,i) * pre(v,i) for the common items, left to
 * right from 0.0.  The row form used here walks u's items in the SAME trie order
 * and, for each item, adds pre(u,i) * pre(v,i) to acc[v] for every rater v of i:
 * for a fixed v the additions into acc[v] are the common items of u and v in
 * ascending trie key, starting from 0.0 — the same operands in the same order,
 * hence the same bits (tests/test_oracle_bulk.py pins the two forms to each
 * other on every user of the ml-100k shape and on random cases).  What changes
 * is the cost: sum_i |U(i)|^2 multiply-adds for all users instead of U * N.
 *
 * stable sortWith(_._2 > _._2).take(k) :610 over (allUsers - u).toSeq in Set
 * order == the k best by (similarity descending, Set position ascending); they
 * are found with a quickselect on the value and sorted with the same stable
 * merge sort the per-user path uses.
 */
#ifdef _OPENMP
#include <omp.h>
#endif

struct orc_knn_table {
    const orc_model* m;
    int32_t k, kk; /* requested k, stored width min(k, U-1) */
    int32_t n_rows;
    int32_t* row_user;    /* [n_rows] dense user of row r */
    int32_t* row_of_user; /* [U] row of a dense user or -1 */
    int32_t* ids;         /* [n_rows * kk] dense neighbour ids, reference order */
    double* sims;         /* [n_rows * kk] */
    int64_t* it_ptr;      /* item-major copy: raters (dense user) and their preprocessed ratings */
    int32_t* it_user;
    double* it_pre;
    double* it_dev;       /* normalized deviation of the same entries (file order inside an item, like i_rows) */
};

static void swap_d(double* a, double* b) { double t = *a; *a = *b; *b = t; }

/* the kth (0-based) LARGEST value of v[0..n) (v is permuted) */
static double select_desc(double* v, int64_t n, int64_t kth) {
    int64_t lo = 0, hi = n - 1;
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        /* median of three -> v[mid] */
        if (v[mid] > v[lo]) swap_d(&v[mid], &v[lo]);
        if (v[hi] > v[lo]) swap_d(&v[hi], &v[lo]);
        if (v[hi] > v[mid]) swap_d(&v[hi], &v[mid]);
        double pv = v[mid];
        int64_t i = lo, j = hi;
        while (i <= j) {
            while (v[i] > pv) ++i;
            while (v[j] < pv) --j;
            if (i <= j) { swap_d(&v[i], &v[j]); ++i; --j; }
        }
        if (kth <= j) hi = j;
        else if (kth >= i) lo = i;
        else return v[kth];
    }
    return v[kth];
}

void orc_knn_table_free(orc_knn_table* t) {
    if (!t) return;
    free(t->row_user); free(t->row_of_user); free(t->ids); free(t->sims);
    free(t->it_ptr); free(t->it_user); free(t->it_pre); free(t->it_dev);
    free(t);
}

orc_knn_table* orc_knn_table_build(const orc_model* m, int32_t k, const int32_t* users_raw, int32_t n_users,
                                   int threads, int* status) {
    if (status) *status = ORC_OK;
    if (!m || k < 0 || (users_raw && n_users < 0)) { if (status) *status = ORC_E_INVALID; return NULL; }
    const int32_t U = m->U;
    for (int32_t u = 0; u < U; ++u)
        if (small_row(m, u)) { if (status) *status = ORC_E_INVALID; return NULL; } /* memo-history dependent: use orc_pipeline */
    orc_knn_table* t = (orc_knn_table*)calloc(1, sizeof(orc_knn_table));
    t->m = m;
    t->k = k;
    t->kk = k < U - 1 ? k : U - 1;
    if (t->kk < 0) t->kk = 0;
    t->n_rows = users_raw ? n_users : U;
    const size_t rows1 = (size_t)(t->n_rows > 0 ? t->n_rows : 1), kk1 = (size_t)(t->kk > 0 ? t->kk : 1);
    t->row_user = (int32_t*)malloc(sizeof(int32_t) * rows1);
    t->row_of_user = (int32_t*)malloc(sizeof(int32_t) * (size_t)(U > 0 ? U : 1));
    for (int32_t u = 0; u < U; ++u) t->row_of_user[u] = -1;
    for (int32_t r = 0; r < t->n_rows; ++r) {
        int32_t d = users_raw ? lookup(m->uid, U, users_raw[r]) : r;
        if (d < 0 || t->row_of_user[d] >= 0) { /* unknown or repeated user */
            if (status) *status = ORC_E_INVALID;
            orc_knn_table_free(t);
            return NULL;
        }
        t->row_user[r] = d;
        t->row_of_user[d] = r;
    }
    t->ids = (int32_t*)malloc(sizeof(int32_t) * rows1 * kk1);
    t->sims = (double*)malloc(sizeof(double) * rows1 * kk1);
    /* item-major copy (any rater order: each acc[v] receives at most one addend per item) */
    const size_t nn = (size_t)(m->n > 0 ? m->n : 1);
    t->it_ptr = (int64_t*)malloc(sizeof(int64_t) * ((size_t)m->I + 1));
    t->it_user = (int32_t*)malloc(sizeof(int32_t) * nn);
    t->it_pre = (double*)malloc(sizeof(double) * nn);
    t->it_dev = (double*)malloc(sizeof(double) * nn);
    memcpy(t->it_ptr, m->i_ptr, sizeof(int64_t) * ((size_t)m->I + 1));
    for (int64_t q = 0; q < m->n; ++q) {
        t->it_user[q] = m->du[m->i_rows[q]];
        t->it_pre[q] = m->pre[m->i_rows[q]];
        t->it_dev[q] = m->dev[m->i_rows[q]];
    }
    if (t->kk == 0 || t->n_rows == 0) return t;
    int nomem = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
    {
        double* acc = (double*)calloc((size_t)U, sizeof(double));
        double* val = (double*)malloc(sizeof(double) * (size_t)U);
        int32_t* pick = (int32_t*)malloc(sizeof(int32_t) * ((size_t)t->kk + 1));
        int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * ((size_t)t->kk + 1));
        if (!acc || !val || !pick || !tmp) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            nomem = 1;
        } else {
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
            for (int32_t r = 0; r < t->n_rows; ++r) {
                const int32_t du = t->row_user[r];
                /* :418-426 for every v at once: u's items in trie order (the owner's iteration order) */
                for (int64_t p = m->u_ptr[du]; p < m->u_ptr[du + 1]; ++p) {
                    const int64_t tu = m->u_sorted[p];
                    const double xu = m->pre[tu];
                    const int32_t it = m->di[tu];
                    for (int64_t q = t->it_ptr[it]; q < t->it_ptr[it + 1]; ++q) {
                        const int32_t v = t->it_user[q];
                        acc[v] = acc[v] + xu * t->it_pre[q];
                    }
                }
                /* :608-610: (allUsers - u).toSeq in Set order, stable sort by similarity descending, take(k):
                 * everything above the kk-th largest value, then the ties at that value in Set order */
                int64_t c = 0;
                for (int32_t x = 0; x < U; ++x)
                    if (x != du) val[c++] = acc[x];
                const double thr = select_desc(val, c, t->kk - 1);
                int32_t n_pick = 0;
                int64_t n_eq = 0; /* ties at thr, collected in val's storage as dense ids */
                int32_t* eq = (int32_t*)val;
                for (int32_t x = 0; x < U; ++x) {
                    if (x == du) continue;
                    if (acc[x] > thr) pick[n_pick++] = x;
                    else if (acc[x] == thr) eq[n_eq++] = x;
                }
                const int32_t need_eq = t->kk - n_pick;
                if (n_eq > need_eq) { /* the need_eq ties that come first in Set order (N3) */
                    keyed* ks = (keyed*)malloc(sizeof(keyed) * (size_t)n_eq);
                    for (int64_t j = 0; j < n_eq; ++j) { ks[j].key = (uint32_t)m->user_pos[eq[j]]; ks[j].idx = eq[j]; }
                    qsort(ks, (size_t)n_eq, sizeof(keyed), cmp_keyed);
                    for (int32_t j = 0; j < need_eq; ++j) pick[n_pick++] = (int32_t)ks[j].idx;
                    free(ks);
                } else {
                    for (int64_t j = 0; j < n_eq; ++j) pick[n_pick++] = eq[j];
                }
                /* into Set order, then the stable sort by value: (similarity descending, Set position ascending) */
                {
                    keyed* ks = (keyed*)malloc(sizeof(keyed) * (size_t)(n_pick > 0 ? n_pick : 1));
                    for (int32_t j = 0; j < n_pick; ++j) { ks[j].key = (uint32_t)m->user_pos[pick[j]]; ks[j].idx = pick[j]; }
                    qsort(ks, (size_t)n_pick, sizeof(keyed), cmp_keyed);
                    for (int32_t j = 0; j < n_pick; ++j) pick[j] = (int32_t)ks[j].idx;
                    free(ks);
                }
                stable_sort_desc(pick, n_pick, acc, tmp);
                for (int32_t j = 0; j < t->kk; ++j) {
                    t->ids[(size_t)r * t->kk + j] = pick[j];
                    t->sims[(size_t)r * t->kk + j] = acc[pick[j]];
                }
                memset(acc, 0, sizeof(double) * (size_t)U);
            }
        }
        free(acc); free(val); free(pick); free(tmp);
    }
    if (nomem) {
        if (status) *status = ORC_E_NOMEM;
        orc_knn_table_free(t);
        return NULL;
    }
    return t;
}

int32_t orc_knn_table_width(const orc_knn_table* t) { return t->kk; }
int32_t orc_knn_table_rows(const orc_knn_table* t) { return t->n_rows; }
const double* orc_knn_table_sims(const orc_knn_table* t) { return t->sims; }
/* raw ids of every stored neighbour, [rows * width]; raw user of every row */
void orc_knn_table_export_ids(const orc_knn_table* t, int32_t* out_ids, int32_t* out_row_user) {
    const size_t cells = (size_t)t->n_rows * (size_t)t->kk;
    for (size_t c = 0; c < cells; ++c) out_ids[c] = t->m->uid[t->ids[c]];
    for (int32_t r = 0; r < t->n_rows; ++r) out_row_user[r] = t->m->uid[t->row_user[r]];
}

/* predictor(train, weightedSumDeviation(train, getSimilarity(train, k, cos))) :557-586 over (users, items) with
 * the table's neighbourhoods; every KNOWN test user must have a row.  out_pred[t] per row (file order); returns
 * the MAE :69-73 as a left fold in file order when ratings != NULL (else 0).  One user per thread; the per-row
 * arithmetic is wsd_dense / orc_pipeline_predict above, unchanged: every rater of the item in file order, the
 * neighbour's similarity or 0.0. */
double orc_knn_table_predict(const orc_knn_table* t, const int32_t* users, const int32_t* items,
                             const double* ratings, int64_t n, int threads, double* out_pred, int* status) {
    const orc_model* m = t->m;
    const int32_t U = m->U;
    if (status) *status = ORC_OK;
    /* rows grouped by dense user (counting sort, file order inside a user) */
    int32_t* du = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int64_t* ptr = (int64_t*)calloc((size_t)U + 2, sizeof(int64_t));
    int64_t* rows = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t r = 0; r < n; ++r) {
        du[r] = lookup(m->uid, U, users[r]);
        if (du[r] >= 0 && t->row_of_user[du[r]] < 0 && t->kk > 0) {
            if (status) *status = ORC_E_INVALID;
            free(du); free(ptr); free(rows);
            return 0.0;
        }
        ptr[(du[r] >= 0 ? du[r] : U) + 1]++;
    }
    for (int32_t g = 0; g <= U; ++g) ptr[g + 1] += ptr[g];
    {
        int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * ((size_t)U + 2));
        memcpy(fill, ptr, sizeof(int64_t) * ((size_t)U + 2));
        for (int64_t r = 0; r < n; ++r) rows[fill[du[r] >= 0 ? du[r] : U]++] = r;
        free(fill);
    }
    for (int64_t j = ptr[U]; j < ptr[U + 1]; ++j) out_pred[rows[j]] = m->global_avg; /* usersAvg.getOrElse(u, -1) < 0 :571-574 */
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads)
#endif
    {
        uint8_t* has = (uint8_t*)calloc((size_t)(U > 0 ? U : 1), 1);
        double* msim = (double*)calloc((size_t)(U > 0 ? U : 1), sizeof(double));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 32)
#endif
        for (int32_t u = 0; u < U; ++u) {
            if (ptr[u] == ptr[u + 1]) continue;
            const int32_t row = t->kk > 0 ? t->row_of_user[u] : -1;
            for (int32_t j = 0; row >= 0 && j < t->kk; ++j) {
                has[t->ids[(size_t)row * t->kk + j]] = 1;
                msim[t->ids[(size_t)row * t->kk + j]] = t->sims[(size_t)row * t->kk + j];
            }
            for (int64_t j = ptr[u]; j < ptr[u + 1]; ++j) {
                const int64_t r = rows[j];
                const int32_t it = lookup(m->iid, m->I, items[r]);
                double w = 0.0;
                if (it >= 0) { /* :513-547 */
                    double num = 0.0, den = 0.0;
                    /* ratedI(i) in file order: (rater, deviation) = (m->du[t], m->dev[t]) for t in i_rows, read from
                     * the table's contiguous copies */
                    for (int64_t q = m->i_ptr[it]; q < m->i_ptr[it + 1]; ++q) {
                        const int32_t x = t->it_user[q];
                        const double s = has[x] ? 0.0 + msim[x] : 0.0; /* getSimilarity :638-641 */
                        num = num + t->it_dev[q] * s;
                        den = den + fabs(s);
                    }
                    w = den > 0 ? num / den : 0.0;
                }
                const double ua = m->user_avg[u];
                out_pred[r] = ua < 0.0 ? m->global_avg : ua + w * orc_scale(ua + w, ua);
            }
            for (int32_t j = 0; row >= 0 && j < t->kk; ++j) has[t->ids[(size_t)row * t->kk + j]] = 0;
        }
        free(has); free(msim);
    }
    double s = 0.0;
    if (ratings)
        for (int64_t r = 0; r < n; ++r) s = fabs(ratings[r] - out_pred[r]) + s;
    free(du); free(ptr); free(rows);
    return ratings ? s / (double)n : 0.0;
}

/* caps the threads of every later OpenMP region of this library (orc_fit's per-user / per-item loops and the bulk form);
 * n <= 0 restores the default (all cores) */
void orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
#else
    (void)n;
#endif
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
