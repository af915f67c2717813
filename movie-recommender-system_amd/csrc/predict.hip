// predict.hip — K7-K9: weighted-sum prediction, the non-personalised predictors and the MAE.
//
// kNN prediction (weightedSumDeviation shared/predictions.scala:504-548 through getSimilarity
// :634-648, then predictor :568-585): for a test pair (u, i)
//     num = sum_{x in U(i)} dev(x,i) * s~(u,x),  den = sum |s~(u,x)|   (file order of i's raters)
// where s~(u,x) is non-zero only for the k neighbours of u.  One wavefront per test rating:
// lanes stride over u's neighbour list, each probes "did neighbour v rate item i" by binary search
// in v's item-sorted row, matches are compacted into LDS with a ballot/mbcnt prefix, ordered by
// the training file row of the matched rating with a wave-level bitonic sort, and folded left in
// fp64 — the same additions in the same order as the reference (terms with s~ = 0 add +-0.0 and
// are identities).  HBM/L2-bound gathers: ~12 k bytes of neighbour list + probes per prediction.
#include <math.h>

#include "engine.h"

namespace knncf {

static constexpr int TPB = 256;
struct PredArgs {
    const int64_t* u_ptr;
    const int32_t* s_col;
    const uint32_t* s_t;
    const double* s_dev;
    const double* user_avg;
    const double* item_avg;
    const double* item_dev_hash;
    const double* item_dev_file;
    double global_avg;
    int32_t own_lo, own_hi;
    // neighbour table
    const int32_t* nbr_uidx;  // neighbour ids sorted ascending
    const double* nbr_usim;
    const int32_t* nbr_cnt;
    int32_t kcap;
    // item-major rows, raters ascending
    const int64_t* i_ptr;
    const int32_t* it_user;
    const double* it_dev;
    const uint32_t* it_t;
    // per-item rater bitmaps (ib_words == 0: not built, binary search instead)
    int64_t ib_words;
    const unsigned long long* item_bits;
    const uint32_t* item_rank;
};

// predictor :568-585 given the user's mean and the weighted-sum deviation
__device__ __forceinline__ double combine(double ua, double wsd) { return ua + wsd * scale_fn(ua + wsd, ua); }

__device__ __forceinline__ void wave_sync() {
    // lanes of one wave exchange data through LDS: order the accesses for the compiler (the LDS
    // queue itself is in order per wave)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int CAP, int WAVES>  // per-wave match capacity (power of two >= kcap), waves per block
__global__ void __launch_bounds__(WAVES * 64) k_predict_knn(PredArgs A, int64_t n, const int32_t* __restrict__ du,
                                                     const int32_t* __restrict__ di, const double* __restrict__ ratings,
                                                     const uint32_t* __restrict__ order, double* __restrict__ pred,
                                                     double* __restrict__ abs_err, uint8_t* __restrict__ owned,
                                                     int unknown_owned) {
    __shared__ uint32_t m_t[WAVES][CAP];
    __shared__ double m_dev[WAVES][CAP];
    __shared__ double m_sim[WAVES][CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * WAVES + wave;
    if (w >= n) return;
    const int64_t t = order ? (int64_t)order[w] : w;
    const int32_t u = du[t], i = di[t];
    const bool mine = (u < 0) ? (unknown_owned != 0) : (u >= A.own_lo && u < A.own_hi);
    if (!mine) {
        if (lane == 0) { owned[t] = 0; abs_err[t] = 0.0; }
        return;
    }
    double p;
    double ua = (u >= 0) ? A.user_avg[u] : -1.0;  // usersAvgValue.getOrElse(u, -1.0) :572
    if (ua < 0.0) {
        p = A.global_avg;
    } else if (i < 0) {
        p = combine(ua, 0.0);  // no rater: den = 0 -> 0.0 :527-529
    } else {
        uint32_t* mt = m_t[wave];
        double* md = m_dev[wave];
        double* ms = m_sim[wave];
        const int32_t cnt = A.nbr_cnt[u];
        const int64_t base = (int64_t)u * A.kcap;
        int32_t total = 0;
        // "which of u's neighbours rated item i": u's neighbour ids (sorted ascending) are looked up in the
        // item's rater list (sorted the same way).  All lanes search the SAME array and neighbouring lanes
        // look for neighbouring ids, so the searches share their cache sectors (the per-neighbour-row
        // searches they replace touched ~7 private sectors each and were bound by L2 sector bandwidth).
        const int64_t rb = A.i_ptr[i], re = A.i_ptr[i + 1];
        for (int32_t j0 = 0; j0 < cnt; j0 += 64) {
            const int32_t j = j0 + lane;
            int64_t lo = rb, hi = rb;
            int32_t x = 0;
            double s = 0.0;
            if (j < cnt) {
                x = A.nbr_uidx[base + j];
                s = A.nbr_usim[base + j];
                hi = re;
            }
            bool found;
            if (A.ib_words > 0) {
                // rater bitmap of item i (U bits) + rank prefixes: one 8-byte read per neighbour, one more on a hit
                found = false;
                if (j < cnt) {
                    const int64_t w = (int64_t)i * A.ib_words + (x >> 6);
                    const unsigned long long word = A.item_bits[w];
                    found = (word >> (x & 63)) & 1ull;
                    if (found) lo = rb + A.item_rank[w] + __popcll(word & ((1ull << (x & 63)) - 1ull));
                }
            } else {
                while (__any(lo < hi)) {
                    if (lo < hi) {
                        const int64_t mid = (lo + hi) >> 1;
                        if (A.it_user[mid] < x) lo = mid + 1;
                        else hi = mid;
                    }
                }
                found = j < cnt && lo < re && A.it_user[lo] == x;
            }
            const unsigned long long hit = __ballot(found);
            if (found) {
                const int32_t slot = total + __popcll(hit & ((1ull << lane) - 1ull));
                mt[slot] = A.it_t[lo];
                md[slot] = A.it_dev[lo];
                ms[slot] = s;
            }
            total += __popcll(hit);
        }
        // order the matches by training file row (the order of ratedI(i) :508-517)
        int32_t m = 1;
        while (m < total) m <<= 1;
        for (int32_t c = total + lane; c < m; c += 64) mt[c] = 0xffffffffu;
        wave_sync();
        for (int32_t size = 2; size <= m; size <<= 1) {
            for (int32_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (int32_t q = lane; q < (m >> 1); q += 64) {
                    int32_t lo = 2 * q - (q & (stride - 1));
                    int32_t hi = lo + stride;
                    bool up = ((lo & size) == 0);
                    uint32_t ta = mt[lo], tb = mt[hi];
                    if ((ta < tb) != up) {
                        mt[lo] = tb; mt[hi] = ta;
                        double x = md[lo]; md[lo] = md[hi]; md[hi] = x;
                        double y = ms[lo]; ms[lo] = ms[hi]; ms[hi] = y;
                    }
                }
                wave_sync();
            }
        }
        double num = 0.0, den = 0.0;
        for (int32_t c = 0; c < total; ++c) {  // every lane folds the same sequence (LDS broadcast)
            double s = ms[c];
            num = num + md[c] * s;
            den = den + fabs(s);
        }
        double wsd = (den > 0) ? num / den : 0.0;
        p = combine(ua, wsd);
    }
    if (lane == 0) {
        pred[t] = p;
        abs_err[t] = ratings ? fabs(ratings[t] - p) : 0.0;
        owned[t] = 1;
    }
}

// the closed-form predictors: computeAvgRating :101, computeUserAvg :120, computeItemAvg :141,
// computePrediction :205-237, baselinePredictorSpark :362-391, and
// predictor(train, weightedSumDeviation(train, similarityOne)) (predict/Personalized.scala:61)
__global__ void k_predict_simple(PredArgs A, int kind, int64_t n, const int32_t* __restrict__ du,
                                 const int32_t* __restrict__ di, const double* __restrict__ ratings,
                                 double* __restrict__ pred, double* __restrict__ abs_err, uint8_t* __restrict__ owned,
                                 int unknown_owned) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int32_t u = du[t], i = di[t];
    const bool mine = (u < 0) ? (unknown_owned != 0) : (u >= A.own_lo && u < A.own_hi);
    if (!mine) {
        owned[t] = 0;
        abs_err[t] = 0.0;
        return;
    }
    double p;
    switch (kind) {
        case KNNCF_PRED_GLOBAL_AVG: p = A.global_avg; break;
        case KNNCF_PRED_USER_AVG: p = (u >= 0) ? A.user_avg[u] : A.global_avg; break;
        case KNNCF_PRED_ITEM_AVG: p = (i >= 0) ? A.item_avg[i] : A.global_avg; break;
        default: {
            double ua = (u >= 0) ? A.user_avg[u] : -1.0;
            if (ua < 0.0) {
                p = A.global_avg;
            } else {
                double d = 0.0;
                if (i >= 0) d = (kind == KNNCF_PRED_BASELINE) ? A.item_dev_hash[i] : A.item_dev_file[i];
                p = combine(ua, d);
            }
        }
    }
    pred[t] = p;
    abs_err[t] = ratings ? fabs(ratings[t] - p) : 0.0;
    owned[t] = 1;
}

void launch_predict(const Train& tr, const NeighborTable* nt, int predictor, int64_t n, const int32_t* d_du,
                    const int32_t* d_di, const double* d_ratings, const uint32_t* d_order, double* d_pred,
                    double* d_abs_err, uint8_t* d_owned, bool unknown_users_owned, hipStream_t st) {
    if (n <= 0) return;
    PredArgs A{};
    A.u_ptr = tr.u_ptr.p; A.s_col = tr.s_col.p; A.s_t = tr.s_t.p; A.s_dev = tr.s_dev.p;
    A.user_avg = tr.user_avg.p; A.item_avg = tr.item_avg.p; A.item_dev_hash = tr.item_dev_hash.p;
    A.item_dev_file = tr.item_dev_file.p; A.global_avg = tr.global_avg;
    A.own_lo = tr.own_lo; A.own_hi = tr.own_hi;
    if (predictor == KNNCF_PRED_KNN) {
        KN_REQUIRE(nt != nullptr, KNNCF_E_STATE, "predict: neighbour table missing");
        A.nbr_uidx = nt->uidx.p; A.nbr_usim = nt->usim.p; A.nbr_cnt = nt->cnt.p; A.kcap = nt->kcap;
        A.i_ptr = tr.i_ptr.p; A.it_user = tr.it_user.p; A.it_dev = tr.it_dev.p; A.it_t = tr.it_t.p;
        A.ib_words = tr.ib_words; A.item_bits = reinterpret_cast<const unsigned long long*>(tr.item_bits.p); A.item_rank = tr.item_rank.p;
#define KN_LAUNCH_KNN(CAPV, WV)                                                                        \
    k_predict_knn<CAPV, WV><<<(unsigned)ceil_div(n, WV), WV * 64, 0, st>>>(                              \
        A, n, d_du, d_di, d_ratings, d_order, d_pred, d_abs_err, d_owned, unknown_users_owned ? 1 : 0)
        if (nt->kcap <= 64) KN_LAUNCH_KNN(64, 4);
        else if (nt->kcap <= 128) KN_LAUNCH_KNN(128, 4);
        else if (nt->kcap <= 256) KN_LAUNCH_KNN(256, 4);
        else if (nt->kcap <= 512) KN_LAUNCH_KNN(512, 4);
        else if (nt->kcap <= 1024) KN_LAUNCH_KNN(1024, 2);
        else if (nt->kcap <= 2048) KN_LAUNCH_KNN(2048, 1);
        else throw Error(KNNCF_E_UNSUPPORTED, "predict: k > 2048 needs the tiled prediction kernel (not built yet)");
#undef KN_LAUNCH_KNN
    } else {
        k_predict_simple<<<(unsigned)ceil_div(n, TPB), TPB, 0, st>>>(A, predictor, n, d_du, d_di, d_ratings, d_pred,
                                                                     d_abs_err, d_owned, unknown_users_owned ? 1 : 0);
    }
    KN_HIP(hipGetLastError());
}

// fixed-shape reduction: block b sums elements b, b + B, ... (each thread a strided slice, then a
// fixed LDS tree), so the result does not depend on scheduling
__global__ void __launch_bounds__(TPB) k_reduce_err(const double* __restrict__ abs_err, const uint8_t* __restrict__ owned,
                                                    int64_t n, double* __restrict__ partial, int64_t* __restrict__ counts) {
    __shared__ double rs[TPB];
    __shared__ long long rc[TPB];
    double s = 0.0;
    long long c = 0;
    for (int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x; t < n; t += (int64_t)gridDim.x * TPB) {
        s += abs_err[t];
        c += owned[t];
    }
    rs[threadIdx.x] = s;
    rc[threadIdx.x] = c;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            rs[threadIdx.x] += rs[threadIdx.x + o];
            rc[threadIdx.x] += rc[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = rs[0];
        counts[blockIdx.x] = rc[0];
    }
}

void launch_reduce_err(const double* d_abs_err, const uint8_t* d_owned, int64_t n, double* d_partials,
                       int64_t* d_counts, int32_t n_blocks, hipStream_t st) {
    k_reduce_err<<<n_blocks, TPB, 0, st>>>(d_abs_err, d_owned, n, d_partials, d_counts);
    KN_HIP(hipGetLastError());
}

}  // namespace knncf
