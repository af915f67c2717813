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
#include <stdio.h>
#include <string.h>

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
    uint32_t bits_bytes;  // extent of item_bits (the grouped kernel addresses it with 32-bit byte offsets)
    uint32_t n_bytes4;    // 4 * number of training ratings
};

// predictor :568-585 given the user's mean and the weighted-sum deviation
__device__ __forceinline__ double combine(double ua, double wsd) { return ua + wsd * scale_fn(ua + wsd, ua); }

__device__ __forceinline__ void wave_sync() {
    // lanes of one wave exchange data through LDS: order the accesses for the compiler (the LDS
    // queue itself is in order per wave)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// a pointer whose value is the same in every lane, moved to scalar registers (buffer descriptors must be uniform)
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}


template <int CAP, int WAVES>  // per-wave match capacity (power of two >= kcap), waves per block
__global__ void __launch_bounds__(WAVES * 64) k_predict_knn(PredArgs A, int64_t n, const int32_t* __restrict__ du,
                                                     const int32_t* __restrict__ di, const double* __restrict__ ratings,
                                                     const uint32_t* __restrict__ order, double* __restrict__ pred,
                                                     double* __restrict__ abs_err, uint8_t* __restrict__ owned,
                                                     int unknown_owned) {
    __shared__ __attribute__((aligned(16))) uint32_t m_t[WAVES][CAP];
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
        const int32_t cnt = __builtin_amdgcn_readfirstlane(A.nbr_cnt[u]);  // (u is the same in every lane)
        const int64_t base = (int64_t)u * A.kcap;
        int32_t total = 0;
        // "which of u's neighbours rated item i": u's neighbour ids (sorted ascending) are looked up in the
        // item's rater list (sorted the same way).  All lanes search the SAME array and neighbouring lanes
        // look for neighbouring ids, so the searches share their cache sectors (the per-neighbour-row
        // searches they replace touched ~7 private sectors each and were bound by L2 sector bandwidth).
        const int64_t rb = A.i_ptr[i], re = A.i_ptr[i + 1];
        if (A.ib_words > 0) {
            // "Which of u's neighbours rated item i, and where is that rating": item i's rater bitmap (U bits) + rank
            // prefixes answer it with one 8-byte + one 4-byte read per neighbour.  A prediction is a chain of three
            // dependent gathers (neighbour ids -> bitmap words -> the matched ratings); with one wave per prediction
            // the kernel is bound by that latency, so each level is issued for ALL neighbours at once, through buffer
            // descriptors rooted at the wave's own rows: the loads are unconditional (a lane without a neighbour, or
            // without a match, uses an out-of-range offset and gets 0) and the compiler can count them instead of
            // waiting for everything at every use.
            constexpr int TR = CAP / 64;
            const int32_t rowlen = __builtin_amdgcn_readfirstlane((int32_t)(re - rb));
            const int32_t ibw = __builtin_amdgcn_readfirstlane((int32_t)A.ib_words);
            const auto r_uidx = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(uniform_ptr(A.nbr_uidx + base)), 0, (uint32_t)cnt * 4u, 0x00020000);
            const auto r_usim = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(uniform_ptr(A.nbr_usim + base)), 0, (uint32_t)cnt * 8u, 0x00020000);
            const auto r_bits = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(uniform_ptr(A.item_bits + (int64_t)i * A.ib_words)), 0, (uint32_t)ibw * 8u, 0x00020000);
            const auto r_rank = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(uniform_ptr(A.item_rank + (int64_t)i * A.ib_words)), 0, (uint32_t)ibw * 4u, 0x00020000);
            const auto r_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(uniform_ptr(A.it_t + rb)), 0, (uint32_t)rowlen * 4u, 0x00020000);
            const auto r_dev = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(uniform_ptr(A.it_dev + rb)), 0, (uint32_t)rowlen * 8u, 0x00020000);
            typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
            uint32_t x[TR], rk[TR], mtv[TR];
            u32x2 sv[TR], wd[TR], dv[TR];
#pragma unroll
            for (int k = 0; k < TR; ++k) {  // level 1: neighbour ids and similarities (offsets past cnt are out of range)
                const uint32_t j = 64u * k + lane;
                x[k] = __builtin_amdgcn_raw_buffer_load_b32(r_uidx, (int)(j * 4u), 0, 0);
                sv[k] = __builtin_amdgcn_raw_buffer_load_b64(r_usim, (int)(j * 8u), 0, 0);
            }
#pragma unroll
            for (int k = 0; k < TR; ++k) {  // level 2: the bitmap word of each neighbour and the raters before that word
                const uint32_t wi = (64u * k + lane < (uint32_t)cnt) ? (x[k] >> 6) : 0x0fffffffu;
                wd[k] = __builtin_amdgcn_raw_buffer_load_b64(r_bits, (int)(wi * 8u), 0, 0);
                rk[k] = __builtin_amdgcn_raw_buffer_load_b32(r_rank, (int)(wi * 4u), 0, 0);
            }
            bool fnd[TR];
#pragma unroll
            for (int k = 0; k < TR; ++k) {  // level 3: the matched ratings (file row, deviation)
                const unsigned long long word = ((unsigned long long)wd[k].y << 32) | wd[k].x;  // (absent neighbour: 0)
                fnd[k] = (word >> (x[k] & 63u)) & 1ull;
                const uint32_t q = fnd[k] ? rk[k] + (uint32_t)__popcll(word & ((1ull << (x[k] & 63u)) - 1ull)) : 0x0fffffffu;
                mtv[k] = __builtin_amdgcn_raw_buffer_load_b32(r_t, (int)(q * 4u), 0, 0);
                dv[k] = __builtin_amdgcn_raw_buffer_load_b64(r_dev, (int)(q * 8u), 0, 0);
            }
#pragma unroll
            for (int k = 0; k < TR; ++k) {
                const unsigned long long hit = __ballot(fnd[k]);
                if (fnd[k]) {
                    const int32_t slot = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(hit >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hit, 0u));
                    mt[slot] = mtv[k];
                    md[slot] = __hiloint2double((int)dv[k].y, (int)dv[k].x);
                    ms[slot] = __hiloint2double((int)sv[k].y, (int)sv[k].x);
                }
                total += __popcll(hit);
            }
        } else {
            // bitmaps not built (they would not fit): u's neighbour ids (sorted ascending) are looked up in the item's
            // rater list (sorted the same way) by binary search
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
                while (__any(lo < hi)) {
                    if (lo < hi) {
                        const int64_t mid = (lo + hi) >> 1;
                        if (A.it_user[mid] < x) lo = mid + 1;
                        else hi = mid;
                    }
                }
                const bool found = j < cnt && lo < re && A.it_user[lo] == x;
                const unsigned long long hit = __ballot(found);
                if (found) {
                    const int32_t slot = total + __popcll(hit & ((1ull << lane) - 1ull));
                    mt[slot] = A.it_t[lo];
                    md[slot] = A.it_dev[lo];
                    ms[slot] = s;
                }
                total += __popcll(hit);
            }
        }
        // order the matches by training file row (the order of ratedI(i) :508-517): rank by counting.  Lane l owns
        // the matches in slots l, l + 64, ...; every lane streams all keys (LDS broadcast reads, 4 keys per read) and
        // counts the smaller ones; then each match moves to its rank.  No dependent LDS round trips, unlike a sorting
        // network's 20-40 compare-exchange stages.
        {
            constexpr int TRS = CAP / 64;
            for (int32_t c = total + lane; c < ((total + 3) & ~3); c += 64) mt[c] = 0xffffffffu;  // pad to a multiple of 4
            wave_sync();
            uint32_t my_t[TRS], rank[TRS];
            double my_d[TRS], my_s[TRS];
#pragma unroll
            for (int k = 0; k < TRS; ++k) {
                const int32_t slot = 64 * k + lane;
                rank[k] = 0;
                my_t[k] = 0;  // (a slot past `total` counts nothing and is not written back)
                my_d[k] = 0.0;
                my_s[k] = 0.0;
                if (slot < total) {
                    my_t[k] = mt[slot];
                    my_d[k] = md[slot];
                    my_s[k] = ms[slot];
                }
            }
            const int nslot = (total + 63) >> 6;  // wave-uniform
            const uint4* keys4 = reinterpret_cast<const uint4*>(mt);
            for (int32_t c = 0; c < ((total + 3) >> 2); ++c) {
                const uint4 kq = keys4[c];
#pragma unroll
                for (int k = 0; k < TRS; ++k) {
                    if (k < nslot) rank[k] += (uint32_t)(kq.x < my_t[k]) + (uint32_t)(kq.y < my_t[k]) + (uint32_t)(kq.z < my_t[k]) + (uint32_t)(kq.w < my_t[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < TRS; ++k) {
                if (64 * k + lane < total) {
                    md[rank[k]] = my_d[k];
                    ms[rank[k]] = my_s[k];
                }
            }
            wave_sync();
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

// ---- the grouped kNN prediction kernel ------------------------------------------------------------------------------
// A prediction is a chain of dependent gathers (row -> user, item -> neighbour ids -> bitmap words -> matched ratings)
// with very little arithmetic behind it (~14 matches on ml-25m shape), so one-wave-per-prediction is bound by six
// memory latencies per wave.  Here a wave takes CHUNK consecutive test rows of a list SORTED BY USER:
//   * the rows' own data (user, item, means, extents) are fetched by one lane per row — three latencies per chunk;
//   * the user's neighbour list stays in registers across all his rows;
//   * G rows are in flight together, and each gather level is issued for all their neighbours at once, through buffer
//     descriptors: unconditional, counted loads (a lane without a neighbour or a match reads out of range and gets 0).
// Matches are ordered by training file row with a counting rank and folded left in fp64 exactly as in k_predict_knn.
template <int TR, int G, int WAVES>  // TR = 64-neighbour trips covering kcap; G rows in flight; waves per block
__global__ void __launch_bounds__(WAVES * 64) k_predict_knn_rows(PredArgs A, int64_t n, const int32_t* __restrict__ du,
                                                             const int32_t* __restrict__ di, const double* __restrict__ ratings,
                                                             const uint32_t* __restrict__ order, double* __restrict__ pred,
                                                             double* __restrict__ abs_err, uint8_t* __restrict__ owned,
                                                             int unknown_owned) {
    constexpr int CAP = TR * 64;
    constexpr int CHUNK = 32;
    __shared__ __attribute__((aligned(16))) uint32_t m_t[WAVES][CAP];
    __shared__ double m_dev[WAVES][CAP];
    __shared__ double m_sim[WAVES][CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = ((int64_t)blockIdx.x * WAVES + wave) * CHUNK;
    if (r0 >= n) return;
    const int nr = (int)min((int64_t)CHUNK, n - r0);
    uint32_t* mt = m_t[wave];
    double* md = m_dev[wave];
    double* ms = m_sim[wave];
    // lane l holds row r0 + l
    int64_t my_row = 0;
    int32_t my_u = -1, my_i = -1, my_cnt = 0;
    uint32_t my_rb = 0, my_len = 0;
    double my_ua = -1.0, my_p = 0.0;
    bool my_mine = false;
    if (lane < nr) {
        my_row = order[r0 + lane];
        my_u = du[my_row];
        my_i = di[my_row];
        my_mine = (my_u < 0) ? (unknown_owned != 0) : (my_u >= A.own_lo && my_u < A.own_hi);
        if (my_u >= 0) {
            my_ua = A.user_avg[my_u];  // usersAvgValue.getOrElse(u, -1.0) :572
            my_cnt = A.nbr_cnt[my_u];
        }
        if (my_i >= 0) {
            my_rb = (uint32_t)A.i_ptr[my_i];
            my_len = (uint32_t)A.i_ptr[my_i + 1] - my_rb;
        }
        my_p = (my_ua < 0.0) ? A.global_avg : combine(my_ua, 0.0);  // unknown user / no rater: den = 0 -> 0.0 :527-529
    }
    const bool my_active = lane < nr && my_mine && my_ua >= 0.0 && my_i >= 0 && my_cnt > 0 && my_len > 0;
    const unsigned long long active = __ballot(my_active);
    const auto r_bits = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(A.item_bits), 0, A.bits_bytes, 0x00020000);
    const auto r_rank = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(A.item_rank), 0, A.bits_bytes / 2, 0x00020000);
    const auto r_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(A.it_t), 0, A.n_bytes4, 0x00020000);
    const auto r_dev = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A.it_dev), 0, A.n_bytes4 * 2, 0x00020000);
    const uint32_t ibw = (uint32_t)A.ib_words;
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    uint32_t x[TR];
    u32x2 sv[TR];
    int32_t cur_u = -1, cur_cnt = 0;
    int r = 0;
    while (r < nr) {
        if (!((active >> r) & 1ull)) { ++r; continue; }  // (wave-uniform)
        const int32_t u = __builtin_amdgcn_readlane(my_u, r);
        if (u != cur_u) {  // a new user: his neighbours (ids ascending) and their similarities
            cur_u = u;
            cur_cnt = __builtin_amdgcn_readlane(my_cnt, r);
            const int64_t base = (int64_t)u * A.kcap;
            const auto r_uidx = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(A.nbr_uidx + base), 0, (uint32_t)cur_cnt * 4u, 0x00020000);
            const auto r_usim = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A.nbr_usim + base), 0, (uint32_t)cur_cnt * 8u, 0x00020000);
#pragma unroll
            for (int k = 0; k < TR; ++k) {
                const uint32_t j = 64u * k + lane;
                x[k] = __builtin_amdgcn_raw_buffer_load_b32(r_uidx, (int)(j * 4u), 0, 0);
                sv[k] = __builtin_amdgcn_raw_buffer_load_b64(r_usim, (int)(j * 8u), 0, 0);
            }
        }
        // up to G consecutive active rows of this user
        int rows[G];
        uint32_t bbase[G], rbase[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            rows[g] = -1;
            bbase[g] = 0x0fffffffu;
            rbase[g] = 0;
            if (r < nr && ((active >> r) & 1ull) && __builtin_amdgcn_readlane(my_u, min(r, 63)) == cur_u) {
                rows[g] = r;
                bbase[g] = (uint32_t)__builtin_amdgcn_readlane(my_i, r) * ibw;  // word offset of the item's bitmap row
                rbase[g] = (uint32_t)__builtin_amdgcn_readlane((int)my_rb, r);
                ++r;
            }
        }
        u32x2 wd[G][TR], dv[G][TR];
        uint32_t rk[G][TR], mtv[G][TR];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < TR; ++k) {  // the bitmap word of each neighbour and the raters before that word
                const uint32_t wi = (rows[g] >= 0 && 64u * k + lane < (uint32_t)cur_cnt) ? bbase[g] + (x[k] >> 6) : 0x0fffffffu;
                wd[g][k] = __builtin_amdgcn_raw_buffer_load_b64(r_bits, (int)(wi * 8u), 0, 0);
                rk[g][k] = __builtin_amdgcn_raw_buffer_load_b32(r_rank, (int)(wi * 4u), 0, 0);
            }
        unsigned long long fmask[G][TR];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < TR; ++k) {  // the matched ratings (file row, deviation)
                const unsigned long long word = ((unsigned long long)wd[g][k].y << 32) | wd[g][k].x;  // (absent: 0)
                const bool f = (word >> (x[k] & 63u)) & 1ull;
                fmask[g][k] = __ballot(f);
                const uint32_t q = f ? rbase[g] + rk[g][k] + (uint32_t)__popcll(word & ((1ull << (x[k] & 63u)) - 1ull)) : 0x0fffffffu;
                mtv[g][k] = __builtin_amdgcn_raw_buffer_load_b32(r_t, (int)(q * 4u), 0, 0);
                dv[g][k] = __builtin_amdgcn_raw_buffer_load_b64(r_dev, (int)(q * 8u), 0, 0);
            }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (rows[g] < 0) continue;  // (wave-uniform)
            int32_t total = 0;
#pragma unroll
            for (int k = 0; k < TR; ++k) {
                const unsigned long long hit = fmask[g][k];
                if ((hit >> lane) & 1ull) {
                    const int32_t slot = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(hit >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hit, 0u));
                    mt[slot] = mtv[g][k];
                    md[slot] = __hiloint2double((int)dv[g][k].y, (int)dv[g][k].x);
                    ms[slot] = __hiloint2double((int)sv[k].y, (int)sv[k].x);
                }
                total += __popcll(hit);
            }
            // order the matches by training file row (the order of ratedI(i) :508-517): rank by counting
            for (int32_t c = total + lane; c < ((total + 3) & ~3); c += 64) mt[c] = 0xffffffffu;
            wave_sync();
            uint32_t key[TR], rank[TR];
            double kd[TR], ks[TR];
#pragma unroll
            for (int k = 0; k < TR; ++k) {
                const int32_t slot = 64 * k + lane;
                rank[k] = 0; key[k] = 0; kd[k] = 0.0; ks[k] = 0.0;
                if (slot < total) { key[k] = mt[slot]; kd[k] = md[slot]; ks[k] = ms[slot]; }
            }
            const int nslot = (total + 63) >> 6;
            const uint4* keys4 = reinterpret_cast<const uint4*>(mt);
            for (int32_t c = 0; c < ((total + 3) >> 2); ++c) {
                const uint4 kq = keys4[c];
#pragma unroll
                for (int k = 0; k < TR; ++k)
                    if (k < nslot) rank[k] += (uint32_t)(kq.x < key[k]) + (uint32_t)(kq.y < key[k]) + (uint32_t)(kq.z < key[k]) + (uint32_t)(kq.w < key[k]);
            }
#pragma unroll
            for (int k = 0; k < TR; ++k)
                if (64 * k + lane < total) { md[rank[k]] = kd[k]; ms[rank[k]] = ks[k]; }
            wave_sync();
            double num = 0.0, den = 0.0;
            for (int32_t c = 0; c < total; ++c) {  // every lane folds the same sequence (LDS broadcast)
                const double sc = ms[c];
                num = num + md[c] * sc;
                den = den + fabs(sc);
            }
            wave_sync();
            const double wsd = (den > 0) ? num / den : 0.0;
            if (lane == rows[g]) my_p = combine(my_ua, wsd);
        }
    }
    if (lane < nr) {
        if (my_mine) {
            pred[my_row] = my_p;
            abs_err[my_row] = ratings ? fabs(ratings[my_row] - my_p) : 0.0;
            owned[my_row] = 1;
        } else {
            owned[my_row] = 0;
            abs_err[my_row] = 0.0;
        }
    }
}

#ifndef KNNCF_PRED_G5
#define KNNCF_PRED_G5 2  // rows in flight per wave of k_predict_knn_items at 257 .. 320 neighbours (A/B switch)
#endif
#ifndef KNNCF_PRED_CHUNK
#define KNNCF_PRED_CHUNK 64  // test rows per workgroup of k_predict_knn_items (A/B switch; at most 256)
#endif
#ifdef KNNCF_PREDICT_PROFILE  /* in-kernel cycle counters of k_predict_knn_items' phases (thread 0 of every workgroup) */
__device__ unsigned long long g_pphase[8];
#define PPH(i) do { if (threadIdx.x == 0) { const long long now_ = clock64(); atomicAdd(&g_pphase[i], (unsigned long long)(now_ - ph_t)); ph_t = now_; } } while (0)
#else
#define PPH(i) do {} while (0)
#endif
// ---- the item-grouped kNN prediction kernel -------------------------------------------------------------------------
// Probing a rater bitmap in global memory costs one cache sector per neighbour: k = 300 probes touch nearly every line
// of the item's 20 KB bitmap row, ~38 KB of L2 -> L1 traffic per prediction, and the kernels above are bound by exactly
// that (19 TB/s of sector traffic at ml-25m shape).  Here the test rows are SORTED BY ITEM and a workgroup keeps the
// current item's bitmap + rank prefixes in LDS (30 KB): it is fetched once per run of rows of that item, the probes
// become LDS reads, and what remains in global memory is the coalesced stream of each row's neighbour ids and the
// gathers of the ~14 matched ratings and similarities.
template <int TR, int G, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_predict_knn_items(PredArgs A, int64_t n, const int32_t* __restrict__ du,
                                                              const int32_t* __restrict__ di, const double* __restrict__ ratings,
                                                              const uint32_t* __restrict__ order, double* __restrict__ pred,
                                                              double* __restrict__ abs_err, uint8_t* __restrict__ owned,
                                                              int unknown_owned) {
    constexpr int CAP = TR * 64;  // matches of a row: up to kcap
    // matches whose (deviation, similarity) sit in LDS at a time: a row's matches are ordered and folded in windows of MCAP
    // (~14 matches on the ml-25m shape; more than 64 for the rows of the few most-rated items only).  The buffers of a full
    // CAP per wave (26 KB per workgroup at k = 300) held the kernel at three workgroups per CU; with 9 KB it runs four, and
    // it waits on gather latency most of the time: 7.25 -> 6.3 ms
    constexpr int MCAP = 64;
    constexpr int CHUNK = KNNCF_PRED_CHUNK;  // rows per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ibw = (int)A.ib_words;
    const int ibw2 = (ibw + 1) >> 1;  // pairs of bitmap words
    // the rank prefixes are kept for every SECOND word only (the odd word adds the popcount of its even neighbour, which
    // comes with the same 16-byte LDS read): 25 instead of 30 KB, which is what lets a third workgroup onto the CU
    unsigned long long* bits = reinterpret_cast<unsigned long long*>(smem);              // [2 * ibw2]
    uint32_t* rnk = reinterpret_cast<uint32_t*>(bits + 2 * ibw2);                         // [ibw2] raters before word 2 j
    double* m_dev = reinterpret_cast<double*>(rnk + ((ibw2 + 1) & ~1));                   // [WAVES][MCAP]
    double* m_sim = m_dev + WAVES * MCAP;                                                 // [WAVES][MCAP]
    uint32_t* m_t = reinterpret_cast<uint32_t*>(m_sim + WAVES * MCAP);                    // [WAVES][CAP] file rows of the matches
    __shared__ int64_t s_row[CHUNK];
    __shared__ int32_t s_u[CHUNK], s_i[CHUNK], s_cnt[CHUNK];
    __shared__ uint32_t s_rb[CHUNK];
    __shared__ double s_ua[CHUNK], s_p[CHUNK];
    __shared__ uint8_t s_mine[CHUNK], s_active[CHUNK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * CHUNK;
    if (r0 >= n) return;
    const int nr = (int)min((int64_t)CHUNK, n - r0);
#ifdef KNNCF_PREDICT_PROFILE
    long long ph_t = clock64();
#endif
    uint32_t* mt = m_t + wave * CAP;
    double* md = m_dev + wave * MCAP;
    double* ms = m_sim + wave * MCAP;
    if ((int)threadIdx.x < nr) {
        const int l = threadIdx.x;
        const int64_t row = order[r0 + l];
        const int32_t u = du[row], i = di[row];
        const bool mine = (u < 0) ? (unknown_owned != 0) : (u >= A.own_lo && u < A.own_hi);
        double ua = -1.0;  // usersAvgValue.getOrElse(u, -1.0) :572
        int32_t cnt = 0;
        uint32_t rb = 0, len = 0;
        if (u >= 0) { ua = A.user_avg[u]; cnt = A.nbr_cnt[u]; }
        if (i >= 0) { rb = (uint32_t)A.i_ptr[i]; len = (uint32_t)A.i_ptr[i + 1] - rb; }
        s_row[l] = row; s_u[l] = u; s_i[l] = i; s_cnt[l] = cnt; s_rb[l] = rb; s_ua[l] = ua;
        s_p[l] = (ua < 0.0) ? A.global_avg : combine(ua, 0.0);  // unknown user / no rater: den = 0 -> 0.0 :527-529
        s_mine[l] = mine ? 1 : 0;
        s_active[l] = (mine && ua >= 0.0 && i >= 0 && cnt > 0 && len > 0) ? 1 : 0;
    }
    __syncthreads();
    PPH(0);  // the rows' own data
    const auto r_t = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(A.it_t), 0, A.n_bytes4, 0x00020000);
    const auto r_dev = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(A.it_dev), 0, A.n_bytes4 * 2, 0x00020000);
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    int ra = 0;
    while (ra < nr) {  // runs of rows of one item (every thread walks the same LDS values: block-uniform)
        const int32_t item = s_i[ra];
        int rbn = ra + 1;
        bool any = s_active[ra] != 0;
        while (rbn < nr && s_i[rbn] == item) { any = any || s_active[rbn] != 0; ++rbn; }
        if (any) {
            const unsigned long long* gb = A.item_bits + (int64_t)item * ibw;
            const uint32_t* gr = A.item_rank + (int64_t)item * ibw;
            // every load of a batch is requested before the first is used (twelve bitmap words and six rank words per thread
            // and batch: ONE batch up to 196 608 users).  One word per trip of a plain loop paid ten dependent global
            // latencies per run — a third of the kernel's time by its phase counters (scripts/phase_profile.sh)
            constexpr int NT = WAVES * 64, NB = 12;
            for (int w0 = 0; w0 < 2 * ibw2; w0 += NB * NT) {
                unsigned long long bv[NB];
                uint32_t rv[NB / 2];
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const int w = w0 + q * NT + (int)threadIdx.x;
                    bv[q] = w < ibw ? gb[w] : 0ull;
                }
#pragma unroll
                for (int q = 0; q < NB / 2; ++q) {
                    const int j = (w0 >> 1) + q * NT + (int)threadIdx.x;
                    rv[q] = 2 * j < ibw ? gr[2 * j] : 0u;
                }
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const int w = w0 + q * NT + (int)threadIdx.x;
                    if (w < 2 * ibw2) bits[w] = bv[q];
                }
#pragma unroll
                for (int q = 0; q < NB / 2; ++q) {
                    const int j = (w0 >> 1) + q * NT + (int)threadIdx.x;
                    if (j < ibw2) rnk[j] = rv[q];
                }
            }
            __syncthreads();
            PPH(1);  // the item's bitmap + ranks into LDS
            for (int rg = ra + wave * G; rg < rbn; rg += WAVES * G) {  // G rows of the run per wave and trip
                int rows[G];
                int32_t cnts[G];
                uint32_t rbase[G];
                uint32_t x[G][TR];
                u32x2 sv[G][TR];
                __amdgpu_buffer_rsrc_t r_usim[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int rr = rg + g;
                    const bool on = rr < rbn && s_active[min(rr, CHUNK - 1)] != 0;
                    rows[g] = on ? rr : -1;
                    const int32_t u = __builtin_amdgcn_readfirstlane(on ? s_u[rr] : 0);
                    cnts[g] = __builtin_amdgcn_readfirstlane(on ? s_cnt[rr] : 0);
                    rbase[g] = (uint32_t)__builtin_amdgcn_readfirstlane(on ? (int)s_rb[rr] : 0);
                    const int64_t base = (int64_t)u * A.kcap;
                    const auto r_uidx = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(uniform_ptr(A.nbr_uidx + base)), 0, (uint32_t)cnts[g] * 4u, 0x00020000);
                    r_usim[g] = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(uniform_ptr(A.nbr_usim + base)), 0, (uint32_t)cnts[g] * 8u, 0x00020000);
#pragma unroll
                    for (int k = 0; k < TR; ++k) {  // the row's neighbours (ids ascending); past cnt: 0
                        const uint32_t j = 64u * k + lane;
                        x[g][k] = __builtin_amdgcn_raw_buffer_load_b32(r_uidx, (int)(j * 4u), 0, 0);
                    }
                }
#ifdef KNNCF_PREDICT_PROFILE
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PPH(2);  // the rows' neighbour ids have arrived
#endif
                u32x2 dv[G][TR];
                uint32_t mtv[G][TR];
                unsigned long long fmask[G][TR];
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int k = 0; k < TR; ++k) {  // probe the LDS bitmap, gather the matched ratings (file row, deviation)
                        const bool have = 64 * k + lane < cnts[g];
                        const uint32_t xi = have ? x[g][k] : 0u;
                        const uint32_t wi = xi >> 6;
                        const ulonglong2 pair = *reinterpret_cast<const ulonglong2*>(bits + (wi & ~1u));
                        const unsigned long long word = (wi & 1u) ? pair.y : pair.x;
                        const bool f = have && ((word >> (xi & 63u)) & 1ull);
                        fmask[g][k] = __ballot(f);
                        const uint32_t before = rnk[wi >> 1] + ((wi & 1u) ? (uint32_t)__popcll(pair.x) : 0u);
                        const uint32_t q = f ? rbase[g] + before + (uint32_t)__popcll(word & ((1ull << (xi & 63u)) - 1ull)) : 0x0fffffffu;
                        mtv[g][k] = __builtin_amdgcn_raw_buffer_load_b32(r_t, (int)(q * 4u), 0, 0);
                        dv[g][k] = __builtin_amdgcn_raw_buffer_load_b64(r_dev, (int)(q * 8u), 0, 0);
                        // the similarity is needed for the ~5 % of neighbours that rated the item only: it is gathered
                        // with the ratings instead of streamed with the ids (a third of the list bytes instead of all)
                        sv[g][k] = __builtin_amdgcn_raw_buffer_load_b64(r_usim[g], f ? (int)((64u * k + lane) * 8u) : -1, 0, 0);
                    }
#ifdef KNNCF_PREDICT_PROFILE
                PPH(3);  // probes issued
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PPH(4);  // the matched ratings have arrived
#endif
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (rows[g] < 0) continue;  // (wave-uniform)
                    int32_t total = 0;
#pragma unroll
                    for (int k = 0; k < TR; ++k) total += __popcll(fmask[g][k]);
                    if (total == 0) continue;  // no neighbour rated the item: the prediction is the user's mean (preset)
                    // order the matches by training file row (the order of ratedI(i) :508-517) — rank by counting — and fold
                    double num = 0.0, den = 0.0;
                    if (total <= MCAP) {
                        // the common case (~14 matches): one match per lane, the keys travel by v_readlane
                        int32_t seen = 0;
#pragma unroll
                        for (int k = 0; k < TR; ++k) {
                            const unsigned long long hit = fmask[g][k];
                            if ((hit >> lane) & 1ull) {
                                const int32_t slot = seen + __builtin_amdgcn_mbcnt_hi((uint32_t)(hit >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hit, 0u));
                                mt[slot] = mtv[g][k];
                                md[slot] = __hiloint2double((int)dv[g][k].y, (int)dv[g][k].x);
                                ms[slot] = __hiloint2double((int)sv[g][k].y, (int)sv[g][k].x);
                            }
                            seen += __popcll(hit);
                        }
                        wave_sync();
                        uint32_t key = 0xffffffffu, rk = 0;
                        double kd = 0.0, ks = 0.0;
                        if (lane < total) { key = mt[lane]; kd = md[lane]; ks = ms[lane]; }
                        for (int32_t c = 0; c < total; ++c) rk += (uint32_t)((uint32_t)__builtin_amdgcn_readlane((int)key, c) < key);
                        if (lane < total) { md[rk] = kd; ms[rk] = ks; }
                        wave_sync();
                        for (int32_t c = 0; c < total; ++c) {  // every lane folds the same sequence (LDS broadcast)
                            const double sc = ms[c];
                            num = num + md[c] * sc;
                            den = den + fabs(sc);
                        }
                        wave_sync();
                    } else {
                        // many matches (rows of the most-rated items): the file rows of all of them in LDS, every match
                        // ranked where it sits (its trip's registers), then windows of MCAP ranks through the buffers
                        int32_t seen = 0;
#pragma unroll
                        for (int k = 0; k < TR; ++k) {
                            const unsigned long long hit = fmask[g][k];
                            if ((hit >> lane) & 1ull)
                                mt[seen + __builtin_amdgcn_mbcnt_hi((uint32_t)(hit >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hit, 0u))] = mtv[g][k];
                            seen += __popcll(hit);
                        }
                        for (int32_t c = total + lane; c < ((total + 3) & ~3); c += 64) mt[c] = 0xffffffffu;
                        wave_sync();
                        uint32_t rank[TR];
#pragma unroll
                        for (int k = 0; k < TR; ++k) rank[k] = 0;
                        const uint4* keys4 = reinterpret_cast<const uint4*>(mt);
                        for (int32_t c = 0; c < ((total + 3) >> 2); ++c) {
                            const uint4 kq = keys4[c];
#pragma unroll
                            for (int k = 0; k < TR; ++k) {
                                const uint32_t key = mtv[g][k];
                                rank[k] += (uint32_t)(kq.x < key) + (uint32_t)(kq.y < key) + (uint32_t)(kq.z < key) + (uint32_t)(kq.w < key);
                            }
                        }
                        for (int32_t b0 = 0; b0 < total; b0 += MCAP) {
#pragma unroll
                            for (int k = 0; k < TR; ++k) {
                                const uint32_t at = rank[k] - (uint32_t)b0;
                                if (((fmask[g][k] >> lane) & 1ull) && at < (uint32_t)MCAP) {
                                    md[at] = __hiloint2double((int)dv[g][k].y, (int)dv[g][k].x);
                                    ms[at] = __hiloint2double((int)sv[g][k].y, (int)sv[g][k].x);
                                }
                            }
                            wave_sync();
                            const int32_t nw = min(MCAP, total - b0);
                            for (int32_t c = 0; c < nw; ++c) {
                                const double sc = ms[c];
                                num = num + md[c] * sc;
                                den = den + fabs(sc);
                            }
                            wave_sync();
                        }
                    }
                    const double wsd = (den > 0) ? num / den : 0.0;
                    if (lane == 0) s_p[rows[g]] = combine(s_ua[rows[g]], wsd);
                }
                PPH(5);  // order by file row + fold
            }
            __syncthreads();  // the bitmap is overwritten by the next run
            PPH(6);  // wait for the other waves
        }
        ra = rbn;
    }
    __syncthreads();
    if ((int)threadIdx.x < nr) {
        const int l = threadIdx.x;
        const int64_t row = s_row[l];
        if (s_mine[l]) {
            const double p = s_p[l];
            pred[row] = p;
            abs_err[row] = ratings ? fabs(ratings[row] - p) : 0.0;
            owned[row] = 1;
        } else {
            owned[row] = 0;
            abs_err[row] = 0.0;
        }
    }
    PPH(7);  // output
}

#ifdef KNNCF_PREDICT_PROFILE
static void predict_profile_dump() {
    unsigned long long h[8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pphase), sizeof(h));
    unsigned long long tot = 0;
    for (int i = 0; i < 8; ++i) tot += h[i];
    static const char* names[8] = {"row data", "bitmap", "ids wait", "probe issue", "gather wait", "rank+fold", "barrier", "output"};
    fprintf(stderr, "[predict profile] total %.3e cycles:", (double)tot);
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * (double)h[i] / (double)tot);
    fprintf(stderr, "\n");
    memset(h, 0, sizeof(h));
    hipMemcpyToSymbol(HIP_SYMBOL(g_pphase), h, sizeof(h));
}
#endif

// sort key of a test row: its dense user / item (unknown ones last)
__global__ void k_user_keys(int64_t n, const int32_t* __restrict__ du, uint32_t limit, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    key[t] = du[t] < 0 ? (uint64_t)limit : (uint64_t)(uint32_t)du[t];
    val[t] = (uint32_t)t;
}

// the same for a shard: rows of other shards' users sort behind everything (key limit + 1) and are counted out, so that the
// prediction kernel only ever sees this shard's rows (the test set is replicated on every rank, the work is not)
__global__ void k_owned_keys(int64_t n, const int32_t* __restrict__ src, const int32_t* __restrict__ du, int32_t own_lo, int32_t own_hi,
                             int unknown_owned, uint32_t limit, uint64_t* __restrict__ key, uint32_t* __restrict__ val, unsigned long long* __restrict__ n_owned) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool mine = false;
    if (t < n) {
        const int32_t u = du[t];
        mine = (u < 0) ? (unknown_owned != 0) : (u >= own_lo && u < own_hi);
        key[t] = !mine ? (uint64_t)limit + 1ull : (src[t] < 0 ? (uint64_t)limit : (uint64_t)(uint32_t)src[t]);
        val[t] = (uint32_t)t;
    }
    // one atomic per BLOCK: 5 M rows / 64 same-address atomics serialise in the L2 (0.34 ms at the ml-25m shape)
    __shared__ unsigned int block_owned;
    if (threadIdx.x == 0) block_owned = 0;
    __syncthreads();
    const unsigned long long b = __ballot(mine);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&block_owned, (unsigned int)__popcll(b));
    __syncthreads();
    if (threadIdx.x == 0 && block_owned) atomicAdd(n_owned, (unsigned long long)block_owned);
}

void launch_owned_keys(int64_t n, const int32_t* d_src, const int32_t* d_du, int32_t own_lo, int32_t own_hi, bool unknown_owned, uint32_t limit,
                       uint64_t* d_key, uint32_t* d_val, unsigned long long* d_n_owned, hipStream_t st) {
    k_owned_keys<<<(unsigned)ceil_div(n, TPB), TPB, 0, st>>>(n, d_src, d_du, own_lo, own_hi, unknown_owned ? 1 : 0, limit, d_key, d_val, d_n_owned);
    KN_HIP(hipGetLastError());
}

void launch_user_keys(int64_t n, const int32_t* d_du, uint32_t limit, uint64_t* d_key, uint32_t* d_val, hipStream_t st) {
    k_user_keys<<<(unsigned)ceil_div(n, TPB), TPB, 0, st>>>(n, d_du, limit, d_key, d_val);
    KN_HIP(hipGetLastError());
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

void launch_predict(const Train& tr, NeighborTable* nt, int predictor, int64_t n, const int32_t* d_du,
                    const int32_t* d_di, const double* d_ratings, const uint32_t* d_order, bool order_by_item,
                    double* d_pred, double* d_abs_err, uint8_t* d_owned, bool unknown_users_owned, hipStream_t st) {
    if (n <= 0) return;
    PredArgs A{};
    A.u_ptr = tr.u_ptr.p; A.s_col = tr.s_col.p; A.s_t = tr.s_t.p; A.s_dev = tr.s_dev.p;
    A.user_avg = tr.user_avg.p; A.item_avg = tr.item_avg.p; A.item_dev_hash = tr.item_dev_hash.p;
    A.item_dev_file = tr.item_dev_file.p; A.global_avg = tr.global_avg;
    A.own_lo = tr.own_lo; A.own_hi = tr.own_hi;
    if (predictor == KNNCF_PRED_KNN) {
        KN_REQUIRE(nt != nullptr, KNNCF_E_STATE, "predict: neighbour table missing");
        A.nbr_cnt = nt->cnt.p; A.kcap = nt->kcap;
        // The item-grouped kernel probes an LDS bitmap and orders its matches by file row: the order of a list does not
        // matter to it, and it streams the lists as the re-rank left them (reference order).  The kernels that probe in global
        // memory want neighbouring lanes on neighbouring ids (cache sectors): they take the id-sorted copies, made here on
        // first need (0.7 ms per step at the ml-25m shape when it was part of every build).
        const bool items_path = d_order && order_by_item && tr.ib_words > 0 && tr.ib_words * 12 <= 48 * 1024 && nt->kcap <= 512;
        if (items_path && nt->idx.p != nullptr) {
            A.nbr_uidx = nt->idx.p; A.nbr_usim = nt->sim.p;
        } else {
            if (!nt->by_id_valid) {
                launch_sort_neighbors(*nt, (int32_t)nt->cnt.n, nullptr, st);
                nt->by_id_valid = true;
            }
            A.nbr_uidx = nt->uidx.p; A.nbr_usim = nt->usim.p;
        }
        A.i_ptr = tr.i_ptr.p; A.it_user = tr.it_user.p; A.it_dev = tr.it_dev.p; A.it_t = tr.it_t.p;
        A.ib_words = tr.ib_words; A.item_bits = reinterpret_cast<const unsigned long long*>(tr.item_bits.p); A.item_rank = tr.item_rank.p;
        const double bits_total = (double)tr.I * (double)tr.ib_words * 8.0;
        if (items_path) {
            A.n_bytes4 = (uint32_t)(tr.n * 4);
            const int trips = (nt->kcap + 63) / 64;
            const unsigned blocks = (unsigned)ceil_div(n, KNNCF_PRED_CHUNK);
#define KN_LAUNCH_ITEMS(TRV, GV)                                                                                          \
    do {                                                                                                                  \
        const size_t ibw2 = (size_t)(tr.ib_words + 1) / 2;                                                                  \
        const size_t smem = ibw2 * 16 + ((ibw2 + 1) & ~(size_t)1) * 4 + (size_t)4 * (64 * 16 + (TRV * 64) * 4);               \
        KN_HIP(hipFuncSetAttribute((const void*)k_predict_knn_items<TRV, GV, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        k_predict_knn_items<TRV, GV, 4><<<blocks, 256, smem, st>>>(A, n, d_du, d_di, d_ratings, d_order, d_pred, d_abs_err,  \
                                                                  d_owned, unknown_users_owned ? 1 : 0);                \
    } while (0)
            if (trips <= 1) KN_LAUNCH_ITEMS(1, 4);
            else if (trips <= 2) KN_LAUNCH_ITEMS(2, 4);
            else if (trips <= 4) KN_LAUNCH_ITEMS(4, 2);
            else if (trips <= 5) KN_LAUNCH_ITEMS(5, KNNCF_PRED_G5);
            else KN_LAUNCH_ITEMS(8, 1);
#undef KN_LAUNCH_ITEMS
            KN_HIP(hipGetLastError());
#ifdef KNNCF_PREDICT_PROFILE
            KN_HIP(hipStreamSynchronize(st));
            predict_profile_dump();
#endif
            return;
        }
        if (d_order && !order_by_item && tr.ib_words > 0 && bits_total < 4.0e9 && nt->kcap <= 512) {
            A.bits_bytes = (uint32_t)bits_total;
            A.n_bytes4 = (uint32_t)(tr.n * 4);
            const int trips = (nt->kcap + 63) / 64;
            const unsigned blocks = (unsigned)ceil_div(ceil_div(n, 32), 4);
#define KN_LAUNCH_ROWS(TRV, GV)                                                                                   \
    k_predict_knn_rows<TRV, GV, 4><<<blocks, 256, 0, st>>>(A, n, d_du, d_di, d_ratings, d_order, d_pred, d_abs_err, \
                                                           d_owned, unknown_users_owned ? 1 : 0)
            if (trips <= 1) KN_LAUNCH_ROWS(1, 4);
            else if (trips <= 2) KN_LAUNCH_ROWS(2, 4);
            else if (trips <= 4) KN_LAUNCH_ROWS(4, 2);
            else if (trips <= 5) KN_LAUNCH_ROWS(5, 2);
            else KN_LAUNCH_ROWS(8, 1);
#undef KN_LAUNCH_ROWS
            KN_HIP(hipGetLastError());
            return;
        }
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
