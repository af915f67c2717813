// select.hip — K6 (top-k neighbour select) and K6b (exact fp64 re-rank).
//
// K6: the bf16 GEMM's S row is only a filter.  With |S[u][v] - s_uv| <= eps for every pair,
// every true top-k member v satisfies S[u][v] >= a_k - 2 eps, a_k = k-th largest S value of the
// row.  One workgroup per row builds a 4096-bin histogram in LDS (wave-level atomics), finds the
// bin holding a_k by a block-wide suffix scan, and compacts every v above (bin lower edge - 2 eps)
// into the row's shortlist.  HBM-bound: one streaming pass over S, the second pass is L2/MALL.
//
// K6b: exact similarities of the shortlist in fp64 IN REFERENCE ORDER (adjustedCosine...
// shared/predictions.scala:418-426: left fold over uItems.intersect(vItems) in the Set iteration
// order of the first argument; memo history N6), then a wave/LDS bitonic sort by
// (similarity desc, HashSet iteration rank asc) == sortWith(_._2 > _._2) on (allUsers - u).toSeq
// :608-610 (stable TimSort, N3), and take(k).
#include <math.h>

#include "engine.h"

namespace knncf {

static constexpr int TPB = 256;
static constexpr int NBINS = 4096;

__device__ __forceinline__ int sim_bin(float x) {
    int b = (int)floorf((x + 1.0f) * (NBINS / 2));
    return min(max(b, 0), NBINS - 1);
}

// rigorous bound on |S[u][v] - s_uv|: operand rounding (eps_base) plus fp32 accumulation — adding a
// zero product is exact, so at most 2 roundings per common item, each <= 2^-24 of a partial sum <= 1.01
__device__ __forceinline__ float row_eps(float eps_base, int64_t row_len) {
    return eps_base + (float)row_len * 2.0f * 6.1e-8f;
}

__global__ void __launch_bounds__(TPB) k_select(const float* __restrict__ S, int64_t ld, int32_t n_rows,
                                                const int32_t* __restrict__ row_user, const int64_t* __restrict__ u_ptr,
                                                int32_t U, int32_t kk, float eps_base, int32_t cap, int32_t* __restrict__ cand_idx,
                                                float* __restrict__ cand_approx, int32_t* __restrict__ cand_cnt) {
    __shared__ uint32_t hist[NBINS];
    __shared__ uint32_t part[TPB];
    __shared__ float s_thr;
    __shared__ uint32_t s_count;
    const int32_t r = blockIdx.x;
    if (r >= n_rows) return;
    const int32_t u = row_user[r];
    const float eps = row_eps(eps_base, u_ptr[u + 1] - u_ptr[u]);
    const float* row = S + (int64_t)r * ld;
    for (int b = threadIdx.x; b < NBINS; b += TPB) hist[b] = 0;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const int32_t U4 = U & ~3;
    for (int32_t v = threadIdx.x * 4; v < U4; v += TPB * 4) {
        float4 x = *reinterpret_cast<const float4*>(row + v);
        if (v + 0 != u) atomicAdd(&hist[sim_bin(x.x)], 1u);
        if (v + 1 != u) atomicAdd(&hist[sim_bin(x.y)], 1u);
        if (v + 2 != u) atomicAdd(&hist[sim_bin(x.z)], 1u);
        if (v + 3 != u) atomicAdd(&hist[sim_bin(x.w)], 1u);
    }
    for (int32_t v = U4 + threadIdx.x; v < U; v += TPB)
        if (v != u) atomicAdd(&hist[sim_bin(row[v])], 1u);
    __syncthreads();
    // suffix counts: thread t owns bins [16 t, 16 t + 16)
    constexpr int PER = NBINS / TPB;
    uint32_t mine = 0;
    for (int j = 0; j < PER; ++j) mine += hist[threadIdx.x * PER + j];
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < TPB; o <<= 1) {  // inclusive suffix scan
        uint32_t add = (threadIdx.x + o < TPB) ? part[threadIdx.x + o] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    // the thread whose bins contain the kk-th largest value
    uint32_t above = (threadIdx.x + 1 < TPB) ? part[threadIdx.x + 1] : 0;  // count in higher bins
    if (above < (uint32_t)kk && part[threadIdx.x] >= (uint32_t)kk) {
        uint32_t c = above;
        int b = threadIdx.x * PER + PER - 1;
        for (; b > threadIdx.x * PER; --b) {
            c += hist[b];
            if (c >= (uint32_t)kk) break;
        }
        // every value in bin b is >= its lower edge (up to one float rounding of x + 1)
        float edge = (float)b / (float)(NBINS / 2) - 1.0f;
        s_thr = (b == 0) ? -INFINITY : edge - 2.0f * eps - 1e-6f;
    }
    __syncthreads();
    const float thr = s_thr;
    int32_t* out_idx = cand_idx + (int64_t)r * cap;
    float* out_apx = cand_approx ? cand_approx + (int64_t)r * cap : nullptr;
    for (int32_t v0 = threadIdx.x * 4; v0 < U4; v0 += TPB * 4) {
        float4 x4 = *reinterpret_cast<const float4*>(row + v0);
        float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int32_t v = v0 + j;
            if (v != u && xs[j] >= thr) {
                uint32_t pos = atomicAdd(&s_count, 1u);
                if (pos < (uint32_t)cap) {
                    out_idx[pos] = v;
                    if (out_apx) out_apx[pos] = xs[j];
                }
            }
        }
    }
    for (int32_t v = U4 + threadIdx.x; v < U; v += TPB) {
        float x = row[v];
        if (v != u && x >= thr) {
            uint32_t pos = atomicAdd(&s_count, 1u);
            if (pos < (uint32_t)cap) {
                out_idx[pos] = v;
                if (out_apx) out_apx[pos] = x;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) cand_cnt[r] = (int32_t)s_count;
}

void launch_select(const float* S, int64_t lds, int32_t n_rows, const int32_t* d_row_user, const int64_t* d_u_ptr,
                   int32_t U, int32_t k, float eps, int32_t cap, int32_t* cand_idx, float* cand_approx,
                   int32_t* cand_cnt, hipStream_t st) {
    if (n_rows <= 0) return;
    int32_t kk = k < U - 1 ? k : U - 1;
    if (kk < 1) kk = 1;
    k_select<<<n_rows, TPB, 0, st>>>(S, lds, n_rows, d_row_user, d_u_ptr, U, kk, eps, cap, cand_idx, cand_approx, cand_cnt);
    KN_HIP(hipGetLastError());
}

// ---- exact similarities in reference order ---------------------------------------------------
struct Rows {
    const int64_t* u_ptr;
    const int32_t* s_col;
    const uint32_t* s_t;
    const double* s_pre;
};

// lower bound of `col` in s_col[lo, hi)
__device__ __forceinline__ int64_t lower_bound_col(const int32_t* __restrict__ s_col, int64_t lo, int64_t hi, int32_t col) {
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (s_col[mid] < col) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// both rows iterate in trie order (dense item index ascending): the common items are visited in
// ascending order and folded left.  Rows of similar length: two-pointer merge; very different
// lengths: walk the short row and binary-search the long one (same visiting order).
__device__ __forceinline__ double merge_dot(const Rows& R, int32_t a, int32_t b) {
    int64_t pa = R.u_ptr[a], ea = R.u_ptr[a + 1], pb = R.u_ptr[b], eb = R.u_ptr[b + 1];
    double s = 0.0;
    if (pa >= ea || pb >= eb) return s;
    if ((ea - pa) > 8 * (eb - pb) || (eb - pb) > 8 * (ea - pa)) {
        if ((ea - pa) > (eb - pb)) {  // make `a` the short row (the product is commutative, the order is not affected)
            int64_t t0 = pa; pa = pb; pb = t0;
            t0 = ea; ea = eb; eb = t0;
        }
        for (; pa < ea && pb < eb; ++pa) {
            int32_t c = R.s_col[pa];
            pb = lower_bound_col(R.s_col, pb, eb, c);
            if (pb < eb && R.s_col[pb] == c) {
                s = s + R.s_pre[pa] * R.s_pre[pb];
                ++pb;
            }
        }
        return s;
    }
    int32_t ca = R.s_col[pa], cb = R.s_col[pb];
    while (true) {
        if (ca == cb) {
            s = s + R.s_pre[pa] * R.s_pre[pb];
            ++pa; ++pb;
            if (pa >= ea || pb >= eb) break;
            ca = R.s_col[pa];
            cb = R.s_col[pb];
        } else if (ca < cb) {
            if (++pa >= ea) break;
            ca = R.s_col[pa];
        } else {
            if (++pb >= eb) break;
            cb = R.s_col[pb];
        }
    }
    return s;
}

__device__ __forceinline__ int64_t find_col(const Rows& R, int32_t user, int32_t col) {
    int64_t lo = R.u_ptr[user], hi = R.u_ptr[user + 1], e = hi;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (R.s_col[mid] < col) lo = mid + 1;
        else hi = mid;
    }
    return (lo < e && R.s_col[lo] == col) ? lo : -1;
}

// similarity evaluated with `w` as the first argument: an immutable.Set of <= 4 items iterates
// in insertion (file) order, larger sets in trie order
__device__ __forceinline__ double owner_dot(const Rows& R, int32_t w, int32_t o) {
    int64_t b = R.u_ptr[w], n = R.u_ptr[w + 1] - b;
    if (n > 4) return merge_dot(R, w, o);
    // visit w's <= 4 entries by ascending file row
    double s = 0.0;
    uint32_t last = 0;
    for (int64_t step = 0; step < n; ++step) {
        int64_t best = -1;
        uint32_t bt = 0xffffffffu;
        for (int64_t q = 0; q < n; ++q) {
            uint32_t t = R.s_t[b + q];
            if ((step == 0 || t > last) && t <= bt) { bt = t; best = b + q; }
        }
        last = bt;
        int64_t po = find_col(R, o, R.s_col[best]);
        if (po >= 0) s = s + R.s_pre[best] * R.s_pre[po];
    }
    return s;
}

// cosine closure :415-432 as seen while building u's neighbourhood (seq_u = u's build number):
// (v,u) is already memoised iff v's neighbourhood was built earlier, and is reused unless < 0.0
__device__ __forceinline__ double pair_sim(const Rows& R, int32_t u, int32_t v, int64_t seq_u, int64_t seq_v) {
    int64_t nu = R.u_ptr[u + 1] - R.u_ptr[u], nv = R.u_ptr[v + 1] - R.u_ptr[v];
    if (nu > 4 && nv > 4) return merge_dot(R, u, v);  // same order whoever owns it
    if (seq_v >= 0 && seq_v < seq_u) {
        double c = owner_dot(R, v, u);
        if (!(c < 0.0)) return c;
    }
    return owner_dot(R, u, v);
}

__device__ __forceinline__ bool ranks_before(double sa, int32_t ia, double sb, int32_t ib) {
    return sa > sb || (sa == sb && ia < ib);
}

// ---- K6b kernel ----------------------------------------------------------------------------------
// One workgroup per panel row u; u's item-sorted row is staged in LDS once.  One WAVE per candidate
// v: the lanes stream v's row 64 entries at a time (coalesced), each lane binary-searches its item in
// u's LDS row, and the matching products are folded left in lane (== item) order through the ballot
// mask — the same additions in the same order as the reference's `.sum` over uItems.intersect(vItems).
// Then an LDS bitonic sort keeps the best kk; shortlists longer than the LDS tile are consumed in
// chunks: [current best kk | next chunk] is sorted and cut to kk again (exact: total order).
static constexpr int RERANK_TILE = 2048;
static constexpr int UROW_LDS = 2048;  // rows up to this many ratings are looked up in LDS, longer ones in L1/L2

struct URow {
    const int32_t* col;
    const double* pre;
    int32_t n;
    int steps;  // binary-search iterations: ceil(log2(n + 1))
};

__device__ __forceinline__ double wave_merge_dot(const Rows& R, const URow& U, int32_t v, int lane) {
    const int64_t pb = R.u_ptr[v], eb = R.u_ptr[v + 1];
    double s = 0.0;
    const int32_t ufirst = U.col[0], ulast = U.col[U.n - 1];
    for (int64_t base = pb; base < eb; base += 64) {
        const int64_t p = base + lane;
        const bool valid = p < eb;
        const int32_t c = valid ? R.s_col[p] : 0x7fffffff;
        const double y = valid ? R.s_pre[p] : 0.0;
        // the whole 64-entry piece lies outside u's item range: nothing to match
        const int32_t cmin = __shfl(c, 0);
        const int64_t last_valid = min(eb - 1, base + 63) - base;
        const int32_t cmax = __shfl(c, (int)last_valid);
        if (cmin > ulast) break;
        if (cmax < ufirst) continue;
        int32_t lo = 0, hi = U.n;
        for (int it = 0; it < U.steps; ++it) {
            int32_t mid = (lo + hi) >> 1;
            bool go = lo < hi && U.col[min(mid, U.n - 1)] < c;
            bool stay = lo < hi;
            lo = go ? mid + 1 : lo;
            hi = (stay && !go) ? mid : hi;
        }
        const bool match = valid && lo < U.n && U.col[lo] == c;
        const double prod = match ? U.pre[lo] * y : 0.0;
        unsigned long long mask = __ballot(match);
        while (mask) {
            int j = __ffsll((long long)mask) - 1;
            s = s + __shfl(prod, j);
            mask &= mask - 1;
        }
    }
    return s;
}

__global__ void __launch_bounds__(TPB) k_rerank(Rows R, const int64_t* __restrict__ seq, int32_t n_rows,
                                                const int32_t* __restrict__ row_user, int32_t cap,
                                                const int32_t* __restrict__ cand_idx, const float* __restrict__ cand_approx,
                                                const int32_t* __restrict__ cand_cnt, int32_t kk, int32_t kcap,
                                                int32_t* __restrict__ nbr_idx, double* __restrict__ nbr_sim,
                                                int32_t* __restrict__ nbr_cnt, float eps_base, double* __restrict__ stats) {
    __shared__ double ssim[RERANK_TILE];
    __shared__ int32_t sidx[RERANK_TILE];
    __shared__ double upre[UROW_LDS];
    __shared__ int32_t ucol[UROW_LDS];
    const int32_t r = blockIdx.x;
    if (r >= n_rows) return;
    const int32_t cnt = cand_cnt[r];
    if (cnt > cap) return;  // overflow: the exact fallback redoes this row
    const int32_t u = row_user[r];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ub = R.u_ptr[u];
    const int32_t nu = (int32_t)(R.u_ptr[u + 1] - ub);
    URow U;
    U.n = nu;
    U.steps = 0;
    while ((1 << U.steps) < nu + 1) ++U.steps;
    if (nu <= UROW_LDS) {
        for (int32_t j = threadIdx.x; j < nu; j += TPB) {
            ucol[j] = R.s_col[ub + j];
            upre[j] = R.s_pre[ub + j];
        }
        U.col = ucol;
        U.pre = upre;
    } else {
        U.col = R.s_col + ub;
        U.pre = R.s_pre + ub;
    }
    const int64_t seq_u = seq[u];
    const float eps = row_eps(eps_base, nu);
    double worst = -1.0;
    int32_t best = 0, pos = 0;
    __syncthreads();
    do {
        const int32_t take = min(RERANK_TILE - best, cnt - pos);
        int32_t m = 1;
        while (m < best + take) m <<= 1;
        for (int32_t c = wave; c < take; c += TPB / 64) {
            const int32_t v = cand_idx[(int64_t)r * cap + pos + c];
            const int32_t nv = (int32_t)(R.u_ptr[v + 1] - R.u_ptr[v]);
            double s;
            if (nu > 4 && nv > 4) {
                s = wave_merge_dot(R, U, v, lane);
            } else {  // Set1..Set4 iterate in file order and the memo history matters (N2, N6): scalar path
                s = 0.0;
                if (lane == 0) s = pair_sim(R, u, v, seq_u, seq[v]);
                s = __shfl(s, 0);
            }
            if (lane == 0) {
                ssim[best + c] = s;
                sidx[best + c] = v;
                if (cand_approx) worst = fmax(worst, fabs((double)cand_approx[(int64_t)r * cap + pos + c] - s) - (double)eps);
            }
        }
        for (int32_t c = take + threadIdx.x; c < m - best; c += TPB) {
            ssim[best + c] = -INFINITY;
            sidx[best + c] = 0x7fffffff;
        }
        __syncthreads();
        for (int32_t size = 2; size <= m; size <<= 1) {
            for (int32_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (int32_t t = threadIdx.x; t < (m >> 1); t += TPB) {
                    int32_t lo = 2 * t - (t & (stride - 1));
                    int32_t hi = lo + stride;
                    bool up = ((lo & size) == 0);  // this sub-sequence ends "best first"
                    double sa = ssim[lo], sb = ssim[hi];
                    int32_t ia = sidx[lo], ib = sidx[hi];
                    bool a_first = ranks_before(sa, ia, sb, ib);
                    if (a_first != up) {
                        ssim[lo] = sb; ssim[hi] = sa;
                        sidx[lo] = ib; sidx[hi] = ia;
                    }
                }
                __syncthreads();
            }
        }
        best = min(kk, best + take);
        pos += take;
    } while (pos < cnt);
    if (cand_approx && worst > -1.0) {
        // max over the grid of (|approx - exact| - eps); must stay <= 0
        unsigned long long* w = reinterpret_cast<unsigned long long*>(stats);
        double shifted = worst + 4.0;  // positive, so the bit pattern orders like the value
        atomicMax(w, (unsigned long long)__double_as_longlong(shifted));
    }
    for (int32_t j = threadIdx.x; j < best; j += TPB) {
        nbr_idx[(int64_t)u * kcap + j] = sidx[j];
        nbr_sim[(int64_t)u * kcap + j] = ssim[j];
    }
    if (threadIdx.x == 0) nbr_cnt[u] = best;
}

void launch_rerank(const Train& tr, NeighborTable& nt, int32_t n_rows, const int32_t* d_row_user, int32_t cap,
                   const int32_t* cand_idx, const float* cand_approx, const int32_t* cand_cnt, float eps,
                   double* d_stats, bool verify, hipStream_t st) {
    if (n_rows <= 0) return;
    KN_REQUIRE(nt.kcap <= RERANK_TILE / 2, KNNCF_E_UNSUPPORTED, "k > 1024 is not supported by the re-rank kernel yet");
    Rows R{tr.u_ptr.p, tr.s_col.p, tr.s_t.p, tr.s_pre.p};
    k_rerank<<<n_rows, TPB, 0, st>>>(R, nt.seq.p, n_rows, d_row_user, cap, cand_idx, verify ? cand_approx : nullptr,
                                     cand_cnt, nt.kcap, nt.kcap, nt.idx.p, nt.sim.p, nt.cnt.p, eps, d_stats);
    KN_HIP(hipGetLastError());
}

// exact similarities of one user against everyone (out[user] = -inf): the fallback for rows whose
// shortlist overflowed and the engine behind scalar queries
__global__ void k_exact_row(Rows R, const int64_t* __restrict__ seq, int32_t U, int32_t user, int64_t user_seq,
                            double* __restrict__ out) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= U) return;
    out[v] = (v == user) ? -INFINITY : pair_sim(R, user, v, user_seq, seq[v]);
}

void launch_exact_row(const Train& tr, const NeighborTable& nt, int32_t user, int64_t user_seq, double* d_out,
                      hipStream_t st) {
    Rows R{tr.u_ptr.p, tr.s_col.p, tr.s_t.p, tr.s_pre.p};
    k_exact_row<<<(unsigned)ceil_div(tr.U, TPB), TPB, 0, st>>>(R, nt.seq.p, tr.U, user, user_seq, d_out);
    KN_HIP(hipGetLastError());
}

__global__ void k_exact_pair(Rows R, int32_t u, int32_t v, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = owner_dot(R, u, v);
}

void launch_exact_pair(const Train& tr, int32_t u, int32_t v, double* d_out, hipStream_t st) {
    Rows R{tr.u_ptr.p, tr.s_col.p, tr.s_t.p, tr.s_pre.p};
    k_exact_pair<<<1, 64, 0, st>>>(R, u, v, d_out);
    KN_HIP(hipGetLastError());
}

}  // namespace knncf
