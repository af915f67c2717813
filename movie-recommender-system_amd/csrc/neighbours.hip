// neighbours.hip — small orchestration kernels around the neighbourhood build: which users need
// a neighbourhood and in which order the reference's lazy closures would have built them
// (SURVEY N6), the exact fallback's ordering keys, and the Jaccard coefficient of one pair.
#include <math.h>

#include <algorithm>

#include <stdlib.h>

#include "engine.h"

namespace knncf {

static constexpr int TPB = 256;

// first test row (file order) at which the reference would call nn(u): the user is in train and
// the item has at least one rater (weightedSumDeviation :508-517 only then evaluates similarities)
__global__ void k_first_rows(int64_t n, const int32_t* __restrict__ du, const int32_t* __restrict__ di,
                             int32_t own_lo, int32_t own_hi, uint32_t* __restrict__ first) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int32_t u = -1;
    if (t < n) {
        u = du[t];
        if (u < own_lo || u >= own_hi || di[t] < 0) u = -1;
    }
    // test files list a user's rows together: of a run of equal users inside the wave only the first row (the smallest t)
    // goes to memory — 5 M atomics on 162 541 addresses became ~0.3 M
    const int32_t prev = __shfl_up(u, 1);
    if (u >= 0 && ((threadIdx.x & 63) == 0 || prev != u)) atomicMin(&first[u], (uint32_t)t);
}

void launch_first_rows(int64_t n, const int32_t* d_du, const int32_t* d_di, int32_t own_lo, int32_t own_hi,
                       uint32_t* d_first, hipStream_t st) {
    if (n <= 0) return;
    k_first_rows<<<(unsigned)ceil_div(n, TPB), TPB, 0, st>>>(n, d_du, d_di, own_lo, own_hi, d_first);
    KN_HIP(hipGetLastError());
}

// users that need a neighbourhood now; their build sequence number orders them like the
// reference's memo history: (call epoch, first test row)
// Sharded handles: EVERY user gets its sequence number (the test rows are replicated on every shard, so every shard
// derives the same numbers with no exchange — rerank.hip: pair_sim needs seq[v] of users another shard builds), but
// only the owned users [own_lo, own_hi) are listed for building.
__global__ void k_collect_new(int32_t U, const uint32_t* __restrict__ first, int64_t* __restrict__ seq, int64_t epoch,
                              int32_t own_lo, int32_t own_hi, int32_t* __restrict__ list, int32_t* __restrict__ count) {
    int32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    if (first[u] == 0xffffffffu || seq[u] >= 0) return;
    seq[u] = (epoch << 32) | (int64_t)first[u];
    if (u >= own_lo && u < own_hi) list[atomicAdd(count, 1)] = u;
}

void launch_collect_new(int32_t U, const uint32_t* d_first, int64_t* d_seq, int64_t epoch, int32_t own_lo, int32_t own_hi,
                        int32_t* d_list, int32_t* d_count, hipStream_t st) {
    k_collect_new<<<(unsigned)ceil_div(U, TPB), TPB, 0, st>>>(U, d_first, d_seq, epoch, own_lo, own_hi, d_list, d_count);
    KN_HIP(hipGetLastError());
}

// sort key of a build list by DESCENDING number of ratings (longest rows first: the select and re-rank kernels run
// one workgroup per row, and a 7000-rating row dispatched last would hold the launch open alone)
__global__ void k_length_keys(int32_t count, const int32_t* __restrict__ list, const int64_t* __restrict__ u_ptr, int32_t max_len,
                              uint64_t* __restrict__ key) {
    int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const int32_t u = list[r];
    key[r] = (uint64_t)max_len - (uint64_t)(u_ptr[u + 1] - u_ptr[u]);
}

void launch_length_keys(int32_t count, const int32_t* d_list, const int64_t* d_u_ptr, int32_t max_len, uint64_t* d_key, hipStream_t st) {
    k_length_keys<<<(unsigned)ceil_div(count, TPB), TPB, 0, st>>>(count, d_list, d_u_ptr, max_len, d_key);
    KN_HIP(hipGetLastError());
}

// ascending key order == descending similarity; a stable sort keeps equal similarities in dense
// (== HashSet iteration) order, i.e. sortWith(_._2 > _._2) on (allUsers - u).toSeq :608-610
__global__ void k_fallback_keys(int32_t U, const double* __restrict__ exact, uint64_t* __restrict__ keys,
                                uint32_t* __restrict__ vals) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= U) return;
    double s = exact[v];
    if (s == 0.0) s = 0.0;  // -0.0 and +0.0 compare equal in the reference's comparator
    uint64_t b = (uint64_t)__double_as_longlong(s);
    uint64_t ordered = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    keys[v] = ~ordered;
    vals[v] = (uint32_t)v;
}

void launch_fallback_keys(int32_t U, const double* d_exact, uint64_t* d_keys, uint32_t* d_vals, hipStream_t st) {
    k_fallback_keys<<<(unsigned)ceil_div(U, TPB), TPB, 0, st>>>(U, d_exact, d_keys, d_vals);
    KN_HIP(hipGetLastError());
}

__global__ void k_fallback_write(int32_t user, int32_t take, int32_t kcap, const uint32_t* __restrict__ sorted_vals,
                                 const double* __restrict__ exact, int32_t* __restrict__ nbr_idx,
                                 double* __restrict__ nbr_sim, int32_t* __restrict__ nbr_cnt) {
    int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) nbr_cnt[user] = take;
    if (j >= take) return;
    uint32_t v = sorted_vals[j];
    nbr_idx[(int64_t)user * kcap + j] = (int32_t)v;
    nbr_sim[(int64_t)user * kcap + j] = exact[v];
}

void launch_fallback_write(int32_t user, int32_t take, int32_t kcap, const uint32_t* d_sorted_vals,
                           const double* d_exact, int32_t* nbr_idx, double* nbr_sim, int32_t* nbr_cnt, hipStream_t st) {
    k_fallback_write<<<(unsigned)ceil_div(std::max(take, 1), TPB), TPB, 0, st>>>(user, take, kcap, d_sorted_vals, d_exact,
                                                                                 nbr_idx, nbr_sim, nbr_cnt);
    KN_HIP(hipGetLastError());
}

// jaccardCoefficient :446-463 (u, v dense or -1 when absent from train)
__global__ void k_jaccard_pair(const int64_t* __restrict__ u_ptr, const int32_t* __restrict__ s_col, int32_t u,
                               int32_t v, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t pa = 0, ea = 0, pb = 0, eb = 0;
    if (u >= 0) { pa = u_ptr[u]; ea = u_ptr[u + 1]; }
    if (v >= 0) { pb = u_ptr[v]; eb = u_ptr[v + 1]; }
    int64_t nu = ea - pa, nv = eb - pb, both = 0;
    while (pa < ea && pb < eb) {
        int32_t ca = s_col[pa], cb = s_col[pb];
        if (ca == cb) { ++both; ++pa; ++pb; }
        else if (ca < cb) ++pa;
        else ++pb;
    }
    *out = (double)both / (double)(nu + nv - both);  // 0.0 / 0 -> NaN, as in Scala
}

void launch_jaccard_pair(const Train& tr, int32_t u, int32_t v, double* d_out, hipStream_t st) {
    k_jaccard_pair<<<1, 64, 0, st>>>(tr.u_ptr.p, tr.s_col.p, u, v, d_out);
    KN_HIP(hipGetLastError());
}

}  // namespace knncf

namespace knncf {

// copy of each built neighbour list sorted by dense neighbour id (prediction probes the item's rater
// list, which is sorted the same way): one workgroup per row, LDS bitonic sort of (id, position) packed in 64 bits;
// the similarities are gathered by position at the end.  Pair t of a stage belongs to the 128-element block t / 64,
// and a wave owns whole blocks: stages of stride <= 64 stay inside one wave's blocks and need no workgroup barrier
// (42 of the 45 stages at k = 300).
__global__ void __launch_bounds__(256) k_sort_neighbors_by_id(int32_t n_rows, const int32_t* __restrict__ row_user, int32_t kcap,
                                                              const int32_t* __restrict__ nbr_idx, const double* __restrict__ nbr_sim,
                                                              const int32_t* __restrict__ nbr_cnt, int32_t* __restrict__ uidx,
                                                              double* __restrict__ usim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int32_t r = blockIdx.x;
    if (r >= n_rows) return;
    const int32_t u = row_user ? row_user[r] : r;
    const int32_t cnt = nbr_cnt[u];
    int32_t m = 128;  // (whole blocks of 128: the wave-local stages assume them)
    while (m < cnt) m <<= 1;
    unsigned long long* key = reinterpret_cast<unsigned long long*>(smem);
    const int64_t base = (int64_t)u * kcap;
    for (int32_t j = threadIdx.x; j < m; j += 256)
        key[j] = j < cnt ? ((unsigned long long)(uint32_t)nbr_idx[base + j] << 32) | (uint32_t)j : ~0ull;
    __syncthreads();
    for (int32_t size = 2; size <= m; size <<= 1)
        for (int32_t stride = size >> 1; stride > 0; stride >>= 1) {
            const bool across = stride > 64;  // partners in different blocks: other waves' elements
            if (across) __syncthreads();
            for (int32_t t = threadIdx.x; t < (m >> 1); t += 256) {
                const int32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const unsigned long long ka = key[lo], kb = key[hi];
                if ((ka < kb) != up) { key[lo] = kb; key[hi] = ka; }
            }
            if (across) __syncthreads();
            else {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    __syncthreads();
    for (int32_t j = threadIdx.x; j < cnt; j += 256) {
        const unsigned long long k = key[j];
        uidx[base + j] = (int32_t)(k >> 32);
        usim[base + j] = nbr_sim[base + (uint32_t)k];
    }
}

// The same copy without a sort network: the ids are distinct integers below U, so the sorted position of an id is its RANK —
// the number of the row's ids below it.  The row's ids are marked in an LDS bitmap over the users, a prefix popcount over
// the bitmap's words gives every word's rank base, and each neighbour lands at base + popcount(bits below it in its word).
// O(U / 64 + k) per row instead of 45 compare-exchange stages (1.28 -> 0.5 ms at the ml-25m shape); needs 12 B of LDS per
// 64 users, so shapes beyond ~340 k users keep the bitonic kernel above.
__global__ void __launch_bounds__(256) k_rank_neighbors_by_id(int32_t n_rows, const int32_t* __restrict__ row_user, int32_t kcap, int32_t words,
                                                              const int32_t* __restrict__ nbr_idx, const double* __restrict__ nbr_sim,
                                                              const int32_t* __restrict__ nbr_cnt, int32_t* __restrict__ uidx,
                                                              double* __restrict__ usim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t wave_tot[4];
    uint32_t* bits = reinterpret_cast<uint32_t*>(smem);  // [2 * words] the bitmap as 32-bit halves (LDS atomics)
    uint32_t* base_of = bits + 2 * words;                // [words] ids of the row below word w
    const int32_t r = blockIdx.x;
    if (r >= n_rows) return;
    const int32_t u = row_user ? row_user[r] : r;
    const int32_t cnt = nbr_cnt[u];
    const int64_t base = (int64_t)u * kcap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int32_t w = threadIdx.x; w < 2 * words; w += 256) bits[w] = 0;
    __syncthreads();
    for (int32_t j = threadIdx.x; j < cnt; j += 256) {
        const uint32_t v = (uint32_t)nbr_idx[base + j];
        atomicOr(&bits[v >> 5], 1u << (v & 31u));
    }
    __syncthreads();
    // exclusive prefix popcount over the 64-bit words: a contiguous run of words per thread, DPP scan, the waves' totals through LDS
    const int32_t per = (words + 255) / 256;
    const int32_t w0 = min(words, (int32_t)threadIdx.x * per), w1 = min(words, w0 + per);
    uint32_t mine = 0;
    for (int32_t w = w0; w < w1; ++w) mine += __popc(bits[2 * w]) + __popc(bits[2 * w + 1]);
    const uint32_t incl = wave_incl_scan(mine);
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
    for (int32_t w = w0; w < w1; ++w) {
        base_of[w] = run;
        run += __popc(bits[2 * w]) + __popc(bits[2 * w + 1]);
    }
    __syncthreads();
    for (int32_t j = threadIdx.x; j < cnt; j += 256) {
        const uint32_t v = (uint32_t)nbr_idx[base + j];
        const uint32_t w = v >> 6, h = (v >> 5) & 1u, b = v & 31u;
        uint32_t pos = base_of[w] + __popc(bits[2 * w + h] & ((1u << b) - 1u));
        if (h) pos += __popc(bits[2 * w]);
        uidx[base + pos] = (int32_t)v;
        usim[base + pos] = nbr_sim[base + j];
    }
}

void launch_sort_neighbors(NeighborTable& nt, int32_t n_rows, const int32_t* d_row_user, hipStream_t st) {
    if (n_rows <= 0 || nt.kcap <= 0) return;
    const size_t cells = nt.cnt.n * (size_t)nt.kcap;
    nt.uidx.ensure(cells);
    nt.usim.ensure(cells);
    {
        const int64_t U = (int64_t)nt.cnt.n;  // (one count per user)
        const int32_t words = (int32_t)ceil_div(U, 64);
        const size_t lds = (size_t)words * 12;
        if (lds <= 48 * 1024 && !getenv("KNNCF_DEBUG_BITONIC_ID_SORT")) {
            k_rank_neighbors_by_id<<<n_rows, 256, lds, st>>>(n_rows, d_row_user, nt.kcap, words, nt.idx.p, nt.sim.p, nt.cnt.p, nt.uidx.p, nt.usim.p);
            KN_HIP(hipGetLastError());
            return;
        }
    }
    int32_t m = 128;
    while (m < nt.kcap) m <<= 1;
    const size_t smem = (size_t)m * 8;
    k_sort_neighbors_by_id<<<n_rows, 256, smem, st>>>(n_rows, d_row_user, nt.kcap, nt.idx.p, nt.sim.p, nt.cnt.p, nt.uidx.p, nt.usim.p);
    KN_HIP(hipGetLastError());
}

}  // namespace knncf
