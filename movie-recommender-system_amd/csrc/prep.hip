// prep.hip — K0-K4: id compaction, user-major / item-major orders, ordered segmented folds
// (means, norms, item deviations), normalized deviations and preprocessed ratings.
//
// Everything here is HBM-bound integer / fp64 work.  The arithmetic follows
// shared/predictions.scala operation by operation and IN THE SAME ORDER: sums
// are left folds along explicit permutations, staged through LDS by coalesced loads and then
// folded sequentially per segment, so results are bit-identical to the fp64 oracle.
#include <math.h>

#include <algorithm>
#include <vector>

#include <stdlib.h>

#include "engine.h"

namespace knncf {

static constexpr int TPB = 256;
static inline int nblocks(int64_t n, int per = TPB) { return (int)std::max<int64_t>(1, ceil_div(n, per)); }

void PrepScratch::ensure_aux() {
    if (aux) return;
    KN_HIP(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
    KN_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    KN_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    KN_HIP(hipEventCreateWithFlags(&ev_commit, hipEventDisableTiming));
}
void PrepScratch::join_commit(hipStream_t st) {
    if (!commit_pending) return;
    KN_HIP(hipStreamWaitEvent(st, ev_commit, 0));
    commit_pending = false;
}
PrepScratch::~PrepScratch() {
    if (aux) { (void)hipStreamSynchronize(aux); (void)hipStreamDestroy(aux); }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_commit) (void)hipEventDestroy(ev_commit);
}

void PrepScratch::release_all() {
    sort.tmp.release();
    k64_a.release(); k64_b.release(); v32_a.release(); v32_b.release();
    k32_a.release(); k32_b.release(); du_row.release(); di_row.release(); perm_f.release(); rec.release(); long_rows.release(); ucnt.release(); utile.release(); perm_iu.release();
}

// ---- K0: ids ---------------------------------------------------------------------------
__global__ void k_row_keys(int64_t n, const int32_t* __restrict__ users, const int32_t* __restrict__ items,
                           uint32_t* __restrict__ ukey, uint32_t* __restrict__ ikey) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    ukey[t] = int_trie_key(users[t]);
    ikey[t] = int_trie_key(items[t]);
}

// range of the raw ids (decides whether the direct id tables can be used): grid-stride, one atomic per block and bound —
// atomics from every wave of a 20 M-row launch on four addresses serialise in the L2 (measured: +14 ms)
__global__ void __launch_bounds__(TPB) k_id_range(int64_t n, const int32_t* __restrict__ users, const int32_t* __restrict__ items,
                                                  int32_t* __restrict__ idrange) {
    __shared__ int32_t part[4][TPB / 64];
    int32_t u = 0x7fffffff, i = 0x7fffffff, um = (int32_t)0x80000000, im = (int32_t)0x80000000;
    for (int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x; t < n; t += (int64_t)gridDim.x * TPB) {
        const int32_t a = users[t], b = items[t];
        u = min(u, a); um = max(um, a);
        i = min(i, b); im = max(im, b);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        u = min(u, __shfl_xor(u, o)); um = max(um, __shfl_xor(um, o));
        i = min(i, __shfl_xor(i, o)); im = max(im, __shfl_xor(im, o));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[0][wave] = u; part[1][wave] = um; part[2][wave] = i; part[3][wave] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; ++w) {
            u = min(u, part[0][w]); um = max(um, part[1][w]);
            i = min(i, part[2][w]); im = max(im, part[3][w]);
        }
        atomicMin(&idrange[0], u); atomicMax(&idrange[1], um);
        atomicMin(&idrange[2], i); atomicMax(&idrange[3], im);
    }
}

// direct id tables: table[raw] = dense index or -1.  Every row marks its id (plain stores of the same value: no race that
// matters), then one thread per table cell resolves the marked cells with the hash + binary search the rows would have done.
__global__ void k_mark_ids(int64_t n, const int32_t* __restrict__ users, const int32_t* __restrict__ items,
                           int32_t* __restrict__ u_table, int32_t* __restrict__ i_table) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    u_table[users[t]] = 0;
    i_table[items[t]] = 0;
}
__global__ void k_resolve_table(int32_t cells, int32_t* __restrict__ table, const uint32_t* __restrict__ keys, int32_t count) {
    int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= cells) return;
    if (table[r] == 0) table[r] = dense_lookup(keys, count, r);
}
// trie keys of the marked cells of a direct table, in any order (they are sorted afterwards): one atomic per wave
__global__ void k_table_keys(int32_t cells, const int32_t* __restrict__ table, uint32_t* __restrict__ keys, uint32_t* __restrict__ count) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool present = r < cells && table[r] == 0;
    const unsigned long long m = __ballot(present);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    uint32_t base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (present) keys[base + __popcll(m & ((1ull << lane) - 1ull))] = int_trie_key(r);
}

// raw id of every dense index, read off a resolved direct table (one thread per cell instead of one store per rating)
__global__ void k_raw_ids_of_table(int32_t cells, const int32_t* __restrict__ table, int32_t* __restrict__ out) {
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < cells && table[r] >= 0) out[table[r]] = r;
}

__global__ void k_dense_ids_table(int64_t n, const int32_t* __restrict__ users, const int32_t* __restrict__ items,
                                  const int32_t* __restrict__ u_table, int32_t u_cells, const int32_t* __restrict__ i_table,
                                  int32_t i_cells, int32_t* __restrict__ du, int32_t* __restrict__ di) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int32_t u = users[t], i = items[t];
    du[t] = (u >= 0 && u < u_cells) ? u_table[u] : -1;
    di[t] = (i >= 0 && i < i_cells) ? i_table[i] : -1;
}

// first file row of each of <= 4 distinct keys (Set1..Set4 keep insertion order)
__global__ void k_first_occurrence(int64_t n, const uint32_t* __restrict__ row_key, const uint32_t* __restrict__ keys,
                                   int32_t count, unsigned long long* __restrict__ first) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint32_t k = row_key[t];
    for (int32_t i = 0; i < count; ++i)
        if (keys[i] == k) atomicMin(&first[i], (unsigned long long)t);
}

__global__ void k_dense_ids(int64_t n, const int32_t* __restrict__ users, const int32_t* __restrict__ items,
                            const uint32_t* __restrict__ ukeys, int32_t U, const uint32_t* __restrict__ ikeys,
                            int32_t I, int32_t* __restrict__ du, int32_t* __restrict__ di) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    du[t] = dense_lookup(ukeys, U, users[t]);
    di[t] = dense_lookup(ikeys, I, items[t]);
}

void launch_dense_ids(const Train& tr, const int32_t* d_users, const int32_t* d_items, int64_t n,
                      int32_t* d_du, int32_t* d_di, hipStream_t st) {
    if (n == 0) return;
    if (tr.u_table_n > 0 && tr.i_table_n > 0)
        k_dense_ids_table<<<nblocks(n), TPB, 0, st>>>(n, d_users, d_items, tr.u_table.p, tr.u_table_n, tr.i_table.p, tr.i_table_n, d_du, d_di);
    else
        k_dense_ids<<<nblocks(n), TPB, 0, st>>>(n, d_users, d_items, tr.ukeys.p, tr.U, tr.ikeys.p, tr.I, d_du, d_di);
    KN_HIP(hipGetLastError());
}

// raw id of every dense index (any row of that user / item carries it)
__global__ void k_scatter_raw_ids(int64_t n, const int32_t* __restrict__ raw, const int32_t* __restrict__ dense,
                                  int32_t* __restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    out[dense[t]] = raw[t];
}

// mode 0: key = du<<lo_bits | di (di < 2^lo_bits: the fewer key bits, the fewer radix passes) ; 1: du ;
// 2: du<<32 | tuple key ; 3: di ; 4: di<<32 | tuple key
__global__ void k_make_keys(int64_t n, int mode, const int32_t* __restrict__ du, const int32_t* __restrict__ di,
                            const int32_t* __restrict__ users, const int32_t* __restrict__ items,
                            uint64_t* __restrict__ key, int lo_bits) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint64_t k;
    switch (mode) {
        case 0: k = ((uint64_t)(uint32_t)du[t] << lo_bits) | (uint32_t)di[t]; break;
        case 1: k = (uint32_t)du[t]; break;
        case 2: k = ((uint64_t)(uint32_t)du[t] << 32) | tuple_trie_key(users[t], items[t]); break;
        case 3: k = (uint32_t)di[t]; break;
        default: k = ((uint64_t)(uint32_t)di[t] << 32) | tuple_trie_key(users[t], items[t]); break;
    }
    key[t] = k;
}

// sharded fits: keys over the owned slice [p0, p0 + m) of the canonical order — (user - lo, file row) with the position
// as the value, then (user - lo, trie key of the (user, item) tuple) along the first sort's result
__global__ void k_slice_file_keys(int64_t p0, int64_t m, int32_t lo, const int32_t* __restrict__ s_user, const uint32_t* __restrict__ s_t,
                                  int tbits, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int64_t p = p0 + j;
    key[j] = ((uint64_t)(uint32_t)(s_user[p] - lo) << tbits) | s_t[p];
    val[j] = (uint32_t)p;
}
__global__ void k_slice_hash_keys(int64_t m, int32_t lo, const uint32_t* __restrict__ pos, const int32_t* __restrict__ s_user,
                                  const uint32_t* __restrict__ s_t, const int32_t* __restrict__ users, const int32_t* __restrict__ items,
                                  uint64_t* __restrict__ key) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t p = pos[j], t = s_t[p];
    key[j] = ((uint64_t)(uint32_t)(s_user[p] - lo) << 32) | tuple_trie_key(users[t], items[t]);
}

// ---- (user, HashMap order) positions by a sort of every user's own segment in LDS ---------------------------------
// perm_uh is the canonical positions re-ordered, inside every user, by (trie key of the (user, item) tuple hash, file row).
// As a global radix sort that is 7 passes over 20 M (64-bit key, value) pairs; but the canonical order already groups the
// positions by user, and a user's segment (123 ratings on average) fits in LDS: one wave sorts one segment (bitonic network
// over 64-bit keys (trie key << 32 | file row) with the local index riding along; no workgroup barrier), a whole workgroup
// the few segments beyond SEG_WAVE_CAP.  A segment longer than SEG_BLOCK_CAP sets ST_LONG_ROW and the fit falls back to
// the global sort.
static constexpr int SEG_SHORT_CAP = 512;    // class 0: one wave per segment, every user (5 KB of LDS per wave)
static constexpr int SEG_MID_CAP = 2048;     // class 1: one workgroup per segment of the middle list (20 KB: eight per CU)
static constexpr int SEG_BLOCK_CAP = 8192;   // class 2: one workgroup per segment of the long list (80 KB)
// (powers of two: a segment is padded to the next one for the network)

// The network's stages with a stride of at most 64 pair elements inside one 128-element chunk, and the chunk of pair index t
// belongs to the wave that t's 64-group maps to (THREADS is a multiple of 64): those stages need no workgroup barrier, only
// the wave's own ordering.  A workgroup barrier is due before a stage whose pairs cross chunks and before the first
// wave-local stage after one (27 of the 91 stages of an 8192-element segment instead of all of them).
template <int THREADS, bool BLOCK, bool VAL>
__device__ __forceinline__ void segment_bitonic(unsigned long long* key, uint16_t* val, int32_t N, int tid) {
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    bool prev_wide = true;  // (the segment was loaded by arbitrary threads)
    for (int32_t size = 2; size <= N; size <<= 1) {
        for (int32_t stride = size >> 1; stride > 0; stride >>= 1) {
            const bool wide = stride >= 128;
            if (BLOCK && (wide || prev_wide)) __syncthreads();
            else wave_sync();
            prev_wide = wide;
            for (int32_t t = tid; t < (N >> 1); t += THREADS) {
                const int32_t lo = 2 * t - (t & (stride - 1));
                const int32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a = key[lo], b = key[hi];
                if ((a > b) == up) {
                    key[lo] = b; key[hi] = a;
                    if (VAL) {
                        const uint16_t va = val[lo], vb = val[hi];
                        val[lo] = vb; val[hi] = va;
                    }
                }
            }
        }
    }
    if (BLOCK) __syncthreads();
    else wave_sync();
}

struct SegLists {
    int32_t* mid;        // users of class 1
    int32_t* longs;      // users of class 2
    uint32_t* counts;    // [0] = |mid|, [1] = |longs|
};

// users beyond the one-wave class, filed by size class (one thread per user; the order inside a list carries no meaning)
__global__ void k_classify_rows(int32_t U, const int64_t* __restrict__ u_ptr, SegLists L) {
    const int32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const int64_t n = u_ptr[u + 1] - u_ptr[u];
    if (n > SEG_SHORT_CAP && n <= SEG_MID_CAP) L.mid[atomicAdd(&L.counts[0], 1u)] = u;
    else if (n > SEG_MID_CAP && n <= SEG_BLOCK_CAP) L.longs[atomicAdd(&L.counts[1], 1u)] = u;
}
static constexpr int seg_threads(int cls) { return cls == 2 ? 512 : TPB; }

// CLASS 0 walks the users [u_lo, u_hi) (one wave each, skipping the longer segments); CLASS 1 / 2 walk their list
// (k_classify_rows; a workgroup per entry, grid-stride: the list's length stays on the device) and keep to the same user
// range.  The three classes touch disjoint users, so prep_fit runs the two list classes on a second stream beside class 0:
// a segment of 8192 is one workgroup's chain of 91 barrier-separated network stages, 0.2 ms during which it needs no CU
// but its own.  The host only takes this path when no segment exceeds SEG_BLOCK_CAP (prep_fit reads the longest row's
// length back); ST_LONG_ROW is a consistency check.
// FILE_ORDER: the key is the file row alone — perm_uf, the (user, file row) order of usersAvg :113.
template <int CLASS, bool FILE_ORDER>
__global__ void __launch_bounds__(seg_threads(CLASS)) k_user_hash_order(const int64_t* __restrict__ u_ptr, int32_t u_lo, int32_t u_hi, SegLists L,
                                                         const int32_t* __restrict__ uid, const int32_t* __restrict__ iid,
                                                         const int32_t* __restrict__ s_col, const uint32_t* __restrict__ s_t,
                                                         uint32_t* __restrict__ perm_out, uint32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char seg_smem[];
    constexpr bool BLOCK = CLASS >= 1;
    constexpr int CAP = CLASS == 0 ? SEG_SHORT_CAP : CLASS == 1 ? SEG_MID_CAP : SEG_BLOCK_CAP;
    constexpr int THREADS = BLOCK ? seg_threads(CLASS) : 64;
    constexpr int SEGS = BLOCK ? 1 : TPB / 64;  // segments in flight per workgroup
    const int wave = BLOCK ? 0 : (int)(threadIdx.x >> 6);
    const int tid = BLOCK ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    unsigned long long* key = reinterpret_cast<unsigned long long*>(seg_smem) + (size_t)wave * CAP;
    uint16_t* val = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned long long*>(seg_smem) + (size_t)SEGS * CAP) + (size_t)wave * CAP;
    const int64_t total = CLASS == 0 ? (int64_t)(u_hi - u_lo) : (int64_t)L.counts[CLASS - 1];
    for (int64_t idx = (int64_t)blockIdx.x * SEGS + wave; idx < total; idx += (int64_t)gridDim.x * SEGS) {
        const int32_t u = CLASS == 0 ? u_lo + (int32_t)idx : (CLASS == 1 ? L.mid[idx] : L.longs[idx]);
        if (CLASS > 0 && (u < u_lo || u >= u_hi)) continue;  // (workgroup-uniform)
        const int64_t b = u_ptr[u], e = u_ptr[u + 1];
        const int64_t n64 = e - b;
        if (n64 > CAP) {  // (only class 0 meets these: they are on the lists)
            if (n64 > SEG_BLOCK_CAP && tid == 0) atomicOr(status, (uint32_t)ST_LONG_ROW);
            continue;
        }
        const int32_t n = (int32_t)n64;
        int32_t N = 2;
        while (N < n) N <<= 1;
        const int32_t user_raw = uid[u];
        for (int32_t i = tid; i < N; i += THREADS) {
            unsigned long long k = ~0ull;
            if (i < n) {
                if (FILE_ORDER) k = s_t[b + i];
                else k = ((unsigned long long)tuple_trie_key(user_raw, iid[s_col[b + i]]) << 32) | s_t[b + i];
            }
            key[i] = k;
            val[i] = (uint16_t)i;
        }
        segment_bitonic<THREADS, BLOCK, true>(key, val, N, tid);
        for (int32_t i = tid; i < n; i += THREADS) perm_out[b + i] = (uint32_t)(b + val[i]);
        // (the next segment's stores into key / val follow this wave's / workgroup's reads in program order; the workgroup
        // form needs its barrier)
        if (BLOCK) __syncthreads();
    }
}

// ---- the canonical (user, item) order by a counting scatter + a sort of every user's own segment in LDS ---------------
// As a global radix sort the canonical order is 5 passes over 20 M (64-bit key, value) pairs.  Dense user indices make the
// first level a counting sort — one histogram, its prefix (= u_ptr), one scatter through per-user cursors, which leaves a
// user's rows together but in no particular order — and a user's rows fit in LDS, where they are sorted by item (the same
// three size classes as the hash order below).  A (user, item) pair is unique, so the result does not depend on the order
// the scatter's atomics happened to run in; a duplicate pair shows up as two equal neighbours and sets ST_DUPLICATE.
// Rating files are usually grouped by user (every MovieLens file is), so the lanes of a wave mostly hold ONE user: 20 M
// atomics of which 64 in a row hit the same address serialise in the L2 (measured: 1.3 ms for the count, 1.8 ms for the
// scatter).  A wave therefore aggregates its RUNS of equal users — a lane that differs from the lane below starts a run and
// issues one atomic for the run's length; equal users that are not neighbours simply make two runs.
struct WaveRun {
    int head_lane;  // first lane of this lane's run
    int len;        // length of the run (meaningful on the head lane)
    bool head;
};
__device__ __forceinline__ WaveRun wave_runs(int32_t u, bool valid) {
    const int lane = threadIdx.x & 63;
    const int32_t below = __shfl_up(u, 1);
    const unsigned long long vmask = __ballot(valid);
    WaveRun r;
    r.head = valid && (lane == 0 || below != u);
    const unsigned long long heads = __ballot(r.head);
    const unsigned long long upto = heads & ((2ull << lane) - 1ull);            // heads at or below this lane
    r.head_lane = upto ? 63 - __clzll((long long)upto) : 0;
    const unsigned long long above = lane == 63 ? 0ull : heads & ~((2ull << lane) - 1ull);  // heads above this lane
    const int n_valid = __popcll(vmask);                                          // (the valid lanes are a prefix)
    r.len = (above ? __ffsll((long long)above) - 1 : n_valid) - lane;
    return r;
}
__global__ void k_count_users(int64_t n, const int32_t* __restrict__ du, uint32_t* __restrict__ cnt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = t < n;
    const int32_t u = valid ? du[t] : -1;
    const WaveRun r = wave_runs(u, valid);
    if (r.head) atomicAdd(&cnt[u], (uint32_t)r.len);
}
static constexpr int SCAN_TILE = TPB * 8;
// per tile of SCAN_TILE users: the number of their ratings; the longest row of all -> *max_len
__global__ void __launch_bounds__(TPB) k_user_tile_sums(int32_t U, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ tile_sum,
                                                        uint32_t* __restrict__ max_len) {
    __shared__ uint32_t ws[TPB / 64], wm[TPB / 64];
    uint32_t s = 0, m = 0;
    for (int j = 0; j < 8; ++j) {
        const int64_t i = (int64_t)blockIdx.x * SCAN_TILE + j * TPB + threadIdx.x;
        const uint32_t c = i < U ? cnt[i] : 0u;
        s += c;
        m = max(m, c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        m = max(m, __shfl_xor(m, o));
    }
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = s; wm[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; ++w) { s += ws[w]; m = max(m, wm[w]); }
        tile_sum[blockIdx.x] = s;
        atomicMax(max_len, m);
    }
}
// u_ptr = exclusive prefix of the counts; the counts are cleared on the way (they become the scatter's cursors)
__global__ void __launch_bounds__(TPB) k_user_ptr(int32_t U, uint32_t* __restrict__ cnt, const uint32_t* __restrict__ tile_sum,
                                                  int64_t* __restrict__ u_ptr) {
    __shared__ unsigned long long wsum[TPB / 64];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long mine = 0;  // ratings of the users in the tiles before this one
    for (int q = threadIdx.x; q < (int)blockIdx.x; q += TPB) mine += tile_sum[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0) wsum[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long b = 0;
        for (int w = 0; w < TPB / 64; ++w) b += wsum[w];
        s_base = b;
    }
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * 8;  // 8 consecutive users per thread
    uint32_t c[8], tot = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        c[j] = i0 + j < U ? cnt[i0 + j] : 0u;
        tot += c[j];
    }
    const uint32_t incl = wave_incl_scan(tot);
    __syncthreads();  // (wsum is reused)
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned long long run = s_base + incl - tot;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (i0 + j < U) {
            u_ptr[i0 + j] = (int64_t)run;
            cnt[i0 + j] = 0;
        }
        run += c[j];
        if (i0 + j == (int64_t)U - 1) u_ptr[U] = (int64_t)run;
    }
}
// row t goes to the next free slot of its user's segment, as (item << 32 | file row)
__global__ void k_scatter_rows(int64_t n, const int32_t* __restrict__ du, const int32_t* __restrict__ di,
                               const int64_t* __restrict__ u_ptr, uint32_t* __restrict__ cursor, unsigned long long* __restrict__ rows) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = t < n;
    const int32_t u = valid ? du[t] : -1;
    const WaveRun r = wave_runs(u, valid);
    uint32_t base = 0;
    if (r.head) base = atomicAdd(&cursor[u], (uint32_t)r.len);
    base = __shfl(base, r.head_lane);
    if (!valid) return;
    const int64_t slot = u_ptr[u] + base + (uint32_t)((int)(threadIdx.x & 63) - r.head_lane);
    rows[slot] = ((unsigned long long)(uint32_t)di[t] << 32) | (uint32_t)t;
}
template <int CLASS>
__global__ void __launch_bounds__(seg_threads(CLASS)) k_user_item_order(const int64_t* __restrict__ u_ptr, int32_t U, SegLists L,
                                                         const unsigned long long* __restrict__ rows, const double* __restrict__ rating,
                                                         int32_t* __restrict__ s_user, int32_t* __restrict__ s_col,
                                                         uint32_t* __restrict__ s_t, double* __restrict__ s_rating,
                                                         uint32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char seg_smem[];
    constexpr bool BLOCK = CLASS >= 1;
    constexpr int CAP = CLASS == 0 ? SEG_SHORT_CAP : CLASS == 1 ? SEG_MID_CAP : SEG_BLOCK_CAP;
    constexpr int THREADS = BLOCK ? seg_threads(CLASS) : 64;
    constexpr int SEGS = BLOCK ? 1 : TPB / 64;
    const int wave = BLOCK ? 0 : (int)(threadIdx.x >> 6);
    const int tid = BLOCK ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    unsigned long long* key = reinterpret_cast<unsigned long long*>(seg_smem) + (size_t)wave * CAP;
    const int64_t total = CLASS == 0 ? (int64_t)U : (int64_t)L.counts[CLASS - 1];
    for (int64_t idx = (int64_t)blockIdx.x * SEGS + wave; idx < total; idx += (int64_t)gridDim.x * SEGS) {
        const int32_t u = CLASS == 0 ? (int32_t)idx : (CLASS == 1 ? L.mid[idx] : L.longs[idx]);
        const int64_t b = u_ptr[u], e = u_ptr[u + 1];
        const int64_t n64 = e - b;
        if (n64 > CAP) {  // (only class 0 meets these: they are on the lists)
            if (n64 > SEG_BLOCK_CAP && tid == 0) atomicOr(status, (uint32_t)ST_LONG_ROW);
            continue;
        }
        const int32_t n = (int32_t)n64;
        int32_t N = 2;
        while (N < n) N <<= 1;
        for (int32_t i = tid; i < N; i += THREADS) key[i] = i < n ? rows[b + i] : ~0ull;
        segment_bitonic<THREADS, BLOCK, false>(key, nullptr, N, tid);
        for (int32_t i = tid; i < n; i += THREADS) {
            const unsigned long long k = key[i];
            const uint32_t t = (uint32_t)k;
            s_user[b + i] = u;
            s_col[b + i] = (int32_t)(k >> 32);
            s_t[b + i] = t;
            s_rating[b + i] = rating[t];
            if (i > 0 && (key[i - 1] >> 32) == (k >> 32)) atomicOr(status, (uint32_t)ST_DUPLICATE);
        }
        if (BLOCK) __syncthreads();
    }
}

// 32-bit keys for the two plain fold orders (a third less sort traffic than 64-bit keys)
__global__ void k_copy_keys_u32(int64_t n, const int32_t* __restrict__ dense, uint32_t* __restrict__ key) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) key[t] = (uint32_t)dense[t];
}

// ptr[s] = first index whose key >= s, s in [0, S]
__global__ void k_segment_ptr_u32(int64_t n, const uint32_t* __restrict__ sorted_key, int32_t S, int64_t* __restrict__ ptr) {
    int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > S) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (sorted_key[mid] < (uint32_t)s) lo = mid + 1;
        else hi = mid;
    }
    ptr[s] = lo;
}

__global__ void k_iota(int64_t n, uint32_t* __restrict__ v) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) v[t] = (uint32_t)t;
}

// unpack the sorted (user, item) keys; flag duplicate (user, item) rows
__global__ void k_unpack_positions(int64_t n, const uint64_t* __restrict__ key, int32_t* __restrict__ s_user,
                                   int32_t* __restrict__ s_col, uint32_t* __restrict__ status, int lo_bits) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint64_t k = key[p];
    s_user[p] = (int32_t)(k >> lo_bits);
    s_col[p] = (int32_t)(k & ((1ull << lo_bits) - 1ull));
    if (p > 0 && key[p - 1] == k) atomicOr(status, (uint32_t)ST_DUPLICATE);
}

__global__ void k_gather_f64(int64_t n, const uint32_t* __restrict__ idx, const double* __restrict__ src,
                             double* __restrict__ dst) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) dst[p] = src[idx[p]];
}

// perm_f[file row] = position
__global__ void k_invert(int64_t n, const uint32_t* __restrict__ s_t, uint32_t* __restrict__ perm_f) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) perm_f[s_t[p]] = (uint32_t)p;
}

// ---- ordered segmented fold ---------------------------------------------------------------
// out[s] = left fold (0.0 + v0) + v1 ... over the segment's elements in perm order
// (perm == nullptr: identity).  SQUARE: v = src*src (math.pow(x, 2) in usersWeights :474).
// One block owns 256 consecutive segments; their element range is staged through LDS in
// chunks by coalesced loads of perm (+ gather of src), then each thread folds its own
// segment's part of the chunk sequentially.
static constexpr int FOLD_CHUNK = 2048;

// `spb` consecutive segments per block (<= TPB): the skewed item segments (one item can hold 0.4 % of all
// ratings) use small groups so that no block serialises several heavy segments; the next chunk's gathers
// are in flight (registers) while the current chunk is folded out of LDS.
template <bool SQUARE>
__global__ void __launch_bounds__(TPB) k_ordered_fold(const int64_t* __restrict__ seg_ptr, int32_t seg_lo,
                                                      int32_t seg_hi, int32_t spb, const uint32_t* __restrict__ perm,
                                                      const double* __restrict__ src, double* __restrict__ out) {
    __shared__ double buf[2][FOLD_CHUNK];
    constexpr int PER = FOLD_CHUNK / TPB;
    int32_t s0 = seg_lo + blockIdx.x * spb;
    int32_t s1 = min(s0 + spb, seg_hi);
    if (s0 >= seg_hi) return;
    int32_t s = s0 + threadIdx.x;
    const bool folder = (int32_t)threadIdx.x < spb && s < s1;
    int64_t e0 = seg_ptr[s0], e1 = seg_ptr[s1];
    int64_t b = 0, e = 0;
    if (folder) {
        b = seg_ptr[s];
        e = seg_ptr[s + 1];
    }
    double acc = 0.0;
    double r[PER];
    auto gather = [&](int64_t c) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int64_t i = c + threadIdx.x + (int64_t)j * TPB;
            r[j] = 0.0;
            if (i < e1) {
                const int64_t idx = perm ? (int64_t)perm[i] : i;
                const double v = src[idx];
                r[j] = SQUARE ? v * v : v;
            }
        }
    };
    auto put = [&](int slot) {
#pragma unroll
        for (int j = 0; j < PER; ++j) buf[slot][threadIdx.x + j * TPB] = r[j];
    };
    gather(e0);
    put(0);
    __syncthreads();
    int slot = 0;
    for (int64_t c = e0; c < e1; c += FOLD_CHUNK, slot ^= 1) {
        const bool more = c + FOLD_CHUNK < e1;
        if (more) gather(c + FOLD_CHUNK);  // in flight while this chunk is folded
        const int64_t lo = max(b, c), hi = min(e, min(e1, c + FOLD_CHUNK));
        {
            // The fold is ONE dependent chain of fp64 additions per segment, and the heaviest segment (an item with
            // 0.4 % of all ratings) sets the kernel's time: keep the chain fed.  The LDS reads of the next 16 values
            // are issued before the current 16 are added, so the chain never waits for a read (a plain loop paid the
            // LDS latency, ~9x the add's, on every element).
            constexpr int FB = 16;
            const double* src_l = &buf[slot][0] - c;
            int64_t q = lo;
            if (q + FB <= hi) {
                double v[FB], w[FB];
#pragma unroll
                for (int j = 0; j < FB; ++j) v[j] = src_l[q + j];
                for (; q + 2 * FB <= hi; q += FB) {
#pragma unroll
                    for (int j = 0; j < FB; ++j) w[j] = src_l[q + FB + j];
#pragma unroll
                    for (int j = 0; j < FB; ++j) acc = acc + v[j];
#pragma unroll
                    for (int j = 0; j < FB; ++j) v[j] = w[j];
                }
#pragma unroll
                for (int j = 0; j < FB; ++j) acc = acc + v[j];
                q += FB;
            }
            for (; q < hi; ++q) acc = acc + src_l[q];
        }
        if (more) put(slot ^ 1);
        __syncthreads();
    }
    if (folder) out[s] = acc;
}

// whole-array left fold in index order (average :94 when the ratings are not dyadic)
__global__ void __launch_bounds__(TPB) k_sequential_sum(int64_t n, const double* __restrict__ src, double* __restrict__ out) {
    __shared__ double buf[FOLD_CHUNK];
    double acc = 0.0;
    for (int64_t c = 0; c < n; c += FOLD_CHUNK) {
        int32_t len = (int32_t)min((int64_t)FOLD_CHUNK, n - c);
        for (int32_t j = threadIdx.x; j < len; j += TPB) buf[j] = src[c + j];
        __syncthreads();
        if (threadIdx.x == 0)
            for (int32_t j = 0; j < len; ++j) acc = acc + buf[j];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = acc;
}

// ratings that are multiples of 1/16 and small sum EXACTLY in fp64 in any order
__global__ void k_check_dyadic(int64_t n, const double* __restrict__ r, uint32_t* __restrict__ status) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    double v = r[t] * 16.0;
    if (!(fabs(v) <= 16777216.0) || v != rint(v)) atomicOr(status, (uint32_t)ST_NOT_DYADIC);
}

// exact (dyadic) sum: order-free block reduction + atomicAdd, every partial is exact
__global__ void __launch_bounds__(TPB) k_exact_sum(int64_t n, const double* __restrict__ src, double* __restrict__ out) {
    __shared__ double red[TPB];
    double acc = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x; t < n; t += (int64_t)gridDim.x * TPB) acc += src[t];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, red[0]);
}

// mean = sum / count for segments [lo, hi)
__global__ void k_divide_by_count(const int64_t* __restrict__ seg_ptr, int32_t lo, int32_t hi,
                                  const double* __restrict__ sum, double* __restrict__ out) {
    int32_t s = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= hi) return;
    out[s] = sum[s] / (double)(seg_ptr[s + 1] - seg_ptr[s]);
}

__global__ void k_sqrt(int32_t lo, int32_t hi, const double* __restrict__ in, double* __restrict__ out) {
    int32_t s = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= hi) return;
    out[s] = sqrt(in[s]);
}

// computeNormalizeDeviation :162-168 for positions [p0, p1)
__global__ void k_deviation(int64_t p0, int64_t p1, const int32_t* __restrict__ s_user,
                            const double* __restrict__ s_rating, const double* __restrict__ user_avg,
                            double* __restrict__ s_dev, uint32_t* __restrict__ status) {
    int64_t p = p0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= p1) return;
    double r = s_rating[p], ua = user_avg[s_user[p]];
    double d = (r - ua) / scale_fn(r, ua);
    s_dev[p] = d;
    if (!isfinite(d)) atomicOr(status, (uint32_t)ST_NONFINITE);
}

// preprocessedRating :476-480
__global__ void k_preprocess(int64_t p0, int64_t p1, const int32_t* __restrict__ s_user,
                             const double* __restrict__ s_dev, const double* __restrict__ user_norm,
                             double* __restrict__ s_pre) {
    int64_t p = p0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= p1) return;
    double w = user_norm[s_user[p]];
    s_pre[p] = (w != 0) ? s_dev[p] / w : 0.0;
}

template <bool SQUARE>
static void fold(const int64_t* seg_ptr, int32_t lo, int32_t hi, const uint32_t* perm, const double* src,
                 double* out, hipStream_t st, int32_t spb = TPB) {
    if (hi <= lo) return;
    k_ordered_fold<SQUARE><<<nblocks(hi - lo, spb), TPB, 0, st>>>(seg_ptr, lo, hi, spb, perm, src, out);
    KN_HIP(hipGetLastError());
}

static uint32_t read_status(PrepScratch& sc, hipStream_t st) {
    uint32_t h = 0;
    KN_HIP(hipMemcpyAsync(&h, sc.status.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    KN_HIP(hipStreamSynchronize(st));
    return h;
}

void prep_fit(Train& tr, PrepScratch& sc, int32_t shard_rank, int32_t shard_count, hipStream_t st) {
    const int64_t n = tr.n;
    KN_REQUIRE(n > 0, KNNCF_E_INVALID, "fit: empty training set");
    KN_REQUIRE(n < (int64_t)1 << 29, KNNCF_E_UNSUPPORTED, "fit: more than 2^29-1 ratings (the kernels address the rating arrays with 32-bit byte offsets)");
    if (sc.aux) KN_HIP(hipStreamSynchronize(sc.aux));  // (a fit that failed between fork and join may have left work there)
    sc.commit_pending = false;
    sc.status.ensure(4);
    KN_HIP(hipMemsetAsync(sc.status.p, 0, 4 * sizeof(uint32_t), st));
    sc.k32_a.ensure(n); sc.k32_b.ensure(n); sc.v32_a.ensure(n); sc.v32_b.ensure(n);
    sc.k64_a.ensure(n); sc.k64_b.ensure(n);

    // distinct users / items in HashSet iteration order.  MovieLens ids are small non-negative integers: then the distinct
    // ids come out of a presence table over the raw id (one plain store per row), and only THEY are hashed and sorted
    // (162 541 + 59 047 keys instead of two sorts of 20 M); any other id space takes the sort + unique of every row's key.
    uint32_t* ukey_row = sc.k32_a.p;
    uint32_t* ikey_row = sc.v32_a.p;
    bool have_row_keys = false;
    sc.idrange.ensure(4);
    {
        const int32_t init[4] = {0x7fffffff, (int32_t)0x80000000, 0x7fffffff, (int32_t)0x80000000};
        KN_HIP(hipMemcpyAsync(sc.idrange.p, init, sizeof(init), hipMemcpyHostToDevice, st));
    }
    k_id_range<<<(unsigned)std::min<int64_t>(1024, ceil_div(n, TPB)), TPB, 0, st>>>(n, tr.user_raw.p, tr.item_raw.p, sc.idrange.p);
    // (K1 / K2 sum ratings: exact in any order when they are dyadic — checked here, read back with the id range)
    k_check_dyadic<<<nblocks(n), TPB, 0, st>>>(n, tr.rating.p, sc.status.p);
    KN_HIP(hipGetLastError());
    int32_t rg[4];
    uint32_t early_status = 0;
    KN_HIP(hipMemcpyAsync(rg, sc.idrange.p, sizeof(rg), hipMemcpyDeviceToHost, st));
    KN_HIP(hipMemcpyAsync(&early_status, sc.status.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    KN_HIP(hipStreamSynchronize(st));
    const bool dyadic = !(early_status & ST_NOT_DYADIC);
    const int32_t ID_LIMIT = 1 << 24;
    const bool small_ids = rg[0] >= 0 && rg[2] >= 0 && rg[1] < ID_LIMIT && rg[3] < ID_LIMIT && !getenv("KNNCF_DEBUG_NO_ID_TABLES");
    tr.u_table_n = tr.i_table_n = 0;
    size_t U = 0, I = 0;
    if (small_ids) {
        tr.u_table_n = rg[1] + 1;
        tr.i_table_n = rg[3] + 1;
        tr.u_table.ensure(tr.u_table_n);
        tr.i_table.ensure(tr.i_table_n);
        KN_HIP(hipMemsetAsync(tr.u_table.p, 0xff, (size_t)tr.u_table_n * sizeof(int32_t), st));  // -1
        KN_HIP(hipMemsetAsync(tr.i_table.p, 0xff, (size_t)tr.i_table_n * sizeof(int32_t), st));
        k_mark_ids<<<nblocks(n), TPB, 0, st>>>(n, tr.user_raw.p, tr.item_raw.p, tr.u_table.p, tr.i_table.p);
        KN_HIP(hipMemsetAsync(sc.status.p + 2, 0, 2 * sizeof(uint32_t), st));  // (words 2, 3: the two counts)
        k_table_keys<<<nblocks(tr.u_table_n), TPB, 0, st>>>(tr.u_table_n, tr.u_table.p, sc.k32_b.p, sc.status.p + 2);
        k_table_keys<<<nblocks(tr.i_table_n), TPB, 0, st>>>(tr.i_table_n, tr.i_table.p, sc.v32_b.p, sc.status.p + 3);
        KN_HIP(hipGetLastError());
        uint32_t cnt[2];
        KN_HIP(hipMemcpyAsync(cnt, sc.status.p + 2, sizeof(cnt), hipMemcpyDeviceToHost, st));
        KN_HIP(hipStreamSynchronize(st));
        U = cnt[0];
        I = cnt[1];
        tr.ukeys.alloc(U);
        tr.ikeys.alloc(I);
        sort_keys_u32(sc.sort, sc.k32_b.p, tr.ukeys.p, U, st);
        sort_keys_u32(sc.sort, sc.v32_b.p, tr.ikeys.p, I, st);
    } else {
        k_row_keys<<<nblocks(n), TPB, 0, st>>>(n, tr.user_raw.p, tr.item_raw.p, ukey_row, ikey_row);
        KN_HIP(hipGetLastError());
        have_row_keys = true;
        sort_keys_u32(sc.sort, ukey_row, sc.k32_b.p, n, st);
        U = unique_u32(sc.sort, sc.k32_b.p, sc.v32_b.p, n, st);
        tr.ukeys.alloc(U);
        KN_HIP(hipMemcpyAsync(tr.ukeys.p, sc.v32_b.p, U * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        sort_keys_u32(sc.sort, ikey_row, sc.k32_b.p, n, st);
        I = unique_u32(sc.sort, sc.k32_b.p, sc.v32_b.p, n, st);
        tr.ikeys.alloc(I);
        KN_HIP(hipMemcpyAsync(tr.ikeys.p, sc.v32_b.p, I * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    }
    KN_REQUIRE(U < (1u << 31) && I < (1u << 31), KNNCF_E_UNSUPPORTED, "fit: too many distinct ids");
    tr.U = (int32_t)U;
    tr.I = (int32_t)I;
    if (U <= 4) {  // Set1..Set4: `ratings.map(_.user).toSet` iterates in first-occurrence order (N3)
        DArr<unsigned long long> first;
        first.alloc(4);
        KN_HIP(hipMemsetAsync(first.p, 0xff, 4 * sizeof(unsigned long long), st));
        if (!have_row_keys) k_row_keys<<<nblocks(n), TPB, 0, st>>>(n, tr.user_raw.p, tr.item_raw.p, ukey_row, ikey_row);
        k_first_occurrence<<<nblocks(n), TPB, 0, st>>>(n, ukey_row, tr.ukeys.p, tr.U, first.p);
        KN_HIP(hipGetLastError());
        unsigned long long hf[4];
        uint32_t hk[4];
        KN_HIP(hipMemcpyAsync(hf, first.p, sizeof(hf), hipMemcpyDeviceToHost, st));
        KN_HIP(hipMemcpyAsync(hk, tr.ukeys.p, U * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        KN_HIP(hipStreamSynchronize(st));
        std::vector<int> ord(U);
        for (size_t i = 0; i < U; ++i) ord[i] = (int)i;
        std::sort(ord.begin(), ord.end(), [&](int a, int b) { return hf[a] < hf[b]; });
        uint32_t nk[4];
        for (size_t i = 0; i < U; ++i) nk[i] = hk[ord[i]];
        KN_HIP(hipMemcpyAsync(tr.ukeys.p, nk, U * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        KN_HIP(hipStreamSynchronize(st));
    }

    sc.du_row.ensure(n); sc.di_row.ensure(n);
    if (small_ids) {  // marked cells -> dense index (ukeys / ikeys are final now, the <= 4 users case included)
        k_resolve_table<<<nblocks(tr.u_table_n), TPB, 0, st>>>(tr.u_table_n, tr.u_table.p, tr.ukeys.p, tr.U);
        k_resolve_table<<<nblocks(tr.i_table_n), TPB, 0, st>>>(tr.i_table_n, tr.i_table.p, tr.ikeys.p, tr.I);
        KN_HIP(hipGetLastError());
    }
    launch_dense_ids(tr, tr.user_raw.p, tr.item_raw.p, n, sc.du_row.p, sc.di_row.p, st);
    tr.uid.alloc(U); tr.iid.alloc(I);
    if (small_ids) {
        k_raw_ids_of_table<<<nblocks(tr.u_table_n), TPB, 0, st>>>(tr.u_table_n, tr.u_table.p, tr.uid.p);
        k_raw_ids_of_table<<<nblocks(tr.i_table_n), TPB, 0, st>>>(tr.i_table_n, tr.i_table.p, tr.iid.p);
    } else {
        k_scatter_raw_ids<<<nblocks(n), TPB, 0, st>>>(n, tr.user_raw.p, sc.du_row.p, tr.uid.p);
        k_scatter_raw_ids<<<nblocks(n), TPB, 0, st>>>(n, tr.item_raw.p, sc.di_row.p, tr.iid.p);
    }
    KN_HIP(hipGetLastError());

    const int ubits = bits_for(U), ibits = bits_for(I);
    tr.s_user.alloc(n); tr.s_col.alloc(n); tr.s_t.alloc(n); tr.s_rating.alloc(n);
    tr.s_dev.alloc(n); tr.s_pre.alloc(n); tr.u_ptr.alloc(U + 1); tr.i_ptr.alloc(I + 1);
    // row counts -> u_ptr, and the longest row: it decides between the per-user LDS sorts and the global radix sorts
    const int32_t u_tiles = nblocks(U, SCAN_TILE);
    sc.ucnt.ensure(U); sc.utile.ensure(u_tiles);
    KN_HIP(hipMemsetAsync(sc.ucnt.p, 0, U * sizeof(uint32_t), st));
    k_count_users<<<nblocks(n), TPB, 0, st>>>(n, sc.du_row.p, sc.ucnt.p);
    k_user_tile_sums<<<u_tiles, TPB, 0, st>>>(tr.U, sc.ucnt.p, sc.utile.p, sc.status.p + 1);
    k_user_ptr<<<u_tiles, TPB, 0, st>>>(tr.U, sc.ucnt.p, sc.utile.p, tr.u_ptr.p);
    KN_HIP(hipGetLastError());

    // owned block of users: ascending dense index, ceil(U / shards) each
    const int32_t per = (int32_t)ceil_div(U, shard_count);
    tr.own_lo = std::min<int64_t>((int64_t)per * shard_rank, U);
    tr.own_hi = std::min<int64_t>((int64_t)per * (shard_rank + 1), U);
    const int32_t lo = tr.own_lo, hi = tr.own_hi;
    int64_t p0 = 0, p1 = n;  // the owned users' positions in the canonical order
    uint32_t max_len = 0;
    {
        int64_t hp[2] = {0, n};
        KN_HIP(hipMemcpyAsync(&max_len, sc.status.p + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        if (shard_count > 1) {
            KN_HIP(hipMemcpyAsync(&hp[0], tr.u_ptr.p + lo, sizeof(int64_t), hipMemcpyDeviceToHost, st));
            KN_HIP(hipMemcpyAsync(&hp[1], tr.u_ptr.p + hi, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        }
        KN_HIP(hipStreamSynchronize(st));
        p0 = hp[0];
        p1 = hp[1];
    }
    tr.own_p0 = p0;
    tr.own_p1 = p1;
    // (KNNCF_DEBUG_GLOBAL_HASH_ORDER: test hook, forces the global sorts that a user with more than SEG_BLOCK_CAP ratings takes)
    const bool seg_sorts = max_len <= (uint32_t)SEG_BLOCK_CAP && !getenv("KNNCF_DEBUG_GLOBAL_HASH_ORDER");
    sc.long_rows.ensure(2 * (size_t)U + 2);
    SegLists L{sc.long_rows.p, sc.long_rows.p + U + 1, sc.status.p + 2};
    constexpr size_t smem_pair0 = (size_t)(TPB / 64) * SEG_SHORT_CAP * 10, smem_pair1 = (size_t)SEG_MID_CAP * 10, smem_pair2 = (size_t)SEG_BLOCK_CAP * 10;
    constexpr size_t smem_key0 = (size_t)(TPB / 64) * SEG_SHORT_CAP * 8, smem_key1 = (size_t)SEG_MID_CAP * 8, smem_key2 = (size_t)SEG_BLOCK_CAP * 8;

    // canonical user-major order: file rows by (user, item)
    bool have_perm_f = false;
    auto need_perm_f = [&] {  // position of every file row: the value sequence of the global fold-order sorts
        if (have_perm_f) return;
        sc.perm_f.ensure(n);
        k_invert<<<nblocks(n), TPB, 0, st>>>(n, tr.s_t.p, sc.perm_f.p);
        have_perm_f = true;
    };
    // fold orders: the positions of every owned user re-ordered inside the user's segment
    // (the item-side fold orders belong to K4, which only the baseline predictors use: prep_item_stats)
    // usersAvg :113 sums a user's ratings in file order; dyadic ratings (every MovieLens file) sum exactly in ANY order, so a
    // whole-file fit then folds them in the canonical order and never builds the (user, file row) order
    const bool need_uf = !dyadic || n <= 4 || shard_count > 1;
    if (need_uf) tr.perm_uf.alloc(n);
    else tr.perm_uf.release();
    if (n > 4) tr.perm_uh.alloc(n);  // a Map of <= 4 entries (Map1..Map4) iterates in insertion = file order (N4)
    else tr.perm_uh.release();
    if (seg_sorts) {
        // Every user's segment sorted in LDS: by item (the canonical order, all users), then — the owned users — by file row
        // and by (trie key of the tuple hash, file row).  Three size classes; the two list classes run on the second stream.
        static PerDeviceState lds2i, lds1h, lds2h, lds1f, lds2f;
        ensure_dynamic_lds(lds2i, (const void*)k_user_item_order<2>, smem_key2);
        ensure_dynamic_lds(lds1h, (const void*)k_user_hash_order<1, false>, smem_pair1);
        ensure_dynamic_lds(lds2h, (const void*)k_user_hash_order<2, false>, smem_pair2);
        ensure_dynamic_lds(lds1f, (const void*)k_user_hash_order<1, true>, smem_pair1);
        ensure_dynamic_lds(lds2f, (const void*)k_user_hash_order<2, true>, smem_pair2);
        sc.ensure_aux();
        hipStream_t sx = sc.aux;
        k_scatter_rows<<<nblocks(n), TPB, 0, st>>>(n, sc.du_row.p, sc.di_row.p, tr.u_ptr.p, sc.ucnt.p,
                                                   reinterpret_cast<unsigned long long*>(sc.k64_a.p));
        KN_HIP(hipMemsetAsync(sc.status.p + 2, 0, 2 * sizeof(uint32_t), st));  // (words 2, 3: the two list lengths)
        k_classify_rows<<<nblocks(U), TPB, 0, st>>>(tr.U, tr.u_ptr.p, L);
        KN_HIP(hipGetLastError());
        KN_HIP(hipEventRecord(sc.ev_fork, st));
        KN_HIP(hipStreamWaitEvent(sx, sc.ev_fork, 0));
        const unsigned long long* rows = reinterpret_cast<const unsigned long long*>(sc.k64_a.p);
        const bool own = hi > lo;
        k_user_item_order<2><<<1024, seg_threads(2), smem_key2, sx>>>(tr.u_ptr.p, tr.U, L, rows, tr.rating.p, tr.s_user.p, tr.s_col.p, tr.s_t.p, tr.s_rating.p, sc.status.p);
        k_user_item_order<1><<<2048, TPB, smem_key1, sx>>>(tr.u_ptr.p, tr.U, L, rows, tr.rating.p, tr.s_user.p, tr.s_col.p, tr.s_t.p, tr.s_rating.p, sc.status.p);
        k_user_item_order<0><<<nblocks(U, TPB / 64), TPB, smem_key0, st>>>(tr.u_ptr.p, tr.U, L, rows, tr.rating.p, tr.s_user.p, tr.s_col.p, tr.s_t.p, tr.s_rating.p, sc.status.p);
        if (need_uf && own) {
            k_user_hash_order<2, true><<<1024, seg_threads(2), smem_pair2, sx>>>(tr.u_ptr.p, lo, hi, L, tr.uid.p, tr.iid.p, tr.s_col.p, tr.s_t.p, tr.perm_uf.p, sc.status.p);
            k_user_hash_order<1, true><<<2048, TPB, smem_pair1, sx>>>(tr.u_ptr.p, lo, hi, L, tr.uid.p, tr.iid.p, tr.s_col.p, tr.s_t.p, tr.perm_uf.p, sc.status.p);
            k_user_hash_order<0, true><<<nblocks(hi - lo, TPB / 64), TPB, smem_pair0, st>>>(tr.u_ptr.p, lo, hi, L, tr.uid.p, tr.iid.p, tr.s_col.p, tr.s_t.p, tr.perm_uf.p, sc.status.p);
        }
        if (n > 4 && own) {
            k_user_hash_order<2, false><<<1024, seg_threads(2), smem_pair2, sx>>>(tr.u_ptr.p, lo, hi, L, tr.uid.p, tr.iid.p, tr.s_col.p, tr.s_t.p, tr.perm_uh.p, sc.status.p);
            k_user_hash_order<1, false><<<2048, TPB, smem_pair1, sx>>>(tr.u_ptr.p, lo, hi, L, tr.uid.p, tr.iid.p, tr.s_col.p, tr.s_t.p, tr.perm_uh.p, sc.status.p);
            k_user_hash_order<0, false><<<nblocks(hi - lo, TPB / 64), TPB, smem_pair0, st>>>(tr.u_ptr.p, lo, hi, L, tr.uid.p, tr.iid.p, tr.s_col.p, tr.s_t.p, tr.perm_uh.p, sc.status.p);
        }
        KN_HIP(hipGetLastError());
        KN_HIP(hipEventRecord(sc.ev_join, sx));
        KN_HIP(hipStreamWaitEvent(st, sc.ev_join, 0));
    } else {
        k_make_keys<<<nblocks(n), TPB, 0, st>>>(n, 0, sc.du_row.p, sc.di_row.p, tr.user_raw.p, tr.item_raw.p, sc.k64_a.p, ibits);
        k_iota<<<nblocks(n), TPB, 0, st>>>(n, sc.v32_a.p);
        KN_HIP(hipGetLastError());
        sort_pairs_u64_u32(sc.sort, sc.k64_a.p, sc.k64_b.p, sc.v32_a.p, tr.s_t.p, n, ibits + ubits, st);
        k_unpack_positions<<<nblocks(n), TPB, 0, st>>>(n, sc.k64_b.p, tr.s_user.p, tr.s_col.p, sc.status.p, ibits);
        k_gather_f64<<<nblocks(n), TPB, 0, st>>>(n, tr.s_t.p, tr.rating.p, tr.s_rating.p);
    }
    KN_HIP(hipGetLastError());
    if (!seg_sorts && shard_count == 1) {  // stable global sorts of the file-order sequence of positions
        need_perm_f();
        if (need_uf) {
            k_copy_keys_u32<<<nblocks(n), TPB, 0, st>>>(n, sc.du_row.p, sc.k32_a.p);
            sort_pairs_u32_u32(sc.sort, sc.k32_a.p, sc.k32_b.p, sc.perm_f.p, tr.perm_uf.p, n, ubits, st);
        }
        if (n > 4) {
            k_make_keys<<<nblocks(n), TPB, 0, st>>>(n, 2, sc.du_row.p, sc.di_row.p, tr.user_raw.p, tr.item_raw.p, sc.k64_a.p, 32);
            sort_pairs_u64_u32(sc.sort, sc.k64_a.p, sc.k64_b.p, sc.perm_f.p, tr.perm_uh.p, n, 32 + ubits, st);
        }
    } else if (!seg_sorts && p1 > p0) {
        // A shard folds its own users only (K2, K3 below), so it orders only their positions: the slice [p0, p1) of the
        // canonical order by (user, file row) — the same segments the whole-file sort yields, 1/shards of the work —
        // and that sequence, stably re-sorted by (user, trie key)
        const int tbits = bits_for((uint64_t)n);
        k_slice_file_keys<<<nblocks(p1 - p0), TPB, 0, st>>>(p0, p1 - p0, lo, tr.s_user.p, tr.s_t.p, tbits, sc.k64_a.p, sc.v32_a.p);
        KN_HIP(hipGetLastError());
        sort_pairs_u64_u32(sc.sort, sc.k64_a.p, sc.k64_b.p, sc.v32_a.p, tr.perm_uf.p + p0, p1 - p0, tbits + bits_for((uint64_t)(hi - lo)), st);
        if (n > 4) {
            k_slice_hash_keys<<<nblocks(p1 - p0), TPB, 0, st>>>(p1 - p0, lo, tr.perm_uf.p + p0, tr.s_user.p, tr.s_t.p, tr.user_raw.p,
                                                               tr.item_raw.p, sc.k64_a.p);
            KN_HIP(hipGetLastError());
            sort_pairs_u64_u32(sc.sort, sc.k64_a.p, sc.k64_b.p, tr.perm_uf.p + p0, tr.perm_uh.p + p0, p1 - p0,
                               32 + bits_for((uint64_t)(hi - lo)), st);
        }
    }
    KN_HIP(hipGetLastError());
    tr.item_stats_ready = false;

    // K1: average :94 — left fold over the file; exact in any order for dyadic ratings
    uint32_t status = read_status(sc, st);
    KN_REQUIRE(!(status & ST_DUPLICATE), KNNCF_E_DUPLICATE, "fit: duplicate (user,item) training rows");
    KN_REQUIRE(!(status & ST_LONG_ROW), KNNCF_E_STATE, "fit: a row longer than the segment sorts take reached them");
    // (A pair with a <= 4-rating user is summed in the order of whichever closure evaluated it first, SURVEY N6.  Sharded
    // handles follow that history too: the test set is replicated, so every shard assigns EVERY user its build sequence
    // number — api.cpp: ensure_neighbors_for_rows — and rerank.hip applies the owner rule with no exchange.)
    sc.dsum.ensure(2);
    KN_HIP(hipMemsetAsync(sc.dsum.p, 0, 2 * sizeof(double), st));
    if (status & ST_NOT_DYADIC) k_sequential_sum<<<1, TPB, 0, st>>>(n, tr.rating.p, sc.dsum.p);
    else k_exact_sum<<<1024, TPB, 0, st>>>(n, tr.rating.p, sc.dsum.p);
    KN_HIP(hipGetLastError());
    double total = 0.0;
    KN_HIP(hipMemcpyAsync(&total, sc.dsum.p, sizeof(double), hipMemcpyDeviceToHost, st));
    KN_HIP(hipStreamSynchronize(st));
    tr.global_avg = total / (double)n;

    tr.user_avg.alloc(U); tr.user_norm.alloc(U);
    if (shard_count > 1) {  // the host all-gathers the other shards' (mean, norm) segments in place; prep_complete_rows then
                            // fills in the other users' deviations and preprocessed ratings
        KN_HIP(hipMemsetAsync(tr.user_avg.p, 0, U * sizeof(double), st));
        KN_HIP(hipMemsetAsync(tr.user_norm.p, 0, U * sizeof(double), st));
    }
    if (hi > lo) {
        // K2: usersAvg :113 (groupBy keeps file order; mean = reduce(_+_) / length)
        fold<false>(tr.u_ptr.p, lo, hi, need_uf ? tr.perm_uf.p : nullptr, tr.s_rating.p, tr.user_norm.p, st);
        k_divide_by_count<<<nblocks(hi - lo), TPB, 0, st>>>(tr.u_ptr.p, lo, hi, tr.user_norm.p, tr.user_avg.p);
        // K3: deviations, norms (HashMap order), preprocessed ratings
        k_deviation<<<nblocks(p1 - p0), TPB, 0, st>>>(p0, p1, tr.s_user.p, tr.s_rating.p, tr.user_avg.p, tr.s_dev.p, sc.status.p);
        fold<true>(tr.u_ptr.p, lo, hi, n > 4 ? tr.perm_uh.p : tr.perm_uf.p, tr.s_dev.p, tr.user_norm.p, st);
        k_sqrt<<<nblocks(hi - lo), TPB, 0, st>>>(lo, hi, tr.user_norm.p, tr.user_norm.p);
        k_preprocess<<<nblocks(p1 - p0), TPB, 0, st>>>(p0, p1, tr.s_user.p, tr.s_dev.p, tr.user_norm.p, tr.s_pre.p);
        KN_HIP(hipGetLastError());
    }
    status = read_status(sc, st);
    KN_REQUIRE(!(status & ST_NONFINITE), KNNCF_E_NONFINITE,
               "fit: non-finite normalized deviation (a user's mean is 1 or 5 with a rating beyond it: scale() == 0)");
}

void prep_complete_rows(Train& tr, PrepScratch& sc, hipStream_t st) {
    const int64_t p0 = tr.own_p0, p1 = tr.own_p1, n = tr.n;
    // (a non-finite deviation among another shard's users failed that shard's fit already; the status word is not re-read)
    if (p0 > 0) {
        k_deviation<<<nblocks(p0), TPB, 0, st>>>(0, p0, tr.s_user.p, tr.s_rating.p, tr.user_avg.p, tr.s_dev.p, sc.status.p);
        k_preprocess<<<nblocks(p0), TPB, 0, st>>>(0, p0, tr.s_user.p, tr.s_dev.p, tr.user_norm.p, tr.s_pre.p);
    }
    if (p1 < n) {
        k_deviation<<<nblocks(n - p1), TPB, 0, st>>>(p1, n, tr.s_user.p, tr.s_rating.p, tr.user_avg.p, tr.s_dev.p, sc.status.p);
        k_preprocess<<<nblocks(n - p1), TPB, 0, st>>>(p1, n, tr.s_user.p, tr.s_dev.p, tr.user_norm.p, tr.s_pre.p);
    }
    KN_HIP(hipGetLastError());
}

// item-major (item, user ascending) copies: (user, preprocessed rating) for the sparse tail of the similarity,
// (user, deviation, file row) for the prediction's "which neighbours rated item i" probes
// The permutation scatters the reads over the whole rating array: gathering four arrays costs four memory sectors per
// entry (1.3 ms at ml-25m shape).  The four values are packed into one 32-byte record first (a streaming pass), so that
// the gather touches one sector.
__global__ void k_pack_records(int64_t n, const int32_t* __restrict__ s_user, const double* __restrict__ s_pre,
                               const double* __restrict__ s_dev, const uint32_t* __restrict__ s_t, uint4* __restrict__ rec) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const unsigned long long a = (unsigned long long)__double_as_longlong(s_pre[p]);
    const unsigned long long b = (unsigned long long)__double_as_longlong(s_dev[p]);
    rec[2 * p] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    rec[2 * p + 1] = make_uint4((uint32_t)s_user[p], s_t[p], 0u, 0u);
}

__global__ void k_item_major(int64_t n, const uint32_t* __restrict__ perm_iu, const uint4* __restrict__ rec,
                             int32_t* __restrict__ it_user, uint32_t* __restrict__ it_pack,
                             double* __restrict__ it_dev, uint32_t* __restrict__ it_t, int ones) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    uint32_t p = perm_iu[q];
    const uint4 r0 = rec[2 * (int64_t)p], r1 = rec[2 * (int64_t)p + 1];
    const double pre = __longlong_as_double((long long)(((unsigned long long)r0.y << 32) | r0.x));
    const double dev = __longlong_as_double((long long)(((unsigned long long)r0.w << 32) | r0.z));
    const int32_t user = (int32_t)r1.x;
    it_user[q] = user;
    {  // Tail word of select.hip: W = (value field, 16 bits, high) | (BYTE address of the column's LDS cell inside its tile, 16
       // bits, low).  The kernel uses the WHOLE word as the rater-side factor (one v_cvt_f32_i32, no field extraction) and
       // masks the address out with one v_and: W / TAIL_SCALE must therefore approximate pre as a whole, so the value field
       // is rounded AFTER subtracting the address bits — |W - pre * TAIL_SCALE| <= 2^15, i.e. |W / TAIL_SCALE - pre| <=
       // 2^-16 (1 + 2^-15): the Q0.16 budget of the row's error band (select.hip: tail_abs * 1.0001 / 65536).  TAIL_SCALE =
       // 2^31 - 2^16 keeps the field inside int16 for |pre| <= 1 without clamping.  The cell address already carries
       // select.hip's accumulator layout (the low and the high 4 columns of every group of 8 live in separate halves:
       // conflict-free 16-byte read-out).
        const uint32_t c = (uint32_t)user & (uint32_t)(SELECT_TCOLS - 1);
        const uint32_t cell = (((c >> 3) << 2) | (c & 3u)) + ((c & 4u) ? (uint32_t)(SELECT_TCOLS / 2) : 0u);
        const uint32_t addr = cell << 2;  // < 2^16 (SELECT_TCOLS <= 16384)
        // Jaccard handles count common items: every tail entry weighs 2^30 and the row-side factor is 2^-30 (select.hip rounds
        // the product to the nearest integer: 1)
        const double target = ones ? 1073741824.0 : fmin(fmax(pre, -1.0), 1.0) * 2147418112.0;
        long long top = __double2ll_rn((target - (double)addr) / 65536.0);
        top = top < -32768 ? -32768 : (top > 32767 ? 32767 : top);
        it_pack[q] = ((uint32_t)(int32_t)top << 16) + addr;
    }
    it_dev[q] = dev;
    it_t[q] = r1.y;
}

// Rater bitmap of every item (bit v of row i <=> dense user v rated dense item i) and the exclusive prefix popcount
// along each row, from the item-major copy: one wave per item walks the item's ascending rater list once, 64 words
// (4096 users) at a time — the entries of the chunk are OR-ed into 64 LDS words, every word of the row is then written
// exactly once (no memset, no global atomics: 20 M scattered 8-byte atomics were 1 ms at ml-25m shape).
__global__ void __launch_bounds__(TPB)
k_item_bits_rank(int32_t I, int64_t words, const int64_t* __restrict__ i_ptr, const int32_t* __restrict__ it_user,
                 const uint32_t* __restrict__ it_tile, int32_t tile_stride, unsigned long long* __restrict__ bits,
                 uint32_t* __restrict__ rank) {
    // One wave per (item, column tile of SELECT_TCOLS users): it_tile says where the item's ascending rater list enters and
    // leaves the tile, so every wave starts at its own entry.  (One wave per ITEM walked the most-rated item's 65 000 raters
    // 64 at a time behind one dependent load each: that single wave took the whole millisecond of the kernel.)
    __shared__ unsigned long long cell[TPB / 64][64];
    constexpr int WPT = SELECT_TCOLS / 64;  // bitmap words per tile
    const int n_tiles = tile_stride - 1;
    const int64_t task = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int32_t item = (int32_t)(task / n_tiles), tile = (int32_t)(task - (int64_t)item * n_tiles);
    const int lane = threadIdx.x & 63;
    if (item >= I) return;  // (the waves of a block only synchronise with themselves)
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    unsigned long long* my = cell[threadIdx.x >> 6];
    const uint32_t* tb = it_tile + (int64_t)item * tile_stride + tile;
    int64_t q = tb[0];
    const int64_t qe = tb[1];
    unsigned long long* b = bits + (int64_t)item * words;
    uint32_t* r = rank + (int64_t)item * words;
    uint32_t run = (uint32_t)(q - i_ptr[item]);  // raters of the item in earlier tiles
    const int64_t w_end = min(words, (int64_t)(tile + 1) * WPT);
    for (int64_t w0 = (int64_t)tile * WPT; w0 < w_end; w0 += 64) {
        my[lane] = 0;
        wave_sync();
        const int64_t v_end = (w0 + 64) * 64;  // users below v_end belong to this chunk or an earlier (finished) one
        for (;;) {
            const int64_t p = q + lane;
            const int64_t v = p < qe ? (int64_t)it_user[p] : INT64_MAX;
            const bool in = v < v_end;
            if (in) atomicOr(&my[(v >> 6) - w0], 1ull << (v & 63));
            const int c = __popcll(__ballot(in));  // the list ascends: the lanes inside the chunk are a prefix
            q += c;
            if (c < 64) break;
        }
        wave_sync();
        const unsigned long long word = my[lane];
        const uint32_t c = (uint32_t)__popcll(word);
        const uint32_t incl = wave_incl_scan(c);
        const int64_t w = w0 + lane;
        if (w < w_end) {
            b[w] = word;
            r[w] = run + incl - c;
        }
        run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        wave_sync();  // (the words are cleared again at the top)
    }
}

// it_tile[i][t] = lower bound of t * SELECT_TCOLS in item i's ascending rater list (absolute entry index)
__global__ void k_item_tiles(int32_t I, int32_t stride, const int64_t* __restrict__ i_ptr, const int32_t* __restrict__ it_user,
                             uint32_t* __restrict__ it_tile) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)I * stride) return;
    const int32_t i = (int32_t)(g / stride), t = (int32_t)(g - (int64_t)i * stride);
    int64_t lo = i_ptr[i], hi = i_ptr[i + 1];
    const int64_t bound = (int64_t)t * SELECT_TCOLS;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (it_user[mid] < bound) lo = mid + 1;
        else hi = mid;
    }
    it_tile[g] = (uint32_t)lo;
}

__global__ void k_col_keys(int64_t n, const int32_t* __restrict__ s_col, uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    key[p] = (uint32_t)s_col[p];
    val[p] = (uint32_t)p;
}

// key = U - count (an item has at most U raters) so that an ascending stable sort lists the most-rated items first (ties:
// dense order) on bits_for(U) key bits
__global__ void k_pop_keys(int32_t I, int32_t U, const int64_t* __restrict__ i_ptr, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I) return;
    key[i] = (uint64_t)U - (uint64_t)(i_ptr[i + 1] - i_ptr[i]);
    val[i] = (uint32_t)i;
}

// Two parts.  Part A, on the caller's stream: the (item, user ascending) order of the positions, i_ptr and the popularity
// order — what the hybrid similarity's head choice and the operand panel need.  Part B — the 32-byte records, the item-major
// copies, the tile table and the rater bitmaps, 1.2 ms of mostly HBM writes (1.8 GB of bitmaps) — is only read by select.hip
// and predict.hip, so it runs on the fit's second stream beside whatever the caller queues next (the operand panel and the
// similarity GEMM); its consumers call sc.join_commit(stream) first (api.cpp).  Part B reads no shared scratch buffer the
// caller's stream may overwrite meanwhile (perm_iu and rec are its own).
void prep_commit(Train& tr, PrepScratch& sc, hipStream_t st) {
    const int64_t n = tr.n;
    const int32_t I = tr.I;
    // item-major copies + popularity order (hybrid similarity: dense head / sparse tail)
    // (item, user ascending) order: a stable sort of the user-major positions by item
    tr.it_user.alloc(n); tr.it_pack.alloc(n); tr.it_dev.alloc(n); tr.it_t.alloc(n); tr.pop_item.alloc(I);
    sc.k64_a.ensure(std::max<int64_t>(n, I)); sc.k64_b.ensure(std::max<int64_t>(n, I));
    sc.v32_a.ensure(std::max<int64_t>(n, I)); sc.perm_iu.ensure(n);
    sc.k32_a.ensure(n); sc.k32_b.ensure(n);
    sc.rec.ensure(2 * (size_t)n);
    tr.tile_stride = (int32_t)ceil_div(tr.U, SELECT_TCOLS) + 1;
    tr.it_tile.ensure((size_t)I * tr.tile_stride);
    // per-item rater bitmaps + rank prefixes for the prediction probes (skipped when they would not fit)
    const int64_t words = ceil_div(tr.U, 64);
    {
        const double bytes = (double)I * (double)words * 12.0;
        size_t free_b = 0, total_b = 0;
        KN_HIP(hipMemGetInfo(&free_b, &total_b));
        // (KNNCF_DEBUG_NO_ITEM_BITMAPS: test hook, forces the no-bitmap prediction path that huge shapes take)
        if (bytes < 0.15 * (double)(free_b + tr.item_bits.bytes() + tr.item_rank.bytes()) && !getenv("KNNCF_DEBUG_NO_ITEM_BITMAPS")) {
            tr.ib_words = words;
            tr.item_bits.ensure((size_t)I * words);
            tr.item_rank.ensure((size_t)I * words);
        } else {
            tr.ib_words = 0;
        }
    }
    sc.ensure_aux();
    // ---- part A
    k_col_keys<<<nblocks(n), TPB, 0, st>>>(n, tr.s_col.p, sc.k32_a.p, sc.v32_a.p);
    KN_HIP(hipGetLastError());
    sort_pairs_u32_u32(sc.sort, sc.k32_a.p, sc.k32_b.p, sc.v32_a.p, sc.perm_iu.p, n, bits_for(I), st);
    k_segment_ptr_u32<<<nblocks((int64_t)I + 1), TPB, 0, st>>>(n, sc.k32_b.p, I, tr.i_ptr.p);
    KN_HIP(hipGetLastError());
    // ---- part B, forked
    {
        hipStream_t sx = sc.aux;
        KN_HIP(hipEventRecord(sc.ev_fork, st));
        KN_HIP(hipStreamWaitEvent(sx, sc.ev_fork, 0));
        k_pack_records<<<nblocks(n), TPB, 0, sx>>>(n, tr.s_user.p, tr.s_pre.p, tr.s_dev.p, tr.s_t.p, sc.rec.p);
        k_item_major<<<nblocks(n), TPB, 0, sx>>>(n, sc.perm_iu.p, sc.rec.p, tr.it_user.p, tr.it_pack.p, tr.it_dev.p, tr.it_t.p, tr.jaccard ? 1 : 0);
        k_item_tiles<<<nblocks((int64_t)I * tr.tile_stride), TPB, 0, sx>>>(I, tr.tile_stride, tr.i_ptr.p, tr.it_user.p, tr.it_tile.p);
        if (tr.ib_words > 0)
            k_item_bits_rank<<<nblocks((int64_t)I * (tr.tile_stride - 1) * 64), TPB, 0, sx>>>(
                I, words, tr.i_ptr.p, tr.it_user.p, tr.it_tile.p, tr.tile_stride, reinterpret_cast<unsigned long long*>(tr.item_bits.p),
                tr.item_rank.p);
        KN_HIP(hipGetLastError());
        KN_HIP(hipEventRecord(sc.ev_commit, sx));
        sc.commit_pending = true;
    }
    // ---- part A, continued
    k_pop_keys<<<nblocks(I), TPB, 0, st>>>(I, tr.U, tr.i_ptr.p, sc.k64_a.p, sc.v32_a.p);
    KN_HIP(hipGetLastError());
    sort_pairs_u64_u32(sc.sort, sc.k64_a.p, sc.k64_b.p, sc.v32_a.p, reinterpret_cast<uint32_t*>(tr.pop_item.p), I, bits_for((uint64_t)tr.U), st);
    {
        std::vector<uint64_t> hk(I);
        KN_HIP(hipMemcpyAsync(hk.data(), sc.k64_b.p, (size_t)I * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        KN_HIP(hipStreamSynchronize(st));
        tr.pop_count.resize(I);
        for (int32_t i = 0; i < I; ++i) tr.pop_count[i] = (int64_t)tr.U - (int64_t)hk[i];
    }
    KN_HIP(hipGetLastError());
}

// K4, the per-item statistics of the baseline predictors — itemsAvg :134, itemsAvgDev :176-186 (foldLeft over the HashMap: trie
// order of the (user, item) tuple hashes), getItemsAvgDev :336-343 (reduceByKey, modelled in file order).  The reference's kNN
// closures (weightedSumDeviation :489-548, predictor :557-585) never evaluate them, and neither does knncf_fit: the first
// predictor or query that reads them calls this (api.cpp: ensure_item_stats).  Ordered fp64 folds over each item's ratings;
// the two fold orders are stable sorts of the positions.
__global__ void k_item_file_keys(int64_t n, const int32_t* __restrict__ s_col, const uint32_t* __restrict__ s_t, int tbits,
                                 uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    key[p] = ((uint64_t)(uint32_t)s_col[p] << tbits) | s_t[p];
    val[p] = (uint32_t)p;
}
__global__ void k_item_hash_keys(int64_t n, const uint32_t* __restrict__ pos, const int32_t* __restrict__ s_col,
                                 const uint32_t* __restrict__ s_t, const int32_t* __restrict__ users, const int32_t* __restrict__ items,
                                 uint64_t* __restrict__ key) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t p = pos[j], t = s_t[p];
    key[j] = ((uint64_t)(uint32_t)s_col[p] << 32) | tuple_trie_key(users[t], items[t]);
}

void prep_item_stats(Train& tr, PrepScratch& sc, hipStream_t st) {
    const int64_t n = tr.n;
    const int32_t I = tr.I;
    tr.item_avg.alloc(I); tr.item_dev_hash.alloc(I); tr.item_dev_file.alloc(I);
    sc.k64_a.ensure(std::max<int64_t>(n, I)); sc.k64_b.ensure(std::max<int64_t>(n, I));
    sc.v32_a.ensure(std::max<int64_t>(n, I)); sc.v32_b.ensure(n); sc.k32_a.ensure(n);
    sc.dsum.ensure((size_t)I + 2);
    const int tbits = bits_for((uint64_t)n), ibits = bits_for((uint64_t)I);
    uint32_t* perm_if = sc.v32_b.p;  // (item, file row)
    uint32_t* perm_ih = sc.k32_a.p;  // (item, trie key of the tuple; ties in file order)
    k_item_file_keys<<<nblocks(n), TPB, 0, st>>>(n, tr.s_col.p, tr.s_t.p, tbits, sc.k64_a.p, sc.v32_a.p);
    KN_HIP(hipGetLastError());
    sort_pairs_u64_u32(sc.sort, sc.k64_a.p, sc.k64_b.p, sc.v32_a.p, perm_if, n, tbits + ibits, st);
    if (n > 4) {  // a Map of <= 4 entries (Map1..Map4) iterates in insertion = file order (N4)
        k_item_hash_keys<<<nblocks(n), TPB, 0, st>>>(n, perm_if, tr.s_col.p, tr.s_t.p, tr.user_raw.p, tr.item_raw.p, sc.k64_a.p);
        KN_HIP(hipGetLastError());
        sort_pairs_u64_u32(sc.sort, sc.k64_a.p, sc.k64_b.p, perm_if, perm_ih, n, 32 + ibits, st);
    } else {
        perm_ih = perm_if;
    }
    // (each fold is bound by the serial fp64 chain of its longest segment; 16 segments per block spread the long ones)
    fold<false>(tr.i_ptr.p, 0, I, perm_if, tr.s_rating.p, sc.dsum.p, st, 16);
    k_divide_by_count<<<nblocks(I), TPB, 0, st>>>(tr.i_ptr.p, 0, I, sc.dsum.p, tr.item_avg.p);
    fold<false>(tr.i_ptr.p, 0, I, perm_if, tr.s_dev.p, sc.dsum.p, st, 16);
    k_divide_by_count<<<nblocks(I), TPB, 0, st>>>(tr.i_ptr.p, 0, I, sc.dsum.p, tr.item_dev_file.p);
    fold<false>(tr.i_ptr.p, 0, I, perm_ih, tr.s_dev.p, sc.dsum.p, st, 16);
    k_divide_by_count<<<nblocks(I), TPB, 0, st>>>(tr.i_ptr.p, 0, I, sc.dsum.p, tr.item_dev_hash.p);
    KN_HIP(hipGetLastError());
    tr.item_stats_ready = true;
}

}  // namespace knncf
