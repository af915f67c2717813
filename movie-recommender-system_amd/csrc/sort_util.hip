// sort_util.hip — K0 plumbing: a stable LSD radix sort (8-bit digits) and sorted-unique for gfx950, hand-written:
// no library call is left on the path.
//
// One pass = three launches over tiles of TILE = 256 threads x ITEMS keys:
//   k_radix_hist      per-tile digit counts                      -> table[tile][256]
//   k_radix_scan      per digit, exclusive prefix along the tiles -> table (in place) + total[256]   (skipped for <= 64 tiles)
//   k_radix_scatter   ranks the tile's keys (stable), re-orders them in LDS so that every digit's keys leave as one
//                     contiguous run, writes them at digit base + tile prefix + rank
// The in-tile rank is wave64 work: the lanes of a wave that hold the same digit find each other with 8 ballots (one per
// digit bit); the lowest of them bumps the wave's private counter of that digit, everyone takes the old value + the number
// of equal-digit lanes below.  A wave walks its slice of the tile 64 keys at a time in input order and LDS executes a
// wave's instructions in order, so the rank is stable without any atomic.
#include <cstring>

#include "engine.h"

namespace knncf {

namespace {

constexpr int RS_BITS = 8;
constexpr int RS_BINS = 1 << RS_BITS;
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_SELF_SCAN_TILES = 64;

template <class K>
struct RsTile {
    // 12 B (64-bit key + value) x 3072 = 36 KiB of LDS: three workgroups per CU; 8 B x 4096 = 32 KiB: four
    static constexpr int ITEMS = sizeof(K) == 8 ? 12 : 16;
    static constexpr int TILE = RS_THREADS * ITEMS;
};

template <class K>
__device__ __forceinline__ uint32_t digit_of(K key, int shift, uint32_t mask) {
    return (uint32_t)(key >> shift) & mask;
}

// element e of wave w's slice: item j, lane l  ->  tile offset w * (ITEMS * 64) + j * 64 + l  (coalesced, input order = (w, j, l))
template <class K>
__global__ void __launch_bounds__(RS_THREADS) k_radix_hist(const K* __restrict__ keys, int64_t n, int shift, uint32_t mask,
                                                           uint32_t* __restrict__ table) {
    constexpr int ITEMS = RsTile<K>::ITEMS, TILE = RsTile<K>::TILE;
    __shared__ uint32_t cnt[RS_WAVES][RS_BINS];
    for (int i = threadIdx.x; i < RS_WAVES * RS_BINS; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)wave * (ITEMS * 64) + lane;
    K k[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int64_t e = base + j * 64;
        k[j] = e < n ? keys[e] : (K)0;
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if (base + j * 64 < n) atomicAdd(&cnt[wave][digit_of(k[j], shift, mask)], 1u);
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) c += cnt[w][threadIdx.x];
    table[(int64_t)blockIdx.x * RS_BINS + threadIdx.x] = c;
}

// grid = RS_BINS / 16 workgroups of 1024 threads: 16 digits (one 64-byte line of a table row) x 64 slices of the tiles
__global__ void __launch_bounds__(1024) k_radix_scan(uint32_t* __restrict__ table, int64_t n_tiles, uint32_t* __restrict__ total) {
    __shared__ uint32_t part[64][16];
    const int dl = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int d = blockIdx.x * 16 + dl;
    const int64_t per = (n_tiles + 63) / 64;
    const int64_t t0 = (int64_t)slice * per, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    uint32_t s = 0;
    for (int64_t t = t0; t < t1; ++t) s += table[t * RS_BINS + d];
    part[slice][dl] = s;
    __syncthreads();
    uint32_t run = 0, all = 0;
    for (int q = 0; q < 64; ++q) {
        const uint32_t v = part[q][dl];
        if (q < slice) run += v;
        all += v;
    }
    if (slice == 0) total[d] = all;
    for (int64_t t = t0; t < t1; ++t) {
        const uint32_t c = table[t * RS_BINS + d];
        table[t * RS_BINS + d] = run;
        run += c;
    }
}

// SELF_SCAN (few tiles): `table` still holds the tiles' raw counts and every workgroup sums the rows before its own itself —
// the scan launch (12 us of mostly latency) is skipped
template <class K, bool VALUES, bool SELF_SCAN>
__global__ void __launch_bounds__(RS_THREADS) k_radix_scatter(const K* __restrict__ kin, K* __restrict__ kout,
                                                              const uint32_t* __restrict__ vin, uint32_t* __restrict__ vout, int64_t n,
                                                              int shift, uint32_t mask, const uint32_t* __restrict__ table,
                                                              const uint32_t* __restrict__ total, int32_t n_tiles) {
    constexpr int ITEMS = RsTile<K>::ITEMS, TILE = RsTile<K>::TILE;
    __shared__ K s_key[TILE];
    __shared__ uint32_t s_val[VALUES ? TILE : 1];
    __shared__ uint32_t wcnt[RS_WAVES][RS_BINS];  // per wave: running count of every digit, then the wave's base inside the digit's run
    __shared__ uint32_t s_start[RS_BINS];         // first slot of the digit's run in the re-ordered tile
    __shared__ uint32_t s_gofs[RS_BINS];          // global index of slot p of digit d = s_gofs[d] + p  (mod 2^32)
    __shared__ uint32_t s_wsum[RS_WAVES];
    for (int i = threadIdx.x; i < RS_WAVES * RS_BINS; i += RS_THREADS) (&wcnt[0][0])[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const int64_t base = tile0 + (int64_t)wave * (ITEMS * 64) + lane;
    const uint64_t lt = (1ull << lane) - 1ull;
    K k[ITEMS];
    uint32_t v[ITEMS];
    uint32_t rk[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int64_t e = base + j * 64;
        k[j] = e < n ? kin[e] : (K)0;
        v[j] = (VALUES && e < n) ? vin[e] : 0u;
    }
    volatile uint32_t* mine = wcnt[wave];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const bool valid = base + j * 64 < n;
        const uint32_t d = digit_of(k[j], shift, mask);
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RS_BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t below = (uint32_t)__popcll(m & lt);
        uint32_t prev = 0;
        if (valid) prev = mine[d];
        if (valid && below == 0) mine[d] = prev + (uint32_t)__popcll(m);
        rk[j] = prev + below;
    }
    __syncthreads();
    {   // thread d: the waves' counts of digit d -> bases inside the digit's run; the runs' starts; the global offsets
        const int d = threadIdx.x;
        uint32_t c[RS_WAVES], tot = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            c[w] = wcnt[w][d];
            wcnt[w][d] = tot;
            tot += c[w];
        }
        const uint32_t incl = wave_incl_scan(tot);
        if (lane == 63) s_wsum[wave] = incl;
        // digit bases over the whole array: exclusive prefix of total[]
        uint32_t g, before;
        if (SELF_SCAN) {
            g = 0;
            before = 0;
            for (int32_t t = 0; t < n_tiles; ++t) {
                const uint32_t ct = table[(int64_t)t * RS_BINS + d];
                if (t < (int32_t)blockIdx.x) before += ct;
                g += ct;
            }
        } else {
            g = total[d];
            before = table[(int64_t)blockIdx.x * RS_BINS + d];
        }
        const uint32_t gincl = wave_incl_scan(g);
        __shared__ uint32_t s_gsum[RS_WAVES];
        if (lane == 63) s_gsum[wave] = gincl;
        __syncthreads();
        uint32_t off = 0, goff = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w)
            if (w < wave) { off += s_wsum[w]; goff += s_gsum[w]; }
        const uint32_t start = off + incl - tot;
        s_start[d] = start;
        s_gofs[d] = (goff + gincl - g) + before - start;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if (base + j * 64 < n) {
            const uint32_t d = digit_of(k[j], shift, mask);
            const uint32_t pos = s_start[d] + wcnt[wave][d] + rk[j];
            s_key[pos] = k[j];
            if (VALUES) s_val[pos] = v[j];
        }
    }
    __syncthreads();
    const int64_t left = n - tile0;
    const uint32_t here = left < TILE ? (uint32_t)left : (uint32_t)TILE;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t pos = (uint32_t)j * RS_THREADS + threadIdx.x;
        if (pos < here) {
            const K key = s_key[pos];
            const uint32_t g = s_gofs[digit_of(key, shift, mask)] + pos;
            kout[g] = key;
            if (VALUES) vout[g] = s_val[pos];
        }
    }
}

// kin/vin are left intact; the last pass lands in kout/vout, the passes before it alternate between (kout, vout) and the
// workspace's pair.  end_bit = number of significant key bits.
template <class K, bool VALUES>
void radix_sort(SortWorkspace& ws, const K* kin, K* kout, const uint32_t* vin, uint32_t* vout, size_t n, int end_bit, hipStream_t st) {
    if (n == 0) return;
    KN_REQUIRE(n < ((size_t)1 << 32), KNNCF_E_UNSUPPORTED, "sort: more than 2^32-1 keys");
    KN_REQUIRE((const void*)kin != (const void*)kout && (!VALUES || vin != vout), KNNCF_E_INVALID, "sort: in place");
    const int max_bits = (int)sizeof(K) * 8;
    if (end_bit < 1) end_bit = 1;
    if (end_bit > max_bits) end_bit = max_bits;
    const int passes = (end_bit + RS_BITS - 1) / RS_BITS;
    constexpr int TILE = RsTile<K>::TILE;
    const int64_t n_tiles = ceil_div((int64_t)n, TILE);
    const size_t table_bytes = ((size_t)n_tiles * RS_BINS + RS_BINS) * sizeof(uint32_t);
    const size_t key_bytes = passes > 1 ? round_up((int64_t)(n * sizeof(K)), 256) : 0;
    const size_t val_bytes = (passes > 1 && VALUES) ? round_up((int64_t)(n * sizeof(uint32_t)), 256) : 0;
    ws.tmp.ensure(round_up((int64_t)table_bytes, 256) + key_bytes + val_bytes);
    uint32_t* table = reinterpret_cast<uint32_t*>(ws.tmp.p);
    uint32_t* total = table + (size_t)n_tiles * RS_BINS;
    K* ktmp = reinterpret_cast<K*>(ws.tmp.p + round_up((int64_t)table_bytes, 256));
    uint32_t* vtmp = reinterpret_cast<uint32_t*>(ws.tmp.p + round_up((int64_t)table_bytes, 256) + key_bytes);
    const K* src_k = kin;
    const uint32_t* src_v = vin;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * RS_BITS;
        const int bits = end_bit - shift < RS_BITS ? end_bit - shift : RS_BITS;
        const uint32_t mask = (1u << bits) - 1u;
        const bool to_out = ((passes - 1 - p) & 1) == 0;
        K* dst_k = to_out ? kout : ktmp;
        uint32_t* dst_v = to_out ? vout : vtmp;
        k_radix_hist<K><<<(unsigned)n_tiles, RS_THREADS, 0, st>>>(src_k, (int64_t)n, shift, mask, table);
        if (n_tiles <= RS_SELF_SCAN_TILES) {
            k_radix_scatter<K, VALUES, true><<<(unsigned)n_tiles, RS_THREADS, 0, st>>>(src_k, dst_k, src_v, dst_v, (int64_t)n, shift, mask, table, total, (int32_t)n_tiles);
        } else {
            k_radix_scan<<<RS_BINS / 16, 1024, 0, st>>>(table, n_tiles, total);
            k_radix_scatter<K, VALUES, false><<<(unsigned)n_tiles, RS_THREADS, 0, st>>>(src_k, dst_k, src_v, dst_v, (int64_t)n, shift, mask, table, total, (int32_t)n_tiles);
        }
        KN_HIP(hipGetLastError());
        src_k = dst_k;
        src_v = dst_v;
    }
}

// ---- distinct values of a sorted array ---------------------------------------------------------------------------
constexpr int UQ_TILE = 2048;
__global__ void __launch_bounds__(RS_THREADS) k_unique_count(const uint32_t* __restrict__ in, int64_t n, uint32_t* __restrict__ tile_cnt) {
    __shared__ uint32_t s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    uint32_t c = 0;
    for (int j = 0; j < UQ_TILE / RS_THREADS; ++j) {
        const int64_t e = (int64_t)blockIdx.x * UQ_TILE + j * RS_THREADS + threadIdx.x;
        if (e < n && (e == 0 || in[e] != in[e - 1])) ++c;
    }
    if (c) atomicAdd(&s, c);
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s;
}
// one workgroup: exclusive prefix of the tile counts (in place), the grand total behind them
__global__ void __launch_bounds__(1024) k_unique_scan(uint32_t* __restrict__ tile_cnt, int64_t n_tiles) {
    __shared__ uint32_t part[1024];
    const int64_t per = (n_tiles + 1023) / 1024;
    const int64_t t0 = (int64_t)threadIdx.x * per, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    uint32_t s = 0;
    for (int64_t t = t0; t < t1; ++t) s += tile_cnt[t];
    part[threadIdx.x] = s;
    __syncthreads();
    uint32_t run = 0, all = 0;
    for (int q = 0; q < 1024; ++q) {
        const uint32_t v = part[q];
        if (q < (int)threadIdx.x) run += v;
        all += v;
    }
    for (int64_t t = t0; t < t1; ++t) {
        const uint32_t c = tile_cnt[t];
        tile_cnt[t] = run;
        run += c;
    }
    if (threadIdx.x == 0) tile_cnt[n_tiles] = all;
}
// one wave per 64 consecutive elements, a workgroup per tile: the heads keep their order
__global__ void __launch_bounds__(RS_THREADS) k_unique_write(const uint32_t* __restrict__ in, int64_t n, const uint32_t* __restrict__ tile_ofs,
                                                             uint32_t* __restrict__ out) {
    __shared__ uint32_t s_run;
    if (threadIdx.x == 0) s_run = tile_ofs[blockIdx.x];
    __syncthreads();
    for (int j = 0; j < UQ_TILE / RS_THREADS; ++j) {
        const int64_t e = (int64_t)blockIdx.x * UQ_TILE + j * RS_THREADS + threadIdx.x;
        const bool head = e < n && (e == 0 || in[e] != in[e - 1]);
        // the workgroup's 256 elements in order: wave prefix, then the waves one after the other
        __shared__ uint32_t wsum[RS_WAVES];
        const uint32_t incl = wave_incl_scan(head ? 1u : 0u);
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t off = s_run, all = 0;
        for (int w = 0; w < RS_WAVES; ++w) {
            if (w < (int)(threadIdx.x >> 6)) off += wsum[w];
            all += wsum[w];
        }
        if (head) out[off + incl - 1] = in[e];
        __syncthreads();
        if (threadIdx.x == 0) s_run += all;
        __syncthreads();
    }
}

}  // namespace

void sort_pairs_u64_u32(SortWorkspace& ws, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                        uint32_t* vout, size_t n, int end_bit, hipStream_t st) {
    radix_sort<uint64_t, true>(ws, kin, kout, vin, vout, n, end_bit, st);
}

void sort_pairs_u32_u32(SortWorkspace& ws, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n,
                        int end_bit, hipStream_t st) {
    radix_sort<uint32_t, true>(ws, kin, kout, vin, vout, n, end_bit, st);
}

void sort_keys_u32(SortWorkspace& ws, const uint32_t* kin, uint32_t* kout, size_t n, hipStream_t st) {
    radix_sort<uint32_t, false>(ws, kin, kout, nullptr, nullptr, n, 32, st);
}

size_t unique_u32(SortWorkspace& ws, const uint32_t* sorted_in, uint32_t* out, size_t n, hipStream_t st) {
    if (n == 0) return 0;
    const int64_t n_tiles = ceil_div((int64_t)n, UQ_TILE);
    ws.count.ensure((size_t)n_tiles + 1);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(ws.count.p);
    k_unique_count<<<(unsigned)n_tiles, RS_THREADS, 0, st>>>(sorted_in, (int64_t)n, cnt);
    k_unique_scan<<<1, 1024, 0, st>>>(cnt, n_tiles);
    k_unique_write<<<(unsigned)n_tiles, RS_THREADS, 0, st>>>(sorted_in, (int64_t)n, cnt, out);
    KN_HIP(hipGetLastError());
    uint32_t h = 0;
    KN_HIP(hipMemcpyAsync(&h, cnt + n_tiles, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    KN_HIP(hipStreamSynchronize(st));
    return h;
}

}  // namespace knncf
