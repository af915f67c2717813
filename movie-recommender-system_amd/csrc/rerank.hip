// rerank.hip — K6b: exact fp64 similarities of the shortlists IN REFERENCE ORDER, and the stable
// top-k.
//
// adjustedCosineSimilarityFunction shared/predictions.scala:418-426 is a left fold over
// uItems.intersect(vItems) in the Set iteration order of the first argument (memo
// history N6).  Dense item index == rank in that iteration order, so for users with > 4 ratings the
// fold runs over the common items in ascending dense index.  getNeighbors :608-610 sorts
// (allUsers - u).toSeq with sortWith(_._2 > _._2) (stable TimSort, N3) and takes k: total order
// (similarity desc, dense user index asc).
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include <stdio.h>
#include <string.h>

#include <type_traits>

#include "engine.h"

namespace knncf {

static constexpr int TPB = 512;

struct Rows {
    const int64_t* u_ptr;
    const int32_t* s_col;
    const uint32_t* s_t;
    const double* s_pre;
    uint32_t col_bytes, pre_bytes;  // buffer-descriptor extents of s_col / s_pre
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

// scalar path: both rows iterate in trie order (dense item index ascending): the common items are
// visited in ascending order and folded left.  Rows of similar length: two-pointer merge; very
// different lengths: walk the short row and binary-search the long one (same visiting order).
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
    int64_t lo = R.u_ptr[user], e = R.u_ptr[user + 1];
    lo = lower_bound_col(R.s_col, lo, e, col);
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

// jaccardCoefficient :446-463: |I(u) & I(v)| / (|I(u)| + |I(v)| - |I(u) & I(v)|), Double / Int; no summation order involved
__device__ __forceinline__ double jaccard_sim(const Rows& R, int32_t u, int32_t v) {
    int64_t pa = R.u_ptr[u], pb = R.u_ptr[v];
    const int64_t ea = R.u_ptr[u + 1], eb = R.u_ptr[v + 1];
    const int64_t nu = ea - pa, nv = eb - pb;
    int64_t both = 0;
    while (pa < ea && pb < eb) {
        const int32_t ca = R.s_col[pa], cb = R.s_col[pb];
        if (ca == cb) { ++both; ++pa; ++pb; }
        else if (ca < cb) ++pa;
        else ++pb;
    }
    return (double)both / (double)(nu + nv - both);
}

__device__ __forceinline__ bool ranks_before(double sa, int32_t ia, double sb, int32_t ib) {
    return sa > sb || (sa == sb && ia < ib);
}


// the value lane (l ^ STRIDE) holds, without LDS: DPP quad permutes (1, 2) and row rotations (4: both directions + a select,
// 8), v_permlane16_swap / v_permlane32_swap (gfx950) for 16 / 32 — ds_bpermute is an LDS round trip per dword
template <int STRIDE>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u2v;
    if constexpr (STRIDE == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);        // quad_perm:[1,0,3,2]
    else if constexpr (STRIDE == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm:[2,3,0,1]
    else if constexpr (STRIDE == 4) {
        const uint32_t below = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);  // row_ror:4: from lane l - 4
        const uint32_t above = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x12C, 0xF, 0xF, false);  // row_ror:12: from lane l + 4
        return (lane & 4) ? below : above;
    } else if constexpr (STRIDE == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);  // row_ror:8
    else if constexpr (STRIDE == 16) {
        const u2v r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // x: rows 0 0 2 2, y: rows 1 1 3 3
        return (lane & 16) ? r.x : r.y;
    } else {
        static_assert(STRIDE == 32, "lane_xor: stride");
        const u2v r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // x: lower half twice, y: upper half twice
        return (lane & 32) ? r.x : r.y;
    }
}
template <int STRIDE>
__device__ __forceinline__ double lane_xor_f64(double v, int lane) {
    const uint32_t lo = lane_xor<STRIDE>((uint32_t)__double2loint(v), lane), hi = lane_xor<STRIDE>((uint32_t)__double2hiint(v), lane);
    return __hiloint2double((int)hi, (int)lo);
}

// the 512-element bitonic network of k_rerank<512> with one element per thread of a 512-thread workgroup held in registers
// (see the call); asim / aidx: the elements (in: any order, out: best first), bsim / bidx: a second exchange buffer
template <int SIZE, int STRIDE>
__device__ __forceinline__ void sort512_stage(double& sv, int32_t& iv, int32_t e, double* asim, int32_t* aidx, double* bsim, int32_t* bidx) {
    double sp;
    int32_t ip;
    if constexpr (STRIDE >= 64) {
        // exchanges in network order: (128, 64) (256, 128) (256, 64) (512, 256) (512, 128) (512, 64) -> buffers a b a b a b
        constexpr int nth = (SIZE == 128 ? 0 : SIZE == 256 ? 1 : 3) + (SIZE / 2 == STRIDE ? 0 : SIZE / 4 == STRIDE ? 1 : 2);
        double* xs = (nth & 1) ? bsim : asim;
        int32_t* xi = (nth & 1) ? bidx : aidx;
        xs[e] = sv;
        xi[e] = iv;
        __syncthreads();
        sp = xs[e ^ STRIDE];
        ip = xi[e ^ STRIDE];
    } else {
        sp = lane_xor_f64<STRIDE>(sv, e & 63);
        ip = (int32_t)lane_xor<STRIDE>((uint32_t)iv, e & 63);
    }
    const bool is_lo = (e & STRIDE) == 0, up = (e & SIZE) == 0;
    const bool a_first = is_lo ? ranks_before(sv, iv, sp, ip) : ranks_before(sp, ip, sv, iv);
    if (a_first != up) { sv = sp; iv = ip; }
}
template <int SIZE, int STRIDE>
__device__ __forceinline__ void sort512_from(double& sv, int32_t& iv, int32_t e, double* asim, int32_t* aidx, double* bsim, int32_t* bidx) {
    sort512_stage<SIZE, STRIDE>(sv, iv, e, asim, aidx, bsim, bidx);
    if constexpr (STRIDE > 1) sort512_from<SIZE, STRIDE / 2>(sv, iv, e, asim, aidx, bsim, bidx);
    else if constexpr (SIZE < 512) sort512_from<SIZE * 2, SIZE>(sv, iv, e, asim, aidx, bsim, bidx);
}
__device__ __attribute__((noinline)) void sort512_in_registers(double* asim, int32_t* aidx, double* bsim, int32_t* bidx) {
    const int32_t e = threadIdx.x;
    double sv = asim[e];
    int32_t iv = aidx[e];
    sort512_from<2, 1>(sv, iv, e, asim, aidx, bsim, bidx);
    // (the sixth and last exchange went through the second buffer: nobody reads the first any more)
    asim[e] = sv;
    aidx[e] = iv;
    __syncthreads();
}

// ---- K6b kernel ----------------------------------------------------------------------------------
// One workgroup per panel row u.  u's item set lives in LDS as a BITMAP over the dense item index
// plus per-word prefix popcounts, so "does u rate item c, and where" costs one LDS read (two more on a
// hit) instead of a binary search; u's preprocessed ratings sit in LDS too (rows up to UPRE_LDS).
// Each wave takes 64 candidates at a time: one lane per candidate fetches its row extent, then the
// wave streams the candidates' rows as a sequence of 64-entry pieces (coalesced), always with the
// next piece already in flight.  Matching products are compacted (ballot + popcount) into the wave's
// LDS buffer in item order; then lane j folds candidate j's products left in fp64 — 64 independent
// chains in parallel, each performing the same additions in the same order as the reference's `.sum`
// over uItems.intersect(vItems).
// Finally an LDS bitonic sort keeps the best kk; shortlists longer than the LDS tile are consumed
// in chunks: [current best kk | next chunk] is sorted and cut to kk again (exact: total order).
// LDS plan per shortlist tile.  TILE = 512 (k <= 384, the common case): 256-product wave buffers and u's ratings in LDS up to
// 512 make a 49 KiB plan, and 85 VGPRs (launch bounds) let THREE workgroups share a CU instead of two — this kernel waits on
// gathers 60 % of the time, so the extra waves pay: 13.5 -> 11.3 ms at the ml-25m shape.  Larger tiles keep the roomier plan.
template <int TILE, bool WIDE = false> struct RerankPlan {
    // products per wave buffer.  A group of four 64-entry pieces may add 256 products, and the buffer is folded whenever it could
    // not take them: with exactly 256 that is before every group that follows a hit.  WIDE (336) folds only once more than 80
    // products are pending — about every second group at the ml-25m shape — and is taken when three workgroups still fit a CU
    // with it (launch_rerank_tile): 10.17 -> 9.87 ms; 352 no longer fits three workgroups at 59 047 items: 13.0 ms.
    static constexpr int WBUF = TILE <= 512 ? (WIDE ? 336 : 256) : 512;
    static constexpr int UPRE_LDS = TILE <= 512 ? 512 : 1024;  // u's preprocessed ratings are kept in LDS up to this row length
    static constexpr int WAVES_PER_EU = TILE <= 512 ? 6 : 4;
};

typedef const __attribute__((address_space(3))) double* lds_cf64;
typedef __attribute__((address_space(3))) double* lds_f64;
typedef const __attribute__((address_space(3))) uint32_t* lds_cu32;

__device__ __forceinline__ void wave_sync() {
    // lanes of one wave exchange data through LDS: order the accesses for the compiler (the LDS queue
    // itself is in order per wave)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#ifdef KNNCF_RERANK_PROFILE
__device__ unsigned long long g_rphase[8];
#define RPH(i) do { if (threadIdx.x == 0) { const long long now_ = clock64(); atomicAdd(&g_rphase[i], (unsigned long long)(now_ - ph_t)); ph_t = now_; } } while (0)
#else
#define RPH(i) do {} while (0)
#endif
#ifdef KNNCF_RERANK_PROFILE
#define FOLD_T0() long long f_t0 = clock64()
#define FOLD_T1() do { if (threadIdx.x == 0) atomicAdd(&g_rphase[5], (unsigned long long)(clock64() - f_t0)); } while (0)
#else
#define FOLD_T0() do {} while (0)
#define FOLD_T1() do {} while (0)
#endif

// exact similarities of this wave's candidates (n_c <= 64); lane j returns candidate j's.
// The candidates' rows are read as ONE stream: position s of the stream belongs to candidate j with
// cstart[j] <= s < cend[j] (per-wave arrays in LDS; each lane walks them forward as its positions grow by 64 per
// piece), so every lane of every piece carries an entry whatever the row lengths are.  PIPE pieces are in flight;
// they are consumed GRP at a time in stage order (bitmap words of all GRP pieces, then u's values, then the writes)
// so that the dependent LDS round trips of different pieces overlap.  The loop is bound by instruction issue and
// LDS latency, not by the volume read, so the per-piece instruction count is what is minimised here.
// Hits are appended to the wave's product buffer in stream order (= candidate by candidate, items ascending); the
// lane holding a candidate's FIRST entry records where that candidate's products start (cofs), which is all the
// fold needs: candidate j's products are [cofs[j], cofs[j+1]).
static constexpr int PIPE = 8;
static constexpr int GRP = 4;
static constexpr int WMETA = 200;  // per-wave LDS words: (cend, cbase)[64] | cofs[65] (193 used)
static constexpr uint32_t NOT_STARTED = 0xffffffffu;

typedef __attribute__((address_space(3))) uint32_t* lds_u32;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef const __attribute__((address_space(3))) u32x2* lds_cu32x2;
typedef __attribute__((address_space(3))) u32x2* lds_u32x2;

// JAC: every hit contributes 1.0 (the fold then yields the exact number of common items)
template <bool JAC, int WBUF, class PreP>
__device__ __forceinline__ double wave_sims(const Rows& R, lds_cu32x2 bp, PreP upre, int32_t nu, lds_f64 wb, lds_u32 meta,
                                            uint32_t my_b, uint32_t my_len, int n_c, int lane) {
    constexpr bool PRE_IN_LDS = !std::is_same<PreP, const double*>::value;
    lds_u32x2 cpair = (lds_u32x2)meta;  // [64] (end of candidate j in the stream, entry of stream position 0 of j)
    lds_u32 cofs = meta + 128;          // [65] first product of candidate j in the product buffer
    const uint32_t incl = wave_incl_scan(my_len);
    const uint32_t E = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    {
        u32x2 pr;
        pr.x = incl;
        pr.y = my_b - (incl - my_len);  // entry of stream position s: cbase[j] + s
        cpair[lane] = pr;
    }
    cofs[lane] = NOT_STARTED;
    if (lane == 0) cofs[64] = NOT_STARTED;
    wave_sync();
    const uint32_t NP = (E + 63u) >> 6;
    double acc = 0.0;
    uint32_t off = 0;
    // stream cursor of this lane: candidate jf = [c_start, c_end), never moves backwards
    int32_t jf = 0;
    uint32_t c_start = 0, c_end, c_base;
    {
        const u32x2 p0 = cpair[0];
        c_end = p0.x;
        c_base = p0.y;
    }
    uint32_t pc[PIPE];
    u32x2 py[PIPE];
    uint32_t pfl[PIPE];  // bit 0: the entry exists, bit 1: it is the first entry of its candidate; bits 8..: candidate
    // The loads go through buffer descriptors and are issued UNCONDITIONALLY (a lane past the stream's end uses an
    // out-of-range offset and gets 0): with loads under a branch the compiler cannot count what is in flight and
    // waits for everything (vmcnt(0)) before the first use, which serialises the whole pipeline.
    const __amdgpu_buffer_rsrc_t col_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(R.s_col), 0, R.col_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t pre_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(R.s_pre), 0, R.pre_bytes, 0x00020000);
    auto request = [&](int d, uint32_t p) {
        const uint32_t s = (p << 6) + lane;
        const bool valid = s < E;
        if (valid) {
            while (s >= c_end) {
                c_start = c_end;
                ++jf;
                const u32x2 pr = cpair[jf];
                c_end = pr.x;
                c_base = pr.y;
            }
        }
        const uint32_t q = valid ? c_base + s : 0x3fffffffu;  // -> byte offsets beyond both arrays
        pc[d] = __builtin_amdgcn_raw_buffer_load_b32(col_rsrc, (int)(q << 2), 0, 0);  // (no use of the loaded values
        py[d] = __builtin_amdgcn_raw_buffer_load_b64(pre_rsrc, (int)(q << 3), 0, 0);  //  here: a use is a wait)
        pfl[d] = valid ? (((uint32_t)jf << 8) | (s == c_start ? 3u : 1u)) : 0u;
    };
    auto fold = [&]() {  // lane j folds candidate j's products left: the reference's `.sum` order
        FOLD_T0();
        wave_sync();
        const uint32_t o0 = min(cofs[lane], off), o1 = min(cofs[lane + 1], off);
        for (uint32_t t = o0; t < o1; ++t) acc = acc + wb[t];
        if (cofs[lane] != NOT_STARTED) cofs[lane] = 0;  // finished or in progress: whatever follows starts at 0
        wave_sync();
        off = 0;
        FOLD_T1();
    };
#pragma unroll
    for (int d = 0; d < PIPE; ++d) request(d, (uint32_t)d);
    for (uint32_t p0 = 0; p0 < NP; p0 += PIPE) {
#pragma unroll
        for (int g = 0; g < PIPE; g += GRP) {
            if (p0 + g < NP) {  // wave-uniform
                if (off + 64 * GRP > WBUF) fold();
                u32x2 wp[GRP];
                unsigned long long mask[GRP];
                bool hits[GRP];
                double prod[GRP];
#pragma unroll
                for (int k = 0; k < GRP; ++k) wp[k] = bp[pc[g + k] >> 5];  // (an absent entry reads word 0: harmless)
#pragma unroll
                for (int k = 0; k < GRP; ++k) {
                    const uint32_t c = pc[g + k];
                    const bool hit = ((wp[k].x >> (c & 31u)) & pfl[g + k] & 1u) != 0u;
                    mask[k] = __ballot(hit);
                    hits[k] = hit;
                    // position of item c in u's row (for a miss: some position of the row, its value is not used; at most nu,
                    // which an LDS copy of the row may read — the next LDS array — but a global row must not)
                    int32_t idx = (int32_t)(wp[k].y + __popc(wp[k].x & ((1u << (c & 31u)) - 1u)));
                    if (!PRE_IN_LDS) idx = min(idx, nu - 1);
                    prod[k] = JAC ? 1.0 : upre[idx] * __hiloint2double((int)py[g + k].y, (int)py[g + k].x);
                }
#pragma unroll
                for (int k = 0; k < GRP; ++k) {
                    const uint32_t pos = off + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask[k], 0u));
                    if (hits[k]) wb[pos] = prod[k];
                    if (pfl[g + k] & 2u) cofs[pfl[g + k] >> 8] = pos;  // this candidate's products start here
                    off += (uint32_t)__popcll(mask[k]);
                }
#pragma unroll
                for (int k = 0; k < GRP; ++k) request(g + k, p0 + g + k + PIPE);
            }
        }
    }
    fold();
    return acc;
}

template <int TILE, bool JAC, bool WIDE>
__global__ void __launch_bounds__(TPB, RerankPlan<TILE>::WAVES_PER_EU) k_rerank(Rows R, const int64_t* __restrict__ seq, int32_t n_rows,
                                                const int32_t* __restrict__ row_user, int32_t cap,
                                                const int32_t* __restrict__ cand_idx, const float* __restrict__ cand_approx,
                                                const int32_t* __restrict__ cand_cnt, int32_t kk, int32_t kcap,
                                                int32_t* __restrict__ nbr_idx, double* __restrict__ nbr_sim,
                                                int32_t* __restrict__ nbr_cnt, const float* __restrict__ cand_eps,
                                                double* __restrict__ stats, uint32_t* __restrict__ row_entries_out, int32_t words,
                                                const Slices sl) {
    // the first sl.n_heavy rows come as sl.P SLICES of their shortlists each: workgroups [0, n_heavy * P) take slice
    // (g % P) of row g / P and write its best k to row g of sl.part_* (partial lists, merged by k_merge_slices); the
    // workgroups behind them take the other rows whole, as ever
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t part[TPB / 64];
    constexpr int WBUF = RerankPlan<TILE, WIDE>::WBUF, UPRE_LDS = RerankPlan<TILE, WIDE>::UPRE_LDS;
    double* ssim = reinterpret_cast<double*>(smem);            // [TILE]
    double* upre = ssim + TILE;                                // [UPRE_LDS]
    double* wbuf = upre + UPRE_LDS;                            // [4][WBUF]
    int32_t* sidx = reinterpret_cast<int32_t*>(wbuf + (TPB / 64) * WBUF);  // [TILE]
    u32x2* bp = reinterpret_cast<u32x2*>(sidx + TILE);         // [words] (bitmap word of u's items, items of u before it)
    uint32_t* wmeta = reinterpret_cast<uint32_t*>(bp + words);  // [TPB / 64][WMETA]
    __shared__ unsigned long long s_entries;
    const int32_t g = blockIdx.x;
    const int32_t n_sliced = sl.n_heavy * sl.P;
    const bool sliced = g < n_sliced;
    const int32_t r = sliced ? g / sl.P : g - n_sliced + sl.n_heavy;
    if (r >= n_rows) return;
    uint32_t* entries_out = sliced ? sl.entries + g : row_entries_out + r;
    int32_t cnt = cand_cnt[r];
    if (cnt > cap) {  // overflow: the exact fallback redoes this row
        if (threadIdx.x == 0) *entries_out = 0;
        return;
    }
    int64_t cbase = (int64_t)r * cap;
    if (sliced) {
        const int32_t len = (cnt + sl.P - 1) / sl.P;
        const int32_t c0 = min(cnt, (g - r * sl.P) * len);
        cbase += c0;
        cnt = min(len, cnt - c0);
    }
    if (threadIdx.x == 0) s_entries = 0;
#ifdef KNNCF_RERANK_PROFILE
    long long ph_t = clock64();
#endif
    const int32_t u = row_user[r];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ub = R.u_ptr[u];
    const int32_t nu = (int32_t)(R.u_ptr[u + 1] - ub);
    const bool pre_lds = nu <= UPRE_LDS;
    // the wave's candidates of the first trip and their row extents: a chain of three dependent loads (count -> ids ->
    // extents) that is requested here so that it runs behind the set-up of u's bitmap instead of in front of the stream
    const int32_t* my_cand = cand_idx + cbase;
    int32_t v_first = 0;
    uint32_t b_first = 0, len_first = 0;
    {
        const int32_t take0 = min(TILE, cnt);
        const int32_t in_trip = min(TPB, take0);
        const int32_t per = (in_trip + TPB / 64 - 1) / (TPB / 64);
        const int32_t c0 = wave * per;
        const int n_c = max(0, min(per, in_trip - c0));
        if (lane < n_c) {
            v_first = my_cand[c0 + lane];
            b_first = (uint32_t)R.u_ptr[v_first];  // n < 2^29 (checked at fit)
            len_first = (uint32_t)R.u_ptr[v_first + 1] - b_first;
        }
    }
    // u's items as a bitmap + exclusive prefix popcounts
    for (int32_t w = threadIdx.x; w < words; w += TPB) bp[w] = u32x2{0u, 0u};
    __syncthreads();
    for (int32_t j = threadIdx.x; j < nu; j += TPB) {
        const int32_t c = R.s_col[ub + j];
        atomicOr(reinterpret_cast<uint32_t*>(bp) + 2 * (c >> 5), 1u << (c & 31));
        if (pre_lds) upre[j] = R.s_pre[ub + j];
    }
    __syncthreads();
    const int32_t per = (words + TPB - 1) / TPB;
    const int32_t w0 = min(words, (int32_t)threadIdx.x * per), w1 = min(words, w0 + per);
    uint32_t mine = 0;
    for (int32_t w = w0; w < w1; ++w) mine += __popc(bp[w].x);
    // exclusive prefix over the threads: DPP scan inside the wave, the waves' totals through LDS (one barrier)
    const uint32_t incl_w = wave_incl_scan(mine);
    if (lane == 63) part[wave] = incl_w;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += part[w];
    uint32_t run = before + incl_w - mine;
    for (int32_t w = w0; w < w1; ++w) {
        bp[w].y = run;
        run += __popc(bp[w].x);
    }
    const int64_t seq_u = seq[u];
    const float eps = cand_eps[r];  // the band select.hip used for this row (KNNCF_FLAG_VERIFY_BOUND)
    lds_f64 wb = (lds_f64)(wbuf + wave * WBUF);
    lds_u32 meta = (lds_u32)(wmeta + wave * WMETA);
    double worst = -1.0;
    int64_t row_entries = 0;
    int32_t best = 0, pos = 0;
    __syncthreads();
    RPH(0);  // bitmap + prefix
    do {
        const int32_t take = min(TILE - best, cnt - pos);
        int32_t m = 1;
        while (m < best + take) m <<= 1;
        // the candidates of a trip (up to 64 per wave) are dealt evenly to the waves
        for (int32_t t0 = 0; t0 < take; t0 += TPB) {
            const int32_t in_trip = min(TPB, take - t0);
            const int32_t per = (in_trip + TPB / 64 - 1) / (TPB / 64);
            const int32_t c0 = t0 + wave * per;
            const int n_c = max(0, min(per, t0 + in_trip - c0));
            if (n_c == 0) continue;
            int32_t v = v_first;
            uint32_t b_v = b_first, len_v = len_first;
            if (pos + t0 > 0) {  // (later trips / rounds: rare for k <= 512)
                v = (lane < n_c) ? my_cand[pos + c0 + lane] : 0;
                b_v = (lane < n_c) ? (uint32_t)R.u_ptr[v] : 0u;
                len_v = (lane < n_c) ? (uint32_t)R.u_ptr[v + 1] - b_v : 0u;
            }
            double s;
            // Set1..Set4 iterate in file order and the memo history matters (N2, N6): scalar path
            row_entries += len_v;  // algorithmic traffic of this kernel: the candidates' rows (12 B per entry)
            const bool small_v = (lane < n_c) && (len_v <= 4);
            if (JAC) {  // exact count of common items (no order involved, any row length), then :461
                const double both = wave_sims<true, WBUF>(R, (lds_cu32x2)bp, (lds_cf64)upre, nu, wb, meta, b_v, len_v, n_c, lane);
                s = both / (double)((int64_t)nu + (int64_t)len_v - (int64_t)both);
            } else if (nu > 4 && !__any(small_v)) {
                s = pre_lds ? wave_sims<false, WBUF>(R, (lds_cu32x2)bp, (lds_cf64)upre, nu, wb, meta, b_v, len_v, n_c, lane)
                            : wave_sims<false, WBUF>(R, (lds_cu32x2)bp, R.s_pre + ub, nu, wb, meta, b_v, len_v, n_c, lane);
            } else {
                s = (lane < n_c) ? pair_sim(R, u, v, seq_u, seq[v]) : 0.0;
            }
            if (lane < n_c) {
                ssim[best + c0 + lane] = s;
                sidx[best + c0 + lane] = v;
                if (cand_approx) worst = fmax(worst, fabs((double)cand_approx[cbase + pos + c0 + lane] - s) - (double)eps);
            }
        }
        RPH(1);  // exact similarities (this wave)
        // TILE = 512: one element per thread, the whole 512-element network (45 stages) with the element in REGISTERS — the 39
        // stages whose partner sits in the same wave exchange through DPP / v_permlane*_swap, the 6 with strides 64 / 128 / 256 through
        // LDS (two buffers in turn — the shortlist arrays and the idle product buffers — so that one barrier per stage is
        // enough).  The all-LDS network below spent ~350 cycles per stage on LDS round trips with half of the threads idle
        // (15 % of the kernel by its phase counters).
        constexpr bool REG_SORT = TILE == 512 && TPB == 512 && (TPB / 64) * WBUF * 8 >= TILE * 12;
        const int32_t m_pad = REG_SORT ? TILE : m;
        for (int32_t c = take + threadIdx.x; c < m_pad - best; c += TPB) {
            ssim[best + c] = -INFINITY;
            sidx[best + c] = 0x7fffffff;
        }
        __syncthreads();
        RPH(2);  // wait for the other waves
        if constexpr (REG_SORT) {
            sort512_in_registers(ssim, sidx, wbuf, reinterpret_cast<int32_t*>(wbuf + TILE));
        } else {
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
                    // a stage with stride <= 64 only moves data inside the 128-element blocks a wave owns (64 consecutive
                    // pairs), so consecutive such stages need no workgroup barrier between them: 6 instead of 45 for m = 512
                    const int32_t next_stride = stride > 1 ? (stride >> 1) : (((size << 1) <= m) ? size : 0);
                    if (stride > 64 || next_stride > 64 || next_stride == 0) __syncthreads();
                    else wave_sync();
                }
            }
        }
        best = min(kk, best + take);
        pos += take;
        RPH(3);  // sort
    } while (pos < cnt);
    // (one global atomic per wave on a single address throttled the whole kernel: -4.8 ms of 24.5 without it.  The
    // counts go to a per-row array instead and are summed by k_sum_row_entries.)
    for (int o = 32; o > 0; o >>= 1) row_entries += __shfl_xor(row_entries, o);
    if (lane == 0) atomicAdd(&s_entries, (unsigned long long)row_entries);  // LDS
    __syncthreads();
    if (threadIdx.x == 0) *entries_out = (uint32_t)min(s_entries, 0xffffffffull);
    if (cand_approx) {
        // max over the grid of (|approx - exact| - eps); must stay <= 0
        for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o));
        if (lane == 0 && worst > -1.0) {
            unsigned long long* w = reinterpret_cast<unsigned long long*>(stats);
            double shifted = worst + 4.0;  // positive, so the bit pattern orders like the value
            atomicMax(w, (unsigned long long)__double_as_longlong(shifted));
        }
    }
    int32_t* out_idx = sliced ? sl.part_idx + (int64_t)g * kcap : nbr_idx + (int64_t)u * kcap;
    double* out_sim = sliced ? sl.part_sim + (int64_t)g * kcap : nbr_sim + (int64_t)u * kcap;
    for (int32_t j = threadIdx.x; j < best; j += TPB) {
        out_idx[j] = sidx[j];
        out_sim[j] = ssim[j];
    }
    if (threadIdx.x == 0) *(sliced ? sl.part_cnt + g : nbr_cnt + u) = best;
    RPH(4);  // output
}

#ifdef KNNCF_RERANK_PROFILE
static void rerank_profile_dump() {
    unsigned long long h[8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rphase), sizeof(h));
    unsigned long long tot = 0;
    for (int i = 0; i < 8; ++i) tot += h[i];
    fprintf(stderr, "[rerank profile] total %.3e cycles:", (double)tot);
    for (int i = 0; i <= 4; ++i) fprintf(stderr, " p%d %.1f%%", i, 100.0 * (double)h[i] / (double)(tot - h[5]));
    fprintf(stderr, " | folds inside p1: %.1f%%", 100.0 * (double)h[5] / (double)(tot - h[5]));
    fprintf(stderr, "\n");
    memset(h, 0, sizeof(h));
    hipMemcpyToSymbol(HIP_SYMBOL(g_rphase), h, sizeof(h));
}
#endif

// stats[1] += sum of the per-row candidate-entry counts (the kernel's algorithmic traffic / 12 B), stats[2] += the shortlist
// lengths min(cand_cnt, cap), stats[3] += the rows whose shortlist overflowed or whose anticipated threshold overshot
// (cand_cnt > cap: the rows api.cpp rebuilds) — one atomic per block and word, so that the host reads four words instead of
// walking 162 541 counts between the re-rank and the prediction
__global__ void __launch_bounds__(1024) k_sum_row_entries(int32_t n_rows, const uint32_t* __restrict__ row_entries, const int32_t* __restrict__ cand_cnt,
                                                          int32_t cap, unsigned long long* __restrict__ total) {
    __shared__ unsigned long long part[3];
    if (threadIdx.x < 3) part[threadIdx.x] = 0;
    __syncthreads();
    const int32_t r = blockIdx.x * 1024 + threadIdx.x;
    unsigned long long v = r < n_rows ? row_entries[r] : 0ull;
    const int32_t c = r < n_rows ? cand_cnt[r] : 0;
    unsigned long long len = (unsigned long long)min(c, cap), over = c > cap ? 1ull : 0ull;
    for (int o = 32; o > 0; o >>= 1) {
        v += __shfl_xor(v, o);
        len += __shfl_xor(len, o);
        over += __shfl_xor(over, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (v) atomicAdd(&part[0], v);
        if (len) atomicAdd(&part[1], len);
        if (over) atomicAdd(&part[2], over);
    }
    __syncthreads();
    if (threadIdx.x < 3 && part[threadIdx.x]) atomicAdd(total + threadIdx.x, part[threadIdx.x]);
}

template <int TILE, bool JAC, bool WIDE>
static void launch_rerank_plan(const Rows& R, const Train& tr, NeighborTable& nt, int32_t n_rows, const int32_t* d_row_user,
                               int32_t cap, const int32_t* cand_idx, const float* cand_approx, const int32_t* cand_cnt,
                               const float* cand_eps, double* d_stats, uint32_t* d_row_entries, hipStream_t st, const Slices& sl) {
    const int32_t words = (int32_t)ceil_div(tr.I, 32);
    constexpr int WBUF = RerankPlan<TILE, WIDE>::WBUF, UPRE_LDS = RerankPlan<TILE, WIDE>::UPRE_LDS;
    const size_t smem = (size_t)TILE * 8 + (size_t)UPRE_LDS * 8 + (size_t)(TPB / 64) * WBUF * 8 + (size_t)TILE * 4 +
                        (size_t)words * 8 + (size_t)(TPB / 64) * WMETA * 4;
    KN_REQUIRE(smem <= 160 * 1024 - 2048, KNNCF_E_UNSUPPORTED, "re-rank: item bitmap does not fit in LDS (too many items)");
    static PerDeviceState lds_state;
    ensure_dynamic_lds(lds_state, (const void*)k_rerank<TILE, JAC, WIDE>, smem);
    const int32_t grid = n_rows + sl.n_heavy * (sl.P - 1);
    k_rerank<TILE, JAC, WIDE><<<grid, TPB, smem, st>>>(R, nt.seq.p, n_rows, d_row_user, cap, cand_idx, cand_approx, cand_cnt, nt.kcap,
                                            nt.kcap, nt.idx.p, nt.sim.p, nt.cnt.p, cand_eps, d_stats, d_row_entries, words, sl);
    k_sum_row_entries<<<(unsigned)ceil_div(n_rows, 1024), 1024, 0, st>>>(n_rows, d_row_entries, cand_cnt, cap, reinterpret_cast<unsigned long long*>(d_stats) + 1);
    KN_HIP(hipGetLastError());
#ifdef KNNCF_RERANK_PROFILE
    KN_HIP(hipStreamSynchronize(st));
    rerank_profile_dump();
#endif
}

template <int TILE, bool JAC>
static void launch_rerank_tile(const Rows& R, const Train& tr, NeighborTable& nt, int32_t n_rows, const int32_t* d_row_user,
                               int32_t cap, const int32_t* cand_idx, const float* cand_approx, const int32_t* cand_cnt,
                               const float* cand_eps, double* d_stats, uint32_t* d_row_entries, hipStream_t st, const Slices& sl) {
    if constexpr (TILE <= 512) {
        // the wide product buffers only where three workgroups still share a CU with them (160 KiB of LDS, 512-byte granules)
        const size_t words = (size_t)ceil_div(tr.I, 32);
        const size_t wide = (size_t)TILE * 12 + (size_t)RerankPlan<TILE, true>::UPRE_LDS * 8 + (size_t)(TPB / 64) * RerankPlan<TILE, true>::WBUF * 8 +
                            words * 8 + (size_t)(TPB / 64) * WMETA * 4 + 64;
        if (3 * round_up((int64_t)wide, 512) <= 160 * 1024) {
            launch_rerank_plan<TILE, JAC, true>(R, tr, nt, n_rows, d_row_user, cap, cand_idx, cand_approx, cand_cnt, cand_eps, d_stats, d_row_entries, st, sl);
            return;
        }
    }
    launch_rerank_plan<TILE, JAC, false>(R, tr, nt, n_rows, d_row_user, cap, cand_idx, cand_approx, cand_cnt, cand_eps, d_stats, d_row_entries, st, sl);
}

// ---- heavy rows as slices ---------------------------------------------------------------------------------------------
// One workgroup per row is the wrong grain for the heaviest rows: their candidates are heavy raters too (up to 33 x the median
// row's work at the ml-25m shape, scripts/analysis/row_work_profile.py), and one such row outlasts a sharded handle's whole
// launch.  The first n_heavy rows of a launch (rows are ordered longest first) are therefore re-ranked as P SLICES of their
// shortlists: P workgroups each return the best k of their slice — the best k of the whole shortlist is the best k of the
// union of those — and k_merge_slices sorts the P partial lists into the row's final list with the same total order.
template <int MT>  // entries sorted at a time (a power of two >= P * k)
__global__ void __launch_bounds__(256) k_merge_slices(int32_t kk, int32_t kcap, int32_t cap, const int32_t* __restrict__ row_user,
                                                      const int32_t* __restrict__ cand_cnt, const Slices sl,
                                                      uint32_t* __restrict__ row_entries_out, int32_t* __restrict__ nbr_idx,
                                                      double* __restrict__ nbr_sim, int32_t* __restrict__ nbr_cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ssim = reinterpret_cast<double*>(smem);        // [MT]
    int32_t* sidx = reinterpret_cast<int32_t*>(ssim + MT);  // [MT]
    const int32_t r = blockIdx.x, P = sl.P;
    if (r >= sl.n_heavy) return;
    if (threadIdx.x == 0) {
        uint64_t e = 0;
        for (int32_t p = 0; p < P; ++p) e += sl.entries[r * P + p];
        row_entries_out[r] = (uint32_t)min(e, (uint64_t)0xffffffffull);
    }
    if (cand_cnt[r] > cap) return;  // (overflow: the exact fallback writes this row)
    const int32_t u = row_user[r];
    int32_t total = 0;
    for (int32_t p = 0; p < P; ++p) {
        const int32_t c = sl.part_cnt[r * P + p];
        for (int32_t j = threadIdx.x; j < c; j += 256) {
            ssim[total + j] = sl.part_sim[(int64_t)(r * P + p) * kcap + j];
            sidx[total + j] = sl.part_idx[(int64_t)(r * P + p) * kcap + j];
        }
        total += c;
    }
    for (int32_t j = total + threadIdx.x; j < MT; j += 256) {
        ssim[j] = -INFINITY;
        sidx[j] = 0x7fffffff;
    }
    __syncthreads();
    for (int32_t size = 2; size <= MT; size <<= 1)
        for (int32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (int32_t t = threadIdx.x; t < (MT >> 1); t += 256) {
                const int32_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const double sa = ssim[lo], sb = ssim[hi];
                const int32_t ia = sidx[lo], ib = sidx[hi];
                if (ranks_before(sa, ia, sb, ib) != up) {
                    ssim[lo] = sb; ssim[hi] = sa;
                    sidx[lo] = ib; sidx[hi] = ia;
                }
            }
            __syncthreads();
        }
    const int32_t best = min(kk, total);
    for (int32_t j = threadIdx.x; j < best; j += 256) {
        nbr_idx[(int64_t)u * kcap + j] = sidx[j];
        nbr_sim[(int64_t)u * kcap + j] = ssim[j];
    }
    if (threadIdx.x == 0) nbr_cnt[u] = best;
}

template <int MT>
static void launch_merge(int32_t cap, const int32_t* d_row_user, const int32_t* cand_cnt, const Slices& sl, uint32_t* d_row_entries,
                         NeighborTable& nt, hipStream_t st) {
    static PerDeviceState lds_state;
    const size_t smem = (size_t)MT * 12;
    ensure_dynamic_lds(lds_state, (const void*)k_merge_slices<MT>, smem);
    k_merge_slices<MT><<<sl.n_heavy, 256, smem, st>>>(nt.kcap, nt.kcap, cap, d_row_user, cand_cnt, sl, d_row_entries, nt.idx.p, nt.sim.p, nt.cnt.p);
    KN_HIP(hipGetLastError());
}

// n_heavy > 0: the first n_heavy rows as `slices` slices each (2 <= slices, slices * k <= 8192; sc holds the partial lists)
void launch_rerank(const Train& tr, NeighborTable& nt, int32_t n_rows, const int32_t* d_row_user, int32_t cap,
                   const int32_t* cand_idx, const float* cand_approx, const int32_t* cand_cnt, const float* cand_eps,
                   double* d_stats, uint32_t* d_row_entries, bool verify, hipStream_t st, int32_t n_heavy, int32_t slices,
                   SliceScratch* sc) {
    if (n_rows <= 0) return;
    // (the shortlist tile must hold the running best k and the next chunk: 2 k; a 4096-entry tile is 113 KiB of LDS at the
    // ml-25m shape's 59 047 items — the wall is the CU's 160 KiB, not a design limit: predict/kNN.scala:73 goes to k = 943)
    KN_REQUIRE(nt.kcap <= 2048, KNNCF_E_UNSUPPORTED, "k > 2048 is not supported by the re-rank kernel (its shortlist tile lives in LDS)");
    Slices sl{0, 1, nullptr, nullptr, nullptr, nullptr};
    if (n_heavy > 0) {
        KN_REQUIRE(sc != nullptr && slices >= 2 && (int64_t)slices * nt.kcap <= 8192 && n_heavy <= n_rows, KNNCF_E_INVALID, "re-rank slices: bad arguments");
        const size_t nv = (size_t)n_heavy * (size_t)slices, cells = nv * (size_t)std::max(nt.kcap, 1);
        sc->entries.ensure(nv); sc->part_cnt.ensure(nv); sc->part_idx.ensure(cells); sc->part_sim.ensure(cells);
        sl = Slices{n_heavy, slices, sc->part_idx.p, sc->part_sim.p, sc->part_cnt.p, sc->entries.p};
    }
    Rows R{tr.u_ptr.p, tr.s_col.p, tr.s_t.p, tr.s_pre.p, (uint32_t)(tr.n * 4), (uint32_t)(tr.n * 8)};
    const float* apx = verify ? cand_approx : nullptr;
#define KN_RERANK(TILEV, JACV) launch_rerank_tile<TILEV, JACV>(R, tr, nt, n_rows, d_row_user, cap, cand_idx, apx, cand_cnt, cand_eps, d_stats, d_row_entries, st, sl)
    if (tr.jaccard) {
        if (nt.kcap <= 384) KN_RERANK(512, true);
        else if (nt.kcap <= 512) KN_RERANK(1024, true);
        else if (nt.kcap <= 1024) KN_RERANK(2048, true);
        else KN_RERANK(4096, true);
    } else {
        if (nt.kcap <= 384) KN_RERANK(512, false);
        else if (nt.kcap <= 512) KN_RERANK(1024, false);
        else if (nt.kcap <= 1024) KN_RERANK(2048, false);
        else KN_RERANK(4096, false);
    }
#undef KN_RERANK
    if (n_heavy > 0) {
        int32_t mt = 512;
        while (mt < slices * nt.kcap) mt <<= 1;
        if (mt <= 512) launch_merge<512>(cap, d_row_user, cand_cnt, sl, d_row_entries, nt, st);
        else if (mt <= 1024) launch_merge<1024>(cap, d_row_user, cand_cnt, sl, d_row_entries, nt, st);
        else if (mt <= 2048) launch_merge<2048>(cap, d_row_user, cand_cnt, sl, d_row_entries, nt, st);
        else if (mt <= 4096) launch_merge<4096>(cap, d_row_user, cand_cnt, sl, d_row_entries, nt, st);
        else launch_merge<8192>(cap, d_row_user, cand_cnt, sl, d_row_entries, nt, st);
    }
}

// exact similarities of one user against everyone (out[user] = -inf): the fallback for rows whose
// shortlist overflowed and the engine behind scalar queries
__global__ void k_exact_row(Rows R, const int64_t* __restrict__ seq, int32_t U, int32_t user, int64_t user_seq,
                            double* __restrict__ out, int jaccard) {
    int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= U) return;
    out[v] = (v == user) ? -INFINITY : (jaccard ? jaccard_sim(R, user, v) : pair_sim(R, user, v, user_seq, seq[v]));
}

void launch_exact_row(const Train& tr, const NeighborTable& nt, int32_t user, int64_t user_seq, double* d_out,
                      hipStream_t st) {
    Rows R{tr.u_ptr.p, tr.s_col.p, tr.s_t.p, tr.s_pre.p, (uint32_t)(tr.n * 4), (uint32_t)(tr.n * 8)};
    k_exact_row<<<(unsigned)ceil_div(tr.U, TPB), TPB, 0, st>>>(R, nt.seq.p, tr.U, user, user_seq, d_out, tr.jaccard ? 1 : 0);
    KN_HIP(hipGetLastError());
}

// ---- Personalized (no k): weightedSumDeviation(train, sim) with sim = the adjusted cosine or the Jaccard coefficient
// itself (predict/Personalized.scala:61-72).  The "neighbour list" of u is every user x with sim(u, x) != 0 — u itself
// included: when (u, i) is also a training pair, u is one of i's raters and contributes sim(u, u) — in ascending dense id,
// which is what the prediction kernels stream.  O(U^2) pairs: the reference only runs it at ml-100k scale; so does this.
// Cosine: every user must have more than 4 ratings (checked by the caller), so the summation order does not depend on
// the memo history and merge_dot is the reference's sum.
template <bool JACCARD>
__global__ void __launch_bounds__(TPB) k_full_rows(Rows R, int32_t U, int32_t* __restrict__ out_idx, double* __restrict__ out_sim,
                                                   int32_t* __restrict__ out_cnt) {
    __shared__ int32_t wsum[TPB / 64];
    __shared__ int32_t s_base;
    const int32_t u = blockIdx.x;
    if (u >= U) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    const int64_t pa0 = R.u_ptr[u], ea = R.u_ptr[u + 1];
    for (int32_t v0 = 0; v0 < U; v0 += TPB) {
        const int32_t v = v0 + threadIdx.x;
        double s = 0.0;
        if (v < U) {
            if (JACCARD) {
                s = jaccard_sim(R, u, v);
            } else {
                s = merge_dot(R, u, v);
            }
        }
        // ordered compaction of the non-zero similarities of this chunk (ids ascending)
        const bool keep = v < U && s != 0.0;
        const unsigned long long km = __ballot(keep);
        if (lane == 0) wsum[wave] = __popcll(km);
        __syncthreads();
        int32_t before = s_base;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (keep) {
            const int64_t pos = (int64_t)u * U + before + __popcll(km & ((1ull << lane) - 1ull));
            out_idx[pos] = v;
            out_sim[pos] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t tot = 0;
            for (int w = 0; w < TPB / 64; ++w) tot += wsum[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out_cnt[u] = s_base;
}

void launch_full_rows(const Train& tr, bool jaccard, int32_t* d_idx, double* d_sim, int32_t* d_cnt, hipStream_t st) {
    Rows R{tr.u_ptr.p, tr.s_col.p, tr.s_t.p, tr.s_pre.p, (uint32_t)(tr.n * 4), (uint32_t)(tr.n * 8)};
    if (jaccard) k_full_rows<true><<<tr.U, TPB, 0, st>>>(R, tr.U, d_idx, d_sim, d_cnt);
    else k_full_rows<false><<<tr.U, TPB, 0, st>>>(R, tr.U, d_idx, d_sim, d_cnt);
    KN_HIP(hipGetLastError());
}

__global__ void k_exact_pair(Rows R, int32_t u, int32_t v, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = owner_dot(R, u, v);
}

void launch_exact_pair(const Train& tr, int32_t u, int32_t v, double* d_out, hipStream_t st) {
    Rows R{tr.u_ptr.p, tr.s_col.p, tr.s_t.p, tr.s_pre.p, (uint32_t)(tr.n * 4), (uint32_t)(tr.n * 8)};
    k_exact_pair<<<1, 64, 0, st>>>(R, u, v, d_out);
    KN_HIP(hipGetLastError());
}

}  // namespace knncf
