// select.hip — K5 sparse tail + K6 top-k neighbour select, fused per similarity row.
//
// The dense MFMA GEMM (gemm.hip) leaves S[u][v] = sum over the H most-rated items (fp16 or fp32
// storage).  For row u this kernel (one 512-thread workgroup per row, two per CU) walks the row ONCE, a column tile
// (16 384 columns = 64 KiB of LDS) at a time; the panel entries of tile t+1 are requested before tile t is
// processed:
//   1. SPARSE TAIL: for every tail item i rated by u and every rater v of i inside the tile,
//      pre(u,i) * pre(v,i) is accumulated in Q7.24 fixed point with integer LDS atomics.  The rater lists
//      are sorted by user and prep.hip tabulates where each list crosses a tile boundary (it_tile), so the
//      tile's work is a set of [begin, end) ranges; they are cut into 64-rater pieces and the pieces are
//      dealt evenly to the 8 waves (a prefix sum over the entries), several pieces in flight per wave;
//   2. the tile's final values S + tail stay in registers; every GROUP of 8 columns whose maximum reaches the
//      threshold known so far is stored whole in a per-row provisional store in global memory (gcap groups, scaled with
//      k).  The threshold — lower edge of the bin that holds the k-th largest value of a 1024-bin histogram, minus 2 eps —
//      is bootstrapped from the first tile at an ANTICIPATED rank (see rank_after below: a guess that the final, exact
//      threshold verifies) and refreshed from the stored values (after 16 and 32 tiles and whenever the store has grown);
//      a loose one merely lets more through;
//   3. after the last tile the threshold is final and the stored values are sorted out once into the shortlist.
// With |S[u][v] - s_uv| <= eps for every pair, every true top-k member v satisfies
// S[u][v] >= a_k - 2 eps (a_k = k-th largest value of the row), so the shortlist provably contains
// the exact top-k; rerank.hip decides.  The panel S is read exactly once (it was written by the GEMM: the round trip
// through HBM is the cost of keeping GEMM and select separate kernels); the kernel itself is bound by
// VALU issue in the tail drain, not by that traffic.
#include <math.h>

#include <type_traits>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "engine.h"

namespace knncf {

#ifdef KNNCF_SELECT_PROFILE
__device__ unsigned long long g_phase[16];
#define PH(i) do { if (threadIdx.x == 0) { const long long now_ = clock64(); atomicAdd(&g_phase[i], (unsigned long long)(now_ - ph_t)); ph_t = now_; } } while (0)
#else
#define PH(i) do {} while (0)
#endif

// Two 512-thread workgroups per CU, 16 384-column tiles (64 KiB each): the phases of a tile are separated by
// barriers and neither workgroup can fill the CU alone; one 1024-thread workgroup with 32 768-column tiles measured
// 10 % slower (and a 48 KiB-tile variant of an earlier version of this kernel 1.7x slower: tiles cost per tile).
// __launch_bounds__(2 * TPB) caps the kernel at 128 VGPRs so that both workgroups fit.
static constexpr int TPB = 512;
#ifndef KNNCF_REFRESH_MASK
#define KNNCF_REFRESH_MASK 0x80008000u  // tiles after which the threshold is refreshed from the stored values (A/B switch): after
                                        // 16 and 32 tiles — none at the ml-25m shape's ten tiles: with the anticipated thresholds
                                        // a refresh there costs more than the groups it saves; the store's fill
                                        // level still triggers one whenever it grows
#endif
static constexpr int NBINS = 1024;
static constexpr int TCOLS = SELECT_TCOLS;  // columns of the row held in LDS at a time (64 KiB)
static constexpr int CPT = TCOLS / TPB;  // columns per thread per tile (32 = 4 groups of 8)
static constexpr int NG = CPT / 8;
static_assert(CPT <= 32, "the survivor mask of a thread is one 32-bit word");
static constexpr int EMAX = 256;     // row positions whose tail entries (16 B each) are held in LDS at a time
#ifndef KNNCF_PMAX
#define KNNCF_PMAX 1024
#endif
static constexpr int PMAX = KNNCF_PMAX;    // pieces per (chunk, tile) with a direct piece -> entry table in LDS (A/B switch: 0 = always the binary search)
static constexpr int MAXT = TCOLS >= 16384 ? 12 : 24;  // tiles whose per-entry rater counts are packed into registers (12 x 16 384 columns: up to 196 608 users)
static constexpr int WAVES_PER_EU = TCOLS >= 16384 ? 4 : 6;  // two / three 512-thread workgroups per CU (LDS: 79 / 47 KiB each)
static constexpr int TAIL_G = 8;           // pieces per group of the drain (two groups in flight per wave)
// the tail is accumulated in Q7.24 fixed point with integer LDS atomics (ds_add_u32; the float form
// ds_add_f32 measured ~1.4x slower here): |sum| <= 1, each product is quantised with error <= 2^-25,
// which the per-common-item term of row_eps covers.  The rater-side factor comes as Q0.16 (4-byte tail
// entries: half the L2/MALL traffic of an (id, fp32) pair); its rounding, <= 2^-16 |pre(u,i)| per product,
// is added to the row's error band exactly (tail_eps).
// it_pack's word W ~ y * (2^31 - 2^16) (prep.hip: k_item_major); the row-side factor is x * 2^24 / (2^31 - 2^16), so that
// W * factor = x y 2^24 (Q7.24); Jaccard handles: W ~ 2^30, factor 2^-30, product 1
static constexpr double TAIL_SCALE = 2147418112.0;
static constexpr float TAIL_FIX = (float)(16777216.0 / TAIL_SCALE);
static constexpr float TAIL_FIX_JAC = 1.0f / 1073741824.0f;
static constexpr float TAIL_UNFIX = 1.0f / 16777216.0f;

// The histogram covers [HIST_LO, HIST_LO + 1) with NBINS bins of width 1/NBINS; values outside land in the end
// bins.  (A k-th largest value below HIST_LO puts the threshold in bin 0 = "everything qualifies" and the row takes
// the exact fallback; one above the range only loosens the threshold to the top bin's edge.)
static constexpr float HIST_LO = -0.125f;
__device__ __forceinline__ int sim_bin(float x) {
    int b = (int)floorf((x - HIST_LO) * (float)NBINS);
    return min(max(b, 0), NBINS - 1);
}

// rigorous bound on |S[u][v] - s_uv|: operand rounding (eps_base) plus, per common item, the roundings of the tail product
// (the rater-side word to fp32, the multiplication, the conversion to Q7.24) and of the dense part's fp32 accumulation —
// adding a zero product is exact, so at most 3 roundings per common item, each <= 2^-24 of a quantity <= 1.01
__device__ __forceinline__ float row_eps(float eps_base, int64_t row_len) {
    return eps_base + (float)row_len * 3.0f * 6.1e-8f;
}

struct TailArgs {
    const int64_t* u_ptr;  // user-major rows
    const int32_t* s_col;
    const double* s_pre;
    const int32_t* colmap;  // < 0: tail item
    const int64_t* i_ptr;   // item-major rows, raters ascending
    const uint32_t* it_pack;  // LDS cell of the column inside its tile << 17 | Q0.16 value (prep.hip: k_item_major)
    uint32_t pack_bytes;
    const uint32_t* it_tile;  // [I][tile_stride]: first entry of the item's list with user >= t * TCOLS
    int32_t tile_stride;
    int32_t has_tail;
    // per-row tail entry lists (k_tail_entries below, rebuilt whenever the head changes): the row's tail entries compacted to
    // the front of its position range, so that a row (or a 256-entry chunk of it) is ONE coalesced read instead of the
    // chain s_col -> colmap -> LDS slot atomic; and the two per-row sums of the error band
    const int32_t* te_cnt;   // [U] tail entries of the row
    const int32_t* te_item;  // [n] item of the j-th tail entry of row u at u_ptr[u] + j
    const float* te_x;       // [n] pre(u, item) * TAIL_FIX
    const float* row_tail_abs;  // [U] sum over the row's tail entries of |pre(u, i)|
    const float* row_head_sq;   // [U] sum over the row's head entries of pre(u, i)^2
    const float* row_len;       // Jaccard handles: [U rounded up to a tile + a tile] |I(v)| as float (1.0 in the padding)
};

// one wave per user: the row's tail entries (colmap < 0) in position order, compacted to the front of the row's range; the
// sums in a fixed order (position -> lane, then xor shuffles), so that the error band is the same on every run
__global__ void k_tail_entries(int32_t u_lo, int32_t U, const int64_t* __restrict__ u_ptr, const int32_t* __restrict__ s_col,
                               const double* __restrict__ s_pre, const int32_t* __restrict__ colmap, int32_t* __restrict__ te_cnt,
                               int32_t* __restrict__ te_item, float* __restrict__ te_x, float* __restrict__ row_tail_abs,
                               float* __restrict__ row_head_sq, int ones) {
    const int32_t u = u_lo + (int32_t)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (u >= U) return;
    const int64_t ub = u_ptr[u], ue = u_ptr[u + 1];
    float tail_abs = 0.f, head_sq = 0.f;
    int32_t n_out = 0;
    for (int64_t p0 = ub; p0 < ue; p0 += 64) {
        const int64_t p = p0 + lane;
        bool is_tail = false;
        int32_t item = 0;
        float x = 0.f;
        if (p < ue) {
            item = s_col[p];
            x = (float)s_pre[p];
            is_tail = colmap[item] < 0;
            if (is_tail) tail_abs += fabsf(x);
            else head_sq = __builtin_fmaf(x, x, head_sq);
        }
        const unsigned long long m = __ballot(is_tail);
        if (is_tail) {
            const int32_t slot = n_out + __popcll(m & ((1ull << lane) - 1ull));
            te_item[ub + slot] = item;
            te_x[ub + slot] = ones ? TAIL_FIX_JAC : x * TAIL_FIX;  // (ones: Jaccard handles count — factor 2^-30 x entry value 2^30)
        }
        n_out += __popcll(m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tail_abs += __shfl_xor(tail_abs, o);
        head_sq += __shfl_xor(head_sq, o);
    }
    if (lane == 0) {
        te_cnt[u] = n_out;
        row_tail_abs[u] = tail_abs;
        row_head_sq[u] = head_sq;
    }
}

// (only the owned users' rows are ever selected from: a shard builds the lists of its own users)
void launch_tail_entries(const Train& tr, const int32_t* d_colmap, int32_t* te_cnt, int32_t* te_item, float* te_x,
                         float* row_tail_abs, float* row_head_sq, hipStream_t st) {
    const int32_t lo = tr.own_lo, hi = tr.own_hi;
    if (hi <= lo) return;
    k_tail_entries<<<(unsigned)ceil_div((int64_t)(hi - lo) * 64, 256), 256, 0, st>>>(lo, hi, tr.u_ptr.p, tr.s_col.p, tr.s_pre.p, d_colmap, te_cnt,
                                                                                     te_item, te_x, row_tail_abs, row_head_sq, tr.jaccard ? 1 : 0);
    KN_HIP(hipGetLastError());
}

// |I(v)| of every user as float, 1.0 in the padding (Jaccard handles: the denominators of select.hip's approximate values)
__global__ void k_row_len(int64_t n_out, int32_t U, const int64_t* __restrict__ u_ptr, float* __restrict__ out) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n_out) out[v] = v < U ? (float)(u_ptr[v + 1] - u_ptr[v]) : 1.0f;
}

int64_t row_len_size(int32_t U) { return round_up(U, SELECT_TCOLS) + SELECT_TCOLS; }

void launch_row_len(const Train& tr, float* d_out, hipStream_t st) {
    const int64_t n_out = row_len_size(tr.U);
    k_row_len<<<(unsigned)ceil_div(n_out, 256), 256, 0, st>>>(n_out, tr.U, tr.u_ptr.p, d_out);
    KN_HIP(hipGetLastError());
}

// threshold from the cumulative histogram: lower edge of the bin holding the kk-th largest value seen so
// far, minus 2 eps (-inf while fewer than kk values have been seen).  Thread t owns bins [4t, 4t+4).
// *s_bin = that bin (0 while fewer than kk values have been seen).
__device__ __forceinline__ void block_threshold(const uint32_t* hist, uint32_t* wtot, float* s_thr, int32_t* s_bin, int32_t kk, float eps) {
    constexpr int PER = NBINS / TPB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t h[PER];
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        h[j] = hist[threadIdx.x * PER + j];
        mine += h[j];
    }
    // inclusive suffix sum inside the wave (lanes above me + me)
    uint32_t suf = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t up = __shfl_down(suf, o);
        if (lane + o < 64) suf += up;
    }
    if (lane == 0) wtot[wave] = suf;
    if (threadIdx.x == 0) { *s_thr = -INFINITY; *s_bin = 0; }
    __syncthreads();
    uint32_t higher = 0;  // values in the waves above mine
    for (int w = wave + 1; w < TPB / 64; ++w) higher += wtot[w];
    const uint32_t incl = suf + higher;        // values in my bins and above
    const uint32_t above = incl - mine;        // values strictly above my bins
    if (above < (uint32_t)kk && incl >= (uint32_t)kk) {
        uint32_t c = above;
        int j = PER - 1;
        for (; j > 0; --j) {
            c += h[j];
            if (c >= (uint32_t)kk) break;
        }
        const int b = threadIdx.x * PER + j;
        // every value in bin b is >= its lower edge (up to one float rounding of x + 1)
        const float edge = (float)b / (float)NBINS + HIST_LO;
        *s_thr = (b == 0) ? -INFINITY : edge - 2.0f * eps - 1e-6f;
        *s_bin = b;
    }
    __syncthreads();
}

// 8 consecutive panel entries starting at a multiple of 8 (rows are padded to ld, a multiple of 256): the
// load and the conversion are separate so that the next tile's entries can be in flight while this one is used
template <class ST>
struct Raw8;
template <>
struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = reinterpret_cast<const float4*>(p)[0];
        b = reinterpret_cast<const float4*>(p)[1];
    }
    __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
    __device__ __forceinline__ void unpack(float* out) const {
        out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
    }
};
template <>
struct Raw8<_Float16> {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    h8 v;
    __device__ __forceinline__ void load(const _Float16* p) { v = *reinterpret_cast<const h8*>(p); }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (_Float16)0.f;
    }
    __device__ __forceinline__ void unpack(float* out) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
    }
};

// JAC: the handle's similarity is the Jaccard coefficient :440-464.  Panel + tail then hold exact COUNTS of common items
// (0/1 operands, tail entries of value 1, read out unscaled) and every column's value becomes
// count / (|I(u)| + |I(v)| - count) in fp32 before the thresholds see it: both operands are exact integers below 2^24, so
// the quotient is the correctly rounded fp32 image of the exact similarity and the error band is a few 1e-7.
template <class ST, bool JAC>
__global__ void __launch_bounds__(TPB, WAVES_PER_EU) k_tail_select(const ST* __restrict__ S, int32_t s_by_user, int64_t ld, int32_t n_rows,
                                                     const int32_t* __restrict__ row_user, const int32_t* __restrict__ row_srow, TailArgs T, int32_t U,
                                                     int32_t kk, float eps_opnd, float eps_rest, int32_t cap, int32_t* __restrict__ cand_idx,
                                                     float* __restrict__ cand_approx, int32_t* __restrict__ cand_cnt,
                                                     float* __restrict__ cand_eps, int32_t* __restrict__ grp_v0,
                                                     float* __restrict__ grp_x, int32_t GCAP, float ant_sigma) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* itile = reinterpret_cast<int32_t*>(smem);            // [TCOLS] tail accumulator, Q7.24, cell layout of it_pack
    uint32_t* hist = reinterpret_cast<uint32_t*>(itile + TCOLS);  // [NBINS]
    float* e_x = reinterpret_cast<float*>(hist + NBINS);          // [EMAX] pre(u, item) * 2^24
    // the per-tile tables exist twice: tile t + 1's are filled while tile t is still being used (see the tile loop)
    uint32_t* e_b0 = reinterpret_cast<uint32_t*>(e_x + EMAX);     // [2][EMAX] the tile's range of the item's rater list
    uint32_t* e_e0 = e_b0 + 2 * EMAX;                             // [2][EMAX]
    uint32_t* e_ps0 = e_e0 + 2 * EMAX;                            // [2][EMAX] exclusive prefix of the 64-rater piece counts
    int32_t* e_item = reinterpret_cast<int32_t*>(e_ps0);          //   (aliased: the item of the entry, until its owner read it)
    uint16_t* piece_e0 = reinterpret_cast<uint16_t*>(e_ps0 + 2 * EMAX);  // [2][PMAX] entry of every piece
    // (no static __shared__: the accumulator must sit at LDS address 0, its cell addresses come straight out of it_pack)
    uint32_t* wtot = reinterpret_cast<uint32_t*>(piece_e0 + 2 * PMAX);   // [TPB / 64]
    uint32_t* wtot2 = wtot + TPB / 64 + 4;                        // [TPB / 64] piece counts per wave (setup)
    uint32_t* e_b = e_b0;
    uint32_t* e_e = e_e0;
    uint32_t* e_ps = e_ps0;
    uint16_t* piece_e = piece_e0;
    float& s_thr = *reinterpret_cast<float*>(wtot + TPB / 64);
    uint32_t& s_count = wtot[TPB / 64 + 1];
    int32_t& s_ne = *reinterpret_cast<int32_t*>(wtot + TPB / 64 + 2);
    int32_t& s_bin = *reinterpret_cast<int32_t*>(wtot2 + TPB / 64);  // bin of the threshold block_threshold found last (scratch cells)
    const int32_t r = blockIdx.x;
    if (r >= n_rows) return;
#ifdef KNNCF_SELECT_PROFILE
    long long ph_t = clock64();
#endif
    const int32_t u = row_user[r];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ub = T.u_ptr[u], ue = T.u_ptr[u + 1];
    // (symmetric path: S is the whole matrix, indexed by dense user; row blocks: by block row — row_srow when the launch
    // covers a subset of the block)
    const ST* row = S + (int64_t)(s_by_user ? u : (row_srow ? row_srow[r] : r)) * ld;
    int32_t* out_idx = cand_idx + (int64_t)r * cap;
    float* out_apx = cand_approx + (int64_t)r * cap;
    // provisional store of the row: whole GROUPS of 8 columns (first column + the 8 values) whose maximum reached the
    // threshold known when their tile was processed; the columns are sorted out once, at the end
    int32_t* g_v0 = grp_v0 + (int64_t)r * GCAP;
    float4* g_x = reinterpret_cast<float4*>(grp_x + (int64_t)r * GCAP * 8);
    // this thread's CPT columns of a tile: group j covers columns t0 + 8 (tid + TPB j) .. + 7
    Raw8<ST> raw[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int64_t v0 = 8 * (threadIdx.x + TPB * j);
        if (v0 < ld) raw[j].load(row + v0);
        else raw[j].zero();
    }
    for (int b = threadIdx.x; b < NBINS; b += TPB) hist[b] = 0;
    if (threadIdx.x == 0) { s_count = 0; s_ne = 0; s_thr = -INFINITY; }
    __syncthreads();

    // tail entries of the row, EMAX of them (a "chunk") at a time, straight out of the row's precomputed list.  Rows of up
    // to EMAX tail entries (all but the heaviest raters) take them once; longer rows re-read a chunk for every tile.
    const int32_t n_te = T.has_tail ? T.te_cnt[u] : 0;
    const int n_chunks = (n_te + EMAX - 1) / EMAX;
    const int n_tiles = T.tile_stride - 1;
    const bool reg_counts = n_chunks == 1 && n_tiles <= MAXT;
    auto collect = [&](int ch) {
        const int32_t c0 = ch * EMAX, cn = min(EMAX, n_te - c0);
        for (int32_t j = threadIdx.x; j < cn; j += TPB) {
            e_item[j] = T.te_item[ub + c0 + j];
            e_x[j] = T.te_x[ub + c0 + j];
        }
        if (threadIdx.x == 0) s_ne = cn;
    };
    // Two per-row quantities of the error band (k_tail_entries):
    //  * tail_abs = sum over the row's tail entries of |pre(u, i)|: the Q0.16 rater-side factors err by 2^-16 each;
    //  * head_sq  = sum over the row's HEAD entries of pre(u, i)^2: the operand roundings of the dense part err by
    //    (2u + u^2) sum_head |x y| <= (2u + u^2) ||x_head|| ||y_head|| <= (2u + u^2) ||x_head||   (||y|| <= 1).
    float tail_abs = 0.f;
    float head_norm = 1.0f;  // no tail: every item is a head column
    if (T.has_tail) {
        tail_abs = T.row_tail_abs[u];
        head_norm = fminf(1.0f, sqrtf(T.row_head_sq[u]) * 1.0001f + 1e-6f);  // (fp32 summation slack)
    }
    if (n_chunks == 1) collect(0);
    __syncthreads();
    const float eps = JAC ? 4e-7f : row_eps(eps_opnd * head_norm + eps_rest, ue - ub) + tail_abs * (1.0001f / 65536.0f);
    if (threadIdx.x == 0) cand_eps[r] = eps;
    const bool any_tail = n_chunks > 0;
    // single-chunk rows: thread e owns entry e for the whole row.  Its item's rater counts per tile (<= 32768 each)
    // are packed into registers, so the tile loop needs no global read for the ranges
    const uint32_t* my_tb = T.it_tile;
    uint32_t cur_b = 0, cw[MAXT / 2];
#pragma unroll
    for (int t = 0; t < MAXT / 2; ++t) cw[t] = 0;
    if (n_chunks == 1 && (int32_t)threadIdx.x < s_ne) {
        my_tb = T.it_tile + (int64_t)e_item[threadIdx.x] * T.tile_stride;
        if (reg_counts) {
            uint32_t tbv[MAXT + 1];
#pragma unroll
            for (int t = 0; t <= MAXT; ++t) tbv[t] = my_tb[min(t, n_tiles)];
            cur_b = tbv[0];
#pragma unroll
            for (int t = 0; t < MAXT; ++t) cw[t >> 1] |= (tbv[t + 1] - tbv[t]) << (16 * (t & 1));
        }
    }
    if (any_tail) {  // the accumulator starts clean; afterwards every tile's read-out clears what it reads
        for (int32_t c = threadIdx.x; c < TCOLS; c += TPB) itile[c] = 0;
    }
    __syncthreads();  // also: every owner has read e_item before e_ps (its alias) is written

    // ---- the tail machinery: setup (ranges, prefix, piece table) / window (piece descriptors of up to 64 pieces of
    // this wave) / tail_group (TAIL_G piece loads and their LDS atomics) ------------------------------------------
    uint32_t P = 0, p_lo = 0, p_hi = 0, n_here = 0;
    int32_t ne = 0;
    // piece descriptors of a window, one piece per lane: d_q = first entry of the piece, d_xs = the row-side factor
    // pre(u, item) * 2^8 as fp32 with its 6 low mantissa bits replaced by (64 - entries of the piece): the word is used
    // as the multiplier as it is (|relative error| < 2^-17, inside the row's error band: api.cpp gemm_eps_rest) and as the
    // shift count of the piece's lane mask (s_lshr_b64 reads the low 6 bits only)
    uint32_t d_q = 0, d_xs = 0;
    // setup of a tile = two halves around ONE barrier: (a) every entry's range inside the tile and the piece counts per
    // wave, (b) the exclusive prefix, the piece -> entry table and this wave's share [p_lo, p_hi) of the pieces.
    // `buf` selects the table copy.  For single-chunk rows the halves of tile t + 1 sit around a barrier tile t needs
    // anyway, so the setup costs no barrier of its own.
    uint32_t su_np = 0, su_incl = 0;  // carried from half (a) to half (b)
    auto setup_a = [&](int tile, int buf) {
        ne = s_ne;
        uint32_t np = 0;
        if ((int32_t)threadIdx.x < ne) {
            uint32_t qb, cnt;
            if (reg_counts) {
                // the counts are consumed in tile order: the current pair of tiles always sits in cw[0], and after every odd
                // tile the words move down one place (static indices only: a select chain over the tile index was turned
                // into a scratch array by the compiler — one dependent scratch load per tile whose vmcnt(0) also drained the
                // panel loads in flight)
                cnt = (cw[0] >> (16 * (tile & 1))) & 0xffffu;
                qb = cur_b;
                cur_b += cnt;
                if (tile & 1) {
#pragma unroll
                    for (int t2 = 0; t2 + 1 < MAXT / 2; ++t2) cw[t2] = cw[t2 + 1];
                }
            } else {
                const uint32_t* tb = (n_chunks > 1 ? T.it_tile + (int64_t)e_item[threadIdx.x] * T.tile_stride : my_tb) + tile;
                qb = tb[0];
                cnt = tb[1] - qb;
            }
            e_b0[buf * EMAX + threadIdx.x] = qb;
            e_e0[buf * EMAX + threadIdx.x] = qb + cnt;
            np = (cnt + 63u) >> 6;
        }
        su_np = np;
        su_incl = wave_incl_scan(np);
        if (lane == 63 && wave < EMAX / 64) wtot2[wave] = su_incl;
    };
    auto setup_b = [&](int buf) {
        uint32_t off = 0;
        P = 0;
#pragma unroll
        for (int w2 = 0; w2 < EMAX / 64; ++w2) {
            const uint32_t sw = wtot2[w2];
            P += sw;
            if (w2 < wave) off += sw;
        }
        if ((int32_t)threadIdx.x < ne) {
            const uint32_t excl = off + su_incl - su_np;
            e_ps0[buf * EMAX + threadIdx.x] = excl;
            if (P <= PMAX)
                for (uint32_t k2 = 0; k2 < su_np; ++k2) piece_e0[buf * PMAX + excl + k2] = (uint16_t)threadIdx.x;
        }
        // the pieces are dealt evenly: wave w takes [p_lo, p_hi).  (wave-uniform values are moved to scalar registers
        // explicitly: the loops below then run on the scalar unit)
        P = __builtin_amdgcn_readfirstlane(P);
        ne = __builtin_amdgcn_readfirstlane(ne);
        const uint32_t wv = __builtin_amdgcn_readfirstlane(wave);
#ifdef KNNCF_TAIL_EVEN_PIECES  /* A/B switch: the pieces dealt evenly, every wave rounds its share up to whole groups itself */
        p_lo = (uint32_t)(((uint64_t)P * wv) / (TPB / 64));
        p_hi = (uint32_t)(((uint64_t)P * (wv + 1)) / (TPB / 64));
#else
        // whole GROUPS of TAIL_G pieces are dealt, so that only the last wave's last group is padded with null pieces (every
        // wave rounding its own share up padded ~4 of ~27 pieces per wave and tile)
        const uint32_t n_groups = (P + TAIL_G - 1) / TAIL_G;
        p_lo = min(P, (uint32_t)(((uint64_t)n_groups * wv) / (TPB / 64)) * TAIL_G);
        p_hi = min(P, (uint32_t)(((uint64_t)n_groups * (wv + 1)) / (TPB / 64)) * TAIL_G);
#endif
    };
    auto use_tables = [&](int buf) {
        e_b = e_b0 + buf * EMAX;
        e_e = e_e0 + buf * EMAX;
        e_ps = e_ps0 + buf * EMAX;
        piece_e = piece_e0 + buf * PMAX;
    };
    auto window = [&](uint32_t pw) {  // lane l looks up the entry of piece pw + l and keeps its descriptor
        n_here = min(64u, p_hi - pw);
        // lanes past the window's pieces hold a NULL piece (one lane, factor 0.0: it adds 0 to the cell of entry 0), so
        // that the drain below only ever runs whole groups
        d_q = 0;
        d_xs = 63u;
        if ((uint32_t)lane < n_here) {
            const uint32_t p2 = pw + lane;
            int32_t e;
            if (P <= PMAX) {
                e = piece_e[p2];
            } else {  // largest e with e_ps[e] <= p2 (a non-empty entry, as p2 < P)
                int32_t lo = 0, hi = ne - 1;
                while (lo < hi) {
                    const int32_t mid = (lo + hi + 1) >> 1;
                    if (e_ps[mid] <= p2) lo = mid;
                    else hi = mid - 1;
                }
                e = lo;
            }
            d_q = e_b[e] + ((p2 - e_ps[e]) << 6);
            const uint32_t len = min(64u, e_e[e] - d_q);  // 1 .. 64
            d_xs = (__float_as_uint(e_x[e]) & ~63u) | (64u - len);
        }
    };
    // The drain: the pieces of a window (<= 64, 8 to a group), as ONE hand-scheduled asm statement per window.  A piece is
    // one 64-lane load of 4-byte tail entries and one LDS-atomic instruction.  Written to the instruction: per piece 6 VALU
    // (two v_readlane for the piece's start and factor; convert / multiply / convert-to-nearest of the rater-side word — used
    // WHOLE, its address bits are pre-compensated in the value field, prep.hip: k_item_major — and ONE v_and for the cell's
    // byte address: the accumulator sits at LDS address 0 and the address comes out of it_pack ready-made; round 2 spent 8:
    // a sign-extraction of a 17-bit field and a shift + mask for the cell), 5 SALU, one buffer load, one ds_add.  The compiler's version of the same loop took 14 VALU per piece:
    // it paid a compare and a select per piece on both sides to keep lanes past the end of a piece harmless; here the
    // loads run unmasked (the entries behind a piece's end are mapped memory: the buffer descriptor covers the whole array
    // and answers 0 beyond it) and the atomics run under the piece's lane mask, set from the scalar unit
    // (s_lshr_b64 exec, -1, 64 - len).  Measured (syn-25m, timing-only ablations): of the kernel's 35.6 ms,
    // the drain is 14.6 — 6.2 of them its loads and atomics, 8.4 instruction issue — and the rest of the tail machinery 4.4.
    // Counting by hand (cdna_hip_programming.md 5.7): every apply waits for its own load with vmcnt(15 - j): a whole
    // younger group is in flight behind it.  Any older load or store of the wave (the next tile's panel entries, the last
    // tile's group stores) retires first, in order, so the count only ever over-waits; the statement ends with vmcnt(0).
    // The ds_adds need no wait (no return value; the tile's barrier drains lgkmcnt).  EXEC is all ones on entry
    // (wave-uniform control flow, 512-thread blocks) and is restored after every atomic.  Hazards (hipcc pads nothing inside
    // the string): every SGPR a buffer_load or a v_readlane lane-select reads is written by the scalar unit (s_lshl /
    // s_add), never directly by a VALU; the factor's v_readlane is three instructions ahead of the v_mul that reads its
    // SGPR (gfx940 family: 2 wait states); SALU writes of EXEC need no wait states before VMEM / LDS; s_add / s_lshl /
    // s_cmp write SCC (declared).
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    // raw buffer descriptor of it_pack (stride 0: byte offsets, range-checked against the array's size; the same words
    // __builtin_amdgcn_make_buffer_rsrc(it_pack, 0, pack_bytes, 0x00020000) builds)
    const uint64_t pack_base = reinterpret_cast<uint64_t>(T.it_pack);
    const u32x4 pack_rsrc = {(uint32_t)pack_base, (uint32_t)(pack_base >> 32) & 0xffffu, T.pack_bytes, 0x00020000u};
    const uint32_t lane4 = (uint32_t)lane << 2;
#ifdef KNNCF_ABL_NOLOAD  /* timing-only ablations (results are wrong): what the drain costs without its loads / atomics */
#define KN_TAIL_LOAD(W) "v_mov_b32 %[" W "], %[l4]\n\t"
#else
#define KN_TAIL_LOAD(W) "buffer_load_dword %[" W "], %[l4], %[rs], %[sq] offen\n\t"
#endif
#ifdef KNNCF_ABL_NOATOMIC
#define KN_TAIL_ADD "s_nop 0\n\t"
#else
#define KN_TAIL_ADD "ds_add_u32 %[a], %[t]\n\t"
#endif
// piece (group base %[sg]) + OFF: start of the piece -> byte offset in a scalar, the load of its 64 entries into W
#ifdef KNNCF_ABL_HOTLOAD  /* timing-only ablation (results are wrong): every piece is read out of the array's first 16 KiB, i.e. the L1/L2 */
#define KN_TAIL_HOT "s_and_b32 %[sq], %[sq], 0x3ffc\n\t"
#else
#define KN_TAIL_HOT
#endif
#define KN_TAIL_ISSUE(W, OFF)                                              \
    "s_add_u32 %[sj], %[sg], " #OFF "\n\t"                                 \
    "v_readlane_b32 %[sq], %[dq], %[sj]\n\t"                               \
    "s_lshl_b32 %[sq], %[sq], 2\n\t"                                       \
    KN_TAIL_HOT                                                            \
    KN_TAIL_LOAD(W)
// piece (group base %[sg]) + OFF: wait until at most N younger loads are in flight, products, cell addresses, the atomics
// under the piece's lane mask
#define KN_TAIL_APPLY(W, OFF, N)                                           \
    "s_add_u32 %[sj], %[sg], " #OFF "\n\t"                                 \
    "v_readlane_b32 %[sx], %[dx], %[sj]\n\t"                               \
    "s_waitcnt vmcnt(" #N ")\n\t"                                          \
    "v_cvt_f32_i32_e32 %[t], %[" W "]\n\t"                                 \
    "v_and_b32_e32 %[a], 0xfffc, %[" W "]\n\t"                             \
    "v_mul_f32_e32 %[t], %[sx], %[t]\n\t"                                  \
    "v_cvt_rpi_i32_f32_e32 %[t], %[t]\n\t"                                 \
    "s_lshr_b64 exec, -1, %[sx]\n\t"                                       \
    KN_TAIL_ADD                                                            \
    "s_mov_b64 exec, -1\n\t"
#define KN_TAIL_ISSUE8(S, B)                                                                                       \
    KN_TAIL_ISSUE(S "0", B + 0) KN_TAIL_ISSUE(S "1", B + 1) KN_TAIL_ISSUE(S "2", B + 2) KN_TAIL_ISSUE(S "3", B + 3) \
    KN_TAIL_ISSUE(S "4", B + 4) KN_TAIL_ISSUE(S "5", B + 5) KN_TAIL_ISSUE(S "6", B + 6) KN_TAIL_ISSUE(S "7", B + 7)
#define KN_TAIL_APPLY8(S)                                                                                          \
    KN_TAIL_APPLY(S "0", 0, 15) KN_TAIL_APPLY(S "1", 1, 14) KN_TAIL_APPLY(S "2", 2, 13) KN_TAIL_APPLY(S "3", 3, 12) \
    KN_TAIL_APPLY(S "4", 4, 11) KN_TAIL_APPLY(S "5", 5, 10) KN_TAIL_APPLY(S "6", 6, 9) KN_TAIL_APPLY(S "7", 7, 8)
#define KN_TAIL_APPLY8_LAST(S) /* no younger group in flight behind this one */                                   \
    KN_TAIL_APPLY(S "0", 0, 7) KN_TAIL_APPLY(S "1", 1, 6) KN_TAIL_APPLY(S "2", 2, 5) KN_TAIL_APPLY(S "3", 3, 4) \
    KN_TAIL_APPLY(S "4", 4, 3) KN_TAIL_APPLY(S "5", 5, 2) KN_TAIL_APPLY(S "6", 6, 1) KN_TAIL_APPLY(S "7", 7, 0)
    // all groups of a window (n8 = its pieces rounded up to whole groups, 8 .. 64), software-pipelined two groups deep:
    // group g + 1's loads are issued before group g's atomics, into the other register set (A / B alternate, the loop is
    // unrolled twice).  The LAST group of a window is applied without a look-ahead group behind it (its own code copy: the
    // vmcnt immediates count 8 loads fewer).  Round 2 always issued a next group — past the window's end a dummy one, so that
    // one set of immediates served every group — and waited for it at the end: a window of 9 .. 16 pieces (a wave's share of a
    // tile is ~23 at the ml-25m shape) then cost two memory latencies instead of one.
    auto tail_window = [&](uint32_t n8) {
        uint32_t a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3, b4, b5, b6, b7, t, a, sq, sx, sj, sg;
        asm volatile(
            "s_mov_b32 %[sg], 0\n\t"
            KN_TAIL_ISSUE8("a", 0)
            "1:\n\t"
            "s_add_u32 %[sj], %[sg], 8\n\t"
            "s_cmp_ge_u32 %[sj], %[n8]\n\t"
            "s_cbranch_scc1 3f\n\t"
            KN_TAIL_ISSUE8("b", 8)
            KN_TAIL_APPLY8("a")
            "s_add_u32 %[sg], %[sg], 8\n\t"
            "s_add_u32 %[sj], %[sg], 8\n\t"
            "s_cmp_ge_u32 %[sj], %[n8]\n\t"
            "s_cbranch_scc1 4f\n\t"
            KN_TAIL_ISSUE8("a", 8)
            KN_TAIL_APPLY8("b")
            "s_add_u32 %[sg], %[sg], 8\n\t"
            "s_branch 1b\n\t"
            "3:\n\t"
            KN_TAIL_APPLY8_LAST("a")
            "s_branch 5f\n\t"
            "4:\n\t"
            KN_TAIL_APPLY8_LAST("b")
            "5:\n\t"
            : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [a4] "=&v"(a4), [a5] "=&v"(a5), [a6] "=&v"(a6),
              [a7] "=&v"(a7), [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [b4] "=&v"(b4), [b5] "=&v"(b5),
              [b6] "=&v"(b6), [b7] "=&v"(b7), [t] "=&v"(t), [a] "=&v"(a), [sq] "=&s"(sq), [sx] "=&s"(sx), [sj] "=&s"(sj),
              [sg] "=&s"(sg)
            : [dq] "v"(d_q), [dx] "v"(d_xs), [n8] "s"(n8), [l4] "v"(lane4), [rs] "s"(pack_rsrc)
            : "memory", "scc");
    };
#undef KN_TAIL_ISSUE8
#undef KN_TAIL_APPLY8
#undef KN_TAIL_APPLY8_LAST
#undef KN_TAIL_ISSUE
#undef KN_TAIL_APPLY
    auto drain = [&](bool) {
#ifdef KNNCF_ABL_NODRAIN  /* timing-only ablation: everything but the drain itself */
        return;
#endif
        for (uint32_t pw = p_lo; pw < p_hi; pw += 64) {
            window(pw);
            tail_window(__builtin_amdgcn_readfirstlane((n_here + 7u) & ~7u));
        }
    };
    const bool pipelined = any_tail && n_chunks == 1;  // single-chunk rows: tile t + 1 is set up inside tile t
    if (pipelined) {
        setup_a(0, 0);
        __syncthreads();
        setup_b(0);
        __syncthreads();
    }
    // ANTICIPATED THRESHOLDS.  The k-th largest value seen so far is a valid threshold but a loose one early in the row:
    // after one tile of ten it sits at the 1.8 % quantile of the row where the final one sits at 0.18 %, and half of all the
    // groups a row stores come from its first two tiles.  Dense user indices are HashSet ranks of the raw ids, i.e. a tile is
    // a pseudo-random sample of the users: of the row's k largest values a fraction f of the columns holds about k f, so
    // after m tiles the emission threshold is taken at rank k f + sigma sqrt(k f (1 - f)) + 3 instead of k (never above k;
    // sigma = 7, launch_tail_select).
    // That is a guess, and the kernel does not trust it: the final threshold is the exact k-th largest of what was stored,
    // and every value of the row in a bin at or above the highest bin ever used for emission IS stored — so if the final
    // threshold's bin is at or above that bin the store provably holds every value >= final threshold - 2 eps, as before;
    // if it is below (the guess overshot: a 7-sigma event per refresh under the sampling model; any distribution is allowed) the row takes the
    // exact fallback.  Rows of fewer than four tiles use the plain rank k throughout.
    const int32_t n_tiles_row = (U + TCOLS - 1) / TCOLS;
    auto rank_after = [&](int m) -> int32_t {  // rank of the emission threshold once m tiles are in the histogram
        if (ant_sigma < 0.0f || n_tiles_row < 4) return kk;
        const float f = fminf(1.0f, (float)m * (float)TCOLS / (float)U);
        const float mu = (float)kk * f;
        const int32_t rk = (int32_t)ceilf(mu + ant_sigma * sqrtf(mu * (1.0f - f)) + 3.0f);
        return rk < kk ? max(rk, 1) : kk;
    };
    int32_t bin_used = 0;  // highest histogram bin an emission threshold was read from (block-uniform)
    // The histogram of the provisional values is kept incrementally: a refresh adds the groups stored since the last one
    // (values >= the threshold of that moment; the threshold only rises, so every value that can still be among the top k
    // passes, and the bins below the threshold are never looked at) instead of zeroing it and re-reading the whole store —
    // four rebuilds per row re-read ~3 x the store's final size, 70 KB per row at the ml-25m shape.
    uint32_t counted = 0;  // groups already in the histogram (block-uniform)
    auto count_new_groups = [&](float floor) {
        const uint32_t G = min(s_count, (uint32_t)GCAP);
        for (uint32_t g = counted + threadIdx.x; g < G; g += TPB) {
            const float4 a = g_x[2 * g], b4 = g_x[2 * g + 1];
            const float x8[8] = {a.x, a.y, a.z, a.w, b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (x8[i] >= floor && x8[i] > -INFINITY) atomicAdd(&hist[sim_bin(x8[i])], 1u);
        }
        counted = G;
        __syncthreads();
    };
    PH(0);  // preamble: collect, clears
    int tile_no = 0;
    uint32_t next_refresh = (uint32_t)GCAP / 4;  // store level that triggers an extra threshold refresh (block-uniform)
    for (int32_t t0 = 0; t0 < U; t0 += TCOLS, ++tile_no) {
        // the tile's panel entries (requested one tile ago) are unpacked and the next tile's requested as soon as this
        // wave's share of the drain is done — before it waits for the other waves at the barrier (the drain ends with
        // vmcnt(0), so the entries have landed; the requests then have the barrier wait, the set-up and this tile's select
        // work to arrive behind)
        float sx[CPT];
        auto take_panel = [&]() {
#pragma unroll
            for (int j = 0; j < NG; ++j) raw[j].unpack(&sx[8 * j]);
            if (t0 + TCOLS < U) {
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const int64_t v0 = (int64_t)t0 + TCOLS + 8 * (threadIdx.x + TPB * j);
                    if (v0 < ld) raw[j].load(row + v0);
                    else raw[j].zero();
                }
            }
            if (t0 + TCOLS > U || (u >= t0 && u < t0 + TCOLS)) {  // the user itself and the padding never qualify
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const int32_t v0 = t0 + 8 * (threadIdx.x + TPB * j);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (v0 + i >= U || v0 + i == u) sx[8 * j + i] = -INFINITY;
                }
            }
        };
        if (pipelined) {
            use_tables(tile_no & 1);
            drain(false);
            PH(4);  // piece descriptors, loads, LDS atomics (this wave)
            take_panel();
            const bool more = t0 + TCOLS < U;
            if (more) setup_a(tile_no + 1, (tile_no + 1) & 1);
            __syncthreads();  // the tile's tail is complete — and the next tile's piece counts are visible
            PH(5);  // wait for the other waves
            if (more) setup_b((tile_no + 1) & 1);  // (published by the barrier at the end of this tile)
            else p_lo = p_hi = 0;
        } else if (any_tail) {
            use_tables(0);
            for (int ch = 0; ch < n_chunks; ++ch) {
                if (n_chunks > 1) {
                    __syncthreads();  // the previous chunk's tables are no longer read
                    collect(ch);
                    __syncthreads();
                }
                setup_a(tile_no, 0);
                __syncthreads();
                setup_b(0);
                __syncthreads();
                PH(2);  // ranges + prefix scan + piece table
                drain(false);
                PH(4);  // piece descriptors, loads, LDS atomics (this wave)
                __syncthreads();
                PH(5);  // wait for the other waves
            }
        }
        if (!pipelined) take_panel();
        PH(1);  // unpack + next tile's requests
        if (any_tail) {
            // read the accumulator out and clear it for the next tile (only this thread touches these cells between
            // the barriers; the scaling by 2^-24 is exact, so the fused multiply-add rounds like multiply + add).  The cell
            // layout (prep.hip: it_pack) keeps both 16-byte halves of a group at a 16-byte lane stride: no bank conflicts
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const int32_t g = threadIdx.x + TPB * j;
                int4* lo4 = reinterpret_cast<int4*>(itile + 4 * g);
                int4* hi4 = reinterpret_cast<int4*>(itile + TCOLS / 2 + 4 * g);
                const int4 a = *lo4, b = *hi4;
                *lo4 = make_int4(0, 0, 0, 0);
                *hi4 = make_int4(0, 0, 0, 0);
                constexpr float UNFIX = JAC ? 1.0f : TAIL_UNFIX;  // (Jaccard: the cells hold counts)
                sx[8 * j + 0] = __builtin_fmaf((float)a.x, UNFIX, sx[8 * j + 0]); sx[8 * j + 1] = __builtin_fmaf((float)a.y, UNFIX, sx[8 * j + 1]);
                sx[8 * j + 2] = __builtin_fmaf((float)a.z, UNFIX, sx[8 * j + 2]); sx[8 * j + 3] = __builtin_fmaf((float)a.w, UNFIX, sx[8 * j + 3]);
                sx[8 * j + 4] = __builtin_fmaf((float)b.x, UNFIX, sx[8 * j + 4]); sx[8 * j + 5] = __builtin_fmaf((float)b.y, UNFIX, sx[8 * j + 5]);
                sx[8 * j + 6] = __builtin_fmaf((float)b.z, UNFIX, sx[8 * j + 6]); sx[8 * j + 7] = __builtin_fmaf((float)b.w, UNFIX, sx[8 * j + 7]);
            }
        }
        if (JAC) {  // counts -> Jaccard coefficients (the masked columns stay -inf)
            const float nu_f = (float)(ue - ub);
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const float4* lp = reinterpret_cast<const float4*>(T.row_len + (int64_t)t0 + 8 * (threadIdx.x + TPB * j));
                const float4 la = lp[0], lb = lp[1];
                const float len8[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float c = sx[8 * j + i];
                    sx[8 * j + i] = c > -INFINITY ? c / (nu_f + len8[i] - c) : c;
                }
            }
        }
        PH(6);  // read-out
        float gm[NG];  // group maxima: a group of 8 is first rejected as a whole (the kernel is VALU/latency-bound)
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const float* x8 = &sx[8 * j];
            gm[j] = fmaxf(fmaxf(fmaxf(x8[0], x8[1]), fmaxf(x8[2], x8[3])), fmaxf(fmaxf(x8[4], x8[5]), fmaxf(x8[6], x8[7])));
        }
        if (tile_no == 0) {
            // First tile: bootstrap a threshold from per-thread (or per-group, or per-column) maxima — the kk-th largest
            // of any set of maxima over disjoint column sets is a lower bound of the kk-th largest value of the row,
            // and a tight one (the top values sit in different threads) — with one histogram atomic per maximum
            // instead of one per column.
            const int32_t kk0 = rank_after(1);  // (the rank this histogram will be asked for)
            if (NG >= 4 && kk0 <= (TCOLS / 32) / 3) {  // maxima over 4 groups = 32 columns
#pragma unroll
                for (int j = 0; j + 3 < NG; j += 4) {
                    const float m = fmaxf(fmaxf(gm[j], gm[j + 1]), fmaxf(gm[j + 2], gm[j + 3]));
                    if (m > -INFINITY) atomicAdd(&hist[sim_bin(m)], 1u);
                }
            } else if (kk0 <= (TCOLS / 8) / 3) {
#pragma unroll
                for (int j = 0; j < NG; ++j)
                    if (gm[j] > -INFINITY) atomicAdd(&hist[sim_bin(gm[j])], 1u);
            } else {
#pragma unroll
                for (int i = 0; i < CPT; ++i)
                    if (sx[i] > -INFINITY) atomicAdd(&hist[sim_bin(sx[i])], 1u);
            }
            __syncthreads();
            block_threshold(hist, wtot, &s_thr, &s_bin, kk0, eps);
            bin_used = max(bin_used, s_bin);
            for (int b = threadIdx.x; b < NBINS; b += TPB) hist[b] = 0;  // the real histogram starts below
            __syncthreads();
        }
        {
            // Emission with the threshold known so far (it can only rise; a stale one just lets more through): a group of
            // 8 columns whose maximum qualifies is stored WHOLE (its 8 values, 32 contiguous bytes, + its first column).
            // Sorting out the individual columns here — per-column masks, slots, histogram updates under divergent
            // branches — cost a third of the kernel's VALU instructions (the kernel is VALU-bound); the few thousand
            // stored values are sorted out once per row instead.  Slots: one LDS atomic per wave.
            const float thr = s_thr;
            uint32_t gmask = 0;
#pragma unroll
            for (int j = 0; j < NG; ++j)
                if (gm[j] >= thr && gm[j] > -INFINITY) gmask |= 1u << j;
            const uint32_t cnt = __popc(gmask);
            const uint32_t incl = wave_incl_scan(cnt);
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            if (total > 0) {
                uint32_t base = 0;
                if (lane == 63) base = atomicAdd(&s_count, total);
                base = __builtin_amdgcn_readlane(base, 63);
                const uint32_t first = base + incl - cnt;
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    if (gmask & (1u << j)) {
                        const uint32_t pos = first + __popc(gmask & ((1u << j) - 1u));
                        if (pos < (uint32_t)GCAP) {
                            g_v0[pos] = t0 + 8 * (threadIdx.x + TPB * j);
                            g_x[2 * pos] = make_float4(sx[8 * j], sx[8 * j + 1], sx[8 * j + 2], sx[8 * j + 3]);
                            g_x[2 * pos + 1] = make_float4(sx[8 * j + 4], sx[8 * j + 5], sx[8 * j + 6], sx[8 * j + 7]);
                        }
                    }
                }
            }
        }
        PH(tile_no == 0 ? 7 : 8);  // histogram + emit
        __syncthreads();
        PH(9);
        // Threshold refresh from the stored values: after the second tile (tightens the rest of the row) and whenever
        // the store is filling up (wide error bands, e.g. bf16 operands)
        {
            const uint32_t prov = s_count;  // (block-uniform after the barrier)
            if (((tile_no < 32 && ((KNNCF_REFRESH_MASK >> tile_no) & 1u)) || prov > next_refresh) && prov <= (uint32_t)GCAP && t0 + TCOLS < U) {
                count_new_groups(s_thr);
                block_threshold(hist, wtot, &s_thr, &s_bin, rank_after(tile_no + 1), eps);
                bin_used = max(bin_used, s_bin);
                next_refresh = max(next_refresh, prov + (uint32_t)GCAP / 8);  // (the store is not compacted: refresh again only after it has grown)
            }
        }
    }

    // ---- the final threshold and the shortlist: every stored value >= T_r - 2 eps_r, with its column -------------------
    if (s_count > (uint32_t)GCAP) {  // the store overflowed: the exact fallback redoes this row
        if (threadIdx.x == 0) cand_cnt[r] = 0x7fffffff;
        return;
    }
    // The stored groups are read ONCE, every load requested before the first is used: a thread keeps its groups of the first
    // FB trips (FB * TPB = 2048 groups; a row of the ml-25m shape stores ~1000) in registers from the histogram pass to the
    // extraction.  (Two passes of one group per thread and trip — the histogram's and the extraction's — each paid a global
    // latency per trip: 13 % of the kernel by its phase counters.)
    constexpr int FB = 4;
    const uint32_t G = s_count;
    float4 fxa[FB], fxb[FB];
    int32_t fv0[FB];
#pragma unroll
    for (int q = 0; q < FB; ++q) {
        const uint32_t g = (uint32_t)q * TPB + threadIdx.x;
        fxa[q] = fxb[q] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        fv0[q] = 0;
        if (g < G) {
            fxa[q] = g_x[2 * g];
            fxb[q] = g_x[2 * g + 1];
            fv0[q] = g_v0[g];
        }
    }
    {
        const float floor = s_thr;
#pragma unroll
        for (int q = 0; q < FB; ++q) {
            const uint32_t g = (uint32_t)q * TPB + threadIdx.x;
            if (g >= counted && g < G) {
                const float x8[8] = {fxa[q].x, fxa[q].y, fxa[q].z, fxa[q].w, fxb[q].x, fxb[q].y, fxb[q].z, fxb[q].w};
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (x8[i] >= floor && x8[i] > -INFINITY) atomicAdd(&hist[sim_bin(x8[i])], 1u);
            }
        }
        counted = max(counted, min(G, (uint32_t)FB * TPB));
        count_new_groups(floor);  // (the groups beyond the registers, if any; ends with the barrier)
    }
    block_threshold(hist, wtot, &s_thr, &s_bin, kk, eps);  // final: every value that can matter is in the histogram
    PH(10);  // the final threshold (the groups stored since the last refresh into the histogram)
    if (s_bin < bin_used) {  // an anticipated threshold overshot (see rank_after): the store may lack a neighbour -> exact fallback
        if (threadIdx.x == 0) cand_cnt[r] = 0x7fffffff;
        return;
    }
    {
        const float thr = s_thr;
        uint32_t& s_out = wtot[TPB / 64 + 3];
        if (threadIdx.x == 0) s_out = 0;
        __syncthreads();
        auto emit = [&](uint32_t m, int32_t v0, const float* x8) {  // (all lanes of the wave call it together)
            const uint32_t cnt = __popc(m);
            const uint32_t incl = wave_incl_scan(cnt);
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
            if (total > 0) {
                uint32_t base = 0;
                if (lane == 63) base = atomicAdd(&s_out, total);
                base = __builtin_amdgcn_readlane(base, 63);
                uint32_t pos = base + incl - cnt;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (m & (1u << i)) {
                        if (pos < (uint32_t)cap) {
                            out_idx[pos] = v0 + i;
                            out_apx[pos] = x8[i];
                        }
                        ++pos;
                    }
                }
            }
        };
#pragma unroll
        for (int q = 0; q < FB; ++q) {
            if ((uint32_t)q * TPB >= G) break;  // (block-uniform)
            const float x8[8] = {fxa[q].x, fxa[q].y, fxa[q].z, fxa[q].w, fxb[q].x, fxb[q].y, fxb[q].z, fxb[q].w};
            uint32_t m = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (x8[i] >= thr && x8[i] > -INFINITY) m |= 1u << i;  // (absent groups hold -inf)
            emit(m, fv0[q], x8);
        }
        for (uint32_t g0 = (uint32_t)FB * TPB; g0 < G; g0 += TPB) {
            const uint32_t g = g0 + threadIdx.x;
            float x8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int32_t v0 = 0;
            uint32_t m = 0;
            if (g < G) {
                v0 = g_v0[g];
                const float4 a = g_x[2 * g], b4 = g_x[2 * g + 1];
                x8[0] = a.x; x8[1] = a.y; x8[2] = a.z; x8[3] = a.w; x8[4] = b4.x; x8[5] = b4.y; x8[6] = b4.z; x8[7] = b4.w;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (x8[i] >= thr && x8[i] > -INFINITY) m |= 1u << i;
            }
            emit(m, v0, x8);
        }
        __syncthreads();
        if (threadIdx.x == 0) cand_cnt[r] = (int32_t)min(s_out, (uint32_t)0x7fffffff);  // (> cap: overflow, exact fallback)
    }
    PH(11);  // the shortlist out of the stored groups
    return;
}

#ifdef KNNCF_SELECT_PROFILE
void select_profile_dump() {
    unsigned long long h[16];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h));
    unsigned long long tot = 0;
    for (int i = 0; i < 16; ++i) tot += h[i];
    fprintf(stderr, "[select profile] cycles per phase (thread 0 of every block), total %.3e\n", (double)tot);
    for (int i = 0; i <= 11; ++i) fprintf(stderr, "  phase %2d: %6.2f %%\n", i, 100.0 * (double)h[i] / (double)tot);
    memset(h, 0, sizeof(h));
    hipMemcpyToSymbol(HIP_SYMBOL(g_phase), h, sizeof(h));
}
#endif

template <class ST, bool JAC>
static void launch_tail_select_t(const TailArgs& T, const ST* S, bool s_by_user, int64_t lds, int32_t n_rows, const int32_t* d_row_user, const int32_t* d_row_srow,
                                 int32_t U, int32_t kk, float eps_opnd, float eps_rest, int32_t cap, int32_t* cand_idx, float* cand_approx,
                                 int32_t* cand_cnt, float* cand_eps, int32_t* grp_v0, float* grp_x, int32_t gcap, bool anticipate, hipStream_t st) {
    const size_t smem = (size_t)TCOLS * 4 + (size_t)NBINS * 4 + (size_t)EMAX * 4 + 2 * ((size_t)EMAX * 12 + (size_t)PMAX * 2) + (2 * (TPB / 64) + 4 + 64) * 4;  // + 64 scratch cells
    static PerDeviceState lds_state;
    ensure_dynamic_lds(lds_state, (const void*)k_tail_select<ST, JAC>, smem);
    // (KNNCF_DEBUG_ANTICIPATE_SIGMA: test hook — a small margin makes the anticipated thresholds overshoot in some rows, which
    // must then be caught by the final check and rebuilt exactly; a negative one switches the anticipation off)
#ifdef KNNCF_NO_ANTICIPATION
    const float ant_sigma = -1.0f;
#else
    const char* sg = getenv("KNNCF_DEBUG_ANTICIPATE_SIGMA");
    const float ant_sigma = !anticipate ? -1.0f : sg ? (float)atof(sg) : 7.0f;
#endif
    k_tail_select<ST, JAC><<<n_rows, TPB, smem, st>>>(S, s_by_user ? 1 : 0, lds, n_rows, d_row_user, d_row_srow, T, U, kk, eps_opnd, eps_rest, cap, cand_idx, cand_approx, cand_cnt, cand_eps, grp_v0, grp_x, gcap, ant_sigma);
    KN_HIP(hipGetLastError());
#ifdef KNNCF_SELECT_PROFILE
    KN_HIP(hipStreamSynchronize(st));
    select_profile_dump();
#endif
}

void launch_tail_select(const Train& tr, const int32_t* d_colmap, const TailEntries& te, bool has_tail, const void* S, bool s_by_user, bool s_fp16, int64_t lds,
                        int32_t n_rows, const int32_t* d_row_user, int32_t k, float eps_opnd, float eps_rest, int32_t cap,
                        int32_t* cand_idx, float* cand_approx, int32_t* cand_cnt, float* cand_eps, int32_t* grp_v0, float* grp_x,
                        int32_t gcap, hipStream_t st, bool anticipate, const int32_t* d_row_srow) {
    if (n_rows <= 0) return;
    KN_REQUIRE(gcap >= 1024 && gcap % 8 == 0, KNNCF_E_INVALID, "select: group store too small");
    const int32_t U = tr.U;
    int32_t kk = k < U - 1 ? k : U - 1;
    if (kk < 1) kk = 1;
    KN_REQUIRE(!has_tail || tr.tile_stride == (int32_t)ceil_div(U, TCOLS) + 1, KNNCF_E_STATE, "select: tile table missing");
    // (KNNCF_DEBUG_SKIP_TAIL: timing-only hook — drops the sparse tail, so the neighbours are WRONG; it measures what the
    // kernel costs without the tail machinery: the panel scan, the thresholds, the group store)
    static const bool skip_tail = getenv("KNNCF_DEBUG_SKIP_TAIL") != nullptr;
    if (skip_tail) has_tail = false;
    KN_REQUIRE(!has_tail || (te.cnt && te.item && te.x && te.tail_abs && te.head_sq), KNNCF_E_STATE, "select: tail entry lists missing");
    TailArgs T{tr.u_ptr.p, tr.s_col.p, tr.s_pre.p, d_colmap, tr.i_ptr.p, tr.it_pack.p, (uint32_t)(tr.n * 4), tr.it_tile.p, tr.tile_stride, has_tail ? 1 : 0,
               te.cnt, te.item, te.x, te.tail_abs, te.head_sq, te.row_len};
    if (tr.jaccard) {
        KN_REQUIRE(te.row_len != nullptr, KNNCF_E_STATE, "select: row lengths missing");
        if (s_fp16) launch_tail_select_t<_Float16, true>(T, static_cast<const _Float16*>(S), s_by_user, lds, n_rows, d_row_user, d_row_srow, U, kk, eps_opnd, eps_rest, cap, cand_idx, cand_approx, cand_cnt, cand_eps, grp_v0, grp_x, gcap, anticipate, st);
        else launch_tail_select_t<float, true>(T, static_cast<const float*>(S), s_by_user, lds, n_rows, d_row_user, d_row_srow, U, kk, eps_opnd, eps_rest, cap, cand_idx, cand_approx, cand_cnt, cand_eps, grp_v0, grp_x, gcap, anticipate, st);
    } else {
        if (s_fp16) launch_tail_select_t<_Float16, false>(T, static_cast<const _Float16*>(S), s_by_user, lds, n_rows, d_row_user, d_row_srow, U, kk, eps_opnd, eps_rest, cap, cand_idx, cand_approx, cand_cnt, cand_eps, grp_v0, grp_x, gcap, anticipate, st);
        else launch_tail_select_t<float, false>(T, static_cast<const float*>(S), s_by_user, lds, n_rows, d_row_user, d_row_srow, U, kk, eps_opnd, eps_rest, cap, cand_idx, cand_approx, cand_cnt, cand_eps, grp_v0, grp_x, gcap, anticipate, st);
    }
}

}  // namespace knncf
