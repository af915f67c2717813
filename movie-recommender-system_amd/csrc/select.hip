// select.hip — K5 sparse tail + K6 top-k neighbour select, fused per similarity row.
//
// The dense MFMA GEMM (gemm.hip) leaves S[u][v] = sum over the H most-rated items (fp16 or fp32
// storage).  For row u this kernel (one 1024-thread workgroup per row) walks the row ONCE, a column tile
// (24 576 columns = 96 KiB of LDS) at a time:
//   1. SPARSE TAIL: for every tail item i rated by u and every rater v of i inside the tile,
//      pre(u,i) * pre(v,i) is accumulated in Q7.24 fixed point with integer LDS atomics (the rater lists
//      are sorted by user, so each list is swept once across the tiles with a per-entry cursor kept in
//      LDS; rows with more than EMAX ratings re-derive their cursors per tile by binary search);
//   2. the tile's final values S + tail stay in registers and enter a cumulative 4096-bin LDS histogram;
//      the bin holding the k-th largest value SEEN SO FAR gives a threshold that can only rise as more
//      columns are seen, so every v of the tile with value >= (bin lower edge - 2 eps) is appended to a
//      provisional shortlist;
//   3. after the last tile the threshold is final and the provisional list is compacted in place.
// With |S[u][v] - s_uv| <= eps for every pair, every true top-k member v satisfies
// S[u][v] >= a_k - 2 eps (a_k = k-th largest value of the row), so the shortlist provably contains
// the exact top-k; rerank.hip decides.  HBM-bound: S is read exactly once and never written back; the
// tail's per-pair products never touch HBM atomics.
#include <math.h>
#include <stdlib.h>

#include "engine.h"

namespace knncf {

// One 1024-thread workgroup per CU with the largest tile that fits.  Measured alternative: two 512-thread
// workgroups per CU with 48 KiB tiles (twice as many tiles per row) ran 1.7x SLOWER — the cost is per tile
// (barrier-separated phases), so fewer, larger tiles win.
static constexpr int TPB = 1024;
static constexpr int NBINS = 4096;
static constexpr int TCOLS = 32768;  // columns of the row held in LDS at a time (128 KiB)
static constexpr int CPT = TCOLS / TPB;  // columns per thread per tile (24 = 3 groups of 8)
static constexpr int EMAX = 1024;    // row positions whose tail cursors (12 B each) are held in LDS at a time
static constexpr int MAX_PER_THREAD = 16;  // provisional entries per thread in the final compaction
static constexpr int TAIL_ILP = 4;         // tail entries a wave keeps in flight
static constexpr int TAIL_CH = 2;          // 64-rater pieces requested ahead per entry
// the tail is accumulated in Q7.24 fixed point with integer LDS atomics (ds_add_u32; the float form
// ds_add_f32 measured ~1.4x slower here): |sum| <= 1, each product is quantised with error <= 2^-25,
// which the per-common-item term of row_eps covers
static constexpr float TAIL_FIX = 16777216.0f;
static constexpr float TAIL_UNFIX = 1.0f / 16777216.0f;

__device__ __forceinline__ int sim_bin(float x) {
    int b = (int)floorf((x + 1.0f) * (NBINS / 2));
    return min(max(b, 0), NBINS - 1);
}

// rigorous bound on |S[u][v] - s_uv|: operand rounding (eps_base) plus fp32 accumulation — adding a
// zero product is exact, so at most 2 roundings per common item, each <= 2^-24 of a partial sum <= 1.01
__device__ __forceinline__ float row_eps(float eps_base, int64_t row_len) {
    return eps_base + (float)row_len * 2.0f * 6.1e-8f;
}

struct TailArgs {
    const int64_t* u_ptr;  // user-major rows
    const int32_t* s_col;
    const double* s_pre;
    const int32_t* colmap;  // < 0: tail item
    const int64_t* i_ptr;   // item-major rows, raters ascending
    const int32_t* it_user;
    const float* it_pre;
    int32_t has_tail;
};

// threshold from the cumulative histogram: lower edge of the bin holding the kk-th largest value seen so
// far, minus 2 eps (-inf while fewer than kk values have been seen).  Thread t owns bins [4t, 4t+4).
__device__ __forceinline__ void block_threshold(const uint32_t* hist, uint32_t* wtot, float* s_thr, int32_t kk, float eps) {
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
    if (threadIdx.x == 0) *s_thr = -INFINITY;
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
        const float edge = (float)b / (float)(NBINS / 2) - 1.0f;
        *s_thr = (b == 0) ? -INFINITY : edge - 2.0f * eps - 1e-6f;
    }
    __syncthreads();
}

// 8 consecutive panel entries starting at a multiple of 8 (rows are padded to ld, a multiple of 128)
__device__ __forceinline__ void load8(const float* p, float* out) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w; out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
}
__device__ __forceinline__ void load8(const _Float16* p, float* out) {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    const h8 v = *reinterpret_cast<const h8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
}

template <class ST>
__global__ void __launch_bounds__(TPB) k_tail_select(const ST* __restrict__ S, int64_t ld, int32_t n_rows,
                                                     const int32_t* __restrict__ row_user, TailArgs T, int32_t U,
                                                     int32_t kk, float eps_base, int32_t cap, int32_t* __restrict__ cand_idx,
                                                     float* __restrict__ cand_approx, int32_t* __restrict__ cand_cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* itile = reinterpret_cast<int32_t*>(smem);            // [TCOLS] tail accumulator, Q7.24
    uint32_t* hist = reinterpret_cast<uint32_t*>(itile + TCOLS);  // [NBINS]
    uint32_t* e_cur = hist + NBINS;                               // [EMAX] cursor into it_user / it_pre (n < 2^32)
    uint32_t* e_end = e_cur + EMAX;                               // [EMAX]
    float* e_x = reinterpret_cast<float*>(e_end + EMAX);          // [EMAX] pre(u, item)
    __shared__ uint32_t wtot[TPB / 64];
    __shared__ float s_thr;
    __shared__ uint32_t s_count;
    __shared__ int32_t s_ne;
    const int32_t r = blockIdx.x;
    if (r >= n_rows) return;
    const int32_t u = row_user[r];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ub = T.u_ptr[u], ue = T.u_ptr[u + 1];
    const float eps = row_eps(eps_base, ue - ub);
    const ST* row = S + (int64_t)r * ld;
    int32_t* out_idx = cand_idx + (int64_t)r * cap;
    float* out_apx = cand_approx + (int64_t)r * cap;
    for (int b = threadIdx.x; b < NBINS; b += TPB) hist[b] = 0;
    if (threadIdx.x == 0) { s_count = 0; s_ne = 0; }
    __syncthreads();

    // tail entries of the row.  Rows of up to EMAX ratings (all but the heaviest raters) collect them once and
    // keep one cursor per entry in LDS across the tiles; longer rows take EMAX positions at a time and
    // re-derive the cursors of every chunk for every tile by binary search.
    const bool single = (ue - ub) <= EMAX;
    const int n_chunks = T.has_tail ? (int)((ue - ub + EMAX - 1) / EMAX) : 0;
    if (single && n_chunks > 0) {
        for (int64_t p = ub + threadIdx.x; p < ue; p += TPB) {
            const int32_t item = T.s_col[p];
            if (T.colmap[item] < 0) {
                const int32_t slot = atomicAdd(&s_ne, 1);
                e_cur[slot] = (uint32_t)T.i_ptr[item];
                e_end[slot] = (uint32_t)T.i_ptr[item + 1];
                e_x[slot] = (float)T.s_pre[p];
            }
        }
        __syncthreads();
    }
    const bool any_tail = n_chunks > 0 && (!single || s_ne > 0);
    if (any_tail) {  // the accumulator starts clean; afterwards every tile's read-out clears what it reads
        for (int32_t c = threadIdx.x; c < TCOLS; c += TPB) itile[c] = 0;
        __syncthreads();
    }

    int tile_no = 0;
    for (int32_t t0 = 0; t0 < U; t0 += TCOLS, ++tile_no) {
        const int32_t t1 = min(U, t0 + TCOLS);
        // this thread's 24 columns of the tile: group j covers columns t0 + 8 (tid + 1024 j) .. + 7.  The loads
        // are issued first so that their HBM latency hides behind the tail accumulation
        float sx[CPT];
#pragma unroll
        for (int j = 0; j < CPT / 8; ++j) {
            const int64_t v0 = (int64_t)t0 + 8 * (threadIdx.x + TPB * j);
            if (v0 < ld) load8(row + v0, &sx[8 * j]);
            else {
#pragma unroll
                for (int i = 0; i < 8; ++i) sx[8 * j + i] = 0.f;
            }
        }
        if (any_tail) {
            for (int ch = 0; ch < n_chunks; ++ch) {
                if (!single) {  // collect this chunk's entries with cursors at the tile's first column
                    __syncthreads();
                    if (threadIdx.x == 0) s_ne = 0;
                    __syncthreads();
                    const int64_t cb = ub + (int64_t)ch * EMAX, ce = min(ue, cb + EMAX);
                    for (int64_t p = cb + threadIdx.x; p < ce; p += TPB) {
                        const int32_t item = T.s_col[p];
                        if (T.colmap[item] < 0) {
                            int64_t lo = T.i_ptr[item], hi = T.i_ptr[item + 1];
                            const int64_t end = hi;
                            while (lo < hi) {
                                const int64_t mid = (lo + hi) >> 1;
                                if (T.it_user[mid] < t0) lo = mid + 1;
                                else hi = mid;
                            }
                            const int32_t slot = atomicAdd(&s_ne, 1);
                            e_cur[slot] = (uint32_t)lo;
                            e_end[slot] = (uint32_t)end;
                            e_x[slot] = (float)T.s_pre[p];
                        }
                    }
                    __syncthreads();
                }
                const int32_t ne = s_ne;
                // one wave per tail entry; TAIL_ILP entries x TAIL_CH 64-rater pieces are requested before any
                // is consumed (the loop is latency-bound: rater lists are short and come from L2/HBM)
                for (int32_t e0 = wave; e0 < ne; e0 += TAIL_ILP * (TPB / 64)) {
                    uint32_t q[TAIL_ILP], qe[TAIL_ILP];
                    float x[TAIL_ILP];
                    bool live[TAIL_ILP];
#pragma unroll
                    for (int j = 0; j < TAIL_ILP; ++j) {
                        const int32_t e = e0 + j * (TPB / 64);
                        live[j] = e < ne;
                        q[j] = live[j] ? e_cur[e] : 0;
                        qe[j] = live[j] ? e_end[e] : 0;
                        x[j] = live[j] ? e_x[e] : 0.f;
                        live[j] = live[j] && q[j] < qe[j];
                    }
                    bool any_live = true;
                    while (any_live) {
                        int32_t v[TAIL_ILP][TAIL_CH];
                        float y[TAIL_ILP][TAIL_CH];
#pragma unroll
                        for (int j = 0; j < TAIL_ILP; ++j)
#pragma unroll
                            for (int k = 0; k < TAIL_CH; ++k) {
                                const uint32_t qq = q[j] + 64 * k + lane;
                                v[j][k] = 0x7fffffff;
                                y[j][k] = 0.f;
                                if (live[j] && qq < qe[j]) {
                                    v[j][k] = T.it_user[qq];
                                    y[j][k] = T.it_pre[qq];
                                }
                            }
                        any_live = false;
#pragma unroll
                        for (int j = 0; j < TAIL_ILP; ++j) {
                            if (!live[j]) continue;  // wave-uniform
#pragma unroll
                            for (int k = 0; k < TAIL_CH; ++k) {
                                if (!live[j]) break;
                                const bool in = v[j][k] < t1;
                                if (in) atomicAdd(&itile[v[j][k] - t0], __float2int_rn(x[j] * y[j][k] * TAIL_FIX));
                                const int n_in = __popcll(__ballot(in));
                                q[j] += n_in;
                                // fewer than 64 inside: reached the tile's end (raters ascend) or the list's end
                                live[j] = (n_in == 64) && q[j] < qe[j];
                            }
                            any_live = any_live || live[j];
                        }
                    }
                    if (single) {
#pragma unroll
                        for (int j = 0; j < TAIL_ILP; ++j) {
                            const int32_t e = e0 + j * (TPB / 64);
                            if (e < ne && lane == 0) e_cur[e] = q[j];
                        }
                    }
                }
                __syncthreads();
            }
            // read the accumulator out (vectorised) and clear it for the next tile: only this thread touches
            // these cells between the barriers
#pragma unroll
            for (int j = 0; j < CPT / 8; ++j) {
                const int32_t c0 = 8 * (threadIdx.x + TPB * j);
                int4* cell = reinterpret_cast<int4*>(itile + c0);
                const int4 a = cell[0], b = cell[1];
                cell[0] = make_int4(0, 0, 0, 0);
                cell[1] = make_int4(0, 0, 0, 0);
                sx[8 * j + 0] += (float)a.x * TAIL_UNFIX; sx[8 * j + 1] += (float)a.y * TAIL_UNFIX;
                sx[8 * j + 2] += (float)a.z * TAIL_UNFIX; sx[8 * j + 3] += (float)a.w * TAIL_UNFIX;
                sx[8 * j + 4] += (float)b.x * TAIL_UNFIX; sx[8 * j + 5] += (float)b.y * TAIL_UNFIX;
                sx[8 * j + 6] += (float)b.z * TAIL_UNFIX; sx[8 * j + 7] += (float)b.w * TAIL_UNFIX;
            }
        }
        if (tile_no == 0) {
            // First tile: a 1/8 subsample bootstraps a valid threshold (the k-th largest of a subset is a lower
            // bound of the k-th largest of the row); then only values above it enter the cumulative histogram,
            // which keeps the LDS atomics off the crowded bins near 0.
            float floor_thr = -INFINITY;
            for (int pass = 0; pass < 2; ++pass) {
                if (((threadIdx.x & 7) == 0) == (pass == 0)) {
#pragma unroll
                    for (int j = 0; j < CPT / 8; ++j) {
                        const int32_t v0 = t0 + 8 * (threadIdx.x + TPB * j);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int32_t v = v0 + i;
                            const float x = sx[8 * j + i];
                            if (v < t1 && v != u && x >= floor_thr) atomicAdd(&hist[sim_bin(x)], 1u);
                        }
                    }
                }
                __syncthreads();
                block_threshold(hist, wtot, &s_thr, kk, eps);
                floor_thr = s_thr;
            }
            const float thr = s_thr;
#pragma unroll
            for (int j = 0; j < CPT / 8; ++j) {
                const int32_t v0 = t0 + 8 * (threadIdx.x + TPB * j);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int32_t v = v0 + i;
                    const float x = sx[8 * j + i];
                    if (v < t1 && v != u && x >= thr) {
                        const uint32_t pos = atomicAdd(&s_count, 1u);
                        if (pos < (uint32_t)cap) {
                            out_idx[pos] = v;
                            out_apx[pos] = x;
                        }
                    }
                }
            }
        } else {
            // Later tiles: one fused pass with the threshold known so far (it can only rise; a stale one just lets a
            // few more provisional entries through).  The kernel is VALU-bound here (2.6e10 panel entries per
            // step), so a group of 8 is first rejected by its maximum; survivors are rare.
            const float thr = s_thr;
#pragma unroll
            for (int j = 0; j < CPT / 8; ++j) {
                const float* x8 = &sx[8 * j];
                const float m = fmaxf(fmaxf(fmaxf(x8[0], x8[1]), fmaxf(x8[2], x8[3])), fmaxf(fmaxf(x8[4], x8[5]), fmaxf(x8[6], x8[7])));
                if (m >= thr) {
                    const int32_t v0 = t0 + 8 * (threadIdx.x + TPB * j);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int32_t v = v0 + i;
                        const float x = x8[i];
                        if (x >= thr && v < t1 && v != u) {
                            atomicAdd(&hist[sim_bin(x)], 1u);
                            const uint32_t pos = atomicAdd(&s_count, 1u);
                            if (pos < (uint32_t)cap) {
                                out_idx[pos] = v;
                                out_apx[pos] = x;
                            }
                        }
                    }
                }
            }
            if (tile_no == 1) {  // one refresh after 2 tiles (~30 % of the row): costs three barriers, tightens the rest
                __syncthreads();
                block_threshold(hist, wtot, &s_thr, kk, eps);
            }
        }
        __syncthreads();
    }


    // ---- compaction of the provisional list by the final threshold (in place) ------------------------
    const uint32_t prov = s_count;
    if (prov > (uint32_t)cap) {  // provisional overflow: the exact fallback redoes this row
        if (threadIdx.x == 0) cand_cnt[r] = (int32_t)min(prov, (uint32_t)0x7fffffff);
        return;
    }
    block_threshold(hist, wtot, &s_thr, kk, eps);  // final: the whole row is in the histogram
    const float thr = s_thr;
    int32_t kv[MAX_PER_THREAD];
    float kx[MAX_PER_THREAD];
#pragma unroll
    for (int j = 0; j < MAX_PER_THREAD; ++j) {
        const uint32_t i = threadIdx.x + (uint32_t)j * TPB;
        kv[j] = -1;
        kx[j] = 0.f;
        if (i < prov) {
            kv[j] = out_idx[i];
            kx[j] = out_apx[i];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAX_PER_THREAD; ++j) {
        if (kv[j] >= 0 && kx[j] >= thr) {
            const uint32_t pos = atomicAdd(&s_count, 1u);
            out_idx[pos] = kv[j];
            out_apx[pos] = kx[j];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) cand_cnt[r] = (int32_t)s_count;
}

template <class ST>
static void launch_tail_select_t(const TailArgs& T, const ST* S, int64_t lds, int32_t n_rows, const int32_t* d_row_user,
                                 int32_t U, int32_t kk, float eps, int32_t cap, int32_t* cand_idx, float* cand_approx,
                                 int32_t* cand_cnt, hipStream_t st) {
    const size_t smem = (size_t)TCOLS * 4 + (size_t)NBINS * 4 + (size_t)EMAX * (4 + 4 + 4);
    static bool attr_set = false;
    if (!attr_set) {
        KN_HIP(hipFuncSetAttribute((const void*)k_tail_select<ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    k_tail_select<ST><<<n_rows, TPB, smem, st>>>(S, lds, n_rows, d_row_user, T, U, kk, eps, cap, cand_idx, cand_approx, cand_cnt);
    KN_HIP(hipGetLastError());
}

void launch_tail_select(const Train& tr, const int32_t* d_colmap, bool has_tail, const void* S, bool s_fp16, int64_t lds,
                        int32_t n_rows, const int32_t* d_row_user, int32_t k, float eps, int32_t cap,
                        int32_t* cand_idx, float* cand_approx, int32_t* cand_cnt, hipStream_t st) {
    if (n_rows <= 0) return;
    KN_REQUIRE(cap <= TPB * MAX_PER_THREAD, KNNCF_E_INVALID, "select: shortlist store larger than the compaction window");
    const int32_t U = tr.U;
    int32_t kk = k < U - 1 ? k : U - 1;
    if (kk < 1) kk = 1;
    TailArgs T{tr.u_ptr.p, tr.s_col.p, tr.s_pre.p, d_colmap, tr.i_ptr.p, tr.it_user.p, tr.it_pre.p, has_tail ? 1 : 0};
    if (s_fp16) launch_tail_select_t(T, static_cast<const _Float16*>(S), lds, n_rows, d_row_user, U, kk, eps, cap, cand_idx, cand_approx, cand_cnt, st);
    else launch_tail_select_t(T, static_cast<const float*>(S), lds, n_rows, d_row_user, U, kk, eps, cap, cand_idx, cand_approx, cand_cnt, st);
}

}  // namespace knncf
