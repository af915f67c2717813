// select.hip — K5 sparse tail + K6 top-k neighbour select, fused per similarity row.
//
// The dense MFMA GEMM (gemm.hip) leaves S[u][v] = sum over the H most-rated items.  For row u this
// kernel (one 1024-thread workgroup per row)
//   1. adds the SPARSE TAIL: for every tail item i rated by u and every rater v of i,
//      pre(u,i) * pre(v,i) is accumulated with LDS float atomics into a column tile of the row
//      (24 576 columns = 96 KiB of LDS at a time; the rater lists are sorted by user, so each list is
//      swept once across the tiles with a per-entry cursor kept in LDS); the tile is added to S and
//      written back (coalesced);
//   2. builds a 4096-bin LDS histogram of the final row in the same pass and finds the bin holding
//      the k-th largest value a_k by a block-wide suffix scan;
//   3. compacts every v with S[u][v] >= (bin lower edge) - 2 eps into the row's shortlist.
// With |S[u][v] - s_uv| <= eps for every pair, every true top-k member v satisfies
// S[u][v] >= a_k - 2 eps, so the shortlist provably contains the exact top-k; rerank.hip decides.
// HBM-bound: S is read + written once and re-read once (second pass mostly L2/MALL); the tail's
// per-pair products never touch HBM atomics.
#include <math.h>

#include "engine.h"

namespace knncf {

static constexpr int TPB = 1024;
static constexpr int NBINS = 4096;
static constexpr int TCOLS = 20480;  // columns of the row held in LDS at a time (80 KiB)
static constexpr int EMAX = 2048;    // row positions whose tail cursors are held in LDS at a time

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

__global__ void __launch_bounds__(TPB) k_tail_select(float* __restrict__ S, int64_t ld, int32_t n_rows,
                                                     const int32_t* __restrict__ row_user, TailArgs T, int32_t U,
                                                     int32_t kk, float eps_base, int32_t cap, int32_t* __restrict__ cand_idx,
                                                     float* __restrict__ cand_approx, int32_t* __restrict__ cand_cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);                // [TCOLS]
    uint32_t* hist = reinterpret_cast<uint32_t*>(tile + TCOLS);   // [NBINS]
    int64_t* e_cur = reinterpret_cast<int64_t*>(hist + NBINS);    // [EMAX] cursor into it_user / it_pre
    int64_t* e_end = e_cur + EMAX;                                // [EMAX]
    float* e_x = reinterpret_cast<float*>(e_end + EMAX);          // [EMAX] pre(u, item)
    __shared__ uint32_t part[TPB];
    __shared__ float s_thr;
    __shared__ uint32_t s_count;
    __shared__ int32_t s_ne;
    const int32_t r = blockIdx.x;
    if (r >= n_rows) return;
    const int32_t u = row_user[r];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ub = T.u_ptr[u], ue = T.u_ptr[u + 1];
    const float eps = row_eps(eps_base, ue - ub);
    float* row = S + (int64_t)r * ld;
    for (int b = threadIdx.x; b < NBINS; b += TPB) hist[b] = 0;
    if (threadIdx.x == 0) s_count = 0;

    // ---- pass 1: tail accumulation (LDS), write-back, histogram --------------------------------
    // the row is taken EMAX positions at a time (one chunk for all but the heaviest raters); every
    // chunk sweeps the column tiles once, the last one also histograms the final values
    int64_t cb = ub;
    bool last_chunk;
    do {
        const int64_t ce = min(ue, cb + EMAX);
        last_chunk = ce >= ue;
        __syncthreads();
        if (threadIdx.x == 0) s_ne = 0;
        __syncthreads();
        if (T.has_tail) {
            for (int64_t p = cb + threadIdx.x; p < ce; p += TPB) {
                const int32_t item = T.s_col[p];
                if (T.colmap[item] < 0) {
                    const int32_t slot = atomicAdd(&s_ne, 1);
                    e_cur[slot] = T.i_ptr[item];
                    e_end[slot] = T.i_ptr[item + 1];
                    e_x[slot] = (float)T.s_pre[p];
                }
            }
        }
        __syncthreads();
        const int32_t ne = s_ne;
        for (int32_t t0 = 0; t0 < U; t0 += TCOLS) {
            const int32_t t1 = min(U, t0 + TCOLS);
            if (ne > 0) {
                for (int32_t c = threadIdx.x; c < t1 - t0; c += TPB) tile[c] = 0.0f;
                __syncthreads();
                for (int32_t e = wave; e < ne; e += TPB / 64) {  // one wave per tail entry
                    int64_t q = e_cur[e];
                    const int64_t qe = e_end[e];
                    const float x = e_x[e];
                    while (q < qe) {
                        const int64_t qq = q + lane;
                        int32_t v = 0x7fffffff;
                        float y = 0.f;
                        if (qq < qe) {
                            v = T.it_user[qq];
                            y = T.it_pre[qq];
                        }
                        const bool in = v < t1;
                        if (in) atomicAdd(&tile[v - t0], x * y);
                        const int n_in = __popcll(__ballot(in));
                        q += n_in;
                        if (n_in < 64) break;  // reached the tile's end (raters ascend) or the list's end
                    }
                    if (lane == 0) e_cur[e] = q;
                }
                __syncthreads();
            }
            if (ne > 0 || last_chunk) {
                // S += tile (if something was accumulated); histogram of the final values on the last chunk
                for (int32_t c = threadIdx.x * 4; c < t1 - t0; c += TPB * 4) {
                    const int32_t v0 = t0 + c;
                    if (v0 + 3 < t1) {
                        float4 x = *reinterpret_cast<const float4*>(row + v0);
                        if (ne > 0) {
                            x.x += tile[c]; x.y += tile[c + 1]; x.z += tile[c + 2]; x.w += tile[c + 3];
                            *reinterpret_cast<float4*>(row + v0) = x;
                        }
                        if (last_chunk) {
                            if (v0 + 0 != u) atomicAdd(&hist[sim_bin(x.x)], 1u);
                            if (v0 + 1 != u) atomicAdd(&hist[sim_bin(x.y)], 1u);
                            if (v0 + 2 != u) atomicAdd(&hist[sim_bin(x.z)], 1u);
                            if (v0 + 3 != u) atomicAdd(&hist[sim_bin(x.w)], 1u);
                        }
                    } else {
                        for (int32_t v = v0; v < t1; ++v) {
                            float x = row[v];
                            if (ne > 0) {
                                x += tile[v - t0];
                                row[v] = x;
                            }
                            if (last_chunk && v != u) atomicAdd(&hist[sim_bin(x)], 1u);
                        }
                    }
                }
                __syncthreads();
            }
        }
        cb = ce;
    } while (!last_chunk);

    // ---- k-th bin: suffix counts, thread t owns bins [4 t, 4 t + 4) --------------------------------
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
    uint32_t above = (threadIdx.x + 1 < TPB) ? part[threadIdx.x + 1] : 0;  // count in higher bins
    if (above < (uint32_t)kk && part[threadIdx.x] >= (uint32_t)kk) {
        uint32_t c = above;
        int b = threadIdx.x * PER + PER - 1;
        for (; b > (int)threadIdx.x * PER; --b) {
            c += hist[b];
            if (c >= (uint32_t)kk) break;
        }
        // every value in bin b is >= its lower edge (up to one float rounding of x + 1)
        float edge = (float)b / (float)(NBINS / 2) - 1.0f;
        s_thr = (b == 0) ? -INFINITY : edge - 2.0f * eps - 1e-6f;
    }
    __syncthreads();

    // ---- pass 2: compaction of the error band ---------------------------------------------------
    const float thr = s_thr;
    int32_t* out_idx = cand_idx + (int64_t)r * cap;
    float* out_apx = cand_approx ? cand_approx + (int64_t)r * cap : nullptr;
    const int32_t U4 = U & ~3;
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

void launch_tail_select(const Train& tr, const int32_t* d_colmap, bool has_tail, float* S, int64_t lds,
                        int32_t n_rows, const int32_t* d_row_user, int32_t k, float eps, int32_t cap,
                        int32_t* cand_idx, float* cand_approx, int32_t* cand_cnt, hipStream_t st) {
    if (n_rows <= 0) return;
    const int32_t U = tr.U;
    int32_t kk = k < U - 1 ? k : U - 1;
    if (kk < 1) kk = 1;
    TailArgs T{tr.u_ptr.p, tr.s_col.p, tr.s_pre.p, d_colmap, tr.i_ptr.p, tr.it_user.p, tr.it_pre.p, has_tail ? 1 : 0};
    const size_t smem = (size_t)TCOLS * 4 + (size_t)NBINS * 4 + (size_t)EMAX * (8 + 8 + 4);
    static bool attr_set = false;
    if (!attr_set) {
        KN_HIP(hipFuncSetAttribute((const void*)k_tail_select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    k_tail_select<<<n_rows, TPB, smem, st>>>(S, lds, n_rows, d_row_user, T, U, kk, eps, cap, cand_idx, cand_approx, cand_cnt);
    KN_HIP(hipGetLastError());
}

}  // namespace knncf
