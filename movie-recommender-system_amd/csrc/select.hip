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

}  // namespace knncf
