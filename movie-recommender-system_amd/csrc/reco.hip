// reco.hip — recommendations shared/predictions.scala:651-674: one user x every train item the user
// has not rated, predicted by any of the predictors, ordered by (prediction descending, raw item id ascending),
// first n.  The predictions come from predict.hip (one batch of I rows); this file marks the rated items and
// produces the order with two stable radix sorts (rocPRIM): by raw id, then by the order-preserving image of the
// fp64 prediction — ties keep the id order, exactly the reference's strict total order.
#include "engine.h"

namespace knncf {

static constexpr int TPB = 256;

__global__ void k_reco_rows(int32_t I, int32_t user_raw, const int32_t* __restrict__ iid, int32_t* __restrict__ users,
                            int32_t* __restrict__ items, uint8_t* __restrict__ rated) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= I) return;
    users[j] = user_raw;
    items[j] = iid[j];
    rated[j] = 0;
}

__global__ void k_reco_mark(const int64_t* __restrict__ u_ptr, const int32_t* __restrict__ s_col, int32_t du,
                            uint8_t* __restrict__ rated) {
    const int64_t b = u_ptr[du], e = u_ptr[du + 1];
    for (int64_t p = b + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < e; p += (int64_t)gridDim.x * blockDim.x) rated[s_col[p]] = 1;
}

__global__ void k_reco_id_keys(int32_t I, const int32_t* __restrict__ iid, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= I) return;
    key[j] = (uint64_t)(uint32_t)(iid[j] ^ 0x80000000);  // signed ids in ascending order
    val[j] = (uint32_t)j;
}

// ascending key <=> descending prediction; rated items last.  -0.0 and +0.0 compare equal in the reference's
// `x._2 == y._2`, so both map to the same key.
__global__ void k_reco_pred_keys(int32_t I, const uint32_t* __restrict__ by_id, const double* __restrict__ pred,
                                 const uint8_t* __restrict__ rated, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= I) return;
    const uint32_t d = by_id[j];
    double p = pred[d];
    if (p == 0.0) p = 0.0;
    const uint64_t bits = (uint64_t)__double_as_longlong(p);
    const uint64_t asc = (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);  // order-preserving for non-NaN doubles
    key[j] = rated[d] ? ~0ull : ~asc;
    val[j] = d;
}

__global__ void k_reco_take(int32_t m, const uint32_t* __restrict__ order, const int32_t* __restrict__ iid,
                            const double* __restrict__ pred, int32_t* __restrict__ out_items, double* __restrict__ out_preds) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t d = order[j];
    out_items[j] = iid[d];
    out_preds[j] = pred[d];
}

void launch_reco_rows(const Train& tr, int32_t user_raw, int32_t du, int32_t* d_users, int32_t* d_items, uint8_t* d_rated, hipStream_t st) {
    k_reco_rows<<<(unsigned)ceil_div(tr.I, TPB), TPB, 0, st>>>(tr.I, user_raw, tr.iid.p, d_users, d_items, d_rated);
    if (du >= 0) k_reco_mark<<<8, TPB, 0, st>>>(tr.u_ptr.p, tr.s_col.p, du, d_rated);
    KN_HIP(hipGetLastError());
}

void launch_reco_order(const Train& tr, SortWorkspace& ws, const double* d_pred, const uint8_t* d_rated, uint64_t* k_a, uint64_t* k_b,
                       uint32_t* v_a, uint32_t* v_b, hipStream_t st) {
    const int32_t I = tr.I;
    k_reco_id_keys<<<(unsigned)ceil_div(I, TPB), TPB, 0, st>>>(I, tr.iid.p, k_a, v_a);
    sort_pairs_u64_u32(ws, k_a, k_b, v_a, v_b, I, 32, st);                       // v_b: dense items by ascending raw id
    k_reco_pred_keys<<<(unsigned)ceil_div(I, TPB), TPB, 0, st>>>(I, v_b, d_pred, d_rated, k_a, v_a);
    sort_pairs_u64_u32(ws, k_a, k_b, v_a, v_b, I, 64, st);                       // stable: ties keep the id order
    KN_HIP(hipGetLastError());
}

void launch_reco_take(const Train& tr, int32_t m, const uint32_t* d_order, const double* d_pred, int32_t* d_items, double* d_preds, hipStream_t st) {
    if (m <= 0) return;
    k_reco_take<<<(unsigned)ceil_div(m, TPB), TPB, 0, st>>>(m, d_order, tr.iid.p, d_pred, d_items, d_preds);
    KN_HIP(hipGetLastError());
}

}  // namespace knncf
