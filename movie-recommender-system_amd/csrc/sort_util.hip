// sort_util.hip — K0 plumbing: stable LSD radix sort and unique via rocPRIM (header-only, ROCm).
// Only the id compaction / CSR construction uses these; every arithmetic kernel of the path is
// hand-written (prep.hip, gemm.hip, select.hip, predict.hip).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

#include "engine.h"

namespace knncf {

void sort_pairs_u64_u32(SortWorkspace& ws, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                        uint32_t* vout, size_t n, int end_bit, hipStream_t st) {
    if (n == 0) return;
    if (end_bit < 1) end_bit = 1;
    if (end_bit > 64) end_bit = 64;
    size_t bytes = 0;
    KN_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, (unsigned)end_bit, st));
    ws.tmp.ensure(bytes);
    KN_HIP(rocprim::radix_sort_pairs(ws.tmp.p, bytes, kin, kout, vin, vout, n, 0, (unsigned)end_bit, st));
}

void sort_pairs_u32_u32(SortWorkspace& ws, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n,
                        int end_bit, hipStream_t st) {
    if (n == 0) return;
    if (end_bit < 1) end_bit = 1;
    if (end_bit > 32) end_bit = 32;
    size_t bytes = 0;
    KN_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, (unsigned)end_bit, st));
    ws.tmp.ensure(bytes);
    KN_HIP(rocprim::radix_sort_pairs(ws.tmp.p, bytes, kin, kout, vin, vout, n, 0, (unsigned)end_bit, st));
}

void sort_keys_u32(SortWorkspace& ws, const uint32_t* kin, uint32_t* kout, size_t n, hipStream_t st) {
    if (n == 0) return;
    size_t bytes = 0;
    KN_HIP(rocprim::radix_sort_keys(nullptr, bytes, kin, kout, n, 0, 32, st));
    ws.tmp.ensure(bytes);
    KN_HIP(rocprim::radix_sort_keys(ws.tmp.p, bytes, kin, kout, n, 0, 32, st));
}

size_t unique_u32(SortWorkspace& ws, const uint32_t* sorted_in, uint32_t* out, size_t n, hipStream_t st) {
    if (n == 0) return 0;
    size_t bytes = 0;
    DArr<size_t>& d_count = ws.count;
    d_count.ensure(1);
    KN_HIP(rocprim::unique(nullptr, bytes, sorted_in, out, d_count.p, n, rocprim::equal_to<uint32_t>(), st));
    ws.tmp.ensure(bytes);
    KN_HIP(rocprim::unique(ws.tmp.p, bytes, sorted_in, out, d_count.p, n, rocprim::equal_to<uint32_t>(), st));
    size_t h = 0;
    KN_HIP(hipMemcpyAsync(&h, d_count.p, sizeof(size_t), hipMemcpyDeviceToHost, st));
    KN_HIP(hipStreamSynchronize(st));
    return h;
}

}  // namespace knncf
