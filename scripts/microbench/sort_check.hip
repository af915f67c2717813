// sort_check.hip — checks sort_util.hip's radix sort / unique against std::stable_sort on the host and times the
// two sorts of the ml-25m shape.  Build + run (GPU box):
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I movie-recommender-system_amd/csrc scripts/microbench/sort_check.hip \
//         -L movie-recommender-system_amd -lknncf -Wl,-rpath,$PWD/movie-recommender-system_amd -o /tmp/sort_check && /tmp/sort_check
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

#include "engine.h"

using namespace knncf;

static int failures = 0;

template <class K>
static void check_pairs(SortWorkspace& ws, size_t n, int bits, uint64_t seed, bool skewed) {
    std::mt19937_64 rng(seed);
    std::vector<K> k(n);
    std::vector<uint32_t> v(n);
    const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    for (size_t i = 0; i < n; ++i) {
        uint64_t x = rng();
        if (skewed) x = (x % 7 == 0) ? x : (x & 0x3ull) * 0x0101010101010101ull;  // long equal runs
        k[i] = (K)(x & mask);
        v[i] = (uint32_t)i;
    }
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
    DArr<K> dk, dko;
    DArr<uint32_t> dv, dvo;
    dk.alloc(n); dko.alloc(n); dv.alloc(n); dvo.alloc(n);
    KN_HIP(hipMemcpy(dk.p, k.data(), n * sizeof(K), hipMemcpyHostToDevice));
    KN_HIP(hipMemcpy(dv.p, v.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (sizeof(K) == 8) sort_pairs_u64_u32(ws, (const uint64_t*)dk.p, (uint64_t*)dko.p, dv.p, dvo.p, n, bits, 0);
    else sort_pairs_u32_u32(ws, (const uint32_t*)dk.p, (uint32_t*)dko.p, dv.p, dvo.p, n, bits, 0);
    KN_HIP(hipDeviceSynchronize());
    std::vector<K> ko(n);
    std::vector<uint32_t> vo(n);
    KN_HIP(hipMemcpy(ko.data(), dko.p, n * sizeof(K), hipMemcpyDeviceToHost));
    KN_HIP(hipMemcpy(vo.data(), dvo.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i)
        if (vo[i] != order[i] || ko[i] != k[order[i]]) ++bad;
    if (bad) {
        ++failures;
        printf("FAIL  pairs<%d> n=%zu bits=%d skew=%d: %zu mismatches\n", (int)sizeof(K) * 8, n, bits, (int)skewed, bad);
    }
}

static void check_keys_unique(SortWorkspace& ws, size_t n, uint32_t range, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::vector<uint32_t> k(n);
    for (auto& x : k) x = (uint32_t)(rng() % range) * 2654435761u;
    DArr<uint32_t> dk, dko, du;
    dk.alloc(n); dko.alloc(n); du.alloc(n);
    KN_HIP(hipMemcpy(dk.p, k.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    sort_keys_u32(ws, dk.p, dko.p, n, 0);
    const size_t cnt = unique_u32(ws, dko.p, du.p, n, 0);
    std::sort(k.begin(), k.end());
    std::vector<uint32_t> ko(n), uo(cnt);
    KN_HIP(hipMemcpy(ko.data(), dko.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    KN_HIP(hipMemcpy(uo.data(), du.p, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost));
    bool ok = ko == k;
    k.erase(std::unique(k.begin(), k.end()), k.end());
    ok = ok && uo == k;
    if (!ok) {
        ++failures;
        printf("FAIL  keys/unique n=%zu range=%u (got %zu distinct, want %zu)\n", n, range, cnt, k.size());
    }
}

template <class K>
static void time_pairs(SortWorkspace& ws, size_t n, int bits, const char* what) {
    std::mt19937_64 rng(7);
    std::vector<K> k(n);
    const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    for (auto& x : k) x = (K)(rng() & mask);
    DArr<K> dk, dko;
    DArr<uint32_t> dv, dvo;
    dk.alloc(n); dko.alloc(n); dv.alloc(n); dvo.alloc(n);
    KN_HIP(hipMemcpy(dk.p, k.data(), n * sizeof(K), hipMemcpyHostToDevice));
    KN_HIP(hipMemset(dv.p, 0, n * sizeof(uint32_t)));
    hipEvent_t e0, e1;
    KN_HIP(hipEventCreate(&e0)); KN_HIP(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep) {
        KN_HIP(hipEventRecord(e0, 0));
        if (sizeof(K) == 8) sort_pairs_u64_u32(ws, (const uint64_t*)dk.p, (uint64_t*)dko.p, dv.p, dvo.p, n, bits, 0);
        else sort_pairs_u32_u32(ws, (const uint32_t*)dk.p, (uint32_t*)dko.p, dv.p, dvo.p, n, bits, 0);
        KN_HIP(hipEventRecord(e1, 0));
        KN_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        KN_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 3) printf("time  %s: n=%zu bits=%d  %.3f ms  (%.3f ms per 8-bit pass)\n", what, n, bits, ms, ms / ((bits + 7) / 8));
    }
}

int main() {
    try {
        SortWorkspace ws;
        const size_t sizes[] = {1, 2, 63, 64, 65, 3071, 3072, 3073, 4095, 4096, 4097, 100003, 1000003};
        uint64_t seed = 1;
        for (size_t n : sizes) {
            for (int bits : {1, 7, 8, 9, 16, 17, 25, 34, 50, 64}) {
                check_pairs<uint64_t>(ws, n, bits, ++seed, false);
                if (bits <= 32) check_pairs<uint32_t>(ws, n, bits, ++seed, false);
            }
            check_pairs<uint64_t>(ws, n, 64, ++seed, true);
            check_pairs<uint32_t>(ws, n, 32, ++seed, true);
            check_keys_unique(ws, n, 1000, ++seed);
            check_keys_unique(ws, n, 1u << 30, ++seed);
        }
        check_pairs<uint64_t>(ws, 20000076, 34, 99, false);
        check_pairs<uint32_t>(ws, 20000076, 16, 98, false);
        printf("%s (%d failures)\n", failures ? "SORT CHECK FAILED" : "sort check ok", failures);
        time_pairs<uint64_t>(ws, 20000076, 34, "canonical (user, item) order");
        time_pairs<uint32_t>(ws, 20000076, 16, "item-major order");
        time_pairs<uint64_t>(ws, 5000019, 32, "test rows by item");
        time_pairs<uint64_t>(ws, 59047, 64, "popularity order");
    } catch (const Error& e) {
        printf("error: %s\n", e.what());
        return 2;
    }
    return failures ? 1 : 0;
}
