// common.h — shared host/device helpers for libknncf (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <stdexcept>
#include <string>

#include "../../include/knncf.h"

namespace knncf {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& msg) : std::runtime_error(msg), status(st) {}
};

#define KN_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            throw ::knncf::Error(KNNCF_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + \
                                                  " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

#define KN_REQUIRE(cond, st, msg)                         \
    do {                                                  \
        if (!(cond)) throw ::knncf::Error((st), (msg));   \
    } while (0)

// ---- Scala 2.11 immutable HashSet/HashMap iteration order ----------------
// HashSet.improve / HashMap.improve
__host__ __device__ inline uint32_t improve(uint32_t h) {
    h = h + ~(h << 9);
    h ^= h >> 14;
    h += h << 4;
    h ^= h >> 10;
    return h;
}
// key whose ascending unsigned order is the trie iteration order (5-bit digits, root = LSB digit)
__host__ __device__ inline uint32_t trie_key(uint32_t h) {
    return ((h & 31u) << 27) | (((h >> 5) & 31u) << 22) | (((h >> 10) & 31u) << 17) |
           (((h >> 15) & 31u) << 12) | (((h >> 20) & 31u) << 7) | (((h >> 25) & 31u) << 2) |
           ((h >> 30) & 3u);
}
__host__ __device__ inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__host__ __device__ inline uint32_t mm3_mix(uint32_t hash, uint32_t data) {
    uint32_t k = data * 0xcc9e2d51u;
    k = rotl32(k, 15);
    k *= 0x1b873593u;
    uint32_t h = hash ^ k;
    h = rotl32(h, 13);
    return h * 5u + 0xe6546b64u;
}
// Tuple2[Int,Int].hashCode = MurmurHash3.productHash(_, 0xcafebabe)
__host__ __device__ inline uint32_t tuple2_hash(int32_t a, int32_t b) {
    uint32_t h = 0xcafebabeu;
    h = mm3_mix(h, (uint32_t)a);
    h = mm3_mix(h, (uint32_t)b);
    h ^= 2u;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
__host__ __device__ inline uint32_t int_trie_key(int32_t id) { return trie_key(improve((uint32_t)id)); }
__host__ __device__ inline uint32_t tuple_trie_key(int32_t a, int32_t b) {
    return trie_key(improve(tuple2_hash(a, b)));
}

// scale shared/predictions.scala:57-61
__host__ __device__ inline double scale_fn(double x, double y) {
    if (x > y) return 5 - y;
    else if (x < y) return y - 1;
    else return 1;
}

// raw id -> dense index.  keys[] holds the trie keys of the distinct ids in DENSE order:
// ascending (binary search) when count > 4, first-occurrence order (linear scan) otherwise.
__host__ __device__ inline int32_t dense_lookup(const uint32_t* keys, int32_t count, int32_t raw) {
    uint32_t key = int_trie_key(raw);
    if (count <= 4) {
        for (int32_t i = 0; i < count; ++i)
            if (keys[i] == key) return i;
        return -1;
    }
    int32_t lo = 0, hi = count;
    while (lo < hi) {
        int32_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return (lo < count && keys[lo] == key) ? lo : -1;
}

// ---- device array RAII -----------------------------------------------------------------
template <class T>
struct DArr {
    T* p = nullptr;
    size_t n = 0;
    DArr() = default;
    DArr(const DArr&) = delete;
    DArr& operator=(const DArr&) = delete;
    ~DArr() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        if (p && count == n && count > 0) return;  // same size as before (a re-fit of the same shape): keep the buffer —
                                                   // hipFree synchronises the device and hipMalloc costs ~0.1-1 ms
        release();
        n = count;
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            n = 0;
            throw Error(KNNCF_E_NOMEM, std::string("hipMalloc(") + std::to_string(count * sizeof(T)) +
                                           " B): " + hipGetErrorString(e));
        }
    }
    void ensure(size_t count) {
        if (count > n || !p) alloc(count);
    }
    size_t bytes() const { return n * sizeof(T); }
};

#ifdef __HIPCC__
// inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / broadcasts (no LDS traffic)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return x;
}
#endif

// Per-device launch state of a kernel, shared by every handle of the process.  hipFuncSetAttribute(MaxDynamicSharedMemorySize)
// and the occupancy query apply to the device that is current when they run, and handles may live on several devices and be
// driven from several threads (one JVM, several GPUs: INTEGRATION.md) — so the "already done" memo is kept per device ordinal
// under a mutex, never in a bare function-level static.
struct PerDeviceState {
    static constexpr int MAX_DEVICES = 64;
    std::mutex mu;
    size_t value[MAX_DEVICES] = {};
};
// runs setup(current value) -> new value when the current device's value is below `want`; returns the device's value
template <class F>
inline size_t per_device_at_least(PerDeviceState& state, size_t want, F&& setup) {
    int dev = 0;
    KN_HIP(hipGetDevice(&dev));
    KN_REQUIRE(dev >= 0 && dev < PerDeviceState::MAX_DEVICES, KNNCF_E_UNSUPPORTED, "device ordinal beyond the per-device tables");
    std::lock_guard<std::mutex> lock(state.mu);
    if (state.value[dev] < want) state.value[dev] = setup(state.value[dev]);
    return state.value[dev];
}
// enables `bytes` of dynamic LDS for `kernel` on the current device (once per device and size)
inline void ensure_dynamic_lds(PerDeviceState& state, const void* kernel, size_t bytes) {
    per_device_at_least(state, bytes, [&](size_t) {
        KN_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        return bytes;
    });
}

// bits of the largest key value: the fewer significant key bits a radix sort is told, the fewer passes it runs
inline int bits_for(uint64_t max_value) {
    int b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

}  // namespace knncf
