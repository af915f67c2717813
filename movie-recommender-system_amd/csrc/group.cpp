// group.cpp — one process, several GPUs: n shard handles + RCCL collectives behind ONE call per step
// (include/knncf.h: knncf_group_*).  SURVEY 5 "distributed communication backend": ncclCommInitAll, one stream per
// GPU, all inside one process — what a JVM driving eight MI355X binds instead of re-implementing sharded.py.
//
// Built on the public C ABI only (knncf_fit / knncf_shard_view_get / knncf_shard_commit / knncf_mae_device): the group
// adds the transport, nothing else.  RCCL is resolved at run time (dlopen) so that libknncf.so keeps loading on hosts
// and in processes that never form a group; inside a PyTorch process the RCCL that torch already loaded is reused.
//
// Reference analogue: distributed/DistributedBaseline.scala:41-47 (one RDD handed to the executors) and the `sum` /
// `count` / `reduceByKey ... collect` actions of shared/predictions.scala:246-268.
#include <dlfcn.h>
#include <link.h>
#include <math.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "common.h"

using namespace knncf;

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;  // why it could not be loaded
};

// process-wide, loaded once; never unloaded (communicators may outlive any one group)
RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // An RCCL that is already in the process wins — PyTorch bundles one (no soname, built against the HIP runtime it
        // bundles); a second copy beside it would bring a second HIP runtime into the process.  It is found by walking the
        // loaded objects, because it may have been loaded under any name.
        std::string loaded;
        dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* data) -> int {
            const char* path = info->dlpi_name;
            if (!path) return 0;
            const char* base = strrchr(path, '/');
            base = base ? base + 1 : path;
            if (strncmp(base, "librccl.so", 10) == 0) {
                *static_cast<std::string*>(data) = path;
                return 1;
            }
            return 0;
        }, &loaded);
        if (!loaded.empty()) api.lib = dlopen(loaded.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (size_t j = 0; j < sizeof(names) / sizeof(names[0]) && !api.lib; ++j) api.lib = dlopen(names[j], RTLD_NOW | RTLD_LOCAL);
        if (!api.lib) {
            const char* e = dlerror();
            api.why = std::string("librccl.so.1 not found: ") + (e ? e : "?");
            return;
        }
        auto sym = [&](const char* n) -> void* {
            void* p = dlsym(api.lib, n);
            if (!p && api.why.empty()) api.why = std::string("RCCL symbol missing: ") + n;
            return p;
        };
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return api;
}

// reusable barrier of the group's host threads (one per GPU)
struct HostBarrier {
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0;
    uint64_t generation = 0;
    explicit HostBarrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lock(mu);
        const uint64_t gen = generation;
        if (++waiting == n) {
            waiting = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lock, [&] { return generation != gen; });
        }
    }
};

}  // namespace

struct knncf_group {
    int n = 0;
    knncf_config cfg{};
    std::vector<int> devices;
    std::vector<knncf_handle*> h;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;         // the collectives' stream of each GPU
    std::vector<DArr<double>> send, recv;    // padded {mean | norm} segment, all ranks' segments
    std::vector<DArr<double>> red;           // [4]: (sum |err|, rows) in, the same out
    std::vector<DArr<int32_t>> t_users, t_items;
    std::vector<DArr<double>> t_ratings, t_pred;
    std::string err;
};

namespace {

struct RankError {
    int status = KNNCF_OK;
    std::string text;
};

#define KG_RCCL(expr)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess)                                                                                     \
            throw Error(KNNCF_E_RCCL, std::string(#expr) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r_) : "?")); \
    } while (0)

// runs body(rank) on n host threads (rank 0 on the caller's), each bound to its device; a throwing rank records its
// error.  Returns the worst status; `collective_point` = the ranks call sync() before a collective: they all learn whether
// every one of them is still fine and skip the collective together otherwise (no rank is left waiting inside RCCL).
struct Team {
    knncf_group* g;
    HostBarrier bar;
    std::vector<RankError> errs;
    std::mutex mu;
    explicit Team(knncf_group* g_) : g(g_), bar(g_->n), errs((size_t)g_->n) {}
    bool all_ok() {  // collective: every rank must call it at the same point
        bar.wait();
        bool ok = true;
        for (auto& e : errs) ok = ok && e.status == KNNCF_OK;
        bar.wait();  // (nobody changes errs between the two waits' reads)
        return ok;
    }
    int run(const std::function<void(int, Team&)>& body) {
        auto one = [&](int r) {
            try {
                // (a failing hipSetDevice must not keep this rank out of the team's barriers: the body's first handle call
                // fails in its own try block and the rank still reaches all_ok())
                if (hipSetDevice(g->devices[r]) != hipSuccess) (void)hipGetLastError();
                body(r, *this);
            } catch (const Error& e) {
                errs[r].status = e.status;
                errs[r].text = e.what();
            } catch (const std::exception& e) {
                errs[r].status = KNNCF_E_INVALID;
                errs[r].text = e.what();
            }
        };
        std::vector<std::thread> th;
        for (int r = 1; r < g->n; ++r) th.emplace_back(one, r);
        int prev = -1;
        (void)hipGetDevice(&prev);
        one(0);
        for (auto& t : th) t.join();
        if (prev >= 0) (void)hipSetDevice(prev);
        int worst = KNNCF_OK;
        for (int r = 0; r < g->n; ++r)
            if (errs[r].status != KNNCF_OK && (worst == KNNCF_OK || errs[r].status < worst)) {
                worst = errs[r].status;
                g->err = "rank " + std::to_string(r) + " (device " + std::to_string(g->devices[r]) + "): " + errs[r].text;
            }
        return worst;
    }
};

// a handle call on rank r: its status / error text become this rank's Error
void check(knncf_group* g, int r, int st) {
    if (st != KNNCF_OK) throw Error(st, knncf_last_error(g->h[r]));
}

}  // namespace

extern "C" {

int knncf_group_create(const knncf_config* cfg, const int32_t* devices, int32_t n_devices, knncf_group** out) {
    if (!cfg || !devices || !out || n_devices < 1 || n_devices > 64 || cfg->struct_size != sizeof(knncf_config)) return KNNCF_E_INVALID;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return KNNCF_E_INVALID;  // one shard per GPU
    RcclApi& api = rccl();
    if (!api.lib || !api.CommInitAll || !api.CommDestroy || !api.AllGather || !api.AllReduce) return KNNCF_E_UNSUPPORTED;
    knncf_group* g = new (std::nothrow) knncf_group();
    if (!g) return KNNCF_E_NOMEM;
    g->n = n_devices;
    g->cfg = *cfg;
    g->devices.assign(devices, devices + n_devices);
    g->h.assign((size_t)n_devices, nullptr);
    g->comm.assign((size_t)n_devices, nullptr);
    g->stream.assign((size_t)n_devices, nullptr);
    g->send = std::vector<DArr<double>>((size_t)n_devices);
    g->recv = std::vector<DArr<double>>((size_t)n_devices);
    g->red = std::vector<DArr<double>>((size_t)n_devices);
    g->t_users = std::vector<DArr<int32_t>>((size_t)n_devices);
    g->t_items = std::vector<DArr<int32_t>>((size_t)n_devices);
    g->t_ratings = std::vector<DArr<double>>((size_t)n_devices);
    g->t_pred = std::vector<DArr<double>>((size_t)n_devices);
    int prev = -1;
    (void)hipGetDevice(&prev);
    int st = KNNCF_OK;
    for (int r = 0; r < n_devices && st == KNNCF_OK; ++r) {
        knncf_config c = *cfg;
        c.device = devices[r];
        c.shard_rank = r;
        c.shard_count = n_devices;
        st = knncf_create(&c, &g->h[r]);
        if (st == KNNCF_OK && (hipSetDevice(devices[r]) != hipSuccess ||
                               hipStreamCreateWithFlags(&g->stream[r], hipStreamNonBlocking) != hipSuccess))
            st = KNNCF_E_HIP;
    }
    if (st == KNNCF_OK) {
        // one communicator per device, all in this process: RCCL picks the xGMI rings / trees itself
        ncclResult_t r = api.CommInitAll(g->comm.data(), n_devices, g->devices.data());
        if (r != ncclSuccess) {
            for (auto& c : g->comm) c = nullptr;
            st = KNNCF_E_RCCL;
        }
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (st != KNNCF_OK) {
        knncf_group_destroy(g);
        return st;
    }
    *out = g;
    return KNNCF_OK;
}

void knncf_group_destroy(knncf_group* g) {
    if (!g) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (int r = 0; r < g->n; ++r) {
        (void)hipSetDevice(g->devices[r]);
        if (g->stream[r]) (void)hipStreamSynchronize(g->stream[r]);
        if (g->comm[r]) (void)rccl().CommDestroy(g->comm[r]);
        g->send[r].release(); g->recv[r].release(); g->red[r].release();
        g->t_users[r].release(); g->t_items[r].release(); g->t_ratings[r].release(); g->t_pred[r].release();
        if (g->stream[r]) (void)hipStreamDestroy(g->stream[r]);
        if (g->h[r]) knncf_destroy(g->h[r]);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete g;
}

const char* knncf_group_last_error(const knncf_group* g) { return g ? g->err.c_str() : "null group"; }

int knncf_group_size(const knncf_group* g, int32_t* n_devices) {
    if (!g || !n_devices) return KNNCF_E_INVALID;
    *n_devices = g->n;
    return KNNCF_OK;
}

int knncf_group_handle(knncf_group* g, int32_t rank, knncf_handle** out) {
    if (!g || !out || rank < 0 || rank >= g->n) return KNNCF_E_INVALID;
    *out = g->h[rank];
    return KNNCF_OK;
}

int knncf_group_fit(knncf_group* g, const int32_t* users, const int32_t* items, const double* ratings, int64_t n) {
    if (!g) return KNNCF_E_INVALID;
    g->err.clear();
    if (n <= 0 || !users || !items || !ratings) {
        g->err = "fit: null or empty input";
        return KNNCF_E_INVALID;
    }
    std::vector<knncf_shard_view> views((size_t)g->n);
    Team team(g);
    return team.run([&](int r, Team& t) {
        try {
            check(g, r, knncf_fit(g->h[r], users, items, ratings, n));  // (PCIe: every device receives the whole file)
            check(g, r, knncf_shard_view_get(g->h[r], &views[r]));
        } catch (const Error& e) {
            t.errs[r].status = e.status;
            t.errs[r].text = e.what();
        }
        if (!t.all_ok()) return;  // the fit STATUS is collective: nobody enters the all-gather if anybody failed
        const knncf_shard_view& v = views[r];
        const int64_t U = v.num_users;
        const int64_t seg = (U + g->n - 1) / g->n;  // ceil(U / n): the partition of prep.hip; segments differ by <= 1 user
        const int64_t mine = (int64_t)v.user_end - v.user_begin;
        KN_REQUIRE(mine <= seg, KNNCF_E_STATE, "group: a shard owns more users than the partition allows");
        hipStream_t st = g->stream[r];
        g->send[r].ensure((size_t)(2 * seg));
        g->recv[r].ensure((size_t)(2 * seg * g->n));
        KN_HIP(hipMemsetAsync(g->send[r].p, 0, (size_t)(2 * seg) * sizeof(double), st));
        if (mine > 0) {
            KN_HIP(hipMemcpyAsync(g->send[r].p, v.d_user_avg + v.user_begin, (size_t)mine * sizeof(double), hipMemcpyDeviceToDevice, st));
            KN_HIP(hipMemcpyAsync(g->send[r].p + seg, v.d_user_norm + v.user_begin, (size_t)mine * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        // THE exchange of the path: 16 B per user (the per-rating deviations are recomputed by knncf_shard_commit)
        KG_RCCL(rccl().AllGather(g->send[r].p, g->recv[r].p, (size_t)(2 * seg), ncclDouble, g->comm[r], st));
        for (int o = 0; o < g->n; ++o) {
            if (o == r) continue;
            const int64_t lo = views[o].user_begin, len = (int64_t)views[o].user_end - lo;
            if (len <= 0) continue;
            const double* src = g->recv[r].p + (int64_t)o * 2 * seg;
            KN_HIP(hipMemcpyAsync(v.d_user_avg + lo, src, (size_t)len * sizeof(double), hipMemcpyDeviceToDevice, st));
            KN_HIP(hipMemcpyAsync(v.d_user_norm + lo, src + seg, (size_t)len * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        KN_HIP(hipStreamSynchronize(st));  // the handle works on its own streams: the segments must have landed
        check(g, r, knncf_shard_commit(g->h[r]));
    });
}

static int group_predict(knncf_group* g, int predictor, const int32_t* users, const int32_t* items, const double* ratings, int64_t n,
                         double* mae, double* out) {
    if (!g) return KNNCF_E_INVALID;
    g->err.clear();
    if (n < 0 || (n > 0 && (!users || !items)) || (mae && n > 0 && !ratings) || (!mae && n > 0 && !out)) {
        g->err = "bad arguments";
        return KNNCF_E_INVALID;
    }
    if (n == 0) {
        if (mae) *mae = NAN;  // 0.0 / 0 in applyAndMean :85
        return KNNCF_OK;
    }
    std::vector<double> total((size_t)g->n * 2, 0.0);
    std::vector<std::vector<double>> parts(out ? (size_t)g->n : 0);
    Team team(g);
    int status = team.run([&](int r, Team& t) {
        double sums[2] = {0.0, 0.0};
        hipStream_t st = g->stream[r];
        try {
            g->t_users[r].ensure((size_t)n); g->t_items[r].ensure((size_t)n);
            KN_HIP(hipMemcpyAsync(g->t_users[r].p, users, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
            KN_HIP(hipMemcpyAsync(g->t_items[r].p, items, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
            double* d_ratings = nullptr;
            if (ratings) {
                g->t_ratings[r].ensure((size_t)n);
                KN_HIP(hipMemcpyAsync(g->t_ratings[r].p, ratings, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
                d_ratings = g->t_ratings[r].p;
            }
            double* d_pred = nullptr;
            if (out) {  // rows of other shards' users stay NaN (a fitted handle never predicts NaN: scale() == 0 fails the fit)
                g->t_pred[r].ensure((size_t)n);
                KN_HIP(hipMemsetAsync(g->t_pred[r].p, 0xff, (size_t)n * sizeof(double), st));
                d_pred = g->t_pred[r].p;
            }
            KN_HIP(hipStreamSynchronize(st));  // "_device" inputs must be complete at the call (include/knncf.h)
            int64_t cnt = 0;
            if (ratings) {
                check(g, r, knncf_mae_device(g->h[r], predictor, g->t_users[r].p, g->t_items[r].p, d_ratings, n, &sums[0], &cnt, d_pred));
            } else {
                check(g, r, knncf_predict_batch_device(g->h[r], predictor, g->t_users[r].p, g->t_items[r].p, n, d_pred));
            }
            sums[1] = (double)cnt;
            if (out) {
                parts[r].resize((size_t)n);
                KN_HIP(hipMemcpyAsync(parts[r].data(), d_pred, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
                KN_HIP(hipStreamSynchronize(st));
            }
        } catch (const Error& e) {
            t.errs[r].status = e.status;
            t.errs[r].text = e.what();
        }
        if (!t.all_ok()) return;  // collective status: nobody enters the all-reduce if anybody failed
        if (!mae) return;
        // MAE :69-73 over all shards: ncclAllReduce of (sum |r - p|, rows) — 16 bytes over xGMI
        g->red[r].ensure(4);
        KN_HIP(hipMemcpyAsync(g->red[r].p, sums, 2 * sizeof(double), hipMemcpyHostToDevice, st));
        KG_RCCL(rccl().AllReduce(g->red[r].p, g->red[r].p + 2, 2, ncclDouble, ncclSum, g->comm[r], st));
        KN_HIP(hipMemcpyAsync(&total[(size_t)r * 2], g->red[r].p + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
        KN_HIP(hipStreamSynchronize(st));
    });
    if (status != KNNCF_OK) return status;
    if (mae) {
        const double s = total[0], c = total[1];  // every rank holds the same pair
        if ((int64_t)llround(c) != n) {
            g->err = "group: the shards own " + std::to_string((long long)llround(c)) + " of " + std::to_string((long long)n) + " test rows";
            return KNNCF_E_STATE;
        }
        *mae = s / (double)n;
    }
    if (out) {
        for (int64_t t = 0; t < n; ++t) {
            double v = std::numeric_limits<double>::quiet_NaN();
            for (int r = 0; r < g->n; ++r)
                if (!isnan(parts[r][(size_t)t])) { v = parts[r][(size_t)t]; break; }
            out[t] = v;
        }
    }
    return KNNCF_OK;
}

int knncf_group_mae(knncf_group* g, int predictor, const int32_t* users, const int32_t* items, const double* ratings, int64_t n,
                    double* mae) {
    if (!mae) return KNNCF_E_INVALID;
    return group_predict(g, predictor, users, items, ratings, n, mae, nullptr);
}

int knncf_group_predict_batch(knncf_group* g, int predictor, const int32_t* users, const int32_t* items, int64_t n, double* out) {
    return group_predict(g, predictor, users, items, nullptr, n, nullptr, out);
}

}  // extern "C"
