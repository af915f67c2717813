// api.cpp — the C ABI of include/knncf.h: handle, orchestration of K0-K9 on one HIP stream,
// scalar queries.  No CPU arithmetic path exists here: every number an entry point returns was
// produced by the HIP kernels (prep.hip, gemm.hip, select.hip, predict.hip).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "engine.h"

using namespace knncf;

namespace knncf {
// small kernels of the orchestrator (neighbours.hip)
void launch_first_rows(int64_t n, const int32_t* d_du, const int32_t* d_di, int32_t own_lo, int32_t own_hi,
                       uint32_t* d_first, hipStream_t st);
void launch_length_keys(int32_t count, const int32_t* d_list, const int64_t* d_u_ptr, int32_t max_len, uint64_t* d_key, hipStream_t st);
void launch_collect_new(int32_t U, const uint32_t* d_first, int64_t* d_seq, int64_t epoch, int32_t own_lo, int32_t own_hi,
                        int32_t* d_list, int32_t* d_count, hipStream_t st);
void launch_fallback_keys(int32_t U, const double* d_exact, uint64_t* d_keys, uint32_t* d_vals, hipStream_t st);
void launch_fallback_write(int32_t user, int32_t take, int32_t kcap, const uint32_t* d_sorted_vals,
                           const double* d_exact, int32_t* nbr_idx, double* nbr_sim, int32_t* nbr_cnt,
                           hipStream_t st);
void launch_jaccard_pair(const Train& tr, int32_t u, int32_t v, double* d_out, hipStream_t st);
}  // namespace knncf

struct StageTimer {
    hipEvent_t a, b;
    double* acc;
};

struct knncf_handle {
    knncf_config cfg{};
    hipStream_t stream = nullptr;
    std::string err;
    Train tr;
    PrepScratch prep;
    bool fitted = false, committed = false;
    NeighborTable nt;
    int64_t epoch = 1;
    // panels
    int64_t U_pad = 0, K_pad = 0;
    DArr<bf16_t> Bpanel;
    bool b_ready = false;
    int32_t head = 0;  // dense head width of the hybrid similarity
    double tail_pairs_full = 0.0;
    DArr<int32_t> colmap;
    // per-row tail entry lists of the current head (rebuilt with the B panel)
    DArr<int32_t> te_cnt, te_item;
    DArr<float> te_x, row_tail_abs, row_head_sq;
    DArr<float> row_len;  // Jaccard handles: |I(v)| as float for select.hip
    // double-buffered row-block panels: a producer stream (densify, GEMM, tail) runs one block ahead
    // of the consumer stream (select, re-rank)
    hipStream_t stream2 = nullptr;
    DArr<bf16_t> Apanel[2];
    DArr<float> S[2];
    DArr<float> S_full;            // symmetric path: the whole U_pad x U_pad similarity panel (raw storage)
    DArr<uint32_t> sym_tiles;      // its tile order (gemm_sym_tile_list), cached per U_pad
    DArr<int32_t> redo_rows;       // users whose select pass is repeated with the plain thresholds (build_neighbors)
    int32_t sym_tiles_n = 0;       // tiles per side the cached list was built for
    hipEvent_t ev_produced[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr}, ev_ready = nullptr;
    int32_t* pinned_cnt = nullptr;
    size_t pinned_cap = 0;
    SelectScratch sel;
    SliceScratch slices;
    NeighborTable pt;       // Personalized (no k): every non-zero similarity of every user (ids ascending, self included)
    bool pt_ready = false;
    DArr<int32_t> reco_users, reco_items, reco_out_items;
    DArr<double> reco_pred, reco_out_preds;
    DArr<uint8_t> reco_rated;
    DArr<int32_t> build_list, build_count;
    DArr<uint32_t> first_row;
    // test scratch
    DArr<int32_t> t_du, t_di, t_users, t_items;
    DArr<double> t_pred, t_err, t_ratings, t_partial, scalar_out;
    DArr<unsigned long long> scalar_u64;
    DArr<uint8_t> t_owned;
    DArr<int64_t> t_counts;
    // host mirrors for scalar queries
    std::vector<uint32_t> h_ukeys, h_ikeys;
    std::vector<int32_t> h_uid;
    knncf_timings tm{};
    std::vector<StageTimer> pending;
    std::vector<hipEvent_t> event_pool;
};

namespace {

hipEvent_t get_event(knncf_handle* h) {
    if (!h->event_pool.empty()) {
        hipEvent_t e = h->event_pool.back();
        h->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    KN_HIP(hipEventCreate(&e));
    return e;
}

struct Stage {  // RAII: times a stage of device work with HIP events on the stream it is launched on
    knncf_handle* h;
    StageTimer t;
    hipStream_t st;
    Stage(knncf_handle* h_, double* acc, hipStream_t st_ = nullptr) : h(h_), st(st_ ? st_ : h_->stream) {
        t.a = get_event(h);
        t.b = get_event(h);
        t.acc = acc;
        (void)hipEventRecord(t.a, st);
    }
    ~Stage() {
        (void)hipEventRecord(t.b, st);
        h->pending.push_back(t);
    }
};

// end of every entry point: the engine's streams are drained (results in caller-provided device buffers are complete on
// return, include/knncf.h) and the stage timers of the call are read
void resolve_timers(knncf_handle* h) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipStreamSynchronize(h->stream2);
    for (auto& t : h->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) *t.acc += ms;
        h->event_pool.push_back(t.a);
        h->event_pool.push_back(t.b);
    }
    h->pending.clear();
}

template <class F>
int guarded(knncf_handle* h, F&& f) {
    if (!h) return KNNCF_E_INVALID;
    // the calling thread's current device is restored on every path (a JVM thread may drive several handles)
    struct DeviceGuard {
        int prev = -1;
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } guard;
    try {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) {
            // (a thread without a usable HIP context, e.g. after a sticky error: nothing to restore, and the failure — if it
            // persists — becomes this call's status instead of the call running on whatever device is current)
            (void)hipGetLastError();
            KN_HIP(hipSetDevice(h->cfg.device));
        } else if (cur != h->cfg.device) {
            guard.prev = cur;
            KN_HIP(hipSetDevice(h->cfg.device));
        }
        f();
        resolve_timers(h);
        return KNNCF_OK;
    } catch (const Error& e) {
        h->err = e.what();
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamSynchronize(h->stream2);
        (void)hipGetLastError();
        h->pending.clear();
        return e.status;
    } catch (const std::bad_alloc&) {
        h->err = "host allocation failed";
        return KNNCF_E_NOMEM;
    } catch (const std::exception& e) {
        h->err = e.what();
        return KNNCF_E_INVALID;
    }
}

void require_fitted(knncf_handle* h, bool committed = true) {
    KN_REQUIRE(h->fitted, KNNCF_E_STATE, "call knncf_fit first");
    if (committed) KN_REQUIRE(h->committed, KNNCF_E_STATE, "sharded handle: call knncf_shard_commit after the exchange");
}

void load_host_ids(knncf_handle* h) {
    if (!h->h_ukeys.empty() || h->tr.U == 0) return;
    Train& tr = h->tr;
    h->h_ukeys.resize(tr.U);
    h->h_ikeys.resize(tr.I);
    h->h_uid.resize(tr.U);
    KN_HIP(hipMemcpyAsync(h->h_ukeys.data(), tr.ukeys.p, tr.U * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipMemcpyAsync(h->h_ikeys.data(), tr.ikeys.p, tr.I * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipMemcpyAsync(h->h_uid.data(), tr.uid.p, tr.U * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
}

int32_t dense_user(knncf_handle* h, int32_t raw) {
    load_host_ids(h);
    return dense_lookup(h->h_ukeys.data(), h->tr.U, raw);
}
int32_t dense_item(knncf_handle* h, int32_t raw) {
    load_host_ids(h);
    return dense_lookup(h->h_ikeys.data(), h->tr.I, raw);
}

template <class T>
T fetch(knncf_handle* h, const T* d, int64_t idx) {
    T v;
    KN_HIP(hipMemcpyAsync(&v, d + idx, sizeof(T), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
    return v;
}

void reset_neighbors(knncf_handle* h) {
    Train& tr = h->tr;
    NeighborTable& nt = h->nt;
    nt.k = h->cfg.k;
    nt.kcap = std::max(0, std::min(nt.k, tr.U - 1));
    size_t cells = (size_t)tr.U * (size_t)std::max(nt.kcap, 1);
    nt.idx.ensure(cells);
    nt.sim.ensure(cells);
    nt.by_id_valid = false;  // (uidx / usim: made by launch_predict when a kernel wants them)
    nt.cnt.ensure(tr.U);
    nt.seq.ensure(tr.U);
    KN_HIP(hipMemsetAsync(nt.cnt.p, 0, tr.U * sizeof(int32_t), h->stream));
    KN_HIP(hipMemsetAsync(nt.seq.p, 0xff, tr.U * sizeof(int64_t), h->stream));  // -1
    h->epoch = 1;
}

// per-pair error bound of the bf16 GEMM entry, excluding the per-row accumulation term that
// select.hip adds from the row length: both operands rounded to bf16 (u = 2^-8, via fp32),
// products exact in fp32, sum |x y| <= ||x|| ||y|| <= 1
// split in two: the relative operand-rounding part (2u + u^2), which select.hip scales by the norm of the row's
// head part (sum_head |x y| <= ||x_head|| ||y_head|| <= ||x_head||), and the absolute rest
float gemm_eps_operand(bool fp16) {
    const double u = ldexp(1.0, fp16 ? -11 : -8) * 1.001;  // bf16: 8 significant bits; fp16: 11
    return (float)(2 * u + u * u);
}
float gemm_eps_rest(bool fp16) {
    // the tail's fp32 operand/product roundings (3 * 2^-24 of sum |x y| <= 1); fp16: |pre| <= 1, below 2^-14 the
    // grid is absolute, 2^-25 per operand: sum (|x| + |y|) 2^-25 <= 2 sqrt(n) 2^-25 < 6e-6 for n < 10^4
    // + the tail's row-side factor travels with its 6 low mantissa bits replaced (select.hip: piece descriptors):
    //   relative 2^-17 of sum |x y| <= 1
    return (float)(4e-7 + (fp16 ? 6e-6 : 0.0) + 8e-6);
}

// per-row shortlist storage: rows whose error band holds more candidates than this take the exact
// fallback.  Heavy raters have compressed similarity distributions (many candidates inside the
// bf16 band), so the store is generous; the re-rank consumes it in LDS-sized chunks.
int32_t shortlist_cap(int32_t k, int32_t U) {
    int64_t want = std::max<int64_t>(16384, 4 * (int64_t)k);
    int64_t cap = 64;
    while (cap < want) cap <<= 1;
    int64_t upper = 64;
    while (upper < U) upper <<= 1;
    return (int32_t)std::min(cap, upper);
}

// Width H of the dense head of the hybrid similarity.  Cost model (measured rates, MI355X): a dense
// column costs 2 * rows * U flops on the MFMA GEMM; a tail item with c raters costs c^2 * rows / U
// LDS accumulator updates in k_tail_select.  Items are in descending popularity, so the optimum is a prefix.
int32_t choose_head(knncf_handle* h, int32_t rows_total, bool symmetric) {
    Train& tr = h->tr;
    const int32_t I = tr.I;
    const std::vector<int64_t>& c = tr.pop_count;
    std::vector<double> tail_sq((size_t)I + 1, 0.0);
    for (int32_t j = I - 1; j >= 0; --j) tail_sq[j] = tail_sq[j + 1] + (double)c[j] * (double)c[j];
    int32_t H;
    if (h->cfg.head_items == KNNCF_HEAD_ALL) {
        H = I;
    } else if (h->cfg.head_items > 0) {
        H = (int32_t)std::min<int64_t>(h->cfg.head_items, I);
    } else {
        // marginal rates measured on MI355X at the ml-25m shape (head sweeps 192 .. 1024, profiles/README.md; re-measured in
        // round 3 with the overlapped GEMM and the 6-VALU drain: 0.0195 ms per dense column on the symmetric path = 2.7e15
        // full-square flops per second, 4.2e-13 s per tail pair product — the optimum stays at 384 / 256): full-square
        // flops per second bought by one more dense column — the symmetric launch computes half of them — and tail pair
        // products per second through k_tail_select
        const double RATE_DENSE = symmetric ? 2.8e15 : 1.2e15;
        const double RATE_SPARSE = 2.3e12;
        const double frac = (double)rows_total / (double)tr.U;
        const double U_pad = (double)round_up(tr.U, 256);
        double best = 1e300;
        H = I;
        for (int64_t cand = 64;; cand += 64) {
            int32_t hc = (int32_t)std::min<int64_t>(cand, I);
            double cost = 2.0 * rows_total * U_pad * (double)round_up(hc, 64) / RATE_DENSE + tail_sq[hc] * frac / RATE_SPARSE;
            if (cost < best) { best = cost; H = hc; }
            if (hc == I) break;
        }
        // Jaccard handles count common items in the panel: fp16 holds the counts exactly up to 2048 (build_neighbors refuses a
        // wider head unless KNNCF_FLAG_F32_PANEL lifts the limit) — the cost model must not pick what the build then refuses;
        // the tail takes the remaining items and the counts stay exact
        if (tr.jaccard && (h->cfg.flags & KNNCF_FLAG_F32_PANEL) == 0) H = std::min(H, 2048);
    }
    h->tail_pairs_full = tail_sq[H];
    return H;
}

// build the neighbourhoods of the users in h->build_list[0 .. count)
void build_neighbors(knncf_handle* h, int32_t count) {
    Train& tr = h->tr;
    NeighborTable& nt = h->nt;
    if (count <= 0 || nt.kcap <= 0) return;
    KN_REQUIRE(h->cfg.similarity != KNNCF_SIM_ONE, KNNCF_E_UNSUPPORTED,
               "kNN neighbourhoods with similarityOne: every similarity is 1.0, the neighbourhood is the first k users in Set order — not built");
    hipStream_t st = h->stream;
    if (count > 256) {
        // longest rows first (LPT): one workgroup per row in select and re-rank, and the row lengths are heavy-tailed
        // (ml-25m shape: mean 123 ratings, maximum 7485) — a long row dispatched last holds the launch open alone.
        // The order of the list carries no meaning (the users' build sequence numbers are already assigned).
        PrepScratch& sc = h->prep;
        sc.k64_a.ensure(count); sc.k64_b.ensure(count); sc.v32_b.ensure(count);
        launch_length_keys(count, h->build_list.p, tr.u_ptr.p, tr.I, sc.k64_a.p, st);  // (a row holds every item at most once)
        sort_pairs_u64_u32(sc.sort, sc.k64_a.p, sc.k64_b.p, reinterpret_cast<const uint32_t*>(h->build_list.p), sc.v32_b.p, count, bits_for((uint64_t)tr.I), st);
        KN_HIP(hipMemcpyAsync(h->build_list.p, sc.v32_b.p, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    }
    const bool fp16 = (h->cfg.flags & KNNCF_FLAG_BF16_FILTER) == 0;
    const bool s_fp16 = (h->cfg.flags & KNNCF_FLAG_F32_PANEL) == 0;  // similarity panel stored as fp16 (half the HBM traffic)
    h->U_pad = round_up(tr.U, 256);
    const int64_t U_pad = h->U_pad;
    size_t free_b = 0, total_b = 0;
    KN_HIP(hipMemGetInfo(&free_b, &total_b));
    const int64_t s_elem = s_fp16 ? 2 : 4;
    // SYMMETRIC PATH (whole-matrix builds on one GPU): S = B B^T is symmetric, so when (nearly) every user's row is wanted
    // the whole U_pad x U_pad panel is produced by ONE launch that computes the tiles on and above the diagonal and stores
    // each of them twice (gemm.hip: SYM) — half the MFMA work of the row-block launches, the same stored values bit for
    // bit — and the row blocks below only run select + re-rank, reading their rows out of it by dense user index.
    // 288 GB of HBM hold it easily at the ml-25m shape (53 GB as fp16); shapes whose square does not fit (syn-1M: 2 TB)
    // and partial / sharded builds take the row-block path.
    const size_t sym_bytes = (size_t)U_pad * (size_t)U_pad * (size_t)s_elem;
    // (what this handle already holds counts as free: the decision must not flip between two builds of the same shape because
    // the first one allocated — measured: a handle built beside another one's 135 GB sat exactly on the edge, released the
    // 53 GB panel in its second build and spent 1.5 s re-allocating)
    size_t own = h->S_full.bytes() + h->sel.cand_idx.bytes() + h->sel.cand_approx.bytes() + h->sel.grp_v0.bytes() + h->sel.grp_x.bytes();
    for (int s = 0; s < 2; ++s) own += h->S[s].bytes() + h->Apanel[s].bytes();
    bool use_sym = h->cfg.shard_count == 1 && (int64_t)count * 2 >= tr.U && U_pad / 256 < 65536 &&
                   sym_bytes <= (free_b + own) / 3 && !getenv("KNNCF_DEBUG_NO_SYMMETRIC_GEMM");
    if (use_sym) {
        try {
            h->S_full.ensure((sym_bytes + 3) / 4);
        } catch (const Error& e) {  // (fragmented / shared device: the row-block path needs far less in one piece)
            if (e.status != KNNCF_E_NOMEM) throw;
            (void)hipGetLastError();
            use_sym = false;
        }
    }
    if (!h->b_ready) {
        // hybrid similarity: the H most-rated items are dense MFMA columns, the rest a sparse tail
        h->head = choose_head(h, count, use_sym);
        h->K_pad = round_up(h->head, 64);
        size_t need = (size_t)U_pad * h->K_pad * sizeof(bf16_t);
        KN_REQUIRE(need < free_b + h->Bpanel.bytes(), KNNCF_E_UNSUPPORTED,
                   "dense bf16 user panel does not fit in HBM; lower head_items");
        Stage s(h, &h->tm.densify_ms);
        h->colmap.ensure(tr.I);
        launch_colmap(tr, h->head, h->colmap.p, st);
        h->Bpanel.ensure((size_t)U_pad * h->K_pad);
        launch_densify(tr, nullptr, 0, tr.U, h->colmap.p, h->Bpanel.p, h->K_pad, U_pad, fp16, st);
        if (tr.jaccard) {
            // the counting GEMM stores exact integers: fp16 holds them up to 2048 (KNNCF_FLAG_F32_PANEL lifts the limit)
            KN_REQUIRE(!s_fp16 || h->head <= 2048, KNNCF_E_UNSUPPORTED, "Jaccard: more than 2048 dense head items need KNNCF_FLAG_F32_PANEL");
            h->row_len.ensure((size_t)row_len_size(tr.U));
            launch_row_len(tr, h->row_len.p, st);
        }
        if (h->head < tr.I) {
            h->te_cnt.ensure(tr.U); h->te_item.ensure(tr.n); h->te_x.ensure(tr.n);
            h->row_tail_abs.ensure(tr.U); h->row_head_sq.ensure(tr.U);
            launch_tail_entries(tr, h->colmap.p, h->te_cnt.p, h->te_item.p, h->te_x.p, h->row_tail_abs.p, h->row_head_sq.p, st);
        }
        h->b_ready = true;
        KN_HIP(hipMemGetInfo(&free_b, &total_b));
    }
    const int64_t K_pad = h->K_pad;
    const int32_t head = h->head;
    h->tm.head_items = head;
    if (use_sym) {
        static const bool tile128 = getenv("KNNCF_GEMM_SYM_TILE128") != nullptr;  // A/B switch for measurements
        const int sym_tile = (tile128 && fp16 && s_fp16) ? 128 : 256;
        const int32_t n_tiles = (int32_t)(U_pad / sym_tile);
        if (h->sym_tiles_n != n_tiles) {
            std::vector<uint32_t> list;
            int group = sym_tile == 128 ? 16 : 8;
            if (const char* g = getenv("KNNCF_GEMM_SYM_GROUP")) group = std::max(1, atoi(g));  // A/B switch for measurements
            gemm_sym_tile_list(n_tiles, list, group);
            h->sym_tiles.ensure(list.size());
            KN_HIP(hipMemcpyAsync(h->sym_tiles.p, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            KN_HIP(hipStreamSynchronize(st));
            h->sym_tiles_n = n_tiles;
        }
        const int64_t n_listed = (int64_t)n_tiles * (n_tiles + 1) / 2;
        Stage s(h, &h->tm.gemm_ms);
        launch_gemm_sym(h->Bpanel.p, h->S_full.p, s_fp16, U_pad, h->K_pad, h->K_pad, U_pad, fp16, !tr.jaccard, h->sym_tiles.p, n_listed, st, sym_tile);
        h->tm.gemm_launches += 1;
        h->tm.gemm_flops_executed += 2.0 * (double)sym_tile * (double)sym_tile * (double)n_listed * (double)h->K_pad;
        // ordered pairs (row, other user) of the rows actually wanted x the dense columns (bench.py halves it)
        h->tm.gemm_flops_algorithmic += 2.0 * (double)count * (double)(tr.U - 1) * (double)h->head;
        KN_HIP(hipMemGetInfo(&free_b, &total_b));
    } else {
        h->S_full.release();
    }
    // rows per block from the similarity-panel budget (two slots: the producer stream runs ahead)
    size_t held = 0;
    for (int s = 0; s < 2; ++s) held += h->S[s].bytes() + h->Apanel[s].bytes();
    held += h->sel.cand_idx.bytes() + h->sel.cand_approx.bytes() + h->sel.grp_v0.bytes() + h->sel.grp_x.bytes();
    int64_t budget = h->cfg.workspace_bytes > 0 ? h->cfg.workspace_bytes / 2
                                                : (int64_t)std::min<size_t>((size_t)48 << 30, (free_b + held) / 4);
    // per panel row: the similarity row, the operand row, the shortlist store and the provisional group store
    int64_t per_row = (use_sym ? 0 : U_pad * s_elem + K_pad * 2) + (int64_t)shortlist_cap(nt.k, tr.U) * 8 + (int64_t)select_gcap(nt.k) * 36;
    // When ALL rows fit a third of what is free (at most 96 GB) they are ONE block: at the ml-25m shape 69 GB of shortlist and
    // group stores beside the 53 GB panel, select and re-rank then run as one launch each and a launch's tail of unfinished
    // rows is paid once (0.5 ms per step against two blocks).  Shapes that need several blocks anyway keep the 48 GB
    // budget: allocating more only costs (syn-1M, one shot: 1.7 s of extra hipMalloc time for 96 GB blocks).
    if (h->cfg.workspace_bytes <= 0) {
        const int64_t whole = round_up(count, 256) * per_row;
        if (whole <= (int64_t)std::min<size_t>((size_t)96 << 30, (free_b + held) / 3)) budget = std::max(budget, whole);
    }
    int64_t R = std::max<int64_t>(256, (budget / per_row) / 256 * 256);
    R = std::min<int64_t>(R, round_up(count, 256));
    const int64_t n_blocks = ceil_div(count, R);
    R = round_up(ceil_div(count, n_blocks), 256);  // equal blocks: no short straggler at the end
    // KNNCF_FLAG_OVERLAP: run the producer one block ahead.  Measured on MI355X (ml-25m shape): -5 % step
    // time, but GEMM and re-rank then contend for LDS/CUs (GEMM 890 -> 506 TFLOP/s), so it is opt-in.
    const bool overlap = (h->cfg.flags & KNNCF_FLAG_OVERLAP) != 0;
    const int slots = (overlap && n_blocks > 1 && !use_sym) ? 2 : 1;
    for (int s = 0; s < slots && !use_sym; ++s) {
        h->S[s].ensure((size_t)(R * U_pad * s_elem + 3) / 4);  // DArr<float> used as raw storage
        h->Apanel[s].ensure((size_t)R * K_pad);
    }
    const int32_t cap = shortlist_cap(nt.k, tr.U);
    const bool verify = (h->cfg.flags & KNNCF_FLAG_VERIFY_BOUND) != 0;
    h->sel.cand_idx.ensure((size_t)R * cap);
    h->sel.cand_approx.ensure((size_t)R * cap);
    h->sel.cand_cnt.ensure(R);
    h->sel.cand_eps.ensure(R);
    h->sel.row_entries.ensure(R);
    h->sel.grp_v0.ensure((size_t)R * select_gcap(nt.k));
    h->sel.grp_x.ensure((size_t)R * select_gcap(nt.k) * 8);
    h->sel.stats.ensure(4);
    KN_HIP(hipMemsetAsync(h->sel.stats.p, 0, 4 * sizeof(double), st));  // [0] bound check, [1] candidate row entries
    if (h->pinned_cap < (size_t)count) {
        if (h->pinned_cnt) KN_HIP(hipHostFree(h->pinned_cnt));
        h->pinned_cnt = nullptr;
        h->pinned_cap = 0;
        KN_HIP(hipHostMalloc((void**)&h->pinned_cnt, (size_t)count * sizeof(int32_t), hipHostMallocDefault));
        h->pinned_cap = (size_t)count;
    }
    // fp16 panel storage rounds the dense head once more: the GEMM clamps it to [-1, 1] first (the exact head sum lies
    // there, so clamping only moves towards it), where half an fp16 ulp is at most 2^-12
    const float eps_opnd = gemm_eps_operand(fp16);
    const float eps_rest = gemm_eps_rest(fp16) + (s_fp16 ? 2.45e-4f : 0.f);
    // producer: densify, GEMM.  Without the overlap it is the consumer's stream itself: an event wait across two
    // hardware queues costs ~0.1 ms each time (three blocks per step at ml-25m shape)
    hipStream_t sp = slots > 1 ? h->stream2 : h->stream;
    hipStream_t sc = h->stream;   // consumer: select, exact re-rank
    KN_HIP(hipEventRecord(h->ev_ready, sc));  // everything queued so far (fit, B panel) precedes the producer
    KN_HIP(hipStreamWaitEvent(sp, h->ev_ready, 0));
    // host copy of the build list (dense users in build order), fetched on first need
    std::vector<int32_t> h_rows;
    auto need_h_rows = [&] {
        if (!h_rows.empty()) return;
        h_rows.resize(count);
        KN_HIP(hipMemcpyAsync(h_rows.data(), h->build_list.p, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        KN_HIP(hipStreamSynchronize(st));
    };
    // rows (positions in the build list) whose anticipated thresholds overshot: once more through select + re-rank with the
    // plain thresholds, out of the panel Sp (whole matrix: indexed by user; a row block based at position rb: by block row)
    auto redo_marked = [&](const std::vector<int32_t>& marked, const void* Sp, bool by_user, int64_t rb) {
        need_h_rows();
        std::vector<int32_t> users(marked.size()), srow(marked.size());
        for (size_t j = 0; j < marked.size(); ++j) {
            users[j] = h_rows[marked[j]];
            srow[j] = (int32_t)(marked[j] - rb);
        }
        TailEntries te{h->te_cnt.p, h->te_item.p, h->te_x.p, h->row_tail_abs.p, h->row_head_sq.p, h->row_len.p};
        const int64_t chunk = std::min<int64_t>((int64_t)marked.size(), R);
        h->redo_rows.ensure(2 * (size_t)chunk);
        for (size_t j0 = 0; j0 < marked.size(); j0 += (size_t)R) {  // (the shortlist / group stores hold R rows)
            const int32_t m = (int32_t)std::min<size_t>((size_t)R, marked.size() - j0);
            KN_HIP(hipMemcpyAsync(h->redo_rows.p, users.data() + j0, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, st));
            KN_HIP(hipMemcpyAsync(h->redo_rows.p + chunk, srow.data() + j0, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, st));
            {
                Stage s(h, &h->tm.select_ms);
                launch_tail_select(tr, h->colmap.p, te, head < tr.I, Sp, by_user, s_fp16, U_pad, m, h->redo_rows.p, nt.k, eps_opnd, eps_rest, cap,
                                   h->sel.cand_idx.p, h->sel.cand_approx.p, h->sel.cand_cnt.p, h->sel.cand_eps.p, h->sel.grp_v0.p, h->sel.grp_x.p,
                                   select_gcap(nt.k), st, /*anticipate=*/false, by_user ? nullptr : h->redo_rows.p + chunk);
                h->tm.select_launches += 1;
            }
            {
                Stage s(h, &h->tm.rerank_ms);
                launch_rerank(tr, nt, m, h->redo_rows.p, cap, h->sel.cand_idx.p, h->sel.cand_approx.p, h->sel.cand_cnt.p, h->sel.cand_eps.p,
                              h->sel.stats.p, h->sel.row_entries.p, verify, st);
            }
            std::vector<int32_t> again(m);
            KN_HIP(hipMemcpyAsync(again.data(), h->sel.cand_cnt.p, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            KN_HIP(hipStreamSynchronize(st));
            for (int32_t j = 0; j < m; ++j) h->pinned_cnt[marked[j0 + j]] = again[j];
        }
    };
    const bool per_block_redo = !use_sym && n_blocks > 1;
    for (int64_t b = 0; b < n_blocks; ++b) {
        const int64_t rb = b * R;
        const int32_t rows = (int32_t)std::min<int64_t>(R, count - rb);
        const int64_t M = round_up(rows, 256);
        const int32_t* d_rows = h->build_list.p + rb;
        const int slot = (int)(b % slots);
        if (b >= slots) KN_HIP(hipStreamWaitEvent(sp, h->ev_consumed[slot], 0));  // S[slot] has been read
        if (!use_sym) {
            Stage s(h, &h->tm.densify_ms, sp);
            launch_densify(tr, d_rows, 0, rows, h->colmap.p, h->Apanel[slot].p, K_pad, M, fp16, sp);
        }
        if (!use_sym) {
            Stage s(h, &h->tm.gemm_ms, sp);
            launch_gemm_nt(h->Apanel[slot].p, h->Bpanel.p, h->S[slot].p, s_fp16, M, U_pad, K_pad, K_pad, K_pad, U_pad, fp16, !tr.jaccard, sp);
            h->tm.gemm_launches += 1;
            h->tm.gemm_flops_executed += 2.0 * (double)M * (double)U_pad * (double)K_pad;
            // SURVEY 8(d) per-unit figure x the units this launch processes: ordered pairs (row, other user)
            // x the dense columns it contracts
            h->tm.gemm_flops_algorithmic += 2.0 * (double)rows * (double)(tr.U - 1) * (double)head;
        }
        KN_HIP(hipEventRecord(h->ev_produced[slot], sp));
        KN_HIP(hipStreamWaitEvent(sc, h->ev_produced[slot], 0));
        TailEntries te{h->te_cnt.p, h->te_item.p, h->te_x.p, h->row_tail_abs.p, h->row_head_sq.p, h->row_len.p};
        h->prep.join_commit(sc);  // the item-major rater lists and the tile table (second part of prep_commit)
        const void* Sblk = use_sym ? (const void*)h->S_full.p : (const void*)h->S[slot].p;
        const int32_t gcap = select_gcap(nt.k);
        // rows [r0, r0 + nr) of this block through select: every per-row store is indexed by the launch row, so a part of the
        // block is the same launch on offset pointers
        auto select_rows = [&](int32_t r0, int32_t nr) {
            // sparse tail (LDS atomics per row tile) + histogram select, fused: one pass over S
            Stage s(h, &h->tm.select_ms, sc);
            const void* Sp = use_sym ? Sblk : (const void*)(static_cast<const char*>(Sblk) + (size_t)r0 * (size_t)U_pad * (size_t)s_elem);
            launch_tail_select(tr, h->colmap.p, te, head < tr.I, Sp, use_sym, s_fp16, U_pad, nr, d_rows + r0, nt.k, eps_opnd, eps_rest, cap,
                               h->sel.cand_idx.p + (size_t)r0 * cap, h->sel.cand_approx.p + (size_t)r0 * cap, h->sel.cand_cnt.p + r0, h->sel.cand_eps.p + r0,
                               h->sel.grp_v0.p + (size_t)r0 * gcap, h->sel.grp_x.p + (size_t)r0 * gcap * 8, gcap, sc);
            h->tm.select_launches += 1;
            h->tm.tail_pair_updates += h->tail_pairs_full * ((double)nr / (double)tr.U);
            h->tm.select_row_bytes += (double)s_elem * (double)nr * (double)tr.U;
        };
        // HEAVY ROWS AS SLICES.  One workgroup per row, rows longest first — but the re-rank of the few heaviest rows (their
        // candidates are heavy raters too: up to 33 x the median row's work at the ml-25m shape, 12 x at the 99.9th percentile,
        // scripts/analysis/row_work_profile.py) outlasts a sharded handle's whole launch: 1.6 - 3.2 ms per shard of config 4 where
        // the work is 1.4.  The top 0.2 % of such a block's rows are therefore re-ranked as P slices of their shortlists + a merge
        // (rerank.hip), P workgroups per row in the same launch: 1.68 ms on every shard (scripts/slice_sweep.sh; 160 rows the
        // same, 640 rows 1.83).  A whole-matrix launch hides that tail by itself, and there the slices' repeated row set-up
        // only costs (+0.2 ms at 256 rows, +0.85 at 2048), so blocks beyond 65 536 rows go unsliced.
        const int32_t slices = std::min<int32_t>(8, 8192 / std::max<int32_t>(nt.kcap, 1));
        // (KNNCF_DEBUG_SLICE_ROWS = n: the first n rows of every block instead — 0 turns slicing off; the small-shape parity tests
        // force it on with this)
        const char* force_slices = getenv("KNNCF_DEBUG_SLICE_ROWS");
        const int32_t n_heavy = slices < 2                         ? 0
                                : force_slices                     ? std::max<int32_t>(0, std::min<int32_t>(rows, atoi(force_slices)))
                                : (rows >= 4096 && rows <= 65536)  ? std::max<int32_t>(16, rows / 512)
                                                                   : 0;
        select_rows(0, rows);
        if (overlap && !per_block_redo) KN_HIP(hipEventRecord(h->ev_consumed[slot], sc));
        {
            Stage s(h, &h->tm.rerank_ms, sc);
            launch_rerank(tr, nt, rows, d_rows, cap, h->sel.cand_idx.p, h->sel.cand_approx.p, h->sel.cand_cnt.p, h->sel.cand_eps.p, h->sel.stats.p,
                          h->sel.row_entries.p, verify, sc, n_heavy, slices, &h->slices);
        }
        if (!overlap) KN_HIP(hipEventRecord(h->ev_consumed[slot], sc));
        if (per_block_redo) KN_HIP(hipMemcpyAsync(h->pinned_cnt + rb, h->sel.cand_cnt.p, rows * sizeof(int32_t), hipMemcpyDeviceToHost, sc));
        if (per_block_redo) {
            // several row blocks (syn-1M, capped workspaces): the block's panel slot is about to be recycled, so rows whose
            // anticipated thresholds overshot are re-selected NOW if they are many (one host round trip per block: the
            // blocks of such builds take tens of milliseconds each)
            KN_HIP(hipStreamSynchronize(sc));
            std::vector<int32_t> marked;
            for (int64_t r = rb; r < rb + rows; ++r)
                if (h->pinned_cnt[r] > cap) marked.push_back((int32_t)r);
            if ((int64_t)marked.size() > std::max<int64_t>(64, rows / 200)) {
                redo_marked(marked, h->S[slot].p, false, rb);
            }
            if (overlap) KN_HIP(hipEventRecord(h->ev_consumed[slot], sc));  // (only now may the producer recycle S[slot])
        }
    }
    // One-launch builds (whole-matrix, or one row block): the device has summed what the host needs to know about the rows
    // (stats[2] shortlist lengths, stats[3] rows to rebuild: k_sum_row_entries), so ONE four-word read-back ends the build; the per-row counts are fetched only if a row has to be rebuilt.
    // (Walking 162 541 counts on the host between the re-rank and the prediction left the GPU idle for 0.25 ms per step.)
    const bool summary = !per_block_redo;
    unsigned long long four[4] = {0, 0, 0, 0};
    nt.by_id_valid = false;  // (the id-sorted copies are made by launch_predict when a kernel wants them)
    if (summary) KN_HIP(hipMemcpyAsync(four, h->sel.stats.p, sizeof(four), hipMemcpyDeviceToHost, sc));
    KN_HIP(hipStreamSynchronize(sc));
    KN_HIP(hipStreamSynchronize(sp));
    auto take_stats = [&](const unsigned long long* w) {
        h->tm.rerank_row_bytes += 12.0 * (double)w[1];
        if (verify && w[0] != 0) {
            double shifted;
            memcpy(&shifted, &w[0], sizeof(double));
            h->tm.max_bound_violation = std::max(h->tm.max_bound_violation, shifted - 4.0);
        }
    };
    if (summary && four[3] == 0) {  // every list is final
        h->tm.shortlist_total += (double)four[2];
        take_stats(four);
        return;
    }
    if (summary) {
        KN_HIP(hipMemcpyAsync(h->pinned_cnt, h->sel.cand_cnt.p, (size_t)count * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        KN_HIP(hipStreamSynchronize(st));
    }
    // select.hip anticipates its emission thresholds and marks a row whose guess overshot (like one whose stores overflowed)
    // for the exact fallback below — a 7-sigma event per row when the dense user order is a pseudo-random column sample,
    // which HashSet ranks of the raw ids are.  Should a data set defeat that (many rows marked), the marked rows are not sent
    // through the per-row exact path (milliseconds each, a U-sized sort each) but once more through select + re-rank with the
    // plain thresholds (redo_marked): the anticipation can then cost at most one extra pass.  Whole-matrix builds and
    // one-block row-block builds do it here (the similarity panel is still there); builds of several row blocks did it per
    // block, before the block's panel slot was recycled (above).
    if (summary) {
        std::vector<int32_t> marked;
        for (int64_t r = 0; r < count; ++r)
            if (h->pinned_cnt[r] > cap) marked.push_back((int32_t)r);
        if ((int64_t)marked.size() > std::max<int64_t>(64, count / 200)) redo_marked(marked, use_sym ? (const void*)h->S_full.p : (const void*)h->S[0].p, use_sym, 0);
    }
    // rows whose shortlist overflowed: exact row + stable descending sort (rare)
    for (int64_t r = 0; r < count; ++r) {
        h->tm.shortlist_total += std::min(h->pinned_cnt[r], cap);
        if (h->pinned_cnt[r] > cap) {
            need_h_rows();
            Stage s(h, &h->tm.rerank_ms);
            int32_t u = h_rows[r];
            h->sel.row_exact.ensure(tr.U);
            h->sel.fb_keys_a.ensure(tr.U); h->sel.fb_keys_b.ensure(tr.U);
            h->sel.fb_vals_a.ensure(tr.U); h->sel.fb_vals_b.ensure(tr.U);
            int64_t seq_u = fetch(h, nt.seq.p, u);
            launch_exact_row(tr, nt, u, seq_u, h->sel.row_exact.p, st);
            launch_fallback_keys(tr.U, h->sel.row_exact.p, h->sel.fb_keys_a.p, h->sel.fb_vals_a.p, st);
            sort_pairs_u64_u32(h->prep.sort, h->sel.fb_keys_a.p, h->sel.fb_keys_b.p, h->sel.fb_vals_a.p,
                               h->sel.fb_vals_b.p, tr.U, 64, st);
            launch_fallback_write(u, nt.kcap, nt.kcap, h->sel.fb_vals_b.p, h->sel.row_exact.p, nt.idx.p, nt.sim.p,
                                  nt.cnt.p, st);
            h->tm.fallback_rows += 1;
        }
    }
    {
        unsigned long long two[2] = {0, 0};
        KN_HIP(hipMemcpyAsync(two, h->sel.stats.p, sizeof(two), hipMemcpyDeviceToHost, st));
        KN_HIP(hipStreamSynchronize(st));
        take_stats(two);
    }
}

void ensure_test_scratch(knncf_handle* h, int64_t n) {
    h->t_du.ensure(n); h->t_di.ensure(n); h->t_pred.ensure(n); h->t_err.ensure(n); h->t_owned.ensure(n);
    h->t_partial.ensure(1024); h->t_counts.ensure(1024);
}

// neighbourhoods needed by the test rows, in the order the reference's lazy closures would build
// them: a user's neighbourhood is built at its first test row whose item has raters
void ensure_neighbors_for_rows(knncf_handle* h, int64_t n) {
    Train& tr = h->tr;
    if (tr.U < 2 || h->nt.kcap <= 0) return;
    KN_REQUIRE(n < (int64_t)0xffffffffll, KNNCF_E_UNSUPPORTED, "more than 2^32-1 test rows");
    hipStream_t st = h->stream;
    h->first_row.ensure(tr.U);
    h->build_list.ensure(tr.U);
    h->build_count.ensure(1);
    KN_HIP(hipMemsetAsync(h->first_row.p, 0xff, tr.U * sizeof(uint32_t), st));
    KN_HIP(hipMemsetAsync(h->build_count.p, 0, sizeof(int32_t), st));
    // (every user's first row, whoever owns it: a shard needs the build sequence numbers of the other shards' users too)
    launch_first_rows(n, h->t_du.p, h->t_di.p, 0, tr.U, h->first_row.p, st);
    launch_collect_new(tr.U, h->first_row.p, h->nt.seq.p, h->epoch, tr.own_lo, tr.own_hi, h->build_list.p, h->build_count.p, st);
    h->epoch += 1;
    int32_t count = fetch(h, h->build_count.p, 0);
    build_neighbors(h, count);
}

// predictor(train, weightedSumDeviation(train, sim)) with sim = adjustedCosineSimilarityFunction(train) or
// jaccardCoefficient(train) (predict/Personalized.scala:61-72): the table of every non-zero similarity, built once per fit
void ensure_personalized_table(knncf_handle* h) {
    if (h->pt_ready) return;
    Train& tr = h->tr;
    KN_REQUIRE(h->cfg.shard_count == 1, KNNCF_E_UNSUPPORTED, "PERSONALIZED is not sharded");
    KN_REQUIRE(tr.U <= 2048, KNNCF_E_UNSUPPORTED,
               "PERSONALIZED with the adjusted cosine / Jaccard similarity keeps U x U similarities: built for U <= 2048 (the reference runs it at ml-100k scale)");
    if (h->cfg.similarity == KNNCF_SIM_COSINE) {
        std::vector<int64_t> ptr((size_t)tr.U + 1);
        KN_HIP(hipMemcpyAsync(ptr.data(), tr.u_ptr.p, ptr.size() * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        KN_HIP(hipStreamSynchronize(h->stream));
        for (int32_t u = 0; u < tr.U; ++u)
            KN_REQUIRE(ptr[u + 1] - ptr[u] > 4, KNNCF_E_UNSUPPORTED,
                       "PERSONALIZED with the adjusted cosine: a user with <= 4 ratings makes the reference's summation order depend on its memo history pair by pair (SURVEY N6); not modelled");
    }
    NeighborTable& pt = h->pt;
    pt.k = pt.kcap = tr.U;
    const size_t cells = (size_t)tr.U * (size_t)tr.U;
    pt.uidx.ensure(cells);
    pt.usim.ensure(cells);
    pt.cnt.ensure(tr.U);
    launch_full_rows(tr, h->cfg.similarity == KNNCF_SIM_JACCARD, pt.uidx.p, pt.usim.p, pt.cnt.p, h->stream);
    pt.by_id_valid = true;
    h->pt_ready = true;
}

// K4 on first use: computeItemAvg :141, computeItemAvgDev :193 and the Spark forms build their item maps when the predictor is
// constructed; the kNN predictor (:489-585) never does, so knncf_fit leaves them out (prep.hip: prep_item_stats)
void ensure_item_stats(knncf_handle* h) {
    if (h->tr.item_stats_ready) return;
    Stage s(h, &h->tm.prep_ms);
    h->prep.join_commit(h->stream);
    prep_item_stats(h->tr, h->prep, h->stream);
}

void run_predict(knncf_handle* h, int predictor, const int32_t* d_users, const int32_t* d_items,
                 const double* d_ratings, int64_t n, double* sum_abs_err, int64_t* count, double* d_pred_out) {
    require_fitted(h);
    Train& tr = h->tr;
    hipStream_t st = h->stream;
    KN_REQUIRE(n >= 0, KNNCF_E_INVALID, "negative row count");
    if (sum_abs_err) *sum_abs_err = 0.0;
    if (count) *count = 0;
    if (n == 0) return;
    KN_REQUIRE(d_users && d_items, KNNCF_E_INVALID, "null test arrays");
    int kind = predictor;
    NeighborTable* table = &h->nt;
    if (predictor == KNNCF_PRED_PERSONALIZED) {
        if (h->cfg.similarity == KNNCF_SIM_ONE) {
            kind = KNNCF_PRED_BASELINE_RDD;  // num/den = file-order mean of the item's deviations (see predict.hip)
        } else {  // the adjusted cosine / the Jaccard coefficient themselves: every user is a "neighbour"
            ensure_personalized_table(h);
            kind = KNNCF_PRED_KNN;
            table = &h->pt;
        }
    }
    KN_REQUIRE(kind >= KNNCF_PRED_GLOBAL_AVG && kind <= KNNCF_PRED_KNN, KNNCF_E_INVALID, "unknown predictor");
    if (kind == KNNCF_PRED_ITEM_AVG || kind == KNNCF_PRED_BASELINE || kind == KNNCF_PRED_BASELINE_RDD) ensure_item_stats(h);
    ensure_test_scratch(h, n);
    {
        Stage s(h, &h->tm.predict_ms);
        launch_dense_ids(tr, d_users, d_items, n, h->t_du.p, h->t_di.p, st);
    }
    if (kind == KNNCF_PRED_KNN && table == &h->nt) ensure_neighbors_for_rows(h, n);
    h->prep.join_commit(st);  // the item-major copies and the rater bitmaps (second part of prep_commit)
    {
        Stage s(h, &h->tm.predict_ms);
        double* pred = d_pred_out ? d_pred_out : h->t_pred.p;
        const uint32_t* d_order = nullptr;
        const bool by_item = tr.ib_words > 0 && tr.ib_words * 12 <= 48 * 1024;
        int64_t n_rows = n;  // rows the prediction kernel walks
        if (kind == KNNCF_PRED_KNN) {  // rows sorted by item (the item's rater bitmap lives in LDS) or else by user
            PrepScratch& sc = h->prep;
            sc.k64_a.ensure(n); sc.k64_b.ensure(n); sc.v32_a.ensure(n); sc.v32_b.ensure(n);
            const uint32_t key_limit = (uint32_t)(by_item ? tr.I : tr.U);  // the key of a row whose item / user the train set lacks
            if (h->cfg.shard_count > 1) {
                // the test set is replicated on every shard, the work is not: this shard's rows sort first and only they are
                // predicted; the other rows' error / ownership cells are cleared here instead of by the kernel
                h->scalar_u64.ensure(1);
                KN_HIP(hipMemsetAsync(h->scalar_u64.p, 0, sizeof(unsigned long long), st));
                KN_HIP(hipMemsetAsync(h->t_err.p, 0, (size_t)n * sizeof(double), st));
                KN_HIP(hipMemsetAsync(h->t_owned.p, 0, (size_t)n, st));
                launch_owned_keys(n, by_item ? h->t_di.p : h->t_du.p, h->t_du.p, tr.own_lo, tr.own_hi, h->cfg.shard_rank == 0, key_limit,
                                  sc.k64_a.p, sc.v32_a.p, h->scalar_u64.p, st);
                sort_pairs_u64_u32(sc.sort, sc.k64_a.p, sc.k64_b.p, sc.v32_a.p, sc.v32_b.p, n, bits_for((uint64_t)key_limit + 1), st);
                n_rows = (int64_t)fetch(h, h->scalar_u64.p, 0);
            } else {
                launch_user_keys(n, by_item ? h->t_di.p : h->t_du.p, key_limit, sc.k64_a.p, sc.v32_a.p, st);
                sort_pairs_u64_u32(sc.sort, sc.k64_a.p, sc.k64_b.p, sc.v32_a.p, sc.v32_b.p, n, bits_for((uint64_t)key_limit), st);
            }
            d_order = sc.v32_b.p;
        }
        if (n_rows > 0)
            launch_predict(tr, table, kind, n_rows, h->t_du.p, h->t_di.p, d_ratings, d_order, by_item, pred, h->t_err.p, h->t_owned.p,
                           h->cfg.shard_rank == 0, st);
        if (sum_abs_err || count) {
            const int32_t nb = 1024;
            launch_reduce_err(h->t_err.p, h->t_owned.p, n, h->t_partial.p, h->t_counts.p, nb, st);
            std::vector<double> hp(nb);
            std::vector<int64_t> hc(nb);
            KN_HIP(hipMemcpyAsync(hp.data(), h->t_partial.p, nb * sizeof(double), hipMemcpyDeviceToHost, st));
            KN_HIP(hipMemcpyAsync(hc.data(), h->t_counts.p, nb * sizeof(int64_t), hipMemcpyDeviceToHost, st));
            KN_HIP(hipStreamSynchronize(st));
            double s_ = 0.0;
            int64_t c_ = 0;
            for (int32_t b = 0; b < nb; ++b) { s_ += hp[b]; c_ += hc[b]; }
            if (sum_abs_err) *sum_abs_err = s_;
            if (count) *count = c_;
        }
    }
}

void do_fit_device(knncf_handle* h, const int32_t* d_users, const int32_t* d_items, const double* d_ratings, int64_t n) {
    KN_REQUIRE(n > 0 && d_users && d_items && d_ratings, KNNCF_E_INVALID, "fit: null or empty input");
    Train& tr = h->tr;
    hipStream_t st = h->stream;
    h->fitted = h->committed = false;
    h->b_ready = false;
    h->pt_ready = false;
    h->h_ukeys.clear(); h->h_ikeys.clear(); h->h_uid.clear();
    tr.n = n;
    tr.jaccard = h->cfg.similarity == KNNCF_SIM_JACCARD;
    {
        Stage s(h, &h->tm.prep_ms);
        if (tr.user_raw.p != d_users) {
            tr.user_raw.alloc(n); tr.item_raw.alloc(n); tr.rating.alloc(n);
            KN_HIP(hipMemcpyAsync(tr.user_raw.p, d_users, n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            KN_HIP(hipMemcpyAsync(tr.item_raw.p, d_items, n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            KN_HIP(hipMemcpyAsync(tr.rating.p, d_ratings, n * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        prep_fit(tr, h->prep, h->cfg.shard_rank, h->cfg.shard_count, st);
    }
    h->fitted = true;
    reset_neighbors(h);
    if (h->cfg.shard_count == 1) {
        Stage s(h, &h->tm.prep_ms);
        prep_commit(tr, h->prep, st);
        h->committed = true;
    }
}

// recommendations(ratings, predictor)(user, n) shared/predictions.scala:651-674
void do_recommend(knncf_handle* h, int predictor, int32_t user, int32_t n, int32_t* out_items, double* out_preds, int32_t* count) {
    require_fitted(h);
    KN_REQUIRE(count && n >= 0 && (n == 0 || (out_items && out_preds)), KNNCF_E_INVALID, "bad arguments");
    *count = 0;
    Train& tr = h->tr;
    if (n == 0 || tr.I == 0) return;
    hipStream_t st = h->stream;
    const int32_t du = dense_user(h, user);
    KN_REQUIRE(du < 0 ? h->cfg.shard_rank == 0 : (du >= tr.own_lo && du < tr.own_hi), KNNCF_E_STATE,
               "recommend: the user belongs to another shard");
    const int32_t I = tr.I;
    h->reco_users.ensure(I); h->reco_items.ensure(I); h->reco_pred.ensure(I); h->reco_rated.ensure(I);
    launch_reco_rows(tr, user, du, h->reco_users.p, h->reco_items.p, h->reco_rated.p, st);
    // one prediction batch over every train item (the rated ones are dropped by the ordering below)
    run_predict(h, predictor, h->reco_users.p, h->reco_items.p, nullptr, I, nullptr, nullptr, h->reco_pred.p);
    PrepScratch& sc = h->prep;
    sc.k64_a.ensure(I); sc.k64_b.ensure(I); sc.v32_a.ensure(I); sc.v32_b.ensure(I);
    launch_reco_order(tr, sc.sort, h->reco_pred.p, h->reco_rated.p, sc.k64_a.p, sc.k64_b.p, sc.v32_a.p, sc.v32_b.p, st);
    int64_t n_rated = 0;
    if (du >= 0) {
        int64_t two[2];
        KN_HIP(hipMemcpyAsync(two, tr.u_ptr.p + du, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        KN_HIP(hipStreamSynchronize(st));
        n_rated = two[1] - two[0];
    }
    const int32_t m = (int32_t)std::min<int64_t>(n, (int64_t)I - n_rated);
    if (m <= 0) return;
    h->reco_out_items.ensure(m); h->reco_out_preds.ensure(m);
    launch_reco_take(tr, m, sc.v32_b.p, h->reco_pred.p, h->reco_out_items.p, h->reco_out_preds.p, st);
    KN_HIP(hipMemcpyAsync(out_items, h->reco_out_items.p, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    KN_HIP(hipMemcpyAsync(out_preds, h->reco_out_preds.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, st));
    KN_HIP(hipStreamSynchronize(st));
    *count = m;
}

// ---- checkpoint / resume of the neighbour table ------------------------------------------------------
struct NbrFileHeader {
    char magic[8];  // "KNNCFNB1"
    int32_t U, kcap, k, similarity;
    int64_t n;
    uint64_t fingerprint;
    int64_t epoch;
};

// FNV-1a over what identifies "the same fit": raw user ids in dense order, row extents, user means
uint64_t fit_fingerprint(knncf_handle* h) {
    Train& tr = h->tr;
    std::vector<int32_t> uid(tr.U);
    std::vector<int64_t> ptr((size_t)tr.U + 1);
    std::vector<double> avg(tr.U);
    KN_HIP(hipMemcpyAsync(uid.data(), tr.uid.p, (size_t)tr.U * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipMemcpyAsync(ptr.data(), tr.u_ptr.p, ((size_t)tr.U + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipMemcpyAsync(avg.data(), tr.user_avg.p, (size_t)tr.U * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
    uint64_t x = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t bytes) {
        const unsigned char* c = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < bytes; ++i) { x ^= c[i]; x *= 1099511628211ull; }
    };
    mix(uid.data(), uid.size() * sizeof(int32_t));
    mix(ptr.data(), ptr.size() * sizeof(int64_t));
    mix(avg.data(), avg.size() * sizeof(double));
    return x;
}

void do_neighbors_save(knncf_handle* h, const char* path) {
    require_fitted(h);
    KN_REQUIRE(path, KNNCF_E_INVALID, "null path");
    KN_REQUIRE(h->cfg.shard_count == 1, KNNCF_E_UNSUPPORTED, "neighbour checkpoints are written by unsharded handles");
    Train& tr = h->tr;
    NeighborTable& nt = h->nt;
    NbrFileHeader hd{};
    memcpy(hd.magic, "KNNCFNB1", 8);
    hd.U = tr.U; hd.kcap = nt.kcap; hd.k = nt.k; hd.similarity = h->cfg.similarity; hd.n = tr.n;
    hd.fingerprint = fit_fingerprint(h);
    hd.epoch = h->epoch;
    const size_t cells = (size_t)tr.U * (size_t)std::max(nt.kcap, 1);
    std::vector<int32_t> cnt(tr.U), idx(cells);
    std::vector<int64_t> seq(tr.U);
    std::vector<double> sim(cells);
    KN_HIP(hipMemcpyAsync(cnt.data(), nt.cnt.p, cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipMemcpyAsync(seq.data(), nt.seq.p, seq.size() * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipMemcpyAsync(idx.data(), nt.idx.p, idx.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipMemcpyAsync(sim.data(), nt.sim.p, sim.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
    FILE* f = fopen(path, "wb");
    KN_REQUIRE(f, KNNCF_E_INVALID, std::string("cannot create ") + path);
    bool ok = fwrite(&hd, sizeof hd, 1, f) == 1 && fwrite(cnt.data(), sizeof(int32_t), cnt.size(), f) == cnt.size() &&
              fwrite(seq.data(), sizeof(int64_t), seq.size(), f) == seq.size() &&
              fwrite(idx.data(), sizeof(int32_t), idx.size(), f) == idx.size() &&
              fwrite(sim.data(), sizeof(double), sim.size(), f) == sim.size();
    ok = (fclose(f) == 0) && ok;
    KN_REQUIRE(ok, KNNCF_E_INVALID, std::string("short write to ") + path);
}

void do_neighbors_load(knncf_handle* h, const char* path) {
    require_fitted(h);
    KN_REQUIRE(path, KNNCF_E_INVALID, "null path");
    KN_REQUIRE(h->cfg.shard_count == 1, KNNCF_E_UNSUPPORTED, "neighbour checkpoints are read by unsharded handles");
    Train& tr = h->tr;
    NeighborTable& nt = h->nt;
    FILE* f = fopen(path, "rb");
    KN_REQUIRE(f, KNNCF_E_INVALID, std::string("cannot open ") + path);
    NbrFileHeader hd{};
    bool ok = fread(&hd, sizeof hd, 1, f) == 1 && memcmp(hd.magic, "KNNCFNB1", 8) == 0;
    if (!ok) { fclose(f); throw Error(KNNCF_E_INVALID, std::string(path) + ": not a neighbour checkpoint"); }
    if (hd.U != tr.U || hd.kcap != nt.kcap || hd.k != nt.k || hd.similarity != h->cfg.similarity || hd.n != tr.n ||
        hd.fingerprint != fit_fingerprint(h)) {
        fclose(f);
        throw Error(KNNCF_E_STATE, std::string(path) + ": checkpoint of a different fit (users, ratings, k or similarity differ)");
    }
    const size_t cells = (size_t)tr.U * (size_t)std::max(nt.kcap, 1);
    std::vector<int32_t> cnt(tr.U), idx(cells);
    std::vector<int64_t> seq(tr.U);
    std::vector<double> sim(cells);
    ok = fread(cnt.data(), sizeof(int32_t), cnt.size(), f) == cnt.size() && fread(seq.data(), sizeof(int64_t), seq.size(), f) == seq.size() &&
         fread(idx.data(), sizeof(int32_t), idx.size(), f) == idx.size() && fread(sim.data(), sizeof(double), sim.size(), f) == sim.size();
    fclose(f);
    KN_REQUIRE(ok, KNNCF_E_INVALID, std::string(path) + ": truncated checkpoint");
    hipStream_t st = h->stream;
    KN_HIP(hipMemcpyAsync(nt.cnt.p, cnt.data(), cnt.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    KN_HIP(hipMemcpyAsync(nt.seq.p, seq.data(), seq.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    KN_HIP(hipMemcpyAsync(nt.idx.p, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    KN_HIP(hipMemcpyAsync(nt.sim.p, sim.data(), sim.size() * sizeof(double), hipMemcpyHostToDevice, st));
    nt.by_id_valid = false;
    KN_HIP(hipStreamSynchronize(st));
    h->epoch = std::max<int64_t>(hd.epoch, 1);
}

}  // namespace

extern "C" {

const char* knncf_version(void) { return "knncf 0.1 (gfx950)"; }

const char* knncf_status_string(int st) {
    switch (st) {
        case KNNCF_OK: return "ok";
        case KNNCF_E_INVALID: return "invalid argument";
        case KNNCF_E_NONFINITE: return "non-finite normalized deviation";
        case KNNCF_E_DUPLICATE: return "duplicate (user,item) rating";
        case KNNCF_E_NOMEM: return "out of memory";
        case KNNCF_E_HIP: return "HIP runtime error";
        case KNNCF_E_STATE: return "invalid call order";
        case KNNCF_E_UNSUPPORTED: return "unsupported configuration";
        case KNNCF_E_NODEVICE: return "no usable gfx950 device";
        case KNNCF_E_RCCL: return "RCCL error";
        default: return "unknown status";
    }
}

int knncf_create(const knncf_config* cfg, knncf_handle** out) {
    if (!cfg || !out || cfg->struct_size != sizeof(knncf_config)) return KNNCF_E_INVALID;
    if (cfg->shard_count < 1 || cfg->shard_rank < 0 || cfg->shard_rank >= cfg->shard_count || cfg->k < 0)
        return KNNCF_E_INVALID;
    if (cfg->similarity < KNNCF_SIM_COSINE || cfg->similarity > KNNCF_SIM_JACCARD) return KNNCF_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        (void)hipGetLastError();
        return KNNCF_E_NODEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return KNNCF_E_NODEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return KNNCF_E_NODEVICE;  // CDNA4 code objects only
    knncf_handle* h = new (std::nothrow) knncf_handle();
    if (!h) return KNNCF_E_NOMEM;
    h->cfg = *cfg;
    if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return KNNCF_E_HIP;
    }
    bool ok = hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming) == hipSuccess;
    for (int s = 0; s < 2; ++s) {
        ok = ok && hipEventCreateWithFlags(&h->ev_produced[s], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&h->ev_consumed[s], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        knncf_destroy(h);
        return KNNCF_E_HIP;
    }
    *out = h;
    return KNNCF_OK;
}

void knncf_destroy(knncf_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    (void)hipStreamSynchronize(h->stream);
    for (auto& t : h->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto e : h->event_pool) (void)hipEventDestroy(e);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
    for (int s = 0; s < 2; ++s) {
        if (h->ev_produced[s]) (void)hipEventDestroy(h->ev_produced[s]);
        if (h->ev_consumed[s]) (void)hipEventDestroy(h->ev_consumed[s]);
    }
    if (h->pinned_cnt) (void)hipHostFree(h->pinned_cnt);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    (void)hipStreamDestroy(h->stream);
    delete h;
}

const char* knncf_last_error(const knncf_handle* h) { return h ? h->err.c_str() : "null handle"; }

int knncf_fit_device(knncf_handle* h, const int32_t* d_users, const int32_t* d_items, const double* d_ratings, int64_t n) {
    return guarded(h, [&] { do_fit_device(h, d_users, d_items, d_ratings, n); });
}

int knncf_fit(knncf_handle* h, const int32_t* users, const int32_t* items, const double* ratings, int64_t n) {
    return guarded(h, [&] {
        KN_REQUIRE(n > 0 && users && items && ratings, KNNCF_E_INVALID, "fit: null or empty input");
        Train& tr = h->tr;
        tr.user_raw.alloc(n); tr.item_raw.alloc(n); tr.rating.alloc(n);
        KN_HIP(hipMemcpyAsync(tr.user_raw.p, users, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        KN_HIP(hipMemcpyAsync(tr.item_raw.p, items, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        KN_HIP(hipMemcpyAsync(tr.rating.p, ratings, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        do_fit_device(h, tr.user_raw.p, tr.item_raw.p, tr.rating.p, n);
    });
}

int knncf_num_users(const knncf_handle* h, int32_t* out) {
    if (!h || !out || !h->fitted) return KNNCF_E_STATE;
    *out = h->tr.U;
    return KNNCF_OK;
}
int knncf_num_items(const knncf_handle* h, int32_t* out) {
    if (!h || !out || !h->fitted) return KNNCF_E_STATE;
    *out = h->tr.I;
    return KNNCF_OK;
}

int knncf_global_avg(knncf_handle* h, double* out) {
    return guarded(h, [&] {
        require_fitted(h, false);
        KN_REQUIRE(out, KNNCF_E_INVALID, "null out");
        *out = h->tr.global_avg;
    });
}

int knncf_user_avg(knncf_handle* h, int32_t user, double* out) {
    return guarded(h, [&] {
        require_fitted(h);
        KN_REQUIRE(out, KNNCF_E_INVALID, "null out");
        int32_t d = dense_user(h, user);
        *out = d >= 0 ? fetch(h, h->tr.user_avg.p, d) : h->tr.global_avg;
    });
}

int knncf_item_avg(knncf_handle* h, int32_t item, double* out) {
    return guarded(h, [&] {
        require_fitted(h);
        KN_REQUIRE(out, KNNCF_E_INVALID, "null out");
        ensure_item_stats(h);
        int32_t d = dense_item(h, item);
        *out = d >= 0 ? fetch(h, h->tr.item_avg.p, d) : h->tr.global_avg;
    });
}

int knncf_item_avg_dev(knncf_handle* h, int32_t item, double* out) {
    return guarded(h, [&] {
        require_fitted(h);
        KN_REQUIRE(out, KNNCF_E_INVALID, "null out");
        ensure_item_stats(h);
        int32_t d = dense_item(h, item);
        *out = d >= 0 ? fetch(h, h->tr.item_dev_hash.p, d) : 0.0;
    });
}

int knncf_item_avg_dev_rdd(knncf_handle* h, int32_t item, double* out) {
    return guarded(h, [&] {
        require_fitted(h);
        KN_REQUIRE(out, KNNCF_E_INVALID, "null out");
        ensure_item_stats(h);
        int32_t d = dense_item(h, item);
        *out = d >= 0 ? fetch(h, h->tr.item_dev_file.p, d) : 0.0;
    });
}

int knncf_similarity(knncf_handle* h, int32_t u, int32_t v, double* out) {
    return guarded(h, [&] {
        require_fitted(h);
        KN_REQUIRE(out, KNNCF_E_INVALID, "null out");
        if (h->cfg.similarity == KNNCF_SIM_ONE) { *out = 1.0; return; }
        int32_t du = dense_user(h, u), dv = dense_user(h, v);
        h->scalar_out.ensure(1);
        if (h->cfg.similarity == KNNCF_SIM_JACCARD) {
            launch_jaccard_pair(h->tr, du, dv, h->scalar_out.p, h->stream);
        } else {
            if (du < 0 || dv < 0) { *out = 0.0; return; }
            launch_exact_pair(h->tr, du, dv, h->scalar_out.p, h->stream);
        }
        *out = fetch(h, h->scalar_out.p, 0);
    });
}

static void neighbors_of(knncf_handle* h, int32_t du, std::vector<int32_t>& ids, std::vector<double>& sims) {
    NeighborTable& nt = h->nt;
    Train& tr = h->tr;
    ids.clear(); sims.clear();
    if (tr.U < 2 || nt.kcap <= 0) return;
    int64_t seq = fetch(h, nt.seq.p, du);
    if (seq < 0) {
        h->build_list.ensure(tr.U);
        int64_t new_seq = h->epoch << 32;
        h->epoch += 1;
        KN_HIP(hipMemcpyAsync(nt.seq.p + du, &new_seq, sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
        KN_HIP(hipMemcpyAsync(h->build_list.p, &du, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        KN_HIP(hipStreamSynchronize(h->stream));
        build_neighbors(h, 1);
    }
    int32_t cnt = fetch(h, nt.cnt.p, du);
    ids.resize(cnt); sims.resize(cnt);
    if (cnt > 0) {
        KN_HIP(hipMemcpyAsync(ids.data(), nt.idx.p + (int64_t)du * nt.kcap, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        KN_HIP(hipMemcpyAsync(sims.data(), nt.sim.p + (int64_t)du * nt.kcap, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        KN_HIP(hipStreamSynchronize(h->stream));
    }
}

int knncf_neighbors(knncf_handle* h, int32_t u, int32_t cap, int32_t* ids, double* sims, int32_t* count) {
    return guarded(h, [&] {
        require_fitted(h);
        KN_REQUIRE(count && cap >= 0 && (cap == 0 || (ids && sims)), KNNCF_E_INVALID, "bad output arguments");
        KN_REQUIRE(h->cfg.similarity != KNNCF_SIM_ONE, KNNCF_E_UNSUPPORTED, "neighbourhoods: adjusted cosine or Jaccard");
        int32_t du = dense_user(h, u);
        load_host_ids(h);
        if (du < 0) {  // user absent from train: every similarity is 0.0, ties keep Set order (N3)
            int32_t c = std::min(h->nt.k, h->tr.U);
            for (int32_t j = 0; j < c && j < cap; ++j) { ids[j] = h->h_uid[j]; sims[j] = 0.0; }
            *count = c;
            return;
        }
        KN_REQUIRE(du >= h->tr.own_lo && du < h->tr.own_hi, KNNCF_E_INVALID, "user belongs to another shard");
        std::vector<int32_t> di;
        std::vector<double> ds;
        neighbors_of(h, du, di, ds);
        for (size_t j = 0; j < di.size() && (int32_t)j < cap; ++j) { ids[j] = h->h_uid[di[j]]; sims[j] = ds[j]; }
        *count = (int32_t)di.size();
    });
}

// getNeighbors(train, k, sim) for many users: the missing neighbourhoods are built in ONE batch (memo history: as if the
// closure had been called for users[0], users[1], ... in this order), then the lists are copied out
static void do_neighbors_batch(knncf_handle* h, const int32_t* users, int64_t n, int32_t cap, int32_t* ids, double* sims, int32_t* counts) {
    require_fitted(h);
    KN_REQUIRE(n >= 0 && cap >= 0 && (n == 0 || (users && counts)) && (n == 0 || cap == 0 || (ids && sims)), KNNCF_E_INVALID, "bad arguments");
    KN_REQUIRE(h->cfg.similarity != KNNCF_SIM_ONE, KNNCF_E_UNSUPPORTED, "neighbourhoods: adjusted cosine or Jaccard");
    if (n == 0) return;
    Train& tr = h->tr;
    NeighborTable& nt = h->nt;
    hipStream_t st = h->stream;
    load_host_ids(h);
    std::vector<int32_t> du((size_t)n);
    for (int64_t j = 0; j < n; ++j) {
        du[j] = dense_lookup(h->h_ukeys.data(), tr.U, users[j]);
        KN_REQUIRE(du[j] < 0 || (du[j] >= tr.own_lo && du[j] < tr.own_hi), KNNCF_E_INVALID, "user belongs to another shard");
    }
    const bool have_lists = tr.U >= 2 && nt.kcap > 0;
    if (have_lists) {
        std::vector<int64_t> seq((size_t)tr.U);
        KN_HIP(hipMemcpyAsync(seq.data(), nt.seq.p, seq.size() * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        KN_HIP(hipStreamSynchronize(st));
        std::vector<int32_t> fresh;
        for (int64_t j = 0; j < n; ++j)
            if (du[j] >= 0 && seq[du[j]] < 0) {
                seq[du[j]] = (h->epoch << 32) | (int64_t)std::min<int64_t>(j, 0xffffffffll);
                fresh.push_back(du[j]);
            }
        if (!fresh.empty()) {
            h->epoch += 1;
            h->build_list.ensure(tr.U);
            KN_HIP(hipMemcpyAsync(nt.seq.p, seq.data(), seq.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
            KN_HIP(hipMemcpyAsync(h->build_list.p, fresh.data(), fresh.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            KN_HIP(hipStreamSynchronize(st));
            build_neighbors(h, (int32_t)fresh.size());
        }
    }
    const size_t kc = (size_t)std::max(nt.kcap, 1);
    std::vector<int32_t> h_cnt, h_idx;
    std::vector<double> h_sim;
    const bool whole = have_lists && n > 1024;  // one big copy instead of 2 n small ones
    if (have_lists) {
        h_cnt.resize(tr.U);
        KN_HIP(hipMemcpyAsync(h_cnt.data(), nt.cnt.p, h_cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (whole) {
            h_idx.resize((size_t)tr.U * kc);
            h_sim.resize((size_t)tr.U * kc);
            KN_HIP(hipMemcpyAsync(h_idx.data(), nt.idx.p, h_idx.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            KN_HIP(hipMemcpyAsync(h_sim.data(), nt.sim.p, h_sim.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        KN_HIP(hipStreamSynchronize(st));
    }
    std::vector<int32_t> row_idx(kc);
    std::vector<double> row_sim(kc);
    for (int64_t j = 0; j < n; ++j) {
        int32_t* oi = ids + (size_t)j * cap;
        double* os = sims + (size_t)j * cap;
        if (du[j] < 0) {  // user absent from train: every similarity is 0.0, ties keep Set order (N3)
            const int32_t c = std::min(nt.k, tr.U);
            for (int32_t q = 0; q < c && q < cap; ++q) { oi[q] = h->h_uid[q]; os[q] = 0.0; }
            counts[j] = c;
            continue;
        }
        const int32_t c = have_lists ? h_cnt[du[j]] : 0;
        const int32_t* src_i = nullptr;
        const double* src_s = nullptr;
        if (whole) {
            src_i = h_idx.data() + (size_t)du[j] * kc;
            src_s = h_sim.data() + (size_t)du[j] * kc;
        } else if (c > 0) {
            KN_HIP(hipMemcpyAsync(row_idx.data(), nt.idx.p + (size_t)du[j] * kc, c * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            KN_HIP(hipMemcpyAsync(row_sim.data(), nt.sim.p + (size_t)du[j] * kc, c * sizeof(double), hipMemcpyDeviceToHost, st));
            KN_HIP(hipStreamSynchronize(st));
            src_i = row_idx.data();
            src_s = row_sim.data();
        }
        for (int32_t q = 0; q < c && q < cap; ++q) { oi[q] = h->h_uid[src_i[q]]; os[q] = src_s[q]; }
        counts[j] = c;
    }
}

int knncf_neighbors_batch(knncf_handle* h, const int32_t* users, int64_t n, int32_t cap, int32_t* ids, double* sims, int32_t* counts) {
    return guarded(h, [&] { do_neighbors_batch(h, users, n, cap, ids, sims, counts); });
}

int knncf_knn_similarity(knncf_handle* h, int32_t u, int32_t v, double* out) {
    return guarded(h, [&] {
        require_fitted(h);
        KN_REQUIRE(out, KNNCF_E_INVALID, "null out");
        KN_REQUIRE(h->cfg.similarity != KNNCF_SIM_ONE, KNNCF_E_UNSUPPORTED, "neighbourhoods: adjusted cosine or Jaccard");
        int32_t du = dense_user(h, u), dv = dense_user(h, v);
        *out = 0.0;
        if (du < 0 || dv < 0) return;  // all of an unseen user's similarities are 0.0
        KN_REQUIRE(du >= h->tr.own_lo && du < h->tr.own_hi, KNNCF_E_INVALID, "user belongs to another shard");
        std::vector<int32_t> di;
        std::vector<double> ds;
        neighbors_of(h, du, di, ds);
        for (size_t j = 0; j < di.size(); ++j)
            if (di[j] == dv) { *out = 0.0 + ds[j]; break; }
    });
}

int knncf_predict_batch_device(knncf_handle* h, int predictor, const int32_t* d_users, const int32_t* d_items,
                               int64_t n, double* d_out) {
    return guarded(h, [&] {
        KN_REQUIRE(d_out || n == 0, KNNCF_E_INVALID, "null output");
        run_predict(h, predictor, d_users, d_items, nullptr, n, nullptr, nullptr, d_out);
    });
}

int knncf_predict_batch(knncf_handle* h, int predictor, const int32_t* users, const int32_t* items, int64_t n, double* out) {
    return guarded(h, [&] {
        KN_REQUIRE(n >= 0 && (n == 0 || (users && items && out)), KNNCF_E_INVALID, "bad arguments");
        if (n == 0) return;
        h->t_users.ensure(n); h->t_items.ensure(n); h->t_pred.ensure(n);
        KN_HIP(hipMemcpyAsync(h->t_users.p, users, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        KN_HIP(hipMemcpyAsync(h->t_items.p, items, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        run_predict(h, predictor, h->t_users.p, h->t_items.p, nullptr, n, nullptr, nullptr, nullptr);
        KN_HIP(hipMemcpyAsync(out, h->t_pred.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        KN_HIP(hipStreamSynchronize(h->stream));
    });
}

int knncf_predict(knncf_handle* h, int predictor, int32_t user, int32_t item, double* out) {
    return knncf_predict_batch(h, predictor, &user, &item, 1, out);
}

int knncf_recommend(knncf_handle* h, int predictor, int32_t user, int32_t n, int32_t* items, double* predictions, int32_t* count) {
    return guarded(h, [&] { do_recommend(h, predictor, user, n, items, predictions, count); });
}

int knncf_neighbors_save(knncf_handle* h, const char* path) {
    return guarded(h, [&] { do_neighbors_save(h, path); });
}

int knncf_neighbors_load(knncf_handle* h, const char* path) {
    return guarded(h, [&] { do_neighbors_load(h, path); });
}

int knncf_mae_device(knncf_handle* h, int predictor, const int32_t* d_users, const int32_t* d_items,
                     const double* d_ratings, int64_t n, double* sum_abs_err, int64_t* count, double* d_pred) {
    return guarded(h, [&] {
        KN_REQUIRE(sum_abs_err && count && (n == 0 || d_ratings), KNNCF_E_INVALID, "bad arguments");
        run_predict(h, predictor, d_users, d_items, d_ratings, n, sum_abs_err, count, d_pred);
    });
}

int knncf_mae(knncf_handle* h, int predictor, const int32_t* users, const int32_t* items, const double* ratings,
              int64_t n, double* mae) {
    return guarded(h, [&] {
        KN_REQUIRE(mae && n >= 0 && (n == 0 || (users && items && ratings)), KNNCF_E_INVALID, "bad arguments");
        KN_REQUIRE(h->cfg.shard_count == 1, KNNCF_E_STATE, "sharded handle: use knncf_mae_device and all-reduce the partial sums");
        if (n == 0) { *mae = NAN; return; }  // 0.0 / 0 in applyAndMean :85
        h->t_users.ensure(n); h->t_items.ensure(n); h->t_ratings.ensure(n);
        KN_HIP(hipMemcpyAsync(h->t_users.p, users, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        KN_HIP(hipMemcpyAsync(h->t_items.p, items, n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        KN_HIP(hipMemcpyAsync(h->t_ratings.p, ratings, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        double s = 0.0;
        int64_t c = 0;
        run_predict(h, predictor, h->t_users.p, h->t_items.p, h->t_ratings.p, n, &s, &c, nullptr);
        *mae = s / (double)n;
    });
}

int knncf_shard_view_get(knncf_handle* h, knncf_shard_view* out) {
    return guarded(h, [&] {
        require_fitted(h, false);
        KN_REQUIRE(out, KNNCF_E_INVALID, "null out");
        Train& tr = h->tr;
        out->user_begin = tr.own_lo;
        out->user_end = tr.own_hi;
        out->nnz_begin = tr.own_p0;  // (read back once by prep_fit: no device round trip per view)
        out->nnz_end = tr.own_p1;
        out->num_users = tr.U;
        out->num_ratings = tr.n;
        out->d_user_avg = tr.user_avg.p;
        out->d_user_norm = tr.user_norm.p;
        out->d_dev = tr.s_dev.p;
        out->d_pre = tr.s_pre.p;
    });
}

int knncf_shard_commit(knncf_handle* h) {
    return guarded(h, [&] {
        require_fitted(h, false);
        if (h->committed) return;
        Stage s(h, &h->tm.prep_ms);
        if (h->cfg.shard_count > 1) prep_complete_rows(h->tr, h->prep, h->stream);
        prep_commit(h->tr, h->prep, h->stream);
        h->committed = true;
    });
}

int knncf_get_timings(const knncf_handle* h, knncf_timings* out) {
    if (!h || !out) return KNNCF_E_INVALID;
    *out = h->tm;
    return KNNCF_OK;
}

int knncf_reset_timings(knncf_handle* h) {
    if (!h) return KNNCF_E_INVALID;
    h->tm = knncf_timings{};
    return KNNCF_OK;
}

int knncf_reset_neighbors(knncf_handle* h) {
    return guarded(h, [&] {
        require_fitted(h, false);
        reset_neighbors(h);
    });
}

int knncf_set_k(knncf_handle* h, int32_t k) {
    return guarded(h, [&] {
        KN_REQUIRE(k >= 0, KNNCF_E_INVALID, "negative k");
        h->cfg.k = k;
        if (h->fitted) reset_neighbors(h);
    });
}

}  // extern "C"
