// engine.h — internal state of one knncf handle and the launcher interface between the host
// orchestrator (api.cpp) and the kernel translation units.  gfx950 only.
#pragma once

#include <string>
#include <vector>

#include "common.h"

namespace knncf {

typedef __bf16 bf16_t;

// columns of a similarity row that select.hip holds in LDS at a time (prep.hip tabulates the tile crossings)
// provisional store of select.hip per similarity row: groups of 8 columns (36 B each); the capacity scales with k
inline int32_t select_gcap(int32_t k) { return 8192 * (int32_t)((k + 511) / 512 > 1 ? (k + 511) / 512 : 1); }
#ifndef KNNCF_TCOLS
#define KNNCF_TCOLS 16384  // (A/B switch: 8192 = half-size tiles, three workgroups of k_tail_select per CU)
#endif
static constexpr int SELECT_TCOLS = KNNCF_TCOLS;  // <= 2^14: it_pack keeps the BYTE address of the column's LDS cell inside its tile in 16 bits
static_assert(SELECT_TCOLS <= 16384 && (SELECT_TCOLS & (SELECT_TCOLS - 1)) == 0, "it_pack: 16-bit cell byte address");

// ---- sort_util.hip (stable LSD radix sort / sorted-unique, hand-written; K0 plumbing) ----
struct SortWorkspace {
    DArr<char> tmp;
    DArr<size_t> count;  // unique_u32's per-tile counts and result cell (raw storage; kept: a hipFree per call synchronises the device)
};
void sort_pairs_u64_u32(SortWorkspace& ws, const uint64_t* kin, uint64_t* kout, const uint32_t* vin,
                        uint32_t* vout, size_t n, int end_bit, hipStream_t st);
void sort_keys_u32(SortWorkspace& ws, const uint32_t* kin, uint32_t* kout, size_t n, hipStream_t st);
void sort_pairs_u32_u32(SortWorkspace& ws, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n,
                        int end_bit, hipStream_t st);
// out must hold n entries; returns the number of distinct values (synchronises the stream)
size_t unique_u32(SortWorkspace& ws, const uint32_t* sorted_in, uint32_t* out, size_t n, hipStream_t st);

// ---- prep.hip: K0-K4 -------------------------------------------------------------------
struct Train {
    int64_t n = 0;
    int32_t U = 0, I = 0;
    // raw rows, file order
    DArr<int32_t> user_raw, item_raw;
    DArr<double> rating;
    // dense ids: trie keys of the distinct raw ids, in dense order (see dense_lookup)
    DArr<uint32_t> ukeys, ikeys;
    DArr<int32_t> uid, iid;  // raw id of dense index
    // raw id -> dense index as a direct table (MovieLens ids are small non-negative integers): one gather per row instead
    // of a hash + binary search; empty (n = 0) when an id is negative or >= 2^24 — then dense_lookup is used
    DArr<int32_t> u_table, i_table;
    int32_t u_table_n = 0, i_table_n = 0;
    // canonical user-major order ("position" p): users ascending, inside a user items ascending.
    // dense item index == rank in HashSet iteration order, so ascending p inside a user IS the
    // reference's summation order N2 for users with > 4 ratings.
    DArr<int64_t> u_ptr;     // [U+1]
    DArr<int32_t> s_user;    // [n]
    DArr<int32_t> s_col;     // [n]
    DArr<uint32_t> s_t;      // [n] file row of position p
    DArr<double> s_rating;   // [n]
    DArr<double> s_dev;      // [n] computeNormalizeDeviation :155-169
    DArr<double> s_pre;      // [n] preprocessedRating :470-481
    // fold orders (permutations of positions)
    DArr<uint32_t> perm_uf;  // (user, file order)            usersAvg :113
    DArr<uint32_t> perm_uh;  // (user, HashMap order, N4)      usersWeights :474
    DArr<int64_t> i_ptr;     // [I+1]
    DArr<double> user_avg, user_norm;  // [U]
    // K4 (itemsAvg :134, itemsAvgDev :176-186, getItemsAvgDev :336-343): built on first use by prep_item_stats — the kNN
    // path of the reference never evaluates them
    DArr<double> item_avg, item_dev_hash, item_dev_file;  // [I]
    bool item_stats_ready = false;
    // item-major copies for the sparse tail of the hybrid similarity (fp32 is enough: it only filters)
    DArr<int32_t> it_user;   // [n] dense user of the q-th entry in (item, user ascending) order
    DArr<uint32_t> it_pack;  // [n] the sparse tail's 4-byte entry: value field (16 bits, high) | byte address of the LDS cell of (user mod SELECT_TCOLS) — prep.hip: k_item_major
    DArr<double> it_dev;     // [n] normalized deviation of that entry (prediction gathers)
    DArr<uint32_t> it_t;     // [n] training file row of that entry (order of ratedI(i) :508-517)
    // per-item rater bitmaps over the dense user index + per-word exclusive rank prefixes: "did user x rate
    // item i, and where is that rating" is one 8-byte read (+ one on a hit) instead of a binary search
    int64_t ib_words = 0;        // 64-bit words per item row = ceil(U / 64); 0 = not built (too large)
    DArr<uint64_t> item_bits;    // [I * ib_words]
    DArr<uint32_t> item_rank;    // [I * ib_words]
    // where each item's rater list crosses the column tiles of select.hip: it_tile[i][t] = first entry q of item i
    // with it_user[q] >= t * SELECT_TCOLS (t = 0 .. tile_stride-1; the last one is the list's end)
    int32_t tile_stride = 0;
    DArr<uint32_t> it_tile;      // [I * tile_stride]
    DArr<int32_t> pop_item;  // [I] dense items by descending number of raters
    std::vector<int64_t> pop_count;  // host: rater counts in that order
    double global_avg = 0.0;
    int32_t own_lo = 0, own_hi = 0;  // owned dense users [lo, hi)
    int64_t own_p0 = 0, own_p1 = 0;  // their positions in the canonical order (everything when not sharded)
    // the handle's similarity is jaccardCoefficient :440-464: the similarity stage then counts common items — 0/1 operand
    // panels, tail entries of value 1 — instead of summing products of preprocessed ratings
    bool jaccard = false;
};

// status word bits written by kernels
enum : uint32_t { ST_NONFINITE = 1u, ST_DUPLICATE = 2u, ST_NOT_DYADIC = 4u, ST_LONG_ROW = 16u };

struct PrepScratch {
    SortWorkspace sort;
    DArr<uint64_t> k64_a, k64_b;
    DArr<uint32_t> v32_a, v32_b, k32_a, k32_b;
    DArr<int32_t> du_row, di_row;
    DArr<uint32_t> perm_f;
    DArr<uint32_t> status;  // [4] device status words
    DArr<int32_t> idrange;  // [4] min / max raw user id, min / max raw item id
    DArr<uint32_t> ucnt, utile;  // [U] ratings per user (then the scatter cursors), [U / 2048] their sums per tile
    DArr<int32_t> long_rows;  // [2 (U + 1)] users whose segment is sorted by the wider classes of k_user_hash_order
    DArr<double> dsum;      // small reduction scratch
    DArr<uint4> rec;        // [2 n] (preprocessed rating, deviation | user, file row) records: one 32-byte gather per entry
    // second stream of the fit: the per-user LDS sorts of the few long rows run beside those of everybody else
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_commit = nullptr;
    // prep_commit leaves its second part (item-major copies, tile table, rater bitmaps) running on `aux`: whoever reads
    // those (select.hip, predict.hip) or re-uses this scratch for more than a sort (prep_item_stats) joins first
    bool commit_pending = false;
    DArr<uint32_t> perm_iu;  // [n] positions in (item, user ascending) order
    void join_commit(hipStream_t st);
    void ensure_aux();
    void release_all();
    PrepScratch() = default;
    PrepScratch(const PrepScratch&) = delete;
    PrepScratch& operator=(const PrepScratch&) = delete;
    ~PrepScratch();
};

// K0 + K1 + owned part of K2/K3.  Throws Error on invalid data.
void prep_fit(Train& tr, PrepScratch& sc, int32_t shard_rank, int32_t shard_count, hipStream_t st);
// sharded handles, after the exchange of the per-user (mean, norm): the deviations and preprocessed ratings of the users
// this shard does NOT own — pure elementwise functions of (rating, the user's mean) and (deviation, the user's norm), so
// every rank recomputes them bit for bit instead of receiving 16 B per rating over xGMI
void prep_complete_rows(Train& tr, PrepScratch& sc, hipStream_t st);
// item-major copies, rater bitmaps, popularity order (need every user's deviations: after the shards' exchange)
void prep_commit(Train& tr, PrepScratch& sc, hipStream_t st);
// K4: the per-item statistics of the baseline predictors (after prep_commit; sets tr.item_stats_ready)
void prep_item_stats(Train& tr, PrepScratch& sc, hipStream_t st);

// raw test ids -> dense (-1 = absent from train)
void launch_dense_ids(const Train& tr, const int32_t* d_users, const int32_t* d_items, int64_t n,
                      int32_t* d_du, int32_t* d_di, hipStream_t st);

// ---- gemm.hip: densify + K5 ------------------------------------------------------------
// rows[r] = dense user of panel row r (nullptr: panel row r == user row_begin + r)
// (tr.jaccard: every present entry is written as 1.0 — the 0/1 panel whose GEMM counts common items)
void launch_densify(const Train& tr, const int32_t* d_rows, int32_t row_begin, int32_t n_rows,
                    const int32_t* d_colmap, bf16_t* panel, int64_t ld, int64_t panel_rows, bool fp16,
                    hipStream_t st);
// colmap[item] = column of the dense head panel (popularity rank < H) or -1 (tail)
void launch_colmap(const Train& tr, int32_t H, int32_t* d_colmap, hipStream_t st);
// C[M][ldc] (fp32 or fp16) = A[M][K] * B[N][K]^T, 16-bit in / fp32 accumulate; M, N multiples of 128, K of 64
// clamp: fp16 C entries are clamped to [-1, 1] before rounding (adjusted cosine: the exact value lies there); false for
// the counting GEMM of the Jaccard path (counts <= 2048 are exact in fp16)
void launch_gemm_nt(const bf16_t* A, const bf16_t* B, void* C, bool c_fp16, int64_t M, int64_t N, int64_t K,
                    int64_t lda, int64_t ldb, int64_t ldc, bool fp16, bool clamp, hipStream_t st);

// the whole symmetric matrix at once: C[N][ldc] = B B^T computed on and above the diagonal (256 x 256 tiles in the order of
// the tile list) and mirrored below it; the stored values are bit for bit those of launch_gemm_nt(B, B, ...)
void gemm_sym_tile_list(int32_t n_tiles, std::vector<uint32_t>& out, int32_t group = 8);
void launch_gemm_sym(const bf16_t* B, void* C, bool c_fp16, int64_t N, int64_t K, int64_t ldb, int64_t ldc, bool fp16, bool clamp,
                     const uint32_t* d_tile_list, int64_t n_listed, hipStream_t st, int tile = 256);

// ---- select.hip: K6 + K6b --------------------------------------------------------------
struct NeighborTable {
    int32_t k = 0;     // requested k
    int32_t kcap = 0;  // min(k, U-1): stored neighbours per user
    DArr<int32_t> idx;   // [U * kcap] dense neighbour ids, reference order
    DArr<double> sim;    // [U * kcap]
    DArr<int32_t> uidx;  // [U * kcap] the same neighbours sorted by dense id: built on demand (launch_predict) for the prediction
    DArr<double> usim;   // [U * kcap]   kernels that probe in global memory; the item-grouped kernel streams idx / sim as they are
    bool by_id_valid = false;  // uidx / usim hold the current lists
    DArr<int32_t> cnt;   // [U] 0 until built
    DArr<int64_t> seq;   // [U] build sequence number (memo history); -1 = not built
};

struct SelectScratch {
    DArr<int32_t> cand_idx;   // [rows * cap]
    DArr<float> cand_approx;  // [rows * cap] (KNNCF_FLAG_VERIFY_BOUND)
    DArr<int32_t> cand_cnt;   // [rows] (> cap == overflow)
    DArr<float> cand_eps;     // [rows] the error band of the row's approximate similarities
    DArr<int32_t> grp_v0;     // [rows * select_gcap(k)] provisional groups: first column
    DArr<float> grp_x;        // [rows * select_gcap(k) * 8] their 8 values
    DArr<double> stats;       // [4]: max bound violation, ...
    DArr<uint32_t> row_entries;  // [rows] ratings of the row's shortlisted candidates (re-rank traffic accounting)
    DArr<double> row_exact;   // fallback: [U] exact similarities of one row
    DArr<uint64_t> fb_keys_a, fb_keys_b;
    DArr<uint32_t> fb_vals_a, fb_vals_b;
};

// per panel row: S[r][:] += sparse tail (items with colmap < 0), then threshold + shortlist:
// candidates v with S[r][v] >= T_r - 2 eps_r; eps_r = eps_opnd * ||head part of row r|| + eps_rest + per-row terms
// per-row tail entry lists for the current head (select.hip: k_tail_entries); device pointers
struct TailEntries {
    const int32_t* cnt = nullptr;   // [U]
    const int32_t* item = nullptr;  // [n], row u's entries at u_ptr[u] ..
    const float* x = nullptr;       // [n]
    const float* tail_abs = nullptr;  // [U]
    const float* head_sq = nullptr;   // [U]
    const float* row_len = nullptr;   // Jaccard handles: [row_len_size(U)] |I(v)| as float
};
int64_t row_len_size(int32_t U);
void launch_row_len(const Train& tr, float* d_out, hipStream_t st);
void launch_tail_entries(const Train& tr, const int32_t* d_colmap, int32_t* te_cnt, int32_t* te_item, float* te_x,
                         float* row_tail_abs, float* row_head_sq, hipStream_t st);
// s_by_user: S is the whole matrix and row r's similarities are S[d_row_user[r]] (symmetric path); else S[r]
void launch_tail_select(const Train& tr, const int32_t* d_colmap, const TailEntries& te, bool has_tail, const void* S, bool s_by_user, bool s_fp16, int64_t lds,
                        int32_t n_rows, const int32_t* d_row_user, int32_t k, float eps_opnd, float eps_rest, int32_t cap,
                        int32_t* cand_idx, float* cand_approx, int32_t* cand_cnt, float* cand_eps, int32_t* grp_v0, float* grp_x,
                        int32_t gcap, hipStream_t st, bool anticipate = true, const int32_t* d_row_srow = nullptr);
// (d_row_srow: row-block panels — the panel row of launch row r when it is not r itself: the re-select of a subset of a block)
// (anticipate = false: the emission thresholds are the plain k-th largest value seen so far — api.cpp re-runs the rows of a
// build whose anticipated thresholds overshot too often that way)
// the first n_heavy rows of a re-rank launch as P slices of their shortlists + a merge (rerank.hip)
struct SliceScratch {
    DArr<int32_t> part_idx, part_cnt;
    DArr<uint32_t> entries;
    DArr<double> part_sim;
};
struct Slices {  // (kernel argument)
    int32_t n_heavy, P;
    int32_t* part_idx;
    double* part_sim;
    int32_t* part_cnt;
    uint32_t* entries;
};
// exact fp64 similarities of the shortlists in reference order, stable top-k; n_heavy > 0: the first n_heavy rows as `slices`
// slices each (2 <= slices, slices * k <= 8192; sc holds their partial lists)
void launch_rerank(const Train& tr, NeighborTable& nt, int32_t n_rows, const int32_t* d_row_user,
                   int32_t cap, const int32_t* cand_idx, const float* cand_approx,
                   const int32_t* cand_cnt, const float* cand_eps, double* d_stats, uint32_t* d_row_entries, bool verify,
                   hipStream_t st, int32_t n_heavy = 0, int32_t slices = 1, SliceScratch* sc = nullptr);
// exact similarities of one user against everyone (fallback + scalar queries)
void launch_exact_row(const Train& tr, const NeighborTable& nt, int32_t user, int64_t user_seq,
                      double* d_out, hipStream_t st);
// Personalized (no k): per user the ids (ascending, the user itself included) and fp64 values of every non-zero
// adjusted-cosine / Jaccard similarity; d_idx / d_sim hold U x U cells, d_cnt the list lengths
void launch_full_rows(const Train& tr, bool jaccard, int32_t* d_idx, double* d_sim, int32_t* d_cnt, hipStream_t st);
// fresh-closure similarity of one pair (owner order = u): writes *d_out
void launch_exact_pair(const Train& tr, int32_t u, int32_t v, double* d_out, hipStream_t st);

// ---- predict.hip: K7-K9 ----------------------------------------------------------------
// per test row: prediction of `predictor`; rows whose user is outside [own_lo, own_hi) are
// skipped (unknown users belong to shard 0).  d_abs_err[t] = |r - p| or 0 for skipped rows.
// id-sorted copies of the lists of the given users (d_row_user == nullptr: of every user)
void launch_sort_neighbors(NeighborTable& nt, int32_t n_rows, const int32_t* d_row_user, hipStream_t st);
void launch_predict(const Train& tr, NeighborTable* nt, int predictor, int64_t n,
                    const int32_t* d_du, const int32_t* d_di, const double* d_ratings,
                    const uint32_t* d_order, bool order_by_item, double* d_pred, double* d_abs_err, uint8_t* d_owned,
                    bool unknown_users_owned, hipStream_t st);
// (key, value) = (dense user / item, or `limit` (= their number) when absent from train; row) of every test row: sorted
// (bits_for(limit + 1) key bits), it is the d_order of the grouped kNN kernels
void launch_user_keys(int64_t n, const int32_t* d_du, uint32_t limit, uint64_t* d_key, uint32_t* d_val, hipStream_t st);
// sharded handles: key = d_src (dense user or item) for the rows of this shard's users, limit + 1 for everybody else's;
// *d_n_owned += number of rows of this shard
void launch_owned_keys(int64_t n, const int32_t* d_src, const int32_t* d_du, int32_t own_lo, int32_t own_hi, bool unknown_owned, uint32_t limit,
                       uint64_t* d_key, uint32_t* d_val, unsigned long long* d_n_owned, hipStream_t st);
// deterministic fixed-shape reduction: sum of d_abs_err and count of d_owned
void launch_reduce_err(const double* d_abs_err, const uint8_t* d_owned, int64_t n, double* d_partials,
                       int64_t* d_counts, int32_t n_blocks, hipStream_t st);

// ---- reco.hip: recommendations :651-674 -------------------------------------------------------------------------
// rows (user, every train item) for the prediction batch + the mask of the items the user rated
void launch_reco_rows(const Train& tr, int32_t user_raw, int32_t du, int32_t* d_users, int32_t* d_items, uint8_t* d_rated, hipStream_t st);
// v_b = dense items ordered by (prediction descending, raw id ascending), rated items last
void launch_reco_order(const Train& tr, SortWorkspace& ws, const double* d_pred, const uint8_t* d_rated, uint64_t* k_a, uint64_t* k_b,
                       uint32_t* v_a, uint32_t* v_b, hipStream_t st);
void launch_reco_take(const Train& tr, int32_t m, const uint32_t* d_order, const double* d_pred, int32_t* d_items, double* d_preds, hipStream_t st);

}  // namespace knncf
