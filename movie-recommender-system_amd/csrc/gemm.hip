// gemm.hip — K5: the user x user adjusted-cosine similarity as a blocked MFMA GEMM on gfx950.
//
//   S[M x N] (fp16 or fp32) = A[M x K] * B[N x K]^T,  A/B = rows of the preprocessed-rating matrix
//   (head columns only) rounded to fp16 (default) or bf16; both operands K-contiguous: the "NT"
//   form, ideal for LDS staging.
//
// The 16-bit result only has to be a FILTER: select.hip keeps every candidate within a rigorous
// error band of the k-th value and the fp64 re-rank decides.
//
// Kernel structure: BK = 64, each wave owns a (WM*32) x 64 sub-tile of v_mfma_f32_32x32x16_{f16,bf16}
// accumulators; 128x128 tile / 4 waves or 256x256 tile / 8 waves.  Operand tiles go
// HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a 2-deep
// ring; the LDS image is lane-linear as the DMA requires, with the bank swizzle applied on the
// SOURCE address (16-B chunk c of row r sits in slot c ^ ((r >> 1) & 7), conflict-free for
// ds_read_b128: the 16 lanes of a read group hit 16 distinct slots of the 256-B bank row).
// Tile t+1 stays in flight across the barrier behind a counted vmcnt.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "engine.h"

namespace knncf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- densify -----------------------------------------------------------------------------
template <bool F16>
__global__ void k_densify(const int64_t* __restrict__ u_ptr, const int32_t* __restrict__ s_col,
                          const double* __restrict__ s_pre, const int32_t* __restrict__ rows, int32_t row_begin,
                          int32_t n_rows, const int32_t* __restrict__ colmap, bf16_t* __restrict__ panel, int64_t ld, int ones) {
    // one wave per panel row
    int32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int32_t lane = threadIdx.x & 63;
    if (wave >= n_rows) return;
    int32_t u = rows ? rows[wave] : row_begin + wave;
    int64_t b = u_ptr[u], e = u_ptr[u + 1];
    bf16_t* out = panel + (int64_t)wave * ld;
    for (int64_t p = b + lane; p < e; p += 64) {
        int32_t c = colmap ? colmap[s_col[p]] : s_col[p];
        if (c >= 0) {
            const float x = ones ? 1.0f : (float)s_pre[p];  // (ones: the 0/1 panel of the Jaccard path)
            if (F16) reinterpret_cast<_Float16*>(out)[c] = (_Float16)x;
            else out[c] = (bf16_t)x;
        }
    }
}

// The same through LDS: a wave assembles its row there (zeros, then the row's head entries) and writes it out whole, in
// 16-byte pieces.  The scatter above writes 2 bytes at a time into a panel that a memset cleared first (8 M partial-sector
// stores): 0.55 ms at the ml-25m shape against 0.40 for this one — both measured beside the forked half of prep_commit, which
// is what they compete with (grids of 1024 .. 50 000 workgroups, 4 or 8 pieces per trip: all 0.37 - 0.43 ms).
template <bool F16>
__global__ void __launch_bounds__(256) k_densify_rows(const int64_t* __restrict__ u_ptr, const int32_t* __restrict__ s_col,
                                                      const double* __restrict__ s_pre, const int32_t* __restrict__ rows, int32_t row_begin,
                                                      int32_t n_rows, const int32_t* __restrict__ colmap, bf16_t* __restrict__ panel, int64_t ld,
                                                      int ones) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int32_t n_waves = (int32_t)((gridDim.x * blockDim.x) >> 6);
    bf16_t* my = reinterpret_cast<bf16_t*>(smem) + (size_t)(threadIdx.x >> 6) * ld;  // ld is a multiple of 64
    auto wave_sync = [] {  // (the waves of a block only synchronise with themselves)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    for (int32_t wave = (int32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); wave < n_rows; wave += n_waves) {
        for (int64_t c = lane; c < ld / 8; c += 64) reinterpret_cast<uint4*>(my)[c] = make_uint4(0u, 0u, 0u, 0u);
        wave_sync();
        const int32_t u = rows ? rows[wave] : row_begin + wave;
        const int64_t b = u_ptr[u], e = u_ptr[u + 1];
        // eight 64-entry pieces of the row at a time, every load of a level requested before the first is used
        for (int64_t p0 = b; p0 < e; p0 += 512) {
            int32_t it[8];
            double xv[8];
            int32_t cc[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int64_t p = p0 + 64 * q + lane;
                it[q] = p < e ? s_col[p] : -1;
                xv[q] = (p < e && !ones) ? s_pre[p] : 1.0;  // (ones: the 0/1 panel of the Jaccard path)
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) cc[q] = it[q] < 0 ? -1 : (colmap ? colmap[it[q]] : it[q]);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (cc[q] >= 0) {
                    if (F16) reinterpret_cast<_Float16*>(my)[cc[q]] = (_Float16)(float)xv[q];
                    else my[cc[q]] = (bf16_t)(float)xv[q];
                }
            }
        }
        wave_sync();
        uint4* out = reinterpret_cast<uint4*>(panel + (int64_t)wave * ld);
        for (int64_t c = lane; c < ld / 8; c += 64) out[c] = reinterpret_cast<const uint4*>(my)[c];
        wave_sync();
    }
}

// panel rows [0, n_rows) = the given users' head columns; rows [n_rows, panel_rows) = 0 (the padding)
void launch_densify(const Train& tr, const int32_t* d_rows, int32_t row_begin, int32_t n_rows,
                    const int32_t* d_colmap, bf16_t* panel, int64_t ld, int64_t panel_rows, bool fp16, hipStream_t st) {
    const size_t lds = (size_t)4 * (size_t)ld * sizeof(bf16_t);
    if (ld % 64 == 0 && lds <= 64 * 1024) {
        if (panel_rows > n_rows) KN_HIP(hipMemsetAsync(panel + (int64_t)n_rows * ld, 0, (size_t)(panel_rows - n_rows) * ld * sizeof(bf16_t), st));
        if (n_rows <= 0) return;
        const int blocks = (int)std::min<int64_t>(ceil_div((int64_t)n_rows * 64, 256), 256 * 16);  // (a resident grid walks the rows)
        if (fp16) k_densify_rows<true><<<blocks, 256, lds, st>>>(tr.u_ptr.p, tr.s_col.p, tr.s_pre.p, d_rows, row_begin, n_rows, d_colmap, panel, ld, tr.jaccard ? 1 : 0);
        else k_densify_rows<false><<<blocks, 256, lds, st>>>(tr.u_ptr.p, tr.s_col.p, tr.s_pre.p, d_rows, row_begin, n_rows, d_colmap, panel, ld, tr.jaccard ? 1 : 0);
        KN_HIP(hipGetLastError());
        return;
    }
    // wide panels (the all-dense formulation): scatter into a cleared panel
    KN_HIP(hipMemsetAsync(panel, 0, (size_t)panel_rows * ld * sizeof(bf16_t), st));
    if (n_rows <= 0) return;
    int blocks = (int)ceil_div((int64_t)n_rows * 64, 256);
    if (fp16) k_densify<true><<<blocks, 256, 0, st>>>(tr.u_ptr.p, tr.s_col.p, tr.s_pre.p, d_rows, row_begin, n_rows, d_colmap, panel, ld, tr.jaccard ? 1 : 0);
    else k_densify<false><<<blocks, 256, 0, st>>>(tr.u_ptr.p, tr.s_col.p, tr.s_pre.p, d_rows, row_begin, n_rows, d_colmap, panel, ld, tr.jaccard ? 1 : 0);
    KN_HIP(hipGetLastError());
}

// ---- hybrid similarity: column map of the dense head (the sparse tail is added in select.hip) --
__global__ void k_colmap(int32_t I, int32_t H, const int32_t* __restrict__ pop_item, int32_t* __restrict__ colmap) {
    int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= I) return;
    colmap[pop_item[j]] = j < H ? j : -1;
}

void launch_colmap(const Train& tr, int32_t H, int32_t* d_colmap, hipStream_t st) {
    k_colmap<<<(unsigned)ceil_div(tr.I, 256), 256, 0, st>>>(tr.I, H, tr.pop_item.p, d_colmap);
    KN_HIP(hipGetLastError());
}

// ---- GEMM --------------------------------------------------------------------------------
// Tile configurations (BK = 64 for both; a wave always owns a WM*32 x 64 sub-tile):
//   small: 128 x 128, 4 waves (2 x 2), WM = WN = 2, 64 KiB LDS, 2 workgroups per CU
//   large: 256 x 256, 8 waves (2 x 4), WM = 4, WN = 2, 132 KiB LDS, 1 workgroup per CU — half the L2 -> LDS operand
//          traffic per flop (at the MFMA peak the 128 x 128 tile asks the L2 for ~39 TB/s, more than it has)
static constexpr int BK = 64;

template <int ROWS>
struct TileGeom {
    static constexpr int TILE_BYTES = ROWS * BK * 2;       // one operand tile: ROWS x 64 16-bit elements
    static constexpr int STAGE_BYTES = 2 * TILE_BYTES;     // A + B
};

// issue this wave's share of one operand tile (ROWS rows x 64 elements) into LDS: 1-KiB pieces of 8 rows
template <int ROWS, int WAVES>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int64_t ld, char* lds_tile, int wave, int lane) {
    constexpr int LOADS = ROWS / 8 / WAVES;
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        int piece = wave * LOADS + j;
        int row = piece * 8 + (lane >> 3);
        int slot = lane & 7;
        int chunk = slot ^ ((row >> 1) & 7);         // swizzle on the source side
        const bf16_t* src = g + (int64_t)row * ld + chunk * 8;
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int row, int chunk) {
    int slot = chunk ^ ((row >> 1) & 7);
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + slot * 16);
}

template <bool F16>
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// LDS plan of the persistent kernel: [PAD][slot 0][slot 1][PAD]; a slot = one k-tile of A and of B.  The epilogue image
// of a tile reuses the slot that was consumed last (the other one already receives the next tile's first k-tile) and
// may spill into the PAD next to it.
template <class OT, int WM, int WAVES_M, int TB>
struct EpiGeom {
    static constexpr int ESZ = (int)sizeof(OT);
    static constexpr int IPP = ESZ == 2 ? WM : (WM >= 2 ? WM / 2 : 1);   // 32-row MFMA blocks of a wave per pass
    static constexpr int PASS_ROWS = IPP * 32;
    static constexpr int PASSES = WAVES_M * (WM / IPP);
    static constexpr int RS = TB * ESZ + 16;                             // LDS row stride of the image
    static constexpr int IMAGE_N = PASS_ROWS * RS;
    // symmetric launches also write the tile's mirror image: the pass's PASS_ROWS rows become PASS_ROWS columns of all TB
    // rows of the transposed tile.  Row stride 272 B for both element sizes: the two 32-lane halves of a wave (columns c
    // and c + 4 of the tile = rows 4 apart of this image) then sit 16 banks apart.
    static constexpr int RS_T = PASS_ROWS * ESZ + 16;
    static constexpr int IMAGE_T = TB * RS_T;
    static constexpr int IMAGE = IMAGE_N > IMAGE_T ? IMAGE_N : IMAGE_T;
    static constexpr int STAGE = TileGeom<TB>::STAGE_BYTES;
    static constexpr int PAD = IMAGE > STAGE ? ((IMAGE - STAGE + 1023) / 1024) * 1024 : 0;
    static constexpr int SMEM = 2 * STAGE + 2 * PAD;
};

// Persistent: one workgroup per CU slot walks its share of the tile sequence.  The k-tiles of consecutive tiles form
// ONE stream through the 2-slot ring, so the first k-tile of the next tile loads under the last MFMA step and the
// epilogue of this one, and the epilogue's global stores drain under the next tile's first MFMA step (measured with
// the stores / the MFMAs ablated: the per-tile kernel spent 10 of its 23 ms waiting for its own stores at K = 256).
// (Splitting the waves into loaders and storers, so that no counted load wait stands behind a store, was slower:
// 9.6 vs 6.9 ms per launch — four waves do not issue the LDS-DMA stream fast enough.)
// SYM (A == B, C square): only the tiles on and above the diagonal are computed, in the order of `tile_list` (tile row in
// the low 16 bits, tile column in the high 16); every off-diagonal tile is stored twice, as it is and mirrored — S is
// symmetric and its mirror is the same sum of the same products in the same order, so the stored values are bit for bit
// what the full-square launch stores, for half the MFMA work.
template <bool F16, class OT, int WM, int WN, int WAVES_M, int WAVES_N, bool SYM>
__global__ void __launch_bounds__(WAVES_M * WAVES_N * 64)
k_gemm_nt_bf16(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, OT* __restrict__ C, int tiles_m,
               int tiles_n, int k_tiles, int64_t lda, int64_t ldb, int64_t ldc, const uint32_t* __restrict__ tile_list, int n_listed,
               float clamp_hi) {
    constexpr int WAVES = WAVES_M * WAVES_N;
    constexpr int TBM = WAVES_M * WM * 32, TBN = WAVES_N * WN * 32;
    static_assert(TBM == TBN, "square block tiles: both operand tiles share one staging routine");
    typedef EpiGeom<OT, WM, WAVES_M, TBM> EG;
    constexpr int TILE_BYTES = TileGeom<TBM>::TILE_BYTES, STAGE_BYTES = TileGeom<TBM>::STAGE_BYTES;
    constexpr int LOADS_PER_STAGE = 2 * (TBM / 8 / WAVES);  // LDS-DMA instructions per wave per stage
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* const ring = lds + EG::PAD;

    // XCD-aware tile order: blocks that share an XCD (equal blockIdx % 8, observed round-robin placement; speed
    // only) own one contiguous range of the tile sequence and walk it side by side, so that A/B panels are reused
    // out of that XCD's L2.  Every tile is visited exactly once for any grid size.
    const int nwg = SYM ? n_listed : tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
    const int xcd_first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int xcd_end = xcd_first + (xcd < r ? q + 1 : q);
    const int stride = ((int)gridDim.x - xcd + 7) >> 3;        // blocks of this XCD
    int t = xcd_first + ((int)blockIdx.x >> 3);
    if (t >= xcd_end) return;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int frow = lane & 31;  // fragment row inside a 32-row MFMA tile
    const int fhalf = lane >> 5; // which 8-wide k half of the 16-wide k-step

    // grouped ordering: 8 tile-rows at a time, column-major inside a group
    auto tile_of = [&](int wg, int& tm, int& tn) {
        if (SYM) {
            const uint32_t packed = tile_list[wg];
            tm = (int)(packed & 0xffffu);
            tn = (int)(packed >> 16);
            return;
        }
        const int GROUP = 8;
        const int group_sz = GROUP * tiles_n;
        const int gid = wg / group_sz;
        const int first_m = gid * GROUP;
        const int gm = min(GROUP, tiles_m - first_m);
        tm = first_m + (wg % group_sz) % gm;
        tn = (wg % group_sz) / gm;
    };

    int tm, tn;
    tile_of(t, tm, tn);
    const bf16_t* Ag = A + (int64_t)tm * TBM * lda;
    const bf16_t* Bg = B + (int64_t)tn * TBN * ldb;
    stage_tile<TBM, WAVES>(Ag, lda, ring, wave, lane);
    stage_tile<TBN, WAVES>(Bg, ldb, ring + TILE_BYTES, wave, lane);
    int slot = 0;
    bool landed = false;  // the current slot's k-tile was already waited for (at the previous tile's epilogue)

    for (;;) {
        const int t_next = t + stride;
        const bool more = t_next < xcd_end;
        int tm2 = 0, tn2 = 0;
        if (more) tile_of(t_next, tm2, tn2);
        const bf16_t* Ag2 = A + (int64_t)tm2 * TBM * lda;
        const bf16_t* Bg2 = B + (int64_t)tn2 * TBN * ldb;

        f32x16 acc[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        for (int kt = 0; kt < k_tiles; ++kt) {
            char* cur = ring + slot * STAGE_BYTES;
            char* nxt = ring + (slot ^ 1) * STAGE_BYTES;
            const bool in_tile = kt + 1 < k_tiles;
            if (in_tile) {
                stage_tile<TBM, WAVES>(Ag + (int64_t)(kt + 1) * BK, lda, nxt, wave, lane);
                stage_tile<TBN, WAVES>(Bg + (int64_t)(kt + 1) * BK, ldb, nxt + TILE_BYTES, wave, lane);
            } else if (more) {
                stage_tile<TBM, WAVES>(Ag2, lda, nxt, wave, lane);
                stage_tile<TBN, WAVES>(Bg2, ldb, nxt + TILE_BYTES, wave, lane);
            }
            if (!landed) {
                // k-tile `cur` landed; the one just issued (LOADS_PER_STAGE DMAs of this wave) stays in flight across
                // the barrier.  (vmcnt also counts the previous tile's stores, all older than the DMAs waited for.)
                static_assert(LOADS_PER_STAGE == 4 || LOADS_PER_STAGE == 8 || LOADS_PER_STAGE == 16, "the counted vmcnt below assumes 4, 8 or 16 LDS-DMA instructions per wave per stage");
                if (in_tile || more) {
                    if (LOADS_PER_STAGE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (LOADS_PER_STAGE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            landed = false;
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                bf16x8 a[WM];
#pragma unroll
                for (int i = 0; i < WM; ++i) a[i] = read_frag(cur, wr * (WM * 32) + i * 32 + frow, ks * 2 + fhalf);
                bf16x8 b[WN];
#pragma unroll
                for (int j = 0; j < WN; ++j) b[j] = read_frag(cur + TILE_BYTES, wc * (WN * 32) + j * 32 + frow, ks * 2 + fhalf);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        // operands swapped (D' = B A^T): a lane then holds ONE row of S and groups of 4 consecutive columns
                        acc[i][j] = mfma<F16>(b[j], a[i], acc[i][j]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // everyone is done reading `cur` before it is restaged
            asm volatile("" ::: "memory");
            slot ^= 1;
        }

        // ---- epilogue: through LDS, so that C leaves in full 16-byte pieces of contiguous rows ----------------
        // D layout of v_mfma_f32_32x32x16 with the operands swapped: lane (frow, fhalf) holds row frow of the 32 x 32
        // block, register e holds column (e & 3) + 8 (e >> 2) + 4 fhalf: four groups of 4 consecutive columns.  Each
        // group is converted and written to an LDS image of PASS_ROWS rows of the tile ([row][col], rows padded by
        // 16 B); then every thread copies 16-byte pieces of rows to global memory (one wave instruction = 1 KiB of
        // contiguous C).  Storing the accumulators directly costs one 2-byte store per element: 1024 wave-wide store
        // instructions per 256 x 256 tile, as long as the tile's MFMA work.
        {
            // the next tile's first k-tile (issued one MFMA step ago) lands before any store is queued behind it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            landed = more;
            constexpr int ESZ = EG::ESZ, RS = EG::RS, IPP = EG::IPP, PASS_ROWS = EG::PASS_ROWS, PASSES = EG::PASSES;
            constexpr int CH = TBN * ESZ / 16;                              // 16-byte pieces per row
            // the consumed slot is slot ^ 1; its image may extend into the PAD on its outer side
            char* const img = (slot ^ 1) ? ring + STAGE_BYTES : lds;
#pragma unroll
            for (int pass = 0; pass < PASSES; ++pass) {
                const int wrp = pass / (WM / IPP), i0 = (pass % (WM / IPP)) * IPP;
                if (wr == wrp) {
#pragma unroll
                    for (int ii = 0; ii < IPP; ++ii)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int i = i0 + ii;
                                char* dst = img + (ii * 32 + frow) * RS + (wc * (WN * 32) + j * 32 + 8 * g + 4 * fhalf) * ESZ;
                                if (ESZ == 2) {
                                    typedef __attribute__((ext_vector_type(4))) _Float16 h4;
                                    h4 v;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = (_Float16)fminf(fmaxf(acc[i][j][4 * g + e], -clamp_hi), clamp_hi);  // (1.0: see api.cpp eps_rest; 65504 for counts)
                                    *reinterpret_cast<h4*>(dst) = v;
                                } else {
                                    *reinterpret_cast<float4*>(dst) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                                }
                            }
                }
                __syncthreads();
                OT* Cg = C + ((int64_t)tm * TBM + wrp * (WM * 32) + i0 * 32) * ldc + (int64_t)tn * TBN;
                for (int idx = threadIdx.x; idx < PASS_ROWS * CH; idx += WAVES * 64) {
                    const int row = idx / CH, ch = idx - row * CH;
                    const uint4 v = *reinterpret_cast<const uint4*>(img + row * RS + ch * 16);
                    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
                    // C is written once and read back only after the whole panel: keep it from displacing A/B in the L2
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(reinterpret_cast<char*>(Cg + (int64_t)row * ldc) + ch * 16));
                }
                __syncthreads();  // the image is rewritten by the next pass / restaged by the next tile
                if (SYM && tm != tn) {
                    // the mirror image of the same rows: element (row, col) of the tile goes to [col][row - pass rows' first];
                    // one element per LDS write (a lane holds one row and runs of 4 columns = 4 rows of this image), then the
                    // same 16-byte copy-out, into the tile (tn, tm) of C
                    constexpr int RS_T = EG::RS_T;
                    constexpr int CHT = PASS_ROWS * ESZ / 16;                  // 16-byte pieces per row of the mirror image
                    if (wr == wrp) {
#pragma unroll
                        for (int ii = 0; ii < IPP; ++ii)
#pragma unroll
                            for (int j = 0; j < WN; ++j)
#pragma unroll
                                for (int g = 0; g < 4; ++g)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const int i = i0 + ii;
                                        const int col = wc * (WN * 32) + j * 32 + 8 * g + 4 * fhalf + e;
                                        OT* dst = reinterpret_cast<OT*>(img + col * RS_T) + (ii * 32 + frow);
                                        if (ESZ == 2) *dst = (OT)fminf(fmaxf(acc[i][j][4 * g + e], -clamp_hi), clamp_hi);
                                        else *dst = (OT)acc[i][j][4 * g + e];
                                    }
                    }
                    __syncthreads();
                    OT* Ct = C + (int64_t)tn * TBN * ldc + (int64_t)tm * TBM + wrp * (WM * 32) + i0 * 32;
                    for (int idx = threadIdx.x; idx < TBN * CHT; idx += WAVES * 64) {
                        const int row = idx / CHT, ch = idx - row * CHT;
                        const uint4 v = *reinterpret_cast<const uint4*>(img + row * RS_T + ch * 16);
                        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
                        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(reinterpret_cast<char*>(Ct + (int64_t)row * ldc) + ch * 16));
                    }
                    __syncthreads();
                }
            }
        }
        if (!more) break;
        t = t_next;
        tm = tm2;
        tn = tn2;
        Ag = Ag2;
        Bg = Bg2;
    }
}

// a pointer whose value is the same in every lane, moved to scalar registers (buffer descriptors must be uniform)
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

// stage_tile with the addressing off the vector registers: the tile origin is a uniform 64-bit pointer (scalar registers),
// the lane part of the source offset is ONE kernel-invariant 32-bit register per operand — voff = (lane >> 3) * ld * 2 +
// (((lane & 7) ^ (lane >> 4)) * 16): row and swizzled chunk of an EVEN 8-row piece; an odd piece's chunk differs in bit 2
// ((row >> 1) & 7 gains 4), i.e. voff ^ 64 (ld * 2 is a multiple of 128) — and the piece's row block goes into the
// scalar base.
template <int ROWS, int WAVES>
__device__ __forceinline__ void stage_tile_u(const bf16_t* g_uniform, uint32_t ld2, char* lds_tile, int wave, uint32_t voff) {
    constexpr int LOADS = ROWS / 8 / WAVES;
    static_assert(LOADS % 2 == 0, "pieces alternate between even and odd");
    const char* base = reinterpret_cast<const char*>(uniform_ptr(g_uniform));
#ifdef KNNCF_OV_ABL_NOLOAD  /* timing-only ablation (results are wrong): no operand loads */
    return;
#endif
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int piece = wave * LOADS + j;  // (wave * LOADS is even: parity of the piece == parity of j)
        const char* src = base + (uint64_t)((uint32_t)piece * 8u * ld2) + (uint64_t)((j & 1) ? (voff ^ 64u) : voff);
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const bf16_t*>(src), (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024), 16, 0, 0);
    }
}

#ifdef KNNCF_OV_ABL_NOSTORE  /* timing-only ablation (results are wrong): the overlapped GEMM without its global stores */
#define KN_OV_STORE "s_nop 0 ; "
#define KN_OV_STORES 0
#else
#define KN_OV_STORE
#define KN_OV_STORES 4
#endif
#define KN_OV_POLICY "nt"  // (measured at H = 384: nt 14.3 ms per launch, plain stores and sc1 17.9 ms)
#ifndef KN_OV_POLICY
#define KN_OV_POLICY "nt"  // (A/B switch: -DKN_OV_POLICY='""' plain, '"sc1"' write-through)
#endif
// ---- the overlapped form (fp16 panel, 256 x 256 tiles) ---------------------------------------------------------------
// k_gemm_nt_bf16 above ADDS a tile's stores to its MFMA work: every store of the epilogue is queued in front of the next
// tile's operand loads, and on gfx9 vmcnt counts loads and stores together and retires them in order, so the counted wait
// for the second k-tile of the next tile also waits for every store of this one.  At K = 384 that is ~5 us of MFMA steps
// plus ~10 us of stores (256 KB per off-diagonal tile at the CU's share of the HBM write rate) per tile.
//
// Here the finished tile does not leave through a workgroup-wide epilogue at all.  Its accumulators are converted to
// fp16 in place (128 -> 64 registers per lane: two waves per SIMD hold 256 registers each, so the next tile's 128
// accumulators fit beside them) and the wave drains them, EIGHT units to a tile, while it computes the next one: two
// units per k-step, each through the wave's PRIVATE 4 KiB of LDS (no workgroup barrier: a wave's LDS operations execute in
// order) —
//   unit 2 i     : the wave's 32 x 64 block i as it is: 8 ds_write_b64 (a lane holds one row and runs of 4 columns), read
//                  back as 16-byte pieces, 4 global stores of 8 rows x 128 contiguous bytes;
//   unit 2 i + 1 : the same block MIRRORED (symmetric launches, off-diagonal tiles): 32 ds_write_b16 into the 64 x 32
//                  transposed image, 4 global stores of 16 rows x 64 contiguous bytes, into tile (tn, tm).
// The stores are spread evenly over the k-steps and the counted waits allow the stores of the previous k-step to stay
// in flight (vmcnt(loads of the next stage + stores issued since)): the launch takes max(MFMA, stores) per tile instead
// of their sum.  LDS: the 128 KiB operand ring + 8 x 4 KiB = all 160 KiB.  The stored values are bit for bit those of
// k_gemm_nt_bf16 (same MFMA order, same clamp, same round-to-nearest-even conversion).
template <bool F16, bool SYM, bool SPREAD>
__global__ void __launch_bounds__(512)
k_gemm_nt_ov(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, _Float16* __restrict__ C, int tiles_m, int tiles_n,
             int k_tiles, int64_t lda, int64_t ldb, int64_t ldc, const uint32_t* __restrict__ tile_list, int n_listed, float clamp_hi) {
    constexpr int WM = 4, WN = 2, WAVES_N = 4, WAVES = 8, TB = 256;
    constexpr int TILE_BYTES = TileGeom<TB>::TILE_BYTES, STAGE_BYTES = TileGeom<TB>::STAGE_BYTES;
    constexpr int LOADS_PER_STAGE = 2 * (TB / 8 / WAVES);  // 8 LDS-DMA instructions per wave per stage
    static_assert(LOADS_PER_STAGE == 8, "the counted waits below assume 8 LDS-DMA instructions per wave per stage");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* const ring = lds;

    const int nwg = SYM ? n_listed : tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
    const int xcd_first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int xcd_end = xcd_first + (xcd < r ? q + 1 : q);
    const int stride = ((int)gridDim.x - xcd + 7) >> 3;
    int t = xcd_first + ((int)blockIdx.x >> 3);
    if (t >= xcd_end) return;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int frow = lane & 31, fhalf = lane >> 5;
    char* const stg = lds + 2 * STAGE_BYTES + wave * 4096;  // this wave's private staging image
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lda2 = (uint32_t)lda * 2u, ldb2 = (uint32_t)ldb * 2u;
    const uint32_t sw16 = (uint32_t)(((lane & 7) ^ (lane >> 4)) * 16);
    uint32_t voff_a = (uint32_t)(lane >> 3) * lda2 + sw16, voff_b = (uint32_t)(lane >> 3) * ldb2 + sw16;
    // (opaque to the optimizer: it otherwise folds the lane part back into the tile origin — (tile row + lane row) * ld as a
    // 64-bit product per lane and piece — and keeps twelve 64-bit addresses per tile in vector registers)
    asm volatile("" : "+v"(voff_a), "+v"(voff_b));

    auto tile_of = [&](int wg, int& tm, int& tn) {
        if (SYM) {
            const uint32_t packed = tile_list[wg];
            tm = (int)(packed & 0xffffu);
            tn = (int)(packed >> 16);
            return;
        }
        const int GROUP = 8;
        const int group_sz = GROUP * tiles_n;
        const int gid = wg / group_sz;
        const int first_m = gid * GROUP;
        const int gm = min(GROUP, tiles_m - first_m);
        tm = first_m + (wg % group_sz) % gm;
        tn = (wg % group_sz) / gm;
    };

    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    // the previous tile, converted: P[i][j][2 g + h] = columns 8 g + 4 fhalf + 2 h, + 1 of block (i, j), row frow
    // the wave's row blocks 2 and 3 wait in P (32 registers) and leave under the next tile; blocks 0 and 1 leave at the tile's end
    constexpr int NP = 2, UNITS = 4;
    uint32_t P[NP][WN][8];
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int h = 0; h < 8; ++h) P[i][j][h] = 0u;
    // Register plan (2 waves per SIMD: 256 registers each, one unified file on gfx950): the 128 accumulators of the running
    // tile in the ACCUMULATION half, P + one fragment set + the addressing in the 128 architectural registers.  hipcc puts
    // MFMA results into accumulation registers only when the kernel is known to use them; this (empty) statement says so.
    int ptm = 0, ptn = 0;
    int unit = 1000;  // next unit of the previous tile to drain (>= UNITS: none left)

    // one unit of the previous tile through the wave's private image; returns the number of global stores it issued.
    // The addressing is kept off the vector registers (P + one fragment set + addresses share 128): the global stores go
    // through a buffer descriptor rooted at the unit's first element (scalar registers), the lane part of the offset is one
    // kernel-invariant register per orientation, the row step rides in the scalar offset; the LDS addresses are one
    // invariant register per access pattern plus immediates (the LDS stores are written as asm for that reason).
    const uint32_t ldc2 = (uint32_t)ldc * 2u;                                              // bytes per row of C
    const uint32_t vo_n = (uint32_t)(lane >> 3) * ldc2 + (uint32_t)(lane & 7) * 16u;       // both orientations: 8 rows x 128 B per store
    const uint32_t stg_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)stg;  // LDS byte address (4 KiB aligned)
    // normal image: [32 rows][128 B], 16-byte slot s of row r at slot s ^ (r & 7); a lane writes 8 bytes of row frow
    const uint32_t aw_n = stg_a + ((uint32_t)frow * 128u | (uint32_t)fhalf * 8u | (uint32_t)(frow & 7) * 16u);  // ^ (slot * 16)
    const uint32_t lr_n = (uint32_t)(lane >> 3) * 128u + (uint32_t)(((lane & 7) ^ (lane >> 3)) * 16);
    // A unit is ONE asm statement: 16 data registers in, 17 temporaries, the addressing in kernel-invariant registers — the
    // compiler's version kept 64-bit per-lane addresses and spilled (a scratch reload is a vector-memory operation: its
    // wait drains every store in flight, the opposite of the point).  A wave's LDS operations execute in order, so the reads
    // see the writes without a wait; every buffer store waits for its own read only (lgkmcnt counts down as the four reads
    // return in order); s_nop 1: a store of more than 8 bytes needs a wait state before its data registers may be rewritten.
    // Both orientations leave as 4 stores of 8 rows x 128 contiguous bytes (scripts/microbench/store_pattern.hip: 5.4 TB/s
    // for that shape against 6.1 linear — and 3.1 for 64-byte pieces, which is why the mirror unit spans TWO row blocks):
    //   normal(I)    : the wave's 32 x 64 block I as it is — image [32 rows][128 B], 16-byte slot s of row r at s ^ (r & 7);
    //   mirror(p, J) : rows 64 p .. + 64 of its 32-column sub-block J, transposed — image [32 rows = columns][128 B = 64 rows].
    const uint32_t lra_n = stg_a + lr_n, lra_t = stg_a + (uint32_t)lane * 16u;
    const uint32_t aw_m = stg_a + (uint32_t)fhalf * 512u + (uint32_t)frow * 2u;
    const uint32_t so_1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(8u * ldc2));
    auto drain_normal = [&](const uint32_t (&R)[WN][8], int I) -> int {
        u32x4 q0, q1, q2, q3;
        uint32_t tmp_a;
        _Float16* Cg = C + ((int64_t)ptm * TB + wr * (WM * 32) + I * 32) * ldc + (int64_t)ptn * TB + wc * (WN * 32);
        const uint64_t cb = reinterpret_cast<uint64_t>(uniform_ptr(Cg));
        const u32x4 rs = {(uint32_t)cb, (uint32_t)(cb >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
        asm volatile(
                "v_xor_b32 %[t], 0, %[aw]\n\tds_write2_b32 %[t], %[r00], %[r01] offset1:1\n\t"
                "v_xor_b32 %[t], 16, %[aw]\n\tds_write2_b32 %[t], %[r02], %[r03] offset1:1\n\t"
                "v_xor_b32 %[t], 32, %[aw]\n\tds_write2_b32 %[t], %[r04], %[r05] offset1:1\n\t"
                "v_xor_b32 %[t], 48, %[aw]\n\tds_write2_b32 %[t], %[r06], %[r07] offset1:1\n\t"
                "v_xor_b32 %[t], 64, %[aw]\n\tds_write2_b32 %[t], %[r10], %[r11] offset1:1\n\t"
                "v_xor_b32 %[t], 80, %[aw]\n\tds_write2_b32 %[t], %[r12], %[r13] offset1:1\n\t"
                "v_xor_b32 %[t], 96, %[aw]\n\tds_write2_b32 %[t], %[r14], %[r15] offset1:1\n\t"
                "v_xor_b32 %[t], 112, %[aw]\n\tds_write2_b32 %[t], %[r16], %[r17] offset1:1\n\t"
                "ds_read_b128 %[q0], %[lr] offset:0\n\t"
                "ds_read_b128 %[q1], %[lr] offset:1024\n\t"
                "ds_read_b128 %[q2], %[lr] offset:2048\n\t"
                "ds_read_b128 %[q3], %[lr] offset:3072\n\t"
                "s_waitcnt lgkmcnt(3)\n\t" KN_OV_STORE "buffer_store_dwordx4 %[q0], %[vo], %[rs], %[so0] offen " KN_OV_POLICY "\n\t"
                "s_waitcnt lgkmcnt(2)\n\t" KN_OV_STORE "buffer_store_dwordx4 %[q1], %[vo], %[rs], %[so1] offen " KN_OV_POLICY "\n\t"
                "s_waitcnt lgkmcnt(1)\n\t" KN_OV_STORE "buffer_store_dwordx4 %[q2], %[vo], %[rs], %[so2] offen " KN_OV_POLICY "\n\t"
                "s_waitcnt lgkmcnt(0)\n\t" KN_OV_STORE "buffer_store_dwordx4 %[q3], %[vo], %[rs], %[so3] offen " KN_OV_POLICY "\n\t"
                "s_nop 1"
            : [t] "=&v"(tmp_a), [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3)
            : [aw] "v"(aw_n), [lr] "v"(lra_n), [vo] "v"(vo_n), [rs] "s"(rs), [so0] "s"(0u), [so1] "s"(so_1), [so2] "s"(2u * so_1), [so3] "s"(3u * so_1),
              [r00] "v"(R[0][0]), [r01] "v"(R[0][1]), [r02] "v"(R[0][2]), [r03] "v"(R[0][3]), [r04] "v"(R[0][4]), [r05] "v"(R[0][5]), [r06] "v"(R[0][6]), [r07] "v"(R[0][7]), [r10] "v"(R[1][0]), [r11] "v"(R[1][1]), [r12] "v"(R[1][2]), [r13] "v"(R[1][3]), [r14] "v"(R[1][4]), [r15] "v"(R[1][5]), [r16] "v"(R[1][6]), [r17] "v"(R[1][7])
            : "memory");
        return KN_OV_STORES;
    };
    auto drain_mirror = [&](const uint32_t (&Ra)[WN][8], const uint32_t (&Rb)[WN][8], int pr, auto J_) -> int {
        constexpr int J = decltype(J_)::value;
        if (!SYM || ptm == ptn) return 0;
        u32x4 q0, q1, q2, q3;
        _Float16* Ct = C + ((int64_t)ptn * TB + wc * (WN * 32) + J * 32) * ldc + (int64_t)ptm * TB + wr * (WM * 32) + pr * 64;
        const uint64_t cb = reinterpret_cast<uint64_t>(uniform_ptr(Ct));
        const u32x4 rs = {(uint32_t)cb, (uint32_t)(cb >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
        asm volatile(
                "ds_write_b16 %[aw], %[r00] offset:0\n\tds_write_b16_d16_hi %[aw], %[r00] offset:128\n\t"
                "ds_write_b16 %[aw], %[r01] offset:256\n\tds_write_b16_d16_hi %[aw], %[r01] offset:384\n\t"
                "ds_write_b16 %[aw], %[r02] offset:1024\n\tds_write_b16_d16_hi %[aw], %[r02] offset:1152\n\t"
                "ds_write_b16 %[aw], %[r03] offset:1280\n\tds_write_b16_d16_hi %[aw], %[r03] offset:1408\n\t"
                "ds_write_b16 %[aw], %[r04] offset:2048\n\tds_write_b16_d16_hi %[aw], %[r04] offset:2176\n\t"
                "ds_write_b16 %[aw], %[r05] offset:2304\n\tds_write_b16_d16_hi %[aw], %[r05] offset:2432\n\t"
                "ds_write_b16 %[aw], %[r06] offset:3072\n\tds_write_b16_d16_hi %[aw], %[r06] offset:3200\n\t"
                "ds_write_b16 %[aw], %[r07] offset:3328\n\tds_write_b16_d16_hi %[aw], %[r07] offset:3456\n\t"
                "ds_write_b16 %[aw], %[r10] offset:64\n\tds_write_b16_d16_hi %[aw], %[r10] offset:192\n\t"
                "ds_write_b16 %[aw], %[r11] offset:320\n\tds_write_b16_d16_hi %[aw], %[r11] offset:448\n\t"
                "ds_write_b16 %[aw], %[r12] offset:1088\n\tds_write_b16_d16_hi %[aw], %[r12] offset:1216\n\t"
                "ds_write_b16 %[aw], %[r13] offset:1344\n\tds_write_b16_d16_hi %[aw], %[r13] offset:1472\n\t"
                "ds_write_b16 %[aw], %[r14] offset:2112\n\tds_write_b16_d16_hi %[aw], %[r14] offset:2240\n\t"
                "ds_write_b16 %[aw], %[r15] offset:2368\n\tds_write_b16_d16_hi %[aw], %[r15] offset:2496\n\t"
                "ds_write_b16 %[aw], %[r16] offset:3136\n\tds_write_b16_d16_hi %[aw], %[r16] offset:3264\n\t"
                "ds_write_b16 %[aw], %[r17] offset:3392\n\tds_write_b16_d16_hi %[aw], %[r17] offset:3520\n\t"
                "ds_read_b128 %[q0], %[lr] offset:0\n\t"
                "ds_read_b128 %[q1], %[lr] offset:1024\n\t"
                "ds_read_b128 %[q2], %[lr] offset:2048\n\t"
                "ds_read_b128 %[q3], %[lr] offset:3072\n\t"
                "s_waitcnt lgkmcnt(3)\n\t" KN_OV_STORE "buffer_store_dwordx4 %[q0], %[vo], %[rs], %[so0] offen " KN_OV_POLICY "\n\t"
                "s_waitcnt lgkmcnt(2)\n\t" KN_OV_STORE "buffer_store_dwordx4 %[q1], %[vo], %[rs], %[so1] offen " KN_OV_POLICY "\n\t"
                "s_waitcnt lgkmcnt(1)\n\t" KN_OV_STORE "buffer_store_dwordx4 %[q2], %[vo], %[rs], %[so2] offen " KN_OV_POLICY "\n\t"
                "s_waitcnt lgkmcnt(0)\n\t" KN_OV_STORE "buffer_store_dwordx4 %[q3], %[vo], %[rs], %[so3] offen " KN_OV_POLICY "\n\t"
                "s_nop 1"
            : [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3)
            : [aw] "v"(aw_m), [lr] "v"(lra_t), [vo] "v"(vo_n), [rs] "s"(rs), [so0] "s"(0u), [so1] "s"(so_1), [so2] "s"(2u * so_1), [so3] "s"(3u * so_1),
              [r00] "v"(Ra[J][0]), [r01] "v"(Ra[J][1]), [r02] "v"(Ra[J][2]), [r03] "v"(Ra[J][3]), [r04] "v"(Ra[J][4]), [r05] "v"(Ra[J][5]), [r06] "v"(Ra[J][6]), [r07] "v"(Ra[J][7]), [r10] "v"(Rb[J][0]), [r11] "v"(Rb[J][1]), [r12] "v"(Rb[J][2]), [r13] "v"(Rb[J][3]), [r14] "v"(Rb[J][4]), [r15] "v"(Rb[J][5]), [r16] "v"(Rb[J][6]), [r17] "v"(Rb[J][7])
            : "memory");
        return KN_OV_STORES;
    };
    // the four units of the row blocks kept in P (blocks 2 and 3 = row pair 1): two per k-step
    auto drain_unit = [&](int u) -> int {
        switch (u) {
            case 0: return drain_normal(P[0], 2);
            case 1: return drain_normal(P[1], 3);
            case 2: return drain_mirror(P[0], P[1], 1, std::integral_constant<int, 0>{});
            default: return drain_mirror(P[0], P[1], 1, std::integral_constant<int, 1>{});
        }
    };

#ifdef KNNCF_OV_ABL_HOTLOAD  /* timing-only ablation (results are wrong): every tile's operands are the panel's first rows (L2-hot) */
#define KN_OV_ROW(x) 0
#else
#define KN_OV_ROW(x) (x)
#endif
    int tm, tn;
    tile_of(t, tm, tn);
    const bf16_t* Ag = A + (int64_t)KN_OV_ROW(tm) * TB * lda;
    const bf16_t* Bg = B + (int64_t)KN_OV_ROW(tn) * TB * ldb;
    stage_tile_u<TB, WAVES>(Ag, lda2, ring, uwave, voff_a);
    stage_tile_u<TB, WAVES>(Bg, ldb2, ring + TILE_BYTES, uwave, voff_b);
    int slot = 0;
    // Counted waits by bookkeeping instead of by position: `ops` counts every vector-memory operation this wave has issued
    // (8 LDS-DMAs per operand stage, 4 stores per unit), mark[s] is its value right after the stage into slot s was issued —
    // so when k-tile `slot` is needed, exactly ops - mark[slot] younger operations may still be in flight (vmcnt retires
    // in order: loads, LDS-DMAs and stores together).  A smaller immediate only waits longer; the immediates are multiples
    // of 4 up to 32.
    uint32_t ops = LOADS_PER_STAGE, mark[2] = {LOADS_PER_STAGE, 0u};
    auto wait_for = [&](int s) {
        const uint32_t allow = ops - mark[s];
        if (allow >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (allow >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
        else if (allow >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (allow >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (allow >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (allow >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (allow >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (allow >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    bool pre_issued = false;  // the second k-tile of this tile was requested at the end of the previous one
    // Where the four deferred units go: with six or more k-steps one per k-step from the third on — the 16 stores of the
    // tile's end then have two k-steps of their own to drain — otherwise two per k-step from the first.
    constexpr bool spread = SPREAD;  // (the launcher: k_tiles >= 6)

    for (;;) {
        const int t_next = t + stride;
        const bool more = t_next < xcd_end;
        int tm2 = 0, tn2 = 0;
        if (more) tile_of(t_next, tm2, tn2);
        const bf16_t* Ag2 = A + (int64_t)KN_OV_ROW(tm2) * TB * lda;
        const bf16_t* Bg2 = B + (int64_t)KN_OV_ROW(tn2) * TB * ldb;

        f32x16 acc[WM][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        for (int kt = 0; kt < k_tiles; ++kt) {
            char* cur = ring + slot * STAGE_BYTES;
            char* nxt = ring + (slot ^ 1) * STAGE_BYTES;
            const bool in_tile = kt + 1 < k_tiles;
            if (kt == 0 && pre_issued) {
                pre_issued = false;  // (k-tile 1 is already on its way into `nxt`)
            } else if (in_tile) {
                stage_tile_u<TB, WAVES>(Ag + (int64_t)(kt + 1) * BK, lda2, nxt, uwave, voff_a);
                stage_tile_u<TB, WAVES>(Bg + (int64_t)(kt + 1) * BK, ldb2, nxt + TILE_BYTES, uwave, voff_b);
                ops += LOADS_PER_STAGE;
                mark[slot ^ 1] = ops;
            } else if (more) {
                stage_tile_u<TB, WAVES>(Ag2, lda2, nxt, uwave, voff_a);
                stage_tile_u<TB, WAVES>(Bg2, ldb2, nxt + TILE_BYTES, uwave, voff_b);
                ops += LOADS_PER_STAGE;
                mark[slot ^ 1] = ops;
            }
            wait_for(slot);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // Fragments: one A fragment ahead of the MFMA pair that uses the current one, the B pair of the next k-substep
            // behind the last A fragment of this one — 24 registers.
            {
                const char* ca = cur;
                const char* cb = cur + TILE_BYTES;
                const int arow = wr * (WM * 32) + frow, brow = wc * (WN * 32) + frow;
                bf16x8 b0 = read_frag(cb, brow, fhalf), b1 = read_frag(cb, brow + 32, fhalf);
                bf16x8 a_cur = read_frag(ca, arow, fhalf);
                const bool unit_step = !spread || kt >= 2;
#pragma unroll
                for (int ks = 0; ks < BK / 16; ++ks) {
                    bf16x8 b0n = b0, b1n = b1;
#pragma unroll
                    for (int i = 0; i < WM; ++i) {
                        bf16x8 a_next = a_cur;
                        if (i + 1 < WM) {
                            a_next = read_frag(ca, arow + (i + 1) * 32, ks * 2 + fhalf);
                        } else if (ks + 1 < BK / 16) {
                            a_next = read_frag(ca, arow, (ks + 1) * 2 + fhalf);
                            b0n = read_frag(cb, brow, (ks + 1) * 2 + fhalf);
                            b1n = read_frag(cb, brow + 32, (ks + 1) * 2 + fhalf);
                        }
#ifndef KNNCF_OV_ABL_NOMFMA  /* timing-only ablation (results are wrong): no MFMA work */
                        acc[i][0] = mfma<F16>(b0, a_cur, acc[i][0]);
                        acc[i][1] = mfma<F16>(b1, a_cur, acc[i][1]);
#else
                        acc[i][0][0] += (float)a_cur[0] + (float)b0[0];
                        acc[i][1][0] += (float)a_cur[1] + (float)b1[0];
#endif
                        a_cur = a_next;
                    }
                    b0 = b0n;
                    b1 = b1n;
                    // units of the previous tile, behind the first (and, when they come two to a k-step, the third) MFMA
                    // group; the next k-substep's fragments are already requested
                    if ((ks == 0 || (ks == 2 && !spread)) && unit_step && unit < UNITS) {
                        ops += (uint32_t)drain_unit(unit);
                        ++unit;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // everyone is done reading `cur` before it is restaged
            asm volatile("" ::: "memory");
            slot ^= 1;
        }
        // The slot just consumed is free: the next tile's SECOND k-tile is requested before this tile's end stores are
        // queued, so that they stand behind it in the in-order count and the next tile's first two k-steps do not wait for them.
        if (more && k_tiles >= 2) {
            char* nxt = ring + (slot ^ 1) * STAGE_BYTES;
            stage_tile_u<TB, WAVES>(Ag2 + BK, lda2, nxt, uwave, voff_a);
            stage_tile_u<TB, WAVES>(Bg2 + BK, ldb2, nxt + TILE_BYTES, uwave, voff_b);
            ops += LOADS_PER_STAGE;
            mark[slot ^ 1] = ops;
            pre_issued = true;
        }
        // short K (fewer k-steps than units): what is left of the previous tile leaves here, not overlapped
        while (unit < UNITS) {
            ops += (uint32_t)drain_unit(unit);
            ++unit;
        }
        ptm = tm;
        ptn = tn;
        // this tile -> fp16 (clamp, round to nearest even: the values k_gemm_nt_bf16 stores).  Row blocks 0 and 1 leave right
        // away (16 stores: they drain under the next tile's first two k-steps), blocks 2 and 3 wait in P
        auto convert = [&](uint32_t (&R)[WN][8], auto I_) {
            constexpr int I = decltype(I_)::value;
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    h2 v;
                    // (v_med3_f32: one instruction per value where fminf(fmaxf()) costs three — its operands are canonicalised
                    // for NaNs first; no NaN can occur here, so the clamp is the one k_gemm_nt_bf16 applies)
                    v[0] = (_Float16)__builtin_amdgcn_fmed3f(acc[I][j][2 * h], -clamp_hi, clamp_hi);
                    v[1] = (_Float16)__builtin_amdgcn_fmed3f(acc[I][j][2 * h + 1], -clamp_hi, clamp_hi);
                    R[j][h] = __builtin_bit_cast(uint32_t, v);
                }
        };
        {
            uint32_t Q0[WN][8], Q1[WN][8];
            convert(Q0, std::integral_constant<int, 0>{});
            ops += (uint32_t)drain_normal(Q0, 0);
            convert(Q1, std::integral_constant<int, 1>{});
            ops += (uint32_t)drain_normal(Q1, 1);
            ops += (uint32_t)drain_mirror(Q0, Q1, 0, std::integral_constant<int, 0>{});
            ops += (uint32_t)drain_mirror(Q0, Q1, 0, std::integral_constant<int, 1>{});
        }
        convert(P[0], std::integral_constant<int, 2>{});
        convert(P[1], std::integral_constant<int, 3>{});
        unit = 0;
        if (!more) break;
        t = t_next;
        tm = tm2;
        tn = tn2;
        Ag = Ag2;
        Bg = Bg2;
    }
    while (unit < UNITS) {  // the last tile of this workgroup
        (void)drain_unit(unit);
        ++unit;
    }
}

template <bool F16, bool SYM, bool SPREAD>
static void launch_gemm_ov_t(const bf16_t* A, const bf16_t* B, _Float16* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                           int64_t ldc, const uint32_t* tile_list, int64_t n_listed, bool clamp, hipStream_t st) {
    constexpr int TB = 256;
    constexpr int SMEM = 2 * TileGeom<TB>::STAGE_BYTES + 8 * 4096;
    static_assert(SMEM <= 160 * 1024, "gemm: LDS plan exceeds the CU's 160 KiB");
    const int64_t tiles = SYM ? n_listed : (M / TB) * (N / TB);
    KN_REQUIRE(tiles > 0 && tiles < (1ll << 31), KNNCF_E_INVALID, "gemm: grid too large");
    static PerDeviceState state;
    const int64_t slots = (int64_t)per_device_at_least(state, 1, [&](size_t) {
        KN_HIP(hipFuncSetAttribute((const void*)k_gemm_nt_ov<F16, SYM, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        int dev = 0, cus = 0;
        KN_HIP(hipGetDevice(&dev));
        KN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        return (size_t)std::max(1, cus);  // all of the CU's LDS: one workgroup per CU
    });
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    k_gemm_nt_ov<F16, SYM, SPREAD><<<grid, 512, SMEM, st>>>(A, B, C, (int)(M / TB), (int)(N / TB), (int)(K / BK), lda, ldb, ldc, tile_list,
                                                    (int)n_listed, clamp ? 1.0f : 65504.0f);
    KN_HIP(hipGetLastError());
}

template <bool F16, bool SYM>
static void launch_gemm_ov(const bf16_t* A, const bf16_t* B, _Float16* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                           int64_t ldc, const uint32_t* tile_list, int64_t n_listed, bool clamp, hipStream_t st) {
    static const bool no_spread = getenv("KNNCF_GEMM_NO_SPREAD") != nullptr;  // A/B switch for measurements
    if (K / BK >= 6 && !no_spread) launch_gemm_ov_t<F16, SYM, true>(A, B, C, M, N, K, lda, ldb, ldc, tile_list, n_listed, clamp, st);
    else launch_gemm_ov_t<F16, SYM, false>(A, B, C, M, N, K, lda, ldb, ldc, tile_list, n_listed, clamp, st);
}

// (A/B switch for measurements: KNNCF_GEMM_NO_OVERLAP=1 takes the epilogue-at-the-end kernel for fp16 panels too)
static bool gemm_overlap_enabled() {
    static const bool off = getenv("KNNCF_GEMM_NO_OVERLAP") != nullptr;
    return !off;
}

template <bool F16, class OT, int WM, int WN, int WAVES_M, int WAVES_N, bool SYM>
static void launch_gemm_cfg(const bf16_t* A, const bf16_t* B, OT* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                            int64_t ldb, int64_t ldc, const uint32_t* tile_list, int64_t n_listed, bool clamp, hipStream_t st) {
    constexpr int TB = WAVES_M * WM * 32;
    constexpr int SMEM = EpiGeom<OT, WM, WAVES_M, TB>::SMEM;
    static_assert(SMEM <= 160 * 1024, "gemm: LDS plan exceeds the CU's 160 KiB");
    const int64_t tiles = SYM ? n_listed : (M / TB) * (N / TB);
    KN_REQUIRE(tiles > 0 && tiles < (1ll << 31), KNNCF_E_INVALID, "gemm: grid too large");
    // resident workgroups of this kernel on the CURRENT device (dynamic-LDS attribute, CU count, occupancy): per device
    static PerDeviceState state;
    const int64_t slots = (int64_t)per_device_at_least(state, 1, [&](size_t) {
        KN_HIP(hipFuncSetAttribute((const void*)k_gemm_nt_bf16<F16, OT, WM, WN, WAVES_M, WAVES_N, SYM>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        int dev = 0, cus = 0, per_cu = 0;
        KN_HIP(hipGetDevice(&dev));
        KN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        KN_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_gemm_nt_bf16<F16, OT, WM, WN, WAVES_M, WAVES_N, SYM>,
                                                            WAVES_M * WAVES_N * 64, SMEM));
        return (size_t)std::max(1, cus * (per_cu > 0 ? per_cu : 1));
    });
    const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
    k_gemm_nt_bf16<F16, OT, WM, WN, WAVES_M, WAVES_N, SYM><<<grid, WAVES_M * WAVES_N * 64, SMEM, st>>>(
        A, B, C, (int)(M / TB), (int)(N / TB), (int)(K / BK), lda, ldb, ldc, tile_list, (int)n_listed, clamp ? 1.0f : 65504.0f);
    KN_HIP(hipGetLastError());
}

template <bool F16, class OT>
static void launch_gemm_t(const bf16_t* A, const bf16_t* B, OT* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                          int64_t ldb, int64_t ldc, bool clamp, hipStream_t st) {
    // (a 4-wave variant of the large tile, each wave a 128 x 128 sub-tile in 256 accumulator registers — a third
    // less LDS read traffic per flop — measured the same: 30.6 vs 30.7 ms at K = 448, 52.8 vs 51.9 ms at K = 1024)
    static const bool force_small = getenv("KNNCF_GEMM_TILE128") != nullptr;  // A/B switch for measurements
    // (16 waves on the 256 x 256 tile, each a 64 x 64 sub-tile, 4 waves per SIMD: 8.3 vs 6.9 ms per launch at K = 256)
    if (M % 256 == 0 && N % 256 == 0 && !force_small) launch_gemm_cfg<F16, OT, 4, 2, 2, 4, false>(A, B, C, M, N, K, lda, ldb, ldc, nullptr, 0, clamp, st);
    else launch_gemm_cfg<F16, OT, 2, 2, 2, 2, false>(A, B, C, M, N, K, lda, ldb, ldc, nullptr, 0, clamp, st);
}

// C is fp16 (c_fp16) or fp32; operands fp16 (fp16) or bf16.  M, N multiples of 128 (256 selects the large tile)
void launch_gemm_nt(const bf16_t* A, const bf16_t* B, void* C, bool c_fp16, int64_t M, int64_t N, int64_t K, int64_t lda,
                    int64_t ldb, int64_t ldc, bool fp16, bool clamp, hipStream_t st) {
    KN_REQUIRE(M % 128 == 0 && N % 128 == 0 && K % BK == 0 && K > 0, KNNCF_E_INVALID, "gemm: shape not tile-aligned");
    KN_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, KNNCF_E_INVALID, "gemm: leading dimensions must be multiples of 8");
    static const bool force_small = getenv("KNNCF_GEMM_TILE128") != nullptr;
    if (c_fp16 && M % 256 == 0 && N % 256 == 0 && !force_small && gemm_overlap_enabled()) {
        if (fp16) launch_gemm_ov<true, false>(A, B, static_cast<_Float16*>(C), M, N, K, lda, ldb, ldc, nullptr, 0, clamp, st);
        else launch_gemm_ov<false, false>(A, B, static_cast<_Float16*>(C), M, N, K, lda, ldb, ldc, nullptr, 0, clamp, st);
        return;
    }
    if (fp16 && c_fp16) launch_gemm_t<true, _Float16>(A, B, static_cast<_Float16*>(C), M, N, K, lda, ldb, ldc, clamp, st);
    else if (fp16) launch_gemm_t<true, float>(A, B, static_cast<float*>(C), M, N, K, lda, ldb, ldc, clamp, st);
    else if (c_fp16) launch_gemm_t<false, _Float16>(A, B, static_cast<_Float16*>(C), M, N, K, lda, ldb, ldc, clamp, st);
    else launch_gemm_t<false, float>(A, B, static_cast<float*>(C), M, N, K, lda, ldb, ldc, clamp, st);
}

// the 256 x 256 tiles on and above the diagonal of an n_tiles x n_tiles grid, 8 tile rows at a time and column by
// column inside such a group (the group's operand rows stay in the L2 while its columns stream): tile row | column << 16
void gemm_sym_tile_list(int32_t n_tiles, std::vector<uint32_t>& out, int32_t group) {
    out.clear();
    out.reserve((size_t)n_tiles * (n_tiles + 1) / 2);
    for (int32_t g = 0; g < n_tiles; g += group)
        for (int32_t tn = g; tn < n_tiles; ++tn)
            for (int32_t tm = g; tm < std::min(g + group, n_tiles) && tm <= tn; ++tm) out.push_back((uint32_t)tm | ((uint32_t)tn << 16));
}

// S[N x N] = B B^T for all N rows at once, N a multiple of 256, computed on and above the diagonal and mirrored
// (fp16 panel storage only: the path that holds the whole similarity matrix)
void launch_gemm_sym(const bf16_t* B, void* C, bool c_fp16, int64_t N, int64_t K, int64_t ldb, int64_t ldc, bool fp16, bool clamp,
                     const uint32_t* d_tile_list, int64_t n_listed, hipStream_t st, int tile) {
    KN_REQUIRE(N % 256 == 0 && N / 128 < 65536 && K % BK == 0 && K > 0 && ldb % 8 == 0 && ldc % 8 == 0, KNNCF_E_INVALID, "symmetric gemm: shape not tile-aligned");
    if (tile == 128) {  // two workgroups per CU: one's stores run under the other's MFMA steps
        KN_REQUIRE(fp16 && c_fp16, KNNCF_E_INVALID, "symmetric gemm: the 128-tile form is built for fp16 operands and panel");
        launch_gemm_cfg<true, _Float16, 2, 2, 2, 2, true>(B, B, static_cast<_Float16*>(C), N, N, K, ldb, ldb, ldc, d_tile_list, n_listed, clamp, st);
        return;
    }
    if (c_fp16 && gemm_overlap_enabled()) {
        if (fp16) launch_gemm_ov<true, true>(B, B, static_cast<_Float16*>(C), N, N, K, ldb, ldb, ldc, d_tile_list, n_listed, clamp, st);
        else launch_gemm_ov<false, true>(B, B, static_cast<_Float16*>(C), N, N, K, ldb, ldb, ldc, d_tile_list, n_listed, clamp, st);
        return;
    }
    if (fp16 && c_fp16) launch_gemm_cfg<true, _Float16, 4, 2, 2, 4, true>(B, B, static_cast<_Float16*>(C), N, N, K, ldb, ldb, ldc, d_tile_list, n_listed, clamp, st);
    else if (fp16) launch_gemm_cfg<true, float, 4, 2, 2, 4, true>(B, B, static_cast<float*>(C), N, N, K, ldb, ldb, ldc, d_tile_list, n_listed, clamp, st);
    else if (c_fp16) launch_gemm_cfg<false, _Float16, 4, 2, 2, 4, true>(B, B, static_cast<_Float16*>(C), N, N, K, ldb, ldb, ldc, d_tile_list, n_listed, clamp, st);
    else launch_gemm_cfg<false, float, 4, 2, 2, 4, true>(B, B, static_cast<float*>(C), N, N, K, ldb, ldb, ldc, d_tile_list, n_listed, clamp, st);
}

}  // namespace knncf

// ---- measurement hook (not part of include/knncf.h): the similarity GEMM alone on a synthetic panel ----------------------
// scripts/microbench/gemm_bench.py: milliseconds per launch of the symmetric (sym = 1) or the row-block (sym = 0, M rows)
// form at N users x K head columns, fp16 operands and panel; the operand panel is filled with a cheap pattern (values in
// [-1/16, 1/16]: the MFMA rate does not depend on the data, the chip's clock under load does a little).
__global__ void k_debug_fill(knncf::bf16_t* p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reinterpret_cast<_Float16*>(p)[i] = (_Float16)((float)((int)((i * 2654435761u) >> 20 & 255) - 128) * (1.0f / 2048.0f));
}

extern "C" int knncf_debug_gemm_bench(int device, int64_t N, int64_t K, int64_t M, int sym, int iters, double* ms_per_launch) {
    using namespace knncf;
    try {
        KN_HIP(hipSetDevice(device));
        KN_REQUIRE(N % 256 == 0 && K % 64 == 0 && M % 256 == 0 && iters > 0 && ms_per_launch, KNNCF_E_INVALID, "bad shape");
        DArr<bf16_t> B;
        DArr<_Float16> C;
        DArr<uint32_t> tiles;
        B.alloc((size_t)N * K);
        const int64_t rows = sym ? N : M;
        C.alloc((size_t)rows * N);
        k_debug_fill<<<(unsigned)ceil_div(N * K, 256), 256>>>(B.p, N * K);
        std::vector<uint32_t> list;
        int64_t n_listed = 0;
        if (sym) {
            gemm_sym_tile_list((int32_t)(N / 256), list, 8);
            n_listed = (int64_t)list.size();
            tiles.alloc(list.size());
            KN_HIP(hipMemcpy(tiles.p, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        hipEvent_t a, b;
        KN_HIP(hipEventCreate(&a));
        KN_HIP(hipEventCreate(&b));
        auto once = [&] {
            if (sym) launch_gemm_sym(B.p, C.p, true, N, K, K, N, true, true, tiles.p, n_listed, nullptr, 256);
            else launch_gemm_nt(B.p, B.p, C.p, true, M, N, K, K, K, N, true, true, nullptr);
        };
        once();
        KN_HIP(hipDeviceSynchronize());
        KN_HIP(hipEventRecord(a, nullptr));
        for (int i = 0; i < iters; ++i) once();
        KN_HIP(hipEventRecord(b, nullptr));
        KN_HIP(hipEventSynchronize(b));
        float ms = 0.f;
        KN_HIP(hipEventElapsedTime(&ms, a, b));
        *ms_per_launch = (double)ms / iters;
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
        return KNNCF_OK;
    } catch (const knncf::Error& e) {
        fprintf(stderr, "knncf_debug_gemm_bench: %s\n", e.what());
        return e.status;
    }
}
