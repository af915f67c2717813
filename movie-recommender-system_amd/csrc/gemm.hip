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
#include <stdlib.h>

#include <algorithm>
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

void launch_densify(const Train& tr, const int32_t* d_rows, int32_t row_begin, int32_t n_rows,
                    const int32_t* d_colmap, bf16_t* panel, int64_t ld, int64_t panel_rows, bool fp16, hipStream_t st) {
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
    if (fp16 && c_fp16) launch_gemm_cfg<true, _Float16, 4, 2, 2, 4, true>(B, B, static_cast<_Float16*>(C), N, N, K, ldb, ldb, ldc, d_tile_list, n_listed, clamp, st);
    else if (fp16) launch_gemm_cfg<true, float, 4, 2, 2, 4, true>(B, B, static_cast<float*>(C), N, N, K, ldb, ldb, ldc, d_tile_list, n_listed, clamp, st);
    else if (c_fp16) launch_gemm_cfg<false, _Float16, 4, 2, 2, 4, true>(B, B, static_cast<_Float16*>(C), N, N, K, ldb, ldb, ldc, d_tile_list, n_listed, clamp, st);
    else launch_gemm_cfg<false, float, 4, 2, 2, 4, true>(B, B, static_cast<float*>(C), N, N, K, ldb, ldb, ldc, d_tile_list, n_listed, clamp, st);
}

}  // namespace knncf
