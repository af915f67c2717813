// store_pattern.hip — what the similarity panel's write patterns cost on MI355X (no compute: stores only).
// A persistent grid (one 512-thread workgroup per CU) walks the 256 x 256 fp16 tiles of an N x N panel (row stride
// 2 N bytes) and writes every tile with 16-byte nontemporal stores in one of the shapes the GEMM epilogues use:
//   0: linear          — the panel as one stream (upper bound)
//   1: 2 rows x 512 B  — a wave instruction covers 2 whole tile rows (k_gemm_nt_bf16's workgroup-wide image, normal)
//   2: 4 rows x 256 B  — its mirror image (128-row passes)
//   3: 8 rows x 128 B  — a wave's private 32 x 64 block, as it is (k_gemm_nt_ov)
//   4: 16 rows x 64 B  — the same block mirrored (k_gemm_nt_ov)
// build: hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip ; run: ./store_pattern [N = 162560]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE>
__global__ void __launch_bounds__(512) k_store(char* C, int64_t N, int64_t n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ld = N * 2;
    const int64_t tiles_per_side = N / 256;
    const u32x4 v = {(uint32_t)threadIdx.x, 1u, 2u, 3u};
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t tm = t / tiles_per_side, tn = t - tm * tiles_per_side;
        char* tile = C + tm * 256 * ld + tn * 512;
        if (MODE == 0) {  // linear: 128 KiB per "tile", 1 KiB per wave instruction
            char* p = C + t * 131072 + wave * 16384;
            for (int it = 0; it < 16; ++it) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + it * 1024 + lane * 16));
        } else if (MODE == 1) {  // 2 rows x 512 B per instruction; a wave owns 32 rows
            for (int it = 0; it < 16; ++it) {
                const int row = wave * 32 + it * 2 + (lane >> 5);
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(tile + row * ld + (lane & 31) * 16));
            }
        } else if (MODE == 2) {  // 4 rows x 256 B per instruction; a wave owns 32 rows x 2 column halves
            for (int it = 0; it < 16; ++it) {
                const int row = wave * 32 + (it >> 1) * 4 + (lane >> 4);
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(tile + row * ld + (it & 1) * 256 + (lane & 15) * 16));
            }
        } else if (MODE == 3) {  // wave (wr, wc) owns rows wr * 128 .. + 128, byte columns wc * 128 .. + 128; 8 rows x 128 B
            const int wr = wave >> 2, wc = wave & 3;
            for (int it = 0; it < 16; ++it) {
                const int row = wr * 128 + it * 8 + (lane >> 3);
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(tile + row * ld + wc * 128 + (lane & 7) * 16));
            }
        } else {  // mirrored block: rows wc * 64 .. + 64, byte columns wr * 256 + i * 64; 16 rows x 64 B
            const int wr = wave >> 2, wc = wave & 3;
            for (int it = 0; it < 16; ++it) {
                const int i = it >> 2;
                const int row = wc * 64 + (it & 3) * 16 + (lane >> 2);
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(tile + row * ld + wr * 256 + i * 64 + (lane & 3) * 16));
            }
        }
    }
}

template <int MODE>
static void run(char* C, int64_t N, const char* name) {
    const int64_t n_tiles = (N / 256) * (N / 256);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k_store<MODE><<<256, 512>>>(C, N, n_tiles);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_store<MODE><<<256, 512>>>(C, N, n_tiles);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)n_tiles * 131072.0;
    printf("%-22s %8.2f ms  %6.2f TB/s  (%.1f GB)\n", name, ms, bytes / ms / 1e9, bytes / 1e9);
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 162560;
    char* C = nullptr;
    if (hipMalloc((void**)&C, (size_t)N * N * 2) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    run<0>(C, N, "linear");
    run<1>(C, N, "2 rows x 512 B");
    run<2>(C, N, "4 rows x 256 B");
    run<3>(C, N, "8 rows x 128 B");
    run<4>(C, N, "16 rows x 64 B");
    hipFree(C);
    return 0;
}
