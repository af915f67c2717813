// lds_atomic_rate.hip — how fast are LDS atomics on gfx950?  (scripts/microbench: measurement only, not product)
// One 1024-thread workgroup per CU; every lane issues N updates to a 32768-word LDS array.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ void __launch_bounds__(1024) k(uint32_t* out, int iters) {
    extern __shared__ uint32_t a[];
    for (int i = threadIdx.x; i < 32768; i += 1024) a[i] = 0;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            uint32_t idx;
            if (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 5) idx = (s >> 10) & 32767;        // random
            else idx = ((it * 8 + j) * 1024 + threadIdx.x) & 32767;                                 // lane-linear
            if (MODE == 0 || MODE == 1) atomicAdd(&a[idx], 1u);
            else if (MODE == 2) a[idx] = s;
            else if (MODE == 3) acc += atomicAdd(&a[idx], 1u);
            else if (MODE == 4) atomicAdd(reinterpret_cast<float*>(a) + idx, 1.0f);
            else if (MODE == 5) a[idx] += 1;  // racy read-modify-write (rate only)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = a[5] + acc;
}

template <int MODE>
void run(const char* name) {
    uint32_t* d;
    hipMalloc(&d, 4096 * 4);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int blocks = 256, iters = 2000;
    k<MODE><<<blocks, 1024, 131072>>>(d, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, 1024, 131072>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 1024 * iters * 8;
    printf("%-36s %8.3f ms  %.3e lane-ops/s  = %.2f lane-ops/clk/CU @2.4GHz\n", name, ms, ops / (ms * 1e-3), ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    run<0>("ds_add_u32 random");
    run<1>("ds_add_u32 lane-linear");
    run<2>("ds_write_b32 random");
    run<3>("ds_add_rtn_u32 random");
    run<4>("ds_add_f32 lane-linear");
    run<5>("read+add+write random (racy)");
    return 0;
}
