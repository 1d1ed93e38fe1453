// probe: sustained fp32 MFMA rate with the 16x16x4 shape (same FLOPs per wave as mfma_peak.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 2) void k(float* o, int iters, float a0, float b0) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32((i & 1) ? a : b, (i & 2) ? a : b, acc[i], 0, 0, 0);
        }
        a = a * 0.999f + 1e-4f;
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    o[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* o; hipMalloc(&o, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1024, 2048, 4096}) {
        for (int rep = 0; rep < 3; ++rep) {
            int iters = 2000;
            hipEventRecord(e0);
            k<<<blocks, 256>>>(o, iters, 0.5f, 0.25f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)blocks * 4 * iters * 128 * 2048.0;
            printf("16x16x4 blocks %d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
