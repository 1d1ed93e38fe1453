// probe: sustained fp32 MFMA rate of this MI355X (no memory traffic), 2 blocks of 4 waves per CU like the GEMM engine
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void k(float* o, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
        }
        a = a * 0.999f + 1e-4f;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    o[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* o; hipMalloc(&o, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {512, 1024, 2048}) {
        for (int rep = 0; rep < 3; ++rep) {
            int iters = 2000;
            hipEventRecord(e0);
            k<<<blocks, 256>>>(o, iters, 0.5f, 0.25f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)blocks * 4 * iters * 64 * 4096.0;
            printf("blocks %d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
