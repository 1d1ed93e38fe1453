// probe: MFMA issue rate of the weight-gradient operand pattern ([k][rows] LDS images).
//   variant 0: 32x32x2 MFMAs, 8 ds_read_b64 per 16 MFMAs      (what gemm_dma_kernel<.., COL, .., ..> does today)
//   variant 1: 16x16x4 MFMAs, 2 ds_read_b128 per 16 MFMAs     (4x4 blocks of 16x16 per wave, lane reads 4 row-blocks at once)
// 64 KB LDS -> 2 workgroups per CU, as the KT = 32 weight-gradient launches
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int V>
__global__ __launch_bounds__(256, 2) void k(float* o, const float* in, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    if (V == 0) {
        f32x16 acc[2][2];
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        const int li = lane & 31, lh = lane >> 5;
        for (int it = 0; it < iters; ++it) {
            const float* as = lds + (it & 3) * 2048;            // [k = 16][128 rows]
            const float* bs = lds + 8192 + (it & 3) * 2048;
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 va = *(const f32x2*)(as + (4 * lh + j) * 128 + 2 * li);
                const f32x2 vb = *(const f32x2*)(bs + (4 * lh + j) * 128 + 2 * li);
                fa[0][j] = va[0]; fa[1][j] = va[1]; fb[0][j] = vb[0]; fb[1][j] = vb[1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
        }
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    } else {
        f32x4 acc[4][4];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
        const int li = lane & 15, lk = lane >> 4;
        for (int it = 0; it < iters; ++it) {                    // one iteration = 8 k = two k4 steps = 32 MFMAs of 16x16x4
            const float* as = lds + (it & 3) * 2048;
            const float* bs = lds + 8192 + (it & 3) * 2048;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 va = *(const f32x4*)(as + (4 * h + lk) * 128 + 4 * li);
                const f32x4 vb = *(const f32x4*)(bs + (4 * h + lk) * 128 + 4 * li);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[a], vb[b], acc[a][b], 0, 0, 0);
            }
        }
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
    }
    o[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *o, *in; hipMalloc(&o, 4096 * 256 * 4); hipMalloc(&in, 65536);
    static float h[16384]; for (int i = 0; i < 16384; ++i) h[i] = (i % 97) * 0.01f - 0.4f; hipMemcpy(in, h, 65536, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant)
        for (int rep = 0; rep < 3; ++rep) {
            const int blocks = 2048, iters = 8000;
            hipEventRecord(e0);
            if (variant == 0) k<0><<<blocks, 256>>>(o, in, iters); else k<1><<<blocks, 256>>>(o, in, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)blocks * 4 * iters * (variant == 0 ? 16 * 4096.0 : 32 * 2048.0);
            printf("%s: %.3f ms  %.1f TFLOP/s\n", variant ? "16x16x4, 2 ds_read_b128 per 16 MFMA" : "32x32x2, 8 ds_read_b64 per 16 MFMA", ms, flop / ms / 1e9);
        }
    return 0;
}
