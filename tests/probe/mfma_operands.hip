// probe: fp32 MFMA rate when every MFMA of a 16-instruction group takes DIFFERENT operand registers (the GEMM pattern:
// fa[a][j] x fb[b][j] -> acc[a][b]), with and without LDS fragment reads between groups; 64 KB LDS -> 2 workgroups per CU
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int LDSREADS>
__global__ __launch_bounds__(256, 2) void k(float* o, const float* in, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = in[i];
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4 fa[2], fb[2];
    const int lane = threadIdx.x & 63;
    for (int a = 0; a < 2; ++a) { fa[a] = *(const f32x4*)&lds[(lane + 64 * a) * 4]; fb[a] = *(const f32x4*)&lds[(lane + 64 * (a + 2)) * 4]; }
    for (int it = 0; it < iters; ++it) {
        if (LDSREADS) {
            const int base = ((it & 15) * 512 + lane) * 4;
#pragma unroll
            for (int a = 0; a < 2; ++a) { fa[a] = *(const f32x4*)&lds[(base + 256 * a) & 16380]; fb[a] = *(const f32x4*)&lds[(base + 256 * (a + 2)) & 16380]; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    o[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *o, *in; hipMalloc(&o, 4096 * 256 * 4); hipMalloc(&in, 65536); hipMemset(in, 0, 65536);
    float h[16384]; for (int i = 0; i < 16384; ++i) h[i] = (i % 97) * 0.01f - 0.4f; hipMemcpy(in, h, 65536, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant)
        for (int rep = 0; rep < 3; ++rep) {
            const int blocks = 2048, iters = 8000;
            hipEventRecord(e0);
            if (variant == 0) k<0><<<blocks, 256>>>(o, in, iters); else k<1><<<blocks, 256>>>(o, in, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)blocks * 4 * iters * 16 * 4096.0;
            printf("%s: %.3f ms  %.1f TFLOP/s\n", variant ? "distinct operands + 4 ds_read_b128 per 16 MFMA" : "distinct operands, registers only", ms, flop / ms / 1e9);
        }
    return 0;
}
