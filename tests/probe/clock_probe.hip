// probe: shader clock (s_memtime ticks per 100 MHz s_memrealtime tick) while a kernel runs, under four kinds of load:
//   0 = one wave per CU doing scalar work, 1 = fp32 MFMA back to back (2 x 4 waves per CU), 2 = MFMA + one ds_read_b128 per 4 MFMAs,
//   3 = like 2 + a streaming global read of 16 bytes per lane and 16 MFMAs
// Answers "is the fp32-MFMA peak (157.3 TFLOP/s at 2.4 GHz) reachable under a mixed load, or does the part give back clock?"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* o, unsigned long long* stamps, const f32x4* src, long long nsrc, int iters, float a0, float b0) {
    __shared__ f32x4 pad[4096];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = threadIdx.x; i < 4096; i += 256) pad[i] = f32x4{a0, b0, a0, b0};
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    if (MODE == 0) {
        if (threadIdx.x < 64) for (int it = 0; it < iters * 64; ++it) a = a * 0.999f + 1e-4f;
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 l = {a, b, a, b};
                if (MODE >= 2) l = pad[(threadIdx.x * 5 + it * 17 + u * 64) & 4095];
                if (MODE == 3 && u == 0) { g += src[gi % nsrc]; gi += (long long)gridDim.x * 256; }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * (j & 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(l[j], b, acc[2 * (j & 1)], 0, 0, 0);
                    acc[2 * (j & 1) + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, l[j], acc[2 * (j & 1) + 1], 0, 0, 0);
                    acc[4 + 2 * (j & 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(l[j], a, acc[4 + 2 * (j & 1)], 0, 0, 0);
                    acc[5 + 2 * (j & 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, l[j], acc[5 + 2 * (j & 1)], 0, 0, 0);
                }
            }
            a = a * 0.999f + 1e-4f;
        }
    }
    float s = a + g[0] + g[1] + g[2] + g[3];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    o[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 4 + 0] = t0; stamps[blockIdx.x * 4 + 1] = r0;
        stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime(); stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}
template <int MODE>
void run(const char* name, int blocks, int iters, float* o, unsigned long long* st, const f32x4* src, long long nsrc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(o, st, src, nsrc, iters, 0.5f, 0.25f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), st, blocks * 32, hipMemcpyDeviceToHost);
        std::vector<double> r;
        for (int i = 0; i < blocks; ++i) r.push_back(100.0 * (double)(h[4 * i + 2] - h[4 * i]) / (double)(h[4 * i + 3] - h[4 * i + 1]));
        std::sort(r.begin(), r.end());
        const double flop = MODE == 0 ? 0.0 : (double)blocks * 4 * iters * 64 * 2048.0;
        printf("%-34s %8.3f ms  %6.1f TFLOP/s  clock (median over workgroups) %6.0f MHz  [min %.0f, max %.0f]\n", name, ms, flop / ms / 1e9,
               r[r.size() / 2], r.front(), r.back());
    }
}
int main() {
    float* o; hipMalloc(&o, 8192 * 256 * 4);
    unsigned long long* st; hipMalloc(&st, 8192 * 32);
    const long long nsrc = 1LL << 28;                    // 4 GiB of float4 to stream
    f32x4* src; hipMalloc(&src, nsrc * 16); hipMemset(src, 0, nsrc * 16);
    run<0>("scalar work, 1 wave per CU", 256, 4000, o, st, src, nsrc);
    run<1>("MFMA only", 2048, 3000, o, st, src, nsrc);
    run<2>("MFMA + LDS reads", 2048, 3000, o, st, src, nsrc);
    run<3>("MFMA + LDS reads + HBM stream", 2048, 3000, o, st, src, nsrc);
    return 0;
}
