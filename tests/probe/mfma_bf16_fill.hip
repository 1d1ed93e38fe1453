// probe (round 6): do plain vector instructions hide beside v_mfma_f32_32x32x16_bf16 (the split-operand GEMM forms' instruction)?
// Per iteration and wave: 8 MFMAs (32 cycles of pipe each) and N plain / packed vector instructions behind each, or the same count lumped.
// What the split forms issue per 24 MFMAs: 4 x split8 = ~176 plain VALU (v_cvt_pk_bf16_f32, v_sub_f32, shifts / masks): N ~ 7 per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum Kind { NONE = 0, FMA, CVT, LUMP, PK };
template <int KIND, int N>
__global__ __launch_bounds__(512) void k(float* o, unsigned long long* cyc, int iters, float a0) {
    extern __shared__ float lds[];
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(a0 + (threadIdx.x & 7) * 0.01f); hb[i] = (__bf16)(0.5f + i * 0.03f); }
    float p[16]; for (int i = 0; i < 16; ++i) p[i] = a0 + i;
    f32x2 q[8]; for (int i = 0; i < 8; ++i) q[i] = f32x2{a0 + i, a0 - i};
    unsigned c[8]; for (int i = 0; i < 8; ++i) c[i] = 0;
    const float ad = 1e-4f, sc = 0.999f;
    const f32x2 ad2 = {1e-4f, 2e-4f}, sc2 = {0.999f, 0.998f};
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ha), "v"(hb));
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int r = (i * N + j) & 15;
                if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(sc), "v"(ad));
                if (KIND == CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(c[r & 7]) : "v"(p[r]), "v"(p[(r + 1) & 15]));
                if (KIND == PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[r & 7]) : "v"(sc2), "v"(ad2));
            }
        }
        if (KIND == LUMP) {
#pragma unroll
            for (int j = 0; j < 8 * N; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 15]) : "v"(sc), "v"(ad));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += p[i];
    for (int i = 0; i < 8; ++i) s += q[i][0] + q[i][1] + __uint_as_float(c[i]);
    o[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x & 7];
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
template <int KIND, int N>
void run(const char* name, float* o, unsigned long long* cyc) {
    const int iters = 4000, blocks = 256;
    hipFuncSetAttribute((const void*)k<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            k<KIND, N><<<blocks, threads, 100 * 1024>>>(o, cyc, iters, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long h[8]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        const int waves = threads / 64;
        printf("%-42s %d wave(s)/SIMD: %6.1f cycles of SIMD time per MFMA (32 = free) | wave 0 %7.1f / last %7.1f per iteration | %.0f TFLOP/s bf16\n", name, waves / 4,
               (double)h[waves - 1] / iters / 8 / (waves / 4), (double)h[0] / iters, (double)h[waves - 1] / iters,
               (double)blocks * waves * iters * 8 * 32768.0 / ms / 1e9);
    }
}
int main() {
    float* o; hipMalloc(&o, 256 * 512 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 64);
    run<NONE, 0>("bf16 32x32x16 only", o, cyc);
    run<FMA, 1>("+ 1 v_fma_f32 per gap", o, cyc);
    run<FMA, 2>("+ 2 v_fma_f32 per gap", o, cyc);
    run<FMA, 4>("+ 4 v_fma_f32 per gap", o, cyc);
    run<FMA, 6>("+ 6 v_fma_f32 per gap", o, cyc);
    run<FMA, 8>("+ 8 v_fma_f32 per gap", o, cyc);
    run<FMA, 12>("+ 12 v_fma_f32 per gap", o, cyc);
    run<CVT, 4>("+ 4 v_cvt_pk_bf16_f32 per gap", o, cyc);
    run<CVT, 8>("+ 8 v_cvt_pk_bf16_f32 per gap", o, cyc);
    run<PK, 2>("+ 2 v_pk_fma_f32 per gap", o, cyc);
    run<PK, 4>("+ 4 v_pk_fma_f32 per gap", o, cyc);
    run<LUMP, 4>("8 MFMA, then 32 v_fma_f32 (lumped)", o, cyc);
    run<LUMP, 8>("8 MFMA, then 64 v_fma_f32 (lumped)", o, cyc);
    return 0;
}
