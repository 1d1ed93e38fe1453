// probe (round 6, second pass): which vector instructions hide in the gaps of v_mfma_f32_16x16x4_f32, and how many per gap?
// mfma_valu_coexec.hip showed v_pk_fma_f32 never hides (interleaved 1:1 it costs ~14 cycles per instruction on top of the MFMA's 32).
// Here: N plain v_fma_f32 / v_add_f32 (VGPR or SGPR multiplier) behind every MFMA, the same count lumped behind the 16 MFMAs, LDS reads
// in the gaps, and packed adds / muls.  One asm statement per instruction, "volatile", in program order.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { NONE = 0, FMA_V, FMA_S, ADD_V, PKFMA, PKADD, PKMUL, FMA_LUMP, FMA_DSB128, FMA_DSB64, MOV, FMA_LIT, PKFMA_S, DSB128_ONLY, DSB64X2_ONLY, DSB128_Q, FMA_DSB64X2, LUMP_S, LUMP_PKS };

template <int KIND, int N>
__global__ __launch_bounds__(512) void k(float* o, unsigned long long* cyc, int iters, float a0, float b0, float sc) {
    extern __shared__ float lds[];
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    float p[16];
    for (int i = 0; i < 16; ++i) p[i] = a + i;
    f32x2 q[8];
    for (int i = 0; i < 8; ++i) q[i] = f32x2{a + i, b - i};
    const f32x2 sc2 = {0.999f, 0.998f}, ad2 = {1e-4f, 2e-4f};
    const float ad = 1e-4f;
    const unsigned long long sc2s = (unsigned long long)__float_as_uint(sc) | ((unsigned long long)__float_as_uint(sc) << 32);
    f32x4 u[4]; f32x2 d[4];
    for (int i = 0; i < 4; ++i) { u[i] = f32x4{0.f, 0.f, 0.f, 0.f}; d[i] = f32x2{0.f, 0.f}; }
    const unsigned la = (threadIdx.x & 63) * 16;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == FMA_DSB128 && (i & 1) == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(u[(i >> 1) & 3]) : "v"(la), "n"(0));
            if (KIND == FMA_DSB64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i & 3]) : "v"(la), "n"(0));
            if (KIND == DSB128_ONLY && (i & 1) == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(u[(i >> 1) & 3]) : "v"(la), "n"(0));
            if (KIND == DSB128_Q && (i & 3) == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(u[(i >> 2) & 3]) : "v"(la), "n"(0));
            if ((KIND == DSB64X2_ONLY || KIND == FMA_DSB64X2) && (i & 1) == 0) {
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i & 3]) : "v"(la), "n"(0));
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[(i + 1) & 3]) : "v"(la), "n"(8));
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int r = (i * N + j) & 15;
                if (KIND == FMA_V || KIND == FMA_DSB128 || KIND == FMA_DSB64 || KIND == FMA_DSB64X2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(b), "v"(ad));
                if (KIND == FMA_S) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "s"(sc), "v"(ad));
                if (KIND == FMA_LIT) asm volatile("v_fmamk_f32 %0, %0, 0x3f7fbe77, %1" : "+v"(p[r]) : "v"(ad));
                if (KIND == PKFMA_S) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(q[r & 7]) : "s"(sc2s), "v"(ad2));
                if (KIND == ADD_V) asm volatile("v_add_f32 %0, %0, %1" : "+v"(p[r]) : "v"(ad));
                if (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(p[r]) : "v"(ad));
                if (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[r & 7]) : "v"(sc2), "v"(ad2));
                if (KIND == PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[r & 7]) : "v"(ad2));
                if (KIND == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[r & 7]) : "v"(sc2));
            }
        }
        if (KIND == FMA_LUMP) {
#pragma unroll
            for (int j = 0; j < 16 * N; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 15]) : "v"(b), "v"(ad));
        }
        if (KIND == LUMP_S) {
#pragma unroll
            for (int j = 0; j < 16 * N; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 15]) : "s"(sc), "v"(ad));
        }
        if (KIND == LUMP_PKS) {
#pragma unroll
            for (int j = 0; j < 16 * N; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(q[j & 7]) : "s"(sc2s), "v"(ad2));
        }
        if (KIND == FMA_DSB128 || KIND == FMA_DSB64 || KIND == DSB128_ONLY || KIND == DSB64X2_ONLY || KIND == DSB128_Q || KIND == FMA_DSB64X2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += p[i];
    for (int i = 0; i < 8; ++i) s += q[i][0] + q[i][1];
    for (int i = 0; i < 4; ++i) s += u[i][0] + u[i][3] + d[i][0] + d[i][1];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int KIND, int N>
void run(const char* name, float* o, unsigned long long* cyc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256;
    hipFuncSetAttribute((const void*)k<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int threads : {256, 512}) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            k<KIND, N><<<blocks, threads, 100 * 1024>>>(o, cyc, iters, 0.5f, 0.25f, 0.999f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        unsigned long long h[8]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        const int waves = threads / 64;
        const double per_mfma_simd = (double)h[waves - 1] / iters / 16 / (waves / 4);   // cycles of SIMD time per MFMA (last wave = both waves done)
        printf("%-44s %d wave(s)/SIMD: %6.1f cycles per MFMA slot (32 = free), wave 0 %7.1f / last wave %7.1f per iteration | %.1f TFLOP/s\n", name, waves / 4,
               per_mfma_simd, (double)h[0] / iters, (double)h[waves - 1] / iters, (double)blocks * waves * iters * 16 * 2048.0 / ms / 1e9);
    }
}

int main() {
    float* o; hipMalloc(&o, 256 * 512 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 64);
    run<NONE, 0>("MFMA only", o, cyc);
    run<FMA_LIT, 2>("+ 2 v_fmamk_f32 (literal multiplier) per gap", o, cyc);
    run<FMA_LIT, 4>("+ 4 v_fmamk_f32 (literal multiplier) per gap", o, cyc);
    run<PKFMA_S, 1>("+ 1 v_pk_fma_f32 (SGPR pair multiplier) per gap", o, cyc);
    run<LUMP_S, 2>("16 MFMA, then 32 v_fma_f32 (SGPR mult., lumped)", o, cyc);
    run<LUMP_PKS, 1>("16 MFMA, then 16 v_pk_fma_f32 (SGPR, lumped)", o, cyc);
    run<LUMP_PKS, 2>("16 MFMA, then 32 v_pk_fma_f32 (SGPR, lumped)", o, cyc);
    run<DSB128_ONLY, 0>("+ 1/2 ds_read_b128 per gap, no VALU", o, cyc);
    run<DSB128_Q, 0>("+ 1/4 ds_read_b128 per gap, no VALU", o, cyc);
    run<DSB64X2_ONLY, 0>("+ 2 ds_read_b64 per 2 gaps, no VALU", o, cyc);
    run<FMA_DSB64X2, 2>("+ 2 v_fma_f32 + 2 ds_read_b64 per 2 gaps", o, cyc);
    run<FMA_V, 1>("+ 1 v_fma_f32 per gap", o, cyc);
    run<FMA_V, 2>("+ 2 v_fma_f32 per gap", o, cyc);
    run<FMA_V, 3>("+ 3 v_fma_f32 per gap", o, cyc);
    run<FMA_V, 4>("+ 4 v_fma_f32 per gap", o, cyc);
    run<FMA_V, 5>("+ 5 v_fma_f32 per gap", o, cyc);
    run<FMA_V, 6>("+ 6 v_fma_f32 per gap", o, cyc);
    run<FMA_V, 8>("+ 8 v_fma_f32 per gap", o, cyc);
    run<FMA_S, 2>("+ 2 v_fma_f32 (SGPR multiplier) per gap", o, cyc);
    run<FMA_S, 4>("+ 4 v_fma_f32 (SGPR multiplier) per gap", o, cyc);
    run<ADD_V, 2>("+ 2 v_add_f32 per gap", o, cyc);
    run<ADD_V, 4>("+ 4 v_add_f32 per gap", o, cyc);
    run<MOV, 4>("+ 4 v_mov_b32 per gap", o, cyc);
    run<FMA_LUMP, 2>("16 MFMA, then 32 v_fma_f32 (lumped)", o, cyc);
    run<FMA_LUMP, 4>("16 MFMA, then 64 v_fma_f32 (lumped)", o, cyc);
    run<PKFMA, 1>("+ 1 v_pk_fma_f32 per gap", o, cyc);
    run<PKADD, 1>("+ 1 v_pk_add_f32 per gap", o, cyc);
    run<PKMUL, 1>("+ 1 v_pk_mul_f32 per gap", o, cyc);
    run<FMA_DSB128, 2>("+ 2 v_fma_f32 + 1/2 ds_read_b128 per gap", o, cyc);
    run<FMA_DSB64, 2>("+ 2 v_fma_f32 + 1 ds_read_b64 per gap", o, cyc);
    run<FMA_DSB128, 4>("+ 4 v_fma_f32 + 1/2 ds_read_b128 per gap", o, cyc);
    return 0;
}
