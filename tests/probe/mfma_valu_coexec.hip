// probe (round 6): do vector instructions hide under fp32 matrix instructions on one SIMD of gfx950?
// Each variant is one asm block per loop iteration, so the instruction order IS what is written here.  Per wave and iteration:
//   16 x v_mfma_f32_16x16x4_f32 (independent accumulators, 32 cycles each on the SIMD's matrix pipe) and NV packed fp32 FMAs,
//   either interleaved (one or two behind every MFMA) or lumped (all MFMAs, then all FMAs), with one or two waves per SIMD, and for
//   two waves per SIMD optionally with the second wave's lump order reversed (a stagger).
// Output: shader cycles per iteration and wave (s_memtime around the loop, wave 0 of workgroup 0) and the chip-wide TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define M4(i) "v_mfma_f32_16x16x4_f32 %" #i ", %16, %17, %" #i "\n\t"
#define PK(r) "v_pk_fma_f32 %" #r ", %" #r ", %26, %27\n\t"
// operands: 0..15 accumulators, 16 a, 17 b, 18..25 packed pairs, 26 scale pair, 27 add pair

enum { MFMA_ONLY = 0, INTER1 = 1, INTER2 = 2, LUMP1 = 3, VALU_ONLY = 4, LUMP2 = 5, LUMP1_STAG = 6, LUMP2_STAG = 7, BF16_ONLY = 8, BF16_INTER1 = 9,
       F32_32X32 = 10, F32_32X32_INTER1 = 11 };

template <int MODE>
__global__ __launch_bounds__(512) void k(float* o, unsigned long long* cyc, int iters, float a0, float b0) {
    extern __shared__ float dummy[];
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    f32x2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = f32x2{a + i, b - i};
    const f32x2 sc = {0.999f, 0.998f}, ad = {1e-4f, 2e-4f};
    const int wave = threadIdx.x >> 6;
    const bool second = wave >= (int)(blockDim.x >> 7);          // the second wave of each SIMD (waves 4..7 of a 512-thread workgroup)
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (short)(0x3f80 + (threadIdx.x & 7)); hb[i] = (short)(0x3f00 + i); }
    f32x4 big[4][4];   // 32x32 accumulators (16 floats each) as 4 x f32x4
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc32[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    (void)big;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define OPS "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), \
            "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15])
#define INS "v"(a), "v"(b), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(sc), "v"(ad)
#define ALLM M4(0) M4(1) M4(2) M4(3) M4(4) M4(5) M4(6) M4(7) M4(8) M4(9) M4(10) M4(11) M4(12) M4(13) M4(14) M4(15)
#define ALLP PK(18) PK(19) PK(20) PK(21) PK(22) PK(23) PK(24) PK(25) PK(18) PK(19) PK(20) PK(21) PK(22) PK(23) PK(24) PK(25)
        // (the packed FMAs write operands 18..25 declared as inputs: the values are junk either way; a "memory"-free asm volatile keeps the block)
        if (MODE == MFMA_ONLY) asm volatile(ALLM : OPS : INS);
        else if (MODE == INTER1)
            asm volatile(M4(0) PK(18) M4(1) PK(19) M4(2) PK(20) M4(3) PK(21) M4(4) PK(22) M4(5) PK(23) M4(6) PK(24) M4(7) PK(25) M4(8) PK(18) M4(9) PK(19)
                         M4(10) PK(20) M4(11) PK(21) M4(12) PK(22) M4(13) PK(23) M4(14) PK(24) M4(15) PK(25) : OPS : INS);
        else if (MODE == INTER2)
            asm volatile(M4(0) PK(18) PK(19) M4(1) PK(20) PK(21) M4(2) PK(22) PK(23) M4(3) PK(24) PK(25) M4(4) PK(18) PK(19) M4(5) PK(20) PK(21) M4(6) PK(22) PK(23)
                         M4(7) PK(24) PK(25) M4(8) PK(18) PK(19) M4(9) PK(20) PK(21) M4(10) PK(22) PK(23) M4(11) PK(24) PK(25) M4(12) PK(18) PK(19) M4(13) PK(20) PK(21)
                         M4(14) PK(22) PK(23) M4(15) PK(24) PK(25) : OPS : INS);
        else if (MODE == LUMP1) asm volatile(ALLM ALLP : OPS : INS);
        else if (MODE == LUMP2) asm volatile(ALLM ALLP ALLP : OPS : INS);
        else if (MODE == VALU_ONLY) asm volatile(ALLP : OPS : INS);
        else if (MODE == LUMP1_STAG) {
            if (second) asm volatile(ALLP ALLM : OPS : INS); else asm volatile(ALLM ALLP : OPS : INS);
        } else if (MODE == LUMP2_STAG) {
            if (second) asm volatile(ALLP ALLP ALLM : OPS : INS); else asm volatile(ALLM ALLP ALLP : OPS : INS);
        } else if (MODE == BF16_ONLY || MODE == BF16_INTER1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ha), "v"(hb));
                if (MODE == BF16_INTER1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "v"(sc), "v"(ad));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc32[i & 3]) : "v"(a), "v"(b));
                if (MODE == F32_32X32_INTER1) {
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(2 * i) & 7]) : "v"(sc), "v"(ad));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(2 * i + 1) & 7]) : "v"(sc), "v"(ad));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc32[i][r];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s + dummy[threadIdx.x & 15];
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, float* o, unsigned long long* cyc, int nvalu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<blocks, threads, 100 * 1024>>>(o, cyc, iters, 0.5f, 0.25f);     // 100 KB of LDS: one workgroup per CU
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[8]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        const int waves = threads / 64;
        const double tf = (double)blocks * waves * iters * 16 * 2048.0 / ms / 1e9;
        if (rep == 1)
            printf("%-34s %d waves/SIMD: %7.1f cycles per iteration and wave (wave 0; last wave %7.1f) | 16 MFMA = 512 cycles of pipe, %2d packed FMAs | %.3f ms, %.1f TFLOP/s\n",
                   name, waves / 4, (double)h[0] / iters, (double)h[waves - 1] / iters, nvalu, ms, MODE == VALU_ONLY ? 0.0 : tf);
    }
}

int main() {
    float* o; hipMalloc(&o, 256 * 512 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 64);
    for (int threads : {256, 512}) {
        run<MFMA_ONLY>("fp32 16x16x4 only", threads, o, cyc, 0);
        run<VALU_ONLY>("16 packed FMAs only", threads, o, cyc, 16);
        run<INTER1>("fp32 16x16x4 + 1 FMA interleaved", threads, o, cyc, 16);
        run<INTER2>("fp32 16x16x4 + 2 FMA interleaved", threads, o, cyc, 32);
        run<LUMP1>("fp32 16x16x4, then 16 FMAs", threads, o, cyc, 16);
        run<LUMP2>("fp32 16x16x4, then 32 FMAs", threads, o, cyc, 32);
        if (threads == 512) {
            run<LUMP1_STAG>("  same, second wave FMAs first", threads, o, cyc, 16);
            run<LUMP2_STAG>("  same (32), second wave FMAs first", threads, o, cyc, 32);
        }
        run<BF16_ONLY>("bf16 16x16x32 only", threads, o, cyc, 0);
        run<BF16_INTER1>("bf16 16x16x32 + 1 FMA interleaved", threads, o, cyc, 16);
        run<F32_32X32>("fp32 32x32x2 only (8 per iter)", threads, o, cyc, 0);
        run<F32_32X32_INTER1>("fp32 32x32x2 + 2 FMA interleaved", threads, o, cyc, 16);
    }
    return 0;
}
