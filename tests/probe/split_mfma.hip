// probe: fp32 products on the bf16 matrix cores through split operands.
//   x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round to nearest even: exact for every fp32 x whose
//   pieces stay normal), a.b ~= hh + (hm + mh) + (mm + hl + lh): six v_mfma_f32_32x32x16_bf16 products, fp32 accumulation.
// Part A: accuracy of a K-long dot product against fp64 for (i) the fp32 MFMA chain, (ii) six products in one accumulator,
//         (iii) six products with the small terms in their own accumulator, (iv) three products (hh, hm, mh).
// Part B: cycles per k16 step of a wave's 64x64 tile: LDS fragment reads + split (VALU) + 24 MFMAs, 4 waves per workgroup.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk(float a, float b) {       // {bf16(a), bf16(b)} round to nearest even, a in the low half
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
// 8 fp32 -> three bf16x8 operands
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& H, bf16x8& M, bf16x8& L) {
    u32x4 h, m, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = x[2 * q], x1 = x[2 * q + 1];
        const unsigned hp = pk(x0, x1);
        const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xFFFF0000u);
        const unsigned mp = pk(r0, r1);
        const float s0 = r0 - __uint_as_float(mp << 16), s1 = r1 - __uint_as_float(mp & 0xFFFF0000u);
        h[q] = hp; m[q] = mp; l[q] = pk(s0, s1);
    }
    H = __builtin_bit_cast(bf16x8, h); M = __builtin_bit_cast(bf16x8, m); L = __builtin_bit_cast(bf16x8, l);
}

// ---------------------------------------------------------------- part A
// C[32][32] = A[32][K] . B[K][32]; one wave.  mode 0: fp32 MFMA, 1: six products one accumulator, 2: six products two accumulators,
// 3: three products, 4: six products, smallest first inside a k-step
__global__ __launch_bounds__(64) void acc_kernel(const float* A, const float* B, float* C, int K, int mode) {
    const int lane = threadIdx.x, i = lane & 31, g = lane >> 5;
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(long long)i * K + k + g], B[(long long)(k + g) * 32 + i], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            float a[8], b[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = A[(long long)i * K + k + 8 * g + j]; b[j] = B[(long long)(k + 8 * g + j) * 32 + i]; }
            bf16x8 ah, am, al, bh, bm, bl;
            split8(a, ah, am, al);
            split8(b, bh, bm, bl);
            if (mode == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            } else if (mode == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2, 0, 0, 0);
            } else if (mode == 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            } else {
                f32x16 t;
                for (int r = 0; r < 16; ++r) t[r] = 0.f;
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, t, 0, 0, 0);
                for (int r = 0; r < 16; ++r) acc[r] += t[r];
            }
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
        C[row * 32 + i] = acc[r] + acc2[r];
    }
}

// ---------------------------------------------------------------- part B
// mode 0: fp32 MFMA 32x32x2 (32 per step)   1: split + 24 bf16 MFMAs   2: 24 bf16 MFMAs, operands split once (no VALU in the loop)
// 3: split only (no MFMA)   4: split + 12 MFMAs of 3 products
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int steps, unsigned long long* cyc) {
    __shared__ float lds[2 * 16 * 128 * 2];          // two operand tiles [16 k][128 rows], two buffers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, g = lane >> 5;
    for (int e = tid; e < 2 * 16 * 128 * 2; e += 256) lds[e] = (float)((e * 2654435761u) >> 8) * (1.f / 16777216.f) - 0.5f;
    __syncthreads();
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    bf16x8 AH[2], AM[2], AL[2], BH[2], BM[2], BL[2];
    float sink = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        const float* as = lds + (s & 1) * (16 * 128 * 2);
        const float* bs = as + 16 * 128;
        float fa[2][8], fb[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x2 va = *reinterpret_cast<const f32x2*>(as + (8 * g + j) * 128 + wm + 2 * li);
            const f32x2 vb = *reinterpret_cast<const f32x2*>(bs + (8 * g + j) * 128 + wn + 2 * li);
            fa[0][j] = va[0]; fa[1][j] = va[1]; fb[0][j] = vb[0]; fb[1][j] = vb[1];
        }
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][j], fb[b][j], acc[a][b], 0, 0, 0);
        } else {
            if (MODE != 2 || s == 0) {
#pragma unroll
                for (int a = 0; a < 2; ++a) { split8(fa[a], AH[a], AM[a], AL[a]); split8(fb[a], BH[a], BM[a], BL[a]); }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) sink += fa[0][j] + fa[1][j] + fb[0][j] + fb[1][j];
            }
            if (MODE == 3) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const u32x4 x = __builtin_bit_cast(u32x4, AH[a]) ^ __builtin_bit_cast(u32x4, AM[a]) ^ __builtin_bit_cast(u32x4, AL[a]) ^
                                    __builtin_bit_cast(u32x4, BH[a]) ^ __builtin_bit_cast(u32x4, BM[a]) ^ __builtin_bit_cast(u32x4, BL[a]);
                    sink += __uint_as_float((x[0] ^ x[1] ^ x[2] ^ x[3]) & 0x3FFFFFFFu);
                }
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[a], BH[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[a], BM[b], acc[a][b], 0, 0, 0);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM[a], BH[b], acc[a][b], 0, 0, 0);
                        if (MODE != 4) {
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM[a], BM[b], acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH[a], BL[b], acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL[a], BH[b], acc[a][b], 0, 0, 0);
                        }
                    }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float v = sink;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) v += acc[a][b][r];
    out[blockIdx.x * 256 + tid] = v;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

static double frand() { return (double)rand() / RAND_MAX; }
static float gauss() { return (float)(sqrt(-2.0 * log(frand() + 1e-12)) * cos(6.283185307179586 * frand())); }

template <int MODE>
static void run_rate(const char* name, float* o, unsigned long long* cyc, int wgs_per_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wgs_per_cu, steps = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        rate_kernel<MODE><<<blocks, 256>>>(o, steps, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(blocks);
        hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double mc = 0; for (auto v : c) mc += (double)v; mc /= blocks;
        // fp32-equivalent flops: every wave 64x64x16 per step
        const double flop = (double)blocks * 4 * steps * 64.0 * 64 * 16 * 2;
        printf("%-34s %d WG/CU: %.3f ms  %.1f TFLOP/s fp32-equivalent  %.0f cyc per step per wave (s_memtime, 100 MHz ticks x?)  clock-free: %.2f us/kstep\n",
               name, wgs_per_cu, ms, flop / ms / 1e9, mc / steps, ms * 1e3 / steps);
    }
}

int main() {
    // ---- part A
    for (int K : {1024, 8192}) {
        for (int dist = 0; dist < 3; ++dist) {
            std::vector<float> A(32 * (size_t)K), B((size_t)K * 32);
            srand(1234 + dist);
            for (auto& v : A) v = dist == 1 ? (float)(frand() + 0.5) : gauss() * (dist == 2 ? expf(4.f * gauss()) : 1.f);
            for (auto& v : B) v = dist == 1 ? (float)(frand() + 0.5) : gauss() * (dist == 2 ? 1e-4f * expf(4.f * gauss()) : 1.f);
            std::vector<double> ref(32 * 32, 0.0), mag(32 * 32, 0.0);
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double s = 0, m = 0;
                    for (int k = 0; k < K; ++k) { const double p = (double)A[(size_t)i * K + k] * B[(size_t)k * 32 + j]; s += p; m += fabs(p); }
                    ref[i * 32 + j] = s; mag[i * 32 + j] = m;
                }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 32 * 32 * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            const char* names[5] = {"fp32 MFMA chain", "6 products, 1 acc", "6 products, 2 acc", "3 products", "6 products small-first"};
            for (int mode = 0; mode < 5; ++mode) {
                acc_kernel<<<1, 64>>>(dA, dB, dC, K, mode);
                std::vector<float> C(32 * 32);
                hipMemcpy(C.data(), dC, 32 * 32 * 4, hipMemcpyDeviceToHost);
                double e2 = 0, r2 = 0, worst = 0, bias = 0;
                for (int e = 0; e < 1024; ++e) {
                    const double d = (double)C[e] - ref[e];
                    e2 += d * d; r2 += ref[e] * ref[e];
                    worst = fmax(worst, fabs(d) / mag[e]);           // relative to sum |a b|: the backward-error scale
                    bias += d / mag[e];
                }
                printf("K %5d dist %d  %-24s rel-L2 %.3e   max |err| / sum|ab| %.3e   mean signed err / sum|ab| %+.3e\n", K, dist, names[mode],
                       sqrt(e2 / r2), worst, bias / 1024);
            }
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    }
    // ---- part B
    float* o; hipMalloc(&o, 4096 * 256 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 4096 * 8);
    for (int w : {1, 2, 3}) {
        run_rate<0>("fp32 MFMA 32x32x2", o, cyc, w);
        run_rate<1>("split + 6 products", o, cyc, w);
        run_rate<2>("6 products, operands pre-split", o, cyc, w);
        run_rate<3>("split only", o, cyc, w);
        run_rate<4>("split + 3 products", o, cyc, w);
    }
    return 0;
}
