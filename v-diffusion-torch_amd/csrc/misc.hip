// misc.hip -- small elementwise / embedding / layout kernels of the hot path (all HBM- or latency-bound).
#include "common.h"
#include <math.h>

namespace {

inline int grid_for(long long total, int per_thread = 1) {
    long long g = (total + 256LL * per_thread - 1) / (256LL * per_thread);
    if (g > 256LL * 32) g = 256LL * 32;
    return (int)(g < 1 ? 1 : g);
}

__global__ void axpby_kernel(const float* x, long long ldx, float alpha, float* y, long long ldy, float beta, long long rows,
                             int C) {
    const int vecs = C >> 2;
    const long long total = rows * vecs;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long r = idx / vecs;
        const int c4 = (int)(idx % vecs) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c4) * alpha;
        float* o = y + r * ldy + c4;
        if (beta != 0.f) v += *reinterpret_cast<const f32x4*>(o) * beta;
        *reinterpret_cast<f32x4*>(o) = v;
    }
}

__global__ void silu_kernel(const float* x, float* y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = v * vd_sigmoid(v);
    }
}

__global__ void silu_bwd_kernel(const float* x, const float* dy, float* dx, long long n, int accumulate) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float z = x[i], s = vd_sigmoid(z);
        const float g = dy[i] * s * (1.f + z * (1.f - s));
        dx[i] = accumulate ? dx[i] + g : g;
    }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one wavefront per row; the row (<= 16 KB) stays in L1/L2 between the passes
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* s, long long rows, int L) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* p = s + row * L;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, p[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) { const float e = __expf(p[j] - mx); p[j] = e; sum += e; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < L; j += 64) p[j] *= inv;
}

// register-resident forms for the row lengths the UNet produces (L = 64 * NPL): one read and one write of the row
template <int NPL>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(float* s, long long rows) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* p = s + row * (64 * NPL);
    float v[NPL];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NPL; ++k) { v[k] = p[lane + 64 * k]; mx = fmaxf(mx, v[k]); }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) { v[k] = __expf(v[k] - mx); sum += v[k]; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int k = 0; k < NPL; ++k) p[lane + 64 * k] = v[k] * inv;
}

template <int NPL>
__global__ __launch_bounds__(256) void softmax_rows_bwd_reg_kernel(const float* pm, float* dp, long long rows, float alpha) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = pm + row * (64 * NPL);
    float* d = dp + row * (64 * NPL);
    float pv[NPL], dv[NPL];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) { pv[k] = p[lane + 64 * k]; dv[k] = d[lane + 64 * k]; dot += pv[k] * dv[k]; }
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < NPL; ++k) d[lane + 64 * k] = alpha * pv[k] * (dv[k] - dot);
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* pm, float* dp, long long rows, int L, float alpha) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = pm + row * L;
    float* d = dp + row * L;
    float dot = 0.f;
    for (int j = lane; j < L; j += 64) dot += p[j] * d[j];
    dot = wave_sum(dot);
    for (int j = lane; j < L; j += 64) d[j] = alpha * p[j] * (d[j] - dot);
}

// NCHW -> NHWC with channel padding (zero-filled up to ldy)
__global__ void nchw_to_nhwc_kernel(const float* x, float* y, int nimg, int C, long long HW, long long ldy) {
    const long long total = (long long)nimg * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long b = idx / HW, pix = idx % HW;
        for (int c = 0; c < (int)ldy; ++c) y[idx * ldy + c] = c < C ? x[(b * C + c) * HW + pix] : 0.f;
    }
}
__global__ void nhwc_to_nchw_kernel(const float* x, long long ldx, float* y, int nimg, int C, long long HW) {
    const long long total = (long long)nimg * C * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx % HW, bc = idx / HW;
        const long long b = bc / C;
        const int c = (int)(bc % C);
        y[idx] = x[(b * HW + pix) * ldx + c];
    }
}

// functions.py:11-29 -- fp64 arithmetic, [sin | cos], optional zero pad for odd dim
__global__ void timestep_embedding_kernel(const double* t, float* out, int n, int dim, double scale) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * dim) return;
    const int b = idx / dim, j = idx % dim;
    float v = 0.f;
    if (j < 2 * half) {
        const int i = j < half ? j : j - half;
        const double step = log(10000.0) / (double)(half - 1);
        const double arg = (scale * t[b]) * exp(-(double)i * step);
        v = (float)(j < half ? sin(arg) : cos(arg));
    }
    out[idx] = v;
}

__global__ void class_embed_kernel(const float* y, const float* w, const float* bias, float* temb, int n, int emb, int ncls) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * emb) return;
    const int b = idx / emb, e = idx % emb;
    const long long lab = (long long)y[b];                      // .long() truncation (modules.py:191-192)
    float v = bias[e];
    if (lab != 0) { long long k = lab - 1; if (k < 0) k = 0; if (k < ncls) v += w[(long long)e * ncls + k]; }
    temb[idx] += v;
}
__global__ void class_embed_bwd_kernel(const float* y, const float* dtemb, float* dw, float* dbias, int n, int emb, int ncls,
                                       int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // over emb * (ncls + 1)
    if (idx >= emb * (ncls + 1)) return;
    const int e = idx / (ncls + 1), k = idx % (ncls + 1);
    float s = 0.f;
    if (k == ncls) {
        for (int b = 0; b < n; ++b) s += dtemb[(long long)b * emb + e];
        dbias[e] = accumulate ? dbias[e] + s : s;
    } else {
        for (int b = 0; b < n; ++b) {
            const long long lab = (long long)y[b];
            if (lab != 0 && (lab - 1 < 0 ? 0 : lab - 1) == k) s += dtemb[(long long)b * emb + e];
        }
        float* o = dw + (long long)e * ncls + k;
        *o = accumulate ? *o + s : s;
    }
}
__global__ void multitag_norm_kernel(const float* y, float* out, int n, int ncls) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    int nnz = 0;
    for (int k = 0; k < ncls; ++k) nnz += y[(long long)b * ncls + k] != 0.f;
    const float d = sqrtf(fmaxf((float)nnz, 1.f));
    for (int k = 0; k < ncls; ++k) out[(long long)b * ncls + k] = y[(long long)b * ncls + k] / d;
}

// generate.py:149  (x*127.5+127.5).clamp(0,255).to(uint8).permute(0,2,3,1): NCHW fp32 -> HWC uint8, one pass
__global__ void to_uint8_hwc_kernel(const float* x, unsigned char* out, int n, int C, long long HW) {
    const long long total = (long long)n * HW * C;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const long long bp = idx / C, pix = bp % HW, b = bp / HW;
        float v = x[(b * C + c) * HW + pix] * 127.5f + 127.5f;
        v = fminf(fmaxf(v, 0.f), 255.f);
        out[idx] = (unsigned char)v;                       // truncation, as torch's float -> uint8 cast
    }
}
// datasets.py:115-120  RandomHorizontalFlip -> ToTensor (u8/255) -> Normalize(0.5, 0.5): HWC uint8 -> NCHW fp32, one pass
__global__ void from_uint8_hwc_kernel(const unsigned char* in, const unsigned char* flip, float* out, int n, int C, int H, int W) {
    const long long HW = (long long)H * W, total = (long long)n * C * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long pix = idx % HW, bc = idx / HW;
        const int c = (int)(bc % C);
        const long long b = bc / C;
        const int y = (int)(pix / W), xx = (int)(pix % W);
        const int xs = (flip && flip[b]) ? W - 1 - xx : xx;
        const float v = (float)in[((b * H + y) * W + xs) * C + c] / 255.0f;
        out[idx] = (v - 0.5f) / 0.5f;
    }
}


// ---- "thin" 3x3 convolutions (reference unet.py:217 in_conv 3->C, :232 out_conv C->3).  With 3-4 channels on one side the
// implicit-GEMM tile engine would burn a 64-wide tile dimension on 3 useful columns; instead the 9 taps are moved to the
// thin side and the wide side becomes a plain GEMM (vd_gemm):
//   Cin thin : xc[p][tap*Cin + ci] = x[p + off(tap)][ci]                      (im2col, K = 9*Cin)   -> y = xc . W^T
//   Cout thin: z[q][co*9 + tap] = sum_ci a[q][ci] w[co][tap][ci]   (GEMM, N = 9*Cout)  -> y[p][co] = b + sum_tap z[p+off(tap)][co*9+tap]
// and the transposed forms for the gradients.
__global__ void im2col3x3_kernel(const float* x, long long ldx, float* xc, int nimg, int H, int W, int C) {
    const int vecs = C >> 2;
    const long long total = (long long)nimg * H * W * 9 * vecs;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(idx % vecs);
        const int tap = (int)((idx / vecs) % 9);
        const long long p = idx / (9 * vecs);
        const int rem = (int)(p % ((long long)H * W)), y = rem / W, xx = rem % W;
        const int yy = y + tap / 3 - 1, xs = xx + tap % 3 - 1;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)yy < (unsigned)H && (unsigned)xs < (unsigned)W)
            val = *reinterpret_cast<const f32x4*>(x + (p + (long long)(tap / 3 - 1) * W + (tap % 3 - 1)) * ldx + 4 * v);
        *reinterpret_cast<f32x4*>(xc + p * (9LL * C) + tap * C + 4 * v) = val;
    }
}

__global__ void tap_gather_kernel(const float* z, long long ldz, const float* bias, float* out, long long ldo, int nimg, int H,
                                  int W, int Cout) {
    const long long total = (long long)nimg * H * W * Cout;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % Cout);
        const long long p = idx / Cout;
        const int rem = (int)(p % ((long long)H * W)), y = rem / W, x = rem % W;
        float acc = bias ? bias[co] : 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                acc += z[(p + (long long)(tap / 3 - 1) * W + (tap % 3 - 1)) * ldz + co * 9 + tap];
        }
        out[p * ldo + co] = acc;
    }
}

// dz[q][co*9 + tap] = dy[q - off(tap)][co] (zero outside the image), columns [9*Cout, ldz) zero-filled
__global__ void tap_spread_kernel(const float* dy, long long lddy, float* dz, long long ldz, int nimg, int H, int W, int Cout) {
    const long long total = (long long)nimg * H * W * ldz;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(idx % ldz);
        const long long q = idx / ldz;
        float v = 0.f;
        if (col < 9 * Cout) {
            const int co = col / 9, tap = col % 9;
            const int rem = (int)(q % ((long long)H * W)), y = rem / W, x = rem % W;
            const int yy = y - (tap / 3 - 1), xx = x - (tap % 3 - 1);
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                v = dy[(q - (long long)(tap / 3 - 1) * W - (tap % 3 - 1)) * lddy + co];
        }
        dz[idx] = v;
    }
}

// g[co][tap*Cin + ci] (one GEMM result) -> dw_oihw[(co*Cin_w + ci)*9 + tap] (+)= ; optional dbias[co] (+)= cs[co*cs_stride + cs_off]
__global__ void thin_wgrad_finish_kernel(const float* g, int Cout_w, int Cin, int Cin_w, float* dw, int accumulate,
                                         const float* cs, int cs_stride, int cs_off, float* dbias) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (dbias && idx < Cout_w) {
        const float c = cs[idx * cs_stride + cs_off];
        dbias[idx] = accumulate ? dbias[idx] + c : c;
    }
    const long long total = (long long)Cout_w * 9 * Cin;
    if (idx >= total) return;
    const int co = (int)(idx / (9 * Cin)), rem = (int)(idx % (9 * Cin));
    const int tap = rem / Cin, ci = rem % Cin;
    if (ci >= Cin_w) return;
    float* o = dw + ((long long)co * Cin_w + ci) * 9 + tap;
    *o = accumulate ? *o + g[idx] : g[idx];
}

}  // namespace

// ---- clock / matrix-rate calibration (bench.py: a figure lines from different boxes can be normalised by).  A fixed register-only loop of
// v_mfma_f32_16x16x4_f32 on every CU (8 waves per CU, two per SIMD, 16 independent accumulators per wave, operands drawn from `seed` so
// that the data is not trivial -- the part's clock depends on it); lane 0 of every workgroup stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) around the loop: in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz, rate = FLOPs / wall time.
namespace {
__global__ __launch_bounds__(512) void mfma_calibrate_kernel(float* sink, unsigned long long* stamps, int iters, unsigned seed) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    float a = (float)(h & 0xffff) * (1.0f / 65536.0f) - 0.5f, b = (float)(h >> 16) * (1.0f / 65536.0f) - 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32((i & 1) ? a : b, (i & 2) ? a : b, acc[i], 0, 0, 0);
        a = a * 0.999f + 1e-4f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123456.789f) sink[0] = s;                     // (keeps the loop; never true for these operands)
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
}  // namespace

extern "C" int vd_mfma_calibrate(float* sink, unsigned long long* stamps, int32_t blocks, int32_t iters, uint32_t seed, void* stream) {
    VD_REQUIRE(sink && stamps && blocks > 0 && iters > 0, "vd_mfma_calibrate: bad arguments");
    hipLaunchKernelGGL(mfma_calibrate_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, sink, stamps, iters, seed);
    VD_LAUNCH_CHECK("mfma_calibrate_kernel");
    return 0;
}

extern "C" int vd_images_to_uint8_hwc(const float* x_nchw, uint8_t* out, int32_t n, int32_t C, int32_t HW, void* stream) {
    hipLaunchKernelGGL(to_uint8_hwc_kernel, dim3(grid_for((long long)n * C * HW)), dim3(256), 0, (hipStream_t)stream, x_nchw, out,
                       n, C, (long long)HW);
    VD_LAUNCH_CHECK("to_uint8_hwc_kernel");
    return 0;
}
extern "C" int vd_images_from_uint8_hwc(const uint8_t* in_hwc, const uint8_t* flip, float* out_nchw, int32_t n, int32_t C,
                                        int32_t H, int32_t W, void* stream) {
    hipLaunchKernelGGL(from_uint8_hwc_kernel, dim3(grid_for((long long)n * C * H * W)), dim3(256), 0, (hipStream_t)stream, in_hwc,
                       flip, out_nchw, n, C, H, W);
    VD_LAUNCH_CHECK("from_uint8_hwc_kernel");
    return 0;
}

extern "C" int vd_axpby(const float* x, int64_t ldx, float alpha, float* y, int64_t ldy, float beta, int64_t rows, int32_t C,
                        void* stream) {
    VD_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "vd_axpby: C/ld must be multiples of 4");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, alpha, y, ldy,
                       beta, rows, C);
    VD_LAUNCH_CHECK("axpby_kernel");
    return 0;
}
extern "C" int vd_silu(const float* x, float* y, int64_t n, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    VD_LAUNCH_CHECK("silu_kernel");
    return 0;
}
extern "C" int vd_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, int32_t accumulate, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n, accumulate);
    VD_LAUNCH_CHECK("silu_bwd_kernel");
    return 0;
}
extern "C" int vd_softmax_rows(float* s, int64_t rows, int32_t L, void* stream) {
    if (rows <= 0) return 0;
    VD_REQUIRE(L > 0, "vd_softmax_rows: L must be positive");
    const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (L == 64) hipLaunchKernelGGL(softmax_rows_reg_kernel<1>, grid, blk, 0, st, s, (long long)rows);
    else if (L == 256) hipLaunchKernelGGL(softmax_rows_reg_kernel<4>, grid, blk, 0, st, s, (long long)rows);
    else if (L == 1024) hipLaunchKernelGGL(softmax_rows_reg_kernel<16>, grid, blk, 0, st, s, (long long)rows);
    else if (L == 4096) hipLaunchKernelGGL(softmax_rows_reg_kernel<64>, grid, blk, 0, st, s, (long long)rows);
    else hipLaunchKernelGGL(softmax_rows_kernel, grid, blk, 0, st, s, rows, L);
    VD_LAUNCH_CHECK("softmax_rows_kernel");
    return 0;
}
extern "C" int vd_softmax_rows_bwd(const float* p, float* dp, int64_t rows, int32_t L, float alpha, void* stream) {
    if (rows <= 0) return 0;
    const dim3 grid((unsigned)((rows + 3) / 4)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (L == 64) hipLaunchKernelGGL(softmax_rows_bwd_reg_kernel<1>, grid, blk, 0, st, p, dp, (long long)rows, alpha);
    else if (L == 256) hipLaunchKernelGGL(softmax_rows_bwd_reg_kernel<4>, grid, blk, 0, st, p, dp, (long long)rows, alpha);
    else if (L == 1024) hipLaunchKernelGGL(softmax_rows_bwd_reg_kernel<16>, grid, blk, 0, st, p, dp, (long long)rows, alpha);
    else if (L == 4096) hipLaunchKernelGGL(softmax_rows_bwd_reg_kernel<64>, grid, blk, 0, st, p, dp, (long long)rows, alpha);
    else hipLaunchKernelGGL(softmax_rows_bwd_kernel, grid, blk, 0, st, p, dp, rows, L, alpha);
    VD_LAUNCH_CHECK("softmax_rows_bwd_kernel");
    return 0;
}
extern "C" int vd_nchw_to_nhwc(const float* x, float* y, int32_t nimg, int32_t C, int32_t H, int32_t W, int64_t ldy,
                               void* stream) {
    VD_REQUIRE(ldy >= C, "vd_nchw_to_nhwc: ldy < C");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long long)nimg * H * W)), dim3(256), 0, (hipStream_t)stream, x, y,
                       nimg, C, (long long)H * W, (long long)ldy);
    VD_LAUNCH_CHECK("nchw_to_nhwc_kernel");
    return 0;
}
extern "C" int vd_nhwc_to_nchw(const float* x, int64_t ldx, float* y, int32_t nimg, int32_t C, int32_t H, int32_t W,
                               void* stream) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long long)nimg * C * H * W)), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)ldx, y, nimg, C, (long long)H * W);
    VD_LAUNCH_CHECK("nhwc_to_nchw_kernel");
    return 0;
}
extern "C" int vd_timestep_embedding(const double* t, float* out, int32_t n, int32_t dim, double scale, void* stream) {
    VD_REQUIRE(dim >= 4, "vd_timestep_embedding: dim too small");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n * dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out, n,
                       dim, scale);
    VD_LAUNCH_CHECK("timestep_embedding_kernel");
    return 0;
}
extern "C" int vd_class_embed(const float* y, const float* w, const float* bias, float* temb, int32_t n, int32_t emb,
                              int32_t ncls, void* stream) {
    hipLaunchKernelGGL(class_embed_kernel, dim3((n * emb + 255) / 256), dim3(256), 0, (hipStream_t)stream, y, w, bias, temb, n,
                       emb, ncls);
    VD_LAUNCH_CHECK("class_embed_kernel");
    return 0;
}
extern "C" int vd_class_embed_bwd(const float* y, const float* dtemb, float* dw, float* dbias, int32_t n, int32_t emb,
                                  int32_t ncls, int32_t accumulate, void* stream) {
    hipLaunchKernelGGL(class_embed_bwd_kernel, dim3((emb * (ncls + 1) + 255) / 256), dim3(256), 0, (hipStream_t)stream, y,
                       dtemb, dw, dbias, n, emb, ncls, accumulate);
    VD_LAUNCH_CHECK("class_embed_bwd_kernel");
    return 0;
}
extern "C" int vd_multitag_norm(const float* y, float* out, int32_t n, int32_t ncls, void* stream) {
    hipLaunchKernelGGL(multitag_norm_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, y, out, n, ncls);
    VD_LAUNCH_CHECK("multitag_norm_kernel");
    return 0;
}

extern "C" int vd_im2col3x3(const float* x, int64_t ldx, float* xc, int32_t nimg, int32_t H, int32_t W, int32_t C, void* stream) {
    VD_REQUIRE(x && xc && C % 4 == 0 && ldx % 4 == 0, "vd_im2col3x3: C/ld must be multiples of 4 (C=%d)", C);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for((long long)nimg * H * W * 9 * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, xc, nimg, H, W, C);
    VD_LAUNCH_CHECK("im2col3x3_kernel");
    return 0;
}

extern "C" int vd_tap_gather(const float* z, int64_t ldz, const float* bias, float* out, int64_t ldo, int32_t nimg, int32_t H,
                             int32_t W, int32_t Cout, void* stream) {
    VD_REQUIRE(z && out && ldz >= 9 * Cout && ldo >= Cout, "vd_tap_gather: bad leading dimensions");
    hipLaunchKernelGGL(tap_gather_kernel, dim3(grid_for((long long)nimg * H * W * Cout)), dim3(256), 0, (hipStream_t)stream, z, ldz,
                       bias, out, ldo, nimg, H, W, Cout);
    VD_LAUNCH_CHECK("tap_gather_kernel");
    return 0;
}

extern "C" int vd_tap_spread(const float* dy, int64_t lddy, float* dz, int64_t ldz, int32_t nimg, int32_t H, int32_t W,
                             int32_t Cout, void* stream) {
    VD_REQUIRE(dy && dz && ldz >= 9 * Cout && lddy >= Cout, "vd_tap_spread: bad leading dimensions");
    hipLaunchKernelGGL(tap_spread_kernel, dim3(grid_for((long long)nimg * H * W * ldz)), dim3(256), 0, (hipStream_t)stream, dy, lddy,
                       dz, ldz, nimg, H, W, Cout);
    VD_LAUNCH_CHECK("tap_spread_kernel");
    return 0;
}

extern "C" int vd_thin_wgrad_finish(const float* g, int32_t Cout_w, int32_t Cin, int32_t Cin_w, float* dw_oihw, int32_t accumulate,
                                    const float* colsum, int32_t cs_stride, int32_t cs_off, float* dbias, void* stream) {
    VD_REQUIRE(g && dw_oihw && Cin_w <= Cin, "vd_thin_wgrad_finish: bad arguments");
    VD_REQUIRE(!dbias || colsum, "vd_thin_wgrad_finish: dbias needs the column sums");
    long long total = (long long)Cout_w * 9 * Cin;
    if (total < Cout_w) total = Cout_w;
    hipLaunchKernelGGL(thin_wgrad_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, Cout_w,
                       Cin, Cin_w, dw_oihw, accumulate, colsum, cs_stride, cs_off, dbias);
    VD_LAUNCH_CHECK("thin_wgrad_finish_kernel");
    return 0;
}
