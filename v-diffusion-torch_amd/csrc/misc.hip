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

}  // namespace

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
